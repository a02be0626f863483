"""Backward of the decoder's dense blocks for the training step (BASELINE configs[4]): nn.Linear, Qwen2RMSNorm and
Qwen2MLP (llava/model/language_model/qwen2/modeling_qwen2.py:76-90, 177-189, 771-789).  Every product runs on v3d_gemm
(out = A . W^T): with y = x . W^T,

    dx = dy . W    = gemm(dy,   W^T)        W^T [K, N] by v3d_transpose
    dW = dy^T . x  = gemm(dy^T, x^T)        dy^T [N, Mp], x^T [K, Mp]: the token rows become the k dimension, zero-padded to 64

so the backward costs two more products of the forward's size and three transposes.  Weights are in the checkpoint's layout
(gate_proj / up_proj stacked as planar [gate | up] rows), not the inference engine's tile-interleaved one: the training forward has
to keep gate and up for the backward, so SwiGLU is a pass of its own here (v3d_swiglu) instead of the GEMM epilogue.
Not here yet: attention backward, the SigLIP tower's backward, the optimizer."""
from . import ops


def _pad64(n):
    return (n + 63) // 64 * 64


def linear_backward(x, w, dy, res=None, need_dx=True, need_dw=True, need_db=False):
    """y = x . w^T (+ b).  x [M, K], w [N, K], dy [M, N] (16-bit, HBM).  Returns (dx [M, K] (+ res) or None, dw [N, K] or None,
    db [N] or None).  `res`: a gradient of dx's shape added in the product's epilogue (a residual branch)."""
    M, K = x.shape
    N = w.shape[0]
    dx = dw = db = None
    if need_dx:
        wt = ops.transpose(w)                                         # [K, N]
        dx = ops.gemm(dy, wt, res=res, epilogue=ops.EPI_RES if res is not None else ops.EPI_NONE)
    if need_dw:
        Mp = _pad64(M)
        dyt = ops.transpose(dy, out_cols=Mp)                          # [N, Mp]
        xt = ops.transpose(x, out_cols=Mp)                            # [K, Mp]
        dw = ops.gemm(dyt, xt)                                        # [N, K]
    if need_db:
        db = ops.colsum(dy)
    return dx, dw, db


def mlp_block_forward(h, ln_w, w_gate_up, w_down, eps=1e-6):
    """The second half of Qwen2DecoderLayer.forward (modeling_qwen2.py:783-789): out = h + down(act(gate(n)) * up(n)), n = norm(h).
    w_gate_up [2 I, H] = rows of gate_proj then rows of up_proj.  Returns (out, saved) - saved keeps what the backward re-reads."""
    n = ops.rmsnorm(h, ln_w, eps)
    gu = ops.gemm(n, w_gate_up)                                       # [S, 2 I] planar
    a = ops.swiglu(gu)
    out = ops.gemm(a, w_down, res=h, epilogue=ops.EPI_RES)
    return out, (h, n, gu, a)


def mlp_block_backward(dout, saved, ln_w, w_gate_up, w_down, eps=1e-6):
    """Gradients of mlp_block_forward: (dh, {"ln": dln_w, "gate_up": dW [2 I, H], "down": dW [H, I]})."""
    h, n, gu, a = saved
    da, dw_down, _ = linear_backward(a, w_down, dout)
    dgu = ops.swiglu_grad(gu, da)
    dn, dw_gu, _ = linear_backward(n, w_gate_up, dgu)
    dh, dln = ops.rmsnorm_grad(h, ln_w, dn, eps, add=dout)            # + the residual branch's gradient
    return dh, {"ln": dln, "gate_up": dw_gu, "down": dw_down}
