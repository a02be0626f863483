"""The training step (BASELINE configs[4]) as a host-side mirror of LlavaQwenForCausalLM.forward(labels=...) +
loss.backward() + the optimizer for one sample (llava_qwen.py:121-205, train_multi.sh:35-90), every operation a C-ABI call:

    sample_forward_backward   SigLIP tower -> mm_projector -> pool + 3-D PE + newline rows spliced between text rows -> Qwen2 with labels
    llm_forward_backward      the language model alone (decoder layers, final norm, LM head, shifted cross-entropy)
    decoder_layer_* / attn_block_* / mlp_block_*, siglip_layer_* / siglip_tower_*, projector_*, inputs_embeds_backward
    AdamW, ZeroAdamW          torch.optim.AdamW's update on f32 master weights; ZeRO-2 partitions over a process group

Every product runs on v3d_gemm (out = A . W^T): with y = x . W^T (nn.Linear),

    dx = dy . W    = gemm(dy,   W^T)        W^T [K, N] by v3d_transpose
    dW = dy^T . x  = gemm(dy^T, x^T)        dy^T [N, Mp], x^T [K, Mp]: the token rows become the k dimension, zero-padded to 128

so the backward costs two more products of the forward's size and three transposes.  Weights are in the checkpoint's layout
(gate_proj / up_proj stacked as planar [gate | up] rows; the tower's 72-wide heads zero-padded to 128 by siglip_pad_layer), not the
inference engine's tile-interleaved one: the training forward has to keep gate and up for the backward, so SwiGLU is a pass of its
own here (v3d_swiglu) instead of the GEMM epilogue.
Attention backward exists in two forms: tiled kernels that recompute the probabilities from the forward's row log-sum-exp and never
write an [S, S] matrix (v3d_attention_backward, csrc/attention_bwd.hip - the default), and the first, MATERIALISED form below
(one head's probability matrix in HBM with the rounding points of the reference's eager attention, modeling_qwen2.py:248-327, all five
products on v3d_gemm), kept as an independent cross-check.  Parity: tests/test_gpu_train_dense.py, tests/test_gpu_zero2.py (autograd in
f32 over the reference's formulae).  ground_sample_forward_backward is the grounding samples' step (infonce loss over object proposals).
Not here: the other grounding head types, LoRA, the HF Trainer surface."""
import math
import os

import torch

from . import ops
from ._native import V3DError


# Weight gradients beside the backward chain.  Nothing in the backward reads a dW or db before the optimizer, so inside a whole-step
# function (llm_forward_backward, sample_forward_backward, ground_sample_forward_backward) linear_backward sends them - two
# transposes, the product, the bias column sums - to a SIDE stream: the HBM-bound transposes run under the main stream's products,
# and products whose tile count leaves part of the chip idle (the tower's dW: 81-243 tiles of 128 x 128 for 512 slots, k = 23 328;
# the last, partly filled round of a persistent launch) share it with the other stream's.  The step functions join the side stream
# before they return.  Called on their own, the block functions stay on the caller's stream.  V3D_TRAIN_WGRAD_STREAM=0: one stream.
_WGRAD = {"stream": None, "depth": 0, "streams": {}}


class _wgrad_overlap:
    def __init__(self, device):
        self.device = torch.device(device)
        self.entered = False

    def __enter__(self):
        if self.device.type != "cuda" or os.environ.get("V3D_TRAIN_WGRAD_STREAM", "1") == "0":
            return self
        _WGRAD["depth"] += 1
        if _WGRAD["depth"] == 1:
            key = self.device.index if self.device.index is not None else torch.cuda.current_device()
            side = _WGRAD["streams"].get(key)
            if side is None:
                side = _WGRAD["streams"][key] = torch.cuda.Stream(device=self.device)
            _WGRAD["stream"] = side
        self.entered = True
        return self

    def __exit__(self, *exc):
        if self.entered:
            _WGRAD["depth"] -= 1
            if _WGRAD["depth"] == 0:
                torch.cuda.current_stream(self.device).wait_stream(_WGRAD["stream"])      # every dW / db is complete for the caller
                _WGRAD["stream"] = None
        return False


def _weight_grads(x, dy, need_dw, need_db):
    """dW = dy^T . x as gemm(dy^T [N, Mp], x^T [K, Mp]) and db = column sums of dy.  Shapes are padded for the product's tiling, never
    its value: the token rows (the k dimension) to 128, i.e. an EVEN number of 64-wide K-steps, which the 256 x 256 kernel's split-K
    tail needs; and where K (dW's columns) is not a multiple of 256 but the reduction is long - the tower's 1152-wide operands over
    23 328 token rows - x^T gets zero rows up to the next multiple of 256: the product then runs on the 256 x 256 kernel with its k
    range split over the idle workgroups (70-85 tiles x 3 chunks) instead of one under-filled round of 128 x 128 tiles, and the zero
    columns it produces are dropped."""
    dw = db = None
    if need_dw:
        M, K = x.shape
        Mp = (M + 127) // 128 * 128
        dyt = ops.transpose(dy, out_cols=Mp)                          # [N, Mp]
        Kp = (K + 255) // 256 * 256
        if Kp != K and M >= 8192 and (Kp - K) * 8 <= K:
            xt = torch.empty((Kp, Mp), dtype=x.dtype, device=x.device)
            xt[K:].zero_()
            ops.transpose(x, out_cols=Mp, out=xt[:K])
            dw = ops.gemm(dyt, xt)[:, :K].contiguous()                # [N, K]
        else:
            xt = ops.transpose(x, out_cols=Mp)                        # [K, Mp]
            dw = ops.gemm(dyt, xt)                                    # [N, K]
    if need_db:
        db = ops.colsum(dy)
    return dw, db


def linear_backward(x, w, dy, res=None, need_dx=True, need_dw=True, need_db=False):
    """y = x . w^T (+ b).  x [M, K], w [N, K], dy [M, N] (16-bit, HBM).  Returns (dx [M, K] (+ res) or None, dw [N, K] or None,
    db [N] or None).  `res`: a gradient of dx's shape added in the product's epilogue (a residual branch).  Inside a step function
    dw / db are produced on the side stream (see _wgrad_overlap): complete when that function returns, not before."""
    dx = dw = db = None
    side = _WGRAD["stream"] if (need_dw or need_db) and x.is_cuda else None
    if side is not None:
        main = torch.cuda.current_stream(x.device)
        side.wait_stream(main)                                        # x and dy are complete
        with torch.cuda.stream(side):
            dw, db = _weight_grads(x, dy, need_dw, need_db)
        x.record_stream(side)                                         # their blocks are not handed out again before the side stream is done
        dy.record_stream(side)
        for t in (dw, db):
            if t is not None:
                t.record_stream(main)                                 # read (optimizer, accumulation) and freed on the caller's stream
    if need_dx:
        wt = ops.transpose(w)                                         # [K, N]
        dx = ops.gemm(dy, wt, res=res, epilogue=ops.EPI_RES if res is not None else ops.EPI_NONE)
    if side is None and (need_dw or need_db):
        dw, db = _weight_grads(x, dy, need_dw, need_db)
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


# ------------------------------------------------------------------------------ attention block


def _pad(n, m):
    return (n + m - 1) // m * m


def attention_backward_materialised(qkv, do, dqkv, S, n_q, n_kv, hd, scale):
    """Causal GQA attention backward over one sequence.  qkv [>= pad128(S), (n_q + 2 n_kv) hd]: rotated q | rotated k | v, rows
    >= S zero (the keys' padding); do [S, n_q hd]; writes dq | dk | dv (gradients of the ROTATED q / k) into dqkv [S, same width].
    Per kv head: for each of its query heads  s = q k^T, p = softmax(causal, s scale), dp = do v^T, ds = p (dp - rowsum(p dp)) scale,
    dq = ds k;  then  dv = [p_h^T ...] [do_h ; ...]  and  dk = [ds_h^T ...] [q_h ; ...]  as ONE product each over the group's heads
    (one f32 accumulation, the sum autograd takes over repeat_kv's copies, modeling_qwen2.py:236-245)."""
    Sp, Sq = _pad(S, 128), _pad(S, 64)
    if qkv.shape[0] < Sp:
        raise ops.V3DError(f"attention_backward_materialised: qkv needs {Sp} rows (keys zero-padded to a multiple of 128)")
    grp = n_q // n_kv
    dev, dt = qkv.device, qkv.dtype
    buf_s = torch.empty((S, Sp), dtype=dt, device=dev)
    buf_p = torch.empty((S, Sp), dtype=dt, device=dev)
    pt = torch.empty((Sp, grp * Sq), dtype=dt, device=dev)
    dst = torch.empty((Sp, grp * Sq), dtype=dt, device=dev)
    dot = torch.empty((hd, grp * Sq), dtype=dt, device=dev)
    qt = torch.empty((hd, grp * Sq), dtype=dt, device=dev)
    for g in range(n_kv):
        k_g = qkv[:Sp, (n_q + g) * hd:(n_q + g + 1) * hd]
        v_g = qkv[:Sp, (n_q + n_kv + g) * hd:(n_q + n_kv + g + 1) * hd]
        kgt = ops.transpose(k_g)                                                  # [hd, Sp]
        for j in range(grp):
            h = g * grp + j
            q_h, do_h = qkv[:S, h * hd:(h + 1) * hd], do[:, h * hd:(h + 1) * hd]
            ops.gemm(q_h, k_g, out=buf_s)
            ops.causal_softmax_rows(buf_s, S, scale, out=buf_p)
            ops.gemm(do_h, v_g, out=buf_s)                                        # dp
            ops.softmax_grad_rows(buf_p, buf_s, scale, out=buf_s)                 # ds (a row is read whole before it is written)
            ops.gemm(buf_s, kgt, out=dqkv[:, h * hd:(h + 1) * hd])
            blk = slice(j * Sq, (j + 1) * Sq)
            ops.transpose(buf_p, out_cols=Sq, out=pt[:, blk])
            ops.transpose(buf_s, out_cols=Sq, out=dst[:, blk])
            ops.transpose(do_h, out_cols=Sq, out=dot[:, blk])
            ops.transpose(q_h, out_cols=Sq, out=qt[:, blk])
        ops.gemm(dst[:S], qt, out=dqkv[:, (n_q + g) * hd:(n_q + g + 1) * hd])
        ops.gemm(pt[:S], dot, out=dqkv[:, (n_q + n_kv + g) * hd:(n_q + n_kv + g + 1) * hd])
    return dqkv


class RopeTables:
    """The rotary table and its inverse (rotation by -theta: the same kernel on a table built from -inv_freq)."""

    def __init__(self, head_dim, n_pos, base, dtype, device):
        self.fwd = ops.RopeTable(head_dim, n_pos, base, dtype, device)
        self.inv = ops.RopeTable(head_dim, n_pos, base, dtype, device, inv_freq=-ops.reference_inv_freq(head_dim, base))


def attn_block_forward(h, ln_w, w_qkv, b_qkv, w_o, rope, n_q, n_kv, hd, eps=1e-6):
    """The first half of Qwen2DecoderLayer.forward (modeling_qwen2.py:771-781): out = h + o_proj(attention(rotary(qkv(norm(h))))).
    w_qkv [(n_q + 2 n_kv) hd, H] = q_proj | k_proj | v_proj rows."""
    S = h.shape[0]
    width = (n_q + 2 * n_kv) * hd
    n = ops.rmsnorm(h, ln_w, eps)
    qkv = torch.zeros((_pad(S, 128), width), dtype=h.dtype, device=h.device)      # rows >= S stay zero: the backward's key padding
    ops.gemm(n, w_qkv, bias=b_qkv, epilogue=ops.EPI_BIAS, out=qkv[:S])
    ops.rope_apply(qkv[:S], n_q + n_kv, hd, rope.fwd)
    o = torch.empty((S, n_q * hd), dtype=h.dtype, device=h.device)
    lse = ops.attention_train(qkv, o, S, n_q, n_kv, 1.0 / math.sqrt(hd))
    out = ops.gemm(o, w_o, res=h, epilogue=ops.EPI_RES)
    return out, (h, n, qkv, o, lse)


def attn_block_backward(dout, saved, ln_w, w_qkv, w_o, rope, n_q, n_kv, hd, eps=1e-6, materialised=False):
    """Gradients of attn_block_forward: (dh, {"ln", "qkv", "qkv_bias", "o"})."""
    h, n, qkv, o, lse = saved
    S = h.shape[0]
    do, dw_o, _ = linear_backward(o, w_o, dout)
    dqkv = torch.empty((S, qkv.shape[1]), dtype=h.dtype, device=h.device)
    if materialised:
        attention_backward_materialised(qkv, do, dqkv, S, n_q, n_kv, hd, 1.0 / math.sqrt(hd))
    else:
        ops.attention_backward(qkv, o, do, lse, dqkv, S, n_q, n_kv, 1.0 / math.sqrt(hd))
    ops.rope_apply(dqkv, n_q + n_kv, hd, rope.inv)                                # the rotation's transpose
    dn, dw_qkv, db_qkv = linear_backward(n, w_qkv, dqkv, need_db=True)
    dh, dln = ops.rmsnorm_grad(h, ln_w, dn, eps, add=dout)
    return dh, {"ln": dln, "qkv": dw_qkv, "qkv_bias": db_qkv, "o": dw_o}


def decoder_layer_forward(h, p, rope, n_q, n_kv, hd, eps=1e-6):
    """Qwen2DecoderLayer.forward (modeling_qwen2.py:727-801) for one sequence; p: dict of the layer's weights (ln1, qkv, qkv_bias, o,
    ln2, gate_up, down)."""
    mid, s1 = attn_block_forward(h, p["ln1"], p["qkv"], p["qkv_bias"], p["o"], rope, n_q, n_kv, hd, eps)
    out, s2 = mlp_block_forward(mid, p["ln2"], p["gate_up"], p["down"], eps)
    return out, (s1, s2)


def decoder_layer_backward(dout, saved, p, rope, n_q, n_kv, hd, eps=1e-6, materialised=False):
    s1, s2 = saved
    dmid, g2 = mlp_block_backward(dout, s2, p["ln2"], p["gate_up"], p["down"], eps)
    dh, g1 = attn_block_backward(dmid, s1, p["ln1"], p["qkv"], p["o"], rope, n_q, n_kv, hd, eps, materialised)
    return dh, {"ln1": g1["ln"], "qkv": g1["qkv"], "qkv_bias": g1["qkv_bias"], "o": g1["o"], "ln2": g2["ln"], "gate_up": g2["gate_up"],
                "down": g2["down"]}


# ------------------------------------------------------------------------------ the language model's step


def llm_forward_backward(params, x, labels, rope, n_q, n_kv, hd, eps=1e-6, recompute=False):
    """Qwen2ForCausalLM.forward with labels (modeling_qwen2.py:1145-1217: decoder layers, final norm, LM head, shifted cross-entropy)
    and its backward for one sequence given as inputs_embeds x [S, H] (what prepare_inputs_labels_for_multimodal hands the LLM,
    llava_qwen.py:121-170).  params: {"layers": [layer dicts of decoder_layer_forward], "norm": [H], "lm_head": [V, H]}; labels [S]
    int64 on the device (-100 = ignored).  Returns (loss (f32 scalar tensor), dx [S, H], grads in params' structure).  All layers'
    activations are kept by default (about 1 GB per 7B layer at S = 6.8 k: 28 GB of the 288); recompute=True keeps only the layers'
    inputs and runs each layer's forward again in the backward, as the reference does under gradient checkpointing
    (train_multi.sh:72) - same numbers, a third more work, 27 GB less."""
    with _wgrad_overlap(x.device):
        h, saved = llm_layers_forward(params, x, rope, n_q, n_kv, hd, eps, recompute)
        n = ops.rmsnorm(h, params["norm"], eps)
        logits = ops.gemm(n, params["lm_head"])
        loss, st = ops.cross_entropy(logits, labels)
        dlogits = ops.cross_entropy_grad(st)
        dn, dw_head, _ = linear_backward(n, params["lm_head"], dlogits)
        dh, dnorm = ops.rmsnorm_grad(h, params["norm"], dn, eps)
        dx, layer_grads = llm_layers_backward(dh, saved, params, rope, n_q, n_kv, hd, eps)
        return loss, dx, {"layers": layer_grads, "norm": dnorm, "lm_head": dw_head}


class _LayerInput:
    """What a layer keeps under activation re-computation (the reference's gradient checkpointing, train_multi.sh:72 ->
    gradient_checkpointing True): only its input; the backward runs the layer's forward again to regenerate the rest.  The kernels
    are deterministic, so the regenerated activations - and therefore every gradient - are bit-identical to the keep-all path."""

    def __init__(self, x):
        self.x = x


def llm_layers_forward(params, x, rope, n_q, n_kv, hd, eps=1e-6, recompute=False):
    """Qwen2Model's decoder layers over inputs_embeds x (modeling_qwen2.py:952-1060): returns the residual stream before the final norm
    and what the backward re-reads (recompute: only each layer's input, 49 MB instead of ~1 GB per 7B layer at S = 6.8 k)."""
    saved, h = [], x
    for p in params["layers"]:
        h_in = h
        h, s = decoder_layer_forward(h_in, p, rope, n_q, n_kv, hd, eps)
        saved.append(_LayerInput(h_in) if recompute else s)
    return h, saved


def llm_layers_backward(dh, saved, params, rope, n_q, n_kv, hd, eps=1e-6):
    layer_grads = [None] * len(saved)
    for i in range(len(saved) - 1, -1, -1):
        s = saved[i]
        if isinstance(s, _LayerInput):
            _, s = decoder_layer_forward(s.x, params["layers"][i], rope, n_q, n_kv, hd, eps)
        dh, layer_grads[i] = decoder_layer_backward(dh, s, params["layers"][i], rope, n_q, n_kv, hd, eps)
        saved[i] = s = None                                           # the layer's activations are no longer needed
    return dh, layer_grads


def accumulate_grads(total, grads):
    """Gradient accumulation over micro-batches (train_multi.sh:31-32: two per optimizer step on 8 GPUs): total += grads, leaf by leaf, in
    the gradients' own 16-bit dtype as torch accumulates .grad; the first micro-batch's tree is taken as it is.  The mean over the
    micro-batches is the optimizer's grad_scale (1 / steps)."""
    if total is None:
        return grads

    def add(t, g, _):
        if t.dtype == torch.float32:
            t.add_(g)                                   # image_newline's [H] f32 row sum (14 KB): left to torch
        else:
            ops.axpy(t, g)

    _tree_zip(add, total, grads, total)
    return total


# ---- what the reference's trainer configures around the update (train_multi.sh:45, 65-68; llava_trainer.py:446-523; zero2.json:36)
LR_KEYWORDS = {"vision_tower": "vision", "mm_projector": "projector"}     # the reference's module keywords -> this tree's top-level keys


def _paths(tree, prefix=""):
    if isinstance(tree, dict):
        out = []
        for k in tree:
            out += _paths(tree[k], f"{prefix}.{k}" if prefix else str(k))
        return out
    if isinstance(tree, list):
        out = []
        for i, v in enumerate(tree):
            out += _paths(v, f"{prefix}.{i}")
        return out
    return [prefix]


def _no_decay(path):
    """llava_trainer.py:459-460: no weight decay for the parameters of nn.LayerNorm modules (get_parameter_names(model,
    ALL_LAYERNORM_LAYERS); the vendored Qwen2RMSNorm is NOT in that list, so the decoder's norm weights ARE decayed) and for every
    parameter whose name contains "bias".  This tree's names: the tower's LayerNorms vision.layers.N.ln{1,2}_{w,b}, the grounding heads'
    LayerNorm `*.ln_w / ln_b`; biases *_b, b0 .. b3, *_bias."""
    leaf = path.rsplit(".", 1)[-1]
    layernorm = leaf in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "ln_w", "ln_b")
    return layernorm or (leaf[:1] == "b" and leaf[1:].isdigit()) or leaf.endswith(("_b", "_bias")) or "bias" in leaf


def param_groups(params, lr, weight_decay=0.0, lr_by_module=None):
    """Per-leaf (lr, weight_decay) in the order of the tree's leaves: the reference's optimizer groups (llava_trainer.py:446-523) -
    `--mm_vision_tower_lr 2e-6` / `--mm_projector_lr` give the tower / projector their own learning rate (train_multi.sh:45), every other
    parameter takes `--learning_rate` (:65); biases and norm weights are not decayed.  lr_by_module: {"vision_tower" | "mm_projector" (the
    reference's keywords) or a top-level key of this tree: lr}."""
    by = {LR_KEYWORDS.get(k, k): float(v) for k, v in (lr_by_module or {}).items() if v is not None}
    out = []
    for path in _paths(params):
        top = path.split(".", 1)[0]
        out.append((by.get(top, float(lr)), 0.0 if _no_decay(path) else float(weight_decay)))
    return out


def cosine_warmup_schedule(total_steps, warmup_ratio=0.03):
    """`--lr_scheduler_type cosine --warmup_ratio 0.03` (train_multi.sh:67-68) as the HF Trainer builds it (get_cosine_schedule_with_warmup,
    warm-up steps = ceil(ratio * total)): multiplier of every group's base lr at optimizer step t = 1, 2, ...  The scheduler is stepped
    AFTER the optimizer, so step t runs with lambda(t - 1): the first step of a run with warm-up has lr 0."""
    import math
    warm = int(math.ceil(total_steps * warmup_ratio))

    def lam(t):
        k = t - 1
        if k < warm:
            return k / max(1, warm)
        prog = (k - warm) / max(1, total_steps - warm)
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
    return lam


def global_grad_norm(grad_leaves):
    """Sum of squares over all gradient tensors as one f32 device scalar [1], deterministic (v3d_sumsq chained over the leaves in tree
    order); its square root is torch.nn.utils.clip_grad_norm_'s total_norm (norm_type 2)."""
    acc = None
    for g in grad_leaves:
        if g is None or g.numel() == 0:
            continue
        acc = ops.sumsq(g.contiguous().view(-1), out=acc, accumulate=acc is not None)
    if acc is None:
        return torch.zeros(1, dtype=torch.float32, device="cuda")
    return acc


def clip_coefficient(total_norm, max_norm):
    """clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1."""
    return min(1.0, float(max_norm) / (float(total_norm) + 1e-6))


class AdamW:
    """f32 master weights and moments for a dict / list tree of 16-bit parameter tensors; step(grads) updates the tree in place
    (one v3d_adamw_step per tensor).  Hyper-parameters as train_multi.sh gives them to the HF Trainer (lr 1e-5, weight decay 0).
    r04 (the trainer-side fidelity of f4): lr_by_module (per-module learning rates, llava_trainer.py:446-523), max_grad_norm (global
    gradient-norm clipping before the update: HF max_grad_norm = 1.0 / zero2.json:36; None = off), schedule (a multiplier of the
    learning rates per step, e.g. cosine_warmup_schedule).  `last_grad_norm` holds the norm of the last step's (scaled) gradient."""

    def __init__(self, params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, lr_by_module=None, max_grad_norm=None, schedule=None):
        self.lr, self.betas, self.eps, self.weight_decay, self.t = lr, betas, eps, weight_decay, 0
        self.groups = param_groups(params, lr, weight_decay, lr_by_module)
        self.max_grad_norm, self.schedule, self.last_grad_norm = max_grad_norm, schedule, None
        self.state = _tree_map(lambda p: (p.detach().float().contiguous(), torch.zeros_like(p, dtype=torch.float32), torch.zeros_like(p, dtype=torch.float32)), params)

    def step(self, params, grads, grad_scale=1.0):
        self.t += 1
        if self.max_grad_norm is not None:             # the norm of the gradient the update sees (after grad_scale: the accumulation mean)
            self.last_grad_norm = float(global_grad_norm(_leaves(grads)).sqrt()) * abs(grad_scale)
            grad_scale = grad_scale * clip_coefficient(self.last_grad_norm, self.max_grad_norm)
        mult = self.schedule(self.t) if self.schedule is not None else 1.0
        groups = iter(self.groups)

        def one(p, g, st):
            lr, wd = next(groups)
            ops.adamw_step(st[0], st[1], st[2], g.contiguous(), p16=p, lr=lr * mult, betas=self.betas, eps=self.eps, weight_decay=wd,
                           step=self.t, grad_scale=grad_scale)
        _tree_zip(one, params, grads, self.state)


def _leaves_of_state(state):
    """The (p32, m, v) tuples of an AdamW state tree, in leaf order."""
    if isinstance(state, dict):
        return [x for k in state for x in _leaves_of_state(state[k])]
    if isinstance(state, list):
        return [x for v in state for x in _leaves_of_state(v)]
    return [state]


def _tree_map(fn, tree):
    if isinstance(tree, dict):
        return {k: _tree_map(fn, v) for k, v in tree.items()}
    if isinstance(tree, list):
        return [_tree_map(fn, v) for v in tree]
    return fn(tree)


def _tree_zip(fn, a, b, c):
    if isinstance(a, dict):
        for k in a:
            _tree_zip(fn, a[k], b[k], c[k])
    elif isinstance(a, list):
        for x, y, z in zip(a, b, c):
            _tree_zip(fn, x, y, z)
    else:
        fn(a, b, c)


# ------------------------------------------------------------------------------ the path from the LLM's input rows back to the projector


def projector_forward(x, w1, b1, w2, b2):
    """mlp2x_gelu (multimodal_projector/builder.py:41-48): Linear, nn.GELU(), Linear on the SigLIP patch features x [rows, 1152]."""
    z = ops.gemm(x, w1, bias=b1, epilogue=ops.EPI_BIAS)
    a = ops.gelu(z)
    return ops.gemm(a, w2, bias=b2, epilogue=ops.EPI_BIAS), (x, z, a)


def projector_backward(dy, saved, w1, w2, need_dx=True):
    x, z, a = saved
    da, dw2, db2 = linear_backward(a, w2, dy, need_db=True)
    dz = ops.gelu_grad(z, da)
    dx, dw1, db1 = linear_backward(x, w1, dz, need_dx=need_dx, need_db=True)
    return dx, {"w1": dw1, "b1": db1, "w2": dw2, "b2": db2}


def inputs_embeds_backward(dx, n_pre, frames, text_rows, text_ids, embed_grad_out, side=27, n=14):
    """Backward of the splice (llava_arch.py:650-836, one sample, one <image>): dx [S, H] = gradient of inputs_embeds
    [n_pre text rows | frames * n * (n + 1) visual rows | text rows].  The visual rows go back through newline / PE add / bilinear pool
    (v3d_visual_tokens_grad) to the projector's output [frames, side * side, H] and to image_newline; the text rows are summed per
    token id into embed_tokens' gradient (embed_grad_out [vocab, H], pre-zeroed).  Returns (dfeat, dnewline)."""
    n_vis = frames * n * (n + 1)
    dfeat, dnl = ops.visual_tokens_grad(dx[n_pre:n_pre + n_vis], frames, side=side, n=n, newline=True)
    ops.embed_grad(dx, text_rows, text_ids, embed_grad_out)
    return dfeat, dnl


# ------------------------------------------------------------------------------ the SigLIP tower (mm_vision_tower is tuned too, train_multi.sh:44)

SIGLIP_HD, SIGLIP_PAD = 72, 128


def siglip_pad_layer(sd, heads=16, inter_pad=4352):
    """A SigLIP encoder layer's checkpoint tensors (siglip_encoder.py:197-305: q/k/v/out_proj, fc1/fc2, two LayerNorms) in the training
    layout: heads zero-padded 72 -> 128 so that attention runs on the head-dim-128 kernels (padded q / k columns add 0 to the scores,
    padded v columns produce zero output columns, and every padded weight row / column receives a zero gradient), fc1 / fc2 padded
    4304 -> 4352 (a multiple of 128 for the GEMM; gelu(0) = 0)."""
    H = sd["q_w"].shape[1]
    dt, dev = sd["q_w"].dtype, sd["q_w"].device

    def pad_rows(w, b):
        wp = torch.zeros((heads * SIGLIP_PAD, H), dtype=dt, device=dev)
        bp = torch.zeros(heads * SIGLIP_PAD, dtype=dt, device=dev)
        wp.view(heads, SIGLIP_PAD, H)[:, :SIGLIP_HD] = w.view(heads, SIGLIP_HD, H)
        bp.view(heads, SIGLIP_PAD)[:, :SIGLIP_HD] = b.view(heads, SIGLIP_HD)
        return wp, bp

    ws, bs = zip(*(pad_rows(sd[n + "_w"], sd[n + "_b"]) for n in ("q", "k", "v")))
    wo = torch.zeros((H, heads * SIGLIP_PAD), dtype=dt, device=dev)
    wo.view(H, heads, SIGLIP_PAD)[:, :, :SIGLIP_HD] = sd["o_w"].view(H, heads, SIGLIP_HD)
    inter = sd["fc1_w"].shape[0]
    fc1 = torch.zeros((inter_pad, H), dtype=dt, device=dev)
    fc1[:inter] = sd["fc1_w"]
    b1 = torch.zeros(inter_pad, dtype=dt, device=dev)
    b1[:inter] = sd["fc1_b"]
    fc2 = torch.zeros((H, inter_pad), dtype=dt, device=dev)
    fc2[:, :inter] = sd["fc2_w"]
    return {"ln1_w": sd["ln1_w"], "ln1_b": sd["ln1_b"], "qkv": torch.cat(ws).contiguous(), "qkv_b": torch.cat(bs).contiguous(), "o": wo, "o_b": sd["o_b"],
            "ln2_w": sd["ln2_w"], "ln2_b": sd["ln2_b"], "fc1": fc1, "fc1_b": b1, "fc2": fc2, "fc2_b": sd["fc2_b"]}


def siglip_unpad_grads(g, heads=16, inter=4304):
    """Gradients of siglip_pad_layer's tensors back in the checkpoint's shapes (the padding's gradients are zero and dropped)."""
    H = g["qkv"].shape[1]
    q, k, v = (t.view(heads, SIGLIP_PAD, H)[:, :SIGLIP_HD].reshape(heads * SIGLIP_HD, H) for t in g["qkv"].view(3, heads * SIGLIP_PAD, H))
    qb, kb, vb = (t.view(heads, SIGLIP_PAD)[:, :SIGLIP_HD].reshape(-1) for t in g["qkv_b"].view(3, heads * SIGLIP_PAD))
    return {"q_w": q, "k_w": k, "v_w": v, "q_b": qb, "k_b": kb, "v_b": vb, "o_w": g["o"].view(H, heads, SIGLIP_PAD)[:, :, :SIGLIP_HD].reshape(H, -1),
            "o_b": g["o_b"], "fc1_w": g["fc1"][:inter], "fc1_b": g["fc1_b"][:inter], "fc2_w": g["fc2"][:, :inter], "fc2_b": g["fc2_b"],
            "ln1_w": g["ln1_w"], "ln1_b": g["ln1_b"], "ln2_w": g["ln2_w"], "ln2_b": g["ln2_b"]}


def siglip_layer_forward(x, p, frames, tokens=729, heads=16, eps=1e-6):
    """SigLipEncoderLayer.forward (siglip_encoder.py:264-305) on x [frames * tokens, 1152]; p from siglip_pad_layer."""
    n1 = ops.layernorm(x, p["ln1_w"], p["ln1_b"], eps)
    qkv = ops.gemm(n1, p["qkv"], bias=p["qkv_b"], epilogue=ops.EPI_BIAS)
    o = torch.empty((x.shape[0], heads * SIGLIP_PAD), dtype=x.dtype, device=x.device)
    lse = ops.attention_train(qkv, o, tokens, heads, heads, SIGLIP_HD ** -0.5, B=frames, causal=False)
    mid = ops.gemm(o, p["o"], bias=p["o_b"], res=x, epilogue=ops.EPI_BIAS_RES)
    n2 = ops.layernorm(mid, p["ln2_w"], p["ln2_b"], eps)
    z = ops.gemm(n2, p["fc1"], bias=p["fc1_b"], epilogue=ops.EPI_BIAS)
    a = ops.gelu(z, tanh_form=True)
    out = ops.gemm(a, p["fc2"], bias=p["fc2_b"], res=mid, epilogue=ops.EPI_BIAS_RES)
    return out, (x, n1, qkv, o, lse, mid, n2, z, a)


def siglip_layer_backward(dout, saved, p, frames, tokens=729, heads=16, eps=1e-6):
    x, n1, qkv, o, lse, mid, n2, z, a = saved
    da, d_fc2, d_b2 = linear_backward(a, p["fc2"], dout, need_db=True)
    dz = ops.gelu_grad(z, da, tanh_form=True)
    dn2, d_fc1, d_b1 = linear_backward(n2, p["fc1"], dz, need_db=True)
    dmid, d_ln2w, d_ln2b = ops.layernorm_grad(mid, p["ln2_w"], dn2, eps, add=dout)
    do, d_o, d_ob = linear_backward(o, p["o"], dmid, need_db=True)
    dqkv = torch.empty_like(qkv)
    ops.attention_backward(qkv, o, do, lse, dqkv, tokens, heads, heads, SIGLIP_HD ** -0.5, B=frames, causal=False)
    dn1, d_qkv, d_qkvb = linear_backward(n1, p["qkv"], dqkv, need_db=True)
    dx, d_ln1w, d_ln1b = ops.layernorm_grad(x, p["ln1_w"], dn1, eps, add=dmid)
    return dx, {"ln1_w": d_ln1w, "ln1_b": d_ln1b, "qkv": d_qkv, "qkv_b": d_qkvb, "o": d_o, "o_b": d_ob, "ln2_w": d_ln2w, "ln2_b": d_ln2b,
                "fc1": d_fc1, "fc1_b": d_b1, "fc2": d_fc2, "fc2_b": d_b2}


def siglip_tower_forward(patches, vp, frames, tokens=729, recompute=False):
    """SigLipVisionEmbeddings (the patch convolution as a GEMM over v3d_patchify's rows + the position embedding, siglip_encoder.py:148-190)
    and the encoder layers the projector reads (hidden_states[-2]: all but the last layer, siglip_encoder.py:576-589).
    vp: {"patch_w" [1152, kpad], "patch_b", "pos" [tokens, 1152], "layers": [siglip_pad_layer dicts]}."""
    h = ops.gemm(patches, vp["patch_w"], bias=vp["patch_b"], res=vp["pos"], res_mod=tokens, epilogue=ops.EPI_BIAS_RES)
    saved = []
    for p in vp["layers"]:
        h_in = h
        h, s = siglip_layer_forward(h_in, p, frames, tokens)
        saved.append(_LayerInput(h_in) if recompute else s)
    return h, (patches, saved)


def siglip_tower_backward(dh, saved, vp, frames, tokens=729):
    patches, layer_saved = saved
    grads = [None] * len(layer_saved)
    for i in range(len(layer_saved) - 1, -1, -1):
        s = layer_saved[i]
        if isinstance(s, _LayerInput):
            _, s = siglip_layer_forward(s.x, vp["layers"][i], frames, tokens)
        dh, grads[i] = siglip_layer_backward(dh, s, vp["layers"][i], frames, tokens)
        layer_saved[i] = s = None
    d_pos = ops.colsum(dh.view(frames, -1)).view(tokens, -1)           # the position embedding is added to every frame
    _, d_patch_w, d_patch_b = linear_backward(patches, vp["patch_w"], dh, need_dx=False, need_db=True)
    return {"patch_w": d_patch_w, "patch_b": d_patch_b, "pos": d_pos, "layers": grads}


def _build_inputs_embeds(params, y, voxel_ids, pe_table, pre_ids, post_ids, frames, tokens, side, n, coord_rows=None, coord_pe=None):
    """prepare_inputs_labels_for_multimodal for one sample with one <image> (llava_arch.py:650-836): text rows | visual rows | text rows;
    coord_rows / coord_pe: the PE of the discretised box centre added to the rows of the <coord> text tokens (Scan2Cap prompts,
    llava_arch.py:416-417, 697-700) - no parameters, so the backward passes those rows' gradient through to the embedding."""
    H = y.shape[1]
    n_pre, n_post, n_vis = pre_ids.numel(), post_ids.numel(), frames * n * (n + 1)
    x = torch.empty((n_pre + n_vis + n_post, H), dtype=y.dtype, device=y.device)
    if n_pre:
        ops.embed_gather(params["embed"], pre_ids, out=x[:n_pre])
    ops.visual_tokens(y.view(frames, tokens, H), voxel_ids, pe_table, params["newline"], side=side, n=n, pool=True, out=x[n_pre:n_pre + n_vis])
    if n_post:
        ops.embed_gather(params["embed"], post_ids, out=x[n_pre + n_vis:])
    if coord_rows is not None and coord_rows.numel():
        ops.add_row(x, coord_rows, coord_pe)
    return x, n_pre, n_vis


def sample_forward_backward(params, patches, voxel_ids, pe_table, pre_ids, post_ids, labels, rope, frames, n_q, n_kv, hd, tokens=729, side=27, n=14,
                            coord_rows=None, coord_pe=None, recompute=False):
    """One training sample end to end on the device (llava_qwen.py:121-205 -> llava_arch.py:336-836 -> modeling_qwen2.py:1145-1217):
    SigLIP tower -> mm_projector -> bilinear pool + 3-D PE + newline rows, spliced between the embedded text rows -> Qwen2 with labels;
    then the backward of all of it.  params: {"vision", "projector": {w1, b1, w2, b2}, "newline" [H], "embed" [vocab, H], "llm"};
    patches [frames * tokens, kpad] (v3d_patchify of the preprocessed frames), voxel_ids [frames, n, n, 3] int32 (the discretised patch
    coordinates: no gradient, llava_arch.py:515), pre_ids / post_ids: the text token ids around <image> (device int64), labels [S];
    coord_rows (device int64) / coord_pe [H]: Scan2Cap's <coord> rows and the box-centre PE added to them.
    recompute: activation re-computation per layer in both towers (the reference's gradient checkpointing, train_multi.sh:72).
    Returns (loss, grads in params' structure; "embed" is a dense [vocab, H] gradient with the text rows' sums)."""
    with _wgrad_overlap(patches.device):
        feat, vsaved = siglip_tower_forward(patches, params["vision"], frames, tokens, recompute)
        pj = params["projector"]
        y, psaved = projector_forward(feat, pj["w1"], pj["b1"], pj["w2"], pj["b2"])
        H = y.shape[1]
        x, n_pre, n_vis = _build_inputs_embeds(params, y, voxel_ids, pe_table, pre_ids, post_ids, frames, tokens, side, n, coord_rows, coord_pe)
        loss, dx, llm_grads = llm_forward_backward(params["llm"], x, labels, rope, n_q, n_kv, hd, recompute=recompute)
        d_embed = torch.zeros_like(params["embed"])
        text_rows = torch.cat([torch.arange(n_pre, device=x.device), torch.arange(n_pre + n_vis, x.shape[0], device=x.device)])
        dfeat, d_newline = inputs_embeds_backward(dx, n_pre, frames, text_rows, torch.cat([pre_ids, post_ids]), d_embed, side=side, n=n)
        dfeat_in, pgrads = projector_backward(dfeat.view(frames * tokens, H), psaved, pj["w1"], pj["w2"])
        vgrads = siglip_tower_backward(dfeat_in, vsaved, params["vision"], frames, tokens)
        return loss, {"vision": vgrads, "projector": pgrads, "newline": d_newline, "embed": d_embed, "llm": llm_grads}


# ------------------------------------------------------------------------------ ZeRO-2 (scripts/zero2.json:22-34)


def _leaves(tree, out=None):
    out = [] if out is None else out
    if isinstance(tree, dict):
        for k in tree:
            _leaves(tree[k], out)
    elif isinstance(tree, list):
        for v in tree:
            _leaves(v, out)
    else:
        out.append(tree)
    return out


class ZeroAdamW:
    """DeepSpeed ZeRO stage 2 as the reference configures it (scripts/zero2.json:22-34: reduce_scatter true, 2e8-element buckets): every
    rank keeps the whole 16-bit model, the ranks' gradients are reduce-scattered (averaged), each rank owns 1 / world of the f32 master
    weights and AdamW moments and updates it (v3d_adamw_step), the updated 16-bit partitions are all-gathered.  The parameters live in ONE
    flat 16-bit buffer (`params` is returned re-bound to views of it), so the gather writes them in place.
    Exchanges: v3d.distributed.reduce_scatter_grads / all_gather_params over the process group ("nccl" = RCCL keeps everything in HBM;
    with "gloo" - the one-GPU rehearsal of tests/test_gpu_zero2.py - the flat buffers are staged through host memory).
    Not yet run on more than one GPU."""

    def __init__(self, params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, bucket_elems=None, algorithm="ring",
                 lr_by_module=None, max_grad_norm=None, schedule=None):
        import torch.distributed as dist
        from . import distributed as D
        self.D, self.dist = D, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.host_staged = dist.get_backend() == "gloo"
        self.lr, self.betas, self.eps, self.weight_decay, self.t = lr, betas, eps, weight_decay, 0
        self.max_grad_norm, self.schedule, self.last_grad_norm = max_grad_norm, schedule, None
        groups = param_groups(params, lr, weight_decay, lr_by_module)
        self.bucket = bucket_elems or D.ZERO2_BUCKET_ELEMS
        self.algorithm = algorithm
        leaves = _leaves(params)
        self.numel = sum(p.numel() for p in leaves)
        self.bounds, self.per = D.partition_bounds(self.numel, self.world)
        dt, dev = leaves[0].dtype, leaves[0].device
        self.flat = torch.zeros(self.per * self.world, dtype=dt, device=dev)            # padded so that every partition has `per` elements
        off = 0
        self.views = []
        for p in leaves:
            v = self.flat[off:off + p.numel()].view(p.shape)
            v.copy_(p)
            self.views.append(v)
            off += p.numel()
        it = iter(self.views)
        self.params = _tree_map(lambda _: next(it), params)                            # the same tree over views of the flat buffer
        b = self.rank * self.per
        self.mine = self.flat[b:b + self.per]
        self.master = self.mine.float()
        self.m = torch.zeros_like(self.master)
        self.v = torch.zeros_like(self.master)
        # this rank's partition cut into runs of constant (lr, weight decay) - the optimizer groups of llava_trainer.py:446-523 laid over
        # the flat buffer (a partition boundary may fall inside a tensor, a tensor boundary inside a partition): [(lo, hi, lr, wd)], local
        self.segments = []
        off = 0
        for p, (g_lr, g_wd) in zip(leaves, groups):
            lo, hi = max(off, b) - b, min(off + p.numel(), b + self.per) - b
            off += p.numel()
            if lo >= hi:
                continue
            if self.segments and self.segments[-1][1] == lo and self.segments[-1][2:] == (g_lr, g_wd):
                self.segments[-1] = (self.segments[-1][0], hi, g_lr, g_wd)
            else:
                self.segments.append((lo, hi, g_lr, g_wd))

    def _flatten_grads(self, grads):
        """grads -> self.flat_g, walking the PARAMETER tree by key: a leaf the sample has no gradient for (the LM head of a
        grounding sample, the grounding heads of a QA sample) contributes zeros; a gradient for an unknown key, or of another shape than
        its parameter, raises; the order of the gradient dict's keys plays no part.  A leaf that already IS its slice of the flat buffer
        (the backward wrote it through grad_views()) is not copied."""
        if getattr(self, "flat_g", None) is None:
            self.flat_g = torch.zeros(self.per * self.world, dtype=self.flat.dtype, device=self.flat.device)
        views = iter(self._g_views())
        seen = [0]

        def walk(p, g, path):
            if isinstance(p, dict):
                if g is not None:
                    if not isinstance(g, dict):
                        raise V3DError(f"ZeroAdamW: gradient at {path or '<root>'} is not a dict like its parameters")
                    extra = set(g) - set(p)
                    if extra:
                        raise V3DError(f"ZeroAdamW: gradient keys {sorted(extra)} at {path or '<root>'} have no parameter")
                for k in p:
                    walk(p[k], None if g is None else g.get(k), f"{path}.{k}" if path else str(k))
            elif isinstance(p, list):
                if g is not None and (not isinstance(g, list) or len(g) != len(p)):
                    raise V3DError(f"ZeroAdamW: gradient list at {path} does not match its {len(p)} parameters")
                for i, x in enumerate(p):
                    walk(x, None if g is None else g[i], f"{path}[{i}]")
            else:
                v = next(views)
                if g is None:
                    v.zero_()
                else:
                    if tuple(g.shape) != tuple(p.shape):
                        raise V3DError(f"ZeroAdamW: gradient of {path} has shape {tuple(g.shape)}, its parameter {tuple(p.shape)}")
                    if g.data_ptr() != v.data_ptr():
                        v.copy_(g)
                seen[0] += p.numel()

        walk(self.params, grads, "")
        if seen[0] != self.numel:
            raise V3DError(f"ZeroAdamW: flattened {seen[0]} gradient elements for {self.numel} parameters")
        return self.flat_g

    def _g_views(self):
        if getattr(self, "_gv", None) is None:
            if getattr(self, "flat_g", None) is None:
                self.flat_g = torch.zeros(self.per * self.world, dtype=self.flat.dtype, device=self.flat.device)
            off, self._gv = 0, []
            for v in self.views:
                self._gv.append(self.flat_g[off:off + v.numel()].view(v.shape))
                off += v.numel()
        return self._gv

    def grad_views(self):
        """The parameter tree over views of the flat GRADIENT buffer: a backward that writes its gradients there (out=...) spares
        step() the copy of that leaf."""
        it = iter(self._g_views())
        return _tree_map(lambda _: next(it), self.params)

    def step(self, grads, grad_scale=1.0):
        self.t += 1
        flat_g = self._flatten_grads(grads)
        if self.host_staged:
            part = self.D.reduce_scatter_grads(flat_g.cpu(), self.bucket, average=True, algorithm=self.algorithm).to(self.flat.device)
        else:
            part = self.D.reduce_scatter_grads(flat_g, self.bucket, average=True, algorithm=self.algorithm)
        part = part.contiguous()
        if self.max_grad_norm is not None:
            # the norm of the AVERAGED gradient: each rank holds one partition of it after the reduce-scatter, so the squared norm is the
            # sum of the partitions' (one all-reduce of a scalar; DeepSpeed's stage-2 clipping does the same)
            sq = global_grad_norm([part])
            if self.world > 1:
                sq = sq.cpu() if self.host_staged else sq
                self.dist.all_reduce(sq, op=self.dist.ReduceOp.SUM)
            self.last_grad_norm = float(sq.sqrt()) * abs(grad_scale)
            grad_scale = grad_scale * clip_coefficient(self.last_grad_norm, self.max_grad_norm)
        mult = self.schedule(self.t) if self.schedule is not None else 1.0
        for lo, hi, g_lr, g_wd in self.segments:
            ops.adamw_step(self.master[lo:hi], self.m[lo:hi], self.v[lo:hi], part[lo:hi], p16=self.mine[lo:hi], lr=g_lr * mult, betas=self.betas,
                           eps=self.eps, weight_decay=g_wd, step=self.t, grad_scale=grad_scale)
        if self.host_staged:
            full = self.D.all_gather_params(self.mine.cpu(), self.per * self.world, self.bucket).to(self.flat.device)
            self.flat.copy_(full)
        else:
            self.D.all_gather_params_into(self.flat, self.rank, self.per, self.bucket)      # in place: no second copy of the model
        return self.params


# ------------------------------------------------------------------------------ grounding samples (ScanRefer / Multi3DRefer)


def _ground_head_forward(x, hp):
    """ground_head_obj / ground_head_query (llava_qwen.py:99-110): Linear, ReLU, LayerNorm (eps 1e-5), Linear."""
    h1 = ops.gemm(x, hp["w0"], bias=hp["b0"], epilogue=ops.EPI_BIAS_RELU)
    hn = ops.layernorm(h1, hp["ln_w"], hp["ln_b"], 1e-5)
    return ops.gemm(hn, hp["w3"], bias=hp["b3"], epilogue=ops.EPI_BIAS), (x, h1, hn)


def _ground_head_backward(dout, saved, hp):
    x, h1, hn = saved
    dhn, d_w3, d_b3 = linear_backward(hn, hp["w3"], dout, need_db=True)
    dh1, d_lnw, d_lnb = ops.layernorm_grad(h1, hp["ln_w"], dhn, 1e-5)
    dz = ops.gelu_grad(h1, dh1, tanh_form=2)                                     # ReLU: h1 > 0 iff its pre-activation was
    dx, d_w0, d_b0 = linear_backward(x, hp["w0"], dz, need_db=True)
    return dx, {"w0": d_w0, "b0": d_b0, "ln_w": d_lnw, "ln_b": d_lnb, "w3": d_w3, "b3": d_b3}


def ground_sample_forward_backward(params, patches, voxel_ids, pe_table, pre_ids, post_ids, ground_row, obj_mask, box_pe, positive, rope, frames,
                                   n_q, n_kv, hd, temperature=0.07, tokens=729, side=27, n=14, eps=1e-6, recompute=False):
    """A grounding sample of the joint training (llava_qwen.py:121-160 -> predict_box :239-310, object features llava_arch.py:351-376,
    479-501): the same tower -> projector -> splice -> decoder as sample_forward_backward, but the loss is the infonce loss between the
    <ground> token's final hidden state and the object proposals' features (masked means of the projector's patch rows + the box-centre
    PE, and the learnt zero-target row), each through its head.  ground_row: the <ground> label token's row in inputs_embeds; obj_mask
    uint8 [n_obj, frames * tokens] (v3d_object_patch_mask); box_pe [n_obj, H] (sin3d PE of the discretised box centres: no gradient);
    positive uint8 [n_obj + 1] (the last entry = the zero-target, set when the sample has no target box).
    params additionally holds "ground": {"obj": head, "query": head, "zero_target" [H]} (head = {w0, b0, ln_w, ln_b, w3, b3}).
    Returns (loss, scores f32 [n_obj + 1], grads) - grads["llm"] has no "lm_head" entry (the LM head takes no part)."""
    with _wgrad_overlap(patches.device):
        feat, vsaved = siglip_tower_forward(patches, params["vision"], frames, tokens, recompute)
        pj, gp = params["projector"], params["ground"]
        y, psaved = projector_forward(feat, pj["w1"], pj["b1"], pj["w2"], pj["b2"])
        H = y.shape[1]
        x, n_pre, n_vis = _build_inputs_embeds(params, y, voxel_ids, pe_table, pre_ids, post_ids, frames, tokens, side, n)
        h, lsaved = llm_layers_forward(params["llm"], x, rope, n_q, n_kv, hd, eps, recompute)
        # predict_box: the query is the final-norm hidden state of the <ground> row; the objects are masked means of the projector rows
        hq = h[ground_row:ground_row + 1]
        query_in = ops.rmsnorm(hq, params["llm"]["norm"], eps)
        obj_feat = ops.masked_mean(y, obj_mask, add=box_pe)
        of = torch.cat([obj_feat, gp["zero_target"][None].to(obj_feat.dtype)], 0).contiguous()
        obj_out, osaved = _ground_head_forward(of, gp["obj"])
        q_out, qsaved = _ground_head_forward(query_in, gp["query"])
        loss, scores, d_obj_out, d_q_out = ops.ground_infonce(obj_out, q_out[0], positive, temperature)
        # backward
        d_of, g_obj = _ground_head_backward(d_obj_out, osaved, gp["obj"])
        d_qin, g_query = _ground_head_backward(d_q_out[None].contiguous(), qsaved, gp["query"])
        d_hq, d_norm = ops.rmsnorm_grad(hq, params["llm"]["norm"], d_qin, eps)
        dh = torch.zeros_like(h)
        ops.copy_rows(d_hq, dh[ground_row:ground_row + 1])
        dx, layer_grads = llm_layers_backward(dh, lsaved, params["llm"], rope, n_q, n_kv, hd, eps)
        d_embed = torch.zeros_like(params["embed"])
        text_rows = torch.cat([torch.arange(n_pre, device=x.device), torch.arange(n_pre + n_vis, x.shape[0], device=x.device)])
        dfeat, d_newline = inputs_embeds_backward(dx, n_pre, frames, text_rows, torch.cat([pre_ids, post_ids]), d_embed, side=side, n=n)
        dy = dfeat.view(frames * tokens, H)
        ops.masked_mean_grad(obj_mask, d_of[:-1].contiguous(), dy, accumulate=True)       # the object features' share of the projector rows' gradient
        dfeat_in, pgrads = projector_backward(dy, psaved, pj["w1"], pj["w2"])
        vgrads = siglip_tower_backward(dfeat_in, vsaved, params["vision"], frames, tokens)
        return loss, scores, {"vision": vgrads, "projector": pgrads, "newline": d_newline, "embed": d_embed,
                              "llm": {"layers": layer_grads, "norm": d_norm},
                              "ground": {"obj": g_obj, "query": g_query, "zero_target": d_of[-1].contiguous()}}
