"""First kernels of the training step (BASELINE configs[4]) against torch autograd on the CPU in f32 (a floating-point kernel:
the torch f32 reference is the checker, tolerances stated per test):
  * v3d_cross_entropy / _grad = the shifted CrossEntropyLoss of Qwen2ForCausalLM.forward (modeling_qwen2.py:1195-1205);
  * v3d_visual_tokens_grad = backward of get_2dPool (bilinear 27 -> 14) + PE add (passes through) + image_newline rows,
    composed as prepare_inputs_labels_for_multimodal does (llava_arch.py:191-210, 307-328, 506-517)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from v3d import ops as _ops
    return _ops


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-6), (torch.float16, 2e-6)])
@pytest.mark.parametrize("S,V", [(37, 320), (300, 1000), (64, 152064)])
def test_cross_entropy_and_gradient(ops, dt, tol, S, V):
    g = torch.Generator().manual_seed(S + V)
    logits = (torch.randn(S, V, generator=g) * 3).to(dt)
    labels = torch.randint(0, V, (S,), generator=g)
    labels[torch.rand(S, generator=g) < 0.4] = -100                      # prompt / visual rows carry IGNORE_INDEX
    labels[1] = 5                                                         # at least one valid target
    ref_in = logits.float().requires_grad_(True)                          # the reference casts the logits to f32 first (:1190-1192)
    ref = F.cross_entropy(ref_in[:-1], labels[1:], ignore_index=-100)
    ref.backward()
    loss, st = ops.cross_entropy(logits.cuda(), labels.cuda())
    assert abs(loss.item() - ref.item()) <= tol * max(1.0, abs(ref.item()))
    assert int(st[3][1].item()) == int((labels[1:] != -100).sum())
    grad = ops.cross_entropy_grad(st, dtype=torch.float32).cpu()
    assert grad.shape == (S, V) and not bool(grad[-1].any())              # nothing predicts past the last position
    assert torch.allclose(grad, ref_in.grad, rtol=1e-4, atol=1e-7)
    assert not bool(grad[:-1][labels[1:] == -100].any())
    if dt != torch.float32:                                               # gradient written in the logits' dtype
        g16 = ops.cross_entropy_grad(st, upstream=2.0).float().cpu()
        assert torch.allclose(g16, 2 * ref_in.grad, rtol=2e-2 if dt == torch.bfloat16 else 2e-3, atol=1e-6)


def test_cross_entropy_all_rows_ignored_is_nan(ops):
    logits = torch.randn(9, 64).cuda()
    labels = torch.full((9,), -100, dtype=torch.int64).cuda()
    loss, st = ops.cross_entropy(logits, labels)
    assert torch.isnan(loss) and int(st[3][1].item()) == 0               # torch's CrossEntropyLoss(mean) gives nan too


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 1.2e-2), (torch.float16, 2e-3)])
def test_visual_tokens_backward(ops, dt, tol):
    V, C, side, n = 3, 256, 27, 14
    g = torch.Generator().manual_seed(7)
    feat = torch.randn(V, side * side, C, generator=g)
    newline = torch.randn(C, generator=g)
    pe = torch.randn(V, n * n, C, generator=g)                            # stands in for PE(voxel ids): no gradient flows into it
    up = torch.randn(V * n * (n + 1), C, generator=g).to(dt)             # upstream gradient d loss / d tokens
    f = feat.clone().requires_grad_(True)
    nl = newline.clone().requires_grad_(True)
    x = f.view(V, side, side, C).permute(0, 3, 1, 2)
    pooled = F.interpolate(x, size=[n, n], mode="bilinear").permute(0, 2, 3, 1)                  # get_2dPool, llava_arch.py:202-204
    tok = pooled + pe.view(V, n, n, C)                                                            # :515
    seq = torch.cat([tok, nl[None, None, None, :].expand(V, n, 1, C)], 2).reshape(-1, C)         # add_token_per_grid, :307-328
    (seq * up.float()).sum().backward()
    dfeat, dnl = ops.visual_tokens_grad(up.cuda(), V, side, n, newline=True)
    want = f.grad
    err = (dfeat.float().cpu() - want).abs()
    assert bool((err <= tol * (want.abs() + 1.0)).all()), err.max()
    assert torch.allclose(dnl.cpu(), nl.grad, rtol=1e-5, atol=1e-4)
    # without newline rows
    f2 = feat.clone().requires_grad_(True)
    pooled2 = F.interpolate(f2.view(V, side, side, C).permute(0, 3, 1, 2), size=[n, n], mode="bilinear").permute(0, 2, 3, 1)
    up2 = torch.randn(V * n * n, C, generator=g).to(dt)
    (pooled2.reshape(-1, C) * up2.float()).sum().backward()
    d2, none = ops.visual_tokens_grad(up2.cuda(), V, side, n, newline=False)
    assert none is None
    assert bool(((d2.float().cpu() - f2.grad).abs() <= tol * (f2.grad.abs() + 1.0)).all())
