"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/v3d.h declares,
rejects bad arguments with a message, and its host-only helpers match the reference goldens."""
import json
import os
import re

import numpy as np
import pytest

from v3d import _native, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "v3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(v3d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.lib()
    names = declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/v3d.h but not exported"
        assert n in _native.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.v3d_abi_version() == 7


def test_invalid_arguments_report_an_error():
    lib = _native.lib()
    rc = lib.v3d_unproject_f32(None, None, None, None, 1, 1, 1, None)
    assert rc == -1
    assert b"null pointer" in lib.v3d_last_error()
    with pytest.raises(_native.V3DError):
        ops.uniform_frame_indices(0, 4)


def test_linear_decode_rows_fuses_norm_is_a_host_rule(monkeypatch):
    """r04: which v3d_linear_decode_rows calls take a norm_weight with more than four rows (host-only query; no GPU call): more than four and
    at most 32 rows, 9..32 K tiles of 128 (K <= 4096), outputs in 16-row groups (SwiGLU: 128-row gate|up tiles), no residual epilogue - and
    never with V3D_DEC_V2=0 / 3 (the other kernels) or V3D_DEC_FUSE_NORM=0."""
    monkeypatch.delenv("V3D_DEC_V2", raising=False)
    monkeypatch.delenv("V3D_DEC_FUSE_NORM", raising=False)
    f = ops.linear_decode_rows_fuses_norm
    assert f(32, 4608, 3584, ops.DEC_BIAS) and f(5, 37888, 3584, ops.DEC_SWIGLU) and f(16, 152064, 3584) and f(8, 256, 1152)
    assert not f(4, 4608, 3584) and not f(33, 4608, 3584)                      # row count
    assert not f(32, 3584, 18944) and not f(32, 4608, 1024) and not f(32, 4608, 3592)      # K: too long, 8 tiles, not a multiple of 128
    assert not f(32, 4600, 3584) and not f(32, 37888 + 64, 3584, ops.DEC_SWIGLU)           # N
    assert not f(32, 3584, 3584, ops.DEC_RES)
    for v in ("0", "3"):
        monkeypatch.setenv("V3D_DEC_V2", v)
        assert not f(32, 4608, 3584)
    monkeypatch.setenv("V3D_DEC_V2", "2")
    assert f(32, 4608, 3584)
    monkeypatch.setenv("V3D_DEC_FUSE_NORM", "0")
    assert not f(32, 4608, 3584)


def test_ops_refuse_cpu_tensors():
    import torch
    with pytest.raises(_native.V3DError, match="no CPU path"):
        ops.unproject(torch.eye(4)[None], torch.eye(4)[None], torch.zeros(1, 4, 4))


def test_uniform_frame_indices_golden():
    with open(os.path.join(ROOT, "tests", "golden", "frame_sampling.json")) as f:
        g = json.load(f)["uniform"]
    for key, want in g.items():
        n = int(key.split("_")[0][1:])
        F = 10 if key.endswith("default") else int(key.split("_F")[1])
        assert ops.uniform_frame_indices(n, F) == want, key
    # against numpy for a sweep
    for n in range(1, 700, 7):
        for F in (1, 2, 8, 10, 32):
            assert ops.uniform_frame_indices(n, F) == np.linspace(0, n - 1, F, dtype=int).tolist()


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_greedy_cover_golden(golden, case):
    g = golden("greedy_cover")
    keys = np.rint(g[case + "_world"].astype(np.float32) / np.float32(0.1)).astype(np.int32)
    sel, gains, n_all, n_sel = ops.greedy_cover(keys, g[case + "_pc"])
    assert sel.tolist() == g[case + "_select"].tolist()
    assert gains.tolist() == g[case + "_voxel_nums"].tolist()
    assert n_all == int(g[case + "_num_all"]) and n_sel == int(g[case + "_num_sel"])


def test_greedy_cover_edge_cases():
    # one frame, empty scene set, duplicate frames (ties -> lowest position)
    keys = np.zeros((3, 4, 3), np.int32)
    keys[1] = 5
    sel, gains, n_all, n_sel = ops.greedy_cover(keys, np.zeros((0, 3), np.int32))
    assert sel.tolist() == [0, 1, 2] and gains.tolist() == [0, 0, 0] and n_all == 0 and n_sel == 0
    sel, gains, n_all, n_sel = ops.greedy_cover(keys, np.array([[0, 0, 0], [5, 5, 5]], np.int32))
    assert sel.tolist() == [0, 1, 2] and gains.tolist() == [1, 1, 0] and n_all == 2 and n_sel == 2
    with pytest.raises(_native.V3DError):
        ops.greedy_cover(np.full((1, 1, 3), 1 << 22, np.int32), np.zeros((1, 3), np.int32))


def test_host_half_under_asan_ubsan():
    """csrc/host.cpp + its self-test driver built with g++ -fsanitize=address,undefined (sanitizers on the CPU
    build): frame-index and greedy-cover helpers on ragged sizes against a brute-force set implementation, and every
    argument-validation branch."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++ on this machine")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "video-3d-llm_amd", "csrc"), "sanitize"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "host_selftest ok" in r.stdout


def _plan(M, N, K, slots=256):
    import ctypes
    out = [ctypes.c_int(0) for _ in range(4)]
    rc = _native.lib().v3d_gemm_plan_host(M, N, K, slots, *[ctypes.byref(o) for o in out])
    assert rc == 0, _native.lib().v3d_last_error()
    return tuple(o.value for o in out)          # kernel, tiles, dp, split


def test_gemm_plan_of_the_paths_shapes():
    """The tile choice for the shapes of the path on a 256-CU chip (host code only): the decoder's gate/up and qkv take whole rounds
    of the 256 x 256 tile, o_proj the 192-row tile, down_proj (378 tiles = 1.48 rounds, K = 18944) the split-K tail, a batch of
    question rows the 128 x 128 tile except for its down_proj (56 tiles cut in four)."""
    assert _plan(6794, 37888, 3584) == (2, 27 * 148, -1, 1)
    assert _plan(6794, 4608, 3584) == (2, 27 * 18, -1, 1)
    assert _plan(6794, 3584, 3584)[0] == 3
    assert _plan(6794, 3584, 18944) == (2, 378, 1, 2)
    assert _plan(960, 3584, 18944) == (2, 56, 0, 4)
    assert _plan(960, 4608, 3584)[0] == 1
    assert _plan(4, 3584, 3584)[0] == 0                        # skinny (decode) path
    assert _plan(23328, 1152, 1152)[0] == 1                    # N % 256 != 0: only the 128 x 128 tile fits


def test_gemm_split_k_plans_are_well_formed():
    """Every split-K plan the launcher can produce must give each K-chunk a workgroup and each chunk-0 workgroup existing partners -
    a plan that does not would leave a workgroup polling a flag nobody raises.  Sweep of shapes and chip sizes; the item list
    is rebuilt here the way the kernel walks it (gemm256pp_kernel: XCD-major id g = xcd * wx + idx, item g = (chunk g / T, tile g mod T))."""
    rng = np.random.default_rng(5)
    seen = 0
    for _ in range(4000):
        M = int(rng.integers(9, 30000))
        N = 256 * int(rng.integers(1, 160))
        K = 64 * int(rng.integers(1, 320))
        slots = int(rng.choice([8, 16, 64, 104, 256, 304]))
        kernel, tiles, dp, split = _plan(M, N, K, slots)
        assert kernel in (1, 2, 3)
        if dp < 0:
            assert split == 1
            continue
        seen += 1
        grid = slots & ~7
        assert kernel == 2 and tiles == ((M + 255) // 256) * (N // 256)
        nt = K // 64
        T = tiles - dp * grid                                  # tiles left after the whole-tile rounds
        assert 2 <= split <= 4 and dp >= 0 and T > 0
        assert T * split <= grid, "a chunk without a workgroup"
        assert nt % 2 == 0 and (nt // 2) >= 2 * split, "a chunk shorter than two granules"
        wx = grid // 8
        owners = {}
        for b in range(grid):                                  # workgroup id -> its tail item
            g = (b & 7) * wx + (b >> 3)
            if g < split * T:
                owners[(g // T, g % T)] = b
        assert len(owners) == split * T                        # every (chunk, tile) exactly once
        for j in range(T):
            b0 = owners[(0, j)]
            g0 = (b0 & 7) * wx + (b0 >> 3)
            for c in range(1, split):
                fg = c * T + g0                                # the kernel's partner formula
                partner = (fg % wx) * 8 + fg // wx
                assert partner == owners[(c, j)] and partner < grid
            gpt = nt // 2
            cuts = [2 * (c * gpt // split) for c in range(split + 1)]
            assert cuts[0] == 0 and cuts[-1] == nt and all(b - a >= 2 for a, b in zip(cuts, cuts[1:]))
    assert seen > 50


def test_training_entry_points_reject_bad_arguments_before_any_launch():
    """Argument checks of the training step's entry points run on the host, before a kernel is launched: no GPU needed."""
    import ctypes
    l = _native.lib()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p = (p + 15) & ~15
    bf16 = 2
    bad = [
        l.v3d_transpose(None, 8, 8, 8, p, 8, 8, bf16, None),                                   # null source
        l.v3d_transpose(p, 12, 8, 12, p, 8, 8, bf16, None),                                    # cols % 8
        l.v3d_transpose(p, 8, 8, 8, p, 8, 8, 0, None),                                         # f32: 16-bit only
        l.v3d_colsum(p, 8, 4, 8, bf16, None, p, 0, None),                                      # null workspace
        l.v3d_rmsnorm_grad(p, 8, p, p, 8, None, 0, p, 8, ctypes.cast(p, ctypes.c_void_p), p, 0, 4, 4096, 1e-6, bf16, None),    # cols > 3584
        l.v3d_layernorm_grad(p, 8, p, p, 8, None, 0, p, 8, ctypes.cast(p, ctypes.c_void_p), p, p, 0, 4, 4096, 1e-6, bf16, None),  # cols > 2048
        l.v3d_swiglu(p, 8, p, 8, 4, 8, bf16, None),                                            # row stride < 2 * inter
        l.v3d_gelu(p, 8, p, 8, 4, 12, 0, bf16, None),                                          # cols % 8
        l.v3d_causal_softmax_rows(p, 8, p, 8, 4, 16, 8, 0, 1.0, bf16, None),                   # n_keys > cols
        l.v3d_attention_backward(p, p, p, p, p, ctypes.cast(p, ctypes.c_void_p), p, p, p, bf16, 1, 64, 3, 2, 512, 512, 512, 512, 512, 512, 512, 512,
                                 0, 0, 0, 0, 0, 0, 0, 0, 1, 1.0, p, 1 << 30, None),             # 3 query heads over 2 kv heads
        l.v3d_attention_backward(p, p, p, p, p, ctypes.cast(p, ctypes.c_void_p), p, p, p, bf16, 1, 64, 4, 2, 512, 512, 512, 512, 512, 512, 512, 512,
                                 0, 0, 0, 0, 0, 0, 0, 0, 1, 1.0, p, 16, None),                  # workspace too small
        l.v3d_adamw_step(ctypes.cast(p, ctypes.c_void_p), ctypes.cast(p, ctypes.c_void_p), ctypes.cast(p, ctypes.c_void_p), p, bf16, None, 0, 8, 1e-3, 0.9, 0.999,
                         1e-8, 0.0, 0, 1.0, None),                                             # step counts from 1
        l.v3d_embed_grad(p, 8, 1, None, None, 1, 8, p, 8, 1, bf16, None),                            # null index arrays
    ]
    assert all(rc < 0 for rc in bad), bad
    assert l.v3d_attention_backward_workspace_bytes(2, 100, 4) == (2 * 4 * 100 + 2 * 2 * 4 * 100 * 128) * 4
    assert l.v3d_colsum_workspace_bytes(65, 16) == 3 * 16 * 4
