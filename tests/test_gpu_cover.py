"""a3 on the device: greedy max-coverage frame selection (scripts/3d/preprocessing/max_coverage_sampling.py:44-94)
against (1) the golden produced by the reference's own loop, (2) the numpy oracle and (3) the exact host C++ version,
all BIT-EXACT (integer / index work): picks, per-pick gains (voxel_nums), num_all_voxels, num_select_voxels."""
import numpy as np
import pytest
import torch

from oracle import v3d_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_device_cover_matches_reference_golden(golden, case):
    from v3d import ops
    g = golden("greedy_cover")
    world = torch.from_numpy(g[case + "_world"]).cuda()
    keys = ops.voxel_keys(world)
    assert np.array_equal(keys.cpu().numpy(), O.voxel_keys(g[case + "_world"]))
    n = keys.shape[0]
    sel, gains, n_all, n_sel = ops.greedy_cover_device(keys.reshape(n, -1, 3), torch.from_numpy(g[case + "_pc"]).cuda(), max_frames=32)
    assert sel.tolist() == g[case + "_select"].tolist()
    assert gains.tolist() == g[case + "_voxel_nums"].tolist()
    assert (n_all, n_sel) == (int(g[case + "_num_all"]), int(g[case + "_num_sel"]))


@pytest.mark.parametrize("n_frames,pts,m,max_frames,seed", [(40, 3000, 5000, 32, 0), (7, 500, 300, 32, 1), (64, 20000, 40000, 16, 2),
                                                              (300, 4096, 60000, 32, 3)])
def test_device_cover_matches_host_and_oracle(n_frames, pts, m, max_frames, seed):
    """Random scenes with many ties (small coordinate range) and frames whose voxels are partly outside the scene set."""
    from v3d import ops
    rng = np.random.default_rng(seed)
    span = int(round(m ** (1 / 3))) + 3
    scene = np.unique(rng.integers(-span, span, size=(m, 3), dtype=np.int32), axis=0)
    keys = np.empty((n_frames, pts, 3), np.int32)
    for f in range(n_frames):                     # each frame sees a window of the scene
        c = rng.integers(-span, span, size=3)
        keys[f] = c + rng.integers(-span // 2 - 1, span // 2 + 2, size=(pts, 3))
    keys[1] = keys[0]                             # exact tie between two frames -> lowest position wins
    want = ops.greedy_cover(keys, scene, max_frames)
    got = ops.greedy_cover_device(torch.from_numpy(keys).cuda(), torch.from_numpy(scene).cuda(), max_frames)
    assert got[0].tolist() == want[0].tolist()
    assert got[1].tolist() == want[1].tolist()
    assert got[2:] == want[2:]
    if n_frames * pts <= 200000:                  # the pure-python oracle only at small sizes
        world = keys.astype(np.float32) * np.float32(0.1)
        assert np.array_equal(O.voxel_keys(world), keys)
        osel, ogain, oall, osel_n = O.greedy_max_coverage(world[:, :, None, :], scene, max_frames=max_frames)
        assert got[0].tolist() == osel.tolist() and got[1].tolist() == ogain.tolist() and got[2:] == (oall, osel_n)


def test_device_cover_edge_cases():
    from v3d import ops
    from v3d._native import V3DError
    keys = torch.zeros((3, 4, 3), dtype=torch.int32, device="cuda")
    sel, gains, n_all, n_sel = ops.greedy_cover_device(keys, torch.zeros((0, 3), dtype=torch.int32, device="cuda"))
    assert sel.tolist() == [0, 1, 2] and gains.tolist() == [0, 0, 0] and (n_all, n_sel) == (0, 0)     # empty scene: order by position
    sel, gains, n_all, n_sel = ops.greedy_cover_device(keys, torch.tensor([[0, 0, 0], [5, 5, 5]], dtype=torch.int32, device="cuda"))
    assert sel.tolist() == [0, 1, 2] and gains.tolist() == [1, 0, 0] and (n_all, n_sel) == (1, 1)
    with pytest.raises(V3DError):
        ops.greedy_cover_device(keys, torch.full((1, 3), 1 << 22, dtype=torch.int32, device="cuda"))
    far = torch.full((2, 2, 3), 1 << 22, dtype=torch.int32, device="cuda")                           # unpackable frame keys: ignored
    sel, gains, n_all, n_sel = ops.greedy_cover_device(far, torch.zeros((1, 3), dtype=torch.int32, device="cuda"))
    assert gains.tolist() == [0, 0] and n_all == 0
