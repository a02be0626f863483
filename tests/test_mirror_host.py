"""CPU-only tests of the `llava` mirror package's host logic against the reference goldens."""
import json
import os

import numpy as np
import pytest
import torch

import llava.video_utils as vu
from llava.utils_3d import convert_pc_to_box

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def bare_processor(**kw):
    vp = object.__new__(vu.VideoProcessor)     # no dataset files in the test container
    vp.video_folder = "data"
    vp.voxel_size, vp.min_xyz_range, vp.max_xyz_range = 0.1, None, None
    vp.frame_sampling_strategy = "uniform"
    for k, v in kw.items():
        setattr(vp, k, v)
    return vp


def test_surface_names_exist():
    import llava.mm_utils as mm
    from v3d.token_ids import IGNORE_INDEX, IMAGE_TOKEN_INDEX
    from llava.model.language_model.llava_qwen import LlavaQwenForCausalLM  # noqa: F401
    from llava.model.llava_arch import LlavaMetaForCausalLM
    from llava.model.position_encoding import PositionEmbeddingSine3D  # noqa: F401
    assert IGNORE_INDEX == -100 and IMAGE_TOKEN_INDEX == -200
    for name in ("unproject", "VideoProcessor", "merge_video_dict", "load_matrix_from_txt"):
        assert hasattr(vu, name)
    for name in ("tokenizer_image_token", "get_model_name_from_path", "KeywordsStoppingCriteria"):
        assert hasattr(mm, name)
    for name in ("get_2dPool", "average_coordinate_in_patch", "discrete_coords", "add_token_per_grid"):
        assert hasattr(LlavaMetaForCausalLM, name)


def test_uniform_and_mc_sampling_golden():
    with open(os.path.join(GOLDEN, "frame_sampling.json")) as f:
        g = json.load(f)
    for key, want in g["uniform"].items():
        n = int(key.split("_")[0][1:])
        vp = bare_processor(scene={"s": {"images": [{"img_path": f"posed_images/s/{i * 10:05d}.jpg"} for i in range(n)]}})
        if key.endswith("default"):
            files = vp.sample_frame_files("s", force_sample=False, frames_upbound=32)
        else:
            files = vp.sample_frame_files("s", force_sample=True, frames_upbound=int(key.split("_F")[1]))
        assert [int(os.path.basename(f).split(".")[0]) // 10 for f in files] == want, key
    for key, want in g["mc"].items():
        strat, F = key.rsplit("_F", 1)
        vp = bare_processor(frame_sampling_strategy=strat, mc_sampling_files={"s": json.loads(json.dumps(g["mc_entry"]))})
        assert vp.sample_frame_files_mc("s", frames_upbound=int(F)) == want, key


def test_discrete_point_golden(golden):
    g = golden("discrete_point")
    vp = bare_processor()
    assert np.array_equal(np.array(vp.discrete_point(g["pts"].tolist()), np.int32), g["ids_norange"])
    vp = bare_processor(min_xyz_range=torch.tensor([-15, -15, -5]), max_xyz_range=torch.tensor([15, 15, 5]))
    assert np.array_equal(np.array(vp.discrete_point(g["pts"].tolist()), np.int32), g["ids_range"])


def test_convert_pc_to_box_golden(golden):
    g = golden("convert_pc_to_box")
    c, s = convert_pc_to_box(g["pc"])
    assert np.array_equal(np.array(c), g["center"]) and np.array_equal(np.array(s), g["size"])


def test_merge_video_dict():
    a = {"world_coords": torch.zeros(2, 4, 4, 3), "images": torch.zeros(2, 3, 4, 4), "objects": torch.zeros(5, 6), "box_input": None}
    out = vu.merge_video_dict([a])
    assert out["world_coords"].shape == (1, 2, 4, 4, 3) and out["box_input"].numel() == 0


def test_mm_utils_helpers():
    import llava.mm_utils as mm

    class Tok:
        bos_token_id = 1

        def __call__(self, s):
            return types_ns(input_ids=[1] + [ord(c) for c in s])

    import types
    types_ns = types.SimpleNamespace
    ids = mm.tokenizer_image_token("ab<image>cd", Tok(), return_tensors="pt")
    assert ids.tolist() == [1, 97, 98, -200, 99, 100]
    assert mm.get_model_name_from_path("/x/llava-qwen/checkpoint-500/") == "llava-qwen_checkpoint-500"
    assert mm.get_model_name_from_path("/x/llava-qwen") == "llava-qwen"


def test_no_cpu_fallback():
    from v3d import V3DError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(V3DError):
        vu.unproject(torch.eye(4)[None], torch.eye(4)[None], torch.zeros(1, 4, 4))


def test_load_depth_pose_matches_reference_loader(tmp_path):
    """a4, host half (video_utils.py:206-230): 16-bit PNG -> the u16 values (incl. > 32767, which must not wrap), pose text ->
    f64, axis_align @ pose in f64 and only then .float() - against the reference's own loader output
    (tests/golden/world_coords.npz, oracle/gen_golden.py g_world_coords), bit for bit; the oracle's unproject on these
    inputs reproduces the reference's world coordinates."""
    from scene_files import write_frames
    from oracle import v3d_oracle as O
    g = np.load(os.path.join(GOLDEN, "world_coords.npz"))
    files = write_frames(str(tmp_path / "scene0000_00"), g["depth"], g["poses"])
    vp = bare_processor(scene={"scannet/scene0000_00": {"axis_align_matrix": g["axis_align"].tolist(), "depth_cam2img": g["cam2img"].tolist()}})
    depth, K, pose = vp._load_depth_pose("scannet/scene0000_00", files)
    assert depth.dtype == torch.int16 and tuple(depth.shape) == (3, 48, 64)
    assert np.array_equal(depth.numpy().view(np.uint16), g["depth"]) and int(depth.numpy().view(np.uint16).max()) == 40000
    assert pose.dtype == torch.float32 and np.array_equal(pose.numpy(), g["aligned_f32"])
    assert K.shape == (3, 4, 4) and np.array_equal(K[1].numpy(), g["cam2img"].astype(np.float32))
    world = O.unproject(K.numpy(), pose.numpy(), depth.numpy().view(np.uint16).astype(np.float32))
    np.testing.assert_allclose(world, g["world"], rtol=2e-6, atol=2e-6)
    b = g["boundry"]
    flat = world.reshape(-1, 3)
    np.testing.assert_allclose([flat[:, 0].min(), flat[:, 0].max(), flat[:, 1].min(), flat[:, 1].max(), flat[:, 2].min(), flat[:, 2].max()],
                               b, rtol=2e-6, atol=2e-6)
