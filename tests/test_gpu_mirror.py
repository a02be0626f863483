"""GPU tests of the `llava` mirror surface: same names / call contracts as the reference, results against its goldens."""
import types

import numpy as np
import pytest
import torch

from oracle import v3d_oracle as O

pytestmark = pytest.mark.gpu


def test_unproject_mirror_host_tensors(golden):
    import llava.video_utils as vu
    g = golden("unproject")
    out = vu.unproject(torch.from_numpy(g["intrinsics"]), torch.from_numpy(g["poses"]), torch.from_numpy(g["depth"].astype(np.float32)))
    assert not out.is_cuda
    np.testing.assert_allclose(out.numpy(), g["world"], rtol=2e-6, atol=2e-6)


def test_position_embedding_module(golden):
    from llava.model.position_encoding import PositionEmbeddingSine3D
    g = golden("sin3d_small")
    pe = PositionEmbeddingSine3D(96)
    out = pe(torch.from_numpy(g["xyz"]).cuda())
    np.testing.assert_allclose(out.cpu().numpy(), g["pe96"], rtol=0, atol=5e-7)
    g2, t = golden("sin3d_tokens_3584"), golden("sin3d_table_3584")
    pe = PositionEmbeddingSine3D(3584)
    assert np.array_equal(pe._dim_t.numpy(), t["dim_t"])
    o16 = pe(torch.from_numpy(g2["ids"].astype(np.float32)).half().cuda()).float().cpu().numpy()
    ref = g2["pe_f16"].astype(np.float32)
    assert np.max(np.abs(o16 - ref)) <= 2.0 ** -10 and (o16 != ref).mean() < 2e-3


def test_arch_mixin_methods(golden):
    from llava.model.llava_arch import LlavaMetaForCausalLM

    class M(LlavaMetaForCausalLM):
        def __init__(self, newline):
            self.config = types.SimpleNamespace(mm_spatial_pool_mode="bilinear", voxel_size=0.1, min_xyz_range=[-15, -15, -5],
                                                max_xyz_range=[15, 15, 5])
            self._m = types.SimpleNamespace(image_newline=newline)

        def get_model(self):
            return self._m

        def get_vision_tower(self):
            return types.SimpleNamespace(num_patches_per_side=27)

    g = golden("add_token_per_grid")
    m = M(torch.from_numpy(g["newline"]).cuda())
    assert np.array_equal(m.add_token_per_grid(torch.from_numpy(g["feat"]).cuda()).cpu().numpy(), g["out"])
    gp = golden("pool2d_bilinear")
    np.testing.assert_allclose(m.get_2dPool(torch.from_numpy(gp["feat"]).cuda()).cpu().numpy(), gp["out_f32"], rtol=1e-6, atol=1e-6)
    gc = golden("coord_pool")
    x16 = torch.from_numpy(gc["coords_f16"]).cuda()
    avg = m.average_coordinate_in_patch(x16)
    assert np.array_equal(avg.cpu().numpy(), gc["avg_f16"])
    assert np.array_equal(m.discrete_coords(avg).cpu().numpy(), gc["vox_f16"])


def test_llava_qwen_generate_and_grounding_call_contract():
    """The overlay's LlavaQwenForCausalLM keeps the 3-D eval drivers' call shapes
    (model_scanqa.py:173-185, model_scanrefer.py:165-173) on a tiny random-init model."""
    from llava.model.language_model.llava_qwen import LlavaQwenConfig, LlavaQwenForCausalLM
    from v3d.engine import EngineConfig, LlmConfig, VitConfig, random_state_dict
    cfg = EngineConfig(vit=VitConfig(hidden=144, inter=272, layers=1, heads=2),
                       llm=LlmConfig(hidden=256, inter=384, layers=1, heads=2, kv_heads=1, vocab=320, max_pos=1024))
    sd = random_state_dict(cfg, torch.float32, "cpu", seed=9, std=0.05, ground_head=True)
    hf = LlavaQwenConfig(vocab_size=320, hidden_size=256, intermediate_size=384, num_hidden_layers=1, num_attention_heads=2,
                         num_key_value_heads=1, max_position_embeddings=1024, rms_norm_eps=1e-6, rope_theta=1000000.0)
    hf.rope_theta, hf.ground_head_type, hf.world_position_embedding_type = 1000000.0, "infonce", "avg-discrete-sin3d"
    model = LlavaQwenForCausalLM(hf, sd, dtype=torch.float16, device="cuda")
    g = torch.Generator().manual_seed(10)
    images = torch.randn(1, 2, 3, 384, 384, generator=g).half().cuda()
    video_dict = {"world_coords": ((torch.rand(1, 2, 384, 384, 3, generator=g) - 0.5) * 10).half().cuda(),
                  "objects": torch.cat([torch.zeros(5, 3), torch.ones(5, 3) * 3], 1)[None].half().cuda(),
                  "box_input": torch.Tensor([])}
    t = torch.randint(0, 300, (1, 16), generator=g)
    input_ids = torch.cat([t[:, :6], torch.tensor([[-200]]), t[:, 6:]], 1).cuda()
    out = model.generate(input_ids, images=images, modalities="video", do_sample=False, num_beams=1, max_new_tokens=5,
                         use_cache=True, video_dict=video_dict)
    assert out.shape == (1, 5) and out.dtype == torch.int64 and int(out.max()) < 320
    # grounding: labels mark the <ground> token position (config.ground_token_ids)
    model.config.ground_token_ids = [310, 311]
    labels = torch.full_like(input_ids, -100)
    labels[0, 12] = 310
    _, scores = model(input_ids, images=images, modalities="video", video_dict=video_dict, labels=labels,
                      use_object_proposals=True, box_labels=None)
    assert scores.shape == (6,) and torch.isfinite(scores.float()).all() and scores.float().abs().max() <= 1.001
    with pytest.raises(NotImplementedError):
        model.generate(input_ids, images=images, video_dict=video_dict, do_sample=True)


def test_siglip_image_processor_mirror(golden):
    """a7 through the reference's class name: PIL frames in, pixel_values out, equal to the reference's own output."""
    from PIL import Image
    from llava.model.multimodal_encoder.siglip_encoder import SigLipImageProcessor
    g = golden("imgproc")
    proc = SigLipImageProcessor(size=(96, 96), crop_size={"height": 96, "width": 96})
    out = proc.preprocess([Image.fromarray(f) for f in g["frames"]], return_tensors="pt")["pixel_values"]
    assert out.dtype == torch.float32 and tuple(out.shape) == (2, 3, 96, 96)
    assert np.array_equal(out.cpu().numpy(), g["pixel_values"])
    # a frame of another size goes through the same PIL bicubic call as the reference's transform chain
    full = SigLipImageProcessor()
    pv = full.preprocess(Image.fromarray(g["small"]))["pixel_values"]
    assert np.array_equal(pv[0, :, ::16, ::16].cpu().numpy(), g["small_pixel_values"])


def test_video_processor_on_scene_files(golden, tmp_path):
    """a4-a6 end to end through the mirror's VideoProcessor on files the test writes (16-bit depth PNGs, pose txt, JPEG frames):
    calculate_world_coords against the REFERENCE's own output on the same files (tests/golden/world_coords.npz: u16 -> f32,
    f64 axis_align @ pose, unproject), preprocess(center_crop): `boundry` = the reference's expression over the FULL-resolution
    coordinates (video_utils.py:268-273), world_coords = the restated cv2.INTER_NEAREST + centre-crop index map
    applied to them (parity unpinned for the index rule itself: cv2 is absent, oracle/v3d_oracle.py:76), images resized and
    cropped as the reference's PIL calls do."""
    import llava.video_utils as vu
    from scene_files import write_frames
    g = golden("world_coords")
    V, H, W = g["depth"].shape
    rgb = np.random.default_rng(3).integers(0, 256, size=(V, H, W, 3), dtype=np.uint8)
    files = write_frames(str(tmp_path / "posed_images" / "scene0000_00"), g["depth"], g["poses"], rgb=rgb, ext=".jpg")
    vp = object.__new__(vu.VideoProcessor)
    vp.video_folder = str(tmp_path)
    vp.frame_sampling_strategy = "uniform"
    vid = "scannet/scene0000_00"
    vp.scene = {vid: {"axis_align_matrix": g["axis_align"].tolist(), "depth_cam2img": g["cam2img"].tolist(),
                      "images": [{"img_path": "posed_images/scene0000_00/" + f.split("/")[-1]} for f in files]}}
    vp.scan2obj = {vid: [[0.0, 0.1, 0.2, 1.0, 1.0, 1.0], [1.0, 1.0, 0.5, 0.4, 0.6, 0.8]]}
    wc = vp.calculate_world_coords(vid, files)["world_coords"]
    assert wc.is_cuda and tuple(wc.shape) == (V, H, W, 3)
    np.testing.assert_allclose(wc.cpu().numpy(), g["world"], rtol=2e-6, atol=2e-6)
    crop = 24                                                   # new_width = int(64 * 24/48) = 32, left = 4
    proc = types.SimpleNamespace(crop_size={"height": crop, "width": crop})
    out = vp.preprocess(vid, proc, force_sample=True, frames_upbound=V)
    np.testing.assert_allclose(out["boundry"].numpy(), g["boundry"], rtol=2e-6, atol=2e-6)
    assert out["video_size"] == V and tuple(out["objects"].shape) == (2, 6)
    want = O.resize_crop_coords(g["world"], crop)
    np.testing.assert_allclose(out["world_coords"].cpu().numpy(), want, rtol=2e-6, atol=2e-6)
    # images: ONE uint8 device tensor [V, crop, crop, 3] = the reference's list of PIL crops, byte for byte (r03: Pillow's bicubic
    # resize runs on the device)
    assert out["images"].is_cuda and out["images"].dtype == torch.uint8 and tuple(out["images"].shape) == (V, crop, crop, 3)
    from PIL import Image
    for v in range(V):
        with Image.open(files[v]) as im:
            ref = im.convert("RGB").resize((32, crop)).crop((4, 0, 4 + crop, crop))
        assert np.array_equal(out["images"][v].cpu().numpy(), np.asarray(ref))
    # strategy "resize" (video_utils.py:293-296): frames to crop x crop, coordinate maps to 384 x 384 by the nearest rule on both axes
    rs = vp.preprocess(vid, proc, force_sample=True, frames_upbound=V, strategy="resize")
    rr = np.minimum(np.floor(np.arange(384) * (H / 384)).astype(int), H - 1)
    cc = np.minimum(np.floor(np.arange(384) * (W / 384)).astype(int), W - 1)
    np.testing.assert_allclose(rs["world_coords"].cpu().numpy(), g["world"][:, rr][:, :, cc], rtol=2e-6, atol=2e-6)
    for v in range(V):
        with Image.open(files[v]) as im:
            assert np.array_equal(rs["images"][v].cpu().numpy(), np.asarray(im.convert("RGB").resize((crop, crop))))
    # the "norm" sampling strategies (calculate_world_coords(do_normalize=True), :232-236): every coordinate clamped to the scene's box
    vp.frame_sampling_strategy = "uniform-norm"
    lo, hi = torch.tensor([-0.5, -1.0, 0.2]), torch.tensor([1.5, 0.75, 1.0])
    vp.pc_min, vp.pc_max = {"scene0000_00": lo}, {"scene0000_00": hi}
    nm = vp.preprocess(vid, proc, force_sample=True, frames_upbound=V)
    np.testing.assert_allclose(nm["world_coords"].cpu().numpy(), np.clip(want, lo.numpy(), hi.numpy()), rtol=2e-6, atol=2e-6)
    clamped = np.clip(g["world"], lo.numpy(), hi.numpy()).reshape(-1, 3)
    np.testing.assert_allclose(nm["boundry"].numpy(), np.stack([clamped.min(0), clamped.max(0)], 1).reshape(-1), rtol=2e-6, atol=2e-6)
    wcn = vp.calculate_world_coords(vid, files, do_normalize=True)["world_coords"]
    np.testing.assert_allclose(wcn.cpu().numpy(), np.clip(g["world"], lo.numpy(), hi.numpy()), rtol=2e-6, atol=2e-6)
    vp.frame_sampling_strategy = "uniform"
    # the asynchronous loader (worker processes writing a pinned shared-memory block, and its inline form) delivers the same arrays
    # as the one-shot load, and np.loadtxt's pose values
    from v3d.pipeline import AsyncSceneLoader
    raw = vp.load_raw(vid, files)
    vp.sample_frame_files = lambda video_id, force_sample=False, frames_upbound=0: files
    from v3d import frame_io
    for workers in (2, 0):
        # (a library caller hands in a pool made before the GPU was touched; this test process has long initialised it, and forks
        # its two PIL-only workers anyway - what AsyncSceneLoader itself no longer does on its own, see the warning test below)
        pool = frame_io.make_pool(workers) if workers else None
        loader = AsyncSceneLoader([vid, vid, vid], lambda v: vp.describe_scene(v, True, V), workers=workers, ahead=1, keep=2, pool=pool)
        try:
            for j in range(3):
                got, _ = loader.get(j)
                assert got["frames"].is_pinned() or not loader.blocks[0].pinned
                for k in ("depth", "frames", "pose", "K"):
                    assert torch.equal(raw[k], got[k]), (workers, j, k)
                ev = torch.cuda.current_stream().record_event()
                got["_done"](ev)
            assert len(loader.blocks) == 1                                   # the three questions shared one decode
            assert set(loader.stage_seconds) == {"depth_png", "pose_txt", "rgb_decode"}
        finally:
            loader.close()
            if pool is not None:
                pool.shutdown(wait=True, cancel_futures=True)
    with pytest.warns(UserWarning, match="already initialised"):          # no pool + live GPU: decode inline rather than fork
        loader = AsyncSceneLoader([vid], lambda v: vp.describe_scene(v, True, V), workers=2)
    assert loader.pool is None
    got, _ = loader.get(0)
    assert torch.equal(raw["frames"], got["frames"])
    got["_done"](None)
    loader.close()
    want_pose = np.stack([g["axis_align"].astype(np.float64) @ np.loadtxt(f.replace("jpg", "txt")) for f in files]).astype(np.float32)
    assert np.array_equal(raw["pose"].numpy(), want_pose)


def test_pipeline_device_inputs_follow_the_depth_maps_aspect(tmp_path):
    """ScanNet's colour frames (1296 x 968) and depth maps (640 x 480) differ in aspect; the reference sizes BOTH
    resizes from world_coords, i.e. from the depth map (video_utils.py:264, 298-299): colour -> 512 x 384, crop at column 64 - not the
    514 / 65 the colour frame's own aspect gives.  v3d.pipeline.ScenePipeline.device_inputs (the default path of every eval runner and
    of bench.py) must equal VideoProcessor.preprocess + the image processor, which in turn equals the reference's PIL calls; and a
    'norm' frame-sampling strategy's clamp (video_utils.py:232-236) must reach the pipelined path too."""
    import llava.video_utils as vu
    from PIL import Image
    from llava.model.multimodal_encoder.siglip_encoder import SigLipImageProcessor
    from scene_files import write_frames
    from v3d.pipeline import AsyncSceneLoader, ScenePipeline, SceneSample
    rng = np.random.default_rng(5)
    V, Hd, Wd, Hc, Wc, crop = 3, 48, 64, 120, 166, 96          # by the colour aspect 166 * 96 / 120 -> 132 (left 18); by the depth's 128 (left 16)
    assert int(Wc * (crop / Hc)) != int(Wd * (crop / Hd))
    depth = rng.integers(500, 4000, size=(V, Hd, Wd)).astype(np.uint16)
    poses = np.tile(np.eye(4), (V, 1, 1))
    poses[:, :3, 3] = rng.normal(size=(V, 3))
    folder = str(tmp_path / "posed_images" / "scene0000_00")
    files = write_frames(folder, depth, poses, rgb=None, ext=".jpg")
    yy, xx = np.mgrid[0:Hc, 0:Wc]
    for v, f in enumerate(files):                               # the colour stream at ITS size
        rgb = np.stack([(xx * (3 + c) + yy * (5 - c) + 40 * v) % 256 for c in range(3)], -1).astype(np.uint8)
        Image.fromarray(rgb).save(f, quality=95)
    vid = "scannet/scene0000_00"
    vp = object.__new__(vu.VideoProcessor)
    vp.video_folder, vp.frame_sampling_strategy = str(tmp_path), "uniform"
    vp.scene = {vid: {"axis_align_matrix": np.eye(4).tolist(), "depth_cam2img": [[57.8, 0, 31.5, 0], [0, 57.8, 23.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
                      "images": [{"img_path": "posed_images/scene0000_00/" + f.split("/")[-1]} for f in files]}}
    vp.scan2obj = {vid: [[0.0, 0.1, 0.2, 1.0, 1.0, 1.0]]}
    vp.sample_frame_files = lambda video_id, force_sample=False, frames_upbound=0: files
    proc = SigLipImageProcessor(size=(crop, crop), crop_size={"height": crop, "width": crop})
    eng = types.SimpleNamespace(dtype=torch.float16, device="cuda", ctx=None, ws=None, MAX_GROUP=32, new_context=lambda: None, new_group=lambda g: None,
                                new_prefill_workspace=lambda: None)
    pipe = ScenePipeline(eng, 1, crop=crop, image_mean=proc.image_mean, image_std=proc.image_std, rescale=proc.rescale_factor)
    for strategy, lo, hi in (("uniform", None, None), ("uniform-norm", [-0.5, -1.0, 0.2], [1.5, 0.75, 1.0])):
        vp.frame_sampling_strategy = strategy
        if lo is not None:
            vp.pc_min, vp.pc_max = {"scene0000_00": torch.tensor(lo)}, {"scene0000_00": torch.tensor(hi)}
        ref = vp.process_3d_video(vid, proc, force_sample=True, frames_upbound=V)
        # (the mirror's own preprocess against the reference's PIL calls at the depth-derived width)
        new_w = int(Wd * (crop / Hd))
        left = (new_w - crop) // 2
        pre = vp.preprocess(vid, proc, force_sample=True, frames_upbound=V)
        for v in range(V):
            with Image.open(files[v]) as im:
                want = np.asarray(im.convert("RGB").resize((new_w, crop)).crop((left, 0, left + crop, crop)))
            assert np.array_equal(pre["images"][v].cpu().numpy(), want)
        loader = AsyncSceneLoader([vid], lambda v: vp.describe_scene(v, True, V), workers=0)
        try:
            raw, _ = loader.get(0)
            images, coords = pipe.device_inputs(SceneSample(input_ids=torch.tensor([-200]), raw=raw, key=None))
            torch.cuda.synchronize()
        finally:
            loader.close()
        assert torch.equal(images, ref["images"].to(torch.float16)), strategy          # model_scanqa.py:163: .half() of the f32 pixel_values
        assert torch.equal(coords, ref["world_coords"].to(torch.float16)), strategy
        if lo is not None:
            c = coords.float().cpu()
            assert bool((c >= torch.tensor(lo).half().float()).all()) and bool((c <= torch.tensor(hi).half().float()).all())
            assert bool((c == torch.tensor(lo).half().float()).any())                  # (the clamp bit somewhere)
