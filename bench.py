#!/usr/bin/env python3
"""bench.py - scenes/sec of the position-aware video->LLM forward path (ScanQA @ 32 frames) on MI355X.

One step = one (scene, question) through the whole hot path with inputs resident in HBM:
  depth u16 + pose -> world coords at the resized/cropped pixels (K1+K2) -> 27x27 patch mean + voxel ids
  (K3+K4) -> SigLIP-so400m tower, 26 layers (K10-K11) -> mlp2x_gelu projector (K12) -> bilinear 27->14 pool
  + 3-D sinusoid PE add + newline rows, written into inputs_embeds (K5-K9) -> Qwen2-7B prefill over
  6720 visual + 74 text tokens (K13-K18) -> 17 greedy tokens = 16 decode passes over the weights.
Random-init weights at the true widths, synthetic inputs, bf16, a DIFFERENT synthetic scene per step (all resident
in HBM before the timed region).  Every scene runs the complete path; scenes are prefilled one by one and decoded in groups
(--decode-group, default 16) that share each pass over the weights.
The default N = 1 line also carries `roofline` (the dominant kernel: the Qwen2 gate/up GEMM, MFMA), `roofline_north_star` (3D-PE +
fusion kernel, HBM), `roofline_attention`, `fp8_config3` (the same step with e4m3 LLM weights, BASELINE configs[3]),
`train_config4` (one training sample end to end on one GPU: tower + projector + Qwen2 with labels, backward, AdamW; v3d/train.py),
`cached_questions` (scene-level reuse, SURVEY 8 f1: further questions about an already prefilled scene), `ground_config2`
(the ScanRefer / Multi3DRefer grounding forward at 32 frames / 50 proposals, BASELINE configs[2], one GPU) and `cpu_baseline`.

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / WORLD_SIZE in the
environment), or started plainly - then THIS process, before it touches the GPU, starts exactly that command as a child (one rank
per GPU, rendezvous on 127.0.0.1), relays rank 0's line and exits with the child's code (the reference's driver starts its own workers
too: model_scanqa.py:242-247, `ray.init(); eval_model.remote(questions[i::n_gpu])`).

Prints ONE JSON line (rank 0).  Scenes shard data-parallel: every rank runs its own scenes, no
data-path collective; one RCCL gather of the generated token ids at the end (eval collation).
V3D_BENCH_DRY=1: no GPU at all - ranks over gloo run the sharding, the record gather and the max-over-ranks timing around a stand-in
for the scene pipeline (CPU test of the launcher and the N > 1 plumbing; the line says so and is not a measurement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "video-3d-llm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FRAMES = 32
TEXT_PRE, TEXT_POST = 14, 60          # 14 system + 60 question ids around one <image> token
NEW_TOKENS = 17                       # 1 from the prefill logits + 16 decode passes over the weights
DECODE_STEPS = NEW_TOKENS
IMAGE_TOKEN_INDEX = -200


def synth_inputs(dev, dtype, seed, frames=FRAMES, text_post=TEXT_POST):
    g = torch.Generator(device=dev).manual_seed(seed)
    F_ = frames
    depth = torch.randint(400, 5000, (F_, 480, 640), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    K = torch.zeros(F_, 4, 4, device=dev)
    K[:, 0, 0] = K[:, 1, 1] = 577.87
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2], K[:, 3, 3] = 319.5, 239.5, 1.0, 1.0
    ang = torch.rand(F_, generator=g, device=dev) * 6.2831853
    P = torch.zeros(F_, 4, 4, device=dev)
    P[:, 0, 0], P[:, 0, 1], P[:, 1, 0], P[:, 1, 1] = torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)
    P[:, 2, 2] = P[:, 3, 3] = 1.0
    P[:, :3, 3] = torch.randn(F_, 3, generator=g, device=dev) * 1.5
    frames = torch.randint(0, 256, (F_, 384, 384, 3), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)   # RGB crops
    text = torch.randint(0, 151000, (TEXT_PRE + text_post,), generator=g, device=dev)
    input_ids = torch.cat([text[:TEXT_PRE], torch.tensor([IMAGE_TOKEN_INDEX], device=dev), text[TEXT_PRE:]]).cpu()
    return dict(depth=depth, K=K, pose=P, P=P, frames=frames, input_ids=input_ids)


class Stamp:
    """HIP event pair on the launch stream around selected kernels, inside the timed region."""

    def __init__(self):
        self.pairs = []

    def __call__(self, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.pairs.append((e0, e1))
        return r

    def mean_us(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs) * 1e3 / max(1, len(self.pairs))


def samples_of(scenes, n, SceneSample):
    """n (scene, question) samples cycling over the resident synthetic scenes; no scene key: every step uploads nothing (inputs are
    resident) but runs the complete device path, geometry and image preprocessing included."""
    for i in range(n):
        sc = scenes[i % len(scenes)]
        yield SceneSample(input_ids=sc["input_ids"], raw=sc, key=None)


def cpu_baseline(threads):
    """The oracle (CPU restatement) timed on this box's host cores on a bounded sample: the 3-D position path
    at full 32-frame shape, plus ONE SigLIP layer and ONE Qwen2 layer at full width, extrapolated x26 / x28."""
    import numpy as np
    from oracle import llm_oracle as L
    from oracle import v3d_oracle as O
    torch.set_num_threads(threads)
    rng = np.random.default_rng(0)
    t = {}
    depth = rng.integers(400, 5000, size=(FRAMES, 480, 640)).astype(np.float32)
    K = np.tile(np.array([[577.87, 0, 319.5, 0], [0, 577.87, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32), (FRAMES, 1, 1))
    P = np.tile(np.eye(4, dtype=np.float32), (FRAMES, 1, 1))
    t0 = time.perf_counter(); world = O.unproject(K, P, depth); t["unproject"] = time.perf_counter() - t0
    t0 = time.perf_counter(); wc = O.resize_crop_coords(world); t["resize_crop"] = time.perf_counter() - t0
    t0 = time.perf_counter(); avg = O.average_coordinate_in_patch(wc, "f32"); vox = O.discrete_coords(avg, "f32"); t["pool_voxel"] = time.perf_counter() - t0
    feat = rng.standard_normal((FRAMES, 729, 3584), dtype=np.float32)
    t0 = time.perf_counter(); pooled = O.get_2dpool_bilinear(feat, "f32"); t["bilinear_pool"] = time.perf_counter() - t0
    t0 = time.perf_counter(); pe = O.sin3d_pe(vox.reshape(FRAMES, -1, 3), 3584, "f32"); t["sin3d_pe"] = time.perf_counter() - t0
    t0 = time.perf_counter(); seq = O.add_token_per_grid(O.add_pe(pooled, pe, "f32"), np.zeros(3584, np.float32)); t["add_newline"] = time.perf_counter() - t0
    geom = sum(t.values())
    g = torch.Generator().manual_seed(0)

    def rn(*s):
        return torch.randn(*s, generator=g) * 0.02
    S = TEXT_PRE + FRAMES * 210 + TEXT_POST
    w = {"input_layernorm.weight": torch.ones(3584), "post_attention_layernorm.weight": torch.ones(3584),
         "self_attn.q_proj.weight": rn(3584, 3584), "self_attn.q_proj.bias": rn(3584),
         "self_attn.k_proj.weight": rn(512, 3584), "self_attn.k_proj.bias": rn(512),
         "self_attn.v_proj.weight": rn(512, 3584), "self_attn.v_proj.bias": rn(512),
         "self_attn.o_proj.weight": rn(3584, 3584), "mlp.gate_proj.weight": rn(18944, 3584),
         "mlp.up_proj.weight": rn(18944, 3584), "mlp.down_proj.weight": rn(3584, 18944)}
    x = torch.randn(1, S, 3584, generator=g)
    t0 = time.perf_counter(); L.qwen2_layer(x, w, "", 28, 4, torch.arange(S), 1e6, 1e-6); llm_layer = time.perf_counter() - t0
    wv = {}
    for n_ in ("q_proj", "k_proj", "v_proj", "out_proj"):
        wv[f"self_attn.{n_}.weight"], wv[f"self_attn.{n_}.bias"] = rn(1152, 1152), rn(1152)
    for n_ in ("layer_norm1", "layer_norm2"):
        wv[n_ + ".weight"], wv[n_ + ".bias"] = torch.ones(1152), torch.zeros(1152)
    wv["mlp.fc1.weight"], wv["mlp.fc1.bias"], wv["mlp.fc2.weight"], wv["mlp.fc2.bias"] = rn(4304, 1152), rn(4304), rn(1152, 4304), rn(1152)
    xv = torch.randn(FRAMES, 729, 1152, generator=g)
    t0 = time.perf_counter(); L.siglip_layer(xv, wv, "", 16); vit_layer = time.perf_counter() - t0
    total = geom + 26 * vit_layer + 28 * llm_layer
    return {"value": 1.0 / total, "unit": "scenes/s", "cores": threads, "kind": "port",
            "sample": ("oracle f32 on CPU: 3-D position path (unproject, resize/crop, patch mean+voxel, bilinear pool, sin3d PE, "
                       "add+newline) at full 32-frame shape = %.2f s; one SigLIP layer (%.2f s) and one Qwen2-7B layer at S=%d "
                       "(%.2f s) at full width, extrapolated x26 / x28; projector, LM head and decode not included" % (geom, vit_layer, S, llm_layer)),
            "seconds_measured": geom + vit_layer + llm_layer}


def usable_cores():
    """Host cores this process can actually run on: sched_getaffinity, limited by the cgroup CPU quota (v2 cpu.max / v1 cfs quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    why = "scheduler affinity: %d of %d host cores" % (n, os.cpu_count() or n)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n = max(1, int(quota))
        why += "; cgroup CPU quota %.1f cores" % quota
    return n, why


def kernel_source_sha(*names):
    """sha1 over the kernel sources a PMC figure belongs to: profiles/pmc_summary.json carries the value it was measured on,
    so that a figure taken on an older kernel is reported as stale instead of passing silently."""
    import hashlib
    h = hashlib.sha1()
    for n in names:
        with open(os.path.join(ROOT, "video-3d-llm_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measure_cached_questions(eng, ops, scene_inp, dev, n_groups, group=16, q_len=TEXT_POST):
    """one scene is prefilled once (geometry, ViT, projector, fusion, the decoder over [system | user | <image>]);
    then n_groups x `group` DIFFERENT questions about it are answered (question rows batched through the decoder over the cached
    prefix, then one decode group per batch) - the reference recomputes the whole prompt per question (model_scanqa.py:130-185).
    Returns (questions/s over the answering alone, ms of the one-off scene prefill)."""
    dt = eng.dtype
    g = torch.Generator(device=dev).manual_seed(4242)
    prefix = scene_inp["input_ids"][: TEXT_PRE + 1]
    qs = [[torch.randint(0, 151000, (q_len,), generator=g, device=dev) for _ in range(group)] for _ in range(n_groups)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    coords = ops.unproject_sampled(scene_inp["depth"], scene_inp["K"], scene_inp["P"], 384, dt)
    images = ops.preprocess_rgb(scene_inp["frames"], dt)
    eng.prefill_scene(prefix, images, coords)
    torch.cuda.synchronize()
    t_scene = time.perf_counter() - t0
    eng.answer_group(qs[0], max_new_tokens=NEW_TOKENS)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for q in qs:
        eng.answer_group(q, max_new_tokens=NEW_TOKENS)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    return n_groups * group / dt_s, t_scene * 1e3


def measure_decode_step(eng, ops, scenes, dev, steps=8):
    """The decode step (one new token per row: four weight-streaming linears per layer + the LM head, rotary / append, split-KV attention)
    against its HBM roofline, in the two forms the product runs: 16 rows = 16 SCENES decoding together, each over its own K/V cache (the
    headline's decode groups), and 32 rows = 32 QUESTIONS about one scene, the prefix K/V read once for all of them (answer batches).
    Algorithmic bytes per step = the decoder's linear weights + the LM head + the K/V rows the step attends over, each once."""
    l = eng.cfg.llm
    wbytes = sum(L[k].numel() * L[k].element_size() for L in eng.l_layers for k in ("wqkv", "wo", "wgu", "wd")) + eng.l_head.numel() * eng.l_head.element_size()
    kv_row = 2 * l.kv_heads * eng.hd * 2 * l.layers                      # bytes of one position's K and V over all layers

    def timed(fn):
        fn(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            fn(1 + i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    out = {"what": "one decode step against its HBM roofline (weights + attended K/V rows once per step; peak 8 TB/s): rows16 = 16 scenes over their own "
                   "caches (the headline's decode groups), rows32_shared_prefix = 32 questions about one prefilled scene (answer batches)"}
    keep = eng.ctx
    # ---- 32 questions about the scene in eng.ctx (prefilled by measure_cached_questions): their caches hold the question rows
    P = keep.prefix_len
    st = eng._answer_state(32)
    ctxs = st.ctxs[:32]
    base = [P + TEXT_POST] * 32
    ms = timed(lambda i: eng.decode_forward_rows(st.rows, ctxs, [b + i for b in base], shared_prefix=P, prefix_kv=keep.kv))
    nbytes = wbytes + (P + 32 * (TEXT_POST + steps // 2)) * kv_row
    out["rows32_shared_prefix"] = {"ms_per_step": ms, "bytes": nbytes, "achieved_TBps": nbytes / ms / 1e9, "frac_of_8TBps": nbytes / ms / 1e9 / 8.0}
    # ---- 16 scenes, each prefilled into its own context
    pool = [eng.new_context() for _ in range(16)]
    grp = eng.new_group(16)
    S = 0
    try:
        for i, c in enumerate(pool):
            inp = scenes[i % len(scenes)]
            eng.use(c)
            feats = eng.encode_images(ops.preprocess_rgb(inp["frames"], eng.dtype))
            ids = eng.voxel_ids(ops.unproject_sampled(inp["depth"], inp["K"], inp["P"], 384, eng.dtype).to(eng.dtype))
            x = eng.build_inputs_embeds(inp["input_ids"], feats, ids)
            S = x.shape[0]
            eng.llm_forward(x, 0, last_rows=[S - 1])
    finally:
        eng.use(keep)
    ms = timed(lambda i: eng.decode_forward_rows(grp, pool, [S + i] * 16))
    nbytes = wbytes + 16 * (S + steps // 2) * kv_row
    out["rows16"] = {"ms_per_step": ms, "bytes": nbytes, "achieved_TBps": nbytes / ms / 1e9, "frac_of_8TBps": nbytes / ms / 1e9 / 8.0}
    del pool, grp
    torch.cuda.empty_cache()
    return out


def measure_grounding(eng, ops, scenes, dev, steps):
    """BASELINE configs[2] on one GPU: the ScanRefer / Multi3DRefer forward (model_scanrefer.py:165-173): geometry, ViT, projector,
    object-proposal patch masks + masked means for 50 proposals (extract_pred_box.py:30), fusion, Qwen2 prefill, infonce head."""
    dt = eng.dtype
    g = torch.Generator(device=dev).manual_seed(777)
    boxes = torch.cat([(torch.rand(50, 3, generator=g, device=dev) - 0.5) * torch.tensor([8.0, 8.0, 2.0], device=dev),
                       torch.rand(50, 3, generator=g, device=dev) * 2.0 + 0.3], 1)
    gpos = TEXT_PRE + 1 + 40                              # the <ground> label token inside the question

    def one(inp):
        coords = ops.unproject_sampled(inp["depth"], inp["K"], inp["P"], 384, dt)
        images = ops.preprocess_rgb(inp["frames"], dt)
        return eng.ground_scores(inp["input_ids"], gpos, images, coords, boxes)

    one(scenes[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        sc = one(scenes[i % len(scenes)])
    torch.cuda.synchronize()
    assert sc.shape == (51,)
    return steps / (time.perf_counter() - t0)


def write_eval_dataset(root, n_scenes, frames=FRAMES):
    """A synthetic scene set in the dataset's own on-disk form (what scripts/3d/preprocessing/generate_image_scannet.py exports and
    llava/video_utils.py:90-105, 196-238, 285-290 read): per frame a 1296 x 968 colour JPEG, a 640 x 480 16-bit depth PNG and a 4 x 4 pose
    text file; EmbodiedScan-style scene index pickles; box JSONs.  Smooth content with texture (camera-like compressibility), not noise."""
    import pickle
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:968, 0:1296].astype(np.float32)
    dy, dx = np.mgrid[0:480, 0:640].astype(np.float32)
    data_list = []
    for s_ in range(n_scenes):
        sid, folder = f"scannet/scene{s_:04d}_00", os.path.join(root, "posed_images", f"scene{s_:04d}_00")
        os.makedirs(folder, exist_ok=True)
        files = []
        for v in range(frames):
            ph = 0.37 * v + s_
            base = os.path.join(folder, f"{v * 10:05d}")
            rgb = np.stack([127 + 90 * np.sin(xx / (41 + 7 * c) + ph) * np.cos(yy / (29 + 5 * c) - ph) for c in range(3)], -1)
            rgb = np.clip(rgb + rng.normal(0, 6, rgb.shape).astype(np.float32), 0, 255).astype(np.uint8)
            Image.fromarray(rgb).save(base + ".jpg", quality=90)
            depth = 2200 + 1500 * np.sin(dx / 97 + ph) * np.cos(dy / 71) + rng.normal(0, 3, dx.shape)
            depth[(dx + dy + 13 * v) % 97 < 3] = 0                                   # holes, as sensor depth has
            Image.fromarray(np.clip(depth, 0, 65535).astype(np.uint16)).save(base + ".png", compress_level=1)
            pose = np.eye(4)
            pose[:3, :3] = [[np.cos(ph), -np.sin(ph), 0], [np.sin(ph), np.cos(ph), 0], [0, 0, 1]]
            pose[:3, 3] = rng.normal(0, 1.5, 3)
            np.savetxt(base + ".txt", pose)
            files.append(os.path.relpath(base + ".jpg", root))
        data_list.append({"sample_idx": sid, "axis_align_matrix": np.eye(4).tolist(),
                          "depth_cam2img": [[577.87, 0, 319.5, 0], [0, 577.87, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
                          "images": [{"img_path": f} for f in files]})
    os.makedirs(os.path.join(root, "embodiedscan"), exist_ok=True)
    os.makedirs(os.path.join(root, "metadata"), exist_ok=True)
    for split in ("train", "val", "test"):
        with open(os.path.join(root, "embodiedscan", f"embodiedscan_infos_{split}.pkl"), "wb") as f:
            pickle.dump({"data_list": data_list if split == "val" else []}, f)
    for name in ("scannet_train_gt_box.json", "scannet_val_pred_box.json"):
        with open(os.path.join(root, "metadata", name), "w") as f:
            json.dump({d["sample_idx"]: [[0, 0, 0, 1, 1, 1]] for d in data_list}, f)
    return [d["sample_idx"] for d in data_list]


def bench_tokenizer():
    """A word-level stand-in tokenizer with the ChatML specials (no Qwen2 tokenizer ships here; the ids only have to be valid)."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    names = {300: "system", 301: "user", 302: "assistant", 303: "\n", 304: "You", 305: "are", 306: "a", 307: "helpful",
             308: "assistant.", 309: "<|im_start|>", 310: "<|im_end|>"}
    vocab = {names.get(i, f"t{i}"): i for i in range(2048)}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="t0"))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split("\n", "isolated"), pre_tokenizers.Split(" ", "removed")])
    return PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|im_end|>", pad_token="t0", unk_token="t0",
                                   additional_special_tokens=["<|im_start|>", "<|im_end|>"])


def measure_eval_runner(eng, dev, n_questions, pool, n_workers, n_scenes=8, reuse_questions_per_scene=32):
    """The PRODUCT's eval loop end to end on files: v3d.eval_scanqa.model_answer_fn (pipelined: asynchronous loader -> upload -> device
    geometry / Pillow-exact resize -> ViT -> ... -> grouped decode with the device-side stop test) over a synthetic on-disk scene set at the
    dataset's true sizes.  Two runs over the same files -> (eval_runner, reuse_runner):
      eval_runner   every question asks about ANOTHER scene than the previous eight (no scene reuse: the loader's worst case);
      reuse_runner  --reuse-scenes on the pipeline (r04): `reuse_questions_per_scene` consecutive questions per scene share ONE scene prefill
                    and are answered in batches of 32, the next scene's load / ViT / prefix prefill running beside them."""
    import contextlib
    import shutil
    import tempfile
    import types
    from llava.model.multimodal_encoder.siglip_encoder import SigLipImageProcessor
    from llava.video_utils import VideoProcessor
    from v3d import eval_scanqa as E
    root = tempfile.mkdtemp(prefix="v3d_bench_scenes_")
    try:
        t0 = time.perf_counter()
        sids = write_eval_dataset(root, n_scenes)
        t_write = time.perf_counter() - t0
        tok = bench_tokenizer()
        with contextlib.redirect_stdout(sys.stderr):
            vp = VideoProcessor(video_folder=root, annotation_dir=os.path.join(root, "embodiedscan"), metadata_dir=os.path.join(root, "metadata"))
        words = [f"t{(7 * k) % 290 + 1}" for k in range(64)]

        def question(i, n_words, scene, salt=0):
            text = " ".join(words[(salt + j) % 64] for j in range(n_words))
            return {"id": f"q{i}", "video": sids[scene % n_scenes],
                    "conversations": [{"from": "human", "value": "<image>\n" + text}, {"from": "gpt", "value": "t42"}],
                    "metadata": {"dataset": "scanqa", "question_type": "what", "answers": ["t42"]}}
        n_words = 40
        while len(E.build_prompt_ids(question(0, n_words, 0), tok)[0]) - 1 < TEXT_PRE + TEXT_POST and n_words < 64:
            n_words += 1
        qs = [question(i, n_words, i) for i in range(n_questions)]
        S = len(E.build_prompt_ids(qs[0], tok)[0]) - 1 + FRAMES * 210
        model = types.SimpleNamespace(engine=eng, device=dev, dtype=eng.dtype, _eos=lambda: None)
        stats = {}
        fn = E.model_answer_fn(model, tok, SigLipImageProcessor(), vp, "bench", FRAMES, NEW_TOKENS, pipeline=True, group_size=16, stats=stats,
                               pool=pool, workers=n_workers)
        fn(qs[: min(4, n_questions)])                     # warm-up: page cache, kernels' first launches, pinned pools
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recs = fn(qs)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        assert len(recs) == n_questions and all(r["sample_id"] == q["id"] for r, q in zip(recs, qs))
        per_q = {k: v / n_questions * 1e3 for k, v in stats["host_thread_seconds"].items()}
        plain = {"what": "the product's eval loop (v3d.eval_scanqa.model_answer_fn, pipelined) end to end on a synthetic ON-DISK scene set: per question "
                         "32 x (1296x968 JPEG + 640x480 16-bit PNG + pose txt) decoded by the asynchronous host loader, uploaded, back-projected, resized "
                         "(Pillow-exact, on the device; colour to 512 x 384 as the depth map's aspect dictates), then the same ViT -> projector -> fusion -> "
                         "Qwen2 prefill S=%d -> %d greedy tokens as the headline; every question about another scene than the eight before it (no reuse), "
                         "files in the page cache" % (S, NEW_TOKENS),
                 "value": n_questions / dt_s, "unit": "scenes/s", "ms_per_step": dt_s / n_questions * 1e3, "questions": n_questions, "seq_len": S,
                 "loader_processes": n_workers, "host_cores": os.cpu_count(),
                 "loader_core_ms_per_question": per_q, "gpu_waited_for_loader_ms_per_question": stats["loader_wait_seconds"] / n_questions * 1e3,
                 "upload_enqueue_ms_per_question": stats["upload_enqueue_seconds"] / n_questions * 1e3,
                 "upload_mb_per_question": (FRAMES * (1296 * 968 * 3 + 640 * 480 * 2)) / 1e6, "dataset_write_s": t_write}
        if os.environ.get("V3D_BENCH_SKIP_REUSE") == "1":      # (development: the eval_runner part alone, e.g. under a kernel trace)
            return plain, {"skipped": True}
        # ---- scene reuse on the pipeline, from the same files: consecutive questions per scene
        QPS = reuse_questions_per_scene
        n_sc = min(n_scenes, 8)      # (the first scene's prefill is exposed in any run: over 8 scenes it is an eighth of the prefills, over r04's 4 a quarter)
        rq = [question(1000 + sc * QPS + j, n_words, sc, salt=3 * j + sc) for sc in range(n_sc) for j in range(QPS)]
        rstats = {}
        rfn = E.model_answer_fn(model, tok, SigLipImageProcessor(), vp, "bench", FRAMES, NEW_TOKENS, reuse_scenes=True, pipeline=True, stats=rstats,
                                pool=pool, workers=n_workers)
        rfn(rq[:32])                                      # warm-up (the answer batches' buffers: 32 per-question caches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rrecs = rfn(rq)
        torch.cuda.synchronize()
        rdt = time.perf_counter() - t0
        assert len(rrecs) == len(rq) and all(r["sample_id"] == q["id"] for r, q in zip(rrecs, rq))
        ids0 = E.build_prompt_ids(rq[0], tok)[0]
        at = int((ids0 == IMAGE_TOKEN_INDEX).nonzero()[0])
        reuse = {"what": "--reuse-scenes ON the pipeline, end to end from the same files (v3d.pipeline.SceneReusePipeline): %d scenes x %d consecutive "
                         "questions; per scene ONE load + upload + ViT + prefix prefill (%d rows), its questions (%d rows + %d new tokens each) answered in "
                         "batches of 32 on another stream while the NEXT scene is loaded and prefilled; compare `cached_questions` (the answering alone, "
                         "scene already prefilled, inputs resident)" % (n_sc, QPS, at + FRAMES * 210, len(ids0) - 1 - at, NEW_TOKENS),
                 "value": len(rq) / rdt, "unit": "questions/s", "ms_per_question": rdt / len(rq) * 1e3, "questions": len(rq), "scenes": n_sc,
                 "questions_per_scene": QPS, "gpu_waited_for_loader_ms_per_scene": rstats["loader_wait_seconds"] / n_sc * 1e3}
        # ScanQA val holds about 66 questions per scene (4675 questions, 71 scenes): the same run at 64 per scene (two answer batches of 32 per prefill)
        rq64 = [question(5000 + sc * 64 + j, n_words, sc, salt=5 * j + sc) for sc in range(n_sc) for j in range(64)]
        t0 = time.perf_counter()
        rrecs = rfn(rq64)
        torch.cuda.synchronize()
        rdt = time.perf_counter() - t0
        assert len(rrecs) == len(rq64)
        reuse["at_64_questions_per_scene"] = {"value": len(rq64) / rdt, "unit": "questions/s", "questions": len(rq64), "scenes": n_sc}
        return plain, reuse
    finally:
        shutil.rmtree(root, ignore_errors=True)


def measure_train_step(dev, steps=2, answer_tokens=64):
    """BASELINE configs[4] on ONE GPU (no ZeRO exchange): one training sample end to end - SigLIP-so400m (26 layers, 32 frames) ->
    mlp2x_gelu -> pool + 3-D PE + newline rows spliced between the text rows -> Qwen2-7B with labels, backward of all of it, AdamW on every
    parameter (v3d/train.py: sample_forward_backward, AdamW); random-init bf16 weights, f32 master weights and moments."""
    from v3d import ops, train
    L, H, I, n_q, n_kv, hd, V = 28, 3584, 18944, 28, 4, 128, 152064
    Lv, Hv, Iv, heads, tokens, kpad = 26, 1152, 4304, 16, 729, 640
    width = (n_q + 2 * n_kv) * hd
    dt = torch.bfloat16

    def mk(*shape, s=1.0):
        return torch.empty(*shape, device=dev, dtype=dt).normal_(0.0, s)

    ones = lambda n_: torch.ones(n_, device=dev, dtype=dt)
    layers = [{"ln1": ones(H), "qkv": mk(width, H, s=H ** -0.5), "qkv_bias": mk(width, s=0.1), "o": mk(H, n_q * hd, s=H ** -0.5),
               "ln2": ones(H), "gate_up": mk(2 * I, H, s=H ** -0.5), "down": mk(H, I, s=I ** -0.5)} for _ in range(L)]
    llm = {"layers": layers, "norm": ones(H), "lm_head": mk(V, H, s=H ** -0.5)}

    def vit_layer():
        sd = {"ln1_w": ones(Hv), "ln1_b": mk(Hv, s=0.02), "ln2_w": ones(Hv), "ln2_b": mk(Hv, s=0.02), "o_w": mk(Hv, Hv, s=Hv ** -0.5), "o_b": mk(Hv, s=0.02),
              "fc1_w": mk(Iv, Hv, s=Hv ** -0.5), "fc1_b": mk(Iv, s=0.02), "fc2_w": mk(Hv, Iv, s=Iv ** -0.5), "fc2_b": mk(Hv, s=0.02)}
        for n_ in ("q", "k", "v"):
            sd[n_ + "_w"], sd[n_ + "_b"] = mk(Hv, Hv, s=Hv ** -0.5), mk(Hv, s=0.02)
        return train.siglip_pad_layer(sd)

    patch_w = mk(Hv, kpad, s=588 ** -0.5)
    patch_w[:, 588:] = 0
    vision = {"patch_w": patch_w, "patch_b": mk(Hv, s=0.02), "pos": mk(tokens, Hv, s=0.02), "layers": [vit_layer() for _ in range(Lv)]}
    params = {"vision": vision, "projector": {"w1": mk(H, Hv, s=Hv ** -0.5), "b1": mk(H, s=0.02), "w2": mk(H, H, s=H ** -0.5), "b2": mk(H, s=0.02)},
              "newline": mk(H, s=0.02), "embed": mk(V, H, s=0.02), "llm": llm}
    rope = train.RopeTables(hd, 8192, 1e6, dt, dev)
    table = ops.Sin3DTable(H, 512, dt, dev)
    opt = train.AdamW(params, lr=1e-5)
    patches = mk(FRAMES * tokens, kpad)
    patches[:, 588:] = 0
    ids = torch.randint(0, 512, (FRAMES, 14, 14, 3), device=dev, dtype=torch.int32)
    pre_ids, post_ids = torch.randint(0, V, (TEXT_PRE,), device=dev), torch.randint(0, V, (TEXT_POST,), device=dev)
    S = TEXT_PRE + FRAMES * 210 + TEXT_POST
    labels = torch.full((S,), -100, dtype=torch.int64, device=dev)
    labels[S - answer_tokens:] = torch.randint(0, V, (answer_tokens,), device=dev)
    # the same step with activation re-computation per layer (the reference trains with gradient checkpointing, train_multi.sh:72): memory and time
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    rc_ms = []
    for i in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, grads = train.sample_forward_backward(params, patches, ids, table, pre_ids, post_ids, labels, rope, FRAMES, n_q, n_kv, hd, recompute=True)
        torch.cuda.synchronize()
        rc_ms.append((time.perf_counter() - t0) * 1e3)
        del grads
    rc_peak = torch.cuda.max_memory_allocated() / 2 ** 30
    torch.cuda.reset_peak_memory_stats()
    fb, ad, first_loss = [], [], None
    for i in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, grads = train.sample_forward_backward(params, patches, ids, table, pre_ids, post_ids, labels, rope, FRAMES, n_q, n_kv, hd)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        opt.step(params, grads)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if first_loss is None:
            first_loss = float(loss)
        del grads
        if i:
            fb.append(t1 - t0)
            ad.append(t2 - t1)
    n_llm = sum(p.numel() for l in layers for p in l.values()) + H + V * H
    n_vit = Lv * (4 * Hv * Hv + 2 * Hv * Iv) + Hv * 588
    n_proj = Hv * H + H * H
    ms = (sum(fb) + sum(ad)) / len(fb) * 1e3
    T = FRAMES * tokens
    flops = (6.0 * S * n_llm + 3.5 * 2.0 * S * S * hd * n_q * L              # linears 6 N S; causal attention: 2 forward + 5 backward products, halved
             + 6.0 * T * (n_vit + n_proj) + 3.5 * 4.0 * FRAMES * tokens * tokens * 72 * heads * Lv)
    return {"what": "BASELINE configs[4] on one GPU (not the headline; no ZeRO exchange): one training sample end to end - SigLIP-so400m (26 layers, "
                    "%d frames) -> mlp2x_gelu -> pool + 3-D PE + newline -> Qwen2-7B with labels over S=%d (%d answer tokens carry labels), backward of "
                    "all of it, AdamW on every parameter; random-init bf16 weights, f32 master weights / moments, all activations kept" % (FRAMES, S, answer_tokens),
            "value": 1e3 / ms, "unit": "samples/s", "ms_per_step": ms, "ms_forward_backward": sum(fb) / len(fb) * 1e3, "ms_adamw": sum(ad) / len(ad) * 1e3,
            "llm_tokens_per_s": S / (ms * 1e-3), "model_tflops": flops / (ms * 1e-3) / 1e12, "mfma_peak_tflops": 2500.0, "first_loss": first_loss,
            "params": n_llm + n_vit + n_proj + V * H + H, "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
            "recompute": {"what": "forward + backward with per-layer activation re-computation (gradient checkpointing as train_multi.sh:72; gradients bit-identical)",
                          "ms_forward_backward": rc_ms[-1], "peak_mem_gb": rc_peak}}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start `torch.distributed.run` with N ranks of this very file as a
    CHILD process (never an exec, and before this process has made any GPU call), let rank 0's JSON line through and return the child's
    exit code.  No retry: a failed child is a failed run.  (v3d.distributed.launch_ranks: the eval runners' --n_gpu uses the same.)"""
    from v3d import distributed as v3dist
    return v3dist.launch_ranks(n, argv, script=__file__, json_only=True)


def dry_run(a, rank, world):
    """V3D_BENCH_DRY=1: the N > 1 plumbing of main() with a stand-in for the scene pipeline and no GPU - the same sharding
    (v3d.distributed.shard), record gather (gather_records), barrier + max-over-ranks timing and line, over gloo on the CPU."""
    import torch.distributed as dist
    from v3d import distributed as v3dist
    multi = world > 1 or os.environ.get("V3D_BENCH_FORCE_DIST") == "1"
    if os.environ.get("V3D_BENCH_DRY_FAIL_RANK") == str(rank):      # (test hook: a rank that dies before the rendezvous)
        raise SystemExit(3)
    if multi:
        dist.init_process_group("gloo")
    my_ids = v3dist.shard(list(range(world * a.steps)), rank, world)
    if multi:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.002 * a.steps)
    answers = [[(7 * sid + k) % 1000 for k in range(NEW_TOKENS)] for sid in my_ids]
    merged = None
    if multi:
        merged = v3dist.gather_records([{"sample_id": sid, "pred_token_ids": row} for sid, row in zip(my_ids, answers)], torch.device("cpu"))
        dist.barrier()
    dt_s = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([dt_s], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_s = tt.item()
    if rank == 0:
        if multi:
            assert [r["sample_id"] for r in merged] == list(range(world * a.steps)), "gathered records are not in question order"
            assert all(r["pred_token_ids"] == [(7 * r["sample_id"] + k) % 1000 for k in range(NEW_TOKENS)] for r in merged)
        print(json.dumps({"metric": "scenes/sec ScanQA @32 frames", "value": world * a.steps / dt_s, "unit": "scenes/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt_s / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "data": "DRY RUN (V3D_BENCH_DRY=1): no GPU, a stand-in for the scene pipeline - not a measurement",
                          "rccl_ranks": dist.get_world_size() if multi else 1, "backend": dist.get_backend() if multi else None,
                          "config": {"workload": "dry run of the N > 1 plumbing", "parallelism": "scene-dp%d" % world}}), flush=True)
    if multi:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="one scene at a time: no decode groups, no prefill/decode overlap")
    ap.add_argument("--decode-group", type=int, default=16, help="scenes decoded together per pass over the weights (1..32)")
    ap.add_argument("--answer-batch", type=int, default=32, help="cached_questions / reuse_runner: questions answered together (1..32)")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[3]: e4m3 linears in the Qwen2 prefill; the headline line becomes that run")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra measurements appended to the default N=1 line (fp8, cached questions, grounding)")
    ap.add_argument("--no-fp8-extra", action="store_true", help="skip only the configs[3] extra")
    ap.add_argument("--no-train-extra", action="store_true", help="skip only the configs[4] (language-model training step) extra")
    ap.add_argument("--prefill-streams", type=int, default=2, help="streams that prefill consecutive scenes side by side (each with its own scratch)")
    ap.add_argument("--eval-runner-only", action="store_true", help="after the headline run only the eval_runner extra (development)")
    ap.add_argument("--scenes", type=int, default=8, help="distinct synthetic scenes resident in HBM, cycled over the steps")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))          # (nothing above has touched the GPU: `import torch` does not)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but the launcher's WORLD_SIZE is {world}")
    if os.environ.get("V3D_BENCH_DRY") == "1":
        return dry_run(a, rank, world)
    # V3D_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend - exercises the N > 1 code path (sharding, record gather,
    # max-over-ranks timing) on a one-GPU box; its numbers mean nothing (the ranks share one GPU) and the line says so
    rehearsal = os.environ.get("V3D_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
        os.environ["V3D_GEMM_STREAMK"] = "0"      # ranks share one card: the GEMM's split-K tail assumes one tail launch on the chip
    loader_pool, n_workers = None, 0
    if world == 1 and (a.eval_runner_only or not a.no_extras):     # the eval_runner extra's decoding processes: forked before the GPU is touched
        from v3d import eval_scanqa as _E, frame_io
        n_workers = _E.default_workers()
        loader_pool = frame_io.make_pool(n_workers)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # V3D_BENCH_FORCE_DIST=1: take the N > 1 code path (RCCL process group, record gather, max-over-ranks all-reduce) with ONE rank -
    # what a one-GPU box can execute of it (launched under torch.distributed.run with --nproc-per-node 1)
    multi = world > 1 or os.environ.get("V3D_BENCH_FORCE_DIST") == "1"
    if multi:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from v3d import distributed as v3dist
    from v3d import ops
    from v3d.engine import Engine, EngineConfig, random_state_dict
    from v3d.pipeline import ScenePipeline, SceneSample
    dtype = torch.bfloat16
    cfg = EngineConfig()
    sd = random_state_dict(cfg, dtype, dev, seed=0, ground_head=True)
    eng = Engine(cfg, sd, dtype=dtype, device=dev, max_frames=FRAMES, llm_fp8=a.fp8)
    # the rank's share of the (scene, question) list: the reference's stride sharding, model_scanqa.py:245
    n_scenes = max(1, min(a.scenes, a.steps))
    my_ids = v3dist.shard(list(range(world * a.steps)), rank, world)
    scenes = [synth_inputs(dev, dtype, seed=1000 + world * i + rank) for i in range(n_scenes)]
    torch.cuda.synchronize()

    def new_stamps():
        return {"pe": Stamp(), "gemm": Stamp(), "attn": Stamp()}
    stamps = new_stamps()

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    G_ALL = 1 if a.no_overlap else max(1, min(32, a.decode_group))

    def measure(eng, stamps):
        """W untimed warm-up scenes, then EXACTLY `steps` scenes between barrier + synchronize; max over ranks.  The timed region IS the
        product's scene pipeline (v3d.pipeline.ScenePipeline.run - the eval runners call the same function): prefill per scene on stream A,
        decode groups on stream B, two context sets."""
        pipe = ScenePipeline(eng, G_ALL, prefill_streams=a.prefill_streams)
        if a.warmup:
            pipe.run(samples_of(scenes, max(a.warmup, min(G_ALL, a.steps)), SceneSample), NEW_TOKENS, overlap=not a.no_overlap, trim=False)
        barrier()
        t0 = time.perf_counter()
        answers = torch.cat(pipe.run(samples_of(scenes, a.steps, SceneSample), NEW_TOKENS, overlap=not a.no_overlap, stamps=stamps, trim=False), 0)
        merged = None
        if multi:   # eval collation: ONE variable-length gather of the answer records to rank 0 (replaces Ray + file lock)
            recs = [{"sample_id": sid, "pred_token_ids": row} for sid, row in zip(my_ids, answers.tolist())]
            merged = v3dist.gather_records(recs, torch.device("cpu") if rehearsal else dev)
        barrier()
        dt_s = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([dt_s], device="cpu" if rehearsal else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_s = tt.item()
            if rank == 0:
                assert [r["sample_id"] for r in merged] == list(range(world * a.steps)), "gathered records are not in question order"
        assert answers.shape == (a.steps, NEW_TOKENS)
        eng.use(pipe.sets[0][0])
        return dt_s

    dt_s = measure(eng, stamps)
    extras = {}
    if world == 1 and not a.no_extras and not a.eval_runner_only:
        # BASELINE.json's stated synthetic proxy (a 4096-token sequence; SURVEY 8(d) asks for it NEXT to the true shape).  No frame count
        # of the real path gives 4096 visual rows (a frame is 14 x 15 = 210 rows), so the proxy is 19 frames x 210 = 3990
        # visual rows + 14 + 92 text ids = exactly 4096; everything else as in the headline.
        PF, PPOST = 19, 4096 - TEXT_PRE - 19 * 210
        keep_scenes = scenes
        scenes = [synth_inputs(dev, dtype, seed=2000 + i, frames=PF, text_post=PPOST) for i in range(min(4, n_scenes))]
        st4 = new_stamps()
        t4 = measure(eng, st4)
        scenes = keep_scenes
        a_us = st4["attn"].mean_us()
        extras["s4096_proxy"] = {
            "what": "BASELINE.json's stated synthetic proxy, a 4096-token sequence: %d frames (384x384; the 3-D code is hard-wired to SigLIP-384, SURVEY F4) "
                    "x 210 visual rows + %d + %d text ids = 4096, %d greedy tokens; the headline is the TRUE 32-frame shape (S = %d)"
                    % (PF, TEXT_PRE, PPOST, NEW_TOKENS, TEXT_PRE + FRAMES * 210 + TEXT_POST),
            "value": a.steps / t4, "unit": "scenes/s", "ms_per_step": t4 / a.steps * 1e3, "frames": PF, "seq_len": 4096,
            "attention_us_per_layer": a_us, "attention_frac_of_2.5PF": 2.0 * 4096 * 4096 * 128 * 28 / a_us / 1e6 / 2500.0,
            "gate_up_gemm_us": st4["gemm"].mean_us(), "gate_up_gemm_frac_of_2.5PF": 2.0 * 4096 * 37888 * 3584 / st4["gemm"].mean_us() / 1e6 / 2500.0}
    if world == 1 and a.eval_runner_only:
        extras["eval_runner"], extras["reuse_runner"] = measure_eval_runner(eng, dev, max(48, 3 * a.steps), loader_pool, n_workers)
    elif world == 1 and not a.no_extras:
        AB = max(1, min(32, a.answer_batch))
        nq, scene_ms = measure_cached_questions(eng, ops, scenes[0], dev, n_groups=max(1, a.steps // 4), group=AB)
        extras["cached_questions"] = {
            "what": "scene-level reuse (SURVEY 8 f1; not the headline): further questions about an ALREADY prefilled scene - %d question rows "
                    "per question run over the cached prefix of %d rows, %d questions per batch, %d new tokens each; the one-off scene prefill "
                    "is reported beside it" % (TEXT_POST, TEXT_PRE + FRAMES * 210, AB, NEW_TOKENS),
            "value": nq, "unit": "questions/s", "scene_prefill_ms": scene_ms}
        try:
            extras["decode_step"] = measure_decode_step(eng, ops, scenes, dev)
        except Exception as e:                          # an extra must never take the headline line down with it
            extras["decode_step"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            extras["eval_runner"], extras["reuse_runner"] = measure_eval_runner(eng, dev, max(48, 3 * a.steps), loader_pool, n_workers)
            extras["reuse_runner"]["vs_cached_questions"] = extras["reuse_runner"]["value"] / nq
        except Exception as e:                          # an extra must never take the headline line down with it
            extras["eval_runner"] = {"error": "%s: %s" % (type(e).__name__, e)}
        extras["ground_config2"] = {
            "what": "BASELINE configs[2] on one GPU: ScanRefer / Multi3DRefer grounding forward, 32 frames, 50 object proposals, "
                    "infonce head (prefill only, no decode)",
            "value": measure_grounding(eng, ops, scenes, dev, max(4, a.steps // 2)), "unit": "scenes/s"}
        if not a.fp8 and not a.no_fp8_extra:
            del eng
            torch.cuda.empty_cache()
            eng8 = Engine(cfg, sd, dtype=dtype, device=dev, max_frames=FRAMES, llm_fp8=True)
            st8 = new_stamps()
            t8 = measure(eng8, st8)
            g8 = st8["gemm"].mean_us()
            extras["fp8_config3"] = {
                "what": "BASELINE configs[3]: same scene step with e4m3 (per-row scaled) weights in the Qwen2 linears and LM head "
                        "(prefill W8A8 on MFMA, decode W8A16 weight streaming); ViT, attention, norms and residual stream stay bf16",
                "value": a.steps / t8, "unit": "scenes/s", "ms_per_step": t8 / a.steps * 1e3,
                "gate_up_gemm_us": g8, "gate_up_gemm_tflops": 2.0 * (TEXT_PRE + FRAMES * 210 + TEXT_POST) * 37888 * 3584 / g8 / 1e6,
                "mfma_peak_tflops": 5000.0}
            del eng8
        if not a.no_train_extra:
            eng = None
            torch.cuda.empty_cache()
            try:
                extras["train_config4"] = measure_train_step(dev)
            except Exception as e:                      # an extra must never take the headline line down with it
                extras["train_config4"] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()

    if rank == 0:
        S = TEXT_PRE + FRAMES * 210 + TEXT_POST
        pe_us = stamps["pe"].mean_us()
        pe_bytes = FRAMES * 729 * 3584 * 2 + FRAMES * 210 * 3584 * 2          # feat read + token rows written (SURVEY 8d, K7+K5-K6+K8)
        gemm_us = stamps["gemm"].mean_us()
        gemm_flops = 2.0 * S * 37888 * 3584                                   # the gate/up GEMM stamped in llm_forward
        gemm_peak = 5000.0 if a.fp8 else 2500.0
        attn_us = stamps["attn"].mean_us()
        attn_flops = 2.0 * S * S * 128 * 28                                   # causal: half of 4*S^2*d*H
        traffic, traffic_note, attn_busy, gemm_busy = None, None, None, None
        gemm_traffic, attn_traffic, mfma_traffic_note = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            pm = json.load(open(pmc))
            attn_busy = pm.get("attention_mfma_busy_frac")       # PMC: MFMA-busy share of SIMD cycles at the actual clock
            gemm_busy = pm.get("gemm_gate_up_mfma_busy_frac")
            # the MFMA-bound kernels' memory-side traffic (fabric requests: Infinity-Cache hits included, so an upper bound of HBM bytes)
            if not a.fp8 and pm.get("gemm_source_sha") == kernel_source_sha("gemm.hip"):
                gemm_traffic = pm.get("gemm_gate_up_fabric_bytes_per_launch")
            if pm.get("attention_source_sha") == kernel_source_sha("attention.hip"):
                attn_traffic = pm.get("attention_fabric_bytes_per_launch")
            mfma_traffic_note = pm.get("gemm_attention_traffic_note")
            sha = kernel_source_sha("visual_tokens.hip")
            if pm.get("visual_tokens_source_sha") == sha:
                traffic = pm.get("visual_tokens_hbm_bytes_per_launch")
                traffic_note = "PMC FETCH_SIZE x2 + WRITE_SIZE (separate rocprofv3 --pmc passes, profiles/), measured on this kernel source (%s)" % sha
            else:
                traffic_note = "stale: profiles/pmc_summary.json was measured on kernel source %s, this build is %s" % (pm.get("visual_tokens_source_sha"), sha)
        line = {
            "metric": "scenes/sec ScanQA @32 frames", "value": world * a.steps / dt_s, "unit": "scenes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt_s / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8" if a.fp8 else "bf16",
            "data": "synthetic" + (" (REHEARSAL: all ranks share cuda:0 over gloo - not a measurement)" if rehearsal else ""),
            "rccl_ranks": dist.get_world_size() if multi else 1, "backend": dist.get_backend() if multi else None,
            "config": {"workload": "ScanQA val, uniform 32 frames, %s, 1xMI355X per rank: 32x(480x640 u16 depth + 384x384 RGB) -> "
                                   "SigLIP-so400m(26L) + mlp2x_gelu + 3D-PE fusion -> Qwen2-7B prefill S=%d + %d greedy tokens (%d decode passes "
                                   "over the weights); random-init weights at true widths; %d distinct synthetic scenes cycled"
                                   % ("fp8 LLM linears (configs[3])" if a.fp8 else "bf16", S, NEW_TOKENS, NEW_TOKENS - 1, n_scenes),
                       "frames": FRAMES, "seq_len": S, "new_tokens": NEW_TOKENS, "decode_weight_passes": NEW_TOKENS - 1,
                       "parallelism": "scene-dp%d" % world, "decode_group": 1 if a.no_overlap else G_ALL,
                       "dead_work_not_computed": "LM head for the last row only (the reference forms all S rows of logits and reads one); the LAST decoder "
                                                 "layer's attention / o_proj / MLP for the last row only (its K/V rows are formed for every row; no later "
                                                 "computation reads the other rows' outputs)",
                       "scheduling": "one scene at a time" if a.no_overlap else
                                     "prefill per scene on %d stream(s) (consecutive scenes side by side); the decode passes of up to %d scenes share each "
                                     "pass over the weights on another stream" % (a.prefill_streams, G_ALL)},
            "roofline": {"kernel": "%s (Qwen2 gate/up + SwiGLU, M=%d N=37888 K=3584): the largest share of the step" %
                                   ("gemm_fp8_kernel" if a.fp8 else "gemm256pp_kernel", S), "bound": "mfma",
                         "achieved": gemm_flops / gemm_us / 1e6, "peak": gemm_peak, "unit": "TFLOP/s",
                         "frac": gemm_flops / gemm_us / 1e6 / gemm_peak, "traffic": gemm_traffic, "traffic_note": mfma_traffic_note if gemm_traffic else None,
                         "us_per_launch": gemm_us, "algorithmic_flops": gemm_flops,
                         "algorithmic_bytes": (S * 3584 + 37888 * 3584) * (1 if a.fp8 else 2) + S * 18944 * 2,
                         "mfma_busy_frac_pmc": None if a.fp8 else gemm_busy},
            "roofline_north_star": {"kernel": "visual_tokens_kernel (bilinear pool + 3D-PE add + newline, K5-K8)", "bound": "hbm",
                                    "achieved": pe_bytes / pe_us / 1e3, "peak": 8000.0, "unit": "GB/s", "frac": pe_bytes / pe_us / 1e3 / 8000.0,
                                    "traffic": traffic, "traffic_note": traffic_note, "us_per_launch": pe_us, "algorithmic_bytes": pe_bytes},
            "roofline_attention": {"kernel": "attn_prefill_kernel (causal GQA, S=%d, 28q/4kv, hd128)" % S, "bound": "mfma",
                                   "achieved": attn_flops / attn_us / 1e6, "peak": 2500.0, "unit": "TFLOP/s",
                                   "frac": attn_flops / attn_us / 1e6 / 2500.0, "traffic": attn_traffic, "us_per_launch": attn_us,
                                   "algorithmic_bytes": 2 * S * 28 * 128 * 2 + 2 * S * 4 * 128 * 2,
                                   "mfma_busy_frac_pmc": attn_busy},
        }
        line.update(extras)
        if world == 1 and not a.no_cpu_baseline:
            # n = ALL host cores this process may run on (printed in the line as `cores`): the scheduler affinity, cut to
            # the container's CPU quota where one is set (256 threads on a 16-core quota ran the same sample 2.6x SLOWER than 16)
            cores, why = usable_cores()
            line["cpu_baseline"] = cpu_baseline(cores)
            line["cpu_baseline"].update({"host_cores_total": os.cpu_count(), "cores_note": why})
        print(json.dumps(line))
    if loader_pool is not None:
        loader_pool.shutdown(wait=True, cancel_futures=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
