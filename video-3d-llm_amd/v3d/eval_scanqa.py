"""ScanQA evaluation runner with the I/O contract of the reference's driver (llava/eval/model_scanqa.py), one process per
GPU instead of Ray actors + a file lock:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m v3d.eval_scanqa \\
        --model-path <ckpt> --video-folder data --embodiedscan-folder data/embodiedscan \\
        --question-file data/processed/scanqa_val_llava_style.json --answer-file out/scanqa.jsonl --max_frame_num 32

* prompt ids: `chatml_ids` restates preprocess_qwen (model_scanqa.py:29-80): ChatML turns with IMAGE_TOKEN_INDEX (-200) where the
  "<image>" placeholder stood (the tokenizer-dependent ids themselves are parity-unpinned: no Qwen2 tokenizer ships with the
  reference; the STRUCTURE is what tests/test_gpu_eval_harness.py checks);
* sharding: `questions[rank::world]` (model_scanqa.py:245);
* per question: VideoProcessor.process_3d_video -> merge_video_dict -> cast to the model dtype on the device (:156-165) ->
  model.generate(..., modalities="video", max_new_tokens=512, use_cache=True, video_dict=...) (:173-185) -> decode, strip the
  stop string (:188-192) -> the record {dataset, sample_id, prompt, pred_response, gt_response, model_id, question_type} (:196-204);
* collation: ONE variable-length gather of the records to rank 0 (v3d.distributed.gather_records over RCCL), which writes
  the JSONL answer file in the original question order (the reference appends under fasteners.InterProcessLock in arrival order).
`--reuse-scenes` (not in the reference): consecutive questions of one scene share the scene's prefill
(Engine.prefill_scene) and are answered in batches of up to 32 (Engine.answer_group); records and order are unchanged.  It runs on the
pipeline too (r04, v3d.pipeline.SceneReusePipeline: asynchronous loader, the next scene's prefill beside this scene's answers);
`--no-pipeline` gives the synchronous form.
"""
import argparse
import json
import os
import re
import time

import torch

from . import distributed as D
from .token_ids import IMAGE_TOKEN_INDEX

EXTRA_PROMPT = ("The video captures 3D spatial information of a scene. Please focus on the spatial relationships in the video and "
                "answer the following questions.\n")
STOP_STR = "<|im_end|>"          # conv_templates["qwen_1_5"].sep (llava/conversation.py:443-452): the ChatML turn terminator


def chatml_ids(turns, tokenizer, has_image=True, system_message="You are a helpful assistant."):
    """Ids of a ChatML conversation as preprocess_qwen builds them (model_scanqa.py:29-80): <|im_start|>system\\n{system}<|im_end|>\\n,
    then per turn <|im_start|>{user|assistant}\\n{text}<|im_end|>\\n; an "<image>" placeholder in a user turn becomes
    IMAGE_TOKEN_INDEX followed by a newline (:53-59); a turn whose value is None is left open (the generation prompt, :62-63).
    turns: [{"from": "human" | "gpt", "value": str | None}, ...].  Returns LongTensor [1, n]."""
    tok = lambda s: tokenizer(s).input_ids       # noqa: E731
    ids_attr = getattr(tokenizer, "additional_special_tokens_ids", None)     # :32; newer transformers dropped the attribute
    im_start, im_end = (ids_attr[:2] if ids_attr else tokenizer.convert_tokens_to_ids(["<|im_start|>", "<|im_end|>"]))
    nl = tok("\n")
    role_ids = {"human": tok("<|im_start|>user"), "gpt": tok("<|im_start|>assistant")}
    if turns and turns[0]["from"] != "human":
        turns = turns[1:]
    ids = [im_start] + tok("system") + nl + tok(system_message) + [im_end] + nl
    for turn in turns:
        head = role_ids[turn["from"]] + nl
        text = turn["value"]
        if text is None:
            ids += head
        elif has_image and "<image>" in text:
            n_img = len(re.findall("<image>", text))
            parts = text.split("<image>")
            cur = list(head)
            for i, part in enumerate(parts):
                cur += tok(part)
                if i < len(parts) - 1:
                    cur += [IMAGE_TOKEN_INDEX] + nl
            cur += [im_end] + nl
            assert cur.count(IMAGE_TOKEN_INDEX) == n_img
            ids += cur
        else:
            ids += head + tok(text) + [im_end] + nl
    return torch.tensor([ids], dtype=torch.long)


def build_prompt_ids(line, tokenizer, mm_use_im_start_end=False):
    """model_scanqa.py:139-154: "<image>\\n" is put in front of the question of the first turn; the ids come from the record's own
    first conversation turn plus an open assistant turn."""
    return chatml_ids([line["conversations"][0], {"from": "gpt", "value": None}], tokenizer, has_image=True)


def make_record(line, pred_text, model_name, extra_prompt=EXTRA_PROMPT):
    """The JSONL record of model_scanqa.py:196-204."""
    return {"dataset": line["metadata"]["dataset"], "sample_id": line["id"], "prompt": extra_prompt + line["conversations"][0]["value"],
            "pred_response": pred_text, "gt_response": line["conversations"][1]["value"], "model_id": model_name,
            "question_type": line["metadata"]["question_type"]}


def clean_answer(text, stop_str=STOP_STR):
    """model_scanqa.py:188-192."""
    text = text.strip()
    if text.endswith(stop_str):
        text = text[: -len(stop_str)]
    return text.strip()


def _video_inputs(video_processor, image_processor, video_id, model, max_frame_num, force_sample=True, box_input=None):
    from llava.video_utils import merge_video_dict
    one = video_processor.process_3d_video(video_id, image_processor, force_sample=force_sample, frames_upbound=max_frame_num)
    if box_input is not None:
        one["box_input"] = box_input                                                          # model_scan2cap.py:170
    vd = merge_video_dict([one])
    images = vd.pop("images").to(device=model.device, dtype=model.dtype)                      # :163
    for k in vd:
        vd[k] = vd[k].to(device=model.device, dtype=model.dtype)                              # :164-165
    return images, vd


def evaluate(questions, answer_fn, rank, world, device, shard="stride"):
    """Shard, answer, collate.  answer_fn(list of question records of ONE rank) -> list of output records in the same order.
    shard "stride": `questions[rank::world]` (model_scanqa.py:245); "scene": whole scenes per rank (v3d.distributed.shard_scene_indices;
    what --reuse-scenes needs to keep one prefill per scene under data parallelism).  Returns on rank 0 every record in the
    original question order, on the other ranks None."""
    if shard == "scene":
        idx = D.shard_scene_indices([q["video"] for q in questions], rank, world)
    elif shard == "stride":
        idx = D.shard_indices(len(questions), rank, world)
    else:
        raise ValueError(f"unknown sharding {shard!r}")
    recs = answer_fn([questions[i] for i in idx])
    if len(recs) != len(idx):
        raise RuntimeError("answer_fn must return one record per question")
    if world == 1:
        return recs
    return D.gather_indexed(recs, idx, device)


def self_launch(n_gpu, module, argv):
    """`--n_gpu N` as the reference's drivers take it (model_scanqa.py:222, 242-247: `ray.init()` and one remote worker per GPU): started
    plainly (no RANK / WORLD_SIZE in the environment) with N > 1, this process - which has not touched the GPU - starts N ranks of `module`
    under torch.distributed.run and returns their exit code; returns None where there is nothing to launch (N <= 1, or already a rank)."""
    if not n_gpu or n_gpu <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return None
    import sys
    return D.launch_ranks(n_gpu, list(sys.argv[1:] if argv is None else argv), module=module)


def rank_setup(n_gpu=None):
    """(rank, world, device, gather device) of this process and its process group.  V3D_EVAL_REHEARSAL=1: every rank on cuda:0 over gloo -
    the N > 1 path (sharding, per-rank loaders, record gather) on a one-GPU box; records gathered through host memory."""
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    if n_gpu and n_gpu > 1 and n_gpu != world:
        raise SystemExit(f"--n_gpu {n_gpu} but the launcher's WORLD_SIZE is {world}")
    rehearsal = os.environ.get("V3D_EVAL_REHEARSAL") == "1"
    if rehearsal:
        local = 0
        os.environ["V3D_GEMM_STREAMK"] = "0"          # ranks share one card: the GEMM's split-K tail assumes one tail launch on the chip
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    return rank, world, dev, (torch.device("cpu") if rehearsal else dev)


def default_workers():
    """Loader processes per rank: the host cores this process may run on, divided among the ranks of the node, at most 16 (one
    question's 32 frames cost about half a core-second to decode; a GPU answers ten questions a second)."""
    local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 8
    return max(2, min(16, cores // max(1, local)))


def model_answer_fn(model, tokenizer, image_processor, video_processor, model_name, max_frame_num=32, max_new_tokens=512,
                    reuse_scenes=False, times=None, pipeline=True, group_size=16, workers=None, stats=None, record_fn=None, pool=None,
                    box_input_fn=None, skip_fn=None):
    """The per-rank loop of model_scanqa.py:130-206 around `model` (the loader-produced LlavaQwenForCausalLM).
    reuse_scenes: one scene prefill per run of consecutive questions about a scene (pipelined or, with pipeline=False, synchronous).
    pipeline (default): v3d.pipeline - asynchronous host loader, prefill / grouped-decode overlap, device-side stop test; the
    records are those of the one-question-at-a-time loop (`pipeline=False`, the reference's own order of operations) up to the
    f32 summation order of the decode linears (tests/test_gpu_eval_harness.py).  stats: dict that receives the host-stage seconds.
    pool: v3d.frame_io.make_pool(...) of decoding processes (make it BEFORE the process initialises the GPU); default: one per call;
    workers = 0 decodes on the calling thread.
    box_input_fn(line) -> [x, y, z] or None: the Scan2Cap prompt's box centre (model_scan2cap.py:137-139, 170; its PE is added to the
    <coord> token rows); skip_fn(line) -> True: the question is not run and its answer is "" (:199-200)."""
    record_fn = record_fn or (lambda line, text: make_record(line, text, model_name))
    coord_ids = getattr(getattr(model, "config", None), "coord_token_ids", None)

    def with_skipped(lines, run):
        """run(active lines) -> texts; skipped lines get the empty answer (model_scan2cap.py:199-200)."""
        active = [l for l in lines if not (skip_fn and skip_fn(l))]
        texts = iter(run(active))
        return [record_fn(l, "" if (skip_fn and skip_fn(l)) else next(texts)) for l in lines]

    def decode(ids):
        return clean_answer(tokenizer.batch_decode(ids.view(1, -1), skip_special_tokens=True)[0])

    def one_by_one(lines):
        def run(active):
            out = []
            for line in active:
                ids = build_prompt_ids(line, tokenizer).to(model.device)
                if int((ids == IMAGE_TOKEN_INDEX).sum()) != 1:
                    raise ValueError("exactly one <image> placeholder per prompt (model_scanqa.py:61)")
                images, vd = _video_inputs(video_processor, image_processor, line["video"], model, max_frame_num,
                                           box_input=box_input_fn(line) if box_input_fn else None)
                t0 = time.time()
                with torch.inference_mode():
                    toks = model.generate(ids, images=images, modalities="video", do_sample=False, temperature=0.0, top_p=None, num_beams=1,
                                          max_new_tokens=max_new_tokens, use_cache=True, video_dict=vd)
                if times is not None:
                    torch.cuda.synchronize()
                    times.append(time.time() - t0)
                out.append(decode(toks[0]))
            return out
        return with_skipped(lines, run)

    def scene_batches(lines):
        """Consecutive questions of one scene: ONE scene prefill, answers in batches of up to 32 (Engine.answer_group)."""
        eng, out, i = model.engine, [], 0
        nb = eng.MAX_GROUP
        eos = model._eos()
        while i < len(lines):
            j = i
            while j < len(lines) and lines[j]["video"] == lines[i]["video"]:
                j += 1
            ids = [build_prompt_ids(l, tokenizer)[0] for l in lines[i:j]]
            at = int((ids[0] == IMAGE_TOKEN_INDEX).nonzero()[0])
            prefix = ids[0][: at + 1]
            if any(not torch.equal(x[: at + 1], prefix) for x in ids):
                raise ValueError("questions of one scene must share the prompt prefix up to <image>")
            images, vd = _video_inputs(video_processor, image_processor, lines[i]["video"], model, max_frame_num)
            with torch.inference_mode():
                P = eng.prefill_scene(prefix, images[0], vd["world_coords"][0])
                for b in range(i, j, nb):
                    qs = [x[at + 1:] for x in ids[b - i: min(j, b + nb) - i]]
                    room = eng.cfg.llm.max_pos - P - max(len(q) for q in qs) + 1
                    answers = eng.answer_group(qs, max_new_tokens=min(max_new_tokens, room), eos_token_id=eos)
                    out += [record_fn(l, decode(a)) for l, a in zip(lines[b: b + len(qs)], answers)]
            i = j
        return out

    def pipelined(lines):
        """v3d.pipeline.ScenePipeline over this rank's questions (module docstring there)."""
        return with_skipped(lines, _pipelined_texts)

    def _pipelined_texts(lines):
        from .pipeline import AsyncSceneLoader, ScenePipeline, SceneSample
        if not lines:
            return []
        eng = model.engine
        pipe = model.__dict__.get("_v3d_pipeline")
        if pipe is None or pipe.G != group_size:
            pipe = model.__dict__["_v3d_pipeline"] = ScenePipeline(
                eng, group_size, crop=image_processor.crop_size["width"], image_mean=image_processor.image_mean,
                image_std=image_processor.image_std, rescale=image_processor.rescale_factor, prefill_streams=2)
        prompts = []
        for line in lines:
            ids = build_prompt_ids(line, tokenizer)[0]
            if int((ids == IMAGE_TOKEN_INDEX).sum()) != 1:
                raise ValueError("exactly one <image> placeholder per prompt (model_scanqa.py:61)")
            prompts.append(ids)
        loader = AsyncSceneLoader([l["video"] for l in lines], lambda vid: video_processor.describe_scene(vid, True, max_frame_num),
                                  workers=default_workers() if workers is None else workers, pool=pool)
        waited = [0.0]

        def samples():
            for j, line in enumerate(lines):
                raw, w = loader.get(j)
                waited[0] += w
                box = box_input_fn(line) if box_input_fn else None
                yield SceneSample(input_ids=prompts[j], raw=raw, key=line["video"],
                                  box_input=None if box is None else torch.tensor([box], dtype=torch.float32),
                                  extra={"coord_token_id": coord_ids[0]} if (box is not None and coord_ids) else {})

        n_vis = max_frame_num * eng.cfg.pool_out * (eng.cfg.pool_out + 1)
        room = eng.cfg.llm.max_pos - (max((len(p) for p in prompts), default=1) - 1 + n_vis) + 1
        t0 = time.time()
        try:
            toks = pipe.run(samples(), min(max_new_tokens, room), eos_token_id=model._eos())
        finally:
            loader.close()
        if times is not None and lines:
            times.extend([(time.time() - t0) / len(lines)] * len(lines))
        if stats is not None:
            stats.update({"host_thread_seconds": dict(loader.stage_seconds), "loader_wait_seconds": waited[0],
                          "upload_enqueue_seconds": pipe.upload_seconds, "questions": len(lines), "wall_seconds": time.time() - t0})
        return [decode(t) for t in toks]

    def scene_pipelined(lines):
        """--reuse-scenes on the pipeline (v3d.pipeline.SceneReusePipeline; r04): asynchronous loader per SCENE, scene i + 1's upload / ViT /
        prefix prefill on one stream while scene i's answer batches run on another; records = scene_batches' records."""
        return with_skipped(lines, _scene_pipelined_texts)

    def _scene_pipelined_texts(lines):
        from .pipeline import AsyncSceneLoader, SceneReusePipeline, SceneSample
        if not lines:
            return []
        eng = model.engine
        pipe = model.__dict__.get("_v3d_reuse_pipeline")
        if pipe is None:
            pipe = model.__dict__["_v3d_reuse_pipeline"] = SceneReusePipeline(
                eng, crop=image_processor.crop_size["width"], image_mean=image_processor.image_mean, image_std=image_processor.image_std,
                rescale=image_processor.rescale_factor)
        groups, i = [], 0                                  # runs of consecutive questions about one scene
        while i < len(lines):
            j = i
            while j < len(lines) and lines[j]["video"] == lines[i]["video"]:
                j += 1
            ids = [build_prompt_ids(l, tokenizer)[0] for l in lines[i:j]]
            if any(int((x == IMAGE_TOKEN_INDEX).sum()) != 1 for x in ids):
                raise ValueError("exactly one <image> placeholder per prompt (model_scanqa.py:61)")
            at = int((ids[0] == IMAGE_TOKEN_INDEX).nonzero()[0])
            prefix = ids[0][: at + 1]
            if any(not torch.equal(x[: at + 1], prefix) for x in ids):
                raise ValueError("questions of one scene must share the prompt prefix up to <image>")
            groups.append((lines[i]["video"], prefix, [x[at + 1:] for x in ids]))
            i = j
        loader = AsyncSceneLoader([g[0] for g in groups], lambda vid: video_processor.describe_scene(vid, True, max_frame_num),
                                  workers=default_workers() if workers is None else workers, pool=pool, ahead=2, keep=2)
        waited = [0.0]

        def scenes():
            for k, (vid, prefix, qs) in enumerate(groups):
                raw, w = loader.get(k)
                waited[0] += w
                yield SceneSample(input_ids=prefix, raw=raw, key=None), qs

        t0 = time.time()
        try:
            answers = pipe.run(scenes(), max_new_tokens, eos_token_id=model._eos())
        finally:
            loader.close()
        if times is not None and lines:
            times.extend([(time.time() - t0) / len(lines)] * len(lines))
        if stats is not None:
            stats.update({"host_thread_seconds": dict(loader.stage_seconds), "loader_wait_seconds": waited[0],
                          "upload_enqueue_seconds": pipe.upload_seconds, "questions": len(lines), "scenes": len(groups),
                          "wall_seconds": time.time() - t0})
        return [decode(t) for scene_answers in answers for t in scene_answers]

    if reuse_scenes:
        return scene_pipelined if pipeline else scene_batches
    return pipelined if pipeline else one_by_one


def load_model(model_path, overwrite_cfg=False):
    """load_pretrained_model (llava/model/builder.py) - the reference's own loader when its checkout is importable behind the
    overlay, else the same call sequence from v3d.loader."""
    from llava.mm_utils import get_model_name_from_path
    name = get_model_name_from_path(model_path)
    overwrite = {"tie_word_embeddings": False, "use_cache": True, "vocab_size": 151649} if overwrite_cfg else {}     # model_scanqa.py:94-99
    try:
        from llava.model.builder import load_pretrained_model
    except ImportError:
        from .loader import load_pretrained_model
    tokenizer, model, image_processor, _ = load_pretrained_model(model_path, None, name, overwrite_config=overwrite)
    return tokenizer, model, image_processor, name


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--video-folder", default="data")
    ap.add_argument("--embodiedscan-folder", default="data/embodiedscan")
    ap.add_argument("--metadata-folder", default="data/metadata")
    ap.add_argument("--question-file", required=True)
    ap.add_argument("--answer-file", default="answer.jsonl")
    ap.add_argument("--test_size", type=int, default=10000000)
    ap.add_argument("--max_frame_num", type=int, default=32)
    ap.add_argument("--max-new-tokens", type=int, default=512)
    ap.add_argument("--frame_sampling_strategy", default="uniform")
    ap.add_argument("--overwrite_cfg", action="store_true")
    ap.add_argument("--reuse-scenes", action="store_true", help="share one scene prefill between consecutive questions of a scene")
    ap.add_argument("--no-pipeline", action="store_true", help="one question at a time on one stream (the reference's order of operations)")
    ap.add_argument("--decode-group", type=int, default=16, help="scenes whose decode steps share a pass over the weights (1..16)")
    ap.add_argument("--loader-workers", type=int, default=0,
                    help="host processes decoding frames ahead of the GPU (0: cores / ranks, at most 16; -1: none, decode on the main thread)")
    ap.add_argument("--shard", choices=("stride", "scene"), default=None,
                    help="stride: questions[rank::world] (the reference); scene: whole scenes per rank (default with --reuse-scenes)")
    ap.add_argument("--n_gpu", type=int, default=None,
                    help="the reference's flag (model_scanqa.py:222): started plainly with N > 1, launch N ranks of this runner on this node")
    a = ap.parse_args(argv)
    if os.path.exists(a.answer_file):                                                         # model_scanqa.py:238-240
        print(f"The {a.answer_file} already exists!!!")
        return 0
    rc = self_launch(a.n_gpu, "v3d.eval_scanqa", argv)
    if rc is not None:
        return rc
    with open(os.path.expanduser(a.question_file)) as f:
        questions = json.load(f)[: a.test_size]
    pool = None
    if not a.no_pipeline and a.loader_workers >= 0:
        from . import frame_io
        pool = frame_io.make_pool(a.loader_workers or default_workers())        # forked before this process touches the GPU
    rank, world, dev, gather_dev = rank_setup(a.n_gpu)
    from llava.video_utils import VideoProcessor
    tokenizer, model, image_processor, name = load_model(os.path.expanduser(a.model_path), a.overwrite_cfg)
    vp = VideoProcessor(video_folder=a.video_folder, annotation_dir=a.embodiedscan_folder, frame_sampling_strategy=a.frame_sampling_strategy,
                        metadata_dir=a.metadata_folder)
    times = []
    fn = model_answer_fn(model, tokenizer, image_processor, vp, name, a.max_frame_num, a.max_new_tokens, a.reuse_scenes, times,
                         pipeline=not a.no_pipeline, group_size=a.decode_group, workers=(0 if a.loader_workers < 0 else a.loader_workers or None), pool=pool)
    records = evaluate(questions, fn, rank, world, gather_dev, shard=a.shard or ("scene" if a.reuse_scenes else "stride"))
    if rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(a.answer_file)), exist_ok=True)
        with open(a.answer_file, "w") as f:
            for r in records:
                f.write(json.dumps(r) + "\n")
        if times:
            print(f"time: {sum(times) / len(times)}")                                        # model_scanqa.py:252
    if pool is not None:
        pool.shutdown(wait=True, cancel_futures=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
