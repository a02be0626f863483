"""Runners for the reference's other 3-D eval drivers, on the same loop as v3d.eval_scanqa (one process per GPU, asynchronous host
loader, one gather to rank 0 that writes the JSONL in question order):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m v3d.eval_3d --task scanrefer \\
        --model-path <ckpt> --video-folder data --embodiedscan-folder data/embodiedscan \\
        --question-file data/processed/scanrefer_vg_val_llava_style.json --answer-file out/scanrefer.jsonl \\
        --frame_sampling_strategy mc-ratio90 --max_frame_num 32

  task          reference driver                              what runs per question                           record (the driver's own keys)
  sqa3d         llava/eval/model_sqa3d.py:126-200             generate (as ScanQA)                             model_scanqa's seven keys
  scan2cap      llava/eval/model_scan2cap.py:129-212          generate; box_input[:3] -> PE on the <coord>     ... + "scene"; gt = line["annotations"];
                                                              token rows; box_input None -> "" without a run     (:137-139, 170, 199-212)
  scanrefer     llava/eval/model_scanrefer.py:130-195         ONE prefill with object proposals -> scores;     pred_response = the arg-max box (the zero-target
                                                              (forward(labels, use_object_proposals=True))       falls back to the best real box, :174-179)
  multi3drefer  llava/eval/model_multi3drefer.py:120-185      the same forward                                  "scores" + "objects" (:171-181)
  scanqa        llava/eval/model_scanqa.py                    = v3d.eval_scanqa

`v3d.eval_ground`, `v3d.eval_scan2cap`, `v3d.eval_sqa3d` are entry points with the task fixed.
"""
import argparse
import json
import os
import time

import torch

from . import eval_scanqa as E
from .token_ids import IGNORE_INDEX, IMAGE_TOKEN_INDEX

GEN_TASKS = ("scanqa", "sqa3d", "scan2cap")
GROUND_TASKS = ("scanrefer", "multi3drefer")


def chatml_ids_labels(turns, tokenizer, system_message="You are a helpful assistant."):
    """preprocess_qwen of the grounding drivers (model_scanrefer.py:28-80): the ids of E.chatml_ids plus the training targets -
    system / user turns masked (IGNORE_INDEX) except their <|im_start|>, <|im_end|> and newline, the assistant turn's tokens kept.
    Only the assistant turn's kept tokens matter here: predict_box finds the <ground> label among them (llava_qwen.py:280-281)."""
    tok = lambda s: tokenizer(s).input_ids       # noqa: E731
    ids_attr = getattr(tokenizer, "additional_special_tokens_ids", None)
    im_start, im_end = (ids_attr[:2] if ids_attr else tokenizer.convert_tokens_to_ids(["<|im_start|>", "<|im_end|>"]))
    nl = tok("\n")
    roles = {"human": tok("<|im_start|>user"), "gpt": tok("<|im_start|>assistant")}
    if turns and turns[0]["from"] != "human":
        turns = turns[1:]
    system = [im_start] + tok("system") + nl + tok(system_message) + [im_end] + nl
    ids = list(system)
    labels = [im_start] + [IGNORE_INDEX] * (len(system) - 3) + [im_end] + nl
    for turn in turns:
        role, text = roles[turn["from"]], turn["value"]
        if text is not None and "<image>" in text:
            parts = text.split("<image>")
            cur = role + nl
            for i, part in enumerate(parts):
                cur = cur + tok(part)
                if i < len(parts) - 1:
                    cur = cur + [IMAGE_TOKEN_INDEX] + nl
            cur = cur + [im_end] + nl
        elif text is None:
            cur = role + nl
        else:
            cur = role + nl + tok(text) + [im_end] + nl
        ids += cur
        if turn["from"] == "human":
            labels += [im_start] + [IGNORE_INDEX] * (len(cur) - 3) + [im_end] + nl
        else:
            labels += [im_start] + [IGNORE_INDEX] * len(role) + cur[len(role) + 1: -2] + [im_end] + nl
    if len(ids) != len(labels):
        raise RuntimeError("preprocess_qwen restatement: ids and labels differ in length")
    return torch.tensor([ids], dtype=torch.long), torch.tensor([labels], dtype=torch.long)


def ground_record(task, line, scores, objects_half, model_name, extra_prompt=E.EXTRA_PROMPT):
    """The JSONL records of model_scanrefer.py:181-189 / model_multi3drefer.py:171-181.  scores: [n_obj + 1] (last = the zero-target);
    objects_half: the proposals as the driver holds them, [n_obj, 6] rounded to f16 (`video_dict[k].half()`, :161-162)."""
    base = {"dataset": line["metadata"]["dataset"], "sample_id": line["id"], "prompt": extra_prompt + line["conversations"][0]["value"]}
    tail = {"gt_response": line["box"], "model_id": model_name, "question_type": line["metadata"]["question_type"]}
    if task == "multi3drefer":
        return {**base, "scores": scores.tolist(), "objects": objects_half.tolist(), **tail}
    pred = int(torch.argmax(scores))                    # torch.max(scores, dim=0): the first maximum
    if pred >= objects_half.shape[0]:                   # the zero-target won: objects[pred] raises in the driver -> best real box (:177-179)
        pred = int(torch.argmax(scores[:-1]))
    return {**base, "pred_response": objects_half[pred].tolist(), **tail}


def ground_answer_fn(task, model, tokenizer, image_processor, video_processor, model_name, max_frame_num=32, workers=None, pool=None,
                     stats=None, sync_every=16):
    """Per-rank loop of the grounding drivers around the engine: host loader -> device inputs -> ONE prefill with the object proposals
    (Engine.ground_scores: patch masks + masked means + box-centre PE, decoder, infonce head) -> scores.  The scores stay on the
    device and are fetched every `sync_every` questions, so the launch queue does not drain per question."""
    from .pipeline import AsyncSceneLoader, ScenePipeline, SceneSample
    gt_ids = getattr(model.config, "ground_token_ids", None)
    if not gt_ids:
        raise ValueError("config.ground_token_ids is needed to locate the <ground> label (train_3d.py:1698-1713 stores it)")

    def run(lines):
        eng = model.engine
        pipe = model.__dict__.get("_v3d_pipeline")
        if pipe is None:
            pipe = model.__dict__["_v3d_pipeline"] = ScenePipeline(
                eng, 16, crop=image_processor.crop_size["width"], image_mean=image_processor.image_mean,
                image_std=image_processor.image_std, rescale=image_processor.rescale_factor, prefill_streams=2)
        loader = AsyncSceneLoader([l["video"] for l in lines], lambda vid: video_processor.describe_scene(vid, True, max_frame_num),
                                  workers=E.default_workers() if workers is None else workers, pool=pool)
        out, pending = [], []
        t0 = time.time()

        def flush():
            for line, sc, obj in pending:
                out.append(ground_record(task, line, sc.cpu(), obj, model_name))
            pending.clear()

        try:
            with torch.inference_mode():
                for j, line in enumerate(lines):
                    ids, labels = chatml_ids_labels([line["conversations"][0], line["conversations"][1]], tokenizer)
                    if int((ids == IMAGE_TOKEN_INDEX).sum()) != 1:
                        raise ValueError("exactly one <image> placeholder per prompt")
                    loc = ((labels[0] >= gt_ids[0]) & (labels[0] <= gt_ids[-1])).nonzero().flatten()
                    if loc.numel() != 1:
                        raise ValueError("exactly one <ground> label token expected (llava_qwen.py:280-281)")
                    raw, _ = loader.get(j)
                    images, coords = pipe.device_inputs(SceneSample(input_ids=ids[0], raw=raw, key=line["video"]))
                    objects = torch.tensor(video_processor.scan2obj[line["video"]]).to(torch.float16)       # the driver's .half()
                    scores = eng.ground_scores(ids[0], int(loc[0]), images, coords, objects.to(eng.device))
                    pending.append((line, scores, objects))
                    if len(pending) >= sync_every:
                        flush()
                flush()
        finally:
            loader.close()
        if stats is not None:
            stats.update({"host_thread_seconds": dict(loader.stage_seconds), "questions": len(lines), "wall_seconds": time.time() - t0})
        return out

    return run


def answer_fn_for(task, model, tokenizer, image_processor, video_processor, model_name, a, pool=None, times=None, stats=None):
    if task in GROUND_TASKS:
        return ground_answer_fn(task, model, tokenizer, image_processor, video_processor, model_name, a.max_frame_num,
                                workers=(0 if a.loader_workers < 0 else a.loader_workers or None), pool=pool, stats=stats)
    kw = dict(max_frame_num=a.max_frame_num, max_new_tokens=a.max_new_tokens, reuse_scenes=a.reuse_scenes, times=times,
              pipeline=not a.no_pipeline, group_size=a.decode_group, workers=(0 if a.loader_workers < 0 else a.loader_workers or None),
              pool=pool, stats=stats)
    if task == "scan2cap":                      # model_scan2cap.py:137-139, 199-212
        if a.reuse_scenes:
            raise NotImplementedError("--reuse-scenes: the Scan2Cap prompt's <coord> rows differ per question inside the prefix's reach")

        def record(line, text):
            r = E.make_record(line, text, model_name)
            r["gt_response"] = line.get("annotations", [line["conversations"][1]["value"]])
            r["scene"] = line["video"]
            return r
        kw.update(record_fn=record, box_input_fn=lambda l: None if l["box_input"] is None else [float(v) for v in l["box_input"][:3]],
                  skip_fn=lambda l: l["box_input"] is None)
    return E.model_answer_fn(model, tokenizer, image_processor, video_processor, model_name, **kw)


def main(argv=None, task=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    if task is None:
        ap.add_argument("--task", required=True, choices=GEN_TASKS + GROUND_TASKS)
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
    ap.add_argument("--reuse-scenes", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true")
    ap.add_argument("--decode-group", type=int, default=16)
    ap.add_argument("--loader-workers", type=int, default=0)
    ap.add_argument("--shard", choices=("stride", "scene"), default=None)
    ap.add_argument("--n_gpu", type=int, default=None, help="the reference's flag: started plainly with N > 1, launch N ranks of this runner")
    a = ap.parse_args(argv)
    fixed_task = task
    task = task or a.task
    if os.path.exists(a.answer_file):
        print(f"The {a.answer_file} already exists!!!")
        return 0
    import sys
    child_argv = list(sys.argv[1:] if argv is None else argv)
    if fixed_task is not None:                       # (the per-task entry points fix the task: their ranks run this module with --task)
        child_argv = ["--task", fixed_task] + child_argv
    rc = E.self_launch(a.n_gpu, "v3d.eval_3d", child_argv)
    if rc is not None:
        return rc
    with open(os.path.expanduser(a.question_file)) as f:
        questions = json.load(f)[: a.test_size]
    pool = None
    if a.loader_workers >= 0 and not (task in GEN_TASKS and a.no_pipeline):
        from . import frame_io
        pool = frame_io.make_pool(a.loader_workers or E.default_workers())          # forked before this process touches the GPU
    rank, world, dev, gather_dev = E.rank_setup(a.n_gpu)
    from llava.video_utils import VideoProcessor
    tokenizer, model, image_processor, name = E.load_model(os.path.expanduser(a.model_path), a.overwrite_cfg)
    vp = VideoProcessor(video_folder=a.video_folder, annotation_dir=a.embodiedscan_folder, frame_sampling_strategy=a.frame_sampling_strategy,
                        metadata_dir=a.metadata_folder)
    times = []
    fn = answer_fn_for(task, model, tokenizer, image_processor, vp, name, a, pool=pool, times=times)
    records = E.evaluate(questions, fn, rank, world, gather_dev, shard=a.shard or ("scene" if a.reuse_scenes else "stride"))
    if rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(a.answer_file)), exist_ok=True)
        with open(a.answer_file, "w") as f:
            for r in records:
                f.write(json.dumps(r) + "\n")
        if times:
            print(f"time: {sum(times) / len(times)}")
    if pool is not None:
        pool.shutdown(wait=True, cancel_futures=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
