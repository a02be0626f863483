"""Host side of the ScanQA eval runner (v3d/eval_scanqa.py) against the I/O contract of llava/eval/model_scanqa.py:
prompt-id structure of preprocess_qwen (:29-80), record schema (:196-204), answer clean-up (:188-192), stride sharding
(:245) + the single gather that replaces Ray + the file lock - the last through a world-size-2 gloo run of `evaluate`
(the same code path the GPU ranks take with RCCL)."""
import json
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import tiny_model_fixture as TM
from v3d import eval_scanqa as E


def _questions(n, scenes=3):
    return [{"id": f"q{i}", "video": f"scannet/scene{i * scenes // n:04d}_00",
             "conversations": [{"from": "human", "value": f"<image>\nt{i + 1} t{i + 2} t{i + 3}"}, {"from": "gpt", "value": f"t{i + 10}"}],
             "metadata": {"dataset": "scanqa", "question_type": "what", "answers": [f"t{i + 10}"]}} for i in range(n)]


def test_chatml_ids_structure(tmp_path):
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(TM.write_checkpoint(str(tmp_path / "ckpt"), TM.load()))
    IM_S, IM_E, NL, SYS, USER, ASSIST = 309, 310, 303, 300, 301, 302
    ids = E.build_prompt_ids(_questions(1)[0], tok)[0].tolist()
    system = [IM_S, SYS, NL, 304, 305, 306, 307, 308, IM_E, NL]                       # <|im_start|>system\nYou are a helpful assistant.<|im_end|>\n
    user = [IM_S, USER, NL, E.IMAGE_TOKEN_INDEX, NL, NL, 1, 2, 3, IM_E, NL]           # the value's own "\n" follows the placeholder's
    assert ids == system + user + [IM_S, ASSIST, NL]                                   # open assistant turn = the generation prompt
    assert ids.count(E.IMAGE_TOKEN_INDEX) == 1
    # a closed assistant turn and a leading non-human turn (dropped, model_scanqa.py:43-44)
    full = E.chatml_ids([{"from": "gpt", "value": "t9"}, {"from": "human", "value": "t4"}, {"from": "gpt", "value": "t5 t6"}], tok)[0].tolist()
    assert full == system + [IM_S, USER, NL, 4, IM_E, NL] + [IM_S, ASSIST, NL, 5, 6, IM_E, NL]


def test_record_schema_and_answer_cleanup():
    line = _questions(1)[0]
    rec = E.make_record(line, "t7", "llava_qwen_tiny")
    assert list(rec) == ["dataset", "sample_id", "prompt", "pred_response", "gt_response", "model_id", "question_type"]
    assert rec["prompt"] == E.EXTRA_PROMPT + line["conversations"][0]["value"] and rec["gt_response"] == "t10" and rec["sample_id"] == "q0"
    assert E.clean_answer("  the chair <|im_end|>\n") == "the chair"
    assert E.clean_answer("the chair") == "the chair"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, shard="stride"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    qs = _questions(11)
    if shard == "scene":                       # interleave the scenes' questions: a scene's questions are NOT consecutive
        qs = qs[0::2] + qs[1::2]

    def answer_fn(lines):                      # stands in for the model: what matters here is who answers what, and the order
        return [E.make_record(l, f"rank{rank}:{l['id']}", "stub") for l in lines]

    recs = E.evaluate(qs, answer_fn, rank, world, torch.device("cpu"), shard=shard)
    if rank == 0:
        with open(out_path, "w") as f:
            for r in recs:
                f.write(json.dumps(r) + "\n")
    else:
        assert recs is None
    dist.barrier()
    dist.destroy_process_group()


def test_evaluate_shards_by_stride_and_collates_in_question_order(tmp_path):
    ctx = mp.get_context("spawn")
    port, out = _free_port(), str(tmp_path / "answers.jsonl")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    recs = [json.loads(l) for l in open(out)]
    assert [r["sample_id"] for r in recs] == [f"q{i}" for i in range(11)]
    assert [r["pred_response"] for r in recs] == [f"rank{i % 2}:q{i}" for i in range(11)]        # questions[rank::world]


def test_evaluate_shards_whole_scenes_and_collates_in_question_order(tmp_path):
    """--shard scene: every scene's questions go to ONE rank (so one prefill per scene survives data parallelism), the
    blocks are balanced by question count, and rank 0 still writes the records in question order."""
    ctx = mp.get_context("spawn")
    port, out = _free_port(), str(tmp_path / "answers.jsonl")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, "scene")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    qs = _questions(11)
    qs = qs[0::2] + qs[1::2]
    recs = [json.loads(l) for l in open(out)]
    assert [r["sample_id"] for r in recs] == [q["id"] for q in qs]
    owner = {}
    for q, r in zip(qs, recs):
        owner.setdefault(q["video"], set()).add(r["pred_response"].split(":")[0])
    assert all(len(v) == 1 for v in owner.values()), owner                   # one rank per scene
    counts = [sum(r["pred_response"].startswith(f"rank{k}:") for r in recs) for k in range(2)]
    assert min(counts) >= 3 and sum(counts) == 11, counts


def test_scene_sharding_partition_properties():
    from v3d import distributed as D
    import random
    rnd = random.Random(1)
    for world in (1, 2, 3, 8):
        for _ in range(20):
            n_scenes = rnd.randint(1, 30)
            keys = [f"s{rnd.randrange(n_scenes)}" for _ in range(rnd.randint(1, 200))]
            parts = [D.shard_scene_indices(keys, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(len(keys)))                                  # a partition
            for r, p in enumerate(parts):
                for s_ in {keys[i] for i in p}:
                    assert all((keys[i] != s_) or (i in p) for i in range(len(keys)))     # whole scenes
            biggest = max(keys.count(k) for k in set(keys))
            assert max(len(p) for p in parts) <= len(keys) / world + biggest          # balanced up to one scene
    # three scenes for three ranks: one each, whatever their sizes (assigning by the first question left the last rank empty)
    assert [D.shard_scene_indices(list("aabccc"), r, 3) for r in range(3)] == [[0, 1], [2], [3, 4, 5]]


def test_prompt_ids_and_labels_equal_the_drivers_own_preprocess_qwen(tmp_path):
    """tests/golden/chatml.json: `preprocess_qwen` of model_scanqa.py:29-80 and model_scanrefer.py:28-80, executed from the drivers' own
    source on the stand-in tokenizer (oracle/gen_golden.py g_chatml).  The runners' restatements give the same ids / labels - the
    tokenizer-DEPENDENT ids stay unpinned (no Qwen2 tokenizer ships with the reference), the structure no longer is."""
    from transformers import AutoTokenizer
    from v3d import eval_3d as E3
    tok = AutoTokenizer.from_pretrained(TM.write_checkpoint(str(tmp_path / "ckpt"), TM.load()))
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chatml.json")))
    for case in g["qa"]:
        assert E.chatml_ids(case["turns"], tok)[0].tolist() == case["ids"]
    for case in g["vg"]:
        ids, labels = E3.chatml_ids_labels(case["turns"], tok)
        assert ids[0].tolist() == case["ids"] and labels[0].tolist() == case["labels"]
        assert E.chatml_ids(case["turns"], tok)[0].tolist() == case["ids"]


def test_grounding_records_follow_the_drivers():
    """model_scanrefer.py:174-189 (arg-max box; the zero-target falls back to the best real box) and model_multi3drefer.py:171-181."""
    from v3d import eval_3d as E3
    line = {"id": "r0", "video": "scannet/scene0000_00", "box": [1, 2, 3, 4, 5, 6], "conversations": [{"from": "human", "value": "<image>\nt1"}, {"from": "gpt", "value": "t318"}],
            "metadata": {"dataset": "scanrefer", "question_type": "unique"}}
    objects = torch.tensor([[0.1, 0.2, 0.3, 1, 1, 1], [1.0, 2.0, 3.0, 0.5, 0.5, 0.5], [3, 3, 3, 2, 2, 2]]).to(torch.float16)
    r = E3.ground_record("scanrefer", line, torch.tensor([0.1, 0.9, 0.3, 0.2]), objects, "m")
    assert list(r) == ["dataset", "sample_id", "prompt", "pred_response", "gt_response", "model_id", "question_type"]
    assert r["pred_response"] == objects[1].tolist() and r["gt_response"] == [1, 2, 3, 4, 5, 6]
    r = E3.ground_record("scanrefer", line, torch.tensor([0.1, 0.2, 0.3, 0.9]), objects, "m")             # the zero-target wins
    assert r["pred_response"] == objects[2].tolist()
    m = E3.ground_record("multi3drefer", line, torch.tensor([0.5, 0.25, 0.125, 0.0]), objects, "m")
    assert list(m) == ["dataset", "sample_id", "prompt", "scores", "objects", "gt_response", "model_id", "question_type"]
    assert m["scores"] == [0.5, 0.25, 0.125, 0.0] and m["objects"] == objects.tolist()
