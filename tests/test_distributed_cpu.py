"""World-size-2 gloo test of the eval sharding + single end-of-run gather (the N>1 path of bench.py / eval)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from v3d import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    questions = [{"id": i, "video": f"scene{i // 3:04d}"} for i in range(11)]      # ragged: 6 vs 5
    mine = D.shard(questions, rank, world)
    recs = [{"sample_id": x["id"], "pred_response": "x" * (x["id"] % 4), "rank": rank} for x in mine]
    merged = D.gather_records(recs, torch.device("cpu"))
    empty = D.gather_bytes(b"" if rank == 1 else b"abc", torch.device("cpu"))
    if rank == 0:
        q.put((merged, empty))
    else:
        assert merged is None and empty is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, empty = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [m["sample_id"] for m in merged] == list(range(11))          # original order restored
    assert [m["rank"] for m in merged] == [i % 2 for i in range(11)]
    assert empty == [b"abc", b""]                                        # empty payloads are legal


def test_shard_is_the_reference_stride():
    items = list(range(10))
    assert D.shard(items, 1, 4) == items[1::4]
