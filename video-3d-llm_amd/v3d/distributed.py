"""Scene-level data parallelism for eval: one process per GPU, `questions[rank::world]`
(the reference's own sharding, llava/eval/model_scanqa.py:245), no collective on the data path, and ONE
variable-length byte gather of the answer records to rank 0 at the end (replaces Ray + file lock,
model_scanqa.py:194-206, 242-247).  Works over RCCL ("nccl", GPU tensors) and gloo (CPU tensors)."""
import json

import torch
import torch.distributed as dist


def shard(items, rank, world):
    return items[rank::world]


def gather_bytes(payload: bytes, device, dst=0):
    """Every rank contributes `payload`; rank `dst` gets the list of all payloads (others get None).
    Two collectives: all_gather of 8-byte lengths, then one gather of length-padded byte tensors."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = max(1, int(max(int(s) for s in sizes)))
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if payload:
        buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    out = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return [bytes(o[: int(s)].cpu().numpy().tobytes()) for o, s in zip(out, sizes)]


def gather_records(records, device, dst=0):
    """records: list of JSON-serialisable dicts (the JSONL lines of model_scanqa.py:196-204).  Rank dst
    gets them all re-interleaved into the original question order of stride sharding."""
    blobs = gather_bytes("\n".join(json.dumps(r) for r in records).encode(), device, dst)
    if blobs is None:
        return None
    per_rank = [[json.loads(l) for l in b.decode().split("\n") if l] for b in blobs]
    merged, i = [], 0
    while any(i < len(p) for p in per_rank):
        for p in per_rank:
            if i < len(p):
                merged.append(p[i])
        i += 1
    return merged
