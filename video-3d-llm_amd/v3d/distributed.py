"""Scene-level data parallelism for eval: one process per GPU, `questions[rank::world]`
(the reference's own sharding, llava/eval/model_scanqa.py:245), no collective on the data path, and ONE
variable-length byte gather of the answer records to rank 0 at the end (replaces Ray + file lock,
model_scanqa.py:194-206, 242-247).  Works over RCCL ("nccl", GPU tensors) and gloo (CPU tensors)."""
import json

import torch
import torch.distributed as dist


def shard(items, rank, world):
    return items[rank::world]


def launch_ranks(n, argv, script=None, module=None, json_only=False):
    """Start N ranks of `script` (a path) or `module` (python -m) on THIS node as a child `torch.distributed.run` and wait for them - what
    the reference's drivers do themselves with `ray.init(); eval_model.remote(questions[i::n_gpu])` (model_scanqa.py:242-247).  Called by a
    parent that has not touched the GPU (never an exec of a process that has; `import torch` does not initialise it); rendezvous on
    127.0.0.1 and a free port; no retry - a failed child is a failed run and its exit code is returned.  json_only: let only lines that
    start with "{" through on stdout (the bench's one result line), everything else goes to stderr."""
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    target = ["-m", module] if module else [os.path.abspath(script)]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + target + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    pkg_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # so that `-m v3d....` resolves in the ranks
    env["PYTHONPATH"] = pkg_root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    print("%d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for ln in proc.stdout:
        if not json_only:
            sys.stdout.write(ln)
            sys.stdout.flush()
        elif ln.lstrip().startswith("{"):
            lines.append(ln.rstrip("\n"))
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    for ln in lines[-1:]:
        print(ln, flush=True)
    if json_only and rc == 0 and not lines:
        print("the ranks exited cleanly but printed no result line", file=sys.stderr)
        return 1
    return rc


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


def shard_indices(n, rank, world):
    """Indices of `questions[rank::world]` (the reference's sharding, model_scanqa.py:245)."""
    return list(range(rank, n, world))


def shard_scene_indices(keys, rank, world):
    """Scene-aware sharding: whole scenes per rank, as contiguous blocks of scenes (in order of first appearance)
    balanced by question count, so that per-scene reuse (one prefill per scene) survives data parallelism - with stride
    sharding a scene's consecutive questions land on `world` different ranks and every rank prefills every scene.
    keys[i] = scene of question i.  Returns this rank's question indices (ascending within a scene, scenes in order).  Every rank
    computes the same partition."""
    order, by_scene = [], {}
    for i, k in enumerate(keys):
        if k not in by_scene:
            by_scene[k] = []
            order.append(k)
        by_scene[k].append(i)
    n, out, done, r = len(keys), [], 0, 0
    for k in order:
        # the scene goes to the rank whose quota [r n / world, (r + 1) n / world) the MIDDLE of its questions falls into: a rank's load
        # is then within half a scene of n / world on either side (by the first question, a short scene in front of a long one dragged
        # the long one along and could leave the last ranks empty)
        while r < world - 1 and done + len(by_scene[k]) / 2 >= (r + 1) * n / world:
            r += 1
        if r == rank:
            out += by_scene[k]
        done += len(by_scene[k])
    return out


def gather_indexed(records, indices, device, dst=0):
    """records[j] answers question indices[j]; rank dst gets every record ordered by question index (any sharding)."""
    if len(records) != len(indices):
        raise ValueError("one record per index")
    blobs = gather_bytes("\n".join(json.dumps([int(i), r]) for i, r in zip(indices, records)).encode(), device, dst)
    if blobs is None:
        return None
    pairs = [json.loads(l) for b in blobs for l in b.decode().split("\n") if l]
    pairs.sort(key=lambda p: p[0])
    if [p[0] for p in pairs] != list(range(len(pairs))):
        raise RuntimeError("gathered records do not cover the question list exactly once")
    return [p[1] for p in pairs]


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


# ------------------------------------------------------------------------------ training collectives (configs[4])
# ZeRO-2 as scripts/zero2.json:22-34 configures DeepSpeed: gradients leave by reduce-scatter in buckets of 2e8 elements
# (`reduce_scatter: true`, `reduce_bucket_size: 2e8`), the updated parameter partitions come back by all-gather in buckets of the
# same size (`allgather_bucket_size: 2e8`).  On the 8-GPU xGMI mesh a 2e8-element bf16 bucket is 400 MB = 50 MB per peer; RCCL's
# reduce_scatter over a fully connected mesh can use all 7 links, a ring is bound by one.

ZERO2_BUCKET_ELEMS = int(2e8)


def partition_bounds(numel, world):
    """Equal contiguous partitions of a flat buffer (padded to a multiple of world): [(begin, end)] per rank."""
    per = (numel + world - 1) // world
    return [(min(r * per, numel), min((r + 1) * per, numel)) for r in range(world)], per


def reduce_scatter_grads(flat_grad, bucket_elems=ZERO2_BUCKET_ELEMS, average=True, algorithm="ring"):
    """flat_grad: this rank's flat gradient buffer [numel] (any float dtype).  Returns this rank's partition of the SUM (or
    mean) over ranks, [per] (zero-padded at the tail), exchanged bucket by bucket so that no more than `bucket_elems` elements are
    in flight.  algorithm:
      "ring"    per bucket ONE reduce-scatter of the backend (RCCL) - or, where the backend has none (gloo), an all-reduce of the
                bucket followed by the local slice (same result, CPU tests only);
      "direct"  per bucket ONE all-to-all (rank r receives every peer's r-th chunk) and a local sum in rank order: on the
                point-to-point xGMI mesh every peer pair has its own link, so the one-shot exchange moves a bucket over all seven
                links at once where a ring is bound by one; the sum order is fixed, so the result does not
                depend on arrival order.  (Not yet run on an 8-GPU node: CPU test with gloo, world 2.)"""
    if algorithm not in ("ring", "direct"):
        raise ValueError(f"unknown reduce-scatter algorithm {algorithm!r}")
    world, rank = dist.get_world_size(), dist.get_rank()
    numel = flat_grad.numel()
    bounds, per = partition_bounds(numel, world)
    out = torch.zeros(per, dtype=flat_grad.dtype, device=flat_grad.device)
    has_rs = dist.get_backend() != "gloo"
    # a bucket covers the same sub-range [o, o + chunk) of every rank's partition, so each bucket is a complete reduce-scatter
    chunk = max(1, min(per, bucket_elems // world))
    for o in range(0, per, chunk):
        n = min(chunk, per - o)
        send = torch.zeros(world * n, dtype=flat_grad.dtype, device=flat_grad.device)
        for r in range(world):
            b = r * per + o
            e = min(b + n, numel)
            if e > b:
                send[r * n: r * n + (e - b)] = flat_grad[b:e]
        if algorithm == "direct":
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send)                      # recv[r * n : (r + 1) * n] = rank r's chunk for this rank
            out[o: o + n] = recv.view(world, n).sum(dim=0)
        elif has_rs:
            dist.reduce_scatter_tensor(out[o: o + n], send, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(send, op=dist.ReduceOp.SUM)
            out[o: o + n] = send[rank * n: (rank + 1) * n]
    if average:
        out /= world
    return out


def all_gather_params(partition, numel, bucket_elems=ZERO2_BUCKET_ELEMS):
    """Inverse exchange after the optimizer step: every rank contributes its updated partition [per]; returns the flat
    parameter buffer [numel], gathered in buckets of at most `bucket_elems` elements."""
    world = dist.get_world_size()
    per = partition.numel()
    full = torch.empty(world * per, dtype=partition.dtype, device=partition.device)
    chunk = max(1, min(per, bucket_elems // world))
    for o in range(0, per, chunk):
        n = min(chunk, per - o)
        parts = [torch.empty(n, dtype=partition.dtype, device=partition.device) for _ in range(world)]
        dist.all_gather(parts, partition[o: o + n].contiguous())
        for r in range(world):
            full[r * per + o: r * per + o + n] = parts[r]
    return full[:numel]


def all_gather_params_into(flat, rank, per, bucket_elems=ZERO2_BUCKET_ELEMS):
    """all_gather_params in place: `flat` [world * per] holds this rank's updated partition at [rank * per, (rank + 1) * per) and
    receives every other rank's, bucket by bucket, without a second buffer of the model's size (ZeroAdamW's parameters are views of
    `flat`)."""
    world = dist.get_world_size()
    chunk = max(1, min(per, bucket_elems // world))
    for o in range(0, per, chunk):
        n = min(chunk, per - o)
        outs = [flat[r * per + o: r * per + o + n] for r in range(world)]
        dist.all_gather(outs, outs[rank].clone() if world > 1 else outs[rank])
    return flat
