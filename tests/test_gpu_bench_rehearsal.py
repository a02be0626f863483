"""`python bench.py --gpus 2` as the driver starts it (no launcher environment), on ONE card: V3D_BENCH_REHEARSAL=1 puts both ranks on cuda:0 with the
gloo backend, so the N > 1 code path of the bench - self-launch of the ranks, the reference's stride sharding of the question list
(model_scanqa.py:245), the scene pipeline per rank, ONE variable-length gather of the answer records, barrier + max-over-ranks timing, rank 0's single
line - executes with the real engine.  Its throughput means nothing (two ranks share a GPU) and the line says so; what is held to here is the plumbing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_two_ranks_on_one_card():
    env = dict(os.environ, V3D_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "V3D_BENCH_DRY"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                           # rank 0's line, once
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]      # whole-job scenes / max-over-ranks time
    assert "rehearsal" in d["data"].lower()
