"""One scene prefill, then N answer batches of 16 questions (60 rows + 17 tokens each) - the `cached_questions` workload of bench.py, for
rocprofv3 --kernel-trace --stats:   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/cached_q_profile.py [batches] [group]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
import bench  # noqa: E402
from v3d import ops  # noqa: E402
from v3d.engine import Engine, EngineConfig, random_state_dict  # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 4
group = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
dtype = torch.bfloat16
cfg = EngineConfig()
sd = random_state_dict(cfg, dtype, dev, seed=0)
eng = Engine(cfg, sd, dtype=dtype, device=dev, max_frames=bench.FRAMES)
sc = bench.synth_inputs(dev, dtype, seed=1000)
g = torch.Generator(device=dev).manual_seed(4242)
prefix = sc["input_ids"][: bench.TEXT_PRE + 1]
qs = [[torch.randint(0, 151000, (bench.TEXT_POST,), generator=g, device=dev) for _ in range(group)] for _ in range(n_batches)]
coords = ops.unproject_sampled(sc["depth"], sc["K"], sc["P"], 384, dtype)
images = ops.preprocess_rgb(sc["frames"], dtype)
eng.prefill_scene(prefix, images, coords)
eng.answer_group(qs[0], max_new_tokens=bench.NEW_TOKENS)
torch.cuda.synchronize()
t0 = time.perf_counter()
for q in qs:
    eng.answer_group(q, max_new_tokens=bench.NEW_TOKENS)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n_batches} batches x {group} questions: {dt / (n_batches * group) * 1e3:.2f} ms per question = {n_batches * group / dt:.1f} questions/s", flush=True)
