"""One whole training sample on one GPU (bench.py's `train_config4` measurement on its own, e.g. under rocprofv3)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
import bench
print(json.dumps(bench.measure_train_step(torch.device("cuda:0"), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 2)))
