"""BASELINE configs[4] on one GPU, the training sample alone (bench.py's train_config4 extra without the rest of the bench):
   python tools/time_train_step.py            (V3D_TRAIN_WGRAD_STREAM=0: weight gradients on the backward's own stream)"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-llm_amd"))
import bench
torch.cuda.set_device(0)
r = bench.measure_train_step(torch.device("cuda:0"))
print(json.dumps({k: v for k, v in r.items() if not isinstance(v, str)}))
