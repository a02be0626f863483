"""Which at::native (torch) kernels run INSIDE the scene loop?  Reads a rocprofv3 --kernel-trace csv of `bench.py --no-extras` and lists, by name,
the kernels that are not v3d's and start after the first LLM prefill attention launch (everything before it is weight / input synthesis and
engine construction).   python tools/native_in_timed.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = next(i for i, r in enumerate(rows) if "attn_prefill_kernel" in r["Kernel_Name"] and "128" in r["Kernel_Name"])
# the scene loop starts with the first scene's ViT: walk back to the first patchify before that launch
start = max(i for i in range(first) if "patchify" in rows[i]["Kernel_Name"])
inside = collections.Counter()
dur = collections.Counter()
n_scenes = sum(1 for r in rows[start:] if "patchify" in r["Kernel_Name"])
for r in rows[start:]:
    n = r["Kernel_Name"]
    if "v3d::" in n or n.startswith("v3d") or "preprocess_rgb" in n or "rmsnorm_decode" in n:
        continue
    inside[n[:150]] += 1
    dur[n[:150]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"{n_scenes} scenes after the loop's first kernel; kernels that are not v3d's inside it:")
for n, c in inside.most_common():
    print(f"  {c:6d} launches ({c / n_scenes:6.2f} per scene) {dur[n] / 1e3 / n_scenes:8.1f} us per scene  {n}")
