"""rocprofv3 kernel_stats.csv -> the markdown table of profiles/README.md: python tools/stats_table.py <kernel_stats.csv> <scenes>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
scenes = float(sys.argv[2])
tot = 0.0
print("| kernel | calls | ms / scene | avg us |\n|---|---|---|---|")
for r in rows:
    name = r["Name"]
    if "v3d::" not in name:
        continue
    short = re.sub(r"\(.*", "", name.replace("void ", "").replace("v3d::", ""))
    ms = float(r["TotalDurationNs"]) / 1e6 / scenes
    tot += ms
    if ms >= 0.005:
        print(f"| `{short}` | {r['Calls']} | {ms:.2f} | {float(r['AverageNs']) / 1e3:.1f} |")
print(f"\nSum of the path's kernels: {tot:.1f} ms per scene")
