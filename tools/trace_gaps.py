"""Reads a rocprofv3 --kernel-trace CSV and prints, for the longest contiguous busy region at the end of the run
(the timed scenes), wall span, union of kernel-busy time, idle gaps and the biggest gaps with the kernels around them."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows), key=lambda e: e[0])
tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
t_end = max(e[1] for e in ev)
ev = [e for e in ev if e[0] >= t_end - tail_ms * 1e6]
span = (max(e[1] for e in ev) - ev[0][0]) / 1e6
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
prev = ev[0]
for e in ev[1:]:
    if e[0] > cur_e:
        busy += cur_e - cur_s
        gaps.append((e[0] - cur_e, prev[2], e[2]))
        cur_s, cur_e = e[0], e[1]
    else:
        cur_e = max(cur_e, e[1])
    if e[1] >= cur_e:
        prev = e
busy += cur_e - cur_s
print(f"last {tail_ms:.0f} ms of the trace: span {span:.1f} ms, kernels busy {busy/1e6:.1f} ms, idle {span - busy/1e6:.1f} ms in {len(gaps)} gaps")
import collections
hist = collections.Counter()
for g, a, b in gaps:
    hist[min(int(g / 1000) // 5 * 5, 100)] += g
print("idle time by gap length (us bucket -> total ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"  {g/1e3:8.1f} us  after {a}  before {b}")
