#!/bin/bash
# r04, second session: memory-side traffic of the pipelined decode kernels and the shared-prefix question attention (separate --pmc passes, as
# MI355X_MICROARCH.md's HBM section prescribes), and the kernel stats of the default bench line.  Run on the GPU box from the repo root.
set -e
OUT=gpurun_out/prof_r04b
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex 'linear_decode_mfma|decode_combine|attn_prefill' --output-format csv -d "$OUT/dec_$C" -- python3 tools/one_decode_rows.py > "$OUT/dec_$C.log" 2>&1
done
echo "pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -- python3 bench.py --steps 32 --no-extras > "$OUT/bench.log" 2>&1
find "$OUT/bench" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/bench_kernel_stats.csv"
tail -1 "$OUT/bench.log" | cut -c1-300
