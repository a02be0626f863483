#!/bin/bash
# Round-3 profiles (run on the GPU box from the repo root: bash tools/profile_r03.sh):
#   1. rocprofv3 --kernel-trace --stats of the default bench command's headline part  -> gpurun_out/prof_r03/stats
#      (--prefill-streams 1: with the default two, kernels of neighbouring scenes share the chip and every duration in the
#       trace is inflated by its neighbour - 180 ms of kernel time per 100 ms scene; one stream gives each kernel's own time)
#   2. SQ counters of the gate/up GEMM and the causal prefill attention (own pass, kernel-trace only)  -> gpurun_out/prof_r03/sq
#   3. FETCH_SIZE / WRITE_SIZE of the 3-D position kernels + the new resize kernel (separate passes)   -> gpurun_out/prof_r03/FETCH_SIZE, WRITE_SIZE
set -e
OUT=gpurun_out/prof_r03
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-extras --prefill-streams 1 > "$OUT/stats.log" 2>&1
if [ "$1" = "stats" ]; then echo "stats done"; find "$OUT/stats" -name '*kernel_stats.csv' | xargs ls -la; exit 0; fi
echo "stats done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d "$OUT/sq" -- python3 tools/one_gemm_attn.py > "$OUT/sq.log" 2>&1
echo "sq done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex 'visual_tokens|coord_pool|unproject_sampled|resize_bicubic' --output-format csv -d "$OUT/$C" -- \
    python3 tools/one_resize.py > "$OUT/$C.log" 2>&1
done
echo "pmc done"
find "$OUT" -name '*.csv' | xargs ls -la
