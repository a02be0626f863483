#!/bin/bash
# Round-4 profiles (run on the GPU box from the repo root: bash tools/profile_r04.sh):
#   1. rocprofv3 --kernel-trace --stats of the default bench command's headline part (ONE prefill stream: a row = the kernel's own time)
#   2. SQ counters of the gate/up GEMM and the causal prefill attention (own pass, kernel-trace only)
#   3. FETCH_SIZE / WRITE_SIZE (separate passes) of the prefill attention / gate-up GEMM and of the r04 decode kernels
set -e
OUT=gpurun_out/prof_r04
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-extras --prefill-streams 1 > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/bench_kernel_stats.csv"
echo "stats done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d "$OUT/sq" -- python3 tools/one_gemm_attn.py > "$OUT/sq.log" 2>&1
echo "sq done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex 'gemm256pp|attn_prefill' --output-format csv -d "$OUT/ga_$C" -- python3 tools/one_gemm_attn.py > "$OUT/ga_$C.log" 2>&1
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex 'linear_decode_mfma|attn_prefill16|attn_decode' --output-format csv -d "$OUT/dec_$C" -- python3 tools/one_decode_rows.py > "$OUT/dec_$C.log" 2>&1
done
echo "pmc done"
find "$OUT" -name '*counter_collection.csv' | xargs ls -la
