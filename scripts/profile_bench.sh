#!/bin/bash
# rocprofv3 passes over the headline bench (run on the GPU box through gpurun). Usage: profile_bench.sh <tag>
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, each in a run of its own (no tracing).
set -e
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_pmc_fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- $BENCH > "$OUT/bench_pmc_l2.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/bench_pmc_sq.log" 2>&1
find "$OUT" -name "*.csv" | head -50
