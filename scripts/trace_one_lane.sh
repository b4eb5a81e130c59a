#!/bin/bash
# rocprofv3 kernel trace of one bench run on ONE lane (no overlapping launches): per-kernel times of the build
# Usage: trace_one_lane.sh <tag> [env assignments...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export RTGO_LANES=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench.log" 2>&1 || exit 1
cp $(find "$OUT" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
head -14 "$OUT/kernel_stats.csv" | cut -c1-160
