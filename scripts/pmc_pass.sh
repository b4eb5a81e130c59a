#!/bin/bash
# One rocprofv3 PMC pass (counters only, no tracing) over a short bench run.
# Usage: pmc_pass.sh <tag> "<counter list>" [bench args...]
set -e
TAG=$1; CTRS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py "$OUT"
