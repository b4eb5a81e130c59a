#!/bin/bash
# rocprofv3 passes over the headline bench (run on the GPU box through gpurun).  Usage: profile_r03.sh <tag>
# Pass 1: kernel trace + stats.  Passes 2-5: PMC counters, each in a run of its own (no tracing).  Then the
# calibration micro-benchmarks on the same box.  scripts/make_bounds.py turns the output into profiles/r03/bounds.json.
set -e
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_units" -- $BENCH > "$OUT/bench_pmc_units.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_pmc_fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- $BENCH > "$OUT/bench_pmc_l2.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_mix" -- $BENCH > "$OUT/bench_pmc_mix.log" 2>&1
# write side of every kernel: store instructions, bytes written to HBM, write requests of the L2 to the fabric
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_FLAT TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum --output-format csv -d "$OUT/pmc_wr" -- $BENCH > "$OUT/bench_pmc_wr.log" 2>&1
timeout -k 10 120 $GRAFT_REPO_ROOT/build/calib > "$OUT/calib.json"
timeout -k 10 120 $GRAFT_REPO_ROOT/build/calib_mix > "$OUT/calib_mix.txt"
for p in units fetch l2 mix wr; do python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py "$OUT/pmc_$p" > "$OUT/pmc_$p.txt"; done
cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
head -12 "$OUT/kernel_stats.csv"
grep k_shadow "$OUT"/pmc_*.txt
tail -1 "$OUT/bench_trace.log" | cut -c1-300
