#!/bin/bash
# Round-2 first GPU look: calibration micro-benchmarks, the counter list, one TA/TCP PMC pass of the bench.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout -k 10 120 $GRAFT_REPO_ROOT/build/calib > $OUT/calib.json
cat $OUT/calib.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 60 rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
grep -c . $OUT/counters_list.txt || true
bash $GRAFT_REPO_ROOT/scripts/pmc_pass.sh ta "TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" > $OUT/pmc_ta.txt 2>&1 || { tail -20 $OUT/pmc_ta/bench.log; tail -5 $OUT/pmc_ta.txt; exit 1; }
grep k_shadow $OUT/pmc_ta.txt
