#!/bin/bash
# PMC pass over the calibration micro-benchmarks: what TA_BUSY / TCP_TOTAL_CACHE_ACCESSES read on known gather patterns.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_calib
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES TD_TD_BUSY_sum --output-format csv -d "$OUT" -- $GRAFT_REPO_ROOT/build/calib > "$OUT/calib.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:40])
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows):
    print(k, {n: f"{v:.4g}" for n, v in sorted(rows[k].items())})
PY
