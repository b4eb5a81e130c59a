#!/bin/bash
# per-kernel time (rocprofv3 kernel trace, one lane) of every library build under raytracer-go_amd/csrc/variants
cd /tmp && export TMPDIR=/tmp
for so in $GRAFT_REPO_ROOT/raytracer-go_amd/csrc/variants/*.so; do
  n=$(basename $so .so)
  RTGO_LANES=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vt_$n -- python3 $GRAFT_REPO_ROOT/scripts/bench_variant.py $so --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  echo "== $n"
  python3 - "$GRAFT_REPO_ROOT/gpurun_out/vt_$n" <<'PY'
import csv,glob,sys,re
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    m=re.search(r"(k_[a-z_]+)",r["Name"])
    if m and float(r["Percentage"])>1: print("  %-16s %8.2f ms per frame"%(m.group(1), float(r["TotalDurationNs"])/1e6/3))
PY
done
