#!/bin/bash
# k_shadow time against the level of the shadow-ray hints (RTGO_HINT_UP, read at scene creation; -1 = no hints)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sweep_hint.txt
: > $OUT
for U in -1 0 1 2 3 4; do
  echo "hint up $U" >> $OUT
  RTGO_HINT_UP=$U timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_frame'])" >> $OUT || exit 1
done
cat $OUT
