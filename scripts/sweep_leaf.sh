#!/bin/bash
# k_shadow time against the SAH leaf size (RTGO_LEAF_SIZE is read at scene creation) + loop statistics of the default
set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sweep_leaf.txt
: > $OUT
for L in ${LEAVES:-3 4 5 6 8}; do
  echo "leaf $L" >> $OUT
  RTGO_LEAF_SIZE=$L timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['alone']['kernel_s_per_frame'], d['config']['tree_blocks'])" >> $OUT
done

cat $OUT
