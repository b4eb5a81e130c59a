#!/bin/bash
# bench.py for several leaf sizes of the SAH tree (RTGO_LEAF_SIZE is read at scene creation)
cd $GRAFT_REPO_ROOT
for ls in ${LEAVES:-4 5 6 7 8}; do
  echo "RTGO_LEAF_SIZE=$ls"
  RTGO_LEAF_SIZE=$ls timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>>gpurun_out/sweep_leaf.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['kernel_ms_per_frame_per_gpu'].items()}, 'alone', round(d['roofline']['alone']['kernel_s_per_frame']*1e3,1), 'one-lane frame', round(d['roofline']['alone']['frame_ms_one_lane'],1))" || exit 1
done
