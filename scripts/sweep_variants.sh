#!/bin/bash
# bench.py against every library build under raytracer-go_amd/csrc/variants (A/B of compile-time knobs)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sweep_variants.txt
: > $OUT
for so in raytracer-go_amd/csrc/librtgo_hip.so raytracer-go_amd/csrc/variants/*.so; do
  echo "$so" >> $OUT
  RTGO_LIB=$PWD/$so timeout -k 10 150 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline 2>>gpurun_out/sweep_variants.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['kernel_ms_per_frame_per_gpu'].items()}, 'alone', round(d['roofline']['alone']['kernel_s_per_frame']*1e3,1), 'one-lane frame', round(d['roofline']['alone']['frame_ms_one_lane'],1))" >> $OUT || exit 1
done
cat $OUT
