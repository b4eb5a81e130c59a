#!/bin/bash
# bench.py against every library build under raytracer-go_amd/csrc/variants (A/B of compile-time knobs)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sweep_variants.txt
: > $OUT
for so in raytracer-go_amd/csrc/librtgo_hip.so raytracer-go_amd/csrc/variants/*.so; do
  echo "$so" >> $OUT
  timeout -k 10 120 python scripts/bench_variant.py $so --steps 3 --warmup 1 --no-cpu-baseline 2>>gpurun_out/sweep_variants.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_frame'], d['roofline']['alone'])" >> $OUT || exit 1
done
cat $OUT
