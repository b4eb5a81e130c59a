#!/bin/bash
# A/B of the render lanes on one GPU: frame time at config 3 with 1 lane, 2 lanes (k_shadow launches in turn), 2 lanes free
set -e
for v in "RTGO_LANES=1" "RTGO_LANES=2 RTGO_SHADOW_SERIAL=1" "RTGO_LANES=2 RTGO_SHADOW_SERIAL=0"; do
  echo "== $v"
  env $v timeout -k 10 150 python bench.py --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['kernel_ms_per_frame'], d['roofline']['frac'], d['sec_per_frame_one_shot'])"
done
