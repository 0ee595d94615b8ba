# experiment: start sampler lane i a fraction of a network evaluation after lane i - 1 (TQDNE_LANE_STAGGER_US), so that the lanes sit at
# different depths of the UNet (memory-bound 64-channel levels next to MFMA-bound 256-channel ones) instead of marching in step
for s in 0 300 600 1100 2200; do
  TQDNE_LANE_STAGGER_US=$s timeout 300 python bench.py --mode sample --steps 5 --warmup 2 --no-cpu-baseline --no-tables --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger_us', $s, 'ms_per_sample', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['parts'].items()})"
done
