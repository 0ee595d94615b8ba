# per-layer small tile (TQDNE_SMALL_TILE_WGS): cfg3's one-lane B = 16 sample and the 4-lane B = 64 headline, same box
for w in 0 64 128; do
  echo "== TQDNE_SMALL_TILE_WGS=$w"
  TQDNE_SMALL_TILE_WGS=$w timeout 300 python tools/cfg3_lanes.py 2>/dev/null | grep "lanes=1 graph=False"
  TQDNE_SMALL_TILE_WGS=$w timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tables --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value'],1), 'wf/s', {k:round(v,2) for k,v in d['parts'].items()})"
done
