#!/bin/bash
# Round 6, lever (c): small-tile convs fold their own GroupNorm (TqConvDesc.gn_fold) against the tq_gn_finalize launches (TQDNE_GN_FOLD_SMALL=0),
# same library, same box, alternated: cfg0 (tiny UNet, B = 4: every ResBlock conv is a small-tile launch), the paper UNet at B = 4, and cfg3's B = 16
# plan (small tile on its T = 512 level only).
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06g; mkdir -p $OUT
summ='
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, d.get("parts"))
'
for rep in 1 2 3; do
for on in 0 1; do
  echo "== fold=$on tiny B=4 rep=$rep" >> $OUT/ab.txt
  TQDNE_GN_FOLD_SMALL=$on python3 bench.py --config tiny --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "$summ" >> $OUT/ab.txt
  echo "== fold=$on paper B=4 rep=$rep" >> $OUT/ab.txt
  TQDNE_GN_FOLD_SMALL=$on python3 bench.py --config paper --batch 4 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "$summ" >> $OUT/ab.txt
done
done
for on in 0 1; do
  LAYER_TABLE_CONFIG=tiny TQDNE_GN_FOLD_SMALL=$on python3 tools/layer_table.py 4 4096 10 > $OUT/layers_tiny_b4_fold_$on.txt 2>/dev/null
done
cat $OUT/ab.txt
