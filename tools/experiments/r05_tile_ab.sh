#!/bin/bash
# Round 5: the 256-channel conv tile as one 8-wave workgroup per CU (default) vs two co-resident 4-wave workgroups (TQDNE_CONV_TILE256=0),
# on the 18-step sample with 4 / 1 / 2 lanes (co-resident workgroups of different lanes are out of phase) and on the train step
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r05g; mkdir -p $OUT
run() {
  python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print({k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('parts'))
"
}
for rep in 1 2; do
for t in 1 0; do
for lanes in 4 2 1; do
  echo "== tile256=$t lanes=$lanes rep $rep" >> $OUT/tile_ab.txt
  TQDNE_CONV_TILE256=$t TQDNE_SAMPLER_LANES=$lanes run --mode sample >> $OUT/tile_ab.txt
done
echo "== tile256=$t train rep $rep" >> $OUT/tile_ab.txt
TQDNE_CONV_TILE256=$t run --mode train >> $OUT/tile_ab.txt
done
done
cat $OUT/tile_ab.txt
