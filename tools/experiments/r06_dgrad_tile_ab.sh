#!/bin/bash
# Round 6: 256-channel data-gradient outputs as two co-resident 4-wave workgroups (TQDNE_DGRAD_TILE256=0) against the 8-wave tile, same library,
# same box, alternated: does a co-resident neighbour's MFMA stream cover the chain epilogue?
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06h; mkdir -p $OUT
for rep in 1 2 3; do
for t in 1 0; do
  echo "== dgrad_tile256=$t rep=$rep" >> $OUT/ab.txt
  TQDNE_DGRAD_TILE256=$t python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parts'])" >> $OUT/ab.txt
done
done
for t in 1 0; do TQDNE_DGRAD_TILE256=$t python3 tools/layer_table.py 64 4096 3 train > $OUT/layers_train_b64_tile256_$t.txt 2>/dev/null; done
cat $OUT/ab.txt
