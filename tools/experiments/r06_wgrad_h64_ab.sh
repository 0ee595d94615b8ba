#!/bin/bash
# Round 6: weight gradients of the 64-output-channel convs in the four-wave shared-tile form (TQDNE_WGRAD_H64=1, default) against the
# half-filled 128-channel tile (=0): per launch, then the B = 64 train step alternated x 3 on one box, then the per-launch table.
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06n; mkdir -p $OUT
for on in 0 1; do
  echo "== per launch, TQDNE_WGRAD_H64=$on" >> $OUT/ab.txt
  TQDNE_WGRAD_H64=$on python3 tools/experiments/r06_wgrad_h64.py >> $OUT/ab.txt 2>&1
done
for s in 256 1024; do
  echo "== per launch, TQDNE_WGRAD_H64=1 TQDNE_WGRAD_SLOTS=$s" >> $OUT/ab.txt
  TQDNE_WGRAD_H64=1 TQDNE_WGRAD_SLOTS=$s python3 tools/experiments/r06_wgrad_h64.py >> $OUT/ab.txt 2>&1
done
for rep in 1 2 3; do for on in 0 1; do
  echo "== train step, TQDNE_WGRAD_H64=$on rep=$rep" >> $OUT/ab.txt
  TQDNE_WGRAD_H64=$on python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parts'])" >> $OUT/ab.txt
done; done
for on in 0 1; do TQDNE_WGRAD_H64=$on python3 tools/layer_table.py 64 4096 5 train > $OUT/layers_train_b64_h64_$on.txt 2>/dev/null; done
cat $OUT/ab.txt
