#!/bin/bash
# Round 6: the default tiles of the fp16 + MX-fp6 scheme fold their own GroupNorm behind the first chunk's loads (TQDNE_GN_FOLD=1) against the
# tq_gn_finalize launches (=0), same library, same box, alternated: bench.py --mode sample / train, B = 64 (4 lanes), and one lane.
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06m; mkdir -p $OUT
for rep in 1 2 3; do for on in 0 1; do
  for mode in sample train; do
    echo "== fold=$on mode=$mode rep=$rep" >> $OUT/ab.txt
    TQDNE_GN_FOLD=$on python3 bench.py --mode $mode --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parts'])" >> $OUT/ab.txt
  done
  echo "== fold=$on mode=sample 1 lane rep=$rep" >> $OUT/ab.txt
  TQDNE_SAMPLER_LANES=1 TQDNE_GN_FOLD=$on python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parts'])" >> $OUT/ab.txt
done; done
for on in 0 1; do TQDNE_GN_FOLD=$on python3 tools/layer_table.py 64 4096 5 > $OUT/layers_inference_b64_fold_$on.txt 2>/dev/null; done
cat $OUT/ab.txt
