#!/bin/bash
# Round 6, second look at TQDNE_GN_FOLD (consumer-side GroupNorm fold on the default fp16 + MX-fp6 tiles): the whole bench step, B = 64, 4 lanes,
# five alternations on one box.
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06ac; mkdir -p $OUT
for rep in 1 2 3 4 5; do for on in 0 1; do
  echo "== fold=$on rep=$rep" >> $OUT/ab.txt
  TQDNE_GN_FOLD=$on python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-tables --no-other-configs --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parts']['train_ms'], d['parts']['sample_ms'])" >> $OUT/ab.txt
done; done
cat $OUT/ab.txt
