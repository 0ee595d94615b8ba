#!/bin/bash
# Round 6: the backward's chain on a high-priority stream (TQDNE_BWD_HIPRIO=1) against the default (chain on the current stream, weight gradients
# on the side stream, equal priorities), and against everything on one stream (TQDNE_BWD_STREAMS=1); same box, alternated.
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06i; mkdir -p $OUT
run() { name=$1; shift
  echo "== $name rep=$rep" >> $OUT/ab.txt
  env "$@" python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parts'])" >> $OUT/ab.txt
}
for rep in 1 2 3; do
  run default TQDNE_BWD_HIPRIO=0
  run hiprio TQDNE_BWD_HIPRIO=1
  run one_stream TQDNE_BWD_STREAMS=1
done
cat $OUT/ab.txt
