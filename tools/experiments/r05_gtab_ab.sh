#!/bin/bash
# Round 5: GroupNorm coefficient table of the scheme-2 convs loaded together with the first chunk (default) vs before it (libtqdne_before_gtab.so
# = the previous build): per layer and on the 18-step sample, alternated on one box
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r05k; mkdir -p $OUT
L=$PWD/tqdne_amd/lib
for rep in 1 2; do
for v in before_gtab hip; do
  echo "== $v rep $rep" >> $OUT/gtab_layers.txt
  TQDNE_HIP_LIB=$L/libtqdne_$v.so python3 tools/experiments/ncb4_ab.py 64 2>/dev/null >> $OUT/gtab_layers.txt
done
done
for rep in 1 2 3; do
for v in before_gtab hip; do
  echo "== $v rep $rep" >> $OUT/gtab_sample.txt
  TQDNE_HIP_LIB=$L/libtqdne_$v.so python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print({k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('parts'))
" >> $OUT/gtab_sample.txt
done
done
cat $OUT/gtab_layers.txt $OUT/gtab_sample.txt
