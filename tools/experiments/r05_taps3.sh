#!/bin/bash
# Round 5: the k = 5 convs with 3 of their 5 taps in the MFMA stream (tools/experiments/taps3_ablation.patch, -DTQ_ABL_TAPS3 -> libtqdne_taps3.so;
# wrong numerics): the matrix + LDS-read work of a Winograd F(2, 5) form with today's staging = the optimistic bound of such a cut
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r05p; mkdir -p $OUT
L=$PWD/tqdne_amd/lib
for rep in 1 2; do
for v in hip taps3; do
  echo "== $v rep $rep" >> $OUT/taps3.txt
  TQDNE_HIP_LIB=$L/libtqdne_$v.so python3 tools/experiments/ncb4_ab.py 64 2>/dev/null >> $OUT/taps3.txt
  TQDNE_HIP_LIB=$L/libtqdne_$v.so python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print({k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('parts'))
" >> $OUT/taps3.txt
done
done
cat $OUT/taps3.txt
