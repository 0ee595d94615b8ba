#!/bin/bash
# Round-5 bounds of the forward conv on the 18-step sample (B = 64, 4 lanes) and per layer -- timing ablations, wrong numerics:
#   nomma    no matrix instructions (operands still loaded / read): the memory + staging + LDS side alone
#   noepi    no output stores
#   mmaonly  no staging in the loop, weights L1-resident, no output stores: the MFMA + LDS-read stream alone
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r05d; mkdir -p $OUT
L=$PWD/tqdne_amd/lib
for rep in 1 2; do
for v in hip nomma noepi mmaonly nostage; do
  echo "== $v rep $rep" >> $OUT/sample_bounds.txt
  TQDNE_HIP_LIB=$L/libtqdne_$v.so python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print({k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('parts'))
" >> $OUT/sample_bounds.txt
done
done
for v in hip nomma noepi mmaonly nostage now; do
  echo "== $v" >> $OUT/layer_bounds.txt
  TQDNE_HIP_LIB=$L/libtqdne_$v.so python3 tools/experiments/ncb4_ab.py 64 2>/dev/null >> $OUT/layer_bounds.txt
done
