#!/bin/bash
# Round-5 ablation gate of the forward-conv levers: per-layer times and the 18-step sample with
#   hip               the default library
#   ncb4              64-channel x 64-position waves, ONE weight buffer (TQ_EXP_NCB4, round 4)
#   ncb4_norefill     the same with the weights never replaced (wrong numerics) = the bound of any weight-prefetch scheme for that tile
#   nostage / now     the default tile without staging in the loop / with L1-resident weights (wrong numerics): bounds of the other levers
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r05b; mkdir -p $OUT
L=$PWD/tqdne_amd/lib
for rep in 1 2; do
for v in hip ncb4 ncb4_norefill; do
  lib=$L/libtqdne_$v.so; [ $v = hip ] && lib=$L/libtqdne_hip.so
  ncb=0; [ $v != hip ] && ncb=1
  echo "== $v rep $rep" >> $OUT/ncb4_layers.txt
  TQDNE_HIP_LIB=$lib TQDNE_CONV_NCB4=$ncb python3 tools/experiments/ncb4_ab.py 64 >> $OUT/ncb4_layers.txt 2>&1
done
done
for rep in 1 2; do
for v in hip ncb4 ncb4_norefill nostage now; do
  lib=$L/libtqdne_$v.so; [ $v = hip ] && lib=$L/libtqdne_hip.so
  ncb=0; case $v in ncb4*) ncb=1;; esac
  echo "== $v rep $rep" >> $OUT/sample_gate.txt
  TQDNE_HIP_LIB=$lib TQDNE_CONV_NCB4=$ncb python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print({k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('parts'), d.get('roofline', {}).get('frac'))
" >> $OUT/sample_gate.txt
done
done
