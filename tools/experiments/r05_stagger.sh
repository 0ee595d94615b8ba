#!/bin/bash
# Round 5: workgroups of a conv launch started apart (-DTQ_EXP_STAGGER, TQDNE_CONV_STAGGER = s_sleep units per hash step, 8 steps):
# per layer (B = 64, one stream) and on the 18-step sample (4 lanes)
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r05i; mkdir -p $OUT
L=$PWD/tqdne_amd/lib/libtqdne_stagger.so
for s in 0 4 8 16 32; do
  echo "== stagger $s (max $((s*7*64)) cycles)" >> $OUT/stagger_layers.txt
  TQDNE_HIP_LIB=$L TQDNE_CONV_STAGGER=$s python3 tools/experiments/ncb4_ab.py 64 2>/dev/null >> $OUT/stagger_layers.txt
done
for rep in 1 2; do
for s in 0 8 16 32; do
  echo "== stagger $s rep $rep" >> $OUT/stagger_sample.txt
  TQDNE_HIP_LIB=$L TQDNE_CONV_STAGGER=$s python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print({k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('parts'))
" >> $OUT/stagger_sample.txt
done
done
cat $OUT/stagger_layers.txt $OUT/stagger_sample.txt
