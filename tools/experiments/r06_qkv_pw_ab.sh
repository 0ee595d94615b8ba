#!/bin/bash
# Round 6: the qkv projection in the channel-tiled form (TQDNE_QKV_PW=0) against the input-stationary one (default above 128 workgroups):
# B = 64 sample on 4 lanes of 16 and on one lane, cfg3-like one-lane plan of 16 samples (layer table), alternated on one box.
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06y; mkdir -p $OUT
for rep in 1 2 3; do for pw in 1 0; do
  echo "== qkv_pw=$pw 4 lanes rep=$rep" >> $OUT/ab.txt
  TQDNE_QKV_PW=$pw python3 bench.py --mode sample --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $OUT/ab.txt
  echo "== qkv_pw=$pw 1 lane rep=$rep" >> $OUT/ab.txt
  TQDNE_SAMPLER_LANES=1 TQDNE_QKV_PW=$pw python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $OUT/ab.txt
done; done
cat $OUT/ab.txt
