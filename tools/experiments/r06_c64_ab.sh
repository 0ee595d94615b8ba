#!/bin/bash
# Round 6, lever (a) of the round-5 verdict: the 64-channel ResBlock convs (and the data gradients with 64 / 192 produced channels) in
# fp16 + MX-fp6 on the 64-channel x 128-position tile against bf16x3 (TQDNE_CONV_MX6_C64=0 / TQDNE_DGRAD_MX6_C64=0: the round-5
# behaviour, same library).  Same box, alternated: the 18-step sample (4 lanes), the train step, the whole step; per-layer tables.
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06b; mkdir -p $OUT
python3 -m tqdne_amd._build > /dev/null || exit 1
summ='
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, d.get("parts"))
'
for rep in 1 2 3; do
for on in 0 1; do
  for mode in sample train step; do
    echo "== c64=$on mode=$mode rep=$rep" >> $OUT/ab.txt
    TQDNE_CONV_MX6_C64=$on TQDNE_DGRAD_MX6_C64=$on python3 bench.py --mode $mode --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "$summ" >> $OUT/ab.txt
  done
done
done
for on in 0 1; do
  TQDNE_CONV_MX6_C64=$on TQDNE_DGRAD_MX6_C64=$on python3 tools/layer_table.py 64 4096 5 > $OUT/layers_inference_b64_c64_$on.txt 2>/dev/null
  TQDNE_CONV_MX6_C64=$on TQDNE_DGRAD_MX6_C64=$on python3 tools/layer_table.py 64 4096 3 train > $OUT/layers_train_b64_c64_$on.txt 2>/dev/null
done
cat $OUT/ab.txt
