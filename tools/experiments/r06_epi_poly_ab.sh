#!/bin/bash
# Round 6, same-box A/B (alternated) of
#   (1) the batched epilogue loads (residual of the forward convs, x / accumulate of the data gradients) against the in-loop loads of
#       rounds 1-5 (library built with -DTQ_ABL_EPI_SERIAL: tqdne_amd/lib/libtq_epi_serial.so, selected with TQDNE_HIP_LIB);
#   (2) Upsample trained in its two-phase k = 3 form against the k = 5 launches over the upsampled gather (TQDNE_POLYPHASE_TRAIN=0).
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06c; mkdir -p $OUT
SER=$PWD/tqdne_amd/lib/libtq_epi_serial.so
summ='
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, d.get("parts"))
'
run() {  # name, env...
  name=$1; shift
  for mode in sample train; do
    echo "== $name mode=$mode rep=$rep" >> $OUT/ab.txt
    env "$@" python3 bench.py --mode $mode --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "$summ" >> $OUT/ab.txt
  done
}
for rep in 1 2 3; do
  run base_serial_k5up TQDNE_HIP_LIB=$SER TQDNE_POLYPHASE_TRAIN=0
  run batched_k5up TQDNE_POLYPHASE_TRAIN=0
  run batched_polytrain TQDNE_POLYPHASE_TRAIN=1
done
TQDNE_HIP_LIB=$SER TQDNE_POLYPHASE_TRAIN=0 python3 tools/layer_table.py 64 4096 5 > $OUT/layers_inference_b64_serial.txt 2>/dev/null
python3 tools/layer_table.py 64 4096 5 > $OUT/layers_inference_b64_batched.txt 2>/dev/null
TQDNE_HIP_LIB=$SER TQDNE_POLYPHASE_TRAIN=0 python3 tools/layer_table.py 64 4096 3 train > $OUT/layers_train_b64_serial_k5up.txt 2>/dev/null
TQDNE_POLYPHASE_TRAIN=0 python3 tools/layer_table.py 64 4096 3 train > $OUT/layers_train_b64_batched_k5up.txt 2>/dev/null
python3 tools/layer_table.py 64 4096 3 train > $OUT/layers_train_b64_batched_polytrain.txt 2>/dev/null
cat $OUT/ab.txt
