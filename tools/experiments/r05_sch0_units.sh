cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r05e; mkdir -p $OUT
python -m pytest tests/test_hip_unet.py -q -m gpu -k "kv_planes or bf16x3_moves" > $OUT/t.log 2>&1; tail -n 3 $OUT/t.log
python -m pytest tests/test_hip_bwd.py tests/test_autoencoder_step.py tests/test_trainer_gpu.py -q -m gpu > $OUT/t2.log 2>&1; tail -n 3 $OUT/t2.log
L=$PWD/tqdne_amd/lib
for rep in 1 2; do
for v in hip sch0_u2 sch0_u1; do
  echo "== $v rep $rep" >> $OUT/sch0_units.txt
  TQDNE_HIP_LIB=$L/libtqdne_$v.so python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print({k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('parts'))
" >> $OUT/sch0_units.txt
done
done
cat $OUT/sch0_units.txt
for rep in 1 2; do python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print('train', d['ms_per_step'], d.get('parts'))
"; done
