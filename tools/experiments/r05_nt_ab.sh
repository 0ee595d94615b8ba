#!/bin/bash
# Round 5: non-temporal activation loads (nt1) / + non-temporal output stores (nt3) in the scheme-2 convs against the default: per layer, the
# 18-step sample, and the PMC traffic of the dominant conv (FETCH_SIZE / WRITE_SIZE, separate passes)
cd ${GRAFT_REPO_ROOT:-.}   # (needs tools/experiments/nontemporal.patch applied and libtqdne_nt1.so / libtqdne_nt3.so built with -DTQ_EXP_NT=1 / =3)
OUT=$PWD/gpurun_out/r05n; mkdir -p $OUT
L=$PWD/tqdne_amd/lib
for rep in 1 2; do
for v in hip nt1 nt3; do
  echo "== $v rep $rep" >> $OUT/nt_layers.txt
  TQDNE_HIP_LIB=$L/libtqdne_$v.so python3 tools/experiments/ncb4_ab.py 64 2>/dev/null >> $OUT/nt_layers.txt
done
done
for rep in 1 2; do
for v in hip nt1 nt3; do
  echo "== $v rep $rep" >> $OUT/nt_sample.txt
  TQDNE_HIP_LIB=$L/libtqdne_$v.so python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print({k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('parts'))
" >> $OUT/nt_sample.txt
done
done
repo=$PWD
cd /tmp && export TMPDIR=/tmp
for v in hip nt1 nt3; do
  export TQDNE_HIP_LIB=$L/libtqdne_$v.so
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$v -- python3 $repo/tools/bench_one.py 256 256 256 5 1024 64 5 > /dev/null 2>&1
  done
  python3 - <<PY >> $OUT/nt_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv1d_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(x) / len(x) for k, x in acc.items()}
print("$v", {k: round(x, 1) for k, x in out.items()}, "read MB (x2 corrected)", round(out.get("FETCH_SIZE", 0) * 2048 / 1e6, 1), "write MB", round(out.get("WRITE_SIZE", 0) * 1024 / 1e6, 1))
PY
  rm -rf $OUT/pmc_$v
done
cd $repo
cat $OUT/nt_layers.txt $OUT/nt_sample.txt $OUT/nt_pmc.txt
