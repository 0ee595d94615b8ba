#!/bin/bash
# Round 6: gn_fold_sample without the clamped loads of an EMPTY remainder batch (default) against with them (library built with
# -DTQ_ABL_FOLD_REMAINDER: tqdne_amd/lib/libtq_fold_rem.so, selected with TQDNE_HIP_LIB): cfg0 sample (every conv folds), paper UNet B = 64
# one-lane sample (51 tq_gn_finalize launches per evaluation), alternated x 3 on one box.
#   python -c "from tqdne_amd import _build; _build.build(force=True, extra_flags=('-DTQ_ABL_FOLD_REMAINDER',), out_name='libtq_fold_rem.so')"
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/${R06_OUT:-r06v}; mkdir -p $OUT
REM=$PWD/tqdne_amd/lib/libtq_fold_rem.so
for rep in 1 2 3; do
  echo "== with-remainder rep=$rep" >> $OUT/ab.txt
  TQDNE_HIP_LIB=$REM python3 tools/experiments/r06_cfg0_gaps.py run 2>/dev/null | grep cfg0 >> $OUT/ab.txt
  echo "== default rep=$rep" >> $OUT/ab.txt
  python3 tools/experiments/r06_cfg0_gaps.py run 2>/dev/null | grep cfg0 >> $OUT/ab.txt
done
for rep in 1 2; do
  echo "== paper B=64 one lane, with-remainder rep=$rep" >> $OUT/ab.txt
  TQDNE_HIP_LIB=$REM TQDNE_SAMPLER_LANES=1 python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $OUT/ab.txt
  echo "== paper B=64 one lane, default rep=$rep" >> $OUT/ab.txt
  TQDNE_SAMPLER_LANES=1 python3 bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $OUT/ab.txt
done
cat $OUT/ab.txt
