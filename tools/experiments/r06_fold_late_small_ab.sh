#!/bin/bash
# Round 6: the small tile of the fp16 + MX-fp6 scheme folds its GroupNorm BEHIND its first staging / weight loads (default) against in
# FRONT of them (library built with -DTQ_ABL_FOLD_EARLY: tqdne_amd/lib/libtq_fold_early.so, selected with TQDNE_HIP_LIB), cfg0 (tiny UNet,
# B = 4) and the paper UNet at B = 4: 18-step sample, alternated x 3 on one box.
#   python -c "from tqdne_amd import _build; _build.build(force=True, extra_flags=('-DTQ_ABL_FOLD_EARLY',), out_name='libtq_fold_early.so')"
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/${R06_OUT:-r06r}; mkdir -p $OUT
EARLY=$PWD/tqdne_amd/lib/libtq_fold_early.so
for rep in 1 2 3; do
  echo "== early rep=$rep" >> $OUT/ab.txt
  TQDNE_HIP_LIB=$EARLY python3 tools/experiments/r06_cfg0_gaps.py run 2>/dev/null | grep cfg0 >> $OUT/ab.txt
  echo "== late rep=$rep" >> $OUT/ab.txt
  python3 tools/experiments/r06_cfg0_gaps.py run 2>/dev/null | grep cfg0 >> $OUT/ab.txt
done
TQDNE_HIP_LIB=$EARLY LAYER_TABLE_CONFIG=tiny python3 tools/layer_table.py 4 4096 5 > $OUT/layers_cfg0_b4_early.txt 2>/dev/null
LAYER_TABLE_CONFIG=tiny python3 tools/layer_table.py 4 4096 5 > $OUT/layers_cfg0_b4_late.txt 2>/dev/null
TQDNE_HIP_LIB=$EARLY python3 tools/layer_table.py 4 4096 5 > $OUT/layers_paper_b4_early.txt 2>/dev/null
python3 tools/layer_table.py 4 4096 5 > $OUT/layers_paper_b4_late.txt 2>/dev/null
cat $OUT/ab.txt; head -1 $OUT/layers_*.txt
