#!/bin/bash
# Round 6: key split of the first-generation attention kernel (head dimension 128: the tiny config's middle block) for grids far below
# the chip: TQDNE_ATTN_KSPLIT=1 (never) / unset (rule: up to 8 workgroups per query tile while the grid stays <= 256) / 2 / 4, cfg0
# (tiny UNet, B = 4) 18-step sample alternated x 3 on one box, and the per-launch table.
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06s; mkdir -p $OUT
for rep in 1 2 3; do for ks in 1 0 2 4; do
  echo "== ksplit=$ks rep=$rep" >> $OUT/ab.txt
  TQDNE_ATTN_KSPLIT=$ks python3 tools/experiments/r06_cfg0_gaps.py run 2>/dev/null | grep cfg0 >> $OUT/ab.txt
done; done
TQDNE_ATTN_KSPLIT=1 LAYER_TABLE_CONFIG=tiny python3 tools/layer_table.py 4 4096 5 2>/dev/null | grep -i "attention\|^#" > $OUT/layers_cfg0_b4_attention.txt
LAYER_TABLE_CONFIG=tiny python3 tools/layer_table.py 4 4096 5 2>/dev/null | grep -i "attention\|^#" >> $OUT/layers_cfg0_b4_attention.txt
cat $OUT/ab.txt $OUT/layers_cfg0_b4_attention.txt
