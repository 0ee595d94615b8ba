#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <tag>   -- the rocprofv3 / PMC / per-launch evidence of one round, into gpurun_out/
tag=$1
repo=$GRAFT_REPO_ROOT
mkdir -p $repo/gpurun_out
(cd $repo && python3 -m tqdne_amd._build > /dev/null) || exit 1   # never build inside a profiled process
bash $repo/tools/profile.sh ${tag}_full_step --steps 5 --warmup 2 --no-tables > $repo/gpurun_out/${tag}_full_step_summary.txt 2>&1
TQDNE_SAMPLER_LANES=1 bash $repo/tools/profile.sh ${tag}_sample_only_1lane --mode sample --steps 3 --warmup 1 --no-tables > $repo/gpurun_out/${tag}_sample_only_1lane_summary.txt 2>&1
bash $repo/tools/profile.sh ${tag}_train_only --mode train --steps 10 --warmup 3 --no-tables > $repo/gpurun_out/${tag}_train_only_summary.txt 2>&1
bash $repo/tools/pmc_dominant.sh ${tag} > /dev/null 2>&1
cd $repo
python3 tools/layer_table.py 64 4096 5 > gpurun_out/${tag}_layers_inference_b64.txt 2>/dev/null
python3 tools/layer_table.py 16 4096 5 > gpurun_out/${tag}_layers_inference_b16.txt 2>/dev/null
python3 tools/layer_table.py 64 4096 3 train > gpurun_out/${tag}_layers_train_b64.txt 2>/dev/null
for d in full_step sample_only_1lane train_only; do
  f=$(find gpurun_out/${tag}_$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/${tag}_${d}_kernel_stats.csv
done
# only the summaries travel back (gpurun merges at most 64 MiB): drop the raw traces
rm -rf gpurun_out/${tag}_full_step gpurun_out/${tag}_sample_only_1lane gpurun_out/${tag}_train_only gpurun_out/pmc_${tag}
