#!/bin/bash
# usage (GPU box, repo root): tools/pmc_forward.sh <tag>  -- per-kernel HBM bytes (FETCH_SIZE x2, WRITE_SIZE) of UNet forwards, B=64, one lane
tag=$1
repo=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export TQDNE_SAMPLER_LANES=1
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $repo/gpurun_out/pmcf_$tag -- python3 $repo/bench.py --no-cpu-baseline --no-train --steps 1 --warmup 0 --sample-steps 2 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$repo/gpurun_out/pmcf_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
rows = []
for k in acc:
    rd = acc[k].get("FETCH_SIZE", 0) * 1024 * 2 / 1e6
    wr = acc[k].get("WRITE_SIZE", 0) * 1024 / 1e6
    n = max(cnt[k].values())
    rows.append((rd + wr, k, n, rd, wr))
tot = sum(r[0] for r in rows)
print(f"total {tot:.0f} MB over the run")
for t, k, n, rd, wr in sorted(rows, reverse=True)[:22]:
    print(f"{k:58s} n={n:4d} read {rd:9.1f} MB write {wr:9.1f} MB  per launch {rd/n:7.1f} / {wr/n:7.1f}")
PY
