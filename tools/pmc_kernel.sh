#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernel.sh <tag> <kernel-name substring> <python script> [args...]
# PMC counters (separate passes) of the launches whose kernel name contains the substring, averaged, into gpurun_out/pmc_<tag>.json
tag=$1; pat=$2; script=$3; shift 3
repo=$GRAFT_REPO_ROOT
[ -f "$script" ] || script=$repo/$script
script=$(readlink -f $script)
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $repo/gpurun_out/pmc_$tag -- python3 $script "$@" > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("$repo/gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$pat" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$repo/gpurun_out/pmc_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$pat" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {k: sum(v) / len(v) for k, v in acc.items()}
out["kernel_us_under_profiler"] = sum(dur) / max(1, len(dur))
out["launches"] = len(dur)
print(json.dumps(out, indent=1))
json.dump(out, open("$repo/gpurun_out/pmc_$tag.json", "w"), indent=1)
PY
rm -rf $repo/gpurun_out/pmc_$tag
