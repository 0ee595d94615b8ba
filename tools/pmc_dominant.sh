#!/bin/bash
# usage (GPU box, repo root): tools/pmc_dominant.sh <tag>   -- PMC counters of the dominant conv launch (512->256, k5, T=1024, B=64)
# separate passes: FETCH_SIZE (3 TCC slots), WRITE_SIZE (2), SQ set
tag=$1
repo=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $repo/gpurun_out/pmc_$tag -- python3 $repo/tools/bench_one.py 256 256 256 5 1024 64 5 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("$repo/gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv1d_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$repo/gpurun_out/pmc_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv1d_mfma" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {k: sum(v) / len(v) for k, v in acc.items()}
out["kernel_us_under_profiler"] = sum(dur) / max(1, len(dur))
# gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2; units are KiB
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["hbm_read_bytes_corrected"] = out["FETCH_SIZE"] * 1024 * 2
    out["hbm_write_bytes"] = out["WRITE_SIZE"] * 1024
    out["hbm_traffic_bytes_per_launch"] = out["hbm_read_bytes_corrected"] + out["hbm_write_bytes"]
out["algorithmic_bytes_per_launch"] = 4 * 64 * 1024 * (512 + 256)   # fp32 input (two sources) + output, weights excluded
import hashlib
out["conv1d_mfma_sha16"] = hashlib.sha256(open("$repo/tqdne_amd/csrc/conv1d_kernel.hpp", "rb").read()).hexdigest()[:16]   # bench.py refuses a file recorded for another build of the kernel template
print(json.dumps(out, indent=1))
json.dump(out, open("$repo/gpurun_out/pmc_$tag.json", "w"), indent=1)
PY
