#!/bin/bash
# usage (GPU box, repo root): tools/pmc_attention.sh <tag>   -- SQ counters of the attention forward kernel (B=64, T=512, 4 x 64)
tag=$1
repo=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $repo/gpurun_out/pmc_$tag -- python3 $repo/tools/bench_attention.py 64 512 4 64 5 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("$repo/gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention_fwd2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$repo/gpurun_out/pmc_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention_fwd2" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {k: sum(v) / len(v) for k, v in acc.items()}
out["kernel_us_under_profiler"] = sum(dur) / max(1, len(dur))
print(json.dumps(out, indent=1))
json.dump(out, open("$repo/gpurun_out/pmc_$tag.json", "w"), indent=1)
PY
