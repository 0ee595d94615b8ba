#!/bin/bash
# usage (GPU box, repo root): tools/pmc_step.sh <tag> <bench.py args...>
# per-kernel-name averages of occupancy use (waves per SIMD), MFMA-busy share, L2 hit rate and LDS bank-conflict share over a
# short bench.py run (kernels are serialised by the counter collection: standalone behaviour of every launch)
tag=$1; shift
repo=$GRAFT_REPO_ROOT
# the profiled program must stay ONE process that nothing re-launches: bench.py --gpus N > 1 starts torch.distributed.run and that
# launcher its rank workers, all from a process the profiler's preloaded library has already GPU-initialised (forbidden on this pool)
for a in "$@"; do
  if [ "$prev" = "--gpus" ] && [ "$a" != "1" ]; then echo "pmc_step.sh: --gpus $a refused (single process only under rocprofv3 --pmc)"; exit 2; fi
  case "$a" in --gpus=1) ;; --gpus=*) echo "pmc_step.sh: $a refused (single process only under rocprofv3 --pmc)"; exit 2;; esac
  prev=$a
done
# build BEFORE entering rocprofv3: a stale library would make __graft_entry__.build() start hipcc children inside the profiled process
python3 -m tqdne_amd._build > /dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $repo/gpurun_out/pmcs_$tag -- python3 $repo/bench.py "$@" --no-tables --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]
for f in glob.glob("$repo/gpurun_out/pmcs_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$repo/gpurun_out/pmcs_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "GRBM_GUI_ACTIVE" not in m or "SQ_WAVE_CYCLES" not in m:
        continue
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    tot_us = sum(dur[k]) / 2.0   # (two passes)
    rows.append((tot_us, k, len(dur[k]) // 2, sum(dur[k]) / max(1, len(dur[k])), m["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024),
                 m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc, m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)),
                 m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 0))))
rows.sort(reverse=True)
print("%-72s %6s %9s %8s %9s %8s %7s %8s" % ("kernel", "n", "total_us", "avg_us", "waves/SIMD", "mfma", "L2hit", "LDSconfl"))
for t, k, n, a, w, mf, l2, lc in rows[:40]:
    print("%-72s %6d %9.0f %8.1f %9.2f %8.2f %7.2f %8.2f" % (k, n, t, a, w, mf, l2, lc))
PY
rm -rf $repo/gpurun_out/pmcs_$tag
