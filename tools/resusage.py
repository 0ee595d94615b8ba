"""List per-kernel register / scratch usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split(" [")[0]] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    print(f'{name:70s} vgpr {r.get("VGPRs"):>4} agpr {r.get("AGPRs"):>4} scratch {r.get("ScratchSize"):>4} spill {r.get("VGPRs Spill"):>3} occ {r.get("Occupancy")} lds {r.get("LDS Size")}')
