#!/bin/bash
# phase stamps (tools/stamps.py, -DTQ_STAMP build in tqdne_amd/lib/stamp.so) of the forward conv on a few paper-config layers
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
out=gpurun_out/${1:-stamps}.txt
{
echo "== 512->256 k5 T1024 B64"; TQ_NWG=512 python3 tools/stamps.py 512 0 256 5 1024 64 5
echo "== 256->256 k5 T1024 B64"; TQ_NWG=512 python3 tools/stamps.py 256 0 256 5 1024 64 5
echo "== 128->128 k5 T2048 B64"; TQ_NWG=1024 python3 tools/stamps.py 128 0 128 5 2048 64 5
echo "== 64->64 k5 T4096 B64"; TQ_NWG=2048 python3 tools/stamps.py 64 0 64 5 4096 64 5
} 2>&1 | grep -v amdgpu.ids > $out
cat $out
