#!/bin/bash
# Round 6: phase stamps (tools/stamps.py, -DTQ_STAMP build in tqdne_amd/lib/stamp.so) of small-tile forward convs at the shapes of a
# launch-bound plan (tiny UNet, B = 4): where do the 12-16 us of such a launch go?
#   python -c "from tqdne_amd import _build; _build.build(force=True, extra_flags=('-DTQ_STAMP',), out_name='stamp.so')"
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06u
out=gpurun_out/r06u/stamps_small.txt
{
for fold in 0 1; do
echo "== 128->128 k5 T512 B4 small tile fold=$fold (64 workgroups)"; TQ_TTILE=32 TQ_FOLD=$fold TQ_NWG=64 python3 tools/stamps.py 128 0 128 5 512 4 5
echo "== 128->128 k5 T1024 B4 small tile fold=$fold (128 workgroups)"; TQ_TTILE=32 TQ_FOLD=$fold TQ_NWG=128 python3 tools/stamps.py 128 0 128 5 1024 4 5
echo "== 64->64 k5 T2048 B4 small tile fold=$fold (256 workgroups)"; TQ_TTILE=32 TQ_FOLD=$fold TQ_NWG=256 python3 tools/stamps.py 64 0 64 5 2048 4 5
echo "== 32->32 k5 T4096 B4 small tile fold=$fold (512 workgroups)"; TQ_TTILE=32 TQ_FOLD=$fold TQ_NWG=512 python3 tools/stamps.py 32 0 32 5 4096 4 5
done
} 2>&1 | grep -v amdgpu.ids > $out
cat $out
