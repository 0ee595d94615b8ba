#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile.sh <outdir-name> <bench args...>
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
repo=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $repo/bench.py --no-cpu-baseline "$@" > $out.log 2>&1 || true
python3 $repo/tools/prof_summary.py $out 24
