#!/bin/bash
# same-box A/B of two builds of the library (box-to-box spread is +-3-5 %): tools/ab_libs.sh <a.so> <b.so> [bench args]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do
    echo "== $L (rep $rep)"
    TQDNE_HIP_LIB=$PWD/tqdne_amd/lib/$L python3 bench.py "$@" 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print({k: d.get(k) for k in ('value', 'ms_per_step', 'ms_per_step_median')}, d.get('roofline', {}).get('parts'), d.get('parts'))
"
  done
done
