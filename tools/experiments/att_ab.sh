#!/bin/bash
# same-box A/B of the attention forward kernel in several builds of the library: tools/experiments/att_ab.sh <a.so> <b.so> ... (names under tqdne_amd/lib/)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for L in "$@"; do
    echo -n "$L: "
    TQDNE_HIP_LIB=$PWD/tqdne_amd/lib/$L python3 tools/experiments/att_time.py 2>/dev/null | tail -1
  done
done
