#!/bin/bash
# Same-box A/B of the round-4 tree (a worktree of commit 47d7f9c under _r04/, its own library) against this tree: the default bench step,
# alternated three times on one box (boxes differ by +-3-5 %, so only this kind of comparison resolves a round's changes)
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r05u; mkdir -p $OUT
for rep in 1 2 3; do
for tree in _r04 .; do
  echo "== tree $tree rep $rep" >> $OUT/ab.txt
  (cd $tree && python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null) | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print({k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('parts'), (d.get('roofline') or {}).get('frac'))
" >> $OUT/ab.txt
done
done
cat $OUT/ab.txt
