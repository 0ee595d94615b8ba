#!/bin/bash
# Same-box A/B of the round-5 tree (a worktree of commit 5eadfca under _r05/, its own library) against this tree: the default bench step
# alternated three times on one box (boxes differ by +-3-5 %, so only this kind of comparison resolves a round's changes), then the other
# BASELINE configurations of both trees once.  The round-6 train step includes the EMA (+0.05 ms), the round-5 one does not.
#   git worktree add _r05 5eadfca && (cd _r05 && python -c "from tqdne_amd import _build; _build.build()")
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out/r06ab; mkdir -p $OUT
summ='
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, d.get("parts"), (d.get("roofline") or {}).get("frac"))
        for oc in d.get("other_configs") or []:
            print("   ", oc.get("config", "")[:70], round(oc.get("value", 0.0), 1), (oc.get("parts") or {}).get("train_ms"), (oc.get("parts") or {}).get("sample_ms"))
'
for rep in 1 2 3; do
  echo "== tree _r05 rep $rep" >> $OUT/ab.txt
  (cd _r05 && python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null) | python3 -c "$summ" >> $OUT/ab.txt
  echo "== tree . rep $rep" >> $OUT/ab.txt
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-tables --no-other-configs --no-pmc 2>/dev/null | python3 -c "$summ" >> $OUT/ab.txt
done
echo "== tree _r05, other configurations" >> $OUT/ab.txt
(cd _r05 && python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables 2>/dev/null) | python3 -c "$summ" >> $OUT/ab.txt
echo "== tree ., other configurations" >> $OUT/ab.txt
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-tables --no-pmc 2>/dev/null | python3 -c "$summ" >> $OUT/ab.txt
cat $OUT/ab.txt
