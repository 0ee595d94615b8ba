cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do for s in 2 1; do
echo "== tiny B=4 streams=$s rep=$rep"
TQDNE_BWD_STREAMS=$s python3 bench.py --config tiny --batch 4 --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parts'])"
echo "== paper B=16 streams=$s rep=$rep"
TQDNE_BWD_STREAMS=$s python3 bench.py --config paper --batch 16 --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-tables --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parts'])"
done; done
