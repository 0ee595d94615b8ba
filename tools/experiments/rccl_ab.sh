# A/B of the forced one-rank RCCL exchange (TQDNE_BENCH_FORCE_RCCL=1) against the plain N = 1 run, same box
run() { # label, env..., command
  label=$1; shift
  env "$@" > gpurun_out/r04_u_$label.json 2> gpurun_out/r04_u_$label.err
  python - "$label" <<'PY'
import json,sys
l=sys.argv[1]
for ln in open(f"gpurun_out/r04_u_{l}.json"):
    if ln.startswith("{"):
        d=json.loads(ln); ex=d.get("gradient_exchange") or {}
        print(l, "ms/step", round(d["ms_per_step"],2), {k:round(v,2) for k,v in d["parts"].items()}, "after", ex.get("train_ms_exchange_after_backward"), "under", ex.get("train_ms_exchange_under_backward"))
PY
}
A="--steps 5 --warmup 2 --no-cpu-baseline --no-tables --no-other-configs"
run step_plain X=1 timeout 300 python bench.py $A
run step_forced_reserve_first TQDNE_BENCH_FORCE_RCCL=1 timeout 300 python bench.py $A
run step_forced_rccl_first TQDNE_BENCH_FORCE_RCCL=1 TQDNE_BENCH_RESERVE_STREAMS_FIRST=0 timeout 300 python bench.py $A
run train_forced_reserve_first TQDNE_BENCH_FORCE_RCCL=1 timeout 300 python bench.py --mode train $A
