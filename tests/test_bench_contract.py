"""bench.py's output contract (the driver parses it): ONE JSON line on stdout with the headline fields, the roofline and CPU-baseline
objects; partial modes are labelled.  Run on the tiny configuration so that it takes seconds."""

import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*args, timeout=600, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout, cwd=ROOT, env=None if env is None else dict(os.environ, **env))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry the JSON line only, got {len(lines)} lines"
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_json_contract_tiny_config():
    d = _run("--config", "tiny", "--batch", "4", "--steps", "2", "--warmup", "1", "--cpu-batch", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "waveforms/s" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and c["one_thread"]["cores"] == 1
    assert set(d["kernel_classes"]) >= {"inference_forward_1lane", "train_forward", "train_backward"}
    assert {"train_ms", "sample_ms", "train_ms_synced", "sample_ms_synced"} <= set(d["parts"]) and "parts_definition" in d
    # same-run parity gate: HIP path vs the CPU oracle child on identical injected inputs, both metrics per checkpoint
    par = d["parity"]
    # (sample_stepK: the sampler's state after K of the 18 steps -- the error growth across the network evaluations)
    assert par["pass"] is True and par["tolerance"] == 1e-3
    assert set(par["checkpoints"]) == {"denoise", "loss", "sample", "sample_step1", "sample_step6", "sample_step12"}
    for c in par["checkpoints"].values():
        assert 0 <= c["max_rel"] < 1e-3 and 0 <= c["allclose_rtol_atol_rms"] <= 1e-3
    # self-evidence of the collective path (a ones tensor summed over RCCL when N > 1), per-rank times, exchange exposure
    assert d["rccl_ranks"] == 1 and d["rank_ms_per_step"]["min"] <= d["rank_ms_per_step"]["max"] and len(d["rank_ms_per_step"]["all"]) == 1
    assert d["gradient_exchange"]["rccl_ranks"] == 1 and d["gradient_exchange"]["hidden_by_overlap_ms"] == 0.0
    # scaling-run self-checks (trivially true at N = 1, but the keys and their shape are what SCALE_rNN.json will carry)
    assert d["rccl_ranks_ok"] is True and d["replicas_equal"] is True
    rep = d["replicas"]
    assert rep["after_broadcast"] and rep["after_timed_steps"] and rep["weights_moved"] is True
    assert len(rep["checksum_after_broadcast"]) == 1 and len(rep["checksum_after_timed_steps"]) == 1


@pytest.mark.timeout(600)
def test_bench_partial_modes_are_labelled():
    d = _run("--config", "tiny", "--batch", "4", "--steps", "1", "--warmup", "1", "--mode", "sample", "--no-cpu-baseline", "--no-tables")
    assert "DEBUG" in d["metric"] and "cpu_baseline" not in d
    d = _run("--config", "tiny", "--batch", "4", "--steps", "1", "--warmup", "1", "--mode", "consistency", "--no-cpu-baseline")
    assert "consistency" in d["metric"] and d["config"]["mode"] == "consistency"


@pytest.mark.timeout(600)
def test_bench_runs_the_rccl_path_over_one_rank_when_forced():
    """TQDNE_BENCH_FORCE_RCCL=1: communicator, rank-0 broadcast, bucketed all-reduce issued from inside the backward sweep on RCCL's
    stream, the exchange-after-backward comparison and the checksum gathers all execute on the 1-GPU box (a sum over one rank)."""
    d = _run("--config", "tiny", "--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-tables",
             env={"TQDNE_BENCH_FORCE_RCCL": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["rccl_forced_at_world1"] is True and d["rccl_ranks"] == 1 and d["rccl_ranks_ok"] is True
    assert d["replicas_equal"] is True and d["replicas"]["weights_moved"] is True
    ex = d["gradient_exchange"]
    assert ex["overlap"] is True and len(ex["buckets_elems"]) >= 1 and ex["hidden_by_overlap_ms"] is not None
    # (round 5: the range-guard pair rides in the last bucket of the step instead of a collective of its own)
    assert ex["tail_words"] == 2
    assert d["value"] > 0


@pytest.mark.timeout(900)
def test_bench_two_ranks_dry_run_on_one_gpu():
    """The WHOLE N > 1 program of bench.py on the 1-GPU box (round-5 verdict item 2): ``python bench.py --gpus 2`` self-launches two
    ranks through torch.distributed.run; with TQDNE_BENCH_BACKEND=gloo + TQDNE_BENCH_SHARE_DEVICE=1 they share cuda:0 and exchange over
    gloo (host staged).  Per-rank shards and seeds, rank-0 broadcast, bucketed exchange issued from inside the backward sweep, the range
    flags in the last bucket, max-over-ranks timing, per-rank times, replica checksums, ONE aggregated JSON line on stdout, exit code 0."""
    d = _run("--gpus", "2", "--config", "tiny", "--batch", "4", "--steps", "2", "--warmup", "1", "--no-tables",
             env={"TQDNE_BENCH_BACKEND": "gloo", "TQDNE_BENCH_SHARE_DEVICE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["rccl_ranks_ok"] is True and d["scaling"] == "weak"
    assert "gloo" in d["collective_backend"] and d["shared_device"] is True and "DRY RUN" in d["metric"]
    assert d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]      # whole-job aggregate over both ranks
    assert len(d["rank_ms_per_step"]["all"]) == 2 and d["rank_ms_per_step"]["max"] <= d["ms_per_step"] * 1.0001
    rep = d["replicas"]
    assert d["replicas_equal"] is True and rep["weights_moved"] is True
    assert len(rep["checksum_after_broadcast"]) == 2 and len(set(rep["checksum_after_timed_steps"])) == 1
    ex = d["gradient_exchange"]
    assert ex["rccl_ranks"] == 2 and ex["overlap"] is True and len(ex["buckets_elems"]) >= 1 and ex["tail_words"] == 2
    assert ex["hidden_by_overlap_ms"] is not None
    assert "cpu_baseline" not in d and "other_configs" not in d   # (rank 0 at N = 1 only)
    assert d["train_wf_s"] > 0 and d["sample_wf_s"] > 0
