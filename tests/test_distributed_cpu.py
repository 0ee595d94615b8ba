"""N > 1 path on CPU (gloo, world_size 2): gradient averaging and batch sharding used by DataParallelTrainer / bench.py.
The data path itself has no collective (independent waveforms); the only exchange is the gradient all-reduce."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tqdne_amd.trainer import allreduce_mean_, shard_batch

    # 1. bucketed mean all-reduce == mean of the per-rank gradients, for sizes that do not divide the bucket
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(100003, generator=g)
    mine = flat.clone()
    allreduce_mean_(flat, world, bucket_elems=4096)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ok1 = torch.allclose(flat, sum(gathered) / world, atol=1e-6)

    # 2. data parallel equivalence on a toy quadratic "model": sharded batch + averaged grads == full-batch grads
    gg = torch.Generator().manual_seed(7)
    w = torch.randn(16, generator=gg)
    batch = {"signal": torch.randn(8, 16, generator=gg), "cond": torch.randn(8, 5, generator=gg)}
    local = shard_batch(batch, rank, world)
    assert local["signal"].shape[0] == 4 and local["cond"].shape[0] == 4

    def grad_of(x):
        ww = w.clone().requires_grad_(True)
        ((x @ ww) ** 2).mean().backward()
        return ww.grad

    gl = grad_of(local["signal"])
    allreduce_mean_(gl, world)
    ok2 = torch.allclose(gl, grad_of(batch["signal"]), atol=1e-6)
    ret[rank] = bool(ok1 and ok2)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gloo_world2_gradient_allreduce_and_sharding():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


# ---------------------------------------------------------------------------------------------------------------------
# DataParallelTrainer itself (not only its helpers), world size 2 over gloo: a stub module with the LightningEDM training surface
# (configure_optimizers / optimizer_params / step_and_backward with the bucket hook) stands in for the HIP-backed module.


class _StubEDM(torch.nn.Module):
    """flat gradient buffer laid out in reverse parameter order; buckets released from inside the "backward" like BackwardPlan.run"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.SiLU(), torch.nn.Linear(32, 32), torch.nn.SiLU(),
                                       torch.nn.Linear(32, 4))
        self.optimizer_params = {"learning_rate": 1e-2, "max_steps": 20, "eta_min": 1e-4}
        ps = list(self.parameters())[::-1]
        self._offs, total = {}, 0
        for p in ps:
            self._offs[id(p)] = total
            total += p.numel()
        self.n_grad = total
        self.flat = torch.zeros(total + 2)   # (+ the two tail words of BackwardPlan's layout, engine_bwd.TAIL_WORDS)
        self.release_order = None  # permutation of the bucket issue order (None: as completed)

    def configure_optimizers(self):
        opt = torch.optim.Adam(self.parameters(), lr=self.optimizer_params["learning_rate"])
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=self.optimizer_params["max_steps"],
                                                         eta_min=self.optimizer_params["eta_min"])
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sch, "interval": "step"}}

    def step_and_backward(self, batch, on_bucket=None, bucket_elems=1 << 20, tail_fill=None):
        self.flat.zero_()
        loss = ((self.net(batch["signal"]) - batch["cond"]) ** 2).mean()
        grads = torch.autograd.grad(loss, list(self.parameters()))
        for p, g in zip(self.parameters(), grads):
            o = self._offs[id(p)]
            self.flat[o:o + p.numel()].copy_(g.reshape(-1))
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        if on_bucket is not None:
            n = self.n_grad
            cuts = [(i, min(i + bucket_elems, n)) for i in range(0, n, bucket_elems)]
            for lo, hi in cuts:
                if hi == n and tail_fill is not None:   # (as BackwardPlan.run: the last bucket takes the caller's tail words along)
                    tail_fill(self.flat[n:n + 2])
                    hi = n + 2
                on_bucket(self.flat[lo:hi])
        return loss.detach(), self.flat[:self.n_grad]


def _trainer_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tqdne_amd.trainer import DataParallelTrainer, shard_batch

    g = torch.Generator().manual_seed(3)
    steps = [{"signal": torch.randn(8, 16, generator=g), "cond": torch.randn(8, 4, generator=g)} for _ in range(3)]

    def run(world_size, **kw):
        m = _StubEDM()
        if world_size > 1 and rank != 0:  # replicas that start different must be overwritten by rank 0's broadcast
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(1.0)
        tr = DataParallelTrainer(m, world_size=world_size, fused_optimizer=False, **kw)
        sizes = []
        for b in steps:
            tr.train_step(shard_batch(b, rank, world_size) if world_size > 1 else b)
            sizes.append(list(tr.last_bucket_sizes))
        return torch.cat([p.detach().reshape(-1) for p in m.parameters()]), sizes

    ref, _ = run(1)                                             # full batch, one rank, torch Adam + cosine LR
    a, sizes_a = run(world, bucket_bytes=4 * 300, overlap=True)   # many small buckets, issued from inside the backward
    b, sizes_b = run(world, bucket_bytes=4 * 300, overlap=False)  # same buckets, issued after the backward
    c, sizes_c = run(world, bucket_bytes=1 << 20, overlap=True)   # one bucket
    ok = (torch.allclose(a, ref, atol=1e-6) and torch.equal(a, b) and torch.allclose(c, ref, atol=1e-6)
          and len(sizes_a[0]) > 3 and len(sizes_c[0]) == 1 and sum(sizes_a[0]) == sum(sizes_c[0]))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gloo_world2_trainer_matches_full_batch_training_for_any_bucketing():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_trainer_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


def _forced_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tqdne_amd.trainer import DataParallelTrainer

    g = torch.Generator().manual_seed(3)
    steps = [{"signal": torch.randn(8, 16, generator=g), "cond": torch.randn(8, 4, generator=g)} for _ in range(3)]

    def run(**kw):
        m = _StubEDM()
        tr = DataParallelTrainer(m, world_size=1, fused_optimizer=False, bucket_bytes=4 * 300, **kw)
        n = 0
        for b in steps:
            tr.train_step(b)
            n += len(tr.last_bucket_sizes)
        return torch.cat([p.detach().reshape(-1) for p in m.parameters()]), n

    ref, n0 = run()
    a, n1 = run(force_exchange=True, overlap=True)
    b, n2 = run(force_exchange=True, overlap=False)
    ret[rank] = bool(n0 == 0 and n1 > 9 and n2 == n1 and torch.equal(a, ref) and torch.equal(b, ref))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_forced_exchange_over_one_rank_is_the_identity():
    """TQDNE_BENCH_FORCE_RCCL's trainer switch: with one rank the broadcast / bucketed all-reduce are issued and change nothing."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_forced_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert ret[0]


def _ipg_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from tqdne_amd.trainer import init_process_group
    init_process_group("gloo", device=None, rank=rank, world_size=world)   # (CPU: nothing to reserve, plain torch call)
    t = torch.ones(1) * (rank + 1)
    dist.all_reduce(t)
    ret[rank] = float(t)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_init_process_group_helper_passes_through_on_cpu():
    """tqdne_amd.trainer.init_process_group = torch's call behind the side-stream reservation a ROCm device needs first"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ipg_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] == ret[1] == 3.0


def test_rank_seeding_gives_distinct_reproducible_streams():
    from tqdne_amd import rng
    seeds = {}
    for rank in (0, 1, 2):
        rng.seed_rank(1234, rank)
        draws = torch.randn(4)
        s1, s2 = rng.next_dropout_seed(), rng.next_dropout_seed()
        rng.seed_rank(1234, rank)
        assert torch.equal(draws, torch.randn(4)) and (s1, s2) == (rng.next_dropout_seed(), rng.next_dropout_seed())
        assert s1 != s2 and 0 <= s1 < 2 ** 64
        seeds[rank] = (s1, s2, tuple(draws.tolist()))
    assert len({v[0] for v in seeds.values()}) == 3 and len({v[2] for v in seeds.values()}) == 3
    rng.seed_rank(0, 0)


# ---------------------------------------------------------------------------------------------------------------- bench.py, N > 1 plumbing
def _bench_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import bench
    n = bench.init_distributed("gloo", rank, world, None, init_timeout_s=60, first_timeout_s=60)
    g = torch.Generator().manual_seed(3)
    params = [torch.randn(1000, generator=g), torch.randn(7, 5, generator=g)]
    same = bench.gather_checksums(bench.replica_checksum(params), world)
    params[1][3, 2] += 1e-7 * (rank + 1)             # one element, one ulp-scale change, on each rank differently
    diff = bench.gather_checksums(bench.replica_checksum(params), world)
    perm = [params[0].flip(0), params[1]]            # same multiset of values in another order
    ret[rank] = dict(n=n, same=len(set(same)) == 1, diff=len(set(diff)) == world,
                     perm=int(bench.replica_checksum(perm)[0]) != int(bench.replica_checksum([params[0], params[1]])[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bench_rendezvous_first_collective_and_replica_checksums_world2():
    """bench.py's N > 1 plumbing on gloo: init + first collective count the ranks; the replica checksum is equal for identical
    weights, differs for a one-element change and for a permutation"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bench_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r] == dict(n=world, same=True, diff=True, perm=True), ret[r]


@pytest.mark.timeout(120)
def test_bench_watchdog_reports_a_hung_rendezvous():
    """rank 0 of a 2-rank job whose peer never shows up: instead of hanging until the driver's clock runs out, the watchdog prints
    ONE diagnostic JSON line (value null, error, stage) and the process exits non-zero"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import bench; bench.init_distributed('gloo', 0, 2, None)" % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), TQDNE_BENCH_INIT_TIMEOUT="4", RANK="0", WORLD_SIZE="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=100)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["value"] is None and d["stage"] == "init_process_group" and "did not complete" in d["error"] and d["n_gpus"] == 2


def _tail_worker(rank, world, port, ret):
    """the range-guard pair folded into the tail of the last gradient bucket (round 5): same predicate on every rank at every step as
    with its own collective, ONE collective fewer per step, and the gradients next to it are untouched"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tqdne_amd.trainer import DataParallelTrainer, shard_batch

    class T(DataParallelTrainer):
        AGREE_EVERY = 4
        flag, auto = None, True

        def _local_range_flag(self):
            return self.flag, self.auto

    g = torch.Generator().manual_seed(3)
    steps = [{"signal": torch.randn(8, 16, generator=g), "cond": torch.randn(8, 4, generator=g)} for _ in range(10)]

    def run(tailed):
        m = _StubEDM()
        tr = T(m, world_size=world, fused_optimizer=False, bucket_bytes=4 * 300)
        tr.flag = torch.zeros(1, dtype=torch.int32)
        tr._tailed = tailed
        calls, log = [], []
        orig = tr._allreduce_async
        tr._allreduce_async = lambda t: (calls.append(t.numel()), orig(t))[1]
        for step, b in enumerate(steps, 1):
            if step == 2 and rank == 1:
                tr.flag.fill_(1)
            if step == 3 and rank == 1:
                tr.flag.zero_(); tr.auto = False
            if step == 5 and rank == 0:
                tr.auto = False
            n0 = len(calls)
            tr.train_step(shard_batch(b, rank, world))
            f = tr.last_skip
            log.append((len(calls) - n0, None if f is None else (float(f[0]) != 0.0, float(f[1]))))
        return torch.cat([p.detach().reshape(-1) for p in m.parameters()]), log

    wa, la = run(True)
    wb, lb = run(False)
    ret[rank] = dict(weights_equal=bool(torch.equal(wa, wb)), tailed=la, separate=lb)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_range_flag_pair_rides_in_the_last_gradient_bucket():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_tail_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0]["tailed"] == ret[1]["tailed"] and ret[0]["separate"] == ret[1]["separate"]
    r = ret[0]
    assert r["weights_equal"]                                            # the gradients next to the tail words are untouched
    assert [x[1] for x in r["tailed"]] == [x[1] for x in r["separate"]]  # the same predicate at every step
    nb = r["tailed"][0][0]
    assert nb >= 3 and r["separate"][0][0] == nb + 1                     # one collective fewer per step while the pair is exchanged
    assert r["tailed"][1][1] == (True, 0.0) and r["tailed"][7][1] == (False, 2.0) and r["tailed"][8][1] is None
    assert r["tailed"][9][0] == nb and r["separate"][9][0] == nb         # after the agreement: gradients only, either way


# ---------------------------------------------------------------------------------------------------------------- range-guard flag, N > 1
def _skip_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tqdne_amd.trainer import DataParallelTrainer

    class T(DataParallelTrainer):
        AGREE_EVERY = 4

        def __init__(self):   # (only what _range_skip_flag touches)
            self.world, self.group, self.fused, self.exchange = world, None, True, True
            self.flag = torch.zeros(1, dtype=torch.int32)
            self.auto = True

        def _local_range_flag(self):
            return self.flag, self.auto

    t = T()
    log = []
    for step in range(1, 11):
        if step == 2 and rank == 1:
            t.flag.fill_(1)            # only rank 1's forward came near the fp16 range
        if step == 3 and rank == 1:
            t.flag.zero_(); t.auto = False    # ... its host has moved its plans to bf16x3
        if step == 5 and rank == 0:
            t.auto = False             # rank 0 follows later
        f = t._range_skip_flag()
        log.append(None if f is None else (float(f[0]) != 0.0, float(f[1])))
    ret[rank] = log
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_range_flag_raised_by_one_rank_skips_on_all_and_the_exchange_ends_by_agreement():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_skip_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0] == ret[1], (ret[0], ret[1])          # every rank sees the same predicate at every step: replicas stay equal
    log = ret[0]
    assert log[0] == (False, 0.0) and log[1] == (True, 0.0)            # step 2: rank 1's flag drops the step on BOTH ranks
    assert log[2] == (False, 1.0) and log[4] == (False, 2.0)           # off-scheme ranks are counted
    assert log[7] == (False, 2.0) and log[8] is None and log[9] is None   # agreed at step 8 (AGREE_EVERY = 4): no more collectives
