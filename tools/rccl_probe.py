"""Where does the gradient exchange's cost at ONE rank come from?  (bench.py with TQDNE_BENCH_FORCE_RCCL=1: train half +4-6 ms although a
sum over one rank launches no RCCL kernel.)  Times, on the paper UNet B = 64 x 4096 training step:
  a) no exchange;  b) forced exchange, all-reduces after the backward;  c) forced, from inside the sweep;
  d) as c with the range-flag exchange off;  e) as c with the bucket all-reduces replaced by nothing (hooks + stream joins only);
and the bare cost of one async all-reduce + wait on an idle and on a busy stream (host time per call, GPU time added)."""

import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.dup2(2, 1)   # (RCCL's banner)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = sys.stderr

    # ---- bare collective
    x = torch.zeros(4 << 20, device=dev)
    y = torch.zeros(1 << 20, device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    for label, busy in (("idle stream", False), ("busy stream (40 x 1M-element add_ between calls)", True)):
        for with_coll in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = 0.0
            for _ in range(20):
                if busy:
                    for _ in range(40):
                        y.add_(1.0)
                if with_coll:
                    t1 = time.perf_counter()
                    dist.all_reduce(x, async_op=True).wait()
                    th += time.perf_counter() - t1
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            print(f"bare: {label}, collectives {'on ' if with_coll else 'off'}: issue {1e3 * t_issue:.2f} ms, done {1e3 * t_all:.2f} ms, "
                  f"host per all_reduce+wait {1e6 * th / 20:.0f} us", file=out)

    # ---- the training step
    from tqdne_amd import LightningEDM, paper_1d_unet_config, rng
    from tqdne_amd.trainer import DataParallelTrainer
    import bench
    cfg = paper_1d_unet_config()
    B, T = 64, 4096
    torch.manual_seed(0)
    rng.seed_rank(0, 0)
    g = torch.Generator().manual_seed(1234)
    batch = {"signal": (0.5 * torch.randn(B, 3, T, generator=g)).to(dev), "cond": torch.randn(B, 5, generator=g).to(dev)}

    def make(**kw):
        edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 100000, "eta_min": 0.0})
        edm.unet.load_state_dict(bench.perturbed_state(edm.unet, 17))
        edm = edm.to(dev).train()
        return DataParallelTrainer(edm, world_size=1, **kw)

    class _Done:
        def wait(self):
            return True

    def timeit(tr, label):
        for _ in range(3):
            tr.train_step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.train_step(batch)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"train: {label}: host issue {1e2 * t_issue:.2f} ms/step, done {1e2 * t_all:.2f} ms/step", file=out)

    timeit(make(), "a) no exchange")
    timeit(make(force_exchange=True, overlap=False), "b) forced, all-reduces after the backward")
    timeit(make(force_exchange=True, overlap=True), "c) forced, all-reduces from inside the sweep")
    tr = make(force_exchange=True, overlap=True)
    tr._range_skip_flag = lambda: None
    timeit(tr, "d) as c, range-flag exchange off")
    tr = make(force_exchange=True, overlap=True)
    tr._allreduce_async = lambda t: _Done()
    timeit(tr, "e) as c, no collective issued at all (hooks + stream joins only)")
    tr = make(force_exchange=True, overlap=False)
    tr._allreduce_async = lambda t: _Done()
    timeit(tr, "f) as b, no collective issued at all")
    timeit(make(), "a) no exchange, again")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
