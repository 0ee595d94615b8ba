#!/usr/bin/env python3
"""One-GPU smoke test of the N > 1 code path of bench.py / DataParallelTrainer: RCCL process group of ONE rank, with the trainer
told the world has 2 ranks so that the bucketed all-reduce, the broadcast and the barriers really run on the GPU.
(Two ranks cannot share one GPU under RCCL; the 2-rank semantics are covered with gloo in tests/test_distributed_cpu.py.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
from tqdne_amd import LightningEDM, tiny_1d_unet_config
from tqdne_amd.trainer import DataParallelTrainer

cfg = tiny_1d_unet_config()
torch.manual_seed(0)
edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 1000, "eta_min": 0.0}).to(dev)
tr = DataParallelTrainer(edm, world_size=1)
tr.world = 2  # force the exchange
for p in edm.parameters():
    dist.broadcast(p.data, src=0)
batch = {"signal": 0.5 * torch.randn(4, 3, 1024, device=dev)}
if cfg["cond_features"]:
    batch["cond"] = torch.randn(4, cfg["cond_features"], device=dev)
edm.train()
l0 = float(tr.train_step(batch))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    loss = tr.train_step(batch)
torch.cuda.synchronize()
dist.barrier()
tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
print(f"dist smoke ok: loss {l0:.4f} -> {float(loss):.4f}, {1e3 * float(tt) / 5:.2f} ms/step incl. RCCL all-reduce of the flat gradient")
dist.destroy_process_group()
