#!/usr/bin/env python3
"""Persistent conv tiles (TQDNE_CONV_PERSIST=2, conv1d_mfma_kernel's NTILE) against the one-tile kernel on the same launches: the
script re-runs itself in two child processes (the switch is read once per process) and compares the outputs bit for bit.
usage (GPU box): python tools/persist_check.py [B]"""
import os, subprocess, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(64, 0, 64, 1024, 0, 0), (64, 0, 64, 1000, 0, 0), (128, 64, 64, 512, 0, 0), (256, 0, 256, 512, 2, 0), (256, 0, 256, 500, 2, 0),
          (128, 128, 128, 256, 2, 0), (256, 128, 256, 1024, 2, 1), (64, 0, 64, 2048, 0, 1), (128, 0, 128, 768, 2, 128), (64, 0, 128, 512, 2, 64)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from tqdne_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B = int(sys.argv[2])
    for (C0, C1, Co, T, wf, extra) in SHAPES:
        K = 5
        x0 = torch.randn(B, T, C0, generator=g); x1 = torch.randn(B, T, C1, generator=g) if C1 else None
        w = torch.randn(Co, C0 + C1, K, generator=g) / (K * (C0 + C1)) ** 0.5
        bias = torch.randn(Co, generator=g); emb = torch.randn(B, Co, generator=g)
        gs = torch.rand(B, C0 + C1, generator=g) + 0.5; gh = torch.randn(B, C0 + C1, generator=g)
        kw = {}
        if extra == 1:
            kw["residual"] = torch.randn(B, T, Co, generator=g).to(dev)
        elif extra > 1:   # fused 1x1 skip conv from a block input of `extra` channels
            kw["skip"] = (torch.randn(B, T, extra, generator=g).to(dev), None, (torch.randn(Co, extra, 1, generator=g) / extra ** 0.5).to(dev),
                          torch.randn(Co, generator=g).to(dev))
        y, st = ops.conv1d(x0.to(dev), w.to(dev), bias.to(dev), x1=None if x1 is None else x1.to(dev), gscale=gs.to(dev),
                           gshift=gh.to(dev), silu=True, emb=emb.to(dev), stats=True, wfmt=wf, **kw)
        xin = torch.cat([x0] + ([x1] if x1 is not None else []), dim=2).double()
        a = torch.nn.functional.silu(xin * gs[:, None, :].double() + gh[:, None, :].double())
        ref = torch.nn.functional.conv1d(a.permute(0, 2, 1), w.double(), bias.double(), padding=K // 2) + emb.double()[:, :, None]
        if extra == 1:
            ref = ref + kw["residual"].cpu().double().permute(0, 2, 1)
        elif extra > 1:
            sx, _, sw, sb = kw["skip"]
            ref = ref + torch.nn.functional.conv1d(sx.cpu().double().permute(0, 2, 1), sw.cpu().double(), sb.cpu().double())
        e = float((y.cpu().double().permute(0, 2, 1) - ref).abs().max() / ref.abs().max())
        h = hashlib.sha1(y.cpu().numpy().tobytes() + st.cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"RESULT {C0}+{C1}->{Co} T{T} wfmt{wf} extra{extra}: err {e:.2e} hash {h}", flush=True)
    sys.exit(0)

B = sys.argv[1] if len(sys.argv) > 1 else "3"
outs = {}
for mode in ("0", "2"):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", B], env=dict(os.environ, TQDNE_CONV_PERSIST=mode),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    outs[mode] = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if r.returncode != 0 or len(outs[mode]) != len(SHAPES):
        print(r.stdout[-3000:])
        sys.exit(1)
ok = True
for a, b in zip(outs["0"], outs["2"]):
    same = a == b
    err = float(b.split("err ")[1].split()[0])
    ok = ok and same and err < 1e-4
    print(("same  " if same else "DIFF  ") + b[7:] + ("" if same else "   one-tile: " + a[7:]))
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
