#!/usr/bin/env python3
"""A/B of the slim 64-channel conv tile (TQDNE_CONV_SLIM=1/0) per layer, in child processes (the switch is read once).  GPU box."""
import os, subprocess, sys
if os.environ.get("SLIM_CHILD"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ctypes as C, torch
    from tqdne_amd import _lib, ops
    lib = _lib.load(); dev = torch.device("cuda:0"); B = 64
    for (C0, C1, Co, T, Cs) in [(64, 0, 64, 4096, 0), (64, 64, 64, 4096, 128), (128, 64, 64, 4096, 192), (64, 0, 64, 4000, 0)]:
        g = torch.Generator().manual_seed(C0 + C1 + T)
        x0 = torch.randn(B, T, C0, generator=g).to(dev); x1 = torch.randn(B, T, C1, generator=g).to(dev) if C1 else None
        w = (torch.randn(Co, C0 + C1, 5, generator=g) / (5 * (C0 + C1)) ** 0.5).to(dev); bias = torch.randn(Co, generator=g).to(dev)
        gs = (torch.rand(B, C0 + C1, generator=g) + 0.5).to(dev); gh = torch.randn(B, C0 + C1, generator=g).to(dev)
        emb = torch.randn(B, Co, generator=g).to(dev)
        y = torch.empty(B, T, Co, device=dev); st = torch.zeros(B, (T + 127) // 128, Co, 2, device=dev)
        wp = ops.pack_conv_weight(w, 0)
        d = _lib.TqConvDesc(); d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, T, T, C0, C1, Co
        d.ktaps, d.stride, d.pad, d.upsample, d.flags, d.wfmt, d.emb_stride = 5, 1, 2, 0, 1 | 2 | 4 | 16, 0, Co
        p = lambda t: None if t is None else t.data_ptr(); s = torch.cuda.current_stream().cuda_stream
        run = lambda: lib.tq_conv1d_fwd(C.byref(d), p(x0), p(x1), p(gs), p(gh), p(wp), p(bias), p(emb), None, p(y), p(st), s)
        for _ in range(5): assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
        for _ in range(7):
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
        print(f"SLIM={os.environ.get('TQDNE_CONV_SLIM')} {C0}+{C1}->{Co} T{T}: {sorted(ts)[3]:7.1f} us  checksum {float(y.double().sum()):.6e} {float(st.double().sum()):.6e}", flush=True)
    sys.exit(0)
for v in ("1", "0", "1", "0"):   # (TQDNE_CONV_SLIM is read once per process)
    subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, SLIM_CHILD="1", TQDNE_CONV_SLIM=v))
