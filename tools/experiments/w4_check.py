#!/usr/bin/env python3
"""conv1d_w4.hip (one wave per SIMD) against conv1d_mfma.hip (two waves per SIMD) on the same launches: outputs and GroupNorm
partial sums must be BIT-IDENTICAL (same operand formats, same accumulation order), then per-layer timing of both, interleaved.
Developer tool, run on the GPU box: python tools/experiments/w4_check.py [B]"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CHILD = os.environ.get("W4_CHILD")
# (C0, C1, Cout, T, fused-skip channels (0 = none))
LAYERS = [(256, 0, 256, 1024, 0), (256, 0, 256, 512, 0), (256, 256, 256, 1024, 0), (256, 128, 256, 1024, 0), (128, 0, 128, 2048, 0),
          (256, 0, 256, 1000, 0), (128, 128, 128, 2048, 0), (256, 0, 256, 128, 0)]


def run_all(timing):
    from tqdne_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    out = {}
    for (C0, C1, Co, T, Cs) in LAYERS:
        g = torch.Generator().manual_seed(C0 * 7 + C1 * 3 + Co + T)
        Bq = B if timing else 3
        x0 = torch.randn(Bq, T, C0, generator=g).to(dev)
        x1 = torch.randn(Bq, T, C1, generator=g).to(dev) if C1 else None
        w = (torch.randn(Co, C0 + C1, 5, generator=g) / (5 * (C0 + C1)) ** 0.5).to(dev)
        bias = torch.randn(Co, generator=g).to(dev)
        emb = torch.randn(Bq, Co, generator=g).to(dev)
        gs = (torch.rand(Bq, C0 + C1, generator=g) + 0.5).to(dev)
        gh = torch.randn(Bq, C0 + C1, generator=g).to(dev)
        y = torch.empty(Bq, T, Co, device=dev)
        st = torch.zeros(Bq, (T + 127) // 128, Co, 2, device=dev)
        wf = _lib.TQ_WFMT_F16_MX6
        wp = ops.pack_conv_weight(w, _lib.PACK_MODE[wf])
        d = _lib.TqConvDesc()
        d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = Bq, T, T, C0, C1, Co
        d.ktaps, d.stride, d.pad, d.upsample = 5, 1, 2, 0
        d.flags = 1 | 2 | 4 | 16
        d.emb_stride = Co
        d.wfmt = wf
        p = lambda t: None if t is None else t.data_ptr()
        stream = torch.cuda.current_stream().cuda_stream

        def run():
            rc = lib.tq_conv1d_fwd(C.byref(d), p(x0), p(x1), p(gs), p(gh), p(wp), p(bias), p(emb), None, p(y), p(st), stream)
            assert rc == 0, rc
        run()
        torch.cuda.synchronize()
        key = f"{C0}+{C1}->{Co} T{T}"
        if not timing:
            out[key] = (y.cpu().clone(), st.cpu().clone())
            continue
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100.0)
        out[key] = sorted(ts)[len(ts) // 2]
    return out


if CHILD:
    res = run_all(CHILD == "time")
    torch.save(res, os.environ["W4_OUT"])
    sys.exit(0)

import tempfile
tmp = tempfile.mkdtemp()
res = {}
for mode in ("check", "time"):
    for w4 in ("1", "0"):
        outp = os.path.join(tmp, f"{mode}{w4}.pt")
        env = dict(os.environ, W4_CHILD=mode, W4_OUT=outp, TQDNE_CONV_W4=w4)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(B)], env=env)
        if r.returncode != 0:
            print(f"child {mode} w4={w4} failed rc={r.returncode}")
            sys.exit(1)
        res[(mode, w4)] = torch.load(outp)
ok = True
for k in res[("check", "1")]:
    y1, s1 = res[("check", "1")][k]
    y0, s0 = res[("check", "0")][k]
    same = torch.equal(y1, y0) and torch.equal(s1, s0)
    err = float((y1 - y0).abs().max() / y0.abs().max())
    serr = float((s1 - s0).abs().max() / s0.abs().max())
    print(f"check {k}: bit-identical {same} (outputs {torch.equal(y1, y0)}, statistics {torch.equal(s1, s0)}); max rel diff outputs {err:.2e}, "
          f"statistics {serr:.2e}; finite {bool(torch.isfinite(y1).all())}")
    ok = ok and same
for k in res[("time", "1")]:
    t1, t0 = res[("time", "1")][k], res[("time", "0")][k]
    C0, rest = k.split("+")
    print(f"time  {k} B={B}: w4 {t1:8.1f} us   two-wave {t0:8.1f} us   ratio {t1 / t0:.3f}")
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
