"""proj_out (256 -> 256, k = 1, T = 512, residual) and qkv-like 1x1 convs: fp16 + mx6 (default for these shapes) against bf16x3 --
the 1x1 convs are bound by staging and latency, not by MFMA cycles, and the bf16 split costs a third of the fp6 packing's VALU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tqdne_amd import ops, _lib
import ctypes as C

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
for (C0, Co, T, res) in [(256, 256, 512, True), (256, 256, 1024, True), (256, 256, 512, False)]:
    for wfmt in (_lib.TQ_WFMT_F16_MX6, _lib.TQ_WFMT_BF16X3):
        x0 = torch.randn(B, T, C0, device=dev)
        r = torch.randn(B, T, Co, device=dev)
        w = torch.randn(Co, C0, 1, device=dev) / C0 ** 0.5
        b = torch.randn(Co, device=dev)
        y = torch.empty(B, T, Co, device=dev)
        wp = ops.pack_conv_weight(w, _lib.PACK_MODE[wfmt])
        d = _lib.TqConvDesc()
        d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, T, T, C0, 0, Co
        d.ktaps, d.stride, d.pad, d.upsample = 1, 1, 0, 0
        d.flags = 8 if res else 0   # TQ_CONV_RES
        d.wfmt = wfmt
        stream = torch.cuda.current_stream().cuda_stream
        p = lambda t: None if t is None else t.data_ptr()
        def run():
            rc = lib.tq_conv1d_fwd(C.byref(d), p(x0), None, None, None, p(wp), p(b), None, p(r) if res else None, p(y), None, stream)
            assert rc == 0, rc
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        ref = x0 @ w[:, :, 0].T + b + (r if res else 0)
        err = float((y - ref).abs().max() / ref.abs().max())
        print(f"B={B} {C0}->{Co} k1 T={T} res={res} wfmt={wfmt}: {1e3 * e0.elapsed_time(e1) / 50:6.1f} us  err {err:.1e}", flush=True)
