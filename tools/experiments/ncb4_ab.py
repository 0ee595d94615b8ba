"""A/B of the 256-channel conv tile as 4 x 2 waves of 64 channels x 64 positions (TQDNE_CONV_NCB4=1: half the LDS reads per MFMA, one weight
buffer) against the default 8 x 1 waves of 32 x 128: us per launch and a checksum of output + statistics.
usage: TQDNE_CONV_NCB4=0|1 python tools/experiments/ncb4_ab.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tqdne_amd import ops, _lib
import ctypes as C

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
LAYERS = [(256, 0, 256, 1024), (256, 256, 256, 1024), (256, 128, 256, 1024), (256, 0, 256, 512), (256, 256, 256, 512), (256, 0, 256, 500)]
lib = _lib.load()
torch.manual_seed(0)
for (C0, C1, Co, T) in LAYERS:
    K = 5
    x0 = torch.randn(B, T, C0, device=dev)
    x1 = torch.randn(B, T, C1, device=dev) if C1 else None
    w = torch.randn(Co, C0 + C1, K, device=dev) / (K * (C0 + C1)) ** 0.5
    b = torch.randn(Co, device=dev)
    gs = torch.rand(B, C0 + C1, device=dev) + 0.5
    gh = torch.randn(B, C0 + C1, device=dev)
    y = torch.empty(B, T, Co, device=dev)
    st = torch.zeros(B, (T + 127) // 128, Co, 2, device=dev)
    d_wfmt = _lib.forward_wfmt(Co, [C0, C1])
    wp = ops.pack_conv_weight(w, _lib.PACK_MODE[d_wfmt])
    d = _lib.TqConvDesc()
    d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = B, T, T, C0, C1, Co
    d.ktaps, d.stride, d.pad, d.upsample = K, 1, K // 2, 0
    d.flags = 3 | 16
    d.wfmt = d_wfmt
    stream = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    def run():
        rc = lib.tq_conv1d_fwd(C.byref(d), p(x0), p(x1), p(gs), p(gh), p(wp), p(b), None, None, p(y), p(st), stream)
        assert rc == 0, rc
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    cs = int(y.view(torch.int32).to(torch.int64).sum()) ^ int(st.view(torch.int32).to(torch.int64).sum())
    print(f"NCB4={os.environ.get('TQDNE_CONV_NCB4', '0')} B={B} {C0}+{C1}->{Co} T={T} wfmt={d_wfmt}: {1e3 * e0.elapsed_time(e1) / n:7.1f} us  checksum {cs & 0xFFFFFFFFFFFF:012x}", flush=True)
