import os, sys, torch, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tqdne_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
p = lambda t: None if t is None else t.data_ptr()
def med(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        a,b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(1e3*a.elapsed_time(b))
    return sorted(ts)[n//2]
st = lambda: torch.cuda.current_stream().cuda_stream
for B in (64, 16):
  for (Cio, T) in ((3, 4096), (6, 4064), (16, 4096)):
    x = torch.randn(B, Cio, T, device=dev); w = torch.randn(64, Cio, 5, device=dev); b = torch.randn(64, device=dev)
    y = torch.empty(B, T, 64, device=dev); stats = torch.empty(B, (T+127)//128, 64, 2, device=dev); sc = torch.rand(B, device=dev)
    t_stem = med(lambda: lib.tq_stem_conv_fwd(p(x), p(sc), p(w), p(b), p(y), p(stats), B, Cio, T, 64, 5, st()))
    h = torch.randn(B, T, 64, device=dev); gs = torch.rand(B, 64, device=dev); gh = torch.randn(B, 64, device=dev)
    wh = torch.randn(Cio, 64, 5, device=dev); bh = torch.randn(Cio, device=dev); out = torch.empty(B, Cio, T, device=dev)
    co, ck = torch.rand(B, device=dev), torch.rand(B, device=dev)
    t_head = med(lambda: lib.tq_head_conv_fwd(p(h), p(gs), p(gh), p(wh), p(bh), p(co), p(ck), p(x), p(out), B, T, 64, Cio, 5, st()))
    print(f"B={B} channels={Cio} T={T}: stem {t_stem:.1f} us, head {t_head:.1f} us")

# backward of the two ends (training): stem weight gradient and head backward
from tqdne_amd import ops
for B in (64,):
  for (Cio, T) in ((3, 4096), (6, 4064)):
    x = torch.randn(B, Cio, T, device=dev); dy = torch.randn(B, T, 64, device=dev); sc = torch.rand(B, device=dev)
    dw = torch.zeros(64, Cio, 5, device=dev); ws = torch.empty(lib.tq_stem_head_bwd_workspace(), dtype=torch.uint8, device=dev)
    t_s = med(lambda: lib.tq_stem_conv_bwd_weight_ws(p(dy), p(x), p(sc), p(dw), B, Cio, T, 64, 5, p(ws), ws.numel(), st()))
    h = torch.randn(B, T, 64, device=dev); gs = torch.rand(B, 64, device=dev); gh = torch.randn(B, 64, device=dev)
    wh = torch.randn(Cio, 64, 5, device=dev); dpred = torch.randn(B, Cio, T, device=dev); co = torch.rand(B, device=dev)
    G = torch.empty(B, T, 64, device=dev); gst = torch.empty(B, (T + 127) // 128, 64, 2, device=dev)
    dwh = torch.zeros(Cio, 64, 5, device=dev); dbh = torch.zeros(Cio, device=dev)
    t_h = med(lambda: lib.tq_head_conv_bwd_ws(p(dpred), p(co), p(h), p(gs), p(gh), p(wh), p(G), p(gst), p(dwh), p(dbh), B, T, 64, Cio, 5, p(ws), ws.numel(), st()))
    print(f"B={B} channels={Cio} T={T}: stem weight gradient {t_s:.1f} us, head backward {t_h:.1f} us")
