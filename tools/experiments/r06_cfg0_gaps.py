#!/usr/bin/env python3
"""Round 6: where a cfg0 (tiny UNet, B = 4) sample spends its time on the GPU.  Run under `rocprofv3 --kernel-trace --output-format csv -d DIR --
python3 tools/experiments/r06_cfg0_gaps.py run`; then `python3 tools/experiments/r06_cfg0_gaps.py report DIR`: kernel time against the
span of the kernels on the device (= kernel time + the gaps between dependent launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if sys.argv[1] == "run":
    import torch
    import bench
    from tqdne_amd import LightningEDM, tiny_1d_unet_config
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    edm = LightningEDM(tiny_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0}, num_sampling_steps=18)
    edm.unet.load_state_dict(bench.perturbed_state(edm.unet, 17))
    edm = edm.to(dev).eval()
    B, T = int(os.environ.get("CFG0_B", "4")), 4096
    g = torch.Generator().manual_seed(4321)
    sig = edm.edm.sampling_sigmas(18).to(dev)
    eps0 = torch.randn(B, 3, T, generator=g, dtype=torch.float64).to(dev) * sig[0]
    for _ in range(3):
        edm.sample_deterministically(eps0, sig, None, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        edm.sample_deterministically(eps0, sig, None, None)
    e1.record()
    torch.cuda.synchronize()
    print(f"cfg0 sample, B = {B}: {e0.elapsed_time(e1) / 5:.3f} ms per 18-step sample (35 network evaluations)")
else:
    import csv, glob, collections
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    n = len(rows)
    rows = rows[n * 3 // 8:]   # (the five timed samples of the eight)
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"{len(rows)} launches, span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms ({100.0 * busy / span:.1f} %), "
          f"gaps: mean {sum(pos) / max(1, len(pos)) / 1e3:.2f} us, median {sorted(pos)[len(pos) // 2] / 1e3:.2f} us")
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, k in rows:
        per[k][0] += 1
        per[k][1] += e - s
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {k[:110]:110s} n={c:6d} avg {t / c / 1e3:7.2f} us  {100.0 * t / busy:5.1f} %")
