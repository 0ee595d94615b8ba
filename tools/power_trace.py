#!/usr/bin/env python3
"""Board power and shader clock while the bench workloads run (developer probe, GPU box): is the step power-limited?

Samples, every ~50 ms from a side thread, the hwmon power reading and the current sclk of the GPU (sysfs: power1_average / power1_input,
pp_dpm_sclk; fallback `rocm-smi --showpower --showclocks --json` at ~1 Hz) while the main thread runs, back to back for a few seconds each:
  idle | the 18-step sampler (B = 64, 4 lanes) | the training step | the dominant conv alone (B = 64) | the same conv on 64 workgroups (B = 8)
and prints per phase: median / max power, power cap, median sclk.  usage: tools/power_trace.py [seconds per phase]"""
import glob, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def _hwmon():
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(card, "pp_dpm_sclk")):
            continue
        hw = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*")))
        if hw:
            return card, hw[0]
    return None, None


CARD, HW = _hwmon()


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except Exception:
        return None


def sample_sysfs():
    p = None
    for name in ("power1_average", "power1_input"):
        v = _read(os.path.join(HW, name)) if HW else None
        if v:
            p = float(v) / 1e6
            break
    clk = None
    s = _read(os.path.join(CARD, "pp_dpm_sclk")) if CARD else None
    if s:
        for line in s.splitlines():
            if line.strip().endswith("*"):
                clk = float(line.split(":")[1].strip().lower().replace("mhz", "").replace("*", "").strip())
    return p, clk


def sample_smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        c = next(iter(d.values()))
        p = next((float(v) for k, v in c.items() if "power" in k.lower() and "(w)" in k.lower()), None)
        clk = next((float(str(v).strip("()").lower().replace("mhz", "")) for k, v in c.items() if k.lower().startswith("sclk")), None)
        return p, clk
    except Exception:
        return None, None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.phase, self.stop = [], "idle", False
        self.use_sysfs = sample_sysfs()[0] is not None

    def run(self):
        while not self.stop:
            p, c = sample_sysfs() if self.use_sysfs else sample_smi()
            self.rows.append((self.phase, p, c))
            time.sleep(0.05 if self.use_sysfs else 0.2)


def med(v):
    v = sorted(x for x in v if x is not None)
    return v[len(v) // 2] if v else None


def main():
    from tqdne_amd import LightningEDM, paper_1d_unet_config, rng
    from tqdne_amd.trainer import DataParallelTrainer
    dev = torch.device("cuda:0")
    cap = _read(os.path.join(HW, "power1_cap")) if HW else None
    print("power cap (W):", None if not cap else float(cap) / 1e6, "| source:", "sysfs " + str(HW) if sample_sysfs()[0] is not None else "rocm-smi")
    torch.manual_seed(0)
    edm = LightningEDM(paper_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 100000, "eta_min": 0.0}, num_sampling_steps=18).to(dev)
    with torch.no_grad():
        for p in edm.unet.parameters():
            if torch.count_nonzero(p) == 0:
                p.normal_(0, 0.02)
    B, T = 64, 4096
    g = torch.Generator().manual_seed(1)
    sig = edm.edm.sampling_sigmas(18).to(dev)
    eps = torch.randn(B, 3, T, generator=g, dtype=torch.float64).to(dev) * sig[0]
    cond = torch.randn(B, 5, generator=g).to(dev)
    batch = {"signal": (0.5 * torch.randn(B, 3, T, generator=g)).to(dev), "cond": cond}
    tr = DataParallelTrainer(edm, world_size=1)
    rng.seed_rank(0, 0)

    def sample():
        edm.eval()
        edm.sample_deterministically(eps, sig, None, cond)

    def train():
        edm.train()
        tr.train_step(batch)

    # the dominant conv alone (tools/experiments/ncb4_ab.py's launch), full grid and a quarter-filled chip
    import ctypes as C
    from tqdne_amd import _lib, ops
    lib = _lib.load()

    def conv_runner(Bc):
        C0, C1, Co, Tc, K = 256, 256, 256, 1024, 5
        x0, x1 = torch.randn(Bc, Tc, C0, device=dev), torch.randn(Bc, Tc, C1, device=dev)
        w = torch.randn(Co, C0 + C1, K, device=dev) / (K * (C0 + C1)) ** 0.5
        gs, gh = torch.rand(Bc, C0 + C1, device=dev) + 0.5, torch.randn(Bc, C0 + C1, device=dev)
        y, st = torch.empty(Bc, Tc, Co, device=dev), torch.zeros(Bc, Tc // 128, Co, 2, device=dev)
        wf = _lib.forward_wfmt(Co, [C0, C1])
        wp = ops.pack_conv_weight(w, _lib.PACK_MODE[wf])
        d = _lib.TqConvDesc()
        d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = Bc, Tc, Tc, C0, C1, Co
        d.ktaps, d.stride, d.pad, d.upsample, d.flags, d.wfmt = K, 1, K // 2, 0, 3 | 16, wf
        keep = (x0, x1, w, gs, gh, y, st, wp, d)
        stream = torch.cuda.current_stream().cuda_stream

        def run():
            for _ in range(20):
                rc = lib.tq_conv1d_fwd(C.byref(d), x0.data_ptr(), x1.data_ptr(), gs.data_ptr(), gh.data_ptr(), wp.data_ptr(), None, None, None,
                                       y.data_ptr(), st.data_ptr(), stream)
                assert rc == 0
            return keep
        return run

    conv64, conv8 = conv_runner(64), conv_runner(8)
    for f in (sample, train, conv64, conv8):
        f()
    torch.cuda.synchronize()
    s = Sampler()
    s.start()
    time.sleep(2.0)
    res = {}
    for name, fn in (("sampler_b64_4lanes", sample), ("train_step_b64", train), ("conv_512to256_T1024_b64 (512 workgroups)", conv64),
                     ("conv_512to256_T1024_b8 (64 workgroups)", conv8)):
        s.phase = name
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < SECS:
            fn()
            n += 1
            if n % 4 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / n
        s.phase = "idle"
        time.sleep(1.0)
    s.stop = True
    s.join()
    print(f"{'phase':45s} {'samples':>7s} {'P median W':>10s} {'P max W':>8s} {'sclk median MHz':>15s}   ms / call")
    for ph in ["idle"] + list(res):
        ps = [r[1] for r in s.rows if r[0] == ph]
        cs = [r[2] for r in s.rows if r[0] == ph]
        pm = med(ps)
        px = max([x for x in ps if x is not None], default=None)
        print(f"{ph:45s} {len(ps):7d} {str(None if pm is None else round(pm, 1)):>10s} {str(None if px is None else round(px, 1)):>8s} "
              f"{str(med(cs)):>15s}   {'' if ph == 'idle' else round(1e3 * res[ph], 2)}")


if __name__ == "__main__":
    main()
