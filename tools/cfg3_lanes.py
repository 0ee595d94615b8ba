"""cfg3 (latent EDM, B = 16, latent 16 x 4096, paper-shape UNet): the 18-step sample with 1 / 2 / 4 sampler lanes, eager and graph-replayed,
and the decode.  (The default puts B = 16 on ONE lane: 2.6 ms per evaluation against 1.14 ms per 16 samples inside the 4-lane B = 64 run.)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from tqdne_amd import LightningAutoencoder, LightningEDM, paper_1d_unet_config
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    base = dict(model_channels=64, channel_mult=(1, 2, 4), attention_resolutions=(), num_res_blocks=2, dims=1, conv_kernel_size=5, dropout=0.1)
    ae = LightningAutoencoder(dict(base, in_channels=3, out_channels=32), dict(base, in_channels=16, out_channels=3),
                              {"learning_rate": 1e-4, "max_steps": 1000, "eta_min": 0.0})
    ae.load_state_dict(bench.perturbed_state(ae, 19))
    ae = ae.to(dev).eval()
    edm = LightningEDM(paper_1d_unet_config(in_channels=16, out_channels=16), {"learning_rate": 1e-4, "max_steps": 100000, "eta_min": 0.0},
                       num_sampling_steps=18, autoencoder=ae)
    edm.unet.load_state_dict(bench.perturbed_state(edm.unet, 17))
    edm = edm.to(dev).eval()
    B = int(os.environ.get("CFG3_B", 16))
    g = torch.Generator().manual_seed(4322)
    cond = torch.randn(B, 5, generator=g).to(dev)
    sig = edm.edm.sampling_sigmas(18).to(dev)
    eps = torch.randn(B, 16, 4096, generator=g, dtype=torch.float64).to(dev) * sig[0]

    def med(fn, n=5):
        fn(); fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        return sorted(ts)[n // 2]

    ref = None
    for lanes in (1, 2, 4):
        for graph in (False, True):
            try:
                out = edm.sample_deterministically(eps, sig, None, cond, use_graph=graph, lanes=lanes)
                ms = med(lambda: edm.sample_deterministically(eps, sig, None, cond, use_graph=graph, lanes=lanes))
                if ref is None:
                    ref = out.clone()
                err = float((out - ref).abs().max() / ref.abs().max())
                print(f"B={B} lanes={lanes} graph={graph}: {ms:.2f} ms per 18-step latent sample ({ms / 35:.3f} ms per evaluation), max rel diff vs 1 lane eager {err:.2e}", flush=True)
            except Exception as e:
                print(f"B={B} lanes={lanes} graph={graph}: {e!r}", flush=True)
    z = torch.randn(B, 16, 4096, device=dev)
    print(f"decode: {med(lambda: ae.decode(z)):.2f} ms", flush=True)


if __name__ == "__main__":
    main()
