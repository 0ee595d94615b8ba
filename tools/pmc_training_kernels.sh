#!/bin/bash
# usage (GPU box, repo root): tools/pmc_training_kernels.sh <tag>  -- PMC counters (separate passes, tools/pmc_kernel.sh) of the training half's
# kernel classes on paper-UNet shapes at B = 64: weight gradient (8-wave shared-tile form; the four-wave form for 64 output channels),
# data gradient with its chain epilogue (fp16 + MX-fp6), attention forward + backward.  Summary table -> gpurun_out/<tag>_pmc_training_kernels.txt
tag=$1
repo=$GRAFT_REPO_ROOT
cd $repo
bash tools/pmc_kernel.sh ${tag}_wgrad_w8 "wgrad_kernel" tools/bench_wgrad_one.py 256 0 256 5 1024 64 5 > /dev/null 2>&1
bash tools/pmc_kernel.sh ${tag}_wgrad_h64 "wgrad_kernel" tools/bench_wgrad_one.py 64 0 64 5 4096 64 5 > /dev/null 2>&1
bash tools/pmc_kernel.sh ${tag}_dgrad_chain "conv1d_mfma" tools/bench_dgrad_one.py 256 256 5 1024 64 5 > /dev/null 2>&1
bash tools/pmc_kernel.sh ${tag}_attn_fwd "attention_fwd2" tools/bench_attn_bwd_one.py 64 512 5 > /dev/null 2>&1
bash tools/pmc_kernel.sh ${tag}_attn_bwd_dkv "attention_bwd2_dkv" tools/bench_attn_bwd_one.py 64 512 5 > /dev/null 2>&1
bash tools/pmc_kernel.sh ${tag}_attn_bwd_dq "attention_bwd2_dq" tools/bench_attn_bwd_one.py 64 512 5 > /dev/null 2>&1
python3 - <<PY
import json
rows = [("wgrad 256 -> 256 k5 T1024 (8-wave shared tile)", "${tag}_wgrad_w8", 2 * 256 * 256 * 5 * 1024 * 64, 4 * 64 * 1024 * (256 + 256)),
        ("wgrad 64 -> 64 k5 T4096 (four-wave form, round 6)", "${tag}_wgrad_h64", 2 * 64 * 64 * 5 * 4096 * 64, 4 * 64 * 4096 * (64 + 64)),
        ("dgrad 256 -> 256 k5 T1024, chain epilogue, f16+mx6", "${tag}_dgrad_chain", 2 * 256 * 256 * 5 * 1024 * 64, 4 * 64 * 1024 * (256 + 256 + 256)),
        ("attention forward (training: bf16x3, lse)", "${tag}_attn_fwd", 4 * 4 * 64 * 512 * 512 * 64, 4 * 64 * 512 * (768 + 256)),
        ("attention backward, dK / dV pass", "${tag}_attn_bwd_dkv", 0, 0), ("attention backward, dQ pass", "${tag}_attn_bwd_dq", 0, 0)]
out = ["# PMC counters per launch (rocprofv3 --pmc, separate passes: tools/pmc_kernel.sh), B = 64, each kernel alone; FETCH_SIZE x 2 (gfx950), KiB units",
       "# busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); traffic = HBM bytes read + written; algo = operands once in fp32",
       f"{'kernel':52s} {'us':>7s} {'TFLOP/s':>8s} {'MFMA busy':>9s} {'traffic MB':>10s} {'algo MB':>8s} {'LDS confl/active':>16s}"]
for name, t, fl, ab in rows:
    try:
        d = json.load(open(f"gpurun_out/pmc_{t}.json"))
    except Exception as e:
        out.append(f"{name:52s} (no data: {e!r})"); continue
    us = d.get("kernel_us_under_profiler", 0.0)
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, 128.0 * d.get("GRBM_GUI_ACTIVE", 0.0))   # (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    tr = (d.get("FETCH_SIZE", 0.0) * 2048 + d.get("WRITE_SIZE", 0.0) * 1024) / 1e6
    lds = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, d.get("SQ_LDS_IDX_ACTIVE", 0.0))
    out.append(f"{name:52s} {us:7.1f} {fl / max(us, 1e-9) / 1e6:8.1f} {busy:9.3f} {tr:10.1f} {ab / 1e6:8.1f} {lds:16.3f}")
open("gpurun_out/${tag}_pmc_training_kernels.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
