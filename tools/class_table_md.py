#!/usr/bin/env python3
"""Print the per-class tables of a bench.py JSON line (kernel_classes) as the markdown used in DESIGN.md section 5."""
import json, sys
d = json.load(open(sys.argv[1]))
for sec, t in d["kernel_classes"].items():
    print(f"| class ({sec}, {t['total_ms']:.2f} ms) | launches | ms | share | algorithmic TFLOP/s ÷ 2500 | algorithmic GB/s ÷ 8000 |")
    print("|---|---|---|---|---|---|")
    for c, v in sorted(t["classes"].items(), key=lambda kv: -kv[1]["ms"]):
        f = lambda x: "" if x is None else f"{x:.3f}"
        print(f"| {c} | {v['launches']} | {v['ms']:.3f} | {v['share']:.3f} | {f(v['mfma_frac'])} | {f(v['hbm_frac'])} |")
    print()
