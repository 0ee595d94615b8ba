#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory per UNet forward (developer tool)."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
nf = max(1, sum(int(r["Calls"]) for r in rows if "embed_kernel" in r["Name"]))
print(f"total {tot/1e6:.1f} ms, {nf} forwards, {tot/1e6/nf:.3f} ms/fwd")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
    print(f"{r['Name'][:84]:84s} n/fwd={int(r['Calls'])/nf:6.2f} ms/fwd={float(r['TotalDurationNs'])/1e6/nf:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
