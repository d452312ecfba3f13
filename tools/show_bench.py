#!/usr/bin/env python3
"""Prints the headline and the extras of bench.py's DETAIL document (bench_detail.json; the compact stdout line carries extras as
[Mrays/s, HBM fraction] pairs only): tools/show_bench.py bench_detail.json"""
import json, sys
d = json.loads(open(sys.argv[1]).readline())
vr = d.get("valu_roofline") or {}
print(f"{d['metric']}: {d['value']:.0f} {d['unit']}  step {d['ms_per_step']:.4f} ms  kernel {d['roofline']['kernel_avg_ms']}  {d['config']['kernel']}"
      f"  hbm frac {d['roofline']['frac']}  valu frac spec/measured {vr.get('frac_vs_spec')} / {vr.get('frac_vs_measured')}")
for k, v in (d.get("extra") or {}).items():
    if isinstance(v, list):   # a compact line
        print(f"  {k:28s} {v[0]:9.0f} Mrays/s  hbm frac {v[1]}")
    elif "Mrays/s" in v:
        r = v.get("valu_roofline") or {}
        print(f"  {k:28s} {v['Mrays/s']:9.0f} Mrays/s  kernel {v.get('kernel_avg_ms')}  {v.get('kernel')}  valu frac {r.get('frac_vs_spec')} / {r.get('frac_vs_measured')}")
    elif "static" in v:  # name@moving: camera still / orbiting / panning, tile-order feedback on and off
        for m, row in v.items():
            if isinstance(row, dict):
                print(f"  {k:28s} {m:10s} feedback on {row['feedback_on']['Mrays/s']:8.0f}  off {row['feedback_off']['Mrays/s']:8.0f} Mrays/s  gain {100 * row['gain']:+5.1f} %")
    else:
        print(f"  {k:28s} {v}")
print("  cpu_baseline", d.get("cpu_baseline"))
