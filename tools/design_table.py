#!/usr/bin/env python3
"""Rows of DESIGN.md section 5.2 from a profiles/<round>/ directory: pmc_*.json of tools/summarize_pmc.py (rocprofv3 kernel stats + PMC passes),
priced by bench.valu_roofline at the profiled kernel time.     usage: tools/design_table.py [profiles/round4]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/round4"
rows = sorted(f[4:-5] for f in os.listdir(d) if f.startswith("pmc_") and f.endswith(".json"))
order = ["direct32x8_1920x1080", "direct32x8_3840x2160", "lut32_1920x1080", "shipped8_1920x1080", "clouds_high_1920x1080", "clouds_high_rm_1920x1080",
         "clouds_high_rm_3840x2160", "clouds_high@lod0_1920x1080", "clouds_high_rm@lod0_1920x1080", "clouds_high_rm@lod0_3840x2160"]
rows = [r for r in order if r in rows] + [r for r in rows if r not in order]
print("| workload | kernel | kernel ms (rocprof avg) | VALU wave-insts | insts / wave | cycles/inst/SIMD | frac vs spec | frac vs measured | VALUUtilization | Occupancy | SALU | "
      "fabric MB (algorithmic) | ratio | HBM frac | VGPR / SGPR / LDS |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    p = json.load(open(os.path.join(d, f"pmc_{r}.json")))
    c = {k: v["mean_per_launch"] for k, v in p["pmc_per_launch"].items()}
    ms = p["kernel_stats"]["avg_ns"] * 1e-6
    w, h = (int(x) for x in r.rsplit("_", 1)[1].split("x"))
    alg = w * h * 20 / 1e6
    hbm = p["derived"].get("hbm_bytes_per_launch", 0) / 1e6
    pm = {"source": r, "counters": c, "profiled_kernel_ns": p["kernel_stats"]["avg_ns"]}
    vr = bench.valu_roofline(pm, ms) or {}
    k = p.get("kernel") or {}
    name = (k.get("kernel") or p["kernel_stats"]["name"]).replace("void atmo::", "").replace("(atmo::RenderConsts)", "")
    print(f"| {r} | `{name}` | {ms:.4f} | {c['SQ_INSTS_VALU'] / 1e6:.1f} M | {c['SQ_INSTS_VALU'] / max(c.get('SQ_WAVES', 1), 1):.0f} | "
          f"{ms * 2.4e6 * 1024 / c['SQ_INSTS_VALU']:.2f} | {vr.get('frac_vs_spec', 0):.2f} | {vr.get('frac_vs_measured', 0):.2f} | {c.get('VALUUtilization', 0):.1f} % | "
          f"{c.get('OccupancyPercent', 0):.0f} % | {c.get('SQ_INSTS_SALU', 0) / 1e6:.1f} M | {hbm:.1f} ({alg:.1f}) | {hbm / alg:.2f} | {alg * 1e6 / (ms * 1e-3) / 8e12:.4f} | "
          f"{k.get('vgpr', '?')} / {k.get('sgpr', '?')} / {k.get('lds', '?')} |")
