#!/usr/bin/env python3
"""Rows of DESIGN.md section 5.2 from a profiles/<round>/ directory (pmc_*.json of tools/summarize_pmc.py + bench_default.json).
usage: tools/design_table.py profiles/round3"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/round3"
bench = json.loads(open(os.path.join(d, "bench_default.json")).read().strip().splitlines()[-1])
live = {"direct32x8_1920x1080": bench}
for k, v in bench.get("extra", {}).items():
    if isinstance(v, dict) and v.get("valu_roofline") and "reforder" not in k:
        name, *opts = k.split("@")
        size = "1920x1080"
        lod = ""
        for o in opts:
            if o == "lod":
                lod = "@lod"
            elif "x" in o:
                size = o
        live[f"{name}{lod}_{size}"] = v
rows = ["direct32x8_1920x1080", "direct32x8_3840x2160", "lut32_1920x1080", "shipped8_1920x1080", "clouds_high_1920x1080", "clouds_high_rm_1920x1080",
        "clouds_high_rm_3840x2160", "clouds_high@lod_1920x1080", "clouds_high_rm@lod_3840x2160"]
print("| workload | kernel ms (rocprof avg) | bench.py kernel ms | VALU wave-insts | cycles/inst/SIMD | frac vs spec | frac vs measured | VALUUtilization | SALU | fabric MB (algorithmic) | ratio | HBM frac |")
for r in rows:
    p = json.load(open(os.path.join(d, f"pmc_{r}.json")))
    c = {k: v["mean_per_launch"] for k, v in p["pmc_per_launch"].items()}
    ms = p["kernel_stats"]["avg_ns"] * 1e-6
    w, h = (int(x) for x in r.rsplit("_", 1)[1].split("x"))
    alg = w * h * 20 / 1e6
    hbm = p["derived"].get("hbm_bytes_per_launch", 0) / 1e6
    b = live.get(r, {})
    vr = b.get("valu_roofline") or {}
    bms = b.get("roofline", {}).get("kernel_avg_ms", b.get("kernel_avg_ms"))
    print(f"| {r} | {ms:.4f} | {bms if bms is None else round(bms, 4)} | {c['SQ_INSTS_VALU'] / 1e6:.1f} M | {ms * 2.4e6 * 1024 / c['SQ_INSTS_VALU']:.2f} | "
          f"{vr.get('frac_vs_spec', 0):.2f} | {vr.get('frac_vs_measured', 0):.2f} | {c.get('VALUUtilization', 0):.1f} % | {c.get('SQ_INSTS_SALU', 0) / 1e6:.1f} M | "
          f"{hbm:.1f} ({alg:.1f}) | {hbm / alg:.2f} | {alg * 1e6 / (ms * 1e-3) / 8e12:.4f} |")
