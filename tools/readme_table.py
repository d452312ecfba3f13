#!/usr/bin/env python3
"""The results table of README.md from profiles/round<N>/bench_default.json (K = 200 run, detail) and bench_k20_line.json (the driver's command):
    python tools/readme_table.py profiles/round6"""
import json
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/round6"
full = json.load(open(f"{d}/bench_default.json"))
k20 = json.load(open(f"{d}/bench_k20_line.json"))
ex, ek = full["extra"], k20["extra"]


def g(v):
    return f"{v:,.0f}".replace(",", " ")


def vr(e):
    v = e.get("valu_roofline") or {}
    return f"{v['frac_vs_spec']:.2f} / {v['frac_vs_measured']:.2f}" if v else ""


def mot(e, key):
    return e[key]["feedback_on"]["Mrays/s"]


rows = [("**headline** direct32x8 1920x1080", k20["value"], full["value"], full["roofline"]["kernel_avg_ms"], vr(full), f"feedback off {g(full['config']['mrays_per_s_feedback_off'])}")]
m = ex["direct32x8@moving"]
rows.append(("  moving 0.1 / 1 / 5 orbit; pan 1", " / ".join(g(ek[f"direct32x8@moving/{k}"][0]) for k in ("orbit:0.1", "orbit:1", "orbit:5")) + "; " + g(ek["direct32x8@moving/pan:1"][0]),
             " / ".join(g(mot(m, k)) for k in ("orbit:0.1", "orbit:1", "orbit:5")) + "; " + g(mot(m, "pan:1")), "", "", ""))
for name in ("direct32x8@3840x2160", "lut32", "shipped8", "clouds_high", "clouds_high@lod0", "clouds_high_rm", "clouds_high_rm@lod0", "clouds_high_rm@3840x2160",
             "clouds_high_rm@lod0@3840x2160", "clouds_high_rm@1280x720", "clouds_high_rm@1280x720@nosplit", "clouds_high_rm@P_limb", "clouds_high_rm@P_limb@nosplit",
             "shipped8@cleared", "lut32@cleared", "direct32x8@reforder", "direct32x8+2vp"):
    e = ex[name]
    rows.append((name, ek[name][0], e["Mrays/s"], e.get("kernel_avg_ms") or 0.0, vr(e), ""))
m = ex["clouds_high_rm@moving"]
rows.append(("clouds_high_rm moving static / 1 / 5 orbit; pan 1", " / ".join(g(ek[f"clouds_high_rm@moving/{k}"][0]) for k in ("static", "orbit:1", "orbit:5")) + "; " + g(ek["clouds_high_rm@moving/pan:1"][0]),
             " / ".join(g(mot(m, k)) for k in ("static", "orbit:1", "orbit:5")) + "; " + g(mot(m, "pan:1")), "", "", ""))
rows.append(("cpu_baseline", k20["cpu_baseline"]["value"], full["cpu_baseline"]["value"], "", "", f"{full['cpu_baseline']['cores']} cores"))
for r in rows:
    a, b = (g(r[1]) if isinstance(r[1], float) else r[1]), (g(r[2]) if isinstance(r[2], float) else r[2])
    k = f"{r[3]:.4f}" if isinstance(r[3], float) and r[3] else ""
    print(f"| {r[0]} | {a} | {b} | {k} | {r[4]} | {r[5]} |")
print("roofline", {k: full["roofline"].get(k) for k in ("frac", "traffic", "traffic_stale", "build_id", "algorithmic_bytes_per_launch")}, "noise_cubemap", ex["noise_cubemap"]["Mtexels/s"])
