#!/usr/bin/env python3
"""Files the outputs of one `tools/r6_final.sh` run (gpurun_out/) as the committed evidence of a round and regenerates every table and sentence of README.md /
DESIGN.md that quotes them, so that the documents cannot drift from the files:
    python tools/file_evidence.py round6"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "round6"
src, fin, dst = os.path.join(ROOT, "gpurun_out", f"profiles_{rnd}"), os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles", rnd)
for f in glob.glob(os.path.join(src, "*")):
    shutil.copy(f, dst)
shutil.copy(os.path.join(ROOT, "gpurun_out", "every_pixel_4k.txt"), os.path.join(dst, "every_pixel_4k.txt"))
shutil.copy(os.path.join(fin, "bench_default_line.json"), os.path.join(dst, "bench_final_default_line.json"))
open(os.path.join(dst, "bench_k20_line.json"), "w").write(open(os.path.join(fin, "bench_driver_cmd_stdout.txt")).read().strip().split("\n")[-1] + "\n")
open(os.path.join(dst, "smoke.txt"), "w").write("".join(ln for ln in open(os.path.join(fin, "smoke.txt")) if "amdgpu.ids" not in ln))
open(os.path.join(dst, "gputest_tail.txt"), "w").write("".join(open(os.path.join(fin, "pytest_gpu.txt")).readlines()[-8:]))

full = json.load(open(os.path.join(dst, "bench_default.json")))
k20 = json.load(open(os.path.join(dst, "bench_k20_line.json")))
fin_line = json.load(open(os.path.join(dst, "bench_final_default_line.json")))
ids = {json.load(open(f)).get("build_id") for f in glob.glob(os.path.join(dst, "pmc_*.json"))}
assert len(ids) == 1 and full["roofline"]["build_id"] in ids and not full["roofline"]["traffic_stale"], (ids, full["roofline"])
build_id = ids.pop()
g = lambda v: f"{v:,.0f}".replace(",", " ")

print(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "readme_table.py"), os.path.join("profiles", rnd), "--apply"], cwd=ROOT, check=True, capture_output=True, text=True).stdout.strip().split("\n")[-1])

# the round's kernel table (the ten rows behind the paragraph that names the build id), the build id, the sentence of section 6
path = os.path.join(ROOT, "DESIGN.md")
text = open(path).read()
rows = [ln for ln in subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_table.py"), os.path.join("profiles", rnd)], cwd=ROOT, check=True, capture_output=True,
                                    text=True).stdout.split("\n") if ln.startswith("| ") and "atmo_render_kernel" in ln]
assert len(rows) == 10
lines = text.split("\n")
start = next(i for i, ln in enumerate(lines) if "every `pmc_*.json` carries the build id" in ln)
n, j = 0, start
while n < 10:
    j += 1
    if lines[j].startswith("| ") and "atmo_render_kernel" in lines[j]:
        lines[j] = rows[n]
        n += 1
text = "\n".join(lines)
old_ids = set(re.findall(r"build id\n?`([0-9a-f]{16})`", text)) | set(re.findall(r"committed build `([0-9a-f]{16})`", text))
for o in old_ids:
    text = text.replace(o, build_id)
r, v = full["roofline"], full["valu_roofline"]
stats = open(os.path.join(dst, "kernel_stats_direct32x8_1920x1080.csv")).read()
m = re.search(r'atmo_render_kernel<4, 8, 1>[^\n"]*",(\d+),(\d+),([\d.]+)', stats)
rocprof_ms = float(m.group(3)) / 1e6 if m else None
sentence = (f"Round 6 (`profiles/round6/bench_default_line.json`, `bench_k20_line.json`): {g(full['value'])} Mrays/s at K = 200 ({g(fin_line['value'])} in the round-end check's second run), "
            f"{g(k20['value'])} at the driver's K = 20, kernel {r['kernel_avg_ms']:.4f} ms" + (f" (rocprofv3 {rocprof_ms:.4f})" if rocprof_ms else "") +
            f", `roofline.frac` {r['frac']:.4f}, traffic {r['traffic'] / 1e6:.1f} MB = {r['traffic'] / r['algorithmic_bytes_per_launch']:.3f}× the algorithmic {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB "
            f"(stamped counters: `traffic_stale` false for the committed build `{build_id}`), `valu_roofline` {v['frac_vs_spec']:.2f} / {v['frac_vs_measured']:.2f}, "
            f"`cpu_baseline` {min(k20['cpu_baseline']['value'], full['cpu_baseline']['value'], fin_line['cpu_baseline']['value']):.1f}–{max(k20['cpu_baseline']['value'], full['cpu_baseline']['value'], fin_line['cpu_baseline']['value']):.1f} Mrays/s on 16 cores.")
text, k = re.subn(r"Round 6 \(`profiles/round6/bench_default_line\.json`, `bench_k20_line\.json`\):.*?Mrays/s on 16 cores\.", lambda _: sentence, text, flags=re.S)
assert k == 1
# the comparison with round 5's table (section 5.6: rocprofv3 averages 0.1782 / 0.4218 / 1.2232 ms)
def avg_ms(workload, kernel):
    t = open(os.path.join(dst, f"kernel_stats_{workload}.csv")).read()
    return float(re.search(re.escape(kernel) + r'[^\n"]*",\d+,\d+,([\d.]+)', t).group(1)) / 1e6
d49, d51, d51k = (avg_ms("clouds_high_1920x1080", "atmo_render_kernel<49, 0, 1>") / 0.1782 - 1) * 100, (avg_ms("clouds_high_rm_1920x1080", "atmo_render_kernel<51, 0, 1>") / 0.4218 - 1) * 100, \
    (avg_ms("clouds_high_rm_3840x2160", "atmo_render_kernel<51, 0, 1>") / 1.2232 - 1) * 100
fmt = lambda x: f"{x:+.1f} %".replace("-", "−")
text, k = re.subn(r"`<49, 0, 1>` [+−][\d.]+ %,\n`<51, 0, 1>` 1080p [+−][\d.]+ % \(tail-bound, ±2 % between boxes\), 4K [+−][\d.]+ %",
                  f"`<49, 0, 1>` {fmt(d49)},\n`<51, 0, 1>` 1080p {fmt(d51)} (tail-bound, ±2 % between boxes), 4K {fmt(d51k)}", text)
assert k == 1
open(path, "w").write(text)
p = os.path.join(dst, "README.md")
t = open(p).read()
t = re.sub(r"stamped with the build id of the profiled library \(`[0-9a-f]{16}`\)", f"stamped with the build id of the profiled library (`{build_id}`)", t)
open(p, "w").write(t)
print(f"DESIGN.md: kernel table, build id {build_id} and the section-6 sentence regenerated; rocprofv3 average of the headline kernel {rocprof_ms}")
