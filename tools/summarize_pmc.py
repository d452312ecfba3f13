#!/usr/bin/env python3
"""Summarises tools/profile.sh output into one JSON per workload under profiles/<round>/.
usage: tools/summarize_pmc.py gpurun_out/prof_<workload>_<WxH> profiles/round1/<name>.json"""
import collections, csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
out = {"source": os.path.basename(src)}
bid = os.path.join(src, "build_id.txt")
if os.path.exists(bid):   # round 6: what the profiled library was built from (build.source_id)
    out["build_id"] = open(bid).read().strip()
stats = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))
if stats:
    for r in csv.DictReader(open(stats[0])):
        if "atmo_render_kernel" in r["Name"]:
            out["kernel_stats"] = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                   "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
counters = {}
meta = None
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv"))):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "atmo_render_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = dict(kernel=r["Kernel_Name"], grid=int(r["Grid_Size"]), workgroup=int(r["Workgroup_Size"]),
                        vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]))
    for k, v in acc.items():
        counters[k] = {"dispatches": len(v), "mean_per_launch": sum(v) / len(v)}
out["kernel"] = meta
out["pmc_per_launch"] = counters
c = {k: v["mean_per_launch"] for k, v in counters.items()}
d = {}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    d["hbm_bytes_per_launch"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    d["note_fetch"] = "FETCH_SIZE/WRITE_SIZE are in KiB; reads here are 4 B/lane dword loads (depth) + texture gathers, for which the x2 wide-load correction of MI355X_MICROARCH.md does not apply (WRITE_SIZE matches the 16 B/ray store stream exactly)"
if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
    d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
if "SQ_INSTS_VALU_TRANS_F32" in c and "SQ_INSTS_VALU" in c:
    d["trans_fraction"] = c["SQ_INSTS_VALU_TRANS_F32"] / c["SQ_INSTS_VALU"]
if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
    d["SQ_ACTIVE_INST_VALU_over_SQ_BUSY_CYCLES"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"]
if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
    d["wait_inst_any_frac_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    d["wait_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
    d["active_inst_any_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
if "TCC_HIT_sum" in c:
    d["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
if "TCP_TCC_READ_REQ_sum" in c and "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
    d["l1_miss_ratio_reqs"] = c["TCP_TCC_READ_REQ_sum"] / max(c["TCP_TOTAL_CACHE_ACCESSES_sum"], 1)
for k in ("VALUBusy", "VALUUtilization", "OccupancyPercent", "MemUnitStalled"):
    if k in c:
        d[k + "_derived_gfx94x_formula"] = c[k]
if "kernel_stats" in out and "SQ_INSTS_VALU" in c:
    d["valu_wave_insts_per_s"] = c["SQ_INSTS_VALU"] / (out["kernel_stats"]["avg_ns"] * 1e-9)
out["derived"] = d
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
if "kernel_stats" in out:
    print(out["kernel_stats"])
