#!/usr/bin/env python3
"""Where does a draw spend its time?  Needs the diagnostic build (every wave records start / end in 100 MHz ticks and its
hardware slot):

    tools/ab_build.sh trace -DATMO_WAVE_TRACE=1
    gpurun -- 'python tools/wave_timeline.py [workload] [W H] [pose]'      # workload as in bench.py --workload

Prints: kernel span, wave-duration distribution, resident waves per SIMD over time (text plot), the time to fill the chip
(ramp), the time from "90 % of the wave-time done" to the end (tail), per-XCD finish times, and how long SIMDs sit with
fewer than 2 resident waves (no partner for the dual-issue of fast-class instructions)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("ATMO_HIP_LIB", os.path.join(ROOT, "godot_atmosphere_shader_amd", "libatmo_hip_trace.so"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_params, demo_textures, make_node  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "direct32x8"
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
    pose = sys.argv[4] if len(sys.argv) > 4 else "P_space"
    config_name, desc = bench.WORKLOADS[wl]
    node = make_node(config_name, demo_textures(), demo_params(), **bench.node_kwargs(wl))
    cam = S.Camera.from_pose(w, h, pose)
    depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
    out = None
    for _ in range(int(os.environ.get("TIMELINE_FRAMES", "12"))):  # paced like a frame loop: the tile-order feedback swaps a new order in when the HOST sees its sort done
        out = node.render(cam, depth, out=out)
        torch.cuda.synchronize()
    fn = node._lib.atmo_debug_wave_trace
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    max_waves = 4 * ((w + 15) // 16) * ((h + 3) // 4)
    buf = np.zeros((max_waves, 4), dtype=np.uint64)
    n = fn(node._ctx, buf.ctypes.data_as(C.c_void_p), max_waves)
    assert n > 0, "no trace: is this the -DATMO_WAVE_TRACE build?"
    t = buf[:n]
    t = t[t[:, 1] > 0]
    t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
    hw, xcc = t[:, 2].astype(np.int64), t[:, 3].astype(np.int64) & 0xF
    pre_us = (t[:, 3].astype(np.int64) >> 8) / 100.0  # kernel entry -> start of the per-pixel work (kernel-argument loads, tile index)
    origin = t0.min()
    s_us, e_us = (t0 - origin) / 100.0, (t1 - origin) / 100.0
    span = e_us.max()
    dur = e_us - s_us
    simd_key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 11) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)
    n_simd = len(np.unique(simd_key))
    print(f"{wl} {w}x{h} {pose}: kernel {node.kernel_name}; {len(t)} waves that reached the end of the kernel on {n_simd} SIMDs, "
          f"span {span:.1f} us (first wave start to last wave end)")
    q = np.percentile(dur, [5, 25, 50, 75, 95, 99, 100])
    print("wave duration us: p5 %.1f  p25 %.1f  p50 %.1f  p75 %.1f  p95 %.1f  p99 %.1f  max %.1f" % tuple(q))
    print("preamble (entry -> per-pixel work) us: p50 %.2f  p95 %.2f  max %.2f" % tuple(np.percentile(pre_us, [50, 95, 100])))
    print(f"wave-time total {dur.sum() / 1e3:.1f} ms = {dur.sum() / span / n_simd:.2f} resident waves per SIMD on average")
    # resident waves over time
    bins = 50
    edges = np.linspace(0.0, span, bins + 1)
    resident = np.zeros(bins)
    for b in range(bins):
        lo, hi = edges[b], edges[b + 1]
        resident[b] = np.clip(np.minimum(e_us, hi) - np.maximum(s_us, lo), 0.0, None).sum() / (hi - lo) / n_simd
    peak = resident.max()
    print("resident waves per SIMD over time (each row %.1f us):" % (span / bins))
    for b in range(bins):
        print(f"  {edges[b]:7.1f} us  {resident[b]:5.2f}  " + "#" * int(round(resident[b] * 8)))
    filled = np.argmax(resident >= 0.9 * peak)
    print(f"ramp: {edges[filled]:.1f} us until 90 % of the peak residency ({peak:.2f} waves per SIMD)")
    order = np.argsort(e_us)
    cum = np.cumsum(dur[order]) / dur.sum()
    t90 = e_us[order][np.searchsorted(cum, 0.9)]
    print(f"tail: 90 % of the wave-time has ended by {t90:.1f} us; the remaining 10 % takes until {span:.1f} us "
          f"({(span - t90) / span * 100:.0f} % of the span)")
    for x in np.unique(xcc):
        m = xcc == x
        print(f"  XCD {x}: {m.sum():6d} waves, last end {e_us[m].max():7.1f} us, wave-time {dur[m].sum() / 1e3:7.2f} ms")
    # per-SIMD: time with no / one resident wave, counting a wave from kernel ENTRY (resident) or from the start of its
    # per-pixel work (past the preamble), and when each SIMD ran out of work
    entry_us = s_us - pre_us
    keys = np.unique(simd_key)[:: max(1, n_simd // 256)]
    for label, start in (("from kernel entry", entry_us), ("past the preamble", s_us)):
        zero = one = 0.0
        for key in keys:
            m = simd_key == key
            ev = np.concatenate([np.stack([start[m], np.ones(m.sum())], 1), np.stack([e_us[m], -np.ones(m.sum())], 1)])
            ev = ev[np.argsort(ev[:, 0], kind="stable")]
            level, last = 0, min(0.0, float(ev[0, 0]))
            for tt, d in ev:
                if level == 0:
                    zero += tt - last
                elif level == 1:
                    one += tt - last
                level += int(d)
                last = tt
            zero += span - last
        print(f"per SIMD, waves counted {label}: {zero / len(keys):5.1f} us with none, {one / len(keys):5.1f} us with exactly one "
              f"(of {span:.1f} us)")
    ends = np.array([e_us[simd_key == key].max() for key in keys])
    print("SIMD runs out of work at us: p5 %.1f  p25 %.1f  p50 %.1f  p75 %.1f  p95 %.1f  max %.1f" %
          tuple(np.percentile(ends, [5, 25, 50, 75, 95, 100])))
    # who is still running during the drain: start time, duration and dispatch rank of the waves that end in the last tenth of the span
    late = e_us >= 0.9 * span
    if late.any():
        rank = np.argsort(np.argsort(entry_us))  # dispatch order
        print(f"waves ending in the last tenth of the span: {late.sum()} of {len(t)}; they entered at us p5 %.1f p50 %.1f p95 %.1f; "
              "duration us p5 %.1f p50 %.1f p95 %.1f; dispatch rank (0 = first) p5 %.0f p50 %.0f p95 %.0f"
              % (tuple(np.percentile(entry_us[late], [5, 50, 95])) + tuple(np.percentile(dur[late], [5, 50, 95]))
                 + tuple(np.percentile(rank[late], [5, 50, 95]))))
        heavy = dur >= np.percentile(dur, 90)
        print("the heaviest tenth of the waves: duration us p50 %.1f; entered at us p5 %.1f p50 %.1f p95 %.1f max %.1f; ended at us p50 %.1f p95 %.1f max %.1f"
              % ((np.percentile(dur[heavy], 50),) + tuple(np.percentile(entry_us[heavy], [5, 50, 95, 100])) + tuple(np.percentile(e_us[heavy], [50, 95, 100]))))
        for lo_, hi_ in ((0.0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 0.9), (0.9, 1.0)):
            m = (entry_us >= lo_ * span) & (entry_us < hi_ * span)
            if m.any():
                print(f"  waves entering in {lo_ * 100:3.0f}-{hi_ * 100:3.0f} % of the span: {m.sum():6d}, duration us p50 {np.percentile(dur[m], 50):6.1f} "
                      f"p95 {np.percentile(dur[m], 95):6.1f} max {dur[m].max():6.1f}")
    # is the launch order sorted by what the waves turn out to cost?  (slot = launch rank: 2 waves per workgroup, blockIdx order)
    if os.environ.get("TIMELINE_RANKS"):
        slot = np.nonzero(buf[:n, 1] > 0)[0]
        nb = 24
        edges_r = np.linspace(0, slot.max() + 1, nb + 1).astype(int)
        print("launch rank (wave slots) -> entry us p50 | duration us p5 p50 p95 max | end us max")
        for b in range(nb):
            m = (slot >= edges_r[b]) & (slot < edges_r[b + 1])
            if m.any():
                print(f"  {edges_r[b]:6d}-{edges_r[b + 1]:6d}: entry {np.percentile(entry_us[m], 50):7.1f} | "
                      "%7.1f %7.1f %7.1f %7.1f | %7.1f" % (tuple(np.percentile(dur[m], [5, 50, 95, 100])) + (e_us[m].max(),)))
        last = np.argsort(-e_us)[:24]
        print("the 24 waves that end last: slot, entry us, duration us, end us")
        for i in last:
            print(f"  {slot[i]:6d}  {entry_us[i]:7.1f}  {dur[i]:7.1f}  {e_us[i]:7.1f}")
    lifetime = e_us - entry_us
    print(f"wave lifetime from entry: mean {lifetime.mean():.1f} us = {lifetime.sum() / span / n_simd:.2f} resident waves per SIMD; "
          f"of which past the preamble {dur.sum() / lifetime.sum() * 100:.0f} %")
    node.close()


if __name__ == "__main__":
    main()
