#!/usr/bin/env python3
"""Benchmark of the atmosphere raymarch hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic frame: BASELINE.json configs[1],
`planet_atmosphere_no_clouds` at 1920x1080 with 32 view steps x 8 light steps (direct light mode), demo
scene, pose P_space, analytic ground-sphere depth buffer, all inputs resident in HBM before the timed
region.  With N > 1 every rank shades its own viewport (weak scaling, BASELINE configs[4] shape: one
viewport per GPU on an orbit of camera poses).  The path has no exchange step, so frames stay resident in
each GPU's HBM (like the inputs) and the timed region ends with ONE RCCL gather of every rank's last frame
to rank 0 (--gather final, default); --gather every gathers each frame (two in flight, overlapped with the
next render; root ingress over xGMI then sets the step time), --gather none skips the collective.
Rank 0 prints ONE JSON line.

Other workloads (--workload): lut32 (reference-exact LUT light, 32 view steps), shipped8 (the shipped
no_clouds shader), clouds_high, clouds_high_rm; --width/--height select the framebuffer (3840x2160 for 4K).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_RAY = 20          # 16 B RGBA32F store + 4 B depth load
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
TIMING_EVERY = 4            # HIP events bracket every 4th kernel launch of the timed region (recording costs ~5 us)
# Measured VALU issue ceilings of MI355X under load (tools/valu_peak.hip, profiles/round1/valu_peak_mi355x.jsonl),
# wave-instructions per second chip-wide: plain f32 FMA/MUL/ADD and transcendental (exp/sqrt/rsq/rcp).
VALU_FMA_WINST_PER_S = 8.68e11
VALU_TRANS_WINST_PER_S = 3.02e11
# rocprofv3 PMC results for the same command line (tools/profile.sh), keyed by (workload, width, height)
PMC_FILES = {
    ("direct32x8", 1920, 1080): "profiles/round1/pmc_final_direct32x8_1920x1080.json",
    ("lut32", 1920, 1080): "profiles/round1/pmc_final_lut32_1920x1080.json",
    ("clouds_high", 1920, 1080): "profiles/round1/pmc_final_clouds_high_1920x1080.json",
    ("clouds_high_rm", 1920, 1080): "profiles/round1/pmc_final_clouds_high_rm_1920x1080.json",
    ("clouds_high_rm", 3840, 2160): "profiles/round1/pmc_final_clouds_high_rm_3840x2160.json",
}


def pmc_summary(workload, w, h):
    """HBM traffic and VALU instruction counts per launch from the committed rocprofv3 PMC passes (collected in
    their own runs, tools/profile.sh); None when this workload/size has not been profiled."""
    path = PMC_FILES.get((workload, w, h))
    if not path or not os.path.exists(os.path.join(ROOT, path)):
        return None
    with open(os.path.join(ROOT, path)) as f:
        d = json.load(f)
    c = {k: v["mean_per_launch"] for k, v in d.get("pmc_per_launch", {}).items()}
    out = {"source": path}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    if "SQ_INSTS_VALU" in c:
        out["valu_winst"] = c["SQ_INSTS_VALU"]
        out["trans_winst"] = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    return out

WORKLOADS = {
    # name: (godot_atmosphere_shader_amd.demo.CONFIGS key, description)
    "direct32x8": ("no_clouds_32x8_direct", "planet_atmosphere_no_clouds, 32 view x 8 light steps (direct light march)"),
    "lut32": ("no_clouds_32_lut", "planet_atmosphere_no_clouds, 32 view steps, baked-LUT light (reference algorithm)"),
    "shipped8": ("no_clouds_8", "planet_atmosphere_no_clouds as shipped: 8 view steps, baked-LUT light"),
    "clouds": ("clouds", "planet_atmosphere_clouds: 8 view, 32 cloud steps"),
    "clouds_high": ("clouds_high", "planet_atmosphere_clouds_high: 8 view, 64 cloud steps, NoiseCubemap coverage"),
    "clouds_high_rm": ("clouds_high_rm", "planet_atmosphere_clouds_high_rm: 8 view, 64 cloud x 6 light steps"),
    # the fast cloud mode (atmo_set_precision 0): fused density expression, error grows with u_cloud_density_scale
    "clouds_high_fast": ("clouds_high", "planet_atmosphere_clouds_high, FAST cloud mode: 8 view, 64 cloud steps"),
    "clouds_high_rm_fast": ("clouds_high_rm", "planet_atmosphere_clouds_high_rm, FAST cloud mode: 8 view, 64 cloud x 6 light steps"),
    "v1_no_clouds": ("v1_no_clouds", "planet_atmosphere_v1_no_clouds (ATMOSPHERE_LITE): 16 view steps"),
    "v1_clouds_high": ("v1_clouds_high", "planet_atmosphere_v1_clouds_high (ATMOSPHERE_LITE): 16 view, 64 cloud steps"),
}


def node_kwargs(workload):
    return dict(precise_clouds=False) if workload.endswith("_fast") else {}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="direct32x8", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--pose", default="P_space")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", default="final", choices=["final", "every", "none"],
                    help="N>1: 'final' = one RCCL gather of every rank's last frame to rank 0 inside the timed region "
                         "(default: the path has no exchange step, frames stay resident like the inputs); "
                         "'every' = gather every frame, two in flight, overlapped with the next render; 'none' = no collective")
    ap.add_argument("--shard", default="viewports", choices=["viewports", "bands"],
                    help="N>1: 'viewports' = one full viewport per GPU (weak scaling, default); 'bands' = ONE viewport cut "
                         "into hit-balanced row bands, one per GPU, gathered in place into the frame on rank 0 (strong scaling)")
    ap.add_argument("--also", default="lut32,shipped8,clouds_high,clouds_high_rm,noise_cubemap",
                    help="comma-separated extra workloads timed at N=1 after the headline and reported under 'extra' "
                         "(SURVEY.md 8d asks for the reference-exact LUT mode and the shipped 8-step shader next to the "
                         "32x8 headline); '' to skip")
    return ap.parse_args()


def usable_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box exposes
    256 logical CPUs but limits the container to 16 CPUs' worth of time; more threads than that only get throttled)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                cores = max(1, min(cores, q // per))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline(config_name, params, textures, cam, depth_np, lut):
    """The oracle (a port of the GDShader, not the reference itself: Godot is a GPU-only path) timed on this
    host's cores over one full frame of the same workload."""
    from godot_atmosphere_shader_amd.demo import CONFIGS, demo_frame
    from oracle.oracle import Oracle

    cores = usable_cores()
    o = Oracle("f32_fast")
    tex = dict(textures, optical_depth=lut)
    frame = demo_frame(cam)
    w, h = cam.width, cam.height
    # bounded sample: every 4th 8-row band of the frame when the full frame would take too long
    t0 = time.perf_counter()
    _, _ = o.render(params, tex, CONFIGS[config_name][1], frame, depth_np, rect=(0, h // 2 - 4, w, h // 2 + 4), nthreads=cores)
    probe = time.perf_counter() - t0
    est_full = probe * h / 8.0
    if est_full <= 30.0:
        rect = (0, 0, w, h)
        sample = f"1 full {w}x{h} frame"
    else:
        rows = max(8, int(h * 20.0 / est_full) // 8 * 8)
        y0 = (h - rows) // 2
        rect = (0, y0, w, y0 + rows)
        sample = f"rows {y0}..{y0 + rows} of one {w}x{h} frame"
    reps, dt, hits = 0, 0.0, 0
    t0 = time.perf_counter()
    while dt < 2.0 and reps < 200:  # repeat the sample until the wall time is measurable on a many-core host
        _, hits = o.render(params, tex, CONFIGS[config_name][1], frame, depth_np, rect=rect, nthreads=cores)
        reps += 1
        dt = time.perf_counter() - t0
    rays = (rect[2] - rect[0]) * (rect[3] - rect[1])
    return {"value": reps * rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x {sample}, {o.precision} build of oracle/atmo_oracle.c, {dt:.2f} s wall, "
                      f"{dt * cores:.0f} core-seconds",
            "hit_fraction": hits / rays}


def bench_noise_cubemap(resolution=256):
    """the NoiseCubemap generator (noise_cubemap.gd:101-140, 'really slow' on the CPU) as a kernel,
    demo-scene settings (resolution 256, scale (100,200,100)); CPU figure = the oracle's scalar loop on one core."""
    from godot_atmosphere_shader_amd import NoiseCubemap, SeededValueNoise
    from oracle.oracle import Oracle

    nz = SeededValueNoise(seed=11, frequency=0.03, fractal_octaves=4, fractal_gain=0.5)
    res = NoiseCubemap(noise=nz, resolution=resolution, scale=(100.0, 200.0, 100.0))
    times = []
    for _ in range(20):
        res._request_update()
        res.process_deferred()
        times.append(res.last_kernel_ms)
    res.close()
    texels = 6 * resolution * resolution
    k_ms = sorted(times)[len(times) // 2]
    o = Oracle("f32")
    t0 = time.perf_counter()
    o.noise_cubemap(resolution, 11, 0.03, 4, 0.5, (100.0, 200.0, 100.0))
    cpu_s = time.perf_counter() - t0
    return {"workload": f"NoiseCubemap 6x{resolution}^2 L8, 4 octaves", "kernel_ms_median": k_ms,
            "Mtexels/s": texels / (k_ms * 1e-3) / 1e6,
            "algorithmic_bytes": texels, "hbm_GBps": texels / (k_ms * 1e-3) / 1e9,
            "cpu_oracle_1core_Mtexels/s": texels / cpu_s / 1e6}


def time_workload(torch, node, cam, depth, steps, warmup, out=None):
    """Single-GPU timed loop; returns (seconds, kernel launches, kernel ms from HIP events)."""
    if out is None:
        out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device=depth.device)
    frame = node.prepare_frame(cam)
    stream = torch.cuda.current_stream().cuda_stream
    dptr, optr = depth.data_ptr(), out.data_ptr()
    for _ in range(warmup):
        node.render_prepared(frame, dptr, optr, stream)
    torch.cuda.synchronize()
    node.set_timing(True, every=TIMING_EVERY)
    t0 = time.perf_counter()
    for _ in range(steps):
        node.render_prepared(frame, dptr, optr, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n, ms = node.get_timing()
    node.set_timing(False)
    return dt, n, ms, out


def timed_loop_distributed(torch, dist, render_into, h, w, device, steps, warmup, gather_mode, timing=None, bands=None):
    """The N > 1 timed region (also run on CPU/gloo by tests/test_distributed_gloo.py with a stub renderer).

    `render_into(buf)` enqueues one frame into the (h, w, 4) float32 tensor `buf`.  W untimed warm-up steps, then
    EXACTLY `steps` steps bracketed by barrier + device synchronisation on both sides; returns (max over ranks of the
    elapsed seconds, kernel launches, kernel ms) -- the last two from `timing = (start, read)` when given.
    gather_mode: "final" (only the last frame is gathered to rank 0, inside the timed region), "every", "none"."""
    from godot_atmosphere_shader_amd.sharding import FrameGather

    is_cuda = device.type == "cuda"

    def sync():
        if is_cuda:
            torch.cuda.synchronize()

    gather = None if gather_mode == "none" else FrameGather(h, w, device, dst=0, depth=2, bands=bands)
    rows = h if bands is None else bands[dist.get_rank()][1] - bands[dist.get_rank()][0]
    scratch = torch.empty((max(rows, 1), w, 4), dtype=torch.float32, device=device)[:rows]

    def step(last):
        # "final": frames stay resident in this GPU's HBM; only the last one is gathered.  "every": each frame.
        if gather is None or (gather_mode == "final" and not last):
            render_into(scratch)
        else:
            buf, slot = gather.next_send_buffer()
            render_into(buf)
            gather.submit(slot)

    if gather is not None:  # set up the communicator / p2p channels outside the timed region whatever W is
        buf, slot = gather.next_send_buffer()
        buf.zero_()
        gather.submit(slot)
        gather.finish()
    for i in range(warmup):
        step(i == warmup - 1)
    if gather is not None:
        gather.finish()
    sync()
    dist.barrier()
    sync()
    if timing is not None:
        timing[0]()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i == steps - 1)
    result = gather.finish() if gather is not None else None
    sync()
    dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    launches, kernel_ms = timing[1]() if timing is not None else (0, 0.0)
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    timed_loop_distributed.last_gathered = result  # (world, h, w, 4) on rank 0 in gather modes, for tests
    return float(tmax.item()), launches, kernel_ms


def main():
    args = parse_args()
    import numpy as np
    import torch

    from godot_atmosphere_shader_amd.demo import demo_params, demo_textures, make_node
    from godot_atmosphere_shader_amd import scene as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("ATMO_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path with a 1-rank group
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    w, h = args.width, args.height
    config_name, desc = WORKLOADS[args.workload]
    textures = demo_textures()
    params = demo_params()
    strong = args.shard == "bands" and (world > 1 or os.environ.get("ATMO_BENCH_FORCE_DIST") == "1")
    pose = args.pose if (world == 1 or rank == 0 or strong) else S.orbit_pose(rank, world)
    cam = S.Camera.from_pose(w, h, pose)
    depth_np = S.depth_ground_sphere(cam)
    depth = torch.from_numpy(depth_np).cuda()
    node = make_node(config_name, textures, params, device=local_rank, **node_kwargs(args.workload))
    rays = w * h

    # ---- timed region ---------------------------------------------------------------------------------
    if world == 1 and not force_dist:
        dt, launches, kernel_ms, out = time_workload(torch, node, cam, depth, args.steps, args.warmup)
        dt_max = dt
        gather_mode = "none (single GPU)"
    else:
        bands = None
        if strong:
            # one frame, row bands balanced by the number of shell-hitting pixels per row (analytic ray/sphere test)
            from godot_atmosphere_shader_amd.sharding import balanced_row_bands, band_rect
            d = cam.pixel_view_dirs()
            d /= np.linalg.norm(d, axis=-1, keepdims=True)
            c = (cam.view @ np.array([0.0, 0.0, 0.0, 1.0]))[:3]
            bq = d @ c
            hit_rows = ((S.DEMO_PLANET_RADIUS + S.DEMO_ATMOSPHERE_HEIGHT) ** 2 - (c @ c - bq * bq) >= 0).sum(axis=1)
            bands = balanced_row_bands(hit_rows + 0.02 * w, world)  # + a small per-row cost for the miss pixels
            frame = node.prepare_frame(cam, rect=band_rect(w, bands[rank]))
        else:
            frame = node.prepare_frame(cam)
        stream = torch.cuda.current_stream().cuda_stream

        def render_into(buf):
            if buf.numel():
                node.render_prepared(frame, depth.data_ptr(), buf.data_ptr(), stream)

        node_timing = (lambda: node.set_timing(True, every=TIMING_EVERY), node.get_timing)
        dt_max, launches, kernel_ms = timed_loop_distributed(
            torch, dist, render_into, h, w, torch.device("cuda", local_rank), args.steps, args.warmup, args.gather, node_timing,
            bands=bands)
        node.set_timing(False)
        gather_mode = {"none": "no collective",
                       "final": "one RCCL gather of each rank's last frame to rank 0, inside the timed region",
                       "every": "RCCL gather of every frame to rank 0, 2 frames in flight, inside the timed region"}[args.gather]

    if rank == 0:
        value = (1 if strong else world) * rays * args.steps / dt_max / 1e6
        pmc = None if strong else pmc_summary(args.workload, w, h)
        kernel_avg_ms = kernel_ms / max(launches, 1)
        launch_rays = rays if not strong else w * (bands[0][1] - bands[0][0])  # rank 0's kernel shades its band only
        achieved_gbs = BYTES_PER_RAY * launch_rays / (kernel_avg_ms * 1e-3) / 1e9
        frame = node.render(cam, depth)
        torch.cuda.synchronize()
        hit_fraction = float((frame.abs().sum(dim=-1) > 0).float().mean().item())
        result = {
            "metric": "Mrays/s at 1920x1080, 32 view x 8 light steps; % HBM roofline",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{desc}; {w}x{h}; demo scene, pose {args.pose}"
                            + ("" if world == 1 else f" on rank 0, orbit poses on ranks 1..{world - 1}; one viewport per GPU"),
                "width": w, "height": h, "rays_per_step_per_gpu": rays,
                "hit_fraction": hit_fraction,
                "mrays_per_s_hit_only": value * hit_fraction,
                "gather": gather_mode,
                "shard": ("one viewport in hit-balanced row bands: " + str(bands)) if strong else "one viewport per GPU",
                "kernel": node.kernel_name,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": None if pmc is None else pmc.get("hbm_bytes"),
                "traffic_source": None if pmc is None else pmc["source"] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bytes per launch)",
                "kernel_avg_ms": kernel_avg_ms,
                "kernel_launches_timed": launches,
                "kernel_timing": f"HIP events around every {TIMING_EVERY}th launch of the timed region, on the launch stream",
                "algorithmic_bytes_per_launch": BYTES_PER_RAY * launch_rays,
                "note": "path is VALU/transcendental-bound, not HBM-bound (20 B/ray); see DESIGN.md",
            },
        }
        if pmc is not None and "valu_winst" in pmc:
            # the resource that actually binds: VALU instruction issue, priced against the measured ceilings
            n_all, n_tr = pmc["valu_winst"], pmc["trans_winst"]
            t_floor = (n_all - n_tr) / VALU_FMA_WINST_PER_S + n_tr / VALU_TRANS_WINST_PER_S
            result["valu_roofline"] = {
                "bound": "valu-issue",
                "valu_wave_insts_per_launch": n_all,
                "transcendental_wave_insts_per_launch": n_tr,
                "achieved_winst_per_s": n_all / (kernel_avg_ms * 1e-3),
                "issue_floor_ms": t_floor * 1e3,
                "frac": t_floor * 1e3 / kernel_avg_ms,
                "peaks": {"fma_winst_per_s": VALU_FMA_WINST_PER_S, "trans_winst_per_s": VALU_TRANS_WINST_PER_S,
                          "source": "tools/valu_peak.hip, profiles/round1/valu_peak_mi355x.jsonl"},
            }
        if world == 1 and args.also:
            extra = {}
            for name in [x for x in args.also.split(",") if x]:
                if name == "noise_cubemap":
                    extra[name] = bench_noise_cubemap()
                    continue
                cfg2, desc2 = WORKLOADS[name]
                node2 = make_node(cfg2, textures, params, device=local_rank, **node_kwargs(name))
                dt2, n2, ms2, _ = time_workload(torch, node2, cam, depth, max(10, args.steps // 4), max(3, args.warmup // 4))
                extra[name] = {"workload": desc2, "Mrays/s": rays * max(10, args.steps // 4) / dt2 / 1e6,
                               "kernel_avg_ms": ms2 / max(n2, 1), "kernel": node2.kernel_name}
                node2.close()
            result["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            from godot_atmosphere_shader_amd.demo import CONFIGS
            ocfg = CONFIGS[config_name][1]
            lut = node.read_optical_depth() if not (ocfg.get("lite") or ocfg.get("light_steps")) else None
            result["cpu_baseline"] = cpu_baseline(config_name, params, textures, cam, depth_np, lut)
        print(json.dumps(result), flush=True)
    node.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
