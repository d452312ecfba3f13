"""Child process of tests/test_distributed_gpu.py: a 1-rank `nccl` (= RCCL) group opened BEFORE anything touches the GPU, then bench.py's N > 1
timed region (timed_loop_distributed + FrameGather) with the HIP kernels filling the send buffers -- what `bench.py --gpus N` runs on every
rank -- and the gathered frame checked bit for bit against a direct render.  Prints one JSON line; exit code 0 = all checks passed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))   # first: no HIP call has happened yet
    import bench
    from godot_atmosphere_shader_amd import scene as S
    from godot_atmosphere_shader_amd.demo import demo_params, demo_textures, make_node
    from godot_atmosphere_shader_amd.sharding import balanced_row_bands

    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    w, h = 640, 360
    tex, params = demo_textures(cube_n=64, shape_n=32), demo_params()
    cam = S.Camera.from_pose(w, h, "P_space")
    depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
    report = {}
    for workload in ("no_clouds_32x8_direct", "clouds_high_rm"):
        node = make_node(workload, tex, params, device=0)
        want = node.render(cam, depth).clone()
        torch.cuda.synchronize()
        frame = node.prepare_frame(cam)
        stream = torch.cuda.current_stream().cuda_stream

        def render_into(buf):
            node.render_prepared(frame, depth.data_ptr(), buf.data_ptr(), stream)

        for mode in ("final", "every", "none"):
            timing = (lambda: node.set_timing(True, every=2), node.get_timing)
            dt, launches, kernel_ms = bench.timed_loop_distributed(torch, dist, render_into, h, w, device, 6, 2, mode, timing)
            node.set_timing(False)
            got = bench.timed_loop_distributed.last_gathered
            assert dt > 0.0 and launches == 3 and kernel_ms > 0.0, (mode, dt, launches, kernel_ms)
            if mode == "none":
                assert got is None
            else:
                assert tuple(got.shape) == (1, h, w, 4) and torch.equal(got[0], want), (workload, mode)
            report[f"{workload}/{mode}"] = dt
        # one viewport in row bands (strong-scaling shape): a 1-rank group holds the only band; gathered in place
        bands = balanced_row_bands(node.measure_row_costs(cam, depth), 1)
        assert bands == [(0, h)]
        dt, _, _ = bench.timed_loop_distributed(torch, dist, render_into, h, w, device, 4, 1, "every", None, bands=bands)
        got = bench.timed_loop_distributed.last_gathered
        assert tuple(got.shape) == (h, w, 4) and torch.equal(got, want), workload
        report[f"{workload}/bands-every"] = dt
        node.close()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"ok": True, "backend": "nccl", "seconds": report}))


if __name__ == "__main__":
    main()
