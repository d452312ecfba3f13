"""world_size-2 gloo test of the multi-GPU path's host logic (sharding + gather), on CPU.

The renderer is stubbed by the CPU oracle here (tests may use it); on the GPU box the same FrameGather runs
over RCCL with the HIP kernels filling the send buffers (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from common import CONFIGS, demo_frame, demo_params, demo_textures
        from godot_atmosphere_shader_amd import scene as S
        from godot_atmosphere_shader_amd.sharding import FrameGather, row_bands
        from oracle.oracle import Oracle

        o = Oracle("f32")
        w, h = 48, 27  # 27 rows over 2 ranks: unequal bands (13 + 14) exercise the padded path
        tex, params = demo_textures(cube_n=16, shape_n=8), demo_params()
        lut = o.bake_optical_depth(100.0, 8.0, 0.5)
        cfg = CONFIGS["clouds"][1]
        dev = torch.device("cpu")

        def render(pose, rect=None):
            cam = S.Camera.from_pose(w, h, pose)
            img, _ = o.render(params, dict(tex, optical_depth=lut), cfg, demo_frame(cam), S.depth_ground_sphere(cam), rect=rect)
            return torch.from_numpy(img)

        if mode == "viewports":
            poses = [S.orbit_pose(k, world) for k in range(world)]
            g = FrameGather(h, w, dev, dst=0, depth=2)
            for it in range(3):  # more submissions than slots: exercises buffer recycling
                buf, slot = g.next_send_buffer()
                buf.copy_(render(poses[rank]))
                g.submit(slot)
            res = g.finish()
            if rank == 0:
                assert res.shape == (world, h, w, 4)
                for k in range(world):
                    assert torch.equal(res[k], render(poses[k]))
            else:
                assert res is None
        else:
            hgt = 28 if mode == "bands_equal" else h
            bands = row_bands(hgt, world)
            cam_pose = "P_limb"

            def render_h(rect=None):
                cam = S.Camera.from_pose(w, hgt, cam_pose)
                img, _ = o.render(params, dict(tex, optical_depth=lut), cfg, demo_frame(cam), S.depth_ground_sphere(cam), rect=rect)
                return torch.from_numpy(img)

            g = FrameGather(hgt, w, dev, dst=0, bands=bands, depth=2)
            for it in range(2):
                buf, slot = g.next_send_buffer()
                y0, y1 = bands[rank]
                buf.copy_(render_h(rect=(0, y0, w, y1)))
                g.submit(slot)
            res = g.finish()
            if rank == 0:
                assert torch.equal(res, render_h())
        with open(os.path.join(tmpdir, f"ok_{mode}_{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["viewports", "bands_equal", "bands_unequal"])
def test_gather_world_size_2(tmp_path, mode):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"ok_{mode}_{r}").exists()


def _bench_worker(rank, world, port, mode, tmpdir):
    """Runs bench.py's N > 1 timed region itself (timed_loop_distributed) over gloo with a stub renderer."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench

        h, w = 6, 5
        calls = {"n": 0}

        def render_into(buf):
            calls["n"] += 1
            buf.fill_(float(100 * rank + calls["n"]))  # frame k of rank r is the constant 100 r + k

        started = []
        timing = (lambda: started.append(calls["n"]), lambda: (7, 1.5))
        steps, warmup = 4, 2
        bands = None
        if mode == "bands":   # strong scaling: one frame in unequal row bands, gathered in place
            bands = [(0, 2), (2, 6)]
            mode = "every"
        dt, launches, kernel_ms = bench.timed_loop_distributed(
            torch, dist, render_into, h, w, torch.device("cpu"), steps, warmup, mode, timing, bands=bands)
        assert dt > 0 and (launches, kernel_ms) == (7, 1.5)
        assert calls["n"] == steps + warmup            # exactly K timed + W warm-up frames rendered
        assert started == [warmup]                      # timing starts after the warm-up steps
        res = bench.timed_loop_distributed.last_gathered
        if bands is not None:
            if rank == 0:
                assert res.shape == (h, w, 4)
                for r, (y0, y1) in enumerate(bands):
                    assert torch.all(res[y0:y1] == float(100 * r + steps + warmup))
            mode = "bands"
        elif mode == "none":
            assert res is None
        elif rank == 0:
            assert res.shape == (world, h, w, 4)
            for r in range(world):                      # the last frame of every rank arrived on rank 0
                assert torch.all(res[r] == float(100 * r + steps + warmup))
        else:
            assert res is None
        with open(os.path.join(tmpdir, f"ok_bench_{mode}_{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["final", "every", "none", "bands"])
def test_bench_distributed_timed_loop_world_size_2(tmp_path, mode):
    world = 2
    port = _free_port()
    mp.spawn(_bench_worker, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"ok_bench_{mode}_{r}").exists()


def _bench_worker4(rank, world, port, mode, tmpdir):
    """World size 4: an EMPTY band (more ranks than rows would give one too) and the band mode with gather="final"."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from godot_atmosphere_shader_amd.sharding import row_bands

        w = 5
        if mode == "empty_band":
            h, bands, gather = 3, row_bands(3, world), "every"      # [(0,0),(0,1),(1,2),(2,3)]: rank 0's band is empty
            assert bands[0] == (0, 0)
        elif mode == "bands_final":
            h, bands, gather = 9, [(0, 1), (1, 5), (5, 5), (5, 9)], "final"   # unequal, one empty in the middle
        else:
            h, bands, gather = 4, None, "every"                       # four viewports
        calls = {"n": 0}

        def render_into(buf):
            calls["n"] += 1
            if buf.numel():
                buf.fill_(float(100 * rank + calls["n"]))

        steps, warmup = 3, 1
        dt, launches, kernel_ms = bench.timed_loop_distributed(
            torch, dist, render_into, h, w, torch.device("cpu"), steps, warmup, gather, None, bands=bands)
        assert dt > 0 and calls["n"] == steps + warmup
        res = bench.timed_loop_distributed.last_gathered
        if rank == 0:
            last = steps + warmup
            if bands is None:
                assert res.shape == (world, h, w, 4)
                for r in range(world):
                    assert torch.all(res[r] == float(100 * r + last))
            else:
                assert res.shape == (h, w, 4)
                for r, (y0, y1) in enumerate(bands):
                    assert torch.all(res[y0:y1] == float(100 * r + last)), (r, y0, y1)
        else:
            assert res is None
        with open(os.path.join(tmpdir, f"ok4_{mode}_{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["viewports", "empty_band", "bands_final"])
def test_bench_distributed_timed_loop_world_size_4(tmp_path, mode):
    world = 4
    port = _free_port()
    mp.spawn(_bench_worker4, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"ok4_{mode}_{r}").exists()


def _bench_worker8(rank, world, port, tmpdir):
    """BASELINE configs[4]'s shape: eight independent 3840 x 2160 RGBA32F viewports, one per rank, gathered to rank 0 every frame
    (132.7 MB per rank and frame), bench.py's own timed loop; the root renders into its slot of the receive buffer."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from godot_atmosphere_shader_amd.sharding import FrameGather

        h, w = 2160, 3840
        calls = {"n": 0, "ptrs": set()}

        def render_into(buf):
            calls["n"] += 1
            calls["ptrs"].add(buf.data_ptr())
            buf[::540, ::960].fill_(float(100 * rank + calls["n"]))   # a sparse signature: the test moves 1 GB per gather as it is

        steps, warmup = 1, 0   # plus the loop's own set-up gather: 2 x 1 GB through gloo
        dt, _, _ = bench.timed_loop_distributed(torch, dist, render_into, h, w, torch.device("cpu"), steps, warmup, "every", None)
        assert dt > 0 and calls["n"] == steps + warmup
        res = bench.timed_loop_distributed.last_gathered
        if rank == 0:
            assert res.shape == (world, h, w, 4) and res.element_size() * res[0].numel() == 132710400
            for r in range(world):
                assert torch.all(res[r, ::540, ::960] == float(100 * r + steps + warmup)), r
            # the root's frames were rendered inside the receive buffers: no send buffer, no self-copy
            g = FrameGather(h, w, torch.device("cpu"), dst=0, depth=1)
            buf, slot = g.next_send_buffer()
            assert buf.data_ptr() == g.recv[slot][0].data_ptr()
        else:
            assert res is None
        with open(os.path.join(tmpdir, f"ok8_{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_bench_distributed_timed_loop_world_size_8_config4_shape(tmp_path):
    world = 8
    port = _free_port()
    mp.spawn(_bench_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"ok8_{r}").exists()


def _bands_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench

        calls = {"n": 0}

        def measure():   # only rank 0 may be asked; other ranks would produce other numbers (measured costs differ per GPU)
            assert rank == 0
            calls["n"] += 1
            rows = np.full(100, 1.0)
            rows[60:80] = 10.0 * calls["n"]   # the last measurement counts
            return rows

        bands = bench.cut_bands_measured(torch, dist, torch.device("cpu"), world, rank, measure)
        assert calls["n"] == (3 if rank == 0 else 0)
        rows = np.full(100, 1.0)
        rows[60:80] = 30.0
        from godot_atmosphere_shader_amd.sharding import balanced_row_bands
        assert bands == balanced_row_bands(rows, world)          # every rank holds rank 0's cuts
        assert bands[0][0] == 0 and bands[-1][1] == 100
        with open(os.path.join(tmpdir, f"okb_{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_measured_band_cuts_are_made_on_rank_0_and_broadcast(tmp_path):
    world = 3
    port = _free_port()
    mp.spawn(_bands_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"okb_{r}").exists()


def test_measured_row_costs_cut_bands_of_equal_work():
    """bench.py --shard bands --band-cost measured: per-row sums of measured tile costs (atmo_measure_tile_costs) cut the frame;
    here with a synthetic cost map shaped like a cloud frame (a heavy blob off-centre on a cheap background)."""
    sys.path.insert(0, ROOT)
    from godot_atmosphere_shader_amd.sharding import balanced_row_bands

    gy, gx, th = 135, 120, 8
    yy, xx = np.mgrid[0:gy, 0:gx]
    cost = 2000.0 + 60000.0 * np.exp(-(((yy - 40) / 12.0) ** 2 + ((xx - 70) / 25.0) ** 2))
    rows = np.repeat(cost.sum(axis=1) / th, th)[:1080]
    for world in (2, 4, 8):
        bands = balanced_row_bands(rows, world)
        assert bands[0][0] == 0 and bands[-1][1] == 1080 and all(bands[k][1] == bands[k + 1][0] for k in range(world - 1))
        work = np.array([rows[a:b].sum() for a, b in bands])
        assert work.max() / work.mean() < 1.0 + 0.02 * world          # within a row's worth of the mean
        equal = np.array([rows[1080 * k // world:1080 * (k + 1) // world].sum() for k in range(world)])
        assert equal.max() / equal.mean() > 1.25                      # equal ROWS would leave one GPU with far more work


def test_cloud_weighted_row_cost_balances_cloud_variants():
    """bench.py --shard bands: rows that cross the cloud shell weigh CLOUD_WEIGHT more for the cloud variants, so the
    bands of a cloud frame are cut by estimated work, not by shell hits alone."""
    sys.path.insert(0, ROOT)
    import bench
    from godot_atmosphere_shader_amd import scene as S
    from godot_atmosphere_shader_amd.sharding import balanced_row_bands

    cam = S.Camera.from_pose(192, 108, "P_space")
    plain = bench.cloud_row_cost(np, S, cam, False)
    cloudy = bench.cloud_row_cost(np, S, cam, True)
    assert plain.shape == (108,) and np.all(cloudy >= plain)
    mid = 54
    assert 5 * plain[mid] < cloudy[mid] <= (1 + bench.CLOUD_WEIGHT) * plain[mid]   # the disc rows carry the cloud weight
    far = S.Camera(192, 108, eye=(0.0, 0.0, 900.0), target=(0.0, 0.0, 0.0), far=2000.0)  # small disc: sky rows above and below
    c_far = bench.cloud_row_cost(np, S, far, True)
    assert c_far[0] == pytest.approx(0.02 * 192) and c_far[54] > 20 * c_far[0]
    for cost in (plain, cloudy, c_far):
        bands = balanced_row_bands(cost, 4)
        assert bands[0][0] == 0 and bands[-1][1] == 108
        sums = [cost[a:b].sum() for a, b in bands]
        assert max(sums) / (sum(sums) / 4) < 1.25


# ---- tile strips dealt longest-processing-time-first (round 4: --shard tiles) ----------------------------------------------------------------

def test_lpt_strips_partition_and_balance():
    """lpt_strips: every strip and every tile belongs to exactly one rank, a rank's tile list is sorted heaviest first, the deal is
    deterministic, and on a cost map with one heavy neighbourhood (a cloud bank) the ranks' loads come out within a few per cent of each other
    where contiguous row bands of equal ROWS would be 4x apart."""
    from godot_atmosphere_shader_amd.sharding import STRIP_TILE_ROWS, lpt_strips, row_bands

    rng = np.random.default_rng(5)
    ty, tx = 135, 120   # 1920 x 1080 in 16 x 8 tiles
    cost = rng.integers(200, 400, (ty, tx)).astype(np.uint32)
    cost[40:70, 30:90] += rng.integers(3000, 30000, (30, 60)).astype(np.uint32)   # the heavy part of the picture
    cost[:12] = 0
    for world in (1, 2, 4, 8):
        strips, tiles = lpt_strips(cost, world)
        assert sorted(k for s in strips for k in s) == list(range((ty + STRIP_TILE_ROWS - 1) // STRIP_TILE_ROWS))
        assert sorted(int(t) for ts in tiles for t in ts) == list(range(ty * tx))
        for r in range(world):
            c = cost.reshape(-1)[tiles[r]].astype(np.int64)
            assert np.all(np.diff(c) <= 0)                                             # heaviest first
            rows = sorted({int(t) // tx // STRIP_TILE_ROWS for t in tiles[r]})
            assert rows == strips[r]                                                  # the tiles of exactly its strips
        again = lpt_strips(cost, world)
        assert again[0] == strips and all(np.array_equal(a, b) for a, b in zip(again[1], tiles))
        load = np.array([cost.reshape(-1)[t].sum(dtype=np.int64) for t in tiles], dtype=np.float64)
        if world > 1:
            bands = [cost[y0 // 8:(y1 + 7) // 8].sum(dtype=np.int64) for y0, y1 in row_bands(ty * 8, world)]
            assert load.max() / load.mean() < 1.06 < max(bands) / np.mean(bands)
    assert lpt_strips(np.zeros((5, 3)), 2)[0] == [[0, 2], [1]]   # nothing measured: still a partition, dealt in turn


def _strips_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from godot_atmosphere_shader_amd.sharding import StripGather, lpt_strips

        h, w, tile_h, strip_tile_rows = 27, 11, 4, 2          # 7 tile rows -> 4 strips of 8 pixel rows, the last one 3 rows tall
        tiles_y, tiles_x = (h + tile_h - 1) // tile_h, 3
        cost = (np.arange(tiles_y * tiles_x).reshape(tiles_y, tiles_x) % 5 + 1).astype(np.uint32)
        strips, tiles = lpt_strips(cost, world, strip_tile_rows)
        g = StripGather(h, w, strips, strip_tile_rows * tile_h, torch.device("cpu"), dst=0)
        want = torch.arange(h * w * 4, dtype=torch.float32).reshape(h, w, 4)
        for it in range(2):   # the buffers are re-used frame after frame
            target = g.render_target()
            target.fill_(-1.0)
            for k in strips[rank]:   # what a tile-list draw writes: the pixels of this rank's strips, nothing else
                y0, y1 = k * strip_tile_rows * tile_h, min((k + 1) * strip_tile_rows * tile_h, h)
                target[y0:y1] = want[y0:y1] + float(it)
            res = g.gather()
            if rank == 0:
                assert res.shape == (h, w, 4) and torch.equal(res, want + float(it))
            else:
                assert res is None
        with open(os.path.join(tmpdir, f"ok_strips_{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_strip_gather_assembles_the_frame(tmp_path, world):
    port = _free_port()
    mp.spawn(_strips_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"ok_strips_{r}").exists()


def test_heavy_tiles_rule_on_exact_costs():
    """sharding.heavy_tiles (round 5): a share whose heaviest wavefront outlives 2 x the share's estimated duration gets the tiles that outlive
    0.3 x it split (at most a third); a share bound by throughput gets none."""
    from godot_atmosphere_shader_amd.sharding import heavy_tiles

    uniform = np.full(4000, 1000.0)                       # 4000 equal tiles: duration 4000 * 1000 * 2 / 6144 = 1302 > any tile
    assert heavy_tiles(uniform) == 0
    few = np.sort(np.concatenate([np.full(60, 400000.0), np.full(2000, 500.0)]))[::-1]   # 60 long tiles among cheap ones: duration ~ 8138
    n = heavy_tiles(few)
    assert n == 60
    assert heavy_tiles(few, trigger=100.0) == 0            # a trigger nothing reaches
    assert heavy_tiles(np.zeros(10)) == 0 and heavy_tiles(np.zeros(0)) == 0
    tail = np.sort(np.concatenate([np.full(900, 300000.0), np.full(100, 10.0)]))[::-1]   # nearly everything heavy: capped at a third
    assert heavy_tiles(tail, trigger=0.0) == 1000 // 3
