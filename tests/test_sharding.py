"""The multi-GPU path on CPU: two processes, gloo backend, the oracle standing in for the
GPU renderer (test only).  Checks that interleaved row tiles / view batches rendered by two
ranks and gathered on rank 0 equal the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from common import setup, small_scene
        from opengl_raytracer_amd import scenes, sharding
        from oracle.rt_oracle import Oracle
        sc = small_scene("mixed")
        w, h, passes = 40, 36, 2
        o = Oracle(w, h)
        setup(o, sc, 2)
        if mode == "tiles":
            job = sharding.TileShardedFrame(o, w, h, "cpu", tile_rows=4)
            frame = job.render(passes)
            # a second frame through the same object (buffers are reused)
            frame = job.render(passes)
        else:
            cams = [scenes.look_at_camera((2.0 * np.cos(a), 0.3, 2.0 * np.sin(a)), (0, 0, 0))
                    for a in np.linspace(0, 2 * np.pi, 5, endpoint=False)]
            job = sharding.FrameBatch(o, w, h, cams, "cpu")
            frame = job.render(passes)
        if rank == 0:
            np.save(out_path, frame.numpy())
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tiles", "views"])
def test_two_rank_gloo_equals_single_process(tmp_path, mode):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import setup, small_scene
    from opengl_raytracer_amd import scenes
    from oracle.rt_oracle import Oracle
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    got = np.load(out)
    sc = small_scene("mixed")
    w, h, passes = 40, 36, 2
    o = Oracle(w, h)
    setup(o, sc, 2)
    if mode == "tiles":
        o.render(passes)
        assert got.shape == (h, w, 4)
        assert np.array_equal(got, o.read_pixels())
    else:
        cams = [scenes.look_at_camera((2.0 * np.cos(a), 0.3, 2.0 * np.sin(a)), (0, 0, 0))
                for a in np.linspace(0, 2 * np.pi, 5, endpoint=False)]
        assert got.shape == (5, h, w, 4)
        for k, cam in enumerate(cams):
            o.set_camera(*cam)
            o.render(passes)
            assert np.array_equal(got[k], o.read_pixels()), k


def test_tile_assignment_covers_every_row_once():
    from opengl_raytracer_amd import sharding
    for h, tr, world in [(2160, 15, 8), (1080, 16, 8), (37, 4, 3), (5, 16, 2)]:
        seen = np.zeros(h, dtype=int)
        for r in range(world):
            seen[sharding.tile_rows_of_rank(h, tr, r, world)] += 1
        assert (seen == 1).all()
    assert sharding.choose_tile_rows(2160, 8) == 15
    assert sharding.choose_tile_rows(2160, 1) == 16
    assert sharding.choose_tile_rows(2160, 2) == 15


def _run_bench(*extra):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_stand_in.py"), "--steps", "1", "--warmup", "0",
                          "--backend", "gloo", "--no-cpu-baseline", *extra],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_and_cuts_one_frame_into_tiles():
    """bench.py --gpus 2 without a launcher around it: two ranks, one frame in row tiles by default
    (strong scaling), the same pixels as one rank; --sharding batch renders a frame per rank (weak)."""
    one = _run_bench("--workload", "sword_800_p2")
    two = _run_bench("--workload", "sword_800_p2", "--gpus", "2")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "strong" and "tiles" in two["config"]["sharding"]
    assert two["config"]["frame_crc32"] == one["config"]["frame_crc32"] is not None
    assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"]
    # a line of more than one rank says how a step splits into the slowest rank's share and the gather
    assert two["share_ms_max"] > 0 and two["gather_ms"] >= 0 and "share_ms_max" not in one
    weak = _run_bench("--workload", "sword_800_p2", "--gpus", "2", "--sharding", "batch")
    assert weak["scaling"] == "weak" and weak["config"]["rays_per_step"] == 2 * one["config"]["rays_per_step"]
