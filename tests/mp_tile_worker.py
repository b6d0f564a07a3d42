"""One rank of test_two_ranks_share_one_gpu_tiles: a process of its own (RANK / WORLD_SIZE /
MASTER_* in the environment, gloo backend) with its own context on cuda:0.  Renders this rank's
interleaved row tiles through sharding.TileShardedFrame, pixels copied through the host (gloo
gathers CPU tensors); rank 0 saves the gathered frame to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, PASSES = 200, 120, 3


def scene_setup(g):
    from opengl_raytracer_amd import scenes
    mesh = scenes.concat_meshes(scenes.triangle_soup(3000, seed=17, extent=1.0, size=0.1),
                                scenes.quad_grid(6, 6, origin=(-1.4, -1.2, 1.2), du=(2.8, 0, 0), dv=(0, 2.4, 0)))
    g.set_mesh(mesh)
    g.set_texture(scenes.noise_texture(64, 64))
    g.set_lights(scenes.default_lights(), 2)
    g.set_camera(*scenes.look_at_camera((0.2, 0.1, -2.5), (0, 0, 0)))


def main():
    import numpy as np
    import torch.distributed as dist
    import opengl_raytracer_amd as rt
    from opengl_raytracer_amd import raytracer, sharding
    raytracer.DEFAULT_OPTIONS["cluster_min_tris"] = 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with rt.RayTracer(W, H, device=0) as g:
            scene_setup(g)
            job = sharding.TileShardedFrame(g, W, H, "cpu", tile_rows=8, bind_device_framebuffer=False)
            frame = job.render(PASSES)
            frame = job.render(PASSES)      # the buffers are reused
            if rank == 0:
                np.save(sys.argv[1], frame.numpy())
            else:
                assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
