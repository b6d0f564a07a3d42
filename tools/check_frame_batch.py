"""Renders a frame batch of two identical views through sharding.FrameBatch on one GPU (frames bound as the
context's framebuffer) and compares both with a plain render: usage: python tools/check_frame_batch.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import opengl_raytracer_amd as rt
from opengl_raytracer_amd import sharding
from bench import workload, setup_scene
wl = workload("sword_800_p2")
dev = torch.device("cuda", 0)
g = rt.RayTracer(wl["width"], wl["height"], device=0)
setup_scene(g, wl)
job = sharding.FrameBatch(g, wl["width"], wl["height"], wl["cams"] * 2, dev)
out = job.render(wl["passes"])
torch.cuda.synchronize()
a, b = out[0].cpu().numpy(), out[1].cpu().numpy()
g2 = rt.RayTracer(wl["width"], wl["height"], device=0)
setup_scene(g2, wl); g2.render(wl["passes"]); ref = g2.read_pixels()
print("frames equal:", np.array_equal(a, b), " equal to a plain render:", np.array_equal(a, ref), a.shape, (a[..., :3] != 26).any(-1).mean())
