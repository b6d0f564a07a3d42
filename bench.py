#!/usr/bin/env python3
"""Benchmark of the three-stage ray tracer on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A step is one whole frame per GPU (or one batch of views): raygen, then P x {intersect, color},
with the scene already resident in HBM, plus -- for N > 1 -- the gather of the RGBA8 frames
to rank 0 over RCCL.  One process per GPU (torch.distributed, backend nccl), scene replicated, no
collective on the data path.  Two ways to shard (--sharding):
  batch  (default) every rank renders one whole frame of the workload per step, N frames per step:
         the work per GPU is fixed ("weak");
  tiles  ONE frame is cut into interleaved row tiles over the ranks: the total work is fixed ("strong").
The 64-view workload deals its fixed set of views over the ranks ("strong").

value = rays actually traced (closest-hit rays entering traversal + shadow rays traced,
summed over passes and ranks) per second of wall time, in Mrays/s.  Rays that missed
everything are dead (the reference skips them in intersect; in color they provably shade
to the background grey, so no shadow ray is traced for them and none is counted).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_TEST = 52.0          # Moller-Trumbore as the reference writes it (trace.glsl:5-20)
PEAK_FP32_TFLOPS = 157.3      # MI355X vector FP32 peak
PEAK_HBM_GBS = 8000.0         # MI355X HBM3E
RAY_PLANE_BYTES = 56.0        # raygen writes 14 float planes per ray
METRIC = "Mrays/s (primary+secondary) at 4K vs triangle count; % FP32 VALU peak"


def asset_mesh(sub, name):
    """Group 0 of an OBJ under assets/, through the host library's loader, as rt_set_mesh records."""
    from opengl_raytracer_amd import host
    return host.Mesh().LoadFromObjFile(os.path.join(ROOT, "assets", sub) + os.sep, name).packed(0)


def asset_texture():
    from opengl_raytracer_amd import host
    return host.load_image(os.path.join(ROOT, "assets", "sword", "sword.png"), invert_y=True)


def workload(name):
    from opengl_raytracer_amd import scenes
    if name == "soup64k_4k_p6":
        return dict(kind="tiles", width=3840, height=2160, passes=6, lights=1,
                    mesh=scenes.triangle_soup(65536, seed=12345), tex=scenes.noise_texture(256, 256),
                    cams=[(np.float32([0, 0, -3]), np.float32([0, 0, 1]), np.float32([0, 1, 0]),
                           np.float32([1, 0, 0]))],
                    desc="synthetic soup, 65536 triangles (seed 12345), 3840x2160, 6 passes (5 bounces), 1 light")
    if name == "soup256k_64views_1080p":
        cams = [scenes.look_at_camera((3.0 * np.cos(a), 0.0, 3.0 * np.sin(a)), (0, 0, 0))
                for a in 2 * np.pi * np.arange(64) / 64]
        return dict(kind="views", width=1920, height=1080, passes=1, lights=1,
                    mesh=scenes.triangle_soup(262144, seed=12346), tex=scenes.noise_texture(256, 256), cams=cams,
                    desc="synthetic soup, 262144 triangles (seed 12346), 64 views of 1920x1080, 1 pass, 1 light")
    if name == "sword_800_p2":
        # the scene, camera and window RunRaytracer sets up (main.cpp:393, 402-414, 607), one bounce
        return dict(kind="tiles", width=800, height=800, passes=2, lights=1, mesh=asset_mesh("sword", "sword_fixed.obj"),
                    tex=asset_texture(), cams=[scenes.default_camera()], data="reference scene (mesh rebuilt from sword_fixed.blend)",
                    desc="sword_fixed.obj (165 triangles) with sword.png, the reference's camera, 800x800, "
                         "2 passes (1 bounce), 1 light")
    if name == "bunny_1080p_p1":
        mesh = asset_mesh("bunny", "bunny.obj")
        return dict(kind="tiles", width=1920, height=1080, passes=1, lights=1, mesh=mesh, tex=scenes.checker_texture(64, 64, 8),
                    cams=[scenes.look_at_camera((0.8, 1.5, -1.9), (0.0, 1.0, -0.3))],
                    data="reference scene (mesh rebuilt from bunny.blend)",
                    desc="bunny.obj (28576 triangles), 1920x1080, 1 pass (primary + shadow rays), 1 light")
    if name == "swordbunny_4k_p4":
        mesh = scenes.concat_meshes(asset_mesh("sword", "sword_fixed.obj"), asset_mesh("bunny", "bunny.obj"))
        return dict(kind="tiles", width=3840, height=2160, passes=4, lights=1, mesh=mesh, tex=asset_texture(),
                    cams=[scenes.default_camera()], data="reference scenes (meshes rebuilt from the .blend sources)",
                    desc="sword_fixed.obj + bunny.obj (165 + 28576 triangles) with sword.png, the reference's camera, "
                         "3840x2160, 4 passes (3 bounces), 1 light")
    raise SystemExit(f"unknown workload {name}")


def setup_scene(obj, wl):
    from opengl_raytracer_amd import scenes
    obj.set_mesh(wl["mesh"])
    obj.set_texture(wl["tex"])
    obj.set_lights(scenes.default_lights(), wl["lights"])
    obj.set_camera(*wl["cams"][0])


def cpu_baseline(wl, seconds_hint=15.0):
    """The oracle (a port of the reference's shaders) on the host cores, on a bounded sample:
    the same scene and camera on a frame scaled down in width and height."""
    from oracle.rt_oracle import Oracle
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    threads = max(1, min(threads, 64))
    # the frame is scaled so that the sample is about 1.7e10 (ray, triangle) tests: 1/20 of the width and
    # height for the 64K-triangle 4K frame, the whole frame for scenes of a few hundred triangles
    tris = max(1, int(wl["mesh"].shape[0] // 3))
    per_frame = float(wl["width"]) * wl["height"] * tris * wl["passes"] * (1 + wl["lights"])
    div = max(1, int(round((per_frame / 1.7e10) ** 0.5)))
    w, h = max(1, wl["width"] // div), max(1, wl["height"] // div)
    o = Oracle(w, h, threads=threads)
    setup_scene(o, wl)
    rays = 0
    t0 = time.perf_counter()
    n_views = 0
    for cam in wl["cams"]:
        o.set_camera(*cam)
        o.raygen()
        for _ in range(wl["passes"]):
            rays += int((o.rays["primitiveID"] != -2).sum())
            o.intersect()
            rays += int(np.isfinite(o.rays["origin"]).all(axis=1).sum()) * wl["lights"]
            o.color()
        n_views += 1
        if time.perf_counter() - t0 > seconds_hint:
            break
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"same scene, {n_views} view(s) at {w}x{h} (1/{div * div} of the rays per view), "
                      f"{wl['passes']} passes, oracle with {threads} OpenMP threads, {dt:.1f} s"}


def measured_traffic(workload_name, kernel_names):
    measured_traffic.valu_issue = None
    """HBM bytes per launch of the named kernels from the committed PMC summary (a separate rocprofv3
    run of this command: counters cannot be read from inside the benchmark), or None."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_per_launch.json")))
        if rec.get("workload") != workload_name:
            return None, None
        measured_traffic.valu_issue = {k: rec["kernels"][k].get("valu_issue_frac") for k in kernel_names}
        return (sum(float(rec["kernels"][k]["hbm_bytes_per_launch"]) for k in kernel_names),
                "profiles/" + rec["source"] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this command)")
    except (OSError, KeyError, ValueError):
        return None, None


def per_pair_roofline(g, job, wl, restore_mode):
    """The brute-force loop proper, measured live: one frame in the per-pair mode (every (ray, triangle)
    pair goes through a per-pair filter, no group rejection, no lists).  The entry is the closest-hit
    kernel of the bounce passes (arbitrary rays, general filter): 52 algorithmic FLOP per pair against
    the FP32 vector peak, durations from hipEvent pairs around each launch."""
    g.set_trace_mode(0)
    try:
        job.render(wl["passes"])
        g.sync()
        g.kernel_log()
        job.render(wl["passes"])
        g.sync()
        log = g.kernel_log()
    finally:
        g.set_trace_mode(restore_mode)
        g.counters()
        g.stage_ms()
    out = []
    for form, label in ((0, "bounce passes, general filter"), (1, "primary pass, shared-origin filter")):
        sel = [r for r in log if int(r["stage"]) == 1 and int(r["form"]) == form and float(r["ms"]) > 0]
        if not sel:
            continue
        ms = sum(float(r["ms"]) for r in sel)
        pairs = sum(float(r["pairs"]) for r in sel)
        tf = pairs * FLOP_PER_TEST / (ms * 1e-3) / 1e12
        out.append({"bound": "valu", "kernel": f"traverse_kernel<K={int(sel[0]['rays_per_lane'])},"
                                                f"{'general' if form == 0 else 'shared-origin'},closest> ({label})",
                    "trace_mode": "filtered", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / PEAK_FP32_TFLOPS, "traffic": None, "launches": len(sel),
                    "avg_launch_ms": ms / len(sel), "pairs_per_launch": pairs / len(sel),
                    "flop_per_pair": FLOP_PER_TEST})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="soup64k_4k_p6")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-per-pair", action="store_true", help="skip the per-pair-mode roofline frame (N = 1)")
    ap.add_argument("--trace-mode", type=int, default=2, help="0 filtered, 1 exact, 2 clustered (default)")
    ap.add_argument("--sharding", choices=("batch", "tiles"), default="batch",
                    help="N > 1: one whole frame per rank per step (default), or one frame in row tiles")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import opengl_raytracer_amd as rt
    from opengl_raytracer_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)

    wl = workload(args.workload)
    g = rt.RayTracer(wl["width"], wl["height"], device=local_rank)
    setup_scene(g, wl)
    g.set_trace_mode(args.trace_mode)
    scaling = "strong"
    if wl["kind"] == "tiles" and (world == 1 or args.sharding == "tiles"):
        job = sharding.TileShardedFrame(g, wl["width"], wl["height"], dev)
        shard_desc = f"image tiles of {job.tile_rows} rows, interleaved over {world} rank(s), gathered to rank 0"
        if world == 1:
            scaling = "weak"  # one rank, one frame: the N = 1 point of the frame-batch series
    elif wl["kind"] == "tiles":
        # one whole frame of the workload per rank per step
        job = sharding.FrameBatch(g, wl["width"], wl["height"], wl["cams"] * world, dev)
        shard_desc = f"frame batch: {world} frames per step, one per rank, gathered to rank 0"
        scaling = "weak"
    else:
        job = sharding.FrameBatch(g, wl["width"], wl["height"], wl["cams"], dev)
        shard_desc = f"frame batch: view k on rank k % {world}, frames gathered to rank 0"

    def sync_all():
        g.sync()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        job.render(wl["passes"])
    sync_all()
    g.counters()
    g.kernel_log()
    g.stage_ms()

    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.render(wl["passes"])
    sync_all()
    dt = time.perf_counter() - t0

    cnt = g.counters()
    log = g.kernel_log()
    stage_ms = g.stage_ms()
    tot = torch.tensor([cnt["closest_rays"], cnt["shadow_rays"], cnt["closest_pairs"], cnt["shadow_pairs"],
                        cnt["closest_candidates"], cnt["shadow_candidates"], cnt["closest_slow_steps"],
                        cnt["shadow_slow_steps"]], dtype=torch.float64, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tot = tot.cpu().numpy()
    dt = float(tmax.item())

    if rank == 0:
        steps = max(args.steps, 1)
        rays = tot[0] + tot[1]
        # dominant kernel on rank 0: group the launch log by kernel variant
        groups = {}
        for r in log:
            key = (int(r["stage"]), int(r["form"]), int(r["rays_per_lane"]), int(r["exact"]))
            if key[1] == 2:
                # the clustered closest-hit traversal covers a pass's pairs with one launch (both sweeps)
                # or with two (band sweep over direction-sorted rays, logged with the pairs; sphere sweep,
                # logged with 0 pairs): one entry, one "launch" per pass, durations added
                key = (key[0], 2, 0, key[3])
            a = groups.setdefault(key, dict(ms=0.0, launches=0, pairs=0.0))
            a["ms"] += float(r["ms"]); a["pairs"] += float(r["pairs"])
            a["launches"] += 1 if (key[1] != 2 or float(r["pairs"]) > 0) else 0
        kernels = []
        for (stage, form, k, exact), a in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
            pairs = a["pairs"]
            if stage == 2:  # shadow traversal: pairs actually executed (early exit), from the counters
                share = a["pairs"] / max(sum(x["pairs"] for kk, x in groups.items() if kk[0] == 2), 1.0)
                pairs = cnt["shadow_pairs"] * share
            name = ("closest_hit_kernel" if stage == 1 else "shade_kernel") + \
                   f"<K={k},{('general', 'shared-origin', 'clustered', 'shared-origin direction cells')[form]}{',exact' if exact else ''}>"
            if form == 2:
                name = "closest_hit_kernel<clustered: band sweep (traverse_bandcell_kernel) + sphere sweep (traverse_cluster_kernel) per pass>"
            a["launches"] = max(a["launches"], 1)
            kernels.append(dict(kernel=name, launches=a["launches"], avg_ms=a["ms"] / a["launches"],
                                total_ms=a["ms"], pairs_per_launch=pairs / a["launches"],
                                tflops=pairs * FLOP_PER_TEST / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0.0))
        dom = kernels[0] if kernels else None
        roofline = None
        if dom:
            names = {"clustered": ["rtk::traverse_bandcell_kernel", "rtk::traverse_cluster_kernel<1, 2>"],
                     "shade_kernel<K=1,shared-origin direction cells>": ["rtk::traverse_grid_kernel<true>"],
                     "closest_hit_kernel<K=1,shared-origin direction cells>": ["rtk::traverse_grid_kernel<false>"]}
            traffic, traffic_src = measured_traffic(
                args.workload if args.trace_mode == 2 and world == 1 else None,
                names["clustered"] if "clustered:" in dom["kernel"] else names.get(dom["kernel"], ["?"]))
            roofline = {"bound": "valu", "kernel": dom["kernel"], "achieved": dom["tflops"],
                        "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_FP32_TFLOPS,
                        "traffic": traffic, "traffic_unit": "bytes of HBM per launch", "traffic_source": traffic_src,
                        # executed VALU wave-instructions / (1024 SIMDs x 0.5 per clock x measured clock), same source
                        "executed_valu_issue_frac": measured_traffic.valu_issue,
                        "avg_launch_ms": dom["avg_ms"], "launches": dom["launches"],
                        "flop_per_pair": FLOP_PER_TEST, "pairs_per_launch": dom["pairs_per_launch"],
                        "note": "FP32 vector ALU bound (no MFMA). achieved = 52 FLOP (the reference's test as "
                                "written) x (ray,triangle) pairs decided per launch / launch duration from "
                                "hipEvent pairs around each launch on the library's stream (a clustered bounce pass: "
                                "from the fork of its two side-by-side kernels to their join). Pairs are mostly "
                                "decided by conservative group tests instead of the 52 FLOPs, so frac exceeds 1; "
                                "the executed VALU issue rate per kernel is in profiles/*rocprof_summary*.txt"}
        out = {
            "metric": METRIC, "value": rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "data": wl.get("data", "synthetic"),
            "config": {"workload": args.workload, "description": wl["desc"], "width": wl["width"],
                       "height": wl["height"], "views": len(getattr(job, "cameras", wl["cams"])), "triangles": int(wl["mesh"].shape[0] // 3),
                       "passes": wl["passes"], "lights": wl["lights"], "sharding": shard_desc,
                       "trace_mode": ("filtered", "exact", "clustered")[args.trace_mode],
                       "rays_per_step": rays / steps,
                       "closest_hit_rays_per_step": tot[0] / steps, "shadow_rays_per_step": tot[1] / steps,
                       # the reference traces a shadow ray per light for every pixel in every pass, dead
                       # pixels included; the value above counts only the rays this build traces
                       "rays_per_step_reference_accounting": tot[0] / steps + float(wl["width"]) * wl["height"] *
                       wl["lights"] * wl["passes"] * len(getattr(job, "cameras", wl["cams"])),
                       "pairs_per_step": (tot[2] + tot[3]) / steps,
                       "filter_pass_rate": (tot[4] + tot[5]) / max(tot[2] + tot[3], 1.0),
                       "slow_steps_per_step": {"closest": tot[6] / steps, "shadow": tot[7] / steps}},
            "pairs_per_s": (tot[2] + tot[3]) / dt,
            "stage_ms_per_step_rank0": {"raygen": stage_ms[0] / steps, "intersect": stage_ms[1] / steps,
                                        "color": stage_ms[2] / steps},
            "roofline": roofline,
            "kernels_rank0": kernels[:6],
        }
        # raygen: 14 planes of one float written per ray, nothing read
        if stage_ms[0] > 0:
            frames = max(1, len(getattr(job, "mine", [0])))  # raygen launches per step on this rank
            gbs = RAY_PLANE_BYTES * g.num_rays * steps * frames / (stage_ms[0] * 1e-3) / 1e9
            rg_traffic, _ = measured_traffic(args.workload if world == 1 else None, ["rtk::raygen_kernel"])
            out["roofline_raygen"] = {"bound": "hbm", "kernel": "raygen_kernel", "achieved": gbs, "peak": PEAK_HBM_GBS,
                                      "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": rg_traffic,
                                      "avg_launch_ms": stage_ms[0] / (steps * frames),
                                      "bytes_per_launch": RAY_PLANE_BYTES * g.num_rays}
        if world == 1 and wl["kind"] == "tiles" and args.trace_mode == 2 and not args.no_per_pair:
            # after the timed region: the same frame in the per-pair mode, for the roofline of the
            # brute-force loop itself (the general filter of the bounce passes does its 20 operations on
            # every pair: frac < 1; the shared-origin filter of the primary pass has a one-operation first level)
            out["roofline_per_pair_mode"] = per_pair_roofline(g, job, wl, args.trace_mode)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    g.close()


if __name__ == "__main__":
    main()
