#!/usr/bin/env python3
"""Benchmark of the three-stage ray tracer on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A step is one whole frame (or one batch of views): raygen, then P x {intersect, color}, with the
scene already resident in HBM, plus -- for N > 1 -- the gather of the RGBA8 pixels to rank 0 over
RCCL.  One process per GPU (torch.distributed, backend nccl), scene replicated, no collective on the
data path.  With --gpus N > 1 and no launcher around it (no WORLD_SIZE in the environment) this
program starts the N ranks itself, as fresh processes, before anything here touches a GPU, and
relays rank 0's line.

Two ways to shard (--sharding):
  tiles  ONE frame is cut into interleaved row tiles over the ranks: the total work is fixed
         ("strong") -- the default for the single-frame workloads;
  batch  every rank renders whole frames: the 64-view workload deals its fixed set of views over the
         ranks ("strong", its default); a single-frame workload renders one frame per rank ("weak").

Frames in flight (--frames-in-flight, default 4, one per hardware queue of the process): every rank
renders with that many contexts, which take the frames (views) of the sequence in turn on streams of
their own, so that the kernel chains of consecutive frames overlap on the GPU; a step is still one
frame (one batch of views), the time is the wall time of the K steps.  After the timed region one
context alone renders a few more frames for the per-kernel figures.

value = rays actually traced (closest-hit rays entering traversal + shadow rays traced, summed over
passes and ranks) per second of wall time, in Mrays/s.  Rays that missed everything are dead (the
reference skips them in intersect; in color they provably shade to the background grey, so no
shadow ray is traced for them and none is counted).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_TEST = 52.0          # Moller-Trumbore as the reference writes it (trace.glsl:5-20)
PEAK_FP32_TFLOPS = 157.3      # MI355X vector FP32 peak: 256 CUs x 64 lanes x 4 FLOP (packed FMA) x 2.4 GHz
PEAK_HBM_GBS = 8000.0         # MI355X HBM3E
NOMINAL_CLOCK_HZ = 2.4e9
# VALU issue rates, measured instruction by instruction (tools/microbench/valu_rates.hip, inline assembly;
# profiles/r03_valu_rates_microbench.txt): packed FP32 (v_pk_fma/mul/add_f32), v_min3/v_max, v_cmp,
# v_cndmask and v_readlane issue one wave64 instruction per SIMD every four clocks -- 1 per clock per CU,
# 256 CUs x 2.4 GHz = 614.4 G/s; plain v_fma/v_mul/v_add_f32 issue at twice that (1.76 per clock per CU
# measured), v_rcp/v_sqrt at half.  So packed and plain FP32 reach the same FLOP rate, and arithmetic that
# may not fuse a multiply into an add (the reference's: fusing changes the bits) has half the FMA peak:
# 78.6 TFLOP/s (60-71 sustained in the microbenchmark, where the clock settles near 2.1 GHz).
# The issue fraction of a kernel is quoted against the 614.4 G/s of the four-clock instructions; a kernel
# made of plain FP32 arithmetic alone could read up to 1.76 on that scale (none here comes near).
PEAK_VALU_GINST = 256 * 4 * 0.25 * NOMINAL_CLOCK_HZ / 1e9
PEAK_UNFUSED_TFLOPS = PEAK_FP32_TFLOPS / 2.0
RAY_PLANE_BYTES = 56.0        # raygen writes 14 float planes per ray
PIXEL_BYTES = 4.0             # color writes one RGBA8 pixel per ray
VISIBLE_GATHER_BYTES = 96.0 + 4.0  # three 32-byte vertices and a texel for a pixel some light reaches
# Contexts that take the frames in turn: one per hardware queue a process has (GPU_MAX_HW_QUEUES, 4 by HIP's
# default); measured 2/3/4/5/6 contexts: 5.63/5.39/5.32/5.56/5.47 ms a frame of soup64k_4k_p6.
FRAMES_IN_FLIGHT = 4
METRIC = "Mrays/s (primary+secondary) at 4K vs triangle count; % FP32 VALU peak"

# rocprof names of the traversal kernels, by (stage, form) of the library's kernel log
KERNEL_NAMES = {
    (1, 3): "rtk::traverse_grid_kernel<false>",
    (2, 3): "rtk::traverse_grid_kernel<true>",
    (1, 2): "rtk::traverse_bandcell_kernel",
    (1, 4): "rtk::traverse_voxel_kernel",
    (1, 5): "rtk::traverse_voxel_long_kernel",
    (2, 6): "rtk::compact_kernel<1>",
    (2, 7): "rtk::color_finish_kernel",
}
EPILOGUE_KERNELS = (KERNEL_NAMES[(2, 6)], KERNEL_NAMES[(2, 7)])
SIDE_BY_SIDE = "bounce pass: rtk::traverse_bandcell_kernel beside rtk::traverse_voxel_kernel + rtk::traverse_voxel_long_kernel"


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


def counters_file(workload_name):
    return os.path.join(ROOT, "profiles", "hbm_traffic_per_launch.json" if workload_name == "soup64k_4k_p6"
                        else f"hbm_traffic_per_launch_{workload_name}.json")


def committed_frame(workload_name):
    """VALU wave-instructions of one whole frame of the workload, from the same committed counter run."""
    try:
        rec = json.load(open(counters_file(workload_name)))
        if rec.get("workload") != workload_name or not rec.get("frame"):
            return None
        return rec["frame"]
    except (OSError, KeyError, ValueError):
        return None


def committed_counters(workload_name):
    """Per-launch hardware counters of the kernels from the committed PMC summary (separate rocprofv3
    runs of this program with --sweeps-in-turn: counters cannot be read from inside the benchmark)."""
    try:
        rec = json.load(open(counters_file(workload_name)))
        if rec.get("workload") != workload_name:
            return {}, None
        return rec["kernels"], "profiles/" + rec["source"]
    except (OSError, KeyError, ValueError):
        return {}, None


def committed_memory_path(workload_name):
    """Per kernel: how busy the address units were, L1 hit rate, L2 round trip (tools/prof_memory_path.sh, committed)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "memory_path_per_launch.json")))
        return rec["kernels"] if rec.get("workload") == workload_name else {}
    except (OSError, KeyError, ValueError):
        return {}


def named_launches(log, in_turn):
    """The kernel log as (rocprof kernel name, record) pairs.  A bounce pass is one record per kernel when
    its kernels run in turn, one record for all three when they run side by side on two streams."""
    out = []
    for r in log:
        stage, form = int(r["stage"]), int(r["form"])
        if form == 2 and not in_turn:
            name = SIDE_BY_SIDE
        elif (stage, form) in KERNEL_NAMES:
            name = KERNEL_NAMES[(stage, form)]
        elif int(r["exact"]):
            name = "rtk::traverse_exact_kernel<%d, %s>" % (int(r["rays_per_lane"]), "false" if stage == 1 else "true")
        else:
            name = "rtk::traverse_kernel<%d, %d, %s>" % (int(r["rays_per_lane"]), form, "false" if stage == 1 else "true")
        out.append((name, r))
    return out


def kernel_rooflines(g, job, wl, workload_name):
    """Two more frames after the timed region, on ONE context with nothing beside it, with the kernels of a
    bounce pass one after the other and an event pair around every traversal launch (each launch then has a
    duration of its own), grouped by kernel.  For each: the executed VALU issue fraction --
    wave-instructions per launch (committed PMC summary) / live duration / issue peak -- which cannot
    exceed 1.  Second result: the shading epilogue of the color stage against the HBM roof."""
    counters, source = committed_counters(workload_name)
    memory_path = committed_memory_path(workload_name)
    g.set_option("sweeps_in_turn", 1)
    g.set_option("kernel_timing", 1)
    try:
        job.render(wl["passes"])
        g.sync()
        g.kernel_log()
        g.counters()
        job.render(wl["passes"])
        g.sync()
        log = g.kernel_log()
        cnt = g.counters()
    finally:
        g.set_option("sweeps_in_turn", 0)
        g.set_option("kernel_timing", 0)
        g.counters()
        g.stage_ms()
    groups = {}
    for name, r in named_launches(log, True):
        a = groups.setdefault(name, dict(ms=0.0, launches=0, pairs=0.0, rays=0.0))
        a["ms"] += float(r["ms"]); a["launches"] += 1; a["pairs"] += float(r["pairs"]); a["rays"] += float(r["rays"])
    # the shading epilogue (raycol_cs.glsl:18-23, 52-127 around the shadow traversal): every pass reads each
    # pixel's ray (56 B) and writes its pixel (4 B); a pixel that a light reaches gathers three vertices and a
    # texel; a ray the stage kills is stored again (56 B).  Two kernels: the pass over all pixels, the shading
    # of the rays it filed.
    color = None
    epi = [groups.pop(k) for k in EPILOGUE_KERNELS if k in groups]
    if len(epi) == 2 and epi[0]["ms"] > 0:
        passes = epi[0]["launches"]
        ms = epi[0]["ms"] + epi[1]["ms"]
        alg = ((RAY_PLANE_BYTES + PIXEL_BYTES) * g.num_rays * passes + VISIBLE_GATHER_BYTES * cnt["lit_pixels"] +
               RAY_PLANE_BYTES * cnt["killed_rays"])
        pmc = [counters.get(k, {}).get("hbm_bytes_per_launch") for k in EPILOGUE_KERNELS]
        color = {"bound": "hbm", "kernel": " + ".join(EPILOGUE_KERNELS), "achieved": alg / (ms * 1e-3) / 1e9,
                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                 "traffic": (pmc[0] + pmc[1]) if all(pmc) else None, "traffic_unit": "bytes of HBM per pass (both kernels)",
                 "bytes_per_pass": alg / passes, "passes": passes, "avg_pass_ms": ms / passes,
                 "avg_launch_ms": {EPILOGUE_KERNELS[0]: epi[0]["ms"] / passes, EPILOGUE_KERNELS[1]: epi[1]["ms"] / passes},
                 "lit_pixels_per_frame": cnt["lit_pixels"], "killed_rays_per_frame": cnt["killed_rays"],
                 "note": "algorithmic bytes: (56 B ray read + 4 B pixel written) x pixels x passes + 100 B of vertex and "
                         "texel gathers per pixel a light reaches + 56 B per ray the stage kills; the shadow traversal "
                         "between the two kernels is VALU work and is not in this figure"}
    out = []
    for name, a in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
        avg_ms = a["ms"] / a["launches"]
        pmc = counters.get(name)
        if pmc and pmc.get("avg_ms") and not 0.8 < avg_ms / float(pmc["avg_ms"]) < 1.25:
            pmc = None  # the counters were taken on a launch of another size (or another build of the kernel)
        entry = {"bound": "valu", "kernel": name, "launches": a["launches"], "avg_launch_ms": avg_ms,
                 "total_ms_per_frame": a["ms"], "rays_per_launch": a["rays"] / a["launches"],
                 "achieved": None, "peak": PEAK_VALU_GINST, "unit": "G VALU wave-instructions/s", "frac": None,
                 "traffic": None, "traffic_unit": "bytes of HBM per launch"}
        if pmc and pmc.get("valu_wave_insts_per_launch"):
            ginst = float(pmc["valu_wave_insts_per_launch"]) / (avg_ms * 1e-3) / 1e9
            if ginst <= PEAK_VALU_GINST:
                entry.update(achieved=ginst, frac=ginst / PEAK_VALU_GINST, traffic=pmc.get("hbm_bytes_per_launch"),
                             counters_source=source)
            else:
                # more instructions per second than the hardware can issue: the committed counters are those of
                # another build of this kernel -- no fraction rather than a wrong one
                entry.update(counters_source=source, counters_stale=True)
        if memory_path.get(name):
            # (from the committed counters of the vector memory path: not measured by this run)
            entry["memory_path"] = memory_path[name]
        out.append(entry)
    return out, color


def exact_mode_roofline(g, job, wl):
    """The reference's loop as written (trace.glsl:135-157): the all-pairs mode -- every ray against every
    triangle, triangles staged in LDS tiles, no acceleration structure -- on every 8th row tile of the frame,
    two passes; its closest-hit launches against the FP32 vector peak.  `achieved` prices every pair at the 52
    FLOP of the shader's test, as a roofline of the brute-force loop does; the kernel executes 26 per pair (the shader's
    first 22 and a four-operation test that proves the shader's own early return) and the whole 52 only on the
    pairs that test leaves open, so `executed` is given beside it."""
    g.set_trace_mode(1)
    g.set_option("kernel_timing", 1)
    try:
        g.set_row_tiles(8, 0, 8)
        g.render(1); g.sync(); g.kernel_log(); g.counters()
        g.render(2); g.sync()
        log = g.kernel_log()
        cnt = g.counters()
    finally:
        g.set_trace_mode(2)
        g.set_option("kernel_timing", 0)
        g.set_row_tiles(job.tile_rows, 0, 1)
        g.counters(); g.stage_ms()
    out = []
    for form, label in ((0, "bounce pass (arbitrary rays)"), (1, "primary pass (rays from the camera)")):
        sel = [r for r in log if int(r["stage"]) == 1 and int(r["form"]) == form and int(r["exact"]) == 1 and float(r["ms"]) > 0]
        if not sel:
            continue
        ms = sum(float(r["ms"]) for r in sel)
        pairs = sum(float(r["pairs"]) for r in sel)
        tf = pairs * FLOP_PER_TEST / (ms * 1e-3) / 1e12
        # executed: 26 FLOP on every pair, 52 more on the pairs left open (all closest-hit launches of the sample)
        open_frac = cnt["closest_candidates"] / max(cnt["closest_pairs"], 1)
        ex = pairs * (26.0 + FLOP_PER_TEST * open_frac) / (ms * 1e-3) / 1e12
        out.append({"bound": "valu", "kernel": "rtk::traverse_exact_kernel<%d, false> (%s)" % (int(sel[0]["rays_per_lane"]), label),
                    "trace_mode": "exact", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s (52 FLOP per pair decided)",
                    "frac": tf / PEAK_FP32_TFLOPS,
                    "executed": {"tflops": ex, "flop_per_pair": 26.0 + FLOP_PER_TEST * open_frac, "pairs_left_open": open_frac,
                                 "unfused_peak": PEAK_UNFUSED_TFLOPS, "frac_of_unfused_peak": ex / PEAK_UNFUSED_TFLOPS},
                    "traffic": None,
                    "launches": len(sel), "avg_launch_ms": ms / len(sel), "pairs_per_launch": pairs / len(sel),
                    "sample": "row tiles of 8 rows, every 8th, of the same frame"})
    return out


def spawn_ranks(args):
    """--gpus N without a launcher: N fresh processes of this program, one per GPU, started before this
    process has touched a GPU; rank 0's output is relayed."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that fails must not leave the others waiting in a collective for ever
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = 0
    while any(p.poll() is None for p in procs):
        bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
        if bad:
            failed = abs(bad[0])
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.05)
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write(b"".join(out).decode())
    sys.stdout.flush()
    return failed or max(abs(c) for c in codes)


def main(stand_in=None):
    """stand_in: a callable (width, height, threads) -> renderer that takes the GPU's place.  Only the tests
    of this program's launcher and sharding pass one (tests/bench_stand_in.py); the benchmark itself has no
    way to render on anything but the HIP library."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="soup64k_4k_p6")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-rooflines", action="store_true",
                    help="skip the frames after the timed region that measure the kernels one by one (N = 1)")
    ap.add_argument("--trace-mode", type=int, default=2, help="0 filtered, 1 exact, 2 clustered (default)")
    ap.add_argument("--sweeps-in-turn", action="store_true",
                    help="the two kernels of a bounce pass one after the other in the timed region too (for profiles)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="contexts per GPU that take the frames (or views) of the sequence in turn, each on a stream "
                         "of its own, so that the kernel chains of consecutive frames overlap (1: one frame at a time; "
                         "0: 4; with more than one, each context runs the two sweeps of a bounce pass in turn on its one "
                         "stream instead of side by side on two -- the frames beside it fill the GPU, and a process has "
                         "four hardware queues)")
    ap.add_argument("--kernel-timing", action="store_true",
                    help="event pairs around every traversal launch of the timed region too (for profiles)")
    ap.add_argument("--sharding", choices=("auto", "batch", "tiles"), default="auto",
                    help="N > 1: one frame in row tiles (default for single-frame workloads), or whole frames per rank")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="context option (a key of opengl_raytracer_amd._native.OPTIONS), e.g. --opt grid_bits=8")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the gather (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r renders on GPU r %% (GPUs present); "
                         "use with --backend gloo (RCCL wants a GPU per rank)")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="one rank: open a process group of one rank all the same and run every collective of the "
                         "multi-rank path (the gather of each frame, barrier, all-reduce) through it")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist
    from opengl_raytracer_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = stand_in is None
    if on_gpu and args.share_gpu:
        local_rank %= max(torch.cuda.device_count(), 1)
    if on_gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        # not HIP's default stream: work on it does not run beside work on other streams
        torch.cuda.set_stream(torch.cuda.Stream(dev))
    # where the pixels are gathered from: device tensors over RCCL; host tensors with any other backend
    shard_dev = dev if (on_gpu and args.backend == "nccl") else torch.device("cpu")
    collectives = world > 1 or args.rehearse_collectives
    if collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        if on_gpu and args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    wl = workload(args.workload)
    free_before = torch.cuda.mem_get_info(dev)[0] if on_gpu else 0
    # frames in flight need framebuffers on the device (pixels that go through the host are copied in turn)
    in_flight = args.frames_in_flight if args.frames_in_flight > 0 else FRAMES_IN_FLIGHT
    if not (on_gpu and shard_dev.type == "cuda"):
        in_flight = 1
    if on_gpu:
        import opengl_raytracer_amd as rt
        opts = {k: int(v) for k, v in (o.split("=", 1) for o in args.opt)}
        gs = [rt.RayTracer(wl["width"], wl["height"], device=local_rank, options=opts) for _ in range(in_flight)]
        for q in gs:
            q.set_trace_mode(args.trace_mode)
            if args.sweeps_in_turn or in_flight > 1:
                q.set_option("sweeps_in_turn", 1)
            if args.kernel_timing:
                q.set_option("kernel_timing", 1)
    else:
        gs = [stand_in(wl["width"], wl["height"], max(1, (os.cpu_count() or 2) // max(world, 1)))]
    t_setup = time.perf_counter()
    for q in gs:
        setup_scene(q, wl)
    if on_gpu:
        for q in gs:
            q.sync()
    setup_scene_ms = (time.perf_counter() - t_setup) * 1e3
    g = gs[0]
    mode = args.sharding
    if mode == "auto":
        mode = "tiles" if wl["kind"] == "tiles" else "batch"
    scaling = "strong"

    def make_job(renderers):
        if wl["kind"] == "tiles" and (world == 1 or mode == "tiles"):
            return sharding.TileShardedFrame(renderers, wl["width"], wl["height"], shard_dev, bind_device_framebuffer=on_gpu,
                                             gather_always=collectives)
        # whole frames: one frame of a single-frame workload per rank per step, or the workload's views dealt out
        cams = wl["cams"] * world if wl["kind"] == "tiles" else wl["cams"]
        return sharding.FrameBatch(renderers, wl["width"], wl["height"], cams, shard_dev, bind_device_framebuffer=on_gpu,
                                   gather_always=collectives)

    job = make_job(gs)
    if isinstance(job, sharding.TileShardedFrame):
        shard_desc = f"image tiles of {job.tile_rows} rows, interleaved over {world} rank(s), gathered to rank 0"
    elif wl["kind"] == "tiles":
        shard_desc = f"frame batch: {world} frames per step, one per rank, gathered to rank 0"
        scaling = "weak"
    else:
        shard_desc = f"frame batch: view k on rank k % {world}, frames gathered to rank 0"
    if in_flight > 1:
        shard_desc += (f"; {in_flight} frames in flight per GPU (contexts that take the frames in turn, each on its own stream)"
                       if isinstance(job, sharding.TileShardedFrame) else
                       f"; {in_flight} views in flight per GPU (contexts that take a rank's views in turn, each on its own stream)")

    def sync_all():
        if on_gpu:
            for q in gs:
                q.sync()
            torch.cuda.synchronize(dev)
        if collectives:
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize(dev)

    # (every context in flight renders at least one untimed frame)
    first_step_ms = 0.0
    t_first = time.perf_counter()
    for k in range(max(args.warmup, in_flight if isinstance(job, sharding.TileShardedFrame) else 0)):
        job.render(wl["passes"])
        if k == 0:
            sync_all()
            first_step_ms = (time.perf_counter() - t_first) * 1e3
    sync_all()
    device_bytes = free_before - torch.cuda.mem_get_info(dev)[0] if on_gpu else 0
    if on_gpu:
        for q in gs:
            q.counters()
            q.kernel_log()
            q.stage_ms()

    frame = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame = job.render(wl["passes"])
    sync_all()
    dt = time.perf_counter() - t0
    # checksum of the gathered pixels of the last step (the same for every N and either renderer)
    frame_crc = None
    if rank == 0 and frame is not None and not (world > 1 and scaling == "weak"):
        import zlib
        frame_crc = "%08x" % zlib.crc32(frame.contiguous().cpu().numpy().tobytes())

    if on_gpu:
        cnt, log, stage_ms = {}, [], [0.0, 0.0, 0.0]
        for q in gs:
            for k, v in q.counters().items():
                cnt[k] = cnt.get(k, 0) + v
            log.extend(q.kernel_log())
            stage_ms = [a + b for a, b in zip(stage_ms, q.stage_ms())]
    else:
        # the stand-in renderer counts nothing: rays = pixels rendered
        n_px = float(getattr(g, "num_rays", wl["width"] * wl["height"])) * len(getattr(job, "mine", [0])) * args.steps
        cnt = dict(closest_rays=n_px, shadow_rays=0, closest_pairs=0, shadow_pairs=0, closest_candidates=0,
                   shadow_candidates=0, closest_slow_steps=0, shadow_slow_steps=0)
        log, stage_ms = [], [0.0, 0.0, 0.0]
    # The same steps with the lights moving as the reference moves them every frame (main.cpp:449:
    # SetLightPosition(lerp(0, 2 pi / 10, sin(time)))), at 60 frames a second: before a step the context
    # whose turn it is gets that frame's lights, so its lists of the light's direction cells are built anew.
    dt_moving = None
    if on_gpu:
        import math
        from opengl_raytracer_amd import scenes
        rigs = [scenes.default_lights(angle_offset=(2 * math.pi / 10) * math.sin((k + 1) / 60.0))
                for k in range(args.steps + in_flight)]

        def moving_step(k):
            targets = [job.rs[job.turn]] if isinstance(job, sharding.TileShardedFrame) else job.rs
            for q in targets:
                q.set_lights(rigs[k], wl["lights"])
            job.render(wl["passes"])

        for k in range(in_flight):      # (the first rebuild of a context's lists sizes them, untimed)
            moving_step(k)
        sync_all()
        t1 = time.perf_counter()
        for k in range(args.steps):
            moving_step(in_flight + k)
        sync_all()
        dt_moving = time.perf_counter() - t1
        for q in gs:                    # back to the lights of the timed region, for what follows
            q.set_lights(scenes.default_lights(), wl["lights"])
        for _ in range(in_flight):
            job.render(wl["passes"])
        sync_all()
        for q in gs:
            q.counters(); q.kernel_log(); q.stage_ms()
        if collectives:
            t_dm = torch.tensor([dt_moving], dtype=torch.float64, device=shard_dev)
            dist.all_reduce(t_dm, op=dist.ReduceOp.MAX)
            dt_moving = float(t_dm.item())

    # with ranks to gather from: a few more steps, one at a time, that split a step into the rank's own
    # rendering and the collective (the slowest rank's share is the ceiling of the split, the rest its cost)
    share_ms, gather_ms = 0.0, 0.0
    if collectives:
        marks = []
        for _ in range(3):
            sync_all()
            m = sharding.StepMarks(shard_dev if on_gpu else "cpu")
            job.render(wl["passes"], marks=m)
            marks.append(m)
        sync_all()
        share_ms = float(np.mean([m.ms()[0] for m in marks]))
        gather_ms = float(np.mean([m.ms()[1] for m in marks]))
        if on_gpu:
            for q in gs:
                q.counters(); q.kernel_log(); q.stage_ms()
    tot = torch.tensor([cnt["closest_rays"], cnt["shadow_rays"], cnt["closest_pairs"], cnt["shadow_pairs"],
                        cnt["closest_candidates"], cnt["shadow_candidates"], cnt["closest_slow_steps"],
                        cnt["shadow_slow_steps"]], dtype=torch.float64, device=shard_dev)
    tmax = torch.tensor([dt, share_ms, gather_ms], dtype=torch.float64, device=shard_dev)
    if collectives:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tot = tot.cpu().numpy()
    dt, share_ms_max, gather_ms_max = (float(v) for v in tmax.cpu().numpy())

    if rank == 0:
        steps = max(args.steps, 1)
        rays = tot[0] + tot[1]
        # traversal launches of the timed region on rank 0, by kernel (a side-by-side bounce pass is one
        # record for its kernels); untimed -- an event pair around every launch costs the stream idle
        # time -- unless --kernel-timing asks for it (the kernels are timed one by one after the region)
        groups = {}
        for name, r in named_launches(log, args.sweeps_in_turn or in_flight > 1):
            a = groups.setdefault(name, dict(ms=0.0, launches=0, pairs=0.0))
            a["ms"] += float(r["ms"]); a["pairs"] += float(r["pairs"]); a["launches"] += 1
        kernels = [dict(kernel=k, launches=a["launches"], pairs_per_launch=a["pairs"] / a["launches"],
                        **(dict(avg_ms=a["ms"] / a["launches"], total_ms=a["ms"]) if args.kernel_timing else {}))
                   for k, a in sorted(groups.items(), key=lambda kv: (-kv[1]["ms"], -kv[1]["launches"]))]
        views = len(getattr(job, "cameras", wl["cams"]))
        out = {
            "metric": METRIC, "value": rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "data": wl.get("data", "synthetic"),
            "config": {"workload": args.workload, "description": wl["desc"], "width": wl["width"],
                       "height": wl["height"], "views": views, "triangles": int(wl["mesh"].shape[0] // 3),
                       "passes": wl["passes"], "lights": wl["lights"], "sharding": shard_desc,
                       "trace_mode": ("filtered", "exact", "clustered")[args.trace_mode], "options": args.opt,
                       "frames_in_flight": in_flight, "rehearse_collectives": bool(args.rehearse_collectives),
                       "rays_per_step": rays / steps,
                       "closest_hit_rays_per_step": tot[0] / steps, "shadow_rays_per_step": tot[1] / steps,
                       # the reference traces a shadow ray per light for every pixel in every pass, dead
                       # pixels included; the value above counts only the rays this build traces
                       "rays_per_step_reference_accounting": tot[0] / steps + float(wl["width"]) * wl["height"] *
                       wl["lights"] * wl["passes"] * views,
                       "pairs_decided_per_step": (tot[2] + tot[3]) / steps,
                       "frame_crc32": frame_crc},
            # what the brute-force loop would have to sustain to decide the same pairs in the same time
            "equivalent_bruteforce_tflops": (tot[2] + tot[3]) * FLOP_PER_TEST / dt / 1e12,
            # (summed over the contexts in flight, whose stages overlap: more than the step's wall time then)
            "stage_ms_per_step_rank0": {"raygen": stage_ms[0] / steps, "intersect": stage_ms[1] / steps,
                                        "color": stage_ms[2] / steps},
            "roofline": None,
            "kernels_rank0": kernels[:6],
        }
        if dt_moving is not None:
            out["ms_per_step_moving_lights"] = dt_moving / steps * 1e3
        if on_gpu:
            # one-off costs, outside every timed region: the scene set-up calls of all contexts in flight (rt_set_mesh
            # builds the filter records, the cells in space and the camera-independent tables), the first step (the
            # lists of the camera's and the light's direction cells, and the band lists on the first bounce pass),
            # and the device memory the rank holds after warm-up
            out["setup_ms"] = {"scene_calls": setup_scene_ms, "first_step": first_step_ms}
            out["device_bytes"] = int(device_bytes)
        if collectives:
            # one step at a time, after the timed region: the slowest rank's rendering of its share (before the
            # gather) and the longest gather (on rank 0 it includes waiting for the slowest rank)
            out["share_ms_max"] = share_ms_max
            out["gather_ms"] = gather_ms_max
        solo = None
        if on_gpu and world == 1 and in_flight > 1:
            # the kernels are timed on one context with nothing beside it (its sweeps side by side again)
            g.set_option("sweeps_in_turn", 1 if args.sweeps_in_turn else 0)
            solo = make_job(g)
            for _ in range(2):
                solo.render(wl["passes"])
            sync_all()
            g.counters(); g.kernel_log(); g.stage_ms()
            solo_steps = 3
            for _ in range(solo_steps):
                solo.render(wl["passes"])
            sync_all()
            stage_ms = g.stage_ms()
            g.counters(); g.kernel_log()
            out["stage_ms_per_step_one_frame_at_a_time"] = {"raygen": stage_ms[0] / solo_steps, "intersect": stage_ms[1] / solo_steps,
                                                            "color": stage_ms[2] / solo_steps}
        if on_gpu and world == 1:
            # raygen: 14 planes of one float written per ray, nothing read.  Its duration inside a frame (the stage's
            # event pair, one context alone; the kernel trace under profiles/ has the same launches), and beside it
            # ten launches back to back, which the 256 MB of Infinity Cache cannot absorb: the sustained write rate
            one = out.get("stage_ms_per_step_one_frame_at_a_time") or out["stage_ms_per_step_rank0"]
            frames = max(1, len(getattr(job, "mine", [0])))  # raygen launches per step
            ms = one["raygen"] / frames
            sync_all()
            g.stage_ms()
            launches = 10
            for _ in range(launches):
                g.raygen()
            g.sync()
            ms_b2b = g.stage_ms()[0] / launches
            gbs = RAY_PLANE_BYTES * g.num_rays / (ms * 1e-3) / 1e9
            gbs_b2b = RAY_PLANE_BYTES * g.num_rays / (ms_b2b * 1e-3) / 1e9
            pmc = committed_counters(args.workload)[0].get("rtk::raygen_kernel", {})
            # the fraction is the sustained one; inside a frame the launch is shorter because part of its
            # write-back is still in the Infinity Cache when the next kernel starts
            out["roofline_raygen"] = {"bound": "hbm", "kernel": "rtk::raygen_kernel", "achieved": gbs_b2b, "peak": PEAK_HBM_GBS,
                                      "unit": "GB/s", "frac": gbs_b2b / PEAK_HBM_GBS, "traffic": pmc.get("hbm_bytes_per_launch"),
                                      "avg_launch_ms": ms_b2b, "launches": launches, "bytes_per_launch": RAY_PLANE_BYTES * g.num_rays}
            if ms > 0.25 * ms_b2b:
                out["roofline_raygen"]["in_a_frame"] = {"avg_launch_ms": ms, "achieved": gbs, "frac": gbs / PEAK_HBM_GBS}
            else:
                # rt_render leaves raygen to the primary traversal, which forms each fresh ray from its pixel in
                # registers: a frame has no raygen launch, and the kernel measured here is the one rt_raygen launches
                out["roofline_raygen"]["in_a_frame"] = None
                out["roofline_raygen"]["note"] = ("rt_raygen's kernel, ten launches back to back; inside rt_render the primary "
                                                  "traversal forms the rays itself and no raygen kernel is launched")
        if on_gpu and world == 1 and args.trace_mode == 2 and not args.no_kernel_rooflines:
            job = solo or job
            per_kernel, color = kernel_rooflines(g, job, wl, args.workload)
            if color:
                out["roofline_color"] = color
            if per_kernel:
                dom = dict(per_kernel[0])
                dom["note"] = ("the kernel with the most time per frame; FP32 vector ALU work, no MFMA; achieved = executed VALU "
                               "wave-instructions per launch (committed PMC summary) / the launch's duration from the hipEvent "
                               "pair around it on the library's stream, kernels of a bounce pass one after the other; peak = "
                               "256 CUs x 1 wave-instruction per clock x 2.4 GHz, the rate of packed, compare, select and "
                               "cross-lane instructions (plain v_fma/v_mul/v_add_f32 issue at twice that: "
                               "profiles/r03_valu_rates_microbench.txt)")
                out["roofline"] = dom
                out["roofline_kernels"] = per_kernel
            fr = committed_frame(args.workload)
            if fr and fr.get("valu_wave_insts"):
                # the whole frame against the same roof: every kernel's instructions (one frame of the committed
                # counter run) over the frame time of the timed region
                ginst = float(fr["valu_wave_insts"]) / (dt / steps) / 1e9
                if ginst <= PEAK_VALU_GINST:
                    out["roofline_frame"] = {"bound": "valu", "achieved": ginst, "peak": PEAK_VALU_GINST,
                                             "unit": "G VALU wave-instructions/s", "frac": ginst / PEAK_VALU_GINST,
                                             "valu_wave_insts_per_frame": fr["valu_wave_insts"],
                                             "kernels_per_frame": fr.get("dispatches"), "ms_per_frame": dt / steps * 1e3,
                                             "frames_in_flight": in_flight,
                                             "note": "all " + str(fr.get("dispatches")) + " kernels of a frame: executed VALU "
                                                     "wave-instructions per frame (committed PMC summary) / the timed "
                                                     "region's time per frame / the issue peak"}
            if wl["kind"] == "tiles":
                out["roofline_exact_mode"] = exact_mode_roofline(g, job, wl)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out))
    if collectives:
        dist.barrier()
        dist.destroy_process_group()
    if on_gpu:
        for q in gs:
            q.close()


if __name__ == "__main__":
    main()
