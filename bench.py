#!/usr/bin/env python3
"""bench.py — rays/s of the trace path on N MI355X GPUs of one node.

A "step" is one pass of the hot path over one batch: rt_photon_generate + rt_render of one frame of
the workload on every rank, the frame quantised on the GPU (Color::scaleOut) and, for N > 1, gathered
to rank 0 over RCCL.  The scene is uploaded (resident in HBM) before the timed region.

Workload (config.workload): BASELINE.json configs[1], examples/example9.xml at 3840x2160 — the
largest single-GPU configuration.  `--workload` selects another config for developer runs.

N > 1: frames are independent units: every rank renders its own full frame per step and
rank 0 collects the RGBA8 frames ("scaling": "weak").  value = rays of all ranks / max-over-ranks time.

The north star's two splits are measured at every N through the product driver (raytracer_amd/driver.py)
and reported under "extra" (fixed total work, i.e. strong scaling):
  c4_tiles   examples3/101_julia_shiny.xml, 8-row bands interleaved over the ranks (driver.render_still),
             float bands gathered to rank 0 and quantised there
  c5_frames  examples3/103_caustic_animation.xml, 20 frames round-robin over the ranks (driver.render_units)
  c5mb8      examples3/103_caustic_animation_mb8.xml, 160 (frame, sub-frame) units round-robin, motion blur
             accumulated on rank 0 in sub-frame order
each with sec/frame, rays/s and every rank's GPU-busy fraction (GPU time of its render calls / wall time).
Those timings cover parse + upload + photon pass + trace + gather + quantise + download of the RGBA8 frames
to the host; the PNG/APNG encode is not in them.  `--workload c5mb` makes the 160-unit file the main line.

Ray counts come from a counters-enabled launch of the same kernels outside the timed region (the
counters are pinned to the oracle's by tests/test_gpu_parity.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "c1": ("examples/example1.xml", 0.0, "C1 examples/example1.xml 512x512"),
    "c2": ("examples/example9_4k.xml", 0.0, "C2 examples/example9.xml @3840x2160, 69 objects, 2 lights, 2 textures"),
    "c3": ("examples2/8_fresnel_ss4.xml", 0.0, "C3 examples2/8_fresnel.xml 512x512, 4x4 supersampling x 8 hues"),
    "c4": ("examples3/101_julia_shiny.xml", 0.0, "C4 examples3/101_julia_shiny.xml 960x600, 2x2 supersampling"),
    "c5": ("examples3/103_caustic_animation.xml", 0.5, "C5 examples3/103_caustic_animation.xml frame t=0.5 (photon pass + render)"),
}

# Algorithmic FP32 operations per unit of work (SURVEY §8d hand counts of the reference source as
# written; lower bounds where the survey gives a range — e.g. every triangle test is priced at its
# cheapest exit).  Transcendentals and the FP64 sub-expressions are not counted.
FLOPS = {"triangle_tests": 29, "sphere_tests": 59, "shaded_hits": 64, "phong_lights": 75, "fresnel_evals": 110,
         "texture_lookups": 40, "primary_rays": 60, "secondary_rays": 37, "julia_de_iterations": 41,
         "julia_steps": 11, "julia_normals": 36, "julia_queries": 30, "photon_segments": 37,
         "bvh_node_visits": 21}   # ours, not the reference's: slab test of one node (6 sub, 6 mul, 6 min/max, 3 compares)
PEAK_FP32_TFLOPS = 157.3   # MI355X FP32 vector peak (MI355X_MICROARCH.md); assumes packed FMA
# Lane-operations per second when every SIMD issues one plain (non-packed, non-fused) FP32 instruction as fast as it
# can: 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz (a wave64 instruction takes 2 clocks; tools/valu_rate.hip
# measures 2.4-2.9 for v_mul/v_add/v_fma, 4.3-5 for packed, v_cmp, v_max/v_min and f64, 8 for rcp/sqrt/exp).
ISSUE_CEILING_LANE_OPS = 256 * 4 * 32 * 2.4e9


def pmc_record(workload):
    """The committed PMC passes of this workload (profiles/traffic.json, made by tools/traffic_json.py from separate
    rocprofv3 --pmc runs of this bench; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(workload, {})
    except (OSError, ValueError):
        return {}


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the PMC passes; None when no pass exists."""
    return pmc_record(workload).get("hbm_bytes_per_launch")


def executed_valu(workload, kernel_ms):
    """What the kernel EXECUTED (not the algorithmic count): SQ_INSTS_VALU x mean active lanes per instruction from
    the committed PMC pass, over the kernel time measured live, against the plain-FP32 issue ceiling."""
    rec = pmc_record(workload)
    insts, lanes = rec.get("valu_wave_instructions"), rec.get("active_lanes_per_valu_instruction")
    if not insts or not lanes:
        return None
    rate = insts * lanes / (kernel_ms * 1e-3)
    return {"lane_ops_per_s": rate, "frac_of_issue_ceiling": rate / ISSUE_CEILING_LANE_OPS,
            "issue_ceiling_lane_ops_per_s": ISSUE_CEILING_LANE_OPS, "valu_wave_instructions": insts, "active_lanes": lanes,
            "note": "SQ_INSTS_VALU x active lanes (profiles/traffic.json) / live kernel time; ceiling = 256 CUs x 4 SIMDs x "
                    "32 lanes/clk x 2.4 GHz, which only plain f32 mul/add/fma/mov reach (tools/valu_rate.hip)"}


def algorithmic_flops(stats_dict):
    return float(sum(stats_dict.get(k, 0) * w for k, w in FLOPS.items()))


def host_cores():
    """Threads for the CPU legs: the GPU box gives one GPU's job 16 of the machine's cores (OMP_NUM_THREADS is set to
    that share there); os.cpu_count() shows the whole machine and is reported beside it."""
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or 16
    return max(1, min(share, os.cpu_count() or 1))


def cpu_baseline(scene, n_rays_frame):
    """The oracle (CPU restatement, "port") timed on this box's host cores on the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    cores = host_cores()
    cam = scene.desc.camera
    # bounded sample: every k-th row of the frame, repeated until about 10 s of CPU time (cores x wall)
    step = 1
    est_full = n_rays_frame / 4e7 / max(1, cores) * 16     # ~4e7 rays/s on 16 cores (C2, measured on the GPU box)
    while est_full / step > 4.0:
        step *= 2
    t0 = time.time()
    photons = oracle_lib.Photons(scene) if scene.desc.photon_steps > 0 else None
    t_photon = time.time() - t0
    rays = 0
    reps = 0
    t0 = time.time()
    while reps == 0 or (time.time() - t0 < 10.0 / cores * 4 and reps < 50):
        _, st = oracle_lib.render(scene, photons, threads=cores, rows=(0, cam.height, step))
        rays += st.total_rays()
        reps += 1
    dt = time.time() - t0 + t_photon * reps
    rays += (photons.stats.photon_segments if photons else 0) * reps
    return {"value": rays / dt, "unit": "rays/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
            "sample": "oracle/liboracle.so (CPU restatement, pinned bit-exact to the reference's O-strict build), every %d-th row "
                      "of the frame (%d rows) x %d repetitions%s, %.2f s wall on %d threads" % (
                step, (cam.height + step - 1) // step, reps, " + serial photon pass" if photons else "", dt, cores)}


def ofast_ratio(key):
    """SURVEY 8d's calibration ratio: how much faster the reference built with its author's flags (-Ofast -flto -march=native)
    is than the O-strict build timed here, per config (profiles/cpu_reference_ofast_ratio.json, tools/ofast_ratio.py)."""
    try:
        with open(os.path.join(ROOT, "profiles", "cpu_reference_ofast_ratio.json")) as f:
            rec = json.load(f)["configs"].get(key)
        return {"ratio": rec["ratio"], "strict_s": rec["strict_s"], "ofast_s": rec["ofast_s"]} if rec else None
    except (OSError, ValueError, KeyError):
        return None


def cpu_reference(scene_path, n_rays_frame, still_at=None, ratio_key=None):
    """The reference's own program (oracle/_ref/raytracer_ostrict: its unmodified sources compiled by
    oracle/Makefile, g++ -O2) on the same scene file with threads = host cores; time = its own
    'Finished in' line (trace + PNG encode, src/main.cpp:23-35).  `still_at` renders one frame of an animated
    file (a <still time=…/> tag, src/sceneparser.cpp).  None where the binary is absent."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "raytracer_ostrict")
    if not os.path.exists(exe):
        return None
    cores = host_cores()
    try:
        with tempfile.TemporaryDirectory() as tmp:
            src_dir = os.path.dirname(scene_path)
            for name in os.listdir(src_dir):
                if name != os.path.basename(scene_path):
                    os.symlink(os.path.join(src_dir, name), os.path.join(tmp, name))
            xml = open(scene_path).read()
            xml = re.sub(r"<scene\b", '<scene threads="%d"' % cores, re.sub(r'\sthreads="[^"]*"', "", xml), count=1)
            if still_at is not None:
                xml = re.sub(r"(<animation\b[^>]*/>)", r'\1<still time="%r"/>' % still_at, xml, count=1)
            path = os.path.join(tmp, "bench_scene.xml")
            with open(path, "w") as f:
                f.write(xml)
            t0 = time.perf_counter()
            out = subprocess.run([exe, path, os.path.join(tmp, "out.png")], capture_output=True, text=True, timeout=240).stdout
            wall = time.perf_counter() - t0
        m = re.search(r"Finished in ([0-9.eE+-]+)\s*s", out)
        if not m:
            return None
        sec = float(m.group(1))
        timer = "its own 'Finished in' timer = %.2f s (trace + PNG encode)" % sec
        if still_at is not None:
            # a still's timer starts AFTER the photon pass (src/main.cpp:18-23): the process's wall clock covers it
            # (+ scene load, milliseconds), as the timers of the reference's animation loops do (src/main.cpp:82-111)
            timer = "wall clock of the process = %.2f s (scene load + serial photon pass + trace + PNG encode; its 'Finished in' timer, which excludes the photon pass, says %.2f s)" % (wall, sec)
            sec = wall
        ratio = ofast_ratio(ratio_key)
        return {"value": n_rays_frame / sec, "unit": "rays/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "reference",
                "sec_per_frame": sec, "ofast_ratio": ratio,
                "sample": "oracle/_ref/raytracer_ostrict (reference sources, g++ -O2 -pthread), whole frame%s, threads=%d, %s; rays = the "
                          "nearest-hit formulation's count.  The author's shipping flags (-Ofast -flto -march=native, host-specific, so that "
                          "build stays off this box) make the reference %s faster on this config (profiles/cpu_reference_ofast_ratio.json: both "
                          "builds timed on the build container's 8 vCPUs, tools/ofast_ratio.py): divide the GPU/CPU ratio by that" % (
                              "" if still_at is None else " at <still time=%r>" % still_at, cores, timer,
                              "%.2fx" % ratio["ratio"] if ratio else "1.0-1.2x (no calibration file)")}
    except Exception:   # noqa: BLE001 - a missing or failing reference binary must not break the bench line
        return None


# ------------------------------------------------------------------------- the north star's splits, through the driver

def count_rays(ctx, capi, path, times):
    """(rays, algorithmic flops, rays not traced) of the units at `times` from counters-enabled launches; rays = primary +
    secondary + shadow + photon segments of the reference's formulation; not traced = the photon roots dropped before the
    pass and the shadow rays that can only add +-0, which a plain launch skips."""
    import torch
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    total, flops, skipped = 0, 0.0, {"photon_roots_culled": 0, "shadow_rays_elided": 0}
    for t in times:
        scene = capi.Scene(path, float(t))
        ctx.upload(scene)
        d = ctx.photon_generate().as_dict() if scene.desc.photon_steps > 0 else {}
        out = torch.empty((scene.width * scene.height, 4), dtype=torch.float32, device="cuda")
        ctx.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        r = ctx.fetch_stats().as_dict()
        total += r["primary_rays"] + r["secondary_rays"] + r["shadow_rays"] + d.get("photon_segments", 0)
        flops += algorithmic_flops({k: v + d.get(k, 0) for k, v in r.items() if k != "kernel_ms"})
        # rays of the reference's formulation that the plain launches do not trace
        skipped["photon_roots_culled"] += d.get("photon_roots_culled", 0)
        skipped["shadow_rays_elided"] += r.get("shadow_rays_elided", 0)
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
    return total, flops, skipped


def timed_job(job, renderer, reps, warmup, world, dev):
    """Runs job() `warmup` + `reps` times; returns (max-over-ranks wall seconds of the `reps`, [busy fraction per rank])."""
    import torch
    import torch.distributed as dist

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        job()
    fence()
    renderer.busy_events = []
    t0 = time.perf_counter()
    for _ in range(reps):
        job()
    fence()
    elapsed = time.perf_counter() - t0
    busy = renderer.busy_seconds()
    renderer.busy_events = None
    both = torch.tensor([elapsed, busy], dtype=torch.float64, device=dev)
    if world > 1:
        rows = [torch.empty_like(both) for _ in range(world)]
        dist.all_gather(rows, both)
    else:
        rows = [both]
    rows = [r.tolist() for r in rows]
    wall = max(r[0] for r in rows)
    return wall, [round(r[1] / wall, 4) for r in rows]


def split_measurements(which, renderer, ctx, capi, driver, world, rank, dev, reps, warmup):
    """The north star's splits (module docstring) at this world size; dict of results on every rank."""
    import torch
    import torch.distributed as dist
    f32 = np.float32
    out = {}

    def shared(x):      # rank 0's count, known to all
        if world == 1:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        dist.broadcast(t, 0)
        return float(t.item())

    if "c4_tiles" in which:
        progress("extra: c4_tiles")
        path = os.path.join(ROOT, "scenes", "examples3", "101_julia_shiny.xml")
        scene = capi.Scene(path, 0.0)
        rays, _flops, skipped = count_rays(ctx, capi, path, [0.0]) if rank == 0 else (0, 0.0, {})
        rays, traced = shared(rays), shared(rays - sum(skipped.values()))
        elided, culled = shared(skipped.get("shadow_rays_elided", 0)), shared(skipped.get("photon_roots_culled", 0))
        host = driver.HostFrames(dev)

        def job():
            full = driver.render_still(renderer, scene)
            if rank == 0:
                host.items.clear()
                host.add(renderer.quantize(full))
                host.frames()
        wall, busy = timed_job(job, renderer, reps, warmup, world, dev)
        out["c4_tiles"] = {"scene": "examples3/101_julia_shiny.xml", "split": "8-row bands interleaved over %d rank(s), float bands gathered to rank 0" % world,
                           "scaling": "strong", "frames": reps, "sec_per_frame": wall / reps, "rays_per_frame": rays,
                           "rays_per_s": rays * reps / wall, "rays_traced_per_frame": traced, "rays_traced_per_s": traced * reps / wall,
                           "shadow_rays_elided_per_frame": elided, "photon_roots_culled_per_frame": culled, "gpu_busy_frac_per_rank": busy}
    if "c4_tiles" in which and world == 1:
        # What bounds C4's 8-GPU band split, measured on this one GPU: the kernel time of each rank's share (bands r, r + 8,
        # r + 16 ... of 8 rows).  The slowest share is the frame time of an 8-GPU run (plus the 9.2 MB float gather).
        progress("extra: c4 band eighths")
        path = os.path.join(ROOT, "scenes", "examples3", "101_julia_shiny.xml")
        scene = capi.Scene(path, 0.0)
        ctx.upload(scene)
        stream = torch.cuda.current_stream().cuda_stream
        eighth_ms = []
        for r in range(8):
            bands = capi.Bands(driver.BAND_ROWS, r, 8)
            buf = torch.empty((ctx.pixel_count(bands), 4), dtype=torch.float32, device=dev)
            ms = []
            for rep in range(1 + reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ctx.render_device(buf.data_ptr(), stream, bands)
                b.record()
                ctx.launch_status()
                torch.cuda.synchronize()
                if rep > 0:
                    ms.append(a.elapsed_time(b))
            eighth_ms.append(round(float(np.mean(ms)), 4))
        whole_ms = out["c4_tiles"]["sec_per_frame"] * 1e3
        out["c4_band_eighth_ms"] = eighth_ms
        out["c4_band_eighth"] = {"max_ms": max(eighth_ms), "mean_ms": round(float(np.mean(eighth_ms)), 4), "whole_frame_ms": round(whole_ms, 4),
                                 "implied_8gpu_speedup_ceiling": round(whole_ms / max(eighth_ms), 3),
                                 "note": "kernel time of rt_bands{8, r, 8} for r = 0..7 on ONE GPU (events on the launch stream): what each of 8 ranks would "
                                         "render; whole frame / slowest share = the strong-scaling ceiling of the band split before the gather"}
    if "c5_frames" in which and world == 1:
        # What bounds C5's 8-GPU frame split: rank 0's host time per gathered unit (every rank renders, rank 0 alone
        # consumes).  World 1 with the collective forced on (RT_DRIVER_FORCE_GATHER=1) runs rank 0's consume loop as an
        # 8-GPU run would; at 8 GPUs it must stay below an eighth of a unit's GPU time.
        progress("extra: c5 rank-0 host cost per unit")
        path = os.path.join(ROOT, "scenes", "examples3", "103_caustic_animation.xml")
        started_group = False
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29655")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            started_group = True
        os.environ["RT_DRIVER_FORCE_GATHER"] = "1"
        consume_s = [0.0, 0]
        real_render_units = driver.render_units

        def timed_render_units(renderer_, path_, units, consume, payload=None):
            def timed_consume(u, pic):
                t0 = time.perf_counter()
                consume(u, pic)
                consume_s[0] += time.perf_counter() - t0
                consume_s[1] += 1
            return real_render_units(renderer_, path_, units, timed_consume, payload)
        driver.render_units = timed_render_units
        try:
            driver.render_frames(renderer, path)      # warm-up
            consume_s[0], consume_s[1] = 0.0, 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                driver.render_frames(renderer, path)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        finally:
            driver.render_units = real_render_units
            os.environ.pop("RT_DRIVER_FORCE_GATHER", None)
            if started_group:
                dist.destroy_process_group()
        units_done = max(1, consume_s[1])
        out["c5_rank0_host_us_per_unit"] = round(consume_s[0] / units_done * 1e6, 2)
        out["c5_rank0_gather"] = {"consume_us_per_unit": round(consume_s[0] / units_done * 1e6, 2), "units": units_done,
                                  "loop_us_per_unit_world1": round(wall / units_done * 1e6, 2),
                                  "note": "one-rank RCCL group with the gather forced on: host time of rank 0's consume call per gathered unit "
                                          "(wait for the gather + hand the RGBA8 frame to the download stream), and the whole loop's wall time per unit "
                                          "at world 1 (parse + upload + enqueue + gather + consume on one rank)"}
    for key, name in (("c5_frames", "103_caustic_animation.xml"), ("c5mb8", "103_caustic_animation_mb8.xml")):
        if key not in which:
            continue
        progress("extra: " + key)
        path = os.path.join(ROOT, "scenes", "examples3", name)
        info = capi.Scene(path, 0.0).info
        if info.sub_frames > 1:
            times = [u[2] for u in driver.motion_blur_units(path, info)]
        else:
            times = [float(f32(f) / f32(info.frames - 1)) for f in range(info.frames)]
        rays, flops, skipped = count_rays(ctx, capi, path, times) if rank == 0 else (0, 0.0, {})
        rays, flops, traced = shared(rays), shared(flops), shared(rays - sum(skipped.values()))
        elided, culled = shared(skipped.get("shadow_rays_elided", 0)), shared(skipped.get("photon_roots_culled", 0))
        wall, busy = timed_job(lambda: driver.render_frames(renderer, path), renderer, reps, warmup, world, dev)
        out[key] = {"scene": "examples3/" + name, "split": "%d units round-robin over %d rank(s), pipelined gathers to rank 0" % (len(times), world),
                    "scaling": "strong", "repetitions": reps, "units": len(times), "frames": info.frames,
                    "units_per_s": len(times) * reps / wall, "sec_per_frame": wall / reps / info.frames, "sec_per_file": wall / reps,
                    "rays_per_file": rays, "flops_per_file": flops, "rays_per_s": rays * reps / wall,
                    "rays_traced_per_file": traced, "rays_traced_per_s": traced * reps / wall,
                    "shadow_rays_elided_per_file": elided, "photon_roots_culled_per_file": culled, "gpu_busy_frac_per_rank": busy}
    return out


def units_line(args, capi, driver, world, rank, local_rank, dev):
    """--workload c5mb: a step is the whole 160-unit file through driver.render_frames (strong scaling over N)."""
    import torch.distributed as dist
    renderer = driver.HipRenderer(local_rank)
    res = split_measurements(("c5mb8",), renderer, renderer.ctx, capi, driver, world, rank, dev, args.steps, args.warmup)["c5mb8"]
    if rank == 0:
        line = {"metric": "rays/sec", "value": res["rays_per_s"], "unit": "rays/s", "rays_traced_per_s": res["rays_traced_per_s"],
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": res["sec_per_file"] * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                "data": "scene-file",
                "config": {"workload": "C5 examples3/103_caustic_animation_mb8.xml: 20 frames x 8 motion-blur sub-frames = 160 units (photon pass + render each)",
                           "scene": res["scene"], "units": res["units"], "units_per_s": res["units_per_s"], "sec_per_frame": res["sec_per_frame"],
                           "rays_per_file": res["rays_per_file"], "rays_traced_per_file": res["rays_traced_per_file"],
                           "shadow_rays_elided_per_file": res["shadow_rays_elided_per_file"], "photon_roots_culled_per_file": res["photon_roots_culled_per_file"],
                           "gpu_busy_frac_per_rank": res["gpu_busy_frac_per_rank"], "parallelism": res["split"]},
                "roofline": None}
        busy_s = sum(res["gpu_busy_frac_per_rank"]) * res["sec_per_file"] * args.steps      # GPU seconds of all ranks
        achieved = res["flops_per_file"] * args.steps / busy_s / 1e12
        line["roofline"] = {"bound": "valu", "kernel": "trace_kernel (+ photon_kernel, ordered photon sum)", "achieved": achieved, "peak": PEAK_FP32_TFLOPS,
                            "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS, "traffic": None,
                            "note": "algorithmic flops of the 160 units / GPU-busy time of the render calls (upload + photon pass + trace per unit, "
                                    "event pairs on the render stream); per-kernel figures are on the c5 line"}
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


_RESULT_FD = None


def claim_stdout():
    """stdout carries ONE JSON line: libraries that write to file descriptor 1 themselves (RCCL prints a version banner
    there when a process group starts) get stderr instead; emit() writes the line to the real stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    data = (json.dumps(line) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def progress(msg):
    """Phase marker on stderr (stdout carries the one JSON line): a run that hangs says where."""
    print("[bench rank %s] %s" % (os.environ.get("RANK", "0"), msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + ["mesh", "c5mb"])
    ap.add_argument("--no-extra", action="store_true", help="skip the C4-tiles / C5-frames measurements under 'extra'")
    ap.add_argument("--extra-reps", type=int, default=3)
    ap.add_argument("--bvh-min", type=int, default=None, help="developer: RT_OPT_BVH_MIN_TRIANGLES before the upload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-call-output", action="store_true",
                    help="developer: rt_render_device + rt_quantize_device per step instead of rt_render_rgba8_device (same bytes)")
    args = ap.parse_args()
    claim_stdout()

    import torch
    import torch.distributed as dist
    from raytracer_amd import capi, driver

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RT_DIST_BACKEND=gloo: rehearsal of the N > 1 code on a box with fewer GPUs than ranks (the ranks share the GPUs
    # there are, the collectives go through gloo); the numbers of such a run mean nothing
    rehearsal = os.environ.get("RT_DIST_BACKEND", "nccl") == "gloo"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # RT_BENCH_FORCE_GATHER=1: a one-rank run goes through the process group and the gathers too (hardware test of the
    # N > 1 step's RCCL calls on a one-GPU box; developer switch, the line then says so in config.parallelism)
    collective = world > 1 or os.environ.get("RT_BENCH_FORCE_GATHER", "0") not in ("", "0")
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    progress("process group ready: world %d, device %s" % (world, dev))

    if args.workload == "c5mb":
        return units_line(args, capi, driver, world, rank, local_rank, dev)

    if args.workload == "mesh":     # developer workload for the BVH: 104k-triangle synthetic mesh, 1920x1080
        import pathlib
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import mesh_scene
        rel, t, label = "mesh.xml", 0.0, "synthetic 104k-triangle mesh 1920x1080 (tests/mesh_scene.py)"
        scene = capi.Scene(mesh_scene.write_mesh_scene(pathlib.Path(tempfile.mkdtemp()), 160, 320, width=1920, height=1080), t)
    else:
        rel, t, label = WORKLOADS[args.workload]
        scene = capi.Scene(os.path.join(ROOT, "scenes", rel), t)
    ctx = capi.Context(local_rank)
    if collective:
        # RCCL's gather kernels of step i sit on the GPU when step i + 1 is dispatched (include/rt_hip.h, RT_OPT_SHARED_GPU)
        ctx.set_option(capi.RT_OPT_SHARED_GPU, 1)
    if args.bvh_min is not None:
        ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, args.bvh_min)
    ctx.upload(scene)
    w, h = scene.width, scene.height
    n_pix = w * h
    frame = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
    # two RGBA8 frames: the gather of step i runs on RCCL's stream while step i+1 renders
    frame8 = [torch.empty((h, w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
    gathered = [[torch.empty_like(frame8[0]) for _ in range(world)] for _ in range(2)] if (collective and rank == 0) else [None, None]
    pending = [None, None]
    stream = torch.cuda.current_stream().cuda_stream
    has_photons = scene.desc.photon_steps > 0

    progress("counters launch")
    # ---- counters (untimed): rays and algorithmic flops of one step on this rank ------------------
    # Algorithmic work = the counts of the reference's formulation (nearest hit by a scan over all
    # objects, then shade once), which the counters of a BVH-less launch reproduce exactly
    # (pinned to the oracle's by tests/test_gpu_parity.py).  Culling and the BVH skip part of that work;
    # they do not change the numerator.
    if args.bvh_min is None:
        ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, 1 << 40)
        ctx.upload(scene)
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    pst = ctx.photon_generate()
    ctx.render_device(frame.data_ptr(), stream)
    torch.cuda.synchronize()
    rst = ctx.fetch_stats()
    stats = {k: v + pst.as_dict()[k] for k, v in rst.as_dict().items() if k != "kernel_ms"}
    rays_per_step = stats["primary_rays"] + stats["secondary_rays"] + stats["shadow_rays"] + stats["photon_segments"]
    flops_per_step = algorithmic_flops(stats)
    flops_note = "algorithmic flops = counts of the reference's formulation (scan over all objects, shade once) x bench.py:FLOPS"
    if args.bvh_min is None:
        ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, capi.RT_DEFAULT_BVH_MIN_TRIANGLES)
        ctx.upload(scene)
        # a scene whose long triangle runs go through the BVH is priced at the work the traversal executes
        # (node visits and leaf tests): against a scan of 10^5 triangles per ray the ratio would say nothing
        ctx.render_device(frame.data_ptr(), stream)
        torch.cuda.synchronize()
        executed = ctx.fetch_stats().as_dict()
        # (only where a scan is no sensible formulation any more: scenes of 256 triangles or more.  C2's 64 triangles go
        # through the BVH too since round 2, by a 4 % margin; its numerator stays the reference's scan, as in round 1)
        if executed.get("bvh_node_visits", 0) > 0 and scene.desc.n_triangles >= 256:
            flops_per_step = algorithmic_flops({k: v + pst.as_dict().get(k, 0) for k, v in executed.items() if k != "kernel_ms"})
            flops_note = "BVH scene: flops of the executed traversal (node visits, leaf tests) x bench.py:FLOPS"
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)

    trace_steps = os.environ.get("RT_BENCH_TRACE_STEPS", "0") not in ("", "0")     # developer: host time of a step's calls on stderr

    def step(i, events=None):
        buf = i % 2
        if trace_steps:      # a step that does not return within 30 s leaves the stack of every Python thread on stderr
            import faulthandler
            faulthandler.cancel_dump_traceback_later()
            faulthandler.dump_traceback_later(30, repeat=False, file=sys.stderr)
        t_in = time.perf_counter()
        if pending[buf] is not None:
            pending[buf].wait()          # the frame gathered two steps ago has left this buffer
            pending[buf] = None
        t_waited = time.perf_counter()
        if has_photons:
            ctx.photon_generate_device(stream)
        if events is not None:
            events[0].record()
        if args.two_call_output:
            ctx.render_device(frame.data_ptr(), stream)
        else:       # the picture leaves the kernel as Color::scaleOut bytes (rt_render_rgba8_device)
            ctx.render_rgba8_device(frame8[buf].data_ptr(), stream)
        if events is not None:
            events[1].record()
        if args.two_call_output:
            ctx.quantize_device(frame.data_ptr(), frame8[buf].data_ptr(), 1.0, n_pix, stream)
        t_enqueued = time.perf_counter()
        if collective:
            driver.settle_for_gloo(frame8[buf])      # (rehearsals over gloo only: see there)
            pending[buf] = dist.gather(frame8[buf], gathered[buf], dst=0, async_op=True)
        if trace_steps:
            progress("step %d: wait for the gather of two steps ago %.3f ms, enqueue %.3f ms, gather call %.3f ms" % (
                i, (t_waited - t_in) * 1e3, (t_enqueued - t_waited) * 1e3, (time.perf_counter() - t_enqueued) * 1e3))

    def fence():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    progress("warm-up")
    for i in range(args.warmup):
        step(i)
    fence()
    progress("timed steps")
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i])
    fence()
    elapsed = time.perf_counter() - t0
    if collective:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    progress("timed steps done: %.3f ms per step" % (elapsed / args.steps * 1e3))
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    ctx.launch_status()      # raises if the Julia kernel's watchdog abandoned the last launch

    extra = None
    if args.workload == "c2" and not args.no_extra:
        extra = split_measurements(("c4_tiles", "c5_frames", "c5mb8"), driver.HipRenderer(local_rank), ctx, capi, driver,
                                   world, rank, dev, args.extra_reps, 1)
        extra["c4_sec_per_frame"] = extra["c4_tiles"]["sec_per_frame"]
        extra["c5_sec_per_frame"] = extra["c5_frames"]["sec_per_frame"]
        extra["note"] = ("through raytracer_amd/driver.py at this world size, fixed total work: parse + upload + photon pass + trace + "
                         "gather + quantise + RGBA8 download; PNG/APNG encode excluded")

    if rank == 0:
        value = rays_per_step * world * args.steps / elapsed
        achieved = flops_per_step / (kernel_ms * 1e-3) / 1e12
        rays_traced = rays_per_step - stats.get("photon_roots_culled", 0) - stats.get("shadow_rays_elided", 0)
        line = {
            "metric": "rays/sec", "value": value, "unit": "rays/s",
            # `value` counts the rays of the reference's formulation (what rays/s is quoted on); this one only those the launch traces
            "rays_traced_per_s": rays_traced * world * args.steps / elapsed,
            "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "scene-file",
            "config": {"workload": label, "scene": rel, "t": t, "rays_per_frame": rays_per_step,
                       "photon_roots_culled_per_frame": stats.get("photon_roots_culled", 0),   # of the photon pass's roots, counted in rays_per_frame, dropped before the pass: DESIGN.md 4.6
                       "shadow_rays_elided_per_frame": stats.get("shadow_rays_elided", 0),   # counted in rays_per_frame (rays of the reference's formulation), not traced: DESIGN.md 4.4
                       "rays_traced_per_frame": rays_traced,   # what the plain launch traces
                       "frames_per_step": world, "sec_per_frame": elapsed / args.steps,
                       "parallelism": "frames sharded over ranks, RGBA8 frames gathered to rank 0 over RCCL (gather of a step overlaps the next step's render; all gathers complete inside the timed region; RT_OPT_SHARED_GPU)" if world > 1 else ("1 GPU, RT_BENCH_FORCE_GATHER: the frame gathered to itself over RCCL" if collective else "1 GPU")},
            "roofline": {"bound": "valu", "kernel": "trace_julia_kernel" if scene.desc.n_julias > 0 else "trace_kernel", "achieved": achieved, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS, "traffic": measured_traffic(args.workload),
                         "executed_valu": executed_valu(args.workload, kernel_ms),
                         "kernel_ms": kernel_ms, "algorithmic_flops_per_launch": flops_per_step,
                         "note": "FP32 VALU bound (no HBM or MFMA roofline applies: scene is KBs, nothing is a contraction); " + flops_note + "; see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, rays_per_step)
            if args.workload != "mesh":
                ref = cpu_reference(os.path.join(ROOT, "scenes", rel), rays_per_step, still_at=t if has_photons else None, ratio_key=args.workload)
                if ref is not None:
                    line["cpu_reference"] = ref
        if extra is not None:
            if world == 1 and not args.no_cpu_baseline:
                c5 = os.path.join(ROOT, "scenes", "examples3", "103_caustic_animation.xml")
                ref = cpu_reference(c5, extra["c5_frames"]["rays_per_file"] / extra["c5_frames"]["frames"], still_at=0.5, ratio_key="c5")
                if ref is not None:
                    extra["c5_cpu_reference"] = ref
            line["extra"] = extra
        emit(line)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
