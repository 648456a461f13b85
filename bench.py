#!/usr/bin/env python3
"""bench.py — rays/s of the trace path on N MI355X GPUs of one node.

A "step" is one pass of the hot path over one batch: rt_photon_generate + rt_render of one frame of
the workload on every rank, the frame quantised on the GPU (Color::scaleOut) and, for N > 1, gathered
to rank 0 over RCCL.  The scene is uploaded (resident in HBM) before the timed region.

Workload (config.workload): BASELINE.json configs[1], examples/example9.xml at 3840x2160 — the
largest single-GPU configuration.  `--workload` selects another config for developer runs.

N > 1: frames are independent units: every rank renders its own full frame per step and
rank 0 collects the RGBA8 frames ("scaling": "weak").  value = rays of all ranks / max-over-ranks time.

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
ISSUE_CEILING_TFLOPS = 39.3   # 256 CUs x 64 lanes x 2.4 GHz: one non-fused mul or add per lane and clock


def measured_traffic(workload):
    """HBM bytes per trace_kernel launch from the committed PMC passes (profiles/traffic.json, made by
    tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs of this bench; FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for gfx950).  None when no pass exists for the workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def algorithmic_flops(stats_dict):
    return float(sum(stats_dict.get(k, 0) * w for k, w in FLOPS.items()))


def cpu_baseline(scene, n_rays_frame):
    """The oracle (CPU restatement, "port") timed on this box's host cores on the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    cores = min(16, os.cpu_count() or 1)
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
    return {"value": rays / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": "oracle/liboracle.so (CPU restatement, pinned bit-exact to the reference's O-strict build), every %d-th row "
                      "of the frame (%d rows) x %d repetitions%s, %.2f s wall on %d threads" % (
                step, (cam.height + step - 1) // step, reps, " + serial photon pass" if photons else "", dt, cores)}


def cpu_reference(scene_path, n_rays_frame):
    """The reference's own program (oracle/_ref/raytracer_ostrict: its unmodified sources compiled by
    oracle/Makefile, g++ -O2) on the same scene file with threads = host cores; time = its own
    'Finished in' line (trace + PNG encode, src/main.cpp:23-35).  None where the binary is absent."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "raytracer_ostrict")
    if not os.path.exists(exe):
        return None
    cores = min(16, os.cpu_count() or 1)
    try:
        with tempfile.TemporaryDirectory() as tmp:
            src_dir = os.path.dirname(scene_path)
            for name in os.listdir(src_dir):
                os.symlink(os.path.join(src_dir, name), os.path.join(tmp, name + ".lnk") if name == os.path.basename(scene_path) else os.path.join(tmp, name))
            xml = open(scene_path).read()
            xml = re.sub(r"<scene\b", '<scene threads="%d"' % cores, re.sub(r'\sthreads="[^"]*"', "", xml), count=1)
            path = os.path.join(tmp, "bench_scene.xml")
            with open(path, "w") as f:
                f.write(xml)
            out = subprocess.run([exe, path, os.path.join(tmp, "out.png")], capture_output=True, text=True, timeout=240).stdout
        m = re.search(r"Finished in ([0-9.eE+-]+)\s*s", out)
        if not m:
            return None
        sec = float(m.group(1))
        return {"value": n_rays_frame / sec, "unit": "rays/s", "cores": cores, "kind": "reference",
                "sample": "oracle/_ref/raytracer_ostrict (reference sources, g++ -O2 -pthread), whole frame, threads=%d, its own "
                          "'Finished in' timer = %.2f s (trace + PNG encode); rays = the nearest-hit formulation's count" % (cores, sec)}
    except Exception:   # noqa: BLE001 - a missing or failing reference binary must not break the bench line
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + ["mesh"])
    ap.add_argument("--bvh-min", type=int, default=None, help="developer: RT_OPT_BVH_MIN_TRIANGLES before the upload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

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
    if args.bvh_min is not None:
        ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, args.bvh_min)
    ctx.upload(scene)
    w, h = scene.width, scene.height
    n_pix = w * h
    frame = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
    # two RGBA8 frames: the gather of step i runs on RCCL's stream while step i+1 renders
    frame8 = [torch.empty((h, w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
    gathered = [[torch.empty_like(frame8[0]) for _ in range(world)] for _ in range(2)] if (world > 1 and rank == 0) else [None, None]
    pending = [None, None]
    stream = torch.cuda.current_stream().cuda_stream
    has_photons = scene.desc.photon_steps > 0

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
        if executed.get("bvh_node_visits", 0) > 0:
            flops_per_step = algorithmic_flops({k: v + pst.as_dict().get(k, 0) for k, v in executed.items() if k != "kernel_ms"})
            flops_note = "BVH scene: flops of the executed traversal (node visits, leaf tests) x bench.py:FLOPS"
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)

    def step(i, events=None):
        buf = i % 2
        if pending[buf] is not None:
            pending[buf].wait()          # the frame gathered two steps ago has left this buffer
            pending[buf] = None
        if has_photons:
            ctx.photon_generate_device(stream)
        if events is not None:
            events[0].record()
        ctx.render_device(frame.data_ptr(), stream)
        if events is not None:
            events[1].record()
        ctx.quantize_device(frame.data_ptr(), frame8[buf].data_ptr(), 1.0, n_pix, stream)
        if world > 1:
            pending[buf] = dist.gather(frame8[buf], gathered[buf], dst=0, async_op=True)

    def fence():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    ctx.launch_status()      # raises if the Julia kernel's watchdog abandoned the last launch

    if rank == 0:
        value = rays_per_step * world * args.steps / elapsed
        achieved = flops_per_step / (kernel_ms * 1e-3) / 1e12
        line = {
            "metric": "rays/sec", "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "scene": rel, "t": t, "rays_per_frame": rays_per_step,
                       "frames_per_step": world, "sec_per_frame": elapsed / args.steps,
                       "parallelism": "frames sharded over ranks, RGBA8 frames gathered to rank 0 over RCCL (gather of a step overlaps the next step's render; all gathers complete inside the timed region)" if world > 1 else "1 GPU"},
            "roofline": {"bound": "valu", "kernel": "trace_julia_kernel" if scene.desc.n_julias > 0 else "trace_kernel", "achieved": achieved, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS, "traffic": measured_traffic(args.workload),
                         "frac_of_unfused_issue_ceiling": achieved / ISSUE_CEILING_TFLOPS,
                         "kernel_ms": kernel_ms, "algorithmic_flops_per_launch": flops_per_step,
                         "note": "FP32 VALU bound (no HBM or MFMA roofline applies: scene is KBs, nothing is a contraction); " + flops_note + "; see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, rays_per_step)
            if args.workload != "mesh" and not has_photons:
                ref = cpu_reference(os.path.join(ROOT, "scenes", rel), rays_per_step)
                if ref is not None:
                    line["cpu_reference"] = ref
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
