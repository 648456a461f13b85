#!/usr/bin/env python3
"""Ad-hoc parity/timing probe: HIP path vs the oracle on scene files (developer tool, GPU box)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import capi  # noqa: E402
import oracle_lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="+")
    ap.add_argument("--t", type=float, default=0.0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--ss", type=int, default=0)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--timing", action="store_true", help="library built with -DRT_PHASE_TIMING: print its section clocks")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--save", default="")
    ap.add_argument("--bvh-min", type=int, default=None)
    args = ap.parse_args()
    ctx = capi.Context(0)
    if args.bvh_min is not None:
        ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, args.bvh_min)
    for path in args.scenes:
        sc = capi.Scene(path, args.t)
        if args.width or args.height or args.ss:
            sc.set_camera(args.width, args.height, args.ss)
        ctx.upload(sc)
        ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1 if args.stats else 0)
        pst = ctx.photon_generate()
        img, st = ctx.render()
        times = []
        for _ in range(args.reps):
            _, s2 = ctx.render()
            times.append(s2.kernel_ms)
        print("%s %dx%d ss%d: photon %.3f ms, render kernel ms %s" % (os.path.basename(path), sc.width, sc.height,
              sc.desc.camera.supersampling, pst.kernel_ms, ["%.3f" % t for t in times]), flush=True)
        if args.stats or args.timing:
            if args.stats:
                print("  hip stats", {k: v for k, v in st.as_dict().items() if v}, flush=True)
            import ctypes as C
            words = (C.c_uint64 * 96)()
            nw = capi.hip_lib().rt_debug_counters(ctx._ctx, words, 96)
            names = ["query_phases", "query_lanes", "march_phases", "march_lanes", "normal_phases", "normal_lanes",
                     "resolve_phases", "resolve_lanes", "de_wave_iter_slots", "query_cycles", "march_cycles",
                     "normal_cycles", "resolve_cycles", "tri_cycles", "sphere_cycles",
                     "r_node", "r_light", "r_shadowsetup", "r_fresnel", "r_return", "r_fetch",
                     "m_tail_cycles", "m_idle_cycles", "m_refill_cycles", "t_wait_cycles", "m_tail_blocks", "m_tail_lanes",
                     "step_blocks_1", "step_blocks_2", "step_blocks_4", "step_blocks_8", "step_blocks_16", "step_blocks_32", "step_blocks_more",
                     "step_class_switch", "step_block_sum", "ring_samples", "ring_march_sum", "ring_normal_sum", "ring_ready_t_sum", "ring_ready_s_sum",
                     "march_has_sum", "march_polls", "chunk_tests", "chunk_passed", "chunk_lanes", "sphere_culls", "sphere_passed",
                     "resolve_mixed", "resolve_minority_lanes", "photon_cull_violations"]
            dbg = dict(zip(names, list(words)[18:nw]))
            print("  scheduler", {k: v for k, v in dbg.items() if v}, flush=True)
            if dbg["march_cycles"]:
                mt = sum(dbg[k] for k in ("march_cycles", "m_tail_cycles", "m_refill_cycles", "m_idle_cycles", "normal_cycles"))
                print("   marcher waves: " + ", ".join("%s %.1f%%" % (k, 100.0 * dbg[k] / mt) for k in
                      ("march_cycles", "m_tail_cycles", "m_refill_cycles", "m_idle_cycles", "normal_cycles")), flush=True)
                blocks = dbg["de_wave_iter_slots"] if args.timing else dbg["de_wave_iter_slots"] // 256
                print("   estimator blocks %d (%.0f clk each), blocks per march phase %.2f, tail phases %d (%.0f clk each, %.1f lanes), lanes holding a job %.1f" % (
                    blocks, dbg["march_cycles"] / max(1, blocks), blocks / max(1, dbg["march_phases"]), dbg["march_phases"],
                    dbg["m_tail_cycles"] / max(1, dbg["march_phases"]), dbg["m_tail_lanes"] / max(1, dbg["march_phases"]), dbg["march_lanes"] / max(1, dbg["march_phases"])), flush=True)
                if args.timing:
                    print("   busy lanes per estimator block %.1f" % (dbg["march_has_sum"] / max(1, blocks)), flush=True)
                st_ = sum(dbg[k] for k in ("resolve_cycles", "query_cycles", "t_wait_cycles"))
                print("   shader waves: " + ", ".join("%s %.1f%%" % (k, 100.0 * dbg[k] / max(1, st_)) for k in ("resolve_cycles", "query_cycles", "t_wait_cycles")), flush=True)
            for ph in ("query", "march", "normal", "resolve"):
                if dbg[ph + "_phases"]:
                    print("   %s: lanes/phase %.1f, cycles/phase %.0f, share of cycles %.1f%%" % (
                        ph, dbg[ph + "_lanes"] / dbg[ph + "_phases"], dbg[ph + "_cycles"] / dbg[ph + "_phases"],
                        100.0 * dbg[ph + "_cycles"] / max(1, sum(dbg[q + "_cycles"] for q in ("query", "march", "normal", "resolve")))), flush=True)
            print("   inside query: triangle runs %.1f%%, sphere runs %.1f%% of query cycles" % (
                100.0 * dbg["tri_cycles"] / max(1, dbg["query_cycles"]), 100.0 * dbg["sphere_cycles"] / max(1, dbg["query_cycles"])), flush=True)
            rc = max(1, dbg["resolve_cycles"])
            print("   inside resolve: " + ", ".join("%s %.1f%%" % (k[2:], 100.0 * dbg[k] / rc) for k in
                  ("r_node", "r_light", "r_shadowsetup", "r_fresnel", "r_return", "r_fetch")), flush=True)
            if dbg["de_wave_iter_slots"] and args.stats:
                print("   DE lane-iterations / wave slots = %.3f" % (st.julia_de_iterations / dbg["de_wave_iter_slots"]), flush=True)
        if args.save:
            capi.png_write(args.save, capi.quantize(img))
        if not args.no_oracle:
            t0 = time.time()
            ph = oracle_lib.Photons(sc) if sc.desc.photon_steps > 0 else None
            t1 = time.time()
            ref, ost = oracle_lib.render(sc, ph, threads=args.threads)
            t2 = time.time()
            neq = img.view(np.uint32) != ref.view(np.uint32)
            print("  oracle photon %.2fs render %.2fs; float words differing %d of %d (pixels %d)" % (
                t1 - t0, t2 - t1, int(neq.sum()), neq.size, int(neq.any(axis=2).sum())), flush=True)
            print("  8-bit:", oracle_lib.image_diff(capi.quantize(img), capi.quantize(ref)), flush=True)
            if args.stats:
                od = ost.as_dict()
                hd = st.as_dict()
                print("  counter diffs", {k: (hd[k], od[k]) for k in od if k != "kernel_ms" and hd[k] != od[k]}, flush=True)
            if ph is not None:
                worst = 0.0
                for o in range(sc.desc.n_objects):
                    a = ctx.photon_download(o)
                    b = ph.texture(o)
                    if b.any() or a.any():
                        worst = max(worst, float(np.abs(a - b).max() / max(1e-30, np.abs(b).max())))
                print("  photon atlas worst rel err %.3e; deposits hip %d oracle %d" % (worst, pst.photon_deposits, ph.stats.photon_deposits), flush=True)


if __name__ == "__main__":
    main()
