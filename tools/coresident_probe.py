#!/usr/bin/env python3
"""Developer probe (GPU box): what does a long-lived kernel of ANOTHER stream cost a render that shares the GPU with it —
the situation of bench.py's N > 1 step, where RCCL's gather kernels of step i sit on rank 0's CUs while step i + 1 renders?
tools/liboccupier.so holds `blocks` workgroups of 256 lanes with `lds` KB of LDS each for `ms` milliseconds on a side
stream; the C2 frame (or the workload given) is rendered on the main stream right behind its launch and timed with events.

  tools/coresident_probe.py [workload] [reps]
      PROBE_HOLD=f       the stand-in stays for f x the render's time alone (default 1.5: it outlives the render)
      PROBE_SHORT=1      three stand-in shapes instead of seven
      PROBE_PIPELINED=1  bench.py's N > 1 step instead: the stand-in is enqueued behind an event of render i, render i + 1
                         follows at once; ms per step with RT_OPT_SHARED_GPU off and on (RT_STATIC_SHARE=n: sixteenths of the
                         tiles owned, developer override of both)
  build the stand-in first: hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/occupier.hip -o tools/liboccupier.so

Prints the render time alone and beside each occupier shape.  trace_kernel is a persistent grid sized to fill the GPU
(three workgroups per CU at C2): a workgroup that finds no room starts when another ends, i.e. at the end of the frame,
and then still owns its static share of the tiles (DevJob::static_blocks) — that tail is what this measures."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from raytracer_amd import capi  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    occ = C.CDLL(os.path.join(ROOT, "tools", "liboccupier.so"))
    occ.occupier_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    rel, t, _label = bench.WORKLOADS[workload]
    scene = capi.Scene(os.path.join(ROOT, "scenes", rel), t)
    ctx = capi.Context(0)
    ctx.upload(scene)
    if scene.desc.photon_steps > 0:
        ctx.photon_generate()
    dev = torch.device("cuda", 0)
    out8 = torch.empty((scene.height, scene.width, 4), dtype=torch.uint8, device=dev)
    main_stream = torch.cuda.current_stream()
    side = torch.cuda.Stream(dev)

    def timed(blocks, lds_kb, ms):
        times = []
        for _ in range(reps):
            torch.cuda.synchronize()
            if blocks:
                rc = occ.occupier_launch(blocks, 256, lds_kb * 1024, ms, C.c_void_p(side.cuda_stream))
                assert rc == 0, rc
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main_stream)
            ctx.render_rgba8_device(out8.data_ptr(), main_stream.cuda_stream)
            b.record(main_stream)
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        ctx.launch_status()
        return float(np.median(times)), float(np.min(times)), float(np.max(times))

    def pipelined(blocks, lds_kb, ms, steps=60):
        """bench.py's N > 1 step with the collective replaced by the occupier: it is enqueued on the side stream behind an
        event of render i (as RCCL's kernel is) and render i + 1 follows on the main stream at once."""
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.render_rgba8_device(out8.data_ptr(), main_stream.cuda_stream)
            if blocks:
                done = torch.cuda.Event()
                done.record(main_stream)
                side.wait_event(done)
                rc = occ.occupier_launch(blocks, 256, lds_kb * 1024, ms, C.c_void_p(side.cuda_stream))
                assert rc == 0, rc
        torch.cuda.synchronize()
        ctx.launch_status()
        return (time.perf_counter() - t0) / steps * 1e3

    if os.environ.get("PROBE_PIPELINED"):
        for shared in (0, 1, 0, 1):
            ctx.set_option(capi.RT_OPT_SHARED_GPU, shared)
            pipelined(0, 0, 0.0, 10)
            base = pipelined(0, 0, 0.0)
            line = "%s RT_OPT_SHARED_GPU=%d: %.3f ms per step alone" % (workload, shared, base)
            for blocks, lds_kb, share in ((16, 16, 0.4), (32, 16, 0.4), (32, 64, 0.4), (64, 32, 0.8)):
                line += "; beside %d x %d KB for %.1f of a frame %.3f" % (blocks, lds_kb, share, pipelined(blocks, lds_kb, share * base))
            print(line, flush=True)
        return

    alone = timed(0, 0, 0.0)
    print("%s alone: median %.3f ms (min %.3f, max %.3f)" % (workload, *alone), flush=True)
    # PROBE_HOLD: how long the occupier stays, as a share of the render's time alone (default 1.5: it outlives the render;
    # a collective that ends mid-frame is 0.4)
    hold = float(os.environ.get("PROBE_HOLD", "1.5")) * alone[0]
    shapes = ((8, 16), (32, 16), (64, 16), (32, 48), (32, 64), (64, 64), (256, 64))
    if os.environ.get("PROBE_SHORT"):
        shapes = ((8, 16), (32, 16), (64, 64))
    for blocks, lds_kb in shapes:
        r = timed(blocks, lds_kb, hold)
        print("%s beside %3d workgroups x 256 lanes x %2d KB LDS held for %.1f ms: median %.3f ms (min %.3f, max %.3f)  %+.1f %%" % (
            workload, blocks, lds_kb, hold, *r, (r[0] / alone[0] - 1.0) * 100.0), flush=True)


if __name__ == "__main__":
    main()
