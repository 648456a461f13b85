#!/usr/bin/env python3
"""Developer tool (GPU box): the driver's unit loop (driver.render_frames: three contexts per GPU, units alternating) on an
animation whose frames are LARGE enough to own tiles — the caustic animation at SIZE x SIZE instead of 300 x 300 — with
RT_OPT_SHARED_GPU off and on: the contexts share the GPU with each other, which is the situation the option is for.

  tools/lanes_big_frames.py [size] [reps]        RT_DRIVER_LANES / RT_DRIVER_OVERLAP as for the driver"""
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi, driver  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    src = os.path.join(ROOT, "scenes", "examples3", "103_caustic_animation.xml")
    text = open(src).read()
    text, n = re.subn(r'horizontal="300" vertical="300"', 'horizontal="%d" vertical="%d"' % (size, size), text)
    assert n == 1
    path = os.path.join(ROOT, "scenes", "examples3", "_lanes_big_%d.xml" % size)
    open(path, "w").write(text)
    try:
        renderer = driver.HipRenderer(0)
        for shared in (0, 1, 0, 1):
            renderer.ctx.set_option(capi.RT_OPT_SHARED_GPU, shared)
            driver.render_frames(renderer, path)          # warm-up (and the twins take the option over: lanes())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frames = 0
            for _ in range(reps):
                frames += len(driver.render_frames(renderer, path)[0])
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            print("%d x %d, RT_OPT_SHARED_GPU=%d: %.3f ms per frame (%d frames)" % (size, size, shared, wall / frames * 1e3, frames), flush=True)
        renderer.close()
    finally:
        os.remove(path)


if __name__ == "__main__":
    main()
