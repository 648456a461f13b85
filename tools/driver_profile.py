"""Developer probe (GPU box): where the host time of driver.render_frames goes (cProfile, top cumulative)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import driver  # noqa: E402

path = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "scenes/examples3/103_caustic_animation_mb8.xml")
r = driver.HipRenderer(0)
driver.render_frames(r, path)
t0 = time.perf_counter()
driver.render_frames(r, path)
print("render_frames: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
driver.render_frames(r, path)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
