#!/usr/bin/env python3
"""Developer tool (GPU box): tools/fuzz_driver.py FIRST LAST  -> the pipelined driver (raytracer_amd/driver.py: three lanes,
photon passes confirmed a round late) against its lock-step loop (RT_DRIVER_SEQUENTIAL=1) on the seeded random scenes of
tests/mesh_scene.py with keyframed attributes, 2-4 frames and motion blur on every other seed (the transformation of
tools/fuzz_cli.py): the RGBA8 frames must be the same bytes.  No oracle and no reference executable: fast."""
import os
import pathlib
import random
import re
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import driver  # noqa: E402
import mesh_scene  # noqa: E402


def animated_scene(tmp, seed):
    scene = mesh_scene.write_random_scene(tmp, seed)
    rnd = random.Random(5000 + seed)

    def animate(m):
        if rnd.random() < 0.2:
            v = float(m.group(2))
            return '%s="%s;%.3f(%s,%.2f);%.3f"' % (m.group(1), m.group(2), v + rnd.uniform(-0.5, 0.5), rnd.choice("liob"), rnd.uniform(0.2, 0.6), v + rnd.uniform(-0.5, 0.5))
        return m.group(0)
    text = re.sub(r'\b(x|y|z|theta|radius)="(-?\d+\.\d+)"', animate, open(scene).read())
    tags = '<animation length="%.2f" fps="10"/>' % rnd.choice([0.2, 0.3, 0.4])
    if seed % 2:
        tags += '<motionblur subframes="%s"/>' % rnd.choice(["2", "3", "2;4"])
    open(scene, "w").write(text.replace("<camera>", tags + "<camera>", 1))
    return scene


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    tmp = pathlib.Path(tempfile.mkdtemp())
    fast, slow = driver.HipRenderer(0), driver.HipRenderer(0)
    bad = []
    for seed in range(first, last):
        if os.environ.get("FUZZ_TRACE"):
            print("begin seed", seed, flush=True)
        scene = animated_scene(tmp, seed)
        os.environ["RT_DRIVER_SEQUENTIAL"] = "0"
        try:
            a = driver.render_frames(fast, scene)[0]
        except RuntimeError as e:
            if "Depth of field" in str(e):
                continue
            raise
        os.environ["RT_DRIVER_SEQUENTIAL"] = "1"
        b = driver.render_frames(slow, scene)[0]
        if len(a) != len(b) or any(not np.array_equal(x, y) for x, y in zip(a, b)):
            bad.append(seed)
            print("seed %d: frames differ" % seed, flush=True)
        if seed % 50 == 0:
            print("... seed", seed, flush=True)
    print("checked seeds %d..%d, differing: %s" % (first, last - 1, bad), flush=True)


if __name__ == "__main__":
    main()
