#!/usr/bin/env python3
"""Developer tool (GPU box): the drop-in CLI (raytracer_amd/rt_amd: loader, kernels, quantiser, PNG writer) against the
reference's own executable (oracle/_ref/raytracer_ostrict, built from the reference's sources in the build container) on the
seeded random scenes of tests/mesh_scene.py: the two PNG files must be the same bytes.
tools/fuzz_cli.py FIRST LAST [animate]   (animate: keyframed attributes, 2-4 frames, motion blur on every other seed: the four
render loops of src/main.cpp and the APNG writer)"""
import os
import pathlib
import random
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mesh_scene  # noqa: E402


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "raytracer_ostrict")
    cli = os.path.join(ROOT, "raytracer_amd", "rt_amd")
    tmp = pathlib.Path(tempfile.mkdtemp())
    bad, n = [], 0
    for seed in range(first, last):
        if seed % 7 == 3:
            continue   # depth of field: the reference is not reproducible
        scene = mesh_scene.write_random_scene(tmp, seed)
        if len(sys.argv) > 3:
            rnd = random.Random(5000 + seed)

            def animate(m):
                if rnd.random() < 0.2:
                    v = float(m.group(2))
                    return '%s="%s;%.3f(%s,%.2f);%.3f"' % (m.group(1), m.group(2), v + rnd.uniform(-0.5, 0.5), rnd.choice("liob"), rnd.uniform(0.2, 0.6), v + rnd.uniform(-0.5, 0.5))
                return m.group(0)
            text = re.sub(r'\b(x|y|z|theta|radius)="(-?\d+\.\d+)"', animate, open(scene).read())
            tags = rnd.choice(['<animation length="%.2f" fps="10"/>' % rnd.choice([0.2, 0.3, 0.4]), ""])
            if seed % 2:
                tags += '<motionblur subframes="%s"/>' % rnd.choice(["2", "3", "2;4"])
            text = text.replace("<camera>", tags + "<camera>", 1)
            open(scene, "w").write(text)
        a, b = str(tmp / "ref.png"), str(tmp / "hip.png")
        if os.environ.get("FUZZ_TRACE"):       # (the reference takes minutes for some animated Julia scenes: say which one is running)
            print("begin seed", seed, flush=True)
        subprocess.check_call([ref_exe, scene, a], stdout=subprocess.DEVNULL)
        subprocess.check_call([cli, scene, b], stdout=subprocess.DEVNULL)
        n += 1
        if open(a, "rb").read() != open(b, "rb").read():
            bad.append(seed)
            print("seed %d: files differ" % seed, flush=True)
        if seed % 50 == 0:
            print("... seed", seed, flush=True)
    print("compared %d scenes (seeds %d..%d), differing: %s" % (n, first, last - 1, bad), flush=True)


if __name__ == "__main__":
    main()
