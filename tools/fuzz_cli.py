#!/usr/bin/env python3
"""Developer tool (GPU box): the drop-in CLI (raytracer_amd/rt_amd: loader, kernels, quantiser, PNG writer) against the
reference's own executable (oracle/_ref/raytracer_ostrict, built from the reference's sources in the build container) on the
seeded random scenes of tests/mesh_scene.py: the two PNG files must be the same bytes.
tools/fuzz_cli.py FIRST LAST"""
import os
import pathlib
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
        a, b = str(tmp / "ref.png"), str(tmp / "hip.png")
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
