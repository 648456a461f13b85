#!/usr/bin/env python3
"""Calibration of the CPU baseline: the reference's own program built with its author's flags
(-Ofast -flto -march=native: host-specific, so it stays off the GPU box) against the O-strict build bench.py times
there (oracle/_ref/raytracer_ostrict, -O2), both on THIS container's cores, same scene files, each program's own
'Finished in' timer.  Writes profiles/cpu_reference_ofast_ratio.json (ratio = O-strict seconds / O-fast seconds:
the shipping build is that much faster than the baseline the bench line states).
usage: python3 tools/ofast_ratio.py [config ...]   (needs oracle/_ref, i.e. /root/reference; minutes of CPU)"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {
    "c1": ("examples/example1.xml", None), "c2": ("examples/example9_4k.xml", None), "c3": ("examples2/8_fresnel_ss4.xml", None),
    "c4": ("examples3/101_julia_shiny.xml", None), "c5": ("examples3/103_caustic_animation.xml", 0.5),
}


def run(exe, rel, still_at, threads):
    scene_path = os.path.join(ROOT, "scenes", rel)
    with tempfile.TemporaryDirectory() as tmp:
        src_dir = os.path.dirname(scene_path)
        for name in os.listdir(src_dir):
            if name != os.path.basename(scene_path):
                os.symlink(os.path.join(src_dir, name), os.path.join(tmp, name))
        xml = open(scene_path).read()
        xml = re.sub(r"<scene\b", '<scene threads="%d"' % threads, re.sub(r'\sthreads="[^"]*"', "", xml), count=1)
        if still_at is not None:
            xml = re.sub(r"(<animation\b[^>]*/>)", r'\1<still time="%r"/>' % still_at, xml, count=1)
        path = os.path.join(tmp, "scene.xml")
        open(path, "w").write(xml)
        t0 = time.perf_counter()
        out = subprocess.run([exe, path, os.path.join(tmp, "out.png")], capture_output=True, text=True, timeout=900).stdout
        wall = time.perf_counter() - t0
    # a still's own timer starts after the photon pass (src/main.cpp:18-23): photon scenes are timed by the process's
    # wall clock, as bench.py's cpu_reference does
    return wall if still_at is not None else float(re.search(r"Finished in ([0-9.eE+-]+)\s*s", out).group(1))


def main():
    threads = os.cpu_count() or 8
    res = {"threads": threads, "host": "build container (%d vCPU)" % threads,
           "strict_build": "g++ -O2 -std=c++17 -pthread", "ofast_build": "g++ -Ofast -flto -march=native -std=c++17 -pthread (reference Makefile:5)",
           "configs": {}}
    only = sys.argv[1:]
    path = os.path.join(ROOT, "profiles", "cpu_reference_ofast_ratio.json")
    if only and os.path.exists(path):
        res = json.load(open(path))
    for key, (rel, still) in CONFIGS.items():
        if only and key not in only:
            continue
        s = min(run(os.path.join(ROOT, "oracle", "_ref", "raytracer_ostrict"), rel, still, threads) for _ in range(2))
        f = min(run(os.path.join(ROOT, "oracle", "_ref", "raytracer_ofast"), rel, still, threads) for _ in range(2))
        res["configs"][key] = {"scene": rel, "strict_s": s, "ofast_s": f, "ratio": round(s / f, 3)}
        print(key, res["configs"][key], flush=True)
    with open(path, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    sys.exit(main())
