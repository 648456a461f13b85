#!/usr/bin/env python3
"""Generates tests/golden/* from the REAL reference (build container only).

Needs oracle/_ref/ (built by `make -C oracle` from /root/reference/src, "O-strict" flags: g++ 11.4
-O2 -std=c++17 -pthread) — the reference itself never travels; these outputs do.

  <name>.png / .apng   the reference executable's own output file for the scene
  <name>.rows.npz      pre-quantisation float32 rows of RayTracer::raytrace's Picture (ref_harness)
  <name>.fp.bin        flattened-scene word stream (ref_harness fingerprint) for the loader test
  <name>.kat.bin       Object::intersect / calcFresnel known answers (ref_harness kat)
  <name>.photon.bin    sparse photon textures after PhotonMapper::generate (ref_harness render)
  x_dof_run{1,2}.png   two runs of the reference on examples2/7_dof.xml: its lens draws come from
                       random_device-seeded engines (src/raytracer.cpp:44,86-89), so the two differ; their
                       distance is the bar of the statistical depth-of-field test: `make_golden.py dof`
  ofast/<name>.png     the same scenes from the reference built with its own Makefile flags
                       (-Ofast -flto -march=native; host-dependent, the SECONDARY parity target of
                       SURVEY 8c): `make_golden.py ofast`
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
HARNESS = os.path.join(REF, "ref_harness")
REF_EXE = os.path.join(REF, "raytracer_ostrict")
REF_EXE_FAST = os.path.join(REF, "raytracer_ofast")
OFAST_CONFIGS = ["c1_example1", "c2_example9_lowres", "c3_fresnel", "c4_julia_shiny", "c5_caustic_animation"]
SCENES = os.path.join(ROOT, "scenes")

# name -> (scene file, times for float rows, number of float rows kept)
CONFIGS = {
    "c1_example1": ("examples/example1.xml", [0.0], 16),
    "c2_example9_4k": ("examples/example9_4k.xml", [0.0], 12),
    "c2_example9_lowres": ("examples/example9_lowres.xml", [0.0], 16),
    "c3_fresnel_ss4": ("examples2/8_fresnel_ss4.xml", [0.0], 16),
    "c3_fresnel": ("examples2/8_fresnel.xml", [0.0], 16),
    "c4_julia_shiny": ("examples3/101_julia_shiny.xml", [0.0], 16),
    "c5_caustic_animation": ("examples3/103_caustic_animation.xml", [0.0, 10.0 / 19.0, 1.0], 10),
    "x_transparent": ("examples2/1_transparent.xml", [0.0], 8),
    "x_julia": ("examples2/4_julia.xml", [0.0], 8),
    "x_supersampling": ("examples2/6_supersampling.xml", [0.0], 8),
}
# the 4K PNG of C2 is 8 MB: not committed (its float rows and the 512x512 sibling are)
PNG_CONFIGS = [c for c in CONFIGS if c != "c2_example9_4k"] + ["c5_caustic_animation_mb8"]
EXTRA_SCENES = {"c5_caustic_animation_mb8": "examples3/103_caustic_animation_mb8.xml"}


def run(cmd):
    t0 = time.time()
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return time.time() - t0


def main():
    only = set(sys.argv[1:])
    timings = {}
    for name, (rel, times, n_rows) in CONFIGS.items():
        if only and name not in only:
            continue
        scene = os.path.join(SCENES, rel)
        run([HARNESS, "fingerprint", scene, "0.25", os.path.join(HERE, name + ".fp.bin")])
        run([HARNESS, "kat", scene, "0", "24", "1234", os.path.join(HERE, name + ".kat.bin")])
        rows_out = {}
        for i, t in enumerate(times):
            tmp = "/tmp/golden_%s_%d.f32" % (name, i)
            photon = "/tmp/golden_%s_%d.photon" % (name, i)
            timings["%s@%g" % (name, t)] = run([HARNESS, "render", scene, repr(float(np.float32(t))), tmp, photon])
            raw = np.fromfile(tmp, dtype=np.uint32)
            w, h = int(raw[0]), int(raw[1])
            img = raw[2:].view(np.float32).reshape(h, w, 4)
            rows = np.unique(np.linspace(0, h - 1, n_rows).astype(np.int64))
            rows_out["t%d" % i] = np.float32(t)
            rows_out["rows%d" % i] = rows
            rows_out["data%d" % i] = img[rows]
            os.remove(tmp)
            if os.path.exists(photon):
                os.remove(photon)
        np.savez_compressed(os.path.join(HERE, name + ".rows.npz"), **rows_out)
        print(name, "rows done", flush=True)
    # sparse photon textures on a reduced scan (steps=200) of the caustic scene at t=0.5
    if not only or "photon" in only:
        src = open(os.path.join(SCENES, "examples3/103_caustic_animation.xml")).read()
        small = os.path.join(SCENES, "examples3/103_caustic_steps200.xml")
        open(small, "w").write(src.replace('steps="1000"', 'steps="200"'))
        run([HARNESS, "render", small, "0.5", "/tmp/golden_photon.f32", os.path.join(HERE, "c5_caustic_steps200.photon.bin")])
        raw = np.fromfile("/tmp/golden_photon.f32", dtype=np.uint32)
        w, h = int(raw[0]), int(raw[1])
        img = raw[2:].view(np.float32).reshape(h, w, 4)
        rows = np.unique(np.linspace(0, h - 1, 10).astype(np.int64))
        np.savez_compressed(os.path.join(HERE, "c5_caustic_steps200.rows.npz"), t0=np.float32(0.5), rows0=rows, data0=img[rows])
    # the reference executable's own output files
    for name in PNG_CONFIGS:
        if only and name not in only:
            continue
        rel = CONFIGS[name][0] if name in CONFIGS else EXTRA_SCENES[name]
        out = os.path.join(HERE, name + ".png")
        timings[name + ".png"] = run([REF_EXE, os.path.join(SCENES, rel), out])
        print(name, "png done %.1fs" % timings[name + ".png"], flush=True)
    if "dof" in only:
        for i in (1, 2):
            out = os.path.join(HERE, "x_dof_run%d.png" % i)
            timings["x_dof_run%d.png" % i] = run([REF_EXE, os.path.join(SCENES, "examples2/7_dof.xml"), out])
            print("dof run", i, "done %.1fs" % timings["x_dof_run%d.png" % i], flush=True)
    if "ofast" in only:
        os.makedirs(os.path.join(HERE, "ofast"), exist_ok=True)
        for name in OFAST_CONFIGS:
            out = os.path.join(HERE, "ofast", name + ".png")
            timings["ofast/" + name + ".png"] = run([REF_EXE_FAST, os.path.join(SCENES, CONFIGS[name][0]), out])
            print(name, "ofast png done %.1fs" % timings["ofast/" + name + ".png"], flush=True)
    meta_path = os.path.join(HERE, "MANIFEST.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    if "ofast" in only:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        meta["reference_build_ofast"] = ("g++ (Ubuntu 11.4.0-1ubuntu1~22.04.2) 11.4.0, -Ofast -flto -march=native -std=c++17 -pthread "
                                         "(the reference Makefile's flags) on " + (cpu[0] if cpu else "unknown CPU"))
    meta.update({
        "reference_build": "g++ (Ubuntu 11.4.0-1ubuntu1~22.04.2) 11.4.0, -O2 -std=c++17 -pthread (O-strict), glibc 2.35",
        "hue_table": "int-abs (printed by `ref_harness hues`)",
    })
    meta.setdefault("reference_seconds_8vcpu", {}).update({k: round(v, 2) for k, v in timings.items()})
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
