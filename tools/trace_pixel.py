#!/usr/bin/env python3
"""Developer probe (GPU box): tools/trace_pixel.py SCENE.xml X Y [X Y ...]
For every castRay the ORACLE makes for the pixel (oracle_pixel_rays: ray, winner, distance), asks the HIP intersectors
(rt_kat_intersect, the functions the render kernels are built from) for the hit of every object with that ray and an infinite
maximum, replays the scan's acceptance rule on those answers, and prints the rays whose winner or distance differs — i.e.
whether an intersector itself disagrees with the oracle (an intersector that agrees here but a picture that differs points
at the render kernel's own logic: culling, the winner rule, the Julia pool)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import capi  # noqa: E402
import oracle_lib  # noqa: E402

sc = capi.Scene(sys.argv[1], 0.0)
pixels = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(2, len(sys.argv) - 1, 2)]
ctx = capi.Context(0)
ctx.upload(sc)
ph = oracle_lib.Photons(sc) if sc.desc.photon_steps > 0 else None
lib = oracle_lib.lib()
lib.oracle_pixel_rays.restype = C.c_uint32
lib.oracle_pixel_rays.argtypes = [C.POINTER(capi.SceneDesc), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
n_obj = sc.desc.n_objects
for (x, y) in pixels:
    cap = 1 << 16
    log = np.zeros((cap, 12), dtype=np.float32)
    n = lib.oracle_pixel_rays(C.byref(sc.desc), ph._h if ph is not None else None, x, y, log.ctypes.data, cap)
    log = log[:n]
    print("pixel (%d, %d): %d rays in the oracle" % (x, y, n), flush=True)
    objs = np.tile(np.arange(n_obj, dtype=np.uint32), n)
    rays = np.zeros((n * n_obj, 7), dtype=np.float32)
    rays[:, 0:6] = np.repeat(log[:, 2:8], n_obj, axis=0)
    rays[:, 6] = np.inf
    out = np.zeros((n * n_obj, 12), dtype=np.float32)
    rc = capi.hip_lib().rt_kat_intersect(ctx._ctx, objs.ctypes.data_as(C.POINTER(C.c_uint32)), rays.ctypes.data_as(C.POINTER(C.c_float)),
                                         n * n_obj, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0
    out = out.reshape(n, n_obj, 12)
    bad = 0
    for i in range(n):
        d = log[i, 5:8]
        best, best_k = np.float32(np.inf), -1
        for k in range(n_obj):
            if out[i, k, 0] == 0 or not out[i, k, 1] <= best:
                continue
            m = sc.desc.materials[sc.desc.objects[k].material]
            cos_rn = np.clip(np.float32(np.dot(d, out[i, k, 5:8])), -1, 1)     # (not the oracle's rounding of the dot product: a sign is all that is used)
            if cos_rn >= 0 and (m.transmittance == 0 or m.refraction_re ** 2 + m.refraction_im ** 2 == 0):
                continue
            best, best_k = out[i, k, 1], k
        o_found, o_obj, o_dist = bool(log[i, 8]), int(log[i, 9]), log[i, 10]
        if (best_k >= 0) != o_found or (o_found and (best_k != o_obj or best != o_dist)):
            bad += 1
            print("  ray %d (depth %d, wavelength %.3f): oracle %s, HIP intersectors %s" % (
                i, int(log[i, 0]), log[i, 1],
                ("object %d at %s" % (o_obj, float.hex(float(o_dist)))) if o_found else "miss",
                ("object %d at %s" % (best_k, float.hex(float(best)))) if best_k >= 0 else "miss"), flush=True)
            print("    origin %s direction %s" % ([float.hex(float(v)) for v in log[i, 2:5]], [float.hex(float(v)) for v in log[i, 5:8]]))
    print("  rays whose winner differs between the oracle and the HIP intersectors: %d" % bad, flush=True)

# detail for rays given as "--ray PIXELINDEX RAYINDEX" is not needed: every object whose answers differ, for the last pixel's rays
def detail(log_row, ks, maxes):
    o = np.ascontiguousarray(log_row[2:5]); d = np.ascontiguousarray(log_row[5:8])
    for k in ks:
        for mx in maxes:
            hit_o, rec_o = oracle_lib.intersect(sc, k, o, d, mx)
            r = np.zeros((1, 7), dtype=np.float32); r[0, 0:3] = o; r[0, 3:6] = d; r[0, 6] = mx
            out1 = np.zeros((1, 12), dtype=np.float32)
            ob = np.array([k], dtype=np.uint32)
            capi.hip_lib().rt_kat_intersect(ctx._ctx, ob.ctypes.data_as(C.POINTER(C.c_uint32)), r.ctypes.data_as(C.POINTER(C.c_float)), 1, out1.ctypes.data_as(C.POINTER(C.c_float)))
            print("    object %d (kind %d) max %s: oracle %s %s | hip %s %s" % (k, sc.desc.objects[k].kind, float.hex(float(mx)), hit_o, float.hex(float(rec_o[0])) if hit_o else "-",
                  bool(out1[0, 0]), float.hex(float(out1[0, 1])) if out1[0, 0] else "-"))

if os.environ.get("TRACE_DETAIL"):
    idx, ks = os.environ["TRACE_DETAIL"].split(":")
    row = log[int(idx)]
    ks = [int(v) for v in ks.split(",")]
    print("  detail of ray", idx)
    detail(row, ks, [np.float32(np.inf), np.float32(row[10])])
