#!/usr/bin/env python3
"""Developer probe (GPU box): tools/diff_pixels.py SCENE.xml [reps]  -> renders the scene `reps` times on the HIP path and once
on the oracle; prints every pixel whose float words differ (position, both values) and whether the HIP renders agree with
each other."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import capi  # noqa: E402
import oracle_lib  # noqa: E402

sc = capi.Scene(sys.argv[1], 0.0)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = capi.Context(0)
ctx.upload(sc)
ctx.photon_generate()
imgs = [ctx.render()[0] for _ in range(reps)]
for i in range(1, reps):
    print("render %d equals render 0:" % i, bool(np.array_equal(imgs[i].view(np.uint32), imgs[0].view(np.uint32))))
ph = oracle_lib.Photons(sc) if sc.desc.photon_steps > 0 else None
ref, _ = oracle_lib.render(sc, ph, threads=16)
bad = np.argwhere(np.any(imgs[0].view(np.uint32) != ref.view(np.uint32), axis=2))
print("differing pixels:", len(bad))
for y, x in bad[:20]:
    print("  (x %d, y %d) hip %s  oracle %s" % (x, y, [float.hex(float(v)) for v in imgs[0][y, x]], [float.hex(float(v)) for v in ref[y, x]]))
if ph is not None:
    for o in range(sc.desc.n_objects):
        if not np.array_equal(ctx.photon_download(o).view(np.uint32), ph.texture(o).view(np.uint32)):
            print("photon texture of object %d differs" % o)
