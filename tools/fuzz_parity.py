#!/usr/bin/env python3
"""Developer tool (GPU box): the random-scene parity sweep of tests/test_gpu_parity.py over many more seeds.
tools/fuzz_parity.py FIRST LAST [BVH_MIN_TRIANGLES [WIDTHxHEIGHT]]  -> prints the seeds whose picture or photon textures differ from the
oracle's.  BVH_MIN_TRIANGLES = 1 sends every triangle run through the BVH traversal instead of the culled scan."""
import os
import pathlib
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import capi  # noqa: E402
import mesh_scene  # noqa: E402
import oracle_lib  # noqa: E402


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    tmp = pathlib.Path(tempfile.mkdtemp())
    ctx = capi.Context(0)
    if len(sys.argv) > 3:
        ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, int(sys.argv[3]))
    bad = []
    for seed in range(first, last):
        if os.environ.get("FUZZ_TRACE"):       # (a run that faults the GPU loses its files: say on stdout where it is)
            print("begin seed", seed, flush=True)
        sc = capi.Scene(mesh_scene.write_random_scene(tmp, seed), 0.0)
        if len(sys.argv) > 4:       # larger pictures: more tiles than waves, so the dynamic part of the work distribution runs
            if sc.desc.n_julias > 0:
                continue            # (the CPU oracle needs minutes for a large Julia picture)
            w, h = (int(v) for v in sys.argv[4].split("x"))
            sc.set_camera(w, h, 0)
            print("seed", seed, flush=True)
        ctx.upload(sc)
        ctx.photon_generate()
        img, _ = ctx.render()
        ph = oracle_lib.Photons(sc) if sc.desc.photon_steps > 0 else None
        ref, _ = oracle_lib.render(sc, ph, threads=16)
        n_bad = int(np.any(img.view(np.uint32) != ref.view(np.uint32), axis=2).sum())
        tex_bad = 0
        if ph is not None:
            tex_bad = sum(int(not np.array_equal(ctx.photon_download(o).view(np.uint32), ph.texture(o).view(np.uint32))) for o in range(sc.desc.n_objects))
        if n_bad or tex_bad:
            bad.append(seed)
            print("seed %d: %d pixels, %d photon textures differ" % (seed, n_bad, tex_bad), flush=True)
        if seed % 50 == 0:
            print("... seed", seed, flush=True)
    print("checked seeds %d..%d, differing: %s" % (first, last - 1, bad), flush=True)


if __name__ == "__main__":
    main()
