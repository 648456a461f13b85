#!/usr/bin/env python3
"""Developer tool (GPU box): tools/fuzz_consistency.py FIRST LAST [WxH]  -> on the seeded random scenes of tests/mesh_scene.py, the
HIP path against ITSELF through its other doors (no oracle, so thousands of scenes per minute): the whole picture against
the picture assembled from rt_bands{8, r, 3} for r = 0..2, against rt_render_device_noalloc, against a second render,
against rt_render_rgba8_device (Color::scaleOut of it), against a render under RT_OPT_SHARED_GPU, and the photon textures
of two passes.  Everything must be the same bits."""
import os
import pathlib
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import capi, driver  # noqa: E402
import mesh_scene  # noqa: E402


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    tmp = pathlib.Path(tempfile.mkdtemp())
    ctx = capi.Context(0)
    dev = torch.device("cuda", 0)
    bad = []
    for seed in range(first, last):
        if os.environ.get("FUZZ_TRACE"):
            print("begin seed", seed, flush=True)
        sc = capi.Scene(mesh_scene.write_random_scene(tmp, seed), 0.0)
        if len(sys.argv) > 3:       # larger pictures (WxH): more tiles than two per wave, so owned tiles and the counter both run
            w_, h_ = (int(v) for v in sys.argv[3].split("x"))
            sc.set_camera(w_, h_, 0)
        ctx.upload(sc)
        ctx.photon_generate()
        tex = [ctx.photon_download(o).copy() for o in range(sc.desc.n_objects)] if sc.desc.photon_steps > 0 else []
        img, _ = ctx.render()
        why = []
        again, _ = ctx.render()
        if not np.array_equal(img.view(np.uint32), again.view(np.uint32)):
            why.append("second render")
        h, w = sc.height, sc.width
        whole = np.zeros_like(img)
        for r in range(3):
            bands = capi.Bands(driver.BAND_ROWS, r, 3)
            part, _ = ctx.render(bands)
            rows = driver.band_rows_of(h, r, 3)
            whole[rows] = part.reshape(len(rows), w, 4)
        if not np.array_equal(img.view(np.uint32), whole.view(np.uint32)):
            why.append("bands")
        out = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
        try:
            ctx.render_device_noalloc(out.data_ptr())
            torch.cuda.synchronize()
            ctx.launch_status()
            if not np.array_equal(img.view(np.uint32), out.cpu().numpy().view(np.uint32)):
                why.append("noalloc")
        except capi.RtError as e:
            if "exceed" not in str(e) and "fit" not in str(e):      # (a frame beyond the fixed workspace is a documented limit)
                why.append("noalloc: %s" % e)
        ctx.set_option(capi.RT_OPT_SHARED_GPU, 1)
        try:
            shared, _ = ctx.render()
        finally:
            ctx.set_option(capi.RT_OPT_SHARED_GPU, 0)
        if not np.array_equal(img.view(np.uint32), shared.view(np.uint32)):
            why.append("RT_OPT_SHARED_GPU")
        out8 = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
        ctx.render_rgba8_device(out8.data_ptr())
        torch.cuda.synchronize()
        ctx.launch_status()
        if not np.array_equal(out8.cpu().numpy().reshape(-1, 4), capi.quantize(img.reshape(-1, 4))):
            why.append("rgba8")
        if tex:
            ctx.photon_generate()
            for o in range(sc.desc.n_objects):
                if not np.array_equal(ctx.photon_download(o).view(np.uint32), tex[o].view(np.uint32)):
                    why.append("photon texture %d of a second pass" % o)
                    break
        if why:
            bad.append(seed)
            print("seed %d: %s" % (seed, ", ".join(why)), flush=True)
        if seed % 100 == 0:
            print("... seed", seed, flush=True)
    print("checked seeds %d..%d, inconsistent: %s" % (first, last - 1, bad), flush=True)


if __name__ == "__main__":
    main()
