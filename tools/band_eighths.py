"""Developer probe (GPU box): kernel time of each eighth of C4's interleaved 8-row bands on one GPU (what each of eight ranks
would render) and of the whole frame, for the library in RT_HIP_LIB (default: the in-tree build)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi  # noqa: E402

ctx = capi.Context(0)
sc = capi.Scene(os.path.join(ROOT, "scenes/examples3/101_julia_shiny.xml"), 0.0)
ctx.upload(sc)
whole = [ctx.render()[1].kernel_ms for _ in range(4)][1:]
eighths = []
for r in range(8):
    bands = capi.Bands(8, r, 8)
    eighths.append(float(np.mean([ctx.render(bands)[1].kernel_ms for _ in range(3)][1:])))
print(os.environ.get("RT_HIP_LIB", "in-tree"), "whole frame %.2f ms; eighths %s; slowest %.2f -> ceiling of the 8-GPU band split %.2fx" % (
    float(np.mean(whole)), ["%.1f" % e for e in eighths], max(eighths), float(np.mean(whole)) / max(eighths)))
