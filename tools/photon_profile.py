"""Developer probe (GPU box, under rocprofv3 --kernel-trace --stats): twenty ordered photon passes of the C5 scene."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi  # noqa: E402

ctx = capi.Context(0)
if os.environ.get("PHOTON_RECORDS"):
    ctx.set_option(capi.RT_OPT_PHOTON_RECORDS, int(os.environ["PHOTON_RECORDS"]))
sc = capi.Scene(os.path.join(ROOT, "scenes/examples3/103_caustic_animation.xml"), 0.5)
ctx.upload(sc)
ts = [ctx.photon_generate().kernel_ms for _ in range(20)]
print("photon ms", ["%.3f" % x for x in ts[1:]], "ordered:", ctx.photon_ordered())
