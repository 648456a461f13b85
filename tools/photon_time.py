"""Developer probe (GPU box): time of the photon pass, reference deposit order against float atomics."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi  # noqa: E402

ctx = capi.Context(0)
for path, t in (("scenes/examples3/103_caustic_animation.xml", 0.5), ("scenes/examples2/9_caustic.xml", 0.0)):
    sc = capi.Scene(os.path.join(ROOT, path), t)
    for ordered, cull in ((1, 1), (1, 0), (0, 1), (0, 0)):
        ctx.set_option(capi.RT_OPT_PHOTON_ORDER, ordered)
        ctx.set_option(capi.RT_OPT_CULL_PHOTON_ROOTS, cull)
        ctx.upload(sc)
        ts = [ctx.photon_generate().kernel_ms for _ in range(6)]
        print(os.path.basename(path), "ordered" if ordered else "atomics", "roots culled" if cull else "every root traced", "photon ms", ["%.3f" % x for x in ts], flush=True)
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    st = ctx.photon_generate()
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
    print("  segments", st.photon_segments, "deposits", st.photon_deposits, "roots outside the cull box", st.photon_roots_culled, flush=True)
