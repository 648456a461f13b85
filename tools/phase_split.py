"""Developer tool (GPU box): share of trace_kernel wall clocks spent in query and in resolve phases, from a library built with
-DRT_PHASE_TIMING (tools/build_variant.sh timing -DRT_PHASE_TIMING; RT_HIP_LIB=$PWD/ab/timing.so python3 tools/phase_split.py scene.xml ...)."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from raytracer_amd import capi
ctx = capi.Context(0)
for path in sys.argv[1:]:
    sc = capi.Scene(path, 0.5 if "caustic" in path else 0.0)
    ctx.upload(sc); ctx.photon_generate()
    ctx.render()
    _, st = ctx.render()
    words = (C.c_uint64 * 96)()
    capi.hip_lib().rt_debug_counters(ctx._ctx, words, 96)
    w = list(words)
    names = ["query_phases", "query_lanes", "march_phases", "march_lanes", "normal_phases", "normal_lanes", "resolve_phases", "resolve_lanes", "de", "query_cycles", "march_cycles", "normal_cycles", "resolve_cycles"]
    d = dict(zip(names, w[18:18 + len(names)]))
    q, r = d["query_cycles"], d["resolve_cycles"]
    print(os.path.basename(path), "kernel ms %.3f" % st.kernel_ms, "query share %.3f" % (q / max(1, q + r)), "phases", d["query_phases"], "clk/query phase %.0f" % (q / max(1, d["query_phases"])), "clk/resolve phase %.0f" % (r / max(1, d["query_phases"])))
