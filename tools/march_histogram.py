"""Developer tool (CPU): Julia marches of a scene by length, from the oracle's diagnostics (oracle_march_histogram,
oracle_set_cost_map).  A march is a chain of dependent steps: the longest ones bound a frame's latency whatever the
machine's throughput.   usage: tools/march_histogram.py scene.xml [width height]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import capi
import oracle_lib
path = sys.argv[1]
sc = capi.Scene(path, 0.0)
if len(sys.argv) > 3:
    sc.set_camera(int(sys.argv[2]), int(sys.argv[3]), 0)
lib = oracle_lib.lib() if hasattr(oracle_lib, "lib") else oracle_lib._lib
cost = np.zeros((sc.height, sc.width), np.float32)
lib.oracle_set_cost_map.argtypes = [C.c_void_p]
lib.oracle_set_cost_map(cost.ctypes.data)
img, st = oracle_lib.render(sc, None, threads=8)
lib.oracle_set_cost_map(None)
counts = (C.c_uint64 * 16)(); steps = (C.c_uint64 * 16)(); longest = C.c_uint64()
lib.oracle_march_histogram(counts, steps, C.byref(longest))
tot = sum(steps)
print("marches by steps (2^(k-1), 2^k]: count, share of all steps")
for k in range(16):
    if counts[k]:
        print("  <= %5d: %9d marches, %5.1f%% of steps" % (1 << k, counts[k], 100.0 * steps[k] / tot))
print("most estimator iterations in one march:", longest.value)
ex = (C.c_uint64 * 6)(); lib.oracle_sample_extremes(ex)
print("sample with the most steps: %d steps, %d iterations, %d Julia queries; with the most iterations: %d steps, %d iterations, %d queries" % tuple(ex))
flat = np.sort(cost.ravel())[::-1]
print("pixels by march steps: max %d, top 10 %s, mean %.0f" % (flat[0], flat[:10].astype(int).tolist(), flat.mean()))
ys, xs = np.unravel_index(np.argsort(cost.ravel())[::-1][:10], cost.shape)
print("costliest pixels (x, y):", list(zip(xs.tolist(), ys.tolist())))
