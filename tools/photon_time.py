import sys, os, time
sys.path.insert(0, os.getcwd())
from raytracer_amd import capi
ctx = capi.Context(0)
for path, t in (("scenes/examples3/103_caustic_animation.xml", 0.5), ("scenes/examples2/9_caustic.xml", 0.0)):
    sc = capi.Scene(path, t); ctx.upload(sc)
    ts = []
    for i in range(6):
        ts.append(ctx.photon_generate().kernel_ms)
    print(os.path.basename(path), "photon ms", ["%.3f" % x for x in ts])
