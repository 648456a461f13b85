import os, sys, torch, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from raytracer_amd import capi
ROOT = os.environ["GRAFT_REPO_ROOT"]
sc = capi.Scene(os.path.join(ROOT, "scenes/examples3/101_julia_shiny.xml"), 0.0)
ctx = capi.Context(0); ctx.upload(sc)
bands = capi.Bands(8, 0, 2)
n = ctx.pixel_count(bands)
dev = torch.device("cuda", 0)
a = torch.empty((n, 4), dtype=torch.float32, device=dev); b = torch.empty_like(a)
def t(fn, out):
    ms = []
    for i in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(out.data_ptr(), None, bands); e1.record(); torch.cuda.synchronize(); ctx.launch_status()
        ms.append(e0.elapsed_time(e1))
    return ms[1:]
print("rt_render_device       ", ["%.2f" % x for x in t(ctx.render_device, a)])
print("rt_render_device_noalloc", ["%.2f" % x for x in t(ctx.render_device_noalloc, b)])
print("same bits:", bool(torch.equal(a.view(torch.int32), b.view(torch.int32))))
