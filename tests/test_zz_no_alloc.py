"""rt_render_device_noalloc (include/rt_hip.h): the render entry point that allocates nothing.

Called through the C ABI and compared bit for bit with rt_render_device — the existing entry point it
is a variant of — on two calls in a row, the second with a smaller input (the fixed workspace keeps
the first call's contents), for a supersampled mesh scene (sample buffer + frame stacks in the
workspace) and for a Julia scene (the other render kernel).  Collected last (test_zz_*): the round's
GPU tier runs with -x.
"""
import numpy as np
import pytest
import torch

from conftest import scene_path
from raytracer_amd import capi

pytestmark = pytest.mark.gpu


def both_ways(ctx, scene):
    """The same frame through rt_render_device and through rt_render_device_noalloc (device tensors)."""
    ctx.upload(scene)
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
    ctx.photon_generate()
    n = scene.width * scene.height
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.full((n, 4), -1.0, dtype=torch.float32, device="cuda:0")
    b = torch.full((n, 4), -2.0, dtype=torch.float32, device="cuda:0")
    ctx.render_device(a.data_ptr(), stream)
    ctx.launch_status()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    ctx.render_device_noalloc(b.data_ptr(), stream)
    ctx.launch_status()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before      # no device memory was taken or returned
    return a.cpu().numpy(), b.cpu().numpy()


@pytest.mark.parametrize("rel,first,second", [
    ("examples/example9_lowres.xml", (256, 192, 2), (96, 64, 1)),     # 4 samples per pixel, then 1: sample buffer, then none
    ("examples2/8_fresnel.xml", (128, 96, 1), (40, 24, 1)),           # dispersion: 8 samples per pixel, deep ray trees
    ("examples2/4_julia.xml", (160, 120, 1), (64, 48, 2)),            # the Julia kernel
])
def test_noalloc_equals_render_device_twice_in_a_row(rel, first, second):
    ctx = capi.Context(0)
    try:
        for w, h, ss in (first, second):
            sc = capi.Scene(scene_path(rel), 0.0)
            sc.set_camera(w, h, ss)
            want, got = both_ways(ctx, sc)
            assert want.min() >= 0.0 and np.isfinite(want).all()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rel, w, h, ss)
    finally:
        ctx.close()


def test_noalloc_with_bands_and_a_photon_scene():
    ctx = capi.Context(0)
    try:
        sc = capi.Scene(scene_path("examples3/103_caustic_steps200.xml"), 0.25)
        sc.set_camera(120, 90, 0)
        ctx.upload(sc)
        ctx.photon_generate()
        bands = capi.Bands(8, 1, 3)
        n = ctx.pixel_count(bands)
        stream = torch.cuda.current_stream().cuda_stream
        a = torch.zeros((n, 4), dtype=torch.float32, device="cuda:0")
        b = torch.zeros((n, 4), dtype=torch.float32, device="cuda:0")
        ctx.render_device(a.data_ptr(), stream, bands)
        ctx.render_device_noalloc(b.data_ptr(), stream, bands)
        torch.cuda.synchronize()
        assert n > 0 and torch.equal(a.view(torch.int32), b.view(torch.int32)) and float(a.max()) > 0
    finally:
        ctx.close()


def test_noalloc_rejects_what_does_not_fit():
    ctx = capi.Context(0)
    try:
        # configs[2]: 512x512 x 128 samples per pixel = 537 MB of samples, beyond the fixed 32 MiB workspace
        sc = capi.Scene(scene_path("examples2/8_fresnel_ss4.xml"), 0.0)
        ctx.upload(sc)
        out = torch.zeros((512 * 512, 4), dtype=torch.float32, device="cuda:0")
        lib = capi.hip_lib()
        rc = lib.rt_render_device_noalloc(ctx._ctx, None, out.data_ptr(), None)
        assert rc == -5      # RT_ERR_LIMIT
        msg = lib.rt_last_error(ctx._ctx).decode()
        assert "sample buffer" in msg and "exceeds" in msg
        assert lib.rt_render_device_noalloc(ctx._ctx, None, None, None) == -1
        # the allocating entry point still renders the scene
        ctx.render_device(out.data_ptr(), None)
        torch.cuda.synchronize()
        assert float(out.max()) > 0
    finally:
        ctx.close()


def test_an_abandoned_launch_is_not_lost_under_later_launches():
    """The watchdog word is sticky in both workspaces until rt_launch_status has seen it: an abandoned Julia launch
    (RT_OPT_DEBUG_FORCE_ABORT) followed, unchecked, by further launches of either entry point is still reported."""
    ctx = capi.Context(0)
    try:
        sc = capi.Scene(scene_path("examples2/4_julia.xml"), 0.0)
        sc.set_camera(64, 48, 1)
        ctx.upload(sc)
        out = torch.zeros((64 * 48, 4), dtype=torch.float32, device="cuda:0")
        for first, second in (("noalloc", "noalloc"), ("device", "noalloc"), ("noalloc", "device")):
            launch = {"noalloc": ctx.render_device_noalloc, "device": ctx.render_device}
            ctx.set_option(capi.RT_OPT_DEBUG_FORCE_ABORT, 1)
            launch[first](out.data_ptr(), None)
            ctx.set_option(capi.RT_OPT_DEBUG_FORCE_ABORT, 0)
            launch[second](out.data_ptr(), None)          # a healthy launch on top, its status unchecked so far
            with pytest.raises(capi.RtError, match="abandoned"):
                ctx.launch_status()
            ctx.launch_status()                             # seen once: cleared
            launch[second](out.data_ptr(), None)
            ctx.launch_status()
            torch.cuda.synchronize()
            assert float(out.max()) > 0
    finally:
        ctx.close()
