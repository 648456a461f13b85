"""GPU parity tests proper: every call goes through the C ABI (librt_hip.so) on cuda:0.

Bars (BASELINE.json north_star): against the reference's own PNG output at most 1 LSB per 8-bit
channel on >= 99.9 % of pixels and PSNR >= 50 dB.  The kernels mirror the O-strict oracle operation
for operation, so against the oracle the tests also assert the much tighter bound they actually
reach (bit-exact float32, photon textures included: deposits are summed in the reference's order).
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib
from conftest import golden_path, scene_path
from raytracer_amd import capi

pytestmark = pytest.mark.gpu

MAX_FRAC_GT1 = 0.001   # <= 0.1 % of pixels may be off by more than 1 LSB
MIN_PSNR = 50.0


def assert_north_star(img8, ref8, what):
    d = oracle_lib.image_diff(img8, ref8)
    assert d["frac_gt1"] <= MAX_FRAC_GT1 and d["psnr"] >= MIN_PSNR, (what, d)
    return d


def hip_render(ctx, scene, stats=False, bands=None):
    ctx.upload(scene)
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1 if stats else 0)
    ctx.photon_generate()
    return ctx.render(bands)


# ---- known answers: one function per call ------------------------------------------------------------

def kat_split(path):
    w = np.fromfile(path, dtype=np.uint32)
    cands = [i for i in np.nonzero(w == 0xFFFFFFFF)[0] if i % 20 == 0 and i + 2 + 4 * int(w[i + 1]) == len(w)]
    m = cands[0]
    return w[:m].reshape(-1, 20), w[m + 2:].reshape(-1, 4)


@pytest.mark.parametrize("name,rel", [("c2_example9_lowres", "examples/example9_lowres.xml"),
                                      ("c4_julia_shiny", "examples3/101_julia_shiny.xml"),
                                      ("c3_fresnel", "examples2/8_fresnel.xml"),
                                      ("x_transparent", "examples2/1_transparent.xml")])
def test_intersectors_against_reference_answers(hip_ctx, name, rel):
    sc = capi.Scene(scene_path(rel), 0.0)
    hip_ctx.upload(sc)
    recs, fres = kat_split(golden_path(name + ".kat.bin"))
    n = len(recs)
    objs = np.ascontiguousarray(recs[:, 0])
    rays = np.ascontiguousarray(recs[:, 1:8]).view(np.float32)
    out = np.zeros((n, 12), dtype=np.float32)
    rc = capi.hip_lib().rt_kat_intersect(hip_ctx._ctx, objs.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         rays.ctypes.data_as(C.POINTER(C.c_float)), n, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0
    assert np.array_equal(out[:, 0] != 0, recs[:, 8] != 0)
    hit = recs[:, 8] != 0
    kinds = np.array([sc.desc.objects[int(o)].kind for o in objs])
    got = out[hit][:, 1:12].view(np.uint32)
    want = recs[hit][:, 9:20]
    # every kind bit-exact, the sphere's uv included: asinf/atan2f are glibc's own operation sequences
    # on the device (raytracer_amd/csrc/rt_libm.h, pinned exhaustively by tests/test_libm_model.py)
    assert np.array_equal(got, want), (kinds[hit][np.any(got != want, axis=1)][:8])
    if len(fres):
        rec5 = np.zeros((len(fres), 5), dtype=np.float32)
        for i, r in enumerate(fres):
            m = sc.desc.materials[sc.desc.objects[int(r[0])].material]
            f = r.view(np.float32)
            rec5[i] = (m.refraction_re, m.refraction_im, m.dispersion, f[1], f[2])
        kr = np.zeros(len(fres), dtype=np.float32)
        assert capi.hip_lib().rt_kat_fresnel(hip_ctx._ctx, rec5.ctypes.data_as(C.POINTER(C.c_float)), len(fres),
                                             kr.ctypes.data_as(C.POINTER(C.c_float))) == 0
        assert np.array_equal(kr.view(np.uint32), fres[:, 3])


def test_fresnel_of_real_indices_against_the_oracle(hip_ctx):
    # fresnel_kr's float route for real refraction indices (rt_intersect.h) against std::complex<float> as the oracle's
    # compiler and libraries evaluate it (src/raytracer.cpp:243-268): indices from 0.001 to 1000 and just outside,
    # both signs of a zero extinction coefficient, every cosine incl. 0, +-1 and the total-reflection boundary
    rng = np.random.default_rng(77)
    n = 60000
    rec5 = np.zeros((n, 5), dtype=np.float32)
    rec5[:, 0] = np.exp(rng.uniform(np.log(5e-4), np.log(2e3), size=n))
    rec5[: n // 2, 0] = rng.uniform(1.0, 2.5, size=n // 2)
    rec5[:, 1] = np.where(rng.uniform(size=n) < 0.5, 0.0, -0.0)
    rec5[:, 2] = np.where(rng.uniform(size=n) < 0.5, 0.0, rng.uniform(-0.3, 0.3, size=n))
    rec5[:, 3] = rng.uniform(-1, 1, size=n)
    rec5[:200, 3] = np.repeat(np.float32([0.0, -0.0, 1.0, -1.0]), 50)
    crit = np.sqrt(np.maximum(0.0, 1.0 - 1.0 / rec5[200:1200, 0].astype(np.float64) ** 2))     # sin_t = 1 from inside
    rec5[200:1200, 3] = np.nextafter(crit.astype(np.float32), rng.choice([0.0, 1.0], size=1000).astype(np.float32))
    rec5[:, 4] = rng.uniform(-1, 1, size=n)
    kr = np.zeros(n, dtype=np.float32)
    assert capi.hip_lib().rt_kat_fresnel(hip_ctx._ctx, rec5.ctypes.data_as(C.POINTER(C.c_float)), n, kr.ctypes.data_as(C.POINTER(C.c_float))) == 0
    m = capi.Material()
    want = np.zeros(n, dtype=np.float32)
    for i in range(n):
        m.refraction_re, m.refraction_im, m.dispersion = float(rec5[i, 0]), float(rec5[i, 1]), float(rec5[i, 2])
        want[i] = oracle_lib.fresnel(m, float(rec5[i, 3]), float(rec5[i, 4]))
    bad = np.nonzero(kr.view(np.uint32) != want.view(np.uint32))[0]
    assert len(bad) == 0, (len(bad), rec5[bad[:5]], kr[bad[:5]], want[bad[:5]])
    assert 0.2 < np.mean(kr == 1.0) < 0.8 or np.mean(kr < 1.0) > 0.3


def test_intersectors_on_seeded_rays_against_oracle(hip_ctx):
    # more rays than the committed fixtures hold, incl. degenerate ones (a == 0 in Möller–Trumbore:
    # axis-parallel rays against axis-aligned walls)
    sc = capi.Scene(scene_path("examples/example9_lowres.xml"), 0.0)
    hip_ctx.upload(sc)
    rng = np.random.default_rng(2024)
    n = 20000
    objs = rng.integers(0, sc.desc.n_objects, size=n).astype(np.uint32)
    rays = np.zeros((n, 7), dtype=np.float32)
    rays[:, 0:3] = rng.uniform(-3, 3, size=(n, 3))
    rays[:, 2] += 1.0
    rays[:, 3:6] = rng.uniform(-1, 1, size=(n, 3)) * [1, 1, 0.3] + [0, 0, -1]
    rays[:, 6] = np.where(rng.uniform(size=n) < 0.5, np.float32(capi.FLT_MAX), rng.uniform(0, 30, size=n)).astype(np.float32)
    rays[:200, 3:6] = [0, 0, -1]      # parallel to the floor plane
    rays[200:400, 3:6] = [0, -1, 0]   # parallel to the wall
    out = np.zeros((n, 12), dtype=np.float32)
    assert capi.hip_lib().rt_kat_intersect(hip_ctx._ctx, objs.ctypes.data_as(C.POINTER(C.c_uint32)),
                                           rays.ctypes.data_as(C.POINTER(C.c_float)), n,
                                           out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    hits = 0
    for i in range(n):
        hit, want = oracle_lib.intersect(sc, int(objs[i]), rays[i, 0:3], rays[i, 3:6], float(rays[i, 6]))
        assert hit == (out[i, 0] != 0), i
        if hit:
            hits += 1
            # sphere uv included: rt_libm.h restates glibc's asinf/atan2f, so no tolerance is left
            assert np.array_equal(out[i, 1:12].view(np.uint32), want.view(np.uint32)), i
    assert hits > 300


# ---- whole frames ------------------------------------------------------------------------------------

FRAMES = [
    # name, scene, t, golden png (frame), reduced size for the oracle comparison
    ("c1_example1", "examples/example1.xml", 0.0, ("c1_example1.png", 0), None),
    ("c2_example9_lowres", "examples/example9_lowres.xml", 0.0, ("c2_example9_lowres.png", 0), None),
    ("c3_fresnel", "examples2/8_fresnel.xml", 0.0, ("c3_fresnel.png", 0), None),
    ("x_transparent", "examples2/1_transparent.xml", 0.0, ("x_transparent.png", 0), None),
    ("x_supersampling", "examples2/6_supersampling.xml", 0.0, ("x_supersampling.png", 0), None),
]


@pytest.mark.parametrize("name,rel,t,png,_", FRAMES)
def test_frame_bit_exact_vs_oracle_and_reference_png(hip_ctx, name, rel, t, png, _):
    sc = capi.Scene(scene_path(rel), t)
    img, st = hip_render(hip_ctx, sc, stats=True)
    ref, ost = oracle_lib.render(sc, None, threads=16)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    # the work counters are the numerator of the roofline: they must be the oracle's
    for k in ("primary_rays", "secondary_rays", "shadow_rays", "sphere_tests", "triangle_tests", "shaded_hits",
              "fresnel_evals", "texture_lookups"):
        if k == "triangle_tests" and st.bvh_node_visits > 0:
            assert st.triangle_tests < ost.triangle_tests     # a BVH run (>= 48 triangles) skips most of them
            continue
        assert getattr(st, k) == getattr(ost, k), k
    gold, _n = capi.png_read(golden_path(png[0]), png[1])
    d = assert_north_star(capi.quantize(img), gold, name)
    assert d["max"] == 0


def test_c2_example9_4k_full_size(hip_ctx):
    """configs[1] at its full 3840x2160 (the bench workload): bit-exact against the reference's golden
    float rows and against the oracle over the whole frame.  (The reference's 4K PNG is 8 MB and is
    not committed; its 512x512 sibling is compared in test_frame_bit_exact_vs_oracle_and_reference_png.)"""
    sc = capi.Scene(scene_path("examples/example9_4k.xml"), 0.0)
    img, _ = hip_render(hip_ctx, sc)
    z = np.load(golden_path("c2_example9_4k.rows.npz"))
    assert np.array_equal(img[z["rows0"]].view(np.uint32), z["data0"].view(np.uint32))
    ref, _ = oracle_lib.render(sc, None, threads=16)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_c3_fresnel_ss4_full_size(hip_ctx):
    """configs[2]: 4x4 supersampling x 8 hues = 128 samples per pixel, summed in reference order."""
    sc = capi.Scene(scene_path("examples2/8_fresnel_ss4.xml"), 0.0)
    img, _ = hip_render(hip_ctx, sc)
    z = np.load(golden_path("c3_fresnel_ss4.rows.npz"))
    assert np.array_equal(img[z["rows0"]].view(np.uint32), z["data0"].view(np.uint32))
    gold, _n = capi.png_read(golden_path("c3_fresnel_ss4.png"))
    d = assert_north_star(capi.quantize(img), gold, "c3 ss4")
    assert d["max"] == 0


def test_c4_julia_full_size(hip_ctx):
    """configs[3]: Julia march through the wave-level phase scheduler."""
    sc = capi.Scene(scene_path("examples3/101_julia_shiny.xml"), 0.0)
    img, st = hip_render(hip_ctx, sc, stats=True)
    z = np.load(golden_path("c4_julia_shiny.rows.npz"))
    assert np.array_equal(img[z["rows0"]].view(np.uint32), z["data0"].view(np.uint32))
    gold, _n = capi.png_read(golden_path("c4_julia_shiny.png"))
    d = assert_north_star(capi.quantize(img), gold, "c4 julia")
    assert d["max"] == 0
    assert st.julia_queries > 0 and st.julia_steps > st.julia_queries


def test_abandoned_julia_launch_is_an_error_on_every_path(hip_ctx):
    """The Julia kernel's watchdog (a wave stuck on the job queues gives the launch up) must reach the caller as an
    error, not as a partly rendered picture: rt_render with stats == NULL (what the CLI shim calls), and
    rt_launch_status after rt_render_device (what the multi-GPU driver and bench.py call)."""
    import torch
    sc = capi.Scene(scene_path("examples2/4_julia.xml"), 0.0)
    sc.set_camera(96, 64, 0)
    hip_ctx.upload(sc)
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
    lib = capi.hip_lib()
    out = np.zeros((64 * 96, 4), dtype=np.float32)
    try:
        hip_ctx.set_option(capi.RT_OPT_DEBUG_FORCE_ABORT, 1)
        rc = lib.rt_render(hip_ctx._ctx, None, out.ctypes.data, None)
        assert rc == -3 and b"abandoned" in lib.rt_last_error(hip_ctx._ctx)
        dev = torch.empty((64 * 96, 4), dtype=torch.float32, device="cuda:0")
        hip_ctx.render_device(dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
        with pytest.raises(capi.RtError, match="abandoned"):
            hip_ctx.launch_status()
    finally:
        hip_ctx.set_option(capi.RT_OPT_DEBUG_FORCE_ABORT, 0)
    # the flag does not outlive its report: the next launch is clean and complete
    img, _ = hip_ctx.render()
    ref, _ = oracle_lib.render(sc, None, threads=16)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_julia_small_vs_oracle(hip_ctx):
    sc = capi.Scene(scene_path("examples2/4_julia.xml"), 0.0)
    sc.set_camera(160, 120, 0)
    img, _ = hip_render(hip_ctx, sc)
    ref, _ = oracle_lib.render(sc, None, threads=16)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("frame", [0, 10, 19])
def test_c5_caustic_frames(hip_ctx, frame):
    """configs[4] frames: photon pass + render, bit-exact against the reference's float rows and 0 LSB from
    its APNG (the photon deposits are summed in the reference's sequential order)."""
    t = float(np.float32(frame) / np.float32(19))
    sc = capi.Scene(scene_path("examples3/103_caustic_animation.xml"), t)
    img, _ = hip_render(hip_ctx, sc)
    gold, n = capi.png_read(golden_path("c5_caustic_animation.png"), frame)
    assert n == 20
    d = assert_north_star(capi.quantize(img), gold, "c5 frame %d" % frame)
    assert d["max"] == 0
    # and the float rows the reference itself produced (photon deposits are summed in its order)
    z = np.load(golden_path("c5_caustic_animation.rows.npz"))
    i = {0: 0, 10: 1, 19: 2}[frame]
    assert float(z["t%d" % i]) == t
    assert np.array_equal(img[z["rows%d" % i]].view(np.uint32), z["data%d" % i].view(np.uint32))


def test_photon_pass_in_slices_sums_to_the_whole(hip_ctx):
    """rt_photon_generate_part_device (the multi-GPU split of one pass): three slices emitted into three
    contexts, textures summed through the tensors that alias the device buffers (what the driver's
    all-reduce does in place), against the whole pass and against the oracle; then a render from the
    summed textures against the render after the whole pass."""
    import torch
    from raytracer_amd import driver
    sc = capi.Scene(scene_path("examples3/103_caustic_steps200.xml"), 0.5)
    ref_img, _ = hip_render(hip_ctx, sc)
    ptr, n = hip_ctx.photon_atlas_device()
    assert n > 0
    whole = torch.as_tensor(driver._DeviceFloats(ptr, n), device="cuda").clone()
    parts = [driver.HipRenderer(0) for _ in range(3)]
    tensors = [r.photon_pass(sc, i, 3) for i, r in enumerate(parts)]
    assert all(t.numel() == n for t in tensors)
    torch.cuda.synchronize()
    assert all(float(t.abs().sum()) > 0 for t in tensors)          # every slice deposits something here
    tensors[0] += tensors[1]
    tensors[0] += tensors[2]
    assert torch.allclose(tensors[0], whole, rtol=2e-5, atol=1e-7)
    img = parts[0].render(sc, None, photons_ready=True).cpu().numpy().reshape(ref_img.shape)
    assert np.allclose(img, ref_img, rtol=1e-5, atol=1e-7)
    assert_north_star(capi.quantize(img), capi.quantize(ref_img), "summed slices")
    # against the oracle's slices: same number of deposits per slice
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    for i in range(3):
        hip_ctx.upload(sc)
        hip_ctx.photon_generate_part_device(i, 3)
        torch.cuda.synchronize()
        st = hip_ctx.fetch_stats()
        want = oracle_lib.Photons(sc, i, 3).stats
        assert (st.photon_deposits, st.photon_segments) == (want.photon_deposits, want.photon_segments), i
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
    # invalid slices are refused
    with pytest.raises(RuntimeError):
        hip_ctx.photon_generate_part_device(3, 3)


def test_rccl_all_reduce_on_the_library_owned_photon_textures(hip_ctx):
    """The driver all-reduces the photon textures in place, in memory the C library allocated (not torch's
    allocator).  One-rank RCCL group on this box: the collective must accept that buffer and leave the
    single contribution unchanged; the two-rank arithmetic is covered on CPU (gloo) and by the test above."""
    import torch
    import torch.distributed as dist
    from raytracer_amd import driver
    sc = capi.Scene(scene_path("examples3/103_caustic_steps200.xml"), 0.5)
    r = driver.HipRenderer(0)
    atlas = r.photon_pass(sc, 0, 1)
    before = atlas.clone()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29643")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dist.all_reduce(atlas, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert torch.equal(atlas, before) and float(before.sum()) > 0
        full = driver.render_still(r, sc)          # world 1: whole pass, no split
        img, _ = hip_render(hip_ctx, sc)
        assert np.allclose(full.cpu().numpy(), img, rtol=1e-5, atol=1e-7)
    finally:
        dist.destroy_process_group()


def test_rccl_gathers_of_the_pipelined_driver(monkeypatch):
    """driver.render_units on a one-rank RCCL group with the collective forced on: async gathers (two buffers in
    flight), the RGBA8 payload of plain animations and the float payload of motion blur all pass through RCCL and
    the side-stream downloads; the frames must be those of the same loops without a process group."""
    import torch
    import torch.distributed as dist
    from raytracer_amd import driver
    r = driver.HipRenderer(0)
    files = [scene_path("examples3/103_caustic_animation.xml"), scene_path("examples3/103_caustic_animation_mb8.xml")]
    want = [driver.render_frames(r, f)[0] for f in files]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29644")
    monkeypatch.setenv("RT_DRIVER_FORCE_GATHER", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for f, frames in zip(files, want):
            got, _info, fps = driver.render_frames(r, f)
            assert len(got) == len(frames) == 20 and len(fps) == 20
            for a, b in zip(got, frames):
                assert np.array_equal(a, b)
    finally:
        dist.destroy_process_group()


def test_units_alternating_over_several_contexts_give_the_same_frames(monkeypatch):
    """driver.render_units alternates its units over three contexts (two with RT_DRIVER_LANES=2) on as many streams (the photon pass of one unit beside the
    trace kernel of the one before) and confirms a unit's photon pass a round late.  The frames must be those of one
    context on one stream and of the lock-step loop — also when every photon pass is found too small for its record
    buffer at first and the unit is rendered again (a fresh pair of contexts with a tiny buffer)."""
    from raytracer_amd import driver
    files = [scene_path("examples3/103_caustic_animation.xml"), scene_path("examples3/103_caustic_animation_mb8.xml")]
    monkeypatch.setenv("RT_DRIVER_SEQUENTIAL", "1")
    want = [driver.render_frames(driver.HipRenderer(0), f)[0] for f in files]
    monkeypatch.setenv("RT_DRIVER_SEQUENTIAL", "0")
    for overlap, n_lanes in (("0", "3"), ("1", "2"), ("1", "3"), ("1", "5")):
        monkeypatch.setenv("RT_DRIVER_OVERLAP", overlap)
        monkeypatch.setenv("RT_DRIVER_LANES", n_lanes)
        r = driver.HipRenderer(0)
        for f, frames in zip(files, want):
            got = driver.render_frames(r, f)[0]
            assert len(got) == len(frames)
            for a, b in zip(got, frames):
                assert np.array_equal(a, b), (overlap, n_lanes)
        r.close()
    monkeypatch.setenv("RT_DRIVER_OVERLAP", "1")
    monkeypatch.delenv("RT_DRIVER_LANES")
    r = driver.HipRenderer(0)
    lanes = r.lanes()
    assert len(lanes) == 3 and lanes[0][0] is r
    reruns = []
    for lane, _stream in lanes:
        lane.ctx.set_option(capi.RT_OPT_PHOTON_RECORDS, 4096)       # far too few: the first pass of each context overflows
        ok = lane.photon_pass_ok
        lane.photon_pass_ok = (lambda ok=ok: reruns.append(ok()) or reruns[-1])
    got = driver.render_frames(r, files[0])[0]
    assert reruns.count(False) >= len(lanes)
    for a, b in zip(got, want[0]):
        assert np.array_equal(a, b)
    # options set on the renderer's context reach the twin that renders the alternate units, whenever they were set
    r.ctx.set_option(capi.RT_OPT_DOF_SEED, 7)
    r.ctx.set_option(capi.RT_OPT_PHOTON_ORDER, 0)
    twin = r.lanes()[1][0]
    assert twin.ctx.options == r.ctx.options and twin.ctx.ordered_photons is False
    r.close()
    assert not twin.ctx._ctx


# The reference built with its OWN Makefile flags (-Ofast -flto -march=native) is the secondary parity
# target: fast-math reorders its arithmetic, so the bar against it is the reference's own
# flag noise — what its O-strict and O-fast builds differ by (tests/test_oracle_golden.py asserts those
# numbers on the two sets of reference files).  Bounds: (fraction of pixels beyond 1 LSB, PSNR dB).
OFAST = [
    ("c1_example1", "examples/example1.xml", 0, 0.0, 99.0),
    ("c2_example9_lowres", "examples/example9_lowres.xml", 0, 0.001, 60.0),
    ("c3_fresnel", "examples2/8_fresnel.xml", 0, 0.0, 80.0),
    ("c4_julia_shiny", "examples3/101_julia_shiny.xml", 0, 0.0025, 55.0),   # the reference's own builds differ by 0.19 % here
    ("c5_caustic_animation", "examples3/103_caustic_animation.xml", 10, 0.001, 65.0),
]


@pytest.mark.parametrize("name,rel,frame,max_gt1,min_psnr", OFAST)
def test_secondary_target_reference_built_with_its_own_flags(hip_ctx, name, rel, frame, max_gt1, min_psnr):
    t = float(np.float32(frame) / np.float32(19))
    img, _ = hip_render(hip_ctx, capi.Scene(scene_path(rel), t))
    fast, _n = capi.png_read(golden_path("ofast/" + name + ".png"), frame)
    d = oracle_lib.image_diff(capi.quantize(img), fast)
    print(name, d)
    assert d["frac_gt1"] <= max_gt1 and d["psnr"] >= min_psnr, (name, d)
    # and never further from it than the reference's own O-strict build is
    strict, _n = capi.png_read(golden_path(name + ".png"), frame)
    d0 = oracle_lib.image_diff(strict, fast)
    assert d["frac_gt1"] <= d0["frac_gt1"] + 1e-4 and d["psnr"] >= d0["psnr"] - 0.5, (name, d, d0)


def test_photon_atlas_vs_oracle(hip_ctx):
    """Default (RT_OPT_PHOTON_ORDER = 1): deposits summed in the reference's sequential order, so the photon
    textures — and the picture that gathers them — are the oracle's bit for bit.  With the option off the
    deposits are float atomics: same addends, another order, equal to rounding."""
    sc = capi.Scene(scene_path("examples3/103_caustic_steps200.xml"), 0.5)
    ph = oracle_lib.Photons(sc)
    ref, _ = oracle_lib.render(sc, ph, threads=16)
    for ordered in (1, 0, 1):
        hip_ctx.set_option(capi.RT_OPT_PHOTON_ORDER, ordered)
        hip_ctx.upload(sc)
        hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
        st = hip_ctx.photon_generate()
        hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
        assert st.photon_segments == ph.stats.photon_segments
        assert st.photon_deposits == ph.stats.photon_deposits and st.photon_deposits > 0
        for o in range(sc.desc.n_objects):
            a = hip_ctx.photon_download(o)
            b = ph.texture(o)
            if ordered:
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), o
            else:
                assert np.array_equal(a != 0, b != 0), o
                assert np.allclose(a, b, rtol=2e-6, atol=0), o
        img, _ = hip_ctx.render()
        if ordered:
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        else:
            assert np.allclose(img, ref, rtol=1e-5, atol=1e-7)
            assert oracle_lib.image_diff(capi.quantize(img), capi.quantize(ref))["frac_gt1"] == 0.0


@pytest.mark.parametrize("size, force_sort", [(6, False), (20, False), (150, False), (20, True), (150, True)])
def test_photon_order_with_long_runs_and_on_both_paths(size, force_sort, monkeypatch):
    """rt_photon_order.hip sums a texel's deposits in key order: one wave per run in LDS (runs up to 320 records: all of
    C5's), against global memory beyond that.  Coarse photon textures make the runs long — 6 x 6 texels take all of the
    scene's 0.1 M deposits: runs of thousands — and exercise that path; RT_PHOTON_ORDER_SORT=1 selects the device-wide
    sort that atlases of 2^31 texels would use.  Textures: the oracle's, bit for bit, on every path."""
    if force_sort:
        monkeypatch.setenv("RT_PHOTON_ORDER_SORT", "1")
    ctx = capi.Context(0)        # (the environment is read by rt_create)
    sc = capi.Scene(scene_path("examples3/103_caustic_steps200.xml"), 0.5)
    sc.desc.photon_texture_size = size
    ph = oracle_lib.Photons(sc)
    assert ph.stats.photon_deposits > 10000
    ctx.upload(sc)
    for _ in range(2):           # twice: the per-texel counts must be back at zero after a pass
        ctx.photon_generate()
        for o in range(sc.desc.n_objects):
            a = ctx.photon_download(o)
            b = ph.texture(o)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (size, o)
    ctx.close()


@pytest.mark.parametrize("source", ["examples3/103_caustic_steps200.xml", 1, 9, 13, 21, 25, 33, 5, 29])
def test_photon_roots_that_can_leave_nothing_are_not_traced(hip_ctx, source, tmp_path):
    """The reference deposits a light ray only after a specular bounce (src/photonmap.cpp:79-83), so a root whose ray
    passes the box around every refracting or reflecting object by leaves nothing.  The plain pass skips those roots
    (RT_OPT_CULL_PHOTON_ROOTS); the counters pass traces them, says how many there are and counts any that met a specular
    object all the same (none may).  Photon textures with the option on and off: the oracle's, bit for bit."""
    if isinstance(source, int):
        from mesh_scene import write_random_scene
        sc = capi.Scene(write_random_scene(tmp_path, source), 0.0)
    else:
        sc = capi.Scene(scene_path(source), 0.5)
    ph = oracle_lib.Photons(sc)
    hip_ctx.upload(sc)
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    st = hip_ctx.photon_generate()
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
    words = (C.c_uint64 * 128)()
    nw = capi.hip_lib().rt_debug_counters(hip_ctx._ctx, words, 128)
    assert words[nw - 1] == 0                      # D_PHOTON_CULL_VIOLATIONS, the last slot
    assert st.photon_segments == ph.stats.photon_segments and st.photon_deposits == ph.stats.photon_deposits
    assert st.photon_roots_culled < st.photon_segments
    if not isinstance(source, int):
        assert st.photon_roots_culled > 0.8 * st.photon_segments      # the glass box is small from where the lights are (segments include the bounces)
    try:
        for on in (1, 0):
            hip_ctx.set_option(capi.RT_OPT_CULL_PHOTON_ROOTS, on)
            hip_ctx.photon_generate()
            for o in range(sc.desc.n_objects):
                assert np.array_equal(hip_ctx.photon_download(o).view(np.uint32), ph.texture(o).view(np.uint32)), (on, o)
    finally:
        hip_ctx.set_option(capi.RT_OPT_CULL_PHOTON_ROOTS, 1)


def test_photon_pass_on_a_fresh_context_does_not_count_on_zeroed_memory(monkeypatch, tmp_path):
    """A context whose buffers grow from scene to scene (tools/fuzz_parity.py's situation; the shared context of this
    file has long reached its largest sizes): photon textures of three scenes with ever larger atlases, then of fuzz
    seeds whose pass once scattered its records outside their buffer — the per-texel counts of the ordered sum were
    zeroed only when the allocation's ADDRESS changed, and the allocator may hand a freed block's address out again.
    RT_POISON_ALLOC=1 fills every new buffer with 0xA5 so that such a dependence shows every time."""
    import mesh_scene
    monkeypatch.setenv("RT_POISON_ALLOC", "1")
    ctx = capi.Context(0)
    try:
        sources = [13, 1001, 1005, 1009, 1, "examples3/103_caustic_steps200.xml"]
        for source in sources:
            if isinstance(source, int):
                sc = capi.Scene(mesh_scene.write_random_scene(tmp_path, source), 0.0)
            else:
                sc = capi.Scene(scene_path(source), 0.5)
            ph = oracle_lib.Photons(sc)
            ctx.upload(sc)
            st = ctx.photon_generate()
            assert ctx.photon_ordered()
            for o in range(sc.desc.n_objects):
                assert np.array_equal(ctx.photon_download(o).view(np.uint32), ph.texture(o).view(np.uint32)), (source, o)
            img, _ = ctx.render()
            ref, _ = oracle_lib.render(sc, ph, threads=8)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), source
    finally:
        ctx.close()


def test_photon_record_buffer_grows_when_a_scene_outgrows_it(hip_ctx):
    """A record buffer far too small for the scene's deposits: rt_photon_generate must notice, resize from
    what the pass reported, run again and return the reference's textures; the asynchronous entry point
    adds the surplus with atomics (equal to rounding) and has the right size from the next pass on."""
    sc = capi.Scene(scene_path("examples3/103_caustic_steps200.xml"), 0.5)
    ph = oracle_lib.Photons(sc)
    fresh = capi.Context(0)
    fresh.set_option(capi.RT_OPT_PHOTON_RECORDS, 4096)
    fresh.upload(sc)
    fresh.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    st = fresh.photon_generate()
    assert st.photon_deposits == ph.stats.photon_deposits > 4096
    for o in range(sc.desc.n_objects):
        assert np.array_equal(fresh.photon_download(o).view(np.uint32), ph.texture(o).view(np.uint32)), o
    import torch
    other = capi.Context(0)
    other.set_option(capi.RT_OPT_PHOTON_RECORDS, 4096)
    other.upload(sc)
    for attempt in range(2):
        other.photon_generate_device()
        assert other.photon_ordered() == (attempt == 1)       # reported, and the buffer has the right size afterwards
        for o in range(sc.desc.n_objects):
            a, b = other.photon_download(o), ph.texture(o)
            if attempt == 0:
                assert np.allclose(a, b, rtol=2e-6, atol=0), o
            else:
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), o
    # the driver's renderer asks and repeats the pass by itself
    from raytracer_amd import driver
    r = driver.HipRenderer(0)
    r.ctx.set_option(capi.RT_OPT_PHOTON_RECORDS, 4096)
    img = r.render(sc).cpu().numpy()
    ref, _ = oracle_lib.render(sc, ph, threads=16)
    assert np.array_equal(img.reshape(ref.shape).view(np.uint32), ref.view(np.uint32))
    hip_ctx.set_option(capi.RT_OPT_PHOTON_ORDER, 0)
    hip_ctx.upload(sc)
    hip_ctx.photon_generate_device()
    assert not hip_ctx.photon_ordered()
    hip_ctx.set_option(capi.RT_OPT_PHOTON_ORDER, 1)


# ---- partitions, accumulation, quantiser -------------------------------------------------------------

def test_band_partition_reassembles_the_frame(hip_ctx):
    sc = capi.Scene(scene_path("examples/example9_lowres.xml"), 0.0)
    sc.set_camera(203, 157, 0)     # ragged: not a multiple of the 8x8 tile nor of the band height
    full, _ = hip_render(hip_ctx, sc)
    out = np.zeros_like(full)
    for band_rows, stride in ((8, 4), (5, 3), (157, 1), (1, 2)):
        out[:] = 0
        for first in range(stride):
            b = capi.Bands(band_rows, first, stride)
            part, _ = hip_ctx.render(b)
            rows = [y for band in range(first, (157 + band_rows - 1) // band_rows, stride)
                    for y in range(band * band_rows, min(157, (band + 1) * band_rows))]
            assert part.shape[0] == len(rows) * 203
            out[rows] = part.reshape(len(rows), 203, 4)
        assert np.array_equal(out.view(np.uint32), full.view(np.uint32)), (band_rows, stride)
    # an empty partition is legal and writes nothing
    empty, _ = hip_ctx.render(capi.Bands(8, 1000, 4))
    assert empty.shape[0] == 0


def test_errors_are_reported_not_thrown(hip_ctx):
    lib = capi.hip_lib()
    ctx = C.c_void_p()
    assert lib.rt_create(C.byref(ctx), 0) == 0
    out = np.zeros(4, dtype=np.float32)
    assert lib.rt_render(ctx, None, out.ctypes.data, None) == -4          # RT_ERR_NO_SCENE
    assert b"before rt_scene_upload" in lib.rt_last_error(ctx)
    sc = capi.Scene(scene_path("examples/example1.xml"), 0.0)
    bad = capi.SceneDesc.from_buffer_copy(sc.desc)
    bad.abi_version = 99
    assert lib.rt_scene_upload(ctx, C.byref(bad)) == -1
    objs = (capi.Object * 1)(capi.Object(capi.RT_OBJ_TRIANGLE, 5, 0))
    bad = capi.SceneDesc.from_buffer_copy(sc.desc)
    bad.n_objects = 1
    bad.objects = objs
    assert lib.rt_scene_upload(ctx, C.byref(bad)) == -1                   # index validated on the host
    assert lib.rt_create(C.byref(C.c_void_p()), 4096) == -1
    lib.rt_destroy(ctx)


def test_mul_add_and_quantize_device(hip_ctx):
    import torch
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 2, size=(1000, 4)).astype(np.float32)
    b = rng.uniform(-1, 2, size=(1000, 4)).astype(np.float32)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    f = np.float32(1.0) / np.float32(8)
    hip_ctx.mul_add_device(da.data_ptr(), db.data_ptr(), float(f), a.size, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(da.cpu().numpy(), a + b * f)                    # Picture::mulAdd
    q = torch.empty((1000, 4), dtype=torch.uint8, device="cuda")
    hip_ctx.quantize_device(da.data_ptr(), q.data_ptr(), 1.0, 1000, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(q.cpu().numpy(), capi.quantize(a + b * f))      # Color::scaleOut


@pytest.mark.parametrize("source,size,supersampling,bands", [
    ("examples/example1.xml", (200, 120), 1, None),            # one sample per pixel: the trace kernel quantises
    ("examples/example1.xml", (96, 64), 2, None),              # several: the kernel that sums them does
    ("examples2/4_julia.xml", (96, 64), 1, None),              # the Julia kernel's shader waves
    ("examples3/103_caustic_steps200.xml", (160, 120), 1, (8, 1, 3)),   # a band share, photon textures read
])
def test_render_rgba8_device_is_render_then_quantize(hip_ctx, source, size, supersampling, bands):
    """rt_render_rgba8_device (the picture leaves the kernel as Color::scaleOut bytes, src/types.h:274-282) against
    rt_render_device + rt_quantize_device, and against scaleOut of the oracle's float picture."""
    import torch
    sc = capi.Scene(scene_path(source), 0.0)
    sc.set_camera(size[0], size[1], supersampling)
    hip_ctx.upload(sc)
    if sc.desc.photon_steps > 0:
        hip_ctx.photon_generate()
    b = capi.Bands(*bands) if bands else None
    n = hip_ctx.pixel_count(b)
    stream = torch.cuda.current_stream().cuda_stream
    rgba = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
    two = torch.empty((n, 4), dtype=torch.uint8, device="cuda:0")
    one = torch.full((n + 16, 4), 0x5A, dtype=torch.uint8, device="cuda:0")      # (16 pixels of canary behind the picture)
    hip_ctx.render_device(rgba.data_ptr(), stream, b)
    hip_ctx.quantize_device(rgba.data_ptr(), two.data_ptr(), 1.0, n, stream)
    hip_ctx.render_rgba8_device(one.data_ptr(), stream, b)
    torch.cuda.synchronize()
    hip_ctx.launch_status()
    assert np.array_equal(one[:n].cpu().numpy(), two.cpu().numpy())
    assert (one[n:].cpu().numpy() == 0x5A).all()
    if bands is None:
        ref, _ = oracle_lib.render(sc, None, threads=16)
        assert np.array_equal(one[:n].cpu().numpy(), capi.quantize(ref.reshape(-1, 4)))
    # a gain other than 1 reaches both kernels
    hip_ctx.quantize_device(rgba.data_ptr(), two.data_ptr(), 0.5, n, stream)
    hip_ctx.render_rgba8_device(one.data_ptr(), stream, b, gain=0.5)
    torch.cuda.synchronize()
    assert np.array_equal(one[:n].cpu().numpy(), two.cpu().numpy())


def test_shared_gpu_option_hands_every_tile_out_through_the_counter_and_changes_no_bit(hip_ctx):
    """RT_OPT_SHARED_GPU (include/rt_hip.h): no tiles owned outright by the waves.  A frame large enough to have owned
    tiles (more than two per wave of the grid), with one sample per pixel and with four."""
    for source, size, supersampling in (("examples/example9.xml", (1280, 720), 1), ("examples/example1.xml", (640, 400), 2)):
        sc = capi.Scene(scene_path(source), 0.0)
        sc.set_camera(size[0], size[1], supersampling)
        hip_ctx.upload(sc)
        plain, _ = hip_ctx.render()
        try:
            hip_ctx.set_option(capi.RT_OPT_SHARED_GPU, 1)
            shared, _ = hip_ctx.render()
            bands = capi.Bands(8, 1, 3)
            part_shared, _ = hip_ctx.render(bands)
        finally:
            hip_ctx.set_option(capi.RT_OPT_SHARED_GPU, 0)
        part, _ = hip_ctx.render(bands)
        assert np.array_equal(plain.view(np.uint32), shared.view(np.uint32))
        assert np.array_equal(part.view(np.uint32), part_shared.view(np.uint32))


# ---- the callers either side of the path: CLI loops and the driver ---------------------------------

def test_cli_renders_the_caustic_animation_like_the_reference(tmp_path):
    """raytracer_amd/rt_amd (C++ host shim over the C ABI, loop of reference src/main.cpp:80-112) on
    configs[4] as shipped: all 20 APNG frames against the reference executable's own file."""
    import os
    import subprocess
    from conftest import ROOT
    out = str(tmp_path / "anim.png")
    subprocess.run([os.path.join(ROOT, "raytracer_amd", "rt_amd"), scene_path("examples3/103_caustic_animation.xml"), out],
                   check=True, timeout=300, stdout=subprocess.DEVNULL)
    for frame in range(20):
        img, n = capi.png_read(out, frame)
        gold, m = capi.png_read(golden_path("c5_caustic_animation.png"), frame)
        assert n == m == 20
        d = assert_north_star(img, gold, "cli frame %d" % frame)
        assert d["max"] == 0, (frame, d)   # ordered photon sums: the reference's file, pixel for pixel


def test_driver_motion_blur_like_the_reference(tmp_path):
    """configs[4] with <motionblur subframes="8"/>: 160 (frame, sub-frame) units, photon pass each,
    accumulated with rt_mul_add_device in sub-frame order (reference src/main.cpp:115-158)."""
    from raytracer_amd import driver
    out = str(tmp_path / "mb8.png")
    frames = driver.render_scene_file(driver.HipRenderer(0), scene_path("examples3/103_caustic_animation_mb8.xml"), out)
    assert len(frames) == 20
    for frame in (0, 7, 19):
        gold, n = capi.png_read(golden_path("c5_caustic_animation_mb8.png"), frame)
        assert n == 20
        assert_north_star(frames[frame], gold, "mb8 frame %d" % frame)
        back, _ = capi.png_read(out, frame)
        assert np.array_equal(back, frames[frame])


def test_driver_still_with_bands(hip_ctx):
    """The still-image loop of the driver on one rank: bands -> compact buffer -> de-interleave."""
    from raytracer_amd import driver
    sc = capi.Scene(scene_path("examples/example9_lowres.xml"), 0.0)
    r = driver.HipRenderer(0)
    full = driver.render_still(r, sc).cpu().numpy()
    ref, _ = hip_render(hip_ctx, sc)
    assert np.array_equal(full.view(np.uint32), ref.view(np.uint32))


def test_depth_of_field(hip_ctx):
    """lens_size != 0: the reference itself is not reproducible (random_device seeds per worker).
    The lens draws are a counter-based hash of (pixel, sub-pixel, seed) here and — the same function — in
    the oracle, so everything around the draw is compared bit for bit, for two seeds; the reference's own
    pictures can only be compared statistically (its two runs differ from each other as much)."""
    sc = capi.Scene(scene_path("examples2/7_dof.xml"), 0.0)
    sc.set_camera(200, 150, 0)
    hip_ctx.upload(sc)
    pictures = []
    try:
        for seed in (1, 2):
            hip_ctx.set_option(capi.RT_OPT_DOF_SEED, seed)
            oracle_lib.set_dof_seed(seed)
            a, _ = hip_ctx.render()
            b, _ = hip_ctx.render()
            ref, _ = oracle_lib.render(sc, None, threads=16)
            assert np.array_equal(a, b)                                   # deterministic for a seed
            assert np.array_equal(a.view(np.uint32), ref.view(np.uint32)), seed
            pictures.append(a)
    finally:
        hip_ctx.set_option(capi.RT_OPT_DOF_SEED, 0)
        oracle_lib.set_dof_seed(0)
    assert not np.array_equal(pictures[0], pictures[1])                   # the seed matters
    assert abs(float(pictures[0][..., :3].mean()) - float(pictures[1][..., :3].mean())) < 0.01


@pytest.mark.parametrize("rel,t", [("examples2/8_fresnel.xml", 0.0), ("examples3/103_caustic_animation.xml", 0.5),
                                   ("examples3/101_julia_shiny.xml", 0.0), ("examples/example9_lowres.xml", 0.0),
                                   ("examples/example8.xml", 0.0), ("examples2/1_transparent.xml", 0.0)])
def test_shadow_rays_that_can_only_add_zero_are_not_traced(hip_ctx, rel, t):
    """Where a light is behind the surface or kd == 0, and its mirrored direction points away from the viewer or ks == 0
    (the back of every object as seen from a light; all of the glass and mirror objects of C3, C4, C5), the reference
    casts a shadow ray and multiplies what it finds by zero.  The plain launch skips those rays
    (RT_OPT_ELIDE_UNLIT_SHADOWS), the counters launch traces them and says how many there are: same float picture bit
    for bit, with the option on or off, as the oracle's."""
    sc = capi.Scene(scene_path(rel), t)
    if sc.desc.n_julias > 0:
        sc.set_camera(240, 150, 0)
    hip_ctx.upload(sc)
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    hip_ctx.photon_generate()
    counted, st = hip_ctx.render()
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
    assert 0 < st.shadow_rays_elided < st.shadow_rays
    try:
        pictures = []
        for on in (1, 0):
            hip_ctx.set_option(capi.RT_OPT_ELIDE_UNLIT_SHADOWS, on)
            pictures.append(hip_ctx.render()[0])
    finally:
        hip_ctx.set_option(capi.RT_OPT_ELIDE_UNLIT_SHADOWS, 1)
    ph = oracle_lib.Photons(sc) if sc.desc.photon_steps > 0 else None
    ref, ost = oracle_lib.render(sc, ph, threads=16)
    assert st.shadow_rays == ost.shadow_rays
    for img in pictures + [counted]:
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_depth_of_field_against_the_reference_statistically(hip_ctx):
    """The HIP picture of examples2/7_dof.xml (full size, 7x7 supersampling) against the reference's own two runs
    (tests/golden/x_dof_run{1,2}.png): PSNR no more than 0.5 dB below what the runs reach against each other, and
    per-channel means within 0.05 LSB (north-star bar for this scene: statistical, the reference is unseeded)."""
    runs = [capi.png_read(golden_path("x_dof_run%d.png" % i), 0)[0] for i in (1, 2)]
    ref_psnr, _ = oracle_lib.dof_statistics(runs[0], runs[1])
    sc = capi.Scene(scene_path("examples2/7_dof.xml"), 0.0)
    hip_ctx.upload(sc)
    try:
        for seed in (0, 3):
            hip_ctx.set_option(capi.RT_OPT_DOF_SEED, seed)
            q = capi.quantize(hip_ctx.render()[0])
            for run in runs:
                psnr, means = oracle_lib.dof_statistics(q, run)
                assert psnr >= ref_psnr - 0.5 and means.max() < 0.05, (seed, psnr, ref_psnr, means)
    finally:
        hip_ctx.set_option(capi.RT_OPT_DOF_SEED, 0)


# ---- GPU-built BVH ------------------------------------------------------------------------------------

from mesh_scene import write_mesh_scene  # noqa: E402


def test_bvh_equals_the_sequential_scan_on_a_large_mesh(tmp_path):
    """~11k triangles: the GPU-built BVH path against the oracle's brute-force scan, bit for bit, and
    against our own brute-force kernel (BVH switched off)."""
    sc = capi.Scene(write_mesh_scene(tmp_path), 0.0)
    assert sc.desc.n_triangles > 10000
    ctx = capi.Context(0)
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, 64)
    ctx.upload(sc)
    img, st = ctx.render()
    assert st.bvh_node_visits > 0
    ref, ost = oracle_lib.render(sc, None, threads=16)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert st.triangle_tests < ost.triangle_tests / 20          # the point of the BVH
    ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, 1 << 40)
    ctx.upload(sc)
    brute, st2 = ctx.render()
    assert st2.bvh_node_visits == 0 and st2.triangle_tests == ost.triangle_tests
    assert np.array_equal(brute.view(np.uint32), img.view(np.uint32))
    ctx.close()


def test_zero_area_triangles_keep_the_sequential_scan(tmp_path):
    """Zero-area triangles make the reference divide by zero (src/objects.cpp:66-78) and accept hits at
    NaN distance, after which the LAST object hit in scene order wins; only the scan reproduces that,
    so such a run gets no BVH — and the picture still equals the oracle's, NaNs included."""
    sc = capi.Scene(write_mesh_scene(tmp_path, 12, 24, poles=True), 0.0)
    ctx = capi.Context(0)
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, 2)
    ctx.upload(sc)
    img, st = ctx.render()
    ref, ost = oracle_lib.render(sc, None, threads=16)
    assert st.triangle_tests == ost.triangle_tests
    assert np.array_equal(np.isnan(img), np.isnan(ref))
    assert np.array_equal(np.nan_to_num(img).view(np.uint32), np.nan_to_num(ref).view(np.uint32))
    ctx.close()


def test_bvh_forced_on_the_configured_scenes():
    """The same pictures with every triangle run (2 triangles and up) behind a BVH."""
    ctx = capi.Context(0)
    ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, 2)
    for rel, t in (("examples/example9_lowres.xml", 0.0), ("examples2/8_fresnel.xml", 0.0), ("examples3/103_caustic_steps200.xml", 0.5)):
        sc = capi.Scene(scene_path(rel), t)
        ctx.upload(sc)
        pst = ctx.photon_generate()
        img, _ = ctx.render()
        ph = oracle_lib.Photons(sc) if sc.desc.photon_steps > 0 else None
        ref, _ = oracle_lib.render(sc, ph, threads=16)
        # photon scenes too: the deposits are summed in the reference's order
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), rel
        assert pst.kernel_ms >= 0.0
    ctx.close()


# ---- wave-level culling and the job queues of the Julia kernel ---------------------------------------

from mesh_scene import write_culling_scene, write_two_julia_scene  # noqa: E402


def test_culling_never_changes_the_answer_far_from_the_origin(tmp_path):
    """Ellipsoids and meshes 6400 units from the world origin (the reference's object-space
    discriminants are dominated by rounding there), a sphere around the camera, a light inside a sphere:
    bounding-sphere and chunk culling must leave every pixel and every counter as the oracle has them."""
    sc = capi.Scene(write_culling_scene(tmp_path), 0.0)
    assert sc.desc.n_spheres == 10 and sc.desc.n_triangles >= 70
    ctx = capi.Context(0)
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    ref, ost = oracle_lib.render(sc, None, threads=16)
    # with the scan (chunk culling) for every run, and as configured by default (the 70-triangle run behind the BVH)
    for bvh_min in (1 << 40, capi.RT_DEFAULT_BVH_MIN_TRIANGLES):
        ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, bvh_min)
        ctx.upload(sc)
        img, st = ctx.render()
        assert np.array_equal(np.isnan(img), np.isnan(ref))
        assert np.array_equal(np.nan_to_num(img).view(np.uint32), np.nan_to_num(ref).view(np.uint32))
        for k in ("primary_rays", "secondary_rays", "shadow_rays", "sphere_tests", "triangle_tests", "shaded_hits"):
            if k == "triangle_tests" and st.bvh_node_visits > 0:
                assert st.triangle_tests < ost.triangle_tests
                continue
            assert getattr(st, k) == getattr(ost, k), k
        assert (st.bvh_node_visits > 0) == (bvh_min == capi.RT_DEFAULT_BVH_MIN_TRIANGLES)
    assert st.secondary_rays > 0 and st.shadow_rays > st.primary_rays
    ctx.close()


def test_two_julia_sets_through_the_job_queues(tmp_path):
    """Queries that march one Julia set and then the next one, reflections between them and shadow rays
    with a finite maximum, against the oracle bit for bit (tracer / marcher hand-over of trace_julia_kernel)."""
    sc = capi.Scene(write_two_julia_scene(tmp_path), 0.0)
    assert sc.desc.n_julias == 2
    ctx = capi.Context(0)
    ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    ctx.upload(sc)
    img, st = ctx.render()
    ref, ost = oracle_lib.render(sc, None, threads=16)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    for k in ("primary_rays", "secondary_rays", "shadow_rays", "julia_queries", "julia_steps", "julia_normals", "shaded_hits"):
        assert getattr(st, k) == getattr(ost, k), k
    assert st.julia_normals > 0 and st.secondary_rays > 0
    # a second render on the same context gives the same picture (queues start clean every launch)
    img2, _ = ctx.render()
    assert np.array_equal(img.view(np.uint32), img2.view(np.uint32))
    ctx.close()


# ---- every example scene the reference ships -----------------------------------------------------------

def _scene_xml(tmp_path, name, camera, lights, surfaces, extra=""):
    xml = tmp_path / (name + ".xml")
    xml.write_text('''<scene output_file="%s.png"><background_color r="0.25" g="0.5" b="0.75"/>%s
  <camera><position x="0" y="0" z="1"/><lookat x="0" y="0" z="-2.5"/><up x="0" y="1" z="0"/><horizontal_fov angle="45"/>%s</camera>
  <lights>%s</lights><surfaces>%s</surfaces></scene>''' % (name, extra, camera, lights, surfaces))
    return str(xml)


GLASS_SPHERE = '''<sphere radius="1.0"><position x="0" y="0" z="-3"/><material_solid><color r="0.2" g="0.6" b="0.3"/>
  <phong ka="0.3" kd="0.9" ks="1.0" exponent="50"/><reflectance r="0.5"/><transmittance t="0.5"/><refraction iof="1.5"/></material_solid></sphere>'''
WALL = '''<mesh name="%s"><material_solid><color r="0.7" g="0.7" b="0.7"/><phong ka="0.3" kd="0.9" ks="0.2" exponent="10"/>
  <reflectance r="0"/><transmittance t="0"/><refraction iof="0"/></material_solid>
  <transform><translate x="0" y="0" z="-6"/><scale x="8" y="8" z="1"/></transform></mesh>'''
POINT = '<ambient_light><color r="1" g="1" b="1"/></ambient_light><point_light><color r="0.8" g="0.8" b="0.8"/><position x="2" y="3" z="1"/></point_light>'
PARALLEL = '<ambient_light><color r="1" g="1" b="1"/></ambient_light><parallel_light><color r="0.8" g="0.8" b="0.8"/><direction x="-1" y="-1" z="-1"/></parallel_light>'


MIRROR = '''<mesh name="%s"><material_solid><color r="0.9" g="0.9" b="0.9"/><phong ka="0.1" kd="0.3" ks="0.5" exponent="30"/>
  <reflectance r="1.0"/><transmittance t="0"/><refraction iof="2.5" ec="3.0"/></material_solid>
  <transform><translate x="0" y="0" z="%s"/>%s<scale x="400" y="400" z="1"/></transform></mesh>'''


@pytest.mark.parametrize("case", ["empty", "one_pixel", "no_bounces", "no_lights", "photons_without_point_light", "wide_thin", "photons_one_receiver",
                                  "deep_mirrors", "photons_deep"])
def test_degenerate_scenes(hip_ctx, tmp_path, case):
    """Edge cases of the path: nothing to hit, a single pixel, recursion cut at the first hit, no lights at all,
    a photon pass with nothing to emit, an image thinner than a tile, a photon pass into a single receiver."""
    wall = WALL % scene_path("examples/plane_small.obj")
    res = lambda w, h, b=4, ss=1: '<resolution horizontal="%d" vertical="%d"/><max_bounces n="%d"/><supersampling subpixels_peraxis="%d"/>' % (w, h, b, ss)
    path = {
        "empty": lambda: _scene_xml(tmp_path, case, res(37, 21), "", ""),
        "one_pixel": lambda: _scene_xml(tmp_path, case, res(1, 1, 4, 3), POINT, GLASS_SPHERE + wall),
        "no_bounces": lambda: _scene_xml(tmp_path, case, res(64, 48, 0), POINT, GLASS_SPHERE + wall),
        "no_lights": lambda: _scene_xml(tmp_path, case, res(64, 48), "", GLASS_SPHERE + wall),
        "photons_without_point_light": lambda: _scene_xml(tmp_path, case, res(64, 48), PARALLEL, GLASS_SPHERE + wall,
                                                           '<caustic steps="50" texture_size="16" factor="0.1"/>'),
        "wide_thin": lambda: _scene_xml(tmp_path, case, res(515, 3), POINT, GLASS_SPHERE + wall),
        # two facing mirrors: reflection chains as deep as the recursion limit allows (64 is the ABI's maximum)
        "deep_mirrors": lambda: _scene_xml(tmp_path, case, res(96, 64, 64), POINT, GLASS_SPHERE.replace('t="0.5"', 't="0"')
                                           + MIRROR % (scene_path("examples/plane_small.obj"), "-7", "")
                                           + MIRROR % (scene_path("examples/plane_small.obj"), "3", '<rotateY theta="180"/>')),
        # photons bouncing between the mirrors for up to 40 segments before they land on the diffuse sphere
        "photons_deep": lambda: _scene_xml(tmp_path, case, res(96, 64, 40), POINT, GLASS_SPHERE.replace('t="0.5"', 't="0"').replace('r="0.5"', 'r="0"').replace('iof="1.5"', 'iof="0"')
                                           + MIRROR % (scene_path("examples/plane_small.obj"), "-7", "")
                                           + MIRROR % (scene_path("examples/plane_small.obj"), "3", '<rotateY theta="180"/>'),
                                           '<caustic steps="200" texture_size="16" factor="0.2"/>'),
        "photons_one_receiver": lambda: _scene_xml(tmp_path, case, res(96, 64), POINT, GLASS_SPHERE + wall,
                                                    '<caustic steps="1000" texture_size="8" factor="0.05"/>'),
    }[case]()
    sc = capi.Scene(path, 0.0)
    hip_ctx.upload(sc)
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    pst = hip_ctx.photon_generate()
    img, st = hip_ctx.render()
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
    ph = oracle_lib.Photons(sc) if sc.desc.photon_steps > 0 else None
    ref, ost = oracle_lib.render(sc, ph, threads=8)
    assert img.shape == ref.shape
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), case
    assert (st.primary_rays, st.secondary_rays, st.shadow_rays) == (ost.primary_rays, ost.secondary_rays, ost.shadow_rays)
    if ph is not None:
        assert (pst.photon_segments, pst.photon_deposits) == (ph.stats.photon_segments, ph.stats.photon_deposits)
        for o in range(sc.desc.n_objects):
            assert np.array_equal(hip_ctx.photon_download(o).view(np.uint32), ph.texture(o).view(np.uint32)), (case, o)
    if case == "empty":
        assert np.all(img == np.array([0.25, 0.5, 0.75, 1.0], dtype=np.float32))
    if case == "photons_one_receiver":
        assert pst.photon_deposits > 32 * 64      # many deposits per texel: the long-run path of rt_photon_order.hip


@pytest.mark.parametrize("seed", list(range(32)) + [78, 5477])      # 5477: a sphere behind a Julia set, 5 000 units from the origin (accept_julia_hit)
def test_random_scenes_vs_oracle(hip_ctx, tmp_path, seed):
    """Seeded random scenes over the whole grammar of the path (tests/mesh_scene.py: write_random_scene):
    pictures, photon textures and ray counts bit-exact against the oracle."""
    import mesh_scene
    sc = capi.Scene(mesh_scene.write_random_scene(tmp_path, seed), 0.0)
    hip_ctx.upload(sc)
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 1)
    pst = hip_ctx.photon_generate()
    img, st = hip_ctx.render()
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
    fast, _ = hip_ctx.render()                     # the kernels without counters are other instantiations
    ph = oracle_lib.Photons(sc) if sc.desc.photon_steps > 0 else None
    ref, ost = oracle_lib.render(sc, ph, threads=8)
    bad = np.argwhere(np.any(img.view(np.uint32) != ref.view(np.uint32), axis=2))
    assert len(bad) == 0, (seed, len(bad), bad[:4].tolist())
    assert np.array_equal(fast.view(np.uint32), ref.view(np.uint32)), seed
    # (julia_queries is not compared: a shadow ray that another object blocks is not marched at all here, while the
    # reference marches it if the Julia set comes first in scene order — same answer, fewer queries)
    for k in ("primary_rays", "secondary_rays", "shadow_rays", "shaded_hits", "fresnel_evals", "texture_lookups"):
        assert getattr(st, k) == getattr(ost, k), (seed, k)
    if ph is not None:
        assert (pst.photon_segments, pst.photon_deposits) == (ph.stats.photon_segments, ph.stats.photon_deposits), seed
        for o in range(sc.desc.n_objects):
            assert np.array_equal(hip_ctx.photon_download(o).view(np.uint32), ph.texture(o).view(np.uint32)), (seed, o)
    # once more with every triangle run behind the BVH traversal (seed 78, twin boxes: exact ties between flat leaf
    # boxes used to go to the earlier object when the slab distance rounded past the triangle test's)
    try:
        hip_ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, 2)
        hip_ctx.upload(sc)
        hip_ctx.photon_generate()
        bvh, _ = hip_ctx.render()
    finally:
        hip_ctx.set_option(capi.RT_OPT_BVH_MIN_TRIANGLES, capi.RT_DEFAULT_BVH_MIN_TRIANGLES)
    assert np.array_equal(bvh.view(np.uint32), ref.view(np.uint32)), (seed, "bvh")


REF_EXECUTABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "raytracer_ostrict")


@pytest.mark.skipif(not os.path.exists(REF_EXECUTABLE), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", [0, 1, 2, 5, 9, 12, 22, 78])
def test_cli_writes_the_file_the_reference_executable_writes(tmp_path, seed):
    """End to end, as a user swaps the programs: the drop-in CLI (loader, kernels, quantiser, PNG writer) and the
    reference's own executable (built from its sources in the build container; it travels as oracle/_ref) on the
    same random scene file — the two PNG files must be the same bytes.  tools/fuzz_cli.py is the long version."""
    import subprocess
    import mesh_scene
    scene = mesh_scene.write_random_scene(tmp_path, seed)
    a, b = str(tmp_path / "reference.png"), str(tmp_path / "hip.png")
    subprocess.check_call([REF_EXECUTABLE, scene, a], stdout=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(os.path.dirname(REF_EXECUTABLE), "..", "..", "raytracer_amd", "rt_amd"), scene, b], stdout=subprocess.DEVNULL)
    assert open(a, "rb").read() == open(b, "rb").read()


RT_AMD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "rt_amd")


@pytest.mark.parametrize("kind,tags,devices", [
    ("still", "", 2), ("still", "", 3),
    ("still_blur", '<motionblur subframes="5"/>', 2),
    ("animation", '<animation length="0.7" fps="10;20"/>', 3),
    ("animation_blur", '<animation length="0.5" fps="8"/><motionblur subframes="2;4"/>', 2),
])
def test_cli_on_several_devices_writes_the_same_file(tmp_path, kind, tags, devices):
    """rt_amd --devices N (one host thread and one rt_ctx per GPU, C++ only): bands of a still, sub-frames of a blurred
    still, frames of an animation (with and without blur, keyframed fps and sub-frame count) come back in order and
    the file is byte for byte the single-device one.  This box has one GPU: RT_AMD_DEVICE_LIST names it N times, which
    exercises the sharding, the ordering and the per-context state; N distinct GPUs differ only in the device index
    handed to rt_create."""
    import subprocess
    from test_driver_distributed import small_scene
    scene = small_scene(tmp_path, tags, size=(64, 44), steps=40)
    one, many = str(tmp_path / "one.png"), str(tmp_path / "many.png")
    subprocess.check_call([RT_AMD, scene, one], stdout=subprocess.DEVNULL)
    env = dict(os.environ, RT_AMD_DEVICE_LIST=",".join(["0"] * devices))
    subprocess.check_call([RT_AMD, "--devices", str(devices), scene, many], stdout=subprocess.DEVNULL, env=env)
    assert open(one, "rb").read() == open(many, "rb").read()
    _img, n = capi.png_read(one, 0)
    assert n == {"still": 1, "still_blur": 1, "animation": 7, "animation_blur": 4}[kind]


ALL_SCENES = [
    ("examples/example2.xml", 0.0), ("examples/example3.xml", 0.0), ("examples/example4.xml", 0.0),
    ("examples/example5.xml", 0.0), ("examples/example6.xml", 0.0), ("examples/example7.xml", 0.0),
    ("examples/example8.xml", 0.0), ("examples/example9.xml", 0.0),
    ("examples2/2_animation.xml", 0.37), ("examples2/3_motionblur.xml", 0.61), ("examples2/5_julia_animation.xml", 0.25),
    ("examples2/7_dof.xml", 0.0), ("examples2/9_caustic.xml", 0.0), ("examples2/10_caustic_texture.xml", 0.0),
    ("examples3/102_fresnel_animation.xml", 0.8), ("examples3/104_julia_animation.xml", 0.5),
]


@pytest.mark.parametrize("rel,t", ALL_SCENES)
def test_every_example_scene_vs_oracle(hip_ctx, rel, t):
    """The remaining example scenes of the reference (the oracle is pinned to the reference itself on
    each of them by tests/test_oracle_golden.py::test_oracle_vs_live_reference).  Float pictures are
    bit-exact, with or without a photon map."""
    sc = capi.Scene(scene_path(rel), t)
    if sc.width * sc.height > 600 * 600:
        sc.set_camera(512, 512, 0)          # example9.xml is 2048x2048; its 4K variant is tested at full size
    hip_ctx.upload(sc)
    hip_ctx.set_option(capi.RT_OPT_COLLECT_STATS, 0)
    hip_ctx.photon_generate()
    img, _ = hip_ctx.render()
    ph = oracle_lib.Photons(sc) if sc.desc.photon_steps > 0 else None
    ref, _ = oracle_lib.render(sc, ph, threads=16)
    # textured spheres included: their texture coordinates go through rt_libm.h's restatement of glibc's
    # atan2f/asinf (src/objects.cpp:47), so example8 is bit-exact like the rest
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), rel
