"""The multi-rank driver (raytracer_amd/driver.py) on CPU: world_size 2 over gloo.

The renderer is a stand-in that computes with the oracle, so what is under test is the sharding
(interleaved bands, frames, (frame, sub-frame) pairs), the gather to rank 0, the motion-blur
accumulation order and the writer — the gathered result must equal the single-process result
bit for bit.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib
from conftest import ROOT, scene_path
from raytracer_amd import capi, driver

ASSETS2 = os.path.join(ROOT, "scenes", "examples2")


def small_scene(tmp_path, extra_scene_tags="", camera_pos='x="0;1.5(b)" y="0;1(b)" z="-2;-3(b)"', size=(40, 28), steps=60):
    xml = tmp_path / "s.xml"
    xml.write_text('''<scene output_file="%s">
  <background_color r="0.0" g="0.0" b="0.0"/>%s
  <caustic steps="%d" texture_size="24" factor="0.5"/>
  <camera><position %s/><lookat x="0" y="0" z="-6.5"/><up x="0" y="1" z="0"/>
    <horizontal_fov angle="45"/><resolution horizontal="%d" vertical="%d"/><max_bounces n="4"/></camera>
  <lights><ambient_light><color r="1" g="1" b="1"/></ambient_light>
    <point_light><color r="0.7" g="0.7" b="0.7"/><position x="2" y="3" z="-2.5"/></point_light></lights>
  <surfaces>
    <mesh name="%s/open_room.obj"><material_solid><color r="0.6" g="0.6" b="0.6"/><phong ka="0.3" kd="0.9" ks="1.0" exponent="20"/>
      <reflectance r="0"/><transmittance t="0"/><refraction iof="0"/></material_solid></mesh>
    <mesh name="%s/box2.obj"><material_solid><color r="0.17" g="0.18" b="0.5"/><phong ka="0" kd="0" ks="0" exponent="200"/>
      <reflectance r="0.8"/><transmittance t="0.8"/><refraction iof="1.5168" ec="9.7525e-9" disp="-0.2"/></material_solid></mesh>
  </surfaces></scene>''' % (tmp_path / "out.png", extra_scene_tags, steps, camera_pos, size[0], size[1], ASSETS2, ASSETS2))
    return str(xml)


def run_two_ranks(scene, tmp_path, **extra_env):
    out, result = str(tmp_path / "dist.png"), str(tmp_path / "dist.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "tests", "dist_worker.py"), scene, out, result]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return np.load(result), out


def single_process_frames(scene):
    """The four loops run sequentially in this process with the oracle (the reference's order)."""
    probe = capi.Scene(scene, 0.0)
    info = probe.info
    h, w = probe.height, probe.width

    def picture(t):
        sc = capi.Scene(scene, float(t))
        ph = oracle_lib.Photons(sc) if sc.desc.photon_steps > 0 else None
        return oracle_lib.render(sc, ph, threads=4)[0]
    animated = info.frames > 1 and info.still_time == np.float32(capi.FLT_MAX)
    if info.sub_frames <= 1:
        times = [np.float32(f) / np.float32(info.frames - 1) for f in range(info.frames)] if animated else [driver.still_time(info)]
        return [capi.quantize(picture(t)) for t in times]
    frames, acc = [], None
    for frame, sub, t, weight, k in driver.motion_blur_units(scene, info):
        if acc is None:
            acc = np.zeros((h, w, 4), dtype=np.float32)
        acc += picture(t) * np.float32(weight)
        if sub == k - 1:
            frames.append(capi.quantize(acc))
            acc = None
    return frames


@pytest.mark.parametrize("kind,tags", [
    ("still", ""),
    ("animation", '<animation length="0.5" fps="6"/>'),
    ("still_blur", '<motionblur subframes="3"/>'),
    ("animation_blur", '<animation length="0.4" fps="5"/><motionblur subframes="2;4"/>'),
])
def test_two_ranks_equal_one(tmp_path, kind, tags):
    scene = small_scene(tmp_path, tags)
    got, png = run_two_ranks(scene, tmp_path)
    want = single_process_frames(scene)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    # what rank 0 wrote decodes to the same frames
    for i, b in enumerate(want):
        img, n = capi.png_read(png, i)
        assert n == len(want) and np.array_equal(img, b)


def test_two_ranks_share_one_photon_pass(tmp_path):
    """A still whose photon pass is split over the ranks (each emits half of the root photons, the photon
    textures are summed with an all-reduce, then each rank traces its bands).  The sum of two partial
    sums is not the reference's sequential sum bit for bit: the picture agrees to float rounding."""
    scene = small_scene(tmp_path, "", steps=90)
    sc = capi.Scene(scene, 0.0)
    # the slices partition the pass: same deposits, and the summed textures are the whole pass's to rounding
    whole = oracle_lib.Photons(sc)
    halves = [oracle_lib.Photons(sc, part, 2) for part in range(2)]
    assert sum(h.stats.photon_deposits for h in halves) == whole.stats.photon_deposits > 0
    assert sum(h.stats.photon_segments for h in halves) == whole.stats.photon_segments
    total = halves[0].atlas() + halves[1].atlas()
    assert np.allclose(total, whole.atlas(), rtol=1e-5, atol=1e-7) and whole.atlas().max() > 0
    one = oracle_lib.Photons(sc, 0, 1)
    assert np.array_equal(one.atlas(), whole.atlas())
    # through the driver on two ranks
    got, _png = run_two_ranks(scene, tmp_path, RT_SPLIT_PHOTONS="1")
    want = single_process_frames(scene)
    assert len(got) == len(want) == 1
    d = oracle_lib.image_diff(got[0], want[0])
    assert d["max"] <= 1 and d["frac_gt0"] < 1e-3, d


def test_photon_split_is_opt_in(monkeypatch):
    """The default keeps a still's photon pass whole on every rank (N-GPU PNG == 1-GPU PNG); RT_SPLIT_PHOTONS=1
    shares passes that are large enough to pay."""
    big = capi.Scene(scene_path("examples3/103_caustic_animation.xml"), 0.0)
    small = capi.Scene(scene_path("examples3/103_caustic_steps200.xml"), 0.0)
    monkeypatch.delenv("RT_SPLIT_PHOTONS", raising=False)
    assert not driver.split_photon_pass(big, 8) and not driver.split_photon_pass(big, 1)
    monkeypatch.setenv("RT_SPLIT_PHOTONS", "1")
    assert driver.split_photon_pass(big, 8) and not driver.split_photon_pass(big, 1)
    assert small.desc.photon_steps < driver.SPLIT_PHOTON_MIN_STEPS
    assert not driver.split_photon_pass(small, 8)


@pytest.mark.parametrize("tags", ['<animation length="0.5" fps="6;12"/>', '<animation length="0.4" fps="5"/><motionblur subframes="2;4"/>'])
def test_pipelined_driver_equals_lock_step(tmp_path, tags):
    """render_units overlaps parse, render, gather and consume; the frames and the file must be those of the
    lock-step loop.  The first scene animates fps: each frame's delay is its own scene's (src/main.cpp:105)."""
    scene = small_scene(tmp_path, tags)
    a, png_a = run_two_ranks(scene, tmp_path)
    raw_a = open(png_a, "rb").read()
    b, png_b = run_two_ranks(scene, tmp_path, RT_DRIVER_SEQUENTIAL="1")
    assert np.array_equal(a, b) and raw_a == open(png_b, "rb").read()
    if "6;12" in tags:
        # fcTL chunks carry delay_num/delay_den = 1/fps of each frame's own scene
        import struct
        dens, pos = [], 8
        while pos < len(raw_a):
            n, kind = struct.unpack(">I4s", raw_a[pos:pos + 8])
            if kind == b"fcTL":
                dens.append(struct.unpack(">HH", raw_a[pos + 8 + 20:pos + 8 + 24]))
            pos += 12 + n
        info = capi.Scene(scene, 0.0).info
        want = [capi.Scene(scene, float(np.float32(f) / np.float32(info.frames - 1))).info.fps for f in range(info.frames)]
        assert len(dens) == info.frames and len(set(want)) > 1
        assert dens == [(int(np.float32(1000.0) / np.float32(x)), 1000) for x in want], (dens, want)
