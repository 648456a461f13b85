"""The C-ABI library loads and exports every symbol include/rt_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT
from raytracer_amd import capi


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("rt_create", "rt_destroy", "rt_scene_upload", "rt_render", "rt_photon_generate", "rt_last_error",
                 "rt_render_device", "rt_mul_add_device", "rt_quantize_device", "rt_photon_download"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(capi.HIP_LIB_PATH), "librt_hip.so must be built in-tree (make hip)"
    lib = ctypes.CDLL(capi.HIP_LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_sizes_match_the_header():
    # sizes computed from the C declarations (natural alignment)
    assert ctypes.sizeof(capi.Mat34) == 48
    assert ctypes.sizeof(capi.Camera) == 48 + 7 * 4
    assert ctypes.sizeof(capi.Light) == 32
    assert ctypes.sizeof(capi.Material) == 56
    assert ctypes.sizeof(capi.Texture) == 16
    assert ctypes.sizeof(capi.Sphere) == 16 + 3 * 48
    assert ctypes.sizeof(capi.Triangle) == 96
    assert ctypes.sizeof(capi.Julia) == 36 + 3 * 48
    assert ctypes.sizeof(capi.Object) == 12
    assert ctypes.sizeof(capi.Stats) == 16 * 8 + 8 + 8 + 8      # 16 counters, kernel_ms, shadow_rays_elided, photon_roots_culled


def test_pure_host_entry_points():
    lib = capi.hip_lib()
    table = ((ctypes.c_float * 3) * 8)()
    assert lib.rt_hue_table(capi.RT_HUE_INT_ABS, table) == 0
    assert [list(r) for r in table][1] == [1.0, 1.0, 0.0]
    assert lib.rt_hue_table(7, table) != 0
    b = capi.Bands(8, 1, 4)
    # 600 rows in bands of 8: bands 1,5,9,... -> 19 bands of 8 rows = 152 rows
    assert lib.rt_bands_pixel_count(960, 600, ctypes.byref(b)) == 960 * 152
    assert lib.rt_bands_pixel_count(960, 600, None) == 960 * 600
    total = sum(lib.rt_bands_pixel_count(960, 600, ctypes.byref(capi.Bands(8, r, 4))) for r in range(4))
    assert total == 960 * 600
