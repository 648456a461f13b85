"""ctypes view of the C ABI (include/rt_hip.h) and of the host library.

Plumbing only: the structs mirror the header field for field, the classes below wrap the entry
points one to one.  The HIP library is loaded lazily and loading FAILS LOUDLY if it is missing —
there is no CPU fallback in the product path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libraytracer_host.so")
HIP_LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")   # override = developer A/B builds

RT_ABI_VERSION = 1
RT_OBJ_SPHERE, RT_OBJ_TRIANGLE, RT_OBJ_JULIA = 0, 1, 2
RT_LIGHT_PARALLEL, RT_LIGHT_POINT = 0, 1
RT_HUE_INT_ABS, RT_HUE_FLOAT_ABS = 0, 1
RT_OPT_COLLECT_STATS, RT_OPT_BVH_MIN_TRIANGLES, RT_OPT_DOF_SEED, RT_OPT_PHOTON_ORDER, RT_OPT_PHOTON_RECORDS = 1, 2, 3, 4, 5
RT_OPT_DEBUG_FORCE_ABORT = 6
RT_OPT_ELIDE_UNLIT_SHADOWS = 7
RT_OPT_CULL_PHOTON_ROOTS = 8
RT_OPT_SHARED_GPU = 9
RT_DEFAULT_BVH_MIN_TRIANGLES = 48
FLT_MAX = float(np.finfo(np.float32).max)


class Mat34(C.Structure):
    _fields_ = [("m", C.c_float * 12)]


class Camera(C.Structure):
    _fields_ = [("camera_to_world", Mat34), ("fov", C.c_float), ("width", C.c_uint32),
                ("height", C.c_uint32), ("supersampling", C.c_uint32), ("max_bounces", C.c_uint32),
                ("focus_distance", C.c_float), ("lens_size", C.c_float)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("position", C.c_float * 3), ("power", C.c_float * 4)]


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 4), ("texture", C.c_int32), ("ka", C.c_float), ("kd", C.c_float),
                ("ks", C.c_float), ("exponent", C.c_float), ("reflectance", C.c_float),
                ("transmittance", C.c_float), ("refraction_re", C.c_float), ("refraction_im", C.c_float),
                ("dispersion", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("offset", C.c_uint64)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("world_to_object", Mat34),
                ("object_to_world", Mat34), ("object_to_world_normals", Mat34)]


class Triangle(C.Structure):
    _fields_ = [("position", (C.c_float * 3) * 3), ("normal", (C.c_float * 3) * 3), ("uv", (C.c_float * 2) * 3)]


class Julia(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("scale", C.c_float), ("c", C.c_float * 4), ("cut_plane", C.c_float),
                ("world_to_object", Mat34), ("object_to_world", Mat34), ("object_to_world_normals", Mat34)]


class Object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_uint32), ("material", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("camera", Camera), ("background", C.c_float * 4),
                ("ambient", C.c_float * 4),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(Light)),
                ("n_objects", C.c_uint32), ("objects", C.POINTER(Object)),
                ("n_spheres", C.c_uint32), ("spheres", C.POINTER(Sphere)),
                ("n_triangles", C.c_uint32), ("triangles", C.POINTER(Triangle)),
                ("n_julias", C.c_uint32), ("julias", C.POINTER(Julia)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
                ("n_texel_bytes", C.c_uint64), ("texels", C.POINTER(C.c_uint8)),
                ("dispersion_mode", C.c_int32), ("hue_rgb", (C.c_float * 3) * 8),
                ("photon_steps", C.c_float), ("photon_texture_size", C.c_uint32), ("photon_factor", C.c_float)]


class Bands(C.Structure):
    _fields_ = [("band_rows", C.c_uint32), ("first_band", C.c_uint32), ("band_stride", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "primary_rays", "secondary_rays", "shadow_rays", "photon_segments", "photon_deposits", "sphere_tests",
        "triangle_tests", "julia_queries", "julia_steps", "julia_de_iterations", "julia_normals", "shaded_hits",
        "phong_lights", "fresnel_evals", "texture_lookups", "bvh_node_visits")] + [("kernel_ms", C.c_double), ("shadow_rays_elided", C.c_uint64), ("photon_roots_culled", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}

    def total_rays(self):
        return self.primary_rays + self.secondary_rays + self.shadow_rays + self.photon_segments


class SceneInfo(C.Structure):
    _fields_ = [("frames", C.c_uint32), ("fps", C.c_float), ("sub_frames", C.c_uint32), ("still_time", C.c_float),
                ("threads", C.c_uint32), ("output_file", C.c_char * 512)]


# ------------------------------------------------------------------------------ host library

_host = None


def host_lib():
    global _host
    if _host is None:
        lib = C.CDLL(HOST_LIB_PATH)
        lib.rth_scene_load.restype = C.c_void_p
        lib.rth_scene_load.argtypes = [C.c_char_p, C.c_float, C.c_int, C.c_char_p, C.c_size_t]
        lib.rth_scene_free.argtypes = [C.c_void_p]
        lib.rth_scene_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        lib.rth_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
        lib.rth_scene_set_camera.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.rth_keyframe_eval.argtypes = [C.c_char_p, C.c_float, C.POINTER(C.c_float), C.c_char_p, C.c_size_t]
        lib.rth_quantize.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]
        lib.rth_png_write.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
        lib.rth_apng_begin.restype = C.c_void_p
        lib.rth_apng_begin.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
        lib.rth_apng_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float]
        lib.rth_apng_end.argtypes = [C.c_void_p]
        lib.rth_png_read.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32), C.c_void_p, C.c_char_p, C.c_size_t]
        lib.rth_scene_fingerprint.restype = C.c_uint64
        lib.rth_scene_fingerprint.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        _host = lib
    return _host


class SceneError(RuntimeError):
    pass


class Scene:
    """A parsed scene file at animation time t (≙ Scene::load, reference src/scene.cpp:8-16)."""

    def __init__(self, path, t=0.0, hue_preset=RT_HUE_INT_ABS):
        err = C.create_string_buffer(1024)
        self._h = host_lib().rth_scene_load(os.fsencode(path), t, hue_preset, err, len(err))
        if not self._h:
            raise SceneError(err.value.decode())
        self.path = path
        self.t = t
        self._refresh()

    def _refresh(self):
        self.desc = SceneDesc()
        host_lib().rth_scene_desc(self._h, C.byref(self.desc))
        self.info = SceneInfo()
        host_lib().rth_scene_get_info(self._h, C.byref(self.info))

    def set_camera(self, width=0, height=0, supersampling=0):
        host_lib().rth_scene_set_camera(self._h, width, height, supersampling)
        self._refresh()

    @property
    def width(self):
        return self.desc.camera.width

    @property
    def height(self):
        return self.desc.camera.height

    def fingerprint(self):
        n = host_lib().rth_scene_fingerprint(self._h, None, 0)
        out = np.zeros(n, dtype=np.uint32)
        host_lib().rth_scene_fingerprint(self._h, out.ctypes.data, n)
        return out

    def close(self):
        if self._h:
            host_lib().rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def keyframe_eval(text, t):
    out = C.c_float()
    err = C.create_string_buffer(256)
    if host_lib().rth_keyframe_eval(text.encode(), t, C.byref(out), err, len(err)) != 0:
        raise SceneError(err.value.decode())
    return out.value


def quantize(rgba, gain=1.0):
    """Color::scaleOut over an (..., 4) float32 array -> uint8 (reference src/types.h:274-282)."""
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.empty(rgba.shape, dtype=np.uint8)
    host_lib().rth_quantize(rgba.ctypes.data, rgba.size // 4, gain, out.ctypes.data)
    return out


def png_write(path, rgba8):
    rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w = rgba8.shape[:2]
    err = C.create_string_buffer(256)
    if host_lib().rth_png_write(os.fsencode(path), rgba8.ctypes.data, w, h, err, len(err)) != 0:
        raise SceneError(err.value.decode())


def apng_write(path, frames_rgba8, fps):
    frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames_rgba8]
    h, w = frames[0].shape[:2]
    err = C.create_string_buffer(256)
    handle = host_lib().rth_apng_begin(os.fsencode(path), w, h, len(frames), err, len(err))
    if not handle:
        raise SceneError(err.value.decode())
    fps_of = fps if isinstance(fps, (list, tuple)) else [fps] * len(frames)
    for i, f in enumerate(frames):
        host_lib().rth_apng_frame(handle, f.ctypes.data, i, fps_of[i])
    host_lib().rth_apng_end(handle)


def png_read(path, frame=0):
    """Returns (rgba8 array HxWx4, frame_count)."""
    w, h, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
    err = C.create_string_buffer(256)
    lib = host_lib()
    if lib.rth_png_read(os.fsencode(path), frame, C.byref(w), C.byref(h), C.byref(n), None, err, len(err)) != 0:
        raise SceneError(err.value.decode())
    out = np.empty((h.value, w.value, 4), dtype=np.uint8)
    lib.rth_png_read(os.fsencode(path), frame, C.byref(w), C.byref(h), C.byref(n), out.ctypes.data, err, len(err))
    return out, n.value


# ------------------------------------------------------------------------------ HIP library

_hip = None


class RtError(RuntimeError):
    pass


def hip_lib():
    """Loads librt_hip.so; raises if it (or the HIP runtime) is missing — never falls back."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RtError("librt_hip.so is not built: run `make hip` (no CPU fallback exists)")
        try:
            # torch bundles its own libamdhip64: load it first so the process holds ONE HIP runtime
            # (librt_hip.so then binds to it; two runtimes in one process cannot both see the GPU)
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(HIP_LIB_PATH)
        lib.rt_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        lib.rt_destroy.argtypes = [C.c_void_p]
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_last_error.argtypes = [C.c_void_p]
        lib.rt_scene_upload.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        lib.rt_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        lib.rt_photon_generate.argtypes = [C.c_void_p, C.POINTER(Stats)]
        lib.rt_render.argtypes = [C.c_void_p, C.POINTER(Bands), C.c_void_p, C.POINTER(Stats)]
        lib.rt_render_device.argtypes = [C.c_void_p, C.POINTER(Bands), C.c_void_p, C.c_void_p]
        lib.rt_render_device_noalloc.argtypes = [C.c_void_p, C.POINTER(Bands), C.c_void_p, C.c_void_p]
        lib.rt_render_rgba8_device.argtypes = [C.c_void_p, C.POINTER(Bands), C.c_float, C.c_void_p, C.c_void_p]
        lib.rt_photon_generate_device.argtypes = [C.c_void_p, C.c_void_p]
        lib.rt_photon_generate_part_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.rt_photon_atlas_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        lib.rt_photon_ordered.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.rt_fetch_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        lib.rt_launch_status.argtypes = [C.c_void_p]
        lib.rt_photon_download.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.rt_mul_add_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint64, C.c_void_p]
        lib.rt_quantize_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint64, C.c_void_p]
        lib.rt_bands_pixel_count.restype = C.c_uint64
        lib.rt_bands_pixel_count.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(Bands)]
        lib.rt_hue_table.argtypes = [C.c_int, C.c_void_p]
        _hip = lib
    return _hip


class Context:
    """One GPU's trace context (rt_ctx)."""

    def __init__(self, device=0):
        self._lib = hip_lib()
        self._ctx = C.c_void_p()
        rc = self._lib.rt_create(C.byref(self._ctx), device)
        if rc != 0:
            raise RtError("rt_create failed (%d): %s" % (rc, self._lib.rt_last_error(None).decode()))
        self.scene = None
        self.ordered_photons = True      # RT_OPT_PHOTON_ORDER as last set through set_option
        self.options = {}                # every option set through set_option, in order (driver.HipRenderer.lanes replays them)

    def _check(self, rc, what):
        if rc != 0:
            raise RtError("%s failed (%d): %s" % (what, rc, self._lib.rt_last_error(self._ctx).decode()))

    def upload(self, scene):
        self._check(self._lib.rt_scene_upload(self._ctx, C.byref(scene.desc)), "rt_scene_upload")
        self.scene = scene

    def set_option(self, option, value):
        self._check(self._lib.rt_set_option(self._ctx, option, value), "rt_set_option")
        self.options.pop(option, None)
        self.options[option] = value
        if option == RT_OPT_PHOTON_ORDER:
            self.ordered_photons = bool(value)

    def photon_generate(self):
        st = Stats()
        self._check(self._lib.rt_photon_generate(self._ctx, C.byref(st)), "rt_photon_generate")
        return st

    def pixel_count(self, bands=None):
        cam = self.scene.desc.camera
        return self._lib.rt_bands_pixel_count(cam.width, cam.height, C.byref(bands) if bands else None)

    def render(self, bands=None):
        """≙ RayTracer::raytrace: returns (float32 array [pixels,4] or [H,W,4], Stats)."""
        n = self.pixel_count(bands)
        out = np.empty((n, 4), dtype=np.float32)
        st = Stats()
        self._check(self._lib.rt_render(self._ctx, C.byref(bands) if bands else None, out.ctypes.data, C.byref(st)),
                    "rt_render")
        if bands is None:
            cam = self.scene.desc.camera
            out = out.reshape(cam.height, cam.width, 4)
        return out, st

    def render_device(self, d_out_ptr, stream=None, bands=None):
        self._check(self._lib.rt_render_device(self._ctx, C.byref(bands) if bands else None, C.c_void_p(d_out_ptr),
                                               C.c_void_p(stream or 0)), "rt_render_device")

    def render_rgba8_device(self, d_out8_ptr, stream=None, bands=None, gain=1.0):
        """rt_render_device + rt_quantize_device in one: the RGBA8 picture only (device memory, 4 bytes per pixel)."""
        self._check(self._lib.rt_render_rgba8_device(self._ctx, C.byref(bands) if bands else None, gain, C.c_void_p(d_out8_ptr),
                                                     C.c_void_p(stream or 0)), "rt_render_rgba8_device")

    def render_device_noalloc(self, d_out_ptr, stream=None, bands=None):
        self._check(self._lib.rt_render_device_noalloc(self._ctx, C.byref(bands) if bands else None, C.c_void_p(d_out_ptr),
                                                       C.c_void_p(stream or 0)), "rt_render_device_noalloc")

    def photon_generate_device(self, stream=None):
        self._check(self._lib.rt_photon_generate_device(self._ctx, C.c_void_p(stream or 0)), "rt_photon_generate_device")

    def photon_generate_part_device(self, part, n_parts, stream=None):
        self._check(self._lib.rt_photon_generate_part_device(self._ctx, C.c_void_p(stream or 0), part, n_parts),
                    "rt_photon_generate_part_device")

    def photon_ordered(self):
        """Waits for the last photon pass; True if all of its deposits were summed in the reference's order."""
        flag = C.c_int()
        self._check(self._lib.rt_photon_ordered(self._ctx, C.byref(flag)), "rt_photon_ordered")
        return bool(flag.value)

    def photon_atlas_device(self):
        """(device pointer, number of floats) of all photon textures of the uploaded scene; (0, 0) without receivers."""
        ptr, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.rt_photon_atlas_device(self._ctx, C.byref(ptr), C.byref(n)), "rt_photon_atlas_device")
        return ptr.value or 0, n.value

    def fetch_stats(self):
        st = Stats()
        self._check(self._lib.rt_fetch_stats(self._ctx, C.byref(st)), "rt_fetch_stats")
        return st

    def launch_status(self):
        """Waits for the last render_device launch; raises if the Julia kernel's watchdog abandoned it."""
        self._check(self._lib.rt_launch_status(self._ctx), "rt_launch_status")

    def photon_download(self, object_index):
        size = self.scene.desc.photon_texture_size
        out = np.zeros((size, size, 4), dtype=np.float32)
        self._check(self._lib.rt_photon_download(self._ctx, object_index, out.ctypes.data), "rt_photon_download")
        return out

    def mul_add_device(self, d_acc, d_src, factor, n_floats, stream=None):
        self._check(self._lib.rt_mul_add_device(self._ctx, C.c_void_p(d_acc), C.c_void_p(d_src), factor, n_floats,
                                                C.c_void_p(stream or 0)), "rt_mul_add_device")

    def quantize_device(self, d_rgba, d_rgba8, gain, n_pixels, stream=None):
        self._check(self._lib.rt_quantize_device(self._ctx, C.c_void_p(d_rgba), C.c_void_p(d_rgba8), gain, n_pixels,
                                                 C.c_void_p(stream or 0)), "rt_quantize_device")

    def close(self):
        if self._ctx:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
