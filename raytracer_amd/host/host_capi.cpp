// C exports of the host library (libraytracer_host.so): scene loading, keyframes, PNG/APNG I/O.
// Used by the Python tests/bench through ctypes and by tools; no GPU dependency.
#include <algorithm>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "png_io.h"
#include "scene_model.h"

namespace rta {
float evaluate_keyframes_for_test(const std::string& text, float t);
}

namespace {

void set_error(char* err, size_t err_len, const std::string& what) {
    if (err == nullptr || err_len == 0) return;
    std::strncpy(err, what.c_str(), err_len - 1);
    err[err_len - 1] = '\0';
}

struct ApngHandle {
    std::ofstream file;
    std::unique_ptr<rta::ApngWriter> writer;
};

}  // namespace

extern "C" {

// Driver-level settings that are not part of rt_scene_desc (reference src/scene.h:14-20)
struct rth_scene_info {
    uint32_t frames;
    float fps;
    uint32_t sub_frames;
    float still_time;       // FLT_MAX when the scene has no <still>
    uint32_t threads;
    char output_file[512];
};

void* rth_scene_load(const char* path, float t, int hue_preset, char* err, size_t err_len) {
    try {
        auto scene = std::make_unique<rta::SceneModel>(rta::load_scene(path, t));
        scene->hue_preset = hue_preset;
        return scene.release();
    } catch (const std::exception& e) {
        set_error(err, err_len, e.what());
        return nullptr;
    }
}

void rth_scene_free(void* scene) { delete static_cast<rta::SceneModel*>(scene); }

// Fills *out with a view borrowing the scene's arrays (valid until rth_scene_free).
void rth_scene_desc(const void* scene, rt_scene_desc* out) {
    *out = static_cast<const rta::SceneModel*>(scene)->desc();
}

void rth_scene_get_info(const void* scene, rth_scene_info* out) {
    const auto* s = static_cast<const rta::SceneModel*>(scene);
    out->frames = s->frames;
    out->fps = s->fps;
    out->sub_frames = s->sub_frames;
    out->still_time = s->still_time;
    out->threads = s->threads;
    std::strncpy(out->output_file, s->output_file.c_str(), sizeof out->output_file - 1);
    out->output_file[sizeof out->output_file - 1] = '\0';
}

// Test/bench overrides of the integer camera attributes (the derived-XML route is preferred; this
// exists so that scaled workloads can be built without writing files).
void rth_scene_set_camera(void* scene, uint32_t width, uint32_t height, uint32_t supersampling) {
    auto* s = static_cast<rta::SceneModel*>(scene);
    if (width) s->camera.width = width;
    if (height) s->camera.height = height;
    if (supersampling) s->camera.supersampling = supersampling;
}

int rth_keyframe_eval(const char* text, float t, float* out, char* err, size_t err_len) {
    try {
        *out = rta::evaluate_keyframes_for_test(text, t);
        return 0;
    } catch (const std::exception& e) {
        set_error(err, err_len, e.what());
        return -1;
    }
}

void rth_quantize(const float* rgba, uint64_t n_pixels, float gain, uint8_t* out) {
    rta::quantize_rgba(rgba, n_pixels, gain, out);
}

int rth_png_write(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height, char* err,
                  size_t err_len) {
    try {
        std::ofstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("output file could not be opened");
        rta::write_png(f, rgba8, width, height);
        return 0;
    } catch (const std::exception& e) {
        set_error(err, err_len, e.what());
        return -1;
    }
}

void* rth_apng_begin(const char* path, uint32_t width, uint32_t height, uint32_t frames, char* err,
                     size_t err_len) {
    try {
        auto h = std::make_unique<ApngHandle>();
        h->file.open(path, std::ios::binary);
        if (!h->file) throw std::runtime_error("output file could not be opened");
        h->writer = std::make_unique<rta::ApngWriter>(h->file, width, height, frames);
        return h.release();
    } catch (const std::exception& e) {
        set_error(err, err_len, e.what());
        return nullptr;
    }
}

int rth_apng_frame(void* handle, const uint8_t* rgba8, uint32_t frame_index, float fps) {
    try {
        static_cast<ApngHandle*>(handle)->writer->frame(rgba8, frame_index, fps);
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

void rth_apng_end(void* handle) {
    auto* h = static_cast<ApngHandle*>(handle);
    h->writer->end();
    delete h;
}

// Decodes frame `frame_index` of a PNG/APNG file.  Call with out == NULL to query the size and the
// frame count; returns 0 on success.
int rth_png_read(const char* path, uint32_t frame_index, uint32_t* width, uint32_t* height,
                 uint32_t* frame_count, uint8_t* out, char* err, size_t err_len) {
    try {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("file could not be opened");
        const std::vector<rta::ImageRGBA8> frames = rta::read_apng_frames(f);
        if (frames.empty()) throw std::runtime_error("no image data");
        if (frame_index >= frames.size()) throw std::runtime_error("frame index out of range");
        const rta::ImageRGBA8& img = frames[frame_index];
        *width = img.width;
        *height = img.height;
        *frame_count = uint32_t(frames.size());
        if (out != nullptr) std::memcpy(out, img.pixels.data(), img.pixels.size());
        return 0;
    } catch (const std::exception& e) {
        set_error(err, err_len, e.what());
        return -1;
    }
}

// Canonical word stream of everything the trace path reads, in the order oracle/ref_harness.cpp
// writes it from the reference's own Scene — the loader parity test compares the two bit for bit.
// Returns the number of 32-bit words (call with out == NULL to size the buffer).
uint64_t rth_scene_fingerprint(const void* scene_ptr, uint32_t* out, uint64_t cap) {
    const auto* s = static_cast<const rta::SceneModel*>(scene_ptr);
    std::vector<uint32_t> w;
    auto u = [&](uint32_t v) { w.push_back(v); };
    auto f = [&](float v) {
        uint32_t b;
        std::memcpy(&b, &v, 4);
        w.push_back(b);
    };
    auto fv = [&](const float* p, int n) { for (int i = 0; i < n; i++) f(p[i]); };
    const rt_camera& cam = s->camera;
    u(cam.width); u(cam.height); u(cam.supersampling); u(cam.max_bounces);
    f(cam.fov); f(cam.focus_distance); f(cam.lens_size);
    fv(cam.camera_to_world.m, 12);
    fv(s->background, 4); fv(s->ambient, 4);
    u(uint32_t(s->lights.size()));
    for (const rt_light& l : s->lights) { u(uint32_t(l.type)); fv(l.position, 3); fv(l.power, 4); }
    u(s->dispersion_mode ? 1 : 0); f(s->photon_steps); u(s->photon_texture_size); f(s->photon_factor);
    u(s->frames); f(s->fps); u(s->sub_frames); f(s->still_time); u(s->threads);
    u(uint32_t(s->objects.size()));
    for (const rt_object& o : s->objects) {
        u(uint32_t(o.kind));
        if (o.kind == RT_OBJ_SPHERE) {
            const rt_sphere& sp = s->spheres[o.index];
            fv(sp.center, 3); f(sp.radius);
            fv(sp.world_to_object.m, 12); fv(sp.object_to_world.m, 12); fv(sp.object_to_world_normals.m, 12);
        } else if (o.kind == RT_OBJ_TRIANGLE) {
            const rt_triangle& t = s->triangles[o.index];
            for (int k = 0; k < 3; k++) { fv(t.position[k], 3); fv(t.normal[k], 3); fv(t.uv[k], 2); }
        } else {
            const rt_julia& j = s->julias[o.index];
            fv(j.position, 3); f(j.scale); fv(j.c, 4); f(j.cut_plane);
            fv(j.world_to_object.m, 12); fv(j.object_to_world.m, 12); fv(j.object_to_world_normals.m, 12);
        }
        const rt_material& m = s->materials[o.material];
        fv(m.color, 4);
        if (m.texture < 0) {
            u(0); u(0); u(0); u(0);
        } else {
            const rt_texture& t = s->textures[m.texture];
            uint32_t h = 2166136261u;   // FNV-1a over the RGBA8 texels
            const uint8_t* p = s->texels.data() + t.offset;
            for (size_t i = 0; i < size_t(t.width) * t.height * 4; i++) h = (h ^ p[i]) * 16777619u;
            u(1); u(t.width); u(t.height); u(h);
        }
        f(m.ka); f(m.kd); f(m.ks); f(m.exponent); f(m.reflectance); f(m.transmittance);
        f(m.refraction_re); f(m.refraction_im); f(m.dispersion);
    }
    if (out != nullptr) std::memcpy(out, w.data(), std::min<uint64_t>(cap, w.size()) * 4);
    return w.size();
}

}  // extern "C"
