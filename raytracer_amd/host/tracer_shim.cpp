#include "tracer_shim.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace rta {

namespace {
// RT_AMD_TIMING=1: seconds spent inside each C-ABI call (stderr at exit; developer aid)
struct CallTimes {
    double create = 0, upload = 0, photons = 0, render = 0;
    bool on = std::getenv("RT_AMD_TIMING") != nullptr;
    ~CallTimes() {
        if (on) std::fprintf(stderr, "timing (C ABI): rt_create %.4f s, rt_scene_upload %.4f s, rt_photon_generate %.4f s, rt_render* %.4f s\n", create, upload, photons, render);
    }
} g_calls;
struct CallLap {
    double& into;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit CallLap(double& d) : into(d) {}
    ~CallLap() { into += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

GpuSession::GpuSession(int device) {
    CallLap lap(g_calls.create);
    const int rc = rt_create(&ctx_, device);
    if (rc != RT_OK) throw std::runtime_error(std::string("GPU backend: ") + rt_last_error(nullptr));
}

GpuSession::~GpuSession() { rt_destroy(ctx_); }

void GpuSession::check(int rc, const char* what) const {
    if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(ctx_));
}

void GpuSession::upload(const SceneModel& scene) {
    const rt_scene_desc d = scene.desc();
    {
        CallLap lap(g_calls.upload);
        check(rt_scene_upload(ctx_, &d), "scene upload");
    }
    width_ = scene.camera.width;
    height_ = scene.camera.height;
    uploaded_ = &scene;
}

void GpuSession::photons() {
    CallLap lap(g_calls.photons);
    check(rt_photon_generate(ctx_, nullptr), "photon map");
}

Picture GpuSession::render() {
    Picture p(width_, height_);
    CallLap lap(g_calls.render);
    check(rt_render(ctx_, nullptr, p.data(), nullptr), "raytrace");
    return p;
}

Picture8 GpuSession::render8() {
    Picture8 p;
    p.width = width_;
    p.height = height_;
    p.pixels.resize(size_t(width_) * height_ * 4);
    CallLap lap(g_calls.render);
    check(rt_render_rgba8(ctx_, nullptr, p.pixels.data(), 1.0f, nullptr), "raytrace");
    return p;
}

Picture8 GpuSession::render8(const rt_bands& bands) {
    Picture8 p;
    p.width = width_;
    const uint64_t n = rt_bands_pixel_count(width_, height_, &bands);
    p.height = width_ ? static_cast<uint32_t>(n / width_) : 0;
    p.pixels.resize(size_t(n) * 4);
    if (n == 0) return p;
    CallLap lap(g_calls.render);
    check(rt_render_rgba8(ctx_, &bands, p.pixels.data(), 1.0f, nullptr), "raytrace");
    return p;
}

Picture8 scale_out(const Picture& picture) {
    Picture8 p;
    p.width = picture.width();
    p.height = picture.height();
    const size_t n = size_t(p.width) * p.height * 4;
    p.pixels.resize(n);
    const float* v = picture.data();
    for (size_t i = 0; i < n; i++) {
        const float c = v[i] < 0.0f ? 0.0f : (v[i] > 1.0f ? 1.0f : v[i]);
        const float s = c * 255.0f;
        p.pixels[i] = static_cast<uint8_t>(s != s ? 0.0f : s);   // truncation; NaN -> 0
    }
    return p;
}

void GpuSession::accumulate(float weight, bool first) {
    CallLap lap(g_calls.render);
    check(rt_render_accumulate(ctx_, nullptr, weight, first ? 1 : 0), "raytrace");
}

Picture8 GpuSession::accumulated8() {
    Picture8 p;
    p.width = width_;
    p.height = height_;
    p.pixels.resize(size_t(width_) * height_ * 4);
    check(rt_accumulated_rgba8(ctx_, nullptr, p.pixels.data(), 1.0f), "raytrace");
    return p;
}

GpuSession& GpuSession::process_default() {
    static GpuSession session(0);
    return session;
}

void PhotonMapper::generate(SceneModel& scene) {
    GpuSession& s = GpuSession::process_default();
    s.upload(scene);
    s.photons();
}

Picture RayTracer::raytrace(const SceneModel& scene) const {
    GpuSession& s = GpuSession::process_default();
    // a scene whose photon pass just ran is already resident (with its photon textures)
    if (s.uploaded_ != &scene) s.upload(scene);
    Picture p = s.render();
    s.uploaded_ = nullptr;   // the reference reloads the scene for every frame; never reuse stale uploads
    return p;
}

Picture8 RayTracer::raytrace8(const SceneModel& scene) const {
    GpuSession& s = GpuSession::process_default();
    if (s.uploaded_ != &scene) s.upload(scene);
    Picture8 p = s.render8();
    s.uploaded_ = nullptr;
    return p;
}

void RayTracer::accumulate(const SceneModel& scene, float weight, bool first) const {
    GpuSession& s = GpuSession::process_default();
    if (s.uploaded_ != &scene) s.upload(scene);
    s.accumulate(weight, first);
    s.uploaded_ = nullptr;
}

Picture8 RayTracer::accumulated8() const { return GpuSession::process_default().accumulated8(); }

}  // namespace rta
