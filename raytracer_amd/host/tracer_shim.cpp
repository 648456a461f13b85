#include "tracer_shim.h"

#include <string>

namespace rta {

GpuSession::GpuSession(int device) {
    const int rc = rt_create(&ctx_, device);
    if (rc != RT_OK) throw std::runtime_error(std::string("GPU backend: ") + rt_last_error(nullptr));
}

GpuSession::~GpuSession() { rt_destroy(ctx_); }

void GpuSession::check(int rc, const char* what) const {
    if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(ctx_));
}

void GpuSession::upload(const SceneModel& scene) {
    const rt_scene_desc d = scene.desc();
    check(rt_scene_upload(ctx_, &d), "scene upload");
    width_ = scene.camera.width;
    height_ = scene.camera.height;
    uploaded_ = &scene;
}

void GpuSession::photons() { check(rt_photon_generate(ctx_, nullptr), "photon map"); }

Picture GpuSession::render() {
    Picture p(width_, height_);
    check(rt_render(ctx_, nullptr, p.data(), nullptr), "raytrace");
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

}  // namespace rta
