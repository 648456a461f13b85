// Host-side mirror of the reference's operator interface for the trace path: the same two calls the
// reference's render loops make, with the same names, argument meaning and error behaviour
// (std::runtime_error), implemented on top of the C ABI of include/rt_hip.h.
//
//   reference src/raytracer.h:7-13   class RayTracer { Picture raytrace(const Scene&) const; }
//   reference src/photonmap.h:5-7    class PhotonMapper { static void generate(Scene&); }
//   reference src/types.h:358-389    class Picture (row-major RGBA float, mulAdd)
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

#include "scene_model.h"
#include "vecmath.h"

namespace rta {

// Row-major RGBA float image (≙ Picture)
class Picture {
public:
    Picture() = default;
    Picture(uint32_t width, uint32_t height) : width_(width), height_(height), data_(size_t(width) * height * 4, 0.0f) {}

    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    bool empty() const { return width_ == 0 || height_ == 0; }
    const float* data() const { return data_.data(); }
    float* data() { return data_.data(); }
    // data[i] += rhs[i] * factor, in element order (reference src/types.h:372-376)
    void mulAdd(const Picture& rhs, float factor) {
        for (size_t i = 0; i < data_.size(); i++) data_[i] += rhs.data_[i] * factor;
    }

private:
    uint32_t width_ = 0, height_ = 0;
    std::vector<float> data_;
};

// A picture as the writers consume it: RGBA8 after Color::scaleOut (reference src/types.h:274-282).
struct Picture8 {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> pixels;
};

// Color::scaleOut of a whole picture on the host (reference src/types.h:274-282, gain 1): u8(clamp(v, 0, 1) * 255.0f)
Picture8 scale_out(const Picture& picture);

// One GPU.  The reference's Scene carries its photon textures inside its objects; here they live on
// the device inside the session, so a Scene handed to PhotonMapper::generate must be rendered
// through the same session (the default session is process-wide, like the reference's free calls).
class GpuSession {
public:
    explicit GpuSession(int device = 0);
    ~GpuSession();
    GpuSession(const GpuSession&) = delete;
    GpuSession& operator=(const GpuSession&) = delete;

    void upload(const SceneModel& scene);      // rt_scene_upload
    void photons();                            // rt_photon_generate
    Picture render();                          // rt_render
    Picture8 render8();                        // rt_render_rgba8
    // rows of the bands only, compact (rt_bands: band b covers image rows [b*band_rows, (b+1)*band_rows))
    Picture8 render8(const rt_bands& bands);
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    void accumulate(float weight, bool first); // rt_render_accumulate
    Picture8 accumulated8();                   // rt_accumulated_rgba8
    static GpuSession& process_default();

private:
    void check(int rc, const char* what) const;
    rt_ctx* ctx_ = nullptr;
    uint32_t width_ = 0, height_ = 0;
    const SceneModel* uploaded_ = nullptr;
    friend class RayTracer;
    friend class PhotonMapper;
};

class PhotonMapper {
public:
    // ≙ static void PhotonMapper::generate(Scene&): runs the caustic pre-pass for `scene`; the
    // textures are consumed by the next RayTracer::raytrace of the same scene object.
    static void generate(SceneModel& scene);
};

class RayTracer {
public:
    // ≙ Picture RayTracer::raytrace(const Scene&) const
    Picture raytrace(const SceneModel& scene) const;
    // The same followed by what every caller in the reference does with the Picture (src/main.cpp:31,64,105,147),
    // done on the GPU so that the float picture never crosses the host link:
    //   raytrace8       raytrace + Color::scaleOut (writePNG / writeAPNGFrame quantise with gain 1)
    //   accumulate      acc.mulAdd(raytrace(scene), weight) with acc zeroed when `first`; acc stays on the device
    //   accumulated8    scaleOut of acc
    Picture8 raytrace8(const SceneModel& scene) const;
    void accumulate(const SceneModel& scene, float weight, bool first) const;
    Picture8 accumulated8() const;
};

}  // namespace rta
