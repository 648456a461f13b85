// In-memory scene as the trace path consumes it: flat arrays in the layout of rt_scene_desc
// (include/rt_hip.h) plus the driver-level settings the four render loops need.
//
// The reference keeps one fat Object (variant + Material copy + texture copy) per primitive
// (reference src/objects.h:104-135); here geometry is split by kind, materials and textures are
// shared by index, and `objects` keeps the scene order the hit rules depend on.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"

namespace rta {

struct SceneModel {
    // driver-level settings (reference src/scene.h:37-43)
    std::string scene_file;
    std::string output_file;
    uint32_t threads = 8;          // parsed and ignored by the HIP backend
    float still_time = 3.402823466e+38f;   // INFINITE = "no <still>"
    uint32_t frames = 1;
    float fps = 25.0f;
    uint32_t sub_frames = 1;

    rt_camera camera{};
    float background[4] = {0, 0, 0, 0};
    float ambient[4] = {0, 0, 0, 0};
    std::vector<rt_light> lights;
    std::vector<rt_object> objects;
    std::vector<rt_sphere> spheres;
    std::vector<rt_triangle> triangles;
    std::vector<rt_julia> julias;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<uint8_t> texels;   // RGBA8 pool
    bool dispersion_mode = false;
    float photon_steps = 0.0f;
    uint32_t photon_texture_size = 0;
    float photon_factor = 0.0f;
    int hue_preset = RT_HUE_INT_ABS;

    // Borrowing view for the C ABI; valid while *this is alive and unmodified.
    rt_scene_desc desc() const;
};

// Parses `path` at animation time `t` in [0,1] (≙ Scene::load, reference src/scene.cpp:8-16).
// Throws std::runtime_error with the reference's message shapes.
SceneModel load_scene(const std::string& path, float t = 0.0f);

}  // namespace rta
