// rt_amd <scene.xml> [<out.png>] — command line compatible with the reference executable
// (reference src/main.cpp:160-207), rendering through the HIP backend.
//
// The four render loops follow the reference's frame/sub-frame time formulas:
//   still image            t = <still time> or 0
//   still + motion blur    t = sub/(K-1)/frames + start,            acc += sub * (1/K)
//   animation              t = frame/(frames-1)
//   animation + blur       t = (frame + sub/(K-1))/frames,          K may change between frames
// The timer brackets the same work as the reference's (still image: starts after scene load and
// photon pass, includes the PNG encode; animations: everything).
#include <chrono>
#include <fstream>
#include <iostream>
#include <stdexcept>

#include "png_io.h"
#include "tracer_shim.h"

using namespace rta;

namespace {

using Clock = std::chrono::high_resolution_clock;

std::vector<uint8_t> quantized(const Picture& p) {
    std::vector<uint8_t> out(size_t(p.width()) * p.height() * 4);
    quantize_rgba(p.data(), uint64_t(p.width()) * p.height(), 1.0f, out.data());
    return out;
}

std::ofstream open_output(const std::string& name) {
    std::ofstream f(name, std::ios::binary);
    if (!f) throw std::runtime_error("output file could not be opened");
    return f;
}

Picture render_at(const std::string& file, float t, uint32_t* sub_frames_out = nullptr) {
    SceneModel scene = load_scene(file, t);
    if (scene.photon_steps > 0.0f) PhotonMapper::generate(scene);
    RayTracer raytracer;
    Picture p = raytracer.raytrace(scene);
    if (sub_frames_out) *sub_frames_out = scene.sub_frames;
    return p;
}

void finished(Clock::time_point begin) {
    const std::chrono::duration<double> dt = Clock::now() - begin;
    std::cout << "\nFinished in " << dt.count() << " s\n";
}

void render_image(const SceneModel& orig) {
    const float start = orig.still_time == kInfinite ? 0.0f : orig.still_time;
    SceneModel scene = load_scene(orig.scene_file, start);
    if (scene.photon_steps > 0.0f) {
        std::cout << "Generating photon map for caustics.. This will take some time.." << std::endl;
        PhotonMapper::generate(scene);
    }
    std::cout << "Rendering image.." << std::endl;
    const auto begin = Clock::now();
    RayTracer raytracer;
    const Picture picture = raytracer.raytrace(scene);
    std::cout << "Writing image to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    write_png(out, quantized(picture).data(), picture.width(), picture.height());
    finished(begin);
}

void render_image_motion_blur(const SceneModel& orig) {
    const float start = orig.still_time == kInfinite ? 0.0f : orig.still_time;
    const SceneModel probe = load_scene(orig.scene_file, start);
    const auto begin = Clock::now();
    const uint32_t k = probe.sub_frames;
    Picture acc(orig.camera.width, orig.camera.height);
    for (uint32_t sub = 0; sub < k; sub++) {
        std::cout << "Rendering image (subframe " << sub + 1 << " of " << k << ")          \r" << std::flush;
        const float t = static_cast<float>(sub) / (k - 1) / orig.frames + start;
        acc.mulAdd(render_at(orig.scene_file, t), 1.0f / k);
    }
    std::cout << std::endl << "Writing image to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    write_png(out, quantized(acc).data(), acc.width(), acc.height());
    finished(begin);
}

void render_video(const SceneModel& orig) {
    const auto begin = Clock::now();
    std::cout << "Writing animation to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    ApngWriter apng(out, orig.camera.width, orig.camera.height, orig.frames);
    for (uint32_t frame = 0; frame < orig.frames; frame++) {
        std::cout << "Rendering frame " << frame + 1 << " of " << orig.frames << "          \r" << std::flush;
        const float t = static_cast<float>(frame) / (orig.frames - 1);
        SceneModel scene = load_scene(orig.scene_file, t);
        if (scene.photon_steps > 0.0f) PhotonMapper::generate(scene);
        RayTracer raytracer;
        const Picture picture = raytracer.raytrace(scene);
        apng.frame(quantized(picture).data(), frame, scene.fps);
    }
    apng.end();
    finished(begin);
}

void render_video_motion_blur(const SceneModel& orig) {
    const auto begin = Clock::now();
    std::cout << "Writing animation to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    ApngWriter apng(out, orig.camera.width, orig.camera.height, orig.frames);
    uint32_t k = orig.sub_frames;
    for (uint32_t frame = 0; frame < orig.frames; frame++) {
        Picture acc(orig.camera.width, orig.camera.height);
        uint32_t next_k = k;   // the scene file may animate the sub-frame count; it applies from the next frame
        for (uint32_t sub = 0; sub < k; sub++) {
            std::cout << "Rendering frame " << frame + 1 << " of " << orig.frames << " (subframe " << sub + 1 << " of " << k
                      << ")          \r" << std::flush;
            const float t = (static_cast<float>(frame) + static_cast<float>(sub) / (k - 1)) / orig.frames;
            acc.mulAdd(render_at(orig.scene_file, t, &next_k), 1.0f / k);
        }
        apng.frame(quantized(acc).data(), frame, orig.fps);
        k = next_k;
    }
    apng.end();
    finished(begin);
}

}  // namespace

int main(int argc, char* argv[]) {
    if (argc < 2) {
        std::cout << "Usage: " << std::endl << argv[0] << " <scene.xml> [<out.png>]" << std::endl;
        return -1;
    }
    try {
        SceneModel scene = load_scene(argv[1]);
        if (argc >= 3) scene.output_file = argv[2];
        if (scene.dispersion_mode) std::cout << "Rendering with dispersion effect. This will increase rendering time." << std::endl;
        if (scene.camera.supersampling > 1) {
            std::cout << "Rendering with supersampling. This will increase rendering time." << std::endl;
        } else if (scene.camera.lens_size != 0.0f) {
            throw std::runtime_error("Depth of field needs supersampling.");
        }
        if (scene.sub_frames > 1) std::cout << "Rendering with motion blur. This will increase rendering time." << std::endl;
        if (scene.photon_steps > 0.0f) std::cout << "Rendering with caustics. This will increase rendering time." << std::endl;
        if (scene.frames > 1 && scene.still_time == kInfinite) {
            if (scene.sub_frames > 1) render_video_motion_blur(scene); else render_video(scene);
        } else {
            if (scene.sub_frames > 1) render_image_motion_blur(scene); else render_image(scene);
        }
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return -1;
    }
    return 0;
}
