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
#include <cstdlib>
#include <deque>
#include <fstream>
#include <future>
#include <iostream>
#include <stdexcept>

#include "png_io.h"
#include "tracer_shim.h"

using namespace rta;

namespace {

using Clock = std::chrono::high_resolution_clock;

std::ofstream open_output(const std::string& name) {
    std::ofstream f(name, std::ios::binary);
    if (!f) throw std::runtime_error("output file could not be opened");
    return f;
}

// RT_AMD_TIMING=1: where the wall time of a run goes (stderr; developer aid, DESIGN.md quotes it)
struct Phases {
    double load = 0, photon = 0, render = 0, encode_wait = 0;
    bool on = std::getenv("RT_AMD_TIMING") != nullptr;
    ~Phases() {
        if (on) std::cerr << "timing: load " << load << " s, photon pass " << photon << " s, render + quantise + copy " << render
                          << " s, waiting for the encoder " << encode_wait << " s\n";
    }
} g_phases;
struct Lap {
    double& into;
    Clock::time_point t0 = Clock::now();
    explicit Lap(double& d) : into(d) {}
    ~Lap() { into += std::chrono::duration<double>(Clock::now() - t0).count(); }
};

SceneModel load_timed(const std::string& file, float t) {
    Lap lap(g_phases.load);
    return load_scene(file, t);
}
void photons_timed(SceneModel& scene) {
    if (scene.photon_steps > 0.0f) {
        Lap lap(g_phases.photon);
        PhotonMapper::generate(scene);
    }
}

// acc.mulAdd(raytrace(scene at t), weight) on the device
void accumulate_at(const std::string& file, float t, float weight, bool first, uint32_t* sub_frames_out = nullptr) {
    SceneModel scene = load_timed(file, t);
    photons_timed(scene);
    {
        Lap lap(g_phases.render);
        RayTracer().accumulate(scene, weight, first);
    }
    if (sub_frames_out) *sub_frames_out = scene.sub_frames;
}

void finished(Clock::time_point begin) {
    const std::chrono::duration<double> dt = Clock::now() - begin;
    std::cout << "\nFinished in " << dt.count() << " s\n";
}

// Frames are compressed (zlib, the costly part of the writer) on worker threads while the next frames render, and
// appended to the file strictly in order: the file is byte for byte what a sequential writer produces.
class FrameSink {
public:
    FrameSink(ApngWriter& apng, uint32_t width, uint32_t height) : apng_(apng), width_(width), height_(height) {}
    void push(Picture8&& picture, uint32_t index, float fps) {
        auto pixels = std::make_shared<std::vector<uint8_t>>(std::move(picture.pixels));
        const uint32_t w = width_, h = height_;
        queue_.push_back({std::async(std::launch::async, [pixels, w, h]() { return deflate_rgba8(pixels->data(), w, h); }), index, fps});
        drain(false);
    }
    void finish() { drain(true); }

private:
    struct Item {
        std::future<std::vector<uint8_t>> z;
        uint32_t index;
        float fps;
    };
    void drain(bool all) {
        while (!queue_.empty()) {
            Item& front = queue_.front();
            const bool ready = front.z.wait_for(std::chrono::seconds(0)) == std::future_status::ready;
            if (!ready && !all && queue_.size() < kMaxInFlight) break;
            Lap lap(g_phases.encode_wait);
            apng_.frame_deflated(front.z.get(), front.index, front.fps);
            queue_.pop_front();
        }
    }
    static constexpr size_t kMaxInFlight = 16;
    ApngWriter& apng_;
    uint32_t width_, height_;
    std::deque<Item> queue_;
};

void render_image(const SceneModel& orig) {
    const float start = orig.still_time == kInfinite ? 0.0f : orig.still_time;
    SceneModel scene = load_timed(orig.scene_file, start);
    if (scene.photon_steps > 0.0f) {
        std::cout << "Generating photon map for caustics.. This will take some time.." << std::endl;
        photons_timed(scene);
    }
    std::cout << "Rendering image.." << std::endl;
    const auto begin = Clock::now();
    Picture8 picture;
    {
        Lap lap(g_phases.render);
        picture = RayTracer().raytrace8(scene);
    }
    std::cout << "Writing image to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    {
        Lap lap(g_phases.encode_wait);
        write_png(out, picture.pixels.data(), picture.width, picture.height);
    }
    finished(begin);
}

void render_image_motion_blur(const SceneModel& orig) {
    const float start = orig.still_time == kInfinite ? 0.0f : orig.still_time;
    const SceneModel probe = load_timed(orig.scene_file, start);
    const auto begin = Clock::now();
    const uint32_t k = probe.sub_frames;
    for (uint32_t sub = 0; sub < k; sub++) {
        std::cout << "Rendering image (subframe " << sub + 1 << " of " << k << ")          \r" << std::flush;
        const float t = static_cast<float>(sub) / (k - 1) / orig.frames + start;
        accumulate_at(orig.scene_file, t, 1.0f / k, sub == 0);
    }
    const Picture8 picture = RayTracer().accumulated8();
    std::cout << std::endl << "Writing image to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    write_png(out, picture.pixels.data(), picture.width, picture.height);
    finished(begin);
}

void render_video(const SceneModel& orig) {
    const auto begin = Clock::now();
    std::cout << "Writing animation to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    ApngWriter apng(out, orig.camera.width, orig.camera.height, orig.frames);
    FrameSink sink(apng, orig.camera.width, orig.camera.height);
    for (uint32_t frame = 0; frame < orig.frames; frame++) {
        std::cout << "Rendering frame " << frame + 1 << " of " << orig.frames << "          \r" << std::flush;
        const float t = static_cast<float>(frame) / (orig.frames - 1);
        SceneModel scene = load_timed(orig.scene_file, t);
        photons_timed(scene);
        Picture8 picture;
        {
            Lap lap(g_phases.render);
            picture = RayTracer().raytrace8(scene);
        }
        sink.push(std::move(picture), frame, scene.fps);
    }
    sink.finish();
    apng.end();
    finished(begin);
}

void render_video_motion_blur(const SceneModel& orig) {
    const auto begin = Clock::now();
    std::cout << "Writing animation to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    ApngWriter apng(out, orig.camera.width, orig.camera.height, orig.frames);
    FrameSink sink(apng, orig.camera.width, orig.camera.height);
    uint32_t k = orig.sub_frames;
    for (uint32_t frame = 0; frame < orig.frames; frame++) {
        uint32_t next_k = k;   // the scene file may animate the sub-frame count; it applies from the next frame
        for (uint32_t sub = 0; sub < k; sub++) {
            std::cout << "Rendering frame " << frame + 1 << " of " << orig.frames << " (subframe " << sub + 1 << " of " << k
                      << ")          \r" << std::flush;
            const float t = (static_cast<float>(frame) + static_cast<float>(sub) / (k - 1)) / orig.frames;
            accumulate_at(orig.scene_file, t, 1.0f / k, sub == 0, &next_k);
        }
        Picture8 picture;
        {
            Lap lap(g_phases.render);
            picture = RayTracer().accumulated8();
        }
        sink.push(std::move(picture), frame, orig.fps);
        k = next_k;
    }
    sink.finish();
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
        // the GPU runtime starts up (some 35 ms: device context, code object) while the scene file is parsed; like the
        // reference's process start-up it lies before every timer
        std::future<void> gpu = std::async(std::launch::async, []() { (void)GpuSession::process_default(); });
        SceneModel scene = load_scene(argv[1]);
        gpu.get();
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
