// rt_amd [--devices N] <scene.xml> [<out.png>] — command line compatible with the reference executable
// (reference src/main.cpp:160-207), rendering through the HIP backend.
//
// --devices N (or RT_AMD_DEVICES=N) shards the work over N GPUs of the node from this one process, one host thread
// and one rt_ctx per GPU (the reference shards rows over CPU threads, src/raytracer.cpp:48-53):
//   still image            8-row bands interleaved over the devices, RGBA8 bands interleaved back on the host
//   still + motion blur    sub-frames round-robin; the float pictures are accumulated on the host in sub-frame order
//   animation              frames round-robin
//   animation + blur       frames round-robin; a frame's sub-frames accumulate on its device in order
// The pictures come back over each GPU's own host link to the process that encodes the file; the files are byte for
// byte those of one device.  RT_AMD_DEVICE_LIST=a,b,.. names the GPUs (default 0..N-1; a GPU may be named twice).
//
// The four render loops follow the reference's frame/sub-frame time formulas:
//   still image            t = <still time> or 0
//   still + motion blur    t = sub/(K-1)/frames + start,            acc += sub * (1/K)
//   animation              t = frame/(frames-1)
//   animation + blur       t = (frame + sub/(K-1))/frames,          K may change between frames
// The timer brackets the same work as the reference's (still image: starts after scene load and
// photon pass, includes the PNG encode; animations: everything).
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <future>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>

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

// ---- several GPUs from one process ----------------------------------------------------------------------------------

// One session (rt_ctx) per device, created one after the other on the calling thread, before this process has any
// other thread: the HIP runtime's start-up must not run beside threads that map or unmap memory (include/rt_hip.h,
// rt_create; tools/hip_init_race.cpp) — and starting a thread maps its stack.
class DevicePool {
public:
    explicit DevicePool(const std::vector<int>& devices) {
        for (int d : devices) sessions_.push_back(std::make_unique<GpuSession>(d));
    }
    size_t size() const { return sessions_.size(); }
    GpuSession& operator[](size_t i) { return *sessions_[i]; }

private:
    std::vector<std::unique_ptr<GpuSession>> sessions_;
};

// Runs job(unit, session) for units 0..n-1, unit u on device u % N (one thread per device, its units in increasing
// order); deliver(unit, result) runs on the calling thread in unit order.  A device runs at most `ahead` units in
// front of the delivery, which bounds the pictures held in memory.  The first exception (of a job or of deliver)
// stops the workers and is rethrown.
template <class R>
void for_units_in_order(DevicePool& pool, size_t n, size_t ahead, const std::function<R(size_t, GpuSession&)>& job,
                        const std::function<void(size_t, R&&)>& deliver) {
    std::vector<std::promise<R>> promises(n);
    std::vector<std::future<R>> futures;
    for (auto& p : promises) futures.push_back(p.get_future());
    std::mutex m;
    std::condition_variable cv;
    size_t delivered = 0;
    bool stop = false;
    std::vector<std::thread> workers;
    for (size_t d = 0; d < pool.size(); d++) {
        workers.emplace_back([&, d]() {
            for (size_t u = d; u < n; u += pool.size()) {
                {
                    std::unique_lock<std::mutex> lock(m);
                    cv.wait(lock, [&]() { return stop || u < delivered + ahead; });
                    if (stop) return;
                }
                try {
                    promises[u].set_value(job(u, pool[d]));
                } catch (...) {
                    promises[u].set_exception(std::current_exception());
                    return;
                }
            }
        });
    }
    std::exception_ptr failure;
    for (size_t u = 0; u < n && !failure; u++) {
        try {
            deliver(u, futures[u].get());
        } catch (...) {
            failure = std::current_exception();
        }
        std::lock_guard<std::mutex> lock(m);
        delivered = u + 1;
        stop = stop || static_cast<bool>(failure);
        cv.notify_all();
    }
    for (auto& w : workers) w.join();
    if (failure) std::rethrow_exception(failure);
}

struct FrameOut {
    Picture8 picture;
    float fps = 0.0f;
};

void render_image_devices(DevicePool& pool, const SceneModel& orig) {
    const float start = orig.still_time == kInfinite ? 0.0f : orig.still_time;
    const SceneModel scene = load_timed(orig.scene_file, start);
    const uint32_t n = static_cast<uint32_t>(pool.size());
    constexpr uint32_t kBandRows = 8;
    auto bands_of = [n](uint32_t d) { rt_bands b; std::memset(&b, 0, sizeof b); b.band_rows = kBandRows; b.first_band = d; b.band_stride = n; return b; };
    if (scene.photon_steps > 0.0f) {
        // every device runs the whole photon pass: summing partial passes would differ from the sequential sum in the
        // last float bits (the reference's pass is sequential, src/photonmap.cpp:32-56)
        std::cout << "Generating photon map for caustics.. This will take some time.." << std::endl;
        for_units_in_order<int>(pool, n, n, [&](size_t, GpuSession& s) { s.upload(scene); s.photons(); return 0; }, [](size_t, int&&) {});
    } else {
        for_units_in_order<int>(pool, n, n, [&](size_t, GpuSession& s) { s.upload(scene); return 0; }, [](size_t, int&&) {});
    }
    std::cout << "Rendering image.." << std::endl;
    const auto begin = Clock::now();
    Picture8 picture;
    picture.width = scene.camera.width;
    picture.height = scene.camera.height;
    picture.pixels.resize(size_t(picture.width) * picture.height * 4);
    const size_t row_bytes = size_t(picture.width) * 4;
    for_units_in_order<Picture8>(pool, n, n, [&](size_t d, GpuSession& s) { return s.render8(bands_of(static_cast<uint32_t>(d))); },
                                 [&](size_t d, Picture8&& part) {
                                     size_t src_row = 0;
                                     for (uint32_t band = static_cast<uint32_t>(d); band * kBandRows < picture.height; band += n) {
                                         for (uint32_t y = band * kBandRows; y < picture.height && y < (band + 1) * kBandRows; y++, src_row++) {
                                             std::memcpy(&picture.pixels[size_t(y) * row_bytes], &part.pixels[src_row * row_bytes], row_bytes);
                                         }
                                     }
                                 });
    std::cout << "Writing image to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    write_png(out, picture.pixels.data(), picture.width, picture.height);
    finished(begin);
}

void render_image_motion_blur_devices(DevicePool& pool, const SceneModel& orig) {
    const float start = orig.still_time == kInfinite ? 0.0f : orig.still_time;
    const SceneModel probe = load_timed(orig.scene_file, start);
    const auto begin = Clock::now();
    const uint32_t k = probe.sub_frames;
    Picture acc(probe.camera.width, probe.camera.height);
    for_units_in_order<Picture>(pool, k, 2 * pool.size(),
                                [&](size_t sub, GpuSession& s) {
                                    const float t = static_cast<float>(sub) / (k - 1) / orig.frames + start;
                                    const SceneModel scene = load_scene(orig.scene_file, t);
                                    s.upload(scene);
                                    if (scene.photon_steps > 0.0f) s.photons();
                                    return s.render();
                                },
                                [&](size_t sub, Picture&& pic) {
                                    std::cout << "Rendering image (subframe " << sub + 1 << " of " << k << ")          \r" << std::flush;
                                    acc.mulAdd(pic, 1.0f / k);      // Picture::mulAdd in sub-frame order (src/main.cpp:64)
                                });
    const Picture8 picture = scale_out(acc);
    std::cout << std::endl << "Writing image to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    write_png(out, picture.pixels.data(), picture.width, picture.height);
    finished(begin);
}

void render_video_devices(DevicePool& pool, const SceneModel& orig) {
    const auto begin = Clock::now();
    std::cout << "Writing animation to " << orig.output_file << std::endl;
    std::ofstream out = open_output(orig.output_file);
    ApngWriter apng(out, orig.camera.width, orig.camera.height, orig.frames);
    FrameSink sink(apng, orig.camera.width, orig.camera.height);
    // with motion blur the scene file may animate the sub-frame count; it applies from the next frame (src/main.cpp:125-151)
    std::vector<uint32_t> k_of(orig.frames, orig.sub_frames);
    if (orig.sub_frames > 1) {
        for (uint32_t frame = 0; frame + 1 < orig.frames; frame++) {
            const uint32_t k = k_of[frame];
            const float t_last = (static_cast<float>(frame) + static_cast<float>(k - 1) / (k - 1)) / orig.frames;
            k_of[frame + 1] = load_scene(orig.scene_file, t_last).sub_frames;
        }
    }
    for_units_in_order<FrameOut>(pool, orig.frames, 4 * pool.size(),
                                 [&](size_t frame, GpuSession& s) {
                                     FrameOut o;
                                     if (orig.sub_frames <= 1) {
                                         const float t = static_cast<float>(frame) / (orig.frames - 1);
                                         const SceneModel scene = load_scene(orig.scene_file, t);
                                         s.upload(scene);
                                         if (scene.photon_steps > 0.0f) s.photons();
                                         o.picture = s.render8();
                                         o.fps = scene.fps;
                                     } else {
                                         const uint32_t k = k_of[frame];
                                         for (uint32_t sub = 0; sub < k; sub++) {
                                             const float t = (static_cast<float>(frame) + static_cast<float>(sub) / (k - 1)) / orig.frames;
                                             const SceneModel scene = load_scene(orig.scene_file, t);
                                             s.upload(scene);
                                             if (scene.photon_steps > 0.0f) s.photons();
                                             s.accumulate(1.0f / k, sub == 0);
                                         }
                                         o.picture = s.accumulated8();
                                         o.fps = orig.fps;
                                     }
                                     return o;
                                 },
                                 [&](size_t frame, FrameOut&& o) {
                                     std::cout << "Rendering frame " << frame + 1 << " of " << orig.frames << "          \r" << std::flush;
                                     sink.push(std::move(o.picture), static_cast<uint32_t>(frame), o.fps);
                                 });
    sink.finish();
    apng.end();
    finished(begin);
}

std::vector<int> device_list(int n) {
    std::vector<int> list;
    if (const char* env = std::getenv("RT_AMD_DEVICE_LIST")) {
        std::stringstream ss(env);
        std::string item;
        while (std::getline(ss, item, ',')) list.push_back(std::stoi(item));
        if (static_cast<int>(list.size()) != n) throw std::runtime_error("RT_AMD_DEVICE_LIST must name exactly --devices GPUs");
    } else {
        for (int d = 0; d < n; d++) list.push_back(d);
    }
    return list;
}

}  // namespace

// RT_AMD_BACKTRACE=1: print the faulting address and the stack of a crashing thread (developer aid)
extern "C" void crash_backtrace(int sig, siginfo_t* info, void*) {
    char line[96];
    const int len = snprintf(line, sizeof line, "signal %d at address %p, si_code %d\n", sig, info ? info->si_addr : nullptr, info ? info->si_code : 0);
    if (len > 0) (void)!write(STDERR_FILENO, line, static_cast<size_t>(len));
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, STDERR_FILENO);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char* argv[]) {
    if (std::getenv("RT_AMD_BACKTRACE")) {
        struct sigaction sa;
        std::memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = crash_backtrace;
        sa.sa_flags = SA_SIGINFO;
        sigaction(SIGSEGV, &sa, nullptr);
        sigaction(SIGABRT, &sa, nullptr);
    }
    int n_devices = 1;
    if (const char* env = std::getenv("RT_AMD_DEVICES")) n_devices = std::atoi(env);
    if (argc >= 3 && std::strcmp(argv[1], "--devices") == 0) {
        n_devices = std::atoi(argv[2]);
        argv += 2;
        argc -= 2;
    }
    if (argc < 2 || n_devices < 1) {
        std::cout << "Usage: " << std::endl << argv[0] << " [--devices N] <scene.xml> [<out.png>]" << std::endl;
        return -1;
    }
    try {
        if (n_devices > 1) {
            // (GPU runtime start-up and the scene parse do not overlap: see below)
            std::unique_ptr<DevicePool> pool = std::make_unique<DevicePool>(device_list(n_devices));
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
                render_video_devices(*pool, scene);
            } else {
                if (scene.sub_frames > 1) render_image_motion_blur_devices(*pool, scene); else render_image_devices(*pool, scene);
            }
            return 0;
        }
        // The GPU runtime starts up first (some 35 ms warm: device context, code object); like the reference's process
        // start-up it lies before every timer.  It is NOT overlapped with the scene parse: with the parse running on
        // this thread while another one sat in the runtime's initialisation, 2 of ~300 runs died with SIGSEGV in a
        // memcpy into a freshly allocated texture buffer (tools/cli_stress.sh, RT_AMD_BACKTRACE=1).  Cause, found with
        // a host-only program that has none of this repository's code in it (tools/hip_init_race.cpp, 2 crashes in 200
        // runs; /proc/self/maps at the fault): during the process's FIRST HIP calls the runtime unmaps a single page
        // that it no longer owns — the fault address is the start of a one-page hole inside the anonymous mapping
        // that malloc had just obtained for the other thread's multi-megabyte buffer.  HIP calls made once the runtime
        // is up (allocations, pinned memory, copies, streams, events: 300 runs) leave other threads' mappings alone.
        // Hence the rule of include/rt_hip.h: the first rt_create runs while no other thread maps or unmaps memory.
        (void)GpuSession::process_default();
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
