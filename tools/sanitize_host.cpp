// Host-side sanitizer check (make asan / make tsan; SURVEY §5: the reference ships a commented-out
// -fsanitize=address,undefined line, /root/reference/Makefile:10-11).  Built with the sanitizers over the host
// loader, the PNG/APNG code and the oracle's CPU restatement, it
//   * loads every scene file under the given directories at several animation times from four threads at once (the
//     asset cache, the keyframe evaluator, OBJ and PNG readers),
//   * renders a small picture of each through the oracle and quantises it,
//   * writes it as PNG and as a two-frame APNG and reads both back,
// and exits 0 if nothing was reported.  usage: sanitize_host <scenes dir>...
#include <atomic>
#include <cstdio>
#include <cstring>
#include <dirent.h>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../include/rt_hip.h"
#include "../raytracer_amd/host/png_io.h"
#include "../raytracer_amd/host/scene_model.h"

struct oracle_photons;
extern "C" {
oracle_photons* oracle_photon_generate(const rt_scene_desc* sc, rt_stats* stats);
void oracle_photon_free(oracle_photons* p);
void oracle_render(const rt_scene_desc* sc, const oracle_photons* photons, int threads, float* rgba_out, rt_stats* stats);
}

static std::vector<std::string> scene_files(const char* dir) {
    std::vector<std::string> out;
    if (DIR* d = opendir(dir)) {
        while (dirent* e = readdir(d)) {
            const std::string n = e->d_name;
            if (n.size() > 4 && n.substr(n.size() - 4) == ".xml" && n.find("dtd") == std::string::npos) out.push_back(std::string(dir) + "/" + n);
        }
        closedir(d);
    }
    return out;
}

int main(int argc, char** argv) {
    std::vector<std::string> files;
    for (int i = 1; i < argc; i++)
        for (const std::string& f : scene_files(argv[i])) files.push_back(f);
    if (files.empty()) {
        std::fprintf(stderr, "no scene files\n");
        return 2;
    }
    std::atomic<size_t> next{0};
    std::atomic<int> loaded{0}, failed{0};
    auto worker = [&]() {
        size_t i;
        while ((i = next.fetch_add(1)) < files.size() * 3) {
            const std::string& path = files[i % files.size()];
            const float t = static_cast<float>(i / files.size()) * 0.5f;
            try {
                rta::SceneModel scene = rta::load_scene(path, t);
                scene.camera.width = 24;
                scene.camera.height = 16;
                if (scene.camera.supersampling > 2) scene.camera.supersampling = 2;
                if (scene.photon_steps > 60.0f) scene.photon_steps = 60.0f;
                const rt_scene_desc desc = scene.desc();
                rt_stats st;
                oracle_photons* ph = scene.photon_steps > 0.0f ? oracle_photon_generate(&desc, &st) : nullptr;
                std::vector<float> rgba(24 * 16 * 4);
                oracle_render(&desc, ph, 1, rgba.data(), &st);
                if (ph) oracle_photon_free(ph);
                std::vector<uint8_t> q(24 * 16 * 4);
                rta::quantize_rgba(rgba.data(), 24 * 16, 1.0f, q.data());
                std::stringstream png, apng;
                rta::write_png(png, q.data(), 24, 16);
                if (rta::read_png(png).pixels != q) throw std::runtime_error("PNG round trip differs");
                rta::ApngWriter w(apng, 24, 16, 2);
                w.frame(q.data(), 0, 10.0f);
                w.frame_deflated(rta::deflate_rgba8(q.data(), 24, 16), 1, 10.0f);
                w.end();
                const std::vector<rta::ImageRGBA8> frames = rta::read_apng_frames(apng);
                if (frames.size() != 2 || frames[1].pixels != q) throw std::runtime_error("APNG round trip differs");
                loaded++;
            } catch (const std::exception& e) {
                std::fprintf(stderr, "%s at t=%g: %s\n", path.c_str(), t, e.what());
                failed++;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int k = 0; k < 4; k++) pool.emplace_back(worker);
    for (std::thread& th : pool) th.join();
    std::printf("sanitize_host: %d scene loads rendered and round-tripped, %d failed\n", loaded.load(), failed.load());
    return failed.load() == 0 ? 0 : 1;
}
