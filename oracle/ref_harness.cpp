// ref_harness.cpp — drives the REAL reference (compiled from /root/reference/src where it lies, see
// oracle/Makefile) to produce golden vectors.  Test infrastructure; runs only in the build
// container (the reference never travels to the GPU box — its outputs do, under tests/golden/).
//
// This file is our own code: it uses the reference's public entry points (Scene::load,
// PhotonMapper::generate, RayTracer::raytrace, Object::intersect) and, by compiling THIS translation
// unit with private->public, reads the geometry members the reference keeps private so that the
// flattened scene can be compared with our own loader's.
//
//   ref_harness fingerprint <scene.xml> <t> <out.bin>
//   ref_harness render      <scene.xml> <t> <out.f32> [photon_out.bin]
//   ref_harness kat         <scene.xml> <t> <rays_per_object> <seed> <out.bin>
//   ref_harness hues
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <istream>
#include <memory>
#include <optional>
#include <random>
#include <string>
#include <variant>
#include <vector>

#define private public
#include "photonmap.h"
#include "raytracer.h"
#include "scene.h"
#undef private

namespace {

struct Words {
    std::vector<uint32_t> w;
    void u(uint32_t v) { w.push_back(v); }
    void f(float v) {
        uint32_t b;
        std::memcpy(&b, &v, 4);
        w.push_back(b);
    }
    void v3(const Vector3& v) { f(v.x); f(v.y); f(v.z); }
    void m34(const Matrix34& m) {
        for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) f(m.m[r][c]);
    }
    void col(const Color& c) { f(c.r); f(c.g); f(c.b); f(c.a); }
};

uint32_t fnv1a(const uint8_t* p, size_t n) {
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 16777619u;
    return h;
}

void put_material(Words& out, const Material& m) {
    out.col(m.color);
    if (m.texture.empty()) {
        out.u(0); out.u(0); out.u(0); out.u(0);
    } else {
        const UDim2 size = m.texture.size();
        std::vector<uint8_t> bytes;
        bytes.reserve(size_t(size.x) * size.y * 4);
        for (u32 y = 0; y < size.y; y++) for (u32 x = 0; x < size.x; x++) {
            const Color c = m.texture.get({x, y});
            for (float ch : {c.r, c.g, c.b, c.a}) bytes.push_back(uint8_t(std::lround(ch * 255.0f)));
        }
        out.u(1); out.u(size.x); out.u(size.y); out.u(fnv1a(bytes.data(), bytes.size()));
    }
    out.f(m.phong.ka); out.f(m.phong.kd); out.f(m.phong.ks); out.f(m.phong.exponent);
    out.f(m.reflectance); out.f(m.transmittance);
    out.f(m.refraction.real()); out.f(m.refraction.imag()); out.f(m.dispersion);
}

// Canonical word stream of everything the trace path reads.  The same order is produced from our
// own loader by rth_scene_fingerprint (raytracer_amd/host/host_capi.cpp).
Words fingerprint(const Scene& s) {
    Words out;
    const Camera& cam = s.camera();
    out.u(cam.resolution().x); out.u(cam.resolution().y); out.u(cam.superSamplingPerAxis()); out.u(cam.maxBounces());
    out.f(cam.fieldOfViewAngle()); out.f(cam.focusDistance()); out.f(cam.lensSize());
    out.m34(cam.cameraTransformation());
    out.col(s.background()); out.col(s.ambientLight());
    out.u(u32(s.lights().size()));
    for (const Light& l : s.lights()) {
        out.u(l.type() == Light::Type::Parallel ? 0 : 1);
        out.v3(l.position()); out.col(l.power());
    }
    out.u(s.dispersionMode() ? 1 : 0); out.f(s.photonMapScanSteps()); out.u(s.photonMapTextureSize()); out.f(s.photonMapFactor());
    out.u(s.frames()); out.f(s.fps()); out.u(s.subFrames()); out.f(s.time()); out.u(s.threads());
    out.u(u32(s.objects().size()));
    for (const Object& o : s.objects()) {
        if (const Sphere* sp = std::get_if<Sphere>(&o.m_object)) {
            out.u(0); out.v3(sp->m_center); out.f(sp->m_radius);
            out.m34(sp->m_world2Object); out.m34(sp->m_object2World); out.m34(sp->m_object2WorldNormals);
        } else if (const Triangle* t = std::get_if<Triangle>(&o.m_object)) {
            out.u(1);
            for (const auto& v : t->m_vertices) { out.v3(v.position); out.v3(v.normal); out.f(v.textureCoordinate.x); out.f(v.textureCoordinate.y); }
        } else {
            const Julia& j = std::get<Julia>(o.m_object);
            out.u(2); out.v3(j.m_position); out.f(j.m_scale);
            out.f(j.m_c.r); out.f(j.m_c.a); out.f(j.m_c.b); out.f(j.m_c.c); out.f(j.m_cutPlane);
            out.m34(j.m_world2Object); out.m34(j.m_object2World); out.m34(j.m_object2WorldNormals);
        }
        put_material(out, o.material());
    }
    return out;
}

void write_words(const std::string& path, const std::vector<uint32_t>& w) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(w.data()), std::streamsize(w.size() * 4));
}

int cmd_fingerprint(const std::string& scene_file, float t, const std::string& out) {
    const Scene s = Scene::load(scene_file, t);
    write_words(out, fingerprint(s).w);
    return 0;
}

int cmd_render(const std::string& scene_file, float t, const std::string& out, const std::string& photon_out) {
    Scene s = Scene::load(scene_file, t);
    if (s.photonMapScanSteps() > 0.0f) PhotonMapper::generate(s);
    RayTracer rt;
    const Picture pic = rt.raytrace(s);
    {
        std::ofstream f(out, std::ios::binary);
        const uint32_t hdr[2] = {pic.size().x, pic.size().y};
        f.write(reinterpret_cast<const char*>(hdr), sizeof hdr);
        for (u32 y = 0; y < pic.size().y; y++) for (u32 x = 0; x < pic.size().x; x++) {
            const Color& c = pic.get({x, y});
            const float px[4] = {c.r, c.g, c.b, c.a};
            f.write(reinterpret_cast<const char*>(px), sizeof px);
        }
    }
    if (!photon_out.empty()) {
        // sparse dump: u32 texture_size, u32 n_entries, then {u32 object, u32 texel, float rgba[4]}
        std::vector<uint32_t> w;
        const u32 size = s.photonMapTextureSize();
        w.push_back(size);
        w.push_back(0);
        u32 entries = 0, oi = 0;
        for (const Object& o : s.objects()) {
            if (!o.m_photonMap.empty()) {
                for (u32 y = 0; y < size; y++) for (u32 x = 0; x < size; x++) {
                    const Color& c = o.m_photonMap.get({x, y});
                    if (c.r != 0 || c.g != 0 || c.b != 0 || c.a != 0) {
                        Words e;
                        e.u(oi); e.u(y * size + x); e.col(c);
                        w.insert(w.end(), e.w.begin(), e.w.end());
                        entries++;
                    }
                }
            }
            oi++;
        }
        w[1] = entries;
        write_words(photon_out, w);
    }
    return 0;
}

// Known-answer vectors: camera rays through random pixels (and rays leaving their hit points in
// random directions) against every object, with infinite and finite max_distance.
// Record = {u32 object, float origin[3], float dir[3] (as passed to the Ray ctor), float max,
//           u32 hit, float distance, point[3], normal[3], uv[2], photon[2]}  = 20 words.
// Then Fresnel records: u32 0xFFFFFFFF marker, u32 count, then {u32 object(material), float cos,
// float wavelength, float kr} each.
int cmd_kat(const std::string& scene_file, float t, int rays_per_object, uint32_t seed, const std::string& out) {
    const Scene s = Scene::load(scene_file, t);
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> uni(0.0f, 1.0f);
    Words w;
    const Camera& cam = s.camera();
    const float tanv = tanf(cam.fieldOfViewAngle());
    const Matrix34 camM = cam.cameraTransformation();
    const Point3 eye = camM * Point3{0, 0, 0};
    auto record = [&](u32 oi, const Object& o, Point3 origin, Vector3 dir, float maxd) {
        const Ray ray(origin, dir);
        const auto hit = o.intersect(ray, maxd);
        w.u(oi); w.v3(origin); w.v3(dir); w.f(maxd);
        w.u(hit.has_value() ? 1 : 0);
        if (hit) {
            w.f(hit->distance); w.v3(hit->point); w.v3(hit->normal);
            w.f(hit->textureCoordinate.x); w.f(hit->textureCoordinate.y);
            w.f(hit->photonCoordinate.x); w.f(hit->photonCoordinate.y);
        } else {
            for (int i = 0; i < 11; i++) w.f(0.0f);
        }
        return hit;
    };
    u32 oi = 0;
    for (const Object& o : s.objects()) {
        for (int i = 0; i < rays_per_object; i++) {
            const float px = (uni(rng) * 2 - 1) * tanv, py = (uni(rng) * 2 - 1) * tanv;
            const Vector3 dir = camM.mulWithoutTranslate(Vector3{px, py, -1.0f});
            const auto hit = record(oi, o, eye, dir, INFINITE);
            if (hit) {
                // finite max around the hit distance (both sides), then a secondary ray from the hit
                record(oi, o, eye, dir, hit->distance * (0.5f + uni(rng)));
                const Vector3 rnd{uni(rng) * 2 - 1, uni(rng) * 2 - 1, uni(rng) * 2 - 1};
                const Point3 off = hit->point + hit->normal * (uni(rng) < 0.5f ? EPSILON : -EPSILON);
                record(oi, o, off, rnd, uni(rng) < 0.5f ? INFINITE : uni(rng) * 20.0f);
            }
        }
        oi++;
    }
    // Fresnel through the private member (this TU sees it as public)
    Picture dummy(UDim2{1, 1});
    RayTracer rt;
    RayTracer::Instance inst(rt, s, dummy);
    RayTracer::Instance::Thread thread(inst);
    std::vector<uint32_t> fres;
    u32 count = 0;
    oi = 0;
    for (const Object& o : s.objects()) {
        const Material& m = o.material();
        if (norm(m.refraction) > 0.0f && (oi == 0 || true)) {
            for (int i = 0; i < std::min(rays_per_object, 64); i++) {
                const float c = uni(rng) * 2 - 1;
                const float wl = s.dispersionMode() ? float(int(uni(rng) * 8)) * 45.0f / 180.0f - 1.0f : 0.0f;
                const float kr = thread.calcFresnel(m, c, wl);
                Words e;
                e.u(oi); e.f(c); e.f(wl); e.f(kr);
                fres.insert(fres.end(), e.w.begin(), e.w.end());
                count++;
            }
        }
        oi++;
    }
    w.u(0xFFFFFFFFu); w.u(count);
    w.w.insert(w.w.end(), fres.begin(), fres.end());
    write_words(out, w.w);
    return 0;
}

int cmd_hues() {
    for (float h = 0.0f; h < 360.0f; h += 45.0f) {
        const Color c = HSVtoRGB(h, 100.0f, 100.0f);
        std::printf("%g: %g %g %g %g\n", h, c.r, c.g, c.b, c.a);
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    try {
        const std::string cmd = argc > 1 ? argv[1] : "";
        if (cmd == "fingerprint" && argc == 5) return cmd_fingerprint(argv[2], std::stof(argv[3]), argv[4]);
        if (cmd == "render" && argc >= 5) return cmd_render(argv[2], std::stof(argv[3]), argv[4], argc > 5 ? argv[5] : "");
        if (cmd == "kat" && argc == 7) return cmd_kat(argv[2], std::stof(argv[3]), std::stoi(argv[4]), uint32_t(std::stoul(argv[5])), argv[6]);
        if (cmd == "hues") return cmd_hues();
        std::fprintf(stderr, "usage: see the header of oracle/ref_harness.cpp\n");
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
}
