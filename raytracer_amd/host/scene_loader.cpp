// Scene file loader: XML subset tokenizer, keyframe evaluator, OBJ subset reader and the scene
// grammar, producing the flat SceneModel the C ABI consumes.
//
// Host code on the reference's side of the boundary.  Written from the file
// formats (reference README.md:63-130, examples/scene.dtd) and the behaviours catalogued in SURVEY
// Appendix A; citations name the reference lines whose observable behaviour is matched.
#include <algorithm>
#include <array>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <istream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>

#include "png_io.h"
#include "scene_model.h"
#include "vecmath.h"

namespace rta {
namespace {

// ------------------------------------------------------------------ markup reader
// Tags only: text content is skipped, `<?..>` and `<!..>` are ignored, attribute values cannot
// contain spaces (behaviour of reference src/xml.cpp:10-71).

enum class TagKind { Open, Close, SelfClosing };

struct Tag {
    std::string name;
    TagKind kind = TagKind::Open;
    std::map<std::string, std::string> attrs;

    bool is(const char* n, TagKind k) const { return kind == k && name == n; }
    const std::string& attr(const std::string& key) const {
        auto it = attrs.find(key);
        if (it == attrs.end()) throw std::runtime_error("Attribute \"" + key + "\" not found");
        return it->second;
    }
    bool has(const std::string& key) const { return attrs.count(key) != 0; }
};

bool is_word_char(char c) { return std::isalnum(static_cast<unsigned char>(c)) || c == '_'; }

class MarkupReader {
public:
    explicit MarkupReader(std::istream& in) : in_(in) {}

    const Tag& next() {
        for (;;) {
            in_.ignore(std::numeric_limits<std::streamsize>::max(), '<');
            std::string body;
            std::getline(in_, body, '>');
            if (body.empty()) {
                throw std::runtime_error(in_.eof() ? "xml file ended unexpectedly"
                                                   : "xml file contains completely empty tag");
            }
            raw_ = body;
            tag_ = Tag();
            if (body[0] == '?' || body[0] == '!') continue;   // prolog, DOCTYPE, comments
            if (body[0] == '/') {
                tag_.kind = TagKind::Close;
                tag_.name = body.substr(1);
                return tag_;
            }
            if (body.back() == '/') {
                tag_.kind = TagKind::SelfClosing;
                body.pop_back();
            }
            std::istringstream words(body);
            if (!(words >> tag_.name)) throw std::runtime_error("xml file contains tag without name");
            std::string word;
            while (words >> word) parse_attribute(word);
            return tag_;
        }
    }

    const Tag& current() const { return tag_; }
    const std::string& raw() const { return raw_; }

private:
    // name="value" or name='value'; name is \w+, value has no quote characters
    void parse_attribute(const std::string& w) {
        size_t eq = 0;
        while (eq < w.size() && is_word_char(w[eq])) eq++;
        auto is_quote = [](char c) { return c == '"' || c == '\''; };
        const bool shape_ok = eq > 0 && w.size() >= eq + 3 && w[eq] == '=' && is_quote(w[eq + 1]) &&
                              is_quote(w.back());
        if (!shape_ok) throw std::runtime_error("xml file contains invalid attribute");
        const std::string value = w.substr(eq + 2, w.size() - eq - 3);
        if (value.find_first_of("\"'") != std::string::npos) {
            throw std::runtime_error("xml file contains invalid attribute");
        }
        tag_.attrs[w.substr(0, eq)] = value;
    }

    std::istream& in_;
    Tag tag_;
    std::string raw_;
};

// ------------------------------------------------------------------ keyframes
// "v0;v1(ease,time);v2(ease);v3(time)" evaluated at t (reference src/sceneparser.cpp:308-386).

float ease_curve(char kind, float x) {
    // the cubic is evaluated in double and narrowed, as std::pow(float,int) does
    switch (kind) {
        case 'l': return x;
        case 'i': return static_cast<float>(std::pow(static_cast<double>(x), 3));
        case 'o': return static_cast<float>(1 - std::pow(static_cast<double>(1 - x), 3));
        case 'b':
            return static_cast<float>(x < 0.5f ? std::pow(static_cast<double>(x * 2), 3) / 2
                                               : 1 - std::pow(static_cast<double>((1 - x) * 2), 3) / 2);
        default: throw std::runtime_error("invalid ease function selected");
    }
}

bool is_number_char(char c) {
    return std::isdigit(static_cast<unsigned char>(c)) || c == '.' || c == 'E' || c == 'e' || c == '+' || c == '-';
}

struct KeyframeCursor {
    const std::string& s;
    size_t pos = 0;

    void skip_space() {
        while (pos < s.size() && std::isspace(static_cast<unsigned char>(s[pos]))) pos++;
    }
    // maximal run of [0-9.Ee+-] (what the reference's token pattern accepts); false if empty
    bool number(std::string* out) {
        size_t p = pos;
        while (p < s.size() && is_number_char(s[p])) p++;
        if (p == pos) return false;
        *out = s.substr(pos, p - pos);
        pos = p;
        return true;
    }
};

float to_float(const std::string& token) {
    size_t used = 0;
    float v;
    try {
        v = std::stof(token, &used);
    } catch (const std::exception&) {
        throw std::runtime_error("stof");
    }
    return v;
}

float evaluate_keyframes(const std::string& text, float t) {
    KeyframeCursor c{text};
    bool first = true;
    float value = 0.0f, time = 0.0f;
    char ease = 'l';
    while (c.pos <= text.size()) {
        // separator: start of string for the first keyframe, ';' otherwise
        if (c.pos < text.size() && text[c.pos] == ';') {
            c.pos++;
        } else if (!first) {
            break;
        }
        c.skip_space();
        std::string value_tok;
        if (!c.number(&value_tok)) break;
        c.skip_space();
        std::string time_tok;
        bool ease_given = false;
        char ease_here = 'l';
        if (c.pos < text.size() && text[c.pos] == '(') {
            const size_t rewind = c.pos;
            c.pos++;
            c.skip_space();
            bool ok = true;
            if (c.pos < text.size() && std::strchr("liob", text[c.pos]) != nullptr) {
                ease_given = true;
                ease_here = text[c.pos++];
                c.skip_space();
                if (c.pos < text.size() && text[c.pos] == ',') {
                    c.pos++;
                    c.skip_space();
                    ok = c.number(&time_tok);
                    c.skip_space();
                }
            } else {
                c.number(&time_tok);   // optional
                c.skip_space();
            }
            if (ok && c.pos < text.size() && text[c.pos] == ')') {
                c.pos++;
            } else {
                c.pos = rewind;   // malformed group: not part of this keyframe
                ease_given = false;
                time_tok.clear();
            }
        }
        c.skip_space();

        const float target_value = to_float(value_tok);
        if (ease_given) ease = ease_here;
        float target_time = first ? 0.0f : 1.0f;
        if (!time_tok.empty()) target_time = to_float(time_tok);
        if (target_time < 0.0f || target_time > 1.0f) throw std::runtime_error("invalid animation time");

        if (first || target_time < t) {
            if (time > target_time) throw std::runtime_error("animation time not in increasing order");
            value = target_value;
            time = target_time;
        } else {
            if (time > t) return value;   // before the (delayed) first keyframe
            return ease_curve(ease, (t - time) / (target_time - time)) * (target_value - value) + value;
        }
        first = false;
        if (c.pos >= text.size()) break;
    }
    return value;
}

// ------------------------------------------------------------------ OBJ subset
// v / vt / vn / f a/b/c triangles (reference src/wavefobj.cpp:24-80)

struct ObjCorner {
    uint32_t v = 0, t = 0, n = 0;
};
struct ObjMesh {
    std::vector<Vec3> positions, normals;
    std::vector<Vec2> uvs;
    std::vector<std::array<ObjCorner, 3>> faces;
};

bool parse_corner(const std::string& text, ObjCorner* out) {
    std::istringstream s(text);
    uint32_t v = 0, t = 0, n = 0;
    char sep = 0;
    if (!(s >> v >> sep) || sep != '/' || v == 0) return false;
    if (!(s >> t)) {
        s.clear();   // "v//n": texture index is optional
        t = 0;
    }
    if (!(s >> sep >> n) || sep != '/' || n == 0) return false;
    *out = {v, t, n};
    return true;
}

ObjMesh read_obj(const std::string& path) {
    std::ifstream file(path);
    if (!file) throw std::runtime_error("mesh obj file \"" + path + "\" could not be opened");
    ObjMesh mesh;
    uint32_t max_v = 0, max_t = 0, max_n = 0;
    std::string line_text;
    while (std::getline(file, line_text)) {
        std::istringstream line(line_text);
        std::string cmd;
        if (!(line >> cmd)) continue;
        if (cmd == "v") {
            Vec3 p;
            if (line >> p.x >> p.y >> p.z) mesh.positions.push_back(p);
        } else if (cmd == "vt") {
            Vec2 p;
            if (line >> p.x >> p.y) mesh.uvs.push_back(p);
        } else if (cmd == "vn") {
            Vec3 p;
            if (line >> p.x >> p.y >> p.z) mesh.normals.push_back(p);
        } else if (cmd == "f") {
            std::string w[3];
            if (!(line >> w[0] >> w[1] >> w[2])) continue;
            std::array<ObjCorner, 3> face;
            bool ok = true;
            for (int i = 0; i < 3; i++) {
                // the index maxima are only updated by corners that parsed, like the reference,
                // where a bad corner returns before the max update (src/wavefobj.cpp:54-68)
                if (parse_corner(w[i], &face[i])) {
                    max_v = std::max(max_v, face[i].v);
                    max_t = std::max(max_t, face[i].t);
                    max_n = std::max(max_n, face[i].n);
                } else {
                    ok = false;
                }
            }
            if (ok) mesh.faces.push_back(face);
        }
    }
    if (max_n > mesh.normals.size() || max_t > mesh.uvs.size() || max_v > mesh.positions.size()) {
        throw std::runtime_error("mesh obj file contains an out of bounds index on a face");
    }
    return mesh;
}

// ------------------------------------------------------------------ asset cache
// The reference re-reads and re-decodes every mesh and texture for every frame and sub-frame
// (src/main.cpp:99,140 call Scene::load afresh); only the keyframed attributes change with t.  Decoded
// assets are therefore kept per file (invalidated by size or modification time), shared by all loads
// of the process, including loads made from a prefetch thread.
struct FileStamp {
    long long size = -1, mtime_ns = -1;
    bool operator==(const FileStamp& o) const { return size == o.size && mtime_ns == o.mtime_ns; }
};
FileStamp stamp_of(const std::string& path) {
    struct stat st;
    FileStamp f;
    if (::stat(path.c_str(), &st) == 0) {
        f.size = static_cast<long long>(st.st_size);
        f.mtime_ns = static_cast<long long>(st.st_mtim.tv_sec) * 1000000000ll + st.st_mtim.tv_nsec;
    }
    return f;
}
template <typename T>
class AssetCache {
public:
    template <typename Load>
    std::shared_ptr<const T> get(const std::string& path, Load load) {
        const FileStamp now = stamp_of(path);
        {
            std::lock_guard<std::mutex> lock(mutex_);
            auto it = entries_.find(path);
            if (it != entries_.end() && now.size >= 0 && it->second.stamp == now) return it->second.value;
        }
        std::shared_ptr<const T> fresh = std::make_shared<const T>(load(path));   // throws like an uncached load
        std::lock_guard<std::mutex> lock(mutex_);
        entries_[path] = {now, fresh};
        return fresh;
    }

private:
    struct Entry {
        FileStamp stamp;
        std::shared_ptr<const T> value;
    };
    std::mutex mutex_;
    std::map<std::string, Entry> entries_;
};
std::shared_ptr<const ObjMesh> cached_obj(const std::string& path) {
    static AssetCache<ObjMesh> cache;
    return cache.get(path, read_obj);
}
std::shared_ptr<const ImageRGBA8> cached_png(const std::string& path) {
    static AssetCache<ImageRGBA8> cache;
    return cache.get(path, read_png_file);
}

// ------------------------------------------------------------------ path helper
// sibling file of the scene file unless absolute (reference src/fspolyfill.h:6-20)
std::string sibling_path(const std::string& scene_path, const std::string& name) {
    if (!name.empty() && std::strchr(":/\\", name.front()) != nullptr) return name;
    if (name.find(':') != std::string::npos) return name;
    const size_t cut = scene_path.find_last_of(":/\\");
    return (cut == std::string::npos ? std::string() : scene_path.substr(0, cut + 1)) + name;
}

// ------------------------------------------------------------------ grammar

void store(const Mat34& m, rt_mat34* out) { std::memcpy(out->m, m.m, sizeof out->m); }

struct Transform {
    Mat34 o2w = Mat34::identity();
    Mat34 o2w_normal = Mat34::identity();
    Mat34 w2o = Mat34::identity();
};

struct Placement {
    Vec3 position;
    rt_material material{};
    Transform transform;
};

class SceneGrammar {
public:
    SceneGrammar(std::istream& in, std::string path, float t) : xml_(in), path_(std::move(path)), t_(t) {}

    SceneModel parse() {
        try {
            if (!xml_.next().is("scene", TagKind::Open)) throw std::runtime_error("scene tag expected");
            scene_element();
        } catch (const std::exception& e) {
            throw std::runtime_error("scene file parse error at tag <" + xml_.raw() + "> Error: " + e.what());
        }
        scene_.scene_file = path_;
        return std::move(scene_);
    }

private:
    const Tag& tag() const { return xml_.current(); }
    bool at(const char* name, TagKind k) const { return tag().is(name, k); }

    float number(const char* key) const { return evaluate_keyframes(tag().attr(key), t_); }
    float number(const char* key, float fallback) const {
        return tag().has(key) ? evaluate_keyframes(tag().attr(key), t_) : fallback;
    }
    uint32_t integer(const char* key) const { return static_cast<uint32_t>(std::stoul(tag().attr(key))); }
    uint32_t integer(const char* key, uint32_t fallback) const {
        return tag().has(key) ? static_cast<uint32_t>(std::stoul(tag().attr(key))) : fallback;
    }
    Vec3 xyz() const { return {number("x"), number("y"), number("z")}; }
    void rgba(float out[4]) const {
        out[0] = number("r");
        out[1] = number("g");
        out[2] = number("b");
        out[3] = number("a", 1.0f);
    }

    void scene_element() {
        scene_.output_file = tag().attr("output_file");
        scene_.threads = integer("threads", scene_.threads);
        // camera defaults (reference src/objects.h:194-204)
        cam_pos_ = {0, 0, 0};
        cam_look_ = {0, 0, -1};
        cam_up_ = {0, 1, 0};
        cam_focus_ = {0, 0, -1};
        scene_.camera.fov = kPi / 4;
        scene_.camera.width = scene_.camera.height = 512;
        scene_.camera.max_bounces = 8;
        scene_.camera.supersampling = 1;
        scene_.camera.lens_size = 0.0f;

        while (!xml_.next().is("scene", TagKind::Close)) {
            if (at("background_color", TagKind::SelfClosing)) {
                rgba(scene_.background);
            } else if (at("animation", TagKind::SelfClosing)) {
                scene_.fps = number("fps");
                scene_.frames = static_cast<uint32_t>(std::ceil(number("length") * scene_.fps));
            } else if (at("still", TagKind::SelfClosing)) {
                scene_.still_time = number("time");
            } else if (at("motionblur", TagKind::SelfClosing)) {
                scene_.sub_frames = static_cast<uint32_t>(std::ceil(number("subframes")));
            } else if (at("caustic", TagKind::SelfClosing)) {
                scene_.photon_steps = number("steps");
                scene_.photon_texture_size = integer("texture_size");
                scene_.photon_factor = number("factor");
            } else if (at("camera", TagKind::Open)) {
                camera_element();
            } else if (at("lights", TagKind::Open)) {
                lights_element();
            } else if (at("surfaces", TagKind::Open)) {
                surfaces_element();
            } else {
                throw std::runtime_error("unknown tag in scene");
            }
        }
        store(Mat34::look_at(cam_pos_, cam_look_, cam_up_), &scene_.camera.camera_to_world);
        scene_.camera.focus_distance = length(cam_focus_ - cam_pos_);
        scene_.dispersion_mode = false;
        for (const rt_object& o : scene_.objects) {
            if (scene_.materials[o.material].dispersion != 0.0f) scene_.dispersion_mode = true;
        }
    }

    void camera_element() {
        // a repeated <camera> starts again from the defaults (reference src/sceneparser.cpp:71-72)
        cam_pos_ = {0, 0, 0};
        cam_look_ = {0, 0, -1};
        cam_up_ = {0, 1, 0};
        cam_focus_ = {0, 0, -1};
        rt_camera& cam = scene_.camera;
        cam.fov = kPi / 4;
        cam.width = cam.height = 512;
        cam.max_bounces = 8;
        cam.supersampling = 1;
        cam.lens_size = 0.0f;
        while (!xml_.next().is("camera", TagKind::Close)) {
            if (at("position", TagKind::SelfClosing)) {
                cam_pos_ = xyz();
            } else if (at("lookat", TagKind::SelfClosing)) {
                cam_look_ = xyz();
            } else if (at("up", TagKind::SelfClosing)) {
                cam_up_ = xyz();
            } else if (at("horizontal_fov", TagKind::SelfClosing)) {
                cam.fov = number("angle") / 360 * 2 * kPi;   // degrees -> radians (src/sceneparser.cpp:81)
            } else if (at("resolution", TagKind::SelfClosing)) {
                cam.width = integer("horizontal");
                cam.height = integer("vertical");
            } else if (at("max_bounces", TagKind::SelfClosing)) {
                cam.max_bounces = static_cast<uint32_t>(std::lroundf(number("n")));
            } else if (at("supersampling", TagKind::SelfClosing)) {
                cam.supersampling = integer("subpixels_peraxis");
            } else if (at("dof", TagKind::SelfClosing)) {
                cam_focus_ = xyz();
                cam.lens_size = number("lenssize");
            } else {
                throw std::runtime_error("unknown tag in camera");
            }
        }
    }

    void lights_element() {
        scene_.lights.clear();
        scene_.ambient[0] = scene_.ambient[1] = scene_.ambient[2] = scene_.ambient[3] = 0.0f;
        while (!xml_.next().is("lights", TagKind::Close)) {
            if (at("ambient_light", TagKind::Open)) {
                const rt_light l = light_element();
                std::memcpy(scene_.ambient, l.power, sizeof l.power);
            } else if (at("parallel_light", TagKind::Open) || at("point_light", TagKind::Open)) {
                scene_.lights.push_back(light_element());
            } else {
                throw std::runtime_error("unknown tag in lights");
            }
        }
    }

    rt_light light_element() {
        const std::string name = tag().name;
        rt_light l{};
        l.type = name == "parallel_light" ? RT_LIGHT_PARALLEL : RT_LIGHT_POINT;
        while (!xml_.next().is(name.c_str(), TagKind::Close)) {
            if (at("color", TagKind::SelfClosing)) {
                rgba(l.power);
            } else if (at("direction", TagKind::SelfClosing) || at("position", TagKind::SelfClosing)) {
                const Vec3 v = xyz();
                l.position[0] = v.x;
                l.position[1] = v.y;
                l.position[2] = v.z;
            } else {
                throw std::runtime_error("unknown tag in " + name);
            }
        }
        return l;
    }

    void surfaces_element() {
        scene_.objects.clear();
        scene_.spheres.clear();
        scene_.triangles.clear();
        scene_.julias.clear();
        while (!xml_.next().is("surfaces", TagKind::Close)) {
            if (at("sphere", TagKind::Open)) {
                rt_sphere s{};
                s.radius = number("radius");
                const Placement p = placement_element();
                s.center[0] = p.position.x;
                s.center[1] = p.position.y;
                s.center[2] = p.position.z;
                store(p.transform.w2o, &s.world_to_object);
                store(p.transform.o2w, &s.object_to_world);
                store(p.transform.o2w_normal, &s.object_to_world_normals);
                scene_.objects.push_back({RT_OBJ_SPHERE, uint32_t(scene_.spheres.size()), add_material(p)});
                scene_.spheres.push_back(s);
            } else if (at("mesh", TagKind::Open)) {
                const std::shared_ptr<const ObjMesh> mesh = cached_obj(sibling_path(path_, tag().attr("name")));
                const Placement p = placement_element();
                add_mesh(*mesh, p);
            } else if (at("julia", TagKind::Open)) {
                rt_julia j{};
                j.scale = number("scale");
                j.c[0] = number("cr");
                j.c[1] = number("ca");
                j.c[2] = number("cb");
                j.c[3] = number("cc");
                j.cut_plane = number("cutplane");
                const Placement p = placement_element();
                j.position[0] = p.position.x;
                j.position[1] = p.position.y;
                j.position[2] = p.position.z;
                store(p.transform.w2o, &j.world_to_object);
                store(p.transform.o2w, &j.object_to_world);
                store(p.transform.o2w_normal, &j.object_to_world_normals);
                scene_.objects.push_back({RT_OBJ_JULIA, uint32_t(scene_.julias.size()), add_material(p)});
                scene_.julias.push_back(j);
            } else {
                throw std::runtime_error("unknown tag in surfaces");
            }
        }
    }

    uint32_t add_material(const Placement& p) {
        scene_.materials.push_back(p.material);
        return uint32_t(scene_.materials.size() - 1);
    }

    // One triangle object per face, vertices and normals moved to world space
    // (reference src/wavefobj.cpp:82-106)
    void add_mesh(const ObjMesh& mesh, const Placement& p) {
        std::vector<Vec3> pos, nrm;
        pos.reserve(mesh.positions.size());
        nrm.reserve(mesh.normals.size());
        for (const Vec3& v : mesh.positions) pos.push_back(p.transform.o2w.apply(v));
        for (const Vec3& n : mesh.normals) nrm.push_back(normalized(p.transform.o2w_normal.apply(n)));
        const uint32_t material = add_material(p);
        for (const auto& face : mesh.faces) {
            rt_triangle tri{};
            for (int k = 0; k < 3; k++) {
                const Vec3& v = pos[face[k].v - 1];
                const Vec3& n = nrm[face[k].n - 1];
                tri.position[k][0] = v.x;
                tri.position[k][1] = v.y;
                tri.position[k][2] = v.z;
                tri.normal[k][0] = n.x;
                tri.normal[k][1] = n.y;
                tri.normal[k][2] = n.z;
                if (face[k].t > 0) {
                    tri.uv[k][0] = mesh.uvs[face[k].t - 1].x;
                    tri.uv[k][1] = mesh.uvs[face[k].t - 1].y;
                }
            }
            scene_.objects.push_back({RT_OBJ_TRIANGLE, uint32_t(scene_.triangles.size()), material});
            scene_.triangles.push_back(tri);
        }
    }

    Placement placement_element() {
        const std::string name = tag().name;
        Placement p;
        p.material.texture = -1;
        while (!xml_.next().is(name.c_str(), TagKind::Close)) {
            if (at("position", TagKind::SelfClosing)) {
                p.position = xyz();
            } else if (at("material_solid", TagKind::Open) || at("material_textured", TagKind::Open)) {
                p.material = material_element();
            } else if (at("transform", TagKind::Open)) {
                p.transform = transform_element();
            } else {
                throw std::runtime_error("unknown tag in " + name);
            }
        }
        return p;
    }

    // Missing sub-tags leave zeros here; the reference leaves them uninitialised
    // (src/sceneparser.cpp:185) — every shipped scene gives all of them.
    rt_material material_element() {
        const std::string name = tag().name;
        rt_material m{};
        m.texture = -1;
        while (!xml_.next().is(name.c_str(), TagKind::Close)) {
            if (at("color", TagKind::SelfClosing)) {
                rgba(m.color);
            } else if (at("texture", TagKind::SelfClosing)) {
                m.texture = texture_index(sibling_path(path_, tag().attr("name")));
            } else if (at("phong", TagKind::SelfClosing)) {
                m.ka = number("ka");
                m.kd = number("kd");
                m.ks = number("ks");
                m.exponent = number("exponent");
            } else if (at("reflectance", TagKind::SelfClosing)) {
                m.reflectance = number("r");
            } else if (at("transmittance", TagKind::SelfClosing)) {
                m.transmittance = number("t");
            } else if (at("refraction", TagKind::SelfClosing)) {
                m.refraction_re = number("iof");
                m.refraction_im = number("ec", 0.0f);
                m.dispersion = number("disp", 0.0f);
            } else {
                throw std::runtime_error("unknown tag in " + name);
            }
        }
        return m;
    }

    // Textures are shared by file name (the reference copies the decoded picture into every
    // object's material, src/objects.h:132-134).
    int32_t texture_index(const std::string& file) {
        auto it = texture_by_file_.find(file);
        if (it != texture_by_file_.end()) return it->second;
        const std::shared_ptr<const ImageRGBA8> cached = cached_png(file);
        const ImageRGBA8& img = *cached;
        rt_texture t{};
        t.width = img.width;
        t.height = img.height;
        t.offset = scene_.texels.size();
        scene_.texels.insert(scene_.texels.end(), img.pixels.begin(), img.pixels.end());
        scene_.textures.push_back(t);
        const int32_t index = int32_t(scene_.textures.size() - 1);
        texture_by_file_[file] = index;
        return index;
    }

    // Steps are applied in file order: o2w = o2w·M, w2o = M⁻¹·w2o, normals = normals·(S⁻¹ or R)
    // (reference src/sceneparser.cpp:216-252)
    Transform transform_element() {
        Transform tr;
        while (!xml_.next().is("transform", TagKind::Close)) {
            if (at("translate", TagKind::SelfClosing)) {
                tr.o2w = tr.o2w * Mat34::translation(xyz());
                tr.w2o = Mat34::translation(xyz() * -1.0f) * tr.w2o;
            } else if (at("scale", TagKind::SelfClosing)) {
                tr.o2w = tr.o2w * Mat34::scaling(xyz());
                tr.w2o = Mat34::scaling(reciprocal(xyz())) * tr.w2o;
                tr.o2w_normal = tr.o2w_normal * Mat34::scaling(reciprocal(xyz()));
            } else if (at("rotateX", TagKind::SelfClosing)) {
                tr.o2w = tr.o2w * Mat34::rotation_x(number("theta") * 2 * kPi / 360.0f);
                tr.w2o = Mat34::rotation_x(-number("theta") * 2 * kPi / 360.0f) * tr.w2o;
                tr.o2w_normal = tr.o2w_normal * Mat34::rotation_x(number("theta") * 2 * kPi / 360.0f);
            } else if (at("rotateY", TagKind::SelfClosing)) {
                tr.o2w = tr.o2w * Mat34::rotation_y(number("theta") * 2 * kPi / 360.0f);
                tr.w2o = Mat34::rotation_y(-number("theta") * 2 * kPi / 360.0f) * tr.w2o;
                tr.o2w_normal = tr.o2w_normal * Mat34::rotation_y(number("theta") * 2 * kPi / 360.0f);
            } else if (at("rotateZ", TagKind::SelfClosing)) {
                tr.o2w = tr.o2w * Mat34::rotation_z(number("theta") * 2 * kPi / 360.0f);
                tr.w2o = Mat34::rotation_z(-number("theta") * 2 * kPi / 360.0f) * tr.w2o;
                tr.o2w_normal = tr.o2w_normal * Mat34::rotation_z(number("theta") * 2 * kPi / 360.0f);
            } else {
                throw std::runtime_error("unknown tag in transform");
            }
        }
        return tr;
    }

    MarkupReader xml_;
    std::string path_;
    float t_;
    SceneModel scene_;
    Vec3 cam_pos_, cam_look_, cam_up_, cam_focus_;
    std::map<std::string, int32_t> texture_by_file_;
};

}  // namespace

float evaluate_keyframes_for_test(const std::string& text, float t) { return evaluate_keyframes(text, t); }

SceneModel load_scene(const std::string& path, float t) {
    std::ifstream file(path);
    if (!file) throw std::runtime_error("scene file \"" + path + "\" could not be opened");
    return SceneGrammar(file, path, t).parse();
}

rt_scene_desc SceneModel::desc() const {
    rt_scene_desc d{};
    d.abi_version = RT_ABI_VERSION;
    d.camera = camera;
    std::memcpy(d.background, background, sizeof background);
    std::memcpy(d.ambient, ambient, sizeof ambient);
    d.n_lights = uint32_t(lights.size());
    d.lights = lights.data();
    d.n_objects = uint32_t(objects.size());
    d.objects = objects.data();
    d.n_spheres = uint32_t(spheres.size());
    d.spheres = spheres.data();
    d.n_triangles = uint32_t(triangles.size());
    d.triangles = triangles.data();
    d.n_julias = uint32_t(julias.size());
    d.julias = julias.data();
    d.n_materials = uint32_t(materials.size());
    d.materials = materials.data();
    d.n_textures = uint32_t(textures.size());
    d.textures = textures.data();
    d.n_texel_bytes = texels.size();
    d.texels = texels.data();
    d.dispersion_mode = dispersion_mode ? 1 : 0;
    // hue table presets live with the ABI (rt_hue_table) but the loader must not depend on the
    // GPU library, so the two constant tables are repeated here.
    static const float kIntAbs[8][3] = {{1, 0, 0}, {1, 1, 0}, {1, 1, 0}, {0, 1, 1},
                                        {0, 1, 1}, {0, 1, 1}, {1, 0, 1}, {1, 0, 1}};
    static const float kFloatAbs[8][3] = {{1, 0, 0},     {1, 0.75f, 0}, {0.5f, 1, 0}, {0, 1, 0.25f},
                                          {0, 1, 1},     {0, 0.25f, 1}, {0.5f, 0, 1}, {1, 0, 0.75f}};
    std::memcpy(d.hue_rgb, hue_preset == RT_HUE_FLOAT_ABS ? kFloatAbs : kIntAbs, sizeof d.hue_rgb);
    d.photon_steps = photon_steps;
    d.photon_texture_size = photon_texture_size;
    d.photon_factor = photon_factor;
    return d;
}

}  // namespace rta
