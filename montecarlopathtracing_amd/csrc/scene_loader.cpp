// .obj / .mtl / .camera readers with the reference's surface and quirks
// (MTPC/sceneManagement.cpp:17-274).  Host-side only; one-off per scene.
//
// Kept quirks: prefix matching at column 0, single-blank field splitting with atof/atoi, faces written
// a/b/c with the 2nd index addressing vn[] and the 3rd vt[] (sceneManagement.cpp:136-165), the third
// corner's vt index cut to the length of its vn index (sceneManagement.cpp:165), mtllib lines ignored
// (the .mtl name is derived from the scene name), per-material face lists in .obj order.
// Deliberate differences: '\r' stripped (D4), Ns/Ni default 1 (D8), malformed input returns an error
// instead of dereferencing NULL, textures are searched next to the scene before the cwd.
// Opt-in (mcpt_scene_load_ex; parity mode = no flag): MCPT_LOAD_STANDARD_OBJ reads faces the way the OBJ
// format defines them (v, v/vt, v//vn, v/vt/vn; relative indices; polygons as fans; any run of blanks separates fields),
// MCPT_LOAD_MTLLIB reads the .mtl files the .obj names, MCPT_LOAD_MORTON_BOUNDS keys the Morton order on the scene's own
// bounding box instead of the fixed [-1,4]^3 that clamps most of veach-mis into a few cells.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>

#include "jpeg_decoder.hpp"
#include "scene.hpp"

namespace mcpt {

int find_material(const Scene& s, const std::string& name)
{
    for (size_t i = 0; i < s.materials.size(); i++)
        if (s.materials[i].name == name) return int(i);
    return -1;
}

namespace {

bool next_line(std::ifstream& in, std::string& line)
{
    if (!std::getline(in, line)) return false;
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
    return true;
}

bool has_prefix(const std::string& s, const char* key) { return s.compare(0, std::strlen(key), key) == 0; }

// substr(n) that tolerates short lines (the reference would throw std::out_of_range)
std::string tail(const std::string& s, size_t n) { return n <= s.size() ? s.substr(n) : std::string(); }

// Splits "a b c" at the first blank twice and converts with atof: fields after the first two keep any
// further text (atof stops at the first non-number).
Vec3 three_numbers(std::string s)
{
    Vec3 r;
    size_t blank = s.find(' ');
    r.x = std::atof(s.substr(0, blank).c_str());
    s = blank == std::string::npos ? s : s.substr(blank + 1);
    blank = s.find(' ');
    r.y = std::atof(s.substr(0, blank).c_str());
    s = blank == std::string::npos ? s : s.substr(blank + 1);
    r.z = std::atof(s.c_str());
    return r;
}

bool read_ppm(const std::string& file, int& w, int& h, std::vector<uint8_t>& bgr)
{
    FILE* fp = std::fopen(file.c_str(), "rb");
    if (!fp) return false;
    char magic[3] = {0, 0, 0};
    int vals[3] = {0, 0, 0}, got = 0;
    bool ok = std::fscanf(fp, "%2s", magic) == 1 && std::strcmp(magic, "P6") == 0;
    while (ok && got < 3) {
        int c = std::fgetc(fp);
        if (c == EOF) ok = false;
        else if (c == '#') { while (c != '\n' && c != EOF) c = std::fgetc(fp); }
        else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
        else { std::ungetc(c, fp); ok = std::fscanf(fp, "%d", &vals[got]) == 1; got++; }
    }
    if (ok) {
        std::fgetc(fp);
        w = vals[0]; h = vals[1];
        ok = vals[2] == 255 && w > 0 && h > 0;
    }
    if (ok) {
        bgr.resize(size_t(w) * h * 3);
        ok = std::fread(bgr.data(), 1, bgr.size(), fp) == bgr.size();
        for (size_t i = 0; ok && i < bgr.size(); i += 3) std::swap(bgr[i], bgr[i + 2]);
    }
    std::fclose(fp);
    return ok;
}

// Material::readinMap (sceneManagement.h:134-143): cv::imread -> 8-bit BGR raster.
// The file named by map_Kd is read with the built-in decoders (JPEG by jpeg_decoder.cpp, binary PPM); it is looked up
// next to the scene first, then relative to the cwd (the reference: cwd only).  A "<name>.ppm" raster is accepted as a
// stand-in for formats that are not built in (PNG, BMP, ...).
bool load_texture(const std::string& dir, const std::string& name, MaterialRec& m, std::string& err)
{
    const std::string bases[2] = {dir + name, name};
    std::string why;
    for (const std::string& b : bases) {
        std::string jerr;
        if (decode_jpeg_file(b, m.map_w, m.map_h, m.bgr, jerr)) return true;
        if (why.empty() && jerr.rfind("cannot open", 0) != 0) why = " (" + jerr + ")";
        if (read_ppm(b, m.map_w, m.map_h, m.bgr)) return true;
        if (read_ppm(b + ".ppm", m.map_w, m.map_h, m.bgr)) return true;
    }
    err = "cannot read texture '" + name + "' next to the scene or in the cwd" + why;
    return false;
}


// scene_data::read_mtl, sceneManagement.cpp:17-74
int read_mtl(const std::string& file, const std::string& dir, Scene& s, std::string& err)
{
    std::ifstream in(file);
    if (!in) { err = "cannot open " + file; return MCPT_ERR_IO; }
    std::string line;
    int cur = -1;
    auto need = [&]() { if (cur < 0) { err = file + ": material property before newmtl"; return false; } return true; };
    while (next_line(in, line)) {
        if (has_prefix(line, "newmtl")) {
            std::string name = tail(line, 7);
            cur = find_material(s, name);               // material_map[name] = pm replaces an older entry
            if (cur < 0) { s.materials.emplace_back(); cur = int(s.materials.size()) - 1; }
            s.materials[cur] = MaterialRec();
            s.materials[cur].name = name;
        } else if (has_prefix(line, "Kd")) {
            if (!need()) return MCPT_ERR_PARSE;
            s.materials[cur].kd = three_numbers(tail(line, 3));
        } else if (has_prefix(line, "Ks")) {
            if (!need()) return MCPT_ERR_PARSE;
            s.materials[cur].ks = three_numbers(tail(line, 3));
        } else if (has_prefix(line, "Ns")) {
            if (!need()) return MCPT_ERR_PARSE;
            s.materials[cur].Ns = std::atof(tail(line, 3).c_str());
        } else if (has_prefix(line, "Ni")) {
            if (!need()) return MCPT_ERR_PARSE;
            s.materials[cur].Ni = std::atof(tail(line, 3).c_str());
        } else if (has_prefix(line, "map_Kd")) {
            if (!need()) return MCPT_ERR_PARSE;
            MaterialRec& m = s.materials[cur];
            m.has_map = true;
            if (!load_texture(dir, tail(line, 7), m, err)) return MCPT_ERR_IO;
        }
    }
    return MCPT_OK;
}

// one face into the scene (Face::calNorm and the Morton key of its centre, sceneManagement.cpp:176-179, :408-412)
void push_face(Scene& s, FaceRec f, int material)
{
    f.material = material;
    f.nrm = normalized(cross(f.v[0] - f.v[1], f.v[2] - f.v[0]));
    const Vec3 center = (f.v[0] + f.v[1] + f.v[2]) / 3;
    f.morton = morton_code(float(center.x), float(center.y), float(center.z));
    s.materials[material].faces.push_back(int32_t(s.faces.size()));
    s.faces.push_back(f);
}

// MCPT_LOAD_STANDARD_OBJ: "f" with any number of corners, each v, v/vt, v//vn or v/vt/vn, 1-based or negative (relative to
// the end of the list so far).  Polygons become the fan (0, i, i+1).  A corner without vn gets the face normal, without
// vt the texture coordinate (0, 0).
int standard_face(const std::string& rest, Scene& s, int material, std::string& err)
{
    struct Corner { long v, t, n; };
    std::vector<Corner> cs;
    const char* p = rest.c_str();
    auto resolve = [](long i, size_t n) -> long { return i > 0 ? i - 1 : (i < 0 ? long(n) + i : -1); };
    while (*p) {
        while (*p == ' ' || *p == '\t') p++;
        if (!*p) break;
        char* end = nullptr;
        Corner c{-1, -1, -1};
        long v = std::strtol(p, &end, 10);
        if (end == p) { err = "face " + std::to_string(s.faces.size()) + ": bad corner '" + std::string(p) + "'"; return MCPT_ERR_PARSE; }
        c.v = resolve(v, s.v.size());
        p = end;
        if (*p == '/') {
            p++;
            if (*p != '/' && *p != ' ' && *p != '\t' && *p) { long t = std::strtol(p, &end, 10); if (end != p) { c.t = resolve(t, s.vt.size()); if (c.t < 0) c.t = -2; } p = end; }
            if (*p == '/') { p++; long n = std::strtol(p, &end, 10); if (end != p) { c.n = resolve(n, s.vn.size()); if (c.n < 0) c.n = -2; } p = end; }
        }
        if (c.v < 0 || c.v >= long(s.v.size()) || c.t == -2 || c.t >= long(s.vt.size()) || c.n == -2 || c.n >= long(s.vn.size())) {
            err = "face " + std::to_string(s.faces.size()) + ": index out of range";
            return MCPT_ERR_PARSE;
        }
        cs.push_back(c);
    }
    if (cs.size() < 3) { err = "face " + std::to_string(s.faces.size()) + ": fewer than three corners"; return MCPT_ERR_PARSE; }
    for (size_t i = 1; i + 1 < cs.size(); i++) {
        const Corner tri[3] = {cs[0], cs[i], cs[i + 1]};
        FaceRec f{};
        for (int k = 0; k < 3; k++) f.v[k] = s.v[size_t(tri[k].v)];
        const Vec3 flat = normalized(cross(f.v[0] - f.v[1], f.v[2] - f.v[0]));
        for (int k = 0; k < 3; k++) {
            f.vn[k] = tri[k].n >= 0 ? s.vn[size_t(tri[k].n)] : flat;
            f.vt[k][0] = tri[k].t >= 0 ? s.vt[size_t(tri[k].t)].first : 0.0;
            f.vt[k][1] = tri[k].t >= 0 ? s.vt[size_t(tri[k].t)].second : 0.0;
        }
        push_face(s, f, material);
    }
    return MCPT_OK;
}

// whitespace-tolerant "x y z" (MCPT_LOAD_STANDARD_OBJ); missing numbers read as 0
Vec3 three_numbers_free(const std::string& t)
{
    const char* p = t.c_str();
    char* end = nullptr;
    double q[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) { q[i] = std::strtod(p, &end); if (end == p) break; p = end; }
    return Vec3{q[0], q[1], q[2]};
}

// scene_data::read_obj, sceneManagement.cpp:76-189
int read_obj(const std::string& file, int load_flags, Scene& s, std::string& err)
{
    std::ifstream in(file);
    if (!in) { err = "cannot open " + file; return MCPT_ERR_IO; }
    std::string line;
    int material = -1;
    const bool standard = (load_flags & MCPT_LOAD_STANDARD_OBJ) != 0;
    while (next_line(in, line)) {
        if (standard) {
            size_t b = line.find_first_not_of(" \t");
            if (b == std::string::npos || line[b] == '#') continue;
            if (b) line = line.substr(b);
            const bool sep2 = line.size() > 2 && (line[2] == ' ' || line[2] == '\t');
            const bool sep1 = line.size() > 1 && (line[1] == ' ' || line[1] == '\t');
            if (line[0] == 'v' && sep1) { s.v.push_back(three_numbers_free(line.substr(2))); continue; }
            if (line[0] == 'v' && line.size() > 1 && line[1] == 'n' && sep2) { s.vn.push_back(three_numbers_free(line.substr(3))); continue; }
            if (line[0] == 'v' && line.size() > 1 && line[1] == 't' && sep2) { const Vec3 q = three_numbers_free(line.substr(3)); s.vt.emplace_back(q.x, q.y); continue; }
            if (line[0] == 'f' && sep1) {
                if (material < 0) { err = "face before any usemtl"; return MCPT_ERR_PARSE; }
                const int rc = standard_face(line.substr(2), s, material, err);
                if (rc) return rc;
                continue;
            }
            if (has_prefix(line, "usemtl")) {
                std::string name = tail(line, 7);
                const size_t nb = name.find_first_not_of(" \t"), ne = name.find_last_not_of(" \t");
                name = nb == std::string::npos ? std::string() : name.substr(nb, ne - nb + 1);
                material = find_material(s, name);
                if (material < 0) { err = "usemtl '" + name + "' is not defined in the .mtl"; return MCPT_ERR_PARSE; }
            }
            continue;
        }
        const char c0 = line.size() > 0 ? line[0] : '\0', c1 = line.size() > 1 ? line[1] : '\0',
                   c2 = line.size() > 2 ? line[2] : '\0';
        if (c0 == 'v' && c1 == ' ') {
            s.v.push_back(three_numbers(line.substr(2)));
        } else if (c0 == 'v' && c1 == 'n' && c2 == ' ') {
            s.vn.push_back(three_numbers(line.substr(3)));
        } else if (c0 == 'v' && c1 == 't' && c2 == ' ') {
            std::string r = line.substr(3);
            size_t blank = r.find(' ');
            double a = std::atof(r.substr(0, blank).c_str());
            r = blank == std::string::npos ? r : r.substr(blank + 1);
            blank = r.find(' ');
            double b = std::atof(r.substr(0, blank).c_str());
            s.vt.emplace_back(a, b);
        } else if (has_prefix(line, "usemtl")) {
            std::string name = tail(line, 7);
            material = find_material(s, name);
            if (material < 0) { err = "usemtl '" + name + "' is not defined in the .mtl"; return MCPT_ERR_PARSE; }
        } else if (c0 == 'f' && c1 == ' ') {
            if (material < 0) { err = "face before any usemtl"; return MCPT_ERR_PARSE; }
            std::string r = line.substr(2);
            int idx[3][3];
            size_t slash = 0;
            for (int corner = 0; corner < 3; corner++) {
                slash = r.find('/');
                idx[corner][0] = std::atoi(r.substr(0, slash).c_str()) - 1;
                r = slash == std::string::npos ? r : r.substr(slash + 1);
                slash = r.find('/');
                idx[corner][1] = std::atoi(r.substr(0, slash).c_str()) - 1;
                r = slash == std::string::npos ? r : r.substr(slash + 1);
                if (corner < 2) {
                    size_t blank = r.find(' ');
                    idx[corner][2] = std::atoi(r.substr(0, blank).c_str()) - 1;
                    r = blank == std::string::npos ? r : r.substr(blank + 1);
                } else {
                    idx[corner][2] = std::atoi(r.substr(0, slash).c_str()) - 1;   // stale 'slash' as length
                }
            }
            FaceRec f{};
            for (int corner = 0; corner < 3; corner++) {
                const int iv = idx[corner][0], in_ = idx[corner][1], it = idx[corner][2];
                if (iv < 0 || iv >= int(s.v.size()) || in_ < 0 || in_ >= int(s.vn.size()) || it < 0 || it >= int(s.vt.size())) {
                    err = "face " + std::to_string(s.faces.size()) + ": index out of range";
                    return MCPT_ERR_PARSE;
                }
                f.v[corner] = s.v[iv];
                f.vn[corner] = s.vn[in_];           // 2nd index -> normals
                f.vt[corner][0] = s.vt[it].first;   // 3rd index -> texture coordinates
                f.vt[corner][1] = s.vt[it].second;
            }
            push_face(s, f, material);
        }
    }
    return MCPT_OK;
}

// scene_data::read_xml, sceneManagement.cpp:191-262
int read_camera(const std::string& file, Scene& s, std::string& err)
{
    std::ifstream in(file);
    if (!in) { err = "cannot open " + file; return MCPT_ERR_IO; }
    std::string line;
    while (next_line(in, line)) {
        if (has_prefix(line, "eye")) s.eye = three_numbers(tail(line, 4));
        else if (has_prefix(line, "lookat")) s.look_at = three_numbers(tail(line, 7));
        else if (has_prefix(line, "up")) s.up = three_numbers(tail(line, 3));
        else if (has_prefix(line, "fovy")) s.fovy = std::atof(tail(line, 5).c_str());
        else if (has_prefix(line, "width")) s.width = std::atoi(tail(line, 6).c_str());
        else if (has_prefix(line, "height")) s.height = std::atoi(tail(line, 7).c_str());
        else if (has_prefix(line, "mtlname")) {
            std::string r = tail(line, 8);
            LightRec l;
            size_t blank = r.find(' ');
            l.name = r.substr(0, blank);
            r = blank == std::string::npos ? r : r.substr(blank + 1);
            l.radiance = three_numbers(r);
            s.lights.push_back(l);
        }
    }
    return MCPT_OK;
}

}  // namespace

// Face::calAera, sceneManagement.cpp:399-406 (law of cosines, not a cross product)
double face_area(const FaceRec& f)
{
    const double a = norm(f.v[1] - f.v[0]), b = norm(f.v[2] - f.v[0]), c = norm(f.v[2] - f.v[1]);
    const double cos_c = (a * a + b * b - c * c) / (2 * a * b);
    const double sin_c = std::sqrt(1 - std::pow(cos_c, 2));
    return a * b * sin_c / 2;
}

// scene_data::read_scene, sceneManagement.cpp:264-274: .mtl, then .obj, then .camera
int load_scene_files(const std::string& path, const std::string& filename, int load_flags, Scene& s, std::string& err)
{
    const std::string base = path + filename;
    int rc = MCPT_OK;
    std::vector<std::string> libs;
    if (load_flags & MCPT_LOAD_MTLLIB) {                 // the files the .obj names, in order; the reference ignores the lines
        std::ifstream in(base + ".obj");
        if (!in) { err = "cannot open " + base + ".obj"; return MCPT_ERR_IO; }
        std::string line;
        while (next_line(in, line)) {
            const size_t b = line.find_first_not_of(" \t");
            if (b == std::string::npos || line.compare(b, 7, "mtllib ") != 0) continue;
            std::string names = line.substr(b + 7);
            size_t pos = 0;
            while (pos < names.size()) {
                const size_t e = names.find(' ', pos);
                const std::string one = names.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
                if (!one.empty()) libs.push_back(one);
                if (e == std::string::npos) break;
                pos = e + 1;
            }
        }
    }
    if (libs.empty()) rc = read_mtl(base + ".mtl", path, s, err);
    for (size_t i = 0; rc == MCPT_OK && i < libs.size(); i++) rc = read_mtl(path + libs[i], path, s, err);
    if (rc) return rc;
    rc = read_obj(base + ".obj", load_flags, s, err);
    if (rc) return rc;
    rc = read_camera(base + ".camera", s, err);
    if (rc) return rc;
    if ((load_flags & MCPT_LOAD_MORTON_BOUNDS) && !s.faces.empty()) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (const FaceRec& f : s.faces)
            for (int k = 0; k < 3; k++) {
                const double q[3] = {f.v[k].x, f.v[k].y, f.v[k].z};
                for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], q[a]); hi[a] = std::max(hi[a], q[a]); }
            }
        for (int a = 0; a < 3; a++) {
            s.morton_lo[a] = float(lo[a]);
            const float span = float(hi[a]) - s.morton_lo[a];
            s.morton_span[a] = span > 0.0f ? span : 1.0f;
        }
        for (FaceRec& f : s.faces) {
            const Vec3 c = (f.v[0] + f.v[1] + f.v[2]) / 3;
            f.morton = morton_code_in(float(c.x), float(c.y), float(c.z), s.morton_lo, s.morton_span);
        }
    }
    return finish_scene(s, base, err);
}

// what read_scene leaves behind once the three files are in: sanity checks, lights -> materials, light area tables
int finish_scene(Scene& s, const std::string& what, std::string& err)
{
    if (s.faces.empty()) { err = what + " has no faces"; return MCPT_ERR_PARSE; }
    if (s.width <= 0 || s.height <= 0) { err = what + " has no width/height"; return MCPT_ERR_PARSE; }
    for (size_t i = 0; i < s.lights.size(); i++) {
        LightRec& l = s.lights[i];
        if (l.material < 0) l.material = find_material(s, l.name);
        if (l.material < 0 || l.material >= int(s.materials.size())) { err = "light '" + l.name + "' names no material"; return MCPT_ERR_PARSE; }
        s.materials[l.material].light = int32_t(i);      // light_map[name]: the last entry wins
        const MaterialRec& m = s.materials[l.material];
        double total = 0;
        l.cdf.resize(m.faces.size());
        l.cdf_sorted = true;
        for (size_t j = 0; j < m.faces.size(); j++) {
            total += face_area(s.faces[m.faces[j]]);
            l.cdf[j] = total;
            if (!(total == total) || (j && !(l.cdf[j] >= l.cdf[j - 1]))) l.cdf_sorted = false;
        }
        l.total_area = total;
    }
    s.area0 = s.lights.empty() ? 0.0 : s.lights[0].total_area;
    return MCPT_OK;
}

// generateImg's camera set-up, pathTracing.cpp:276-294
CameraFrame camera_frame(const Scene& s)
{
    const double pi = 3.1415926;                                  // pathTracing.h:11
    CameraFrame cf;
    const Vec3 up = normalized(s.up);
    const Vec3 dir = s.look_at - s.eye;
    const double l = norm(dir);
    const double dy = std::tan(s.fovy / 2 / 180 * pi) * l;
    const double dx = dy / s.height * s.width;
    const double pdx = 2 * dx / s.width, pdy = 2 * dy / s.height;
    const Vec3 screen_x_dir = normalized(cross(dir, up));
    cf.screen_pdy = up * pdy;
    cf.screen_pdx = screen_x_dir * pdx;
    cf.start_point = (s.look_at - screen_x_dir * dx) + up * dy;
    cf.eye = s.eye;
    return cf;
}

}  // namespace mcpt
