// Host-side scene model of libmcpt: what the reference keeps in scene_data + BVH
// (MTPC/sceneManagement.h:173-199, MTPC/BVH.h:22-47), flattened to index-based records so it can be
// mirrored into HBM without pointer chasing or std::string compares.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mcpt.h"

namespace mcpt {

// fp64 3-vector with the reference's operation order (MTPC/sceneManagement.h:18-86).
struct Vec3 {
    double x = 0, y = 0, z = 0;
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, double t) { return {a.x * t, a.y * t, a.z * t}; }
inline Vec3 operator/(Vec3 a, double m) { return {a.x / m, a.y / m, a.z / m}; }
inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y}; }
inline double norm(Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline Vec3 normalized(Vec3 a) { double d = norm(a); return {a.x / d, a.y / d, a.z / d}; }

struct FaceRec {              // MTPC/sceneManagement.h:110-121
    Vec3 v[3], vn[3];
    double vt[3][2];
    Vec3 nrm;                 // Face::norm
    int32_t material;         // index instead of the reference's std::string
    uint32_t morton;
};

struct MaterialRec {          // MTPC/sceneManagement.h:123-148
    std::string name;
    Vec3 kd, ks;
    double Ns = 1, Ni = 1;    // D8
    bool has_map = false;
    int map_w = 0, map_h = 0;
    std::vector<uint8_t> bgr; // cv::Mat layout: rows x cols x BGR
    std::vector<int32_t> faces;   // Material::f, .obj order
    int32_t light = -1;       // light_map lookup
};

struct LightRec {             // MTPC/sceneManagement.h:158-163 (+ what shade() rebuilds per call)
    std::string name;
    Vec3 radiance;
    int32_t material = -1;
    double total_area = 0;
    std::vector<double> cdf;  // running triangle-area sum, pathTracing.cpp:177-184
    bool cdf_sorted = true;   // finite and non-decreasing -> device may binary-search it
};

struct NodeBox { double max_x, max_y, max_z, min_x, min_y, min_z; };   // boundingBox, sceneManagement.h:165-171

struct Scene {
    std::vector<Vec3> v, vn;
    std::vector<std::pair<double, double>> vt;
    std::vector<FaceRec> faces;           // .obj order
    std::vector<MaterialRec> materials;
    std::vector<LightRec> lights;
    Vec3 eye, look_at, up;
    double fovy = 0;
    int width = 0, height = 0;
    // after build_accel():
    std::vector<int32_t> order;           // leaf k -> .obj face (stable Morton order)
    mcpt_bvh_info bi{};
    std::vector<NodeBox> nodes;           // Nr real nodes, compact level order
    std::vector<int32_t> node_level, node_leaf;
    float morton_lo[3] = {-1.0f, -1.0f, -1.0f};   // Morton domain: the reference's fixed [-1,4]^3 (morton code.h:6-7) unless
    float morton_span[3] = {5.0f, 5.0f, 5.0f};    // the scene was loaded with MCPT_LOAD_MORTON_BOUNDS
    double area0 = 0;                     // total area of lights[0] (Q1)
    bool accel_built = false;             // false: Morton sort + BVH are left to the device (mcpt_device_create_ex)
};

// scene_loader.cpp
int load_scene_files(const std::string& path, const std::string& filename, int load_flags, Scene& out, std::string& err);
int finish_scene(Scene& s, const std::string& what, std::string& err);
int find_material(const Scene& s, const std::string& name);
// bvh_build.cpp
uint32_t morton_code(float x, float y, float z);
uint32_t morton_code_in(float x, float y, float z, const float lo[3], const float span[3]);
int find_index(const mcpt_bvh_info& b, int i, int l);
bool has_right_child(const mcpt_bvh_info& b, int node, int l);
int build_accel(Scene& s, std::string& err);
mcpt_bvh_info bvh_shape(int t);
double face_area(const FaceRec& f);
// png_writer.cpp
int64_t png_encode(const uint8_t* rgb8, int w, int h, uint8_t* out, int64_t cap);
// output_formats.cpp
int64_t png_encode_deflate(const uint8_t* rgb8, int w, int h, uint8_t* out, int64_t cap);
int write_pfm(const char* file, const double* img, int w, int h, std::string& err);
int checkpoint_save(const char* file, const double* img, int w, int h, int spp, uint64_t seed, uint64_t scene_tag, int parts,
                    const uint8_t* done, std::string& err);
int checkpoint_load(const char* file, double* img, int w, int h, int spp, uint64_t seed, uint64_t scene_tag, int parts, uint8_t* done,
                    std::string& err);
// camera frame of generateImg (pathTracing.cpp:276-294), shared by host and device code
struct CameraFrame { Vec3 eye, start_point, screen_pdx, screen_pdy; };
CameraFrame camera_frame(const Scene& s);

}  // namespace mcpt
