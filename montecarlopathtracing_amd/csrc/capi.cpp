// C ABI of libmcpt.so (include/mcpt.h): host scene handles, device residency, and the launch sequences that
// stand in for ray_intersect / generateImg / imshow / render_scene of the reference.
#include <hip/hip_runtime_api.h>
#include <hip/hip_version.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <functional>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "accel_build.hpp"
#include "build_kernels.hpp"
#include "jpeg_decoder.hpp"
#include "kernels.hpp"
#include "knobs.hpp"
#include "scene.hpp"
#include "wavefront.hpp"

using namespace mcpt;

namespace {
thread_local std::string g_error;
int fail(int code, const std::string& msg) { g_error = msg; return code; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(MCPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

template <class T, class A>
int upload(const std::vector<T, A>& h, T** d)
{
    *d = nullptr;
    const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(d), bytes));
    if (!h.empty()) HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return MCPT_OK;
}
template <class T>
int grow(T** ptr, int64_t* cap, int64_t need)
{
    if (*cap >= need) return MCPT_OK;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr; *cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(ptr), size_t(need) * sizeof(T)));
    *cap = need;
    return MCPT_OK;
}

}  // namespace

// the calling thread's error message, for the other translation units of the library (multi_device.cpp)
namespace mcpt { int set_error(int code, const std::string& msg) { return fail(code, msg); } }

struct mcpt_scene {
    Scene s;
    // The fast walk's culling hierarchy depends on the scene and the leaf order only: built once, shared by every device
    // created from this scene (one SAH build for the 8 GPUs of a node, not 8).
    mutable std::atomic<int> devices_created{0};      // mcpt_scene_set_resolution is refused once a device holds the camera
    // Shared ownership: the caller's handle and every device created from the scene hold one reference each; the scene goes with the
    // last of mcpt_scene_free / mcpt_device_free, in whichever order they come (a device keeps using the handle: the shared culling
    // hierarchy, the counter above).
    mutable std::atomic<int> refs{1};
    mutable std::mutex fast_mu;
    mutable std::shared_ptr<const FastBvh> fast_cached;
    mutable std::vector<int32_t> fast_order;
    mutable int fast_leaf = 0;
    mutable double fast_ct = 0;
};

// The hierarchy is always built for the deep stack (the better tree); MCPT_FAST_STACK_LIMIT builds it for a shallower one (A/B runs).
static int stack_limit_for(const Knobs& k) { return k.fast_stack_limit ? k.fast_stack_limit : kFastMaxDepth; }
static FastBuildOpts build_opts_for(const Knobs& k)
{
    FastBuildOpts o;
    if (k.fast_leaf) o.max_leaf = std::max(1, std::min(kFastMaxLeaf, k.fast_leaf));
    if (k.fast_ct > 0) o.cost_tri = k.fast_ct;
    o.serial = k.build_serial != 0; o.talk = k.print_diag != 0;
    return o;
}

// (built with the knobs of the device creation that asks first; a later one with other builder knobs rebuilds)
static std::shared_ptr<const FastBvh> shared_fast_bvh(const mcpt_scene* h, const std::vector<int32_t>& order, const Knobs& k)
{
    std::lock_guard<std::mutex> lock(h->fast_mu);
    const int limit = stack_limit_for(k);
    const FastBuildOpts o = build_opts_for(k);
    if (!h->fast_cached || h->fast_order != order || h->fast_cached->stack_limit != limit || h->fast_leaf != o.max_leaf || h->fast_ct != o.cost_tri) {
        auto fb = std::make_shared<FastBvh>();
        build_fast_bvh(h->s.faces, order.data(), int(h->s.faces.size()), *fb, limit, o);
        h->fast_cached = fb;
        h->fast_order = order;
        h->fast_leaf = o.max_leaf; h->fast_ct = o.cost_tri;
    }
    return h->fast_cached;
}

struct mcpt_device {
    int ordinal = 0;
    Knobs knobs;                           // the environment as it was when this device was created (knobs.hpp)
    DScene ds{};
    hipStream_t stream = nullptr;          // library stream for the host-pointer entry points
    // scene arrays
    DNode* nodes = nullptr; DTri* tris = nullptr; DTriShade* shade = nullptr; DMaterial* materials = nullptr;
    DLight* lights = nullptr; DLightTri* light_tris = nullptr; double* light_cdf = nullptr; uint8_t* texels = nullptr;
    FastNode* fast_nodes = nullptr; DTri* fast_tris = nullptr; CwNode* cw_nodes = nullptr; DTriPre* fast_pre = nullptr;
    size_t n_cw_nodes = 0;          // nodes in cw_nodes when the hierarchy was built on the device
    int trace_mode = MCPT_TRACE_FAST;
    int32_t* d_order = nullptr;            // leaf -> .obj face (device build keeps it for read-back)
    mcpt_bvh_info bi{};
    // frame state
    int width = 0, height = 0;
    double* dirs = nullptr;                // W*H*3 primary directions
    bool dirs_ready = false;
    // render workspace
    int32_t* pixels = nullptr; int64_t n_pixels = 0; int part_key[4] = {-1, -1, -1, -1};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t look_stream = nullptr;     // the host's looks at a path count travel here, so that they wait for the logic pass that wrote
    hipEvent_t look_ev = nullptr;          // the count and for nothing enqueued after it (the finishing kernel above all)
    unsigned int* h_look = nullptr;        // pinned host word the looks land in (never a pageable stack address: an async copy into
                                           // pageable memory goes through the runtime's pin-on-the-fly / staging paths)
    const mcpt_scene* scene = nullptr;     // the handle this device was created from (devices_created is given back in mcpt_device_free)
    // closest-hit and test entry points (mcpt_trace_closest*, mcpt_sample_radiance) have counters, queue words and a deferred-ray
    // list of their own: a frame in flight on another stream keeps using its frame slot's
    DCounters* aux_ctr = nullptr; TraceQueue* aux_queue = nullptr; long long* aux_slow_list = nullptr;
    size_t sample_budget_bytes = size_t(4) << 30;   // megakernel path: radiance staging buffer per chunk
    size_t wf_budget_bytes = 0;                     // path state + rays per frame slot; 0 = a share of the free HBM (MCPT_WORKSPACE_GB overrides)
    size_t wf_auto_budget = 0;                      // that share, asked for once (hipMemGetInfo costs a few hundred microseconds)
    // Everything a frame in flight owns.  Two slots: with MCPT_RENDER_PIPELINE consecutive frames alternate between them, so the
    // latency-bound tail of one frame (the finishing kernel's last long paths, the fold) overlaps the head of the next on another stream.
    struct FrameSlot {
        PrimaryHit* hits = nullptr; int64_t hits_cap = 0;
        double* rad = nullptr; size_t rad_cap = 0;
        void* wf_ws = nullptr; size_t wf_ws_bytes = 0;
        int32_t* hit_slots = nullptr; int64_t hit_slots_cap = 0;
        PrimarySurface* surf = nullptr; int64_t surf_cap = 0;   // first-vertex record per hit pixel of the chunk
        unsigned int* alive_base = nullptr; int64_t alive_base_cap = 0;   // shaded pixels before each group of 64 hit slots
        WfCounts* wf_counts = nullptr;                  // MCPT_WF_COUNT_SLOTS slots
        TraceQueue* queue = nullptr;                    // persistent trace kernels: chunk queue head + deferred-ray list
        long long* slow_list = nullptr;
        char* path_area = nullptr;                      // records and exact-walk stacks of the pool form of the finishing pass (finish_pool_bytes)
        DCounters* ctr = nullptr;
        hipEvent_t done = nullptr;                      // recorded after the slot's last kernel of a frame
        bool used = false;
        bool keeping = false;                           // ctr holds kept statistics of earlier frames (must not be cleared)
    } slot[2];
    int next_slot = 0;
    bool pipelined = false;                         // set by the first MCPT_RENDER_PIPELINE frame (sizes the workspace budget)
    // statistics kept on the device side until mcpt_device_collect_stats (MCPT_RENDER_KEEP_STATS)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;   // start/stop pairs around trace launches
    size_t ev_used = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> frame_ev;  // start/stop of every kept frame
    size_t frame_ev_used = 0;
    uint64_t kept_samples = 0, kept_primary = 0; int kept_launches = 0;
    unsigned int slow_cap = 1u << 20;
    LaunchCfg cfg;                                  // this GPU's resident grids and knobs
    long long finish_threshold = 500000;            // paths left at which the finishing pass takes over (MCPT_FINISH_PATHS; sweep: flat from 2e5 to 1e6)
};

extern "C" {

int mcpt_version(void) { return MCPT_VERSION; }
const char* mcpt_last_error(void) { return g_error.c_str(); }

const char* mcpt_knobs_describe(void) { return knobs_table(); }

int mcpt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------------ which HIP runtime is this?
// libmcpt.so's code objects are built by one hipcc; the libamdhip64.so they run on is whichever the process loaded first under that
// soname (a Python process that imported torch has the wheel's bundled runtime, not /opt/rocm's).  Kernels built by a newer compiler on
// an older runtime are the likeliest cause of the abort DESIGN 8a records, and nothing used to notice: now the first device creation
// compares the two versions and refuses when major.minor differ, unless the caller has said it knows (mcpt_allow_runtime_mismatch).
static std::atomic<int> g_allow_runtime_mismatch{0};

void mcpt_allow_runtime_mismatch(int32_t allow) { g_allow_runtime_mismatch.store(allow ? 1 : 0); }

// Pure comparison (CPU unit test): 0 when a library compiled against HIP `compiled` may run on runtime `runtime` (both encoded as
// HIP_VERSION: major * 10^7 + minor * 10^5 + patch), else MCPT_ERR_HIP with a message naming both and the runtime's file.
int mcpt_hip_runtime_check(int32_t compiled, int32_t runtime, const char* runtime_path, char* msg, int64_t cap)
{
    const int cmaj = compiled / 10000000, cmin = compiled / 100000 % 100, rmaj = runtime / 10000000, rmin = runtime / 100000 % 100;
    const bool ok = compiled > 0 && runtime > 0 && cmaj == rmaj && cmin == rmin;
    if (msg && cap > 0) {
        if (ok) msg[0] = 0;
        else
            std::snprintf(msg, size_t(cap),
                          "libmcpt.so was compiled against HIP %d.%d (%d) but this process runs on HIP runtime %d.%d (%d) loaded from %s: "
                          "another libamdhip64.so was loaded first (a Python process that imported torch has the wheel's).  Load libmcpt.so "
                          "before it, or call mcpt_allow_runtime_mismatch(1) / set MCPT_ALLOW_RUNTIME_MISMATCH=1 to run anyway",
                          cmaj, cmin, compiled, rmaj, rmin, runtime, (runtime_path && runtime_path[0]) ? runtime_path : "(unknown)");
    }
    return ok ? MCPT_OK : MCPT_ERR_HIP;
}

int mcpt_hip_runtime_info(int32_t* compiled, int32_t* runtime, char* path, int64_t cap)
{
    if (compiled) *compiled = HIP_VERSION;
    int rv = 0;
    if (hipRuntimeGetVersion(&rv) != hipSuccess) { (void)hipGetLastError(); rv = 0; }
    if (runtime) *runtime = rv;
    if (path && cap > 0) {
        path[0] = 0;
        Dl_info info;
        if (dladdr(reinterpret_cast<const void*>(static_cast<hipError_t (*)(int*)>(&hipRuntimeGetVersion)), &info) && info.dli_fname) std::snprintf(path, size_t(cap), "%s", info.dli_fname);
    }
    return MCPT_OK;
}

// what mcpt_device_create / mcpt_multi_create ask before they touch a device
static int runtime_gate()
{
    int32_t compiled = 0, runtime = 0;
    char path[512], msg[1024];
    mcpt_hip_runtime_info(&compiled, &runtime, path, sizeof path);
    if (mcpt_hip_runtime_check(compiled, runtime, path, msg, sizeof msg) == MCPT_OK) return MCPT_OK;
    if (g_allow_runtime_mismatch.load() || read_knobs().allow_runtime_mismatch) {
        static std::atomic<int> told{0};
        if (!told.exchange(1)) std::fprintf(stderr, "libmcpt: %s -- running anyway, as asked\n", msg);
        return MCPT_OK;
    }
    return fail(MCPT_ERR_HIP, msg);
}

// ------------------------------------------------------------------------------------------------ scene
int mcpt_scene_load(const char* path, const char* filename, mcpt_scene** out) { return mcpt_scene_load_ex(path, filename, 0, out); }

int mcpt_scene_load_ex(const char* path, const char* filename, int32_t load_flags, mcpt_scene** out)
{
    if (!path || !filename || !out) return fail(MCPT_ERR_ARG, "null argument");
    *out = nullptr;
    if (load_flags & ~(MCPT_LOAD_STANDARD_OBJ | MCPT_LOAD_MTLLIB | MCPT_LOAD_MORTON_BOUNDS)) return fail(MCPT_ERR_ARG, "unknown load flag");
    std::unique_ptr<mcpt_scene> h(new mcpt_scene);
    std::string err;
    int rc = load_scene_files(path, filename, load_flags, h->s, err);
    if (rc) return fail(rc, err);
    rc = build_accel(h->s, err);
    if (rc) return fail(rc, err);
    *out = h.release();
    return MCPT_OK;
}
static void scene_release(const mcpt_scene* s) { if (s && s->refs.fetch_sub(1) == 1) delete s; }
void mcpt_scene_free(mcpt_scene* s) { scene_release(s); }

int mcpt_scene_create(const mcpt_scene_desc* dsc, int32_t flags, mcpt_scene** out)
{
    if (!dsc || !out) return fail(MCPT_ERR_ARG, "null argument");
    *out = nullptr;
    if (dsc->num_faces <= 0 || dsc->num_faces > 0x3fffffff || !dsc->v || !dsc->vn || !dsc->material || dsc->num_materials <= 0 ||
        !dsc->material_rec || dsc->num_lights < 0 || (dsc->num_lights && (!dsc->light_material || !dsc->light_radiance)))
        return fail(MCPT_ERR_ARG, "incomplete scene description");
    std::unique_ptr<mcpt_scene> h(new mcpt_scene);
    Scene& s = h->s;
    s.materials.resize(size_t(dsc->num_materials));
    for (int m = 0; m < dsc->num_materials; m++) {
        MaterialRec& r = s.materials[m];
        const double* q = dsc->material_rec + size_t(m) * 8;
        r.name = (dsc->material_names && dsc->material_names[m]) ? dsc->material_names[m] : ("material" + std::to_string(m));
        r.kd = Vec3{q[0], q[1], q[2]}; r.ks = Vec3{q[3], q[4], q[5]}; r.Ns = q[6]; r.Ni = q[7];
    }
    const int64_t t = dsc->num_faces;
    s.faces.resize(size_t(t));
    for (int64_t i = 0; i < t; i++) {
        FaceRec& f = s.faces[size_t(i)];
        const int m = dsc->material[i];
        if (m < 0 || m >= dsc->num_materials) return fail(MCPT_ERR_PARSE, "face material index out of range");
        for (int c = 0; c < 3; c++) {
            f.v[c] = Vec3{dsc->v[i * 9 + c * 3], dsc->v[i * 9 + c * 3 + 1], dsc->v[i * 9 + c * 3 + 2]};
            f.vn[c] = Vec3{dsc->vn[i * 9 + c * 3], dsc->vn[i * 9 + c * 3 + 1], dsc->vn[i * 9 + c * 3 + 2]};
            f.vt[c][0] = dsc->vt ? dsc->vt[i * 6 + c * 2] : 0.0; f.vt[c][1] = dsc->vt ? dsc->vt[i * 6 + c * 2 + 1] : 0.0;
        }
        f.material = m;
        f.nrm = normalized(cross(f.v[0] - f.v[1], f.v[2] - f.v[0]));              // Face::calNorm
        const Vec3 center = (f.v[0] + f.v[1] + f.v[2]) / 3;
        f.morton = morton_code(float(center.x), float(center.y), float(center.z));
        s.materials[m].faces.push_back(int32_t(i));
    }
    s.lights.resize(size_t(dsc->num_lights));
    for (int l = 0; l < dsc->num_lights; l++) {
        LightRec& r = s.lights[l];
        r.material = dsc->light_material[l];
        if (r.material < 0 || r.material >= dsc->num_materials) return fail(MCPT_ERR_PARSE, "light material index out of range");
        r.name = s.materials[r.material].name;
        r.radiance = Vec3{dsc->light_radiance[l * 3], dsc->light_radiance[l * 3 + 1], dsc->light_radiance[l * 3 + 2]};
    }
    s.eye = Vec3{dsc->eye[0], dsc->eye[1], dsc->eye[2]}; s.look_at = Vec3{dsc->look_at[0], dsc->look_at[1], dsc->look_at[2]};
    s.up = Vec3{dsc->up[0], dsc->up[1], dsc->up[2]}; s.fovy = dsc->fovy; s.width = dsc->width; s.height = dsc->height;
    std::string err;
    int rc = finish_scene(s, "scene description", err);
    if (rc) return fail(rc, err);
    s.bi = bvh_shape(int(t));
    if (!(flags & MCPT_SCENE_DEFER_BUILD)) {
        rc = build_accel(s, err);
        if (rc) return fail(rc, err);
    }
    *out = h.release();
    return MCPT_OK;
}

int mcpt_scene_set_resolution(mcpt_scene* s, int32_t w, int32_t h)
{
    if (!s || w <= 0 || h <= 0) return fail(MCPT_ERR_ARG, "bad resolution");
    // a device caches the camera frame, the primary directions and its frame size when it is created; changing the resolution
    // under it would make callers size their frame buffers for another picture than the device writes
    if (s->devices_created.load() > 0 && (w != s->s.width || h != s->s.height))
        return fail(MCPT_ERR_ARG, "the resolution cannot change after a device has been created from the scene");
    s->s.width = w; s->s.height = h;
    return MCPT_OK;
}

int mcpt_scene_get_info(const mcpt_scene* h, mcpt_scene_info* o)
{
    if (!h || !o) return fail(MCPT_ERR_ARG, "null argument");
    const Scene& s = h->s;
    o->num_faces = int32_t(s.faces.size()); o->num_materials = int32_t(s.materials.size()); o->num_lights = int32_t(s.lights.size());
    o->width = s.width; o->height = s.height;
    o->eye[0] = s.eye.x; o->eye[1] = s.eye.y; o->eye[2] = s.eye.z;
    o->look_at[0] = s.look_at.x; o->look_at[1] = s.look_at.y; o->look_at[2] = s.look_at.z;
    o->up[0] = s.up.x; o->up[1] = s.up.y; o->up[2] = s.up.z;
    o->fovy = s.fovy; o->bvh = s.bi;
    return MCPT_OK;
}

int mcpt_scene_get_faces(const mcpt_scene* h, double* g, int32_t* material, uint32_t* morton)
{
    if (!h) return fail(MCPT_ERR_ARG, "null scene");
    const Scene& s = h->s;
    for (size_t i = 0; i < s.faces.size(); i++) {
        const FaceRec& f = s.faces[i];
        if (g) {
            double* o = g + i * 27;
            for (int c = 0; c < 3; c++) { o[c * 3] = f.v[c].x; o[c * 3 + 1] = f.v[c].y; o[c * 3 + 2] = f.v[c].z; }
            for (int c = 0; c < 3; c++) { o[9 + c * 3] = f.vn[c].x; o[9 + c * 3 + 1] = f.vn[c].y; o[9 + c * 3 + 2] = f.vn[c].z; }
            for (int c = 0; c < 3; c++) { o[18 + c * 2] = f.vt[c][0]; o[18 + c * 2 + 1] = f.vt[c][1]; }
            o[24] = f.nrm.x; o[25] = f.nrm.y; o[26] = f.nrm.z;
        }
        if (material) material[i] = f.material;
        if (morton) morton[i] = f.morton;
    }
    return MCPT_OK;
}

int mcpt_scene_get_leaf_order(const mcpt_scene* h, int32_t* o)
{
    if (!h || !o) return fail(MCPT_ERR_ARG, "null argument");
    if (!h->s.accel_built) return fail(MCPT_ERR_ARG, "scene has no host build (MCPT_SCENE_DEFER_BUILD): read the device's copy");
    std::copy(h->s.order.begin(), h->s.order.end(), o);
    return MCPT_OK;
}

int mcpt_scene_get_bvh_nodes(const mcpt_scene* h, double* box6, int32_t* level, int32_t* leaf_face)
{
    if (!h) return fail(MCPT_ERR_ARG, "null scene");
    const Scene& s = h->s;
    if (!s.accel_built) return fail(MCPT_ERR_ARG, "scene has no host build (MCPT_SCENE_DEFER_BUILD): read the device's copy");
    for (int i = 0; i < s.bi.Nr; i++) {
        const NodeBox& b = s.nodes[i];
        if (box6) { double* o = box6 + size_t(i) * 6; o[0] = b.max_x; o[1] = b.max_y; o[2] = b.max_z; o[3] = b.min_x; o[4] = b.min_y; o[5] = b.min_z; }
        if (level) level[i] = s.node_level[i];
        if (leaf_face) leaf_face[i] = s.node_leaf[i] >= 0 ? s.order[s.node_leaf[i]] : -1;
    }
    return MCPT_OK;
}

int mcpt_scene_find_index(const mcpt_scene* h, int32_t i, int32_t l) { return h ? find_index(h->s.bi, i, l) : -1; }

int mcpt_scene_get_material(const mcpt_scene* h, int32_t m, char name[64], double r[8], int32_t fl[4])
{
    if (!h || m < 0 || m >= int(h->s.materials.size())) return fail(MCPT_ERR_ARG, "material index");
    const MaterialRec& mt = h->s.materials[m];
    if (name) { std::memset(name, 0, 64); std::strncpy(name, mt.name.c_str(), 63); }
    if (r) { r[0] = mt.kd.x; r[1] = mt.kd.y; r[2] = mt.kd.z; r[3] = mt.ks.x; r[4] = mt.ks.y; r[5] = mt.ks.z; r[6] = mt.Ns; r[7] = mt.Ni; }
    if (fl) { fl[0] = mt.has_map; fl[1] = mt.map_w; fl[2] = mt.map_h; fl[3] = mt.light; }
    return MCPT_OK;
}

int mcpt_scene_get_light(const mcpt_scene* h, int32_t i, char name[64], double rad[3], int32_t* material, double* area)
{
    if (!h || i < 0 || i >= int(h->s.lights.size())) return fail(MCPT_ERR_ARG, "light index");
    const LightRec& l = h->s.lights[i];
    if (name) { std::memset(name, 0, 64); std::strncpy(name, l.name.c_str(), 63); }
    if (rad) { rad[0] = l.radiance.x; rad[1] = l.radiance.y; rad[2] = l.radiance.z; }
    if (material) *material = l.material;
    if (area) *area = l.total_area;
    return MCPT_OK;
}

uint32_t mcpt_morton_code(float x, float y, float z) { return morton_code(x, y, z); }

// Engine of the fast walk for a scene of t triangles (include/mcpt.h: mcpt_scene_trace_engine).  Measured on MI355X, frame times pool /
// vote: cornell-box (15 k triangles) 82.0 / 92.3 ms, veach-mis 147.5 / 160.7, one eighth of a cornell-box frame 13.9 / 15.3; the 204 k
// triangle interior 253 / 250, 10 M triangles 56.3 / 52.8: where the walk waits for memory, the pool engine's longer chain of dependent
// LDS and memory round trips per step costs what its fuller lanes save, or more.
static int trace_engine_for(long long t, const Knobs& k)
{
    if (k.trace_engine == 1) return (mcpt_device_count() > 0 && !pool_engine_available()) ? MCPT_ENGINE_VOTE : MCPT_ENGINE_POOL;
    if (k.trace_engine == 0) return MCPT_ENGINE_VOTE;
    if (t > k.pool_max_tris) return MCPT_ENGINE_VOTE;
    // (a device that cannot hold the pool engine's workgroup -- 1024 threads, 159 KB of LDS -- runs the voting engine; without a device
    // the answer is the policy's)
    if (mcpt_device_count() > 0 && !pool_engine_available()) {
        static std::atomic<int> told{0};
        if (!told.exchange(1)) std::fprintf(stderr, "libmcpt: this device cannot hold the pool engine's workgroup; the voting engine runs instead\n");
        return MCPT_ENGINE_VOTE;
    }
    return MCPT_ENGINE_POOL;
}

int mcpt_scene_trace_engine(const mcpt_scene* h)
{
    if (!h) return fail(MCPT_ERR_ARG, "null argument");
    return trace_engine_for((long long)h->s.faces.size(), read_knobs());      // (what a device created now would use)
}

int mcpt_scene_fast_bvh_stats(const mcpt_scene* h, int32_t* n_nodes, int32_t* max_depth, int32_t* leaf_order, int32_t* nesting_ok)
{
    if (!h) return fail(MCPT_ERR_ARG, "null scene");
    FastBvh fb;
    if (!h->s.accel_built) return fail(MCPT_ERR_ARG, "scene was created without a host build");
    { const Knobs k = read_knobs(); build_fast_bvh(h->s.faces, h->s.order.data(), h->s.bi.t, fb, stack_limit_for(k), build_opts_for(k)); }
    if (n_nodes) *n_nodes = int32_t(fb.nodes.size());
    if (max_depth) *max_depth = fb.max_depth;
    if (leaf_order) std::copy(fb.leaf_tris.begin(), fb.leaf_tris.end(), leaf_order);
    if (nesting_ok) {
        // every child box must contain what hangs below it: inner children by their own child boxes, leaves by the
        // reference's leaf boxes of their triangles
        const Scene& s = h->s;
        const int leaf0 = find_index(s.bi, (1 << s.bi.Level) - 1, s.bi.Level);
        bool ok = true;
        for (const FastNode& nd : fb.nodes)
            for (int c = 0; c < 2; c++) {
                const int32_t ref = nd.child[c];
                if (ref == kFastEmpty) continue;
                auto inside = [&](const double lo[3], const double hi[3]) {
                    for (int a = 0; a < 3; a++) if (lo[a] < nd.lo[c][a] || hi[a] > nd.hi[c][a]) ok = false;
                };
                if (ref >= 0) { inside(fb.nodes[ref].lo[0], fb.nodes[ref].hi[0]); if (fb.nodes[ref].child[1] != kFastEmpty) inside(fb.nodes[ref].lo[1], fb.nodes[ref].hi[1]); }
                else {
                    const int r = -1 - ref, first = r >> 4, count = (r & 7) + 1;
                    for (int i = 0; i < count; i++) {
                        const NodeBox& b = s.nodes[leaf0 + fb.leaf_tris[first + i]];
                        const double lo[3] = {b.min_x, b.min_y, b.min_z}, hi[3] = {b.max_x, b.max_y, b.max_z};
                        inside(lo, hi);
                    }
                }
            }
        // compressed nodes: every decoded child box must contain the fp64 box of what it refers to
        {
            std::vector<std::array<double, 6>> cwbox(fb.cw.size());     // fp64 box of each CwNode (union of its children's true boxes)
            std::vector<int> bin_of(fb.cw.size(), -1);
            // recompute true boxes bottom-up through the binary tree: box of a FastNode child is stored in its parent
            std::function<void(int, int, const double*, const double*)> walk;   // (cw node, unused, lo, hi)
            auto leaf_box = [&](int32_t ref, double lo[3], double hi[3]) {
                const int r = -1 - ref, first = r >> 4, count = (r & 7) + 1;
                for (int a = 0; a < 3; a++) { lo[a] = 1e300; hi[a] = -1e300; }
                for (int i = 0; i < count; i++) {
                    const NodeBox& b = s.nodes[leaf0 + fb.leaf_tris[first + i]];
                    const double l[3] = {b.min_x, b.min_y, b.min_z}, h2[3] = {b.max_x, b.max_y, b.max_z};
                    for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h2[a]); }
                }
            };
            std::function<void(int, double*, double*)> true_box = [&](int n, double* lo, double* hi) {
                for (int a = 0; a < 3; a++) { lo[a] = 1e300; hi[a] = -1e300; }
                const CwNode& nd = fb.cw[n];
                for (int c = 0; c < 4; c++) {
                    if (nd.child[c] == kFastEmpty) continue;
                    double cl[3], ch[3];
                    if (nd.child[c] >= 0) true_box(nd.child[c], cl, ch); else leaf_box(nd.child[c], cl, ch);
                    for (int a = 0; a < 3; a++) {
                        const double sc = std::ldexp(1.0, nd.e[a]);
                        const double dl = double(nd.p[a]) + double((nd.qlo[a] >> (8 * c)) & 255u) * sc;
                        const double dh = double(nd.p[a]) + double((nd.qhi[a] >> (8 * c)) & 255u) * sc;
                        if (dl > cl[a] || dh < ch[a]) ok = false;
                        lo[a] = std::min(lo[a], cl[a]); hi[a] = std::max(hi[a], ch[a]);
                    }
                }
            };
            double lo[3], hi[3];
            if (!fb.cw.empty()) true_box(0, lo, hi);
            // every triangle slot must be reachable exactly once
            std::vector<int> seen(fb.leaf_tris.size(), 0);
            for (const CwNode& nd : fb.cw)
                for (int c = 0; c < 4; c++)
                    if (nd.child[c] < 0 && nd.child[c] != kFastEmpty) {
                        const int r = -1 - nd.child[c], first = r >> 4, count = (r & 7) + 1;
                        for (int i = 0; i < count; i++) seen[first + i]++;
                    }
            for (int v : seen) if (v != 1) ok = false;
            if (fb.cw_stack_need >= kFastMaxDepth) ok = false;
        }
        *nesting_ok = ok ? 1 : 0;
    }
    return MCPT_OK;
}

// ------------------------------------------------------------------------------------------------ partition
static void tile_shape(const mcpt_render_params* p, int& tw, int& th, int& rank, int& world)
{
    tw = (p && p->tile_w > 0) ? p->tile_w : 32;
    th = (p && p->tile_h > 0) ? p->tile_h : 8;
    world = (p && p->world > 1) ? p->world : 1;
    rank = (p && world > 1) ? p->rank : 0;
}

// Tile (tx, ty) belongs to rank (tx + shift*ty) mod world, shift = the first integer >= world/2 that is coprime with
// world: consecutive tiles of a row go round-robin over the ranks and every tile row starts on a different rank, so no
// rank ends up with a fixed set of image columns (a plain "tile index mod world" does when the row length is a multiple
// of world -- 1280/32 = 40 tiles per row with 8 ranks -- and the empty sides of a frame then unbalance the ranks).
static int tile_shift(int world)
{
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    for (int s = std::max(1, world / 2); s < world; s++) if (gcd(s, world) == 1) return s;
    return 1;
}

static void owned_pixel_list(int W, int H, int tw, int th, int rank, int world, std::vector<int32_t>& out)
{
    out.clear();
    const int shift = tile_shift(world);
    for (int y = 0; y < H; y++) {
        const int ty = y / th;
        for (int x = 0; x < W; x++) {
            const int tx = x / tw;
            if ((tx + shift * ty) % world == rank) out.push_back(y * W + x);
        }
    }
}

int64_t mcpt_owned_pixels(const mcpt_scene* h, const mcpt_render_params* p, int32_t* pixels)
{
    if (!h) return fail(MCPT_ERR_ARG, "null scene");
    int tw, th, rank, world;
    tile_shape(p, tw, th, rank, world);
    if (rank < 0 || rank >= world) return fail(MCPT_ERR_ARG, "rank outside world");
    std::vector<int32_t> v;
    owned_pixel_list(h->s.width, h->s.height, tw, th, rank, world, v);
    if (pixels) std::copy(v.begin(), v.end(), pixels);
    return int64_t(v.size());
}

// ------------------------------------------------------------------------------------------------ device
void mcpt_device_free(mcpt_device* d)
{
    if (!d) return;
    (void)hipSetDevice(d->ordinal);
    (void)hipDeviceSynchronize();          // frames of a sequence may still be in flight on the caller's streams
    void* ptrs[] = {d->nodes, d->tris, d->shade, d->materials, d->lights, d->light_tris, d->light_cdf, d->texels, d->fast_nodes, d->fast_tris, d->fast_pre, d->cw_nodes, d->d_order,
                    d->dirs, d->pixels};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& f : d->slot) {
        void* q[] = {f.hits, f.rad, f.wf_ws, f.hit_slots, f.surf, f.alive_base, f.wf_counts, f.queue, f.slow_list, f.path_area, f.ctr};
        for (void* p : q) if (p) (void)hipFree(p);
        if (f.done) (void)hipEventDestroy(f.done);
    }
    for (hipEvent_t e : d->ev) if (e) (void)hipEventDestroy(e);
    for (auto& pr : d->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& pr : d->frame_ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (void* p : {static_cast<void*>(d->aux_ctr), static_cast<void*>(d->aux_queue), static_cast<void*>(d->aux_slow_list)}) if (p) (void)hipFree(p);
    if (d->h_look) (void)hipHostFree(d->h_look);
    if (d->look_stream) (void)hipStreamDestroy(d->look_stream);
    if (d->look_ev) (void)hipEventDestroy(d->look_ev);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    if (d->scene) { d->scene->devices_created.fetch_sub(1); scene_release(d->scene); }
    delete d;
}

int mcpt_device_create(const mcpt_scene* h, int32_t ordinal, mcpt_device** out)
{
    return mcpt_device_create_ex(h, ordinal, (h && !h->s.accel_built) ? MCPT_BUILD_DEVICE : MCPT_BUILD_HOST, out);
}

int mcpt_device_create_ex(const mcpt_scene* h, int32_t ordinal, int32_t build_mode, mcpt_device** out)
{
    if (!h || !out) return fail(MCPT_ERR_ARG, "null argument");
    *out = nullptr;
    if (build_mode != MCPT_BUILD_HOST && build_mode != MCPT_BUILD_DEVICE && build_mode != MCPT_BUILD_DEVICE_FAST && build_mode != MCPT_BUILD_DEVICE_SAH)
        return fail(MCPT_ERR_ARG, "bad build mode");
    const Scene& s = h->s;
    if (build_mode == MCPT_BUILD_HOST && !s.accel_built) return fail(MCPT_ERR_ARG, "scene has no host build; use MCPT_BUILD_DEVICE");
    int ndev = mcpt_device_count();
    if (ndev <= 0) return fail(MCPT_ERR_NO_DEVICE, "no HIP device available (libmcpt has no CPU fallback)");
    if (const int gate = runtime_gate()) return gate;          // kernels of one hipcc on another release's runtime: refused
    if (ordinal < 0 || ordinal >= ndev) return fail(MCPT_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(ordinal));
    std::unique_ptr<mcpt_device, void (*)(mcpt_device*)> d(new mcpt_device, mcpt_device_free);
    d->ordinal = ordinal;
    d->knobs = read_knobs();
    const Knobs& K = d->knobs;
    HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    for (auto& e : d->ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipStreamCreateWithFlags(&d->look_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&d->look_ev, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&d->h_look), 64, hipHostMallocDefault));

    const int t = int(s.faces.size());
    const mcpt_bvh_info bi = bvh_shape(t);
    d->bi = bi;
    int rc;
    const bool talk = K.print_diag && t >= (1 << 17);
    const auto t_create = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (talk) std::fprintf(stderr, "device create: %s at %.2f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_create).count());
    };
    std::vector<int32_t> order;                     // leaf -> .obj face
    const bool fast_on_device = build_mode == MCPT_BUILD_DEVICE_FAST || build_mode == MCPT_BUILD_DEVICE_SAH;
    if (build_mode == MCPT_BUILD_HOST) {
        std::vector<DNode> nodes(bi.Nr);
        for (int i = 0; i < bi.Nr; i++) {
            const NodeBox& b = s.nodes[i];
            DNode n{};
            n.mn[0] = b.min_x; n.mn[1] = b.min_y; n.mn[2] = b.min_z; n.mx[0] = b.max_x; n.mx[1] = b.max_y; n.mx[2] = b.max_z;
            nodes[i] = n;
        }
        std::vector<DTri> tris(t);
        std::vector<DTriShade> shade(t);
        for (int k = 0; k < t; k++) {
            const FaceRec& f = s.faces[s.order[k]];
            DTri q{};
            DTriShade a{};
            const Vec3* vs[3] = {&f.v[0], &f.v[1], &f.v[2]};
            double* dst[3] = {q.v1, q.v2, q.v3};
            for (int c = 0; c < 3; c++) { dst[c][0] = vs[c]->x; dst[c][1] = vs[c]->y; dst[c][2] = vs[c]->z; }
            q.n[0] = f.nrm.x; q.n[1] = f.nrm.y; q.n[2] = f.nrm.z;
            q.material = f.material; q.face = s.order[k]; q.leaf = k;
            double* nd[3] = {a.vn1, a.vn2, a.vn3};
            for (int c = 0; c < 3; c++) { nd[c][0] = f.vn[c].x; nd[c][1] = f.vn[c].y; nd[c][2] = f.vn[c].z; }
            a.vt1[0] = f.vt[0][0]; a.vt1[1] = f.vt[0][1]; a.vt2[0] = f.vt[1][0]; a.vt2[1] = f.vt[1][1]; a.vt3[0] = f.vt[2][0]; a.vt3[1] = f.vt[2][1];
            tris[k] = q; shade[k] = a;
        }
        order = s.order;
        if ((rc = upload(nodes, &d->nodes)) || (rc = upload(tris, &d->tris)) || (rc = upload(shade, &d->shade)) || (rc = upload(order, &d->d_order)))
            return rc;
    } else {
        // faces in .obj order -> HBM, then Morton keys, stable sort, leaf records and the level-by-level union on the GPU
        // (no zero fill: 2.2 GB at 10 M triangles, every element is written below)
        std::vector<double, default_init_alloc<double>> v9(size_t(t) * 9), vn9(size_t(t) * 9), vt6(size_t(t) * 6), nrm3(size_t(t) * 3);
        std::vector<int32_t, default_init_alloc<int32_t>> mat(static_cast<size_t>(t));
        parallel_pieces(t, [&](long long ib, long long ie) {
        for (long long i = ib; i < ie; i++) {
            const FaceRec& f = s.faces[size_t(i)];
            for (int c = 0; c < 3; c++) {
                v9[size_t(i) * 9 + c * 3] = f.v[c].x; v9[size_t(i) * 9 + c * 3 + 1] = f.v[c].y; v9[size_t(i) * 9 + c * 3 + 2] = f.v[c].z;
                vn9[size_t(i) * 9 + c * 3] = f.vn[c].x; vn9[size_t(i) * 9 + c * 3 + 1] = f.vn[c].y; vn9[size_t(i) * 9 + c * 3 + 2] = f.vn[c].z;
                vt6[size_t(i) * 6 + c * 2] = f.vt[c][0]; vt6[size_t(i) * 6 + c * 2 + 1] = f.vt[c][1];
            }
            nrm3[size_t(i) * 3] = f.nrm.x; nrm3[size_t(i) * 3 + 1] = f.nrm.y; nrm3[size_t(i) * 3 + 2] = f.nrm.z;
            mat[size_t(i)] = f.material;
        }
        });
        lap("faces staged");
        double *d_v9 = nullptr, *d_vn9 = nullptr, *d_vt6 = nullptr, *d_nrm3 = nullptr;
        int32_t* d_mat = nullptr;
        auto drop = [&]() { (void)hipFree(d_v9); (void)hipFree(d_vn9); (void)hipFree(d_vt6); (void)hipFree(d_nrm3); (void)hipFree(d_mat); };
        if ((rc = upload(v9, &d_v9)) || (rc = upload(vn9, &d_vn9)) || (rc = upload(vt6, &d_vt6)) || (rc = upload(nrm3, &d_nrm3)) || (rc = upload(mat, &d_mat))) { drop(); return rc; }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&d->nodes), size_t(bi.Nr) * sizeof(DNode));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d->tris), size_t(t) * sizeof(DTri));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d->shade), size_t(t) * sizeof(DTriShade));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d->d_order), size_t(t) * sizeof(int32_t));
        if (e == hipSuccess) {
            BuildInputs in{d_v9, d_vn9, d_vt6, d_nrm3, d_mat, t, {s.morton_lo[0], s.morton_lo[1], s.morton_lo[2]},
                           {s.morton_span[0], s.morton_span[1], s.morton_span[2]}};
            e = device_build_reference(in, bi, d->nodes, d->tris, d->shade, d->d_order, d->stream);
        }
        order.resize(t);
        if (e == hipSuccess) e = hipMemcpy(order.data(), d->d_order, size_t(t) * sizeof(int32_t), hipMemcpyDeviceToHost);
        drop();
        if (e != hipSuccess) return fail(MCPT_ERR_HIP, std::string("device build: ") + hipGetErrorString(e));
    }

    lap("reference structures in HBM");
    std::vector<uint8_t> texels;
    std::vector<DMaterial> mats(s.materials.size());
    for (size_t i = 0; i < s.materials.size(); i++) {
        const MaterialRec& m = s.materials[i];
        DMaterial dm{};
        dm.kd[0] = m.kd.x; dm.kd[1] = m.kd.y; dm.kd[2] = m.kd.z; dm.ks[0] = m.ks.x; dm.ks[1] = m.ks.y; dm.ks[2] = m.ks.z;
        dm.Ns = m.Ns; dm.Ni = m.Ni; dm.has_map = m.has_map; dm.map_w = m.map_w; dm.map_h = m.map_h; dm.light = m.light;
        dm.tex_offset = int64_t(texels.size());
        texels.insert(texels.end(), m.bgr.begin(), m.bgr.end());
        mats[i] = dm;
    }
    std::vector<DLight> lights(s.lights.size());
    std::vector<DLightTri> ltris;
    std::vector<double> lcdf;
    for (size_t i = 0; i < s.lights.size(); i++) {
        const LightRec& l = s.lights[i];
        const MaterialRec& m = s.materials[l.material];
        DLight dl{};
        dl.radiance[0] = l.radiance.x; dl.radiance[1] = l.radiance.y; dl.radiance[2] = l.radiance.z;
        dl.total_area = l.total_area; dl.material = l.material; dl.ntri = int32_t(m.faces.size());
        dl.first = int32_t(ltris.size()); dl.cdf_sorted = l.cdf_sorted ? 1 : 0;
        for (size_t j = 0; j < m.faces.size(); j++) {
            const FaceRec& f = s.faces[m.faces[j]];
            DLightTri q{};
            double* pv[3] = {q.v1, q.v2, q.v3};
            double* pn[3] = {q.vn1, q.vn2, q.vn3};
            for (int c = 0; c < 3; c++) {
                pv[c][0] = f.v[c].x; pv[c][1] = f.v[c].y; pv[c][2] = f.v[c].z;
                pn[c][0] = f.vn[c].x; pn[c][1] = f.vn[c].y; pn[c][2] = f.vn[c].z;
            }
            ltris.push_back(q);
            lcdf.push_back(l.cdf[j]);
        }
        lights[i] = dl;
    }
    if ((rc = upload(mats, &d->materials)) || (rc = upload(lights, &d->lights)) || (rc = upload(ltris, &d->light_tris)) ||
        (rc = upload(lcdf, &d->light_cdf)) || (rc = upload(texels, &d->texels)))
        return rc;

    // result-identical fast structure: SAH hierarchy built on the host from the leaf order (accel_build.cpp), permuted triangle
    // copy gathered on the GPU -- or, MCPT_BUILD_DEVICE_FAST, a 4-wide tree over the Morton order built on the GPU in place
    std::shared_ptr<const FastBvh> fb_host;
    if (!fast_on_device) { fb_host = shared_fast_bvh(h, order, K); lap("culling hierarchy on the host"); }
    FastBvh fb_dev;                              // MCPT_BUILD_DEVICE_FAST: shape figures of the hierarchy built on this GPU
    const FastBvh& fb_ro = fast_on_device ? fb_dev : *fb_host;
    FastBvh& fb = fb_dev;
    bool coords_ok = true;                       // every coordinate zero or within [1e-150, 1e150]
    for (const FaceRec& f : s.faces)
        for (int c = 0; c < 3; c++)
            for (double v : {f.v[c].x, f.v[c].y, f.v[c].z}) {
                const double a = std::fabs(v);
                if (!(a == 0.0 || (a >= 1e-150 && a <= 1e150))) coords_ok = false;
            }
    if (fast_on_device) {
        int n_cw = 0, levels = 0;
        double amax = 0;
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        for (const FaceRec& f : s.faces)
            for (int c = 0; c < 3; c++) {
                const double q[3] = {f.v[c].x, f.v[c].y, f.v[c].z};
                for (int a = 0; a < 3; a++) { if (q[a] < blo[a]) blo[a] = q[a]; if (q[a] > bhi[a]) bhi[a] = q[a]; }
            }
        // Clusters of Morton-consecutive triangles on the GPU (by default one compressed node over four single-triangle leaves:
        // every triangle keeps its own quantised box), a SAH tree over the clusters' boxes on the host.  Sweep on MI355X
        // (MCPT_CLUSTER_LEAF x MCPT_CLUSTER_LEVELS, ms per frame synthetic 10 M SPP 16 / cornell-box): 1x1 87 / 143, 1x2 93 / 161,
        // 1x3 103 / 182, 2x1 116 / 177, 4x2 163 / 238; the host's full SAH tree: 56 / 110.
        const int kPerLeaf = K.cluster_leaf, kClusterLevels = K.cluster_levels;
        CwNode* d_lower = nullptr;
        int n_top = 0;
        std::vector<double> top_boxes;
        bool ploc_fell_back = false;
        if (build_mode == MCPT_BUILD_DEVICE_SAH) {
            // clusters grown by locally-ordered clustering on the GPU (build_kernels.hip: device_build_ploc), the host's SAH tree over them
            const int kCluster = K.ploc_cluster, kHeightEnv = K.ploc_height;
            // how tall a cluster may grow: what the walk's stack leaves once the tree over the expected number of clusters has its levels
            int kHeight = kHeightEnv;
            if (!kHeight) {
                const long long est = std::max<long long>(1, 2ll * t / kCluster);
                int lv = 1;
                while ((1ll << lv) < est) lv++;
                kHeight = std::max(6, std::min(20, 35 - 5 - lv));
            }
            const int kRadius = K.ploc_radius, kLeaf = K.ploc_leaf ? K.ploc_leaf : kFastDefaultLeaf, kBudget = K.ploc_budget;
            const double kAreaDen = K.ploc_area, kCt = K.ploc_ct, kCl = K.ploc_cl;
            std::vector<int32_t> top_roots;
            int lower_need = 0, rounds = 0;
            hipError_t e = device_build_ploc(d->tris, t, blo, bhi, kCluster, kHeight, kRadius, kLeaf, kAreaDen > 0 ? 1.0 / kAreaDen : 0.0, kCt, kCl, kBudget, &d_lower, &d->fast_tris, &n_cw, &n_top, &top_boxes,
                                             &top_roots, &lower_need, &amax, &rounds, d->stream);
            if (e == hipErrorNotSupported) ploc_fell_back = true;
            else {
            if (e != hipSuccess) return fail(MCPT_ERR_HIP, std::string("device build of the fast hierarchy (clustering): ") + hipGetErrorString(e));
            lap("clusters on the GPU");
            if (talk) std::fprintf(stderr, "device create: %d clusters in %d rounds, %d nodes below them, stack need below a cluster root %d\n", n_top, rounds, n_cw, lower_need);
            fb.scene_absmax = amax;
            if (n_top == 1) {
                d->cw_nodes = d_lower;
                d->n_cw_nodes = size_t(n_cw);
                fb.max_depth = lower_need;
                fb.cw_stack_need = lower_need;
            } else {
                FastBvh up;
                build_fast_upper(top_boxes.data(), n_top, lower_need, up);
                const int n_up = int(up.cw.size());
                for (CwNode& nd : up.cw)
                    for (int c = 0; c < 4; c++)
                        if (nd.child[c] < 0 && nd.child[c] != kFastEmpty) {            // cluster -> its root node, or its triangles if it is one leaf
                            const int32_t r = top_roots[size_t(-1 - nd.child[c])];
                            nd.child[c] = r >= 0 ? n_up + r : r;
                        }
                e = hipMalloc(reinterpret_cast<void**>(&d->cw_nodes), size_t(n_up + n_cw) * sizeof(CwNode));
                if (e == hipSuccess) e = hipMemcpy(d->cw_nodes, up.cw.data(), size_t(n_up) * sizeof(CwNode), hipMemcpyHostToDevice);
                if (e == hipSuccess) e = hipMemcpyAsync(d->cw_nodes + n_up, d_lower, size_t(n_cw) * sizeof(CwNode), hipMemcpyDeviceToDevice, d->stream);
                if (e == hipSuccess) e = device_offset_children(d->cw_nodes + n_up, n_cw, n_up, d->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
                (void)hipFree(d_lower);
                if (e != hipSuccess) return fail(MCPT_ERR_HIP, std::string("device build of the fast hierarchy: ") + hipGetErrorString(e));
                d->n_cw_nodes = size_t(n_up) + size_t(n_cw);
                fb.max_depth = up.max_depth + lower_need;
                fb.cw_stack_need = up.cw_stack_need;             // includes lower_need
            }
            }
        }
        if (build_mode != MCPT_BUILD_DEVICE_SAH || ploc_fell_back) {
        hipError_t e = device_build_fast(d->tris, t, blo, bhi, kPerLeaf, kClusterLevels, &d_lower, &d->fast_tris, &n_cw, &levels, &n_top, &top_boxes, &amax, d->stream);
        if (e != hipSuccess) return fail(MCPT_ERR_HIP, std::string("device build of the fast hierarchy: ") + hipGetErrorString(e));
        fb.scene_absmax = amax;
        if (n_top == 1) {                        // small scene: the GPU's tree is the whole tree
            d->cw_nodes = d_lower;
            d->n_cw_nodes = size_t(n_cw);
            fb.max_depth = levels;
            fb.cw_stack_need = 3 * levels;       // three siblings pushed per level on the way down
        } else {
            FastBvh up;
            build_fast_upper(top_boxes.data(), n_top, 3 * levels, up);
            const int n_up = int(up.cw.size());
            for (CwNode& nd : up.cw)
                for (int c = 0; c < 4; c++)
                    if (nd.child[c] < 0 && nd.child[c] != kFastEmpty) nd.child[c] = n_up + (-1 - nd.child[c]);   // cluster -> its root node
            e = hipMalloc(reinterpret_cast<void**>(&d->cw_nodes), size_t(n_up + n_cw) * sizeof(CwNode));
            if (e == hipSuccess) e = hipMemcpy(d->cw_nodes, up.cw.data(), size_t(n_up) * sizeof(CwNode), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpyAsync(d->cw_nodes + n_up, d_lower, size_t(n_cw) * sizeof(CwNode), hipMemcpyDeviceToDevice, d->stream);
            if (e == hipSuccess) e = device_offset_children(d->cw_nodes + n_up, n_cw, n_up, d->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
            (void)hipFree(d_lower);
            if (e != hipSuccess) return fail(MCPT_ERR_HIP, std::string("device build of the fast hierarchy: ") + hipGetErrorString(e));
            d->n_cw_nodes = size_t(n_up) + size_t(n_cw);
            fb.max_depth = up.max_depth + levels;
            fb.cw_stack_need = up.cw_stack_need;             // includes the clusters' 3 * levels
        }
        }
    } else {
        int32_t* d_slots = nullptr;
        if ((rc = upload(fb_ro.cw, &d->cw_nodes)) || (rc = upload(fb_ro.leaf_tris, &d_slots))) { (void)hipFree(d_slots); return rc; }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&d->fast_tris), std::max<size_t>(fb_ro.leaf_tris.size(), 1) * sizeof(DTri));
        if (e == hipSuccess) e = device_gather_tris(d->tris, d_slots, int(fb_ro.leaf_tris.size()), d->fast_tris, d->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
        (void)hipFree(d_slots);
        if (e != hipSuccess) return fail(MCPT_ERR_HIP, std::string("fast triangle gather: ") + hipGetErrorString(e));
    }
    // The pre-test pays where the walk is bound by instruction issue, i.e. where nodes and triangles come out of L1 / L2 / the 256-MB
    // Infinity Cache (cornell-box: 7.38 -> 7.25 ms per k_wf_trace launch; veach-mis and the 204 k-triangle interior alike).  On the
    // 10 M-triangle scene the walk waits for memory, and a second dependent fetch per leaf (48-B record, then the 128-B record of a
    // survivor) costs more than the skipped arithmetic saves: 6.90 vs 6.44 ms per launch.  So: records only for scenes of at most
    // MCPT_PRE_TEST_MAX_TRIS triangles (default 2^20: ~200 B per triangle of nodes, records and triangles stay cache-resident).
    const long long pre_max = K.pre_test_max_tris;
    if (t <= pre_max) {
        // fp32 records of the triangle phase's pre-test, one per slot of the fast triangle array
        const int n_slots = fast_on_device ? t : int(fb_ro.leaf_tris.size());
        // (four records of padding: the pre-test reads its triangles in rounds of up to four slots, used or not)
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&d->fast_pre), (size_t(n_slots) + 4) * sizeof(DTriPre));
        if (e == hipSuccess) e = hipMemsetAsync(d->fast_pre + n_slots, 0, 4 * sizeof(DTriPre), d->stream);
        if (e == hipSuccess) e = device_build_pre(d->fast_tris, n_slots, fb_ro.scene_absmax, d->fast_pre, d->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
        if (e != hipSuccess) return fail(MCPT_ERR_HIP, std::string("pre-test records: ") + hipGetErrorString(e));
    }
    lap("culling hierarchy in HBM");
    if (K.slow_list) d->slow_cap = unsigned(K.slow_list);   // tests shrink it to force the overflow path
    init_launch_cfg(d->cfg, K.logic_grid, K.trace_block_rays, K.trace_min_chunk, K.trace_max_chunk);
    d->cfg.trace_pool = trace_engine_for(t, K) == MCPT_ENGINE_POOL ? 1 : 0;
    // the pool engine keeps the stack entries of a ray beyond those it has in LDS in an area behind the deferred-ray list of the launch
    const size_t spill_bytes = d->cfg.trace_pool ? pool_spill_bytes(d->cfg.cus) : 0;
    // ... and finishes a frame's last paths in path mode (MCPT_FINISH_ENGINE=lane: the one-lane-per-path kernel, for A/B runs)
    d->cfg.finish_pool = d->cfg.trace_pool;
    if (K.finish_engine == 0) d->cfg.finish_pool = 0;
    const size_t path_bytes = d->cfg.finish_pool ? finish_pool_bytes(d->cfg.cus, int(s.lights.size())) : 0;     // (0: a path's rays do not fit a lane's slots)
    if (!path_bytes) d->cfg.finish_pool = 0;
    for (auto& f : d->slot) {
        if (path_bytes) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f.path_area), path_bytes));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f.ctr), sizeof(DCounters)));
        HIP_TRY(hipMemset(f.ctr, 0, sizeof(DCounters)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f.wf_counts), sizeof(WfCounts) * MCPT_WF_COUNT_SLOTS));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f.queue), sizeof(TraceQueue)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f.slow_list), size_t(d->slow_cap) * sizeof(long long) + spill_bytes));
        HIP_TRY(hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->aux_ctr), sizeof(DCounters)));
    HIP_TRY(hipMemset(d->aux_ctr, 0, sizeof(DCounters)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->aux_queue), sizeof(TraceQueue)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->aux_slow_list), size_t(d->slow_cap) * sizeof(long long) + spill_bytes));
    // paths left at which the finishing pass takes over: the pool form holds the wavefront kernels' pace further up (sweep on one eighth of
    // the headline frame, ms: 250 k 14.2, 500 k 13.3, 1 M 13.1, 2 M 13.0, 4 M 13.5, 8 M 14.8; whole frame 81.0 / 80.3 at 500 k / 2 M), the
    // one-lane-per-path form is flat from 2e5 to 1e6
    d->finish_threshold = path_bytes ? 1500000 : 500000;
    if (K.finish_paths >= 0) d->finish_threshold = K.finish_paths;
    if (K.workspace_gb > 0) d->wf_budget_bytes = size_t(K.workspace_gb * double(size_t(1) << 30));

    DScene& S = d->ds;
    S.nodes = d->nodes; S.tris = d->tris; S.shade = d->shade; S.materials = d->materials; S.lights = d->lights;
    S.light_tris = d->light_tris; S.light_cdf = d->light_cdf; S.texels = d->texels;
    S.t = t; S.Lv = bi.Lv; S.Level = bi.Level; S.Nr = bi.Nr;
    S.num_lights = int32_t(s.lights.size()); S.num_materials = int32_t(s.materials.size());
    S.area0 = s.area0;
    S.fast.cw = d->cw_nodes; S.fast.nodes = nullptr; S.fast.tris = d->fast_tris; S.fast.pre = d->fast_pre; S.fast.absmax = fb_ro.scene_absmax;
    S.fast.enabled = (coords_ok && fb_ro.max_depth < kFastMaxDepth && fb_ro.cw_stack_need < kFastMaxDepth && fb_ro.scene_absmax >= 1e-15 &&
                      fb_ro.scene_absmax <= 1e15) ? 1 : 0;
    // Which shape of the trace engine walks it (wavefront.hip): by default the short-stack one at 4 waves per SIMD -- the hierarchy may
    // need up to kFastMaxDepth - 1 entries in the worst case, but a ray that would push past entry 27 is simply handed to the one-lane
    // walk (deep stack), and on every scene measured none does (10 M triangles: 0 of 1.5e8 rays).  MCPT_SHORT_KERNEL=0: the deep-stack
    // engine at 3 waves per SIMD.
    S.fast.stack_limit = kFastShortStack;
    {
        // (any prefix of the node array may be mirrored; the host builder puts the top of the tree there)
        const size_t n_cw_total = fast_on_device ? d->n_cw_nodes : fb_ro.cw.size();
        S.fast.cached = int32_t(std::min<size_t>(n_cw_total, size_t(kFastTopNodes)));
        if (K.node_cache >= 0 && K.node_cache < S.fast.cached) S.fast.cached = K.node_cache;
    }
    if (!K.short_kernel) S.fast.stack_limit = kFastMaxDepth;
    S.fast.stack_cap = S.fast.stack_limit;
    if (K.test_stack_cap >= 4 && K.test_stack_cap < S.fast.stack_cap) S.fast.stack_cap = K.test_stack_cap;
    const CameraFrame cf = camera_frame(s);
    S.cam.eye[0] = cf.eye.x; S.cam.eye[1] = cf.eye.y; S.cam.eye[2] = cf.eye.z;
    S.cam.start_point[0] = cf.start_point.x; S.cam.start_point[1] = cf.start_point.y; S.cam.start_point[2] = cf.start_point.z;
    S.cam.pdx[0] = cf.screen_pdx.x; S.cam.pdx[1] = cf.screen_pdx.y; S.cam.pdx[2] = cf.screen_pdx.z;
    S.cam.pdy[0] = cf.screen_pdy.x; S.cam.pdy[1] = cf.screen_pdy.y; S.cam.pdy[2] = cf.screen_pdy.z;
    S.cam.width = s.width; S.cam.height = s.height;
    d->width = s.width; d->height = s.height;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->dirs), size_t(s.width) * s.height * 3 * sizeof(double)));
    h->devices_created.fetch_add(1);
    h->refs.fetch_add(1);
    d->scene = h;
    *out = d.release();
    return MCPT_OK;
}

// what the device holds, read back (parity of the device build against the host build)
int mcpt_device_get_bvh_nodes(mcpt_device* d, double* box6, int32_t* leaf_face)
{
    if (!d) return fail(MCPT_ERR_ARG, "null device");
    HIP_TRY(hipSetDevice(d->ordinal));
    const mcpt_bvh_info& bi = d->bi;
    if (box6) {
        std::vector<DNode> nodes(bi.Nr);
        HIP_TRY(hipMemcpy(nodes.data(), d->nodes, size_t(bi.Nr) * sizeof(DNode), hipMemcpyDeviceToHost));
        for (int i = 0; i < bi.Nr; i++) {
            double* o = box6 + size_t(i) * 6;
            o[0] = nodes[i].mx[0]; o[1] = nodes[i].mx[1]; o[2] = nodes[i].mx[2]; o[3] = nodes[i].mn[0]; o[4] = nodes[i].mn[1]; o[5] = nodes[i].mn[2];
        }
    }
    if (leaf_face) {
        std::vector<int32_t> order(bi.t);
        HIP_TRY(hipMemcpy(order.data(), d->d_order, size_t(bi.t) * sizeof(int32_t), hipMemcpyDeviceToHost));
        const int leaf0 = find_index(bi, (1 << bi.Level) - 1, bi.Level);
        for (int i = 0; i < bi.Nr; i++) leaf_face[i] = (i >= leaf0 && i < leaf0 + bi.t) ? order[i - leaf0] : -1;
    }
    return MCPT_OK;
}

int mcpt_device_get_leaf_order(mcpt_device* d, int32_t* leaf_to_face)
{
    if (!d || !leaf_to_face) return fail(MCPT_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(d->ordinal));
    HIP_TRY(hipMemcpy(leaf_to_face, d->d_order, size_t(d->bi.t) * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

int mcpt_device_set_trace_mode(mcpt_device* d, int32_t mode)
{
    if (!d || (mode != MCPT_TRACE_FAST && mode != MCPT_TRACE_REFERENCE)) return fail(MCPT_ERR_ARG, "bad trace mode");
    d->trace_mode = mode;
    return MCPT_OK;
}

static int ensure_dirs(mcpt_device* d, hipStream_t st)
{
    if (!d->dirs_ready) {
        launch_primary_dirs(d->ds.cam, d->dirs, st);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
        d->dirs_ready = true;
    }
    return MCPT_OK;
}

static void counters_to_stats(const DCounters& c, mcpt_stats* s, bool print_diag)
{
    s->rays_primary = c.rays_primary; s->rays_shadow = c.rays_shadow; s->rays_bounce = c.rays_bounce;
    s->node_visits = c.node_visits; s->tri_tests = c.tri_tests; s->shade_calls = c.shade_calls; s->samples = c.samples;
    s->shadow_skipped = c.shadow_skipped;
    s->dom_rays = c.trace_rays; s->dom_node_visits = c.trace_nodes; s->dom_tri_tests = c.trace_tris;
    if (print_diag) {
        const double tot = double(c.pad[8] + c.pad[9] + c.pad[10] + c.pad[11]);
        const double iters = double(c.pad[0] + c.pad[2] + c.pad[4]);
        auto per = [](unsigned long long a, unsigned long long b) { return b ? double(a) / double(b) : 0.0; };
        std::fprintf(stderr, "trace diag: inner iters %llu lanes %.1f/64 | pre-test iters %llu lanes %.1f/64 | exact iters %llu lanes %.1f/64 | idle lanes/iter %.1f | "
                             "wave time: refill %.1f%% inner %.1f%% pre-test %.1f%% exact %.1f%% | cycles per iter: inner %.0f pre-test %.0f exact %.0f\n",
                     c.pad[0], per(c.pad[1], c.pad[0]), c.pad[2], per(c.pad[3], c.pad[2]), c.pad[4], per(c.pad[5], c.pad[4]), iters ? double(c.pad[6]) / iters : 0.0,
                     tot ? 100.0 * c.pad[8] / tot : 0.0, tot ? 100.0 * c.pad[9] / tot : 0.0, tot ? 100.0 * c.pad[10] / tot : 0.0, tot ? 100.0 * c.pad[11] / tot : 0.0,
                     per(c.pad[9], c.pad[0]), per(c.pad[10], c.pad[2]), per(c.pad[11], c.pad[4]));
        std::fprintf(stderr, "k_wf_trace: %llu rays, %.3f nodes, %.3f triangles visited, %.3f exact tests per ray (%.1f %% of the visited triangles survive the pre-test)\n",
                     c.trace_rays, per(c.trace_nodes, c.trace_rays), per(c.trace_tris, c.trace_rays), per(c.trace_exact, c.trace_rays), 100.0 * per(c.trace_exact, c.trace_tris));
        std::fprintf(stderr, "rays deferred to the exact walk by k_wf_trace: %llu of %llu\n", c.pad[12], c.trace_rays);
#ifdef MCPT_POOL_DEBUG
        if (c.pp[19]) {
            static const char* nm[5] = {"node", "leaf", "exact", "result", "shade"};
            const double life = double(c.pp[18]);
            for (int i = 0; i < 5; i++)
                std::fprintf(stderr, "pool %-6s: %10llu steps, %5.1f lanes per step, %7.0f cycles per step, %5.1f %% of wave time\n", nm[i], c.pp[i],
                             c.pp[i] ? double(c.pp[5 + i]) / c.pp[i] : 0.0, c.pp[i] ? double(c.pp[12 + i]) / c.pp[i] : 0.0, life ? 100.0 * c.pp[12 + i] / life : 0.0);
            std::fprintf(stderr, "pool: %llu waves, %.0f cycles per wave, vote + claim + sleep %.1f %% of wave time, %llu sleeps, %llu steps that claimed nothing\n", c.pp[19],
                         life / c.pp[19], life ? 100.0 * c.pp[17] / life : 0.0, c.pp[10], c.pp[11]);
        }
        for (int i = 0; i < 4; i++) std::fprintf(stderr, "pool class %d: %llu steps, %.1f lanes per step (%.1f could before the claim)\n", i, c.dbg[8 + i], c.dbg[8 + i] ? double(c.dbg[12 + i]) / c.dbg[8 + i] : 0.0, c.dbg[8 + i] ? double(c.dbg[16 + i]) / c.dbg[8 + i] : 0.0);
        std::fprintf(stderr, "pool: %llu sleeps, %llu steps that claimed nothing\n", c.dbg[20], c.dbg[21]);
        std::fprintf(stderr, "pool debug: %llu launches, %llu slots in all, %llu consumed in %llu refill steps, %llu rays among them, %llu started, %llu slots retired, %llu tickets\n", c.dbg[5], c.dbg[4], c.dbg[0], c.dbg[2], c.dbg[1], c.dbg[7], c.dbg[3], c.dbg[6]);
#endif
        if (c.pad[20]) {
            std::fprintf(stderr, "PRE-TEST SELF-CHECK: %llu rejected triangles are candidates by the exact test\n", c.pad[20]);
            double g[24]; std::memcpy(g, c.dbg, sizeof g);
            std::fprintf(stderr, "  first: margins beta %.6g gamma %.6g alpha %.6g behind %.6g beyond %.6g clear %.6g | t32 %.9g |det| %.6g | t_k %.17g leader %.17g limit_f %.9g margin %.6g eta4 %.6g slot %.0f of %.0f\n"
                                 "  ray o %.17g %.17g %.17g d %.17g %.17g %.17g\n",
                         g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[13], g[14], g[15], g[16], g[17], g[18], g[19], g[20], g[21]);
        }
        if (c.pad[13]) std::fprintf(stderr, "finish diag: longest wave %llu steps, %.0f us alive, %.0f us of it in the ray walks (100 MHz ticks; maxima over waves and launches)\n",
                                    c.pad[13], double(c.pad[14]) / 100.0, double(c.pad[15]) / 100.0);
        const double lt = double(c.pad[16] + c.pad[17] + c.pad[18]);
        std::fprintf(stderr, "logic diag: resolve %.1f%% compaction %.1f%% shade %.1f%% | cycles per wave: %.0f / %.0f / %.0f (waves %llu)\n",
                     lt ? 100.0 * c.pad[16] / lt : 0.0, lt ? 100.0 * c.pad[17] / lt : 0.0, lt ? 100.0 * c.pad[18] / lt : 0.0,
                     c.pad[19] ? double(c.pad[16]) / c.pad[19] : 0.0, c.pad[19] ? double(c.pad[17]) / c.pad[19] : 0.0, c.pad[19] ? double(c.pad[18]) / c.pad[19] : 0.0, c.pad[19]);
    }
}

// ------------------------------------------------------------------------------------------------ closest hit
int mcpt_trace_closest_device(mcpt_device* d, const double* d_rays, int64_t n, int32_t* d_face, double* d_t, double* d_p,
                              double* d_pn, void* stream)
{
    if (!d || (n > 0 && !d_rays) || n < 0) return fail(MCPT_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(d->ordinal));
    if (!d_face || !d_t || !d_p) return fail(MCPT_ERR_ARG, "d_face, d_t and d_p are required by the device form");
    launch_trace_closest(d->ds, d->trace_mode == MCPT_TRACE_FAST, d_rays, n, d_face, d_t, d_p, d_pn, d->aux_ctr, d->aux_queue, d->aux_slow_list, d->slow_cap,
                         static_cast<hipStream_t>(stream), d->cfg);
    HIP_TRY(hipGetLastError());
    return MCPT_OK;
}

int mcpt_trace_closest(mcpt_device* d, const double* rays, int64_t n, int32_t* face, double* t, double* p, double* pn, mcpt_stats* stats)
{
    if (!d || (n > 0 && !rays) || n < 0) return fail(MCPT_ERR_ARG, "bad argument");
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return MCPT_OK;
    HIP_TRY(hipSetDevice(d->ordinal));
    double *d_rays = nullptr, *d_t = nullptr, *d_p = nullptr, *d_pn = nullptr;
    int32_t* d_face = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_rays); (void)hipFree(d_t); (void)hipFree(d_p); (void)hipFree(d_pn); (void)hipFree(d_face); };
#define TRY_OR_CLEAN(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(MCPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_rays), size_t(n) * 6 * sizeof(double)));
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_face), size_t(n) * sizeof(int32_t)));
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_t), size_t(n) * sizeof(double)));
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_p), size_t(n) * 3 * sizeof(double)));
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_pn), size_t(n) * 3 * sizeof(double)));
    // host buffers are pageable: blocking copies (the runtime stages them), ordered around the kernels by stream synchronisation
    TRY_OR_CLEAN(hipMemcpy(d_rays, rays, size_t(n) * 6 * sizeof(double), hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemsetAsync(d->aux_ctr, 0, sizeof(DCounters), d->stream));
    TRY_OR_CLEAN(hipEventRecord(d->ev[0], d->stream));
    launch_trace_closest(d->ds, d->trace_mode == MCPT_TRACE_FAST, d_rays, n, d_face, d_t, d_p, d_pn, d->aux_ctr, d->aux_queue, d->aux_slow_list, d->slow_cap, d->stream, d->cfg);
    TRY_OR_CLEAN(hipGetLastError());
    TRY_OR_CLEAN(hipEventRecord(d->ev[1], d->stream));
    TRY_OR_CLEAN(hipStreamSynchronize(d->stream));
    if (face) TRY_OR_CLEAN(hipMemcpy(face, d_face, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (t) TRY_OR_CLEAN(hipMemcpy(t, d_t, size_t(n) * sizeof(double), hipMemcpyDeviceToHost));
    if (p) TRY_OR_CLEAN(hipMemcpy(p, d_p, size_t(n) * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (pn) TRY_OR_CLEAN(hipMemcpy(pn, d_pn, size_t(n) * 3 * sizeof(double), hipMemcpyDeviceToHost));
    DCounters c{};
    TRY_OR_CLEAN(hipMemcpy(&c, d->aux_ctr, sizeof c, hipMemcpyDeviceToHost));
    if (stats) {
        counters_to_stats(c, stats, d->knobs.print_diag != 0);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, d->ev[0], d->ev[1]);
        stats->ms_trace = ms; stats->ms_total = ms; stats->launches = 1;
    }
    cleanup();
#undef TRY_OR_CLEAN
    return MCPT_OK;
}

// ------------------------------------------------------------------------------------------------ integrator
static int prepare_partition(mcpt_device* d, const mcpt_render_params* p, hipStream_t st)
{
    int tw, th, rank, world;
    tile_shape(p, tw, th, rank, world);
    if (rank < 0 || rank >= world) return fail(MCPT_ERR_ARG, "rank outside world");
    const int key[4] = {tw, th, rank, world};
    if (std::memcmp(key, d->part_key, sizeof key) == 0 && d->pixels) return MCPT_OK;
    std::vector<int32_t> v;
    owned_pixel_list(d->width, d->height, tw, th, rank, world, v);
    if (d->pixels) {
        HIP_TRY(hipDeviceSynchronize());       // a frame of the previous partition may still be in flight (MCPT_RENDER_KEEP_STATS / PIPELINE)
        (void)hipFree(d->pixels); d->pixels = nullptr;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->pixels), std::max<size_t>(v.size(), 1) * sizeof(int32_t)));
    if (!v.empty()) HIP_TRY(hipMemcpy(d->pixels, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));   // blocking: v is pageable
    d->n_pixels = int64_t(v.size());
    std::memcpy(d->part_key, key, sizeof key);
    return MCPT_OK;
}

// megakernel path: one lane per camera sample, the whole path in one kernel (kept for A/B runs and as a second
// implementation the wavefront path is checked against)
static int render_megakernel(mcpt_device* d, mcpt_device::FrameSlot& f, const mcpt_render_params* p, double* d_img, bool timed, hipStream_t st,
                             double& ms_trace, int& launches)
{
    const int64_t npx = d->n_pixels;
    const int spp = p->spp;
    const size_t per_pixel = size_t(spp) * 3 * sizeof(double);
    int64_t chunk = int64_t(std::max<size_t>(d->sample_budget_bytes / per_pixel, 64));
    chunk = std::min<int64_t>(chunk, npx);
    if (f.rad_cap < size_t(chunk) * per_pixel) {
        if (f.rad) (void)hipFree(f.rad);
        f.rad = nullptr; f.rad_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f.rad), size_t(chunk) * per_pixel));
        f.rad_cap = size_t(chunk) * per_pixel;
    }
    for (int64_t first = 0; first < npx; first += chunk) {
        const int n_slots = int(std::min<int64_t>(chunk, npx - first));
        if (timed) HIP_TRY(hipEventRecord(d->ev[2], st));
        launch_shade_samples(d->ds, p->seed, d->dirs, d->pixels, f.hits, int(first), n_slots, spp, f.rad, f.ctr, st);
        HIP_TRY(hipGetLastError());
        if (timed) {
            HIP_TRY(hipEventRecord(d->ev[3], st));
            HIP_TRY(hipEventSynchronize(d->ev[3]));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, d->ev[2], d->ev[3]));
            ms_trace += ms;
        }
        launches++;
        launch_fold_samples(f.rad, d->pixels, f.hits, int(first), n_slots, spp, d_img, st);
        HIP_TRY(hipGetLastError());
    }
    return MCPT_OK;
}

// wavefront path (wavefront.hpp): per chunk, lockstep iterations of logic + trace over compacted path state in HBM.
// timed: event pairs around the trace launches, summed here (one stream synchronisation at the end); keep: the pairs are recorded
// and left in d->ev_pool for mcpt_device_collect_stats -- the frame ends without the host waiting for it.
static int render_wavefront(mcpt_device* d, mcpt_device::FrameSlot& f, const mcpt_render_params* p, double* d_img, bool timed, bool keep, hipStream_t st,
                            double& ms_trace, int& launches)
{
    const int64_t npx = d->n_pixels;
    const int spp = p->spp;
    const int nl = d->ds.num_lights;
    const bool fast = d->trace_mode == MCPT_TRACE_FAST;
    const size_t bpp = wf_bytes_per_path(nl);
    // chunk: as many pixels as the workspace budget holds paths for (every pixel may hit)
    const size_t overhead = 64 * 1024;
    // Fewer, larger chunks are cheaper (every chunk ends in a tail of small launches): by default a frame slot may use half of
    // the HBM that is free (a third when two frames are pipelined), which holds a whole 1280x720 SPP-256 frame (83 GB) on a
    // 288-GB device.
    size_t budget = d->wf_budget_bytes;
    if (!budget) {
        if (!d->wf_auto_budget) {
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            size_t mine = 0;
            for (const auto& q : d->slot) mine += q.wf_ws_bytes + q.rad_cap;
            d->wf_auto_budget = std::max<size_t>((free_b + mine) / (d->pipelined ? 3 : 2), size_t(1) << 30);
        }
        budget = d->wf_auto_budget;
    }
    int64_t cap = int64_t((budget - overhead) / (bpp + 24));      // + 24 B radiance per sample
    cap = std::min<int64_t>(cap, npx * int64_t(spp));
    cap = std::min<int64_t>(cap, (int64_t(1) << 31) - 4096);                 // 32-bit compaction counter / sample ids
    int64_t chunk_slots = std::max<int64_t>(cap / spp, 1);
    chunk_slots = std::min<int64_t>(chunk_slots, npx);
    cap = chunk_slots * spp;
    const size_t ws_need = size_t(cap) * bpp + overhead;
    if (f.wf_ws_bytes < ws_need) {
        if (f.wf_ws) (void)hipFree(f.wf_ws);
        f.wf_ws = nullptr; f.wf_ws_bytes = 0;
        HIP_TRY(hipMalloc(&f.wf_ws, ws_need));
        f.wf_ws_bytes = ws_need;
    }
    const size_t rad_need = size_t(cap) * 3 * sizeof(double);
    if (f.rad_cap < rad_need) {
        if (f.rad) (void)hipFree(f.rad);
        f.rad = nullptr; f.rad_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f.rad), rad_need));
        f.rad_cap = rad_need;
    }
    int rc = grow(&f.hit_slots, &f.hit_slots_cap, chunk_slots);
    if (rc) return rc;
    if ((rc = grow(&f.surf, &f.surf_cap, chunk_slots))) return rc;
    if ((rc = grow(&f.alive_base, &f.alive_base_cap, chunk_slots / 64 + 2))) return rc;
    WfArgs a{};
    WfState A, B;
    if (!wf_carve(f.wf_ws, f.wf_ws_bytes, cap, nl, A, B, a.rays)) return fail(MCPT_ERR_NOMEM, "wavefront workspace too small");
    a.cap = cap; a.nl = nl; a.spp = spp; a.seed = p->seed; a.pixels = d->pixels; a.hit_slots = f.hit_slots; a.surf = f.surf; a.alive_base = f.alive_base; a.hits = f.hits;
    a.dirs = d->dirs; a.rad = f.rad; a.counts = f.wf_counts; a.ctr = f.ctr; a.tris = d->tris; a.materials = d->materials; a.queue = fast ? f.queue : nullptr;
    a.finish_below = fast ? unsigned(std::min<long long>(std::max<long long>(d->finish_threshold, 0), 1ll << 30)) : 0u;
    // Iterations are enqueued without waiting for their counts: every kernel reads its input count from the device slot the
    // previous one wrote.  The host looks at a count only every few iterations (to stop, and to size the next grids).
    const size_t ev_first = d->ev_used;
    auto next_pair = [&](std::pair<hipEvent_t, hipEvent_t>*& out) -> int {
        if (d->ev_used == d->ev_pool.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            d->ev_pool.emplace_back(e0, e1);
        }
        out = &d->ev_pool[d->ev_used++];
        return MCPT_OK;
    };
    const int kSyncEvery = 4;
    for (int64_t first = 0; first < npx; first += chunk_slots) {
        const int n_slots = int(std::min<int64_t>(chunk_slots, npx - first));
        HIP_TRY(hipMemsetAsync(f.wf_counts, 0, sizeof(WfCounts) * MCPT_WF_COUNT_SLOTS, st));
        launch_hit_slots(f.hits, int(first), n_slots, f.hit_slots, &f.wf_counts[0].n_next, st);
        HIP_TRY(hipGetLastError());
        long long n_upper = (long long)n_slots * spp;        // upper bound of the live paths, refined at every look
        double n_grid = double(n_upper);                     // grid-sizing estimate between looks (kernels stride, any grid is correct)
        a.first_slot = int(first);
        a.in = A; a.out = B;
        a.counts_in = &f.wf_counts[0];
        launch_primary_surface(d->ds, a, f.surf, f.alive_base, &f.wf_counts[0].pad[2], n_slots, st);      // what the samples of a pixel share at their first vertex
        HIP_TRY(hipGetLastError());
        for (int depth = 0; depth < MCPT_MAX_DEPTH && n_upper > 0; depth++) {
            a.depth = depth;
            a.counts_in = &f.wf_counts[depth]; a.count_mul = depth == 0 ? unsigned(spp) : 1u;
            a.counts = &f.wf_counts[depth + 1];
            const long long n_launch = std::max<long long>(1, (long long)n_grid);
            launch_wf_logic(d->ds, a, n_launch, depth == 0, st, d->cfg);
            HIP_TRY(hipGetLastError());
            // The host looks at this pass's count every few iterations, and at every iteration once the hand-over to the finishing
            // kernel is near.  The look waits for this logic pass only (event + side stream): when it finds the hand-over, the
            // finishing kernel is launched and the call returns while it runs -- the next frame's head can overlap it.
            const bool look = (depth + 1) % kSyncEvery == 0 || (a.finish_below && n_grid * 0.6 <= 6.0 * double(a.finish_below));
            if (look) {
                HIP_TRY(hipEventRecord(d->look_ev, st));
                HIP_TRY(hipStreamWaitEvent(d->look_stream, d->look_ev, 0));
                HIP_TRY(hipMemcpyAsync(d->h_look, &f.wf_counts[depth + 1].n_next, sizeof(unsigned int), hipMemcpyDeviceToHost, d->look_stream));
                HIP_TRY(hipStreamSynchronize(d->look_stream));
                const unsigned int n_now = *d->h_look;
                if (n_now <= a.finish_below) {
                    if (n_now > 0) { launch_wf_finish(d->ds, a, (long long)n_now, st, d->cfg, f.path_area, f.slow_list, d->slow_cap); HIP_TRY(hipGetLastError()); }
                    n_upper = 0;
                    break;
                }
                n_upper = n_now;
                n_grid = double(n_now);
            } else if (a.finish_below) {
                // few paths left (decided on the device from this pass's count): one lane per path runs them to the end
                launch_wf_finish(d->ds, a, std::min<long long>(n_launch, (long long)a.finish_below), st, d->cfg, f.path_area, f.slow_list, d->slow_cap);
                HIP_TRY(hipGetLastError());
            }
            const long long n_trace = look ? (long long)n_grid : n_launch;
            std::pair<hipEvent_t, hipEvent_t>* pr = nullptr;
            if (timed || keep) { if ((rc = next_pair(pr))) return rc; HIP_TRY(hipEventRecord(pr->first, st)); }
            launch_wf_trace(d->ds, a, n_trace, fast, f.queue, f.slow_list, d->slow_cap, st, d->cfg);
            HIP_TRY(hipGetLastError());
            if (timed || keep) HIP_TRY(hipEventRecord(pr->second, st));
            launches++;
            std::swap(a.in, a.out);
            if (!look) n_grid *= 0.75;   // paths die at >= 40 % per bounce (Russian roulette 0.6)
        }
        // paths still alive at the depth cap cannot exist: logic(MAX_DEPTH-1) emits no bounce ray; a last logic pass resolves them
        if (n_upper > 0) {
            a.depth = MCPT_MAX_DEPTH;
            a.counts_in = &f.wf_counts[MCPT_MAX_DEPTH]; a.count_mul = 1u; a.counts = &f.wf_counts[MCPT_MAX_DEPTH + 1];
            launch_wf_logic(d->ds, a, n_upper, false, st, d->cfg);
            HIP_TRY(hipGetLastError());
        }
        launch_fold_samples(f.rad, d->pixels, f.hits, int(first), n_slots, spp, d_img, st);
        HIP_TRY(hipGetLastError());
    }
    if (timed && !keep) {
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t i = ev_first; i < d->ev_used; i++) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, d->ev_pool[i].first, d->ev_pool[i].second));
            ms_trace += ms;
        }
        d->ev_used = ev_first;
    }
    return MCPT_OK;
}

static int render_device_impl(mcpt_device* d, const mcpt_render_params* p, double* d_img, mcpt_stats* stats, hipStream_t st, int& slot_used);

int mcpt_render_device(mcpt_device* d, const mcpt_render_params* p, double* d_img, mcpt_stats* stats, void* stream)
{
    if (!d || !p || !d_img || p->spp <= 0) return fail(MCPT_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(d->ordinal));
    if (stats) std::memset(stats, 0, sizeof *stats);
    // a frame that fails half-way must not leave half-recorded event pairs behind: mcpt_device_collect_stats would trip over them
    const size_t ev_used0 = d->ev_used, frame_ev_used0 = d->frame_ev_used;
    int slot_used = -1;
    const int rc = render_device_impl(d, p, d_img, stats, static_cast<hipStream_t>(stream), slot_used);
    if (rc != MCPT_OK) {
        d->ev_used = ev_used0; d->frame_ev_used = frame_ev_used0;
        if (slot_used >= 0) d->slot[slot_used].keeping = false;      // its counters hold part of a frame: cleared by the next one
    }
    return rc;
}

static int render_device_impl(mcpt_device* d, const mcpt_render_params* p, double* d_img, mcpt_stats* stats, hipStream_t st, int& slot_used)
{
    const bool keep = (p->flags & MCPT_RENDER_KEEP_STATS) != 0 && !(p->flags & MCPT_RENDER_MEGAKERNEL);
    const bool timed = stats != nullptr && !keep;
    // frame slot: consecutive pipelined frames alternate; a slot's previous frame (possibly on another stream) must be over
    if ((p->flags & MCPT_RENDER_PIPELINE) && !d->pipelined) {
        HIP_TRY(hipDeviceSynchronize());
        d->pipelined = true; d->wf_auto_budget = 0;                  // the budget now has to hold two frames
        for (auto& q : d->slot) { if (q.wf_ws) (void)hipFree(q.wf_ws); q.wf_ws = nullptr; q.wf_ws_bytes = 0; }
    }
    const int si = (p->flags & MCPT_RENDER_PIPELINE) ? (d->next_slot ^= 1) : 0;
    slot_used = si;
    mcpt_device::FrameSlot& f = d->slot[si];
    if (f.used) HIP_TRY(hipStreamWaitEvent(st, f.done, 0));
    int rc = ensure_dirs(d, st);
    if (rc) return rc;
    rc = prepare_partition(d, p, st);
    if (rc) return rc;
    const int64_t npx = d->n_pixels;
    if (npx == 0) return MCPT_OK;
    if ((rc = grow(&f.hits, &f.hits_cap, npx))) return rc;
    if (!keep || !f.keeping) HIP_TRY(hipMemsetAsync(f.ctr, 0, sizeof(DCounters), st));    // kept statistics accumulate until they are collected
    f.keeping = keep;
    std::pair<hipEvent_t, hipEvent_t>* fe = nullptr;
    if (keep) {
        if (d->frame_ev_used == d->frame_ev.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            d->frame_ev.emplace_back(e0, e1);
        }
        fe = &d->frame_ev[d->frame_ev_used++];
        HIP_TRY(hipEventRecord(fe->first, st));
    } else HIP_TRY(hipEventRecord(d->ev[0], st));
    launch_primary_hits(d->ds, d->trace_mode == MCPT_TRACE_FAST, d->dirs, d->pixels, int(npx), f.hits, f.ctr, f.queue, f.slow_list, d->slow_cap, st, d->cfg);
    HIP_TRY(hipGetLastError());
    double ms_trace = 0;
    int launches = 0;
    if (p->flags & MCPT_RENDER_MEGAKERNEL) rc = render_megakernel(d, f, p, d_img, timed, st, ms_trace, launches);
    else rc = render_wavefront(d, f, p, d_img, timed, keep, st, ms_trace, launches);
    if (rc) return rc;
    if (keep) {
        HIP_TRY(hipEventRecord(fe->second, st));
        d->kept_samples += uint64_t(npx) * uint64_t(p->spp); d->kept_primary += uint64_t(npx); d->kept_launches += launches;
    } else HIP_TRY(hipEventRecord(d->ev[1], st));
    HIP_TRY(hipEventRecord(f.done, st));
    f.used = true;
    if (timed) {
        DCounters c{};
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(&c, f.ctr, sizeof c, hipMemcpyDeviceToHost));
        counters_to_stats(c, stats, d->knobs.print_diag != 0);
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, d->ev[0], d->ev[1]));
        stats->ms_total = ms; stats->ms_trace = ms_trace; stats->launches = launches;
        stats->samples = uint64_t(npx) * uint64_t(p->spp);      // camera samples covered (a primary miss is a finished sample)
        stats->rays_primary = uint64_t(npx);
    }
    return MCPT_OK;
}

// Statistics of every MCPT_RENDER_KEEP_STATS frame since the last call: waits for those frames, sums the device counters of both
// frame slots, the event pairs around every k_wf_trace launch (ms_trace) and around every frame (ms_total = sum of frame times;
// pipelined frames overlap, so this can exceed the wall time), then starts over.
int mcpt_device_collect_stats(mcpt_device* d, mcpt_stats* stats)
{
    if (!d || !stats) return fail(MCPT_ERR_ARG, "null argument");
    std::memset(stats, 0, sizeof *stats);
    HIP_TRY(hipSetDevice(d->ordinal));
    HIP_TRY(hipDeviceSynchronize());
    DCounters sum{};
    for (auto& f : d->slot) {
        DCounters c{};
        HIP_TRY(hipMemcpy(&c, f.ctr, sizeof c, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(f.ctr, 0, sizeof(DCounters)));
        unsigned long long* a = reinterpret_cast<unsigned long long*>(&sum);
        const unsigned long long* b = reinterpret_cast<const unsigned long long*>(&c);
        for (size_t i = 0; i < sizeof(DCounters) / sizeof(unsigned long long); i++) a[i] += b[i];
        sum.max_depth = std::max(sum.max_depth - c.max_depth, c.max_depth);      // a maximum, not a sum
    }
    counters_to_stats(sum, stats, d->knobs.print_diag != 0);
    // the bookkeeping starts over whatever the queries say: a pair that cannot be read is left out and reported
    hipError_t bad = hipSuccess;
    for (size_t i = 0; i < d->ev_used; i++) {
        float ms = 0;
        const hipError_t e = hipEventElapsedTime(&ms, d->ev_pool[i].first, d->ev_pool[i].second);
        if (e == hipSuccess) stats->ms_trace += ms; else bad = e;
    }
    for (size_t i = 0; i < d->frame_ev_used; i++) {
        float ms = 0;
        const hipError_t e = hipEventElapsedTime(&ms, d->frame_ev[i].first, d->frame_ev[i].second);
        if (e == hipSuccess) stats->ms_total += ms; else bad = e;
    }
    stats->launches = d->kept_launches; stats->samples = d->kept_samples; stats->rays_primary = d->kept_primary;
    d->ev_used = 0; d->frame_ev_used = 0; d->kept_launches = 0; d->kept_samples = 0; d->kept_primary = 0;
    for (auto& f : d->slot) f.keeping = false;
    if (bad != hipSuccess) { (void)hipGetLastError(); return fail(MCPT_ERR_HIP, std::string("an event pair of a kept frame could not be read: ") + hipGetErrorString(bad)); }
    return MCPT_OK;
}

int mcpt_render(mcpt_device* d, const mcpt_render_params* p, double* img, mcpt_stats* stats)
{
    if (!d || !p || !img) return fail(MCPT_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(d->ordinal));
    const size_t bytes = size_t(d->width) * d->height * 3 * sizeof(double);
    double* d_img = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_img), bytes));
    // The caller's frame is pageable host memory: blocking copies on either side of the frame, which itself is ordered on d->stream.
    hipError_t e = hipMemcpy(d_img, img, bytes, hipMemcpyHostToDevice);   // untouched pixels keep the caller's values
    int rc = e == hipSuccess ? mcpt_render_device(d, p, d_img, stats, d->stream) : fail(MCPT_ERR_HIP, hipGetErrorString(e));
    // also on failure: nothing of this frame may still be running when d_img goes
    e = hipStreamSynchronize(d->stream);
    if (rc == MCPT_OK && e == hipSuccess) e = hipMemcpy(img, d_img, bytes, hipMemcpyDeviceToHost);
    if (rc == MCPT_OK && e != hipSuccess) rc = fail(MCPT_ERR_HIP, hipGetErrorString(e));
    (void)hipFree(d_img);
    return rc;
}

int mcpt_sample_radiance(mcpt_device* d, uint64_t seed, const int32_t* pix, const int32_t* k, int64_t n, double* rgb)
{
    if (!d || !pix || !k || !rgb || n < 0) return fail(MCPT_ERR_ARG, "bad argument");
    if (n == 0) return MCPT_OK;
    for (int64_t i = 0; i < n; i++)
        if (pix[i] < 0 || pix[i] >= d->width * d->height) return fail(MCPT_ERR_ARG, "pixel index out of range");
    HIP_TRY(hipSetDevice(d->ordinal));
    int rc = ensure_dirs(d, d->stream);
    if (rc) return rc;
    int32_t *d_pix = nullptr, *d_k = nullptr;
    double* d_rgb = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_pix); (void)hipFree(d_k); (void)hipFree(d_rgb); };
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_pix), size_t(n) * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_k), size_t(n) * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_rgb), size_t(n) * 24);
    if (e == hipSuccess) e = hipMemcpy(d_pix, pix, size_t(n) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_k, k, size_t(n) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_sample_radiance(d->ds, seed, d->dirs, d_pix, d_k, n, d_rgb, d->aux_ctr, d->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    if (e == hipSuccess) e = hipMemcpy(rgb, d_rgb, size_t(n) * 24, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(MCPT_ERR_HIP, hipGetErrorString(e));
    return MCPT_OK;
}

// ------------------------------------------------------------------------------------------------ output
int mcpt_quantize_rgb8(const double* img, int64_t n, uint8_t* rgb8)
{
    if (!img || !rgb8 || n < 0) return fail(MCPT_ERR_ARG, "bad argument");
    for (int64_t i = 0; i < n; i++) {
        double v = img[i] * 255;                  // imshow, MTPC.cpp:26-28: (unsigned char)glm::clamp(v*255, 0.0, 255.0)
        v = std::max(v, 0.0);
        v = std::min(v, 255.0);
        rgb8[i] = static_cast<uint8_t>(v);
    }
    return MCPT_OK;
}

int64_t mcpt_png_encode(const uint8_t* rgb8, int32_t w, int32_t h, uint8_t* out, int64_t cap)
{
    if (!rgb8 || !out) return fail(MCPT_ERR_ARG, "null argument");
    const int64_t n = png_encode(rgb8, w, h, out, cap);
    if (n < 0) return fail(MCPT_ERR_ARG, "png: bad size or buffer too small");
    return n;
}

int mcpt_write_png(const char* file, const uint8_t* rgb8, int32_t w, int32_t h)
{
    if (!file || !rgb8 || w <= 0 || h <= 0) return fail(MCPT_ERR_ARG, "bad argument");
    const int64_t cap = 8 + 25 + 12 + 2 + int64_t(h) * (int64_t(w) * 3 + 6) + 4 + 12 + 16;
    std::vector<uint8_t> buf(static_cast<size_t>(cap));
    const int64_t n = png_encode(rgb8, w, h, buf.data(), cap);
    if (n < 0) return fail(MCPT_ERR_ARG, "png: width too large for one stored block per row");
    FILE* fp = std::fopen(file, "wb");
    if (!fp) return fail(MCPT_ERR_IO, std::string("cannot open ") + file);
    const bool ok = std::fwrite(buf.data(), 1, size_t(n), fp) == size_t(n);
    std::fclose(fp);                              // the reference never closes it (truncated veach-mis PNGs)
    return ok ? MCPT_OK : fail(MCPT_ERR_IO, std::string("short write to ") + file);
}

int64_t mcpt_png_encode_deflate(const uint8_t* rgb8, int32_t w, int32_t h, uint8_t* out, int64_t cap)
{
    if (!rgb8 || w <= 0 || h <= 0) { fail(MCPT_ERR_ARG, "bad argument"); return MCPT_ERR_ARG; }
    const int64_t n = png_encode_deflate(rgb8, w, h, out, cap);
    if (n < 0) { fail(MCPT_ERR_ARG, "png: buffer too small"); return MCPT_ERR_ARG; }
    return n;
}

int mcpt_write_png_deflate(const char* file, const uint8_t* rgb8, int32_t w, int32_t h)
{
    if (!file || !rgb8 || w <= 0 || h <= 0) return fail(MCPT_ERR_ARG, "bad argument");
    const int64_t need = png_encode_deflate(rgb8, w, h, nullptr, 0);
    std::vector<uint8_t> buf(static_cast<size_t>(need));
    const int64_t n = png_encode_deflate(rgb8, w, h, buf.data(), need);
    if (n != need) return fail(MCPT_ERR_ARG, "png: encoder size mismatch");
    FILE* fp = std::fopen(file, "wb");
    if (!fp) return fail(MCPT_ERR_IO, std::string("cannot open ") + file);
    const bool ok = std::fwrite(buf.data(), 1, size_t(n), fp) == size_t(n);
    return (std::fclose(fp) == 0 && ok) ? MCPT_OK : fail(MCPT_ERR_IO, std::string("short write to ") + file);
}

int mcpt_write_pfm(const char* file, const double* img, int32_t w, int32_t h)
{
    if (!file || !img || w <= 0 || h <= 0) return fail(MCPT_ERR_ARG, "bad argument");
    std::string err;
    const int rc = write_pfm(file, img, w, h, err);
    return rc ? fail(rc, err) : MCPT_OK;
}

// Identity of the frame a checkpoint belongs to: FNV-1a over everything the picture depends on besides spp / seed / parts
// (which the file header carries): geometry, normals, texture coordinates and material of every face in leaf order, material
// records and texels, lights, camera, resolution, Morton domain.  Version 2 of the tag (version 1 hashed three counts).
static uint64_t scene_tag(const Scene& s)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
    auto mixd = [&](double v) { mix(&v, sizeof v); };
    auto mixi = [&](int64_t v) { mix(&v, sizeof v); };
    mixi(2); mixi(int64_t(s.faces.size())); mixi(int64_t(s.materials.size())); mixi(int64_t(s.lights.size()));
    for (const FaceRec& f : s.faces) {
        for (int c = 0; c < 3; c++) { mixd(f.v[c].x); mixd(f.v[c].y); mixd(f.v[c].z); mixd(f.vn[c].x); mixd(f.vn[c].y); mixd(f.vn[c].z); mixd(f.vt[c][0]); mixd(f.vt[c][1]); }
        mixi(f.material); mixi(f.morton);
    }
    for (const MaterialRec& m : s.materials) {
        mixd(m.kd.x); mixd(m.kd.y); mixd(m.kd.z); mixd(m.ks.x); mixd(m.ks.y); mixd(m.ks.z); mixd(m.Ns); mixd(m.Ni);
        mixi(m.has_map); mixi(m.map_w); mixi(m.map_h);
        if (!m.bgr.empty()) mix(m.bgr.data(), m.bgr.size());
    }
    for (const LightRec& l : s.lights) { mixi(l.material); mixd(l.radiance.x); mixd(l.radiance.y); mixd(l.radiance.z); }
    for (const Vec3* v : {&s.eye, &s.look_at, &s.up}) { mixd(v->x); mixd(v->y); mixd(v->z); }
    mixd(s.fovy); mixi(s.width); mixi(s.height);
    for (int a = 0; a < 3; a++) { mixd(s.morton_lo[a]); mixd(s.morton_span[a]); }
    return h;
}

int mcpt_checkpoint_save(const char* file, const mcpt_scene* h, const double* img, int32_t spp, uint64_t seed, int32_t parts, const uint8_t* done)
{
    if (!file || !h || !img || !done || spp <= 0 || parts <= 0 || parts > 65536) return fail(MCPT_ERR_ARG, "bad argument");
    std::string err;
    const int rc = checkpoint_save(file, img, h->s.width, h->s.height, spp, seed, scene_tag(h->s), parts, done, err);
    return rc ? fail(rc, err) : MCPT_OK;
}

int mcpt_checkpoint_load(const char* file, const mcpt_scene* h, double* img, int32_t spp, uint64_t seed, int32_t parts, uint8_t* done)
{
    if (!file || !h || !img || !done || spp <= 0 || parts <= 0 || parts > 65536) return fail(MCPT_ERR_ARG, "bad argument");
    std::string err;
    const int rc = checkpoint_load(file, img, h->s.width, h->s.height, spp, seed, scene_tag(h->s), parts, done, err);
    return rc ? fail(rc, err) : MCPT_OK;
}

int mcpt_decode_jpeg(const char* file, int32_t* width, int32_t* height, uint8_t* bgr, int64_t cap)
{
    if (!file || !width || !height) return fail(MCPT_ERR_ARG, "null argument");
    int w = 0, h = 0;
    std::vector<uint8_t> px;
    std::string err;
    if (!decode_jpeg_file(file, w, h, px, err)) return fail(MCPT_ERR_IO, err);
    *width = w; *height = h;
    if (bgr) {
        if (cap < int64_t(px.size())) return fail(MCPT_ERR_ARG, "buffer too small");
        std::memcpy(bgr, px.data(), px.size());
    }
    return MCPT_OK;
}

// ------------------------------------------------------------------------------------------------ render_scene
// The options struct grew with the library version (100: seed .. output_prefix; 101: .. reserved; 102: .. devices) and carries no size
// of its own.  mcpt_render_scene_ex was the only entry point through version 102 and reads the struct as it stood then -- every field
// of it: a caller that sets load_flags, a checkpoint or num_devices through it gets what it asked for, not a silently different
// render -- so a caller compiled against a 100 / 101 header must hand over a zero-extended struct of that size.  Fields added after
// 102 are reached through mcpt_render_scene_opts only, which takes the caller's sizeof and reads exactly that many bytes.
static constexpr int64_t kOptionsBytesV102 = int64_t(offsetof(mcpt_render_scene_options, devices) + sizeof(const int32_t*));
int mcpt_render_scene_ex(const char* path, const char* filename, int32_t spp, const mcpt_render_scene_options* opt, mcpt_stats* stats)
{
    return mcpt_render_scene_opts(path, filename, spp, opt, opt ? kOptionsBytesV102 : 0, stats);
}

int mcpt_render_scene_opts(const char* path, const char* filename, int32_t spp, const mcpt_render_scene_options* opt, int64_t opt_bytes, mcpt_stats* stats)
{
    if (!path || !filename || spp <= 0 || opt_bytes < 0 || (opt_bytes > 0 && !opt)) return fail(MCPT_ERR_ARG, "bad argument");
    mcpt_render_scene_options o{};
    if (opt) std::memcpy(&o, opt, std::min<size_t>(size_t(opt_bytes), sizeof o));
    const bool talk = !o.quiet;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    mcpt_scene* sc = nullptr;
    int rc = mcpt_scene_load_ex(path, filename, o.load_flags, &sc);
    if (rc) return rc;
    if (o.width > 0 && o.height > 0) mcpt_scene_set_resolution(sc, o.width, o.height);
    const Scene& s = sc->s;
    if (talk) {
        std::printf("%s%s.obj\nnumber of materials = %zu\nnumber of vertices = %zu\nnumber of faces = %zu\n", path, filename,
                    s.materials.size(), s.v.size(), s.faces.size());
        std::printf("Total real = %d\nBuild BVH success\n", s.bi.Nr);
    }
    mcpt_device* dev = nullptr;
    mcpt_multi* multi = nullptr;
    const bool many = o.num_devices > 0 || o.num_devices == -1;
    if (many) rc = mcpt_multi_create(sc, o.num_devices > 0 ? o.devices : nullptr, o.num_devices > 0 ? o.num_devices : 0, MCPT_BUILD_HOST, o.gather, &multi);
    else rc = mcpt_device_create(sc, o.device, &dev);
    if (rc) { mcpt_scene_free(sc); return rc; }
    if (talk && many) std::printf("rendering on %d GPUs\n", mcpt_multi_num_devices(multi));
    if (many && o.checkpoint) {
        mcpt_multi_free(multi); mcpt_scene_free(sc);
        return fail(MCPT_ERR_ARG, "a checkpointed frame is rendered partition by partition on one GPU: leave num_devices at 0");
    }
    const auto t1 = clk::now();
    if (talk) std::printf("Phase 1(read scene + bvh build) time cost = %.3f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    std::vector<double> img(size_t(s.width) * s.height * 3, 0.0);
    mcpt_render_params rp{};
    rp.spp = spp; rp.seed = o.seed; rp.world = 1;
    mcpt_stats local{};
    if (!o.checkpoint) {
        rc = many ? mcpt_multi_render(multi, &rp, img.data(), &local) : mcpt_render(dev, &rp, img.data(), &local);
    } else {
        // the frame in `parts` tile partitions, saved after each; partitions a matching checkpoint already holds are skipped
        const int parts = o.checkpoint_parts > 0 ? o.checkpoint_parts : 8;
        std::vector<uint8_t> done(size_t(parts), 0);
        const int lrc = mcpt_checkpoint_load(o.checkpoint, sc, img.data(), spp, o.seed, parts, done.data());
        if (lrc != MCPT_OK) { std::fill(img.begin(), img.end(), 0.0); std::fill(done.begin(), done.end(), uint8_t(0)); }
        if (talk && lrc == MCPT_OK) {
            int have = 0;
            for (uint8_t v : done) have += v ? 1 : 0;
            std::printf("resuming from %s: %d of %d partitions done\n", o.checkpoint, have, parts);
        }
        rp.world = parts;
        for (int part = 0; part < parts && rc == MCPT_OK; part++) {
            if (done[size_t(part)]) continue;
            rp.rank = part;
            mcpt_stats one{};
            rc = mcpt_render(dev, &rp, img.data(), &one);
            if (rc != MCPT_OK) break;
            local.rays_primary += one.rays_primary; local.rays_shadow += one.rays_shadow; local.rays_bounce += one.rays_bounce;
            local.node_visits += one.node_visits; local.tri_tests += one.tri_tests; local.shade_calls += one.shade_calls;
            local.samples += one.samples; local.shadow_skipped += one.shadow_skipped; local.ms_trace += one.ms_trace;
            local.ms_total += one.ms_total; local.launches += one.launches;
            local.max_depth = std::max(local.max_depth, one.max_depth);
            done[size_t(part)] = 1;
            rc = mcpt_checkpoint_save(o.checkpoint, sc, img.data(), spp, o.seed, parts, done.data());
        }
    }
    const auto t2 = clk::now();
    if (rc == MCPT_OK) {
        if (talk) std::printf("Phase 2(ray tracing) = %.3f ms\n", std::chrono::duration<double, std::milli>(t2 - t1).count());
        std::vector<uint8_t> rgb(img.size());
        mcpt_quantize_rgb8(img.data(), int64_t(img.size()), rgb.data());
        const std::string prefix = o.output_prefix ? std::string(o.output_prefix) : std::string("../result/") + filename;
        const std::string stem = prefix + "-SPP" + std::to_string(spp);                 // imshow, MTPC.cpp:17-20
        rc = (o.output_flags & MCPT_OUT_PNG_DEFLATE) ? mcpt_write_png_deflate((stem + ".png").c_str(), rgb.data(), s.width, s.height)
                                                      : mcpt_write_png((stem + ".png").c_str(), rgb.data(), s.width, s.height);
        if (rc == MCPT_OK && (o.output_flags & MCPT_OUT_PFM)) rc = mcpt_write_pfm((stem + ".pfm").c_str(), img.data(), s.width, s.height);
    }
    if (stats) *stats = local;
    if (dev) mcpt_device_free(dev);
    if (multi) mcpt_multi_free(multi);
    mcpt_scene_free(sc);
    return rc;
}

int mcpt_render_scene(const char* path, const char* filename, int32_t spp)
{
    return mcpt_render_scene_ex(path, filename, spp, nullptr, nullptr);
}

}  // extern "C"
