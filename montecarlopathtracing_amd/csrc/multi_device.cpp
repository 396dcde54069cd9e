// generateImg on several GPUs of one node behind the C ABI (mcpt_multi_*): one host thread per GPU, scene resident on every GPU,
// tiles dealt by mcpt_render_params.rank/world, end-of-frame exchange of compact pixel buffers into the first GPU's HBM over
// xGMI -- hipMemcpyPeerAsync by default, RCCL send/recv on request.  The reference has nothing like it (one OpenMP process,
// MTPC/pathTracing.cpp:303); this is what lets render_scene(path, filename, N) of MTPC/MTPC.cpp:35 use the whole node without
// a Python launcher.  Built only on the library's public entry points plus two pack/unpack kernels.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>          // types only: the library itself is loaded with dlopen when MCPT_GATHER_RCCL is asked for

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcpt.h"
#include "kernels.hpp"

namespace mcpt { int set_error(int code, const std::string& msg); }
static int fail(int code, const std::string& msg) { return mcpt::set_error(code, msg); }

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err)
    {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) { err = "librccl lacks an expected symbol"; return false; }
        return true;
    }
};

struct Rank {
    int ordinal = 0;
    mcpt_device* dev = nullptr;
    hipStream_t stream = nullptr;
    double* d_frame = nullptr;          // this rank's full-size frame (only its own pixels are written); rank 0's is THE frame
    int32_t* d_pixels = nullptr;        // its pixel list, on its GPU
    double* d_compact = nullptr;        // [n][3] on its GPU
    int64_t n = 0;
    // on devices[0]:
    double* d_stage = nullptr;          // [n][3] where the compact buffer lands
    int32_t* d_pixels0 = nullptr;       // the same pixel list on devices[0]
    mcpt_stats stats{};
    int rc = MCPT_OK;
    std::string err;
    hipEvent_t ev_start = nullptr, ev_rendered = nullptr;   // on the rank's stream: before its render / after its render (timing report)
    float render_ms = 0;
};

// One long-lived host thread per GPU (ranks 1..n-1; rank 0 works on the caller's thread): a frame hands each of them one job and
// waits for all.  Threads started per frame would pay their creation and the runtime's per-thread set-up inside every frame.
struct Workers {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::function<void(int)> job;
    uint64_t generation = 0;
    int pending = 0;
    bool quit = false;
    std::vector<std::thread> threads;

    void start(int n_ranks)
    {
        for (int r = 1; r < n_ranks; r++)
            threads.emplace_back([this, r]() {
                uint64_t seen = 0;
                for (;;) {
                    std::function<void(int)> fn;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_job.wait(lk, [&] { return quit || generation != seen; });
                        if (quit) return;
                        seen = generation;
                        fn = job;
                    }
                    fn(r);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    // fn(r) for every rank; rank 0 on the calling thread
    void run(const std::function<void(int)>& fn)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = fn; pending = int(threads.size()); generation++;
        }
        cv_job.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_job.notify_all();
        for (std::thread& t : threads) t.join();
        threads.clear();
    }
};

}  // namespace

struct mcpt_multi {
    const mcpt_scene* scene = nullptr;
    int width = 0, height = 0;
    int gather = MCPT_GATHER_PEER;
    int part_key[2] = {-1, -1};         // tile shape the pixel lists were made for ({-1,-1}: none)
    std::vector<Rank> ranks;
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_gather = nullptr;   // on devices[0]'s stream: frame start, frame end, its own render done
    float gather_ms = 0;                // last frame: from rank 0's render being done to the last unpack (what the exchange adds)
    Workers workers;
};

#define HIP_OR_FAIL(expr)                                                                               \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(MCPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

static void free_lists(mcpt_multi* m)
{
    for (Rank& r : m->ranks) {
        (void)hipSetDevice(r.ordinal);
        if (r.d_pixels) (void)hipFree(r.d_pixels);
        if (r.d_compact) (void)hipFree(r.d_compact);
        (void)hipSetDevice(m->ranks[0].ordinal);
        if (r.d_stage) (void)hipFree(r.d_stage);
        if (r.d_pixels0) (void)hipFree(r.d_pixels0);
        r.d_pixels = r.d_pixels0 = nullptr; r.d_compact = r.d_stage = nullptr; r.n = 0;
    }
}

// pixel lists of every rank for this tile shape, on the rank's GPU and on devices[0]
static int prepare_lists(mcpt_multi* m, const mcpt_render_params* p)
{
    const int key[2] = {p->tile_w, p->tile_h};
    if (m->ranks[0].d_pixels && std::memcmp(key, m->part_key, sizeof key) == 0) return MCPT_OK;
    free_lists(m);
    m->part_key[0] = m->part_key[1] = -1;       // whatever fails below, no list of this handle counts as valid
    const int world = int(m->ranks.size());
    for (int r = 0; r < world; r++) {
        Rank& R = m->ranks[size_t(r)];
        mcpt_render_params q = *p;
        q.rank = r; q.world = world;
        const int64_t n = mcpt_owned_pixels(m->scene, &q, nullptr);
        if (n < 0) return int(n);
        std::vector<int32_t> pix(size_t(std::max<int64_t>(n, 1)));
        if (n > 0 && mcpt_owned_pixels(m->scene, &q, pix.data()) != n) return fail(MCPT_ERR_ARG, "pixel partition changed between two calls");
        R.n = n;
        const size_t lb = pix.size() * sizeof(int32_t), cb = pix.size() * 3 * sizeof(double);
        HIP_OR_FAIL(hipSetDevice(R.ordinal));
        HIP_OR_FAIL(hipMalloc(reinterpret_cast<void**>(&R.d_pixels), lb));
        HIP_OR_FAIL(hipMemcpy(R.d_pixels, pix.data(), lb, hipMemcpyHostToDevice));
        if (r > 0) {
            HIP_OR_FAIL(hipMalloc(reinterpret_cast<void**>(&R.d_compact), cb));
            HIP_OR_FAIL(hipSetDevice(m->ranks[0].ordinal));
            HIP_OR_FAIL(hipMalloc(reinterpret_cast<void**>(&R.d_stage), cb));
            HIP_OR_FAIL(hipMalloc(reinterpret_cast<void**>(&R.d_pixels0), lb));
            HIP_OR_FAIL(hipMemcpy(R.d_pixels0, pix.data(), lb, hipMemcpyHostToDevice));
        }
    }
    std::memcpy(m->part_key, key, sizeof key);
    return MCPT_OK;
}

extern "C" {

void mcpt_multi_free(mcpt_multi* m)
{
    if (!m) return;
    m->workers.stop();
    if (!m->ranks.empty()) free_lists(m);
    for (size_t i = 0; i < m->comms.size(); i++)
        if (m->comms[i] && m->rccl.CommDestroy) { (void)hipSetDevice(m->ranks[i].ordinal); (void)m->rccl.CommDestroy(m->comms[i]); }
    for (Rank& r : m->ranks) {
        (void)hipSetDevice(r.ordinal);
        if (r.d_frame) (void)hipFree(r.d_frame);
        if (r.ev_start) (void)hipEventDestroy(r.ev_start);
        if (r.ev_rendered) (void)hipEventDestroy(r.ev_rendered);
        if (r.stream) (void)hipStreamDestroy(r.stream);
        if (r.dev) mcpt_device_free(r.dev);
    }
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->ev_gather) (void)hipEventDestroy(m->ev_gather);
    if (m->rccl.lib) dlclose(m->rccl.lib);
    delete m;
}

int mcpt_multi_num_devices(const mcpt_multi* m) { return m ? int(m->ranks.size()) : 0; }

int mcpt_multi_create(const mcpt_scene* scene, const int32_t* devices, int32_t num_devices, int32_t build_mode, int32_t gather, mcpt_multi** out)
{
    if (!scene || !out) return fail(MCPT_ERR_ARG, "null argument");
    *out = nullptr;
    if (gather != MCPT_GATHER_PEER && gather != MCPT_GATHER_RCCL) return fail(MCPT_ERR_ARG, "unknown gather mode");
    const int visible = mcpt_device_count();
    if (visible <= 0) return fail(MCPT_ERR_NO_DEVICE, "no HIP device available (libmcpt has no CPU fallback)");
    std::vector<int32_t> ord;
    if (!devices || num_devices <= 0) {
        const int n = num_devices > 0 ? num_devices : visible;
        for (int i = 0; i < n; i++) ord.push_back(i);
    } else ord.assign(devices, devices + num_devices);
    for (int32_t o : ord) if (o < 0 || o >= visible) return fail(MCPT_ERR_NO_DEVICE, "device ordinal out of range");
    if (gather == MCPT_GATHER_RCCL) {
        std::vector<int32_t> u = ord;
        std::sort(u.begin(), u.end());
        if (std::adjacent_find(u.begin(), u.end()) != u.end()) return fail(MCPT_ERR_ARG, "MCPT_GATHER_RCCL needs distinct device ordinals");
    }
    mcpt_scene_info info;
    int rc = mcpt_scene_get_info(scene, &info);
    if (rc) return rc;
    std::unique_ptr<mcpt_multi, void (*)(mcpt_multi*)> m(new mcpt_multi, mcpt_multi_free);
    m->scene = scene; m->width = info.width; m->height = info.height; m->gather = gather;
    m->ranks.resize(ord.size());
    for (size_t i = 0; i < ord.size(); i++) m->ranks[i].ordinal = ord[i];
    // one thread per GPU: upload + (device) build of the scene, the rank's frame and stream
    const size_t frame_bytes = size_t(info.width) * info.height * 3 * sizeof(double);
    auto setup = [&](Rank& R) {
        R.rc = mcpt_device_create_ex(scene, R.ordinal, build_mode, &R.dev);
        if (R.rc) { R.err = mcpt_last_error(); return; }
        hipError_t e = hipSetDevice(R.ordinal);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&R.d_frame), frame_bytes);
        if (e == hipSuccess) e = hipMemset(R.d_frame, 0, frame_bytes);
        if (e == hipSuccess) e = hipEventCreate(&R.ev_start);
        if (e == hipSuccess) e = hipEventCreate(&R.ev_rendered);
        if (e != hipSuccess) { R.rc = MCPT_ERR_HIP; R.err = std::string("multi-device setup: ") + hipGetErrorString(e); }
    };
    {
        std::vector<std::thread> pool;
        for (size_t i = 1; i < m->ranks.size(); i++) pool.emplace_back(setup, std::ref(m->ranks[i]));
        setup(m->ranks[0]);
        for (std::thread& t : pool) t.join();
    }
    for (Rank& R : m->ranks) if (R.rc) return fail(R.rc, R.err);
    // peer access towards devices[0] where the hardware offers it (copies fall back to staging otherwise)
    for (size_t i = 1; i < m->ranks.size(); i++) {
        const int a = m->ranks[i].ordinal, b = m->ranks[0].ordinal;
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
            (void)hipSetDevice(a);
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
    }
    HIP_OR_FAIL(hipSetDevice(m->ranks[0].ordinal));
    HIP_OR_FAIL(hipEventCreate(&m->ev0));
    HIP_OR_FAIL(hipEventCreate(&m->ev1));
    HIP_OR_FAIL(hipEventCreate(&m->ev_gather));
    m->workers.start(int(m->ranks.size()));
    if (gather == MCPT_GATHER_RCCL) {
        std::string err;
        if (!m->rccl.load(err)) return fail(MCPT_ERR_IO, err);
        m->comms.assign(ord.size(), nullptr);
        std::vector<int> devlist(ord.begin(), ord.end());
        const ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), int(devlist.size()), devlist.data());
        if (r != ncclSuccess) return fail(MCPT_ERR_HIP, std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(r));
    }
    *out = m.release();
    return MCPT_OK;
}

int mcpt_multi_render_device(mcpt_multi* m, const mcpt_render_params* p, double** d_img, mcpt_stats* stats)
{
    if (!m || !p || !d_img || p->spp <= 0) return fail(MCPT_ERR_ARG, "bad argument");
    *d_img = nullptr;
    if (stats) std::memset(stats, 0, sizeof *stats);
    int rc = prepare_lists(m, p);
    if (rc) return rc;
    const int world = int(m->ranks.size());
    Rank& R0 = m->ranks[0];
    HIP_OR_FAIL(hipSetDevice(R0.ordinal));
    HIP_OR_FAIL(hipEventRecord(m->ev0, R0.stream));
    const bool rccl = m->gather == MCPT_GATHER_RCCL && world > 1;
    const bool keep = (p->flags & MCPT_RENDER_KEEP_STATS) != 0 && !(p->flags & MCPT_RENDER_MEGAKERNEL);
    // one thread per GPU: render the rank's tiles, pack them, (peer mode) send them to devices[0]
    auto work = [&](int r) {
        Rank& R = m->ranks[size_t(r)];
        mcpt_render_params q = *p;
        q.rank = r; q.world = world;
        R.rc = MCPT_OK; R.err.clear();
        hipError_t e = hipSetDevice(R.ordinal);
        if (e == hipSuccess) e = hipEventRecord(R.ev_start, R.stream);
        if (e != hipSuccess) { R.rc = MCPT_ERR_HIP; R.err = std::string("multi-device render: ") + hipGetErrorString(e); return; }
        // MCPT_RENDER_KEEP_STATS: the ranks' statistics stay on their devices (no read-back, no event queries inside the frame) until
        // mcpt_multi_collect_stats; one frame at a time all the same (MCPT_RENDER_PIPELINE is not passed on)
        q.flags &= ~MCPT_RENDER_PIPELINE;
        R.rc = mcpt_render_device(R.dev, &q, R.d_frame, keep ? nullptr : &R.stats, R.stream);
        if (R.rc) { R.err = mcpt_last_error(); return; }
        e = hipSetDevice(R.ordinal);
        if (e == hipSuccess) e = hipEventRecord(R.ev_rendered, R.stream);
        if (e == hipSuccess && r == 0) e = hipEventRecord(m->ev_gather, R.stream);
        if (e == hipSuccess && r > 0 && R.n > 0) {
            mcpt::launch_pack_pixels(R.d_frame, R.d_pixels, R.n, R.d_compact, R.stream);
            e = hipGetLastError();
            if (e == hipSuccess && !rccl)
                e = hipMemcpyPeerAsync(R.d_stage, R0.ordinal, R.d_compact, R.ordinal, size_t(R.n) * 3 * sizeof(double), R.stream);
        }
        if (e == hipSuccess && !rccl) e = hipStreamSynchronize(R.stream);
        if (e != hipSuccess) { R.rc = MCPT_ERR_HIP; R.err = std::string("multi-device render: ") + hipGetErrorString(e); }
    };
    m->workers.run(work);
    for (Rank& R : m->ranks) if (R.rc) return fail(R.rc, R.err);
    if (rccl) {
        // every rank's send and rank 0's receives as ONE group: each send is ordered after the rank's render + pack on its stream,
        // the receives after rank 0's render on its stream
        ncclResult_t nr = m->rccl.GroupStart();
        for (int r = 1; r < world && nr == ncclSuccess; r++) {
            Rank& R = m->ranks[size_t(r)];
            if (R.n <= 0) continue;
            nr = m->rccl.Send(R.d_compact, size_t(R.n) * 3, ncclDouble, 0, m->comms[size_t(r)], R.stream);
            if (nr == ncclSuccess) nr = m->rccl.Recv(R.d_stage, size_t(R.n) * 3, ncclDouble, r, m->comms[0], R0.stream);
        }
        const ncclResult_t ge = m->rccl.GroupEnd();
        if (nr == ncclSuccess) nr = ge;
        if (nr != ncclSuccess) return fail(MCPT_ERR_HIP, std::string("RCCL gather: ") + m->rccl.GetErrorString(nr));
        for (int r = 1; r < world; r++) { HIP_OR_FAIL(hipSetDevice(m->ranks[size_t(r)].ordinal)); HIP_OR_FAIL(hipStreamSynchronize(m->ranks[size_t(r)].stream)); }
    }
    HIP_OR_FAIL(hipSetDevice(R0.ordinal));
    for (int r = 1; r < world; r++) {
        Rank& R = m->ranks[size_t(r)];
        mcpt::launch_unpack_pixels(R.d_stage, R.d_pixels0, R.n, R0.d_frame, R0.stream);
        HIP_OR_FAIL(hipGetLastError());
    }
    HIP_OR_FAIL(hipEventRecord(m->ev1, R0.stream));
    HIP_OR_FAIL(hipStreamSynchronize(R0.stream));
    if (stats && !keep) {
        for (const Rank& R : m->ranks) {
            const mcpt_stats& s = R.stats;
            stats->rays_primary += s.rays_primary; stats->rays_shadow += s.rays_shadow; stats->rays_bounce += s.rays_bounce;
            stats->node_visits += s.node_visits; stats->tri_tests += s.tri_tests; stats->shade_calls += s.shade_calls;
            stats->samples += s.samples; stats->shadow_skipped += s.shadow_skipped;
            stats->dom_rays += s.dom_rays; stats->dom_node_visits += s.dom_node_visits; stats->dom_tri_tests += s.dom_tri_tests;
            stats->launches += s.launches;
            stats->ms_trace = std::max(stats->ms_trace, s.ms_trace);
            stats->max_depth = std::max(stats->max_depth, s.max_depth);
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, m->ev0, m->ev1);      // rank 0's stream from before its render to after the last unpack
        stats->ms_total = ms;
    }
    m->gather_ms = 0;
    (void)hipEventElapsedTime(&m->gather_ms, m->ev_gather, m->ev1);
    for (Rank& R : m->ranks) {
        R.render_ms = 0;
        (void)hipSetDevice(R.ordinal);
        if (hipEventSynchronize(R.ev_rendered) == hipSuccess) (void)hipEventElapsedTime(&R.render_ms, R.ev_start, R.ev_rendered);
    }
    (void)hipGetLastError();
    (void)hipSetDevice(R0.ordinal);
    *d_img = R0.d_frame;
    return MCPT_OK;
}

// what the last frame's parts took (HIP events on each rank's own stream): render_ms[num_devices] (may be NULL), *gather_ms (may
// be NULL) = from devices[0]'s own render being done to the last rank's pixels being in place; *comm_ranks (may be NULL) = ranks the
// RCCL communicator reports (0 with MCPT_GATHER_PEER)
int mcpt_multi_last_timing(const mcpt_multi* m, double* render_ms, double* gather_ms, int32_t* comm_ranks)
{
    if (!m) return fail(MCPT_ERR_ARG, "null handle");
    if (render_ms) for (size_t i = 0; i < m->ranks.size(); i++) render_ms[i] = m->ranks[i].render_ms;
    if (gather_ms) *gather_ms = m->gather_ms;
    if (comm_ranks) {
        int n = 0;
        if (!m->comms.empty() && m->comms[0] && m->rccl.CommCount) (void)m->rccl.CommCount(m->comms[0], &n);
        *comm_ranks = n;
    }
    return MCPT_OK;
}

// statistics of every MCPT_RENDER_KEEP_STATS frame since the last call, summed over the GPUs (ms_trace, ms_total: the slowest GPU's sums)
int mcpt_multi_collect_stats(mcpt_multi* m, mcpt_stats* stats)
{
    if (!m || !stats) return fail(MCPT_ERR_ARG, "null argument");
    std::memset(stats, 0, sizeof *stats);
    for (Rank& R : m->ranks) {
        mcpt_stats s{};
        const int rc = mcpt_device_collect_stats(R.dev, &s);
        if (rc) return rc;
        stats->rays_primary += s.rays_primary; stats->rays_shadow += s.rays_shadow; stats->rays_bounce += s.rays_bounce;
        stats->node_visits += s.node_visits; stats->tri_tests += s.tri_tests; stats->shade_calls += s.shade_calls;
        stats->samples += s.samples; stats->shadow_skipped += s.shadow_skipped;
        stats->dom_rays += s.dom_rays; stats->dom_node_visits += s.dom_node_visits; stats->dom_tri_tests += s.dom_tri_tests;
        stats->launches += s.launches;
        stats->ms_trace = std::max(stats->ms_trace, s.ms_trace);
        stats->ms_total = std::max(stats->ms_total, s.ms_total);
        stats->max_depth = std::max(stats->max_depth, s.max_depth);
    }
    return MCPT_OK;
}

int mcpt_multi_render(mcpt_multi* m, const mcpt_render_params* p, double* img, mcpt_stats* stats)
{
    if (!img) return fail(MCPT_ERR_ARG, "null image");
    double* d = nullptr;
    const int rc = mcpt_multi_render_device(m, p, &d, stats);
    if (rc) return rc;
    HIP_OR_FAIL(hipSetDevice(m->ranks[0].ordinal));
    HIP_OR_FAIL(hipMemcpy(img, d, size_t(m->width) * m->height * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

}  // extern "C"
