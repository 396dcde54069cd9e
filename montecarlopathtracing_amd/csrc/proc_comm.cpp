// One process per GPU without torch in the process (mcpt_comm_*): the end-of-frame gather of a rank's pixels into rank 0's frame, a
// barrier and a small all-reduce, over an RCCL communicator that spans the PROCESSES of a launch (python -m torch.distributed.run starts
// them and hands out RANK / WORLD_SIZE; the ranks exchange RCCL's unique id themselves: montecarlopathtracing_amd/procs.py).  Why not
// torch.distributed for this: importing torch puts the wheel's HIP runtime under libmcpt.so's kernels -- here every rank
// runs the runtime the library was compiled against and /opt/rocm's own librccl.  The reference has nothing like it (one OpenMP process,
// MTPC/pathTracing.cpp:303); the in-process form of the same exchange is multi_device.cpp.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>          // types only: librccl is loaded with dlopen

#include <dlfcn.h>

#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mcpt.h"
#include "kernels.hpp"

namespace mcpt { int set_error(int code, const std::string& msg); }
static int fail(int code, const std::string& msg) { return mcpt::set_error(code, msg); }

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err)
    {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !AllReduce || !GetErrorString) {
            err = "librccl lacks an expected symbol";
            return false;
        }
        return true;
    }
};

}  // namespace

struct mcpt_comm {
    Rccl rccl;
    ncclComm_t comm = nullptr;
    int ordinal = 0, rank = 0, world = 1;
    hipStream_t stream = nullptr;               // the exchange's own stream
    hipEvent_t ev = nullptr;
    // pixel lists of the partition the buffers were made for
    int key[2] = {-1, -1};
    int64_t n_own = 0;
    int32_t* d_pixels_own = nullptr;            // this rank's pixels
    double* d_compact = nullptr;                // [n_own][3]
    std::vector<int64_t> n_of;                  // rank 0: pixels of every rank
    std::vector<int32_t*> d_pixels_of;          // rank 0: their lists, on this GPU
    std::vector<double*> d_stage_of;            // rank 0: where their buffers land
    double* d_red = nullptr;                    // all-reduce scratch (64 doubles)
};

#define HIP_OR_FAIL(expr)                                                                               \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(MCPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NCCL_OR_FAIL(c, expr)                                                                           \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) return fail(MCPT_ERR_HIP, std::string(#expr) + ": " + (c)->rccl.GetErrorString(r_)); \
    } while (0)

static void free_lists(mcpt_comm* c)
{
    if (c->d_pixels_own) (void)hipFree(c->d_pixels_own);
    if (c->d_compact) (void)hipFree(c->d_compact);
    for (int32_t* p : c->d_pixels_of) if (p) (void)hipFree(p);
    for (double* p : c->d_stage_of) if (p) (void)hipFree(p);
    c->d_pixels_own = nullptr; c->d_compact = nullptr; c->d_pixels_of.clear(); c->d_stage_of.clear(); c->n_of.clear();
    c->key[0] = c->key[1] = -1; c->n_own = 0;
}

static int prepare_lists(mcpt_comm* c, const mcpt_scene* scene, const mcpt_render_params* p)
{
    const int key[2] = {p->tile_w, p->tile_h};
    if (c->d_pixels_own && std::memcmp(key, c->key, sizeof key) == 0) return MCPT_OK;
    free_lists(c);
    auto list_of = [&](int r, std::vector<int32_t>& pix) -> int64_t {
        mcpt_render_params q = *p;
        q.rank = r; q.world = c->world;
        const int64_t n = mcpt_owned_pixels(scene, &q, nullptr);
        if (n < 0) return n;
        pix.assign(size_t(n > 0 ? n : 1), 0);
        if (n > 0 && mcpt_owned_pixels(scene, &q, pix.data()) != n) return fail(MCPT_ERR_ARG, "pixel partition changed between two calls");
        return n;
    };
    std::vector<int32_t> pix;
    int64_t n = list_of(c->rank, pix);
    if (n < 0) return int(n);
    c->n_own = n;
    HIP_OR_FAIL(hipMalloc(reinterpret_cast<void**>(&c->d_pixels_own), pix.size() * sizeof(int32_t)));
    HIP_OR_FAIL(hipMemcpy(c->d_pixels_own, pix.data(), pix.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_OR_FAIL(hipMalloc(reinterpret_cast<void**>(&c->d_compact), pix.size() * 3 * sizeof(double)));
    if (c->rank == 0) {
        c->n_of.assign(size_t(c->world), 0); c->d_pixels_of.assign(size_t(c->world), nullptr); c->d_stage_of.assign(size_t(c->world), nullptr);
        for (int r = 1; r < c->world; r++) {
            n = list_of(r, pix);
            if (n < 0) return int(n);
            c->n_of[size_t(r)] = n;
            HIP_OR_FAIL(hipMalloc(reinterpret_cast<void**>(&c->d_pixels_of[size_t(r)]), pix.size() * sizeof(int32_t)));
            HIP_OR_FAIL(hipMemcpy(c->d_pixels_of[size_t(r)], pix.data(), pix.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_OR_FAIL(hipMalloc(reinterpret_cast<void**>(&c->d_stage_of[size_t(r)]), pix.size() * 3 * sizeof(double)));
        }
    }
    std::memcpy(c->key, key, sizeof key);
    return MCPT_OK;
}

extern "C" {

int mcpt_comm_unique_id(uint8_t* id, int64_t cap)
{
    if (!id || cap < int64_t(sizeof(ncclUniqueId))) return fail(MCPT_ERR_ARG, "the id buffer needs 128 bytes");
    Rccl r;
    std::string err;
    if (!r.load(err)) return fail(MCPT_ERR_IO, err);
    ncclUniqueId u;
    const ncclResult_t rc = r.GetUniqueId(&u);
    if (rc != ncclSuccess) return fail(MCPT_ERR_HIP, std::string("ncclGetUniqueId: ") + r.GetErrorString(rc));
    std::memcpy(id, &u, sizeof u);
    return int(sizeof u);           // (the library stays loaded: the id refers to its bootstrap state)
}

void mcpt_comm_free(mcpt_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->ordinal);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_lists(c);
    if (c->d_red) (void)hipFree(c->d_red);
    if (c->comm && c->rccl.CommDestroy) (void)c->rccl.CommDestroy(c->comm);
    if (c->ev) (void)hipEventDestroy(c->ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int mcpt_comm_create(int32_t ordinal, int32_t rank, int32_t world, const uint8_t* id, int64_t id_bytes, mcpt_comm** out)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world || id_bytes != int64_t(sizeof(ncclUniqueId))) return fail(MCPT_ERR_ARG, "bad argument");
    *out = nullptr;
    const int visible = mcpt_device_count();
    if (visible <= 0) return fail(MCPT_ERR_NO_DEVICE, "no HIP device available (libmcpt has no CPU fallback)");
    if (ordinal < 0 || ordinal >= visible) return fail(MCPT_ERR_NO_DEVICE, "device ordinal out of range");
    std::unique_ptr<mcpt_comm, void (*)(mcpt_comm*)> c(new mcpt_comm, mcpt_comm_free);
    c->ordinal = ordinal; c->rank = rank; c->world = world;
    std::string err;
    if (!c->rccl.load(err)) return fail(MCPT_ERR_IO, err);
    HIP_OR_FAIL(hipSetDevice(ordinal));
    HIP_OR_FAIL(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_OR_FAIL(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
    HIP_OR_FAIL(hipMalloc(reinterpret_cast<void**>(&c->d_red), 64 * sizeof(double)));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    NCCL_OR_FAIL(c, c->rccl.CommInitRank(&c->comm, world, u, rank));
    *out = c.release();
    return MCPT_OK;
}

int mcpt_comm_size(const mcpt_comm* c)
{
    if (!c) return 0;
    int n = 0;
    if (c->rccl.CommCount && c->comm && c->rccl.CommCount(c->comm, &n) == ncclSuccess) return n;
    return c->world;
}

// Every rank: its own pixels of d_frame (frame layout, H*W*3 doubles on its GPU) travel as one compact buffer to rank 0, which puts them
// at their positions in ITS d_frame.  The exchange waits for what `stream` (the stream the frame was rendered on; NULL: the default
// stream) holds when the call is made, and the call returns when rank 0's frame is complete (ranks > 0: when their buffer is sent).
int mcpt_comm_gather_frame(mcpt_comm* c, const mcpt_scene* scene, const mcpt_render_params* p, double* d_frame, void* stream)
{
    if (!c || !scene || !p || !d_frame) return fail(MCPT_ERR_ARG, "null argument");
    HIP_OR_FAIL(hipSetDevice(c->ordinal));
    if (c->world == 1) return MCPT_OK;
    int rc = prepare_lists(c, scene, p);
    if (rc) return rc;
    HIP_OR_FAIL(hipEventRecord(c->ev, static_cast<hipStream_t>(stream)));
    HIP_OR_FAIL(hipStreamWaitEvent(c->stream, c->ev, 0));
    if (c->rank > 0) {
        if (c->n_own > 0) {
            mcpt::launch_pack_pixels(d_frame, c->d_pixels_own, c->n_own, c->d_compact, c->stream);
            HIP_OR_FAIL(hipGetLastError());
        }
        NCCL_OR_FAIL(c, c->rccl.GroupStart());
        if (c->n_own > 0) NCCL_OR_FAIL(c, c->rccl.Send(c->d_compact, size_t(c->n_own) * 3, ncclDouble, 0, c->comm, c->stream));
        NCCL_OR_FAIL(c, c->rccl.GroupEnd());
    } else {
        NCCL_OR_FAIL(c, c->rccl.GroupStart());
        for (int r = 1; r < c->world; r++)
            if (c->n_of[size_t(r)] > 0) NCCL_OR_FAIL(c, c->rccl.Recv(c->d_stage_of[size_t(r)], size_t(c->n_of[size_t(r)]) * 3, ncclDouble, r, c->comm, c->stream));
        NCCL_OR_FAIL(c, c->rccl.GroupEnd());
        for (int r = 1; r < c->world; r++)
            if (c->n_of[size_t(r)] > 0) {
                mcpt::launch_unpack_pixels(c->d_stage_of[size_t(r)], c->d_pixels_of[size_t(r)], c->n_of[size_t(r)], d_frame, c->stream);
                HIP_OR_FAIL(hipGetLastError());
            }
    }
    HIP_OR_FAIL(hipStreamSynchronize(c->stream));
    return MCPT_OK;
}

// v[n] (n <= 64, host) reduced over the ranks in place: op 0 = sum, 1 = max.  Doubles as the barrier of the launch (n = 0 is allowed).
int mcpt_comm_allreduce(mcpt_comm* c, double* v, int32_t n, int32_t op)
{
    if (!c || n < 0 || n > 64 || (n > 0 && !v) || (op != 0 && op != 1)) return fail(MCPT_ERR_ARG, "bad argument");
    HIP_OR_FAIL(hipSetDevice(c->ordinal));
    double buf[64] = {0};
    const int m = n > 0 ? n : 1;
    if (n > 0) std::memcpy(buf, v, size_t(n) * sizeof(double));
    HIP_OR_FAIL(hipMemcpy(c->d_red, buf, size_t(m) * sizeof(double), hipMemcpyHostToDevice));          // blocking: buf is pageable
    NCCL_OR_FAIL(c, c->rccl.AllReduce(c->d_red, c->d_red, size_t(m), ncclDouble, op == 0 ? ncclSum : ncclMax, c->comm, c->stream));
    HIP_OR_FAIL(hipStreamSynchronize(c->stream));
    HIP_OR_FAIL(hipMemcpy(buf, c->d_red, size_t(m) * sizeof(double), hipMemcpyDeviceToHost));
    if (n > 0) std::memcpy(v, buf, size_t(n) * sizeof(double));
    return MCPT_OK;
}

}  // extern "C"
