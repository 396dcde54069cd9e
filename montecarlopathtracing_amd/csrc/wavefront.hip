// Wavefront integrator, trace side: the closest-hit launches of an iteration (see wavefront.hpp; the logic and finishing kernels are in
// wavefront_logic.hip, which is compiled with other code-generation options: Makefile).  -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "accel_build.hpp"
#include "dev_common.hpp"
#include "shade_common.hpp"
#include "trace_persistent.hpp"
#include "trace_pool.hpp"
#include "vertex.hpp"
#include "wavefront.hpp"
#include "wf_ray_source.hpp"

namespace mcpt {

// ---------------------------------------------------------------------------------------------- layout helpers (ldc / stc: dev_common.hpp)
size_t wf_bytes_per_path(int nl)
{
    const size_t state = 4 + (6 + 3 * nl + (nl == 1 ? 3 : 6)) * 8 + nl * 4 + 4 + nl * 4 + 4 + 3 * 8;   // WfState (no w with one light)
    const size_t rays = nl * 3 * 8;                                                  // WfRays
    return 2 * state + rays;
}

bool wf_carve(void* base, size_t bytes, long long cap, int nl, WfState& A, WfState& B, WfRays& R)
{
    char* p = static_cast<char*>(base);
    char* end = p + bytes;
    auto take = [&](size_t n) -> void* { char* r = p; p += (n + 255) & ~size_t(255); return r; };
    auto state = [&](WfState& s) {
        s.id = static_cast<int32_t*>(take(cap * 4));
        s.T = static_cast<double*>(take(cap * 24)); s.L = static_cast<double*>(take(cap * 24));
        s.c = static_cast<double*>(take(size_t(cap) * 24 * nl)); s.expect = static_cast<int32_t*>(take(size_t(cap) * 4 * nl));
        s.w = nl == 1 ? nullptr : static_cast<double*>(take(cap * 24)); s.bdir = static_cast<double*>(take(cap * 24));
        s.btype = static_cast<int32_t*>(take(cap * 4));
        s.hit_mat = static_cast<int32_t*>(take(size_t(cap) * 4 * nl)); s.hit_leaf = static_cast<int32_t*>(take(cap * 4));
        s.p = static_cast<double*>(take(cap * 24));
    };
    state(A); state(B);
    R.d = static_cast<double*>(take(size_t(cap) * 24 * nl));
    return p <= end;
}

// ---------------------------------------------------------------------------------------------- trace kernels
#ifndef MCPT_KARG_SOURCE
#define MCPT_KARG_SOURCE 1          /* the trace kernels read WfArgs from the kernarg segment where they need it (wf_ray_source.hpp: WfRaySourceK); 0: held in registers */
#endif
// where the second kernel parameter (WfArgs a) of k_wf_trace / k_wf_trace_pool lies in the kernarg segment: explicit arguments are laid
// out in order at their natural alignment, DScene first
__device__ __forceinline__ WfArgsKernarg wf_kernarg_args()
{
    constexpr size_t off = (sizeof(DScene) + alignof(WfArgs) - 1) / alignof(WfArgs) * alignof(WfArgs);
    typedef const char __attribute__((address_space(4)))* kbytes;
    return (WfArgsKernarg)((kbytes)__builtin_amdgcn_kernarg_segment_ptr() + off);
}

// persistent fast walk
__device__ __forceinline__ long long wf_chunk(long long total, int min_chunk, int max_chunk)
{
    const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
    long long c = total / (waves * 4);
    c = (c / 64) * 64;
    return c < min_chunk ? min_chunk : (c > max_chunk ? max_chunk : c);
}

// Two shapes of the kernel: SHORT = a kFastShortStack-entry stack (27 KB of LDS per block, 128 VGPRs: 4 waves per SIMD) for hierarchies
// built to fit it, else MCPT_FAST_STACK entries at 3 waves per SIMD (accel_build.hpp).
template <int STACK, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_wf_trace(DScene S, WfArgs a, TraceQueue* queue, long long* slow_list, unsigned int slow_cap, int min_chunk, int max_chunk)
{
    const long long n_paths = a.counts->n_next;
    if (n_paths <= (long long)a.finish_below) return;                  // nothing left, or k_wf_finish has taken the paths
    const long long chunk = wf_chunk(n_paths * (a.nl + 1), min_chunk, max_chunk);
    __shared__ int lds_stack[STACK * 256];
    __shared__ double lds_rays[4 * MCPT_RAYBUF_BYTES / 8];
#if MCPT_KARG_SOURCE
    WfRaySourceK src; src.ap = wf_kernarg_args(); src.n_paths = n_paths; src.nl = a.nl;
#else
    WfRaySource src; src.a = a; src.n_paths = n_paths;
#endif
    LaneStats ls;
    Work w = {0, 0};
#ifdef MCPT_PRE_CHECK
    if (a.ctr) w.dbg = a.ctr->dbg;
#endif
    trace_persistent(S, src, queue, slow_list, slow_cap, chunk, lds_stack + threadIdx.x, 256, lds_rays + (threadIdx.x >> 6) * (MCPT_RAYBUF_BYTES / 8), w, S.fast.stack_cap < STACK ? S.fast.stack_cap : STACK);
    ls.nodes = w.nodes; ls.tris = w.tris;
    if (a.ctr) {        // the dominant kernel's own work, for its roofline
        const unsigned long long tn = wave_sum(w.nodes), tt = wave_sum(w.tris), tr = wave_sum(w.rays), te = wave_sum(w.exact);
        if ((threadIdx.x & 63) == 0 && tr) {
            atomicAdd(&a.ctr->trace_nodes, tn); atomicAdd(&a.ctr->trace_tris, tt); atomicAdd(&a.ctr->trace_rays, tr); atomicAdd(&a.ctr->trace_exact, te);
        }
        const unsigned long long tw = wave_sum(w.pre_wrong);
        if ((threadIdx.x & 63) == 0 && tw) atomicAdd(&a.ctr->pad[20], tw);
    }
#ifdef MCPT_TRACE_DIAG
    if ((threadIdx.x & 63) == 0 && a.ctr) for (int i = 0; i < 12; i++) atomicAdd(&a.ctr->pad[i], w.diag[i]);
#endif
    flush_stats(a.ctr, ls);
}

// The pool engine (trace_pool.hpp): one workgroup per CU, its rays resident in LDS.
template <int NW, int KT, int SCAP>
__global__ void __launch_bounds__(NW * 64, 1) k_wf_trace_pool(DScene S, WfArgs a, TraceQueue* queue, long long* slow_list, unsigned int slow_cap, int min_chunk, int max_chunk)
{
    const long long n_paths = a.counts->n_next;
    if (n_paths <= (long long)a.finish_below) return;
    const long long chunk = wf_chunk(n_paths * (a.nl + 1), min_chunk, max_chunk);
    __shared__ PoolLds<NW, KT, SCAP> L;
#if MCPT_KARG_SOURCE
    typedef WfRaySourceK Src;
    Src src; src.ap = wf_kernarg_args(); src.n_paths = n_paths; src.nl = a.nl;
#else
    typedef WfRaySource Src;
    Src src; src.a = a; src.n_paths = n_paths;
#endif
    LaneStats ls;
    Work w = {0, 0};
#ifdef MCPT_POOL_DEBUG
    if (a.ctr) w.dbg = a.ctr->dbg;
#endif
    trace_pool<Src, NW, KT, SCAP>(S, src, queue, slow_list, slow_cap, chunk, L, w, reinterpret_cast<int*>(slow_list + slow_cap));
    ls.nodes = w.nodes; ls.tris = w.tris;
    if (a.ctr) {
        const unsigned long long tn = wave_sum(w.nodes), tt = wave_sum(w.tris), tr = wave_sum(w.rays), te = wave_sum(w.exact);
        if ((threadIdx.x & 63) == 0 && (tn | tt | tr | te)) {
            atomicAdd(&a.ctr->trace_nodes, tn); atomicAdd(&a.ctr->trace_tris, tt); atomicAdd(&a.ctr->trace_rays, tr); atomicAdd(&a.ctr->trace_exact, te);
        }
        const unsigned long long tw = wave_sum(w.pre_wrong);
        if ((threadIdx.x & 63) == 0 && tw) atomicAdd(&a.ctr->pad[20], tw);
    }
    flush_stats(a.ctr, ls);
}

__global__ void __launch_bounds__(256) k_wf_trace_slow(DScene S, WfArgs a, const TraceQueue* queue, const long long* slow_list, unsigned int slow_cap)
{
    const long long n_paths = a.counts->n_next;
    if (n_paths <= (long long)a.finish_below || queue->slow_count == 0) return;
    if (a.ctr && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&a.ctr->pad[12], (unsigned long long)queue->slow_count);   // diagnostics
    __shared__ int lds_stack[MCPT_FAST_STACK * 256];
    WfRaySource src; src.a = a; src.n_paths = n_paths;
    LaneStats ls;
    Work w = {0, 0};
    trace_slow_list(S, src, queue, slow_list, slow_cap, w, lds_stack + threadIdx.x);
    ls.nodes = w.nodes; ls.tris = w.tris;
    flush_stats(a.ctr, ls);
}

// reference-shaped walk for every ray (MCPT_TRACE_REFERENCE): one thread per slot
__global__ void __launch_bounds__(256) k_wf_trace_reference(DScene S, WfArgs a)
{
    const long long n_paths = a.counts->n_next;
    WfRaySource src; src.a = a; src.n_paths = n_paths;
    const long long total = src.total();
    LaneStats ls;
    Work w = {0, 0};
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        Ray r;
        if (!src.fetch(q, r)) continue;
        Hit h;
        const bool ok = trace_closest(S, r, h, w);
        src.store(q, ok, h);
    }
    ls.nodes = w.nodes; ls.tris = w.tris;
    flush_stats(a.ctr, ls);
}

// ---------------------------------------------------------------------------------------------- launchers
static unsigned grid_for(long long n, int block, unsigned cap_blocks)
{
    long long b = (n + block - 1) / block;
    if (b < 1) b = 1;
    return (unsigned)(b > cap_blocks ? cap_blocks : b);
}

// Can this device hold a workgroup of the pool engine (1024 threads, 159 KB of LDS)?  Asked once per process; the closest-hit forms of
// the kernel (kernels.hip) have the same footprint.
bool pool_engine_available()
{
    static const int ok = [] {
        int per_cu = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &per_cu, reinterpret_cast<const void*>(k_wf_trace_pool<MCPT_POOL_WAVES, MCPT_POOL_KT, MCPT_POOL_STACK>), MCPT_POOL_WAVES * 64, 0);
        if (e != hipSuccess) (void)hipGetLastError();
        return (e == hipSuccess && per_cu >= 1 && pool_engine_available_closest()) ? 1 : 0;
    }();
    return ok != 0;
}

size_t pool_spill_bytes(int cus)
{
    return size_t(cus > 0 ? cus : 256) * size_t(MCPT_POOL_SPILL) * size_t(MCPT_POOL_KT * 64) * sizeof(int);
}

int persistent_grid(const void* kernel, int cus)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
    return cus * per_cu;
}

// the current device's numbers (mcpt_device_create calls this once per device, with that device current); forced_logic_grid,
// trace_block_rays and the chunk bounds come from the handle's knobs (knobs.hpp)
void init_launch_cfg(LaunchCfg& cfg, unsigned forced_logic_grid, long long trace_block_rays, int min_chunk, int max_chunk)
{
    int dev = 0;
    hipDeviceProp_t prop;
    cfg.cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cfg.cus = prop.multiProcessorCount;
    if (cfg.cus <= 0) cfg.cus = 256;
    init_launch_cfg_logic(cfg, forced_logic_grid);             // wavefront_logic.hip: resident grids of k_wf_logic and k_wf_finish
    cfg.trace_grid = persistent_grid(reinterpret_cast<const void*>(k_wf_trace<MCPT_FAST_STACK, 3>), cfg.cus);
    cfg.trace_grid_short = persistent_grid(reinterpret_cast<const void*>(k_wf_trace<kFastShortStack, 4>), cfg.cus);
    // a wave that starts pays one atomic on the queue head and a few on the counters: give every block >= 2048 rays
    cfg.trace_block_rays = trace_block_rays >= 256 ? trace_block_rays : 2048;
    // every claim is an atomic on one word (~88 per microsecond on this chip): below this many rays per claim the queue head,
    // not the walk, bounds a launch of a million rays
    cfg.min_chunk = min_chunk >= 64 ? min_chunk / 64 * 64 : 256;
    cfg.max_chunk = max_chunk >= 64 ? max_chunk / 64 * 64 : 2048;
    if (cfg.max_chunk < cfg.min_chunk) cfg.max_chunk = cfg.min_chunk;
    cfg.trace_pool = 0;             // (mcpt_device_create picks the engine from the scene: capi.cpp: trace_engine_for)
    init_launch_cfg_closest(cfg);
}

long long persistent_chunk(long long total, int grid_blocks)
{
    const long long waves = (long long)grid_blocks * 4;
    long long c = total / (waves * 4);
    c = (c / 64) * 64;
    if (c < 64) c = 64;
    if (c > 2048) c = 2048;
    return c;
}

void launch_wf_trace(const DScene& S, const WfArgs& a, long long n_upper, bool fast, TraceQueue* queue, long long* slow_list,
                     unsigned int slow_cap, hipStream_t st, const LaunchCfg& cfg)
{
    if (n_upper <= 0) return;
    const long long total = n_upper * (a.nl + 1);
    if (!fast) {
        hipLaunchKernelGGL(k_wf_trace_reference, dim3(grid_for(total, 256, 1u << 20)), dim3(256), 0, st, S, a);
        return;
    }
    const bool shallow = S.fast.stack_limit <= kFastShortStack;
    const int resident = shallow ? cfg.trace_grid_short : cfg.trace_grid;
    const long long blocks_needed = (total + cfg.trace_block_rays - 1) / cfg.trace_block_rays;
    const int g = (int)(blocks_needed < resident ? blocks_needed : resident);
    if (a.queue != queue) (void)hipMemsetAsync(queue, 0, sizeof(TraceQueue), st);      // (else the logic pass before this launch has cleared it)
    if (cfg.trace_pool && total < (1ll << 32)) {        // (the pool engine keeps a ray's slot number in 32 bits)
        const long long per_block = cfg.trace_block_rays * (MCPT_POOL_WAVES / 4);
        const long long nb = (total + per_block - 1) / per_block;
        const int gp = (int)(nb < cfg.cus ? nb : cfg.cus);
        hipLaunchKernelGGL((k_wf_trace_pool<MCPT_POOL_WAVES, MCPT_POOL_KT, MCPT_POOL_STACK>), dim3(gp), dim3(MCPT_POOL_WAVES * 64), 0, st, S, a, queue, slow_list, slow_cap, cfg.min_chunk, cfg.max_chunk);
    } else if (shallow) hipLaunchKernelGGL((k_wf_trace<kFastShortStack, 4>), dim3(g), dim3(256), 0, st, S, a, queue, slow_list, slow_cap, cfg.min_chunk, cfg.max_chunk);
    else hipLaunchKernelGGL((k_wf_trace<MCPT_FAST_STACK, 3>), dim3(g), dim3(256), 0, st, S, a, queue, slow_list, slow_cap, cfg.min_chunk, cfg.max_chunk);
    hipLaunchKernelGGL(k_wf_trace_slow, dim3(g < 512 ? g : 512), dim3(256), 0, st, S, a, queue, slow_list, slow_cap);   // (blocks without work leave at once)
}

}  // namespace mcpt
