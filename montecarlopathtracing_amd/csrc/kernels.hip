// Hand-written HIP kernels for gfx950 (MI355X): the reference's per-pixel integrator loop
// generateImg -> ray_intersect/bvh_intersect -> shade -> nextRay (MTPC/pathTracing.cpp), in fp64 with the
// reference's operation order.  Built with -ffp-contract=off: a fused multiply-add would change the last
// bit of the hit tests and with it the closest-hit triangle.
//
// No MFMA: this is per-lane tree walking and branching, not a contraction.  wave64 throughout.
#include <hip/hip_runtime.h>

#include "accel_build.hpp"
#include "dev_common.hpp"
#include "kernels.hpp"
#include "shade_common.hpp"
#include "shade_path.hpp"
#include "trace_persistent.hpp"
#include "trace_pool.hpp"

namespace mcpt {

// ------------------------------------------------------------------------------------------------ kernels
// mcpt_trace_closest with the reference-shaped walk: one lane per ray.
__global__ void __launch_bounds__(256) k_trace_closest_reference(DScene S, const double* __restrict__ rays, long long n,
                                                       int32_t* __restrict__ face, double* __restrict__ t_out,
                                                       double* __restrict__ p_out, double* __restrict__ pn_out, DCounters* ctr)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    LaneStats ls;
    if (gid < n) {
        Ray r;
        r.o = ld3(rays + gid * 6); r.d = ld3(rays + gid * 6 + 3);
        Hit h; Work w = {0, 0};
        const bool ok = trace_closest(S, r, h, w);
        ls.nodes = w.nodes; ls.tris = w.tris; ls.primary = 1;
        V3 pn = mk(0, 0, 0);
        if (ok) pn = hit_normal(S, h);
        if (face) face[gid] = ok ? S.tris[h.leaf].face : -1;
        if (t_out) t_out[gid] = ok ? h.t : 0.0;
        if (p_out) { p_out[gid * 3] = h.p.x; p_out[gid * 3 + 1] = h.p.y; p_out[gid * 3 + 2] = h.p.z; }
        if (pn_out) { pn_out[gid * 3] = pn.x; pn_out[gid * 3 + 1] = pn.y; pn_out[gid * 3 + 2] = pn.z; }
    }
    flush_stats(ctr, ls);
}

// ---- persistent fast walk over a plain ray array (mcpt_trace_closest) and over the primary rays
struct ArrayRaySource {
    static constexpr bool kWantsPoint = true;
    const double* rays; long long n;
    int32_t* leaf_out; double* t_out; double* p_out;     // leaf_out receives the LEAF index; k_finish_hits turns it into a face
    __device__ __forceinline__ long long total() const { return n; }
    __device__ __forceinline__ bool fetch(long long q, Ray& r) const
    {
        r.o = ld3(rays + q * 6); r.d = ld3(rays + q * 6 + 3);
#ifdef MCPT_DBG_INVALID        /* debugging builds: a pseudo-random quarter of the slots holds no ray */
        return ((((unsigned int)q * 2654435761u) >> 7) & 3u) != 0u;
#else
        return true;
#endif
    }
    __device__ __forceinline__ void store(long long q, bool ok, const Hit& h) const
    {
        leaf_out[q] = ok ? h.leaf : -1;
        t_out[q] = ok ? h.t : 0.0;
        p_out[q * 3] = h.p.x; p_out[q * 3 + 1] = h.p.y; p_out[q * 3 + 2] = h.p.z;
    }
};

struct PrimaryRaySource {
    static constexpr bool kWantsPoint = true;
    const double* dirs; const int32_t* pixels; int n_pixels; double eye[3]; PrimaryHit* hits;
    __device__ __forceinline__ long long total() const { return n_pixels; }
    __device__ __forceinline__ bool fetch(long long q, Ray& r) const
    {
        const int pix = pixels ? pixels[q] : (int)q;
        r.o = ld3(eye); r.d = ld3(dirs + (size_t)pix * 3);
        return true;
    }
    __device__ __forceinline__ void store(long long q, bool ok, const Hit& h) const
    {
        PrimaryHit ph;
        ph.leaf = ok ? h.leaf : -1; ph.pad = 0; ph.t = h.t; ph.p[0] = h.p.x; ph.p[1] = h.p.y; ph.p[2] = h.p.z;
        hits[q] = ph;
    }
};

template <class Src, int STACK, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_trace_persistent(DScene S, Src src, TraceQueue* queue, long long* slow_list, unsigned int slow_cap,
                                                                long long chunk, DCounters* ctr)
{
    __shared__ int lds_stack[STACK * 256];
    __shared__ double lds_rays[4 * MCPT_RAYBUF_BYTES / 8];
    LaneStats ls;
    Work w = {0, 0};
    trace_persistent(S, src, queue, slow_list, slow_cap, chunk, lds_stack + threadIdx.x, 256, lds_rays + (threadIdx.x >> 6) * (MCPT_RAYBUF_BYTES / 8), w, S.fast.stack_cap < STACK ? S.fast.stack_cap : STACK);
    ls.nodes = w.nodes; ls.tris = w.tris;
    { const unsigned long long tw = wave_sum(w.pre_wrong); if ((threadIdx.x & 63) == 0 && tw && ctr) atomicAdd(&ctr->pad[20], tw); }
    flush_stats(ctr, ls);
}

template <class Src, int NW, int KT, int SCAP>
__global__ void __launch_bounds__(NW * 64, 1) k_trace_pool(DScene S, Src src, TraceQueue* queue, long long* slow_list, unsigned int slow_cap,
                                                          long long chunk, DCounters* ctr)
{
    __shared__ PoolLds<NW, KT, SCAP> L;
    LaneStats ls;
    Work w = {0, 0};
    trace_pool<Src, NW, KT, SCAP>(S, src, queue, slow_list, slow_cap, chunk, L, w, reinterpret_cast<int*>(slow_list + slow_cap));
    ls.nodes = w.nodes; ls.tris = w.tris;
    { const unsigned long long tw = wave_sum(w.pre_wrong); if ((threadIdx.x & 63) == 0 && tw && ctr) atomicAdd(&ctr->pad[20], tw); }
    flush_stats(ctr, ls);
}

template <class Src>
__global__ void __launch_bounds__(256) k_trace_slow(DScene S, Src src, const TraceQueue* queue, const long long* slow_list, unsigned int slow_cap,
                                                    DCounters* ctr)
{
    __shared__ int lds_stack[MCPT_FAST_STACK * 256];
    LaneStats ls;
    Work w = {0, 0};
    trace_slow_list(S, src, queue, slow_list, slow_cap, w, lds_stack + threadIdx.x);
    ls.nodes = w.nodes; ls.tris = w.tris;
    flush_stats(ctr, ls);
}

// leaf index -> .obj face index, interpolated normal of the accepted hit
__global__ void k_finish_hits(DScene S, long long n, int32_t* __restrict__ face, const double* __restrict__ p, double* __restrict__ pn_out, DCounters* ctr)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    LaneStats ls;
    if (gid < n) {
        ls.primary = 1;
        const int leaf = face[gid];
        V3 pn = mk(0, 0, 0);
        if (leaf >= 0) {
            Hit h; h.leaf = leaf; h.t = 0; h.p = ld3(p + gid * 3);
            pn = hit_normal(S, h);
            face[gid] = S.tris[leaf].face;
        }
        if (pn_out) { pn_out[gid * 3] = pn.x; pn_out[gid * 3 + 1] = pn.y; pn_out[gid * 3 + 2] = pn.z; }
    }
    flush_stats(ctr, ls);
}

// Pixel positions: pos(i,0) = start - pdy*i, pos(i,j+1) = pos(i,j) + pdx -- a running sum along each row
// (pathTracing.cpp:297,326), so one thread walks one row.  Writes the normalised primary direction.
__global__ void k_primary_dirs(DCamera cam, double* __restrict__ dirs)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= cam.height) return;
    const V3 eye = ld3(cam.eye), pdx = ld3(cam.pdx);
    V3 pos = ld3(cam.start_point) - ld3(cam.pdy) * (double)row;
    double* out = dirs + (size_t)row * cam.width * 3;
    for (int j = 0; j < cam.width; j++) {
        const V3 d = normalized(pos - eye);
        out[j * 3] = d.x; out[j * 3 + 1] = d.y; out[j * 3 + 2] = d.z;
        pos = pos + pdx;
    }
}

// One lane per owned pixel: the primary ray is the same for every sample of a pixel (no jitter,
// pathTracing.cpp:306-308), so it is traced once.
__global__ void __launch_bounds__(256) k_primary_hits_reference(DScene S, const double* __restrict__ dirs, const int32_t* __restrict__ pixels,
                                                      int n_pixels, PrimaryHit* __restrict__ hits, DCounters* ctr)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    LaneStats ls;
    if (gid < n_pixels) {
        const int pix = pixels ? pixels[gid] : gid;
        Ray r;
        r.o = ld3(S.cam.eye); r.d = ld3(dirs + (size_t)pix * 3);
        Hit h; Work w = {0, 0};
        const bool ok = trace_closest(S, r, h, w);
        ls.nodes = w.nodes; ls.tris = w.tris; ls.primary = 1;
        PrimaryHit ph;
        ph.leaf = ok ? h.leaf : -1; ph.pad = 0; ph.t = h.t; ph.p[0] = h.p.x; ph.p[1] = h.p.y; ph.p[2] = h.p.z;
        hits[gid] = ph;
    }
    flush_stats(ctr, ls);
}

// One lane per camera sample.  Samples of one pixel are consecutive lanes, so a wave starts from one shared
// primary hit (coherent first vertex and shadow rays).  Radiance goes to rad[(slot*spp + k)*3].
__global__ void __launch_bounds__(256) k_shade_samples(DScene S, unsigned long long seed, const double* __restrict__ dirs,
                                                       const int32_t* __restrict__ pixels, const PrimaryHit* __restrict__ hits,
                                                       int first_slot, long long n_samples, int spp, double* __restrict__ rad, DCounters* ctr)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    LaneStats ls;
    if (gid < n_samples) {
        const int slot = first_slot + (int)(gid / spp);
        const int k = (int)(gid % spp);
        const int pix = pixels ? pixels[slot] : slot;
        const PrimaryHit ph = hits[slot];
        ls.samples = 1;
        double r[3] = {0, 0, 0};
        if (ph.leaf >= 0) {
            RngKey key; key.k0 = (uint32_t)seed; key.k1 = (uint32_t)(seed >> 32); key.pixel = (uint32_t)pix; key.sample = (uint32_t)k;
            Hit h; h.leaf = ph.leaf; h.t = ph.t; h.p = mk(ph.p[0], ph.p[1], ph.p[2]);
            shade_path(S, key, ld3(dirs + (size_t)pix * 3), h, r, ls);
        }
        rad[gid * 3] = r[0]; rad[gid * 3 + 1] = r[1]; rad[gid * 3 + 2] = r[2];
    }
    flush_stats(ctr, ls);
}

// mcpt_sample_radiance: arbitrary (pixel, k) pairs, primary ray traced per sample.
__global__ void __launch_bounds__(256) k_sample_radiance(DScene S, unsigned long long seed, const double* __restrict__ dirs,
                                                         const int32_t* __restrict__ pix, const int32_t* __restrict__ ks, long long n,
                                                         double* __restrict__ rgb, DCounters* ctr)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    LaneStats ls;
    if (gid < n) {
        Ray r; r.o = ld3(S.cam.eye); r.d = ld3(dirs + (size_t)pix[gid] * 3);
        Hit h; Work w = {0, 0};
        double out[3] = {0, 0, 0};
        ls.primary = 1; ls.samples = 1;
        if (trace_closest(S, r, h, w)) {
            RngKey key; key.k0 = (uint32_t)seed; key.k1 = (uint32_t)(seed >> 32); key.pixel = (uint32_t)pix[gid]; key.sample = (uint32_t)ks[gid];
            shade_path(S, key, r.d, h, out, ls);
        }
        ls.nodes += w.nodes; ls.tris += w.tris;
        rgb[gid * 3] = out[0]; rgb[gid * 3 + 1] = out[1]; rgb[gid * 3 + 2] = out[2];
    }
    flush_stats(ctr, ls);
}

// Per pixel: acc(float) += radiance/N for k = 0..N-1 in order (pathTracing.cpp:301,316-318 with D3), widened
// to double for image::img (sceneManagement.h:221).  One lane per (pixel, channel).
__global__ void k_fold_samples(const double* __restrict__ rad, const int32_t* __restrict__ pixels, const PrimaryHit* __restrict__ hits,
                               int first_slot, int n_slots, int spp, double* __restrict__ img)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)n_slots * 3) return;
    const int s = (int)(gid / 3), c = (int)(gid % 3);
    const double* src = rad + (size_t)s * spp * 3 + c;
    const int slot = first_slot + s;
    float acc = 0.0f;
    if (hits[slot].leaf >= 0)            // a primary miss adds nothing (pathTracing.cpp:311)
        for (int k = 0; k < spp; k++) acc = (float)((double)acc + src[(size_t)k * 3] / spp);
    const int pix = pixels ? pixels[slot] : slot;
    img[(size_t)pix * 3 + c] = (double)acc;
}

// End-of-frame exchange of the multi-GPU entry (multi_device.cpp): a rank's pixels leave its frame as one compact buffer
// (pack, on the rank's GPU) and are put at their frame positions on GPU 0 (unpack).  One lane per (pixel, channel).
__global__ void k_pack_pixels(const double* __restrict__ frame, const int32_t* __restrict__ pixels, long long n3, double* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) out[i] = frame[(size_t)pixels[i / 3] * 3 + (i % 3)];
}
__global__ void k_unpack_pixels(const double* __restrict__ in, const int32_t* __restrict__ pixels, long long n3, double* __restrict__ frame)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) frame[(size_t)pixels[i / 3] * 3 + (i % 3)] = in[i];
}

// ------------------------------------------------------------------------------------------------ launchers
static inline unsigned blocks_for(long long n, int block) { return (unsigned)((n + block - 1) / block); }

template <class Src>
static void launch_persistent(const DScene& S, const Src& src, long long total, TraceQueue* queue, long long* slow_list, unsigned int slow_cap,
                              DCounters* ctr, hipStream_t st, int grid, int grid_short, const LaunchCfg& cfg)
{
    const bool shallow = S.fast.stack_limit <= kFastShortStack;
    const int resident = shallow ? grid_short : grid;
    const long long blocks_needed = (total + 255) / 256;
    const int g = (int)(blocks_needed < resident ? blocks_needed : resident);
    (void)hipMemsetAsync(queue, 0, sizeof(TraceQueue), st);
    if (cfg.trace_pool && total < (1ll << 32)) {        // (the pool engine keeps a ray's slot number in 32 bits)
        const long long per_block = cfg.trace_block_rays * (MCPT_POOL_WAVES / 4);
        const long long nb = (total + per_block - 1) / per_block;
        const int gp = (int)(nb < cfg.cus ? nb : cfg.cus);
        long long c = total / ((long long)gp * MCPT_POOL_WAVES * 4);
        c = (c / 64) * 64; c = c < cfg.min_chunk ? cfg.min_chunk : (c > cfg.max_chunk ? cfg.max_chunk : c);
        hipLaunchKernelGGL((k_trace_pool<Src, MCPT_POOL_WAVES, MCPT_POOL_KT, MCPT_POOL_STACK>), dim3(gp), dim3(MCPT_POOL_WAVES * 64), 0, st, S, src, queue, slow_list, slow_cap, c, ctr);
    } else if (shallow) hipLaunchKernelGGL((k_trace_persistent<Src, kFastShortStack, 4>), dim3(g), dim3(256), 0, st, S, src, queue, slow_list, slow_cap, persistent_chunk(total, g), ctr);
    else hipLaunchKernelGGL((k_trace_persistent<Src, MCPT_FAST_STACK, 3>), dim3(g), dim3(256), 0, st, S, src, queue, slow_list, slow_cap, persistent_chunk(total, g), ctr);
    hipLaunchKernelGGL(k_trace_slow<Src>, dim3(256), dim3(256), 0, st, S, src, queue, slow_list, slow_cap, ctr);
}

bool pool_engine_available_closest()
{
    int a = 0, b = 0;
    const hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &a, reinterpret_cast<const void*>(k_trace_pool<ArrayRaySource, MCPT_POOL_WAVES, MCPT_POOL_KT, MCPT_POOL_STACK>), MCPT_POOL_WAVES * 64, 0);
    const hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &b, reinterpret_cast<const void*>(k_trace_pool<PrimaryRaySource, MCPT_POOL_WAVES, MCPT_POOL_KT, MCPT_POOL_STACK>), MCPT_POOL_WAVES * 64, 0);
    if (e1 != hipSuccess || e2 != hipSuccess) (void)hipGetLastError();
    return e1 == hipSuccess && e2 == hipSuccess && a >= 1 && b >= 1;
}

void init_launch_cfg_closest(LaunchCfg& cfg)
{
    cfg.array_grid = persistent_grid(reinterpret_cast<const void*>(k_trace_persistent<ArrayRaySource, MCPT_FAST_STACK, 3>), cfg.cus);
    cfg.primary_grid = persistent_grid(reinterpret_cast<const void*>(k_trace_persistent<PrimaryRaySource, MCPT_FAST_STACK, 3>), cfg.cus);
    cfg.array_grid_short = persistent_grid(reinterpret_cast<const void*>(k_trace_persistent<ArrayRaySource, kFastShortStack, 4>), cfg.cus);
    cfg.primary_grid_short = persistent_grid(reinterpret_cast<const void*>(k_trace_persistent<PrimaryRaySource, kFastShortStack, 4>), cfg.cus);
}

// d_face, d_t, d_p must be non-null device buffers (the C-ABI layer always allocates them); d_pn may be null
void launch_trace_closest(const DScene& S, bool fast, const double* d_rays, long long n, int32_t* d_face, double* d_t, double* d_p,
                          double* d_pn, DCounters* ctr, TraceQueue* queue, long long* slow_list, unsigned int slow_cap, hipStream_t st,
                          const LaunchCfg& cfg)
{
    if (n <= 0) return;
    if (!fast) {
        hipLaunchKernelGGL(k_trace_closest_reference, dim3(blocks_for(n, 256)), dim3(256), 0, st, S, d_rays, n, d_face, d_t, d_p, d_pn, ctr);
        return;
    }
    ArrayRaySource src; src.rays = d_rays; src.n = n; src.leaf_out = d_face; src.t_out = d_t; src.p_out = d_p;
    launch_persistent(S, src, n, queue, slow_list, slow_cap, ctr, st, cfg.array_grid, cfg.array_grid_short, cfg);
    hipLaunchKernelGGL(k_finish_hits, dim3(blocks_for(n, 256)), dim3(256), 0, st, S, n, d_face, d_p, d_pn, ctr);
}
void launch_pack_pixels(const double* d_frame, const int32_t* d_pixels, long long n_pixels, double* d_out, hipStream_t st)
{
    if (n_pixels <= 0) return;
    hipLaunchKernelGGL(k_pack_pixels, dim3(blocks_for(n_pixels * 3, 256)), dim3(256), 0, st, d_frame, d_pixels, n_pixels * 3, d_out);
}
void launch_unpack_pixels(const double* d_in, const int32_t* d_pixels, long long n_pixels, double* d_frame, hipStream_t st)
{
    if (n_pixels <= 0) return;
    hipLaunchKernelGGL(k_unpack_pixels, dim3(blocks_for(n_pixels * 3, 256)), dim3(256), 0, st, d_in, d_pixels, n_pixels * 3, d_frame);
}
void launch_primary_dirs(const DCamera& cam, double* d_dirs, hipStream_t st)
{
    hipLaunchKernelGGL(k_primary_dirs, dim3(blocks_for(cam.height, 64)), dim3(64), 0, st, cam, d_dirs);
}
void launch_primary_hits(const DScene& S, bool fast, const double* d_dirs, const int32_t* d_pixels, int n_pixels, PrimaryHit* d_hits,
                         DCounters* ctr, TraceQueue* queue, long long* slow_list, unsigned int slow_cap, hipStream_t st, const LaunchCfg& cfg)
{
    if (n_pixels <= 0) return;
    if (!fast) {
        hipLaunchKernelGGL(k_primary_hits_reference, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, st, S, d_dirs, d_pixels, n_pixels, d_hits, ctr);
        return;
    }
    PrimaryRaySource src; src.dirs = d_dirs; src.pixels = d_pixels; src.n_pixels = n_pixels; src.hits = d_hits;
    src.eye[0] = S.cam.eye[0]; src.eye[1] = S.cam.eye[1]; src.eye[2] = S.cam.eye[2];
    launch_persistent(S, src, n_pixels, queue, slow_list, slow_cap, ctr, st, cfg.primary_grid, cfg.primary_grid_short, cfg);
}
void launch_shade_samples(const DScene& S, unsigned long long seed, const double* d_dirs, const int32_t* d_pixels,
                          const PrimaryHit* d_hits, int first_slot, int n_slots, int spp, double* d_rad, DCounters* ctr, hipStream_t st)
{
    const long long n = (long long)n_slots * spp;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_shade_samples, dim3(blocks_for(n, 256)), dim3(256), 0, st, S, seed, d_dirs, d_pixels, d_hits, first_slot, n, spp, d_rad, ctr);
}
void launch_sample_radiance(const DScene& S, unsigned long long seed, const double* d_dirs, const int32_t* d_pix, const int32_t* d_k,
                            long long n, double* d_rgb, DCounters* ctr, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sample_radiance, dim3(blocks_for(n, 256)), dim3(256), 0, st, S, seed, d_dirs, d_pix, d_k, n, d_rgb, ctr);
}
void launch_fold_samples(const double* d_rad, const int32_t* d_pixels, const PrimaryHit* d_hits, int first_slot, int n_slots, int spp,
                         double* d_img, hipStream_t st)
{
    if (n_slots <= 0) return;
    hipLaunchKernelGGL(k_fold_samples, dim3(blocks_for((long long)n_slots * 3, 256)), dim3(256), 0, st, d_rad, d_pixels, d_hits, first_slot, n_slots, spp, d_img);
}

}  // namespace mcpt
