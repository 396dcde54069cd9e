// Wavefront integrator, path side: the logic kernel (resolve + shade + compaction), the finishing pass in its two forms and the small
// kernels around them (see wavefront.hpp).  -ffp-contract=off; compiled with -mllvm -disable-machine-licm (Makefile: these kernels use
// dozens of fp64 constants of the device math library; hoisted out of the block loop they cost ~40 registers, parked in scratch memory).
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

// The path state is a stream: every word is read once and written once per pass.  The logic kernel's accesses to it are marked
// non-temporal, so that they do not push the scene's triangles and shading records -- which the shade rounds gather at random -- out of
// L1 / L2 (round 4: -1 % of the frame on four scenes; MCPT_LOGIC_NT = 0 for A/B runs).
#ifndef MCPT_LOGIC_NT
#define MCPT_LOGIC_NT 1
#endif
#if MCPT_LOGIC_NT
template <class T> __device__ __forceinline__ T sld(const T* p) { return __builtin_nontemporal_load(p); }
template <class T> __device__ __forceinline__ void sst(T* p, T v) { __builtin_nontemporal_store(v, p); }
#else
template <class T> __device__ __forceinline__ T sld(const T* p) { return *p; }
template <class T> __device__ __forceinline__ void sst(T* p, T v) { *p = v; }
#endif
__device__ __forceinline__ V3 sldc(const double* __restrict__ a, long long cap, long long i) { return mk(sld(a + i), sld(a + cap + i), sld(a + 2 * cap + i)); }
__device__ __forceinline__ void sstc(double* __restrict__ a, long long cap, long long i, V3 v) { sst(a + i, v.x); sst(a + cap + i, v.y); sst(a + 2 * cap + i, v.z); }

#ifndef MCPT_FINISH_WAVES
#define MCPT_FINISH_WAVES 2  /* blocks of the finishing kernel per CU the compiler plans for: 2 = 256 registers per lane (76 bytes of them in scratch memory), 1 = 512 */
#endif
#ifndef MCPT_LOGIC_WAVES
#define MCPT_LOGIC_WAVES 4   /* waves per SIMD the logic kernel is compiled for: 128 VGPRs, 20 / 40 spilled registers.  Round 1 (ms per frame): 2: 111.5,
                               3: 108.0, 4: 107.2, 5: 113.5, 6: 121.2; with the 4-wave trace engine: 3: 101.6, 4: 100.4 (cornell-box), equal within
                               0.3 % on veach-mis, the interior and the 10 M-triangle scene */
#endif
#ifndef MCPT_LOGIC_WAVES_FIRST
#define MCPT_LOGIC_WAVES_FIRST MCPT_LOGIC_WAVES     /* ... its first pass (no resolve: fewer values alive) */
#endif

// ---------------------------------------------------------------------------------------------- logic kernel
// Shade vertex `depth` of sample `id` at path position j (pathTracing.cpp:147-241; the pieces are in vertex.hpp).
// What is known goes out at once and the bounce is sampled before the lights (every uniform has its own counter: the order of
// evaluation is free): L, p and the sample id are stored before anything is computed, the incoming direction dies with
// bounce_sample -- the later passes' values that are alive at the same time, and with them the registers the compiler had to
// park in scratch memory (39 at 4 waves per SIMD), are what this order is about.
template <bool FIRST>
__device__ __forceinline__ void wf_shade_vertex(const DScene& S, const WfArgs& a, long long j, int id, int leaf, const V3& p, const V3& dir, const V3& T, const V3& L,
                                                int mat_first, int pix_first, const V3& pn_first, const V3& kd_first, LaneStats& ls)
{
    const long long cap = a.cap;
    const int nl = a.nl;
    const bool folded = nl == 1;
    const uint32_t depth = (uint32_t)a.depth;
    sst(a.out.id + j, id);
    if (!FIRST) { sstc(a.out.L, cap, j, L); sstc(a.out.p, cap, j, p); }       // first pass: L = 0; p is the pixel's primary hit (a.hits)
    const DMaterial* m = S.materials + (FIRST ? mat_first : S.tris[leaf].material);
    V3 pn = pn_first, kd = kd_first;
    if (!FIRST) vertex_surface(S, leaf, p, m, pn, kd);

    RngKey key;
    key.k0 = (uint32_t)a.seed; key.k1 = (uint32_t)(a.seed >> 32);
    if (FIRST) key.pixel = (uint32_t)pix_first;
    else { const int slot = a.first_slot + id / a.spp; key.pixel = (uint32_t)(a.pixels ? a.pixels[slot] : slot); }
    key.sample = (uint32_t)(id % a.spp);

    {
        V3 nd = mk(0, 0, 0), wgt = mk(1, 1, 1);
        const int btype = bounce_sample(key, depth, nl, m, dir, pn, kd, nd, wgt);
        if (btype >= 0) { sstc(a.out.bdir, cap, j, nd); ls.bounce++; }
        sst(a.out.btype + j, btype);
        if (folded) sstc(a.out.T, cap, j, mk(T.x * wgt.x * MCPT_INV_P_RR, T.y * wgt.y * MCPT_INV_P_RR, T.z * wgt.z * MCPT_INV_P_RR));
        else { sstc(a.out.w, cap, j, wgt); if (!FIRST) sstc(a.out.T, cap, j, T); }
    }

    int sample_mat = -1;
    for (int l = 0; l < nl; l++) {
        V3 direction, c;
        const int expect = light_sample(S, key, depth, l, p, pn, kd, sample_mat, direction, c);
        if (expect != -2) {
            sstc(a.out.c + (long long)l * 3 * cap, cap, j, folded ? mk(T.x * c.x, T.y * c.y, T.z * c.z) : c);
            sstc(a.rays.d + (long long)l * 3 * cap, cap, j, direction);       // origin p + direction * 0.01: WfRaySource
            ls.shadow++;
        } else ls.skipped++;
        sst(a.out.expect + ((long long)l * cap + j), expect);
    }
}

#ifndef MCPT_LOGIC_PER
#define MCPT_LOGIC_PER 2            /* positions a thread of the later passes resolves per round (their words are requested together) */
#endif

// One thread per path position of the previous iteration.  FIRST: positions enumerate (hit slot, k).
//
// Later passes (FIRST == false).  Of the positions a pass resolves about half go on to a next vertex (Russian roulette at 0.6, rays that
// leave the scene, emitters), and shading is the longer half of the kernel: shaded where they were resolved, the surviving vertices ran
// in waves that were half empty (25 of 64 lanes per vector instruction over the whole kernel, round 4's counters).  A block therefore
// RESOLVES in rounds of 256 x MCPT_LOGIC_PER positions -- visibility of the shadow rays into the radiance so far, does the bounce ray
// lead to a surface that is not an emitter -- and parks the survivors in a ring in LDS: position, leaf, radiance (32 bytes).  It SHADES
// in rounds of 256 vertices taken off the ring, every wave full, as soon as 256 are there (and what is left at the end): the words the
// next vertex is made of (sample id, throughput, the bounce ray it was reached by) are read from the position then -- for the
// survivors only (the bytes moved stay what they were: a survivor's words share their 128-byte lines with the dead positions').  One
// barrier per resolve round (the
// block-wide prefix), one per shade round (ring and output base visible); a block's output positions are one atomic per shade round.
// Ring slots are written after the barrier of a resolve round, which every wave reaches only after its reads of the shade round
// before: no slot is overwritten while it is read.
template <bool FIRST>
__global__ void __launch_bounds__(256, FIRST ? MCPT_LOGIC_WAVES_FIRST : MCPT_LOGIC_WAVES) k_wf_logic(DScene S, WfArgs a)
{
    const long long n_prev = (long long)a.counts_in->n_next * a.count_mul;
    if (a.queue && blockIdx.x == 0 && threadIdx.x == 0) { a.queue->head = 0ull; a.queue->slow_count = 0u; a.queue->redo_all = 0u; }   // for trace(depth)
    if (!FIRST && n_prev <= (long long)a.finish_below) return;         // those paths went to k_wf_finish
    const long long cap = a.cap;
    const int nl = a.nl;
    // One light (the usual scene): T * c and T * w / P_RR are formed when the vertex is shaded instead of when it is resolved --
    // the same products, one pass earlier -- so the bounce weight never goes through memory.
    const bool folded = nl == 1;
    const uint32_t depth = (uint32_t)a.depth;           // depth of the vertex shaded in this pass
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    LaneStats ls;
#ifdef MCPT_TRACE_DIAG
    unsigned long long dg[4] = {0, 0, 0, 0};
    unsigned long long tl = __builtin_amdgcn_s_memtime();
#define MCPT_LSTAMP(k) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); dg[k] += tn - tl; tl = tn; }
#else
#define MCPT_LSTAMP(k)
#endif
    const long long n_round = (n_prev + 255) / 256 * 256;
    if constexpr (FIRST) {
        // ---- first pass: no resolve and no compaction -- a pixel's samples live or die together, so k_primary_surface has numbered the
        // shaded pixels and sample k of pixel number n sits at n * spp + k (no ballot, no atomic, no barrier)
        if (blockIdx.x == 0 && threadIdx.x == 0) a.counts->n_next = a.counts_in->pad[2] * (unsigned int)a.spp;   // shaded pixels x samples
        for (long long base = (long long)blockIdx.x * 256; base < n_round; base += (long long)gridDim.x * 256) {
            const long long i = base + threadIdx.x;
            if (i >= n_prev) continue;
            const PrimarySurface* ps = a.surf + i / a.spp;          // the same record for all samples of a pixel
            const int k = (int)(i % a.spp);
            // Path positions in exact slot order trace measurably slower on a rank's share of a frame, so the pixels are shuffled
            // within windows of 2^MCPT_SHUFFLE_LOG2 (an odd multiplier modulo a power of two is a bijection) -- close to the order
            // the block-wise compaction used to leave.  One eighth of the frame, ms per frame by window: none (slot order) 18.8,
            // 2^7 17.3, 2^10 16.5, 2^12 18.0, 2^14 18.7, 2^16 18.8; the whole frame is within 0.5 % for all of them.
            unsigned int an = a.alive_base[(i / a.spp) >> 6] + (unsigned int)ps->alive_index;
            const unsigned int n_alive = a.counts_in->pad[2];
#ifndef MCPT_SHUFFLE_LOG2
#define MCPT_SHUFFLE_LOG2 10
#endif
            constexpr unsigned int kWin = (1u << MCPT_SHUFFLE_LOG2) - 1u;
            if ((an | kWin) < n_alive) an = (an & ~kWin) | ((an & kWin) * 40503u & kWin);      // (not in the last, partial window)
            const long long j = (long long)an * a.spp + k;
            const int id = (ps->slot - a.first_slot) * a.spp + k;
            const int leaf = ps->leaf, mat_first = ps->material, pix_first = ps->pixel;
            const V3 p = ld3(ps->p), dir = ld3(ps->dir), pn_first = ld3(ps->pn), kd_first = ld3(ps->kd);
            ls.samples = 1;
            ls.shades++;
            if (depth > ls.depth) ls.depth = depth;
            const DMaterial* m = S.materials + mat_first;
            if (m->light >= 0) {                                             // emitter: pathTracing.cpp:141-144
                const V3 rad = ld3(S.lights[m->light].radiance);
                a.rad[(size_t)id * 3] = rad.x; a.rad[(size_t)id * 3 + 1] = rad.y; a.rad[(size_t)id * 3 + 2] = rad.z;
                MCPT_LSTAMP(0)
                continue;
            }
            MCPT_LSTAMP(0)
            wf_shade_vertex<true>(S, a, j, id, leaf, p, dir, mk(1, 1, 1), mk(0, 0, 0), mat_first, pix_first, pn_first, kd_first, ls);
            MCPT_LSTAMP(2)
        }
    } else {
        constexpr int kPer = MCPT_LOGIC_PER;
        constexpr unsigned int kRing = 512u * kPer;          // holds the 255 vertices that may wait and a round's 256 x kPer
        static_assert(kPer == 1 || kPer == 2, "ring size: a power of two");
        // (two sets of wave totals, used in turn: the values of one resolve round are still being read by its slower waves while the
        // faster ones write the next round's; the output base in turn as well: two shade rounds may follow each other directly)
        __shared__ unsigned int wave_tot[2][kPer][4];
        __shared__ unsigned int block_base[2];
        __shared__ int ring_pos[kRing], ring_leaf[kRing];
        __shared__ double ring_L[3][kRing];
        int turn = 0, bturn = 0;
        unsigned int head = 0, count = 0;                   // the ring: the same values in every thread of the block
        long long base = (long long)blockIdx.x * 256;
        for (;;) {
            while (count < 256u && base < n_round) {
                // ---- resolve vertex depth-1 of 256 x kPer positions (pathTracing.cpp:213-231, 244-261)
                // Every word is requested before any is looked at (what a dead path or an unused shadow slot holds is stale but
                // harmless): one memory latency per round instead of a chain of three.
                long long pos[kPer];
                int id[kPer], bt[kPer], hl[kPer];
                V3 L[kPer], T[kPer], wgt[kPer], L_dir[kPer];
                bool alive[kPer];
#pragma unroll
                for (int u = 0; u < kPer; u++) {
                    pos[u] = base + threadIdx.x;
                    base += (long long)gridDim.x * 256;
                    const long long i = pos[u] < n_prev ? pos[u] : 0;       // (past the end: position 0's words, not used)
                    id[u] = sld(a.in.id + i); bt[u] = sld(a.in.btype + i); hl[u] = sld(a.in.hit_leaf + i);
                    // folded (one light): in.T already is the throughput after the bounce and in.c is T * c (see the stores of
                    // wf_shade_vertex) -- the resolve needs T only where a specular chain ends on an emitter
                    T[u] = mk(1, 1, 1); L[u] = mk(0, 0, 0); wgt[u] = mk(1, 1, 1); L_dir[u] = mk(0, 0, 0);
                    if (!folded && depth > 1) T[u] = sldc(a.in.T, cap, i);
                    if (depth > 1) L[u] = sldc(a.in.L, cap, i);
                    if (!folded) wgt[u] = sldc(a.in.w, cap, i);
                    for (int l = 0; l < nl; l++) {
                        const int expect = sld(a.in.expect + ((long long)l * cap + i));
                        const int hm = sld(a.in.hit_mat + ((long long)l * cap + i));
                        const V3 c = sldc(a.in.c + (long long)l * 3 * cap, cap, i);
                        if (expect == -2) continue;
                        const bool vis = hm == expect;
                        L_dir[u].x += vis ? c.x : c.x * 0.0;
                        L_dir[u].y += vis ? c.y : c.y * 0.0;
                        L_dir[u].z += vis ? c.z : c.z * 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < kPer; u++) {
                    alive[u] = false;
                    if (pos[u] < n_prev) {
                        const bool have_vertex = bt[u] >= 0 && hl[u] >= 0;
                        if (folded) L[u] = L[u] + L_dir[u];
                        else {
                            L[u] = L[u] + mk(T[u].x * L_dir[u].x, T[u].y * L_dir[u].y, T[u].z * L_dir[u].z);
                            if (have_vertex) T[u] = mk(T[u].x * wgt[u].x * MCPT_INV_P_RR, T[u].y * wgt[u].y * MCPT_INV_P_RR, T[u].z * wgt[u].z * MCPT_INV_P_RR);
                        }
                        if (have_vertex) {
                            ls.shades++;
                            if (depth > ls.depth) ls.depth = depth;
                            // (is the surface an emitter: the trace kernel said so with the leaf, MCPT_HIT_EMITTER)
                            if (hl[u] & MCPT_HIT_EMITTER) {                              // emitter: pathTracing.cpp:141-144
                                if ((bt[u] & 7) != RT_DIFFUSE) {
                                    const DMaterial* m = S.materials + S.tris[hl[u] & MCPT_HIT_LEAF_MASK].material;
                                    if (folded) T[u] = sldc(a.in.T, cap, pos[u]);
                                    const V3 rad = ld3(S.lights[m->light].radiance);
                                    L[u] = L[u] + mk(T[u].x * rad.x, T[u].y * rad.y, T[u].z * rad.z);
                                }
                            } else alive[u] = true;
                        }
                        if (!alive[u]) { sst(a.rad + (size_t)id[u] * 3, L[u].x); sst(a.rad + (size_t)id[u] * 3 + 1, L[u].y); sst(a.rad + (size_t)id[u] * 3 + 2, L[u].z); }
                    }
                }
                MCPT_LSTAMP(0)
                // ---- the survivors go on the ring: wave ballots + prefix over the block's four waves, sub-round by sub-round
                unsigned int before[kPer];
                turn ^= 1;
#pragma unroll
                for (int u = 0; u < kPer; u++) {
                    const unsigned long long bal = __ballot(alive[u]);
                    before[u] = __popcll(bal & ((1ull << lane) - 1ull));
                    if (lane == 0) wave_tot[turn][u][wv] = (unsigned int)__popcll(bal);
                }
                __syncthreads();
                unsigned int at = head + count;
#pragma unroll
                for (int u = 0; u < kPer; u++) {
                    const unsigned int t0 = wave_tot[turn][u][0], t1 = wave_tot[turn][u][1], t2 = wave_tot[turn][u][2], t3 = wave_tot[turn][u][3];
                    if (alive[u]) {
                        const unsigned int s = (at + before[u] + (wv > 0 ? t0 : 0u) + (wv > 1 ? t1 : 0u) + (wv > 2 ? t2 : 0u)) & (kRing - 1u);
                        ring_pos[s] = (int)pos[u]; ring_leaf[s] = hl[u];        // (alive: no flag in it)
                        ring_L[0][s] = L[u].x; ring_L[1][s] = L[u].y; ring_L[2][s] = L[u].z;
                    }
                    at += t0 + t1 + t2 + t3;
                }
                count = at - head;
                MCPT_LSTAMP(1)
            }
            if (count == 0u) break;                          // every position resolved, every vertex shaded
            // ---- shade a round of vertices off the ring: positions block_base .. + m of the pass's output
            const unsigned int m_round = count < 256u ? count : 256u;
            // (one atomic per 256 paths on one word: with the word sharded 16 ways this kernel's first pass takes 6.0 instead of 6.5 ms --
            // the counter is not its floor)
            bturn ^= 1;
            if (threadIdx.x == 0) block_base[bturn] = atomicAdd(&a.counts->n_next, m_round);
            __syncthreads();
            MCPT_LSTAMP(1)
            if (threadIdx.x < m_round) {
                const unsigned int s = (head + threadIdx.x) & (kRing - 1u);
                const long long i = ring_pos[s];
                const int leaf = ring_leaf[s];
                const V3 L = mk(ring_L[0][s], ring_L[1][s], ring_L[2][s]);
                const long long j = (long long)block_base[bturn] + threadIdx.x;
                // the sample, its throughput and the bounce ray that reached this vertex, from the position
                const int id = sld(a.in.id + i);
                const int bt = sld(a.in.btype + i);
                V3 T = mk(1, 1, 1);
                if (depth > 1 || folded) T = sldc(a.in.T, cap, i);
                if (!folded) { const V3 wgt = sldc(a.in.w, cap, i); T = mk(T.x * wgt.x * MCPT_INV_P_RR, T.y * wgt.y * MCPT_INV_P_RR, T.z * wgt.z * MCPT_INV_P_RR); }
                const V3 bd = sldc(a.in.bdir, cap, i);
                // the vertex the bounce ray left from: the pixel's primary hit after the first pass, in.p afterwards
                V3 pv;
                if (depth == 1) { const PrimaryHit* ph = a.hits + (a.first_slot + id / a.spp); pv = mk(ph->p[0], ph->p[1], ph->p[2]); }
                else pv = sldc(a.in.p, cap, i);
                // the hit point of the bounce ray, as the reference's test formed it when the trace kernel accepted the triangle
                // (sceneManagement.cpp:318-320: t = ((v1 - o) . n) / (n . d), p = o + d t; same operands, same operations)
                const V3 ro = (bt & MCPT_BT_NO_OFFSET) ? pv : pv + bd * 0.01;
                const DTri* tr = S.tris + leaf;
                const V3 v1 = ld3(tr->v1), n = ld3(tr->n);
                const double t = dot(v1 - ro, n) / dot(n, bd);
                const V3 p = ro + bd * t;
                wf_shade_vertex<false>(S, a, j, id, leaf, p, neg(bd), T, L, 0, 0, mk(0, 0, 0), mk(0, 0, 0), ls);
            }
            head = (head + m_round) & (kRing - 1u);
            count -= m_round;
            MCPT_LSTAMP(2)
        }
    }
#ifdef MCPT_TRACE_DIAG
    if (!FIRST && (threadIdx.x & 63) == 0 && a.ctr) { for (int k = 0; k < 3; k++) atomicAdd(&a.ctr->pad[16 + k], dg[k]); atomicAdd(&a.ctr->pad[19], 1ull); }     // (the later passes' account)
#endif
    flush_stats(a.ctr, ls);
}

// ---------------------------------------------------------------------------------------------- finishing pass
// When few paths are left, a launch pair per bounce costs more than the paths: every launch starts with cold caches and runs at
// memory latency.  One persistent launch takes over the state logic(depth) has just written (its rays not yet traced) and runs
// every remaining path to its end.  A wave advances its 64 paths one vertex per iteration, in step: trace the shadow rays,
// resolve, trace the bounce ray, shade the next vertex (vertex.hpp: the arithmetic of k_wf_logic).  Lanes whose path has ended
// take the next unclaimed path (one atomic per refill on the pass's own count slot), so a wave is as long as its share of
// the work, not as its longest path.  A lane that has just adopted a path finds the rays of its first step in the wavefront
// state instead of computing them; from the second step on everything lives in registers.
__global__ void __launch_bounds__(256, MCPT_FINISH_WAVES) k_wf_finish(DScene S, WfArgs a)
{
    __shared__ int lds_stack[MCPT_FAST_STACK * 256];
    const long long n = a.counts->n_next;
    if (n == 0 || n > (long long)a.finish_below) return;               // nothing left, or still wavefront work
    const long long cap = a.cap;
    const int nl = a.nl;
    const bool folded = nl == 1;                                       // see k_wf_logic
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    LaneStats ls;
    Work w = {0, 0};
    int* stack = lds_stack + threadIdx.x;
    WfRaySource src; src.a = a; src.n_paths = n;
    bool queue_empty = false;

    enum { M_IDLE = 0, M_ADOPTED = 1, M_VERTEX = 2 };
#ifdef MCPT_TRACE_DIAG
    const unsigned long long fin_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long fin_iters = 0, fin_t_trace = 0;
#endif
    int mode = M_IDLE;
    long long j = 0;                    // M_ADOPTED: position in the wavefront state
    int id = 0, leaf = -1, in_type = RT_TRANSMISSION;
    uint32_t depth = 0;                 // M_ADOPTED: vertex whose rays are pending; M_VERTEX: vertex about to be shaded
    V3 T = mk(1, 1, 1), L = mk(0, 0, 0), dir = mk(0, 0, 0), p = mk(0, 0, 0);
    RngKey key;
    key.k0 = (uint32_t)a.seed; key.k1 = (uint32_t)(a.seed >> 32); key.pixel = 0; key.sample = 0;

    for (;;) {
        // ---- idle lanes adopt the next paths
        const unsigned long long idle = __ballot(mode == M_IDLE);
        if (idle && !queue_empty && (__popcll(idle) >= 16 || idle == ~0ull)) {
            const unsigned int want = (unsigned int)__popcll(idle);
            unsigned int got = 0;
            if (lane == 0) got = atomicAdd(&a.counts->pad[0], want);
            got = __shfl(got, 0, 64);
            if ((long long)got + want >= n) queue_empty = true;
            const long long mine = (long long)got + __popcll(idle & lt_mask);
            if (mode == M_IDLE && mine < n) {
                mode = M_ADOPTED; j = mine;
                id = a.out.id[j];
                depth = (uint32_t)a.depth;
                T = mk(1, 1, 1); L = mk(0, 0, 0);
                if (a.depth > 0 || folded) T = ldc(a.out.T, cap, j);
                if (a.depth > 0) L = ldc(a.out.L, cap, j);
                p = src.vertex(j);
                const int slot = a.first_slot + id / a.spp;
                key.pixel = (uint32_t)(a.pixels ? a.pixels[slot] : slot); key.sample = (uint32_t)(id % a.spp);
            }
        }
        if (!__ballot(mode != M_IDLE)) {
            if (queue_empty) break;
            continue;
        }

        // ---- lanes at a vertex: emitter test, surface (pathTracing.cpp:141-160)
        const DMaterial* m = nullptr;
        V3 pn = mk(0, 0, 0), kd = mk(0, 0, 0);
        bool ended = false;
        if (mode == M_VERTEX) {
            ls.shades++;
            if (depth > ls.depth) ls.depth = depth;
            m = S.materials + S.tris[leaf].material;
            if (m->light >= 0) {
                const V3 rad = ld3(S.lights[m->light].radiance);
                if (depth == 0) L = rad;
                else if (in_type != RT_DIFFUSE) L = L + mk(T.x * rad.x, T.y * rad.y, T.z * rad.z);
                ended = true;
            } else vertex_surface(S, leaf, p, m, pn, kd);
        }

        // ---- the rays of this step: one shadow ray per light and the bounce ray (Russian roulette + nextRay, or the stored ones)
        V3 L_dir = mk(0, 0, 0);
        int sample_mat = -1, expect0 = -2;
        V3 c0 = mk(0, 0, 0);
        Ray rs; rs.o = p; rs.d = mk(1, 1, 1);
        for (int l = 0; l < nl; l++) {
            int expect = -2;
            V3 c = mk(0, 0, 0);
            Ray r; r.o = p; r.d = mk(1, 1, 1);
            if (mode == M_VERTEX && !ended) {
                expect = light_sample(S, key, depth, l, p, pn, kd, sample_mat, r.d, c);
                if (expect != -2) ls.shadow++; else ls.skipped++;
            } else if (mode == M_ADOPTED) {
                expect = a.out.expect[(long long)l * cap + j];
                if (expect != -2) { c = ldc(a.out.c + (long long)l * 3 * cap, cap, j); r.d = ldc(a.rays.d + (long long)l * 3 * cap, cap, j); }
            }
            if (expect != -2) r.o = p + r.d * 0.01;
            if (nl == 1) { expect0 = expect; c0 = c; rs = r; break; }       // traced below, together with the bounce ray
            if (expect != -2) {
                Hit h;
                const bool ok = trace_lane_fast(S, r, h, w, stack, 256);
                const bool vis = (ok ? S.tris[h.leaf].material : -1) == expect;
                L_dir.x += vis ? c.x : c.x * 0.0;
                L_dir.y += vis ? c.y : c.y * 0.0;
                L_dir.z += vis ? c.z : c.z * 0.0;
            }
        }
        int bt = -1;
        V3 wgt = mk(1, 1, 1);
        Ray br; br.o = p; br.d = mk(1, 1, 1);
        if (mode == M_VERTEX && !ended) {
            bt = bounce_sample(key, depth, nl, m, dir, pn, kd, br.d, wgt);
            if (bt >= 0) ls.bounce++;
        } else if (mode == M_ADOPTED) {
            bt = a.out.btype[j];
            if (bt >= 0) { br.d = ldc(a.out.bdir, cap, j); if (!folded) wgt = ldc(a.out.w, cap, j); }
        }
        if (bt >= 0 && !(bt & MCPT_BT_NO_OFFSET)) br.o = p + br.d * 0.01;

#ifdef MCPT_TRACE_DIAG
        fin_iters++;
        const unsigned long long fin_ta = __builtin_amdgcn_s_memtime();
#endif
        // ---- closest hits.  One light: a lane with both rays hands its shadow ray to lane ^ 32 when that lane has no path, so
        // the two walks of a vertex run side by side -- it is the last few long paths, alone in their waves, that decide how long
        // this kernel runs.  Lanes without a free partner walk the shadow ray first and the bounce ray in a second round.
        bool b_ok = false;
        Hit b_hit; b_hit.leaf = -1; b_hit.t = 0; b_hit.p = mk(0, 0, 0);
        if (nl == 1) {
            const bool have_s = expect0 != -2, have_b = bt >= 0;
            const int partner = lane ^ 32;
            const unsigned long long idle_now = __ballot(mode == M_IDLE);
            const bool give = have_s && have_b && ((idle_now >> partner) & 1ull);
            Ray rin;
            rin.o = mk(__shfl(rs.o.x, partner, 64), __shfl(rs.o.y, partner, 64), __shfl(rs.o.z, partner, 64));
            rin.d = mk(__shfl(rs.d.x, partner, 64), __shfl(rs.d.y, partner, 64), __shfl(rs.d.z, partner, 64));
            const int partner_gives = __shfl((int)give, partner, 64);      // every lane takes part: not under a short-circuit
            const bool helping = mode == M_IDLE && partner_gives != 0;
            // first round: the helper's ray, or the bounce ray if the shadow ray was handed over, or the own shadow ray
            const int kind = helping ? 2 : (give ? 1 : (have_s ? 0 : (have_b ? 1 : -1)));
            Ray r1 = helping ? rin : (kind == 1 ? br : rs);
            Hit h1; h1.leaf = -1; h1.t = 0; h1.p = mk(0, 0, 0);
            bool ok1 = false;
            if (kind >= 0) ok1 = trace_lane_fast(S, r1, h1, w, stack, 256);
            const int mat1 = (kind == 0 || kind == 2) ? (ok1 ? S.tris[h1.leaf].material : -1) : -1;
            const int mat_helped = __shfl(mat1, partner, 64);
            if (have_s) {
                const bool vis = (give ? mat_helped : mat1) == expect0;
                L_dir.x += vis ? c0.x : c0.x * 0.0;
                L_dir.y += vis ? c0.y : c0.y * 0.0;
                L_dir.z += vis ? c0.z : c0.z * 0.0;
            }
            if (kind == 1) { b_ok = ok1; b_hit = h1; }
            const bool second = have_b && kind == 0;
            if (__ballot(second)) { if (second) b_ok = trace_lane_fast(S, br, b_hit, w, stack, 256); }
        } else if (bt >= 0) b_ok = trace_lane_fast(S, br, b_hit, w, stack, 256);
#ifdef MCPT_TRACE_DIAG
        fin_t_trace += __builtin_amdgcn_s_memtime() - fin_ta;
#endif
        if (mode == M_ADOPTED && folded) L = L + L_dir;                 // its c was stored as T * c
        else L = L + mk(T.x * L_dir.x, T.y * L_dir.y, T.z * L_dir.z);

        bool goes_on = false;
        if (bt >= 0 && b_ok) {
            // an adopted path with one light already holds the throughput after its bounce
            if (!(mode == M_ADOPTED && folded)) T = mk(T.x * wgt.x * MCPT_INV_P_RR, T.y * wgt.y * MCPT_INV_P_RR, T.z * wgt.z * MCPT_INV_P_RR);
            leaf = b_hit.leaf; p = b_hit.p; dir = neg(br.d); in_type = bt & 7; depth++;
            goes_on = true;
        }
        if (mode != M_IDLE) {
            if (goes_on) mode = M_VERTEX;
            else { a.rad[(size_t)id * 3] = L.x; a.rad[(size_t)id * 3 + 1] = L.y; a.rad[(size_t)id * 3 + 2] = L.z; mode = M_IDLE; }
        }
    }
#ifdef MCPT_TRACE_DIAG
    if (lane == 0 && a.ctr) {
        const unsigned long long life = __builtin_amdgcn_s_memtime() - fin_t0;
        atomicMax(&a.ctr->pad[13], fin_iters);
        atomicMax(&a.ctr->pad[14], life);
        atomicMax(&a.ctr->pad[15], fin_t_trace);
    }
#endif
    ls.nodes += w.nodes; ls.tris += w.tris;
    flush_stats(a.ctr, ls);
}

// ---------------------------------------------------------------------------------------------- finishing pass, pool form
// The same hand-over, run by the pool engine in path mode (trace_pool.hpp): one workgroup per CU whose LDS holds the rays of the
// paths it owns, a SHADE class beside the walk's classes, every path at its own pace.  What k_wf_finish does with 64 paths in
// lock-step per wave -- each step as long as its slowest walk -- this does with stateless steps at 45-50 lanes.
#ifndef MCPT_POOL_NPC
#define MCPT_POOL_NPC (MCPT_POOL_KT / 2)           /* path slots per lane the record planes are laid out for (one light: KT / 2 paths) */
#endif
struct WfPaths {
    static constexpr bool kPaths = true;
    WfArgs a;
    long long n;                // paths handed over (positions 0 .. n-1 of the wavefront state a.out)
    int nl, npc;
    float inv_r;                // 1 / (nl + 1)
    double* recd;               // [block][9 + 3 nl][npc * 64]: T, L, bounce weight, c per light
    int* reci;                  // [block][3 + nl][npc * 64]: sample id, depth, bounce type, expect per light
    int* gstack;                // [block][wave][lane][MCPT_FAST_STACK]: stack of the one-lane exact walk (rare rays)
    __device__ __forceinline__ int* lane_stack(int wave, int lane) const
    {
        return gstack + ((size_t)(blockIdx.x * MCPT_POOL_WAVES + wave) * 64 + lane) * MCPT_FAST_STACK;
    }
};

size_t finish_pool_bytes(int cus, int nl)
{
    if (nl < 1 || nl + 1 > MCPT_POOL_KT / 2) return 0;                 // (at least two paths per lane)
    const size_t blocks = size_t(cus > 0 ? cus : 256), plane = size_t(MCPT_POOL_NPC) * 64;
    return blocks * (size_t(9 + 3 * nl) * plane * sizeof(double) + size_t(3 + nl) * plane * sizeof(int) + size_t(MCPT_POOL_WAVES) * 64 * MCPT_FAST_STACK * sizeof(int));
}

template <int NW, int KT, int SCAP>
__global__ void __launch_bounds__(NW * 64, 1) k_wf_finish_pool(DScene S, WfArgs a, char* area, int* spill)
{
    const long long n = a.counts->n_next;
    if (n == 0 || n > (long long)a.finish_below) return;               // nothing left, or still wavefront work
    __shared__ PoolLds<NW, KT, SCAP> L;
    const size_t blocks = gridDim.x, plane = size_t(MCPT_POOL_NPC) * 64;
    WfPaths pp;
    pp.a = a; pp.n = n; pp.nl = a.nl; pp.npc = MCPT_POOL_NPC; pp.inv_r = 1.0f / (float)(a.nl + 1);
    pp.recd = reinterpret_cast<double*>(area);
    pp.reci = reinterpret_cast<int*>(pp.recd + blocks * size_t(9 + 3 * a.nl) * plane);
    pp.gstack = pp.reci + blocks * size_t(3 + a.nl) * plane;
    WfRaySource src; src.a = a; src.n_paths = n;                       // (path mode never fetches or stores through it)
    Work w = {0, 0};
    trace_pool<WfRaySource, NW, KT, SCAP, WfPaths>(S, src, nullptr, nullptr, 0u, 64, L, w, spill, pp);
    LaneStats ls;
    ls.nodes = w.nodes; ls.tris = w.tris;
    __syncthreads();
    if (threadIdx.x == 0 && a.ctr) {
        if (L.stat[0]) atomicAdd(&a.ctr->shade_calls, (unsigned long long)L.stat[0]);
        if (L.stat[1]) atomicAdd(&a.ctr->rays_shadow, (unsigned long long)L.stat[1]);
        if (L.stat[2]) atomicAdd(&a.ctr->rays_bounce, (unsigned long long)L.stat[2]);
        if (L.stat[3]) atomicAdd(&a.ctr->shadow_skipped, (unsigned long long)L.stat[3]);
        if (L.stat[4]) atomicMax(&a.ctr->max_depth, (unsigned long long)L.stat[4]);
    }
    flush_stats(a.ctr, ls);
}

// one PrimarySurface per hit pixel of the chunk (same arithmetic as the per-sample code it replaces: vertex_surface)
__global__ void k_primary_surface(DScene S, WfArgs a, PrimarySurface* __restrict__ surf, unsigned int* __restrict__ alive_count)
{
    const unsigned int n = a.counts_in->n_next;
    const unsigned int n_round = (n + 63u) / 64u * 64u;              // whole waves take part in the ballot below
    const int lane = threadIdx.x & 63;
    for (unsigned int h = blockIdx.x * blockDim.x + threadIdx.x; h < n_round; h += gridDim.x * blockDim.x) {
        bool shaded = false;
        PrimarySurface r;
        if (h < n) {
        const int slot = a.hit_slots[h];
        const PrimaryHit ph = a.hits[slot];
        const int pix = a.pixels ? a.pixels[slot] : slot;
        const V3 p = mk(ph.p[0], ph.p[1], ph.p[2]), dir = neg(ld3(a.dirs + (size_t)pix * 3));
        r.p[0] = p.x; r.p[1] = p.y; r.p[2] = p.z; r.dir[0] = dir.x; r.dir[1] = dir.y; r.dir[2] = dir.z;
        r.leaf = ph.leaf; r.material = S.tris[ph.leaf].material; r.pixel = pix; r.slot = slot;
        V3 pn = mk(0, 0, 0), kd = mk(0, 0, 0);
        const DMaterial* m = S.materials + r.material;
        if (m->light < 0) { vertex_surface(S, ph.leaf, p, m, pn, kd); shaded = true; }
        r.pn[0] = pn.x; r.pn[1] = pn.y; r.pn[2] = pn.z; r.kd[0] = kd.x; r.kd[1] = kd.y; r.kd[2] = kd.z;
        r.pad[0] = r.pad[1] = r.pad[2] = 0;
        }
        // the shaded pixels of the chunk are numbered in slot order (paths in another order trace measurably slower on a rank's
        // share of a frame): here the rank within the wave and the wave's count, k_alive_scan turns the counts into offsets
        const unsigned long long bal = __ballot(shaded);
        if (lane == 0) alive_count[h >> 6] = (unsigned int)__popcll(bal);
        if (h < n) {
            r.alive_index = shaded ? (int32_t)__popcll(bal & ((1ull << lane) - 1ull)) : -1;
            surf[h] = r;
        }
    }
}

// exclusive prefix over the per-wave counts of k_primary_surface (at most a few thousand: one block), total -> *total_out
__global__ void __launch_bounds__(1024) k_alive_scan(const WfCounts* __restrict__ counts_in, unsigned int* __restrict__ wave_count, unsigned int* __restrict__ total_out)
{
    __shared__ unsigned int part[1024];
    const unsigned int n_waves = (counts_in->n_next + 63u) / 64u;
    const unsigned int per = (n_waves + 1023u) / 1024u;
    const unsigned int b = threadIdx.x * per, e = b + per < n_waves ? b + per : n_waves;
    unsigned int sum = 0;
    for (unsigned int i = b; i < e; i++) sum += wave_count[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned int off = 1; off < 1024; off <<= 1) {
        const unsigned int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned int run = part[threadIdx.x] - sum;              // exclusive prefix of this thread's range
    for (unsigned int i = b; i < e; i++) { const unsigned int c = wave_count[i]; wave_count[i] = run; run += c; }
    if (threadIdx.x == 1023) *total_out = part[1023];
}


// slots of this chunk whose primary ray hit something, in slot order within a wave
__global__ void k_hit_slots(const PrimaryHit* __restrict__ hits, int first_slot, int n_slots, int32_t* __restrict__ hit_slots, unsigned int* count)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const bool hit = s < n_slots && hits[first_slot + s].leaf >= 0;
    const unsigned long long bal = __ballot(hit);
    const int lane = threadIdx.x & 63;
    unsigned int base = 0;
    if (lane == 0 && bal) base = atomicAdd(count, (unsigned int)__popcll(bal));
    base = __shfl(base, 0, 64);
    if (hit) hit_slots[base + __popcll(bal & ((1ull << lane) - 1ull))] = first_slot + s;
}

__global__ void k_zero(double* __restrict__ p, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0.0;
}

// ---------------------------------------------------------------------------------------------- launchers
static unsigned grid_for(long long n, int block, unsigned cap_blocks)
{
    long long b = (n + block - 1) / block;
    if (b < 1) b = 1;
    return (unsigned)(b > cap_blocks ? cap_blocks : b);
}

void launch_primary_surface(const DScene& S, const WfArgs& a, PrimarySurface* surf, unsigned int* alive_count, unsigned int* alive_total, int n_slots_upper,
                            hipStream_t st)
{
    if (n_slots_upper <= 0) return;
    hipLaunchKernelGGL(k_primary_surface, dim3(grid_for(n_slots_upper, 256, 4096)), dim3(256), 0, st, S, a, surf, alive_count);
    hipLaunchKernelGGL(k_alive_scan, dim3(1), dim3(1024), 0, st, a.counts_in, alive_count, alive_total);
}

// A resident-size grid whose blocks stride over the paths: starting a block of this kernel is expensive (large kernarg,
// 160+ VGPRs, scratch), so 768 long-lived blocks beat 16 k short ones by 15 % of a frame at N=1 and 30 % at one eighth
// of a frame (measured: MCPT_LOGIC_GRID sweep).
void launch_wf_logic(const DScene& S, const WfArgs& a, long long n_upper, bool first, hipStream_t st, const LaunchCfg& cfg)
{
    if (n_upper <= 0) return;
    // small inputs get small grids (>= 1024 paths per block): every wave that starts costs a few atomics on shared counters
    unsigned g = grid_for(n_upper, 1024, first ? cfg.logic_first : cfg.logic_rest);
    if (first) hipLaunchKernelGGL(k_wf_logic<true>, dim3(g), dim3(256), 0, st, S, a);
    else hipLaunchKernelGGL(k_wf_logic<false>, dim3(g), dim3(256), 0, st, S, a);
}

void launch_wf_finish(const DScene& S, const WfArgs& a, long long n_upper, hipStream_t st, const LaunchCfg& cfg, char* path_area, long long* slow_list,
                      unsigned int slow_cap)
{
    if (n_upper <= 0) return;
    // the pool form where the pool engine runs (small scenes) and a path's rays fit a lane's slots; a block per 64 x NP paths, at most one per CU
    if (cfg.finish_pool && path_area && a.nl + 1 <= MCPT_POOL_KT / 2) {
        const long long per_block = 64ll * (MCPT_POOL_KT / (a.nl + 1));
        const long long nb = (n_upper + per_block - 1) / per_block;
        const int g = (int)(nb < cfg.cus ? nb : cfg.cus);
        hipLaunchKernelGGL((k_wf_finish_pool<MCPT_POOL_WAVES, MCPT_POOL_KT, MCPT_POOL_STACK>), dim3(g), dim3(MCPT_POOL_WAVES * 64), 0, st, S, a, path_area,
                           reinterpret_cast<int*>(slow_list + slow_cap));
        return;
    }
    hipLaunchKernelGGL(k_wf_finish, dim3(grid_for(n_upper, 256, unsigned(cfg.finish_grid))), dim3(256), 0, st, S, a);
}

void launch_hit_slots(const PrimaryHit* hits, int first_slot, int n_slots, int32_t* hit_slots, unsigned int* count, hipStream_t st)
{
    if (n_slots <= 0) return;
    hipLaunchKernelGGL(k_hit_slots, dim3((n_slots + 255) / 256), dim3(256), 0, st, hits, first_slot, n_slots, hit_slots, count);
}

void launch_zero_rad(double* rad, long long n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_zero, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, rad, n);
}

void init_launch_cfg_logic(LaunchCfg& cfg, unsigned forced_grid)
{
    cfg.logic_first = forced_grid ? forced_grid : unsigned(persistent_grid(reinterpret_cast<const void*>(k_wf_logic<true>), cfg.cus));
    cfg.logic_rest = forced_grid ? forced_grid : unsigned(persistent_grid(reinterpret_cast<const void*>(k_wf_logic<false>), cfg.cus));
    cfg.finish_grid = persistent_grid(reinterpret_cast<const void*>(k_wf_finish), cfg.cus);
}

}  // namespace mcpt
