// Rays of a wavefront iteration as the trace engines see them (Src of trace_persistent.hpp / trace_pool.hpp): slot q = l * n_paths + j,
// l in [0, nl] (l == nl: the bounce ray of path j), rebuilt from the path state k_wf_logic wrote (wavefront.hpp).
#pragma once
#include "dev_common.hpp"
#include "vertex.hpp"
#include "wavefront.hpp"

namespace mcpt {

// ray slot q = l*n_paths + j, l in [0, nl] (l == nl: the bounce ray of path j)
struct WfRaySource {
    static constexpr bool kWantsPoint = false;      // results are a leaf or a material: the hit point is formed again by the next logic pass
    WfArgs a;
    long long n_paths;
    __device__ __forceinline__ long long total() const { return n_paths * (a.nl + 1); }
    // (l, j) of slot q without a 64-bit division: nl is small
    __device__ __forceinline__ void split(long long q, int& l, long long& j) const
    {
        l = 0; j = q;
        while (j >= n_paths) { j -= n_paths; l++; }
    }
    // the vertex the rays of path j leave from: after the first pass every sample of a pixel still sits on its primary hit
    __device__ __forceinline__ V3 vertex(long long j) const
    {
        if (a.depth == 0) {
            const PrimaryHit* ph = a.hits + (a.first_slot + a.out.id[j] / a.spp);
            return mk(ph->p[0], ph->p[1], ph->p[2]);
        }
        return ldc(a.out.p, a.cap, j);
    }
    __device__ __forceinline__ bool fetch(long long q, Ray& r) const
    {
        int l; long long j;
        split(q, l, j);
        // branch-free: the ray words are loaded whether or not the slot is in use, so nothing waits on the flag
        const bool bounce = l == a.nl;
        const int flag = bounce ? a.out.btype[j] : a.out.expect[(long long)l * a.cap + j];
        const V3 p = vertex(j);
        r.d = ldc(bounce ? a.out.bdir : a.rays.d + (long long)l * 3 * a.cap, a.cap, j);
        r.o = (bounce && (flag & MCPT_BT_NO_OFFSET)) ? p : p + r.d * 0.01;
        return bounce ? flag >= 0 : flag != -2;
    }
    __device__ __forceinline__ void store(long long q, bool ok, const Hit& h) const
    {
        int l; long long j;
        split(q, l, j);
        if (l == a.nl) {
            int v = -1;
            if (ok) { const int mat = h.mat >= 0 ? h.mat : a.tris[h.leaf].material; v = h.leaf | (a.materials[mat].light >= 0 ? MCPT_HIT_EMITTER : 0); }
            a.out.hit_leaf[j] = v;
        } else {
            // (the persistent engine hands the material over with the leaf: one dependent fetch less per shadow ray in its store batch)
            a.out.hit_mat[(long long)l * a.cap + j] = ok ? (h.mat >= 0 ? h.mat : a.tris[h.leaf].material) : -1;
        }
    }
};

// The same source reading the launch's arguments where the runtime put them -- the kernarg segment -- each time a ray is fetched or
// stored, instead of holding them in scalar registers for the length of the kernel.  A trace kernel keeps ~25 pointers and a dozen
// scalars of WfArgs alive across its walk loop although only the refill / store section (one step in nine) looks at them; with the
// loop's own uniform state that is more than the 102 SGPRs a wave has, and the excess lives in VGPR lanes (29 spilled SGPRs in the
// pool kernel, 49 in the voting engine's: 172 v_readlane_b32 in its ISA, each with its wait states, several of them in the node step).
// Scalar loads from the kernarg segment hit the scalar cache; the empty asm keeps the compiler from hoisting them out of the loop again.
typedef const WfArgs __attribute__((address_space(4)))* WfArgsKernarg;
template <class T> __device__ __forceinline__ const T __attribute__((address_space(1)))* wf_glob(const T* p)
{
    return (const T __attribute__((address_space(1)))*)p;
}
template <class T> __device__ __forceinline__ T __attribute__((address_space(1)))* wf_glob_mut(T* p)
{
    return (T __attribute__((address_space(1)))*)p;
}
// MCPT_TRACE_NT = 1: the trace kernels' reads of the path state (a stream: every ray is fetched once) and their answers go past the caches'
// normal replacement (non-temporal), so that they do not push nodes and triangles out of L1 / L2.  Measured in round 4: no difference
// (7.16 against 7.13 ms per launch; the logic kernel's own accesses, wavefront_logic.hip, are where the hint pays).
#ifndef MCPT_TRACE_NT
#define MCPT_TRACE_NT 0
#endif
#if MCPT_TRACE_NT
template <class P> __device__ __forceinline__ auto wf_sld(P p) { return __builtin_nontemporal_load(p); }
template <class P, class T> __device__ __forceinline__ void wf_sst(P p, T v) { __builtin_nontemporal_store(v, p); }
#else
template <class P> __device__ __forceinline__ auto wf_sld(P p) { return *p; }
template <class P, class T> __device__ __forceinline__ void wf_sst(P p, T v) { *p = v; }
#endif
struct WfRaySourceK {
    static constexpr bool kWantsPoint = false;
    WfArgsKernarg ap;
    long long n_paths;
    int nl;
    __device__ __forceinline__ WfArgsKernarg args() const { WfArgsKernarg p = ap; __asm__ volatile("" : "+s"(p)); return p; }
    __device__ __forceinline__ long long total() const { return n_paths * (nl + 1); }
    __device__ __forceinline__ void split(long long q, int& l, long long& j) const
    {
        l = 0; j = q;
        while (j >= n_paths) { j -= n_paths; l++; }
    }
    __device__ __forceinline__ bool fetch(long long q, Ray& r) const
    {
        int l; long long j;
        split(q, l, j);
        const WfArgsKernarg A = args();
        const long long cap = A->cap;
        const bool bounce = l == nl;
        const int flag = bounce ? wf_sld(wf_glob(A->out.btype) + j) : wf_sld(wf_glob(A->out.expect) + ((long long)l * cap + j));
        V3 p;
        if (A->depth == 0) {
            const auto* ph = wf_glob(A->hits) + (A->first_slot + wf_glob(A->out.id)[j] / A->spp);
            p = mk(ph->p[0], ph->p[1], ph->p[2]);
        } else {
            const auto* g = wf_glob(A->out.p);
            p = mk(wf_sld(g + j), wf_sld(g + cap + j), wf_sld(g + 2 * cap + j));
        }
        const auto* d = wf_glob(bounce ? A->out.bdir : A->rays.d + (long long)l * 3 * cap);
        r.d = mk(wf_sld(d + j), wf_sld(d + cap + j), wf_sld(d + 2 * cap + j));
        r.o = (bounce && (flag & MCPT_BT_NO_OFFSET)) ? p : p + r.d * 0.01;
        return bounce ? flag >= 0 : flag != -2;
    }
    __device__ __forceinline__ void store(long long q, bool ok, const Hit& h) const
    {
        int l; long long j;
        split(q, l, j);
        const WfArgsKernarg A = args();
        if (l == nl) {
            int v = -1;
            if (ok) { const int mat = h.mat >= 0 ? h.mat : wf_glob(A->tris)[h.leaf].material; v = h.leaf | (wf_glob(A->materials)[mat].light >= 0 ? MCPT_HIT_EMITTER : 0); }
            wf_sst(wf_glob_mut(A->out.hit_leaf) + j, v);
        }
        else wf_sst(wf_glob_mut(A->out.hit_mat) + ((long long)l * A->cap + j), ok ? (h.mat >= 0 ? h.mat : wf_glob(A->tris)[h.leaf].material) : -1);
    }
};

}  // namespace mcpt
