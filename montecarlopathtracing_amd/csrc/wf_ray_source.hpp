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
            a.out.hit_leaf[j] = ok ? h.leaf : -1;
        } else {
            // (the persistent engine hands the material over with the leaf: one dependent fetch less per shadow ray in its store batch)
            a.out.hit_mat[(long long)l * a.cap + j] = ok ? (h.mat >= 0 ? h.mat : a.tris[h.leaf].material) : -1;
        }
    }
};

}  // namespace mcpt
