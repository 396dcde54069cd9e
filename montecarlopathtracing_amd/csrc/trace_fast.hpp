// Fast closest hit: identical results to the reference-shaped walk (dev_common.hpp: trace_closest), far fewer steps.
//
// Contract (proved in accel_build.cpp's header, checked ray-for-ray by tests/test_gpu_parity.py):
// for a ray with every |d_k| in [1e-100,1e100] and every o_k zero or in [1e-150,1e150] in magnitude, on a scene
// whose coordinates obey the same bound, ray_intersect's answer is the lexicographic minimum of (t_k, k) over the
// leaves k whose OWN box passes the reference's slab test, whose triangle test passes and whose t_k > 0.
// This walk
//   * culls inner nodes with a conservative slab test: the same (b-o) numerators, multiplied by 1/d instead of
//     divided, differ from the reference's quotients by < 3 ulp, so comparing with a 2^-48 relative slack never
//     rejects a box the reference would accept, and by monotonicity never loses a leaf below it;
//   * prunes by distance only beyond best_t + margin, margin = 1e-9 * scale / min|d_k| (>= 1e6 times the rounding
//     error of t_k = (p.x-o.x)/d.x relative to the slab entry), and not at all when min|d_k| < 1e-6;
//   * decides a triangle's own box from the same cheap interval when the outcome is certain (outside the 2^-48
//     band) and with the reference's six true divisions otherwise;
//   * runs the reference's triangle test and t_k computation unchanged.
// Every other ray falls back to trace_closest().  Per-lane traversal stack: 32 entries in LDS, [depth][lane] layout.
#pragma once
#include "dev_common.hpp"

namespace mcpt {

#define MCPT_FAST_STACK 32
#define MCPT_FAST_EMPTY (-2147483647 - 1)

struct Slab { double entry, exit; };

// per-axis parametric interval of a box from reciprocals; entry = max of the mins, exit = min of the maxes
__device__ __forceinline__ Slab slab_interval(const double lo[3], const double hi[3], const V3& o, const V3& r)
{
    const double ax = (lo[0] - o.x) * r.x, bx = (hi[0] - o.x) * r.x;
    const double ay = (lo[1] - o.y) * r.y, by = (hi[1] - o.y) * r.y;
    const double az = (lo[2] - o.z) * r.z, bz = (hi[2] - o.z) * r.z;
    Slab s;
    s.entry = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
    s.exit = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz));
    return s;
}

// conservative acceptance of an enclosing box: true whenever any box inside it passes the reference's test
__device__ __forceinline__ bool slab_may_hit(const Slab& s)
{
    return s.exit >= 0.0 && s.entry <= s.exit + s.exit * 0x1p-48;
}

// intersect(Ray&, boundingBox&) of the reference on explicit planes (true divisions)
__device__ __forceinline__ bool box_hit_exact(const double lo[3], const double hi[3], const Ray& r)
{
    double txmin = (lo[0] - r.o.x) / r.d.x, txmax = (hi[0] - r.o.x) / r.d.x;
    double tymin = (lo[1] - r.o.y) / r.d.y, tymax = (hi[1] - r.o.y) / r.d.y;
    double tzmin = (lo[2] - r.o.z) / r.d.z, tzmax = (hi[2] - r.o.z) / r.d.z;
    if (txmin > txmax) { const double tmp = txmin; txmin = txmax; txmax = tmp; }
    if (tymin > tymax) { const double tmp = tymin; tymin = tymax; tymax = tmp; }
    if (tzmin > tzmax) { const double tmp = tzmin; tzmin = tzmax; tzmax = tmp; }
    if (txmax < 0 || tymax < 0 || tzmax < 0) return false;
    if (txmin <= 0 && tymin <= 0 && tzmin <= 0) return true;
    return dmax3(txmin, tymin, tzmin) <= dmin3(txmax, tymax, tzmax);
}

__device__ __forceinline__ bool fast_path_ok(const DFast& F, const Ray& r)
{
    const double ax = fabs(r.d.x), ay = fabs(r.d.y), az = fabs(r.d.z);
    const bool d_ok = ax >= 1e-100 && ax <= 1e100 && ay >= 1e-100 && ay <= 1e100 && az >= 1e-100 && az <= 1e100;
    const double ox = fabs(r.o.x), oy = fabs(r.o.y), oz = fabs(r.o.z);
    const bool o_ok = (ox == 0.0 || (ox >= 1e-150 && ox <= 1e150)) && (oy == 0.0 || (oy >= 1e-150 && oy <= 1e150)) &&
                      (oz == 0.0 || (oz >= 1e-150 && oz <= 1e150));
    return F.enabled && d_ok && o_ok;      // NaNs fail every comparison -> false
}

// stack: LDS words, this lane's slots are stack[i * stride]
__device__ __forceinline__ bool trace_closest_fast(const DScene& S, const Ray& r, Hit& best, Work& w, int* __restrict__ stack, int stride)
{
    const DFast& F = S.fast;
    if (!fast_path_ok(F, r)) return trace_closest(S, r, best, w);

    const V3 rcp = mk(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    const double dmin = fmin(fmin(fabs(r.d.x), fabs(r.d.y)), fabs(r.d.z));
    const double scale = fmax(fmax(F.absmax, fabs(r.o.x)), fmax(fabs(r.o.y), fabs(r.o.z)));
    const double margin = dmin >= 1e-6 ? 1e-9 * scale / dmin : __builtin_inf();
    const FastNode* __restrict__ nodes = F.nodes;
    const DTri* __restrict__ tris = F.tris;

    bool found = false;
    best.leaf = -1; best.t = 0; best.p = mk(0, 0, 0);
    double limit = __builtin_inf();          // best.t + margin once something is found
    int sp = 0;
    int cur = 0;                             // root
    for (;;) {
        while (cur >= 0) {                   // inner nodes
            const FastNode* nd = nodes + cur;
            w.nodes++;
            const Slab s0 = slab_interval(nd->lo[0], nd->hi[0], r.o, rcp);
            const Slab s1 = slab_interval(nd->lo[1], nd->hi[1], r.o, rcp);
            const int c0 = nd->child[0], c1 = nd->child[1];
            const bool h0 = c0 != MCPT_FAST_EMPTY && slab_may_hit(s0) && !(s0.entry > limit);
            const bool h1 = c1 != MCPT_FAST_EMPTY && slab_may_hit(s1) && !(s1.entry > limit);
            if (h0 && h1) {
                const bool first0 = s0.entry <= s1.entry;
                stack[sp * stride] = first0 ? c1 : c0;
                sp++;
                cur = first0 ? c0 : c1;
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else if (sp > 0) { sp--; cur = stack[sp * stride]; }
            else cur = MCPT_FAST_EMPTY;
        }
        if (cur == MCPT_FAST_EMPTY) break;
        {                                    // leaf
            const int ref = -1 - cur;
            const int first = ref >> 4, count = (ref & 15) + 1;
            for (int i = 0; i < count; i++) {
                const DTri* tr = tris + first + i;
                // the reference's leaf box of this triangle (findBondingBox(Face&), BVH.cpp:87-97)
                double lo[3], hi[3];
                lo[0] = dmin3(tr->v1[0], tr->v2[0], tr->v3[0]); hi[0] = dmax3(tr->v1[0], tr->v2[0], tr->v3[0]);
                lo[1] = dmin3(tr->v1[1], tr->v2[1], tr->v3[1]); hi[1] = dmax3(tr->v1[1], tr->v2[1], tr->v3[1]);
                lo[2] = dmin3(tr->v1[2], tr->v2[2], tr->v3[2]); hi[2] = dmax3(tr->v1[2], tr->v2[2], tr->v3[2]);
                const Slab s = slab_interval(lo, hi, r.o, rcp);
                if (s.exit < 0.0) continue;                                  // a tmax < 0: the sign of a quotient is exact
                if (s.entry > limit) continue;                               // cannot beat the current best
                bool pass;
                if (s.entry <= 0.0) pass = true;                             // every tmin <= 0 (signs exact)
                else if (s.entry + s.entry * 0x1p-48 <= s.exit) pass = true; // certainly dmax(tmin) <= dmin(tmax)
                else if (s.entry > s.exit + s.exit * 0x1p-48) pass = false;  // certainly not
                else pass = box_hit_exact(lo, hi, r);                        // inside the rounding band: ask the reference
                if (!pass) continue;
                V3 p;
                w.tris++;
                if (tri_hit(tr, r, p)) {
                    const double t = (p.x - r.o.x) / r.d.x;                  // pathTracing.cpp:347
                    const int k = tr->leaf;
                    if (t > 0 && (!found || t < best.t || (t == best.t && k < best.leaf))) {
                        found = true; best.leaf = k; best.t = t; best.p = p;
                        limit = t + margin;
                    }
                }
            }
        }
        if (sp > 0) { sp--; cur = stack[sp * stride]; }
        else break;
    }
    return found;
}

}  // namespace mcpt
