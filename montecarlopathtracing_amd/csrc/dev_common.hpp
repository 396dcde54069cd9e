// Device-side building blocks shared by every kernel: fp64 vectors in the reference's operation order, the RNG
// seam, the reference's hit tests, and the reference-shaped (stackless) closest-hit walk.
// Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.hpp"

namespace mcpt {

#define MCPT_PI 3.1415926      /* pathTracing.h:11 */
#define MCPT_P_RR 0.6          /* pathTracing.cpp:237 */

// ------------------------------------------------------------------------------------------------ vectors
struct V3 { double x, y, z; };
__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, double t) { return mk(a.x * t, a.y * t, a.z * t); }
__device__ __forceinline__ V3 operator/(V3 a, double m) { return mk(a.x / m, a.y / m, a.z / m); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Vertex::cross, sceneManagement.h:68-74
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return mk(a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y); }
__device__ __forceinline__ double norm(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 normalized(V3 a) { double d = norm(a); return mk(a.x / d, a.y / d, a.z / d); }

// component-major SoA ([component][cap]: the wavefront path state, wavefront.hpp): element i of a 3-vector array
__device__ __forceinline__ V3 ldc(const double* __restrict__ a, long long cap, long long i)
{
    return mk(a[i], a[cap + i], a[2 * cap + i]);
}
__device__ __forceinline__ void stc(double* __restrict__ a, long long cap, long long i, V3 v)
{
    a[i] = v.x; a[cap + i] = v.y; a[2 * cap + i] = v.z;
}

// dmin / dmax, sceneManagement.cpp:3-15 (if-chains with their NaN fall-through)
__device__ __forceinline__ double dmin3(double p1, double p2, double p3)
{
    if (p1 <= p2 && p1 <= p3) return p1;
    else if (p2 <= p1 && p2 <= p3) return p2;
    else return p3;
}
__device__ __forceinline__ double dmax3(double p1, double p2, double p3)
{
    if (p1 >= p2 && p1 >= p3) return p1;
    else if (p2 >= p1 && p2 >= p3) return p2;
    else return p3;
}

// ------------------------------------------------------------------------------------------------ RNG seam (D1)
// Philox4x32-10, counter (pixel, sample, depth<<16 | block, 'MCPT'), key = seed.  A path vertex numbers its uniforms by slot
// (light l: 4l..4l+3, then RR, FRESNEL, LOBE, PHI, THETA); slot s is word s & 3 of block s >> 2 and a word w becomes
// (w + 0.5) * 2^-32, a uniform on the open interval (0,1) with 32 random bits -- nl + 2 blocks per vertex (round 1 spent a
// block on two 53-bit uniforms, 2 nl + 3 blocks per vertex; the generator was a fifth of the logic kernels' instructions).
struct Philox { uint32_t v[4]; };
__device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#ifndef MCPT_PHILOX_ROUNDS
#define MCPT_PHILOX_ROUNDS 10          /* anything else is a timing experiment, not a generator */
#endif
#pragma unroll
    for (int r = 0; r < MCPT_PHILOX_ROUNDS; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox p; p.v[0] = c0; p.v[1] = c1; p.v[2] = c2; p.v[3] = c3;
    return p;
}
__device__ __forceinline__ double word_to_unit(uint32_t w) { return __builtin_fma((double)w, 0x1p-32, 0x1p-33); }   // exact
struct RngKey { uint32_t k0, k1, pixel, sample; };
// slots 4b .. 4b+3
__device__ __forceinline__ void uniform4(const RngKey& k, uint32_t depth, uint32_t block, double& u0, double& u1, double& u2, double& u3)
{
    const Philox p = philox4x32_10(k.pixel, k.sample, (depth << 16) | block, 0x4D435054u, k.k0, k.k1);
    u0 = word_to_unit(p.v[0]); u1 = word_to_unit(p.v[1]); u2 = word_to_unit(p.v[2]); u3 = word_to_unit(p.v[3]);
}
// slot 4b
__device__ __forceinline__ double uniform1(const RngKey& k, uint32_t depth, uint32_t block)
{
    return word_to_unit(philox4x32_10(k.pixel, k.sample, (depth << 16) | block, 0x4D435054u, k.k0, k.k1).v[0]);
}

// ------------------------------------------------------------------------------------------------ hit tests
struct Ray { V3 o, d; };

// intersect(Ray&, boundingBox&), sceneManagement.cpp:340-391: six true divisions, swap, reject if any
// tmax < 0, accept if every tmin <= 0, else dmax(tmin) <= dmin(tmax).
__device__ __forceinline__ bool box_hit(const DNode* __restrict__ nd, const Ray& r)
{
    const double2* q = reinterpret_cast<const double2*>(nd);
    const double2 a = q[0], b = q[1], c = q[2];      // mn.x mn.y | mn.z mx.x | mx.y mx.z
    double txmin = (a.x - r.o.x) / r.d.x;
    double txmax = (b.y - r.o.x) / r.d.x;
    double tymin = (a.y - r.o.y) / r.d.y;
    double tymax = (c.x - r.o.y) / r.d.y;
    double tzmin = (b.x - r.o.z) / r.d.z;
    double tzmax = (c.y - r.o.z) / r.d.z;
    if (txmin > txmax) { const double tmp = txmin; txmin = txmax; txmax = tmp; }
    if (tymin > tymax) { const double tmp = tymin; tymin = tymax; tymax = tmp; }
    if (tzmin > tzmax) { const double tmp = tzmin; tzmin = tzmax; tzmax = tmp; }
    if (txmax < 0 || tymax < 0 || tzmax < 0) return false;
    if (txmin <= 0 && tymin <= 0 && tzmin <= 0) return true;
    return dmax3(txmin, tymin, tzmin) <= dmin3(txmax, tymax, tzmax);
}

// intersect(Ray&, Face&, Vertex&), sceneManagement.cpp:316-338: plane hit + three same-side edge tests.
__device__ __forceinline__ bool tri_hit(const DTri* __restrict__ tr, const Ray& r, V3& p)
{
    const V3 v1 = ld3(tr->v1), v2 = ld3(tr->v2), v3 = ld3(tr->v3), n = ld3(tr->n);
    const double t = dot(v1 - r.o, n) / dot(n, r.d);
    p = r.o + r.d * t;
    const V3 ap = p - v1, bp = p - v2, cp = p - v3;
    const V3 ab = v2 - v1, bc = v3 - v2, ca = v1 - v3;
    const double dir1 = dot(cross(ab, ap), n), dir2 = dot(cross(bc, bp), n), dir3 = dot(cross(ca, cp), n);
    const double j1 = dir1 * dir2, j2 = dir1 * dir3, j3 = dir2 * dir3;
    return j1 >= 0 && j2 >= 0 && j3 >= 0;
}

// findGarCor, pathTracing.cpp:394-432
__device__ __forceinline__ V3 barycentric(V3 v1, V3 v2, V3 v3, V3 p)
{
    const V3 e1 = v3 - v2, e2 = v1 - v3, e3 = v2 - v1;
    const V3 d1 = p - v1, d2 = p - v2, d3 = p - v3;
    const V3 n = cross(e1, e2);
    const double an = dot(n, n);
    return mk(dot(cross(e1, d3), n) / an, dot(cross(e2, d1), n) / an, dot(cross(e3, d2), n) / an);
}

struct Hit {
    int leaf; double t; V3 p;
    int mat = -1;       // material of the hit triangle where the walk has it at hand (the persistent engine: same record as `leaf`), else -1
};
struct Work {
    uint32_t nodes, tris;       // compressed nodes stepped on, triangles visited (persistent engine: put through the pre-test)
    uint32_t rays = 0;          // rays started by the persistent engine
    uint32_t exact = 0;         // persistent engine: triangles that needed the reference's fp64 test
    unsigned long long* dbg = nullptr;      // MCPT_PRE_CHECK builds: 24 words for the first offending triangle
    uint32_t pre_wrong = 0;     // MCPT_PRE_CHECK builds: triangles the pre-test rejected although the exact test makes them candidates (must be 0)
#ifdef MCPT_TRACE_DIAG
    // [0..5] iterations and waiting lanes of the inner / pre-test / exact phase, [6] idle lanes, [8..11] cycles in refill / inner / pre-test / exact
    unsigned long long diag[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
};

// ray_intersect / bvh_intersect (pathTracing.cpp:334-390) without recursion and without a stack.
// The tree is the reference's implicit complete tree: node i (heap numbering) at level l has children 2i+1,
// 2i+2; its record sits at i - Nv(l) (BVH::findIndex).  Pre-order "both children, left first" is walked by
// index arithmetic alone: descend = (2i+1, l+1); leaving a finished subtree = strip the trailing 1-bits of
// i+1 (climb while we are a right child) and step to the right sibling, which is virtual only when everything
// further right is virtual too (virtual nodes are the tail of every level), i.e. when the walk is over.
// Virtual children are skipped (D5).  Closest = smallest t_x with strict '<', so ties keep the earlier leaf.
__device__ __forceinline__ bool trace_closest(const DScene& S, const Ray& r, Hit& best, Work& w)
{
    const int Level = S.Level, Lv = S.Lv;
    const DNode* __restrict__ nodes = S.nodes;
    const DTri* __restrict__ tris = S.tris;
    bool flag = false;
    best.leaf = -1; best.t = 0; best.p = mk(0, 0, 0);
    uint32_t i = 0;
    int l = 0;
    const uint32_t leaf0 = (1u << Level) - 1u;
    for (;;) {
        const int lvl = Lv >> (Level - l + 1);
        const uint32_t idx = i - (uint32_t)(2 * lvl - __popc(lvl));
        w.nodes++;
        const bool inside = box_hit(nodes + idx, r);
        if (inside && l == Level) {
            const int k = (int)(i - leaf0);
            V3 p;
            w.tris++;
            if (tri_hit(tris + k, r, p)) {
                const double t = (p.x - r.o.x) / r.d.x;          // pathTracing.cpp:347
                if (!flag) { if (t > 0) { flag = true; best.leaf = k; best.t = t; best.p = p; } }
                else if (t > 0 && t < best.t) { best.leaf = k; best.t = t; best.p = p; }
            }
        }
        if (inside && l < Level) { i = 2u * i + 1u; l++; continue; }
        // leave this subtree
        uint32_t x = i + 1u;
        const int up = __ffs((int)~x) - 1;                        // trailing ones of x
        x >>= up; l -= up;
        if (x == 0u) break;                                       // came up the right spine: done
        const uint32_t end_l = (2u << l) - 1u - (uint32_t)(Lv >> (Level - l));
        if (x >= end_l) break;                                    // right sibling is virtual: done
        i = x;                                                    // 0-based index of the sibling (x+1)-1
    }
    return flag;
}

// interpolated, un-normalised normal of the accepted hit (pathTracing.cpp:350-351)
__device__ __forceinline__ V3 hit_normal(const DScene& S, const Hit& h)
{
    const DTri* tr = S.tris + h.leaf;
    const DTriShade* sh = S.shade + h.leaf;
    const V3 g = barycentric(ld3(tr->v1), ld3(tr->v2), ld3(tr->v3), h.p);
    return (ld3(sh->vn1) * g.x + ld3(sh->vn2) * g.y) + ld3(sh->vn3) * g.z;
}

// ------------------------------------------------------------------------------------------------ counters
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ unsigned int wave_max(unsigned int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { unsigned int o = __shfl_down(v, off, 64); v = o > v ? o : v; }
    return v;
}
struct LaneStats { uint32_t nodes = 0, tris = 0, shadow = 0, bounce = 0, primary = 0, shades = 0, samples = 0, depth = 0, skipped = 0; };
__device__ __forceinline__ void flush_stats(DCounters* c, const LaneStats& s)
{
    if (!c) return;
    const unsigned long long n = wave_sum(s.nodes), t = wave_sum(s.tris), sh = wave_sum(s.shadow), bo = wave_sum(s.bounce),
                             pr = wave_sum(s.primary), sc = wave_sum(s.shades), sa = wave_sum(s.samples), sk = wave_sum(s.skipped);
    const unsigned int md = wave_max(s.depth);
    if ((threadIdx.x & 63) == 0) {
        if (n) atomicAdd(&c->node_visits, n);
        if (t) atomicAdd(&c->tri_tests, t);
        if (sh) atomicAdd(&c->rays_shadow, sh);
        if (bo) atomicAdd(&c->rays_bounce, bo);
        if (pr) atomicAdd(&c->rays_primary, pr);
        if (sc) atomicAdd(&c->shade_calls, sc);
        if (sa) atomicAdd(&c->samples, sa);
        if (sk) atomicAdd(&c->shadow_skipped, sk);
        if (md) atomicMax(&c->max_depth, (unsigned long long)md);
    }
}


}  // namespace mcpt
