// Fast closest hit: identical results to the reference-shaped walk (dev_common.hpp: trace_closest), far fewer steps.
//
// Contract (proved in accel_build.cpp's header, checked ray-for-ray by tests/test_gpu_parity.py):
// for a ray with every |d_k| in [1e-100,1e100] and every o_k zero or in [1e-150,1e150] in magnitude, on a scene
// whose coordinates obey the same bound, ray_intersect's answer is the lexicographic minimum of (t_k, k) over the
// leaves k whose OWN box passes the reference's slab test, whose triangle test passes and whose t_k > 0.
// This walk
//   * culls inner nodes conservatively on a compressed 4-wide hierarchy (CwNode: 8-bit child boxes rounded outward,
//     tested in fp32 with a rigorous error pad, see make_rayf): a box the reference would accept is never rejected,
//     and by monotonicity no leaf below it is lost;
//   * prunes by distance only beyond best_t + margin, margin = 1e-9 * scale / min|d_k| (>= 1e6 times the rounding
//     error of t_k = (p.x-o.x)/d.x relative to the slab entry), and not at all when min|d_k| < 1e-6;
//   * decides a triangle's own box from reciprocals when the outcome is certain (outside the 2^-48 band, see
//     own_box_hit) and with the reference's six true divisions otherwise;
//   * runs the reference's triangle test and t_k computation unchanged.
// Every other ray takes trace_closest().  The walk itself is in trace_persistent.hpp (per-lane stack: 32 entries in LDS).
#pragma once
#include "dev_common.hpp"

namespace mcpt {

#ifndef MCPT_FAST_STACK
#define MCPT_FAST_STACK 36     /* per-lane traversal stack entries in LDS (36 KB per block); accel_build.hpp bounds the
                                  hierarchy to it: 35 = 3 x 11 levels of the device-built 4-wide tree (16.7 M triangles) + slack */
#endif
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

// the reference's box test certainly passes for THIS box (not merely for something inside it): every tmax >= 0 and
// either every tmin <= 0 or dmax(tmin) <= dmin(tmax) with the rounding band excluded
__device__ __forceinline__ bool slab_certain_pass(const Slab& s)
{
    return s.exit >= 0.0 && (s.entry <= 0.0 || s.entry + s.entry * 0x1p-48 <= s.exit);
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

// v_min_f64 / v_max_f64 as they are (the compiler's fmin/fmax first quiets each operand with an extra instruction; the operands
// here are finite)
__device__ __forceinline__ double vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// The reference's box decision for a triangle's OWN box (bvh_intersect tests the leaf's box before the triangle,
// pathTracing.cpp:334-345), from reciprocals wherever the outcome is certain.  The signs of (b-o)*(1/d) and (b-o)/d agree
// (no underflow inside fast_path_ok's ranges), which settles "some tmax < 0" and "every tmin <= 0" exactly.  What remains is
// dmax3(tmin) <= dmin3(tmax), i.e. tmin_a <= tmax_b for all axes a, b; for a == b it holds by construction (the reference
// swaps), so only the six cross-axis pairs are compared -- this is what keeps a flat box (an axis-aligned triangle:
// tmin == tmax on one axis) out of the six-division fallback.  Products and quotients differ by < 2^-50 relative; a pair is
// decided from the products when they are more than 2^-48 apart.
__device__ __forceinline__ bool own_box_hit(const DTri* __restrict__ tr, const Ray& r, const V3& rcp)
{
    // the leaf's box (BVH.cpp:87-97); vertices are finite (fast_path_ok's scene condition), so min/max instructions give the
    // reference's if-chain values up to the sign of a zero, which no comparison below can see
    double lo[3], hi[3];
    lo[0] = vmin(vmin(tr->v1[0], tr->v2[0]), tr->v3[0]); hi[0] = vmax(vmax(tr->v1[0], tr->v2[0]), tr->v3[0]);
    lo[1] = vmin(vmin(tr->v1[1], tr->v2[1]), tr->v3[1]); hi[1] = vmax(vmax(tr->v1[1], tr->v2[1]), tr->v3[1]);
    lo[2] = vmin(vmin(tr->v1[2], tr->v2[2]), tr->v3[2]); hi[2] = vmax(vmax(tr->v1[2], tr->v2[2]), tr->v3[2]);
    const double ax = (lo[0] - r.o.x) * rcp.x, bx = (hi[0] - r.o.x) * rcp.x;
    const double ay = (lo[1] - r.o.y) * rcp.y, by = (hi[1] - r.o.y) * rcp.y;
    const double az = (lo[2] - r.o.z) * rcp.z, bz = (hi[2] - r.o.z) * rcp.z;
    const double nx = vmin(ax, bx), fx = vmax(ax, bx);
    const double ny = vmin(ay, by), fy = vmax(ay, by);
    const double nz = vmin(az, bz), fz = vmax(az, bz);
    if (fx < 0.0 || fy < 0.0 || fz < 0.0) return false;
    if (nx <= 0.0 && ny <= 0.0 && nz <= 0.0) return true;
    const double ex = vmin(fy, fz), ey = vmin(fx, fz), ez = vmin(fx, fy);          // exits of the other two axes, all >= 0
    if (nx + fabs(nx) * 0x1p-48 <= ex && ny + fabs(ny) * 0x1p-48 <= ey && nz + fabs(nz) * 0x1p-48 <= ez) return true;
    if (nx > ex + ex * 0x1p-48 || ny > ey + ey * 0x1p-48 || nz > ez + ez * 0x1p-48) return false;
    return box_hit_exact(lo, hi, r);
}

// A triangle's test passed at p.  Does it replace the best candidate?  The remaining conditions of the reference are a
// conjunction of pure tests (own box, t > 0, (t, k) lexicographically below the best), evaluated cheapest first:
// sign and rank of t from the reciprocal where that is certain, then the box, then t itself with the reference's division.
__device__ __forceinline__ bool better_candidate(const DTri* __restrict__ tr, const Ray& r, const V3& rcp, const V3& p, bool found,
                                                 const Hit& best, double& t, int& k)
{
    const double ta = (p.x - r.o.x) * rcp.x;                // same sign as t, within 2^-50 of it
    if (!(ta > 0.0)) return false;
    if (found && ta > best.t + best.t * 0x1p-48) return false;
    if (!own_box_hit(tr, r, rcp)) return false;
    t = (p.x - r.o.x) / r.d.x;                              // pathTracing.cpp:347
    k = tr->leaf;
    return t > 0 && (!found || t < best.t || (t == best.t && k < best.leaf));
}

// Rays the fast walk may take.  fp64 side (exact decisions at the leaves): no under/overflow in (b-o)*(1/d);
// fp32 side (conservative culling on compressed nodes): every product stays finite and above the denormal range.
__device__ __forceinline__ bool fast_path_ok(const DFast& F, const Ray& r)
{
    const double ax = fabs(r.d.x), ay = fabs(r.d.y), az = fabs(r.d.z);
    const bool d_ok = ax >= 1e-15 && ax <= 1e15 && ay >= 1e-15 && ay <= 1e15 && az >= 1e-15 && az <= 1e15;
    const double ox = fabs(r.o.x), oy = fabs(r.o.y), oz = fabs(r.o.z);
    const bool o_ok = (ox == 0.0 || (ox >= 1e-150 && ox <= 1e15)) && (oy == 0.0 || (oy >= 1e-150 && oy <= 1e15)) &&
                      (oz == 0.0 || (oz >= 1e-150 && oz <= 1e15));
    return F.enabled && d_ok && o_ok;      // NaNs fail every comparison -> false
}

// Per-ray constants of the conservative fp32 test on compressed nodes.
// For a plane at p + q*2^e the parametric distance is t(q) = q*(2^e r) + (p - o) r.  Computed in fp32 from
// o32 = fl(o), r32 = fl(1/d) it differs from the real value by less than 6 eps |r| (|o| + 3S) (S = largest scene
// coordinate; q*2^e <= 2S, |p| <= S), so subtracting / adding pad = 16 eps |r| (|o| + 3S) gives a rigorous lower bound
// of the slab entry and upper bound of the slab exit of the decoded box, which itself contains the child's fp64 box.
struct RayF {
    float o[3], r[3], pad[3];
};
__device__ __forceinline__ RayF make_rayf(const DFast& F, const Ray& ray, const V3& rcp)
{
    RayF f;
    f.o[0] = (float)ray.o.x; f.o[1] = (float)ray.o.y; f.o[2] = (float)ray.o.z;
    f.r[0] = (float)rcp.x; f.r[1] = (float)rcp.y; f.r[2] = (float)rcp.z;
    const double s3 = 3.0 * F.absmax;
    f.pad[0] = __double2float_ru(16.0 * 0x1p-24 * (fabs(ray.o.x) + s3) * fabs(rcp.x) * 1.0000002);
    f.pad[1] = __double2float_ru(16.0 * 0x1p-24 * (fabs(ray.o.y) + s3) * fabs(rcp.y) * 1.0000002);
    f.pad[2] = __double2float_ru(16.0 * 0x1p-24 * (fabs(ray.o.z) + s3) * fabs(rcp.z) * 1.0000002);
    return f;
}

// 1/x to within a few ulp (v_rcp_f64 + two Newton steps).  Only feeds the conservative culling and the certainty bands
// (2^-48 relative slack = 32 ulp); every value that reaches an output is computed with true divisions.
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

// ---- conservative fp32 pre-test of a leaf triangle ------------------------------------------------------------------------------
// About four visited triangles in five fail the reference's test (sceneManagement.cpp:316-338), and that test is ~160 fp64
// instructions.  This one (~55 fp32 instructions, 48 B instead of 104 B fetched) REJECTS a triangle only when the reference's test
// provably fails or the triangle provably cannot become the closest hit; everything else goes to the exact test unchanged, so the set
// of decisions -- and every result bit -- stays the reference's.
//
// With tvec = o - v1, pvec = d x e2, det = e1 . pvec, u = tvec . pvec, qvec = tvec x e1, v = d . qvec, tq = e2 . qvec (Moeller-
// Trumbore), the plane hit has barycentric weights beta = u/det (at v2), gamma = v/det (at v3), alpha = 1 - beta - gamma and
// parameter t = tq/det.  The reference's dir1, dir2, dir3 are 2 * area * (gamma, alpha, beta) up to one common sign, so it accepts
// exactly when the weights have one sign -- and since they sum to 1, a weight that is clearly negative sits beside one that is at
// least 1/3: the pairwise product the reference forms from those two is negative.
//
// Error budget (eps = 2^-24; O, D, V0, E1, E2 the fp32 values; S = largest scene coordinate, omax = max |o_i|; dm = max |D_i|;
// a1, a2 >= the 1-norms of the edges; T = max |tvec_i|):
//   tvec: conversions of o and v1 and one subtraction  -> |err| <= 2^-23 (omax + S) per component =: eta/2, eta = 2^-22 (omax + S)
//   pvec: two products with four roundings each        -> |err| <= 2^-22 dm a2,  |pvec| <= dm a2
//   u   : <= 3 (eta/2) dm a2 + |tvec|_1 2^-22 dm a2 + 3 eps |tvec|_1 dm a2     <= dm a2 (3 eta + 2^-19 T)
//   v   : qvec err <= a1 (eta/2 + 2^-22 T) per component                       <= dm a1 (3 eta + 2^-19 T)
//   det : <= 2^-21 dm a1 a2;   tq: <= a1 a2 (3 eta + 2^-19 T)
// The code uses 4 eta and 2^-18 T (a third more than needed: the bounds themselves are evaluated in fp32) and 2^-19 dm a1 a2 for det.
// Against the reference's own fp64 rounding (u64 = 2^-53): with rho = |d| / |n . d| its computed dir_k is within
// 100 u64 rho^2 |e| (omax + S) of the true value (t, p, three differences, a cross and a dot product, each bounded in turn), while a
// rejected weight has |true dir_k| > sqrt(3) 2^-22 rho |e_other| (omax + S).  The test therefore only rejects when
//   (i)  |det| > 2^-9 dm a1 a2           (rho < 2^11: the ray is not within ~0.03 degrees of the triangle's plane), and
//   (ii) the build found the triangle's shortest edge >= 2^-16 S and >= 2^-12 of its longest (DTriPre::a1 finite), and
//   (iii) omax <= 4 S, and S and dm within [1e-6, 1e6] (no fp32 product under- or overflows; else eta = inf: nothing is rejected),
// which leaves four to seven orders of magnitude between the two error scales, keeps the weight of at least 1/3 clearly positive
// in the reference's arithmetic too, and every product far from underflow (coordinates >= 1e-15 in magnitude by fast_path_ok).
// Distance: t* = tq/det is the true parameter of the plane hit; a triangle is skipped when t* < -margin (the reference's t_k is then
// negative: not a candidate) or t* > limit (it cannot beat the leader) -- the same margin and limit, with the same meaning, as the
// box culling of cw_step.
struct PreRay { float o[3], d[3], dm, eta4, margin; };
__device__ __forceinline__ PreRay make_pre_ray(const DFast& F, const Ray& r, const float of[3], float margin_ru)
{
    PreRay q;
    q.o[0] = of[0]; q.o[1] = of[1]; q.o[2] = of[2];
    q.d[0] = (float)r.d.x; q.d[1] = (float)r.d.y; q.d[2] = (float)r.d.z;
    q.dm = fmaxf(fmaxf(fabsf(q.d[0]), fabsf(q.d[1])), fabsf(q.d[2]));
    const float s = __double2float_ru(F.absmax);
    const float omax = fmaxf(fmaxf(fabsf(of[0]), fabsf(of[1])), fabsf(of[2]));
    // (iii), and magnitudes for which no intermediate product leaves fp32's normal range (outside them nothing is rejected)
    const bool in_range = omax <= 4.0f * s && s >= 1e-6f && s <= 1e6f && q.dm >= 1e-6f && q.dm <= 1e6f;
    q.eta4 = in_range ? 0x1p-20f * (omax + s) * 1.0001f : __builtin_inff();
    q.margin = margin_ru;
    return q;
}
__device__ __forceinline__ bool tri_pre_reject(const DTriPre* __restrict__ q, const PreRay& R, float limit_f, float* dbg = nullptr)
{
    const float4* w = reinterpret_cast<const float4*>(q);
    const float4 A = w[0], B = w[1], C = w[2];             // v0 a1 | e1 a2 | e2 -
    const float a1 = A.w, a2 = B.w;
    const float tx = R.o[0] - A.x, ty = R.o[1] - A.y, tz = R.o[2] - A.z;
    const float px = fmaf(R.d[1], C.z, -(R.d[2] * C.y)), py = fmaf(R.d[2], C.x, -(R.d[0] * C.z)), pz = fmaf(R.d[0], C.y, -(R.d[1] * C.x));
    const float det = fmaf(B.z, pz, fmaf(B.y, py, B.x * px));
    const float u = fmaf(tz, pz, fmaf(ty, py, tx * px));
    const float qx = fmaf(ty, B.z, -(tz * B.y)), qy = fmaf(tz, B.x, -(tx * B.z)), qz = fmaf(tx, B.y, -(ty * B.x));
    const float v = fmaf(R.d[2], qz, fmaf(R.d[1], qy, R.d[0] * qx));
    const float tq = fmaf(C.z, qz, fmaf(C.y, qy, C.x * qx));
    const float T = fmaxf(fmaxf(fabsf(tx), fabsf(ty)), fabsf(tz));
    const float base = fmaf(T, 0x1p-18f, R.eta4);
    const float da1 = R.dm * a1, da2 = R.dm * a2;
    const float Eu = da2 * base, Ev = da1 * base, X = da2 * a1, Etq = (a1 * a2) * base;
    const float Dt = fabsf(det);
    const bool neg = det < 0.0f;
    const float U = neg ? -u : u, V = neg ? -v : v, TQ = neg ? -tq : tq;
    const bool clear = Dt > 0x1p-9f * X;                    // (i); false as well when a1 = +inf or anything is NaN
    bool rej = U < -Eu || V < -Ev || (U + V) - Dt > (Eu + Ev) + 0x1p-19f * X;
    rej = rej || TQ + Etq < -((R.margin * Dt) * 1.002f) || TQ - Etq > (limit_f * Dt) * 1.002f;
    if (dbg) { dbg[0] = U + Eu; dbg[1] = V + Ev; dbg[2] = ((Eu + Ev) + 0x1p-19f * X) - ((U + V) - Dt); dbg[3] = TQ + Etq + (R.margin * Dt) * 1.002f;
               dbg[4] = (limit_f * Dt) * 1.002f - (TQ - Etq); dbg[5] = Dt - 0x1p-9f * X; dbg[6] = TQ / Dt; dbg[7] = Dt; }
    return clear && rej;
}

#ifndef MCPT_PARTIAL_SORT
#define MCPT_PARTIAL_SORT 0
#endif
#ifndef MCPT_CW_PACKED
#define MCPT_CW_PACKED 0          /* near and far plane of an axis as one v_pk_fma_f32: 12 packed instead of 24 scalar fmas per node, 9 instructions fewer after
                                     the moves it needs -- and 0.6 % slower on three scenes (measured twice): not kept */
#endif
struct CwHits { float key[4]; int ref[4]; };

// One step on a compressed node: which children may contain a candidate, sorted by lower bound of entry distance
// (absent / culled children get key = +inf, ref = EMPTY).
__device__ __forceinline__ CwHits cw_step_words(const uint4& w0, const uint4& w1, const uint4& w2, const uint4& w3, const RayF& f, float limit_f);
__device__ __forceinline__ CwHits cw_step(const CwNode* __restrict__ nd, const RayF& f, float limit_f)
{
    const uint4* q = reinterpret_cast<const uint4*>(nd);
    const uint4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
    return cw_step_words(w0, w1, w2, w3, f, limit_f);
}

// The top of the tree mirrored in LDS (the block copies nodes [0, n) there when it starts): nearly every ray steps on these nodes, and a
// node fetch is four 16-byte gathers per lane through the vector memory path, which the walk keeps as busy as the ALUs.
#ifndef MCPT_NODE_CACHE_N
#define MCPT_NODE_CACHE_N 96         /* nodes of it the engines hold (6 KB per block) */
#endif
struct NodeCache { const uint4* lds; int n; };
__device__ __forceinline__ void fill_node_cache(uint4* lds, const CwNode* __restrict__ nodes, int n)
{
    const uint4* g = reinterpret_cast<const uint4*>(nodes);
    for (int i = threadIdx.x; i < n * 4; i += blockDim.x) lds[i] = g[i];
    __syncthreads();
}
__device__ __forceinline__ CwHits cw_step(const CwNode* __restrict__ nodes, int cur, const NodeCache& nc, const RayF& f, float limit_f)
{
    uint4 w0, w1, w2, w3;
    if (cur < nc.n) { const uint4* q = nc.lds + cur * 4; w0 = q[0]; w1 = q[1]; w2 = q[2]; w3 = q[3]; }
    else { const uint4* q = reinterpret_cast<const uint4*>(nodes + cur); w0 = q[0]; w1 = q[1]; w2 = q[2]; w3 = q[3]; }
    return cw_step_words(w0, w1, w2, w3, f, limit_f);
}
__device__ __forceinline__ CwHits cw_step_words(const uint4& w0, const uint4& w1, const uint4& w2, const uint4& w3, const RayF& f, float limit_f)
{
    const float p[3] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z)};
    const int e[3] = {(int)(signed char)(w0.w & 255u), (int)(signed char)((w0.w >> 8) & 255u), (int)(signed char)((w0.w >> 16) & 255u)};
    const unsigned qlo[3] = {w1.x, w1.y, w1.z}, qhi[3] = {w1.w, w2.x, w2.y};
    const int child[4] = {(int)w2.z, (int)w2.w, (int)w3.x, (int)w3.y};
    float sr[3], prlo[3], prhi[3];
    unsigned nearw[3], farw[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float scale = __uint_as_float((unsigned)(e[a] + 127) << 23);
        const float prb = (p[a] - f.o[a]) * f.r[a];
        sr[a] = scale * f.r[a];
        prlo[a] = prb - f.pad[a];
        prhi[a] = prb + f.pad[a];
        const bool pos = f.r[a] >= 0.0f;
        nearw[a] = pos ? qlo[a] : qhi[a];
        farw[a] = pos ? qhi[a] : qlo[a];
    }
    CwHits h;
#if MCPT_CW_PACKED
    typedef float cw_f2 __attribute__((ext_vector_type(2)));
    cw_f2 sr2[3], pr2[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { sr2[a] = cw_f2{sr[a], sr[a]}; pr2[a] = cw_f2{prlo[a], prhi[a]}; }
#endif
#pragma unroll
    for (int c = 0; c < 4; c++) {
#if MCPT_CW_PACKED
        // near and far plane of an axis as one packed fma (v_pk_fma_f32: the same two roundings as two fmaf)
        const cw_f2 tx = __builtin_elementwise_fma(cw_f2{(float)((nearw[0] >> (8 * c)) & 255u), (float)((farw[0] >> (8 * c)) & 255u)}, sr2[0], pr2[0]);
        const cw_f2 ty = __builtin_elementwise_fma(cw_f2{(float)((nearw[1] >> (8 * c)) & 255u), (float)((farw[1] >> (8 * c)) & 255u)}, sr2[1], pr2[1]);
        const cw_f2 tz = __builtin_elementwise_fma(cw_f2{(float)((nearw[2] >> (8 * c)) & 255u), (float)((farw[2] >> (8 * c)) & 255u)}, sr2[2], pr2[2]);
        const float tnx = tx.x, tfx = tx.y, tny = ty.x, tfy = ty.y, tnz = tz.x, tfz = tz.y;
#else
        const float tnx = fmaf((float)((nearw[0] >> (8 * c)) & 255u), sr[0], prlo[0]);
        const float tny = fmaf((float)((nearw[1] >> (8 * c)) & 255u), sr[1], prlo[1]);
        const float tnz = fmaf((float)((nearw[2] >> (8 * c)) & 255u), sr[2], prlo[2]);
        const float tfx = fmaf((float)((farw[0] >> (8 * c)) & 255u), sr[0], prhi[0]);
        const float tfy = fmaf((float)((farw[1] >> (8 * c)) & 255u), sr[1], prhi[1]);
        const float tfz = fmaf((float)((farw[2] >> (8 * c)) & 255u), sr[2], prhi[2]);
#endif
        const float entry = fmaxf(fmaxf(tnx, tny), tnz);
        const float exit = fminf(fminf(tfx, tfy), tfz);
        const bool hit = child[c] != MCPT_FAST_EMPTY && exit >= 0.0f && entry <= exit && entry <= limit_f;
        h.key[c] = hit ? entry : __builtin_inff();
        h.ref[c] = hit ? child[c] : MCPT_FAST_EMPTY;
    }
#if MCPT_PARTIAL_SORT
    // only the nearest child is brought to the front (three conditional swaps); the other hits keep their slot order
#define MCPT_CSWAP(i, j)                                                                        \
    {                                                                                           \
        const bool sw = h.key[j] < h.key[i];                                                    \
        const float ka = sw ? h.key[j] : h.key[i], kb = sw ? h.key[i] : h.key[j];               \
        const int ra = sw ? h.ref[j] : h.ref[i], rb = sw ? h.ref[i] : h.ref[j];                 \
        h.key[i] = ka; h.key[j] = kb; h.ref[i] = ra; h.ref[j] = rb;                             \
    }
    MCPT_CSWAP(0, 1) MCPT_CSWAP(2, 3) MCPT_CSWAP(0, 2)
#undef MCPT_CSWAP
    return h;
#endif
    // sorting network for 4 keys
#define MCPT_CSWAP(i, j)                                                                        \
    {                                                                                           \
        const bool sw = h.key[j] < h.key[i];                                                    \
        const float ka = sw ? h.key[j] : h.key[i], kb = sw ? h.key[i] : h.key[j];               \
        const int ra = sw ? h.ref[j] : h.ref[i], rb = sw ? h.ref[i] : h.ref[j];                 \
        h.key[i] = ka; h.key[j] = kb; h.ref[i] = ra; h.ref[j] = rb;                             \
    }
    MCPT_CSWAP(0, 1) MCPT_CSWAP(2, 3) MCPT_CSWAP(0, 2) MCPT_CSWAP(1, 3) MCPT_CSWAP(1, 2)
#undef MCPT_CSWAP
    return h;
}

#ifndef MCPT_LANE_PRE_TEST
#define MCPT_LANE_PRE_TEST 0           /* the pre-test in the one-lane walk too: measured, no difference (the finishing kernel is bound by its chain of dependent steps, 15.51 vs 15.52 ms per 1/8 frame) */
#endif
// One ray per lane, start to finish (while-while over the compressed hierarchy): the same decisions, in the same order
// per candidate, as the persistent engine -- used where only a few thousand rays remain and a launch per bounce would
// cost more than the rays themselves.  stack: this lane's LDS words, stack[i * stride].
__device__ __forceinline__ bool trace_lane_fast(const DScene& S, const Ray& r, Hit& best, Work& w, int* __restrict__ stack, int stride)
{
    const DFast& F = S.fast;
    if (!fast_path_ok(F, r)) return trace_closest(S, r, best, w);
    const CwNode* __restrict__ nodes = F.cw;
    const DTri* __restrict__ tris = F.tris;
    const V3 rcp = mk(fast_rcp(r.d.x), fast_rcp(r.d.y), fast_rcp(r.d.z));
    const double rmax = fmax(fmax(fabs(rcp.x), fabs(rcp.y)), fabs(rcp.z));
    const double scale = fmax(fmax(F.absmax, fabs(r.o.x)), fmax(fabs(r.o.y), fabs(r.o.z)));
    const double margin = rmax <= 1e6 ? 1.0000001e-9 * scale * rmax : __builtin_inf();
    const RayF rf = make_rayf(F, r, rcp);
    double limit = __builtin_inf();
    float limit_f = __builtin_inff();
    bool found = false;
    best.leaf = -1; best.t = 0; best.p = mk(0, 0, 0);
    int sp = 0, cur = 0;
    for (;;) {
        while (cur >= 0) {
            w.nodes++;
            const CwHits h = cw_step(nodes + cur, rf, limit_f);
            if (h.ref[3] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[3]; sp++; }
            if (h.ref[2] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[2]; sp++; }
            if (h.ref[1] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[1]; sp++; }
            cur = h.ref[0];
            if (cur == MCPT_FAST_EMPTY && sp > 0) { sp--; cur = stack[sp * stride]; }
        }
        if (cur == MCPT_FAST_EMPTY) break;
        const int ref = -1 - cur;
        const int first = ref >> 4, count = (ref & 7) + 1;
        // the leaf's triangles through the conservative fp32 pre-test first (tri_pre_reject: their records are requested together,
        // one memory latency for the whole leaf), the reference's test for the survivors only
        unsigned int surv = 0;
#if MCPT_LANE_PRE_TEST
        if (!F.pre) surv = (1u << count) - 1u;
        else {
            const PreRay pr = make_pre_ray(F, r, rf.o, __double2float_ru(margin));
#pragma unroll
            for (int i = 0; i < 4; i++) {        // (a slot past the leaf's last is the next leaf's or the array's padding: tested, not used)
                const bool rej = tri_pre_reject(F.pre + first + i, pr, limit_f);
                if (i < count && !rej) surv |= 1u << i;
            }
            for (int i = 4; i < count; i++) if (!tri_pre_reject(F.pre + first + i, pr, limit_f)) surv |= 1u << i;
        }
#else
        surv = (1u << count) - 1u;
#endif
        w.tris += count;
        while (surv) {
            const int i = __ffs(surv) - 1;
            surv &= surv - 1;
            const DTri* tr = tris + first + i;
            V3 p;
            if (!tri_hit(tr, r, p)) continue;
            double t; int k;
            if (better_candidate(tr, r, rcp, p, found, best, t, k)) {
                found = true; best.leaf = k; best.t = t; best.p = p;
                limit = t + margin;
                limit_f = __double2float_ru(limit);
            }
        }
        if (sp > 0) { sp--; cur = stack[sp * stride]; }
        else break;
    }
    return found;
}

}  // namespace mcpt
