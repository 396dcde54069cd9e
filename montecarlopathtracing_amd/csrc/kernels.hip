// Hand-written HIP kernels for gfx950 (MI355X): the reference's per-pixel integrator loop
// generateImg -> ray_intersect/bvh_intersect -> shade -> nextRay (MTPC/pathTracing.cpp), in fp64 with the
// reference's operation order.  Built with -ffp-contract=off: a fused multiply-add would change the last
// bit of the hit tests and with it the closest-hit triangle.
//
// No MFMA: this is per-lane tree walking and branching, not a contraction.  wave64 throughout.
#include <hip/hip_runtime.h>

#include "device_scene.hpp"
#include "kernels.hpp"

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
// Philox4x32-10, counter (pixel, sample, depth<<16 | slot>>1, 'MCPT'), key = seed.  Even slots use words 0,1
// of the block, odd slots words 2,3; 53 bits -> [0,1).
struct Philox { uint32_t v[4]; };
__device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox p; p.v[0] = c0; p.v[1] = c1; p.v[2] = c2; p.v[3] = c3;
    return p;
}
__device__ __forceinline__ double bits_to_unit(uint32_t h, uint32_t l)
{
    const unsigned long long bits = ((static_cast<unsigned long long>(h) << 32) | l) >> 11;
    return static_cast<double>(bits) * (1.0 / 9007199254740992.0);
}
struct RngKey { uint32_t k0, k1, pixel, sample; };
__device__ __forceinline__ double uniform(const RngKey& k, uint32_t depth, uint32_t slot)
{
    const Philox p = philox4x32_10(k.pixel, k.sample, (depth << 16) | (slot >> 1), 0x4D435054u, k.k0, k.k1);
    return (slot & 1u) ? bits_to_unit(p.v[2], p.v[3]) : bits_to_unit(p.v[0], p.v[1]);
}
// two consecutive slots (2b, 2b+1) from one block
__device__ __forceinline__ void uniform2(const RngKey& k, uint32_t depth, uint32_t block, double& u0, double& u1)
{
    const Philox p = philox4x32_10(k.pixel, k.sample, (depth << 16) | block, 0x4D435054u, k.k0, k.k1);
    u0 = bits_to_unit(p.v[0], p.v[1]);
    u1 = bits_to_unit(p.v[2], p.v[3]);
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

struct Hit { int leaf; double t; V3 p; };
struct Work { uint32_t nodes, tris; };

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
struct LaneStats { uint32_t nodes = 0, tris = 0, shadow = 0, bounce = 0, primary = 0, shades = 0, samples = 0, depth = 0; };
__device__ __forceinline__ void flush_stats(DCounters* c, const LaneStats& s)
{
    if (!c) return;
    const unsigned long long n = wave_sum(s.nodes), t = wave_sum(s.tris), sh = wave_sum(s.shadow), bo = wave_sum(s.bounce),
                             pr = wave_sum(s.primary), sc = wave_sum(s.shades), sa = wave_sum(s.samples);
    const unsigned int md = wave_max(s.depth);
    if ((threadIdx.x & 63) == 0) {
        if (n) atomicAdd(&c->node_visits, n);
        if (t) atomicAdd(&c->tri_tests, t);
        if (sh) atomicAdd(&c->rays_shadow, sh);
        if (bo) atomicAdd(&c->rays_bounce, bo);
        if (pr) atomicAdd(&c->rays_primary, pr);
        if (sc) atomicAdd(&c->shade_calls, sc);
        if (sa) atomicAdd(&c->samples, sa);
        if (md) atomicMax(&c->max_depth, (unsigned long long)md);
    }
}

// ------------------------------------------------------------------------------------------------ kernels
// mcpt_trace_closest: one lane per ray.
__global__ void __launch_bounds__(256) k_trace_closest(DScene S, const double* __restrict__ rays, long long n,
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
__global__ void __launch_bounds__(256) k_primary_hits(DScene S, const double* __restrict__ dirs, const int32_t* __restrict__ pixels,
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

// ---- shading pieces ------------------------------------------------------------------------------------------
enum { RT_DIFFUSE = 0, RT_SPECULAR = 1, RT_TRANSMISSION = 2 };   // sceneManagement.h:203-205

// Refract, pathTracing.cpp:13-27 (cosi and cost2 are floats in the reference)
__device__ __forceinline__ bool refract_dir(V3 i, V3 n, double eta, V3& out)
{
    const float cosi = (float)dot(i, n);
    const float cost2 = (float)(1.0f - eta * eta * (1.0f - cosi * cosi));
    if (cost2 >= 0.0f) {
        out = i * eta - n * (eta * cosi + sqrtf(cost2));
        return true;
    }
    return false;
}

// BRDFImportanceSampling, pathTracing.cpp:30-64
__device__ __forceinline__ V3 brdf_sample(double u_phi, double u_theta, V3 direction, int type, double Ns)
{
    const double phi = u_phi * 2 * MCPT_PI;
    double theta;
    if (type == RT_DIFFUSE) theta = asin(sqrt(u_theta));
    else theta = acos(pow(u_theta, (double)1 / (Ns + 1)));
    const V3 sample = mk(sin(theta) * cos(phi), cos(theta), sin(theta) * sin(phi));
    V3 front;
    if (fabs(direction.x) > fabs(direction.y)) front = normalized(mk(direction.z, 0, -direction.x));
    else front = normalized(mk(0, -direction.z, direction.y));
    const V3 right = cross(direction, front);
    return normalized((right * sample.x + direction * sample.y) + front * sample.z);
}

// first j with rnd < cdf[j] (pathTracing.cpp:189-190), or -1
__device__ __forceinline__ int pick_light_triangle(const double* __restrict__ cdf, int n, bool sorted, double rnd)
{
    if (sorted) {
        int lo = 0, hi = n;                       // smallest j with rnd < cdf[j]
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rnd < cdf[mid]) hi = mid; else lo = mid + 1; }
        return lo < n ? lo : -1;
    }
    for (int j = 0; j < n; j++) if (rnd < cdf[j]) return j;
    return -1;
}

// shade() (pathTracing.cpp:137-266) with the recursion unrolled into a loop: the recursion is a chain
// (one bounce per vertex), so L = sum_d T_d * Ldir_d with T_{d+1} = T_d * w_d / 0.6.
__device__ void shade_path(const DScene& S, const RngKey& key, V3 view_dir /* ray.direction of the primary ray */,
                           Hit hit, double out[3], LaneStats& ls)
{
    const uint32_t nl = (uint32_t)S.num_lights;
    V3 L = mk(0, 0, 0), T = mk(1, 1, 1);
    V3 dir = neg(view_dir);                 // "dir": from the hit toward where the path came from
    int in_type = RT_TRANSMISSION;          // depth 0: an emitter is returned as is
    Work w = {0, 0};
    for (uint32_t depth = 0;; depth++) {
        ls.shades++;
        if (depth > ls.depth) ls.depth = depth;
        const DTri* tr = S.tris + hit.leaf;
        const DMaterial* m = S.materials + tr->material;
        if (m->light >= 0) {                                                    // :141-144
            const V3 rad = ld3(S.lights[m->light].radiance);
            if (depth == 0) L = rad;
            else if (in_type != RT_DIFFUSE) L = L + mk(T.x * rad.x, T.y * rad.y, T.z * rad.z);   // :247-261
            break;
        }
        const V3 tv1 = ld3(tr->v1), tv2 = ld3(tr->v2), tv3 = ld3(tr->v3);
        const DTriShade* sh = S.shade + hit.leaf;
        const V3 g = barycentric(tv1, tv2, tv3, hit.p);
        const V3 pn = (ld3(sh->vn1) * g.x + ld3(sh->vn2) * g.y) + ld3(sh->vn3) * g.z;
        V3 kd;
        if (m->has_map) {                                                       // :147-160 (Q9)
            const double row = sh->vt1[0] * g.x + sh->vt2[0] * g.y + sh->vt3[0] * g.z;
            const double col = sh->vt1[1] * g.x + sh->vt2[1] * g.y + sh->vt3[1] * g.z;
            const double irow = row - floor(row), icol = col - floor(col);
            int rr = (int)(irow * m->map_h), cc = (int)(icol * m->map_w);
            rr = rr < 0 ? 0 : (rr > m->map_h - 1 ? m->map_h - 1 : rr);          // D7
            cc = cc < 0 ? 0 : (cc > m->map_w - 1 ? m->map_w - 1 : cc);
            const uint8_t* px = S.texels + m->tex_offset + ((size_t)rr * m->map_w + cc) * 3;
            kd = mk((double)px[2] / 255, (double)px[1] / 255, (double)px[0] / 255);
        } else kd = ld3(m->kd);

        // direct illumination, :166-232
        V3 L_dir = mk(0, 0, 0);
        int sample_mat = -1;
        for (uint32_t i = 0; i < nl; i++) {
            const DLight* lt = S.lights + i;
            V3 xl = mk(0, 0, 0), vn = mk(0, 0, 0);
            double u0, u1, u2, u3;
            uniform2(key, depth, 2u * i, u0, u1);
            const double rnd = u0 * S.area0;                                    // frozen static u1 range (Q1)
            const int j = pick_light_triangle(S.light_cdf + lt->first, lt->ntri, lt->cdf_sorted != 0, rnd);
            if (j >= 0) {
                uniform2(key, depth, 2u * i + 1u, u2, u3);
                const DLightTri* q = S.light_tris + lt->first + j;
                sample_mat = lt->material;
                const double rnd1 = u1, rnd2 = u2, rnd3 = u3;
                const double p1 = rnd1 / (rnd1 + rnd2 + rnd3), p2 = rnd2 / (rnd1 + rnd2 + rnd3), p3 = rnd3 / (rnd1 + rnd2 + rnd3);
                xl = (ld3(q->v1) * p1 + ld3(q->v2) * p2) + ld3(q->v3) * p3;
                vn = (ld3(q->vn1) * p1 + ld3(q->vn2) * p2) + ld3(q->vn3) * p3;
            }
            const V3 direction = normalized(xl - hit.p);
            double visibility = 1;
            Ray rl; rl.o = hit.p + direction * 0.01; rl.d = direction;
            Hit inter;
            const bool got = trace_closest(S, rl, inter, w);
            ls.shadow++;
            const int inter_mat = got ? S.tris[inter.leaf].material : -1;
            if (inter_mat != sample_mat) visibility = 0;                        // :213
            if (dot(direction, pn) > 0) {
                const double pdf_light = (double)1 / lt->total_area;
                const double cos_theta = fabs(dot(direction, vn) / norm(direction) / norm(vn));
                const double cos_theta_hat = fabs(dot(direction, pn) / norm(direction) / norm(pn));
                const double dd = norm(xl - hit.p);
                const double dist = (1.0 < dd) ? dd : 1.0;                      // std::max(1.0, distance)
                const V3 intensity = ((((ld3(lt->radiance) * cos_theta) * cos_theta_hat) / pow(dist, 2.0)) / pdf_light) * visibility;
                const double kd_dots = dot(direction, pn);
                if (kd_dots > 0) {
                    L_dir.x += kd.x * intensity.x * kd_dots / MCPT_PI;
                    L_dir.y += kd.y * intensity.y * kd_dots / MCPT_PI;
                    L_dir.z += kd.z * intensity.z * kd_dots / MCPT_PI;
                }
            }
        }
        L = L + mk(T.x * L_dir.x, T.y * L_dir.y, T.z * L_dir.z);

        // indirect illumination, :234-263
        if (depth + 1 >= MCPT_MAX_DEPTH_DEV) break;                             // D6
        double u_rr, u_fresnel;
        uniform2(key, depth, 2u * nl, u_rr, u_fresnel);                         // slots 4nl (RR), 4nl+1 (FRESNEL)
        if (!(u_rr < MCPT_P_RR)) break;                                         // russian_Roulette :3-11
        // nextRay, :66-134
        Ray nr; int type = -1;
        const V3 ks = ld3(m->ks);
        if (m->Ni > 1) {
            double n1, n2;
            const double cos_in = dot(neg(dir), pn);
            V3 normal;
            if (cos_in > 0) { normal = neg(pn); n1 = m->Ni; n2 = 1.0; }
            else { normal = pn; n1 = 1.0; n2 = m->Ni; }
            const double rf0 = pow((n1 - n2) / (n1 + n2), 2.0);
            const double fresnel = rf0 + (1.0f - rf0) * pow(1.0f - fabs(cos_in), 5.0);
            if (fresnel < u_fresnel) {
                V3 direction;
                if (refract_dir(neg(dir), normal, n1 / n2, direction)) { nr.o = hit.p; nr.d = direction; type = RT_TRANSMISSION; }
                else {
                    const V3 incoming = neg(dir);
                    nr.o = hit.p; nr.d = incoming - (normal * dot(incoming, normal)) * 2; type = RT_SPECULAR;
                }
            }
        }
        if (type < 0) {
            double u_lobe, u_phi, u_theta, unused;
            uniform2(key, depth, 2u * nl + 1u, u_lobe, u_phi);                  // slots 4nl+2 (LOBE), 4nl+3 (PHI)
            uniform2(key, depth, 2u * nl + 2u, u_theta, unused);                // slot 4nl+4 (THETA)
            const double kd_norm = norm(kd), ks_norm = norm(ks);
            V3 direction;
            if (ks_norm != 0 && kd_norm / ks_norm < u_lobe) {
                const V3 incoming = neg(dir);
                const V3 reflect = incoming - (pn * dot(incoming, pn)) * 2;
                direction = brdf_sample(u_phi, u_theta, reflect, RT_SPECULAR, m->Ns);
                type = RT_SPECULAR;
            } else {
                direction = brdf_sample(u_phi, u_theta, pn, RT_DIFFUSE, m->Ns);
                type = RT_DIFFUSE;
            }
            nr.o = hit.p + direction * 0.01; nr.d = direction;
        }
        Hit next;
        ls.bounce++;
        if (!trace_closest(S, nr, next, w)) break;
        const V3 wgt = type == RT_DIFFUSE ? kd : (type == RT_SPECULAR ? ks : mk(1, 1, 1));
        T = mk(T.x * wgt.x / MCPT_P_RR, T.y * wgt.y / MCPT_P_RR, T.z * wgt.z / MCPT_P_RR);
        hit = next; dir = neg(nr.d); in_type = type;
    }
    ls.nodes += w.nodes; ls.tris += w.tris;
    out[0] = L.x; out[1] = L.y; out[2] = L.z;
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
__global__ void k_fold_samples(const double* __restrict__ rad, const int32_t* __restrict__ pixels, int first_slot, int n_slots,
                               int spp, double* __restrict__ img)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)n_slots * 3) return;
    const int s = (int)(gid / 3), c = (int)(gid % 3);
    const double* src = rad + (size_t)s * spp * 3 + c;
    float acc = 0.0f;
    for (int k = 0; k < spp; k++) acc = (float)((double)acc + src[(size_t)k * 3] / spp);
    const int slot = first_slot + s;
    const int pix = pixels ? pixels[slot] : slot;
    img[(size_t)pix * 3 + c] = (double)acc;
}

// ------------------------------------------------------------------------------------------------ launchers
static inline unsigned blocks_for(long long n, int block) { return (unsigned)((n + block - 1) / block); }

void launch_trace_closest(const DScene& S, const double* d_rays, long long n, int32_t* d_face, double* d_t, double* d_p,
                          double* d_pn, DCounters* ctr, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_trace_closest, dim3(blocks_for(n, 256)), dim3(256), 0, st, S, d_rays, n, d_face, d_t, d_p, d_pn, ctr);
}
void launch_primary_dirs(const DCamera& cam, double* d_dirs, hipStream_t st)
{
    hipLaunchKernelGGL(k_primary_dirs, dim3(blocks_for(cam.height, 64)), dim3(64), 0, st, cam, d_dirs);
}
void launch_primary_hits(const DScene& S, const double* d_dirs, const int32_t* d_pixels, int n_pixels, PrimaryHit* d_hits,
                         DCounters* ctr, hipStream_t st)
{
    if (n_pixels <= 0) return;
    hipLaunchKernelGGL(k_primary_hits, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, st, S, d_dirs, d_pixels, n_pixels, d_hits, ctr);
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
void launch_fold_samples(const double* d_rad, const int32_t* d_pixels, int first_slot, int n_slots, int spp, double* d_img, hipStream_t st)
{
    if (n_slots <= 0) return;
    hipLaunchKernelGGL(k_fold_samples, dim3(blocks_for((long long)n_slots * 3, 256)), dim3(256), 0, st, d_rad, d_pixels, first_slot, n_slots, spp, d_img);
}

}  // namespace mcpt
