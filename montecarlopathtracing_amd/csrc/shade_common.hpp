// Pieces of shade()/nextRay() shared by the megakernel and the wavefront logic kernel.
#pragma once
#include "dev_common.hpp"

namespace mcpt {

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

// ---- shading arithmetic -----------------------------------------------------------------------------------------------
// The hit tests (dev_common.hpp) keep IEEE divisions: their last bit decides which triangle is hit.  Shading is continuous
// in its inputs, so here a quotient is formed as a * (1/b) with 1/b from v_rcp_f64 + two Newton steps (<= 2 ulp from the
// IEEE quotient; ~7 instructions instead of ~30).  The per-sample radiance stays within the stated 1e-9 relative of the
// oracle by six orders of magnitude.
__device__ __forceinline__ double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
// The same for square roots: v_rsq_f64 and one coupled Newton step plus a residual correction (<= 1 ulp from the IEEE root, 8
// instructions instead of ~20: no scaling of denormal or huge arguments -- the arguments here are squared lengths at scene scale and
// uniforms).  The clamp turns rsq(0) = inf into a finite number so that fsqrt(0) = 0 as for the IEEE root.
__device__ __forceinline__ double fsqrt(double x)
{
    const double y = fmin(__builtin_amdgcn_rsq(x), 0x1p1000);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}
__device__ __forceinline__ double norm_s(V3 a) { return fsqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 normalized_s(V3 a) { const double inv = frcp(norm_s(a)); return mk(a.x * inv, a.y * inv, a.z * inv); }
// findGarCor (pathTracing.cpp:394-432) for shading
__device__ __forceinline__ V3 barycentric_s(V3 v1, V3 v2, V3 v3, V3 p)
{
    const V3 e1 = v3 - v2, e2 = v1 - v3, e3 = v2 - v1;
    const V3 d1 = p - v1, d2 = p - v2, d3 = p - v3;
    const V3 n = cross(e1, e2);
    const double inv = frcp(dot(n, n));
    return mk(dot(cross(e1, d3), n) * inv, dot(cross(e2, d1), n) * inv, dot(cross(e3, d2), n) * inv);
}
#define MCPT_INV_PI (1.0 / 3.1415926)
#define MCPT_INV_P_RR (1.0 / 0.6)
#define MCPT_INV_255 (1.0 / 255.0)

// x^2 and x^5 by multiplication (pow(dist,2), pow(.,2), pow(.,5) of pathTracing.cpp:222,98,99): within 2 ulp of libm's pow
__device__ __forceinline__ double sqr(double x) { return x * x; }
__device__ __forceinline__ double pow5(double x) { const double x2 = x * x; return x2 * x2 * x; }

// BRDFImportanceSampling, pathTracing.cpp:30-64
__device__ __forceinline__ V3 brdf_sample(double u_phi, double u_theta, V3 direction, int type, double Ns)
{
    const double phi = u_phi * 2 * MCPT_PI;
    // The reference forms theta = asin(sqrt(u)) or acos(u^(1/(Ns+1))) and then takes sin(theta), cos(theta).  Those are
    // sqrt(u), sqrt(1-u) resp. sqrt(1-x*x), x: evaluated directly (same values to within libm rounding, without
    // four fp64 transcendental calls per bounce).
    double sin_t, cos_t;
    if (type == RT_DIFFUSE) { sin_t = fsqrt(u_theta); cos_t = fsqrt(1.0 - u_theta); }
    else { cos_t = pow(u_theta, (double)1 / (Ns + 1)); sin_t = fsqrt(fmax(1.0 - cos_t * cos_t, 0.0)); }
    double sin_p, cos_p;
    sincos(phi, &sin_p, &cos_p);
    const V3 sample = mk(sin_t * cos_p, cos_t, sin_t * sin_p);
    V3 front;
    if (fabs(direction.x) > fabs(direction.y)) front = normalized_s(mk(direction.z, 0, -direction.x));
    else front = normalized_s(mk(0, -direction.z, direction.y));
    const V3 right = cross(direction, front);
    return normalized_s((right * sample.x + direction * sample.y) + front * sample.z);
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

}  // namespace mcpt
