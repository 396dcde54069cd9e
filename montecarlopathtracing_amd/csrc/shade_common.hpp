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

}  // namespace mcpt
