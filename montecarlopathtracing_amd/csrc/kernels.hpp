// Launch interface between the C-ABI layer (capi.cpp) and the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include "device_scene.hpp"

namespace mcpt {

#define MCPT_MAX_DEPTH_DEV 64      /* == MCPT_MAX_DEPTH in mcpt.h (D6) */

struct PrimaryHit {                 // closest hit of a pixel's (sample-independent) primary ray
    int32_t leaf, pad;
    double t;
    double p[3];
};

struct TraceQueue;
void launch_trace_closest(const DScene& S, bool fast, const double* d_rays, long long n, int32_t* d_face, double* d_t, double* d_p,
                          double* d_pn, DCounters* ctr, TraceQueue* queue, long long* slow_list, unsigned int slow_cap, hipStream_t st);
void launch_primary_dirs(const DCamera& cam, double* d_dirs, hipStream_t st);
void launch_primary_hits(const DScene& S, bool fast, const double* d_dirs, const int32_t* d_pixels, int n_pixels, PrimaryHit* d_hits,
                         DCounters* ctr, TraceQueue* queue, long long* slow_list, unsigned int slow_cap, hipStream_t st);
void launch_shade_samples(const DScene& S, unsigned long long seed, const double* d_dirs, const int32_t* d_pixels,
                          const PrimaryHit* d_hits, int first_slot, int n_slots, int spp, double* d_rad, DCounters* ctr, hipStream_t st);
void launch_sample_radiance(const DScene& S, unsigned long long seed, const double* d_dirs, const int32_t* d_pix, const int32_t* d_k,
                            long long n, double* d_rgb, DCounters* ctr, hipStream_t st);
void launch_fold_samples(const double* d_rad, const int32_t* d_pixels, const PrimaryHit* d_hits, int first_slot, int n_slots, int spp,
                         double* d_img, hipStream_t st);

}  // namespace mcpt
