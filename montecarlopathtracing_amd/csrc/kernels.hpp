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

// Resident grid sizes and tuning knobs of the persistent kernels, per GPU: filled once by init_launch_cfg() while that device is
// current (mcpt_device_create), carried by the mcpt_device -- nothing about a launch is process-wide, so one process can drive
// several GPUs from several threads.
struct LaunchCfg {
    int cus = 0;
    unsigned logic_first = 0, logic_rest = 0;       // k_wf_logic<true> / <false>
    int trace_grid = 0, trace_grid_short = 0, finish_grid = 0;   // k_wf_trace (deep / short stack), k_wf_finish
    int array_grid = 0, primary_grid = 0;           // k_trace_persistent<ArrayRaySource> / <PrimaryRaySource>, deep stack
    int array_grid_short = 0, primary_grid_short = 0;   // ... short stack
    long long trace_block_rays = 2048;              // MCPT_TRACE_BLOCK_RAYS: a block of k_wf_trace is started per this many rays
    int min_chunk = 256, max_chunk = 2048;          // MCPT_TRACE_MIN_CHUNK / MAX_CHUNK: ray slots per queue claim
    int trace_pool = 0;                             // the pool engine (rays resident in LDS: k_wf_trace_pool, k_trace_pool) instead of the voting engine
    int finish_pool = 0;                            // the finishing pass in its pool form (k_wf_finish_pool: paths resident with their rays)
};
bool pool_engine_available();                      // wavefront.hip: the current device can hold a workgroup of the pool engine
bool pool_engine_available_closest();              // kernels.hip: ... of its closest-hit forms
size_t pool_spill_bytes(int cus);                  // wavefront.hip: bytes the pool engine wants behind a launch's deferred-ray list (stack entries beyond its LDS part)
void init_launch_cfg(LaunchCfg& cfg, unsigned forced_logic_grid, long long trace_block_rays, int min_chunk, int max_chunk);   // wavefront.hip (calls init_launch_cfg_closest of kernels.hip)
void init_launch_cfg_closest(LaunchCfg& cfg);
void launch_trace_closest(const DScene& S, bool fast, const double* d_rays, long long n, int32_t* d_face, double* d_t, double* d_p,
                          double* d_pn, DCounters* ctr, TraceQueue* queue, long long* slow_list, unsigned int slow_cap, hipStream_t st,
                          const LaunchCfg& cfg);
void launch_pack_pixels(const double* d_frame, const int32_t* d_pixels, long long n_pixels, double* d_out, hipStream_t st);
void launch_unpack_pixels(const double* d_in, const int32_t* d_pixels, long long n_pixels, double* d_frame, hipStream_t st);
void launch_primary_dirs(const DCamera& cam, double* d_dirs, hipStream_t st);
void launch_primary_hits(const DScene& S, bool fast, const double* d_dirs, const int32_t* d_pixels, int n_pixels, PrimaryHit* d_hits,
                         DCounters* ctr, TraceQueue* queue, long long* slow_list, unsigned int slow_cap, hipStream_t st, const LaunchCfg& cfg);
void launch_shade_samples(const DScene& S, unsigned long long seed, const double* d_dirs, const int32_t* d_pixels,
                          const PrimaryHit* d_hits, int first_slot, int n_slots, int spp, double* d_rad, DCounters* ctr, hipStream_t st);
void launch_sample_radiance(const DScene& S, unsigned long long seed, const double* d_dirs, const int32_t* d_pix, const int32_t* d_k,
                            long long n, double* d_rgb, DCounters* ctr, hipStream_t st);
void launch_fold_samples(const double* d_rad, const int32_t* d_pixels, const PrimaryHit* d_hits, int first_slot, int n_slots, int spp,
                         double* d_img, hipStream_t st);

}  // namespace mcpt
