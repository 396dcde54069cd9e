// Wavefront formulation of generateImg's sample loop (pathTracing.cpp:303-320) for one chunk of camera samples.
//
// All paths of a chunk advance in lockstep, one path vertex per iteration:
//     logic(d):  resolve the rays of vertex d-1 (visibility -> direct light, bounce hit -> next vertex),
//                shade vertex d (emitter test, texture, light sampling, Russian roulette, BSDF sample),
//                emit its shadow rays and its bounce ray, compact the surviving paths (wave ballot + prefix).
//     trace(d):  closest hit of every emitted ray (the dominant kernel; fast walk of trace_fast.hpp).
// Path state lives in HBM as component-major SoA indexed by the compacted position, so every load/store of a wave
// is one contiguous 512-byte run.  Two copies (A/B) because compaction moves paths between iterations.
#pragma once
#include <hip/hip_runtime_api.h>

#include "device_scene.hpp"
#include "kernels.hpp"

namespace mcpt {

struct WfState {            // one side of the double buffer; every array has `cap` entries per component
    int32_t* id;            // chunk-local sample id = (slot - first_slot) * spp + k
    double* T;              // [3][cap] throughput at the vertex that was just shaded (one light: after its bounce, and
                            //          c = T * c); several lights: not stored by the first pass (T = 1)
    double* L;              // [3][cap] radiance gathered before that vertex; not stored by the first pass (L = 0)
    // what shade(d) leaves for resolve(d):
    double* c;              // [nl][3][cap] direct-light contribution of light l if visible
    int32_t* expect;        // [nl][cap] material the shadow ray must hit to be visible; -2 = no shadow ray
    double* w;              // [3][cap] weight of the bounce (kd / ks / 1); with one light it is folded into T and c instead
    double* bdir;           // [3][cap] direction of the bounce ray
    int32_t* btype;         // [cap] ray_type of the bounce ray (| MCPT_BT_NO_OFFSET: it starts at the vertex itself), -1 = none
    // what trace(d) adds:
    int32_t* hit_mat;       // [nl][cap] material of the shadow ray's closest hit, -1 = miss
    int32_t* hit_leaf;      // [cap]     bounce ray: leaf (| MCPT_HIT_EMITTER when its material is a light) or -1.  The hit point is not stored: the next logic pass forms it again from the
                            //           ray it rebuilds and the leaf's plane -- the first two lines of the reference's triangle test on the
                            //           same operands, hence the same bits -- which takes 24 bytes per bounce ray off the trace kernel's
                            //           scattered stores and ~45 instructions per finished ray off the instruction-bound kernel
    double* p;              // [3][cap]  the vertex that was shaded (origin of its rays before the 0.01 offset); not stored by the first
                            //           pass (the pixel's primary hit)
};

// Rays of vertex d, written by logic(d), consumed by trace(d); slot (l, j), l = nl for the bounce ray.  All rays of a vertex
// leave from p (+ 0.01 d, pathTracing.cpp:196,128; refraction and total reflection start at p itself, :97,:110), so p is
// stored once (WfState::p) and the origin is rebuilt where the ray is fetched; the bounce direction is WfState::bdir.
struct WfRays {
    double* d;              // [nl][3][cap] shadow-ray directions
};

// What all samples of a pixel share at their first vertex (the primary ray has no jitter): computed once per hit pixel by
// k_primary_surface, read by the first logic pass as broadcast 16-byte loads instead of ~40 scattered ones per sample.
struct alignas(16) PrimarySurface {
    double p[3], dir[3];        // hit point, direction back to the eye
    double pn[3], kd[3];        // interpolated normal and diffuse colour there (unset on an emitter)
    int32_t leaf, material, pixel, slot;
    int32_t alive_index, pad[3];    // rank of the pixel among the shaded (non-emitter) pixels of its group of 64 hit slots, -1 on an emitter:
                                    // the first logic pass puts sample k at path position (alive_base[group] + alive_index) * spp + k -- no compaction
};

// A bounce ray's answer carries whether the surface it reached is a light: the trace kernels have the triangle's material at hand, and
// the next logic pass then knows which paths go on from the words of the path alone (no triangle and no material fetched before its
// block-wide prefix -- two dependent round trips per resolve round).
#define MCPT_HIT_EMITTER 0x40000000
#define MCPT_HIT_LEAF_MASK 0x3fffffff

struct TraceQueue {                 // persistent trace kernels; device words, zeroed before each launch
    unsigned long long head;        // next ticket (trace_persistent.hpp maps tickets to ray slots)
    unsigned int slow_count;        // rays deferred to the reference-shaped walk (may exceed the list capacity)
    unsigned int redo_all;          // set when a ray the fast walk could not decide found the list full
    unsigned int pad[12];
};

struct WfCounts {           // one per iteration, on the device: slot 0 = hit pixels of the chunk, slot d+1 = paths alive after
    unsigned int n_next;    // logic(d).  Kernels read their input count from the previous slot, so the host need not know it.
    unsigned int pad[15];
};
#define MCPT_WF_COUNT_SLOTS 72

struct WfArgs {
    WfState in, out;
    WfRays rays;
    long long cap;
    int nl, spp, depth;
    unsigned long long seed;
    const int32_t* pixels;      // slot -> pixel index (NULL: identity)
    const int32_t* hit_slots;   // first pass: compacted list of slots whose primary ray hit something
    const PrimarySurface* surf; // first pass: one record per entry of hit_slots
    const unsigned int* alive_base;   // first pass: shaded pixels before each group of 64 entries of hit_slots (k_alive_scan)
    const PrimaryHit* hits;     // first pass: primary hit per slot
    const double* dirs;         // primary directions per pixel
    int first_slot;
    double* rad;                // [chunk samples][3] finished radiance
    WfCounts* counts;           // this iteration's output slot
    const WfCounts* counts_in;  // the slot holding this iteration's input count
    unsigned int count_mul;     // input paths = counts_in->n_next * count_mul (spp for the first pass, 1 afterwards)
    DCounters* ctr;
    const DTri* tris;           // S.tris (material of a shadow ray's hit)
    const DMaterial* materials; // S.materials (is the surface a bounce ray reached an emitter: MCPT_HIT_EMITTER)
    // Hand-over to the finishing kernel, decided on the device: when logic(d) leaves at most this many paths, k_wf_finish
    // (launched after every logic pass) runs them to their end and trace(d), logic(d+1), ... find nothing to do.  0 = never.
    unsigned int finish_below;
    TraceQueue* queue;          // the frame slot's queue words: k_wf_logic clears them for the trace launch that follows it (a fill dispatch per
                                // iteration, ~13 us with its gap, is a percent of a rank's share of the frame)
};

size_t wf_bytes_per_path(int nl);
// carve the workspace; returns false if it does not fit
bool wf_carve(void* base, size_t bytes, long long cap, int nl, WfState& a, WfState& b, WfRays& r);

// n_upper: host-side upper bound of the input count (sizes the grid only)
void launch_wf_logic(const DScene& S, const WfArgs& a, long long n_upper, bool first, hipStream_t st, const LaunchCfg& cfg);
void launch_wf_trace(const DScene& S, const WfArgs& a, long long n_upper, bool fast, TraceQueue* queue, long long* slow_list,
                     unsigned int slow_cap, hipStream_t st, const LaunchCfg& cfg);
// path_area: finish_pool_bytes() of device memory for the pool form of the finishing pass (or null: the one-lane-per-path form);
// slow_list / slow_cap: the frame slot's deferred-ray list, behind which the pool engine's stack spill area lies
void launch_wf_finish(const DScene& S, const WfArgs& a, long long n_upper, hipStream_t st, const LaunchCfg& cfg, char* path_area, long long* slow_list,
                      unsigned int slow_cap);
size_t finish_pool_bytes(int cus, int nl);
int persistent_grid(const void* kernel, int cus);   // blocks of 256 threads of `kernel` resident on the current device
void init_launch_cfg_logic(LaunchCfg& cfg, unsigned forced_grid);   // wavefront_logic.hip: logic_first / logic_rest / finish_grid of cfg
long long persistent_chunk(long long total, int grid_blocks);

void launch_primary_surface(const DScene& S, const WfArgs& a, PrimarySurface* surf, unsigned int* alive_count, unsigned int* alive_total, int n_slots_upper,
                            hipStream_t st);
void launch_hit_slots(const PrimaryHit* hits, int first_slot, int n_slots, int32_t* hit_slots, unsigned int* count, hipStream_t st);
void launch_zero_rad(double* rad, long long n_doubles, hipStream_t st);

}  // namespace mcpt
