// Device build of Morton keys, Morton ordering, leaf records and the implicit tree's boxes (build_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <vector>

#include "../../include/mcpt.h"
#include "device_scene.hpp"

namespace mcpt {

struct BuildInputs {            // faces in .obj order, device pointers
    const double* v9;           // [t][9]  v1 v2 v3
    const double* vn9;          // [t][9]
    const double* vt6;          // [t][6]
    const double* nrm3;         // [t][3]  Face::norm
    const int32_t* material;    // [t]
    int t;
    float morton_lo[3], morton_span[3];     // key domain (Scene::morton_lo / morton_span)
};

// fills nodes[Nr] (compact level order), tris[t], shade[t] (leaf order) and d_order[t] (leaf -> .obj face)
hipError_t device_build_reference(const BuildInputs& in, const mcpt_bvh_info& bi, DNode* nodes, DTri* tris, DTriShade* shade,
                                  int32_t* d_order, hipStream_t st);
// The lower part of the fast hierarchy on the device (MCPT_BUILD_DEVICE_FAST): the triangle records sorted by a 63-bit Morton
// code on the scene's bounds [lo, hi] and a complete 4-ary tree over leaves of per_leaf consecutive ones, at most max_levels levels
// high, written as the compressed nodes the walk kernels read.  With t <= 4^(max_levels+1) that is the whole tree (n_top = 1);
// otherwise it is a forest of n_top clusters (nodes 0 .. n_top-1 are their roots, top_boxes their exact boxes, lo[3] hi[3] each)
// for the host to put a SAH tree over (accel_build.cpp: build_fast_upper).  Only ever used to cull (trace_fast.hpp), so its
// shape cannot change a result.  *cw and *fast_tris are hipMalloc'ed here; the walk needs 3 stack entries per level.
hipError_t device_build_fast(const DTri* leaf_tris, int t, const double lo[3], const double hi[3], int per_leaf, int max_levels, CwNode** cw, DTri** fast_tris,
                             int* n_nodes, int* levels, int* n_top, std::vector<double>* top_boxes, double* absmax, hipStream_t st);
// MCPT_BUILD_DEVICE_SAH: the lower part grown on the device by parallel locally-ordered clustering into subtrees of at most max_cluster
// triangles and max_height binary levels, each collapsed into compressed 4-wide nodes (leaves of up to max_leaf triangles where the
// surface-area heuristic, evaluated bottom up with cost_leaf + cost_tri per triangle for a leaf and 1 for a node, prefers a leaf to a split).  Cluster c's
// nodes are a run of *cw starting at top_roots[c] (its root; < 0: the cluster is one leaf and this is its reference); top_boxes as above;
// hipErrorNotSupported: the clusters leave the tree above them too few of the walk's stack entries (take another builder); lower_need = stack entries a walk below a cluster
// root can hold.  rounds = clustering rounds it took.
hipError_t device_build_ploc(const DTri* leaf_tris, int t, const double lo[3], const double hi[3], int max_cluster, int max_height, int radius, int max_leaf,
                             double area_fraction /* of the scene box's area a cluster's box may have; 0: no limit */, double cost_tri, double cost_leaf,
                             int collapse_budget /* stack entries below a cluster root; 0: chosen from the number of clusters */, CwNode** cw, DTri** fast_tris, int* n_nodes, int* n_top, std::vector<double>* top_boxes, std::vector<int32_t>* top_roots,
                             int* lower_need, double* absmax, int* rounds, hipStream_t st);
hipError_t device_offset_children(CwNode* nodes, int n, int off, hipStream_t st);     // child >= 0 -> child + off
hipError_t device_gather_tris(const DTri* tris, const int32_t* d_slots, int n, DTri* out, hipStream_t st);
// the pre-test's fp32 record of every slot of the fast triangle array (absmax = largest |coordinate| of the scene)
hipError_t device_build_pre(const DTri* fast_tris, int n, double absmax, DTriPre* out, hipStream_t st);

}  // namespace mcpt
