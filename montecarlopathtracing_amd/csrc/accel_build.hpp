// Traversal structure of the fast closest-hit kernel (see accel_build.cpp for why it is result-identical).
#pragma once
#include <cstdint>
#include <vector>

#include "device_scene.hpp"
#include "scene.hpp"

namespace mcpt {

#ifndef MCPT_FAST_STACK
#define MCPT_FAST_STACK 36
#endif
constexpr int kFastMaxDepth = MCPT_FAST_STACK;          // inner levels; bounds the per-lane LDS stack of the deep-stack kernels
// The trace engine exists in two shapes (wavefront.hip): a 27-entry stack leaves LDS and registers for 4 waves per SIMD, the 36-entry
// one for 3.  Hierarchies are built for the deep stack; the short-stack engine hands a ray that would overflow to the one-lane walk.
constexpr int kFastShortStack = 27;
constexpr int kFastMaxLeaf = 4;            // most triangles a leaf may hold (3 bits of the reference; bit 3 is a runtime flag)
constexpr int kFastDefaultLeaf = 4;        // default leaf size (measured: 4 beats 1 and 2 on MI355X; inner steps cost more than leaf boxes)
constexpr int32_t kFastEmpty = INT32_MIN;  // child reference of an absent child

struct FastBvh {
    std::vector<CwNode> cw;                // compressed 4-wide collapse of `nodes` (what the kernels walk)
    int cw_stack_need = 0;                 // worst-case traversal stack entries
    int stack_limit = kFastMaxDepth;       // what it was built to stay below
    std::vector<FastNode> nodes;
    std::vector<int32_t> leaf_tris;        // reference leaf index k of every slot of the leaf triangle list
    double scene_absmax = 0;               // largest |coordinate| of any leaf box
    int max_depth = 0;
};

// order[k] = .obj face held by reference leaf k
void build_fast_bvh(const std::vector<FaceRec>& faces, const int32_t* order, int t, FastBvh& out, int stack_limit = kFastMaxDepth);
// upper part over n GPU-built clusters given by their boxes (lo[3], hi[3]); see accel_build.cpp
void build_fast_upper(const double* boxes6, int n, int lower_need, FastBvh& out, int stack_limit = kFastMaxDepth);

}  // namespace mcpt
