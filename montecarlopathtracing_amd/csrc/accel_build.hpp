// Traversal structure of the fast closest-hit kernel (see accel_build.cpp for why it is result-identical).
#pragma once
#include <cstdint>
#include <memory>
#include <new>
#include <thread>
#include <utility>
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
constexpr int kFastTopNodes = 256;        // nodes the builder puts first, top of the tree breadth first (the engines keep a prefix of them in LDS)
constexpr int kFastMaxLeaf = 8;            // most triangles a leaf may hold (3 bits of the reference: count - 1)
constexpr int kFastDefaultLeaf = 4;        // default leaf size (measured: 4 beats 1 and 2 on MI355X; inner steps cost more than leaf boxes)
constexpr int32_t kFastEmpty = INT32_MIN;  // child reference of an absent child

// fn(begin, end) over contiguous pieces of [0, n) on up to 16 threads (the caller's included); small n: one call
template <class F>
inline void parallel_pieces(long long n, F fn, long long min_per_thread = 1 << 16)
{
    unsigned hw = std::thread::hardware_concurrency();
    long long parts = hw ? (hw < 16u ? hw : 16u) : 1;
    if (parts > n / min_per_thread) parts = n / min_per_thread;
    if (parts <= 1) { if (n > 0) fn(0ll, n); return; }
    std::vector<std::thread> pool;
    for (long long p = 1; p < parts; p++) pool.emplace_back([=]() { fn(n * p / parts, n * (p + 1) / parts); });
    fn(0ll, n / parts);
    for (std::thread& th : pool) th.join();
}

// std::vector<T, ...>::resize without the zero fill (10 M binary nodes are 1.3 GB that the worker threads overwrite at once)
template <class T>
struct default_init_alloc : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_alloc<U>; };
    template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};

struct FastBvh {
    std::vector<CwNode> cw;                // compressed 4-wide collapse of `nodes` (what the kernels walk)
    int cw_stack_need = 0;                 // worst-case traversal stack entries
    int stack_limit = kFastMaxDepth;       // what it was built to stay below
    std::vector<FastNode, default_init_alloc<FastNode>> nodes;   // the binary SAH tree (host only: what `cw` is collapsed from)
    std::vector<int32_t> leaf_tris;        // reference leaf index k of every slot of the leaf triangle list
    double scene_absmax = 0;               // largest |coordinate| of any leaf box
    int max_depth = 0;
};

// what the host builder takes from the handle's knobs (knobs.hpp)
struct FastBuildOpts {
    int max_leaf = kFastDefaultLeaf;       // MCPT_FAST_LEAF
    double cost_tri = 1.6;                 // MCPT_FAST_CT: relative cost of one leaf triangle (cheap box reject + some exact tests)
    bool serial = false;                   // MCPT_BUILD_SERIAL
    bool talk = false;                     // MCPT_PRINT_DIAG
};
// order[k] = .obj face held by reference leaf k
void build_fast_bvh(const std::vector<FaceRec>& faces, const int32_t* order, int t, FastBvh& out, int stack_limit = kFastMaxDepth,
                    const FastBuildOpts& opts = FastBuildOpts());
// upper part over n GPU-built clusters given by their boxes (lo[3], hi[3]); see accel_build.cpp
void build_fast_upper(const double* boxes6, int n, int lower_need, FastBvh& out, int stack_limit = kFastMaxDepth, const FastBuildOpts& opts = FastBuildOpts());

}  // namespace mcpt
