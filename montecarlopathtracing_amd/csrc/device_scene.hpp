// Records of a scene as they sit in HBM.  Traversal touches one node or one triangle per lane per step,
// each lane somewhere else in the tree, so the unit of access is a whole record that fills exactly one or two
// 64-byte memory segments (array-of-records per node / per triangle, SoA across kinds of data: boxes, hit-test
// geometry, shading attributes, materials, light tables and textures live in separate arrays so that a step
// only pulls the bytes it needs).
#pragma once
#include <cstdint>

namespace mcpt {

// One real BVH node, compact level order (index = BVH::findIndex).  48 bytes of box padded to one 64-B segment.
struct alignas(64) DNode {
    double mn[3];
    double mx[3];
    double pad[2];
};

// Hit-test geometry of leaf k (Morton order): 96 bytes used by intersect(Ray,Face) + ids.  128 B = 2 segments.
struct alignas(128) DTri {
    double v1[3], v2[3], v3[3];
    double n[3];               // Face::norm
    int32_t material;
    int32_t face;              // .obj index
    int32_t leaf;              // k, the reference's leaf (Morton-order) index: tie-break key and index into shade[]
    int32_t pad[5];
};

// Shading attributes of leaf k, read once per accepted closest hit.
struct alignas(128) DTriShade {
    double vn1[3], vn2[3], vn3[3];
    double vt1[2], vt2[2], vt3[2];
    double pad;
};

// Fast structure: one inner node = both children's boxes + their references, 128 B = one fetch per step.
// child >= 0: inner node; child < 0: leaf, -1-child = (first << 4) | (count-1) into the permuted triangle array.
struct alignas(128) FastNode {
    double lo[2][3];
    double hi[2][3];
    int32_t child[2];
    int32_t pad[6];
};

// Compressed 4-wide node, 64 B = one memory segment = four 16-B loads per lane per step (the walk is bound by L1 line
// throughput, not by ALU or HBM): child boxes quantised to 8 bits per plane on a per-node grid
//     plane = p[axis] + q * 2^e[axis],   q in 0..255, rounded outward at build time,
// so the decoded box contains the child's fp64 box.  Only used to CULL; candidates are decided by the reference's
// own fp64 tests.  child >= 0: node index; child < 0: leaf, -1-child = (first << 4) | (count-1); MCPT_FAST_EMPTY: none.
struct alignas(64) CwNode {
    float p[3];
    int8_t e[3];
    uint8_t nchild;
    uint32_t qlo[3];           // [axis]: byte c = child c
    uint32_t qhi[3];
    int32_t child[4];
    uint32_t pad[2];
};

// fp32 companion of a fast-leaf triangle slot, 48 B = three 16-B loads: what the conservative pre-test of the triangle phase
// reads (trace_fast.hpp: tri_pre_reject).  v0 = fl32(v1), e1 = fl32(v2 - v1), e2 = fl32(v3 - v1) (differences formed in fp64);
// a1 >= |e1.x| + |e1.y| + |e1.z|, a2 likewise (rounded up).  a1 = +inf switches the pre-test off for this triangle (an edge so
// short against the scene's extent, or a sliver so thin, that the error bounds of the test would not cover the reference's own
// fp64 rounding: build_kernels.hip: k_build_pre).  Only ever used to SKIP the exact test of a triangle that cannot pass it.
struct alignas(16) DTriPre {
    float v0[3], a1;
    float e1[3], a2;
    float e2[3], pad;
};
static_assert(sizeof(DTriPre) == 48, "three 16-byte loads");

struct DFast {
    const CwNode* cw;          // compressed wide hierarchy (root = 0)
    const FastNode* nodes;
    const DTri* tris;          // DTri records permuted into fast-leaf order (leaf field = reference leaf index)
    const DTriPre* pre;        // same order: fp32 records of the pre-test
    double absmax;             // largest |coordinate| in the scene
    int32_t enabled;           // 0: scene has coordinates outside [1e-150,1e150] -> reference-shaped walk only
    int32_t stack_limit;       // per-lane stack entries of the trace engine that walks it: picks the short-stack or the deep-stack kernels
    int32_t stack_cap;         // entries of that stack the engine may use (= stack_limit; tests shrink it to force the overflow hand-over)
    int32_t cached;            // cw[0 .. cached) is the top of the tree (accel_build.cpp: cw_top_first): the engines mirror a prefix of it in LDS
};

struct alignas(16) DMaterial {
    double kd[3], ks[3];
    double Ns, Ni;
    int32_t has_map, map_w, map_h, light;
    int64_t tex_offset;        // byte offset into the texel pool (BGR rows)
    int64_t pad;
};

// One triangle of an emitter, in the order of Material::f (the order shade() walks the area CDF in).
struct alignas(16) DLightTri {
    double v1[3], v2[3], v3[3];
    double vn1[3], vn2[3], vn3[3];
};

struct alignas(16) DLight {
    double radiance[3];
    double total_area;
    int32_t material, ntri, first, cdf_sorted;   // first = offset into light_tris / light_cdf
};

struct DCamera {                                   // generateImg's frame, pathTracing.cpp:276-294
    double eye[3], start_point[3], pdx[3], pdy[3];
    int32_t width, height;
};

struct DScene {
    const DNode* nodes;
    const DTri* tris;
    const DTriShade* shade;
    const DMaterial* materials;
    const DLight* lights;
    const DLightTri* light_tris;
    const double* light_cdf;
    const uint8_t* texels;
    DFast fast;
    int32_t t, Lv, Level, Nr, num_lights, num_materials;
    double area0;                                  // range of the frozen static u1 (Q1)
    DCamera cam;
};

// device-side counters (one cache line)
struct DCounters {
    unsigned long long rays_primary, rays_shadow, rays_bounce, node_visits, tri_tests, shade_calls, samples, max_depth;
    unsigned long long shadow_skipped;   // shadow rays the reference traces although their result is never used (light behind the surface)
    unsigned long long trace_rays, trace_nodes, trace_tris;   // work done inside the dominant kernel (k_wf_trace) only
    unsigned long long trace_exact;                           // ... triangles of those that survived the pre-test (exact fp64 tests)
    unsigned long long dbg[24];   // MCPT_PRE_CHECK builds: what the pre-test saw of the first triangle it should not have rejected
    unsigned long long pp[24];    // MCPT_POOL_DEBUG builds, pool form of the finishing pass: [0..4] steps per class (node, leaf, exact, result, shade),
                                  // [5..9] lanes that claimed, [10] sleeps, [11] steps that claimed nothing, [12..16] wave cycles per class,
                                  // [17] cycles voting / claiming / sleeping, [18] wave lifetimes, [19] waves
    unsigned long long pad[24];   // diagnostics: [0..11] trace engine (MCPT_TRACE_DIAG builds), [12] rays k_wf_trace handed to the exact walk,
                                  // [13..15] finishing kernel, [16..19] logic kernel
};

}  // namespace mcpt
