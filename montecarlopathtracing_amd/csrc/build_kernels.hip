// Device build of the reference's acceleration inputs: Morton keys (MTPC/morton code.cpp:3-32), the
// Morton ordering of the faces (MTPC/MTPC.cpp:44, stable), the per-leaf boxes and the bottom-up union of the implicit
// complete tree (MTPC/BVH.cpp:56-124), written straight into the records the walk kernels read.  Results are
// bit-identical to the host build (bvh_build.cpp); tests compare them node for node.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>
#include <rocprim/rocprim.hpp>

#include "build_kernels.hpp"
#include "dev_common.hpp"

namespace mcpt {

__device__ __forceinline__ uint32_t spread3(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t quant10(float unit)
{
    float s = unit * 1024.0f;
    s = (s < 0.0f) ? 0.0f : s;          // std::max(s, 0.0f)
    s = (1023.0f < s) ? 1023.0f : s;    // std::min(s, 1023.0f)
    return (uint32_t)s;
}

// key of face i = getMortonCode(centre), centre = ((v1+v2)+v3)/3 in fp64, narrowed to float at the call
struct MortonDomain { float lo[3], span[3]; };
__global__ void k_morton_keys(const double* __restrict__ v9, int t, MortonDomain dom, uint32_t* __restrict__ keys, int32_t* __restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t) return;
    const double* p = v9 + (size_t)i * 9;
    const float cx = (float)(((p[0] + p[3]) + p[6]) / 3), cy = (float)(((p[1] + p[4]) + p[7]) / 3), cz = (float)(((p[2] + p[5]) + p[8]) / 3);
    const uint32_t xx = spread3(quant10((cx - dom.lo[0]) / dom.span[0]));      // reference: lo = -1, span = 5
    const uint32_t yy = spread3(quant10((cy - dom.lo[1]) / dom.span[1]));
    const uint32_t zz = spread3(quant10((cz - dom.lo[2]) / dom.span[2]));
    keys[i] = xx * 4 + yy * 2 + zz;
    idx[i] = i;
}

// leaf k <- face order[k]: hit-test record, shading record and the leaf's box (findBondingBox(Face&), BVH.cpp:87-97)
__global__ void k_fill_leaves(const double* __restrict__ v9, const double* __restrict__ vn9, const double* __restrict__ vt6,
                              const double* __restrict__ nrm3, const int32_t* __restrict__ material, const int32_t* __restrict__ order, int t,
                              DTri* __restrict__ tris, DTriShade* __restrict__ shade, DNode* __restrict__ leaf_nodes)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= t) return;
    const int f = order[k];
    const double* p = v9 + (size_t)f * 9; const double* n = vn9 + (size_t)f * 9; const double* q = vt6 + (size_t)f * 6;
    DTri tr;
    for (int i = 0; i < 3; i++) { tr.v1[i] = p[i]; tr.v2[i] = p[3 + i]; tr.v3[i] = p[6 + i]; tr.n[i] = nrm3[(size_t)f * 3 + i]; }
    tr.material = material[f]; tr.face = f; tr.leaf = k;
    for (int i = 0; i < 5; i++) tr.pad[i] = 0;
    tris[k] = tr;
    DTriShade sh;
    for (int i = 0; i < 3; i++) { sh.vn1[i] = n[i]; sh.vn2[i] = n[3 + i]; sh.vn3[i] = n[6 + i]; }
    sh.vt1[0] = q[0]; sh.vt1[1] = q[1]; sh.vt2[0] = q[2]; sh.vt2[1] = q[3]; sh.vt3[0] = q[4]; sh.vt3[1] = q[5]; sh.pad = 0;
    shade[k] = sh;
    DNode nd;
    for (int i = 0; i < 3; i++) { nd.mn[i] = dmin3(p[i], p[3 + i], p[6 + i]); nd.mx[i] = dmax3(p[i], p[3 + i], p[6 + i]); }
    nd.pad[0] = nd.pad[1] = 0;
    leaf_nodes[k] = nd;
}

__device__ __forceinline__ int dev_find_index(int Lv, int Level, int i, int l)
{
    const int lvl = Lv >> (Level - l + 1);
    return i - (2 * lvl - __popc(lvl));
}

// one level of the bottom-up pass: parent = union of its children, or a copy of the left child when the right is virtual
__global__ void k_build_level(DNode* __restrict__ nodes, int Lv, int Level, int l)
{
    const int first = (1 << l) - 1;
    const int end = (1 << (l + 1)) - 1 - (Lv >> (Level - l));
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end) return;
    const int end_child = (1 << (l + 2)) - 1 - (Lv >> (Level - l - 1));
    const DNode c1 = nodes[dev_find_index(Lv, Level, 2 * i + 1, l + 1)];
    DNode out = c1;
    if (2 * i + 2 < end_child) {
        const DNode c2 = nodes[dev_find_index(Lv, Level, 2 * i + 2, l + 1)];
        for (int a = 0; a < 3; a++) {
            out.mx[a] = (c1.mx[a] < c2.mx[a]) ? c2.mx[a] : c1.mx[a];      // std::max
            out.mn[a] = (c2.mn[a] < c1.mn[a]) ? c2.mn[a] : c1.mn[a];      // std::min
        }
    }
    nodes[dev_find_index(Lv, Level, i, l)] = out;
}

__global__ void k_gather_tris(const DTri* __restrict__ tris, const int32_t* __restrict__ slots, int n, DTri* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = tris[slots[i]];
}

// ---------------------------------------------------------------------------------------------- fast hierarchy on the device
struct FBox { double lo[3], hi[3]; };

// The fast hierarchy keeps its own order: 63-bit Morton codes (21 bits per axis) of the triangle centres on the scene's bounding
// box.  The reference's 30-bit keys on the fixed [-1,4]^3 cube leave thousands of triangles of a large scene on one key, in
// .obj order -- groups of four of those make useless leaves (measured: 174 triangle tests per ray on the 10 M-triangle scene).
__device__ __forceinline__ unsigned long long spread21(unsigned long long x)
{
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
struct FastDomain { double lo[3], inv[3]; };
__global__ void k_fast_keys(const DTri* __restrict__ tris, int t, FastDomain dom, unsigned long long* __restrict__ keys, int32_t* __restrict__ idx)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= t) return;
    const DTri* tr = tris + k;
    unsigned long long key = 0;
    for (int a = 0; a < 3; a++) {
        const double c = (tr->v1[a] + tr->v2[a] + tr->v3[a]) * (1.0 / 3.0);
        double u = (c - dom.lo[a]) * dom.inv[a] * 2097152.0;
        u = u >= 0.0 ? u : 0.0;                   // NaN -> 0
        u = u <= 2097151.0 ? u : 2097151.0;
        key |= spread21((unsigned long long)u) << (2 - a);
    }
    keys[k] = key; idx[k] = k;
}

// box of fast leaf g = triangles 4g .. 4g+3 of the sorted copy (each triangle's own box is the reference's, BVH.cpp:87-97)
__global__ void k_fast_leaf_boxes(const DTri* __restrict__ tris, int t, int per_leaf, int groups, FBox* __restrict__ boxes, unsigned long long* __restrict__ absmax_bits)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    double am = 0.0;
    if (g < groups) {
        FBox b;
        for (int a = 0; a < 3; a++) { b.lo[a] = __builtin_inf(); b.hi[a] = -__builtin_inf(); }
        const int first = per_leaf * g, end = first + per_leaf < t ? first + per_leaf : t;
        for (int k = first; k < end; k++) {
            const DTri* tr = tris + k;
            for (int a = 0; a < 3; a++) {
                const double lo = fmin(fmin(tr->v1[a], tr->v2[a]), tr->v3[a]), hi = fmax(fmax(tr->v1[a], tr->v2[a]), tr->v3[a]);
                b.lo[a] = fmin(b.lo[a], lo); b.hi[a] = fmax(b.hi[a], hi);
                if (isfinite(lo)) am = fmax(am, fabs(lo));
                if (isfinite(hi)) am = fmax(am, fabs(hi));
            }
        }
        boxes[g] = b;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmax(am, __shfl_down(am, off, 64));
    if ((threadIdx.x & 63) == 0 && am > 0.0) atomicMax(absmax_bits, (unsigned long long)__double_as_longlong(am));   // non-negative doubles order like integers
}

// One level: node i takes children 4i .. 4i+3 of the level below (nodes, or fast leaves when leaf_level), stores its own exact
// box for the level above and its compressed record: per axis a grid origin p (fp32, rounded down), a power-of-two step and the
// children's planes as 8-bit offsets rounded outward, verified in fp64 (the same rule as the host builder, accel_build.cpp).
__global__ void k_fast_level(const FBox* __restrict__ child_boxes, int n_children, int child_base, int per_leaf /* 0: children are nodes */, int t,
                             FBox* __restrict__ my_boxes, CwNode* __restrict__ nodes, int node_base, int n_nodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    FBox kid[4];
    int nk = 0;
    for (int c = 0; c < 4; c++) if (4 * i + c < n_children) kid[nk++] = child_boxes[4 * i + c];
    FBox own;
    for (int a = 0; a < 3; a++) {
        own.lo[a] = __builtin_inf(); own.hi[a] = -__builtin_inf();
        for (int c = 0; c < nk; c++) { own.lo[a] = fmin(own.lo[a], kid[c].lo[a]); own.hi[a] = fmax(own.hi[a], kid[c].hi[a]); }
    }
    my_boxes[i] = own;
    CwNode nd;
    nd.nchild = (uint8_t)nk;
    nd.pad[0] = nd.pad[1] = 0;
    for (int a = 0; a < 3; a++) {
        const float pf = __double2float_rd(own.lo[a]);
        const double p = (double)pf;
        int e = -126;
        const double ext = own.hi[a] - p;
        if (ext > 0) { const int want = (int)ceil(log2(ext / 255.0)); e = want > -126 ? want : -126; }
        uint32_t wlo = 0, whi = 0;
        for (;; e++) {
            const double sc = ldexp(1.0, e);
            bool ok = p + 255.0 * sc >= own.hi[a];
            wlo = 0; whi = 0;
            for (int c = 0; ok && c < nk; c++) {
                double ql = floor((kid[c].lo[a] - p) / sc), qh = ceil((kid[c].hi[a] - p) / sc);
                ql = fmin(fmax(ql, 0.0), 255.0); qh = fmin(fmax(qh, 0.0), 255.0);
                while (ql > 0 && p + ql * sc > kid[c].lo[a]) ql -= 1;
                while (qh < 255 && p + qh * sc < kid[c].hi[a]) qh += 1;
                if (p + ql * sc > kid[c].lo[a] || p + qh * sc < kid[c].hi[a]) ok = false;
                wlo |= (uint32_t)ql << (8 * c); whi |= (uint32_t)qh << (8 * c);
            }
            if (ok || e >= 127) break;          // e = 127 cannot fail for finite boxes; non-finite scenes never use this structure
        }
        nd.p[a] = pf; nd.e[a] = (int8_t)e; nd.qlo[a] = wlo; nd.qhi[a] = whi;
    }
    for (int c = 0; c < 4; c++) {
        const int ci = 4 * i + c;
        if (ci >= n_children) nd.child[c] = (int32_t)0x80000000;               // MCPT_FAST_EMPTY
        else if (per_leaf) { const int first = per_leaf * ci, count = (first + per_leaf < t ? per_leaf : t - first); nd.child[c] = -1 - ((first << 4) | (count - 1)); }
        else nd.child[c] = child_base + ci;
    }
    nodes[node_base + i] = nd;
}

__global__ void k_offset_children(CwNode* __restrict__ nodes, int n, int off)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < 4; c++) { const int32_t r = nodes[i].child[c]; if (r >= 0) nodes[i].child[c] = r + off; }
}

hipError_t device_offset_children(CwNode* nodes, int n, int off, hipStream_t st)
{
    if (n > 0 && off != 0) hipLaunchKernelGGL(k_offset_children, dim3((n + 255) / 256), dim3(256), 0, st, nodes, n, off);
    return hipGetLastError();
}

hipError_t device_build_fast(const DTri* leaf_tris, int t, const double lo[3], const double hi[3], int per_leaf, int max_levels, CwNode** cw, DTri** fast_tris,
                             int* n_nodes, int* levels, int* n_top, std::vector<double>* top_boxes, double* absmax, hipStream_t st)
{
    *cw = nullptr; *fast_tris = nullptr; *n_nodes = 0; *levels = 0; *n_top = 0; *absmax = 0;
    if (t <= 0 || t > (1 << 27)) return hipErrorInvalidValue;                  // leaf references hold first << 4
    // ---- order: sort (63-bit Morton code, leaf index), gather the triangle records into that order
    DTri* tris = nullptr;
    {
        unsigned long long *keys = nullptr, *keys_out = nullptr;
        int32_t *idx = nullptr, *idx_out = nullptr;
        void* tmp = nullptr;
        size_t tmp_bytes = 0;
        auto drop = [&]() { (void)hipFree(keys); (void)hipFree(keys_out); (void)hipFree(idx); (void)hipFree(idx_out); (void)hipFree(tmp); };
        hipError_t rc = hipMalloc(reinterpret_cast<void**>(&keys), size_t(t) * 8);
        if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&keys_out), size_t(t) * 8);
        if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&idx), size_t(t) * 4);
        if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&idx_out), size_t(t) * 4);
        if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&tris), size_t(t) * sizeof(DTri));
        if (rc != hipSuccess) { drop(); (void)hipFree(tris); return rc; }
        FastDomain dom;
        for (int a = 0; a < 3; a++) { dom.lo[a] = lo[a]; const double ext = hi[a] - lo[a]; dom.inv[a] = ext > 0 ? 1.0 / ext : 0.0; }
        hipLaunchKernelGGL(k_fast_keys, dim3((t + 255) / 256), dim3(256), 0, st, leaf_tris, t, dom, keys, idx);
        rc = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_out, idx, idx_out, t, 0, 63, st);
        if (rc == hipSuccess) rc = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
        if (rc == hipSuccess) rc = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_out, idx, idx_out, t, 0, 63, st);
        if (rc == hipSuccess) { hipLaunchKernelGGL(k_gather_tris, dim3((t + 255) / 256), dim3(256), 0, st, leaf_tris, idx_out, t, tris); rc = hipGetLastError(); }
        if (rc == hipSuccess) rc = hipStreamSynchronize(st);
        drop();
        if (rc != hipSuccess) { (void)hipFree(tris); return rc; }
    }
    const int groups = (t + per_leaf - 1) / per_leaf;
    int size[16], L = 0;                                                       // size[d] = nodes of inner level d, bottom first
    for (int n = groups;;) { n = (n + 3) / 4; size[L++] = n; if (n == 1 || L == max_levels) break; }   // stops below the root: a forest
    int total = 0;
    for (int d = 0; d < L; d++) total += size[d];
    FBox *a = nullptr, *b = nullptr;
    unsigned long long* am = nullptr;
    CwNode* nodes = nullptr;
    auto cleanup = [&]() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(am); };
    hipError_t rc = hipMalloc(reinterpret_cast<void**>(&a), size_t(groups) * sizeof(FBox));
    if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&b), size_t(size[0]) * sizeof(FBox));
    if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&am), sizeof(unsigned long long));
    if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&nodes), size_t(total) * sizeof(CwNode));
    if (rc == hipSuccess) rc = hipMemsetAsync(am, 0, sizeof(unsigned long long), st);
    if (rc != hipSuccess) { cleanup(); (void)hipFree(nodes); (void)hipFree(tris); return rc; }
    hipLaunchKernelGGL(k_fast_leaf_boxes, dim3((groups + 255) / 256), dim3(256), 0, st, tris, t, per_leaf, groups, a, am);
    // root = node 0: level bases run top-down while the levels are built bottom-up
    int n_children = groups;
    for (int d = 0; d < L; d++) {
        int base = 0, child_base = 0;
        for (int u = L - 1; u > d; u--) base += size[u];
        child_base = base + size[d];                                           // the level below follows this one
        hipLaunchKernelGGL(k_fast_level, dim3((size[d] + 255) / 256), dim3(256), 0, st, a, n_children, child_base, d == 0 ? per_leaf : 0, t, b, nodes, base,
                           size[d]);
        FBox* tmp = a; a = b; b = tmp;                                         // b (size[0] entries) is large enough for every later level
        n_children = size[d];
    }
    unsigned long long bits = 0;
    rc = hipGetLastError();
    // (pageable host memory -- a stack word, a vector -- is only ever touched by blocking copies after the stream has drained: an
    // asynchronous copy into it goes through the runtime's pin-on-the-fly / staging paths, and an early return would leave a DMA
    // pending into a dead frame)
    if (rc == hipSuccess) rc = hipStreamSynchronize(st);
    if (rc == hipSuccess) rc = hipMemcpy(&bits, am, sizeof bits, hipMemcpyDeviceToHost);
    // exact boxes of the top level built here (nodes 0 .. size[L-1]-1): what the host's builder sees of each cluster; after the
    // last swap they are in `a`
    if (rc == hipSuccess && top_boxes) {
        top_boxes->resize(size_t(size[L - 1]) * 6);
        rc = hipMemcpy(top_boxes->data(), a, size_t(size[L - 1]) * sizeof(FBox), hipMemcpyDeviceToHost);
    }
    cleanup();
    if (rc != hipSuccess) { (void)hipFree(nodes); (void)hipFree(tris); return rc; }
    double v; std::memcpy(&v, &bits, sizeof v);
    *cw = nodes; *fast_tris = tris; *n_nodes = total; *levels = L; *n_top = size[L - 1]; *absmax = v;
    return hipSuccess;
}


// ---------------------------------------------------------------------------------------------- SAH-quality lower hierarchy on the device
// MCPT_BUILD_DEVICE_SAH.  Parallel locally-ordered clustering (Meister & Bittner 2017: agglomerative, each cluster looks `radius`
// places up and down the Morton order for the partner with the smallest joint surface area, mutual choices merge): the tree it grows
// from the triangles is of the quality of a top-down SAH build where the SAH matters most, at the bottom.  It is grown only as far
// as subtrees of max_cluster triangles and max_height binary levels, so that (a) the walk's stack need below a cluster is bounded
// by construction and (b) what remains -- a few percent of the primitives -- is a job of milliseconds for the host's binned-SAH
// builder with its stack budget (accel_build.cpp: build_fast_upper), exactly as for MCPT_BUILD_DEVICE_FAST.  Every finished
// cluster is collapsed by one thread into compressed 4-wide nodes with the host collapser's rules (open the child of largest
// area while the heights below still fit the budget; subtrees of at most MCPT leaf-size triangles become leaves; planes quantised
// outward and verified in fp64), its triangles laid out in depth-first order so that every leaf is a run of the triangle array.
// Deterministic: clusters, node order and triangle order depend on the input only (ordered compaction, no atomics in the layout).
struct PlocArrays {
    FBox* box;              // [2t] node boxes: triangle k of the sorted copy = node k; inner nodes from t
    int32_t* left; int32_t* right;      // [2t] (inner nodes only)
    int32_t* cnt;           // [2t] triangles below
    int32_t* hgt;           // [2t] binary height (a triangle: 0)
    float* cost;            // [2t] SAH cost of the best way to finish the subtree (leaf or split), in units of area
    int32_t* leaf;          // [2t] 1: that best way is one leaf
    int32_t* first;         // [2t] smallest triangle position below: a name of the cluster that does not depend on the order in which the
                            //      merges of a round got their node numbers (ties between equal areas are broken by it)
};

__device__ __forceinline__ float ploc_area(const FBox& b)
{
    const float dx = (float)(b.hi[0] - b.lo[0]), dy = (float)(b.hi[1] - b.lo[1]), dz = (float)(b.hi[2] - b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}
__global__ void k_ploc_init(int t, int32_t* __restrict__ cid, PlocArrays A, float cost_tri, float cost_leaf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t) { cid[i] = i; A.cnt[i] = 1; A.hgt[i] = 0; A.cost[i] = (cost_leaf + cost_tri) * ploc_area(A.box[i]); A.leaf[i] = 1; A.first[i] = i; }
}

__device__ __forceinline__ float ploc_joint_area(const FBox& a, const FBox& b)
{
    const float dx = (float)(fmax(a.hi[0], b.hi[0]) - fmin(a.lo[0], b.lo[0]));
    const float dy = (float)(fmax(a.hi[1], b.hi[1]) - fmin(a.lo[1], b.lo[1]));
    const float dz = (float)(fmax(a.hi[2], b.hi[2]) - fmin(a.lo[2], b.lo[2]));
    return dx * dy + dy * dz + dz * dx;
}

__device__ __forceinline__ unsigned int ploc_pair_hash(int a, int b)
{
    const unsigned int lo = (unsigned int)(a < b ? a : b), hi = (unsigned int)(a < b ? b : a);
    unsigned int h = lo * 0x9e3779b1u ^ (hi * 0x85ebca77u + 0x165667b1u);
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    return h;
}
// nn[i] = position of the partner cluster i would merge with (-1: none allowed any more -> the cluster is finished)
__global__ void k_ploc_nn(const int32_t* __restrict__ cid, int n, PlocArrays A, int radius, int max_cluster, int max_height, float max_area, int32_t* __restrict__ nn)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = cid[i];
    const int ca = A.cnt[a], ha = A.hgt[a], fa = A.first[a];
    int best = -1;
    float best_area = __builtin_inff();
    unsigned int best_hash = 0xffffffffu;
    if (ca < max_cluster && ha < max_height) {
        const FBox ba = A.box[a];
        const int j0 = i - radius > 0 ? i - radius : 0, j1 = i + radius < n - 1 ? i + radius : n - 1;
        for (int j = j0; j <= j1; j++) {
            if (j == i) continue;
            const int b = cid[j];
            if (ca + A.cnt[b] > max_cluster || A.hgt[b] >= max_height) continue;
            const float ar = ploc_joint_area(ba, A.box[b]);
            if (ar > max_area) continue;            // (keeps the clusters compact and a wall-sized triangle on its own: the top-down builder places those)
            // equal areas (regular meshes are full of them) are ordered by a hash of the pair: with "the lower position wins" a row of
            // equal triangles would merge one pair per round
            const unsigned int hs = ploc_pair_hash(fa, A.first[b]);
            if (ar < best_area || (ar == best_area && hs < best_hash)) { best_area = ar; best_hash = hs; best = j; }
        }
    }
    nn[i] = best;
}

// mutual choices merge (the new node takes the lower position); cid[i] becomes: node >= 0 still active, -1 gone, -2 - node finished
__global__ void k_ploc_merge(int32_t* __restrict__ cid, int n, const int32_t* __restrict__ nn, PlocArrays A, int t, int32_t* __restrict__ n_inner, int max_leaf,
                             float cost_tri, float cost_node, float cost_leaf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = nn[i];
    const int a = cid[i];
    if (j < 0) { cid[i] = -2 - a; return; }
    if (nn[j] != i) return;
    if (i > j) return;                      // (the partner at the lower position does the merge; this slot is cleared below)
    const int b = cid[j];
    const int id = t + atomicAdd(n_inner, 1);
    FBox u;
    const FBox ba = A.box[a], bb = A.box[b];
    for (int k = 0; k < 3; k++) { u.lo[k] = fmin(ba.lo[k], bb.lo[k]); u.hi[k] = fmax(ba.hi[k], bb.hi[k]); }
    A.box[id] = u; A.left[id] = a; A.right[id] = b;
    A.cnt[id] = A.cnt[a] + A.cnt[b];
    const int ha = A.hgt[a], hb = A.hgt[b];
    A.hgt[id] = 1 + (ha > hb ? ha : hb);
    // the surface-area heuristic, bottom up: one leaf of all its triangles, or this node over the best of both sides
    const int c = A.cnt[a] + A.cnt[b];
    const float ar = ploc_area(u);
    const float as_leaf = (cost_leaf + cost_tri * (float)c) * ar, as_split = cost_node * ar + A.cost[a] + A.cost[b];
    const bool leaf = c <= max_leaf && as_leaf <= as_split;
    A.cost[id] = leaf ? as_leaf : as_split;
    A.leaf[id] = leaf ? 1 : 0;
    { const int f1 = A.first[a], f2 = A.first[b]; A.first[id] = f1 < f2 ? f1 : f2; }
    cid[i] = id;
}
__global__ void k_ploc_clear_partner(int32_t* __restrict__ cid, int n, const int32_t* __restrict__ nn)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = nn[i];
    if (j >= 0 && j < i && nn[j] == i) cid[i] = -1;
}
struct PlocIsActive { __device__ __forceinline__ bool operator()(const int32_t& v) const { return v >= 0; } };
struct PlocIsDone { __device__ __forceinline__ bool operator()(const int32_t& v) const { return v <= -2; } };

__global__ void k_ploc_cluster_counts(const int32_t* __restrict__ done, int n, const int32_t* __restrict__ cnt, int32_t* __restrict__ tri_cnt)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) tri_cnt[c] = cnt[-2 - done[c]];
}

// One cluster per thread.  WRITE = false: count the wide nodes it will need.  WRITE = true: emit them at node_base[c] (the root
// first), the triangle order at tri_base[c], the cluster's exact box and the largest number of stack entries a walk below its
// root can hold.
struct PlocItem { int32_t node, budget, parent, slot; };
template <bool WRITE>
__global__ void k_ploc_collapse(const int32_t* __restrict__ done, int n_clusters, PlocArrays A, int t, int max_leaf, int budget0,
                                const int32_t* __restrict__ node_base, const int32_t* __restrict__ tri_base, int32_t* __restrict__ node_cnt,
                                CwNode* __restrict__ nodes, int32_t* __restrict__ perm, FBox* __restrict__ top_boxes, int32_t* __restrict__ top_refs,
                                int32_t* __restrict__ need_max)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clusters) return;
    const int root = -2 - done[c];
    if (A.leaf[root] && n_clusters > 1) {
        // the whole cluster is one leaf (a large triangle on its own, a few triangles nothing would merge with): no node of its own, the
        // tree above refers to the triangles directly
        if (!WRITE) { node_cnt[c] = 0; return; }
        const int tb0 = tri_base[c];
        int32_t ls[8]; int lsp = 0, at = tb0;
        ls[lsp++] = root;
        while (lsp > 0) {
            const int32_t n = ls[--lsp];
            if (n < t) perm[at++] = n;
            else { ls[lsp++] = A.right[n]; ls[lsp++] = A.left[n]; }
        }
        top_boxes[c] = A.box[root];
        top_refs[c] = -1 - ((tb0 << 4) | (A.cnt[root] - 1));
        return;
    }
    PlocItem stack[80];
    int sp = 0;
    int emitted = 0, tri_next = 0, need = 0;
    const int nb = WRITE ? node_base[c] : 0, tb = WRITE ? tri_base[c] : 0;
    stack[sp++] = PlocItem{root, budget0, -1, 0};
    while (sp > 0) {
        const PlocItem it = stack[--sp];
        int32_t kid[4];
        int nk = 0;
        auto is_leaf = [&](int32_t n) { return A.leaf[n] != 0; };
        if (is_leaf(it.node)) kid[nk++] = it.node;          // (a cluster that is one leaf: a node with that child alone)
        else { kid[nk++] = A.left[it.node]; kid[nk++] = A.right[it.node]; }
        // widen: open the inner child of largest area while the heights below still fit the stack budget
        while (nk < 4) {
            int pick = -1; float best = -1.0f;
            for (int i = 0; i < nk; i++) {
                if (is_leaf(kid[i])) continue;
                const FBox b = A.box[kid[i]];
                const float dx = (float)(b.hi[0] - b.lo[0]), dy = (float)(b.hi[1] - b.lo[1]), dz = (float)(b.hi[2] - b.lo[2]);
                const float ar = dx * dy + dy * dz + dz * dx;
                if (ar > best) { best = ar; pick = i; }
            }
            if (pick < 0) break;
            const int pushes = nk;              // nk + 1 children after opening
            bool fits = true;
            for (int i = 0; i < nk; i++) {
                if (i == pick) {
                    const int32_t l = A.left[kid[i]], r = A.right[kid[i]];
                    if ((!is_leaf(l) && A.hgt[l] > it.budget - pushes) || (!is_leaf(r) && A.hgt[r] > it.budget - pushes)) fits = false;
                } else if (!is_leaf(kid[i]) && A.hgt[kid[i]] > it.budget - pushes) fits = false;
            }
            if (!fits) break;
            const int32_t open = kid[pick];
            for (int i = pick; i + 1 < nk; i++) kid[i] = kid[i + 1];
            nk--;
            kid[nk++] = A.left[open]; kid[nk++] = A.right[open];
        }
        const int self = emitted++;
        const int pushes = nk - 1;
        const int depth_used = (budget0 - it.budget) + pushes;
        need = depth_used > need ? depth_used : need;
        if (WRITE && it.parent >= 0) nodes[nb + it.parent].child[it.slot] = nb + self;
        int32_t refs[4] = {(int32_t)0x80000000, (int32_t)0x80000000, (int32_t)0x80000000, (int32_t)0x80000000};
        for (int i = 0; i < nk; i++) {
            if (!is_leaf(kid[i])) { if (sp < 80) stack[sp++] = PlocItem{kid[i], it.budget - pushes, self, i}; continue; }
            // a leaf: its triangles, depth first, become the next run of the cluster's part of the triangle array
            const int count = A.cnt[kid[i]];
            if (WRITE) {
                int32_t ls[8]; int lsp = 0, at = tb + tri_next;
                ls[lsp++] = kid[i];
                while (lsp > 0) {
                    const int32_t n = ls[--lsp];
                    if (n < t) perm[at++] = n;
                    else { ls[lsp++] = A.right[n]; ls[lsp++] = A.left[n]; }
                }
                refs[i] = -1 - (((tb + tri_next) << 4) | (count - 1));
            }
            tri_next += count;
        }
        if (WRITE) {
            CwNode nd;
            nd.nchild = (uint8_t)nk;
            nd.pad[0] = nd.pad[1] = 0;
            FBox kb[4];
            for (int i = 0; i < nk; i++) kb[i] = A.box[kid[i]];
            for (int a = 0; a < 3; a++) {
                double lo = __builtin_inf(), hi = -__builtin_inf();
                for (int i = 0; i < nk; i++) { lo = fmin(lo, kb[i].lo[a]); hi = fmax(hi, kb[i].hi[a]); }
                const float pf = __double2float_rd(lo);
                const double p = (double)pf;
                int e = -126;
                const double ext = hi - p;
                if (ext > 0) { const int want = (int)ceil(log2(ext / 255.0)); e = want > -126 ? want : -126; }
                uint32_t wlo = 0, whi = 0;
                for (;; e++) {
                    const double sc = ldexp(1.0, e);
                    bool ok = p + 255.0 * sc >= hi;
                    wlo = 0; whi = 0;
                    for (int i = 0; ok && i < nk; i++) {
                        double ql = floor((kb[i].lo[a] - p) / sc), qh = ceil((kb[i].hi[a] - p) / sc);
                        ql = fmin(fmax(ql, 0.0), 255.0); qh = fmin(fmax(qh, 0.0), 255.0);
                        while (ql > 0 && p + ql * sc > kb[i].lo[a]) ql -= 1;
                        while (qh < 255 && p + qh * sc < kb[i].hi[a]) qh += 1;
                        if (p + ql * sc > kb[i].lo[a] || p + qh * sc < kb[i].hi[a]) ok = false;
                        wlo |= (uint32_t)ql << (8 * i); whi |= (uint32_t)qh << (8 * i);
                    }
                    if (ok || e >= 127) break;
                }
                nd.p[a] = pf; nd.e[a] = (int8_t)e; nd.qlo[a] = wlo; nd.qhi[a] = whi;
            }
            for (int i = 0; i < 4; i++) nd.child[i] = refs[i];          // (inner children: patched in when they are emitted)
            nodes[nb + self] = nd;
        }
    }
    if (!WRITE) node_cnt[c] = emitted;
    else { top_boxes[c] = A.box[root]; top_refs[c] = nb; atomicMax(need_max, need); }
}

hipError_t device_build_ploc(const DTri* leaf_tris, int t, const double lo[3], const double hi[3], int max_cluster, int max_height, int radius, int max_leaf,
                             double area_fraction, double cost_tri, double cost_leaf, int collapse_budget, CwNode** cw, DTri** fast_tris, int* n_nodes, int* n_top, std::vector<double>* top_boxes, std::vector<int32_t>* top_roots,
                             int* lower_need, double* absmax, int* rounds, hipStream_t st)
{
    *cw = nullptr; *fast_tris = nullptr; *n_nodes = 0; *n_top = 0; *absmax = 0; *lower_need = 0;
    if (rounds) *rounds = 0;
    if (t <= 0 || t > (1 << 27)) return hipErrorInvalidValue;
    std::vector<void*> owned;
    auto drop = [&]() { for (void* q : owned) (void)hipFree(q); owned.clear(); };
    auto take = [&](void** ptr, size_t bytes) { const hipError_t e = hipMalloc(ptr, bytes ? bytes : 1); if (e == hipSuccess) owned.push_back(*ptr); return e; };
#define PL_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { drop(); return e_; } } while (0)
    // ---- Morton order of the triangle records
    DTri* tris = nullptr;
    unsigned long long *keys = nullptr, *keys_out = nullptr;
    int32_t *idx = nullptr, *idx_out = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    PL_TRY(take(reinterpret_cast<void**>(&keys), size_t(t) * 8));
    PL_TRY(take(reinterpret_cast<void**>(&keys_out), size_t(t) * 8));
    PL_TRY(take(reinterpret_cast<void**>(&idx), size_t(t) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&idx_out), size_t(t) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&tris), size_t(t) * sizeof(DTri)));
    FastDomain dom;
    for (int a = 0; a < 3; a++) { dom.lo[a] = lo[a]; const double ext = hi[a] - lo[a]; dom.inv[a] = ext > 0 ? 1.0 / ext : 0.0; }
    hipLaunchKernelGGL(k_fast_keys, dim3((t + 255) / 256), dim3(256), 0, st, leaf_tris, t, dom, keys, idx);
    PL_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_out, idx, idx_out, t, 0, 63, st));
    PL_TRY(take(&tmp, tmp_bytes));
    PL_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_out, idx, idx_out, t, 0, 63, st));
    hipLaunchKernelGGL(k_gather_tris, dim3((t + 255) / 256), dim3(256), 0, st, leaf_tris, idx_out, t, tris);
    // ---- clustering
    PlocArrays A;
    unsigned long long* am = nullptr;
    int32_t *cid = nullptr, *cid2 = nullptr, *nn = nullptr, *done = nullptr, *counters = nullptr;
    PL_TRY(take(reinterpret_cast<void**>(&A.box), size_t(2) * t * sizeof(FBox)));
    PL_TRY(take(reinterpret_cast<void**>(&A.left), size_t(2) * t * 4));
    PL_TRY(take(reinterpret_cast<void**>(&A.right), size_t(2) * t * 4));
    PL_TRY(take(reinterpret_cast<void**>(&A.cnt), size_t(2) * t * 4));
    PL_TRY(take(reinterpret_cast<void**>(&A.hgt), size_t(2) * t * 4));
    PL_TRY(take(reinterpret_cast<void**>(&A.cost), size_t(2) * t * 4));
    PL_TRY(take(reinterpret_cast<void**>(&A.leaf), size_t(2) * t * 4));
    PL_TRY(take(reinterpret_cast<void**>(&A.first), size_t(2) * t * 4));
    PL_TRY(take(reinterpret_cast<void**>(&cid), size_t(t) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&cid2), size_t(t) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&nn), size_t(t) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&done), size_t(t) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&counters), 16 * 4));          // [0] inner nodes, [1] selected (active), [2] selected (done), [3] stack need
    PL_TRY(take(reinterpret_cast<void**>(&am), 8));
    PL_TRY(hipMemsetAsync(counters, 0, 16 * 4, st));
    PL_TRY(hipMemsetAsync(am, 0, 8, st));
    hipLaunchKernelGGL(k_fast_leaf_boxes, dim3((t + 255) / 256), dim3(256), 0, st, tris, t, 1, t, A.box, am);
    hipLaunchKernelGGL(k_ploc_init, dim3((t + 255) / 256), dim3(256), 0, st, t, cid, A, (float)cost_tri, (float)cost_leaf);
    const double sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
    const float max_area = area_fraction > 0 ? (float)((sx * sy + sy * sz + sz * sx) * area_fraction) : __builtin_inff();
    size_t sel_bytes = 0;
    PL_TRY(rocprim::select(nullptr, sel_bytes, cid, cid2, counters + 1, size_t(t), PlocIsActive(), st));
    void* sel_tmp = nullptr;
    PL_TRY(take(&sel_tmp, sel_bytes));
    int n_active = t, n_done = 0, n_rounds = 0;
    while (n_active > 0) {
        const unsigned g = unsigned((n_active + 255) / 256);
        hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(256), 0, st, cid, n_active, A, radius, max_cluster, max_height, max_area, nn);
        hipLaunchKernelGGL(k_ploc_merge, dim3(g), dim3(256), 0, st, cid, n_active, nn, A, t, counters, max_leaf, (float)cost_tri, 1.0f, (float)cost_leaf);
        hipLaunchKernelGGL(k_ploc_clear_partner, dim3(g), dim3(256), 0, st, cid, n_active, nn);
        size_t b1 = sel_bytes;
        PL_TRY(rocprim::select(sel_tmp, b1, cid, done + n_done, counters + 2, size_t(n_active), PlocIsDone(), st));
        size_t b2 = sel_bytes;
        PL_TRY(rocprim::select(sel_tmp, b2, cid, cid2, counters + 1, size_t(n_active), PlocIsActive(), st));
        int32_t h[2] = {0, 0};
        PL_TRY(hipStreamSynchronize(st));
        PL_TRY(hipMemcpy(h, counters + 1, 8, hipMemcpyDeviceToHost));       // blocking: h is a stack array
        n_active = h[0]; n_done += h[1];
        int32_t* sw = cid; cid = cid2; cid2 = sw;
        if (++n_rounds > 4096) { drop(); return hipErrorUnknown; }      // (every round finishes or merges at least one cluster)
    }
    if (rounds) *rounds = n_rounds;
    // ---- collapse the clusters
    const int nc = n_done;
    int32_t *tri_cnt = nullptr, *tri_base = nullptr, *node_cnt = nullptr, *node_base = nullptr, *perm = nullptr;
    FBox* d_top = nullptr;
    int32_t* d_refs = nullptr;
    PL_TRY(take(reinterpret_cast<void**>(&tri_cnt), size_t(nc) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&tri_base), size_t(nc) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&node_cnt), size_t(nc) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&node_base), size_t(nc) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&perm), size_t(t) * 4));
    PL_TRY(take(reinterpret_cast<void**>(&d_top), size_t(nc) * sizeof(FBox)));
    PL_TRY(take(reinterpret_cast<void**>(&d_refs), size_t(nc) * 4));
    const unsigned gc = unsigned((nc + 127) / 128);
    // Stack entries a walk may hold below a cluster root: at least the height the clusters were grown to (then the collapse can always
    // proceed); more lets it open more children per node, which shortens every walk -- as long as the tree over the nc clusters, which
    // the host builds into what is left of kFastMaxDepth, keeps ~4 levels more than a balanced binary tree needs.
    int levels_above = 1;
    while ((1ll << levels_above) < (long long)nc) levels_above++;
    int budget = 35 - (levels_above + 4);
    budget = budget < max_height ? max_height : (budget > 24 ? 24 : budget);
    if (collapse_budget > 0) budget = collapse_budget < max_height ? max_height : collapse_budget;
    // (a scene that left far more clusters than a scene of its size should -- everything too large or too far apart to merge -- leaves
    // no room for the tree above them: the caller takes another builder)
    if (nc > 1 && budget + levels_above + 2 > 35) { drop(); return hipErrorNotSupported; }
    hipLaunchKernelGGL(k_ploc_cluster_counts, dim3((nc + 255) / 256), dim3(256), 0, st, done, nc, A.cnt, tri_cnt);
    hipLaunchKernelGGL(k_ploc_collapse<false>, dim3(gc), dim3(128), 0, st, done, nc, A, t, max_leaf, budget, nullptr, nullptr, node_cnt, nullptr, nullptr, nullptr, nullptr, nullptr);
    size_t scan_bytes = 0;
    PL_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, tri_cnt, tri_base, int32_t(0), size_t(nc), rocprim::plus<int32_t>(), st));
    void* scan_tmp = nullptr;
    PL_TRY(take(&scan_tmp, scan_bytes));
    size_t sb = scan_bytes;
    PL_TRY(rocprim::exclusive_scan(scan_tmp, sb, tri_cnt, tri_base, int32_t(0), size_t(nc), rocprim::plus<int32_t>(), st));
    sb = scan_bytes;
    PL_TRY(rocprim::exclusive_scan(scan_tmp, sb, node_cnt, node_base, int32_t(0), size_t(nc), rocprim::plus<int32_t>(), st));
    int32_t last[2] = {0, 0};
    PL_TRY(hipStreamSynchronize(st));
    PL_TRY(hipMemcpy(&last[0], node_base + (nc - 1), 4, hipMemcpyDeviceToHost));
    PL_TRY(hipMemcpy(&last[1], node_cnt + (nc - 1), 4, hipMemcpyDeviceToHost));
    const int total = last[0] + last[1];
    CwNode* nodes = nullptr;
    DTri* out_tris = nullptr;
    hipError_t rc = hipMalloc(reinterpret_cast<void**>(&nodes), size_t(total > 0 ? total : 1) * sizeof(CwNode));
    if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&out_tris), size_t(t) * sizeof(DTri));
    if (rc != hipSuccess) { (void)hipFree(nodes); (void)hipFree(out_tris); drop(); return rc; }
    hipLaunchKernelGGL(k_ploc_collapse<true>, dim3(gc), dim3(128), 0, st, done, nc, A, t, max_leaf, budget, node_base, tri_base, nullptr, nodes, perm, d_top, d_refs, counters + 3);
    hipLaunchKernelGGL(k_gather_tris, dim3((t + 255) / 256), dim3(256), 0, st, tris, perm, t, out_tris);
    unsigned long long bits = 0;
    int32_t need = 0;
    rc = hipGetLastError();
    if (rc == hipSuccess) rc = hipStreamSynchronize(st);
    if (rc == hipSuccess) rc = hipMemcpy(&bits, am, 8, hipMemcpyDeviceToHost);
    if (rc == hipSuccess) rc = hipMemcpy(&need, counters + 3, 4, hipMemcpyDeviceToHost);
    if (rc == hipSuccess && top_boxes) { top_boxes->resize(size_t(nc) * 6); rc = hipMemcpy(top_boxes->data(), d_top, size_t(nc) * sizeof(FBox), hipMemcpyDeviceToHost); }
    if (rc == hipSuccess && top_roots) { top_roots->resize(size_t(nc)); rc = hipMemcpy(top_roots->data(), d_refs, size_t(nc) * 4, hipMemcpyDeviceToHost); }
    drop();
    if (rc != hipSuccess) { (void)hipFree(nodes); (void)hipFree(out_tris); return rc; }
    double v; std::memcpy(&v, &bits, sizeof v);
    *cw = nodes; *fast_tris = out_tris; *n_nodes = total; *n_top = nc; *absmax = v; *lower_need = need;
    return hipSuccess;
#undef PL_TRY
}


#define BK_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

hipError_t device_build_reference(const BuildInputs& in, const mcpt_bvh_info& bi, DNode* nodes, DTri* tris, DTriShade* shade,
                                  int32_t* d_order, hipStream_t st)
{
    const int t = in.t;
    uint32_t *keys = nullptr, *keys_out = nullptr;
    int32_t* idx = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t rc = hipSuccess;
    auto cleanup = [&]() { (void)hipFree(keys); (void)hipFree(keys_out); (void)hipFree(idx); (void)hipFree(tmp); };
    if ((rc = hipMalloc(reinterpret_cast<void**>(&keys), size_t(t) * 4)) != hipSuccess || (rc = hipMalloc(reinterpret_cast<void**>(&keys_out), size_t(t) * 4)) != hipSuccess ||
        (rc = hipMalloc(reinterpret_cast<void**>(&idx), size_t(t) * 4)) != hipSuccess) { cleanup(); return rc; }
    MortonDomain dom;
    for (int a = 0; a < 3; a++) { dom.lo[a] = in.morton_lo[a]; dom.span[a] = in.morton_span[a]; }
    hipLaunchKernelGGL(k_morton_keys, dim3((t + 255) / 256), dim3(256), 0, st, in.v9, t, dom, keys, idx);
    // stable LSD radix sort of (key, face index) on the 30 key bits: equal keys keep .obj order (D2)
    rc = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_out, idx, d_order, t, 0, 30, st);
    if (rc == hipSuccess) rc = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (rc == hipSuccess) rc = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_out, idx, d_order, t, 0, 30, st);
    if (rc != hipSuccess) { cleanup(); return rc; }
    const int leaf0 = ((1 << bi.Level) - 1) - (2 * (bi.Lv >> 1) - __builtin_popcount(unsigned(bi.Lv >> 1)));   // findIndex(2^Level - 1, Level)
    hipLaunchKernelGGL(k_fill_leaves, dim3((t + 255) / 256), dim3(256), 0, st, in.v9, in.vn9, in.vt6, in.nrm3, in.material, d_order, t, tris, shade,
                       nodes + leaf0);
    for (int l = bi.Level - 1; l >= 0; l--) {
        const int count = (1 << l) - (bi.Lv >> (bi.Level - l));
        if (count > 0) hipLaunchKernelGGL(k_build_level, dim3((count + 255) / 256), dim3(256), 0, st, nodes, bi.Lv, bi.Level, l);
    }
    rc = hipGetLastError();
    if (rc == hipSuccess) rc = hipStreamSynchronize(st);
    cleanup();
    return rc;
}

// fp32 records of the triangle phase's pre-test (device_scene.hpp: DTriPre, trace_fast.hpp: tri_pre_reject), one per slot of the
// fast triangle array.  Edges are differenced in fp64 and rounded once; a1, a2 are rounded up.  a1 = +inf (pre-test off) when
// the error analysis of the test does not cover the triangle: a non-finite coordinate, a shortest edge below 2^-16 of the
// scene's largest coordinate, or a sliver whose shortest edge is below 2^-12 of its longest.
__global__ void k_build_pre(const DTri* __restrict__ tris, int n, double absmax, DTriPre* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DTri* t = tris + i;
    const double e1[3] = {t->v2[0] - t->v1[0], t->v2[1] - t->v1[1], t->v2[2] - t->v1[2]};
    const double e2[3] = {t->v3[0] - t->v1[0], t->v3[1] - t->v1[1], t->v3[2] - t->v1[2]};
    const double e3[3] = {e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2]};
    auto len = [](const double* e) { return sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]); };
    const double l1 = len(e1), l2 = len(e2), l3 = len(e3);
    const double lmin = fmin(l1, fmin(l2, l3)), lmax = fmax(l1, fmax(l2, l3));
    DTriPre q;
    for (int a = 0; a < 3; a++) { q.v0[a] = (float)t->v1[a]; q.e1[a] = (float)e1[a]; q.e2[a] = (float)e2[a]; }
    // 1-norms of the fp32 edges and of the true ones (the two differ by 2^-24 relative): rounded up with room to spare
    q.a1 = __double2float_ru((fabs(e1[0]) + fabs(e1[1]) + fabs(e1[2])) * (1.0 + 0x1p-20));
    q.a2 = __double2float_ru((fabs(e2[0]) + fabs(e2[1]) + fabs(e2[2])) * (1.0 + 0x1p-20));
    q.pad = 0.0f;
    const bool ok = lmax < __builtin_inf() && lmin >= 0x1p-16 * absmax && lmin >= 0x1p-12 * lmax;     // (false for NaN)
    if (!ok) q.a1 = __builtin_inff();
    out[i] = q;
}

hipError_t device_build_pre(const DTri* fast_tris, int n, double absmax, DTriPre* out, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_build_pre, dim3((n + 255) / 256), dim3(256), 0, st, fast_tris, n, absmax, out);
    return hipGetLastError();
}

hipError_t device_gather_tris(const DTri* tris, const int32_t* d_slots, int n, DTri* out, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_gather_tris, dim3((n + 255) / 256), dim3(256), 0, st, tris, d_slots, n, out);
    return hipGetLastError();
}

}  // namespace mcpt
