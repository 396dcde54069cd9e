// Device build of the reference's acceleration inputs: Morton keys (MTPC/morton code.cpp:3-32), the
// Morton ordering of the faces (MTPC/MTPC.cpp:44, stable), the per-leaf boxes and the bottom-up union of the implicit
// complete tree (MTPC/BVH.cpp:56-124), written straight into the records the walk kernels read.  Results are
// bit-identical to the host build (bvh_build.cpp); tests compare them node for node.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

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
    rc = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys_out, idx, d_order, t, 0, 30, st);
    if (rc == hipSuccess) rc = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (rc == hipSuccess) rc = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_out, idx, d_order, t, 0, 30, st);
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

hipError_t device_gather_tris(const DTri* tris, const int32_t* d_slots, int n, DTri* out, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_gather_tris, dim3((n + 255) / 256), dim3(256), 0, st, tris, d_slots, n, out);
    return hipGetLastError();
}

}  // namespace mcpt
