// Morton keys, stable Morton ordering and the reference's implicit complete-tree BVH
// (MTPC/morton code.cpp:3-32, MTPC/MTPC.cpp:44, MTPC/BVH.cpp:37-132), built on the host.
#include <algorithm>
#include <numeric>

#include "scene.hpp"

namespace mcpt {

namespace {
// 10 bits -> every third bit (morton code.cpp:3-10)
inline uint32_t spread3(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
inline uint32_t quantize10(float unit)
{
    float s = unit * 1024.0f;
    s = std::max(s, 0.0f);
    s = std::min(s, 1023.0f);
    return static_cast<uint32_t>(s);
}
inline int popcount32(int x) { return __builtin_popcount(static_cast<unsigned>(x)); }
inline double tri_max(double a, double b, double c)   // dmax, sceneManagement.cpp:10-15
{
    if (a >= b && a >= c) return a;
    if (b >= a && b >= c) return b;
    return c;
}
inline double tri_min(double a, double b, double c)   // dmin, sceneManagement.cpp:3-8
{
    if (a <= b && a <= c) return a;
    if (b <= a && b <= c) return b;
    return c;
}
}  // namespace

// getMortonCode: fixed domain [-1,4]^3, float arithmetic (morton code.h:6-7, morton code.cpp:22-32)
uint32_t morton_code(float x, float y, float z)
{
    const float lo = -1.0f, span = 5.0f;
    const uint32_t xx = spread3(quantize10((x - lo) / span));
    const uint32_t yy = spread3(quantize10((y - lo) / span));
    const uint32_t zz = spread3(quantize10((z - lo) / span));
    return xx * 4 + yy * 2 + zz;
}

// the same key on another axis-aligned domain (MCPT_LOAD_MORTON_BOUNDS); lo = -1, span = 5 gives morton_code()
uint32_t morton_code_in(float x, float y, float z, const float lo[3], const float span[3])
{
    const uint32_t xx = spread3(quantize10((x - lo[0]) / span[0]));
    const uint32_t yy = spread3(quantize10((y - lo[1]) / span[1]));
    const uint32_t zz = spread3(quantize10((z - lo[2]) / span[2]));
    return xx * 4 + yy * 2 + zz;
}

// BVH::findIndex (BVH.cpp:99-104): implicit index -> compact index = i - (virtual nodes above level l)
int find_index(const mcpt_bvh_info& b, int i, int l)
{
    const int lvl = b.Lv >> (b.Level - l + 1);
    return i - (2 * lvl - popcount32(lvl));
}

// BVH::haveRightSubtree (BVH.cpp:126-132)
bool has_right_child(const mcpt_bvh_info& b, int node, int l)
{
    const long first_virtual = (1L << (l + 2)) - 1 - (b.Lv >> (b.Level - l - 1));
    return 2L * node + 2 < first_virtual;
}

// t, Lc, Lv, Nc, Nv, Nr, Level of BVH.cpp:46-52 for t faces
mcpt_bvh_info bvh_shape(int t)
{
    mcpt_bvh_info b{};
    b.t = t;
    b.Lc = 1;
    b.Level = 0;
    while (b.Lc < t) { b.Lc <<= 1; b.Level++; }    // Lc = 2^ceil(log2 t); Level = floor(log2(2 Lc - 1)) = log2 Lc
    b.Lv = b.Lc - t;
    b.Nc = 2 * b.Lc - 1;
    b.Nv = 2 * b.Lv - popcount32(b.Lv);
    b.Nr = 2 * t - 1 + popcount32(b.Lv);
    return b;
}

int build_accel(Scene& s, std::string& err)
{
    const int t = int(s.faces.size());
    if (t <= 0) { err = "scene has no faces"; return MCPT_ERR_PARSE; }
    // MTPC.cpp:44 with a stable order for equal keys (D2)
    s.order.resize(t);
    std::iota(s.order.begin(), s.order.end(), 0);
    std::stable_sort(s.order.begin(), s.order.end(),
                     [&](int a, int b) { return s.faces[a].morton < s.faces[b].morton; });

    s.bi = bvh_shape(t);
    mcpt_bvh_info& b = s.bi;

    s.nodes.assign(b.Nr, NodeBox{});
    s.node_level.assign(b.Nr, 0);
    s.node_leaf.assign(b.Nr, -1);
    for (int l = b.Level; l >= 0; l--) {           // BVH.cpp:56-84, bottom-up
        const int first = (1 << l) - 1;
        const int end = (1 << (l + 1)) - 1 - (b.Lv >> (b.Level - l));
        for (int i = first; i < end; i++) {
            const int self = find_index(b, i, l);
            NodeBox& nb = s.nodes[self];
            s.node_level[self] = l;
            if (l == b.Level) {
                const int k = i - first;
                const FaceRec& f = s.faces[s.order[k]];
                s.node_leaf[self] = k;
                nb.max_x = tri_max(f.v[0].x, f.v[1].x, f.v[2].x);   // findBondingBox(Face&), BVH.cpp:87-97
                nb.max_y = tri_max(f.v[0].y, f.v[1].y, f.v[2].y);
                nb.max_z = tri_max(f.v[0].z, f.v[1].z, f.v[2].z);
                nb.min_x = tri_min(f.v[0].x, f.v[1].x, f.v[2].x);
                nb.min_y = tri_min(f.v[0].y, f.v[1].y, f.v[2].y);
                nb.min_z = tri_min(f.v[0].z, f.v[1].z, f.v[2].z);
            } else {
                const NodeBox& c1 = s.nodes[find_index(b, 2 * i + 1, l + 1)];
                if (has_right_child(b, i, l)) {                      // BVH.cpp:106-114
                    const NodeBox& c2 = s.nodes[find_index(b, 2 * i + 2, l + 1)];
                    nb.max_x = std::max(c1.max_x, c2.max_x);
                    nb.max_y = std::max(c1.max_y, c2.max_y);
                    nb.max_z = std::max(c1.max_z, c2.max_z);
                    nb.min_x = std::min(c1.min_x, c2.min_x);
                    nb.min_y = std::min(c1.min_y, c2.min_y);
                    nb.min_z = std::min(c1.min_z, c2.min_z);
                } else {
                    nb = c1;                                         // BVH.cpp:116-124
                }
            }
        }
    }
    s.accel_built = true;
    return MCPT_OK;
}

}  // namespace mcpt
