// Host build of the traversal structure the fast closest-hit kernel walks.
//
// Why a second structure may be used at all: for a ray whose direction components are all non-zero the reference's
// result does not depend on the shape of its tree.  Its box test (sceneManagement.cpp:340-391) is monotone under
// box inclusion in IEEE arithmetic -- a parent's box contains its children's, "(b-o)/d" is a composition of monotone
// correctly-rounded operations, so a leaf whose own box passes has every ancestor passing as well -- hence
//     ray_intersect(ray) = lexicographic min over leaves k of (t_k, k)
//                          among { k : box test of leaf k passes, triangle test passes, t_k > 0 }      (pathTracing.cpp:334-374)
// (the reference visits leaves in ascending k and replaces only on strict '<').  Any structure that (a) reaches every
// leaf that can be that minimum and (b) evaluates the reference's own fp64 tests on it returns the identical answer.
// Rays with a zero/denormal/non-finite component (0/0 -> NaN breaks monotonicity) take the reference-shaped walk.
//
// This file builds (a): a binary SAH hierarchy over the reference's leaf boxes, at most kMaxLeaf triangles per leaf,
// depth-bounded so the traversal stack fits the LDS budget, children's boxes stored in the parent so one record
// fetch tests both children.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <numeric>
#include <thread>

#include "accel_build.hpp"

namespace mcpt {
namespace {

constexpr int kBins = 16;
double kCostNode = 1.0;   // relative cost of one inner step (two conservative box tests)

struct Box {
    double lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<double>::infinity(); hi[a] = -lo[a]; } }
    void grow(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    double half_area() const
    {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

// fn(begin, end, part) for `parts` contiguous pieces of [b, e), one thread each (the caller's included); returns when all are done
template <class F>
static void fork_join(int b, int e, int parts, F fn)
{
    const long long n = e - b;
    std::vector<std::thread> pool;
    for (int p = 1; p < parts; p++) pool.emplace_back([=]() { fn(int(b + n * p / parts), int(b + n * (p + 1) / parts), p); });
    fn(b, int(b + n / parts), 0);
    for (std::thread& th : pool) th.join();
}

// A subtree the top-level pass leaves to a worker thread: primitives idx[b, e) at `depth`, to be hung under parent.child[slot].
struct Subtree { int b, e, depth; int32_t parent; int slot; };

struct Builder {
    const std::vector<Box>& prim;       // exact leaf boxes, indexed by leaf (Morton) order k
    const std::vector<Vec3>& cen;       // their centres
    std::vector<int32_t>& idx;          // permutation being partitioned (workers own disjoint ranges of it)
    FastBvh& out;
    std::vector<Subtree>* defer = nullptr;   // top-level pass only: subtrees of at most `cut` primitives are recorded, not built
    int cut = 0;
    int max_leaf = kFastDefaultLeaf;         // most primitives a leaf may hold
    double cost_tri = 1.6;                   // relative cost of one leaf triangle (FastBuildOpts)
    int depth_limit = kFastMaxDepth;         // binary depth the tree must stay below (the walk's stack)
    // Top-level pass of a large scene: the loops over a node's primitives (bounds, bins, partition) run on `threads` threads while the
    // node has at least kParallelMin of them -- minima, maxima and counts are merged exactly, so the splits are those of the serial
    // build; only the order inside a range differs, which nothing depends on (leaves sort their triangles, the median split orders by
    // (value, index)).  10 M triangles on 16 threads: 1.0 s of single-threaded top levels -> see DESIGN.md 6.
    int threads = 1;
    std::vector<int32_t>* scratch = nullptr; // as long as idx: the partition's second buffer
    static constexpr int kParallelMin = 1 << 16;

    Builder(const std::vector<Box>& p, const std::vector<Vec3>& c, std::vector<int32_t>& i, FastBvh& o) : prim(p), cen(c), idx(i), out(o) {}

    static int ceil_log2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

    Box bounds(int b, int e) const
    {
        Box r; r.reset();
        for (int i = b; i < e; i++) r.grow(prim[idx[i]]);
        return r;
    }

    bool wide(int n) const { return threads > 1 && scratch && n >= kParallelMin; }

    Box bounds_wide(int b, int e) const
    {
        if (!wide(e - b)) return bounds(b, e);
        std::vector<Box> part(size_t(threads), Box{});
        fork_join(b, e, threads, [&](int cb_, int ce_, int p) { part[size_t(p)] = bounds(cb_, ce_); });
        Box r; r.reset();
        for (const Box& q : part) r.grow(q);
        return r;
    }

    int32_t make_leaf(int b, int e)
    {
        const int32_t start = int32_t(out.leaf_tris.size());
        // ascending k inside a leaf: ties are then met in the reference's order
        std::sort(idx.begin() + b, idx.begin() + e);
        for (int i = b; i < e; i++) out.leaf_tris.push_back(idx[i]);
        return -1 - ((start << 4) | (e - b - 1));
    }

    int32_t child(int b, int e, int depth, int32_t parent, int slot)
    {
        if (defer && e - b <= cut && e - b > 1) { defer->push_back(Subtree{b, e, depth, parent, slot}); return kFastEmpty; }
        return build(b, e, depth);
    }

    // returns child reference (>=0 inner node index, <0 leaf) for prims [b,e)
    int32_t build(int b, int e, int depth)
    {
        const int n = e - b;
        out.max_depth = std::max(out.max_depth, depth);
        if (n <= 1) return make_leaf(b, e);
        const bool force_balanced = depth + ceil_log2(n) >= depth_limit - 1;
        const bool par = wide(n);
        auto centre_bounds = [&](int cb_, int ce_) {
            Box r; r.reset();
            for (int i = cb_; i < ce_; i++) {
                const Vec3& c = cen[idx[i]];
                r.lo[0] = std::min(r.lo[0], c.x); r.hi[0] = std::max(r.hi[0], c.x);
                r.lo[1] = std::min(r.lo[1], c.y); r.hi[1] = std::max(r.hi[1], c.y);
                r.lo[2] = std::min(r.lo[2], c.z); r.hi[2] = std::max(r.hi[2], c.z);
            }
            return r;
        };
        Box cb; cb.reset();
        if (par) {
            std::vector<Box> part(size_t(threads), Box{});
            fork_join(b, e, threads, [&](int cb_, int ce_, int p) { part[size_t(p)] = centre_bounds(cb_, ce_); });
            for (const Box& q : part) cb.grow(q);
        } else cb = centre_bounds(b, e);
        int mid = -1;
        bool have_kid_boxes = false;
        Box lb, rb;
        if (!force_balanced) {
            // bins of the three axes in one pass over the primitives (an axis without extent has none)
            struct Bins { Box bb[3][kBins]; int cnt[3][kBins]; };
            double scale[3]; bool use[3];
            for (int a = 0; a < 3; a++) { const double ext = cb.hi[a] - cb.lo[a]; use[a] = ext > 0; scale[a] = use[a] ? kBins * (1.0 - 1e-12) / ext : 0.0; }
            auto clear_bins = [](Bins& q) { for (int a = 0; a < 3; a++) for (int k = 0; k < kBins; k++) { q.bb[a][k].reset(); q.cnt[a][k] = 0; } };
            auto fill_bins = [&](int cb_, int ce_, Bins& q) {
                for (int i = cb_; i < ce_; i++) {
                    const int32_t pi = idx[i];
                    const Vec3& c = cen[pi];
                    const double v[3] = {c.x, c.y, c.z};
                    for (int a = 0; a < 3; a++) {
                        if (!use[a]) continue;
                        int k = int((v[a] - cb.lo[a]) * scale[a]);
                        k = std::min(std::max(k, 0), kBins - 1);
                        q.bb[a][k].grow(prim[pi]); q.cnt[a][k]++;
                    }
                }
            };
            Bins B;                              // (on the stack: this runs once per node, ten million times)
            clear_bins(B);
            if (par) {
                std::vector<Bins> part;
                part.resize(size_t(threads));
                fork_join(b, e, threads, [&](int cb_, int ce_, int p) { clear_bins(part[size_t(p)]); fill_bins(cb_, ce_, part[size_t(p)]); });
                for (const Bins& q : part)
                    for (int a = 0; a < 3; a++) for (int k = 0; k < kBins; k++) { B.bb[a][k].grow(q.bb[a][k]); B.cnt[a][k] += q.cnt[a][k]; }
            } else fill_bins(b, e, B);
            // the node's own box: the union of any used axis' bins (every primitive is in exactly one bin per axis), else by a pass
            Box nb; nb.reset();
            { int a0 = use[0] ? 0 : (use[1] ? 1 : (use[2] ? 2 : -1));
              if (a0 >= 0) { for (int k = 0; k < kBins; k++) if (B.cnt[a0][k]) nb.grow(B.bb[a0][k]); } else nb = bounds_wide(b, e); }
            const double parent_area = nb.half_area();
            double best = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                if (!use[a]) continue;
                const Box* bb = B.bb[a]; const int* cnt = B.cnt[a];
                // Only the occupied bins take part: a split behind an empty bin has the sides, hence the cost, of the split before
                // it and is never strictly better.  Most nodes of a tree are tiny (ten million of them hold one or two triangles).
                int nz[kBins], m = 0;
                for (int k = 0; k < kBins; k++) if (cnt[k]) nz[m++] = k;
                if (m < 2) continue;
                double right_area[kBins]; int right_cnt[kBins];
                Box acc; acc.reset(); int c = 0;
                for (int j = m - 1; j > 0; j--) { acc.grow(bb[nz[j]]); c += cnt[nz[j]]; right_area[j] = acc.half_area(); right_cnt[j] = c; }
                acc.reset(); c = 0;
                for (int j = 0; j + 1 < m; j++) {
                    acc.grow(bb[nz[j]]); c += cnt[nz[j]];
                    const double cost = acc.half_area() * c + right_area[j + 1] * right_cnt[j + 1];
                    if (cost < best) { best = cost; best_axis = a; best_bin = nz[j]; }
                }
            }
            if (best_axis >= 0) {
                const double split_cost = kCostNode + cost_tri * best / std::max(parent_area, 1e-300);
                if (n <= max_leaf && cost_tri * n <= split_cost) return make_leaf(b, e);
                const int a = best_axis;
                auto goes_left = [&](int32_t p) {
                    const Vec3& c = cen[p];
                    const double v = a == 0 ? c.x : (a == 1 ? c.y : c.z);
                    int k = int((v - cb.lo[a]) * scale[a]);
                    k = std::min(std::max(k, 0), kBins - 1);
                    return k <= best_bin;
                };
                if (par) {
                    // left and right parts of every thread's piece, in piece order, through the second buffer
                    std::vector<int> nleft(size_t(threads), 0), first(size_t(threads), 0), last(size_t(threads), 0);
                    fork_join(b, e, threads, [&](int cb_, int ce_, int p) {
                        int c = 0;
                        for (int i = cb_; i < ce_; i++) c += goes_left(idx[i]) ? 1 : 0;
                        nleft[size_t(p)] = c; first[size_t(p)] = cb_; last[size_t(p)] = ce_;
                    });
                    int total_left = 0;
                    for (int c : nleft) total_left += c;
                    std::vector<int> lo_at(size_t(threads), 0), hi_at(size_t(threads), 0);
                    { int l = b, r = b + total_left;
                      for (int p = 0; p < threads; p++) { lo_at[size_t(p)] = l; hi_at[size_t(p)] = r; l += nleft[size_t(p)]; r += (last[size_t(p)] - first[size_t(p)]) - nleft[size_t(p)]; } }
                    std::vector<int32_t>& tmp = *scratch;
                    fork_join(b, e, threads, [&](int cb_, int ce_, int p) {
                        int l = lo_at[size_t(p)], r = hi_at[size_t(p)];
                        for (int i = cb_; i < ce_; i++) { const int32_t v = idx[i]; if (goes_left(v)) tmp[size_t(l++)] = v; else tmp[size_t(r++)] = v; }
                    });
                    fork_join(b, e, threads, [&](int cb_, int ce_, int) { std::copy(tmp.begin() + cb_, tmp.begin() + ce_, idx.begin() + cb_); });
                    mid = b + total_left;
                } else {
                    auto it = std::partition(idx.begin() + b, idx.begin() + e, goes_left);
                    mid = int(it - idx.begin());
                }
                if (mid == b || mid == e) mid = -1;
                else {
                    // the children's boxes are the unions of the bins on either side of the split (the same minima and maxima a pass
                    // over their primitives finds)
                    lb.reset(); rb.reset();
                    for (int k = 0; k < kBins; k++) if (B.cnt[a][k]) (k <= best_bin ? lb : rb).grow(B.bb[a][k]);
                    have_kid_boxes = true;
                }
            } else if (n <= max_leaf) {
                return make_leaf(b, e);     // all centroids coincide
            }
        }
        if (mid < 0) {                       // balanced split on the widest centroid axis
            int a = 0;
            if (cb.hi[1] - cb.lo[1] > cb.hi[a] - cb.lo[a]) a = 1;
            if (cb.hi[2] - cb.lo[2] > cb.hi[a] - cb.lo[a]) a = 2;
            mid = b + n / 2;
            std::nth_element(idx.begin() + b, idx.begin() + mid, idx.begin() + e, [&](int32_t p, int32_t q) {
                const double vp = a == 0 ? cen[p].x : (a == 1 ? cen[p].y : cen[p].z);
                const double vq = a == 0 ? cen[q].x : (a == 1 ? cen[q].y : cen[q].z);
                return vp < vq || (vp == vq && p < q);
            });
        }
        const int32_t self = int32_t(out.nodes.size());
        out.nodes.emplace_back(FastNode{});
        if (!have_kid_boxes) { lb = bounds_wide(b, mid); rb = bounds_wide(mid, e); }
        const int32_t l = child(b, mid, depth + 1, self, 0);
        const int32_t r = child(mid, e, depth + 1, self, 1);
        FastNode& nd = out.nodes[self];
        for (int a = 0; a < 3; a++) { nd.lo[0][a] = lb.lo[a]; nd.hi[0][a] = lb.hi[a]; nd.lo[1][a] = rb.lo[a]; nd.hi[1][a] = rb.hi[a]; }
        nd.child[0] = l; nd.child[1] = r;
        return self;
    }
};

}  // namespace

namespace {

// ---- collapse the binary tree into compressed 4-wide nodes -----------------------------------------------------------
// A binary subtree the top-level collapse leaves to a worker thread
struct CollapseTask { int node, budget; int32_t parent; int slot; };

struct Collapser {
    const FastBvh& in;
    std::vector<CwNode>& out;
    std::vector<int>& height;           // binary height below each FastNode
    std::vector<int>& count;            // FastNodes in the subtree of each FastNode
    std::vector<CollapseTask>* defer = nullptr;   // top-level pass only: subtrees of at most `cut` binary nodes are recorded
    int cut = 0;

    int compute_height(int n)
    {
        if (height[n] > 0) return height[n];        // done already (a subtree a worker has been through; every inner node has height >= 1)
        int h = 0, cnt = 1;
        for (int c = 0; c < 2; c++) {
            const int32_t r = in.nodes[n].child[c];
            if (r >= 0) { h = std::max(h, 1 + compute_height(r)); cnt += count[r]; }
            else h = std::max(h, 1);
        }
        count[n] = cnt;
        return height[n] = h;
    }

    struct Kid { int32_t ref; double lo[3], hi[3]; };

    static double area(const Kid& k)
    {
        const double dx = k.hi[0] - k.lo[0], dy = k.hi[1] - k.lo[1], dz = k.hi[2] - k.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }

    // budget = stack entries still available on the path to this node; guarantees need(node) <= budget
    int32_t emit(int n, int budget, int& need)
    {
        std::vector<Kid> kids;
        auto kid_of = [&](int parent, int c) {
            Kid k; k.ref = in.nodes[parent].child[c];
            for (int a = 0; a < 3; a++) { k.lo[a] = in.nodes[parent].lo[c][a]; k.hi[a] = in.nodes[parent].hi[c][a]; }
            return k;
        };
        for (int c = 0; c < 2; c++) if (in.nodes[n].child[c] != kFastEmpty) kids.push_back(kid_of(n, c));
        // widen: open the inner child with the largest area while the stack budget allows one more sibling
        while (kids.size() < 4) {
            int pick = -1; double best = -1;
            for (size_t i = 0; i < kids.size(); i++)
                if (kids[i].ref >= 0 && area(kids[i]) > best) { best = area(kids[i]); pick = int(i); }
            if (pick < 0) break;
            // after opening there are kids.size()+1 children -> kids.size() pushes at this node; every inner child must
            // still fit: height(child) <= budget - pushes
            const int pushes = int(kids.size());
            bool fits = true;
            for (size_t i = 0; i < kids.size(); i++) {
                if (int(i) == pick) {
                    for (int c = 0; c < 2; c++) { const int32_t r = in.nodes[kids[i].ref].child[c]; if (r >= 0 && height[r] > budget - pushes) fits = false; }
                } else if (kids[i].ref >= 0 && height[kids[i].ref] > budget - pushes) fits = false;
            }
            if (!fits) break;
            const int open = kids[pick].ref;
            kids.erase(kids.begin() + pick);
            for (int c = 0; c < 2; c++) if (in.nodes[open].child[c] != kFastEmpty) kids.push_back(kid_of(open, c));
        }
        const int32_t self = int32_t(out.size());
        out.emplace_back();
        const int pushes = int(kids.size()) - 1;
        int below = 0;
        int32_t refs[4] = {kFastEmpty, kFastEmpty, kFastEmpty, kFastEmpty};
        for (size_t i = 0; i < kids.size(); i++) {
            if (kids[i].ref >= 0) {
                if (defer && count[kids[i].ref] <= cut) { defer->push_back(CollapseTask{kids[i].ref, budget - pushes, self, int(i)}); continue; }
                int nd = 0; refs[i] = emit(kids[i].ref, budget - pushes, nd); below = std::max(below, nd);
            } else refs[i] = kids[i].ref;
        }
        need = pushes + below;
        // quantise
        CwNode nd{};
        nd.nchild = uint8_t(kids.size());
        for (int a = 0; a < 3; a++) {
            double lo = std::numeric_limits<double>::infinity(), hi = -lo;
            for (const Kid& k : kids) { lo = std::min(lo, k.lo[a]); hi = std::max(hi, k.hi[a]); }
            float pf = float(lo);
            if (double(pf) > lo) pf = std::nextafter(pf, -std::numeric_limits<float>::infinity());
            const double p = double(pf);
            int e = -126;
            const double ext = hi - p;
            if (ext > 0) e = std::max(-126, int(std::ceil(std::log2(ext / 255.0))));
            for (;;) {
                const double sc = std::ldexp(1.0, e);
                bool ok = p + 255.0 * sc >= hi;
                uint32_t wlo = 0, whi = 0;
                for (size_t i = 0; ok && i < kids.size(); i++) {
                    double ql = std::floor((kids[i].lo[a] - p) / sc), qh = std::ceil((kids[i].hi[a] - p) / sc);
                    ql = std::min(std::max(ql, 0.0), 255.0); qh = std::min(std::max(qh, 0.0), 255.0);
                    while (ql > 0 && p + ql * sc > kids[i].lo[a]) ql -= 1;
                    while (qh < 255 && p + qh * sc < kids[i].hi[a]) qh += 1;
                    if (p + ql * sc > kids[i].lo[a] || p + qh * sc < kids[i].hi[a]) ok = false;
                    wlo |= uint32_t(ql) << (8 * i); whi |= uint32_t(qh) << (8 * i);
                }
                if (ok) { nd.qlo[a] = wlo; nd.qhi[a] = whi; break; }
                e++;
            }
            nd.p[a] = pf; nd.e[a] = int8_t(e);
        }
        for (int i = 0; i < 4; i++) nd.child[i] = refs[i];
        out[self] = nd;
        return self;
    }
};

}  // namespace

static void build_from_boxes(const std::vector<Box>& prim, int max_leaf, int budget0, FastBvh& out, const FastBuildOpts& opts);

void build_fast_bvh(const std::vector<FaceRec>& faces, const int32_t* order, int t, FastBvh& out, int stack_limit, const FastBuildOpts& opts)
{
    out = FastBvh();
    std::vector<Box> prim(t);
    double amax = 0;
    auto mx3 = [](double a, double b, double c) { if (a >= b && a >= c) return a; if (b >= a && b >= c) return b; return c; };   // dmax
    auto mn3 = [](double a, double b, double c) { if (a <= b && a <= c) return a; if (b <= a && b <= c) return b; return c; };   // dmin
    const auto t_start = std::chrono::steady_clock::now();
    std::mutex amax_mu;
    parallel_pieces(t, [&](long long kb, long long ke) {
        double am = 0;
        for (long long k = kb; k < ke; k++) {
            const FaceRec& f = faces[size_t(order[k])];    // the reference's own leaf box of leaf k (BVH.cpp:87-97)
            Box& q = prim[size_t(k)];
            q.lo[0] = mn3(f.v[0].x, f.v[1].x, f.v[2].x); q.hi[0] = mx3(f.v[0].x, f.v[1].x, f.v[2].x);
            q.lo[1] = mn3(f.v[0].y, f.v[1].y, f.v[2].y); q.hi[1] = mx3(f.v[0].y, f.v[1].y, f.v[2].y);
            q.lo[2] = mn3(f.v[0].z, f.v[1].z, f.v[2].z); q.hi[2] = mx3(f.v[0].z, f.v[1].z, f.v[2].z);
            for (int a = 0; a < 3; a++) {
                if (std::isfinite(q.lo[a])) am = std::max(am, std::fabs(q.lo[a]));
                if (std::isfinite(q.hi[a])) am = std::max(am, std::fabs(q.hi[a]));
            }
        }
        std::lock_guard<std::mutex> lock(amax_mu);
        amax = std::max(amax, am);
    });
    if (opts.talk) std::fprintf(stderr, "fast hierarchy (host): leaf boxes %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    out.scene_absmax = amax;
    build_from_boxes(prim, std::max(1, std::min(kFastMaxLeaf, opts.max_leaf)), stack_limit - 1, out, opts);
    out.stack_limit = stack_limit;
}

// Upper part of a two-part hierarchy (MCPT_BUILD_DEVICE_FAST): the SAH tree over the boxes of clusters the GPU has built, one
// cluster per leaf.  lower_need = traversal stack entries a cluster's own subtree needs.  A leaf child of out.cw comes back as
// -1 - cluster; the caller turns it into the index of that cluster's root node.
void build_fast_upper(const double* boxes6, int n, int lower_need, FastBvh& out, int stack_limit, const FastBuildOpts& opts)
{
    out = FastBvh();
    std::vector<Box> prim(static_cast<size_t>(n), Box{});
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { prim[size_t(i)].lo[a] = boxes6[size_t(i) * 6 + a]; prim[size_t(i)].hi[a] = boxes6[size_t(i) * 6 + 3 + a]; }
    build_from_boxes(prim, 1, stack_limit - 1 - lower_need, out, opts);
    out.stack_limit = stack_limit;
    for (CwNode& nd : out.cw)
        for (int c = 0; c < 4; c++)
            if (nd.child[c] < 0 && nd.child[c] != kFastEmpty) nd.child[c] = -1 - out.leaf_tris[size_t((-1 - nd.child[c]) >> 4)];
    out.cw_stack_need += lower_need;
}

// The trace engines mirror the first nodes of the array in LDS (trace_fast.hpp: NodeCache): relabel so that these are the top of the
// tree, breadth first -- the nodes nearly every ray steps on.  The others keep their relative order.  A walk does not depend on node
// numbers (children are ordered by entry distance, pushed by slot): same visits, same results.
static void cw_top_first(std::vector<CwNode>& cw, int n_top)
{
    const int n = int(cw.size());
    if (n <= 1 || n_top <= 1) return;
    std::vector<int32_t> top;
    top.reserve(size_t(std::min(n, n_top)));
    top.push_back(0);
    for (size_t head = 0; head < top.size() && int(top.size()) < n_top; head++)
        for (int c = 0; c < 4 && int(top.size()) < n_top; c++) {
            const int32_t r = cw[size_t(top[head])].child[c];
            if (r >= 0 && r < n) top.push_back(r);
        }
    std::vector<int32_t> new_index(size_t(n), -1);
    for (size_t i = 0; i < top.size(); i++) new_index[size_t(top[i])] = int32_t(i);
    int32_t at = int32_t(top.size());
    for (int i = 0; i < n; i++) if (new_index[size_t(i)] < 0) new_index[size_t(i)] = at++;
    std::vector<CwNode> moved(static_cast<size_t>(n));
    parallel_pieces(n, [&](long long b, long long e) {
        for (long long i = b; i < e; i++) {
            CwNode nd = cw[size_t(i)];
            for (int c = 0; c < 4; c++) if (nd.child[c] >= 0 && nd.child[c] < n) nd.child[c] = new_index[size_t(nd.child[c])];
            moved[size_t(new_index[size_t(i)])] = nd;
        }
    });
    cw.swap(moved);
}

// binned-SAH binary tree over the boxes, collapsed to compressed 4-wide nodes whose walk needs at most budget0 stack entries
static void build_from_boxes(const std::vector<Box>& prim, int max_leaf, int budget0, FastBvh& out, const FastBuildOpts& opts)
{
    const int t = int(prim.size());
    const bool talk = opts.talk;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<Vec3> cen(static_cast<size_t>(t), Vec3{});
    std::vector<int32_t> idx(static_cast<size_t>(t), 0);
    parallel_pieces(t, [&](long long ib, long long ie) {
        for (long long i = ib; i < ie; i++) {
            cen[size_t(i)] = Vec3{0.5 * (prim[size_t(i)].lo[0] + prim[size_t(i)].hi[0]), 0.5 * (prim[size_t(i)].lo[1] + prim[size_t(i)].hi[1]),
                                  0.5 * (prim[size_t(i)].lo[2] + prim[size_t(i)].hi[2])};
            idx[size_t(i)] = int32_t(i);
        }
    });
    Builder bld(prim, cen, idx, out);
    bld.max_leaf = max_leaf; bld.depth_limit = budget0 + 1; bld.cost_tri = opts.cost_tri;
    out.nodes.reserve(size_t(t));
    out.leaf_tris.reserve(size_t(t));
    // Large scenes: the top of the tree is split here, subtrees of <= t/64 primitives are built by worker threads (each into
    // its own arrays, over its own range of idx) and appended in a fixed order, so the result does not depend on timing.
    std::vector<Subtree> subtrees;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int workers = int(std::min(16u, hw));
    std::vector<int32_t> scratch;
    if (t >= (1 << 17) && workers > 1 && !opts.serial) {
        bld.defer = &subtrees; bld.cut = std::max(4096, t / 64);
        if (t >= Builder::kParallelMin) { scratch.resize(size_t(t)); bld.scratch = &scratch; bld.threads = workers; }
    }
    const int32_t root = bld.build(0, t, 0);
    const auto t_top = std::chrono::steady_clock::now();
    if (talk) std::fprintf(stderr, "fast hierarchy (host): top of the tree %.2f s, %zu subtrees for %d threads\n",
                           std::chrono::duration<double>(t_top - t0).count(), subtrees.size(), workers);
    if (!subtrees.empty()) {
        std::vector<FastBvh> part(subtrees.size());
        std::vector<int32_t> part_root(subtrees.size(), kFastEmpty);
        std::vector<size_t> by_size(subtrees.size());        // largest first: the last tasks to start are the short ones
        std::iota(by_size.begin(), by_size.end(), size_t(0));
        std::stable_sort(by_size.begin(), by_size.end(), [&](size_t x, size_t y) { return subtrees[x].e - subtrees[x].b > subtrees[y].e - subtrees[y].b; });
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t turn = next.fetch_add(1); turn < subtrees.size(); turn = next.fetch_add(1)) {
                const size_t i = by_size[turn];
                Builder wb(prim, cen, idx, part[i]);
                part[i].nodes.reserve(size_t(subtrees[i].e - subtrees[i].b));
                part[i].leaf_tris.reserve(size_t(subtrees[i].e - subtrees[i].b));
                wb.max_leaf = max_leaf; wb.depth_limit = budget0 + 1; wb.cost_tri = opts.cost_tri;
                part_root[i] = wb.build(subtrees[i].b, subtrees[i].e, subtrees[i].depth);
            }
        };
        std::vector<std::thread> pool;
        for (int w = 1; w < workers; w++) pool.emplace_back(work);
        work();
        for (std::thread& th : pool) th.join();
        if (talk) std::fprintf(stderr, "fast hierarchy (host): subtrees built %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_top).count());
        // appended in task order (so the result does not depend on timing), the copies themselves on the worker threads
        std::vector<int32_t> node_off(subtrees.size()), leaf_off(subtrees.size());
        {
            size_t nn = out.nodes.size(), nl = out.leaf_tris.size();
            for (size_t i = 0; i < subtrees.size(); i++) { node_off[i] = int32_t(nn); leaf_off[i] = int32_t(nl); nn += part[i].nodes.size(); nl += part[i].leaf_tris.size(); }
            out.nodes.resize(nn);
            out.leaf_tris.resize(nl);
        }
        next = 0;
        auto place = [&]() {
            for (size_t i = next.fetch_add(1); i < subtrees.size(); i = next.fetch_add(1)) {
                const int32_t no = node_off[i], lo = leaf_off[i];
                auto moved = [&](int32_t r) -> int32_t {
                    if (r == kFastEmpty) return r;
                    if (r >= 0) return r + no;
                    const int32_t v = -1 - r;
                    return -1 - ((((v >> 4) + lo) << 4) | (v & 15));
                };
                FastNode* dst = out.nodes.data() + no;
                for (FastNode nd : part[i].nodes) { nd.child[0] = moved(nd.child[0]); nd.child[1] = moved(nd.child[1]); *dst++ = nd; }
                std::copy(part[i].leaf_tris.begin(), part[i].leaf_tris.end(), out.leaf_tris.begin() + lo);
                out.nodes[size_t(subtrees[i].parent)].child[subtrees[i].slot] = moved(part_root[i]);     // (a node of the top part: no two tasks share a slot)
            }
        };
        pool.clear();
        for (int w = 1; w < workers; w++) pool.emplace_back(place);
        place();
        for (std::thread& th : pool) th.join();
        for (size_t i = 0; i < subtrees.size(); i++) { out.max_depth = std::max(out.max_depth, part[i].max_depth); part[i] = FastBvh(); }
        if (talk) std::fprintf(stderr, "fast hierarchy (host): subtrees in place %.2f s after the start\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    if (root < 0) {
        // a single leaf: wrap it in a root whose second child is an empty leaf with an inverted box
        FastNode nd{};
        Box b = bld.bounds(0, t);
        for (int a = 0; a < 3; a++) {
            nd.lo[0][a] = b.lo[a]; nd.hi[0][a] = b.hi[a];
            nd.lo[1][a] = std::numeric_limits<double>::infinity(); nd.hi[1][a] = -std::numeric_limits<double>::infinity();
        }
        nd.child[0] = root; nd.child[1] = kFastEmpty;
        out.nodes.push_back(nd);
    }
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<int> height(out.nodes.size(), 0), count(out.nodes.size(), 0);
    Collapser col{out, out.cw, height, count};
    if (!subtrees.empty()) {
        // the subtrees the workers built, again on the workers; the pass from the root then stops at their roots
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t i = next.fetch_add(1); i < subtrees.size(); i = next.fetch_add(1)) {
                const int32_t r = out.nodes[size_t(subtrees[i].parent)].child[subtrees[i].slot];
                if (r >= 0) col.compute_height(r);
            }
        };
        std::vector<std::thread> pool;
        for (int w = 1; w < workers; w++) pool.emplace_back(work);
        work();
        for (std::thread& th : pool) th.join();
    }
    col.compute_height(0);
    if (talk) std::fprintf(stderr, "fast hierarchy (host): heights %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    out.cw.reserve(out.nodes.size());
    std::vector<CollapseTask> ctasks;
    if (!subtrees.empty()) { col.defer = &ctasks; col.cut = std::max<int>(4096, int(out.nodes.size() / 64)); }
    int need = 0;
    col.emit(0, budget0, need);
    if (talk) std::fprintf(stderr, "fast hierarchy (host): top of the collapse %.2f s after the heights' start, %zu tasks\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count(), ctasks.size());
    if (!ctasks.empty()) {
        // same scheme as the SAH pass: workers collapse whole subtrees into their own arrays, appended in task order
        std::vector<std::vector<CwNode>> part(ctasks.size());
        std::vector<int> part_need(ctasks.size(), 0);
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t i = next.fetch_add(1); i < ctasks.size(); i = next.fetch_add(1)) {
                Collapser wc{out, part[i], height, count};
                part[i].reserve(size_t(count[ctasks[i].node]));
                wc.emit(ctasks[i].node, ctasks[i].budget, part_need[i]);
            }
        };
        std::vector<std::thread> pool;
        for (int w = 1; w < workers; w++) pool.emplace_back(work);
        work();
        for (std::thread& th : pool) th.join();
        std::vector<int32_t> off(ctasks.size());
        {
            size_t nn = out.cw.size();
            for (size_t i = 0; i < ctasks.size(); i++) { off[i] = int32_t(nn); nn += part[i].size(); }
            out.cw.resize(nn);
        }
        next = 0;
        auto place = [&]() {
            for (size_t i = next.fetch_add(1); i < ctasks.size(); i = next.fetch_add(1)) {
                CwNode* dst = out.cw.data() + off[i];
                for (CwNode nd : part[i]) {
                    for (int c = 0; c < 4; c++) if (nd.child[c] >= 0) nd.child[c] += off[i];
                    *dst++ = nd;
                }
                out.cw[size_t(ctasks[i].parent)].child[ctasks[i].slot] = off[i];         // a worker's root is its node 0
                std::vector<CwNode>().swap(part[i]);
            }
        };
        pool.clear();
        for (int w = 1; w < workers; w++) pool.emplace_back(place);
        place();
        for (std::thread& th : pool) th.join();
        for (size_t i = 0; i < ctasks.size(); i++) need = std::max(need, (budget0 - ctasks[i].budget) + part_need[i]);   // pushes above the subtree + below
    }
    out.cw_stack_need = need;
    cw_top_first(out.cw, kFastTopNodes);
    if (talk) {
        const auto t2 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "fast hierarchy (host): %d triangles, SAH build %.2f s, collapse %.2f s, %zu binary / %zu wide nodes, depth %d, stack need %d\n", t,
                     std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), out.nodes.size(), out.cw.size(),
                     out.max_depth, need);
    }
}

}  // namespace mcpt
