// Persistent-threads closest-hit engine: the fast walk of trace_fast.hpp reorganised so that a wavefront keeps
// its 64 lanes busy although rays take very different numbers of steps.
//
//   * Every wave owns a private range of ray slots (a chunk claimed with ONE atomic on the queue head; a single word
//     saturates near 88 dequeues/us on this chip, so claims are per chunk, never per ray).  Lanes that finish a ray
//     take the next slots of the range: the refill is a ballot + prefix count, no memory traffic.
//   * Each lane is a small state machine: IDLE, INNER (about to test the two children of an inner node),
//     TRI (walking the triangles of a leaf).  Per wave iteration the wave executes ONE phase -- the one most lanes are
//     waiting for (majority vote over __ballot masks) -- so a phase always runs with at least a third of the lanes
//     that have work, instead of every phase running for whoever happens to need it.
//   * Per-lane traversal stack in LDS ([depth][lane], conflict free); rays that need the reference-shaped walk
//     (a zero / denormal / non-finite component) are not walked here: their slot goes to a side list that a second,
//     tiny launch handles, so one such ray cannot hold 63 lanes for the length of an exhaustive walk.
// Results are bit-identical to trace_closest_fast() and therefore to the reference (same tests, same order of
// evaluation per candidate, (t, k) lexicographic minimum).
#pragma once
#include "trace_fast.hpp"
#include "wavefront.hpp"

namespace mcpt {

#define MCPT_REFILL_LANES 16        /* refill as soon as this many lanes are idle */

// Src must provide:  long long total() const;
//                    bool fetch(long long q, Ray& r) const;        // false: slot holds no ray
//                    void store(long long q, bool hit, const Hit& h) const;
template <class Src>
__device__ __forceinline__ void trace_persistent(const DScene& S, const Src& src, TraceQueue* queue, long long* __restrict__ slow_list,
                                                 unsigned int slow_cap, long long chunk, int* __restrict__ stack, int stride, Work& w)
{
    const DFast& F = S.fast;
    const FastNode* __restrict__ nodes = F.nodes;
    const DTri* __restrict__ tris = F.tris;
    const long long total = src.total();
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // wave-uniform range of slots still to hand out
    long long next = 0, range_end = 0;
    bool queue_empty = false;

    // lane state
    enum { ST_IDLE = 0, ST_INNER = 1, ST_TRI = 2 };
    int state = ST_IDLE;
    long long slot = -1;
    Ray r; r.o = mk(0, 0, 0); r.d = mk(1, 1, 1);
    V3 rcp = mk(1, 1, 1);
    double margin = 0, limit = 0;
    bool found = false;
    Hit best; best.leaf = -1; best.t = 0; best.p = mk(0, 0, 0);
    int sp = 0, cur = 0, tri_i = 0, tri_end = 0;

    for (;;) {
        // ------------------------------------------------------------------ refill idle lanes
        unsigned long long idle = __ballot(state == ST_IDLE);
        if (idle && !queue_empty && (__popcll(idle) >= MCPT_REFILL_LANES || idle == ~0ull)) {
            for (;;) {                                   // until every idle lane has a ray or the queue is dry
                idle = __ballot(state == ST_IDLE);
                if (!idle) break;
                if (next >= range_end) {                 // claim a new chunk (wave-uniform)
                    unsigned long long got = 0;
                    if (lane == 0) got = atomicAdd(&queue->head, (unsigned long long)chunk);
                    got = __shfl(got, 0, 64);
                    next = (long long)got;
                    range_end = next + chunk < total ? next + chunk : total;
                    if (next >= total) { queue_empty = true; break; }
                }
                const int want = __popcll(idle);
                const long long avail = range_end - next;
                const int give = want < avail ? want : (int)avail;
                const int rank = __popcll(idle & lt_mask);
                if (state == ST_IDLE && rank < give) {
                    const long long q = next + rank;
                    Ray nr;
                    if (src.fetch(q, nr)) {
                        if (fast_path_ok(F, nr)) {
                            slot = q; r = nr;
                            rcp = mk(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
                            const double dmin = fmin(fmin(fabs(r.d.x), fabs(r.d.y)), fabs(r.d.z));
                            const double scale = fmax(fmax(F.absmax, fabs(r.o.x)), fmax(fabs(r.o.y), fabs(r.o.z)));
                            margin = dmin >= 1e-6 ? 1e-9 * scale / dmin : __builtin_inf();
                            limit = __builtin_inf();
                            found = false; best.leaf = -1; best.t = 0; best.p = mk(0, 0, 0);
                            sp = 0; cur = 0; state = ST_INNER;
                        } else {
                            const unsigned int at = atomicAdd(&queue->slow_count, 1u);
                            if (at < slow_cap) slow_list[at] = q;    // list full: the second pass scans every slot instead
                        }
                    }
                }
                next += give;
            }
        }
        // ------------------------------------------------------------------ pick the phase most lanes wait for
        const unsigned long long m_inner = __ballot(state == ST_INNER);
        const unsigned long long m_tri = __ballot(state == ST_TRI);
        if (!m_inner && !m_tri) {
            if (queue_empty) break;
            continue;                                    // everyone idle but the queue is not dry: refill next round
        }
        if (__popcll(m_inner) >= __popcll(m_tri)) {
            // -------------------------------------------------------------- one inner step
            if (state == ST_INNER) {
                const FastNode* nd = nodes + cur;
                w.nodes++;
                const Slab s0 = slab_interval(nd->lo[0], nd->hi[0], r.o, rcp);
                const Slab s1 = slab_interval(nd->lo[1], nd->hi[1], r.o, rcp);
                const int c0 = nd->child[0], c1 = nd->child[1];
                const bool h0 = c0 != MCPT_FAST_EMPTY && slab_may_hit(s0) && !(s0.entry > limit);
                const bool h1 = c1 != MCPT_FAST_EMPTY && slab_may_hit(s1) && !(s1.entry > limit);
                int nxt;
                if (h0 && h1) {
                    const bool first0 = s0.entry <= s1.entry;
                    stack[sp * stride] = first0 ? c1 : c0;
                    sp++;
                    nxt = first0 ? c0 : c1;
                } else if (h0) nxt = c0;
                else if (h1) nxt = c1;
                else if (sp > 0) { sp--; nxt = stack[sp * stride]; }
                else nxt = MCPT_FAST_EMPTY;
                if (nxt >= 0) cur = nxt;
                else if (nxt == MCPT_FAST_EMPTY) { src.store(slot, found, best); state = ST_IDLE; }
                else { const int ref = -1 - nxt; tri_i = ref >> 4; tri_end = tri_i + (ref & 15) + 1; state = ST_TRI; }
            }
        } else {
            // -------------------------------------------------------------- one triangle of the current leaf
            if (state == ST_TRI) {
                const DTri* tr = tris + tri_i;
                tri_i++;
                double lo[3], hi[3];
                lo[0] = dmin3(tr->v1[0], tr->v2[0], tr->v3[0]); hi[0] = dmax3(tr->v1[0], tr->v2[0], tr->v3[0]);
                lo[1] = dmin3(tr->v1[1], tr->v2[1], tr->v3[1]); hi[1] = dmax3(tr->v1[1], tr->v2[1], tr->v3[1]);
                lo[2] = dmin3(tr->v1[2], tr->v2[2], tr->v3[2]); hi[2] = dmax3(tr->v1[2], tr->v2[2], tr->v3[2]);
                const Slab s = slab_interval(lo, hi, r.o, rcp);
                bool pass = false;
                if (!(s.exit < 0.0) && !(s.entry > limit)) {
                    if (s.entry <= 0.0) pass = true;
                    else if (s.entry + s.entry * 0x1p-48 <= s.exit) pass = true;
                    else if (s.entry > s.exit + s.exit * 0x1p-48) pass = false;
                    else pass = box_hit_exact(lo, hi, r);
                }
                if (pass) {
                    V3 p;
                    w.tris++;
                    if (tri_hit(tr, r, p)) {
                        const double t = (p.x - r.o.x) / r.d.x;
                        const int k = tr->leaf;
                        if (t > 0 && (!found || t < best.t || (t == best.t && k < best.leaf))) {
                            found = true; best.leaf = k; best.t = t; best.p = p;
                            limit = t + margin;
                        }
                    }
                }
                if (tri_i >= tri_end) {                  // leaf done: pop
                    if (sp > 0) {
                        sp--;
                        const int nxt = stack[sp * stride];
                        if (nxt >= 0) { cur = nxt; state = ST_INNER; }
                        else { const int ref = -1 - nxt; tri_i = ref >> 4; tri_end = tri_i + (ref & 15) + 1; }
                    } else { src.store(slot, found, best); state = ST_IDLE; }
                }
            }
        }
    }
}

// second pass: the deferred rays, one lane each, reference-shaped walk.  If more rays were deferred than the side list
// holds (pathological input), every slot is scanned and the rays that failed fast_path_ok() are recognised again.
template <class Src>
__device__ __forceinline__ void trace_slow_list(const DScene& S, const Src& src, const TraceQueue* queue, const long long* __restrict__ slow_list,
                                                unsigned int slow_cap, Work& w)
{
    const unsigned int n = queue->slow_count;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n <= slow_cap) {
        for (long long i = first; i < n; i += stride) {
            const long long q = slow_list[i];
            Ray r;
            if (!src.fetch(q, r)) continue;
            Hit h;
            const bool ok = trace_closest(S, r, h, w);
            src.store(q, ok, h);
        }
    } else {
        const long long total = src.total();
        for (long long q = first; q < total; q += stride) {
            Ray r;
            if (!src.fetch(q, r) || fast_path_ok(S.fast, r)) continue;
            Hit h;
            const bool ok = trace_closest(S, r, h, w);
            src.store(q, ok, h);
        }
    }
}

}  // namespace mcpt
