// Persistent-threads closest-hit engine: the fast walk of trace_fast.hpp reorganised so that a wavefront keeps
// its 64 lanes busy although rays take very different numbers of steps.
//
//   * Every wave owns a private range of ray slots (a chunk claimed with ONE atomic on the queue head; a single word
//     saturates near 88 dequeues/us on this chip, so claims are per chunk, never per ray).  The head counts tickets, not
//     slots: the first seven eighths of the slots go out in large chunks, the rest in chunks of MCPT_TAIL_CHUNK, so that
//     the waves run dry within one small chunk of each other (with 2048-slot chunks to the end, the last waves of a
//     6-ms launch worked alone for most of a millisecond: 2.5 % of the frame).  Lanes that finish a ray
//     take the next slots of the range: the refill is a ballot + prefix count, no memory traffic.
//   * Each lane is a small state machine: IDLE, INNER (about to test the four children of a compressed node), TRI (about to
//     put the triangles of a leaf through the conservative fp32 pre-test), EXACT (holding triangles that survived it and need
//     the reference's own fp64 test).  Per wave iteration the wave executes ONE phase -- the one most lanes are waiting for
//     (vote over __ballot masks, weighted a little toward the cheaper phases) -- so a phase always runs with a good share of
//     the lanes that have work, instead of every phase running for whoever happens to need it.  Round 2 had two phases and
//     ran the ~160-instruction fp64 test on every visited triangle (four in five fail it) at 28-36 of 64 lanes; now a leaf
//     visit is one pre-test pass over its <= 4 triangles (~55 fp32 instructions each, 48-byte records) and only the
//     survivors queue for the exact test.
//   * Per-lane traversal stack in LDS ([depth][lane], conflict free); rays that need the reference-shaped walk
//     (a zero / denormal / non-finite component) are not walked here: their slot goes to a side list that a second,
//     tiny launch handles, so one such ray cannot hold 63 lanes for the length of an exhaustive walk.
// Results are bit-identical to trace_closest_fast() and therefore to the reference (same exact tests on every triangle that
// can pass them, (t, k) lexicographic minimum).
//
// Variants that were built, verified bit-exact, measured slower and removed from this file (they are in the
// history): pair redistribution in the triangle phase, pop-time culling with 16-bit keys, ray supply a batch ahead in registers,
// near-tie modes 0-2, own-box test and division per passing triangle.
#pragma once
#include "trace_fast.hpp"
#include "wavefront.hpp"

namespace mcpt {

#ifndef MCPT_REFILL_LANES
#define MCPT_REFILL_LANES 24        /* refill as soon as this many lanes are idle (sweep: 8: +6 %, 16: +1 %, 32: +1 %) */
#endif
#ifndef MCPT_TAIL_CHUNK
#define MCPT_TAIL_CHUNK 256         /* slots per claim in the last eighth of a launch */
#endif
// Phase vote: the phase with the largest weight x (lanes waiting for it) runs.
#ifndef MCPT_W_INNER
#define MCPT_W_INNER 4
#endif
#ifndef MCPT_W_TRI
#define MCPT_W_TRI 3
#endif
#ifndef MCPT_W_EXACT
#define MCPT_W_EXACT 3
#endif
#ifndef MCPT_LEAF_CLASS
#define MCPT_LEAF_CLASS 1           /* 1: the vote knows two classes -- lanes at a node, lanes at a leaf; a leaf iteration runs the pre-test
                                       for the lanes that enter a leaf and then one exact test for every lane that holds a survivor (its own
                                       fresh ones included).  0: three classes, the exact test as a phase of its own. */
#endif
#ifndef MCPT_EXACT_MIN
#define MCPT_EXACT_MIN 1            /* MCPT_LEAF_CLASS: lanes holding a survivor before a leaf iteration runs its exact block */
#endif
#ifndef MCPT_PRE_UNROLL
#define MCPT_PRE_UNROLL 2           /* triangles per round of the pre-test (their records are requested together) */
#endif
#ifndef MCPT_PRE_TEST
#define MCPT_PRE_TEST 1             /* 0: every visited triangle survives the pre-test (A/B runs; same results) */
#endif

// Src must provide:  long long total() const;
//                    bool fetch(long long q, Ray& r) const;        // false: slot holds no ray (loads may be speculative)
//                    void store(long long q, bool hit, const Hit& h) const;
//                    static constexpr bool kWantsPoint;             // false: store() does not look at h.p
//
// Ray supply: when the wave's LDS batch is used up, the next 64 slots of its chunk are fetched at once (one slot per lane,
// branch-free, so nothing waits on a flag) into LDS ([component][lane], 52 B per ray); idle lanes take entries by ballot rank.
#define MCPT_RAYBUF_DOUBLES 6
#define MCPT_RAYBUF_BYTES (64 * (MCPT_RAYBUF_DOUBLES * 8 + 4))      /* per wave: 64 rays and their flags */

template <class Src>
__device__ __forceinline__ void trace_persistent(const DScene& S, const Src& src, TraceQueue* queue, long long* __restrict__ slow_list,
                                                 unsigned int slow_cap, long long chunk, int* __restrict__ stack, int stride,
                                                 double* __restrict__ raybuf /* this wave's MCPT_RAYBUF_BYTES of LDS */, Work& w,
                                                 int stack_cap = MCPT_FAST_STACK /* entries of `stack` per lane */)
{
    const DFast& F = S.fast;
    const CwNode* __restrict__ nodes = F.cw;
    const DTri* __restrict__ tris = F.tris;
    const DTriPre* __restrict__ pre = F.pre;
    const long long total = src.total();
    // ticket k < big_tickets: slots [k * chunk, (k + 1) * chunk); later tickets: MCPT_TAIL_CHUNK slots each
    const long long small = chunk < MCPT_TAIL_CHUNK ? chunk : MCPT_TAIL_CHUNK;
    const long long big_tickets = (total - total / 8) / chunk;
    const int lane = threadIdx.x & 63;
    // work counters that are the same for every lane of a phase: wave-uniform, in scalar registers (per-lane they would cost three
    // VGPRs the walk does not have); wctr: a few LDS words behind the rays for the rare per-lane corrections
    unsigned int c_nodes = 0, c_rays = 0, c_exact = 0;
    unsigned int* __restrict__ wctr = reinterpret_cast<unsigned int*>(raybuf + MCPT_RAYBUF_DOUBLES * 64);
    if (lane < 4) wctr[lane] = 0u;        // [0] inner steps refused (stack full), [3] MCPT_PRE_CHECK

    // wave-uniform supply state
    long long next = 0, range_end = 0;          // unclaimed part of the wave's chunk
    bool queue_empty = false;                   // no more slots anywhere
    long long reg_base = 0; int reg_count = 0;  // the next <= 64 slots to fetch
    long long lds_base = 0; int lds_count = 0, lds_taken = 0;   // the batch in LDS
    unsigned long long lds_valid = 0;           // bit e: entry e of the batch holds a ray

    // lane state
    enum { ST_IDLE = 0, ST_INNER = 1, ST_TRI = 2, ST_EXACT = 3 };
    int state = ST_IDLE;
    long long slot = -1;
    Ray r; r.o = mk(0, 0, 0); r.d = mk(1, 1, 1);
    RayF rf; for (int a = 0; a < 3; a++) { rf.o[a] = 0; rf.r[a] = 1; rf.pad[a] = 0; }
    float limit_f = 0;                          // upper bound (rounded up) of leader's product x (1 + 2^-47) + margin
    float margin_f = 0;                         // the pruning margin of trace_fast.hpp, rounded up (+inf: no pruning by distance)
    bool found = false;
    // The leading candidate: best_leaf = its slot in the fast triangle array, best_t = the product (p.x - o.x) * (1 / d.x) (within
    // 2^-50 of its t_k), best_px = its hit point's x (what the reference's t_k is divided out of).  The candidate is verified (own
    // box), t_k divided out and the hit point formed again in finish_ray(): three registers per lane instead of nine through the
    // whole walk, for ~45 instructions per ray.
    int best_leaf = -1;
    double best_t = 0, best_px = 0;
    bool ambiguous = false;             // the ray goes to the exact one-lane walk (stack overflow, leader's own box fails)
    int sp = 0;
    int cur = 0;                        // ST_INNER: the node to step on; ST_TRI / ST_EXACT: first slot of the leaf's triangles
    int tri_m = 0;                      // ST_TRI: number of triangles of the leaf; ST_EXACT: bit k set = triangle cur + k awaits the exact test

    // A finished ray's result goes to memory (called for a batch of idle lanes at a time).
    auto finish_ray = [&]() __attribute__((always_inline)) {
        // The leading candidate was chosen by rank alone.  It is the reference's answer iff its own box passes the reference's
        // slab test (a candidate that fails it must not have displaced anything); otherwise the ray is re-walked exactly.
        Hit h; h.leaf = -1; h.t = 0; h.p = mk(0, 0, 0);
        if (found) {
            const V3 rc = mk(fast_rcp(r.d.x), fast_rcp(r.d.y), fast_rcp(r.d.z));
            const DTri* tr = tris + best_leaf;
            if (!own_box_hit(tr, r, rc)) ambiguous = true;
            h.leaf = tr->leaf;
            h.mat = tr->material;
            h.t = (best_px - r.o.x) / r.d.x;                    // pathTracing.cpp:347
            // the hit point, for sources that store it: the first two lines of intersect(Ray&, Face&, Vertex&) again -- the same
            // operations on the same operands give the same bits as when the triangle was tested
            if constexpr (Src::kWantsPoint) {
                const V3 v1 = ld3(tr->v1), n = ld3(tr->n);
                const double t = dot(v1 - r.o, n) / dot(n, r.d);
                h.p = r.o + r.d * t;
            }
        }
        if (ambiguous) {
            const unsigned int at = atomicAdd(&queue->slow_count, 1u);
            if (at < slow_cap) slow_list[at] = slot;
            else queue->redo_all = 1u;                          // list full: the second pass re-walks every slot
        } else src.store(slot, found, h);
    };

    // claim the next <= 64 slots
    auto request = [&]() __attribute__((always_inline)) {
        reg_count = 0;
        if (queue_empty) return;
        if (next >= range_end) {
            unsigned long long got = 0;
            if (lane == 0) got = atomicAdd(&queue->head, 1ull);
            got = __shfl(got, 0, 64);
            const long long ticket = (long long)got;
            const long long size = ticket < big_tickets ? chunk : small;
            next = ticket < big_tickets ? ticket * chunk : big_tickets * chunk + (ticket - big_tickets) * small;
            range_end = next + size < total ? next + size : total;
            if (next >= total) { queue_empty = true; return; }
        }
        const long long avail = range_end - next;
        reg_count = avail < 64 ? (int)avail : 64;
        reg_base = next;
        next += reg_count;
    };
    request();

    // the next entry of this lane's stack becomes its work; nothing left: the ray is finished (its result is stored at the next refill)
    auto pop_next = [&]() __attribute__((always_inline)) {
        // (selects, not branches that assign different variables: the compiler merges such stores into one through a computed
        // address, which pins the variables to scratch memory)
        if (sp > 0) {
            sp--;
            const int nxt = stack[sp * stride];
            const bool node = nxt >= 0;
            const int ref = -1 - nxt;
            cur = node ? nxt : ref >> 4;
            tri_m = node ? tri_m : (ref & 7) + 1;
            state = node ? ST_INNER : ST_TRI;
        } else state = ST_IDLE;
    };

#ifdef MCPT_TRACE_DIAG
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
    int last_phase = 0;
#define MCPT_STAMP(ph) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); if (lane == 0) w.diag[8 + last_phase] += t_now - t_prev; t_prev = t_now; last_phase = ph; }
#else
#define MCPT_STAMP(ph)
#endif
    for (;;) {
        // ------------------------------------------------------------------ refill idle lanes
        MCPT_STAMP(0)
        unsigned long long idle = __ballot(state == ST_IDLE);
        const bool supply = lds_taken < lds_count || reg_count > 0;
        if (idle && supply && (__popcll(idle) >= MCPT_REFILL_LANES || idle == ~0ull)) {
            // results of the rays that ended since the last refill go out in one batch: a store between two node loads
            // would put its acknowledge time on the walking lanes' critical path (loads and stores share one counter)
            if (state == ST_IDLE && slot >= 0) { finish_ray(); slot = -1; }
            for (;;) {
                idle = __ballot(state == ST_IDLE);
                if (!idle) break;
                if (lds_taken >= lds_count) {            // the next batch into LDS
                    if (reg_count == 0) break;           // nothing left anywhere
                    Ray reg_ray; reg_ray.o = mk(0, 0, 0); reg_ray.d = mk(1, 1, 1);
                    const bool reg_valid = lane < reg_count && src.fetch(reg_base + lane, reg_ray);
                    raybuf[0 * 64 + lane] = reg_ray.o.x; raybuf[1 * 64 + lane] = reg_ray.o.y; raybuf[2 * 64 + lane] = reg_ray.o.z;
                    raybuf[3 * 64 + lane] = reg_ray.d.x; raybuf[4 * 64 + lane] = reg_ray.d.y; raybuf[5 * 64 + lane] = reg_ray.d.z;
                    lds_valid = __ballot(reg_valid);
                    lds_base = reg_base; lds_count = reg_count; lds_taken = 0;
                    request();
                }
                const int want = __popcll(idle);
                const int avail = lds_count - lds_taken;
                const int give = want < avail ? want : avail;
                // set bits of `idle` below this lane (v_mbcnt: no per-lane mask register to keep)
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)idle, 0u));
                bool started = false;
                if (state == ST_IDLE && rank < give) {
                    const int e = lds_taken + rank;
                    if ((lds_valid >> e) & 1ull) {
                        Ray nr;
                        nr.o = mk(raybuf[0 * 64 + e], raybuf[1 * 64 + e], raybuf[2 * 64 + e]);
                        nr.d = mk(raybuf[3 * 64 + e], raybuf[4 * 64 + e], raybuf[5 * 64 + e]);
                        const long long q = lds_base + e;
                        if (fast_path_ok(F, nr)) {
                            slot = q; r = nr;
                            const V3 rcp = mk(fast_rcp(r.d.x), fast_rcp(r.d.y), fast_rcp(r.d.z));
                            const double rmax = fmax(fmax(fabs(rcp.x), fabs(rcp.y)), fabs(rcp.z));     // = 1 / min|d_k|
                            const double scale = fmax(fmax(F.absmax, fabs(r.o.x)), fmax(fabs(r.o.y), fabs(r.o.z)));
                            margin_f = rmax <= 1e6 ? __double2float_ru(1.0000001e-9 * scale * rmax) : __builtin_inff();
                            limit_f = __builtin_inff();
                            rf = make_rayf(F, r, rcp);
                            found = false; best_leaf = -1; best_t = 0; best_px = 0;
                            ambiguous = false;
                            sp = 0; cur = 0; state = ST_INNER;
                            started = true;
                        } else {
                            const unsigned int at = atomicAdd(&queue->slow_count, 1u);
                            if (at < slow_cap) slow_list[at] = q;    // list full: the second pass scans every slot instead
                        }
                    }
                }
                lds_taken += give;
                c_rays += (unsigned int)__popcll(__ballot(started));
            }
        }
        // ------------------------------------------------------------------ pick the phase most lanes wait for
        const int n_inner = __popcll(__ballot(state == ST_INNER));
        const int n_tri = __popcll(__ballot(state == ST_TRI));
        const int n_exact = __popcll(__ballot(state == ST_EXACT));
        if (!(n_inner | n_tri | n_exact)) {
            if (lds_taken >= lds_count && reg_count == 0) break;      // no walking lane and no ray left to hand out
            continue;
        }
#if MCPT_LEAF_CLASS
        // two classes: lanes at a node, lanes at a leaf (pre-test or exact stage); the leaf phase runs both of its blocks
        const int phase = (MCPT_W_INNER * n_inner >= MCPT_W_TRI * (n_tri + n_exact)) ? ST_INNER : ST_TRI;
#else
        const int s_inner = MCPT_W_INNER * n_inner, s_tri = MCPT_W_TRI * n_tri, s_exact = MCPT_W_EXACT * n_exact;
        const int phase = (s_inner >= s_tri && s_inner >= s_exact) ? ST_INNER : (s_tri >= s_exact ? ST_TRI : ST_EXACT);
#endif
#ifdef MCPT_TRACE_DIAG
        if (lane == 0) {
            const int k = phase - 1;                                   // 0 inner, 1 tri, 2 exact
            w.diag[2 * k] += 1; w.diag[2 * k + 1] += phase == ST_INNER ? n_inner : (phase == ST_TRI ? n_tri : n_exact);
            w.diag[6] += 64 - n_inner - n_tri - n_exact;
#if MCPT_LEAF_CLASS
            if (phase == ST_TRI) { w.diag[4] += 1; w.diag[5] += n_exact; }      // exact-stage lanes at the start of a leaf iteration
#endif
        }
#endif
        MCPT_STAMP(phase)
        if (phase == ST_INNER) {
            // -------------------------------------------------------------- one step on a compressed node
            // three pushes must fit: a ray whose stack would overflow (the hierarchy is built not to need that) goes to the exact walk
            c_nodes += (unsigned int)n_inner;
            if (state == ST_INNER) {
                if (sp > stack_cap - 3) { ambiguous = true; state = ST_IDLE; atomicAdd(&wctr[0], 1u); }       // (not a step: taken off the count)
                else {
                    const CwHits h = cw_step(nodes + cur, rf, limit_f);
                    // nearest first; the others go on the stack so that the next nearest is on top
                    if (h.ref[3] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[3]; sp++; }
                    if (h.ref[2] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[2]; sp++; }
                    if (h.ref[1] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[1]; sp++; }
                    int nxt = h.ref[0];
                    if (nxt == MCPT_FAST_EMPTY && sp > 0) { sp--; nxt = stack[sp * stride]; }
                    const bool node = nxt >= 0, none = nxt == MCPT_FAST_EMPTY;
                    const int ref = -1 - nxt;
                    cur = node ? nxt : ref >> 4;
                    tri_m = node ? tri_m : (ref & 7) + 1;
                    state = node ? ST_INNER : (none ? ST_IDLE : ST_TRI);     // (ST_IDLE: the result is stored at the next refill)
                }
            }
        }
#if MCPT_LEAF_CLASS
        if (phase == ST_TRI && n_tri) {
#else
        else if (phase == ST_TRI) {
#endif
            // -------------------------------------------------------------- the triangles of a leaf through the fp32 pre-test
            if (state == ST_TRI) {
                unsigned int surv = 0;
#if MCPT_PRE_TEST
                w.tris += tri_m;
                if (!pre) surv = (1u << tri_m) - 1u;       // scenes for which the pre-test is switched off (capi.cpp: too large for the caches)
                else {
                const PreRay pr = make_pre_ray(F, r, rf.o, margin_f);
                // MCPT_PRE_UNROLL triangles per round: their records are requested together, so a round costs one memory latency
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (int k0 = 0; k0 < tri_m; k0 += MCPT_PRE_UNROLL) {
#pragma unroll
                    for (int j = 0; j < MCPT_PRE_UNROLL; j++) {
                        const int k = k0 + j;
                        // (a slot past the leaf's last is a triangle of the next leaf or the array's padding: tested, not used)
                        const bool rej = tri_pre_reject(pre + cur + k, pr, limit_f);
                        if (k < tri_m && !rej) surv |= 1u << k;
                    }
                }
#ifdef MCPT_PRE_CHECK
                // self-check build: every REJECTED triangle through the exact test as well; one that passes it with a positive t_k not
                // behind the leader should have survived (counted, and made to survive)
                for (int k = 0; k < tri_m; k++) {
                    if ((surv >> k) & 1u) continue;
                    V3 pc;
                    if (tri_hit(tris + cur + k, r, pc)) {
                        const double tc = (pc.x - r.o.x) / r.d.x;
                        if (tc > 0.0 && (!found || tc <= best_t * (1.0 + 0x1p-40))) {
                            atomicAdd(&wctr[3], 1u); surv |= 1u << k;
                            if (w.dbg && atomicCAS(w.dbg, 0ull, 1ull) == 0ull) {      // the first one: what the pre-test saw
                                float h[8];
                                (void)tri_pre_reject(pre + cur + k, pr, limit_f, h);
                                double* o = reinterpret_cast<double*>(w.dbg);
                                for (int i = 0; i < 8; i++) o[1 + i] = h[i];
                                o[9] = tc; o[10] = found ? best_t : -1.0; o[11] = limit_f; o[12] = pr.margin; o[13] = pr.eta4; o[14] = cur + k; o[15] = tri_m;
                                o[16] = r.o.x; o[17] = r.o.y; o[18] = r.o.z; o[19] = r.d.x; o[20] = r.d.y; o[21] = r.d.z;
                            }
                        }
                    }
                }
#endif
                }
#else
                w.tris += tri_m;
                surv = (1u << tri_m) - 1u;
#endif
                if (surv) { tri_m = (int)surv; state = ST_EXACT; }
                else pop_next();
            }
        }
#if MCPT_LEAF_CLASS
        // the exact block runs once enough lanes hold a survivor (or nothing else is waiting at a leaf): below that they wait, as
        // members of the leaf class, for the next leaf iteration
        const int n_hold = phase == ST_TRI ? __popcll(__ballot(state == ST_EXACT)) : 0;
        if (n_hold >= MCPT_EXACT_MIN || (n_hold && (!n_tri || n_hold >= n_tri + n_exact))) {
            c_exact += (unsigned int)n_hold;
#else
        else if (phase == ST_EXACT) {
            c_exact += (unsigned int)n_exact;
#endif
            // -------------------------------------------------------------- one surviving triangle through the reference's test
            if (state == ST_EXACT) {
                const int k = __ffs(tri_m) - 1;
                tri_m &= tri_m - 1;
                const int ti = cur + k;
                const DTri* tr = tris + ti;
                // Candidate = own box passes AND triangle test passes AND t > 0 -- a conjunction of pure tests, so the
                // order of evaluation is free: the triangle test goes first, the own box is looked at once per ray (finish_ray).
                V3 p;
                if (tri_hit(tr, r, p)) {
                    // t_k = (p.x - o.x) / d.x is within 2^-50 (relative) of this product and has its sign: two candidates whose
                    // products differ by more than 2^-47 are ranked like their t_k
                    const double ta = (p.x - r.o.x) * fast_rcp(r.d.x);
                    if (ta > 0.0) {
                        const double band = best_t * 0x1p-47;
                        if (!found || ta < best_t - band) {
                            found = true; best_leaf = ti; best_t = ta; best_px = p.x;
                            limit_f = __double2float_ru((ta + ta * 0x1p-47) + (double)margin_f);
                        } else if (!(ta > best_t + band)) {
                            // closer to the leader than the products resolve (a shared edge, a face listed twice, two sides of
                            // a sheet): the reference's own order of the two, (t_k, k).  Whether either is a candidate at all (its
                            // own box) is not looked at here -- a non-candidate that takes or keeps the lead here can only be
                            // displaced by something closer still, and if it is still leading when the ray is finished the own-box
                            // test there sends the ray to the exact walk
                            const double t_new = (p.x - r.o.x) / r.d.x, t_old = (best_px - r.o.x) / r.d.x;
                            if (t_new < t_old || (t_new == t_old && tr->leaf < tris[best_leaf].leaf)) {
                                best_leaf = ti; best_t = ta; best_px = p.x;
                            }
                        }
                    }
                }
                if (!tri_m) pop_next();
            }
        }
    }
    if (slot >= 0) finish_ray();
    if (lane == 0) { w.nodes += c_nodes - wctr[0]; w.rays += c_rays; w.exact += c_exact; w.pre_wrong += wctr[3]; }     // (summed over the wave by the caller)
}

// second pass: the deferred rays, one lane each, reference-shaped walk.  If more rays were deferred than the side list
// holds (pathological input), every slot is scanned and the rays that failed fast_path_ok() are recognised again.
// A deferred ray is either one the fast walk may not take at all (fast_path_ok fails: reference-shaped walk) or one whose candidates
// the engine could not rank from products alone (a near-tie, a stack overflow): that one is walked again on the fast hierarchy, one
// lane per ray, with every candidate decided exactly on the spot (trace_lane_fast) -- a few times the cost of an ordinary ray, where
// the exhaustive walk costs ~1 300 node visits.  stack: this lane's LDS words (stride 256), or nullptr (then always the exhaustive walk).
__device__ __forceinline__ bool trace_deferred(const DScene& S, const Ray& r, Hit& h, Work& w, int* __restrict__ stack)
{
    if (stack && fast_path_ok(S.fast, r)) return trace_lane_fast(S, r, h, w, stack, 256);
    return trace_closest(S, r, h, w);
}

template <class Src>
__device__ __forceinline__ void trace_slow_list(const DScene& S, const Src& src, const TraceQueue* queue, const long long* __restrict__ slow_list,
                                                unsigned int slow_cap, Work& w, int* __restrict__ stack = nullptr)
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
            const bool ok = trace_deferred(S, r, h, w, stack);
            src.store(q, ok, h);
        }
    } else {
        const long long total = src.total();
        const bool redo_all = queue->redo_all != 0;        // a ray the fast walk could not decide did not fit into the list
        for (long long q = first; q < total; q += stride) {
            Ray r;
            if (!src.fetch(q, r) || (!redo_all && fast_path_ok(S.fast, r))) continue;
            Hit h;
            const bool ok = trace_deferred(S, r, h, w, stack);
            src.store(q, ok, h);
        }
    }
}

}  // namespace mcpt
