// Persistent-threads closest-hit engine: the fast walk of trace_fast.hpp reorganised so that a wavefront keeps
// its 64 lanes busy although rays take very different numbers of steps.
//
//   * Every wave owns a private range of ray slots (a chunk claimed with ONE atomic on the queue head; a single word
//     saturates near 88 dequeues/us on this chip, so claims are per chunk, never per ray).  The head counts tickets, not
//     slots: the first seven eighths of the slots go out in large chunks, the rest in chunks of MCPT_TAIL_CHUNK, so that
//     the waves run dry within one small chunk of each other (with 2048-slot chunks to the end, the last waves of a
//     6-ms launch worked alone for most of a millisecond: 2.5 % of the frame).  Lanes that finish a ray
//     take the next slots of the range: the refill is a ballot + prefix count, no memory traffic.
//   * Each lane is a small state machine: IDLE, INNER (about to test the two children of an inner node),
//     TRI (walking the triangles of a leaf).  Per wave iteration the wave executes ONE phase -- the one most lanes are
//     waiting for (vote over __ballot masks, slightly biased to the cheaper inner step) -- so a phase always runs with at least a third of the lanes
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

#ifndef MCPT_REFILL_LANES
#define MCPT_REFILL_LANES 24        /* refill as soon as this many lanes are idle (sweep: 8: +6 %, 16: +1 %, 32: +1 %) */
#endif
#ifndef MCPT_TAIL_CHUNK
#define MCPT_TAIL_CHUNK 256         /* slots per claim in the last eighth of a launch */
#endif
#ifndef MCPT_TRI_BIAS_NUM
#define MCPT_TRI_BIAS_NUM 3         /* the triangle phase runs when NUM * (lanes waiting for it) > DEN * (lanes waiting for an */
#endif                              /* inner step).  Sweep NUM/4 (ms per frame): 1: 113.2, 2: 110.1, 3: 108.7, 4 (plain majority): */
                                    /* 110.5, 5: 111.4, 6: 112.2, 8: 115.1 -- the cheaper phase may run with a few lanes less */
#ifndef MCPT_TRI_BIAS_DEN
#define MCPT_TRI_BIAS_DEN 4
#endif
#ifndef MCPT_INNER_BURST
#define MCPT_INNER_BURST 1          /* inner-node steps per scheduling vote */
#endif
#ifndef MCPT_SLIM_STATE
#define MCPT_SLIM_STATE 1           /* 1: a register diet for a fourth wave per SIMD -- the next 64 slots are fetched when the LDS batch runs
                                       out (not a batch ahead, in registers), and of 1/d only the x component is kept (the other two are
                                       recomputed where a ray is finished) */
#endif
#ifndef MCPT_TRI_SHARE
#define MCPT_TRI_SHARE 0            /* 1: the triangle phase hands the pending (ray, triangle) pairs of all its lanes out to all 64 lanes
                                       (a lane with a 4-triangle leaf gets three helpers) instead of every lane walking its own leaf one
                                       triangle per iteration at ~28 of 64 lanes.  Measured, not kept: triangle iterations 79.5 M -> 51.0 M
                                       per frame (all iterations -16 %), but the ~45 cross-lane moves per iteration and 24 spilled
                                       registers make one such iteration 1.75x as long: 7.75 instead of 6.27 ms per launch. */
#endif
#ifndef MCPT_POP_CULL
#define MCPT_POP_CULL 0             /* 1: every stack entry carries a lower bound of its entry distance (16 bits: the upper half of the
                                       fp32 bound, i.e. rounded down); an entry popped after the ray's limit has moved in front of it is
                                       dropped instead of visited */
#endif
#ifndef MCPT_INBAND_INPLACE
#define MCPT_INBAND_INPLACE 3       /* how two candidates the products cannot rank are told apart: 1 = by the reference's own t_k and leaf index on
                                       the spot; 2 = the contender is remembered and the two are ranked where the ray is finished (fewer
                                       registers in the triangle phase; a second contender sends the ray to the exact walk); 3 = by t_k and leaf
                                       index on the spot without the newcomer's own-box test (the leader's is checked at the end anyway);
                                       0 = the ray goes to the exact walk.  At 4 waves per SIMD (128 VGPRs) every register in the triangle
                                       phase counts: mode 1 spills 20 registers, mode 2 27; mode 3 with two spilled registers was 11 % slower
                                       per launch than mode 0 (whose deferred rays, 0.05 % on cornell-box, cost 0.13 ms per launch), with one
                                       (after the per-lane rank mask went: v_mbcnt) it is 2.5 % faster: 7.37 vs 7.56 ms */
#endif
#ifndef MCPT_LAZY_VERIFY
#define MCPT_LAZY_VERIFY 1          /* 1: a triangle whose test passes only has the RANK of its distance looked at (two multiplies);
                                       the own-box test and the division of t_k are done once per ray, for the winner, when the
                                       results of a batch of finished rays are stored (>= MCPT_REFILL_LANES lanes at once).  Done per
                                       passing triangle they ran for ~7 lanes of 64 and were a third of the kernel's instructions. */
#endif

// Src must provide:  long long total() const;
//                    bool fetch(long long q, Ray& r) const;        // false: slot holds no ray (loads may be speculative)
//                    void store(long long q, bool hit, const Hit& h) const;
//
// Ray supply is double-buffered so that its memory latency never stalls the lanes that are walking:
//   stage 1: the next 64 slots of the wave's chunk are loaded into registers (one slot per lane, issued early, not waited for);
//   stage 2: when the LDS batch is used up, stage 1 is written to LDS ([component][lane], 52 B per ray) and the following
//            64 slots are requested at once; idle lanes take entries of the LDS batch by ballot rank.
#define MCPT_RAYBUF_DOUBLES 6
#if MCPT_TRI_SHARE
#define MCPT_RAYBUF_BYTES (64 * (MCPT_RAYBUF_DOUBLES * 8 + 4 + 4))  /* per wave: 64 rays, their flags, and the owner table of MCPT_TRI_SHARE */
#else
#define MCPT_RAYBUF_BYTES (64 * (MCPT_RAYBUF_DOUBLES * 8 + 4))      /* per wave: 64 rays and their flags */
#endif

template <class Src>
__device__ __forceinline__ void trace_persistent(const DScene& S, const Src& src, TraceQueue* queue, long long* __restrict__ slow_list,
                                                 unsigned int slow_cap, long long chunk, int* __restrict__ stack, int stride,
                                                 double* __restrict__ raybuf /* this wave's MCPT_RAYBUF_BYTES of LDS */, Work& w,
                                                 unsigned short* __restrict__ kstack = nullptr /* MCPT_POP_CULL: [depth][lane] like stack */,
                                                 int stack_cap = MCPT_FAST_STACK /* entries of `stack` per lane */)
{
    const DFast& F = S.fast;
    const CwNode* __restrict__ nodes = F.cw;
    const DTri* __restrict__ tris = F.tris;
    const long long total = src.total();
    // ticket k < big_tickets: slots [k * chunk, (k + 1) * chunk); later tickets: MCPT_TAIL_CHUNK slots each
    const long long small = chunk < MCPT_TAIL_CHUNK ? chunk : MCPT_TAIL_CHUNK;
    const long long big_tickets = (total - total / 8) / chunk;
    const int lane = threadIdx.x & 63;
    int* __restrict__ rayflag = reinterpret_cast<int*>(raybuf + MCPT_RAYBUF_DOUBLES * 64);
    int* __restrict__ owner_of = rayflag + 64;      // MCPT_TRI_SHARE: pair slot -> owning lane | (triangle offset << 8)
    (void)owner_of;

    // wave-uniform supply state
    long long next = 0, range_end = 0;          // unclaimed part of the wave's chunk
    bool queue_empty = false;                   // no more slots anywhere
    long long reg_base = 0; int reg_count = 0;  // stage 1: slots [reg_base, reg_base+reg_count) in flight / in registers
    long long lds_base = 0; int lds_count = 0, lds_taken = 0;   // stage 2
#if !MCPT_SLIM_STATE
    Ray reg_ray; reg_ray.o = mk(0, 0, 0); reg_ray.d = mk(1, 1, 1);
    bool reg_valid = false;
#endif

    // lane state
    enum { ST_IDLE = 0, ST_INNER = 1, ST_TRI = 2 };
    int state = ST_IDLE;
    long long slot = -1;
    Ray r; r.o = mk(0, 0, 0); r.d = mk(1, 1, 1);
#if MCPT_SLIM_STATE
    double rcp_x = 1;
#define MCPT_RCP_X rcp_x
#define MCPT_FULL_RCP() mk(rcp_x, fast_rcp(r.d.y), fast_rcp(r.d.z))
#else
    V3 rcp = mk(1, 1, 1);
#define MCPT_RCP_X rcp.x
#define MCPT_FULL_RCP() rcp
#endif
    RayF rf; for (int a = 0; a < 3; a++) { rf.o[a] = 0; rf.r[a] = 1; rf.pad[a] = 0; }
    float limit_f = 0;
    double margin = 0, limit = 0;
    bool found = false;
    Hit best; best.leaf = -1; best.t = 0; best.p = mk(0, 0, 0);
#if MCPT_LAZY_VERIFY
    // best.t holds the product (p.x - o.x) * (1 / d.x) of the leading candidate (within 2^-50 of its t_k), best.leaf its slot in
    // the fast triangle array; the candidate is verified (own box) and t_k divided out in finish_ray().
    bool ambiguous = false;             // two candidates closer than the products can tell apart: the ray goes to the exact walk
#if MCPT_INBAND_INPLACE == 2
    int alt = -1;                       // a contender within the products' resolution of the leader (slot in the fast triangle array)
#endif
    bool solo = false;                  // MCPT_TRI_SHARE: this lane walks the rest of its leaf itself (a near-tie needs the exact comparison)
    (void)solo;
#endif
    int sp = 0, cur = 0, tri_i = 0, tri_end = 0;

    // A finished ray's result goes to memory (called for a batch of idle lanes at a time).
    auto finish_ray = [&]() {
#if MCPT_LAZY_VERIFY
        // The leading candidate was chosen by rank alone.  It is the reference's answer iff its own box passes the reference's
        // slab test (a candidate that fails it must not have displaced anything) and no other candidate came within the
        // resolution of the products; otherwise the ray is re-walked exactly (reference-shaped walk, second launch).
        Hit h; h.leaf = -1; h.t = 0; h.p = best.p;
        if (found) {
            const V3 rc = MCPT_FULL_RCP();
            const DTri* tr = tris + best.leaf;
            if (!own_box_hit(tr, r, rc)) ambiguous = true;
            h.leaf = tr->leaf;
            h.t = (best.p.x - r.o.x) / r.d.x;                   // pathTracing.cpp:347
#if MCPT_INBAND_INPLACE == 2
            if (alt >= 0 && !ambiguous) {
                // the contender: its hit point again (the same arithmetic gives the same bits), its own box, then the reference's order
                const DTri* ta_ = tris + alt;
                V3 pa;
                if (tri_hit(ta_, r, pa) && own_box_hit(ta_, r, rc)) {
                    const double t_alt = (pa.x - r.o.x) / r.d.x;
                    if (t_alt > 0 && (t_alt < h.t || (t_alt == h.t && ta_->leaf < h.leaf))) { h.leaf = ta_->leaf; h.t = t_alt; h.p = pa; }
                }
            }
#endif
        }
        if (ambiguous) {
            const unsigned int at = atomicAdd(&queue->slow_count, 1u);
            if (at < slow_cap) slow_list[at] = slot;
            else queue->redo_all = 1u;                          // list full: the second pass re-walks every slot
        } else src.store(slot, found, h);
#else
        src.store(slot, found, best);
#endif
    };

    // claim the next <= 64 slots and issue their loads (stage 1)
    auto request = [&]() {
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
#if !MCPT_SLIM_STATE
        reg_valid = lane < reg_count && src.fetch(reg_base + lane, reg_ray);
#endif
    };
    request();

#ifdef MCPT_TRACE_DIAG
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
    int last_phase = 0;
#define MCPT_STAMP(ph) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); if (lane == 0) w.diag[5 + last_phase] += t_now - t_prev; t_prev = t_now; last_phase = ph; }
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
                if (lds_taken >= lds_count) {            // stage 1 -> stage 2, and ask for the batch after it
                    if (reg_count == 0) break;           // nothing left anywhere
#if MCPT_SLIM_STATE
                    Ray reg_ray; reg_ray.o = mk(0, 0, 0); reg_ray.d = mk(1, 1, 1);
                    const bool reg_valid = lane < reg_count && src.fetch(reg_base + lane, reg_ray);
#endif
                    raybuf[0 * 64 + lane] = reg_ray.o.x; raybuf[1 * 64 + lane] = reg_ray.o.y; raybuf[2 * 64 + lane] = reg_ray.o.z;
                    raybuf[3 * 64 + lane] = reg_ray.d.x; raybuf[4 * 64 + lane] = reg_ray.d.y; raybuf[5 * 64 + lane] = reg_ray.d.z;
                    rayflag[lane] = reg_valid ? 1 : 0;
                    lds_base = reg_base; lds_count = reg_count; lds_taken = 0;
                    request();
                }
                const int want = __popcll(idle);
                const int avail = lds_count - lds_taken;
                const int give = want < avail ? want : avail;
                // set bits of `idle` below this lane (v_mbcnt: no per-lane mask register to keep)
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)idle, 0u));
                if (state == ST_IDLE && rank < give) {
                    const int e = lds_taken + rank;
                    if (rayflag[e]) {
                        Ray nr;
                        nr.o = mk(raybuf[0 * 64 + e], raybuf[1 * 64 + e], raybuf[2 * 64 + e]);
                        nr.d = mk(raybuf[3 * 64 + e], raybuf[4 * 64 + e], raybuf[5 * 64 + e]);
                        const long long q = lds_base + e;
                        if (fast_path_ok(F, nr)) {
                            w.rays++;
                            slot = q; r = nr;
#if MCPT_SLIM_STATE
                            const V3 rcp = mk(fast_rcp(r.d.x), fast_rcp(r.d.y), fast_rcp(r.d.z));
                            rcp_x = rcp.x;
#else
                            rcp = mk(fast_rcp(r.d.x), fast_rcp(r.d.y), fast_rcp(r.d.z));
#endif
                            const double rmax = fmax(fmax(fabs(rcp.x), fabs(rcp.y)), fabs(rcp.z));     // = 1 / min|d_k|
                            const double scale = fmax(fmax(F.absmax, fabs(r.o.x)), fmax(fabs(r.o.y), fabs(r.o.z)));
                            margin = rmax <= 1e6 ? 1.0000001e-9 * scale * rmax : __builtin_inf();
                            limit = __builtin_inf(); limit_f = __builtin_inff();
                            rf = make_rayf(F, r, rcp);
                            found = false; best.leaf = -1; best.t = 0; best.p = mk(0, 0, 0);
#if MCPT_LAZY_VERIFY
                            ambiguous = false;
#if MCPT_INBAND_INPLACE == 2
                            alt = -1;
#endif
#endif
                            sp = 0; cur = 0; state = ST_INNER;
                        } else {
                            const unsigned int at = atomicAdd(&queue->slow_count, 1u);
                            if (at < slow_cap) slow_list[at] = q;    // list full: the second pass scans every slot instead
                        }
                    }
                }
                lds_taken += give;
            }
        }
        // ------------------------------------------------------------------ pick the phase most lanes wait for
        const unsigned long long m_inner = __ballot(state == ST_INNER);
        const unsigned long long m_tri = __ballot(state == ST_TRI);
        if (!m_inner && !m_tri) {
            if (lds_taken >= lds_count && reg_count == 0) break;      // no walking lane and no ray left to hand out
            continue;
        }
#ifdef MCPT_TRACE_DIAG
        if (lane == 0) {
            const bool in = __popcll(m_inner) >= __popcll(m_tri);
            w.diag[in ? 0 : 2] += 1; w.diag[in ? 1 : 3] += in ? __popcll(m_inner) : __popcll(m_tri);
            w.diag[4] += 64 - __popcll(m_inner) - __popcll(m_tri);
        }
#endif
        const bool run_inner = m_inner && MCPT_TRI_BIAS_DEN * __popcll(m_inner) >= MCPT_TRI_BIAS_NUM * __popcll(m_tri);
        MCPT_STAMP(run_inner ? 1 : 2)
        if (run_inner) {
            // -------------------------------------------------------------- inner steps (a short burst per vote)
#pragma unroll 1
            for (int burst = 0; burst < MCPT_INNER_BURST; burst++)
            if (state == ST_INNER) {
#if MCPT_LAZY_VERIFY
                // three pushes must fit: a ray whose stack would overflow (the hierarchy is built not to need that) goes to the exact walk
                if (sp > stack_cap - 3) { ambiguous = true; state = ST_IDLE; }
                else {
#else
                {
#endif
                w.nodes++;
                const CwHits h = cw_step(nodes + cur, rf, limit_f);
                // nearest first; the others go on the stack so that the next nearest is on top
#if MCPT_POP_CULL
#define MCPT_KEY16(x) ((unsigned short)(__float_as_uint(fmaxf((x), 0.0f)) >> 16))
                if (h.ref[3] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[3]; kstack[sp * stride] = MCPT_KEY16(h.key[3]); sp++; }
                if (h.ref[2] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[2]; kstack[sp * stride] = MCPT_KEY16(h.key[2]); sp++; }
                if (h.ref[1] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[1]; kstack[sp * stride] = MCPT_KEY16(h.key[1]); sp++; }
                int nxt = h.ref[0];
                if (nxt == MCPT_FAST_EMPTY) {
                    while (sp > 0) {
                        sp--;
                        if (__uint_as_float((unsigned)kstack[sp * stride] << 16) <= limit_f) { nxt = stack[sp * stride]; break; }
                    }
                }
#else
                if (h.ref[3] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[3]; sp++; }
                if (h.ref[2] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[2]; sp++; }
                if (h.ref[1] != MCPT_FAST_EMPTY) { stack[sp * stride] = h.ref[1]; sp++; }
                int nxt = h.ref[0];
                if (nxt == MCPT_FAST_EMPTY && sp > 0) { sp--; nxt = stack[sp * stride]; }
#endif
                if (nxt >= 0) cur = nxt;
                else if (nxt == MCPT_FAST_EMPTY) state = ST_IDLE;                 // result stored at the next refill
                else { const int ref = -1 - nxt; tri_i = ref >> 4; tri_end = tri_i + (ref & 7) + 1; state = ST_TRI; }
                }
            }
        } else {
            // -------------------------------------------------------------- triangles of the current leaves
            // one triangle of this lane's own leaf, the sequential way
            auto test_own_triangle = [&]() {
                const DTri* tr = tris + tri_i;
                tri_i++;
                // Candidate = own box passes AND triangle test passes AND t > 0 -- a conjunction of pure tests, so the
                // order of evaluation is free: the triangle test goes first (about one visited triangle in five passes it),
                // the rest (better_candidate) is evaluated only for those.
                V3 p;
                w.tris++;
                if (tri_hit(tr, r, p)) {
#if MCPT_LAZY_VERIFY
                    // t_k = (p.x - o.x) / d.x is within 2^-50 (relative) of this product and has its sign: two candidates whose
                    // products differ by more than 2^-47 are ranked like their t_k
                    const double ta = (p.x - r.o.x) * MCPT_RCP_X;
                    if (ta > 0.0) {
                        const double band = best.t * 0x1p-47;
                        if (!found || ta < best.t - band) {
#if MCPT_INBAND_INPLACE == 2
                            // a remembered contender lies within one band of the old leader: it stays clearly behind the new leader
                            // only if that one leads by more than four bands
                            if (found && alt >= 0) { if (ta < best.t - 4.0 * band) alt = -1; else ambiguous = true; }
#endif
                            found = true; best.leaf = tri_i - 1; best.t = ta; best.p = p;
                            limit = (ta + ta * 0x1p-47) + margin;
                            limit_f = __double2float_ru(limit);
                        } else if (!(ta > best.t + band)) {
                            // closer to the leader than the products resolve (a shared edge, a face listed twice, two sides of
                            // a sheet): rank the two by the reference's own t_k and leaf index, provided this one is a candidate
                            // at all.  The leader's own box is looked at when the ray is finished, like any leader's.
#if !MCPT_INBAND_INPLACE
                            ambiguous = true;
#elif MCPT_INBAND_INPLACE == 3
                            // the reference's own order of the two, (t_k, k): whether either is a candidate at all (its own box) is not
                            // looked at here -- a non-candidate that takes or keeps the lead here can only be displaced by something
                            // closer still, and if it is still leading when the ray is finished the own-box test there sends the ray
                            // to the exact walk
                            {
                                const double t_new = (p.x - r.o.x) / r.d.x, t_old = (best.p.x - r.o.x) / r.d.x;
                                if (t_new < t_old || (t_new == t_old && tr->leaf < tris[best.leaf].leaf)) {
                                    best.leaf = tri_i - 1; best.t = ta; best.p = p;
                                }
                            }
#elif MCPT_INBAND_INPLACE == 2
                            if (alt < 0) alt = tri_i - 1; else ambiguous = true;
#else
                            if (own_box_hit(tr, r, MCPT_FULL_RCP())) {
                                const double t_new = (p.x - r.o.x) / r.d.x, t_old = (best.p.x - r.o.x) / r.d.x;
                                if (t_new < t_old || (t_new == t_old && tr->leaf < tris[best.leaf].leaf)) {
                                    best.leaf = tri_i - 1; best.t = ta; best.p = p;
                                }
                            }
#endif
                        }
                    }
#else
                    double t; int k;
                    if (better_candidate(tr, r, rcp, p, found, best, t, k)) {
                        found = true; best.leaf = k; best.t = t; best.p = p;
                        limit = t + margin;
                        limit_f = __double2float_ru(limit);
                    }
#endif
                }
            };
#if MCPT_TRI_SHARE && MCPT_LAZY_VERIFY
            {
                // pending pairs of this wave: lane L owns c(L) <= 4 of them; pair g = excl(L) + j is tested by lane g (g < 64)
                int c = (state == ST_TRI && !solo) ? tri_end - tri_i : 0;
                c = c > 4 ? 4 : c;
                int incl = c;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
                const int excl = incl - c;
                const int total = __shfl(incl, 63, 64);
#pragma unroll
                for (int k = 0; k < 4; k++) if (k < c && excl + k < 64) owner_of[excl + k] = lane | (k << 8);
                const bool work = lane < total;
                int ow = lane, kk = 0;
                if (work) { const int e = owner_of[lane]; ow = e & 255; kk = e >> 8; }
                // the owner's ray, as far as the test and the rank of the distance need it
                Ray hr;
                hr.o = mk(__shfl(r.o.x, ow, 64), __shfl(r.o.y, ow, 64), __shfl(r.o.z, ow, 64));
                hr.d = mk(__shfl(r.d.x, ow, 64), __shfl(r.d.y, ow, 64), __shfl(r.d.z, ow, 64));
                const double h_rcpx = __shfl(MCPT_RCP_X, ow, 64), h_best = __shfl(best.t, ow, 64);
                const int h_found = __shfl((int)found, ow, 64);
                const int h_tri = __shfl(tri_i, ow, 64) + kk;
                V3 hp = mk(0, 0, 0);
                double h_ta = __builtin_inf();          // +inf: no candidate from this pair
                if (work) {
                    w.tris++;
                    if (tri_hit(tris + h_tri, hr, hp)) {
                        const double ta = (hp.x - hr.o.x) * h_rcpx;
                        // certainly farther than the owner's leader: not a candidate (same rule as the sequential code)
                        if (ta > 0.0 && !(h_found && ta > h_best + h_best * 0x1p-47)) h_ta = ta;
                    }
                }
                // owners collect: the smallest product among their pairs, and whether anything comes within the products' resolution
                // of it (another pair, or the current leader) -- then the lane walks this leaf itself, with the exact comparison
                const int done = c < 64 - excl ? c : (64 - excl > 0 ? 64 - excl : 0);      // pairs of this lane that were tested
                double m1 = __builtin_inf(), m2 = __builtin_inf();
                int src = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int from = (excl + k) & 63;
                    const double tk = __shfl(h_ta, from, 64);
                    if (k < done) {
                        if (tk < m1) { m2 = m1; m1 = tk; src = from; }
                        else if (tk < m2) m2 = tk;
                    }
                }
                const V3 wp = mk(__shfl(hp.x, src, 64), __shfl(hp.y, src, 64), __shfl(hp.z, src, 64));
                if (done > 0) {
                    const bool any = m1 < __builtin_inf();
                    const bool near_other = m2 <= m1 + m1 * 0x1p-46;                              // (inf <= inf: only when any)
                    const bool near_leader = found && !(m1 < best.t - best.t * 0x1p-47);           // a candidate is never certainly farther
                    if (any && (near_other || near_leader)) solo = true;                          // nothing taken, nothing skipped
                    else {
                        if (any) {
                            found = true; best.leaf = tri_i + ((src - excl) & 63); best.t = m1; best.p = wp;
                            limit = (m1 + m1 * 0x1p-47) + margin;
                            limit_f = __double2float_ru(limit);
                        }
                        tri_i += done;
                    }
                }
                if (__ballot(state == ST_TRI && solo)) { if (state == ST_TRI && solo) test_own_triangle(); }
            }
#else
            if (state == ST_TRI) test_own_triangle();
#endif
            if (state == ST_TRI && tri_i >= tri_end) {   // leaf done: pop
#if MCPT_LAZY_VERIFY
                solo = false;
#endif
#if MCPT_POP_CULL
                int nxt = MCPT_FAST_EMPTY;
                while (sp > 0) {
                    sp--;
                    if (__uint_as_float((unsigned)kstack[sp * stride] << 16) <= limit_f) { nxt = stack[sp * stride]; break; }
                }
                if (nxt >= 0) { cur = nxt; state = ST_INNER; }
                else if (nxt == MCPT_FAST_EMPTY) state = ST_IDLE;
                else { const int ref = -1 - nxt; tri_i = ref >> 4; tri_end = tri_i + (ref & 7) + 1; }
#else
                if (sp > 0) {
                    sp--;
                    const int nxt = stack[sp * stride];
                    if (nxt >= 0) { cur = nxt; state = ST_INNER; }
                    else { const int ref = -1 - nxt; tri_i = ref >> 4; tri_end = tri_i + (ref & 7) + 1; }
                } else state = ST_IDLE;
#endif
            }
        }
    }
    if (slot >= 0) finish_ray();
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
