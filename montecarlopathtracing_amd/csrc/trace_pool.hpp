// Closest-hit engine with the rays of a workgroup resident in LDS ("pool" engine): the persistent engine of trace_persistent.hpp keeps
// one ray per lane in registers and lets the wave vote for a phase, so a phase runs with the lanes that happen to wait for it (58 % of
// the lanes on cornell-box, 26 % in the exact block).  Here a workgroup of MCPT_POOL_WAVES waves owns 64 x KT ray slots whose whole
// state -- ray, culling constants, leader, traversal stack -- lives in LDS, and a wave step is stateless: the wave picks a class
// (node step / leaf pre-test / exact triangle test / finish + refill), every lane claims one slot of that class, loads what the step
// needs, runs it, writes back what changed and files the slot under its next class.
//
//   * Slot s = k * 64 + lane is only ever handled by lane `lane` (of any wave): every LDS access of a step is [field][k][lane], i.e.
//     conflict free, and there is no queue between waves -- per lane and class one 32-bit LDS word holds the class's set
//     (bit k of word [c][lane]: slot k * 64 + lane waits for a step of class c).  Claim = atomic AND that clears the bit (the caller owns
//     the slot iff the bit was set in the value returned), filing = atomic OR.  A claimed slot is in no set, so nothing else touches it.
//     (Rounds 3 and 4a packed two classes into a 64-bit word: the vote, the claim and the filing were 64-bit shifts, masks and compares
//     on every lane -- a quarter of the engine's vector instructions went into choosing what to do.)
//     No wave ever waits for another one: no barrier after the start, no spinning on data (the only sleep is "nothing claimable now").
//   * A lane has work in class c if any of its KT slots (not held by another wave) waits for c: with a few slots per class that is
//     nearly always, which is where the lanes per instruction come from -- so a slot is kept small (120 bytes: fp64 ray, 1/d as
//     floats, limit, leader, eight stack entries): every slot more per lane is worth one to two lanes per step.
//   * Ordering between the wave that leaves a slot and the wave that takes it: the OR that files the slot is a release, the AND that
//     claims it an acquire (workgroup scope) -- everything the step wrote, in LDS or in the global spill area, happens-before
//     everything the next step reads (MCPT_POOL_ORDER).
//   * Only the first MCPT_POOL_STACK entries of a slot's traversal stack are in LDS; the deeper ones (a tenth of the rays get there)
//     live in a global spill area behind the launch's deferred-ray list, [block][entry][slot].  A ray that would pass the walk's
//     stack_cap goes to the deferred list (one-lane walk), like in the persistent engine.
// The decisions are the persistent engine's, test for test (same cw_step, tri_pre_reject, tri_hit, ranking, own-box check at the end),
// so results are bit-identical to it and to the reference-shaped walk.
//
// Path mode (template parameter PP with kPaths = true; what k_wf_finish_pool runs): the same workgroup also OWNS the paths its rays belong
// to.  A lane's KT ray slots are dealt to KT / (lights + 1) path slots -- one slot per shadow ray and one for the bounce ray -- and a
// fifth class, SHADE, takes a path whose rays have all come back: resolve the vertex (visibility -> direct light, bounce hit -> next
// vertex), shade the next one with the logic kernel's own functions (vertex.hpp), write its rays straight into the path's ray slots
// and file them under the node class; a path that ends hands its slot to the next path of the wavefront state (adoption: its pending
// rays are read from the state k_wf_logic left).  No ray ever goes through memory, the result step neither fetches nor stores (it notes
// leaf / material in the slot and clears the ray's bit in the lane's busy word; whoever clears the last bit of a path files it under
// SHADE), and every path advances at its own pace: the one-lane-per-path finishing kernel this replaces ran 64 paths in lock-step, a
// step as long as its slowest walk.
#pragma once
#include "trace_persistent.hpp"
#include "vertex.hpp"

namespace mcpt {

struct NoPaths { static constexpr bool kPaths = false; };

#ifndef MCPT_POOL_WAVES
#define MCPT_POOL_WAVES 16
#endif
#ifndef MCPT_POOL_KT
#define MCPT_POOL_KT 20             /* ray slots per lane (<= 32: one 32-bit set per class) */
#endif
#ifndef MCPT_POOL_STACK
#define MCPT_POOL_STACK 8           /* stack entries per slot in LDS */
#endif
#define MCPT_POOL_SPILL (MCPT_FAST_STACK - MCPT_POOL_STACK)      /* ... and beyond them in global memory (few rays go that deep) */
// vote: the class with the largest weight x (lanes that can claim a slot of it) runs; ties go downstream (finish > exact > leaf > node)
#ifndef MCPT_POOL_CLAIMS
#define MCPT_POOL_CLAIMS 2             /* attempts of a lane to claim a slot in one step */
#endif
/* (a claim that takes every slot of the class the word shows and hands back all but one was measured in round 3: 85.0 vs 81.7 ms, the
   hidden slots starve the other waves) */
#ifndef MCPT_POOL_PREFETCH
#define MCPT_POOL_PREFETCH 0
#endif
#ifndef MCPT_POOL_CACHE_N
#define MCPT_POOL_CACHE_N 0         /* nodes of the top of the tree held in LDS; 52 is what the 160 KB leave beside the ray slots -- measured: 7.11 ms per trace
                                       launch with them, 7.07 without (round 4: the kernel is bound by vector instruction issue, not by its gathers) */
#endif
#ifndef MCPT_POOL_STICKY
#define MCPT_POOL_STICKY 0          /* a lane whose slot stays at a node keeps it for the wave's next node step (no filing, no claim) */
#endif
#ifndef MCPT_POOL_PREF
#define MCPT_POOL_PREF 0            /* > 0: every wave has a class it prefers (its score counts (4 + PREF) / 4): simultaneous voters spread out */
#endif
// Hand-over of a slot from one wave to another.  What a step writes (LDS state, and stack entries beyond the eighth: plain stores to
// the global spill area) must be visible to the wave that claims the slot next.  1 (default): the filing OR is a RELEASE and the claiming
// AND an ACQUIRE at workgroup scope -- the memory model's own guarantee, for LDS and for the spill area alike.  0: relaxed atomics
// between compiler barriers, which leans on the LDS unit executing a wave's instructions in order and says nothing about the spill
// stores (they travel through the vector memory path, counted by vmcnt, not lgkmcnt) -- what rounds 3 ran; kept for A/B runs.
#ifndef MCPT_POOL_ORDER
#define MCPT_POOL_ORDER 1
#endif
#if MCPT_POOL_ORDER
#define MCPT_POOL_CLAIM_ORDER __ATOMIC_ACQUIRE
#define MCPT_POOL_FILE_ORDER __ATOMIC_RELEASE
#else
#define MCPT_POOL_CLAIM_ORDER __ATOMIC_RELAXED
#define MCPT_POOL_FILE_ORDER __ATOMIC_RELAXED
#endif
#ifndef MCPT_POOL_FASTPUSH
#define MCPT_POOL_FASTPUSH 1        /* 0: a branch per pushed child (A/B runs) */
#endif
#ifndef MCPT_PW_INNER
#define MCPT_PW_INNER 4
#endif
#ifndef MCPT_PW_LEAF
#define MCPT_PW_LEAF 4
#endif
#ifndef MCPT_PW_EXACT
#define MCPT_PW_EXACT 4
#endif
#ifndef MCPT_PW_FIN
#define MCPT_PW_FIN 4
#endif
#ifndef MCPT_PW_SHADE
#define MCPT_PW_SHADE 4             /* path mode: weight of the paths waiting for their next vertex */
#endif

struct alignas(16) PoolOxy { double ox, oy; };
struct alignas(16) PoolOzDx { double oz, dx; };
struct alignas(16) PoolDyz { double dy, dz; };
struct alignas(16) PoolRcp { float rx, ry, rz, limit; };    // 1/d as floats; the walk's current limit
// A 16-byte group is read as ONE 16-byte access: left to itself the compiler splits a struct of two doubles into two 8-byte loads and
// merges them again as ds_read2_b64 offset1:1 -- two 8-byte halves at a 16-byte lane stride, which the LDS serves at a quarter of the rate
// of ds_read_b128 (8 bank-conflict cycles per instruction: tools/probes/lds_conflict_probe.hip; that was most of the 22 % of
// SQ_LDS_BANK_CONFLICT in SQ_LDS_IDX_ACTIVE the round-3 profile of this kernel shows).
typedef double pool_d2 __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ T pool_ld16(const T* p)
{
    static_assert(sizeof(T) == 16, "16-byte group");
    const pool_d2 v = *reinterpret_cast<const pool_d2*>(p);
    T r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

// Every array is [k][lane] (a 16-byte group per lane where a step wants the words together, else one word per lane): consecutive lanes
// touch consecutive banks whatever their k, so a 4-byte access of a step has no bank conflict.  (Single words inside 16-byte groups
// were 4-way conflicts: measured, a third of the LDS cycles.)  The 16-byte groups and the 8-byte words are not conflict-free in the
// counter's sense: a 64-lane ds_read_b128 / ds_write_b128 covers 1 KB = eight passes over the 32 banks by construction, and
// SQ_LDS_BANK_CONFLICT counts the passes beyond the first of every instruction (DESIGN section 6 has the account by instruction).
template <int NW, int KT, int SCAP>
struct PoolLds {
    PoolOxy oxy[KT * 64];
    PoolOzDx ozdx[KT * 64];
    PoolDyz dyz[KT * 64];
    PoolRcp rcp[KT * 64];
    double best_t[KT * 64];                 // the leader's product (p.x - o.x) * (1 / d.x); its hit point is formed again when the ray is finished
    unsigned int q[KT * 64];                // the ray's slot in the source (launches of 2^32 slots and more take the voting engine)
    int cur[KT * 64];                       // node to step on / first triangle slot of the leaf
    int best_leaf[KT * 64];
    int spf[KT * 64];                       // stack entries (bits 0-7) | flags | leaf: number of triangles, exact class: survivors (bits 16-23)
    int stack[SCAP * KT * 64];              // [entry][k][lane]
    unsigned int cls[5 * 64];               // [class][lane]: bit k = slot k * 64 + lane waits for a step of that class (class 4, path mode: bit k = PATH slot k waits for its next vertex)
    unsigned int busy[64];                  // path mode: bit k = ray slot k of this lane is still walking
    int tbl[NW * 64];                       // refill: rank among the fetched rays -> lane that holds it
    uint4 nodes[MCPT_POOL_CACHE_N ? MCPT_POOL_CACHE_N * 5 : 1];     // the top of the tree, 80 bytes apart (64 of node, 16 unused: lanes on eight consecutive nodes read without a bank conflict)
    unsigned int stat[8];                   // path mode: shade calls, shadow rays, bounce rays, shadow rays skipped, deepest vertex (flushed by the kernel)
    unsigned int live;                      // slots that may still carry a ray (path mode: path slots that may still carry a path)
    unsigned int dry;                       // waves whose supply of source slots has run out
};

// values that are the same in every lane, told to the compiler (scalar registers, scalar branches)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni(long long v)
{
    const int lo = __builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)v);
    const int hi = __builtin_amdgcn_readfirstlane((int)(unsigned int)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}

template <class Src, int NW, int KT, int SCAP, class PP = NoPaths>
__device__ __forceinline__ void trace_pool(const DScene& S, const Src& src, TraceQueue* queue, long long* __restrict__ slow_list,
                                           unsigned int slow_cap, long long chunk, PoolLds<NW, KT, SCAP>& L, Work& w,
                                           int* __restrict__ spill /* [block][MCPT_POOL_SPILL][KT * 64] stack entries beyond SCAP, or null */,
                                           const PP& pp = PP())
{
    static_assert(KT <= 32, "one 32-bit set per class");
    enum { C_INNER = 0, C_LEAF = 1, C_EXACT = 2, C_FIN = 3, C_SHADE = 4, C_DEAD = 5 };
    enum { F_FOUND = 256, F_AMBIG = 512, F_RAY = 1024, F_OWNFAIL = 2048 /* the leader's own box does not pass the reference's box test */ };
    const DFast& F = S.fast;
    const CwNode* __restrict__ nodes = F.cw;
    const DTri* __restrict__ tris = F.tris;
    const DTriPre* __restrict__ pre = F.pre;
    const long long total = src.total();
    const long long small = chunk < MCPT_TAIL_CHUNK ? chunk : MCPT_TAIL_CHUNK;
    const long long big_tickets = (total - total / 8) / chunk;
    const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));      // (wave: the same in every lane, and the compiler is told so)
    const int stack_max = spill ? SCAP + MCPT_POOL_SPILL : SCAP;
    const int stack_cap = F.stack_cap < stack_max ? F.stack_cap : stack_max;
    int* __restrict__ my_spill = spill ? spill + (size_t)blockIdx.x * MCPT_POOL_SPILL * (KT * 64) : nullptr;
    // entry e of slot (k, lane): in LDS below SCAP, else in this block's part of the spill area
    auto st_put = [&](int e, int k, int v) __attribute__((always_inline)) {
        if (e < SCAP) L.stack[(e * KT + k) * 64 + lane] = v;
        else my_spill[((e - SCAP) * KT + k) * 64 + lane] = v;
    };
    auto st_get = [&](int e, int k) __attribute__((always_inline)) -> int {
        return e < SCAP ? L.stack[(e * KT + k) * 64 + lane] : my_spill[((e - SCAP) * KT + k) * 64 + lane];
    };

    // every slot starts in the finish class without a ray: the first steps of every wave are refills
    // (path mode: every path slot starts free in the SHADE class -- the first steps are adoptions -- and no ray slot is filed)
    for (int k = wave; k < KT; k += NW) { L.spf[k * 64 + lane] = 0; if constexpr (PP::kPaths) L.q[k * 64 + lane] = 0u; }
    // More than half of all node steps are on the top three levels of the tree: MCPT_POOL_CACHE_N > 0 reads those nodes from LDS.  Two
    // explicit address spaces and a branch between them -- one pointer that may be either is a flat load, which goes down BOTH paths.
    const int n_cached = F.cached < MCPT_POOL_CACHE_N ? F.cached : MCPT_POOL_CACHE_N;
    { const uint4* g = reinterpret_cast<const uint4*>(nodes); for (int i = threadIdx.x; i < n_cached * 4; i += NW * 64) L.nodes[(i >> 2) * 5 + (i & 3)] = g[i]; }
    // path mode: R ray slots per path (one per light and the bounce ray), NP path slots per lane
    int R = 1, NP = KT;
    if constexpr (PP::kPaths) { R = pp.nl + 1; NP = KT / R; }
    if (wave == 0) {
        L.cls[C_INNER * 64 + lane] = 0u; L.cls[C_LEAF * 64 + lane] = 0u; L.cls[C_EXACT * 64 + lane] = 0u;
        L.cls[C_FIN * 64 + lane] = PP::kPaths ? 0u : (unsigned int)((1ull << KT) - 1ull);
        L.cls[C_SHADE * 64 + lane] = PP::kPaths ? (unsigned int)((1ull << NP) - 1ull) : 0u;
        L.busy[lane] = 0u;
    }
    if (threadIdx.x == 0) { L.live = (PP::kPaths ? NP : KT) * 64; L.dry = 0; }
    if (threadIdx.x < 8) L.stat[threadIdx.x] = 0u;
    __syncthreads();

    unsigned int c_nodes = 0, c_rays = 0, c_exact = 0;
#ifdef MCPT_POOL_DEBUG
    unsigned long long d_used = 0, d_okc = 0, d_steps = 0, d_kill = 0, d_tickets = 0;
    unsigned long long d_cs[5] = {0, 0, 0, 0, 0}, d_cl[5] = {0, 0, 0, 0, 0}, d_sleep = 0, d_miss = 0, d_want[5] = {0, 0, 0, 0, 0};
    unsigned long long d_cyc[6] = {0, 0, 0, 0, 0, 0};      // wave cycles per class, [5]: voting, claiming, sleeping
    unsigned long long d_t = __builtin_amdgcn_s_memtime();
    const unsigned long long d_t0 = d_t;
#endif
    long long next = 0, range_end = 0;          // unclaimed part of the wave's chunk of source slots
    bool queue_empty = false;
    int rot = wave % KT;                        // where a lane starts to look for a set bit: rotates, so no slot waits forever --
    // every pair of waves with a stride of its own (11, 13, 17, 19: odd, and coprime to the 20 slots), so that two waves after the same
    // class do not keep looking at the same slot first (-0.5 % of the kernel against a common stride of 1)
    const int rot_stride = (int)((0x13110d0bu >> (8 * ((wave >> 1) & 3))) & 255u) % KT | 1;
    bool keep = false;                          // MCPT_POOL_STICKY: this lane still owns slot keep_k, which waits for a node step
    int keep_k = 0;
#if MCPT_POOL_PREF
    // nine waves of sixteen lean to the node step, three to the pre-test, two each to the exact test and the refill (their shares of the steps)
    const int pref_slot = (wave * 16 / NW) & 15;
    const int pref = pref_slot < 9 ? C_INNER : (pref_slot < 12 ? C_LEAF : (pref_slot < 14 ? C_EXACT : C_FIN));
#endif

    // What the slot's next step will read from memory is requested now -- by whichever wave runs that step, from the CU's L1 instead of
    // from L2.  The loaded word is not looked at; it is folded into `junk` behind a later load of this wave (loads return in order), so
    // nothing ever waits for it.
    unsigned int pf = 0, junk = 0;
#if MCPT_POOL_PREFETCH
#define MCPT_TOUCH(ptr) pf += *reinterpret_cast<const volatile unsigned int*>(ptr)
#else
#define MCPT_TOUCH(ptr)
#endif
    auto touch_next = [&](bool node, int first, int cnt) __attribute__((always_inline)) {
#if MCPT_POOL_PREFETCH
        if (node) MCPT_TOUCH(nodes + first);
        else if (pre) { MCPT_TOUCH(pre + first); if (cnt > 2) MCPT_TOUCH(pre + first + cnt - 1); }
#endif
    };

    // Culling pads and pruning margin are recomputed from the floats a step has at hand instead of being carried in the slot (16 bytes
    // of 184): in fp32, rounded so that they are never below the values make_rayf / the voting engine use (those have a factor 2.6 of
    // slack over the error they cover) -- a larger pad or margin only culls less; the decisions at the leaves are untouched.
    const float s3f = __double2float_ru(3.0 * F.absmax), absmax_f = __double2float_ru(F.absmax);
    auto pad_of = [&](float of, float rf_) __attribute__((always_inline)) { return (0x1p-20f * 1.00001f) * ((fabsf(of) + s3f) * fabsf(rf_)); };
    auto margin_of = [&](const float of[3], const PoolRcp& a4) __attribute__((always_inline)) {
        const float rmax = fmaxf(fmaxf(fabsf(a4.rx), fabsf(a4.ry)), fabsf(a4.rz));
        const float scale = fmaxf(fmaxf(absmax_f, fabsf(of[0])), fmaxf(fabsf(of[1]), fabsf(of[2])));
        return rmax <= 0.99e6f ? (1.0000001e-9f * 1.00001f) * (scale * rmax) : __builtin_inff();
    };
    // the top of the slot's stack becomes its work (cur, spf are written); returns the class it waits for
    auto pop_next = [&](int idx, int k, int spf) __attribute__((always_inline)) -> int {
        int sp = spf & 255;
        if (sp == 0) return C_FIN;
        sp--;
#if MCPT_POOL_FASTPUSH
        int nxt;
        // (the empty asm keeps the compiler from merging the two reads into one flat load through a selected address)
        if (!__ballot(sp >= SCAP)) { nxt = L.stack[(sp * KT + k) * 64 + lane]; __asm__ volatile("" : "+v"(nxt)); }
        else nxt = st_get(sp, k);
#else
        const int nxt = st_get(sp, k);
#endif
        const bool node = nxt >= 0;
        const int ref = -1 - nxt;
        const int first = node ? nxt : ref >> 4, cnt = (ref & 7) + 1;
        touch_next(node, first, cnt);
        L.cur[idx] = first;
        L.spf[idx] = (spf & 0xff00) | sp | (node ? 0 : cnt << 16);
        return node ? C_INNER : C_LEAF;
    };

    for (;;) {
        const unsigned int m_inner = __hip_atomic_load(&L.cls[C_INNER * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned int m_leaf = __hip_atomic_load(&L.cls[C_LEAF * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned int m_exact = __hip_atomic_load(&L.cls[C_EXACT * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned int m_fin = __hip_atomic_load(&L.cls[C_FIN * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int n_inner = __popcll(__ballot(keep || m_inner != 0u));
        const int n_leaf = __popcll(__ballot(m_leaf != 0u));
        const int n_exact = __popcll(__ballot(m_exact != 0u));
        // A wave's claim on source slots is private (a chunk per ticket): once the tickets are gone, a wave without a chunk leaves the
        // finish class to the waves that still have rays to hand out; a slot is retired only when every wave of the block is dry.
        const bool fin_ok = PP::kPaths || !queue_empty || uni((int)__hip_atomic_load(&L.dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == NW;
        const int n_fin = fin_ok ? __popcll(__ballot(m_fin != 0u)) : 0;
        unsigned int m_shade = 0u;
        int n_shade = 0;
        if constexpr (PP::kPaths) {
            m_shade = __hip_atomic_load(&L.cls[C_SHADE * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            n_shade = __popcll(__ballot(m_shade != 0u));
        }
        if (!(n_inner | n_leaf | n_exact | n_fin | n_shade)) {
            // (no lane keeps a slot here: n_inner counts them)
            if (uni((int)__hip_atomic_load(&L.live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) break;
            __builtin_amdgcn_s_sleep(4);         // the other waves hold what is left
#ifdef MCPT_POOL_DEBUG
            d_sleep++;
#endif
            continue;
        }
        int c = C_FIN, best_score = MCPT_PW_FIN * n_fin;
        if (MCPT_PW_EXACT * n_exact > best_score) { c = C_EXACT; best_score = MCPT_PW_EXACT * n_exact; }
        if (MCPT_PW_LEAF * n_leaf > best_score) { c = C_LEAF; best_score = MCPT_PW_LEAF * n_leaf; }
        if (MCPT_PW_INNER * n_inner > best_score) { c = C_INNER; best_score = MCPT_PW_INNER * n_inner; }
        // (a waiting path keeps up to R ray slots idle: the SHADE class goes first on a tie)
        if (PP::kPaths && MCPT_PW_SHADE * n_shade >= best_score && n_shade) { c = C_SHADE; best_score = MCPT_PW_SHADE * n_shade; }
#if MCPT_POOL_PREF
        {
            const int s0 = MCPT_PW_INNER * n_inner * (pref == C_INNER ? 4 + MCPT_POOL_PREF : 4), s1 = MCPT_PW_LEAF * n_leaf * (pref == C_LEAF ? 4 + MCPT_POOL_PREF : 4);
            const int s2 = MCPT_PW_EXACT * n_exact * (pref == C_EXACT ? 4 + MCPT_POOL_PREF : 4), s3 = MCPT_PW_FIN * n_fin * (pref == C_FIN ? 4 + MCPT_POOL_PREF : 4);
            c = C_FIN; best_score = s3;
            if (s2 > best_score) { c = C_EXACT; best_score = s2; }
            if (s1 > best_score) { c = C_LEAF; best_score = s1; }
            if (s0 > best_score) { c = C_INNER; best_score = s0; }
        }
#endif
        c = uni(c);
        // slots kept for a node step that is not the next step after all are filed now
        if (c != C_INNER && __ballot(keep)) {
            if (keep) __hip_atomic_fetch_or(&L.cls[C_INNER * 64 + lane], 1u << keep_k, MCPT_POOL_FILE_ORDER, __HIP_MEMORY_SCOPE_WORKGROUP);
            keep = false;
        }

        // Claim: another wave's lane of the same index may be after the same slot (it read the same mask a moment ago).  Odd waves look
        // from the top, even ones from the bottom, and a lane that lost tries once more on what the atomic returned (the fresh mask).
        bool have = keep;
        int k = keep_k;
        keep = false;
        unsigned int cm = c == C_INNER ? m_inner : (c == C_LEAF ? m_leaf : (c == C_EXACT ? m_exact : (c == C_FIN ? m_fin : m_shade)));
        unsigned int* const cword = &L.cls[c * 64 + lane];
        // (rot and the wave's parity are scalars: the two masks are too, and the choice between them is a scalar branch)
        const unsigned int below = 0xffffffffu >> (31 - rot), above = 0xffffffffu << rot;
#pragma unroll
        for (int attempt = 0; attempt < MCPT_POOL_CLAIMS; attempt++) {
            const bool want = !have && cm != 0u;
            if (attempt && !__ballot(want)) break;
            if (want) {
                int kk;
                if (wave & 1) { const unsigned int lo = cm & below; kk = 31 - __clz((int)(lo ? lo : cm)); }
                else { const unsigned int hi = cm & above; kk = __ffs((int)(hi ? hi : cm)) - 1; }
                const unsigned int bit = 1u << kk;
                const unsigned int old = __hip_atomic_fetch_and(cword, ~bit, MCPT_POOL_CLAIM_ORDER, __HIP_MEMORY_SCOPE_WORKGROUP);
                have = (old & bit) != 0u;
                k = kk;
                cm = old & ~bit;
            }
        }
        { rot += rot_stride; if (rot >= KT) rot -= KT; }
        const unsigned long long hv = __ballot(have);
#ifdef MCPT_POOL_DEBUG
        if (!hv) d_miss++;
#endif
        if (!hv) continue;
        const int n_have = __popcll(hv);
#ifdef MCPT_POOL_DEBUG
        { const unsigned long long tn = __builtin_amdgcn_s_memtime(); d_cyc[5] += tn - d_t; d_t = tn; }
        d_cs[c]++; d_cl[c] += n_have; d_want[c] += c == C_INNER ? n_inner : (c == C_LEAF ? n_leaf : (c == C_EXACT ? n_exact : (c == C_FIN ? n_fin : n_shade)));
#endif
        const int idx = k * 64 + lane;
        int nc = C_DEAD;
        __asm__ volatile("" ::: "memory");

        if (c == C_INNER) {
            // ---------------------------------------------------------------- one step on a compressed node
            // (a lane that stays on nodes taking its next node step at once -- no filing, vote, claim and reload in between, at the lanes
            // that happen to go on -- was measured in round 4: no difference at two, three or four steps in a row)
            bool refused = false;
            if (have) {
                const int cur = L.cur[idx];
                int spf = L.spf[idx];
                const PoolOxy a0 = pool_ld16(&L.oxy[idx]); const PoolOzDx a1 = pool_ld16(&L.ozdx[idx]); const PoolRcp a4 = L.rcp[idx];
                const float limit = a4.limit;
                RayF rf;
                rf.o[0] = (float)a0.ox; rf.o[1] = (float)a0.oy; rf.o[2] = (float)a1.oz;
                rf.r[0] = a4.rx; rf.r[1] = a4.ry; rf.r[2] = a4.rz;
                rf.pad[0] = pad_of(rf.o[0], a4.rx); rf.pad[1] = pad_of(rf.o[1], a4.ry); rf.pad[2] = pad_of(rf.o[2], a4.rz);
                int sp = spf & 255;
                if (sp > stack_cap - 3) {        // three pushes must fit: the ray goes to the one-lane walk
                    L.spf[idx] = spf | F_AMBIG; nc = C_FIN; refused = true;
                } else {
#if MCPT_POOL_CACHE_N
                    // (the LDS side first: the memory side's loads go into the same registers and would otherwise be waited for before the
                    // LDS reads may even be issued)
                    typedef unsigned int pool_u4 __attribute__((ext_vector_type(4)));
                    pool_u4 v0 = 0u, v1 = 0u, v2 = 0u, v3 = 0u;
                    const bool in_lds = cur < n_cached;
                    if (in_lds) {
                        typedef const pool_u4 __attribute__((address_space(3)))* lds_words;
                        const lds_words q = (lds_words)&L.nodes[cur * 5];
                        v0 = q[0]; v1 = q[1]; v2 = q[2]; v3 = q[3];
                    }
                    __asm__ volatile("" ::: "memory");
                    if (!in_lds) {
                        typedef const pool_u4 __attribute__((address_space(1)))* mem_words;
                        const mem_words q = (mem_words)(nodes + cur);
                        v0 = q[0]; v1 = q[1]; v2 = q[2]; v3 = q[3];
                    }
                    const uint4 w0 = make_uint4(v0.x, v0.y, v0.z, v0.w), w1 = make_uint4(v1.x, v1.y, v1.z, v1.w), w2 = make_uint4(v2.x, v2.y, v2.z, v2.w),
                                w3 = make_uint4(v3.x, v3.y, v3.z, v3.w);
                    const CwHits h = cw_step_words(w0, w1, w2, w3, rf, limit);
#else
                    const CwHits h = cw_step(nodes + cur, rf, limit);        // (a global load: a pointer that may be LDS or memory is a flat one)
#endif

                    junk += pf; pf = 0;
#if MCPT_POOL_FASTPUSH
                    // The children come back sorted with the culled ones last: n hits, the n - 1 farther ones go on the stack, farthest
                    // first.  When every lane's pushes stay in the LDS part of its stack (nine steps in ten) they are three predicated
                    // stores at computed positions -- no branch per push, none between LDS and the spill area.
                    const int n_hit = (h.ref[0] != MCPT_FAST_EMPTY) + (h.ref[1] != MCPT_FAST_EMPTY) + (h.ref[2] != MCPT_FAST_EMPTY) + (h.ref[3] != MCPT_FAST_EMPTY);
                    if (!__ballot(sp + n_hit - 1 > SCAP)) {
                        int* const col = &L.stack[k * 64 + lane];
                        if (n_hit >= 4) col[sp * (KT * 64)] = h.ref[3];
                        if (n_hit >= 3) col[(sp + n_hit - 3) * (KT * 64)] = h.ref[2];
                        if (n_hit >= 2) col[(sp + n_hit - 2) * (KT * 64)] = h.ref[1];
                        sp += n_hit > 1 ? n_hit - 1 : 0;
                    } else {
                        if (h.ref[3] != MCPT_FAST_EMPTY) { st_put(sp, k, h.ref[3]); sp++; }
                        if (h.ref[2] != MCPT_FAST_EMPTY) { st_put(sp, k, h.ref[2]); sp++; }
                        if (h.ref[1] != MCPT_FAST_EMPTY) { st_put(sp, k, h.ref[1]); sp++; }
                    }
#else
                    if (h.ref[3] != MCPT_FAST_EMPTY) { st_put(sp, k, h.ref[3]); sp++; }
                    if (h.ref[2] != MCPT_FAST_EMPTY) { st_put(sp, k, h.ref[2]); sp++; }
                    if (h.ref[1] != MCPT_FAST_EMPTY) { st_put(sp, k, h.ref[1]); sp++; }
#endif
                    int nxt = h.ref[0];
#if MCPT_POOL_FASTPUSH
                    {
                        const bool pop = nxt == MCPT_FAST_EMPTY && sp > 0;
                        if (pop) sp--;
                        if (!__ballot(pop && sp >= SCAP)) { if (pop) { nxt = L.stack[(sp * KT + k) * 64 + lane]; __asm__ volatile("" : "+v"(nxt)); } }      // (an LDS read, not a flat one)
                        else if (pop) nxt = st_get(sp, k);
                    }
#else
                    if (nxt == MCPT_FAST_EMPTY && sp > 0) { sp--; nxt = st_get(sp, k); }
#endif
                    const bool node = nxt >= 0, none = nxt == MCPT_FAST_EMPTY;
                    const int ref = -1 - nxt;
                    const int first = node ? nxt : ref >> 4, cnt = (ref & 7) + 1;
                    if (!none) { touch_next(node, first, cnt); L.cur[idx] = first; }
                    spf = (spf & 0xff00) | sp | ((!node && !none) ? cnt << 16 : 0);
                    L.spf[idx] = spf;
                    nc = node ? C_INNER : (none ? C_FIN : C_LEAF);
                }
            }
            c_nodes += (unsigned int)(n_have - __popcll(__ballot(refused)));
        } else if (c == C_LEAF) {
            // ---------------------------------------------------------------- the triangles of a leaf through the fp32 pre-test
            if (have) {
                const int cur = L.cur[idx];
                const int spf0 = L.spf[idx];
                const int cnt = (spf0 >> 16) & 255;
                const PoolOxy a0 = pool_ld16(&L.oxy[idx]); const PoolOzDx a1 = pool_ld16(&L.ozdx[idx]); const PoolDyz a2 = pool_ld16(&L.dyz[idx]); const PoolRcp a4 = L.rcp[idx];
                const float of[3] = {(float)a0.ox, (float)a0.oy, (float)a1.oz};
                const float limit = a4.limit, margin = margin_of(of, a4);
                unsigned int surv = 0;
                w.tris += cnt;
#if MCPT_PRE_TEST
                if (!pre) surv = (1u << cnt) - 1u;
                else {
                    Ray r; r.o = mk(a0.ox, a0.oy, a1.oz); r.d = mk(a1.dx, a2.dy, a2.dz);
                    const PreRay pr = make_pre_ray(F, r, of, margin);
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                    for (int k0 = 0; k0 < cnt; k0 += MCPT_PRE_UNROLL) {
#pragma unroll
                        for (int j = 0; j < MCPT_PRE_UNROLL; j++) {
                            const int kk = k0 + j;
                            const bool rej = tri_pre_reject(pre + cur + kk, pr, limit);
                            if (kk < cnt && !rej) surv |= 1u << kk;
                        }
                    }
                    junk += pf; pf = 0;
#ifdef MCPT_PRE_CHECK
                    const double bt = L.best_t[idx];
                    const bool found = (spf0 & F_FOUND) != 0;
                    for (int kk = 0; kk < cnt; kk++) {
                        if ((surv >> kk) & 1u) continue;
                        V3 pc;
                        if (tri_hit(tris + cur + kk, r, pc)) {
                            const double tc = (pc.x - r.o.x) / r.d.x;
                            if (tc > 0.0 && (!found || tc <= bt * (1.0 + 0x1p-40))) { w.pre_wrong++; surv |= 1u << kk; }
                        }
                    }
#endif
                }
#else
                surv = (1u << cnt) - 1u;
#endif
                if (surv) {
#if MCPT_POOL_PREFETCH
                    { const DTri* tn = tris + cur + (__ffs((int)surv) - 1); MCPT_TOUCH(tn); MCPT_TOUCH(reinterpret_cast<const char*>(tn) + 64); }
#endif
                    L.spf[idx] = (spf0 & 0xffff) | ((int)surv << 16); nc = C_EXACT;
                }
                else nc = pop_next(idx, k, spf0);
            }
        } else if (c == C_EXACT) {
            // ---------------------------------------------------------------- one surviving triangle through the reference's test
            c_exact += (unsigned int)n_have;
            if (have) {
                const int cur = L.cur[idx];
                int spf = L.spf[idx];
                int surv = (spf >> 16) & 255;
                const PoolOxy a0 = pool_ld16(&L.oxy[idx]); const PoolOzDx a1 = pool_ld16(&L.ozdx[idx]); const PoolDyz a2 = pool_ld16(&L.dyz[idx]);
                const double best_t = L.best_t[idx];
                Ray r; r.o = mk(a0.ox, a0.oy, a1.oz); r.d = mk(a1.dx, a2.dy, a2.dz);
                const int kk = __ffs(surv) - 1;
                surv &= surv - 1;
                const int ti = cur + kk;
                const DTri* tr = tris + ti;
                const bool found = (spf & F_FOUND) != 0;
                V3 p;
                const bool hit = tri_hit(tr, r, p);
                junk += pf; pf = 0;
                if (hit) {
                    const double ta = (p.x - r.o.x) * fast_rcp(r.d.x);
                    if (ta > 0.0) {
                        const double band = best_t * 0x1p-47;
                        // The reference tests a leaf's own box before its triangle (bvh_intersect): a leader whose own box fails sends the ray
                        // to the exact walk when it is finished.  The box is tested here, where the triangle is in registers -- in the result
                        // step it was five more gathers and their address arithmetic per ray (-1 % of the kernel).
                        if (!found || ta < best_t - band) {
                            const PoolRcp a4 = L.rcp[idx];
                            const float of[3] = {(float)a0.ox, (float)a0.oy, (float)a1.oz};
                            L.best_t[idx] = ta; L.best_leaf[idx] = ti;
                            L.rcp[idx].limit = __double2float_ru((ta + ta * 0x1p-47) + (double)margin_of(of, a4));
                            const bool own = own_box_hit(tr, r, mk(fast_rcp(r.d.x), fast_rcp(r.d.y), fast_rcp(r.d.z)));
                            spf = (spf & ~F_OWNFAIL) | F_FOUND | (own ? 0 : F_OWNFAIL);
                        } else if (!(ta > best_t + band)) {
                            // (the leader's hit point again: the first two lines of the reference's test on its triangle, same operands, same bits)
                            const DTri* lt = tris + L.best_leaf[idx];
                            const V3 lv1 = ld3(lt->v1), ln = ld3(lt->n);
                            const double tl = dot(lv1 - r.o, ln) / dot(ln, r.d);
                            const double old_px = (r.o + r.d * tl).x;
                            const double t_new = (p.x - r.o.x) / r.d.x, t_old = (old_px - r.o.x) / r.d.x;
                            if (t_new < t_old || (t_new == t_old && tr->leaf < lt->leaf)) {
                                L.best_t[idx] = ta; L.best_leaf[idx] = ti;
                                const bool own = own_box_hit(tr, r, mk(fast_rcp(r.d.x), fast_rcp(r.d.y), fast_rcp(r.d.z)));
                                spf = (spf & ~F_OWNFAIL) | (own ? 0 : F_OWNFAIL);
                            }
                        }
                    }
                }
                if (surv) {
#if MCPT_POOL_PREFETCH
                    { const DTri* tn = tris + cur + (__ffs(surv) - 1); MCPT_TOUCH(tn); MCPT_TOUCH(reinterpret_cast<const char*>(tn) + 64); }
#endif
                    spf = (spf & 0xffff) | (surv << 16);
                    L.spf[idx] = spf; nc = C_EXACT;
                }
                else nc = pop_next(idx, k, spf);
                if (nc == C_FIN) L.spf[idx] = spf & 0xffff;
            }
        } else if (PP::kPaths && c == C_FIN) {
            if constexpr (PP::kPaths) {
                // ---------------------------------------------------------------- path mode: the ray's answer stays in its slot
                // (reference leaf in best_leaf, its material in cur; the ray itself stays where it is: the SHADE step forms the hit point from it)
                if (have) {
                    const int spf = L.spf[idx];
                    const PoolOxy a0 = pool_ld16(&L.oxy[idx]); const PoolOzDx a1 = pool_ld16(&L.ozdx[idx]); const PoolDyz a2 = pool_ld16(&L.dyz[idx]);
                    Ray r; r.o = mk(a0.ox, a0.oy, a1.oz); r.d = mk(a1.dx, a2.dy, a2.dz);
                    bool found = (spf & F_FOUND) != 0;
                    bool ambiguous = (spf & F_AMBIG) != 0;
                    int leaf_ref = -1, mat = -1;
                    if (found) {
                        const DTri* tr = tris + L.best_leaf[idx];
                        if (spf & F_OWNFAIL) ambiguous = true;              // (tested when the triangle became the leader)
                        leaf_ref = tr->leaf; mat = tr->material;
                    }
                    if (ambiguous) {
                        // rare (a leader whose own box fails, a walk past the stack, a ray the fast walk may not take): the one-lane exact
                        // walk on the spot -- what the trace kernels leave to their deferred-ray pass
                        Hit hh;
                        Work w2 = {0, 0};
                        const bool ok = fast_path_ok(F, r) ? trace_lane_fast(S, r, hh, w2, pp.lane_stack(wave, lane), 1) : trace_closest(S, r, hh, w2);
                        w.nodes += w2.nodes; w.tris += w2.tris;
                        found = ok; leaf_ref = ok ? hh.leaf : -1; mat = ok ? S.tris[hh.leaf].material : -1;
                    }
                    L.best_leaf[idx] = found ? leaf_ref : -1;
                    L.cur[idx] = found ? mat : -1;
                    // whoever clears the last busy bit of a path files the path (acquire + release: the other rays' answers, written by other
                    // waves before they cleared theirs, are visible to the SHADE step this leads to)
                    const unsigned int bit = 1u << k;
                    const unsigned int old = __hip_atomic_fetch_and(&L.busy[lane], ~bit, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int pk = (int)(((float)k + 0.5f) * pp.inv_r);
                    const unsigned int pm = ((1u << R) - 1u) << (pk * R);
                    if (((old & ~bit) & pm) == 0u)
                        __hip_atomic_fetch_or(&L.cls[C_SHADE * 64 + lane], 1u << pk, MCPT_POOL_FILE_ORDER, __HIP_MEMORY_SCOPE_WORKGROUP);
                    nc = C_DEAD;                                        // the ray slot is idle until its path's next vertex
                }
            }
        } else if (PP::kPaths) {
            if constexpr (PP::kPaths) {
                // ---------------------------------------------------------------- a path whose rays are all back, or a free path slot
                // k = path slot; its ray slots are k * R + l, l = R - 1 for the bounce ray.  The arithmetic is k_wf_logic's / k_wf_finish's
                // (vertex.hpp), operation for operation; a path's record lives in this block's part of a global area, [field][path slot][lane].
                const WfArgs& a = pp.a;
                const int nl = pp.nl;
                const bool folded = nl == 1;                             // see k_wf_logic
                const long long cap = a.cap;
                const int s0 = k * R;
                const size_t plane = (size_t)pp.npc * 64;
                double* __restrict__ rd = pp.recd + (size_t)blockIdx.x * (size_t)(9 + 3 * nl) * plane + (size_t)k * 64 + lane;
                int* __restrict__ ri = pp.reci + (size_t)blockIdx.x * (size_t)(3 + nl) * plane + (size_t)k * 64 + lane;
                enum { P_FREE = 0, P_ADOPTED = 1, P_VERTEX = 2 };
                enum { RI_ID = 0, RI_DEPTH = 1, RI_BT = 2, RI_EXPECT = 3 };
                enum { RD_T = 0, RD_L = 3, RD_W = 6, RD_C = 9 };
                auto ldp = [&](int f) __attribute__((always_inline)) { return mk(rd[(size_t)f * plane], rd[(size_t)(f + 1) * plane], rd[(size_t)(f + 2) * plane]); };
                auto stp = [&](int f, const V3& v) __attribute__((always_inline)) { rd[(size_t)f * plane] = v.x; rd[(size_t)(f + 1) * plane] = v.y; rd[(size_t)(f + 2) * plane] = v.z; };
                int mode = have ? (int)L.q[s0 * 64 + lane] : P_FREE;
                int id = 0, leaf = -1, in_type = RT_TRANSMISSION;
                uint32_t depth = 0;
                V3 T = mk(1, 1, 1), Lr = mk(0, 0, 0), p = mk(0, 0, 0), dir = mk(0, 0, 0);
                bool at_vertex = false;
                // ---- resolve the vertex whose rays have come back (pathTracing.cpp:213-231, 244-261)
                if (have && mode != P_FREE) {
                    id = ri[RI_ID * plane]; depth = (uint32_t)ri[RI_DEPTH * plane];
                    const int bt = ri[RI_BT * plane];
                    T = ldp(RD_T); Lr = ldp(RD_L);
                    V3 L_dir = mk(0, 0, 0);
                    for (int l = 0; l < nl; l++) {
                        const int expect = ri[(size_t)(RI_EXPECT + l) * plane];
                        if (expect == -2) continue;
                        const V3 cc = ldp(RD_C + 3 * l);
                        const bool vis = L.cur[(s0 + l) * 64 + lane] == expect;
                        L_dir.x += vis ? cc.x : cc.x * 0.0;
                        L_dir.y += vis ? cc.y : cc.y * 0.0;
                        L_dir.z += vis ? cc.z : cc.z * 0.0;
                    }
                    if (mode == P_ADOPTED && folded) Lr = Lr + L_dir;            // its c was stored as T * c
                    else Lr = Lr + mk(T.x * L_dir.x, T.y * L_dir.y, T.z * L_dir.z);
                    const int bidx = (s0 + nl) * 64 + lane;
                    const int hl = bt >= 0 ? L.best_leaf[bidx] : -1;
                    if (hl >= 0) {
                        // an adopted path with one light already holds the throughput after its bounce
                        if (!(mode == P_ADOPTED && folded)) { const V3 wgt = ldp(RD_W); T = mk(T.x * wgt.x * MCPT_INV_P_RR, T.y * wgt.y * MCPT_INV_P_RR, T.z * wgt.z * MCPT_INV_P_RR); }
                        // the hit point: the first two lines of the reference's triangle test on the ray that is still in its slot
                        const PoolOxy b0 = pool_ld16(&L.oxy[bidx]); const PoolOzDx b1 = pool_ld16(&L.ozdx[bidx]); const PoolDyz b2 = pool_ld16(&L.dyz[bidx]);
                        const V3 ro = mk(b0.ox, b0.oy, b1.oz), bd = mk(b1.dx, b2.dy, b2.dz);
                        const DTri* tr = S.tris + hl;
                        const V3 v1 = ld3(tr->v1), n = ld3(tr->n);
                        const double t = dot(v1 - ro, n) / dot(n, bd);
                        p = ro + bd * t; dir = neg(bd); in_type = bt & 7; depth++; leaf = hl;
                        at_vertex = true;
                    }
                }
                bool ended = have && mode != P_FREE && !at_vertex;       // no bounce ray, or it left the scene
                // ---- the next vertex: emitter test, surface (pathTracing.cpp:141-160)
                const DMaterial* m = nullptr;
                V3 pn = mk(0, 0, 0), kd = mk(0, 0, 0);
                const int n_shades = __popcll(__ballot(at_vertex));
                const unsigned int deepest = wave_max(at_vertex ? depth : 0u);
                if (at_vertex) {
                    m = S.materials + S.tris[leaf].material;
                    if (m->light >= 0) {
                        const V3 rad = ld3(S.lights[m->light].radiance);
                        if (depth == 0) Lr = rad;
                        else if (in_type != RT_DIFFUSE) Lr = Lr + mk(T.x * rad.x, T.y * rad.y, T.z * rad.z);
                        ended = true; at_vertex = false;
                    } else vertex_surface(S, leaf, p, m, pn, kd);
                }
                if (ended) { a.rad[(size_t)id * 3] = Lr.x; a.rad[(size_t)id * 3 + 1] = Lr.y; a.rad[(size_t)id * 3 + 2] = Lr.z; mode = P_FREE; }
                // ---- rays go straight into the path's slots
                unsigned int nb_inner = 0u, nb_fin = 0u;
                auto emit = [&](int sk, const V3& o, const V3& d) __attribute__((always_inline)) {
                    const int i2 = sk * 64 + lane;
                    Ray r; r.o = o; r.d = d;
                    const bool okf = fast_path_ok(F, r);
                    const V3 rcp = mk(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
                    PoolOxy b0; b0.ox = o.x; b0.oy = o.y;
                    PoolOzDx b1; b1.oz = o.z; b1.dx = d.x;
                    PoolDyz b2; b2.dy = d.y; b2.dz = d.z;
                    PoolRcp b4; b4.rx = (float)rcp.x; b4.ry = (float)rcp.y; b4.rz = (float)rcp.z; b4.limit = __builtin_inff();
                    L.oxy[i2] = b0; L.ozdx[i2] = b1; L.dyz[i2] = b2; L.rcp[i2] = b4;
                    L.best_t[i2] = 0;
                    L.cur[i2] = 0; L.best_leaf[i2] = -1; L.spf[i2] = okf ? F_RAY : (F_RAY | F_AMBIG);
                    if (okf) nb_inner |= 1u << sk; else nb_fin |= 1u << sk;
                };
                int n_shadow = 0, n_skipped = 0, n_bounce = 0;
                if (at_vertex) {
                    RngKey key;
                    key.k0 = (uint32_t)a.seed; key.k1 = (uint32_t)(a.seed >> 32);
                    const int slot = a.first_slot + id / a.spp;
                    key.pixel = (uint32_t)(a.pixels ? a.pixels[slot] : slot); key.sample = (uint32_t)(id % a.spp);
                    int sample_mat = -1;
                    for (int l = 0; l < nl; l++) {
                        V3 direction, cc;
                        const int expect = light_sample(S, key, depth, l, p, pn, kd, sample_mat, direction, cc);
                        ri[(size_t)(RI_EXPECT + l) * plane] = expect;
                        if (expect != -2) { stp(RD_C + 3 * l, cc); emit(s0 + l, p + direction * 0.01, direction); n_shadow++; }
                        else n_skipped++;
                    }
                    V3 nd = mk(0, 0, 0), wgt = mk(1, 1, 1);
                    const int bt = bounce_sample(key, depth, nl, m, dir, pn, kd, nd, wgt);
                    ri[RI_BT * plane] = bt;
                    if (bt >= 0) { stp(RD_W, wgt); emit(s0 + nl, (bt & MCPT_BT_NO_OFFSET) ? p : p + nd * 0.01, nd); n_bounce++; }
                    ri[RI_DEPTH * plane] = (int)depth;
                    stp(RD_T, T); stp(RD_L, Lr);
                    mode = P_VERTEX;
                }
                // ---- a free path slot adopts the next path of the wavefront state (its rays are pending there: k_wf_logic wrote them)
                const bool want_path = have && mode == P_FREE;
                const unsigned long long wb = __ballot(want_path);
                bool dead = false;
                if (wb) {
                    long long mine = pp.n;
                    if (!queue_empty) {
                        const unsigned int want = (unsigned int)__popcll(wb);
                        unsigned int got = 0;
                        if (lane == 0) got = atomicAdd(&a.counts->pad[0], want);
                        got = (unsigned int)uni((int)got);
                        if ((long long)got + want >= pp.n) queue_empty = true;
                        mine = (long long)got + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(wb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)wb, 0u));
                    }
                    if (want_path) {
                        if (mine < pp.n) {
                            const long long j = mine;
                            id = a.out.id[j];
                            depth = (uint32_t)a.depth;
                            T = mk(1, 1, 1); Lr = mk(0, 0, 0);
                            if (a.depth > 0 || folded) T = ldc(a.out.T, cap, j);
                            if (a.depth > 0) Lr = ldc(a.out.L, cap, j);
                            if (a.depth == 0) { const PrimaryHit* ph = a.hits + (a.first_slot + id / a.spp); p = mk(ph->p[0], ph->p[1], ph->p[2]); }
                            else p = ldc(a.out.p, cap, j);
                            for (int l = 0; l < nl; l++) {
                                const int expect = a.out.expect[(long long)l * cap + j];
                                ri[(size_t)(RI_EXPECT + l) * plane] = expect;
                                if (expect != -2) {
                                    const V3 cc = ldc(a.out.c + (long long)l * 3 * cap, cap, j), d = ldc(a.rays.d + (long long)l * 3 * cap, cap, j);
                                    stp(RD_C + 3 * l, cc); emit(s0 + l, p + d * 0.01, d);
                                }
                            }
                            const int bt = a.out.btype[j];
                            ri[RI_BT * plane] = bt;
                            if (bt >= 0) {
                                const V3 bd = ldc(a.out.bdir, cap, j);
                                if (!folded) stp(RD_W, ldc(a.out.w, cap, j));
                                emit(s0 + nl, (bt & MCPT_BT_NO_OFFSET) ? p : p + bd * 0.01, bd);
                            }
                            ri[RI_ID * plane] = id; ri[RI_DEPTH * plane] = (int)depth;
                            stp(RD_T, T); stp(RD_L, Lr);
                            mode = P_ADOPTED;
                        } else dead = true;                                 // no path left, in any block
                    }
                }
                if (have) {
                    L.q[s0 * 64 + lane] = (unsigned int)mode;
                    const unsigned int nb = nb_inner | nb_fin;
                    // (the busy bits before the filing: a ray's result step, in another wave, clears its bit)
                    if (nb) __hip_atomic_fetch_or(&L.busy[lane], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (nb_inner) __hip_atomic_fetch_or(&L.cls[C_INNER * 64 + lane], nb_inner, MCPT_POOL_FILE_ORDER, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (nb_fin) __hip_atomic_fetch_or(&L.cls[C_FIN * 64 + lane], nb_fin, MCPT_POOL_FILE_ORDER, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // a path without a ray in flight (nothing to trace at its vertex, or adopted without rays) is resolved by the next step;
                    // a free slot that found no path is retired
                    nc = (mode != P_FREE && !nb) ? C_SHADE : C_DEAD;
                    w.rays += (unsigned int)__popc(nb);
                }
                const int n_dead = __popcll(__ballot(have && dead));
                if (n_dead && lane == 0) __hip_atomic_fetch_sub(&L.live, (unsigned int)n_dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // the logic kernel's counters, once per step
                const unsigned int t_sh = (unsigned int)wave_sum((unsigned long long)n_shadow), t_sk = (unsigned int)wave_sum((unsigned long long)n_skipped),
                                   t_bo = (unsigned int)wave_sum((unsigned long long)n_bounce);
                if (lane == 0) {
                    if (n_shades) atomicAdd(&L.stat[0], (unsigned int)n_shades);
                    if (t_sh) atomicAdd(&L.stat[1], t_sh);
                    if (t_bo) atomicAdd(&L.stat[2], t_bo);
                    if (t_sk) atomicAdd(&L.stat[3], t_sk);
                    if (deepest) atomicMax(&L.stat[4], deepest);
                }
            }
        } else {
            // ---------------------------------------------------------------- results out, new rays in
            if (have) {
                const int spf = L.spf[idx];
                if (spf & F_RAY) {
                    const PoolOxy a0 = pool_ld16(&L.oxy[idx]); const PoolOzDx a1 = pool_ld16(&L.ozdx[idx]); const PoolDyz a2 = pool_ld16(&L.dyz[idx]);
                    const long long slot = (long long)L.q[idx];
                    Ray r; r.o = mk(a0.ox, a0.oy, a1.oz); r.d = mk(a1.dx, a2.dy, a2.dz);
                    const bool found = (spf & F_FOUND) != 0;
                    bool ambiguous = (spf & F_AMBIG) != 0;
                    Hit h; h.leaf = -1; h.t = 0; h.p = mk(0, 0, 0);
                    if (found) {
                        const DTri* tr = tris + L.best_leaf[idx];
                        if (spf & F_OWNFAIL) ambiguous = true;              // (tested when the triangle became the leader)
                        h.leaf = tr->leaf;
                        h.mat = tr->material;
                        // the hit point: the first two lines of intersect(Ray&, Face&, Vertex&) again -- same operands, same bits as when the
                        // triangle was tested; t_k is divided out of its x (pathTracing.cpp:347)
                        const V3 v1 = ld3(tr->v1), n = ld3(tr->n);
                        const double t = dot(v1 - r.o, n) / dot(n, r.d);
                        h.p = r.o + r.d * t;
                        h.t = (h.p.x - r.o.x) / r.d.x;
                    }
                    if (ambiguous) {
                        const unsigned int at = atomicAdd(&queue->slow_count, 1u);
                        if (at < slow_cap) slow_list[at] = slot;
                        else queue->redo_all = 1u;
                    } else src.store(slot, found, h);
                }
            }
            bool got = false;
            bool all_dry = queue_empty;             // (a dry wave is only here when every wave is)
            if (!queue_empty && next >= range_end) {
                unsigned long long tk = 0;
                if (lane == 0) tk = atomicAdd(&queue->head, 1ull);
                const long long ticket = uni((long long)tk);
                const long long size = ticket < big_tickets ? chunk : small;
                next = ticket < big_tickets ? ticket * chunk : big_tickets * chunk + (ticket - big_tickets) * small;
                range_end = next + size < total ? next + size : total;
#ifdef MCPT_POOL_DEBUG
                d_tickets++;
#endif
                if (next >= total) {
                    queue_empty = true;
                    unsigned int before = 0;
                    if (lane == 0) before = __hip_atomic_fetch_add(&L.dry, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    all_dry = uni((int)before) + 1 == NW;
                }
            }
            if (!queue_empty) {
                // the next <= 64 source slots, one per lane; the rays among them go to the lanes that need one, in order
                const long long left = range_end - next;
                const int avail = left < 64 ? (int)left : 64;
                Ray nr; nr.o = mk(0, 0, 0); nr.d = mk(1, 1, 1);
                const bool valid = lane < avail && src.fetch(next + lane, nr);
                const bool ok = valid && fast_path_ok(F, nr);
                junk += pf; pf = 0;
                const unsigned long long V = __ballot(ok);
                const int n_ok = __popcll(V);
                const int rank_ok = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(V >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)V, 0u));
                int used = avail;                // source slots consumed by this step
                if (n_ok > n_have) used = __ffsll((long long)__ballot(ok && rank_ok == n_have - 1));
                used = uni(used);
                if (valid && !ok && lane < used) {      // a ray the fast walk may not take
                    const unsigned int at = atomicAdd(&queue->slow_count, 1u);
                    if (at < slow_cap) slow_list[at] = next + lane;
                }
                if (ok && rank_ok < n_have) L.tbl[wave * 64 + rank_ok] = lane;
                const int rank_need = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(hv >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)hv, 0u));
                got = have && rank_need < n_ok;
                __asm__ volatile("" ::: "memory");
                const int from = got ? L.tbl[wave * 64 + rank_need] : lane;
                Ray r;
                r.o.x = __shfl(nr.o.x, from, 64); r.o.y = __shfl(nr.o.y, from, 64); r.o.z = __shfl(nr.o.z, from, 64);
                r.d.x = __shfl(nr.d.x, from, 64); r.d.y = __shfl(nr.d.y, from, 64); r.d.z = __shfl(nr.d.z, from, 64);
                if (got) {
                    const V3 rcp = mk(fast_rcp(r.d.x), fast_rcp(r.d.y), fast_rcp(r.d.z));
                    PoolOxy b0; b0.ox = r.o.x; b0.oy = r.o.y;
                    PoolOzDx b1; b1.oz = r.o.z; b1.dx = r.d.x;
                    PoolDyz b2; b2.dy = r.d.y; b2.dz = r.d.z;
                    PoolRcp b4; b4.rx = (float)rcp.x; b4.ry = (float)rcp.y; b4.rz = (float)rcp.z; b4.limit = __builtin_inff();
                    L.oxy[idx] = b0; L.ozdx[idx] = b1; L.dyz[idx] = b2; L.rcp[idx] = b4;
                    L.best_t[idx] = 0;
                    L.cur[idx] = 0; L.best_leaf[idx] = -1; L.spf[idx] = F_RAY;
                    L.q[idx] = (unsigned int)(next + from);
                    MCPT_TOUCH(nodes);
                    nc = C_INNER;
                }
                next += used;
#ifdef MCPT_POOL_DEBUG
                d_used += used; d_okc += __popcll(__ballot(ok && lane < used)); d_steps++;
#endif
                c_rays += (unsigned int)__popcll(__ballot(got));
            }
            if (have && !got) {
                if (queue_empty && all_dry) nc = C_DEAD;           // no ray left for this slot, in any wave
                else { L.spf[idx] = 0; nc = C_FIN; }               // it asks again
            }
            const int n_dead = __popcll(__ballot(have && nc == C_DEAD));
#ifdef MCPT_POOL_DEBUG
            d_kill += n_dead;
#endif
            if (n_dead && lane == 0) __hip_atomic_fetch_sub(&L.live, (unsigned int)n_dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __asm__ volatile("" ::: "memory");
#ifdef MCPT_POOL_DEBUG
        { const unsigned long long tn = __builtin_amdgcn_s_memtime(); d_cyc[c] += tn - d_t; d_t = tn; }
#endif
#if MCPT_POOL_STICKY
        if (have && c == C_INNER && nc == C_INNER) { keep = true; keep_k = k; nc = C_DEAD; }      // (not filed: it stays with this lane)
#endif
        if (have && nc != C_DEAD)
            __hip_atomic_fetch_or(&L.cls[nc * 64 + lane], 1u << k, MCPT_POOL_FILE_ORDER, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#undef MCPT_TOUCH
    junk += pf;
    if (lane == 0) { w.nodes += c_nodes + (junk == 0x9e3779b9u ? 1u : 0u); w.rays += c_rays; w.exact += c_exact; }
#ifdef MCPT_POOL_DEBUG
    if constexpr (PP::kPaths) {
        if (lane == 0 && pp.a.ctr) {
            unsigned long long* o = pp.a.ctr->pp;
            for (int i = 0; i < 5; i++) { atomicAdd(&o[i], d_cs[i]); atomicAdd(&o[5 + i], d_cl[i]); atomicAdd(&o[12 + i], d_cyc[i]); }
            atomicAdd(&o[10], d_sleep); atomicAdd(&o[11], d_miss); atomicAdd(&o[17], d_cyc[5]);
            atomicAdd(&o[18], __builtin_amdgcn_s_memtime() - d_t0); atomicAdd(&o[19], 1ull);
        }
    } else
    if (lane == 0 && w.dbg) {
        {   // (the same account for the trace launches: DCounters::pp follows DCounters::dbg)
            unsigned long long* o = w.dbg + 24;
            for (int i = 0; i < 4; i++) { atomicAdd(&o[i], d_cs[i]); atomicAdd(&o[5 + i], d_cl[i]); atomicAdd(&o[12 + i], d_cyc[i]); }
            atomicAdd(&o[10], d_sleep); atomicAdd(&o[11], d_miss); atomicAdd(&o[17], d_cyc[5]);
            atomicAdd(&o[18], __builtin_amdgcn_s_memtime() - d_t0); atomicAdd(&o[19], 1ull);
        }
        atomicAdd(&w.dbg[0], d_used); atomicAdd(&w.dbg[1], d_okc); atomicAdd(&w.dbg[2], d_steps); atomicAdd(&w.dbg[3], d_kill); atomicAdd(&w.dbg[6], d_tickets);
        atomicAdd(&w.dbg[7], (unsigned long long)c_rays);
        for (int i = 0; i < 4; i++) { atomicAdd(&w.dbg[8 + i], d_cs[i]); atomicAdd(&w.dbg[12 + i], d_cl[i]); atomicAdd(&w.dbg[16 + i], d_want[i]); }
        atomicAdd(&w.dbg[20], d_sleep); atomicAdd(&w.dbg[21], d_miss);
        if (threadIdx.x == 0 && blockIdx.x == 0) { atomicAdd(&w.dbg[4], (unsigned long long)total); atomicAdd(&w.dbg[5], 1ull); }
    }
#endif
}

}  // namespace mcpt
