// bf_hamming.hip — brute-force Hamming top-2 over 256-bit ORB descriptors on
// gfx950 (MI355X).  Replaces cv2.BFMatcher(NORM_HAMMING).match / knnMatch(k=2)
// behind BruteForceFeatureMatcher.match (reference feature_matchers.py:33-39).
//
// Design:
//   * lane <-> query: every lane keeps R query descriptors in registers
//     (8*R VGPRs), so the per-query top-2 lives in registers and never needs a
//     cross-lane reduction.
//   * inner op per 32-bit word: v_xor_b32 + accumulating v_bcnt_u32_b32 =
//     16 VALU ops per 256-bit pair, the floor for this ISA without MFMA.  A
//     gfx950 SIMD runs two wave64 VALU instructions at once when they come from
//     different waves and at most one of them is a v_bcnt, so the floor is
//     16 x 4 / 2 = 32 cycles per wave-row; the wave raises its priority for its
//     v_bcnt half so that the v_bcnt of one wave pairs with the v_xor of another
//     (row_acc below; measured in shader cycles by tools/ubench/cycles.hip).
//   * train rows are wave-uniform.  Long chunks (>= 512 rows) stream them through
//     SGPRs: s_load_dwordx16 + v_xor with the SGPR operand, no LDS, no barrier
//     (bf_scan_sgpr.h, one generated asm statement per stretch of rows).  Short
//     chunks and frame-sized calls stage them global -> LDS with coalesced 16-B
//     loads (one 256-row / 8-KiB tile per step, double buffered) and read them
//     back as wave-uniform (broadcast) ds_read_b128.
//   * top-2 maintenance is filtered: each accumulator starts at
//     2^31 - (current threshold), so "this pair improved" is the sign bit of
//     the accumulator.  The accumulators of 16 consecutive train rows are
//     combined (AND tree in the LDS form, unsigned min3 chain in the SGPR form)
//     and one compare + one wave-uniform branch on the ballot (update block laid
//     out as unlikely) serves the whole group.  The update itself is branch-free
//     on packed keys (dist << 23 | train index): 2nd = med3, 1st = min, which
//     keeps OpenCV's (distance asc, index asc) order because keys are unique and
//     compare lexicographically.
//   * the train axis is split into chunks so any N fills 256 CUs.  Chunk boundaries come from a small table built on
//     the host (make_plan_core).  Large searches run a QUEUE plan (round 4): as many worker blocks per query block as are
//     resident at once, whose waves draw the chunks by ticket (one returning atomic add per chunk) and keep their top-2
//     from chunk to chunk; the chunks shrink towards the end of the queue, so all waves run out of work together, and
//     nobody waits for anybody.  Other shapes run one block per chunk (grid.y = chunks): leader chunk(s) first, uniform
//     ones, then a run of shrinking ones so that the blocks dispatched last have the least to do.
//   * blocks that scan different chunks for the same queries tighten each other's thresholds once per chunk / 256 rows:
//     through a per-query bound in global memory (returning atomicMin + relaxed agent-scope load; a stale bound is only
//     looser, never wrong), or - queue plans with many workers per query - by folding their pair into the result slot
//     and reading its 2nd key back: the exact 2nd-best distance of everything folded in so far (share_union).
//   * a chunk that starts before anybody has published a bound for its queries
//     (the first dispatch round of a big search, every block of a small one)
//     folds its first rows - 128, or the whole chunk of a small train set - into
//     its top-2 UNFILTERED: no compare, no ballot, no branch per row (a wave
//     takes the update path when any of its 64 lanes improves, i.e. on about
//     128 / s of its rows after s rows: early on the filter rejects nothing).
//   * every block folds its top-2 into a per-query 64-bit slot with two 32-bit
//     atomic minima (1st key; then the loser or the own 2nd key); the last
//     block to arrive for a query block (agent-scope ticket, no release fence:
//     every contribution is a returned memory-side atomic; pinned on the ISA by
//     tests/test_isa_handoff_cpu.py) decodes the slots to (int32 idx, int32 dist),
//     makes the selections that need no reduction over the queries (bf_select) and
//     restores the merge state, so a call is ONE kernel: no memset, no partial
//     tables, no merge kernel.
//   * several independent searches can share one launch (bf_top2_batch_kernel).
#include "internal.h"
#include <chrono>
#include "bf_scan_sgpr.h"
#include <stdio.h>
#include <vector>
#include <atomic>

typedef uint32_t u32;

#define SLAM_TILE_ROWS 256          // train rows per LDS tile (8 KiB)
#define SLAM_COLD_ROWS 128          // rows folded in unfiltered at a cold chunk start (make_plan; slam_bf_set_tuning [6])
#define SLAM_GROUP_PAIRS 16         // (train rows) x (queries per lane) covered by one filter test
#define SLAM_KEY_IDX_BITS 23
#define SLAM_KEY_IDX_MASK 0x7FFFFFu
#define SLAM_KEY_NONE 0xFFFFFFFFu
#define SLAM_ACC_BIAS 0x80000000u
#define SLAM_BOUND_IDLE 0x7F7F7F7Fu  // bound[] between launches (one byte pattern: restored by memset as well)
#define SLAM_L2_RESIDENT_ROWS 65536 // train sets up to this many rows (2 MiB) stay in every XCD's 4 MiB L2 once touched
#define SLAM_CURSOR_STRIDE 32       // u32 words between the cursors of two query waves (one 128-byte line each)
#define SLAM_BF_RESIDENT 6          // blocks of bf_top2_kernel<1, true, true> a CU holds at once (106 SGPRs: 6 waves per SIMD); queue plans

// D = popcount(x) + acc in ONE instruction.  Written as asm because hipcc
// re-associates __builtin_popcount(x)+acc chains into bcnt(x,0)+v_add3 (20
// instead of 16 VALU ops per pair).
__device__ __forceinline__ u32 bcnt_acc(u32 x, u32 acc) {
    u32 d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}
__device__ __forceinline__ u32 umed3(u32 a, u32 b, u32 c) { return max(min(a, b), min(max(a, b), c)); }

// LDS byte address of a __shared__ object, and a 16-byte LDS read at address + constant (becomes the offset: field)
__device__ __forceinline__ u32 lds_addr(const uint4* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u32)(uintptr_t)(const __attribute__((address_space(3))) uint4*)p;
#else
    return 0;
#endif
}
__device__ __forceinline__ uint4 lds_read16(u32 addr, int byte_offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const uint4 lds_u4;
    return *(lds_u4*)(uintptr_t)(addr + byte_offset);
#else
    return make_uint4(0, 0, 0, 0);
#endif
}

// distances of one train row (a = words 0-3, b = words 4-7) to the lane's R queries, biased:
// acc[r] = 2^31 - th[r] + d  =>  d < th[r]  <=>  acc[r] < 2^31 (sign bit clear)
//
// R = 1 issues the row as ONE block: the eight v_xor first (each overwrites its train word, which is dead afterwards,
// so no extra registers), then the eight accumulating v_bcnt, with the wave's PRIORITY raised for the v_bcnt half.
//
// Why the priority (round 3, tools/ubench/cycles.hip, profiles/r03_ubench_cycles.log - all in measured shader cycles,
// s_memtime against s_memrealtime, in-kernel clock 2.35-2.39 GHz): a gfx950 SIMD executes a wave64 VALU instruction
// in 4 cycles and can run TWO at once when they come from different waves and at least one of them is a "simple" op
// (v_xor/and/or/add/mov ...): a stream of v_xor from 8 waves costs 2.08 cycles per instruction, a stream of v_bcnt
// 4.05 - but four waves of pure v_xor next to four waves of pure v_bcnt cost 2.17 per instruction TOGETHER when the
// v_bcnt waves have the higher priority (2.65 with equal priorities, 3.6 when the v_xor waves have it).  v_bcnt can
// only use one of the two issue slots; whoever is served first takes that slot, so a v_xor served first blocks the
// v_bcnt of the next wave for 4 cycles while the second slot idles.  Arbitration is by priority, then age: with equal
// priorities the oldest wave is served first whatever it holds, and every bare loop of this row measured 3.7-4.0 cycles
// per instruction (59-64 per row; the round-2 kernel: 54.9, its waves drift apart at waits and branches).  With
// s_setprio 2 in front of the v_bcnt half and s_setprio 0 in front of the v_xor half, a wave in its v_bcnt half is
// served first and the v_xor of the other waves fill the second slot: the same bare loop runs at 2.24 cycles per
// instruction (35.8 per row), and this kernel went from 1.514 to 1.24-1.26 ms at 64k x 64k (profiles/r03_ab_setprio.log).
// The priority stays raised through the LDS reads and the filter that follow the row (placing s_setprio 0 right
// behind the last v_bcnt measured 2 % slower at 64k x 64k) and is dropped when the scan ends.
template <int R>
__device__ __forceinline__ void row_acc(const u32 (&q)[R][8], uint4 a, uint4 b, const u32 (&init)[R],
                                        u32 (&acc)[R]) {
    if constexpr (R == 1) {
        asm volatile("s_setprio 0\n\t"
            "v_xor_b32 %1, %9, %1\n\tv_xor_b32 %2, %10, %2\n\tv_xor_b32 %3, %11, %3\n\tv_xor_b32 %4, %12, %4\n\t"
            "v_xor_b32 %5, %13, %5\n\tv_xor_b32 %6, %14, %6\n\tv_xor_b32 %7, %15, %7\n\tv_xor_b32 %8, %16, %8\n\t"
            "s_setprio 2\n\t"
            "v_bcnt_u32_b32 %0, %1, %17\n\tv_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %0, %3, %0\n\t"
            "v_bcnt_u32_b32 %0, %4, %0\n\tv_bcnt_u32_b32 %0, %5, %0\n\tv_bcnt_u32_b32 %0, %6, %0\n\t"
            "v_bcnt_u32_b32 %0, %7, %0\n\tv_bcnt_u32_b32 %0, %8, %0"
            : "=v"(acc[0]), "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w)
            : "v"(q[0][0]), "v"(q[0][1]), "v"(q[0][2]), "v"(q[0][3]), "v"(q[0][4]), "v"(q[0][5]), "v"(q[0][6]),
              "v"(q[0][7]), "v"(init[0]));
        return;
    }
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][0] ^ a.x, init[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][1] ^ a.y, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][2] ^ a.z, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][3] ^ a.w, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][4] ^ b.x, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][5] ^ b.y, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][6] ^ b.z, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][7] ^ b.w, acc[r]);
}

// Filter + update for U consecutive train rows.  "Some pair of this lane improved" <=> the AND of
// all U*R accumulators has its sign bit clear: v_and_b32 is a 2-cycle op where v_min/v_cmp are 4,
// and one compare + one wave-uniform branch (on the ballot, update block laid out as unlikely)
// serves U rows.  The thresholds stay fixed inside a group, which only makes them looser.
template <int R, int U>
__device__ __forceinline__ void filter_update(const u32 (&acc)[U][R], u32 first_train_idx, u32 (&b1)[R],
                                              u32 (&b2)[R], u32 (&init)[R]) {
    u32 m = acc[0][0];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int r = 0; r < R; r++)
            if (u || r) m &= acc[u][r];
    if (__builtin_expect(__ballot((int)m >= 0) != 0ull, 0)) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            // usually one row of the group is responsible: skip the others (wave-uniform again)
            u32 mu = acc[u][0];
#pragma unroll
            for (int r = 1; r < R; r++) mu &= acc[u][r];
            if (U > 1 && __ballot((int)mu >= 0) == 0ull) continue;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u32 key = ((acc[u][r] - init[r]) << SLAM_KEY_IDX_BITS) | (first_train_idx + u);
                b2[r] = umed3(b1[r], b2[r], key);   // second smallest of {b1, b2, key}
                b1[r] = min(b1[r], key);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++)   // tighten only: may already hold a smaller bound learnt from other chunks
            init[r] = max(init[r], SLAM_ACC_BIAS - (b2[r] >> SLAM_KEY_IDX_BITS));
    }
}

// The unfiltered form for the first rows of a chunk that starts without any bound: every row is folded into (b1, b2),
// no test, no ballot, no branch.  A wave takes the update path for a row when ANY of its 64 lanes improves, i.e. on about
// min(1, 128 / s) of its rows once s rows stand behind the thresholds: until a few hundred rows are known the filter
// rejects nothing and costs a compare and a branch per row (a lone wave scanning a cold 256-row chunk measured 216 cycles
// per row, profiles/r03_cycles.log 4096 x 4096, against 64 for its sixteen instructions).  The accumulators of these rows
// start at 0, so the key is (acc << 23) | index: three instructions per row and pair.
template <int R, int U>
__device__ __forceinline__ void insert_rows(const u32 (&acc)[U][R], u32 first_train_idx, u32 (&b1)[R], u32 (&b2)[R]) {
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u32 key = (acc[u][r] << SLAM_KEY_IDX_BITS) | (first_train_idx + u);
            b2[r] = umed3(b1[r], b2[r], key);
            b1[r] = min(b1[r], key);
        }
}

// Exchange the 2nd-best distance of the lane's queries with the blocks scanning other train chunks.
// bound[q] only ever holds the 2nd-best distance over some subset of the train rows, i.e. an upper
// bound of the final 2nd-best distance g: rows with d > g can be dropped, rows with d == g must stay
// (they may win the tie on index), hence the "+ 1".  The load may be stale; that only loosens it.
// (Measured and dropped: issuing the load at one share point and consuming it at the next, so that a wave never
// sits on the round trip, makes every applied bound one segment staler - 64k x 64k 1528 -> 1551 us,
// 8192 x 65536 212 -> 222 us, 2000 x 2000 21 -> 33 us.)
// gk[r] keeps the last bound seen (an upper bound of the final 2nd-best distance for good: bound[] only decreases
// during a launch); the epilogue uses it to skip merges that cannot matter, without another round trip.
// (Forms measured against each other in round 4, profiles/r04_ab_queue.log "share_bound forms": the returning atomic waited for
// on the spot, round 3's load + no-return atomic, and this one with and without the "nothing before a 2nd neighbour" rule - all
// within 1 % of each other; this is the one whose writes are known to be complete.)
template <int R>
__device__ __forceinline__ void share_bound(u32* __restrict__ bound, int qbase, int N, const u32 (&b2)[R],
                                            u32 (&init)[R], u32 (&gk)[R], u32 (&pend)[R]) {
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int qi = qbase + r * 64;
        if (qi < N) {
            const u32 own = b2[r] >> SLAM_KEY_IDX_BITS;
            // The bound is READ with a relaxed agent-scope load.  A lane whose own 2nd-best beats it publishes it with an
            // atomic minimum in the RETURNING form, whose value nobody waits for here: it is parked in pend[] and consumed
            // at the lane's next exchange or in front of the arrival ticket (vector-memory operations return in issue
            // order, so by then it has long arrived).  Returning on purpose: every write to bound[] is then known to have
            // been performed at the memory side before the block's ticket is drawn, so none can land behind the last
            // arriver's reset (a no-return atomic is only known to have been sent).
            asm volatile("" ::"v"(pend[r]));
            const u32 g = __hip_atomic_load(&bound[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Nothing is published while the lane has no 2nd neighbour yet (own = 511): a block that starts must still see
            // "nobody has published" (SLAM_BOUND_IDLE) then.
            if (own < g && own <= 256u) pend[r] = atomicMin(&bound[qi], own);
            gk[r] = g;
            init[r] = SLAM_ACC_BIAS - min(own, g + 1);
        }
    }
}

// The exchange as a MERGE (round 4): instead of publishing its own 2nd-best distance, a lane whose pair (b1, b2) changed since
// it last did so folds the pair into best[q] - the two atomic minima of the final merge (see the epilogue) - and takes the
// slot's 2nd key as the bound: the exact 2nd-best distance over EVERYTHING the blocks of this query have folded in so far,
// where the minimum of their own 2nd-best distances is that of a set k times smaller (k blocks that have scanned s rows each:
// the quantile 2 / (k s) instead of about sqrt(1.4 / k) / s - at k = 48, the 1/8 shard's workers, what the old bound reaches
// after 4 s rows).  Folding a pair twice is harmless: keys are distinct rows, a lane that meets its own 1st key in the slot
// (o1 == b1) pushes no loser, and every push is a real key other than the smallest one.  A lane whose pair did not change
// reads the 2nd key with a relaxed agent-scope load, as it read the bound before.
// sent[r]: the lane's b2 at its last fold (b2 changes whenever the pair does: a new key either becomes b2 or moves b1 there).
template <int R>
__device__ __forceinline__ void share_union(unsigned long long* __restrict__ best, int qbase, int N, const u32 (&b1)[R],
                                            const u32 (&b2)[R], u32 (&init)[R], u32 (&gk)[R], u32 (&sent)[R]) {
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int qi = qbase + r * 64;
        if (qi < N) {
            u32* half = (u32*)&best[qi];                          // little endian: [0] = 2nd key, [1] = 1st key
            u32 k2;
            if (b2[r] != sent[r]) {
                const u32 o1 = atomicMin(half + 1, b1[r]);
                const u32 loser = o1 == b1[r] ? SLAM_KEY_NONE : max(o1, b1[r]);
                const u32 push = min(loser, b2[r]);                // b2 != sent implies b2 != none: there is something to push
                k2 = min(atomicMin(half, push), push);
                sent[r] = b2[r];
            } else {
                k2 = __hip_atomic_load(half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const u32 g = k2 >> SLAM_KEY_IDX_BITS;                 // 511 while the slot holds fewer than two keys
            gk[r] = g;
            init[r] = SLAM_ACC_BIAS - min(b2[r] >> SLAM_KEY_IDX_BITS, g + 1);
        }
    }
}

// Per-query merge state shared by the blocks of one launch (device memory owned by the ctx).
// Invariant between launches: best[] = ~0, bound[] = 0x7F7F7F7F, arrivals[] = 0, cursor[] = 0; the last block
// of every query block restores it after decoding, so no memset or merge kernel runs per call.
struct bf_state {
    unsigned long long* best;   // [N]  (1st key << 32 | 2nd key), keys = dist << 23 | train row, ~0 = none
    u32* bound;                 // [N]  upper bound of the final 2nd-best distance (share_bound)
    u32* arrivals;              // [query blocks]  how many chunk blocks have merged their result
    u32* cursor;                // [4 x query blocks] x SLAM_CURSOR_STRIDE  queue plans: the next chunk ticket of each query wave (0
                                // between launches), every cursor on a 128-byte line of its own: returning atomics on one line
                                // serialise at the memory side (with the 128 cursors of the 1/8 shard packed into four lines a
                                // 256-row chunk cost 168 us where one block per chunk took 157: profiles/r04_ab_queue.log)
};

// Selection fused into the decode (round 4): the last arriver of a query block has (k1, k2) of its queries
// in hand, so the selections that need NO reduction over the queries - "has a neighbour" (bf.match, feature_matchers.py:39,44)
// and the Lowe ratio test - are made there: one flag per query, and how many each wave kept (a plain store per wave: nothing
// to zero beforehand, the host adds them up).  A search + ratio test on resident rows is ONE launch instead of two.
struct bf_select {
    uint8_t* keep;        // [N] 1 = kept; null = no selection
    int* wave_kept;       // [ceil(N / 64)] rows kept in each group of 64 consecutive queries (device or pinned host memory)
    double param;         // mode 2: the ratio
    int mode;             // 0 = keep every query that has a neighbour, 2 = Lowe ratio: dist0 < param * dist1 (needs two neighbours)
    unsigned epoch;       // what the last arriver of query block x stores into done[x] once its results are visible to the host
    unsigned* done;       // pinned host memory, or null: the caller synchronises the stream instead of polling (slam_wait_done)
};

// grid.x = query blocks of 256*R rows, grid.y = train chunks: block (x, y) scans rows [tbl[y], tbl[y+1]).  Every block
// merges its top-2 into st.best; the last block to arrive for a query block decodes (idx + train_base, dist) and
// restores the merge state.  Blocks with y < lead are the leaders: short chunks at the head of the dispatch order that,
// after their merge, publish the exact 2nd-best distance of everything merged so far.  Nobody waits for them: the other
// blocks pick the bound up at their next share point.  Measured and dropped: holding the other blocks back until their
// leaders are done (-15..20 %: the leaders alone cannot keep the SIMDs busy) and raising the leaders' wave priority
// with s_setprio (no change: as the oldest waves of their SIMDs they are served first anyway).
// (bx, by) = the block's place in its search's grid of qblocks x S blocks: blockIdx of a single search, decoded from the
// linear block index of a batch launch (bf_top2_batch_kernel).
template <int R, bool SFEED, bool QUEUE>
__device__ __forceinline__ void bf_top2_block(const uint4* __restrict__ q, int N, const uint4* __restrict__ t,
                                              const int* __restrict__ tbl, int lead, bf_state st, int train_base,
                                              int2* __restrict__ out_idx, int2* __restrict__ out_dist,
                                              uint4* __restrict__ keep, const int bx, const int by, const int S,
                                              const int cold_arg, const int uni_arg, const int M, const int nchunks, const int merge_arg,
                                              const bf_select sel) {
    // QUEUE is a template parameter, not a run-time flag: with both forms in one kernel the one-block-per-chunk plans ran
    // 1-2.5 % slower than without the queue code (30 SGPR spills instead of 3; profiles/r04_ab_queue.log).  A queue plan has
    // a boundary table, exchanges bounds and has no leaders.
    static_assert(!QUEUE || (SFEED && R == 1), "queue plans run the SGPR-fed scan at one query per lane");
    const int uni = QUEUE ? 0 : uni_arg;
    __shared__ uint4 tile[2][SLAM_TILE_ROWS * 2 + 4];
    __shared__ u32 s_last;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int qbase = bx * (256 * R) + wave * (64 * R) + lane;
    const bool leader = !QUEUE && by < lead;
    // cold < 0 (make_plan): every chunk fits into its unfiltered start of -cold rows and the whole grid is resident at
    // once - no block could ever use a bound published by another, so none is read or written (one round trip less in
    // front of the scan).
    const bool nobound = !QUEUE && cold_arg < 0;
    const int cold = cold_arg < 0 ? -cold_arg : cold_arg;

    u32 qr[R][8];
#pragma unroll
    for (int r = 0; r < R; r++) {
        int qi = qbase + r * 64;
        qi = qi < N ? qi : N - 1;  // clamp: tail lanes compute a duplicate and never store
        const uint4 a = q[2 * (size_t)qi], b = q[2 * (size_t)qi + 1];
        if (keep && by == 0 && qbase + r * 64 < N) {
            // the caller wants the query rows left in device memory (they are the next frame's train side,
            // frontend.py:181-187): the first chunk's blocks have them in registers anyway
            keep[2 * (size_t)qi] = a;
            keep[2 * (size_t)qi + 1] = b;
        }
        qr[r][0] = a.x; qr[r][1] = a.y; qr[r][2] = a.z; qr[r][3] = a.w;
        qr[r][4] = b.x; qr[r][5] = b.y; qr[r][6] = b.z; qr[r][7] = b.w;
    }
    // How the blocks of a query exchange what they know: as a bound (share_bound) or as a merge (share_union; queue plans
    // with many workers per query block - a wave-uniform run-time choice inside the queue kernel only).
    const bool merging = QUEUE && merge_arg != 0;
    const u32 nobody = merging ? (SLAM_KEY_NONE >> SLAM_KEY_IDX_BITS) : SLAM_BOUND_IDLE;   // what gk reads while nothing is published
    u32 b1[R], b2[R], init[R], gk[R], pend[R];                     // pend: parked atomic returns (bound form) / b2 at the last fold (merge form)
#pragma unroll
    for (int r = 0; r < R; r++) {
        pend[r] = merging ? SLAM_KEY_NONE : 0u;
        b1[r] = SLAM_KEY_NONE;
        b2[r] = SLAM_KEY_NONE;
        init[r] = SLAM_ACC_BIAS - (SLAM_KEY_NONE >> SLAM_KEY_IDX_BITS);  // "distance 511": everything enters
        gk[r] = nobody;
    }
    auto share = [&]() {
        if (merging) share_union<R>(st.best, qbase, N, b1, b2, init, gk, pend);
        else share_bound<R>(st.bound, qbase, N, b2, init, gk, pend);
    };
    // A chunk starts COLD when nobody has published a bound for any of the wave's queries yet (the first dispatch round,
    // every block of a frame-sized search): its first `cold` rows are then folded in without a filter (insert_rows).
    // Blocks of later rounds pick up a bound at their first exchange and go straight to the filtered scan.
    auto nobody_published = [&]() -> bool {
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; r++) any = any || gk[r] != nobody;
        return __ballot(any) == 0ull;
    };
    auto tighten = [&]() {
#pragma unroll
        for (int r = 0; r < R; r++) init[r] = max(init[r], SLAM_ACC_BIAS - (b2[r] >> SLAM_KEY_IDX_BITS));
    };

    // What the epilogue needs of the kernel arguments, parked in VECTOR registers in front of the scan: the SGPR-fed scan statement
    // clobbers two thirds of the SGPRs, so the compiler would fetch N, the grid's height and the result pointers again from the
    // argument segment behind it - scalar loads on the hand-off's critical path whose lines the scan's row stream has pushed out of
    // the scalar cache by then (200 x 1000: 7.2 us against 6.3 in round 3, whose kernel had fewer arguments; profiles/
    // r04_ab_queue.log "parked arguments").  Every use below is a per-lane compare or address, so vector registers serve as they are.
    int N_e = N, S_e = S, base_e = train_base;
    int2* oi_e = out_idx;
    int2* od_e = out_dist;
    if constexpr (SFEED) {
        asm volatile("" : "+v"(N_e), "+v"(S_e), "+v"(base_e));
        asm volatile("" : "+v"(oi_e), "+v"(od_e));
    }
    // uni > 0: the plan is `uni` rows per chunk throughout (single-round launches: frame-sized searches and everything up
    // to a few thousand rows a side) - no boundary table to read, and none to upload in front of the launch
    if constexpr (SFEED) {
        // Train rows reach the lanes through SGPRs (SLAM_SCAN_GROUPS_ASM, bf_scan_sgpr.h): no LDS tile, no barrier, every
        // wave scans on its own.  The bound is exchanged where the LDS form does it: at the start, after 16 / 32 / 64 /
        // 128 rows of a wave's FIRST chunk (it starts with no threshold of its own, so early on nearly every group takes
        // the update path and what the sibling blocks have scanned meanwhile tightens it), then once per 256 rows.
        // The scalar loads run only four rows ahead, which covers an L2 hit but not a miss: each wave therefore TOUCHES
        // the 128-byte lines of the 256 rows after the stretch it is scanning with one vector load (lanes 0-15 of wave w:
        // lines 16 w .. 16 w + 15 of those 8 KiB) - what the LDS form's tile loads do as a side effect.  Measured
        // (profiles/r03_ab_sgpr_feed.log): 1000 x 3000 17.5 -> 13.9 us, 8192 x 8192 54.6 -> 48.9 us with the touch.
        //
        // QUEUE plans (nchunks > 0; round 4): grid.y is not the number of chunks but the number of WORKER blocks per query
        // block - as many as are resident at once - and every wave draws its chunks itself: a ticket from the cursor of
        // its query wave (one returning agent-scope atomic add, issued behind a chunk and waited for together with the
        // exchange in front of the next), chunk = tbl[ticket] .. tbl[ticket + 1], until the tickets run out.  A wave keeps its top-2 and
        // its threshold from chunk to chunk: ONE start without a threshold, ONE merge and ONE arrival per worker instead
        // of one per chunk, no dispatch of a new block between chunks, and the waves of a SIMD stay busy until the queue
        // of their query wave is empty - the chunks shrink towards its end (make_plan_core), so they all run out of work
        // within a few dozen rows of each other.  Nobody waits for anybody: the only exit condition is ticket >= nchunks,
        // which every wave reaches because the cursor only grows.  Blocks that are dispatched after the queue has run
        // dry (a device shared with another process) draw a ticket past the end, merge nothing and arrive.
        static_assert(!SFEED || R == 1, "the SGPR-fed scan holds one query per lane");
        const char* tbytes = (const char*)t;
        constexpr bool queue = QUEUE;
        u32* const cursor = st.cursor + (size_t)(4 * bx + wave) * SLAM_CURSOR_STRIDE;
        // One round trip per chunk: a queue worker draws its next ticket right BEHIND a chunk (not a chunk ahead: a wave that
        // sits on a drawn chunk while it scans another keeps it from a wave that has run dry - tickets one chunk ahead
        // measured 1045 against 1022 us at 64k x 64k, 534 against 522 at 32768 x 65536) and resolves it behind the bound
        // exchange of the next one, whose wait covers the draw; a block plan touches its chunk's first lines in front of the
        // exchange instead (its chunk is known from blockIdx).  Forms measured: profiles/r04_ab_queue.log "queue loop forms".
        // (Nothing but the one touch of the rows behind the current stretch may be in flight when a scan statement starts:
        // with a second vector load outstanding the compiler puts an s_waitcnt vmcnt(0) in front of the statement.)
        auto draw = [&]() -> u32 {
            u32 v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add(cursor, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return v;
        };
        int ci = by;
        u32 ticket = 0;
        if (queue) ticket = draw();
        bool fresh = true;                              // the wave has no threshold of its own yet (its first chunk)
        int chunks_done = 0;
        while (true) {
            int c0 = 0, c1 = 0;
            if (!queue) { c0 = uni ? ci * uni : tbl[ci]; c1 = uni ? min(M, c0 + uni) : tbl[ci + 1]; }
            auto touch = [&](int first) -> u32 {
                // block plans: the four waves of a block share a chunk, each touches a quarter of the lines; queue plans:
                // every wave has a chunk of its own and touches all 64 lines
                const long long off = (long long)first * SLAM_DESC_BYTES + (queue ? lane : wave * 16 + lane) * 128;
                u32 v = 0;
                if ((queue || lane < 16) && off < (long long)c1 * SLAM_DESC_BYTES) v = *(const u32*)(tbytes + off);
                return v;
            };
            u32 warm = 0;
            if (!queue) warm = touch(c0);
            // Queue workers that exchange through bound[] (few workers per query: each has thousands of rows of its own) do so
            // on their first four chunks and then on every 4th: the reads are agent-scope loads of lines that the publishing
            // atomics keep dropping from the L2s, i.e. fabric traffic for nothing once the thresholds have settled (64k x 64k:
            // FETCH_SIZE 49.0 -> 34.3 MB raw per launch at 1018.7 -> 1016.4 us; every 16th: 27.9 MB but 1024.5 us;
            // profiles/r04_ab_queue.log "exchange").  The merge form keeps every chunk (the shard is where thresholds matter).
            const bool exchange = !queue || merging || chunks_done < 4 || (chunks_done & 3) == 0;
            chunks_done++;
            if (!nobound && exchange) share();
            if (queue) {
                ci = __builtin_amdgcn_readfirstlane(ticket);      // drawn behind the previous chunk; its wait was the exchange's
                if (ci >= nchunks) break;
                c0 = tbl[ci];
                c1 = tbl[ci + 1];
                // the chunk's first lines: on a wave's first chunk, and on every chunk of a train set that does not stay in an
                // XCD's L2 (4 MiB; beyond that the front of the queues keeps moving into rows no L2 holds, and without the touch
                // the scan's first scalar loads of every chunk wait for HBM: 131072 x 2^20, the per-rank problem of the loop-closure
                // run, 34.5 ms without against 31.7 ms in round 3)
                if (fresh || M > SLAM_L2_RESIDENT_ROWS) warm = touch(c0);
            }
            asm volatile("" ::"v"(warm));                        // the chunk's first lines are there: the scan starts right away
            int row = c0;
            bool is_cold = fresh && cold >= 16 && (nobound || nobody_published());
            bool shared = true;                            // the exchange in front of the chunk's first stretch is done
            while (row < c1) {
                if (!shared && !nobound) share();
                shared = false;
                const int done = row - c0;
                is_cold = is_cold && done == 0;
                const int seg = is_cold ? cold
                                        : (!fresh ? SLAM_TILE_ROWS : (done < 16 ? 16 : (done < SLAM_TILE_ROWS ? done : SLAM_TILE_ROWS)));
                const int end = min(row + seg, c1);
                warm = (done == 0 || done >= SLAM_TILE_ROWS) ? touch(row + SLAM_TILE_ROWS) : 0u;
                int ng = __builtin_amdgcn_readfirstlane((end - row) >> 4);
                if (ng > 0) {
                    const uint4* tp = t + 2 * (size_t)row;
                    u32 idx = (u32)__builtin_amdgcn_readfirstlane(row);
                    row += ng << 4;
                    if (is_cold) {
                        SLAM_SCAN_GROUPS_COLD_ASM(qr[0], tp, ng, idx, b1[0], b2[0]);
                        tighten();
                    } else {
                        SLAM_SCAN_GROUPS_ASM(qr[0], tp, ng, idx, b1[0], b2[0], init[0]);
                    }
                }
                is_cold = false;
                for (; row < end; row++) {                 // fewer than 16 rows left: only at the end of the train set
                    const uint4 x0 = t[2 * (size_t)row], y0 = t[2 * (size_t)row + 1];
                    u32 acc1[1][R];
                    row_acc<R>(qr, x0, y0, init, acc1[0]);
                    filter_update<R, 1>(acc1, (u32)row, b1, b2, init);
                }
                asm volatile("" ::"v"(warm));              // the touch has landed (this keeps its register until here)
            }
            if (!queue) break;
            fresh = false;
            ticket = draw();
        }
    } else {
        const int t0 = uni ? by * uni : tbl[by];
        const int t1 = uni ? min(M, t0 + uni) : tbl[by + 1];
        // prologue: first tile -> LDS buffer 0
    #pragma unroll
        for (int i = 0; i < 2; i++) {
            const int g = 2 * t0 + tid + i * 256;
            tile[0][tid + i * 256] = g < 2 * t1 ? t[(size_t)g] : make_uint4(0, 0, 0, 0);
        }
        __syncthreads();

        int buf = 0;
        for (int tb = t0; tb < t1; tb += SLAM_TILE_ROWS) {
            // issue the next tile's global loads before computing on this one
            const int nb = tb + SLAM_TILE_ROWS;
            uint4 nxt[2];
            if (nb < t1) {
    #pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int g = 2 * nb + tid + i * 256;
                    nxt[i] = g < 2 * t1 ? t[(size_t)g] : make_uint4(0, 0, 0, 0);
                }
            }
            if (!nobound) share();
            const int cnt = __builtin_amdgcn_readfirstlane(min(SLAM_TILE_ROWS, t1 - tb));
            const uint4* tp = tile[buf];
            // software pipeline: the next row is read from LDS while the current one is computed (the last
            // read-ahead lands in the tile's padding and is discarded).  Measured: letting the compiler batch
            // a whole group's reads behind immediate offsets is 6 % slower than this rolling form.
            int j = 0;
            uint4 a0 = tp[0], c0 = tp[1];
            constexpr int U = SLAM_GROUP_PAIRS / R;   // 16 rows at R = 1 (59 VGPRs, still 8 waves/SIMD)
            static_assert(U >= 2 && U % 2 == 0, "row pipeline handles rows in pairs");
            // one group of U rows: rows are read through ONE running LDS address in a VGPR plus immediate offsets (the tile is
            // padded by two rows so the read-ahead needs no wrap); the empty asm after every row block ties that address to the
            // row's result, so the reads stay where they are written - rolling, one row ahead - instead of being
            // batched by the scheduler (batched: 67 VGPRs, 7 waves/SIMD).  Against per-read scalar address arithmetic
            // + v_mov from SGPR this saves ~1.5 VALU and ~8 SALU instructions per row: no change where the VALU is
            // saturated (64k x 64k), 3-12 % on latency-bound sizes (4096 x 4096 30.5 -> 26.8 us).
            auto group = [&](const u32 (&ini)[R], u32 (&acc)[U][R]) {
                u32 base = lds_addr(tp + 2 * j);
#pragma unroll
                for (int u = 0; u < U; u += 2) {
                    const uint4 a1 = lds_read16(base, (2 * u + 2) * 16), c1 = lds_read16(base, (2 * u + 3) * 16);
                    row_acc<R>(qr, a0, c0, ini, acc[u]);
                    asm volatile("" : "+v"(base) : "v"(acc[u][0]));
                    a0 = lds_read16(base, (2 * u + 4) * 16);
                    c0 = lds_read16(base, (2 * u + 5) * 16);
                    row_acc<R>(qr, a1, c1, ini, acc[u + 1]);
                    asm volatile("" : "+v"(base) : "v"(acc[u + 1][0]));
                }
            };
            // A block starts with no threshold of its own.  When no sibling has published one either, its first `cold` rows
            // are folded in unfiltered (insert_rows); after that - or from the start, when a bound was there - nearly every
            // group still takes the update path for a while, and the bound is re-read after 16, 32, 64 and 128 rows (what
            // the sibling blocks have scanned meanwhile tightens it), later once per tile.
            int seg_end = tb == t0 ? 16 : cnt;
            if (tb == t0 && cold >= U && (nobound || nobody_published())) {
                const int lim = min(cold, cnt);
                u32 zero[R];
#pragma unroll
                for (int r = 0; r < R; r++) zero[r] = 0u;
                for (; j + U <= lim; j += U) {
                    u32 acc[U][R];
                    group(zero, acc);
                    insert_rows<R, U>(acc, (u32)(tb + j), b1, b2);
                }
                tighten();
                if (j < cnt && !nobound) share();
                while (seg_end <= j) seg_end *= 2;
            }
            while (true) {
                const int lim = min(seg_end, cnt);
                for (; j + U <= lim; j += U) {
                    u32 acc[U][R];
                    group(init, acc);
                    filter_update<R, U>(acc, (u32)(tb + j), b1, b2, init);
                }
                if (lim >= cnt) break;
                if (!nobound) share();
                seg_end *= 2;
            }
            for (; j < cnt; j++) {
                const uint4 x0 = tp[2 * j], y0 = tp[2 * j + 1];
                u32 acc1[1][R];
                row_acc<R>(qr, x0, y0, init, acc1[0]);
                filter_update<R, 1>(acc1, (u32)(tb + j), b1, b2, init);
            }
            if (nb < t1) {
    #pragma unroll
                for (int i = 0; i < 2; i++) tile[buf ^ 1][tid + i * 256] = nxt[i];
            }
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- epilogue: merge, then the last arriver of this query block decodes ----------------------
    __builtin_amdgcn_s_setprio(0);   // the scan raised it (row_acc); merges and tickets run at the default priority
    // Merge into best[q] = (1st key << 32 | 2nd key) with TWO 32-bit atomic minima instead of a 64-bit CAS loop: the 1st
    // key goes into the high half with a returning atomicMin; whichever of (what was there, mine) lost, or my 2nd key if
    // that is smaller, goes into the low half.  Every key is distinct, so: the high half ends as the smallest key m; m is
    // never pushed into the low half (its holder pushes min(loser, own 2nd), both > m; everybody else pushes keys > m);
    // and the second smallest key s always is (if s is somebody's 1st key it either loses against m directly or is
    // displaced by m later, and the displacing thread pushes it; if it is the 2nd key of m's holder, that thread pushes
    // it).  One round trip, no retries when the 16-64 chunk blocks of a query finish together - the CAS loop paid a
    // load plus one round trip per lost race (epilogue of a 4096 x 4096 block: 6.6 us mean, profiles/r02_block_timeline.log).
    // gk >= the final 2nd-best distance (share_bound): a block whose best row is farther than that cannot contribute and
    // skips the atomics without looking (rows AT that distance may still win the tie on index).
    if (merging) {
    // (merge form of the exchange: a lane whose pair has not changed since its last fold has nothing left to merge)
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int qi = qbase + r * 64;
        if (qi < N_e && b1[r] != SLAM_KEY_NONE && (b1[r] >> SLAM_KEY_IDX_BITS) <= gk[r] &&
            (b2[r] != pend[r] || b2[r] == SLAM_KEY_NONE)) {       // (a lone row: b2 is none, and so was it at the last fold)
            u32* half = (u32*)&st.best[qi];
            const u32 o1 = atomicMin(half + 1, b1[r]);
            const u32 loser = o1 == b1[r] ? SLAM_KEY_NONE : max(o1, b1[r]);
            const u32 push = min(loser, b2[r]);
            if (push != SLAM_KEY_NONE) {
                // returning form on purpose: "the value is back" means the minimum has been taken at the memory side, which is
                // what the arrival ticket below relies on (a no-return atomic is only known to have been sent)
                const u32 o2 = atomicMin(half, push);
                asm volatile("" ::"v"(o2));
            }
        }
    }
    } else {
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int qi = qbase + r * 64;
        if (qi < N_e && b1[r] != SLAM_KEY_NONE && (b1[r] >> SLAM_KEY_IDX_BITS) <= gk[r]) {
            u32* half = (u32*)&st.best[qi];                       // little endian: [0] = 2nd key, [1] = 1st key
            const u32 o1 = atomicMin(half + 1, b1[r]);
            const u32 push = min(max(o1, b1[r]), b2[r]);
            if (leader) {
                // leaders leave the 2nd key of everything merged so far as the bound (an upper bound of the final one:
                // the low half only ever holds keys other than the smallest)
                const u32 k2 = min(atomicMin(half, push), push);
                if ((k2 >> SLAM_KEY_IDX_BITS) < gk[r]) {           // returning form, like every write to bound[] (share_bound)
                    const u32 ob = atomicMin(&st.bound[qi], k2 >> SLAM_KEY_IDX_BITS);
                    asm volatile("" ::"v"(ob));
                }
            } else if (push != SLAM_KEY_NONE) {
                // returning form on purpose: "the value is back" means the minimum has been taken at the memory side, which is
                // what the arrival ticket below relies on (a no-return atomic is only known to have been sent)
                const u32 o2 = atomicMin(half, push);
                asm volatile("" ::"v"(o2));
            }
        }
    }
    }
    // Arrival ticket.  Everything a block contributes travels in agent-scope INTEGER atomics in the RETURNING form
    // (global_atomic_umin ... sc0: the merges above, every write to bound[] in share_bound / share_union), and what is relied
    // on is this, no more:
    //   (1) a returned value means the read-modify-write has been performed at the point all XCDs share - an agent-scope
    //       atomic is never satisfied from an XCD's own L2 (the guide measures that for float atomics, "Global float atomics";
    //       for the integer minima used here it is what 140 000 + 46 000 fuzzed searches and the state soaks of rounds 3 / 4
    //       show, and what makes cross-XCD merges come out right at all), and nothing of it stays dirty in an L2;
    //   (2) every wave waits for all its returns (s_waitcnt vmcnt(0): vector-memory operations return in issue order), then
    //       the block barrier, then ONE lane's relaxed agent-scope ticket - the guide's hand-off table, first row: "an
    //       agent-scope atomic add ... after every storing wave's vmcnt(0) wait ... the workgroup whose add came last, told by
    //       the value its add returned";
    //   (3) the block that draws the last ticket passes a barrier and TAKES the slots with returning 8-byte atomic exchanges
    //       (which also put them back to idle): agent-scope atomics on BOTH sides of the hand-off - the guide's "{8-B agent
    //       atomics both sides}" form - so no load is involved that an L1 or an XCD's L2 could serve, and no acquire fence is
    //       needed (round 3 read the slots with sc1 loads behind a buffer_inv sc1: 4096 x 4096 16.2 -> 15.6 us without it, other
    //       sizes equal; profiles/r04_ab_queue.log "swap").
    // The G16 counter recipe has an agent-scope RELEASE fence (buffer_wbl2 sc1) in front of the ticket; it writes back dirty L2
    // lines, of which this hand-off has none, and cost 1.7 us on the critical path of every block: 4096 x 4096 18.4 -> 16.7 us,
    // 2000 x 2000 10.7 -> 9.4 us, the 1/8 shard 159.9 -> 154.8 us without it (profiles/r03_ab_cold_start.log).
    // tests/test_isa_handoff_cpu.py holds every build to (1)-(3) on the disassembly (return forms, wait and barrier in front
    // of the ticket, returning exchanges behind it; profiles/r04_isa_handoff_excerpt.txt), and every search leaves the
    // whole state idle, which the GPU tests assert on the state itself (slam_bf_state_dirty).  Placement-independent: nothing
    // relies on which XCD a block runs on.
    if (!merging) {
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("" ::"v"(pend[r]));   // the parked returns of share_bound
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const u32 ticket = __hip_atomic_fetch_add(&st.arrivals[bx], 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == (u32)S_e - 1 ? 1u : 0u;
        asm volatile("" ::"s"(ticket));                 // (the ticket is back: its value was just used)
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int qi = qbase + r * 64;
        int kept = 0;
        if (qi < N_e) {
            // every contribution was made by an agent-scope atomic; it is taken - and the slot put back to idle for the next
            // launch - by one more (all other blocks are done with these queries)
            const unsigned long long v = __hip_atomic_exchange(&st.best[qi], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 k1 = (u32)(v >> 32), k2 = (u32)v;
            int2 oi, od;
            oi.x = k1 == SLAM_KEY_NONE ? SLAM_NO_MATCH_IDX : (int)(k1 & SLAM_KEY_IDX_MASK) + base_e;
            od.x = k1 == SLAM_KEY_NONE ? SLAM_NO_MATCH_DIST : (int)(k1 >> SLAM_KEY_IDX_BITS);
            oi.y = k2 == SLAM_KEY_NONE ? SLAM_NO_MATCH_IDX : (int)(k2 & SLAM_KEY_IDX_MASK) + base_e;
            od.y = k2 == SLAM_KEY_NONE ? SLAM_NO_MATCH_DIST : (int)(k2 >> SLAM_KEY_IDX_BITS);
            oi_e[qi] = oi;
            od_e[qi] = od;
            if (sel.keep) {
                // (double) comparisons of integers <= 256: exact, and the same arithmetic as filter_keep_kernel
                const bool k = oi.x >= 0 && (sel.mode == 0 || (oi.y >= 0 && (double)od.x < sel.param * (double)od.y));
                sel.keep[qi] = k ? 1 : 0;
                kept = k ? 1 : 0;
            }
            if (!merging) __hip_atomic_store(&st.bound[qi], SLAM_BOUND_IDLE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (sel.keep && qi - lane < N_e) {                   // (wave-uniform: the group's first query exists)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
            if (lane == 0) sel.wave_kept[qi >> 6] = kept;
        }
    }
    if (tid == 0) __hip_atomic_store(&st.arrivals[bx], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (QUEUE && tid < 4)   // every worker of this query block has arrived, so nobody draws a ticket any more
        __hip_atomic_store(&st.cursor[(size_t)(4 * bx + tid) * SLAM_CURSOR_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__builtin_expect(sel.done != nullptr, 0)) {   // (laid out behind the common path: tests/test_isa_handoff_cpu.py reads the text in order)
        // A host thread polls done[bx] instead of synchronising the stream (frame-sized calls: 4 us less per call,
        // tools/ubench/sync_vs_poll.hip).  Every thread makes its own result stores visible at system scope, the block meets,
        // one lane releases the flag.  The state restores above need no such care: the next launch is ordered behind this
        // one by the stream.
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&sel.done[bx], sel.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int R, bool SFEED, bool QUEUE>
__global__ __launch_bounds__(256) void bf_top2_kernel(const uint4* __restrict__ q, int N,
                                                      const uint4* __restrict__ t, const int* __restrict__ tbl,
                                                      int lead, bf_state st, int train_base,
                                                      int2* __restrict__ out_idx, int2* __restrict__ out_dist,
                                                      uint4* __restrict__ keep, int cold, int uni, int M,
                                                      int nchunks, int merge, bf_select sel) {
    // Queue plans rotate the query blocks from one row of workers to the next: worker y of query block x is block
    // ((x - y) mod grid.x, y).  Blocks go to the XCDs round robin by their linear index, so without the rotation ALL workers
    // of a query block sit on one XCD (grid.x is a multiple of 8 for the big grids) - and the XCDs do not run at the same
    // pace: the queues of the query blocks of one XCD ran dry 5 % later than those of another (65536 x 65536: 1070 .. 1125 us
    // by XCD, profiles/r04_block_timeline.log), and nobody can help a queue but its own workers.  Rotated, every queue is
    // served from all XCDs and the faster ones simply draw more tickets.  The dispatch order, hence the age mix of a
    // queue's workers (the VALU arbiter serves older waves first), stays what it was: row y of the grid is worker y of
    // every query block.
    int bx = (int)blockIdx.x;
    if (QUEUE) bx = (bx + (int)blockIdx.y) % (int)gridDim.x;
    bf_top2_block<R, SFEED, QUEUE>(q, N, t, tbl, lead, st, train_base, out_idx, out_dist, keep, bx, (int)blockIdx.y,
                            (int)gridDim.y, cold, uni, M, nchunks, merge, sel);
}

// ---- several independent searches in ONE launch -------------------------------------------------------------------
// A frame-sized search is a cold, half-empty grid: launch + drain latency, not the scan, is what it costs (4096 x 4096:
// 27 us for 17 M pairs).  crossCheck needs two such searches (query -> train and train -> query), candidate
// verification a handful: instead of queueing cold grids behind one another, the searches share one grid.  The
// descriptors travel BY VALUE in the kernel argument segment (no upload in front of the launch); a block finds its
// search by walking the prefix sums of the searches' block counts (wave-uniform, a few scalar compares).
struct bf_search {
    const uint4* q; const uint4* t; const int* tbl; int2* out_idx; int2* out_dist; uint4* keep;
    bf_state st;
    int N, lead, S, qblocks, train_base, first_block, cold, uni, M, sfeed;
};
struct bf_batch {
    int count;
    unsigned epoch;              // polled completion (bf_select.done / .epoch): one word per query block of every search
    unsigned* done;
    bf_search s[SLAM_BF_BATCH_MAX];
};

__global__ __launch_bounds__(256) void bf_top2_batch_kernel(const bf_batch b) {
    int i = 0;
    const int id = (int)blockIdx.x;
    int done_base = 0;                              // the search's first completion word: one per query block, search after search
    while (i + 1 < b.count && id >= b.s[i + 1].first_block) { done_base += b.s[i].qblocks; i++; }
    const bf_search& p = b.s[i];
    const int local = id - p.first_block;           // x fastest, as in the single search: consecutive blocks = consecutive query blocks
    const int bx = local % p.qblocks, by = local / p.qblocks;
    bf_select sel{};
    sel.epoch = b.epoch;
    sel.done = b.done ? b.done + done_base : nullptr;
    if (p.sfeed)   // block-uniform: the search's train rows travel through SGPRs (short or long chunks in device memory)
        bf_top2_block<1, true, false>(p.q, p.N, p.t, p.tbl, p.lead, p.st, p.train_base, p.out_idx, p.out_dist, p.keep, bx, by, p.S, p.cold,
                               p.uni, p.M, 0, 0, sel);
    else
        bf_top2_block<1, false, false>(p.q, p.N, p.t, p.tbl, p.lead, p.st, p.train_base, p.out_idx, p.out_dist, p.keep, bx, by, p.S, p.cold,
                                p.uni, p.M, 0, 0, sel);
}

// merge G decoded tables by (dist, idx)
__global__ __launch_bounds__(256) void bf_merge_tables_kernel(const int2* __restrict__ idx_parts,
                                                              const int2* __restrict__ dist_parts, int G, int N,
                                                              int2* __restrict__ idx, int2* __restrict__ dist) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const uint64_t NONE = ~0ull;
    uint64_t k1 = NONE, k2 = NONE;
    for (int g = 0; g < G; g++) {
        const int2 pi = idx_parts[(size_t)g * N + n], pd = dist_parts[(size_t)g * N + n];
        const uint64_t ka = pi.x < 0 ? NONE : ((uint64_t)(u32)pd.x << 32) | (u32)pi.x;
        const uint64_t kb = pi.y < 0 ? NONE : ((uint64_t)(u32)pd.y << 32) | (u32)pi.y;
        // insert ka then kb, keeping the two smallest
        uint64_t lo = k1 < ka ? k1 : ka, hi = k1 < ka ? ka : k1;
        k1 = lo; k2 = k2 < hi ? k2 : hi;
        lo = k1 < kb ? k1 : kb; hi = k1 < kb ? kb : k1;
        k1 = lo; k2 = k2 < hi ? k2 : hi;
    }
    int2 oi, od;
    oi.x = k1 == NONE ? SLAM_NO_MATCH_IDX : (int)(u32)k1;
    od.x = k1 == NONE ? SLAM_NO_MATCH_DIST : (int)(k1 >> 32);
    oi.y = k2 == NONE ? SLAM_NO_MATCH_IDX : (int)(u32)k2;
    od.y = k2 == NONE ? SLAM_NO_MATCH_DIST : (int)(k2 >> 32);
    idx[n] = oi;
    dist[n] = od;
}

__global__ __launch_bounds__(256) void bf_fill_none_kernel(int N, int2* __restrict__ idx, int2* __restrict__ dist) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    idx[n] = make_int2(SLAM_NO_MATCH_IDX, SLAM_NO_MATCH_IDX);
    dist[n] = make_int2(SLAM_NO_MATCH_DIST, SLAM_NO_MATCH_DIST);
}

// ---- host side -----------------------------------------------------------

struct bf_plan {
    int R;           // queries per lane
    int qblocks;     // grid.x
    int S;           // grid.y = number of chunks (entries of the boundary table - 1)
    int chunk;       // rows of a uniform chunk (a multiple of the LDS tile, or of 32 rows for frame-sized inputs)
    int lead;        // leader chunks (the first `lead` entries of the table)
    int lead_rows;   // rows they cover together
    int tail;        // shrinking chunks at the end of the table
    int sfeed;       // 1: train rows through SGPRs (bf_scan_sgpr.h), 0: through an LDS tile
    int cold;        // rows a chunk that starts without any bound folds in unfiltered (a multiple of 16; 0 = none)
    int uni;         // > 0: every chunk has this many rows (no leaders, no tail) and the kernel needs no boundary table
    int merge;       // queue plans: 1 = the workers exchange by MERGING their pairs into best[] (share_union), 0 = through bound[]
    int workers;     // > 0: a QUEUE plan - grid.y = this many worker blocks per query block, whose waves draw the S chunks
                     // of the table by ticket (bf_top2_block); 0: one block per chunk, grid.y = S
};

static int bf_check_knobs(const int* k) {
    SLAM_REQUIRE(k[0] == 0 || k[0] == 1 || k[0] == 2 || k[0] == 4 || k[0] == 8, "R must be 0, 1, 2, 4 or 8");
    SLAM_REQUIRE(k[1] >= 0 && k[1] <= 4096, "blocks_per_cu out of range");
    SLAM_REQUIRE(k[2] >= -1 && k[2] <= (1 << 22), "lead_rows out of range");
    SLAM_REQUIRE(k[3] >= 0 && k[3] <= (1 << 22) && k[3] % 32 == 0, "lead_chunk must be a multiple of 32 rows");
    SLAM_REQUIRE(k[4] >= -1 && k[4] <= 4096, "tail out of range");
    SLAM_REQUIRE(k[5] >= -1 && k[5] <= 1, "feed must be 0 (heuristic), 1 (SGPRs) or -1 (LDS tile)");
    SLAM_REQUIRE(!(k[5] == 1 && k[0] > 1), "the SGPR-fed scan holds one query per lane (R = 1)");
    SLAM_REQUIRE(k[6] >= -1 && k[6] <= (1 << 20) && (k[6] < 0 || k[6] % 16 == 0), "cold must be -1 or a multiple of 16 rows");
    SLAM_REQUIRE(k[7] >= 0 && k[7] <= (1 << 22) && k[7] % 32 == 0, "chunk must be a multiple of 32 rows");
    SLAM_REQUIRE(k[8] >= -1 && k[8] <= 1, "queue must be 0 (heuristic), 1 (workers draw chunks by ticket) or -1 (one block per chunk)");
    SLAM_REQUIRE(!(k[8] == 1 && (k[0] > 1 || k[5] == -1)), "a queue plan runs the SGPR-fed scan at one query per lane");
    SLAM_REQUIRE(k[9] >= -1 && k[9] <= 1, "merge must be 0 (heuristic), 1 (queue workers exchange by merging) or -1 (through bounds)");
    return SLAM_OK;
}

// Tuning overrides live in the context (0 = heuristic): set through slam_bf_set_tuning, read under ctx->mu.
extern "C" int slam_bf_set_tuning(slam_ctx* ctx, const int32_t* h_knobs, int count) {
    SLAM_REQUIRE(ctx, "slam_bf_set_tuning: null ctx");
    SLAM_REQUIRE(count >= 0 && count <= SLAM_BF_KNOBS && (h_knobs || count == 0), "bad knob array");
    int k[SLAM_BF_KNOBS] = {};
    for (int i = 0; i < count; i++) k[i] = h_knobs[i];
    if (int rc = bf_check_knobs(k)) return rc;
    std::lock_guard<std::mutex> g(ctx->mu);
    for (int i = 0; i < SLAM_BF_KNOBS; i++) ctx->bf_knob[i] = k[i];
    return SLAM_OK;
}

// uniform chunks of the rows [0, M) for a given R: aim at `blocks_per_cu` blocks per CU, a chunk being at least one LDS tile.
// qb_all: the query blocks of ALL searches that share the launch (a batch), or 0 for a search that has the grid to itself.
static void plan_uniform(int num_cu, const int* knob, int64_t N, int64_t M, int R, int blocks_per_cu, bool forced, int64_t qb_all,
                         bf_plan* p) {
    const int64_t tiles = (M + SLAM_TILE_ROWS - 1) / SLAM_TILE_ROWS;
    p->R = R;
    p->qblocks = (int)((N + 256 * R - 1) / (256 * R));
    if (qb_all < p->qblocks) qb_all = p->qblocks;
    // Blocks per CU.  Round 2's LDS-only kernel wanted many short chunks (32-64 blocks per CU: finished waves keep being
    // replaced, so the under-occupied tail is short).  The SGPR-fed scan prefers FEWER, longer chunks - every chunk costs a
    // start without a threshold, a merge and a ticket, and the shrinking tail takes care of the drain: swept again on the
    // final kernel (profiles/r03_blocks_per_cu_sweep.log), 12-18 blocks per CU are best or equal from the 1/8 shard to
    // the headline grid as long as there are no more query blocks than CUs; 16 ships (64k x 64k 1072 -> 1058 us against
    // 32; 8192 x 65536 153.8 -> 151.2 us against 24; 49152 x 49152 620.8 -> 614.6 us; 32768 x 32768, 20000 x 20000,
    // 50000 x 20000 equal; 40000 x 40000 +0.8 %; 14 is worth another 0.1 % at 64k x 64k and costs 2 % at 50000 x 20000).
    // Above that every block already scans tens of thousands of rows and 32 stays (2^20 x 131072: 33.0 ms against
    // 34.7 ms at anything lower).
    if (!blocks_per_cu) blocks_per_cu = p->qblocks > num_cu ? 32 : 16;
    int64_t S = (int64_t)num_cu * blocks_per_cu / qb_all;
    if (S > tiles) S = tiles;
    if (S < 1) S = 1;
    int64_t chunk = (M + S - 1) / S;
    chunk = (chunk + SLAM_TILE_ROWS - 1) / SLAM_TILE_ROWS * SLAM_TILE_ROWS;
    // No chunk longer than 131072 rows: with thousands of query blocks the formula above gives two to four chunks of half a
    // million rows, and a grid of a few very long dispatch rounds ends on a round that is mostly empty (2^20 x 2^20, the
    // loop-closure grid: 524288-row chunks 268.3 ms, 262144 256.1, 131072 253.3, 65536 254.0, 16384 255.9 ms;
    // profiles/r03_blocks_per_cu_sweep.log).
    if (!forced && chunk > 131072) chunk = 131072;
    if (!forced && M < 16384) {
        // Train sets below the leader regime: a small launch costs launch + drain latency on a grid that may not fill the
        // chip.  Since round 3 a chunk that starts before anybody has published a bound folds its first rows in unfiltered
        // (insert_rows: no compare, no branch per row), so a cold start is cheap and what counts is parallelism - a lone
        // wave on a SIMD cannot pair its v_xor with another wave's v_bcnt and exposes every LDS read.  Measured over
        // single searches from 200 x 200 to 12000 x 12000 and 65536 x 4096 (tools/ab_time.py with the chunk knob,
        // profiles/r03_small_chunk_sweep.log): with c1 = the chunk that gives every CU ONE block, the best chunk is c1 up
        // to 128 rows (frame-sized searches, the reference matches <= 200 x 200, slam.py:23: 32-64 rows per block) and
        // about 8 sqrt(c1) beyond - 4096 x 4096: 128 rows (2 blocks per CU, 23.0 -> 17.6 us), 8192 x 8192: 256 (4 per CU,
        // 40.2 -> 33.0 us), 12000 x 12000: 384, 65536 x 4096 and sixteen 4096 x 4096 in one launch: 512 (8 per CU).
        int64_t c1 = (M * qb_all + num_cu - 1) / num_cu;
        int64_t c2 = c1;
        if (c1 > 128) {
            int64_t r = 1;
            while ((r + 1) * (r + 1) <= c1) r++;
            c2 = 8 * r < 128 ? 128 : 8 * r;
        }
        c2 = (c2 + 31) / 32 * 32;
        chunk = c2 < 32 ? 32 : (c2 > 512 ? 512 : c2);
        if (knob[7]) chunk = knob[7];
    }
    p->chunk = (int)chunk;
}

// The chunk boundary table for N x M (tbl[0] = 0 ... tbl[S] = M) and the plan that goes with it.
//   leaders:  `lead` short chunks covering the first lead_rows rows; dispatched first, they publish exact bounds.
//             The 2nd-best distance over s rows is the 2/s quantile of a query's distances and a wave takes the update
//             path for a row when any of its 64 lanes beats its threshold, so a block starting from the bound of s
//             rows fires on about 128/s of its first rows.
//   uniform:  chunks of p.chunk rows.
//   tail:     the last `tail` chunks shrink linearly, so the blocks dispatched last (the youngest waves, which the
//             age-ordered VALU arbiter serves last) have the least left to do when the grid drains.
// rows_on_host: the train rows lie in the context's pinned staging block (frame-sized host calls let the kernel read them
// over PCIe): they must not travel through scalar loads.
static bool bf_rows_on_host(const slam_ctx* ctx, const void* p) {
    const char* c = (const char*)p;
    return ctx->io_host && c >= (const char*)ctx->io_host && c < (const char*)ctx->io_host + ctx->io_host_bytes;
}

// The planner proper: a pure function of the device's CU count, the knobs and the shape (slam_bf_plan_describe exposes it
// without a device, so the CPU test suite can hold it to its invariants).
static bf_plan make_plan_core(int num_cu, const int* k, int64_t N, int64_t M, std::vector<int>* tbl, int64_t qb_all,
                              bool rows_on_host) {
    bf_plan p;
    // R = 1 query per lane measured fastest at every size tried (64k x 64k: 1.68 ms vs 1.79 ms for R = 2,
    // 1.96 ms for R = 4; 58 VGPRs, 8 waves/SIMD); R = 2 / 4 / 8 stay available through slam_bf_set_tuning.
    plan_uniform(num_cu, k, N, M, k[0] ? k[0] : 1, k[1], k[1] != 0, qb_all, &p);
    const int64_t slots = (int64_t)num_cu * 8;                      // resident blocks of 4 waves at 8 waves/SIMD
    // ---- leaders
    int64_t lead_rows = k[2] < 0 ? 0 : k[2];
    if (k[2] == 0 && M >= 16384) {
        // measured (profiles/r02_plan_sweep.log): M/8 up to 8192 rows; 64k x 64k 1595 -> 1532 us with the tail below,
        // the 1/8 query shard 8192 x 65536 252 -> 213 us, 20000 x 20000 181 -> 168 us
        lead_rows = M / 8;
        if (lead_rows > 8192) lead_rows = 8192;
    }
    lead_rows = lead_rows / 32 * 32;
    if (lead_rows + p.chunk > M) lead_rows = 0;                          // leave the rest of the grid real work
    int64_t lead_chunk = k[3];
    if (lead_rows && !lead_chunk) {
        // about one leader wave per SIMD (one block per CU) across all query blocks, chunks of at least 128 rows
        int64_t per_q = (num_cu + p.qblocks - 1) / p.qblocks;
        if (per_q > lead_rows / 128) per_q = lead_rows / 128;
        if (per_q < 1) per_q = 1;
        lead_chunk = (lead_rows / per_q + 31) / 32 * 32;
    }
    p.lead = lead_rows ? (int)((lead_rows + lead_chunk - 1) / lead_chunk) : 0;
    p.lead_rows = (int)lead_rows;
    // ---- tail
    const int64_t rest = M - lead_rows;
    int64_t n_uniform = (rest + p.chunk - 1) / p.chunk;
    int64_t tail = k[4] < 0 ? 0 : (k[4] ? k[4] : 32);                    // shipped: 32 (round 2: 16; with 14 blocks per CU 32 is
                                                                          // worth 2 % on the 1/8 shard and equal elsewhere)
    if (tail > 0) {
        const int64_t last_round = (slots + p.qblocks - 1) / p.qblocks;  // chunk indices in flight when the grid drains
        if (tail > last_round) tail = last_round;
        if (tail > n_uniform / 2) tail = n_uniform / 2;
    }
    // a grid that is resident all at once has no drain to shape: uniform chunks, which also need no boundary table
    // (measured, profiles/r03_small_chunk_sweep.log: up to two blocks per CU the shrinking tail gains nothing - 200 x 200
    // to 3000 x 3000 are 0.4-2.4 us faster without it and a search whose shape changes from frame to frame uploads no
    // table; fuller one-round grids are dispatched over a longer time and keep it: 12000 x 12000 61 -> 57 us)
    const int64_t qb_launch = qb_all > p.qblocks ? qb_all : p.qblocks;
    const bool one_round = lead_rows == 0 && qb_launch * n_uniform <= slots;
    if (one_round && qb_launch * n_uniform <= 2 * (int64_t)num_cu && k[4] == 0) tail = 0;
    std::vector<int>& b = *tbl;
    b.clear();
    b.push_back(0);
    for (int i = 0; i < p.lead; i++) {
        const int64_t e = (int64_t)(i + 1) * lead_chunk;
        b.push_back((int)(e < lead_rows ? e : lead_rows));
    }
    // sizes of the shrinking chunks: chunk * (tail - i) / (tail + 1), multiples of 32, at least 32 rows
    std::vector<int> ts;
    int64_t tail_rows = 0;
    for (int64_t i = 0; i < tail; i++) {
        int64_t sz = (int64_t)p.chunk * (tail - i) / (tail + 1) / 32 * 32;
        if (sz < 32) sz = 32;
        ts.push_back((int)sz);
        tail_rows += sz;
    }
    if (tail_rows >= rest) { ts.clear(); tail_rows = 0; }
    int64_t pos = lead_rows;
    const int64_t uniform_end = M - tail_rows;
    while (pos < uniform_end) {
        pos = pos + p.chunk < uniform_end ? pos + p.chunk : uniform_end;
        b.push_back((int)pos);
    }
    for (int sz : ts) {
        pos += sz;
        b.push_back((int)pos);
    }
    b.back() = (int)M;
    p.tail = (int)ts.size();
    p.S = (int)b.size() - 1;
    // Which way the train rows travel.  Through SGPRs the scan needs no LDS read (the chip holds a higher clock) but every
    // call of the scan exposes a scalar-load latency: measured (profiles/r03_ab_sgpr_feed.log) it wins where a block scans
    // many rows - 64k x 64k (chunks of 2048) 1217 -> 1080-1091 us, 2^20 x 2^20 304 -> 272 ms, 32768 x 65536 623 -> 599 us -
    // and lost where chunks were one tile and the filter still ran from the first row (20000 x 20000, 256 rows: 145 -> 158 us).
    // With the unfiltered start a chunk of up to 128 rows is ONE call of the scan with no exchange in between, and there
    // the SGPR form wins again (no tile staging, no barrier, two waves per SIMD do not expose LDS latency:
    // profiles/r03_ab_cold_start.log feed=1, 4096 x 4096 18.4 -> 17.3 us, 2000 x 2000 10.7 -> 9.8 us; 8192 x 8192 with
    // 256-row chunks 32.8 -> 35.6 us stays on the LDS form).  Rows that lie in pinned host memory (frame-sized host calls)
    // always take the LDS form: every scalar load would be a PCIe round trip.
    // The unfiltered start: 128 rows, and the WHOLE chunk for train sets below the leader regime whose chunks have at most
    // 384 rows (6000 x 6000 to 12000 x 12000: all the chunks of a query start together, so by the time a filter could
    // reject anything the chunk is over - 6000 x 6000 25.3 -> 20.6 us, 8192 x 8192 32.6 -> 28.9 us, 10000 x 10000 47.7 ->
    // 40.6 us with the chunk unfiltered throughout; 12000 x 12000 and 65536 x 4096 with their 384- / 512-row chunks equal
    // or better filtered; profiles/r03_small_chunk_sweep.log).
    p.cold = k[6] < 0 ? 0 : (k[6] ? k[6] : SLAM_COLD_ROWS);
    if (k[6] == 0 && lead_rows == 0 && M < 16384 && p.chunk <= 384 && p.cold < p.chunk) p.cold = (p.chunk + 15) / 16 * 16;
    const bool short_chunks = p.chunk <= p.cold && lead_rows == 0 && !rows_on_host;
    // Since the scan's fired groups go by triples (round 4, second half) the SGPR form is at least as fast wherever the rows lie in
    // device memory - also on the one-tile chunks of the leader regime that stayed on the LDS tile until then (2048 x 40000 45.7 ->
    // 34.3 us, 200 x 20000 21.2 -> 17.9, 64 x 65536 27.0 -> 24.1, 8192 x 16384 50.5 -> 48.4; profiles/r04_ab_queue.log "feed"):
    // the LDS tile is left for rows in pinned host memory and for several queries per lane.
    p.sfeed = p.R == 1 && (k[5] == 1 || (k[5] == 0 && (p.chunk >= 2 * SLAM_TILE_ROWS || short_chunks || !rows_on_host))) ? 1 : 0;
    p.uni = p.tail == 0 && p.lead == 0 ? p.chunk : 0;
    // one round and no chunk longer than the unfiltered start: the kernel exchanges no bounds at all (passed as -cold)
    if (qb_launch * (int64_t)p.S <= slots && p.chunk <= p.cold && p.lead == 0) p.cold = -p.cold;   // (S: the tail's extra chunks counted)
    p.workers = 0;
    p.merge = 0;
    // ---- queue plan (round 4): as many worker blocks per query block as are resident at once; their waves draw the chunks
    // of the table by ticket and keep their top-2 from chunk to chunk (bf_top2_block).  What it removes is what made the
    // 1/8 query shard of an 8-GPU run 27 % less efficient than the full grid: 137 starts without a
    // threshold, merges and arrivals per query where 48 do, the gaps between a block's end and its successor's first row,
    // and a drain of 40 us in which the last dispatch round thinned out (profiles/r03_block_timeline.log).
    // The table: uniform chunks of c rows (256 for a shard, up to 1024 when a worker has thousands of rows to itself), and
    // from the point where the rest would give every worker fewer than two of those, chunks of (rest / 2 workers) rows,
    // never fewer than 32: the waves of a query run out of work within a few dozen rows of each other.
    const int64_t resident = (int64_t)num_cu * SLAM_BF_RESIDENT;
    const int64_t W = resident / p.qblocks > 1 ? resident / p.qblocks : 1;   // (forced on a grid of many dispatch rounds: one)
    const bool can_queue = p.R == 1 && k[5] != -1 && !rows_on_host && qb_all <= p.qblocks && (M >= 16384 || k[8] == 1) && W >= 1;
    // (fill: workers x query blocks must come to >= 96 % of the resident slots - the slots a queue plan leaves empty stay empty for
    // the whole launch, where the one-block-per-chunk plans refill them: 120000 x 65536 at 91.6 % fill ran 5 % slower as a queue,
    // 50000 x 20000 at 89 % 1.4 % slower, 20000 x 20000 at 97.7 % 7 % faster; profiles/r04_ab_queue.log "shapes")
    // (and not two or three workers a query block over a train set far beyond the L2s: 131072 x 2^20 - the per-rank problem of the
    // 8-GPU loop-closure run, three workers a query block - 32.25 ms as a queue against 31.80 ms one block per chunk, whose
    // blocks of a dispatch round stream the same 4 MiB chunk together)
    // (and at most 256 workers a query block: with 384 - 1000 queries - the one-block-per-chunk plan wins again, 129 against 138 us
    // at 1000 x 400000; profiles/r04_ab_queue.log "many workers")
    const bool want_queue = k[8] == 1 || (k[8] == 0 && W >= 2 && W <= 256 && W * p.qblocks * 100 >= resident * 96 && M >= 1024 * W &&
                                          (W >= 4 || M <= 131072));
    if (can_queue && want_queue) {
        // 256-row chunks; 1024 where a worker has 16384 rows or more to itself (fewer tickets and exchanges: 131072 x 65536 2078 ->
        // 2063 us, 20000 x 400000 1977 -> 1965, 16384 x 524288 2059 -> 2049; at 64k x 64k, 10923 rows a worker, 256 is still the
        // better one: 1020.8 against 1027.3; profiles/r04_ab_queue.log "chunk length")
        int64_t c = k[7] ? k[7] : (M / W >= 16384 ? 1024 : 256);
        const int64_t c_floor = (M / 3000 + 255) / 256 * 256;               // keeps the table within one ring slot
        if (c < c_floor) c = c_floor;
        const bool guided = k[4] >= 0;
        const int64_t cmin = k[4] > 0 ? (k[4] + 15) / 16 * 16 : 64;
        b.clear();
        b.push_back(0);
        int shrinking = 0;
        for (int64_t at = 0; at < M;) {
            int64_t len = c;
            if (guided) {
                const int64_t g = (M - at) / (2 * W) / 16 * 16;
                if (g < len) { len = g < cmin ? cmin : g; shrinking++; }
            }
            at = at + len < M ? at + len : M;
            b.push_back((int)at);
        }
        p.chunk = (int)c;
        p.S = (int)b.size() - 1;
        p.lead = 0; p.lead_rows = 0; p.tail = shrinking; p.uni = 0; p.sfeed = 1;
        p.cold = k[6] < 0 ? 0 : (k[6] ? k[6] : SLAM_COLD_ROWS);
        p.workers = (int)(W < p.S ? W : p.S);
        // the exact bound of a merge pays where many workers share a query (48 on the 1/8 shard: 148.5 -> 145.3 us; 24: 273 ->
        // 266; six at 64k x 64k: 1015 -> 1022, where the minimum of six 2nd-best distances is nearly as good and costs less) -
        // but not where VERY many do: from 128 workers up their atomics on one query's slot lines cost more than the tighter
        // thresholds give (3000 x 200000, 128 workers: 182 us merging against 176 through bounds; 2048 x 262144, 192: 171 / 160)
        p.merge = k[9] == 1 || (k[9] == 0 && p.workers >= 12 && p.workers <= 96) ? 1 : 0;
    }
    return p;
}

static bf_plan make_plan(slam_ctx* ctx, int64_t N, int64_t M, std::vector<int>* tbl, int64_t qb_all = 0,
                         bool rows_on_host = false) {
    std::lock_guard<std::mutex> g(ctx->mu);
    return make_plan_core(ctx->num_cu, ctx->bf_knob, N, M, tbl, qb_all, rows_on_host);
}

// The launch plan for N x M on a device with `num_cu` CUs under `count` knobs (as slam_bf_set_tuning; NULL / 0 = shipped),
// WITHOUT a device: h_plan int32 [14] = slam_bf_plan_info's ten entries (h_plan[7] = num_cu) + {table-free (1 = the kernel
// computes its chunk from the block index), bound-free (1 = no block reads or writes a bound), worker blocks per query
// block of a queue plan (0 = one block per chunk), resident blocks per CU the planner counts on}; the chunk boundary table
// goes to h_tbl (up to tbl_cap entries; may be NULL) and its length to *tbl_len.  rows_on_host: the train rows lie in
// pinned host memory (frame-sized host calls).  qb_all: query blocks of all searches sharing the launch (0 = alone).
extern "C" int slam_bf_plan_describe(int num_cu, const int32_t* h_knobs, int count, int64_t N, int64_t M, int64_t qb_all,
                                     int rows_on_host, int32_t* h_plan, int32_t* h_tbl, int64_t tbl_cap, int64_t* tbl_len) {
    SLAM_REQUIRE(num_cu >= 1 && num_cu <= 4096, "num_cu out of range");
    SLAM_REQUIRE(count >= 0 && count <= SLAM_BF_KNOBS && (h_knobs || count == 0), "bad knob array");
    SLAM_REQUIRE(N >= 1 && M >= 1 && M <= SLAM_MAX_TRAIN_PER_PASS && N <= (1ll << 30) && qb_all >= 0, "bad sizes");
    SLAM_REQUIRE(h_plan && tbl_len && (h_tbl || tbl_cap == 0) && tbl_cap >= 0, "slam_bf_plan_describe: null argument");
    int k[SLAM_BF_KNOBS] = {};
    for (int i = 0; i < count; i++) k[i] = h_knobs[i];
    if (int rc = bf_check_knobs(k)) return rc;
    std::vector<int> tbl;
    const bf_plan p = make_plan_core(num_cu, k, N, M, &tbl, qb_all, rows_on_host != 0);
    h_plan[0] = p.R; h_plan[1] = p.qblocks; h_plan[2] = p.chunk; h_plan[3] = p.S;
    h_plan[4] = p.lead_rows; h_plan[5] = p.lead; h_plan[6] = p.tail; h_plan[7] = num_cu;
    h_plan[8] = p.sfeed; h_plan[9] = p.cold < 0 ? -p.cold : p.cold; h_plan[10] = p.uni ? 1 : 0; h_plan[11] = p.cold < 0 ? 1 : 0;
    h_plan[12] = p.workers; h_plan[13] = SLAM_BF_RESIDENT | (p.merge << 8);
    *tbl_len = (int64_t)tbl.size();
    for (int64_t i = 0; i < (int64_t)tbl.size() && i < tbl_cap; i++) h_tbl[i] = tbl[(size_t)i];
    return SLAM_OK;
}

extern "C" int slam_bf_plan_info(slam_ctx* ctx, int64_t N, int64_t M, int32_t* h_plan /*[10]*/) {
    SLAM_REQUIRE(ctx && h_plan, "slam_bf_plan_info: null argument");
    SLAM_REQUIRE(N >= 1 && M >= 1 && M <= SLAM_MAX_TRAIN_PER_PASS && N <= (1ll << 30), "bad sizes");
    std::vector<int> tbl;
    const bf_plan p = make_plan(ctx, N, M, &tbl);
    h_plan[0] = p.R; h_plan[1] = p.qblocks; h_plan[2] = p.chunk; h_plan[3] = p.S;
    h_plan[4] = p.lead_rows; h_plan[5] = p.lead; h_plan[6] = p.tail; h_plan[7] = ctx->num_cu;
    h_plan[8] = p.sfeed; h_plan[9] = p.cold < 0 ? -p.cold : p.cold;
    return SLAM_OK;
}

// merge state for up to N queries, kept clean between launches (see bf_state), followed by the chunk boundary table
#define SLAM_BF_TBL_MAX 65536   // entries; grid.y <= 65535
// (SLAM_BF_TBL_RING slots of SLAM_BF_TBL_SLOT entries + one slot of SLAM_BF_TBL_MAX: internal.h)
static size_t bf_state_bytes(size_t rows) {      // best u64 [rows] + bound u32 [rows] + arrivals u32 [blocks] + padded cursors + alignment slack
    const size_t blocks = (rows + 255) / 256;   // query blocks at R = 1, the finest split
    return rows * 12 + blocks * 4 + blocks * 16 * SLAM_CURSOR_STRIDE + 128;
}

static int bf_state_fill(slam_ctx* ctx) {
    char* p = (char*)ctx->bf_state_mem;
    const size_t rows = (size_t)ctx->bf_state_rows;
    SLAM_HIP(hipMemsetAsync(p, 0xFF, rows * 8, ctx->stream));
    SLAM_HIP(hipMemsetAsync(p + rows * 8, 0x7F, rows * 4, ctx->stream));
    SLAM_HIP(hipMemsetAsync(p + rows * 12, 0, bf_state_bytes(rows) - rows * 12, ctx->stream));   // arrivals, then the cursors
    return SLAM_OK;
}

static int bf_state_get(slam_ctx* ctx, int64_t N, bf_state* out) {
    std::lock_guard<std::mutex> g(ctx->mu);
    if (N > ctx->bf_state_rows) {
        SLAM_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->bf_state_mem) SLAM_HIP(hipFree(ctx->bf_state_mem));
        ctx->bf_state_mem = nullptr;
        ctx->bf_state_rows = 0;
        const int64_t rows = N + (N >> 2) + 1024;                 // headroom: slightly larger calls do not realloc
        SLAM_HIP(hipMalloc(&ctx->bf_state_mem, bf_state_bytes((size_t)rows)));
        ctx->bf_state_rows = rows;
        if (int rc = bf_state_fill(ctx)) return rc;
    }
    char* p = (char*)ctx->bf_state_mem;
    const size_t rows = (size_t)ctx->bf_state_rows;
    out->best = (unsigned long long*)p;
    out->bound = (u32*)(p + rows * 8);
    out->arrivals = (u32*)(p + rows * 12);
    out->cursor = (u32*)(((uintptr_t)(out->arrivals + (rows + 255) / 256) + 127) & ~(uintptr_t)127);
    return SLAM_OK;
}

// The boundary table on the device.  A search whose table equals the previous one reuses it (back-to-back searches of
// one shape upload nothing).  Otherwise the table goes into the next slot of a small ring of (pinned host, device) slot
// pairs with one asynchronous copy and NO synchronisation: frame-sized searches change shape on every frame
// (frontend.py:181-187 matches whatever ORB found), and crossCheck alternates N x M with M x N.  The device side of a
// slot is safe by stream order (the copy into it queues behind every launch that read it); the pinned host side is
// rewritten by the CPU, so an event recorded behind each upload is checked before the slot's next turn,
// SLAM_BF_TBL_RING changes of shape later (it has long fired).
// Tables with more than SLAM_BF_TBL_SLOT entries (train sets cut into > 4095 chunks) take the one big slot, which
// is rewritten behind a stream synchronisation.
static int bf_table_get(slam_ctx* ctx, const std::vector<int>& tbl, const int** d_tbl) {
    std::lock_guard<std::mutex> g(ctx->mu);
    const size_t n = tbl.size();
    SLAM_REQUIRE(n <= SLAM_BF_TBL_MAX, "train set needs %zu chunks, more than one launch can index", n - 1);
    if (!ctx->bf_tbl_ready) {
        // all or nothing: the table storage counts as initialised only when the device block, the pinned block and every
        // event exist; a partial failure frees what it got, so the next call starts over instead of meeting a null pointer
        const size_t ints = (size_t)SLAM_BF_TBL_RING * SLAM_BF_TBL_SLOT + SLAM_BF_TBL_MAX;
        hipError_t e = hipMalloc(&ctx->bf_tbl_dev, ints * sizeof(int));
        if (e == hipSuccess) e = hipHostMalloc(&ctx->bf_tbl_host, ints * sizeof(int), hipHostMallocDefault);
        int made = 0;
        for (; e == hipSuccess && made < SLAM_BF_TBL_RING; made++)
            e = hipEventCreateWithFlags(&ctx->bf_tbl_ev[made], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int i = 0; i < made - 1; i++) (void)hipEventDestroy(ctx->bf_tbl_ev[i]);
            if (ctx->bf_tbl_host) (void)hipHostFree(ctx->bf_tbl_host);
            if (ctx->bf_tbl_dev) (void)hipFree(ctx->bf_tbl_dev);
            ctx->bf_tbl_host = ctx->bf_tbl_dev = nullptr;
            return slam_set_error(SLAM_ERR_HIP, "chunk table storage: %s", hipGetErrorString(e));
        }
        for (int i = 0; i <= SLAM_BF_TBL_RING; i++) {
            ctx->bf_tbl_n[i] = 0;
            ctx->bf_tbl_busy[i] = false;
        }
        ctx->bf_tbl_cur = 0;
        ctx->bf_tbl_ready = true;
    }
    int* host = (int*)ctx->bf_tbl_host;
    int* dev = (int*)ctx->bf_tbl_dev;
    const bool big = n > SLAM_BF_TBL_SLOT;
    int slot = big ? SLAM_BF_TBL_RING : ctx->bf_tbl_cur;
    if (!big)   // any slot that still holds this table will do (shapes that alternate, e.g. crossCheck's two searches)
        for (int i = 0; i < SLAM_BF_TBL_RING; i++)
            if (ctx->bf_tbl_n[i] == (int)n && memcmp(host + (size_t)i * SLAM_BF_TBL_SLOT, tbl.data(), n * sizeof(int)) == 0) {
                slot = i;
                break;
            }
    size_t off = big ? (size_t)SLAM_BF_TBL_RING * SLAM_BF_TBL_SLOT : (size_t)slot * SLAM_BF_TBL_SLOT;
    if (ctx->bf_tbl_n[slot] != (int)n || memcmp(host + off, tbl.data(), n * sizeof(int)) != 0) {
        if (big) {
            SLAM_HIP(hipStreamSynchronize(ctx->stream));      // a queued launch may still read the big slot
        } else {
            slot = (ctx->bf_tbl_cur + 1) % SLAM_BF_TBL_RING;
            off = (size_t)slot * SLAM_BF_TBL_SLOT;
            if (ctx->bf_tbl_busy[slot]) {                      // its last upload: SLAM_BF_TBL_RING changes of shape ago
                SLAM_HIP(hipEventSynchronize(ctx->bf_tbl_ev[slot]));
                ctx->bf_tbl_busy[slot] = false;
            }
            ctx->bf_tbl_cur = slot;
        }
        // the slot is recorded as holding this table only once the copy (and its event) are queued: after a failure it
        // holds nothing, so a later search of the same shape uploads again instead of launching on stale bounds
        ctx->bf_tbl_n[slot] = 0;
        memcpy(host + off, tbl.data(), n * sizeof(int));
        SLAM_HIP(hipMemcpyAsync(dev + off, host + off, n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        if (!big) {
            SLAM_HIP(hipEventRecord(ctx->bf_tbl_ev[slot], ctx->stream));
            ctx->bf_tbl_busy[slot] = true;
        }
        ctx->bf_tbl_n[slot] = (int)n;
    }
    *d_tbl = dev + off;
    return SLAM_OK;
}

// Put the merge state back to its idle values (best = none, bound = loose, arrivals = 0).  The kernel restores
// it itself at the end of every search; this is for the case where one did not finish (a failed launch, a device
// error reported by a later call): stream-ordered, so it may be issued right behind whatever is still queued.
extern "C" int slam_bf_reset_state(slam_ctx* ctx) {
    SLAM_REQUIRE(ctx, "slam_bf_reset_state: null ctx");
    SLAM_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->bf_state_mem) return SLAM_OK;
    return bf_state_fill(ctx);
}

// How many 32-bit words of the merge state are NOT at their idle value (best = ~0, bound = 0x7F7F7F7F, arrivals = 0,
// cursor = 0) once everything queued on the context has finished: 0 after any completed search.  For tests and soak tools:
// a key, bound, ticket or cursor that survived a search would corrupt the next one silently.
extern "C" int slam_bf_state_dirty(slam_ctx* ctx, int64_t* h_words) {
    SLAM_REQUIRE(ctx && h_words, "slam_bf_state_dirty: null argument");
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<u32> host;
    size_t rows;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        rows = (size_t)ctx->bf_state_rows;
        *h_words = 0;
        if (!ctx->bf_state_mem) return SLAM_OK;
        host.resize(bf_state_bytes(rows) / 4);
        SLAM_HIP(hipMemcpy(host.data(), ctx->bf_state_mem, host.size() * sizeof(u32), hipMemcpyDeviceToHost));
    }
    int64_t bad = 0;
    for (size_t i = 0; i < host.size(); i++) {
        const u32 idle = i < rows * 2 ? 0xFFFFFFFFu : (i < rows * 3 ? SLAM_BOUND_IDLE : 0u);
        bad += host[i] != idle;
    }
    *h_words = bad;
    return SLAM_OK;
}

// Waiting for a search by polling: the last arriver of every query block stores the call's epoch into a pinned word once its
// results are visible to the host (bf_top2_block, sel.done).  Frame-sized calls return ~4 us earlier than through
// hipStreamSynchronize (11.1 -> 6.7 us for an empty kernel, tools/ubench/sync_vs_poll.hip).  A search that is not done
// after 2 ms of spinning - a long one, or a launch that failed - is waited for the ordinary way, which also surfaces errors;
// and every 256th polled call synchronises anyway, so the runtime retires its completion signals at a bounded distance.
// The context's pinned block: SLAM_BF_DONE_FLAGS completion words, then `extra` bytes (the fused selection's per-wave counts).
int slam_done_block(slam_ctx* ctx, uint64_t extra) {
    std::lock_guard<std::mutex> g(ctx->mu);
    if (SLAM_BF_DONE_BYTES + extra > ctx->sel_host_bytes) {
        SLAM_HIP(hipStreamSynchronize(ctx->stream));          // a queued search may still write the old block
        if (ctx->sel_host) SLAM_HIP(hipHostFree(ctx->sel_host));
        ctx->sel_host = nullptr;
        ctx->sel_host_bytes = 0;
        const uint64_t bytes = (SLAM_BF_DONE_BYTES + extra + 4095) / 4096 * 4096 * 2;
        SLAM_HIP(hipHostMalloc(&ctx->sel_host, bytes, hipHostMallocDefault));
        memset(ctx->sel_host, 0, bytes);
        ctx->sel_host_bytes = bytes;
    }
    return SLAM_OK;
}
unsigned slam_done_epoch(slam_ctx* ctx) {
    unsigned e = ++ctx->done_epoch;
    if (e == 0) e = ++ctx->done_epoch;                        // (0 is what a fresh block holds)
    return e;
}
int slam_wait_done(slam_ctx* ctx, const unsigned* flags, int count, unsigned epoch) {
    const auto t0 = std::chrono::steady_clock::now();
    bool done = false;
    for (unsigned spins = 1; !done; spins++) {
        int i = 0;
        // (== and not "this epoch or a later one": two threads on one context may draw their epochs and launch in different
        // orders; a word overwritten by the other thread's call is then simply never seen, and the wait falls back to the stream)
        while (i < count && __atomic_load_n(&flags[i], __ATOMIC_ACQUIRE) == epoch) i++;
        done = i == count;
        if (done) break;
        if ((spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        __builtin_ia32_pause();
    }
    if (!done || (++ctx->polled_calls & 255u) == 0) SLAM_HIP(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

// one pass over at most 2^23 train rows
static int bf_pass(slam_ctx* ctx, const void* d_query, int64_t N, const void* d_train, int64_t M,
                   int64_t train_base, int32_t* d_idx, int32_t* d_dist, void* d_keep, const bf_select sel = bf_select{},
                   int* qblocks_out = nullptr) {
    std::vector<int> tbl;
    const bf_plan p = make_plan(ctx, N, M, &tbl, 0, bf_rows_on_host(ctx, d_train));
    bf_state st;
    if (int rc = bf_state_get(ctx, N, &st)) return rc;
    const int* d_tbl = nullptr;
    if (!p.uni)
        if (int rc = bf_table_get(ctx, tbl, &d_tbl)) return rc;
    int workers = p.workers;
    if (workers) {
        // the plan counts on SLAM_BF_RESIDENT blocks per CU; a build or device that holds fewer gets fewer workers (the
        // table is the same: the waves simply draw more tickets each), never a second dispatch round of idle workers
        static std::atomic<int> occ_once{0};     // a property of the kernel and the architecture: asked once per process
        int occ = occ_once.load(std::memory_order_relaxed);
        if (!occ) {
            int o = 0;
            SLAM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, bf_top2_kernel<1, true, true>, 256, 0));
            occ = o > 0 ? o : SLAM_BF_RESIDENT;
            occ_once.store(occ, std::memory_order_relaxed);
        }
        const int64_t fit = (int64_t)ctx->num_cu * occ / p.qblocks;
        if (fit >= 1 && fit < workers) workers = (int)fit;
    }
    const dim3 grid(p.qblocks, workers ? workers : p.S), block(256);
    if (qblocks_out) *qblocks_out = p.qblocks;
    const int nchunks = workers ? p.S : 0;
    const uint4* q = (const uint4*)d_query;
    const uint4* t = (const uint4*)d_train;
    int2* oi = (int2*)d_idx;
    int2* od = (int2*)d_dist;
    const int tb = (int)train_base;
    SLAM_HIP(hipGetLastError());
    if (int rc = slam_prof_begin(ctx)) return rc;
#define SLAM_BF_LAUNCH(R_, F_, Q_) \
    bf_top2_kernel<R_, F_, Q_><<<grid, block, 0, ctx->stream>>>(q, (int)N, t, d_tbl, p.lead, st, tb, oi, od, (uint4*)d_keep, p.cold, p.uni, (int)M, nchunks, p.merge, sel)
    switch (p.R) {
        case 8: SLAM_BF_LAUNCH(8, false, false); break;
        case 4: SLAM_BF_LAUNCH(4, false, false); break;
        case 2: SLAM_BF_LAUNCH(2, false, false); break;
        default:
            if (workers) SLAM_BF_LAUNCH(1, true, true);
            else if (p.sfeed) SLAM_BF_LAUNCH(1, true, false);
            else SLAM_BF_LAUNCH(1, false, false);
            break;
    }
#undef SLAM_BF_LAUNCH
    if (int rc = slam_prof_end(ctx)) return rc;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        (void)slam_bf_reset_state(ctx);
        return slam_set_error(SLAM_ERR_HIP, "top-2 kernel launch failed: %s", hipGetErrorString(e));
    }
    return SLAM_OK;
}

extern "C" int slam_bf_knn2_u256(slam_ctx* ctx, const void* d_query, int64_t N, const void* d_train,
                                 int64_t M, int64_t train_base, int32_t* d_idx, int32_t* d_dist) {
    return slam_bf_knn2_keep(ctx, d_query, N, d_train, M, train_base, d_idx, d_dist, nullptr);
}

// slam_bf_knn2_u256 that also leaves a copy of the query rows at d_keep (device memory, 32*N bytes) when M > 0
int slam_bf_knn2_keep(slam_ctx* ctx, const void* d_query, int64_t N, const void* d_train, int64_t M,
                      int64_t train_base, int32_t* d_idx, int32_t* d_dist, void* d_keep) {
    SLAM_REQUIRE(ctx, "slam_bf_knn2_u256: null ctx");
    SLAM_REQUIRE(N >= 0 && M >= 0, "negative size (N=%lld, M=%lld)", (long long)N, (long long)M);
    SLAM_REQUIRE(N <= (1ll << 30), "N=%lld exceeds 2^30 query rows per call", (long long)N);
    SLAM_REQUIRE(train_base >= 0 && train_base + M <= 0x7FFFFFFFll, "train_base + M must fit int32");
    if (N == 0) return SLAM_OK;
    SLAM_REQUIRE(d_query && d_idx && d_dist, "slam_bf_knn2_u256: null device pointer");
    SLAM_REQUIRE(((uintptr_t)d_query & 15) == 0 && ((uintptr_t)d_train & 15) == 0,
                 "descriptor pointers must be 16-byte aligned");
    SLAM_HIP(hipSetDevice(ctx->device));
    if (M == 0) {
        bf_fill_none_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream>>>(
            (int)N, (int2*)d_idx, (int2*)d_dist);
        SLAM_HIP(hipGetLastError());
        return SLAM_OK;
    }
    SLAM_REQUIRE(d_train, "slam_bf_knn2_u256: null train pointer");
    const int64_t PASS = SLAM_MAX_TRAIN_PER_PASS;
    const int64_t passes = (M + PASS - 1) / PASS;
    SLAM_REQUIRE(((uintptr_t)d_keep & 15) == 0, "d_keep must be 16-byte aligned");
    if (passes == 1) return bf_pass(ctx, d_query, N, d_train, M, train_base, d_idx, d_dist, d_keep);
    // train set larger than one key range: run passes into per-pass tables, then merge them
    const uint64_t tb = (uint64_t)passes * N * 2 * sizeof(int32_t);
    void* ws = nullptr;
    if (int rc = slam_workspace(ctx, 2 * tb, &ws)) return rc;
    int32_t* idx_parts = (int32_t*)ws;
    int32_t* dist_parts = (int32_t*)((char*)ws + tb);
    for (int64_t p = 0; p < passes; p++) {
        const int64_t m0 = p * PASS, m = (M - m0) < PASS ? (M - m0) : PASS;
        if (int rc = bf_pass(ctx, d_query, N, (const char*)d_train + m0 * SLAM_DESC_BYTES, m, train_base + m0,
                             idx_parts + p * N * 2, dist_parts + p * N * 2, p == 0 ? d_keep : nullptr))
            return rc;
    }
    return slam_bf_merge_top2(ctx, idx_parts, dist_parts, passes, N, d_idx, d_dist);
}

// slam_bf_knn2_u256 + a selection that needs no reduction over the queries, in ONE launch (see bf_select).  d_sel_keep u8 [N];
// the rows kept by each wave land in pinned host memory owned by the context (written by the kernel itself: no copy command),
// and *h_count is their sum once the search is done (polled completion words up to SLAM_BF_DONE_FLAGS query blocks, else one
// synchronisation).  h_count may be null: then nothing is waited for (asynchronous, the
// flags are in d_sel_keep when the stream gets there).  Train sets that need several passes and empty ones take the two-step
// way (search, then slam_filter_launch).
int slam_bf_knn2_select(slam_ctx* ctx, const void* d_query, int64_t N, const void* d_train, int64_t M, int64_t train_base,
                        int32_t* d_idx, int32_t* d_dist, void* d_query_keep, int mode, double param, uint8_t* d_sel_keep,
                        int64_t* h_count) {
    SLAM_REQUIRE(ctx, "slam_bf_knn2_select_u256: null ctx");
    SLAM_REQUIRE(mode == 0 || mode == 2, "slam_bf_knn2_select_u256: mode %d (0 = has a neighbour, 2 = Lowe ratio; the min-distance filter "
                 "needs a reduction over the queries: slam_bf_match_filter)", mode);
    SLAM_REQUIRE(N >= 0 && N <= (1ll << 30), "bad N=%lld", (long long)N);
    if (h_count) *h_count = 0;
    if (N == 0) return SLAM_OK;
    SLAM_REQUIRE(d_sel_keep, "slam_bf_knn2_select_u256: null d_keep");
    const bool fused = M > 0 && M <= SLAM_MAX_TRAIN_PER_PASS;
    if (!fused) {
        if (int rc = slam_bf_knn2_keep(ctx, d_query, N, d_train, M, train_base, d_idx, d_dist, d_query_keep)) return rc;
        if (!h_count) return slam_filter_launch(ctx, d_idx, d_dist, N, mode, param, d_sel_keep);
        return slam_bf_match_filter(ctx, d_idx, d_dist, N, mode, param, d_sel_keep, h_count, nullptr);
    }
    SLAM_REQUIRE(M >= 0 && d_query && d_train && d_idx && d_dist, "slam_bf_knn2_select_u256: null device pointer");
    SLAM_REQUIRE(((uintptr_t)d_query & 15) == 0 && ((uintptr_t)d_train & 15) == 0 && ((uintptr_t)d_query_keep & 15) == 0,
                 "descriptor pointers must be 16-byte aligned");
    SLAM_REQUIRE(train_base >= 0 && train_base + M <= 0x7FFFFFFFll, "train_base + M must fit int32");
    SLAM_HIP(hipSetDevice(ctx->device));
    const int64_t waves = (N + 63) / 64;                            // groups of 64 consecutive queries
    if (int rc = slam_done_block(ctx, (uint64_t)waves * 4)) return rc;
    // the pinned block: SLAM_BF_DONE_FLAGS completion words (one per query block of a search that is waited for by polling),
    // then the per-wave counts
    bf_select sel;
    sel.keep = d_sel_keep; sel.wave_kept = (int*)((char*)ctx->sel_host + SLAM_BF_DONE_BYTES); sel.param = param; sel.mode = mode;
    sel.epoch = 0; sel.done = nullptr;
    const bool poll = h_count && (N + 255) / 256 <= SLAM_BF_DONE_FLAGS;
    if (poll) {
        sel.done = (unsigned*)ctx->sel_host;
        sel.epoch = slam_done_epoch(ctx);
    }
    int qblocks = 0;
    if (int rc = bf_pass(ctx, d_query, N, d_train, M, train_base, d_idx, d_dist, d_query_keep, sel, &qblocks)) return rc;
    if (!h_count) return SLAM_OK;
    if (poll) {
        if (int rc = slam_wait_done(ctx, sel.done, qblocks, sel.epoch)) return rc;
    } else {
        SLAM_HIP(hipStreamSynchronize(ctx->stream));
    }
    int64_t c = 0;
    const volatile int* wk = (const volatile int*)sel.wave_kept;
    for (int64_t w = 0; w < waves; w++) c += wk[w];
    *h_count = c;
    return SLAM_OK;
}

extern "C" int slam_bf_knn2_select_u256(slam_ctx* ctx, const void* d_query, int64_t N, const void* d_train, int64_t M,
                                        int64_t train_base, int32_t* d_idx, int32_t* d_dist, int mode, double param,
                                        uint8_t* d_keep, int64_t* h_count) {
    return slam_bf_knn2_select(ctx, d_query, N, d_train, M, train_base, d_idx, d_dist, nullptr, mode, param, d_keep, h_count);
}

// B independent searches in one launch (see bf_top2_batch_kernel).  h_keep: per search, where to leave a device copy of
// its query rows (or null); may itself be null.
int slam_bf_knn2_batch_keep(slam_ctx* ctx, int64_t B, const slam_bf_search* h_searches, void* const* h_keep, bool wait) {
    SLAM_REQUIRE(ctx, "slam_bf_knn2_batch_u256: null ctx");
    SLAM_REQUIRE(B >= 0 && B <= SLAM_BF_BATCH_MAX, "B=%lld searches, at most %d per call", (long long)B, SLAM_BF_BATCH_MAX);
    if (B == 0) return SLAM_OK;
    SLAM_REQUIRE(h_searches, "slam_bf_knn2_batch_u256: null search table");
    // wait: the call returns when every search is done and its results are visible to the host - by polling the searches'
    // completion words where that is possible (at most SLAM_BF_DONE_FLAGS query blocks, no search without train rows),
    // otherwise by synchronising the stream
    bool poll = wait;
    int64_t rows = 0, qb_all = 0;
    for (int64_t i = 0; i < B; i++) {
        const slam_bf_search& h = h_searches[i];
        SLAM_REQUIRE(h.N >= 0 && h.M >= 0 && h.N <= (1ll << 30), "search %lld: bad sizes (N=%lld, M=%lld)", (long long)i, (long long)h.N, (long long)h.M);
        SLAM_REQUIRE(h.M <= SLAM_MAX_TRAIN_PER_PASS, "search %lld: a batched search covers at most 2^23 train rows", (long long)i);
        SLAM_REQUIRE(h.train_base >= 0 && h.train_base + h.M <= 0x7FFFFFFFll, "search %lld: train_base + M must fit int32", (long long)i);
        if (h.N == 0) continue;
        SLAM_REQUIRE(h.d_query && h.d_idx && h.d_dist && (h.d_train || h.M == 0), "search %lld: null device pointer", (long long)i);
        SLAM_REQUIRE(((uintptr_t)h.d_query & 15) == 0 && ((uintptr_t)h.d_train & 15) == 0 &&
                     (!h_keep || ((uintptr_t)h_keep[i] & 15) == 0), "search %lld: descriptor pointers must be 16-byte aligned", (long long)i);
        rows += (h.N + 255) / 256 * 256;            // every search's state starts on a query-block boundary
        if (h.M) qb_all += (h.N + 255) / 256;
    }
    SLAM_HIP(hipSetDevice(ctx->device));
    bf_batch batch;
    memset(&batch, 0, sizeof(batch));
    poll = poll && qb_all <= SLAM_BF_DONE_FLAGS;
    int done_words = 0;
    std::vector<int> tables;
    std::vector<size_t> tbl_at;
    int64_t row0 = 0;
    int blocks = 0;
    bf_state st;
    if (rows)
        if (int rc = bf_state_get(ctx, rows, &st)) return rc;
    for (int64_t i = 0; i < B; i++) {
        const slam_bf_search& h = h_searches[i];
        if (h.N == 0) continue;
        if (h.M == 0) {                                 // no train rows: every query reports "no neighbour"
            poll = false;
            bf_fill_none_kernel<<<dim3((unsigned)((h.N + 255) / 256)), dim3(256), 0, ctx->stream>>>((int)h.N, (int2*)h.d_idx,
                                                                                                 (int2*)h.d_dist);
            SLAM_HIP(hipGetLastError());
            continue;
        }
        std::vector<int> tbl;
        const bf_plan p = make_plan(ctx, h.N, h.M, &tbl, qb_all, bf_rows_on_host(ctx, h.d_train));
        SLAM_REQUIRE(p.R == 1, "slam_bf_knn2_batch_u256 runs at one query per lane: clear the R override (slam_bf_set_tuning)");
        bf_search& d = batch.s[batch.count++];
        d.q = (const uint4*)h.d_query; d.t = (const uint4*)h.d_train;
        d.out_idx = (int2*)h.d_idx; d.out_dist = (int2*)h.d_dist;
        d.keep = h_keep ? (uint4*)h_keep[i] : nullptr;
        d.st.best = st.best + row0; d.st.bound = st.bound + row0; d.st.arrivals = st.arrivals + row0 / 256; d.st.cursor = st.cursor + row0 / 64 * SLAM_CURSOR_STRIDE;
        d.N = (int)h.N; d.lead = p.lead; d.S = p.S; d.qblocks = p.qblocks; d.train_base = (int)h.train_base;
        d.first_block = blocks; d.cold = p.cold; d.uni = p.uni; d.M = (int)h.M; d.sfeed = p.sfeed;
        done_words += p.qblocks;
        tbl_at.push_back(tables.size());
        if (!p.uni) tables.insert(tables.end(), tbl.begin(), tbl.end());
        blocks += p.qblocks * p.S;
        row0 += (h.N + 255) / 256 * 256;
    }
    if (batch.count == 0) {
        if (wait) SLAM_HIP(hipStreamSynchronize(ctx->stream));
        return SLAM_OK;
    }
    if (poll) {
        if (int rc = slam_done_block(ctx, 0)) return rc;
        batch.done = (unsigned*)ctx->sel_host;
        batch.epoch = slam_done_epoch(ctx);
    }
    const int* d_tbl = nullptr;
    if (!tables.empty())
        if (int rc = bf_table_get(ctx, tables, &d_tbl)) return rc;
    for (int i = 0; i < batch.count; i++) batch.s[i].tbl = batch.s[i].uni ? nullptr : d_tbl + tbl_at[i];
    SLAM_HIP(hipGetLastError());
    if (int rc = slam_prof_begin(ctx)) return rc;
    bf_top2_batch_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(batch);
    if (int rc = slam_prof_end(ctx)) return rc;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        (void)slam_bf_reset_state(ctx);
        return slam_set_error(SLAM_ERR_HIP, "batched top-2 kernel launch failed: %s", hipGetErrorString(e));
    }
    if (poll) return slam_wait_done(ctx, batch.done, done_words, batch.epoch);
    if (wait) SLAM_HIP(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_bf_knn2_batch_u256(slam_ctx* ctx, int64_t B, const slam_bf_search* h_searches) {
    return slam_bf_knn2_batch_keep(ctx, B, h_searches, nullptr, false);
}

extern "C" int slam_bf_merge_top2(slam_ctx* ctx, const int32_t* d_idx_parts, const int32_t* d_dist_parts,
                                  int64_t G, int64_t N, int32_t* d_idx, int32_t* d_dist) {
    SLAM_REQUIRE(ctx, "slam_bf_merge_top2: null ctx");
    SLAM_REQUIRE(G >= 1 && N >= 0 && N <= (1ll << 30), "bad sizes (G=%lld, N=%lld)", (long long)G, (long long)N);
    if (N == 0) return SLAM_OK;
    SLAM_REQUIRE(d_idx_parts && d_dist_parts && d_idx && d_dist, "slam_bf_merge_top2: null device pointer");
    SLAM_HIP(hipSetDevice(ctx->device));
    bf_merge_tables_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream>>>(
        (const int2*)d_idx_parts, (const int2*)d_dist_parts, (int)G, (int)N, (int2*)d_idx, (int2*)d_dist);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}
