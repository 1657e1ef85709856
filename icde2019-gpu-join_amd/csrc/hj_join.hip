// hj_join.hip — gfx950 device code of the join phase: work-item planning (decompose_chains, jp.cu:843-874), the LDS chained-table
// build + probe that counts (join_partitioned_aggregate, jp.cu:885-1095) or late-materialises (jp.cu:1420-1557), the one-probe
// materialiser (join_partitioned_results, jp.cu:1107-1416), general items for a skewed or larger build side (role flip,
// jp.cu:929-1003), and the result reductions.  What the file does instead of the reference's kernels: see hj_part.hip's header
// and DESIGN.md §3.
#include "hj_device.h"

namespace hj {

// ------------------------------------------------------------------------------------------------
// join: plan → (count) → scan → (materialise)
// ------------------------------------------------------------------------------------------------

// items per partition: probe partition cut into chunks of <= chunk tuples (decompose_chains,
// jp.cu:843-874, threshold = 2*bucket_size at hjcp.cu:904); no item when either side is empty.
// bflag / pflag: overflow flags of relations whose histogram-free passes were queued (nullptr otherwise).  A raised
// flag means the ranges are not valid: no items, the join kernels then do nothing and the host redoes the relation.
// Thread 0 also zeroes the two result accumulators of k_sum2 and the output cursor of k_join_mat_reg.
// Probe side in RANGES with several ranges per partition (sampled path): the ranges of partition p are r0[p] + j * stride,
// j < nr[p].  Whole ranges are packed into LIST items of <= chunk probe tuples (the table of the partition is built once for all
// of them); a range longer than a chunk is cut into chunk items as above.  emit(index, list, q0, q1): list items carry
// (first range, number of ranges) in (q0, q1).
// A range longer than a chunk goes to emit_big(index of its first item, b, e, number of chunk items) as a whole: the heavy hitter of
// config 4 is one partition of ~1700 chunk items, which ONE thread used to write one after the other (k_join_expand 100 us).
template <class F, class G>
__device__ inline uint32_t walk_ranges(const uint64_t *__restrict__ pbeg, const uint64_t *__restrict__ pend, uint32_t r0, uint32_t nr,
                                       uint32_t stride, uint32_t chunk, F emit, G emit_big) {
    uint32_t items = 0, run0 = 0, runlen = 0;
    uint64_t acc = 0;
    // the ranges are fetched eight at a time, ahead of the data-dependent packing below: one thread walks its partition's list alone
    // and a dependent global load per range (~2 us each) made the two planning kernels of config 4 80-105 us long (the 64 partitions
    // of the pass-1 digit that holds the heavy hitter have ~28 ranges each)
    constexpr uint32_t WR_BATCH = 8;
    uint64_t bb[WR_BATCH], ee[WR_BATCH];
    for (uint32_t j = 0; j < nr; j++) {
        if ((j & (WR_BATCH - 1)) == 0) {
#pragma unroll
            for (uint32_t t = 0; t < WR_BATCH; t++) {
                const uint32_t rt = r0 + (j + t < nr ? j + t : nr - 1) * stride;
                bb[t] = pbeg[rt]; ee[t] = pend[rt];
            }
        }
        const uint32_t r = r0 + j * stride;
        uint64_t b = bb[0], e = ee[0];
#pragma unroll
        for (uint32_t t = 1; t < WR_BATCH; t++) if ((j & (WR_BATCH - 1)) == t) { b = bb[t]; e = ee[t]; }
        const uint64_t len = e - b;
        if (len > chunk) {
            if (runlen && acc) { emit(items, true, (uint64_t)run0, (uint64_t)runlen); items++; }
            runlen = 0; acc = 0;
            const uint32_t nck = (uint32_t)((len + chunk - 1) / chunk);
            emit_big(items, b, e, nck);
            items += nck;
        } else {
            if (acc + len > chunk) { emit(items, true, (uint64_t)run0, (uint64_t)runlen); items++; runlen = 0; acc = 0; }
            if (!runlen) run0 = r;
            runlen++; acc += len;
        }
    }
    if (runlen && acc) { emit(items, true, (uint64_t)run0, (uint64_t)runlen); items++; }
    return items;
}

__global__ void k_join_plan(const uint64_t *__restrict__ bbeg, const uint64_t *__restrict__ bend,
                            const uint64_t *__restrict__ pbeg, const uint64_t *__restrict__ pend,
                            uint32_t nparts, uint32_t chunk, uint32_t *__restrict__ items_cnt,
                            const uint32_t *__restrict__ bflag, const uint32_t *__restrict__ pflag,
                            uint64_t *__restrict__ zero2, uint64_t *__restrict__ zero_cursor, const uint32_t *__restrict__ rpart,
                            const uint32_t *__restrict__ pr0, const uint32_t *__restrict__ pnr, uint32_t rstride) {
    // nparts = probe RANGES; the build partition of range i is rpart[i] (sampled path, one item list per range) or i.
    // pr0 != nullptr: nparts = PARTITIONS, each with a list of ranges (walk_ranges)
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { zero2[0] = 0; zero2[1] = 0; *zero_cursor = 0; }
    if (i >= nparts) return;
    if ((bflag && *bflag) || (pflag && *pflag)) { items_cnt[i] = 0; return; }
    if (pr0) {
        items_cnt[i] = bend[i] != bbeg[i] ? walk_ranges(pbeg, pend, pr0[i], pnr[i], rstride, chunk, [](uint32_t, bool, uint64_t, uint64_t) {}, [](uint32_t, uint64_t, uint64_t, uint32_t) {}) : 0u;
        return;
    }
    const uint32_t p = rpart ? rpart[i] : i;
    uint64_t nb = bend[p] - bbeg[p], np = pend[i] - pbeg[i];
    items_cnt[i] = (nb && np) ? (uint32_t)((np + chunk - 1) / chunk) : 0u;
}

// items_cnt has been scanned (local + chunk prefix): write the item list and the item count.
__global__ void k_join_expand(const uint64_t *__restrict__ bbeg, const uint64_t *__restrict__ bend,
                              const uint64_t *__restrict__ pbeg, const uint64_t *__restrict__ pend,
                              uint32_t nparts, uint32_t chunk, const uint32_t *__restrict__ items_scanned,
                              const uint64_t *__restrict__ chunk_prefix, JoinItem *__restrict__ items,
                              const uint32_t *__restrict__ bflag, const uint32_t *__restrict__ pflag, const uint32_t *__restrict__ rpart,
                              const uint32_t *__restrict__ pr0, const uint32_t *__restrict__ pnr, uint32_t rstride) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if ((bflag && *bflag) || (pflag && *pflag)) return; // ranges not valid (k_join_plan counted no items)
    if (pr0) { // (no lane leaves before the wave-wide part below)
        const bool act = i < nparts;
        const uint64_t at = act ? (uint64_t)items_scanned[i] + chunk_prefix[i >> SCAN_CHUNK_LOG] : 0;
        const uint64_t b0 = act ? bbeg[i] : 0, nb = act ? bend[i] - b0 : 0;
        const uint32_t r0 = act ? pr0[i] : 0u, nr = (act && nb) ? pnr[i] : 0u;
        bool has_big = false;
        // the thread's own walk writes the list items; ranges longer than a chunk are only noted ...
        walk_ranges(pbeg, pend, r0, nr, rstride, chunk, [&](uint32_t idx, bool list, uint64_t q0, uint64_t q1) {
            JoinItem it;
            it.b0 = b0; it.nb = (uint32_t)nb; it.p = i | (list ? JOIN_ITEM_LIST : 0u);
            it.q0 = q0; it.q1 = q1;
            items[at + idx] = it;
        }, [&](uint32_t, uint64_t, uint64_t, uint32_t) { has_big = true; });
        // ... and written by the whole wave: every lane repeats the walk of a partition that has some (uniform values: scalar loads),
        // the chunk items of a long range are dealt to the 64 lanes
        const uint32_t ln = lane_id();
        for (uint64_t pending = __ballot(has_big); pending; pending &= pending - 1) {
            const int L = __ffsll((unsigned long long)pending) - 1;
            const uint64_t atL = uniform64(__shfl(at, L)), b0L = uniform64(__shfl(b0, L));
            const uint32_t nbL = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((uint32_t)nb, L));
            const uint32_t iL = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(i, L));
            const uint32_t r0L = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(r0, L)), nrL = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(nr, L));
            walk_ranges(pbeg, pend, r0L, nrL, rstride, chunk, [](uint32_t, bool, uint64_t, uint64_t) {},
                        [&](uint32_t idx0, uint64_t b, uint64_t e, uint32_t nck) {
                for (uint32_t ck = ln; ck < nck; ck += 64) {
                    JoinItem it;
                    it.b0 = b0L; it.nb = nbL; it.p = iL;
                    it.q0 = b + (uint64_t)ck * chunk; it.q1 = it.q0 + chunk < e ? it.q0 + chunk : e;
                    items[atL + idx0 + ck] = it;
                }
            });
        }
        return;
    }
    if (i >= nparts) return;
    uint64_t at = (uint64_t)items_scanned[i] + chunk_prefix[i >> SCAN_CHUNK_LOG];
    const uint32_t p = rpart ? rpart[i] : i;
    uint64_t nb = bend[p] - bbeg[p], np = pend[i] - pbeg[i];
    uint32_t c = (nb && np) ? (uint32_t)((np + chunk - 1) / chunk) : 0u;
    // a self-contained descriptor per item: the join workgroup reads ONE 32-byte record and goes straight to the data
    // (not item -> partition -> four range loads: every dependent global load is ~2 us under load)
    for (uint32_t j = 0; j < c; j++) {
        JoinItem it;
        it.b0 = bbeg[p]; it.nb = (uint32_t)nb; it.p = p;
        it.q0 = pbeg[i] + (uint64_t)j * chunk;
        it.q1 = it.q0 + chunk < pend[i] ? it.q0 + chunk : pend[i];
        items[at + j] = it;
    }
}

// ---- general items (a skewed build relation: JoinArgs.general) ----
// A side of the join as the planner sees it: partition i is range i (r0 == nullptr) or the list of ranges r0[i] + j * stride, j < nr[i].
struct SideRef { const uint64_t *beg, *end; const uint32_t *r0, *nr; uint32_t stride; };
__device__ inline uint64_t side_size(const SideRef &s, uint32_t i) {
    if (!s.r0) return s.end[i] - s.beg[i];
    uint64_t tot = 0;
    for (uint32_t j = 0; j < s.nr[i]; j++) { const uint32_t r = s.r0[i] + j * s.stride; tot += s.end[r] - s.beg[r]; }
    return tot;
}
// The items of partition i.  With a skewed build relation the table side is chosen PER PARTITION: the smaller of the two partitions
// builds (the reference flips the roles for build partitions that do not fit its table, jp.cu:929-1003).  That turns ONE workgroup
// looping over hundreds of table chunks of a heavy hitter into one item per chunk of the streamed side, and it keeps a key with
// thousands of duplicates out of the table, where it would be one chain that every probe of that key walks link by link (count) or
// one output round per duplicate (materialisation): measured at 2^24 x 2^27 Zipf with the Zipf side designated to build, 1.5 s
// per step without flipping, 6.8 ms flipping only oversize partitions, 1.5 ms with this rule (profiles/r4_skewed_build.txt).
// The streamed side is cut into items exactly as the probe side always was (walk_ranges / chunks).  emit(index, item).
template <class F>
__device__ inline uint32_t general_items(const JoinArgs &a, uint32_t i, F emit) {
    const SideRef B{a.bbeg, a.bend, a.br0, a.bnr, a.bstride}, P{a.pbeg, a.pend, a.pr0, a.pnr, a.rstride};
    const uint64_t nb = side_size(B, i), np = side_size(P, i);
    if (!nb || !np) return 0;
    const bool swap = np < nb;
    const SideRef &T = swap ? P : B, &S = swap ? B : P;
    JoinItem base;
    base.p = i | (swap ? JOIN_ITEM_SWAP : 0u) | (T.r0 ? JOIN_ITEM_BLIST : 0u);
    if (T.r0) { base.b0 = T.r0[i]; base.nb = T.nr[i]; }
    else { base.b0 = T.beg[i]; base.nb = (uint32_t)(T.end[i] - T.beg[i]); }
    base.q0 = 0; base.q1 = 0;
    if (S.r0)
        return walk_ranges(S.beg, S.end, S.r0[i], S.nr[i], S.stride, a.chunk, [&](uint32_t idx, bool list, uint64_t q0, uint64_t q1) {
            JoinItem it = base;
            it.q0 = q0; it.q1 = q1; if (list) it.p |= JOIN_ITEM_LIST;
            emit(idx, it);
        }, [&](uint32_t idx0, uint64_t b, uint64_t e, uint32_t nck) {
            for (uint32_t ck = 0; ck < nck; ck++) {
                JoinItem it = base;
                it.q0 = b + (uint64_t)ck * a.chunk; it.q1 = it.q0 + a.chunk < e ? it.q0 + a.chunk : e;
                emit(idx0 + ck, it);
            }
        });
    uint32_t n = 0;
    for (uint64_t q = S.beg[i]; q < S.end[i]; q += a.chunk, n++) {
        JoinItem it = base;
        it.q0 = q; it.q1 = q + a.chunk < S.end[i] ? q + a.chunk : S.end[i];
        emit(n, it);
    }
    return n;
}

__global__ void k_join_plan_gen(JoinArgs a, uint32_t nparts, uint32_t *__restrict__ items_cnt, uint64_t *__restrict__ zero2, uint64_t *__restrict__ zero_cursor) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { zero2[0] = 0; zero2[1] = 0; *zero_cursor = 0; }
    if (i >= nparts) return;
    if ((a.bflag && *a.bflag) || (a.pflag && *a.pflag)) { items_cnt[i] = 0; return; }
    items_cnt[i] = general_items(a, i, [](uint32_t, const JoinItem &) {});
}
__global__ void k_join_expand_gen(JoinArgs a, uint32_t nparts, const uint32_t *__restrict__ items_scanned, const uint64_t *__restrict__ chunk_prefix,
                                  JoinItem *__restrict__ items) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nparts) return;
    if ((a.bflag && *a.bflag) || (a.pflag && *a.pflag)) return;
    const uint64_t at = (uint64_t)items_scanned[i] + chunk_prefix[i >> SCAN_CHUNK_LOG];
    general_items(a, i, [&](uint32_t idx, const JoinItem &it) { items[at + idx] = it; });
}

// the probe ranges of an item: one chunk [q0, q1), or (list items) range rr of it.q1 whole ranges starting at range it.q0
__device__ inline uint32_t item_nranges(const JoinItem &it) { return (it.p & JOIN_ITEM_LIST) ? (uint32_t)it.q1 : 1u; }
__device__ inline void item_range(const JoinArgs &a, const JoinItem &it, uint32_t rr, uint64_t &q0, uint64_t &q1) {
    if (it.p & JOIN_ITEM_LIST) { const uint32_t r = (uint32_t)it.q0 + rr * a.rstride; q0 = a.pbeg[r]; q1 = a.pend[r]; }
    else { q0 = it.q0; q1 = it.q1; }
}

// k_join_plan + scan + k_join_expand in ONE single-workgroup launch, for partition counts where three dependent launches
// cost more than the work (a step at <= 2^24 tuples is launch-latency bound): chunks of 1024 partitions with a running carry.
__global__ __launch_bounds__(1024) void k_join_plan_fused(const uint64_t *__restrict__ bbeg, const uint64_t *__restrict__ bend,
                                                          const uint64_t *__restrict__ pbeg, const uint64_t *__restrict__ pend,
                                                          uint32_t nparts, uint32_t chunk, JoinItem *__restrict__ items,
                                                          const uint32_t *__restrict__ bflag, const uint32_t *__restrict__ pflag,
                                                          uint64_t *__restrict__ zero2, uint64_t *__restrict__ zero_cursor,
                                                          uint64_t *__restrict__ n_items) {
    __shared__ uint32_t scratch[17];
    if (threadIdx.x == 0) { zero2[0] = 0; zero2[1] = 0; *zero_cursor = 0; }
    const bool invalid = (bflag && *bflag) || (pflag && *pflag);
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nparts; base += 1024) {
        const uint32_t p = base + threadIdx.x;
        uint64_t nb = 0, np = 0;
        if (p < nparts && !invalid) { nb = bend[p] - bbeg[p]; np = pend[p] - pbeg[p]; }
        const uint32_t c = (nb && np) ? (uint32_t)((np + chunk - 1) / chunk) : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan<uint32_t>(c, scratch, &total);
        for (uint32_t j = 0; j < c; j++) {
            JoinItem it;
            it.b0 = bbeg[p]; it.nb = (uint32_t)nb; it.p = p;
            it.q0 = pbeg[p] + (uint64_t)j * chunk;
            it.q1 = it.q0 + chunk < pend[p] ? it.q0 + chunk : pend[p];
            items[carry + ex + j] = it;
        }
        carry += total;
    }
    if (threadIdx.x == 0) *n_items = carry;
}

// The same in one launch for ANY partition count: 1024 partitions per workgroup, item slots reserved with ONE atomic per workgroup
// on *n_items (zeroed before the launch: by the pass-2 kernel, FastArgs.zero_items, or by the host).  The items of a partition stay
// together, partitions of a workgroup stay in order, workgroups land in the order their atomics arrive: no consumer depends on the
// order of the item list.  2^15 partitions (2^27 x 2^27): one launch of 32 workgroups instead of plan + scan (2) + expand;
// 2^18 partitions: 256 atomics on one word (~3 us at the ~88 returning atomics per us one address sustains).
__global__ __launch_bounds__(1024) void k_join_plan_atomic(const uint64_t *__restrict__ bbeg, const uint64_t *__restrict__ bend,
                                                           const uint64_t *__restrict__ pbeg, const uint64_t *__restrict__ pend,
                                                           uint32_t nparts, uint32_t chunk, JoinItem *__restrict__ items,
                                                           const uint32_t *__restrict__ bflag, const uint32_t *__restrict__ pflag,
                                                           uint64_t *__restrict__ zero2, uint64_t *__restrict__ zero_cursor,
                                                           unsigned long long *__restrict__ n_items) {
    __shared__ uint32_t scratch[17];
    __shared__ unsigned long long base_s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { zero2[0] = 0; zero2[1] = 0; *zero_cursor = 0; }
    if ((bflag && *bflag) || (pflag && *pflag)) return; // ranges not valid: no items (n_items stays 0)
    const uint32_t p = blockIdx.x * 1024 + threadIdx.x;
    uint64_t b0 = 0, nb = 0, q0 = 0, q1 = 0;
    if (p < nparts) { b0 = bbeg[p]; nb = bend[p] - b0; q0 = pbeg[p]; q1 = pend[p]; }
    const uint32_t c = (nb && q1 > q0) ? (uint32_t)((q1 - q0 + chunk - 1) / chunk) : 0u;
    uint32_t total;
    const uint32_t ex = block_excl_scan<uint32_t>(c, scratch, &total);
    if (threadIdx.x == 0) base_s = total ? atomicAdd(n_items, (unsigned long long)total) : 0ull;
    __syncthreads();
    const uint64_t at = base_s + ex;
    for (uint32_t j = 0; j < c; j++) {
        JoinItem it;
        it.b0 = b0; it.nb = (uint32_t)nb; it.p = p;
        it.q0 = q0 + (uint64_t)j * chunk;
        it.q1 = it.q0 + chunk < q1 ? it.q0 + chunk : q1;
        items[at + j] = it;
    }
}

hipError_t launch_join_plan_atomic(hipStream_t st, const JoinArgs &a, uint32_t nparts, JoinItem *items, uint64_t *zero2, uint64_t *zero_cursor,
                                   uint64_t *n_items) {
    hipLaunchKernelGGL(k_join_plan_atomic, dim3((nparts + 1023) / 1024), dim3(1024), 0, st, a.bbeg, a.bend, a.pbeg, a.pend, nparts, a.chunk, items,
                       a.bflag, a.pflag, zero2, zero_cursor, reinterpret_cast<unsigned long long *>(n_items));
    return hipGetLastError();
}

hipError_t launch_join_plan_fused(hipStream_t st, const JoinArgs &a, uint32_t nparts, JoinItem *items, uint64_t *zero2, uint64_t *zero_cursor,
                                  uint64_t *n_items) {
    hipLaunchKernelGGL(k_join_plan_fused, dim3(1), dim3(1024), 0, st, a.bbeg, a.bend, a.pbeg, a.pend, nparts, a.chunk, items, a.bflag, a.pflag,
                       zero2, zero_cursor, n_items);
    return hipGetLastError();
}

// LDS layout (dynamic): head[nh] u32 | entries[cap] 8 bytes ({tag16 << 16 | next16, payload} or {key, payload}) | with full
// keys: next[cap] u16.  The reference's table: elem int16 tag, payload int32, next int16, head int32[1024] (jp.cu:899-902):
// the same 8 bytes per tuple, here laid out so that one 8-byte LDS load per chain hop fetches tag, link and payload.
// GEN (general items: the build relation is skewed — host: hj_api.hip plan_join): the TABLE side of an item may be a LIST of ranges
// (a sampled build relation: JOIN_ITEM_BLIST, b0 = first range, nb = number of ranges) that is built into the LDS table piece by
// piece, the table taking the next cap tuples of the concatenated ranges per chunk; and the roles may be FLIPPED for the item
// (JOIN_ITEM_SWAP): the table is built from the relation the host calls probe side and the designated build side is streamed —
// what the reference does for build partitions that do not fit its table (jp.cu:929-1003).  Count and aggregate are symmetric.
template <bool TAG16, int JM, bool GEN = false>
__global__ __launch_bounds__(JOIN_THREADS) void k_join(JoinArgs a) {
    static_assert(JM == 0 || JM == 2, "JM: 0 = count + aggregate, 2 = late materialisation (the second probe of round 2's two-probe materialiser, JM 1, is gone)");
    static_assert(!GEN || JM == 0, "general items: count kernel and k_join_mat_reg only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t item = blockIdx.x;
    if (item >= *a.n_items) return;
    // LDS: head[nh] u32 | ent[cap] 8-byte entries | (full keys only) next[cap] u16.  An entry is read with ONE 8-byte LDS
    // load per chain hop: TAG16 {tag << 16 | next, payload}; full keys {key, payload} (+ the separate next link).
    uint32_t *head = reinterpret_cast<uint32_t *>(smem);
    uint2 *ent = reinterpret_cast<uint2 *>(smem + (size_t)a.nh * 4);
    uint16_t *lnext = reinterpret_cast<uint16_t *>(smem + (size_t)a.nh * 4 + (size_t)a.cap * 8);

    const uint32_t tid = threadIdx.x, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)); // scalar: the probe stream state stays in SGPRs
    const JoinItem it = a.items[item];
    const uint32_t nr = item_nranges(it); // probe ranges of the item (list items: several whole ranges share one table build)
    const uint32_t bits = a.radix_bits, nhm = a.nh - 1, tsh = a.radix_bits > 16u ? a.radix_bits : 16u; // tag = key >> tsh: see plan_join
    // the two sides of the item (wave-uniform; without GEN: the build relation is the table, the probe relation the stream)
    const bool swap = GEN && (it.p & JOIN_ITEM_SWAP), blist = GEN && (it.p & JOIN_ITEM_BLIST);
    const int32_t *const tk = swap ? a.pk : a.bk, *const tp = swap ? a.pp : a.bp, *const sk = swap ? a.bk : a.pk, *const sp = swap ? a.bp : a.pp;
    const uint64_t t_nalloc = swap ? a.p_nalloc : a.b_nalloc, s_nalloc = swap ? a.b_nalloc : a.p_nalloc;
    const uint64_t *const tbeg = swap ? a.pbeg : a.bbeg, *const tend = swap ? a.pend : a.bend, *const sbeg = swap ? a.bbeg : a.pbeg, *const send = swap ? a.bend : a.pend;
    const uint32_t tstride = swap ? a.rstride : a.bstride, sstride = swap ? a.bstride : a.rstride;
    auto stream_range = [&](uint32_t rr_, uint64_t &q0_, uint64_t &q1_) {
        if (it.p & JOIN_ITEM_LIST) { const uint32_t r = (uint32_t)it.q0 + rr_ * sstride; q0_ = sbeg[r]; q1_ = send[r]; }
        else { q0_ = it.q0; q1_ = it.q1; }
    };
    // table cursor: range tr of ntr, [tb, te) = what is left of it
    const uint32_t ntr = blist ? it.nb : 1u;
    uint32_t tr = 0;
    uint64_t tb, te;
    auto table_range = [&](uint32_t j) {
        if (blist) { const uint32_t r = (uint32_t)it.b0 + j * tstride; tb = tbeg[r]; te = tend[r]; }
        else { tb = it.b0; te = it.b0 + it.nb; }
    };
    table_range(0);
    if (GEN) while (tb == te && tr + 1 < ntr) table_range(++tr);
    // TAG16: the entry stores key >> max(bits, 16) — 16 bits.  The key bits below that which are not radix bits, [bits, 16), are part of
    // the bucket index (key >> bits) & (nh - 1), so two keys of one chain that agree in the tag are equal: exact whenever
    // bits + log2(nh) >= 16 (round 6; until then: only at >= 16 radix bits).  The reference takes its tag shortcut, jp.cu:1029, at any
    // bit count (D2).  Otherwise the table stores full keys
    auto hidx = [&](uint32_t key) -> uint32_t { return (key >> bits) & nhm; };

    uint64_t my_matches = 0, my_agg = 0;
    while (tb < te) { // one table chunk per iteration: the next cap tuples of the table side
        uint64_t gb = tb;
        uint32_t nbc = (uint32_t)(te - tb < a.cap ? te - tb : a.cap), filled = 0;
        // ---- build: tuple j of the chunk lives in slot j; LIFO chain insert by atomic exchange on
        // the bucket head (jp.cu:1021-1048).  Loads of three iterations (6144 tuples: a whole default-size table) are
        // in flight at a time; the first three are issued BEFORE the heads are initialised (they need no LDS) ----
        uint64_t i0 = (gb & ~(uint64_t)3) + (uint64_t)tid * 4;
        int4 bkv[3], bpv[3];
        auto bload = [&]() {
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const uint64_t i = i0 + (uint64_t)r * JOIN_THREADS * 4;
                if (i < gb + nbc) { bkv[r] = load4(tk, i, t_nalloc); bpv[r] = load4(tp, i, t_nalloc); }
            }
        };
        bload();
        // the wave's probe stream: 256 tuples at w0, w0 + 2048, ... of range rr, then on into the item's next range (list items) —
        // (rr, nq0, nq1, w0) is wave-uniform.  The loads of the NEXT position are always in flight while the current one is probed.
        uint32_t rr = 0;
        uint64_t nq0, nq1;
        stream_range(0, nq0, nq1);
        uint64_t w0 = (nq0 & ~(uint64_t)3) + (uint64_t)wave * 256;
        auto skip_empty = [&]() { // this wave has nothing (left) in range rr: on to the next one
            while (w0 >= nq1 && rr + 1 < nr) { rr++; stream_range(rr, nq0, nq1); w0 = (nq0 & ~(uint64_t)3) + (uint64_t)wave * 256; }
        };
        skip_empty();
        // TWO blocks of the stream are requested ahead of the one being probed, the first two before the table is built: their loads fly
        // during the build (same box, three rounds: k_join 0.454-0.461 -> 0.435-0.441 ms at 2^27, 2.86-2.88 -> 2.80-2.81 at 2^30; a long
        // stream — config 4 — does not care: 3.25 both)
        uint64_t w1, q01, q11, w2, q02, q12; // the staged blocks: position and range (wave-uniform)
        int4 nk = make_int4(0, 0, 0, 0), np = make_int4(0, 0, 0, 0), nk2 = make_int4(0, 0, 0, 0), np2 = make_int4(0, 0, 0, 0);
        auto issue = [&](int4 &k_, int4 &p_, uint64_t &w_, uint64_t &q0_, uint64_t &q1_) { // the block at the cursor; the cursor moves on
            w_ = w0; q0_ = nq0; q1_ = nq1;
            k_ = make_int4(0, 0, 0, 0); p_ = make_int4(0, 0, 0, 0);
            const uint64_t i_ = w0 + (uint64_t)lane_id() * 4;
            if (i_ < nq1) { k_ = load4(sk, i_, s_nalloc); p_ = load4(sp, i_, s_nalloc); }
            if (w0 < nq1) { w0 += (uint64_t)JOIN_THREADS * 4; skip_empty(); }
        };
        issue(nk, np, w1, q01, q11);
        if (!GEN) issue(nk2, np2, w2, q02, q12); // (general items: one block ahead — the second one costs the third workgroup per CU: 85 VGPRs)
        for (uint32_t i = tid; i < a.nh; i += JOIN_THREADS) head[i] = 0xFFFFFFFFu;
        __syncthreads();
        for (;;) { // the pieces of this chunk: [gb, gb + nbc) goes to slots filled ... (one piece unless the table side is a list)
            while (i0 < gb + nbc) {
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const uint64_t i = i0 + (uint64_t)r * JOIN_THREADS * 4;
                    if (i < gb + nbc) {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            uint64_t idx = i + e;
                            if (idx >= gb && idx < gb + nbc) {
                                const uint32_t slot = filled + (uint32_t)(idx - gb), key = (uint32_t)elem(bkv[r], e);
                                const uint32_t old = atomicExch(&head[hidx(key)], slot);
                                if (TAG16) ent[slot] = make_uint2(((key >> tsh) << 16) | (old & 0xFFFFu), (uint32_t)elem(bpv[r], e));
                                else { ent[slot] = make_uint2(key, (uint32_t)elem(bpv[r], e)); lnext[slot] = (uint16_t)old; }
                            }
                        }
                    }
                }
                i0 += (uint64_t)JOIN_THREADS * 4 * 3;
                if (i0 < gb + nbc) bload();
            }
            filled += nbc; tb += nbc;
            if (!GEN) break;
            while (tb == te && tr + 1 < ntr) table_range(++tr); // the next range that holds something
            if (tb == te || filled == a.cap) break;              // table side exhausted, or the table is full
            gb = tb;
            nbc = (uint32_t)(te - tb < a.cap - filled ? te - tb : a.cap - filled);
            i0 = (gb & ~(uint64_t)3) + (uint64_t)tid * 4;
            bload();
        }
        __syncthreads();
        // ---- probe ----
        // the loop bound is wave-uniform (w0), so every lane of a wave stays in the loop together:
        // the ballot ranks and the wave's output cursor depend on it.  The next iteration's loads are issued before
        // this iteration's chains are walked.
        while (w1 < q11) {
            const uint64_t i = w1 + (uint64_t)lane_id() * 4, q0 = q01, q1 = q11;
            const int4 kv = nk, pv = np;
            if (!GEN) { nk = nk2; np = np2; w1 = w2; q01 = q02; q11 = q12; issue(nk2, np2, w2, q02, q12); }
            else issue(nk, np, w1, q01, q11);
            if (JM == 0) {
                // count-only: the four bucket heads of this lane's four tuples are fetched first and the four
                // chains are walked in lockstep, so their LDS reads overlap instead of completing one by one
                uint32_t pos4[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint64_t idx = i + e;
                    const bool valid = idx >= q0 && idx < q1;
                    pos4[e] = (valid ? head[hidx((uint32_t)elem(kv, e))] : 0xFFFFFFFFu) & 0xFFFFu;
                }
                while ((pos4[0] & pos4[1] & pos4[2] & pos4[3]) != 0xFFFFu) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const uint32_t pos = pos4[e];
                        if (pos != 0xFFFFu) {
                            const uint32_t key = (uint32_t)elem(kv, e);
                            const uint2 en = ent[pos];
                            const bool eq = TAG16 ? ((en.x >> 16) == (key >> tsh)) : (en.x == key);
                            if (eq) {
                                my_matches++;
                                my_agg += (uint64_t)((int64_t)(int32_t)en.y * (int64_t)elem(pv, e));
                            }
                            pos4[e] = TAG16 ? (en.x & 0xFFFFu) : (uint32_t)lnext[pos];
                        }
                    }
                }
            } else
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint64_t idx = i + e;
                const bool valid = idx >= q0 && idx < q1;
                const uint32_t key = (uint32_t)elem(kv, e);
                const int32_t ppay = elem(pv, e);
                uint32_t pos = valid ? head[hidx(key)] : 0xFFFFFFFFu;
                pos &= 0xFFFFu; // chain links are 16 bit; 0xFFFF = end
                while (pos != 0xFFFFu) {
                    const uint2 en = ent[pos];
                    const bool eq = TAG16 ? ((en.x >> 16) == (key >> tsh)) : (en.x == key);
                    if (eq) {
                        my_matches++;
                        if (JM == 2) {
                            // late materialisation (join_partitioned_varpayload, jp.cu:1524-1533): payloads are
                            // row ids; gather the extra columns of both sides and add them up
                            const int32_t bval = (int32_t)en.y;
                            int64_t acc = 0;
                            for (uint32_t z = 0; z < a.ncp; z++) acc += a.Dp[(uint64_t)(uint32_t)ppay + z * a.sp];
                            for (uint32_t z = 0; z < a.ncb; z++) acc += a.Db[(uint64_t)(uint32_t)bval + z * a.sb];
                            my_agg += (uint64_t)acc;
                        } else {
                            my_agg += (uint64_t)((int64_t)(int32_t)en.y * (int64_t)ppay);
                        }
                    }
                    pos = TAG16 ? (en.x & 0xFFFFu) : (uint32_t)lnext[pos];
                }
            }
        }
        __syncthreads();
    }
    my_matches = wave_sum64(my_matches);
    my_agg = wave_sum64(my_agg);
    if (lane_id() == 0) {
        a.wave_counts[(uint64_t)item * JOIN_WAVES + wave] = my_matches;
        a.wave_agg[(uint64_t)item * JOIN_WAVES + wave] = my_agg;
    }
#ifdef HJ_STAMPS
    if (a.stamps && threadIdx.x == 0) {
        unsigned long long *s = a.stamps + (size_t)item * 4;
        // END stamps only: a clock read at the start of this kernel — kept in a register, parked in LDS or stored at once — takes it from 72 to
        // 94 VGPRs, i.e. from three workgroups per CU to two (measured: 2.83 -> 3.55 ms at 2^30); the timeline is rebuilt from the ends per CU
        s[0] = 0; s[1] = 0; s[2] = hj_now(); s[3] = hj_where();
    }
#endif
}

// ---- materialisation in ONE probe, matches held in REGISTERS ----
// The reference's lead timed run writes its output in the same probe that finds the matches: matching lanes are ranked by a ballot
// into a small shared-memory staging block and one reservation on a global counter is taken per flush (join_partitioned_results,
// jp.cu:1228-1261, 1358-1388: 16 pairs per warp, one atomicAdd per 32 ints).  Sized for gfx950: ONE exact reservation on the output
// cursor per round for the whole workgroup (an item of the default shape = one ~4096-tuple partition pair takes one returning atomic
// of its exact match count: 2^30 matches = 2.6e5 atomics instead of the reference's 3.4e7), output gap-free, in no particular order
// (as in the reference: atomics decide).  Round 3's first version staged the matches in a 27-KiB LDS block (k_join_mat: two
// workgroups per CU instead of the count kernel's three, a resumable probe loop; 5.9-6.6 ms against 5.2-5.3 here; removed in round 4,
// profiles/r3_materialize_breakdown.txt).  Here nothing is staged in LDS: a lane keeps the probe
// tuples of a sub-chunk in registers and, per ROUND, the slot of the next match of
// each of them.  A round = every tuple advances to its next match (four chains in lockstep) -> ballots rank the matches inside the
// wave, the eight wave totals meet in LDS -> ONE exact reservation on the output cursor for the whole workgroup -> every wave
// writes its matches as runs of coalesced 4-byte-per-lane stores (key and probe payload from registers, build payload from the
// table entry, whose link is also where the next round starts).  Unique build keys: one productive round per sub-chunk and one
// that finds nothing — not run where the table was checked for repeated keys behind its build (tables that serve many sub-chunks);
// duplicates take as many rounds as the longest run of equal keys.  LDS = the table alone: 3 workgroups per CU.
// tuples per lane per sub-chunk: two 16-byte groups + one 8-byte group = 10 -> 5120 probe tuples per sub-chunk: a ~4096-tuple
// partition plus 8 sigma (4608) in ONE sub-chunk (one reservation), at 30 state registers instead of the 36 that three 16-byte
// groups need (which spill at 80 VGPRs = three workgroups per CU)
constexpr int MR_IT = 3;
#define MR_NE(t) ((t) == 2 ? 2 : 4)
constexpr uint32_t MR_SUB = 2 * JOIN_THREADS * 4 + JOIN_THREADS * 2;
// The body is a function of its own so that flipped roles (SWAP) are a template parameter: either way every column pointer is a
// kernel argument the compiler can fetch where it is used, instead of twenty selected pointers held in registers for the whole
// item (40-56 bytes per lane of scratch that way).
template <bool TAG16, bool LISTS, bool GEN, bool SWAP>
__device__ __forceinline__ void join_mat_reg_item(const JoinArgs &a, const JoinItem &it, unsigned char *smem) {
    uint32_t *head = reinterpret_cast<uint32_t *>(smem);
    uint2 *ent = reinterpret_cast<uint2 *>(smem + (size_t)a.nh * 4);
    uint16_t *lnext = reinterpret_cast<uint16_t *>(smem + (size_t)a.nh * 4 + (size_t)a.cap * 8);
    const size_t tbl = ((size_t)a.nh * 4 + (size_t)a.cap * 8 + (TAG16 ? 0 : (size_t)a.cap * 2) + 15) & ~(size_t)15;
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + tbl); // [2][JOIN_WAVES] wave totals (round parity) | [16],[17] base lo/hi

    // general items: two 16-byte groups per lane (4096-tuple sub-chunks) — the third group's 9 registers are what the table and
    // stream cursors of both roles need to stay out of scratch at three workgroups per CU
    constexpr int IT = GEN ? 2 : MR_IT;
    constexpr uint32_t SUB = GEN ? 2 * JOIN_THREADS * 4 : MR_SUB;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, ln = lane_id();
    const uint32_t nr = (LISTS || GEN) ? item_nranges(it) : 1u;
    const uint32_t bits = a.radix_bits, nhm = a.nh - 1, tsh = a.radix_bits > 16u ? a.radix_bits : 16u; // tag = key >> tsh: see plan_join
    auto hidx = [&](uint32_t key) -> uint32_t { return (key >> bits) & nhm; }; // see k_join
    const uint64_t lt_mask = ((uint64_t)1 << ln) - 1;
    constexpr uint32_t END = 0xFFFFu;
    // the two sides of the item, as in k_join; flipped roles also flip the two payload columns of the output
    constexpr bool swap = SWAP;
    const bool blist = GEN && (it.p & JOIN_ITEM_BLIST);
    const int32_t *const tk = swap ? a.pk : a.bk, *const tp = swap ? a.pp : a.bp, *const sk = swap ? a.bk : a.pk, *const sp = swap ? a.bp : a.pp;
    const uint64_t t_nalloc = swap ? a.p_nalloc : a.b_nalloc, s_nalloc = swap ? a.b_nalloc : a.p_nalloc;
    const uint64_t *const tbeg = swap ? a.pbeg : a.bbeg, *const tend = swap ? a.pend : a.bend, *const sbeg = swap ? a.bbeg : a.pbeg, *const send = swap ? a.bend : a.pend;
    const uint32_t tstride = swap ? a.rstride : a.bstride, sstride = swap ? a.bstride : a.rstride;
    int32_t *const out_tpay = swap ? a.out_ppay : a.out_bpay, *const out_spay = swap ? a.out_bpay : a.out_ppay;
    const uint32_t ntr = blist ? it.nb : 1u;
    uint32_t tr = 0;
    uint64_t tb, te;
    auto table_range = [&](uint32_t j) {
        if (blist) { const uint32_t r = (uint32_t)it.b0 + j * tstride; tb = uniform64(tbeg[r]); te = uniform64(tend[r]); }
        else { tb = it.b0; te = it.b0 + it.nb; }
    };
    table_range(0);
    if (GEN) while (tb == te && tr + 1 < ntr) table_range(++tr);

    const uint64_t b0_ = it.b0, nb_ = it.nb;
    for (uint64_t bc = 0; GEN ? tb < te : bc < nb_; bc += a.cap) { // one table chunk per iteration: the next cap tuples of the table side
        bool built = false;
        uint32_t par = 0;
        const uint64_t gb0 = GEN ? tb : b0_ + bc;
        const uint32_t nbc0 = GEN ? 0u : (uint32_t)(nb_ - bc < a.cap ? nb_ - bc : a.cap);
        for (uint32_t rr = 0; rr < nr; rr++) { // list items: whole ranges, one after the other, against the same table
        uint64_t q0, q1;
        if (LISTS || GEN) { // the range cursor is wave-uniform: SGPRs, as in k_join (28 B/lane of scratch otherwise)
            if (it.p & JOIN_ITEM_LIST) { const uint32_t r = (uint32_t)it.q0 + rr * sstride; q0 = sbeg[r]; q1 = send[r]; }
            else { q0 = it.q0; q1 = it.q1; }
            q0 = uniform64(q0); q1 = uniform64(q1);
        } else { q0 = it.q0; q1 = it.q1; }
        for (uint64_t s0 = q0 & ~(uint64_t)3; s0 < q1; s0 += (uint64_t)SUB) {
            // the sub-chunk's probe tuples: issued first, so that they fly while the table is built
            // the first 2048 tuples are requested before the table is built (they fly during the build); the rest behind it — the
            // build keeps four 16-byte loads of its own in flight and the register file is what limits the workgroups per CU
            int4 kk[IT], pp[IT];
            auto tuple_index = [&](int t, int e) -> uint64_t { // group t < 2: 4 tuples per lane; group 2: 2 tuples per lane
                return t < 2 ? s0 + (uint64_t)t * JOIN_THREADS * 4 + (uint64_t)tid * 4 + e : s0 + (uint64_t)2 * JOIN_THREADS * 4 + (uint64_t)tid * 2 + e;
            };
            auto fetch = [&](int t) {
                const uint64_t i = tuple_index(t, 0);
                kk[t] = make_int4(0, 0, 0, 0); pp[t] = make_int4(0, 0, 0, 0);
                if (i >= q1) return;
                if (t < 2) { kk[t] = load4(sk, i, s_nalloc); pp[t] = load4(sp, i, s_nalloc); }
                else { // 8-byte group (i is even; the second element may lie beyond the allocation)
                    kk[t].x = sk[i]; pp[t].x = sp[i];
                    if (i + 1 < s_nalloc) { kk[t].y = sk[i + 1]; pp[t].y = sp[i + 1]; }
                }
            };
            fetch(0);
            if (built) {
#pragma unroll
                for (int t = 1; t < IT; t++) fetch(t);
            }
            if (!built) {
                for (uint32_t i = tid; i < a.nh; i += JOIN_THREADS) head[i] = 0xFFFFFFFFu;
                if (tid == 0) red[18] = 0; // "some key occurs twice in this table" (set by the check behind the build)
                __syncthreads();
                uint32_t filled = 0;
                for (;;) { // the pieces of this table chunk (one piece unless the table side is a list)
                    const uint64_t gb = GEN ? tb : gb0;
                    const uint32_t nbc = GEN ? (uint32_t)(te - tb < a.cap - filled ? te - tb : a.cap - filled) : nbc0;
                    for (uint64_t i0 = (gb & ~(uint64_t)3) + (uint64_t)tid * 4; i0 < gb + nbc; i0 += (uint64_t)JOIN_THREADS * 4 * 2) {
                        int4 bkv[2], bpv[2]; // two loads per column in flight (the probe tuples are live in registers already)
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            const uint64_t i = i0 + (uint64_t)r * JOIN_THREADS * 4;
                            if (i < gb + nbc) { bkv[r] = load4(tk, i, t_nalloc); bpv[r] = load4(tp, i, t_nalloc); }
                        }
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            const uint64_t i = i0 + (uint64_t)r * JOIN_THREADS * 4;
                            if (i < gb + nbc) {
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const uint64_t idx = i + e;
                                    if (idx >= gb && idx < gb + nbc) {
                                        const uint32_t slot = (GEN ? filled : 0u) + (uint32_t)(idx - gb), key = (uint32_t)elem(bkv[r], e);
                                        const uint32_t old = atomicExch(&head[hidx(key)], slot);
                                        if (TAG16) ent[slot] = make_uint2(((key >> tsh) << 16) | (old & 0xFFFFu), (uint32_t)elem(bpv[r], e));
                                        else { ent[slot] = make_uint2(key, (uint32_t)elem(bpv[r], e)); lnext[slot] = (uint16_t)old; }
                                    }
                                }
                            }
                        }
                    }
                    if (!GEN) break;
                    // the table cursor is wave-uniform, but it moves inside nested loops: pinned to SGPRs by hand
                    filled = (uint32_t)__builtin_amdgcn_readfirstlane((int)(filled + nbc)); tb = uniform64(tb + nbc);
                    while (tb == te && tr + 1 < ntr) { tr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tr + 1)); table_range(tr); }
                    if (tb == te || filled == a.cap) break;
                }
                __syncthreads();
                built = true;
                // Unique keys in the table = at most one match per streamed tuple = the round that would find nothing (a chain walk, a
                // reduction and a barrier per sub-chunk) need not run.  Whether they are unique costs about one such round to find
                // out (every entry looks down the rest of its chain for its own key), so it is asked only where the table serves
                // many sub-chunks: list items, general items, chunks of a long streamed side (config 4: 13 sub-chunks per table).
                if (LISTS || GEN || it.q1 - it.q0 > 2 * (uint64_t)SUB) {
                    const uint32_t nfill = GEN ? filled : nbc0;
                    bool dup = false;
                    for (uint32_t sl = tid; sl < nfill; sl += JOIN_THREADS) {
                        const uint2 me = ent[sl];
                        uint32_t s_ = TAG16 ? (me.x & 0xFFFFu) : (uint32_t)lnext[sl];
                        while (s_ != END) {
                            const uint2 en = ent[s_];
                            if (TAG16 ? ((en.x >> 16) == (me.x >> 16)) : (en.x == me.x)) { dup = true; break; }
                            s_ = TAG16 ? (en.x & 0xFFFFu) : (uint32_t)lnext[s_];
                        }
                    }
                    if (dup) red[18] = 1; // (read behind the first barrier of the rounds)
                } else if (tid == 0) red[18] = 1; // not asked: as if
#pragma unroll
                for (int t = 1; t < IT; t++) fetch(t);
            }
            // chain position of tuple j (END: exhausted): 16 bits each, two per register; mm bit j: tuple j sits ON a match that is
            // not written yet.  (Packed: the register file, not LDS, is what limits this kernel to three workgroups per CU.)
            uint32_t sp2[IT * 2], mm = 0;
#pragma unroll
            for (int z = 0; z < IT * 2; z++) sp2[z] = 0xFFFFFFFFu;
            auto getp = [&](int j) -> uint32_t { return (j & 1) ? sp2[j >> 1] >> 16 : sp2[j >> 1] & 0xFFFFu; };
            auto setp = [&](int j, uint32_t v) { sp2[j >> 1] = (j & 1) ? ((sp2[j >> 1] & 0xFFFFu) | (v << 16)) : ((sp2[j >> 1] & 0xFFFF0000u) | v); };
#pragma unroll
            for (int t = 0; t < IT; t++)
#pragma unroll
                for (int e = 0; e < MR_NE(t); e++) {
                    const uint64_t idx = tuple_index(t, e);
                    const bool valid = idx >= q0 && idx < q1;
                    setp(t * 4 + e, (valid ? head[hidx((uint32_t)elem(kk[t], e))] : 0xFFFFFFFFu) & END);
                }
            for (;; par ^= 1u) { // rounds
                // every tuple advances to its next match; the chains of a load group in lockstep
#pragma unroll
                for (int t = 0; t < IT; t++) {
                    for (;;) {
                        bool walking = false;
#pragma unroll
                        for (int e = 0; e < MR_NE(t); e++) {
                            const int j = t * 4 + e;
                            const uint32_t s_ = getp(j);
                            if (s_ != END && !((mm >> j) & 1u)) {
                                const uint32_t key = (uint32_t)elem(kk[t], e);
                                const uint2 en = ent[s_];
                                const bool eq = TAG16 ? ((en.x >> 16) == (key >> tsh)) : (en.x == key);
                                if (eq) mm |= 1u << j;
                                else {
                                    const uint32_t nx = TAG16 ? (en.x & 0xFFFFu) : (uint32_t)lnext[s_];
                                    setp(j, nx);
                                    walking |= nx != END;
                                }
                            }
                        }
                        if (!walking) break;
                    }
                }
                // wave total -> LDS; every thread then knows its wave's offset and the workgroup's total
                uint32_t wtot = (uint32_t)__popc(mm);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) wtot += __shfl_xor(wtot, o, 64);
                if (ln == 0) red[par * JOIN_WAVES + wave] = wtot;
                __syncthreads();
                uint32_t T = 0, wbase = 0;
#pragma unroll
                for (uint32_t w = 0; w < (uint32_t)JOIN_WAVES; w++) { const uint32_t v = red[par * JOIN_WAVES + w]; wbase += w < wave ? v : 0u; T += v; }
                if (!T) break; // workgroup-uniform: nobody found anything this round
                const bool unique_table = red[18] == 0;
                if (tid == 0) {
                    const unsigned long long base = atomicAdd(a.out_cursor, (unsigned long long)T); // ONE reservation per round
                    red[16] = (uint32_t)base; red[17] = (uint32_t)(base >> 32);
                }
                __syncthreads();
                uint64_t o = ((uint64_t)red[16] | ((uint64_t)red[17] << 32)) + wbase;
#pragma unroll
                for (int t = 0; t < IT; t++)
#pragma unroll
                    for (int e = 0; e < MR_NE(t); e++) {
                        const int j = t * 4 + e;
                        const bool m = (mm >> j) & 1u;
                        const uint64_t mask = __ballot(m);
                        if (m) {
                            const uint32_t slot = getp(j);
                            const uint2 en = ent[slot];
                            const uint64_t at = o + (uint64_t)__popcll(mask & lt_mask);
                            if (at < a.out_cap) {
                                a.out_key[at] = elem(kk[t], e);
                                out_tpay[at] = (int32_t)en.y;
                                out_spay[at] = elem(pp[t], e);
                            }
                            setp(j, TAG16 ? (en.x & 0xFFFFu) : (uint32_t)lnext[slot]); // the next round starts behind the match
                        }
                        o += (uint64_t)__popcll(mask);
                    }
                mm = 0;
                // red[16..17] are rewritten only behind the next round's first barrier; the totals alternate by parity
                if (unique_table) break; // nothing further down any chain: the next round would find nothing
            }
            par ^= 1u;
        }
        }
        if (GEN && !built) break; // nothing to probe in any range of the item: no table was built, the cursor did not move
        __syncthreads(); // the table is rebuilt (next build chunk): every wave must be through with it
    }
}

template <bool TAG16, bool LISTS, bool GEN = false> // LISTS: the items may be list items (sampled probe side); GEN: general items (see k_join)
__global__ __launch_bounds__(JOIN_THREADS, 6) void k_join_mat_reg(JoinArgs a) { // 6 waves per SIMD = three workgroups per CU: <= 85 VGPRs
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t item = blockIdx.x;
    if (item >= *a.n_items) return;
    const JoinItem it = a.items[item];
    if (GEN && (it.p & JOIN_ITEM_SWAP)) join_mat_reg_item<TAG16, LISTS, GEN, GEN>(a, it, smem); // (GEN as SWAP: no flipped instance without GEN)
    else join_mat_reg_item<TAG16, LISTS, GEN, false>(a, it, smem);
}

// count-only result: sums of the per-wave match counts and aggregates into out[0], out[1] (zeroed by k_join_plan)
// extra (optional): two more words added in by one thread — what the heavy-hitter bypass of pass 1 counted (hj_part.hip, HOT 1)
__global__ __launch_bounds__(256) void k_sum2(const uint64_t *__restrict__ cnt, const uint64_t *__restrict__ agg,
                                              const uint32_t *__restrict__ len_ptr, uint64_t mul, unsigned long long *__restrict__ out,
                                              const uint64_t *__restrict__ extra) {
    const uint64_t L = (uint64_t)(*len_ptr) * mul;
    uint64_t s = 0, t = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (uint64_t)gridDim.x * blockDim.x) { s += cnt[i]; t += agg[i]; }
    s = wave_sum64(s);
    t = wave_sum64(t);
    // one pair of atomics per WORKGROUP: a single word takes ~88 returning atomics per microsecond, and 4096 of them (one pair
    // per wave of a 512-workgroup grid) cost this kernel ~50 us whatever the input size
    __shared__ uint64_t red[2][4];
    if (lane_id() == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = t; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        t = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (extra && blockIdx.x == 0) { s += extra[0]; t += extra[1]; }
        if (s) atomicAdd(out, (unsigned long long)s);
        if (t) atomicAdd(out + 1, (unsigned long long)t);
    }
}

// sum of a[i] * b[i] (int32 x int32 -> int64, mod 2^64) over the first min(*n_ptr, cap) elements, added to *out: the aggregate
// (sum payR * payS) of a materialised output whose length is known on the device only (streaming materialising probe)
__global__ __launch_bounds__(256) void k_dot(const int32_t *__restrict__ a, const int32_t *__restrict__ b, const unsigned long long *__restrict__ n_ptr,
                                             uint64_t cap, unsigned long long *__restrict__ out) {
    const uint64_t n = *n_ptr < cap ? *n_ptr : cap;
    uint64_t s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        s += (uint64_t)((int64_t)a[i] * (int64_t)b[i]);
    s = wave_sum64(s);
    if (lane_id() == 0 && s) atomicAdd(out, (unsigned long long)s);
}
hipError_t launch_dot(hipStream_t st, const int32_t *a, const int32_t *b, const uint64_t *n_ptr, uint64_t cap, uint64_t *out) {
    hipLaunchKernelGGL(k_dot, dim3(1024), dim3(256), 0, st, a, b, reinterpret_cast<const unsigned long long *>(n_ptr), cap, reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (host)
// ------------------------------------------------------------------------------------------------

hipError_t launch_join_plan(hipStream_t st, const JoinArgs &a, uint32_t nparts, uint32_t *items_cnt, uint64_t *zero2, uint64_t *zero_cursor) {
    if (a.general) {
        hipLaunchKernelGGL(k_join_plan_gen, dim3((nparts + 255) / 256), dim3(256), 0, st, a, nparts, items_cnt, zero2, zero_cursor);
        HJ_LAUNCH_CHECK();
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_join_plan, dim3((nparts + 255) / 256), dim3(256), 0, st, a.bbeg, a.bend, a.pbeg, a.pend, nparts, a.chunk, items_cnt,
                       a.bflag, a.pflag, zero2, zero_cursor, a.rpart, a.pr0, a.pnr, a.rstride);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_join_expand(hipStream_t st, const JoinArgs &a, uint32_t nparts, const uint32_t *items_scanned,
                              const uint64_t *chunk_prefix, JoinItem *items) {
    if (a.general) {
        hipLaunchKernelGGL(k_join_expand_gen, dim3((nparts + 255) / 256), dim3(256), 0, st, a, nparts, items_scanned, chunk_prefix, items);
        HJ_LAUNCH_CHECK();
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_join_expand, dim3((nparts + 255) / 256), dim3(256), 0, st, a.bbeg, a.bend, a.pbeg, a.pend, nparts, a.chunk,
                       items_scanned, chunk_prefix, items, a.bflag, a.pflag, a.rpart, a.pr0, a.pnr, a.rstride);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

size_t join_lds_bytes(uint32_t nh, uint32_t cap, bool tag16) {
    size_t b = (size_t)nh * 4 + (size_t)cap * 4 + (size_t)cap * (tag16 ? 2 : 4) + (size_t)cap * 2;
    return (b + 15) & ~(size_t)15;
}

hipError_t join_set_lds_limit(int device, size_t bytes) {
    static size_t limit[64] = {};
    std::lock_guard<std::mutex> lock(g_attr_mutex);
    if (device >= 0 && device < 64 && bytes <= limit[device]) return hipSuccess;
    const void *fns[] = {reinterpret_cast<const void *>(&k_join<true, 0>), reinterpret_cast<const void *>(&k_join<true, 2>),
                         reinterpret_cast<const void *>(&k_join<false, 0>), reinterpret_cast<const void *>(&k_join<false, 2>),
                         reinterpret_cast<const void *>(&k_join<true, 0, true>), reinterpret_cast<const void *>(&k_join<false, 0, true>)};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    if (device >= 0 && device < 64) limit[device] = bytes;
    return hipSuccess;
}

// LDS of the one-probe materialising kernel: the table + the per-round wave totals and the reserved output base
size_t join_mat_lds_bytes(uint32_t nh, uint32_t cap, bool tag16) {
    const size_t tbl = ((size_t)nh * 4 + (size_t)cap * 8 + (tag16 ? 0 : (size_t)cap * 2) + 15) & ~(size_t)15;
    return tbl + 128;
}

hipError_t launch_join_mat_reg(hipStream_t st, const JoinArgs &a, uint32_t max_items, bool tag16) {
    static size_t limit[64] = {};
    const size_t lds = join_mat_lds_bytes(a.nh, a.cap, tag16);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(g_attr_mutex);
        if (dev < 0 || dev >= 64 || lds > limit[dev]) {
            const void *fns[] = {reinterpret_cast<const void *>(&k_join_mat_reg<true, false>), reinterpret_cast<const void *>(&k_join_mat_reg<false, false>),
                                 reinterpret_cast<const void *>(&k_join_mat_reg<true, true>), reinterpret_cast<const void *>(&k_join_mat_reg<false, true>),
                                 reinterpret_cast<const void *>(&k_join_mat_reg<true, true, true>), reinterpret_cast<const void *>(&k_join_mat_reg<false, true, true>)};
            for (const void *f : fns) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
            }
            if (dev >= 0 && dev < 64) limit[dev] = lds;
        }
    }
    dim3 g(max_items ? max_items : 1), b(JOIN_THREADS);
    if (a.general) {
        if (tag16) hipLaunchKernelGGL((k_join_mat_reg<true, true, true>), g, b, lds, st, a);
        else hipLaunchKernelGGL((k_join_mat_reg<false, true, true>), g, b, lds, st, a);
    } else if (a.pr0) {
        if (tag16) hipLaunchKernelGGL((k_join_mat_reg<true, true>), g, b, lds, st, a);
        else hipLaunchKernelGGL((k_join_mat_reg<false, true>), g, b, lds, st, a);
    } else {
        if (tag16) hipLaunchKernelGGL((k_join_mat_reg<true, false>), g, b, lds, st, a);
        else hipLaunchKernelGGL((k_join_mat_reg<false, false>), g, b, lds, st, a);
    }
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_join(hipStream_t st, const JoinArgs &a, uint32_t max_items, bool tag16, int jm) {
    size_t lds = join_lds_bytes(a.nh, a.cap, tag16);
    dim3 g(max_items ? max_items : 1), b(JOIN_THREADS);
    if (a.general) {
        if (jm != 0) return hipErrorInvalidValue; // general items: count and one-probe materialisation only (the host sees to it)
        if (tag16) hipLaunchKernelGGL((k_join<true, 0, true>), g, b, lds, st, a);
        else hipLaunchKernelGGL((k_join<false, 0, true>), g, b, lds, st, a);
        HJ_LAUNCH_CHECK();
        return hipSuccess;
    }
    if (jm != 0 && jm != 2) return hipErrorInvalidValue;
    if (tag16) {
        if (jm == 0) hipLaunchKernelGGL((k_join<true, 0>), g, b, lds, st, a);
        else hipLaunchKernelGGL((k_join<true, 2>), g, b, lds, st, a);
    } else {
        if (jm == 0) hipLaunchKernelGGL((k_join<false, 0>), g, b, lds, st, a);
        else hipLaunchKernelGGL((k_join<false, 2>), g, b, lds, st, a);
    }
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_sum2(hipStream_t st, const uint64_t *cnt, const uint64_t *agg, const uint32_t *len_ptr, uint64_t mul, uint64_t *out2, const uint64_t *extra2) {
    hipLaunchKernelGGL(k_sum2, dim3(256), dim3(256), 0, st, cnt, agg, len_ptr, mul, reinterpret_cast<unsigned long long *>(out2), extra2);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace hj
