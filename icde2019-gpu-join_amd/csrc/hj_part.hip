// hj_part.hip — gfx950 (MI355X, CDNA4) device code of the radix-partitioned hash join: the PARTITION passes (the join kernels: hj_join.hip).
//
// What the reference does on this path (all GPU kernels live in src/join-primitives.cu):
//   partition_pass_one/two   jp.cu:58-283, 338-535   fused histogram + slot reservation + LDS tile
//                                                    reorder + key/payload scatter into chained
//                                                    4096-tuple buckets (atomic bump allocation)
//   compute_bucket_info      jp.cu:294-312           chain walk between the passes
//   decompose_chains         jp.cu:843-874           probe-side work decomposition (<= 8192 tuples)
//   join_partitioned_aggregate / _results  jp.cu:885-1095, 1107-1416   LDS chained hash build+probe
//
// What this file does instead (MI355X-first):
//   * a partitioned relation is two columns + per-partition ranges [beg, end).  Two ways to get there:
//       - histogram-free passes (k_part1_fast, k_part2_fast; default for two-pass partitioning): every workgroup owns
//         a span of the input (pass 1) or one pass-1 digit (pass 2) and writes each digit into a fixed-capacity slot
//         of its own — the reference's bump-allocated fixed-size buckets (jp.cu:138-192) with one owner per bucket, so
//         no histogram, no scan, no global atomic.  A slot that would overflow (skew) raises a device flag;
//       - exact passes (k_plan, k_hist, k_scan_*, k_offsets, k_scatter_wc): keys-only histogram, device-side scan,
//         scatter to gap-free positions.  The host queues them for a relation whose flag came back raised (the flag
//         travels with the join's result block; the join kernels do nothing on flagged partitions).
//     All three scatter kernels share one round machinery (wc_fast): a 1024-thread workgroup (wave64) appends
//     (key,payload) pairs to per-digit 128-byte LDS write-combining lines and flushes only whole, 128-byte-aligned
//     lines with 16-byte stores, at any fan-out from 2 to 512 (the multi-GPU shard split uses the same kernel).
//     Nothing is read back to the host between kernels; grids are launched at their upper bound.
//   * the join kernels build a chained hash table in LDS per partition (16-bit tags at >= 16 radix bits, full keys
//     otherwise) and probe it with coalesced 16-byte loads; work items split the streamed side (<= probe_chunk tuples);
//     k_join counts per wave; k_join_mat_reg materialises (key,payR,payS) in the same probe, one exact reservation on
//     an output cursor per round, every tuple kept (no FOLD ring: jp.cu:1097-1101 D6); general items (GEN): tables
//     built from range lists, the smaller partition builds (role flip, jp.cu:929-1003).
//   All arithmetic is integer; there is no MFMA-shaped work on this path.
#include "hj_device.h"

namespace hj {

// ------------------------------------------------------------------------------------------------
// partition pass: plan → histogram → scan → offsets → scatter
// ------------------------------------------------------------------------------------------------

// Input of a pass = a list of SEGMENTS (contiguous tuple ranges [sbeg[i], send[i]) of the input columns);
// `spp` consecutive segments form one PARENT partition.  A single-GPU pass over contiguous partitions has
// one segment per parent (sbeg = offsets, send = offsets + 1); after the multi-GPU exchange a parent is the
// run of that pass-1 digit received from each peer (spp = number of GPUs) — the reference feeds its pass 1
// a segment list the same way (jp.cu:84-99,112-117).
// One thread per segment: number of spans (<= span tuples each) it is cut into, and the exclusive prefix
// of that.  span_start has nseg+1 entries.  Single workgroup, any nseg (chunks of 1024 with a carry).
__global__ __launch_bounds__(1024) void k_plan(const uint64_t *__restrict__ sbeg, const uint64_t *__restrict__ send,
                                               uint32_t nseg, uint32_t span, uint32_t *__restrict__ span_start) {
    __shared__ uint32_t scratch[17];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nseg; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        uint32_t c = 0;
        if (i < nseg) {
            uint64_t cnt = send[i] - sbeg[i];
            c = (uint32_t)((cnt + span - 1) / span);
        }
        uint32_t total;
        uint32_t ex = block_excl_scan<uint32_t>(c, scratch, &total);
        if (i < nseg) span_start[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) span_start[nseg] = carry;
}

struct SpanInfo {
    uint32_t parent, s, nsp, first; // parent id, span index inside it, spans of the parent, span_start[parent]
    uint64_t lo, hi;                // tuple range of this span
};

// Decode blockIdx.x into a span.  Every thread runs the same search (wave-uniform scalar loads).
__device__ __forceinline__ bool decode_span(const uint64_t *__restrict__ sbeg, const uint64_t *__restrict__ send,
                                            uint32_t nseg, uint32_t spp, const uint32_t *__restrict__ span_start,
                                            uint32_t span, SpanInfo &si) {
    const uint32_t b = blockIdx.x;
    if (b >= span_start[nseg]) return false;
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (span_start[mid] <= b) lo = mid; else hi = mid;
    }
    // segments with zero spans share span_start with their successor: take the last one <= b
    const uint32_t seg = lo;
    si.parent = seg / spp;
    si.first = span_start[si.parent * spp];
    si.nsp = span_start[(si.parent + 1) * spp] - si.first;
    si.s = b - si.first;
    uint64_t p0 = sbeg[seg], p1 = send[seg];
    si.lo = p0 + (uint64_t)(b - span_start[seg]) * span;
    si.hi = si.lo + span < p1 ? si.lo + span : p1;
    return true;
}

// Histogram of one span: hist[(first*P) + d*nsp + s].  4 B/tuple read, nothing else.
template <int MODE>
__global__ __launch_bounds__(PART_THREADS) void k_hist(const int32_t *__restrict__ keys, uint64_t nalloc,
                                                       const uint64_t *__restrict__ sbeg, const uint64_t *__restrict__ send,
                                                       uint32_t nseg, uint32_t spp,
                                                       const uint32_t *__restrict__ span_start, uint32_t span,
                                                       uint32_t shift, uint32_t P, uint32_t mask_or_n,
                                                       uint32_t *__restrict__ hist, const uint32_t *__restrict__ remap) {
    __shared__ uint32_t h[MAX_PARTS];
    SpanInfo si;
    if (!decode_span(sbeg, send, nseg, spp, span_start, span, si)) return;
    for (uint32_t d = threadIdx.x; d < P; d += PART_THREADS) h[d] = 0;
    __syncthreads();
    const uint64_t a0 = si.lo & ~(uint64_t)3;
    // wave-uniform trip count (rank_in_digit needs the whole wave)
    for (uint64_t w0 = a0 + (uint64_t)(threadIdx.x & ~63u) * 4; w0 < si.hi; w0 += (uint64_t)PART_THREADS * 4) {
        const uint64_t i = w0 + (uint64_t)lane_id() * 4;
        int4 v = (i < si.hi) ? load4(keys, i, nalloc) : make_int4(0, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            uint64_t idx = i + e;
            const bool valid = idx >= si.lo && idx < si.hi;
            (void)rank_in_digit(h, digit_of<MODE>((uint32_t)elem(v, e), shift, mask_or_n, remap), valid, P <= 2);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < P; d += PART_THREADS)
        hist[(uint64_t)si.first * P + (uint64_t)d * si.nsp + si.s] = h[d];
}

// ---- scan over a device-sized array: data[i] becomes the exclusive prefix inside its 4096-entry
// chunk, chunk_sums[c] the chunk total; k_scan_top turns the sums into exclusive chunk prefixes.
// The value of entry i is then data[i] + chunk_prefix[i >> 12].  L = (*len_ptr) * mul, or mul. ----
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_local(T *__restrict__ data, const uint32_t *__restrict__ len_ptr,
                                                             uint64_t mul, uint64_t *__restrict__ chunk_sums,
                                                             uint64_t *__restrict__ single, uint64_t *__restrict__ total_out) {
    __shared__ T scratch[17];
    const uint64_t L = len_ptr ? (uint64_t)(*len_ptr) * mul : mul;
    const uint64_t start = (uint64_t)blockIdx.x * SCAN_CHUNK;
    if (start >= L) {
        if (single && threadIdx.x == 0) { single[0] = 0; single[1] = 0; if (total_out) *total_out = 0; } // empty scan
        return;
    }
    const uint64_t i0 = start + (uint64_t)threadIdx.x * SCAN_PER;
    T v[SCAN_PER];
    T sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER; j++) {
        v[j] = (i0 + j < L) ? data[i0 + j] : (T)0;
        sum += v[j];
    }
    T total;
    T ex = block_excl_scan<T>(sum, scratch, &total);
#pragma unroll
    for (int j = 0; j < SCAN_PER; j++) {
        if (i0 + j < L) data[i0 + j] = ex;
        ex += v[j];
    }
    if (threadIdx.x == 0) {
        chunk_sums[blockIdx.x] = (uint64_t)total;
        // a scan that fits one chunk needs no second kernel (launch_scan_*: one launch instead of two, small inputs)
        if (single) { single[0] = 0; single[1] = (uint64_t)total; if (total_out) *total_out = (uint64_t)total; }
    }
}

// Single workgroup: exclusive scan of the chunk sums; chunk_prefix[nchunks] and *total_out = total.
__global__ __launch_bounds__(1024) void k_scan_top(const uint64_t *__restrict__ chunk_sums, const uint32_t *__restrict__ len_ptr,
                                                   uint64_t mul, uint64_t *__restrict__ chunk_prefix,
                                                   uint64_t *__restrict__ total_out) {
    __shared__ uint64_t scratch[17];
    const uint64_t L = len_ptr ? (uint64_t)(*len_ptr) * mul : mul;
    const uint64_t nchunks = (L + SCAN_CHUNK - 1) / SCAN_CHUNK;
    uint64_t carry = 0;
    for (uint64_t base = 0; base < nchunks; base += 1024) {
        uint64_t i = base + threadIdx.x;
        uint64_t v = i < nchunks ? chunk_sums[i] : 0;
        uint64_t total;
        uint64_t ex = block_excl_scan<uint64_t>(v, scratch, &total);
        if (i < nchunks) chunk_prefix[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        chunk_prefix[nchunks] = carry;
        if (total_out) *total_out = carry;
    }
}

// Child partition offsets: coff[parent*P + d] = output position of (parent, d, span 0); coff[last] = n.
// The same values go to beg[]/end[] (partition p = [beg[p], end[p])), the form the join reads.
__global__ void k_offsets(const uint32_t *__restrict__ hist, const uint64_t *__restrict__ chunk_prefix,
                          const uint32_t *__restrict__ span_start, uint32_t nparents, uint32_t spp, uint32_t P,
                          uint64_t n, uint64_t *__restrict__ coff, uint64_t *__restrict__ beg, uint64_t *__restrict__ end) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nchild = (uint64_t)nparents * P;
    if (t > nchild) return;
    uint64_t v = n;
    if (t < nchild) {
        const uint32_t parent = (uint32_t)(t / P), d = (uint32_t)(t % P);
        const uint64_t total = (uint64_t)span_start[nparents * spp] * P;
        const uint32_t first = span_start[parent * spp], nsp = span_start[(parent + 1) * spp] - first;
        const uint64_t idx = (uint64_t)first * P + (uint64_t)d * nsp;
        v = idx < total ? (uint64_t)hist[idx] + chunk_prefix[idx >> SCAN_CHUNK_LOG] : n;
    }
    coff[t] = v;
    if (beg && t < nchild) beg[t] = v;
    if (end && t > 0) end[t - 1] = v;
}

// Write-combining scatter (k_scatter_wc, k_part1_fast, k_part2_fast): the software write-combining idea of the
// reference's CPU partitioner (partition-primitives.cu:40-125) re-expressed in LDS.  Every digit owns 128-byte
// lines in LDS (32 tuples); tuples are appended to their digit's line and a line leaves the CU only when it is
// full, as one aligned 128-byte store per column.  On MI355X a two-column stream copy runs at ~5.2 TB/s and the
// same copy with every 128-byte line scattered to a random aligned position at ~5.1 TB/s (hj_ubench), while
// unaligned 16-tuple runs reach 1-2.5 TB/s: whole aligned lines are the point.  The round machinery is wc_fast
// below; the exact (histogram-backed) kernel k_scatter_wc follows it.
constexpr int WC_THREADS = 1024;
constexpr int WC_LINE = 32;   // tuples per 128-byte line
constexpr int WC_HSTRIDE = MAX_PARTS + 64; // arrival counters per parity + 64 per-lane trash counters (branch-free ranking)

// ------------------------------------------------------------------------------------------------
// histogram-free ("optimistic") passes
// ------------------------------------------------------------------------------------------------
// The exact pass above pays a keys-only histogram kernel (4 B/tuple of extra HBM reads, 15 % of a 2^30 x 2^30
// step) for exact, gap-free output offsets.  The reference has no such pass: its partition kernels bump-allocate
// fixed-size buckets as they go (jp.cu:138-192).  The same idea, kept contention-free: every output SLOT has a
// fixed capacity, sized from the expected count plus 8 standard deviations, and belongs to exactly one
// workgroup, so there is no histogram, no scan and no global atomic:
//   pass 1: workgroup s owns one contiguous span of the input; its tuples of digit d go to slot (d, s)
//           = out[(d*nspans + s)*cap ...], filled front to back in whole 128-byte lines (slots are line-aligned);
//   pass 2: workgroup d owns pass-1 digit d = the nspans slots (d, *) and splits them again; child c goes to slot
//           (d, c) = out[(d*P + c)*cap ...] — the final partition.
// Each pass writes the ranges [beg, end) of its slots; the join reads partitions as ranges.  A slot that would
// overflow (skewed keys) raises *ovf and nothing is written past a slot; the join's planning kernel reads the flags
// and produces no work items, the host sees them with the result block and redoes that relation with the exact passes
// (hj_api.hip).  Uniform and near-uniform inputs never overflow.
// The per-round machinery (rank by one LDS atomic, per-digit write-combining lines, full lines leave as aligned
// 128-byte stores) is that of k_scatter_wc; all lines are aligned here, so there is no first-line masking.
constexpr int WF_TRASH_LINES = WC_THREADS / WC_LINE;   // one trash slot per thread (branch-free placement)
constexpr int WF_LINES = MAX_PARTS + WF_TRASH_LINES;
constexpr int WF_MAXSEG = 1024;                        // segments of one parent (= pass-1 workgroups)
constexpr uint32_t WF_NONE = 0xFFFFFFFFu;

struct WfLds {
    int2 *buf;                   // [WF_LINES][32] (key, payload) pairs: K = 512/P lines per digit + one trash slot per thread
    uint32_t *hh, *line;         // per digit and round parity: (line fill at round start << 16) | arrivals; output position of slot 0
    uint32_t *wlist;             // [16 waves][32] flush work lists
    uint32_t *pc4, *sb;          // pass 2: prefix of 4-tuple units per segment [nseg+1]; segment start | padding
    uint32_t *lo;                // exact pass: first valid slot of a digit's first line (aliases pc4: never both)
    uint32_t *lt, *own;          // exact pass under skew: lines << 16 | first line per digit; owner digit per line
    uint32_t *hk, *hot;          // heavy-hitter bypass (pass 1 only: aliases the segment tables): hk[HOT_SLOTS] (key, payload) pairs; hot[64] reduction words
};

// digit d's output slot has index slotA + d*slotB (that is where its range [beg, end) is written) and, with uniform capacities,
// lies at index*cap.  Variable capacities (the sampled path of skewed relations): the slot of digit d starts at
// voff + vbase[d] + vs*vcap[d] and holds vcap[d] tuples.
// Output positions inside the kernels are kept in LINES (32 tuples = 128 bytes per column): every slot starts on a line and holds
// whole lines, so 32-bit line numbers address 2^37 tuples — a relation of 2^32 tuples and more (288 GB of HBM, not 32-bit
// positions, is the size limit of one GPU; the reference's CLI accepts up to ULONG_MAX/4 tuples, main.cu:491-514) — and byte
// addresses are formed in 64 bits where a line is stored.
struct FastGeom {
    uint32_t slotA, slotB, cap;
    const uint32_t *vbase = nullptr, *vcap = nullptr;
    uint32_t voff = 0, vs = 0;
};
__device__ __forceinline__ uint32_t slot_line(const FastGeom &g, uint32_t d) { // first line of the slot
    return g.vbase ? (g.voff + g.vbase[d] + g.vs * g.vcap[d]) / WC_LINE : (g.slotA + d * g.slotB) * (g.cap / WC_LINE);
}
__device__ __forceinline__ uint32_t slot_lines(const FastGeom &g, uint32_t d) { return (g.vbase ? g.vcap[d] : g.cap) / WC_LINE; }

// One workgroup, one input stream, P digits with K = 512/P LDS lines each.  Per round of 8192 tuples:
//   A  kept tuples of the previous round open their digit's next line; digit owners advance the output position by
//      the lines flushed last round; every tuple takes its slot with ONE returning LDS atomic on a word that holds
//      (fill of the digit's lines at round start << 16 | arrivals this round), so the returned value IS the slot;
//   B  with the round's totals known (one LDS read per tuple): slots inside lines that leave this round, or in a
//      digit none of whose lines leaves, are stored now ((key,payload) as one 8-byte LDS write); slots past the
//      leaving lines are kept in registers for phase A; the digit owners seed the next round's counters;
//   C  every full line leaves: 8 lanes move one line (two 16-byte LDS reads, a 16-byte store to each column).
// All LDS traffic of a phase is issued back to back: operations that must not happen are pointed at per-thread
// trash slots / per-lane trash counters instead of being branched around.
// EXACT = true: the histogram-backed pass (k_scatter_wc): every digit's output run starts at an exact, arbitrarily
// aligned position (lo[d] = first valid slot of the run's first line), nothing can overflow, no slot ranges are
// written.  HEAVY: the span's histogram says one digit holds more than a quarter of it — ranks are taken with the
// wave-aggregated atomic (same-address LDS atomics serialise per lane).
// VAR (exact pass under skew): the 512 LDS lines are dealt to the digits in proportion to what the span's histogram
// says each will receive per round (L_.lt[d] = lines << 16 | first line, L_.own[line] = digit), instead of K each.
// MODE 1 (exact pass only): the digit is the multi-GPU shard of the key (digit_of<1>: hash, optional position table), P
// need not be a power of two (K = the largest power of two <= 512/P lines per digit).
// HOT (pass 1 of a relation known to be skewed; SRC 0): the heavy-hitter bypass.  L_.hk is a direct-mapped table of (key, payload the
// OTHER relation has for it) pairs; a tuple whose key sits in its slot is joined here and takes no slot of a partition.  HOT 1: counted
// (matches and payload products, one atomic pair per workgroup at the end).  HOT 2: written to the join's output columns — the hits of
// round r are ranked by ballots, the workgroup reserves their exact number on the output cursor with ONE returning atomic issued in
// phase B, and they are written at the start of round r + 1 (the atomic has phases B and C to return; the tuples are still in kk/pp).
// (HOT 3, round 6, removed: the hits as digit number P through the LDS lines, whole-line stores to the output, one reservation per round by
// the digit's owner — every hit pays phases B and C again: 8.8 ms against HOT 2's 8.35 on config 4, profiles/r6_hot_bypass.txt.)
template <int U, int KFIX, int SRC, bool EXACT = false, bool HEAVY = false, bool VAR = false, int MODE = 0, int HOT = 0>
__device__ __forceinline__ void wc_fast(const WfLds &L_, const int32_t *__restrict__ keys, const int32_t *__restrict__ pays,
                                        uint64_t lo64, uint64_t hi64, uint64_t nalloc, uint32_t nseg, uint32_t shift,
                                        uint32_t P, const FastGeom g, int32_t *__restrict__ out_keys,
                                        int32_t *__restrict__ out_pays, uint64_t *__restrict__ obeg,
                                        uint64_t *__restrict__ oend, uint32_t *__restrict__ ovf,
                                        const uint32_t *__restrict__ remap = nullptr, const HotArgs *hotp = nullptr) {
    static_assert(HOT == 0 || (SRC == 0 && !EXACT && MODE == 0), "the bypass belongs to pass 1 of the histogram-free passes");
    constexpr bool BALANCED = VAR && !EXACT; // the flush of phase C from ONE list per workgroup (the sampled passes)
    int2 *buf = L_.buf;
    uint32_t *hh = L_.hh, *line = L_.line;
    const uint32_t kshift = KFIX ? (uint32_t)__builtin_ctz((unsigned)KFIX) : 31u - (uint32_t)__builtin_clz((uint32_t)MAX_PARTS / P);
    const uint32_t K = 1u << kshift, capS = K * WC_LINE; // lines / slots per digit in LDS
    const uint32_t tid = threadIdx.x, wv = tid >> 6, ln = tid & 63u;
    const uint32_t mask = P - 1;
    constexpr uint32_t ROUND = WC_THREADS * 4 * U;
    // ---- input feeder ----
    // SRC 0: the contiguous tuples [lo64, hi64): thread t of round r loads the 16 bytes at a0 + r*ROUND + (u*1024+t)*4.
    // SRC 1: the parent's segments as one stream of 4-tuple units (segments are 16-byte aligned and padded to whole
    //        units); the stream is cut into 16 contiguous ranges, one per wave, so that a wave's loads are consecutive
    //        1-KiB pieces and a lane's segment cursor moves rarely.
    const uint64_t a0 = lo64 & ~(uint64_t)3;
    const uint32_t rlo = (uint32_t)(lo64 - a0), rhi = (uint32_t)(hi64 - a0);
    const int32_t *kin = keys + a0, *pin = pays + a0;
    const uint64_t navail = nalloc - a0;
    uint32_t nrounds, wbeg = 0, wend = 0, ci = 0, clo = 0, chi = 0, csb = 0;
    if (SRC == 0) {
        nrounds = (rhi + ROUND - 1) / ROUND;
    } else {
        const uint32_t T4 = L_.pc4[nseg];
        const uint32_t R = ((T4 + (WC_THREADS / 64) * 64 - 1) / ((WC_THREADS / 64) * 64)) * 64; // units per wave, multiple of 64
        nrounds = (R + 64 * U - 1) / (64 * U);
        wbeg = wv * R;
        wend = wbeg + R < T4 ? wbeg + R : T4;
        const uint32_t unit0 = wbeg + ln;
        if (unit0 < wend) { // segment of the lane's first unit: the last i with pc4[i] <= unit0
            uint32_t a = 0, b = nseg;
            while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (L_.pc4[m] <= unit0) a = m; else b = m; }
            ci = a; clo = L_.pc4[a]; chi = L_.pc4[a + 1]; csb = L_.sb[a];
        }
    }
    auto fetch = [&](uint32_t round, int u, int4 &kv, int4 &pv, uint32_t &vm) {
        kv = make_int4(0, 0, 0, 0); pv = make_int4(0, 0, 0, 0); vm = 0;
        if (SRC == 0) {
            const uint32_t r = round * ROUND + (u * WC_THREADS + tid) * 4;
            if (round < nrounds && r < rhi) {
                kv = load4(kin, r, navail);
                pv = load4(pin, r, navail);
#pragma unroll
                for (int e = 0; e < 4; e++) vm |= (r + e >= rlo && r + e < rhi) ? (1u << e) : 0u;
            }
        } else {
            const uint32_t unit = wbeg + (round * U + u) * 64 + ln;
            if (round < nrounds && unit < wend) {
                while (unit >= chi) { ci++; clo = chi; chi = L_.pc4[ci + 1]; csb = L_.sb[ci]; }
                const uint64_t addr = (uint64_t)(csb >> 2) * WC_LINE + (uint64_t)(unit - clo) * 4; // segments start on a line
                kv = *reinterpret_cast<const int4 *>(keys + addr);
                pv = *reinterpret_cast<const int4 *>(pays + addr);
                vm = (unit + 1 == chi) ? (0xFu >> (csb & 3u)) : 0xFu; // the segment's last unit may be padded
            }
        }
    };
    int4 kv[U], pv[U];
    uint32_t vm[U];
#pragma unroll
    for (int u = 0; u < U; u++) fetch(0, u, kv[u], pv[u], vm[u]);
    int4 kk[U], pp[U];     // the previous round's tuples: the kept ones are stored one phase later
    uint32_t keep[U * 4];  // LDS pair index of a kept tuple, WF_NONE = none
#pragma unroll
    for (int j = 0; j < U * 4; j++) keep[j] = WF_NONE;
#pragma unroll
    for (int u = 0; u < U; u++) { kk[u] = make_int4(0, 0, 0, 0); pp[u] = make_int4(0, 0, 0, 0); }
    // the digit this thread owns (tid < P): slot geometry
    // (in lines) the slot of the digit this thread owns; a slot counts as full one granule (the digit's LDS lines) early
    const uint32_t my_base = tid < P ? slot_line(g, tid) : 0u, my_lim = my_base + (tid < P ? slot_lines(g, tid) : 0u);
    const uint32_t my_gran = (VAR && tid < P) ? (L_.lt[tid] >> 16) : K;
    const uint32_t trash = MAX_PARTS * WC_LINE + tid;
    // ---- heavy-hitter bypass state ----
    const uint2 *hk2 = reinterpret_cast<const uint2 *>(L_.hk);
    uint32_t *hwtot = L_.hot, *hwpre = L_.hot + 16;                                   // HOT 2: hits per wave / their exclusive prefix
    unsigned long long *hbase = reinterpret_cast<unsigned long long *>(L_.hot + 32); // HOT 2: the round's reservation
    uint32_t hcnt = 0, hw = 0; // HOT 1: this WAVE's hits (wave-uniform) / hit bits of the round's 8 tuples
    uint64_t hagg = 0;
    unsigned long long hres = 0;
    // HOT 2: the hits of the round whose tuples are in (kk, pp) go out — every wave runs the same ballots
    auto hot_emit = [&]() {
        uint64_t wb = (uint64_t)(*hbase) + hwpre[wv];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int j = u * 4 + e;
                const bool hit = (hw >> j) & 1u;
                const uint64_t m = __ballot(hit);
                if (hit) {
                    const uint64_t pos = wb + (uint32_t)__popcll(m & (((uint64_t)1 << ln) - 1));
                    if (pos < hotp->out_cap) {
                        const uint32_t key = (uint32_t)elem(kk[u], e);
                        hotp->out_key[pos] = (int32_t)key;
                        hotp->out_str[pos] = elem(pp[u], e);
                        hotp->out_tab[pos] = (int32_t)hk2[hot_slot(key)].y;
                    }
                }
                wb += (uint32_t)__popcll(m);
            }
    };
#ifdef HJ_STAMPS
    unsigned long long stA = 0, stB = 0, stC = 0, stH = 0; // (wave-uniform: scalar registers)
#endif
    uint32_t par = 0;
    for (uint32_t round = 0; round < nrounds; round++, par ^= 1) {
        uint32_t *h = hh + par * WC_HSTRIDE, *hprev = hh + (par ^ 1) * WC_HSTRIDE;
#ifdef HJ_STAMPS
        const unsigned long long ts0 = HOT ? hj_now() : 0ull;
#endif
        // ---- A ----
        // HOT 2: the reservation issued in phase B of the last round is awaited HERE — vector memory operations return in order, so waiting for
        // the atomic also waits for the loads issued before it, i.e. for THIS round's prefetched tuples, which are needed now anyway; awaited
        // at the end of phase C (round 6's first version) it held the whole workgroup at the barrier until the next round's data had arrived:
        // 1.5 ms of config 4's pass 1 (profiles/r6_hot_bypass.txt)
        if (HOT == 2 && round) {
            if (tid == (uint32_t)(WC_THREADS / 64) - 1) *hbase = hres;
            __syncthreads();
#if !(defined(HJ_EXP) && HJ_EXP == 2)
            hot_emit();
#endif
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t k = keep[u * 4 + e];
                buf[k != WF_NONE ? k : trash] = make_int2(elem(kk[u], e), elem(pp[u], e));
            }
        // another workgroup gave up (a slot overflowed somewhere): stop moving data that will be thrown away.  One
        // thread polls the flag, the workgroup learns it through LDS behind the round's barriers (uniform exit).
        if (!EXACT && tid == 0 && (round & 3u) == 0) L_.wlist[(WC_THREADS / 64) * 32] = __hip_atomic_load(ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (BALANCED && tid == 0) L_.wlist[(WC_THREADS / 64) * 32 + 1] = 0; // the flush list of phase C is empty (read last behind the last barrier of the round before)
        if (tid < P) { // the lines flushed last round move this digit's output position
            const uint32_t w = hprev[tid];
            const uint32_t full = ((w >> 16) + (w & 0xFFFFu)) & ~(uint32_t)(WC_LINE - 1);
            if (full) {
                uint32_t nl = line[tid] + full / WC_LINE;
                if (!EXACT && nl + my_gran > my_lim) { *ovf = 1u | (tid << 8) | (round << 20); nl = my_base; } // slot full: give up (the exact passes redo it)
                line[tid] = nl;
                if (EXACT) L_.lo[tid] = 0; // only the run's first line starts mid-line
            }
        }
        // the kept tuples are stored: this round's tuples move to (kk, pp) — the ranks below and phase B work from there, phase A of
        // the next round stores the kept ones — and the NEXT round's loads are issued right away: they fly through the rest of
        // A, B and C (round 3; they used to be issued at the end of B and had only C to arrive)
        uint32_t vmc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            kk[u] = kv[u]; pp[u] = pv[u]; vmc[u] = vm[u];
            fetch(round + 1, u, kv[u], pv[u], vm[u]);
        }
        uint32_t code[U * 4]; // digit << 16 | slot in the digit's lines ; WF_NONE = not a tuple
        uint32_t htot = 0;    // HOT 2: the wave's hits this round (wave-uniform)
        if (HOT) { // one 8-byte LDS read per tuple: the (key, payload) pair of its slot; hw = hit bits.  (Looking the NEXT round's tuples up
                   // at the end of phase C instead — one LDS round trip less in phase A — measured 2.5 % slower.)
            hw = 0;
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int j = u * 4 + e;
                    const uint32_t key = (uint32_t)elem(kk[u], e);
                    const uint2 en = hk2[hot_slot(key)];
                    const bool hit = ((vmc[u] >> e) & 1u) && en.x == key;
                    hw |= hit ? (1u << j) : 0u;
                    if (HOT == 1) hagg += hit ? (uint64_t)((int64_t)(int32_t)en.y * (int64_t)elem(pp[u], e)) : (uint64_t)0;
                    if (HOT == 2) htot += (uint32_t)__popcll(__ballot(hit));
                }
            if (HOT == 1) { // the wave's hits, counted in scalar registers (the kernel has no vector register to spare)
#pragma unroll
                for (int j = 0; j < U * 4; j++) hcnt += (uint32_t)__popcll(__ballot((hw >> j) & 1u));
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool hit = HOT && ((hw >> (u * 4 + e)) & 1u);
                const bool valid = ((vmc[u] >> e) & 1u) && !hit;
                const uint32_t d = MODE == 0 ? (((uint32_t)elem(kk[u], e) >> shift) & mask) : digit_of<1>((uint32_t)elem(kk[u], e), 0, P, remap);
                // (HOT: more than a third of the tuples may be hits, which take no slot: their LDS operations are masked off, not pointed at trash)
                const uint32_t old = HEAVY ? rank_in_digit(h, d, valid, P <= 2)
                                   : HOT ? (valid ? atomicAdd(&h[d], 1u) : 0u)
                                           : atomicAdd(&h[valid ? d : (uint32_t)MAX_PARTS + ln], 1u); // invalid: a trash counter
                code[u * 4 + e] = valid ? ((d << 16) | ((old >> 16) + (old & 0xFFFFu))) : WF_NONE;
            }
        if (HOT == 2 && ln == 0) hwtot[wv] = htot;
        __syncthreads();
#ifdef HJ_STAMPS
        const unsigned long long ts1 = HOT ? hj_now() : 0ull;
#endif
        // ---- B ----
        if (HOT == 2 && tid < (uint32_t)(WC_THREADS / 64)) { // lanes 0..15 of wave 0: prefix of the waves' hits, one reservation for the workgroup
            const uint32_t mine = hwtot[tid];
            uint32_t v = mine;
#pragma unroll
            for (int o = 1; o < WC_THREADS / 64; o <<= 1) { const uint32_t t = __shfl_up(v, o, 64); if ((int)tid >= o) v += t; }
            hwpre[tid] = v - mine;
#if defined(HJ_EXP) && HJ_EXP == 1
            if (tid == (uint32_t)(WC_THREADS / 64) - 1) hres = 0ull;
#elif defined(HJ_EXP) && HJ_EXP == 5
            if (tid == (uint32_t)(WC_THREADS / 64) - 1) hres = v ? atomicAdd(reinterpret_cast<unsigned long long *>(hotp->out_key) + (size_t)blockIdx.x * 32, (unsigned long long)v) : 0ull; // a word of its own: no contention
#elif defined(HJ_EXP) && HJ_EXP == 6
            if (tid == (uint32_t)(WC_THREADS / 64) - 1 && v) __hip_atomic_fetch_add(hotp->cursor, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // same word, result unused
#else
            if (tid == (uint32_t)(WC_THREADS / 64) - 1) hres = v ? atomicAdd(hotp->cursor, (unsigned long long)v) : 0ull; // (the wave goes on: the value is awaited at the start of the next round)
#endif
        }
        const uint32_t stop = EXACT ? 0u : L_.wlist[(WC_THREADS / 64) * 32];
        uint32_t hw[U * 4];
#pragma unroll
        for (int j = 0; j < U * 4; j++) hw[j] = h[code[j] != WF_NONE ? code[j] >> 16 : 0u]; // all LDS reads first
        if (tid < P) { // next round's counter starts at the fill the digit's open line will have
            const uint32_t w = h[tid];
            hprev[tid] = (((w >> 16) + (w & 0xFFFFu)) & (uint32_t)(WC_LINE - 1)) << 16;
        }
        bool any_bypass = false;
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int j = u * 4 + e;
                const uint32_t c = code[j];
                const bool valid = c != WF_NONE;
                const uint32_t d = valid ? c >> 16 : 0u, q = c & 0xFFFFu;
                const uint32_t full = ((hw[j] >> 16) + (hw[j] & 0xFFFFu)) & ~(uint32_t)(WC_LINE - 1); // slots that leave this round
                const bool leaves = q < full;
                const uint32_t lt = VAR ? L_.lt[d] : 0u;
                const uint32_t capd = VAR ? (lt >> 16) * WC_LINE : capS, based = VAR ? (lt & 0xFFFFu) * WC_LINE : d * capS;
                const bool now = valid && (leaves ? q < capd : full == 0);
                any_bypass |= valid && leaves && q >= capd;
                if (!HOT) buf[now ? based + q : trash] = make_int2(elem(kk[u], e), elem(pp[u], e));
                else if (now) buf[based + q] = make_int2(elem(kk[u], e), elem(pp[u], e));
                keep[j] = (valid && !leaves && full != 0) ? based + (q - full) : WF_NONE;
            }
        if (any_bypass) { // rare: a digit received more than its K lines in one round; straight to HBM
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int j = u * 4 + e;
                    const uint32_t c = code[j];
                    if (c != WF_NONE) {
                        const uint32_t d = c >> 16, q = c & 0xFFFFu;
                        const uint32_t full = ((hw[j] >> 16) + (hw[j] & 0xFFFFu)) & ~(uint32_t)(WC_LINE - 1);
                        if (q < full && q >= (VAR ? (L_.lt[d] >> 16) * WC_LINE : capS)) {
                            const uint64_t o = (uint64_t)line[d] * WC_LINE + q;
                            if (EXACT || o < (uint64_t)(slot_line(g, d) + slot_lines(g, d)) * WC_LINE) { out_keys[o] = elem(kk[u], e); out_pays[o] = elem(pp[u], e); }
                            else *ovf = 2u;
                        }
                    }
                }
        }
        __syncthreads();
#ifdef HJ_STAMPS
        const unsigned long long ts2 = HOT ? hj_now() : 0ull;
#endif
        // ---- C: the wave owns 32 of the 512 LDS lines (line ls belongs to digit ls >> kshift); the full ones are
        //         compacted into a list and flushed 8 lanes per line ----
        {
            uint32_t *wlist = L_.wlist + wv * 32;
            const uint32_t lsl = wv * 32 + (ln & 31u);
            const uint32_t dq = VAR ? L_.own[lsl] : lsl >> kshift;          // VAR: 0xFFFF = a line nobody owns
            const uint32_t ltq = (VAR && dq < P) ? L_.lt[dq] : 0u;
            const uint32_t wq = dq < P ? h[dq] : 0u;
            uint32_t fullq_n = ((wq >> 16) + (wq & 0xFFFFu)) & ~(uint32_t)(WC_LINE - 1);
            const uint32_t capq = VAR ? (ltq >> 16) * WC_LINE : capS;
            fullq_n = (fullq_n < capq ? fullq_n : capq) >> 5; // full lines of that digit
            const bool fullq = (ln < 32u) && ((VAR ? lsl - (ltq & 0xFFFFu) : (lsl & (K - 1))) < fullq_n);
            const uint64_t m = __ballot(fullq);
            uint32_t nfull = (uint32_t)__popcll(m), t0 = 0, tstep = 8;
            if (BALANCED) { // lines dealt by need (VAR): the full lines of a round sit with the waves that own the heavy digits' lines — one
                            // list for the workgroup, every wave takes every 16th group of 8 lines (one more barrier; config 4's pass 1:
                            // 1.1 us of 6.5 per round were waves waiting for the slowest one's lines, profiles/r6_hot_bypass.txt)
                uint32_t *gn = L_.wlist + (WC_THREADS / 64) * 32 + 1;
                uint32_t gbase = 0;
                if (ln == 0 && nfull) gbase = atomicAdd(gn, nfull);
                gbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)gbase);
                if (fullq) L_.wlist[gbase + (uint32_t)__popcll(m & (((uint64_t)1 << ln) - 1))] = lsl;
                __syncthreads();
                nfull = *gn; t0 = wv * 8; tstep = (WC_THREADS / 64) * 8;
                wlist = L_.wlist;
            } else {
                if (fullq) wlist[__popcll(m & (((uint64_t)1 << ln) - 1))] = lsl;
                __builtin_amdgcn_wave_barrier(); // DS operations of one wave execute in order
            }
            const uint32_t c4 = (ln & 7u) * 4;
            for (uint32_t t = t0; t < nfull; t += tstep) {
                const uint32_t idx = t + (ln >> 3);
                if (idx < nfull) {
                    const uint32_t ls = wlist[idx];
                    const uint32_t df = VAR ? L_.own[ls] : ls >> kshift;
                    const uint32_t jf = VAR ? ls - (L_.lt[df] & 0xFFFFu) : (ls & (K - 1));
                    const uint64_t gpos = ((uint64_t)line[df] + jf) * WC_LINE + c4;
                    const int4 x = *reinterpret_cast<const int4 *>(buf + ls * WC_LINE + c4);     // k0 p0 k1 p1
                    const int4 y = *reinterpret_cast<const int4 *>(buf + ls * WC_LINE + c4 + 2); // k2 p2 k3 p3
                    const int4 kq = make_int4(x.x, x.z, y.x, y.z), pq = make_int4(x.y, x.w, y.y, y.w);
                    const uint32_t first_valid = (EXACT && jf == 0) ? L_.lo[df] : 0u;
                    if (first_valid == 0) {
                        *reinterpret_cast<int4 *>(out_keys + gpos) = kq;
                        *reinterpret_cast<int4 *>(out_pays + gpos) = pq;
                    } else { // the span's first line of this digit starts mid-line: element-wise
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (c4 + e >= first_valid) { out_keys[gpos + e] = elem(kq, e); out_pays[gpos + e] = elem(pq, e); }
                    }
                }
            }
        }
#ifdef HJ_STAMPS
        const unsigned long long ts3 = HOT ? hj_now() : 0ull; // (ts3 .. ts4: this wave waits for the other waves' lines)
#endif
        __syncthreads();
#ifdef HJ_STAMPS
        if (HOT) { const unsigned long long ts4 = hj_now(); stA += ts1 - ts0; stB += ts2 - ts1; stC += ts3 - ts2; stH += ts4 - ts3; }
#endif
        if (stop) return; // workgroup-uniform: every thread read the same LDS word between the same barriers
    }
#ifdef HJ_STAMPS
    if (HOT && hotp->stamps && tid == 0) { unsigned long long *sp_ = hotp->stamps + (size_t)blockIdx.x * 8; sp_[0] = stA; sp_[1] = stB; sp_[2] = stC; sp_[3] = stH; sp_[4] = nrounds; }
#endif
    if (HOT == 2 && nrounds) { // the last round's hits
        if (tid == (uint32_t)(WC_THREADS / 64) - 1) *hbase = hres;
        __syncthreads();
        hot_emit();
    }
    if (HOT == 1) { // one atomic pair per workgroup
        const uint64_t wc = (uint64_t)hcnt, wa = wave_sum64(hagg); // (hcnt is the WAVE's count already)
        unsigned long long *red = reinterpret_cast<unsigned long long *>(L_.hot);
        if (ln == 0) { red[wv] = wc; red[16 + wv] = wa; }
        __syncthreads();
        if (tid == 0) {
            uint64_t sc = 0, sa = 0;
            for (int w = 0; w < WC_THREADS / 64; w++) { sc += red[w]; sa += red[16 + w]; }
            if (sc) atomicAdd(hotp->acc, (unsigned long long)sc);
            if (sa) atomicAdd(hotp->acc + 1, (unsigned long long)sa);
        }
    }
    // ---- epilogue: phase A of the last round, the partially filled last line of every digit, the slot ranges ----
    uint32_t *hlast = hh + (par ^ 1) * WC_HSTRIDE;
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (keep[u * 4 + e] != WF_NONE) buf[keep[u * 4 + e]] = make_int2(elem(kk[u], e), elem(pp[u], e));
    if (tid < P) {
        const uint32_t w = hlast[tid];
        const uint32_t full = ((w >> 16) + (w & 0xFFFFu)) & ~(uint32_t)(WC_LINE - 1);
        if (full) {
            uint32_t nl = line[tid] + full / WC_LINE;
            if (!EXACT && nl + my_gran > my_lim) { *ovf = 3u | (tid << 8) | (blockIdx.x << 16); nl = my_base; }
            line[tid] = nl;
            if (EXACT) L_.lo[tid] = 0;
        }
    }
    __syncthreads();
    for (uint32_t d = wv; d < P; d += WC_THREADS / 64) {
        const uint32_t s = ln & (WC_LINE - 1);
        const uint32_t w = hlast[d];
        const uint32_t cur = ((w >> 16) + (w & 0xFFFFu)) & (uint32_t)(WC_LINE - 1);
        if (s < cur && (!EXACT || s >= L_.lo[d])) {
            const int2 v = buf[(VAR ? (L_.lt[d] & 0xFFFFu) * WC_LINE : d * capS) + s];
            if (ln < (uint32_t)WC_LINE) out_keys[(uint64_t)line[d] * WC_LINE + s] = v.x;
            else out_pays[(uint64_t)line[d] * WC_LINE + s] = v.y;
        }
    }
    if (!EXACT && tid < P) {
        const uint32_t slot = g.slotA + tid * g.slotB;
        const uint32_t w = hlast[tid];
        obeg[slot] = (uint64_t)my_base * WC_LINE;
        oend[slot] = (uint64_t)line[tid] * WC_LINE + (((w >> 16) + (w & 0xFFFFu)) & (uint32_t)(WC_LINE - 1));
    }
}

__device__ __forceinline__ void wf_carve(WfLds &L_, unsigned char *smem) {
    L_.buf = reinterpret_cast<int2 *>(smem);
    L_.hh = reinterpret_cast<uint32_t *>(L_.buf + WF_LINES * WC_LINE);
    L_.line = L_.hh + 2 * WC_HSTRIDE;
    L_.wlist = L_.line + MAX_PARTS;
    L_.pc4 = L_.wlist + (WC_THREADS / 64) * 32 + 4; // + the "stop" word
    L_.sb = L_.pc4 + WF_MAXSEG + 4;
    L_.lo = L_.pc4;
    L_.lt = L_.sb + WF_MAXSEG;       // lines dealt by need: an area of their own (the segment tables of pass 2 are in use then)
    L_.own = L_.lt + MAX_PARTS;
    L_.hk = L_.pc4;                  // 2 * HOT_SLOTS words of the WF_MAXSEG + 4 + WF_MAXSEG the two segment tables hold (16-byte aligned)
    L_.hot = L_.own + MAX_PARTS;
}
static_assert(2 * HOT_SLOTS <= 2 * WF_MAXSEG + 4, "the hot-key table lives in the segment tables of pass 2");
// the candidate table of the heavy-hitter bypass as a workgroup uses it: (key, payload) pairs; a candidate counts only while the other
// relation holds it exactly once (cnt == 1, k_hot_build of this step); the others become fillers, which no key of their slot equals
__device__ __forceinline__ void hot_load(uint32_t *hk, const uint32_t *__restrict__ cand, const uint32_t *__restrict__ cnt, const int32_t *__restrict__ pay) {
    for (uint32_t i = threadIdx.x; i < HOT_SLOTS; i += blockDim.x) {
        hk[2 * i] = cnt[i] == 1u ? cand[i] : hot_filler(i);
        hk[2 * i + 1] = (uint32_t)pay[i];
    }
}
size_t fast_lds_bytes_impl() {
    return (size_t)WF_LINES * WC_LINE * 8 + (size_t)WC_HSTRIDE * 4 * 2 + (size_t)MAX_PARTS * 4 +
           ((WC_THREADS / 64) * 32 + 4) * 4 + (size_t)(WF_MAXSEG + 4) * 4 + (size_t)WF_MAXSEG * 4 + (size_t)MAX_PARTS * 4 * 2 + 64 * 4;
}

// The exact pass: scatter one span to the positions the histogram + scan assigned.  The span's private output run of
// digit d starts at g0 (any alignment); the LDS lines of d mirror the 128-byte output lines being filled.
// Algorithmic traffic: 8 B read + 8 B written per tuple.
template <int U, int MODE>
__global__ __launch_bounds__(WC_THREADS) void k_scatter_wc(const int32_t *__restrict__ keys, const int32_t *__restrict__ pays,
                                                           uint64_t nalloc, const uint64_t *__restrict__ sbeg,
                                                           const uint64_t *__restrict__ send, uint32_t nseg, uint32_t spp,
                                                           const uint32_t *__restrict__ span_start,
                                                           uint32_t span, uint32_t shift, uint32_t P,
                                                           const uint32_t *__restrict__ hist,
                                                           const uint64_t *__restrict__ chunk_prefix,
                                                           int32_t *__restrict__ out_keys, int32_t *__restrict__ out_pays,
                                                           uint64_t n_out, const uint32_t *__restrict__ remap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WfLds L_;
    wf_carve(L_, smem);
    SpanInfo si;
    if (!decode_span(sbeg, send, nseg, spp, span_start, span, si)) return;
    const uint32_t tid = threadIdx.x;
    uint32_t *scratch = L_.hh; // 17 words, before hh is zeroed
    const uint64_t L = (uint64_t)span_start[nseg] * P;
    uint32_t cnt = 0, lo0 = 0;
    if (tid < P) {
        const uint64_t idx = (uint64_t)si.first * P + (uint64_t)tid * si.nsp + si.s;
        const uint64_t g0 = (uint64_t)hist[idx] + chunk_prefix[idx >> SCAN_CHUNK_LOG];
        const uint64_t g1 = idx + 1 < L ? (uint64_t)hist[idx + 1] + chunk_prefix[(idx + 1) >> SCAN_CHUNK_LOG] : n_out;
        cnt = (uint32_t)(g1 - g0);
        L_.line[tid] = (uint32_t)(g0 / WC_LINE); // in lines (wc_fast); the run's first line starts at slot lo0 of it
        lo0 = (uint32_t)g0 & (WC_LINE - 1);
        L_.lo[tid] = lo0;
    }
    // One workgroup reduction tells every thread three things about the span's histogram:
    //  - does one digit hold more than a quarter of the span (same-address LDS atomics would serialise per lane: HEAVY);
    //  - how many lines the digits want in total, a digit wanting its expected arrivals per round with 30 % headroom;
    //  - does some digit want more than the K = 512/P lines every digit has by default (then the lines are dealt by need).
    constexpr uint32_t ROUND = WC_THREADS * 4 * U;
    const uint32_t K = (uint32_t)MAX_PARTS / P;
    const uint64_t len = si.hi - si.lo;
    uint32_t need = 0;
    uint64_t packed = 0;
    if (tid < P) {
        need = (uint32_t)(((uint64_t)cnt * ROUND * 13 / 10) / (len ? len : 1)) / WC_LINE + 1;
        if (need > (uint32_t)MAX_PARTS) need = MAX_PARTS;
        packed = (uint64_t)need | ((uint64_t)((uint64_t)cnt * 4 > len ? 1 : 0) << 32) | ((uint64_t)(need > K ? 1 : 0) << 48);
    }
    uint64_t tot64;
    (void)block_excl_scan<uint64_t>(packed, reinterpret_cast<uint64_t *>(scratch), &tot64);
    const uint32_t total_need = (uint32_t)tot64, any_heavy = (uint32_t)(tot64 >> 32) & 0xFFFFu, any_over = (uint32_t)(tot64 >> 48);
    const bool var = any_over != 0 && P < (uint32_t)MAX_PARTS; // 512 digits leave no line to deal
    if (var) {
        uint32_t nl = need;
        if (total_need > (uint32_t)MAX_PARTS) // not enough lines: one each, the rest in proportion to the wish beyond one
            nl = 1 + (uint32_t)((uint64_t)(need - (tid < P ? 1u : 0u)) * ((uint32_t)MAX_PARTS - P) / (total_need - P));
        if (tid >= P) nl = 0;
        uint32_t dummy;
        const uint32_t first = block_excl_scan<uint32_t>(nl, scratch, &dummy);
        if (tid < (uint32_t)MAX_PARTS) L_.own[tid] = 0xFFFFu;
        __syncthreads();
        if (tid < P) {
            L_.lt[tid] = (nl << 16) | first;
            for (uint32_t j = 0; j < nl; j++) L_.own[first + j] = tid;
        }
    }
    __syncthreads();
    for (uint32_t d = tid; d < 2 * WC_HSTRIDE; d += WC_THREADS) L_.hh[d] = 0;
    __syncthreads();
    if (tid < P) L_.hh[tid] = lo0 << 16; // round 0 starts at the fill the run's first line already has
    __syncthreads();
    FastGeom g{};
    g.slotA = 0; g.slotB = 0; g.cap = 0;
    // block_excl_scan hands the workgroup totals to every thread: the branches are workgroup-uniform
#define HJ_WC(KF, HV, VR) wc_fast<U, KF, 0, true, HV, VR, MODE>(L_, keys, pays, si.lo, si.hi, nalloc, 0, shift, P, g, out_keys, out_pays, nullptr, nullptr, nullptr, remap)
    if (var) { if (any_heavy) HJ_WC(0, true, true); else HJ_WC(0, false, true); }
    else if (any_heavy) HJ_WC(0, true, false);
    else if (MODE == 0 && P == (uint32_t)MAX_PARTS) HJ_WC(1, false, false);
    else HJ_WC(0, false, false);
#undef HJ_WC
}

// pass 1: one workgroup per span of the contiguous input.  MODE 1: the digit is the multi-GPU shard of the key (hash, any
// fan-out 1..512) — the histogram-free level-0 split of the sliced exchange (hj_dist.hip): shard g's slots (g, *) form one
// contiguous region of fixed size, which is what travels to GPU g.  FEW: at most 4 digits — ranks are taken with the
// wave-aggregated atomic (nearly every lane would otherwise queue on one of a few LDS words).
template <int U, int MODE, bool FEW>
__device__ __forceinline__ void part1_fast_body(const FastArgs &a, const uint32_t s, unsigned char *smem) {
    WfLds L_;
    wf_carve(L_, smem);
    if (__hip_atomic_load(a.ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return; // an earlier workgroup gave up already
    const uint32_t tid = threadIdx.x;
    const uint64_t lo = (uint64_t)s * a.span < a.n ? (uint64_t)s * a.span : a.n; // a span past the end (short slice of a multi-GPU split): empty
    const uint64_t hi = lo + a.span < a.n ? lo + a.span : a.n;
    FastGeom g{s, a.nspans, a.cap};
    for (uint32_t d = tid; d < 2 * WC_HSTRIDE; d += WC_THREADS) L_.hh[d] = 0;
    if (tid < a.P) L_.line[tid] = slot_line(g, tid);
    if (tid == 0) L_.wlist[(WC_THREADS / 64) * 32] = 0;
    __syncthreads();
    if (MODE == 0 && a.P == (uint32_t)MAX_PARTS) wc_fast<U, 1, 0, false, false, false, 0>(L_, a.keys, a.pays, lo, hi, a.n, 0, a.shift, a.P, g, a.out_keys, a.out_pays, a.obeg, a.oend, a.ovf);
    else wc_fast<U, 0, 0, false, FEW, false, MODE>(L_, a.keys, a.pays, lo, hi, a.n, 0, a.shift, a.P, g, a.out_keys, a.out_pays, a.obeg, a.oend, a.ovf);
}
template <int U, int MODE, bool FEW>
__global__ __launch_bounds__(WC_THREADS) void k_part1_fast(FastArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    part1_fast_body<U, MODE, FEW>(a, blockIdx.x, smem);
}
// Both relations of a join in ONE launch per pass (hj_api.hip partition_both, small and medium inputs): workgroups [0, a.nspans) are
// relation a's spans, the rest relation b's.  A step is then three data-moving launches on one stream instead of five on two
// (the reference's shape: launch after launch on one stream, jp.cu:1582-1613), and the second relation's workgroups fill the CUs
// the first one's tail leaves idle without an event fork and join.
template <int U>
__global__ __launch_bounds__(WC_THREADS) void k_part1_fast2(FastArgs a, FastArgs b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x < a.nspans) part1_fast_body<U, 0, false>(a, blockIdx.x, smem);
    else part1_fast_body<U, 0, false>(b, blockIdx.x - a.nspans, smem);
}

// ---- the sampled path of skewed relations: histogram-free passes with per-digit slot capacities and LDS lines dealt by
// need, both derived (on the host, once per binding) from a joint sample histogram of the final partition ids.  The
// reference partitions any distribution in one launch per pass (bump-allocated buckets, jp.cu:138-192); here a skewed
// relation keeps the one-launch passes of the uniform case, with slots sized to what the sample says each digit receives.
// A slot that still overflows raises the flag: the exact passes are the fallback. ----
// More than 15 radix bits: 2^bits counters no longer fit a workgroup's LDS (128 KiB = 2^15), so the id space is cut into 2^(bits-15)
// SLICES and a workgroup counts the ids of ONE slice only; workgroups of the same slice share the sample between them, every slice
// reads the whole sample (keys only, 1/stride of the relation: 2-8 x ~1 GiB for 2^31 tuples at 16-18 bits, once per binding).
constexpr uint32_t SAMPLE_LDS_BITS = 15;
// hot_cand / hot_cnt (optional): the heavy-hitter bypass will take the tuples of these keys out of the relation in pass 1 — they are
// counted as sampled (the shares stay fractions of the WHOLE relation) but enter no bin
__global__ __launch_bounds__(1024) void k_sample_joint(const int32_t *__restrict__ keys, uint64_t n, uint32_t mask, uint32_t stride, uint32_t slice_bits,
                                                       uint32_t *__restrict__ hist, unsigned long long *__restrict__ sampled,
                                                       const uint32_t *__restrict__ hot_cand, const uint32_t *__restrict__ hot_cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    const uint32_t slice = blockIdx.x & ((1u << slice_bits) - 1), wg = blockIdx.x >> slice_bits, nwg = gridDim.x >> slice_bits;
    const uint32_t lmask = slice_bits ? (1u << SAMPLE_LDS_BITS) - 1 : mask; // the bins of one workgroup
    for (uint32_t i = threadIdx.x; i <= lmask; i += 1024) h[i] = 0;
    const uint2 *hk2 = reinterpret_cast<const uint2 *>(h + lmask + 1); // behind the bins
    if (hot_cand) hot_load(h + lmask + 1, hot_cand, hot_cnt, reinterpret_cast<const int32_t *>(hot_cand));
    __syncthreads();
    // blocks of 4096 tuples, every stride-th one; the heavy key would serialise a wave's LDS atomics: aggregated rank
    uint64_t cnt = 0;
    for (uint64_t b = (uint64_t)wg * stride; b * 4096 < n; b += (uint64_t)nwg * stride) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t i = b * 4096 + (uint64_t)u * 1024 + threadIdx.x;
            const bool valid = i < n;
            const uint32_t key = valid ? (uint32_t)keys[i] : 0u;
            const uint32_t id = key & mask;
            bool cold = true;
            if (hot_cand) cold = hk2[hot_slot(key)].x != key;
            const bool mine = valid && cold && (id >> SAMPLE_LDS_BITS) == (slice_bits ? slice : id >> SAMPLE_LDS_BITS);
            (void)rank_in_digit(h, mine ? (id & lmask) : 0u, mine);
            cnt += valid;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= lmask; i += 1024)
        if (h[i]) atomicAdd(&hist[slice_bits ? ((slice << SAMPLE_LDS_BITS) | i) : i], h[i]);
    cnt = wave_sum64(cnt);
    if (slice == 0 && lane_id() == 0 && cnt) atomicAdd(sampled, (unsigned long long)cnt); // (every slice sees the whole sample: counted once)
}

hipError_t launch_sample_joint(hipStream_t st, const int32_t *keys, uint64_t n, uint32_t bits, uint32_t stride, uint32_t *hist, uint64_t *sampled,
                               const uint32_t *hot_cand, const uint32_t *hot_cnt) {
    static bool set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (bits > SAMPLE_LDS_BITS + 3) return hipErrorInvalidValue;
    const uint32_t slice_bits = bits > SAMPLE_LDS_BITS ? bits - SAMPLE_LDS_BITS : 0u;
    const size_t lds = ((size_t)1 << (slice_bits ? SAMPLE_LDS_BITS : bits)) * 4 + (hot_cand ? (size_t)2 * HOT_SLOTS * 4 : 0);
    {
        std::lock_guard<std::mutex> lock(g_attr_mutex);
        if (dev < 0 || dev >= 64 || !set[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sample_joint), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024 + 2 * HOT_SLOTS * 4);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) set[dev] = true;
        }
    }
    hipLaunchKernelGGL(k_sample_joint, dim3(256), dim3(1024), lds, st, keys, n, (1u << bits) - 1, stride, slice_bits, hist, reinterpret_cast<unsigned long long *>(sampled), hot_cand, hot_cnt);
    return hipGetLastError();
}

template <int U, bool HEAVY, int HOT>
__global__ __launch_bounds__(WC_THREADS) void k_part1_var(FastArgs a, VarArgs v) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WfLds L_;
    wf_carve(L_, smem);
    if (__hip_atomic_load(a.ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const uint32_t tid = threadIdx.x, s = blockIdx.x;
    const uint64_t lo = (uint64_t)s * a.span < a.n ? (uint64_t)s * a.span : a.n;
    const uint64_t hi = lo + a.span < a.n ? lo + a.span : a.n;
    FastGeom g{s, a.nspans, 0};
    g.vbase = v.vbase; g.vcap = v.vcap; g.voff = 0; g.vs = s;
    for (uint32_t d = tid; d < 2 * WC_HSTRIDE; d += WC_THREADS) L_.hh[d] = 0;
    if (tid < a.P) { L_.line[tid] = slot_line(g, tid); L_.lt[tid] = v.lt[tid]; }
    if (tid < (uint32_t)MAX_PARTS) L_.own[tid] = v.own[tid];
    if (tid == 0) L_.wlist[(WC_THREADS / 64) * 32] = 0;
    if (HOT) hot_load(L_.hk, v.hot.cand, v.hot.cnt, v.hot.pay);
    __syncthreads();
    // 512 digits: every digit has exactly one of the 512 lines, there is nothing to deal — the fixed-geometry rounds (no per-digit
    // line table in the inner loops) with the sampled slot capacities (a 512-way pass under skew: 9.8 -> see r4_sampled_16_17_bits.txt)
    if (a.P == (uint32_t)MAX_PARTS) wc_fast<U, 1, 0, false, HEAVY, false, 0, HOT>(L_, a.keys, a.pays, lo, hi, a.n, 0, a.shift, a.P, g, a.out_keys, a.out_pays, a.obeg, a.oend, a.ovf, nullptr, &v.hot);
    else wc_fast<U, 0, 0, false, HEAVY, true, 0, HOT>(L_, a.keys, a.pays, lo, hi, a.n, 0, a.shift, a.P, g, a.out_keys, a.out_pays, a.obeg, a.oend, a.ovf, nullptr, &v.hot);
}

// ---- a look at a large relation BEFORE its first optimistic attempt (round 6) ----
// nsamp keys, one per stratum of n / nsamp tuples at a hashed offset inside it (a fixed stride would alias with sorted or strided keys), counted by
// pass-1 digit (hist[0 .. P1)) and by pass-2 digit (hist[P1 .. P1 + P2)): the host reads the P1 + P2 counters and sends a relation with a digit far
// above its share straight to the sampled path — no failed attempt, no partition buffers allocated for it.  P1 + P2 <= 1024; hist zero at launch.
__global__ __launch_bounds__(1024) void k_skew_probe(const int32_t *__restrict__ keys, uint64_t n, uint32_t nsamp, uint32_t b1, uint32_t b2, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[1024];
    const uint32_t tid = threadIdx.x, i = blockIdx.x * 1024 + tid, P1 = 1u << b1, P2 = 1u << b2;
    h[tid] = 0;
    __syncthreads();
    if (i < nsamp) {
        const uint64_t stride = n / nsamp;
        const uint64_t pos = (uint64_t)i * stride + (uint64_t)fmix32(i * 0x9E3779B1u + 0x7F4A7C15u) % stride;
        const uint32_t key = (uint32_t)keys[pos];
        atomicAdd(&h[(key >> b2) & (P1 - 1)], 1u);
        atomicAdd(&h[P1 + (key & (P2 - 1))], 1u);
    }
    __syncthreads();
    if (tid < P1 + P2 && h[tid]) atomicAdd(&hist[tid], h[tid]);
}
hipError_t launch_skew_probe(hipStream_t st, const int32_t *keys, uint64_t n, uint32_t nsamp, uint32_t b1, uint32_t b2, uint32_t *hist) {
    hipLaunchKernelGGL(k_skew_probe, dim3((nsamp + 1023) / 1024), dim3(1024), 0, st, keys, n, nsamp, b1, b2, hist);
    return hipGetLastError();
}

// ---- the heavy-hitter bypass: finding the candidates (once per binding) and what the other relation holds for them (every step) ----
// k_hot_sample: nsamp keys of the relation (runs of 16, evenly spread) counted in an open-addressing table in HBM (slots >= 2 * nsamp)
__global__ __launch_bounds__(256) void k_hot_sample(const int32_t *__restrict__ keys, uint64_t n, uint32_t nsamp, uint32_t *__restrict__ tkey,
                                                    uint32_t *__restrict__ tcnt, uint32_t mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsamp) return;
    const uint64_t stride = n / (nsamp >> 4); // >= 16: the runs do not overlap
    const uint64_t pos = (uint64_t)(i >> 4) * stride + (i & 15u);
    if (pos >= n) return;
    const uint32_t key = (uint32_t)keys[pos];
    if ((int32_t)key == HOT_NEVER) return; // the table's "empty"
    uint32_t slot = fmix32(key) & mask;
    for (;;) {
        const uint32_t prev = atomicCAS(&tkey[slot], (uint32_t)HOT_NEVER, key);
        if (prev == (uint32_t)HOT_NEVER || prev == key) { atomicAdd(&tcnt[slot], 1u); return; }
        slot = (slot + 1) & mask;
    }
}
// the keys sampled at least thr times, as (key, count) pairs in no particular order; *nout counts all of them (it may exceed cap)
__global__ __launch_bounds__(256) void k_hot_collect(const uint32_t *__restrict__ tkey, const uint32_t *__restrict__ tcnt, uint32_t slots, uint32_t thr,
                                                     uint2 *__restrict__ out, uint32_t *__restrict__ nout, uint32_t cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    const uint32_t c = tcnt[i];
    if (c >= thr) {
        const uint32_t at = atomicAdd(nout, 1u);
        if (at < cap) out[at] = make_uint2(tkey[i], c);
    }
}
// every step: cnt[slot] = tuples of the scanned relation whose key is candidate `slot`, pay[slot] = the payload of one of them.
// 4 B per tuple read (keys only; the payload of a hit is fetched on its own).  cnt must be zero at launch; *zero_acc (two words) is
// zeroed for the pass that follows on the same stream.
__global__ __launch_bounds__(256) void k_hot_build(const int32_t *__restrict__ keys, const int32_t *__restrict__ pays, uint64_t n,
                                                   const uint32_t *__restrict__ cand, uint32_t *__restrict__ cnt, int32_t *__restrict__ pay,
                                                   unsigned long long *__restrict__ zero_acc) {
    __shared__ uint32_t hk[HOT_SLOTS];
    for (uint32_t i = threadIdx.x; i < HOT_SLOTS; i += blockDim.x) hk[i] = cand[i];
    if (zero_acc && blockIdx.x == 0 && threadIdx.x < 2) zero_acc[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (uint64_t)gridDim.x * blockDim.x * 4) {
        const int4 kv = load4(keys, i, n);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (i + e >= n) break;
            const uint32_t key = (uint32_t)elem(kv, e);
            const uint32_t slot = hot_slot(key);
            if (hk[slot] == key) {
                atomicAdd(&cnt[slot], 1u);
                pay[slot] = pays[i + e];
            }
        }
    }
}
hipError_t launch_hot_sample(hipStream_t st, const int32_t *keys, uint64_t n, uint32_t nsamp, uint32_t *tkey, uint32_t *tcnt, uint32_t slots) {
    hipLaunchKernelGGL(k_hot_sample, dim3((nsamp + 255) / 256), dim3(256), 0, st, keys, n, nsamp, tkey, tcnt, slots - 1);
    return hipGetLastError();
}
hipError_t launch_hot_collect(hipStream_t st, const uint32_t *tkey, const uint32_t *tcnt, uint32_t slots, uint32_t thr, uint2 *out, uint32_t *nout, uint32_t cap) {
    hipLaunchKernelGGL(k_hot_collect, dim3((slots + 255) / 256), dim3(256), 0, st, tkey, tcnt, slots, thr, out, nout, cap);
    return hipGetLastError();
}
hipError_t launch_hot_build(hipStream_t st, const int32_t *keys, const int32_t *pays, uint64_t n, const uint32_t *cand, uint32_t *cnt, int32_t *pay, unsigned long long *zero_acc) {
    const uint64_t want = (n / 4 + 255) / 256;
    const uint32_t blocks = (uint32_t)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(k_hot_build, dim3(blocks), dim3(256), 0, st, keys, pays, n, cand, cnt, pay, zero_acc);
    return hipGetLastError();
}

template <int U>
__global__ __launch_bounds__(WC_THREADS) void k_part2_var(FastArgs a, VarArgs v) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (*a.ovf) return;
    const uint4 w = v.wg[blockIdx.x]; // {parent, first span, spans, output position}
    WfLds L_;
    wf_carve(L_, smem);
    const uint32_t tid = threadIdx.x, d = w.x;
    FastGeom g{blockIdx.x * a.P, 1u, 0};
    g.vbase = v.vbase + (uint64_t)d * a.P; g.vcap = v.vcap + (uint64_t)d * a.P; g.voff = w.w; g.vs = 0;
    uint32_t *scratch = L_.hh;
    uint32_t units = 0;
    if (tid < w.z) {
        const uint64_t b = a.sbeg[(uint64_t)d * a.spp + w.y + tid], e = a.send[(uint64_t)d * a.spp + w.y + tid];
        const uint32_t cnt = (uint32_t)(e - b);
        units = (cnt + 3) >> 2;
        L_.sb[tid] = ((uint32_t)(b / WC_LINE) << 2) | (units * 4 - cnt);
    }
    uint32_t total;
    const uint32_t ex = block_excl_scan<uint32_t>(units, scratch, &total);
    if (tid < w.z) L_.pc4[tid] = ex;
    if (tid == 0) L_.pc4[w.z] = total;
    __syncthreads();
    for (uint32_t i = tid; i < 2 * WC_HSTRIDE; i += WC_THREADS) L_.hh[i] = 0;
    if (tid < a.P) { L_.line[tid] = slot_line(g, tid); L_.lt[tid] = v.lt[(uint64_t)d * a.P + tid]; }
    if (tid < (uint32_t)MAX_PARTS) L_.own[tid] = v.own[(uint64_t)d * MAX_PARTS + tid];
    if (tid == 0) L_.wlist[(WC_THREADS / 64) * 32] = 0;
    __syncthreads();
    // workgroup-uniform: parents dominated by one child rank with the wave-aggregated atomic (as k_scatter_wc does per span)
    if (a.P == (uint32_t)MAX_PARTS) { // one line per child: the fixed-geometry rounds (see k_part1_var)
        if (v.heavy[d]) wc_fast<U, 1, 1, false, true, false, 0>(L_, a.keys, a.pays, 0, 0, 0, w.z, a.shift, a.P, g, a.out_keys, a.out_pays, a.obeg, a.oend, a.ovf);
        else wc_fast<U, 1, 1, false, false, false, 0>(L_, a.keys, a.pays, 0, 0, 0, w.z, a.shift, a.P, g, a.out_keys, a.out_pays, a.obeg, a.oend, a.ovf);
        return;
    }
    if (v.heavy[d]) wc_fast<U, 0, 1, false, true, true, 0>(L_, a.keys, a.pays, 0, 0, 0, w.z, a.shift, a.P, g, a.out_keys, a.out_pays, a.obeg, a.oend, a.ovf);
    else wc_fast<U, 0, 1, false, false, true, 0>(L_, a.keys, a.pays, 0, 0, 0, w.z, a.shift, a.P, g, a.out_keys, a.out_pays, a.obeg, a.oend, a.ovf);
}

// multi-GPU: segment table of one received slice.  Peer q's split wrote its slots (me, s) into a region of nsp slots of cap
// tuples, which arrived at keys[base + q*nsp*cap ...]; oend[q*nsp + s] = end position of that slot in the SENDER's buffer,
// whose slot (me, s) started at (me*nsp + s)*cap.  Workgroup s of the local pass 1 reads the G segments (*, s):
// sbeg/send[s*G + q].  A fill beyond cap (a corrupted message) raises the flag.
// me == 0xFFFFFFFF (phantom world, a one-GPU measurement mode): region q holds this rank's OWN shard q, i.e. sender slot (q, s).
__global__ void k_dist_segments(const uint64_t *__restrict__ oend, uint32_t G, uint32_t nsp, uint32_t cap, uint32_t me, uint64_t base,
                                uint64_t *__restrict__ sbeg, uint64_t *__restrict__ send, uint32_t *__restrict__ flag,
                                unsigned long long *__restrict__ received, unsigned long long *__restrict__ received_self) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * nsp) return;
    const uint32_t q = t / nsp, sidx = t % nsp;
    const uint64_t sender_beg = ((uint64_t)(me == 0xFFFFFFFFu ? q : me) * nsp + sidx) * cap;
    const uint64_t e = oend[t];
    uint64_t fill = e >= sender_beg ? e - sender_beg : 0;
    if (e < sender_beg || fill > cap) { *flag = 1u; fill = 0; }
    const uint64_t b = base + ((uint64_t)q * nsp + sidx) * cap;
    sbeg[(uint64_t)sidx * G + q] = b;
    send[(uint64_t)sidx * G + q] = b + fill;
    if (fill) atomicAdd(received, (unsigned long long)fill);
    if (fill && received_self && q == (me == 0xFFFFFFFFu ? 0u : me)) atomicAdd(received_self, (unsigned long long)fill); // tuples that never crossed a link
}

hipError_t launch_dist_segments(hipStream_t st, const uint64_t *oend, uint32_t G, uint32_t nsp, uint32_t cap, uint32_t me, uint64_t base,
                                uint64_t *sbeg, uint64_t *send, uint32_t *flag, uint64_t *received, uint64_t *received_self) {
    hipLaunchKernelGGL(k_dist_segments, dim3((G * nsp + 255) / 256), dim3(256), 0, st, oend, G, nsp, cap, me, base, sbeg, send, flag,
                       reinterpret_cast<unsigned long long *>(received), reinterpret_cast<unsigned long long *>(received_self));
    return hipGetLastError();
}

// pass 2: one workgroup per parent = the spp input segments [sbeg, send) of that parent
template <int U>
__device__ __forceinline__ void part2_fast_body(const FastArgs &a, const uint32_t parent, unsigned char *smem) {
    // (the join's planning kernel reserves item slots on a counter: zeroed here, by the kernel that always runs just before it)
    if (a.zero_items && parent == 0 && threadIdx.x == 0) *a.zero_items = 0;
    if (*a.ovf) return; // pass 1 gave up: the exact passes take over
#ifdef HJ_STAMPS
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 4] = hj_now();
#endif
    WfLds L_;
    wf_carve(L_, smem);
    const uint32_t tid = threadIdx.x;
    // pass 2: child c of parent d -> slot d*P + c.  seg_pass1 (multi-GPU, hj_dist.hip): the workgroup is span span0 + parent of
    // a pass 1 whose input arrives as segments (the slots received from every peer): digit d -> slot (d, span0 + parent)
    const FastGeom g = a.seg_pass1 ? FastGeom{a.span0 + parent, a.nspans, a.cap} : FastGeom{parent * a.P, 1u, a.cap};
    uint32_t *scratch = L_.hh; // 17 words, before hh is zeroed
    // segment table of this parent: 4-tuple units per segment, scanned
    uint32_t units = 0;
    if (tid < a.spp) {
        const uint64_t b = a.sbeg[(uint64_t)parent * a.spp + tid], e = a.send[(uint64_t)parent * a.spp + tid];
        const uint32_t cnt = (uint32_t)(e - b);
        units = (cnt + 3) >> 2;
        L_.sb[tid] = ((uint32_t)(b / WC_LINE) << 2) | (units * 4 - cnt); // segment start in lines (slots start on a line) | padding of its last unit
    }
    uint32_t total;
    const uint32_t ex = block_excl_scan<uint32_t>(units, scratch, &total);
    if (tid < a.spp) L_.pc4[tid] = ex;
    if (tid == 0) L_.pc4[a.spp] = total;
    __syncthreads();
    for (uint32_t d = tid; d < 2 * WC_HSTRIDE; d += WC_THREADS) L_.hh[d] = 0;
    if (tid < a.P) L_.line[tid] = slot_line(g, tid);
    if (tid == 0) L_.wlist[(WC_THREADS / 64) * 32] = 0;
    __syncthreads();
    if (a.P == (uint32_t)MAX_PARTS) wc_fast<U, 1, 1>(L_, a.keys, a.pays, 0, 0, 0, a.spp, a.shift, a.P, g, a.out_keys, a.out_pays, a.obeg, a.oend, a.ovf);
    else wc_fast<U, 0, 1>(L_, a.keys, a.pays, 0, 0, 0, a.spp, a.shift, a.P, g, a.out_keys, a.out_pays, a.obeg, a.oend, a.ovf);
#ifdef HJ_STAMPS
    if (a.stamps && threadIdx.x == 0) {
        unsigned long long *s = a.stamps + (size_t)blockIdx.x * 4;
        s[1] = 0; s[2] = hj_now(); s[3] = hj_where();
    }
#endif
}
template <int U>
__global__ __launch_bounds__(WC_THREADS) void k_part2_fast(FastArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    part2_fast_body<U>(a, blockIdx.x, smem);
}
template <int U>
__global__ __launch_bounds__(WC_THREADS) void k_part2_fast2(FastArgs a, FastArgs b) { // both relations: see k_part1_fast2
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x < a.nparents) part2_fast_body<U>(a, blockIdx.x, smem);
    else part2_fast_body<U>(b, blockIdx.x - a.nparents, smem);
}

// tuples per shard (MODE 1 digit, no remap): per-workgroup LDS histogram, one global atomic per shard per workgroup
__global__ __launch_bounds__(512) void k_shard_count(const int32_t *__restrict__ keys, uint64_t n, uint32_t nshards,
                                                     unsigned long long *__restrict__ counts) {
    __shared__ uint32_t h[MAX_PARTS];
    for (uint32_t d = threadIdx.x; d < nshards; d += blockDim.x) h[d] = 0;
    __syncthreads();
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint64_t w0 = lo + (threadIdx.x & ~63u); w0 < hi; w0 += blockDim.x) { // wave-uniform trip count
        const uint64_t i = w0 + lane_id();
        const bool valid = i < hi;
        (void)rank_in_digit(h, valid ? digit_of<1>((uint32_t)keys[i], 0, nshards) : 0u, valid);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nshards; d += blockDim.x)
        if (h[d]) atomicAdd(&counts[d], (unsigned long long)h[d]);
}

hipError_t launch_shard_count(hipStream_t st, const int32_t *keys, uint64_t n, uint32_t nshards, uint64_t *counts) {
    const uint32_t blocks = (uint32_t)(n / 65536 < 1 ? 1 : (n / 65536 > 4096 ? 4096 : n / 65536));
    hipLaunchKernelGGL(k_shard_count, dim3(blocks), dim3(512), 0, st, keys, n, nshards, reinterpret_cast<unsigned long long *>(counts));
    return hipGetLastError();
}

// ---- on-box ceilings for the roofline (bench.py): what the HBM system gives the two access patterns of a radix
// pass, with no partitioning work at all.  KIND 0: stream copy of a column pair, 16 bytes per lane.  KIND 1: the same
// streaming reads, but every 128-byte line (8 lanes x 16 bytes) is stored at a pseudo-random line position of the
// output (bijection on the line index: odd multiplier modulo a power of two) — aligned whole-line scatter, the
// write pattern of the write-combining flush. ----
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 ld_nt(const int4 *p) { v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(p)); return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st_nt(int4 *p, int4 a) { v4i_t v = {a.x, a.y, a.z, a.w}; __builtin_nontemporal_store(v, reinterpret_cast<v4i_t *>(p)); }

template <int KIND, int UNR>
__global__ __launch_bounds__(256) void k_ubench(const int4 *__restrict__ ik, const int4 *__restrict__ ip, int4 *__restrict__ ok,
                                                int4 *__restrict__ op, uint64_t n16, uint64_t line_mask, uint64_t mul, int nt) {
    // a workgroup owns contiguous chunks of 256*UNR units; every thread issues its UNR loads of both columns before
    // the first store, so 2*UNR 16-byte loads per lane are in flight
    const uint64_t chunk = (uint64_t)256 * UNR;
    for (uint64_t base = (uint64_t)blockIdx.x * chunk; base < n16; base += (uint64_t)gridDim.x * chunk) {
        int4 a[UNR], b[UNR];
#pragma unroll
        for (int j = 0; j < UNR; j++) {
            const uint64_t u = base + (uint64_t)j * 256 + threadIdx.x;
            if (u < n16) { if (nt & 2) { a[j] = ld_nt(ik + u); b[j] = ld_nt(ip + u); } else { a[j] = ik[u]; b[j] = ip[u]; } }
        }
#pragma unroll
        for (int j = 0; j < UNR; j++) {
            const uint64_t u = base + (uint64_t)j * 256 + threadIdx.x;
            if (u < n16) {
                uint64_t o = u;
                if (KIND == 1) {
                    const int lg = (nt >> 4) == 15 ? -1 : (nt >> 4); // experiments: log2 lines per scattered chunk; 15 = half lines (64 B)
                    if (lg >= 0) o = ((((u >> (3 + lg)) * mul) & (line_mask >> lg)) << (3 + lg)) | (u & ((8u << lg) - 1));
                    else o = ((((u >> 2) * mul) & (line_mask * 2 + 1)) << 2) | (u & 3);
                }
                if (nt & 1) { st_nt(ok + o, a[j]); st_nt(op + o, b[j]); } else { ok[o] = a[j]; op[o] = b[j]; }
            }
        }
    }
}

// KIND 2 / 3: both columns streamed in only (folded into a word that is practically never stored) / streamed out only (a constant):
// what HBM gives pure reads and pure writes of this shape.  A kernel that reads R bytes and writes W bytes cannot beat
// (R + W) / (R / read_rate + W / write_rate): the ceiling of ITS mix (the materialising join writes 59 % of its bytes at config 4).
__global__ __launch_bounds__(256) void k_ubench_oneway(const int4 *__restrict__ ik, const int4 *__restrict__ ip, int4 *__restrict__ ok,
                                                       int4 *__restrict__ op, uint64_t n16, bool write) {
    uint32_t acc = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * 512; base < n16; base += (uint64_t)gridDim.x * 512) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint64_t u = base + (uint64_t)j * 256 + threadIdx.x;
            if (u >= n16) continue;
            if (write) { const int4 v = make_int4((int)u, 1, 2, 3); ok[u] = v; op[u] = v; }
            else { const int4 a = ik[u], b = ip[u]; acc += (uint32_t)(a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w); }
        }
    }
    if (!write && acc == 0x9E3779B9u) ok[0] = make_int4(0, 0, 0, 0); // practically never: keeps the loads alive
}

// KIND 4-7 (round 6, the layout gate): the SAME bytes as kinds 0/1 moved as ONE array per side — the caller's two columns are the
// halves of one allocation (ip == ik + n, op == ok + n).  Four 16-byte loads per lane in flight before the first store, like kinds 0/1.
//   4  plain one-array copy (the guide's float4-copy shape): two streams instead of four
//   5  line-interleaved pairs: a tuple line is 256 contiguous bytes, its 32 keys then its 32 payloads; 8 lanes move one line pair
//      (loads at a and a + 128): what the INTERMEDIATE partitions would look like if phase C of wc_fast stored its two halves adjacently
//   6  kind 5 with every 256-byte line pair stored at a pseudo-random line-pair position of the one output array (kind 1's scatter
//      with 256-byte chunks into one array)
//   7  two columns in (the API layout), line-interleaved pairs out, scattered as in 6: pass 1 with interleaved intermediates
template <int KIND>
__global__ __launch_bounds__(256) void k_ubench1(const int4 *__restrict__ in, int4 *__restrict__ out, uint64_t np16, uint64_t stride16,
                                                 uint64_t pair_mask, uint64_t mul) {
    // np16 = 16-byte units per column that are moved; stride16 = distance between the two input columns (kind 7)
    if (KIND == 4) {
        const uint64_t n16 = np16 * 2;
        for (uint64_t base = (uint64_t)blockIdx.x * 1024; base < n16; base += (uint64_t)gridDim.x * 1024) {
            int4 a[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { const uint64_t u = base + (uint64_t)j * 256 + threadIdx.x; if (u < n16) a[j] = in[u]; }
#pragma unroll
            for (int j = 0; j < 4; j++) { const uint64_t u = base + (uint64_t)j * 256 + threadIdx.x; if (u < n16) out[u] = a[j]; }
        }
        return;
    }
    for (uint64_t base = (uint64_t)blockIdx.x * 512; base < np16; base += (uint64_t)gridDim.x * 512) {
        int4 a[2], b[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint64_t u = base + (uint64_t)j * 256 + threadIdx.x;
            if (u < np16) {
                if (KIND == 7) { a[j] = in[u]; b[j] = in[stride16 + u]; }
                else { const uint64_t p = (u >> 3) * 16 + (u & 7); a[j] = in[p]; b[j] = in[p + 8]; }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint64_t u = base + (uint64_t)j * 256 + threadIdx.x;
            if (u < np16) {
                const uint64_t pair = KIND == 5 ? (u >> 3) : (((u >> 3) * mul) & pair_mask);
                const uint64_t p = pair * 16 + (u & 7);
                out[p] = a[j]; out[p + 8] = b[j];
            }
        }
    }
}

hipError_t launch_ubench(hipStream_t st, int kind, const int32_t *ik, const int32_t *ip, int32_t *ok, int32_t *op, uint64_t n) {
    const uint64_t n16 = n / 4;
    if (kind >= 4) { // one array per side: the columns are the halves of one allocation (checked by hj_ubench)
        uint64_t pairs = n16 / 8, p2 = 1;
        while (p2 * 2 <= pairs) p2 *= 2;
        const uint64_t np16 = (kind == 6 || kind == 7) ? p2 * 8 : pairs * 8; // the scatter covers a power-of-two number of line pairs
        const uint64_t mul1 = 0x9E3779B97F4A7C15ULL | 1;
        dim3 g1(16384), b1(256);
#define UB1(K_) hipLaunchKernelGGL((k_ubench1<K_>), g1, b1, 0, st, (const int4 *)ik, (int4 *)ok, np16, n16, p2 - 1, mul1)
        if (kind == 4) UB1(4); else if (kind == 5) UB1(5); else if (kind == 6) UB1(6); else UB1(7);
#undef UB1
        return hipGetLastError();
    }
    uint64_t lines = n16 / 8, pow2 = 1;
    while (pow2 * 2 <= lines) pow2 *= 2;
    const uint64_t used16 = kind == 1 ? pow2 * 8 : n16; // the scatter covers the largest power-of-two number of lines
    static int blocks = 0, nt = 0;
    static std::once_flag once; // experiment knobs, read once (contexts on several host threads may get here together)
    std::call_once(once, [] {
        const char *e = getenv("HJ_UB_BLOCKS"); blocks = e ? atoi(e) : 16384;
        const char *h = getenv("HJ_UB_NT"); nt = h ? atoi(h) : 0;
    });
    dim3 g(blocks), b(256);
    const uint64_t mul = 0x9E3779B97F4A7C15ULL | 1;
#define UB(K_, U_) hipLaunchKernelGGL((k_ubench<K_, U_>), g, b, 0, st, (const int4 *)ik, (const int4 *)ip, (int4 *)ok, (int4 *)op, used16, K_ ? pow2 - 1 : (uint64_t)0, mul, nt)
    if (kind == 2 || kind == 3) { // read-only / write-only streams of both columns: the two ends of a kernel's read:write mix
        hipLaunchKernelGGL(k_ubench_oneway, g, b, 0, st, (const int4 *)ik, (const int4 *)ip, (int4 *)ok, (int4 *)op, n16, kind == 3);
        return hipGetLastError();
    }
    // two 16-byte loads of each column in flight per lane (1, 4 and 8 were swept in round 3: no better; the instances are gone)
    if (kind == 0) UB(0, 2); else UB(1, 2);
#undef UB
    return hipGetLastError();
}

// gap-free copy of a partitioned relation given as ranges: partition p moves to [off[p], off[p+1]) (introspection)
__global__ __launch_bounds__(256) void k_compact(const int32_t *__restrict__ k, const int32_t *__restrict__ p,
                                                 const uint64_t *__restrict__ beg, const uint64_t *__restrict__ end, uint32_t nparts,
                                                 const uint64_t *__restrict__ off, int32_t *__restrict__ ok, int32_t *__restrict__ op) {
    for (uint32_t q = blockIdx.x; q < nparts; q += gridDim.x) {
        const uint64_t b = beg[q], n = end[q] - b, o = off[q];
        for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) { ok[o + i] = k[b + i]; op[o + i] = p[b + i]; }
    }
}

// root of the offsets of an unpartitioned relation; resets the relation's overflow flag
__global__ void k_set_root(uint64_t *poff, uint64_t n, uint32_t *flag) { poff[0] = 0; poff[1] = n; if (flag) *flag = 0; }

// ------------------------------------------------------------------------------------------------
// launch wrappers (host)
// ------------------------------------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (device, function) pair, not to a context: the
// high-water marks are kept per device and only ever raised, under a lock (one context per host thread).

hipError_t launch_set_root(hipStream_t st, uint64_t *poff, uint64_t n, uint32_t *flag) {
    hipLaunchKernelGGL(k_set_root, dim3(1), dim3(1), 0, st, poff, n, flag);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_plan(hipStream_t st, const PassArgs &pa) {
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, st, pa.sbeg, pa.send, pa.nseg, pa.span, pa.span_start);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_hist(hipStream_t st, int mode, const PassArgs &pa) {
    dim3 g(pa.max_spans), b(PART_THREADS);
    if (mode == 0)
        hipLaunchKernelGGL(k_hist<0>, g, b, 0, st, pa.keys, pa.nalloc, pa.sbeg, pa.send, pa.nseg, pa.spp, pa.span_start, pa.span, pa.shift, pa.P, pa.mask_or_n, pa.hist, (const uint32_t *)nullptr);
    else
        hipLaunchKernelGGL(k_hist<1>, g, b, 0, st, pa.keys, pa.nalloc, pa.sbeg, pa.send, pa.nseg, pa.spp, pa.span_start, pa.span, pa.shift, pa.P, pa.mask_or_n, pa.hist, pa.remap);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_scan_u32(hipStream_t st, uint32_t *data, const uint32_t *len_ptr, uint64_t mul, uint64_t max_len,
                           uint64_t *chunk_sums, uint64_t *chunk_prefix, uint64_t *total_out) {
    uint32_t nchunks = (uint32_t)((max_len + SCAN_CHUNK - 1) / SCAN_CHUNK);
    if (nchunks == 0) nchunks = 1;
    if (nchunks == 1) { // the whole scan in one workgroup: chunk_prefix[0..1] and the total come from the same launch
        hipLaunchKernelGGL(k_scan_local<uint32_t>, dim3(1), dim3(SCAN_THREADS), 0, st, data, len_ptr, mul, chunk_sums, chunk_prefix, total_out);
        HJ_LAUNCH_CHECK();
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_scan_local<uint32_t>, dim3(nchunks), dim3(SCAN_THREADS), 0, st, data, len_ptr, mul, chunk_sums, (uint64_t *)nullptr, (uint64_t *)nullptr);
    HJ_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, chunk_sums, len_ptr, mul, chunk_prefix, total_out);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_offsets(hipStream_t st, const PassArgs &pa, uint64_t n, uint64_t *coff) {
    uint64_t nthreads = (uint64_t)pa.nparents * pa.P + 1;
    hipLaunchKernelGGL(k_offsets, dim3((uint32_t)((nthreads + 255) / 256)), dim3(256), 0, st, pa.hist, pa.chunk_prefix,
                       pa.span_start, pa.nparents, pa.spp, pa.P, n, coff, pa.beg, pa.end);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

size_t fast_lds_bytes() { return fast_lds_bytes_impl(); }

template <int U, int MODE>
static hipError_t launch_scatter_wc_t(hipStream_t st, const PassArgs &pa) {
    static bool attr_set[64] = {}; // per device
    const size_t lds = fast_lds_bytes();
    auto fn = k_scatter_wc<U, MODE>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(g_attr_mutex);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(fn, dim3(pa.max_spans), dim3(WC_THREADS), lds, st, pa.keys, pa.pays, pa.nalloc, pa.sbeg, pa.send, pa.nseg, pa.spp,
                       pa.span_start, pa.span, pa.shift, pa.P, pa.hist, pa.chunk_prefix, pa.out_keys, pa.out_pays,
                       pa.n_out, pa.remap);
    return hipGetLastError();
}

// the exact scatter: radix digits (mode 0) or multi-GPU shards (mode 1: hash, optional position table, any fan-out <= 512)
hipError_t launch_scatter(hipStream_t st, int mode, const PassArgs &pa) {
    return mode == 0 ? launch_scatter_wc_t<2, 0>(st, pa) : launch_scatter_wc_t<2, 1>(st, pa);
}

// slot capacity of the histogram-free passes: expected count + 8 standard deviations (Poisson), rounded to the
// digit's LDS lines (K = 512/P lines of 32 tuples), plus one such granule (a slot counts as full one granule early)
uint32_t fast_slot_cap(uint64_t expected, uint32_t P) {
    const uint64_t gran = (uint64_t)(MAX_PARTS / P) * WC_LINE;
    uint64_t sd = 1;
    while (sd * sd < expected) sd++;
    uint64_t c = expected + 8 * sd + 32;
    c = ((c + gran - 1) / gran) * gran + gran;
    return c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c;
}

template <typename F>
static hipError_t fast_attr(F fn, bool *flags) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_attr_mutex);
    if (dev < 0 || dev >= 64 || !flags[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast_lds_bytes());
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) flags[dev] = true;
    }
    return hipSuccess;
}

hipError_t launch_part1_fast(hipStream_t st, const FastArgs &fa) {
    static bool set[3][64] = {};
    hipError_t e;
    // mode 1 = the multi-GPU level-0 split (shard digit); few digits take the wave-aggregated rank
    if (fa.mode == 0) {
        auto fn = k_part1_fast<2, 0, false>;
        if ((e = fast_attr(fn, set[0])) != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(fa.nspans), dim3(WC_THREADS), fast_lds_bytes(), st, fa);
    } else if (fa.P <= 4) {
        auto fn = k_part1_fast<2, 1, true>;
        if ((e = fast_attr(fn, set[1])) != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(fa.nspans), dim3(WC_THREADS), fast_lds_bytes(), st, fa);
    } else {
        auto fn = k_part1_fast<2, 1, false>;
        if ((e = fast_attr(fn, set[2])) != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(fa.nspans), dim3(WC_THREADS), fast_lds_bytes(), st, fa);
    }
    return hipGetLastError();
}

hipError_t launch_part1_var(hipStream_t st, const FastArgs &fa, const VarArgs &va, bool heavy) {
    static bool set[4][64] = {};
    hipError_t e;
#define HJ_P1V(IDX, HV, HT) do { auto fn = k_part1_var<2, HV, HT>; if ((e = fast_attr(fn, set[IDX])) != hipSuccess) return e; \
        hipLaunchKernelGGL(fn, dim3(fa.nspans), dim3(WC_THREADS), fast_lds_bytes(), st, fa, va); } while (0)
    if (va.hot.mode && heavy) return hipErrorInvalidValue; // (the host does not bypass where the residue still has a dominant digit)
    if (va.hot.mode == 1) HJ_P1V(2, false, 1);
    else if (va.hot.mode == 2) HJ_P1V(3, false, 2);
    else if (heavy) HJ_P1V(0, true, 0);
    else HJ_P1V(1, false, 0);
#undef HJ_P1V
    return hipGetLastError();
}

hipError_t launch_part2_var(hipStream_t st, const FastArgs &fa, const VarArgs &va, uint32_t nwg, bool, bool) {
    static bool set[64] = {};
    auto fn = k_part2_var<2>;
    hipError_t e = fast_attr(fn, set);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(nwg), dim3(WC_THREADS), fast_lds_bytes(), st, fa, va);
    return hipGetLastError();
}

hipError_t launch_part2_fast(hipStream_t st, const FastArgs &fa) {
    static bool set[64] = {};
    auto fn = k_part2_fast<2>;
    hipError_t e = fast_attr(fn, set);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(fa.nparents), dim3(WC_THREADS), fast_lds_bytes(), st, fa);
    return hipGetLastError();
}

hipError_t launch_part1_fast2(hipStream_t st, const FastArgs &fa, const FastArgs &fb) {
    static bool set[64] = {};
    auto fn = k_part1_fast2<2>;
    hipError_t e = fast_attr(fn, set);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(fa.nspans + fb.nspans), dim3(WC_THREADS), fast_lds_bytes(), st, fa, fb);
    return hipGetLastError();
}

hipError_t launch_part2_fast2(hipStream_t st, const FastArgs &fa, const FastArgs &fb) {
    static bool set[64] = {};
    auto fn = k_part2_fast2<2>;
    hipError_t e = fast_attr(fn, set);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(fa.nparents + fb.nparents), dim3(WC_THREADS), fast_lds_bytes(), st, fa, fb);
    return hipGetLastError();
}


hipError_t launch_compact(hipStream_t st, const int32_t *k, const int32_t *p, const uint64_t *beg, const uint64_t *end,
                          uint32_t nparts, const uint64_t *off, int32_t *ok, int32_t *op) {
    hipLaunchKernelGGL(k_compact, dim3(nparts < 8192 ? nparts : 8192), dim3(256), 0, st, k, p, beg, end, nparts, off, ok, op);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace hj
