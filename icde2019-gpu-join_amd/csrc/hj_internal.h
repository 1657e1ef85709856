// hj_internal.h — shared between the kernel files (hj_part.hip, hj_join.hip, hj_util.hip; device helpers: hj_device.h) and the host side (hj_api.hip).
#ifndef HJ_INTERNAL_H_
#define HJ_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hj {

// partition kernels: 512-thread workgroups (8 wave64), 4 x 16-byte loads per thread per tile
constexpr int PART_THREADS = 512;
constexpr int TILE_U = 4;
constexpr int TILE = PART_THREADS * 4 * TILE_U; // default tile: 8192 tuples (span granularity)
constexpr int MAX_PARTS = 512;                  // fan-out limit of one pass (9 bits)
constexpr int MAX_PARENTS = 1024;               // k_plan is a single workgroup

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER = 16;
constexpr int SCAN_CHUNK_LOG = 12;
constexpr int SCAN_CHUNK = 1 << SCAN_CHUNK_LOG; // = SCAN_THREADS * SCAN_PER

constexpr int JOIN_THREADS = 512;
constexpr int JOIN_WAVES = JOIN_THREADS / 64;

constexpr int MAX_SEGS = 8192;                  // segments of one exact pass (parents x segments per parent)

// One exact (histogram + scan + scatter) radix pass.  The input is a list of segments [sbeg[i], send[i]) of the
// input columns; spp consecutive segments form one parent partition (contiguous partitions: sbeg = offsets,
// send = offsets + 1, spp = 1).
struct PassArgs {
    const int32_t *keys, *pays; // input columns
    uint64_t nalloc;            // true length of the input arrays
    const uint64_t *sbeg, *send; // segment ranges [nseg]
    uint32_t nseg, spp;         // nparents = nseg / spp
    uint32_t nparents;
    uint32_t *span_start;       // [nseg+1]
    uint32_t span;              // tuples per span
    uint32_t max_spans;         // launch bound
    uint32_t shift, P, mask_or_n;
    uint32_t *hist;             // [max_spans * P]
    uint64_t *chunk_sums, *chunk_prefix;
    int32_t *out_keys, *out_pays;
    uint64_t n_out;             // tuples of the pass (end of the last child partition)
    uint64_t *beg, *end;        // optional: child partition ranges for the join [nparents * P]
    const uint32_t *remap;      // optional (shard digits): output position of each shard
};

// The histogram-free ("optimistic") passes.  Output slots have a fixed capacity: pass 1 gives every
// (digit, span) pair a slot of cap tuples, pass 2 every final partition; a slot that would overflow raises
// *ovf and the exact passes redo the relation.  See hj_part.hip.
struct FastArgs {
    const int32_t *keys, *pays;  // input columns
    uint64_t n;                  // pass 1: tuples of the (contiguous) input
    uint32_t span, nspans;       // pass 1: tuples per workgroup, number of workgroups
    const uint64_t *sbeg, *send; // pass 2: input segments [nparents * spp]
    uint32_t nparents, spp;
    uint32_t shift, P;           // digit = (key >> shift) & (P - 1)
    uint32_t cap;                // slot capacity in tuples (multiple of the digit's LDS lines)
    int32_t *out_keys, *out_pays;
    uint64_t *obeg, *oend;       // slot ranges written by the pass: pass 1 [P * nspans] (digit major), pass 2 [nparents * P]
    uint32_t *ovf;               // overflow flag (also read: a set flag makes the kernel return at once)
    uint32_t mode;               // pass 1: 0 = radix digit, 1 = multi-GPU shard of the key (hash; P = number of GPUs)
    uint32_t seg_pass1, span0;   // launch_part2_fast as a pass 1 over received segments: workgroup b is span span0 + b of nspans
    uint64_t *zero_items;        // pass 2, optional: the join's item counter, zeroed by workgroup 0 (k_join_plan_atomic reserves on it)
    unsigned long long *stamps;  // experiment builds only (-DHJ_STAMPS, `make stamps`): per workgroup {start, end, hw id, -} in 100-MHz ticks
};

// one work item of the join: build partition [b0, b0+nb), probe chunk [q0, q1) of partition p
// p bit 31 (JOIN_ITEM_LIST): the probe side of the item is a LIST of whole ranges — q0 = first range, q1 = number of ranges,
// range j = q0 + j * JoinArgs.rstride — that share the one table build
struct JoinItem {
    uint64_t b0, q0, q1;
    uint32_t nb, p;
};
constexpr uint32_t JOIN_ITEM_LIST = 0x80000000u;
// general items (JoinArgs.general: the build relation is skewed):
//   JOIN_ITEM_BLIST  the TABLE side of the item is a list of ranges: b0 = first range, nb = number of ranges (stride of that relation)
//   JOIN_ITEM_SWAP   roles flipped for this item: the table is built from the relation the host calls probe side, the designated
//                    build side is streamed (b0/nb and q0/q1 then index that relation's arrays)
constexpr uint32_t JOIN_ITEM_BLIST = 0x40000000u, JOIN_ITEM_SWAP = 0x20000000u;

struct JoinArgs {
    const int32_t *bk, *bp;  // build side, partitioned
    const uint64_t *bbeg, *bend; // partition p = [bbeg[p], bend[p])
    uint64_t b_nalloc;
    const int32_t *pk, *pp;  // probe side, partitioned
    const uint64_t *pbeg, *pend;
    uint64_t p_nalloc;
    const uint32_t *bflag, *pflag; // overflow flags of histogram-free partitions (nullptr: ranges known good)
    const uint32_t *rpart;   // probe side given as RANGES (sampled path): partition id of range i; nullptr: range i = partition i
    const uint32_t *pr0, *pnr; // ... or PARTITIONS with a list of ranges each: first range, number of ranges (stride rstride)
    uint32_t rstride;
    const uint32_t *br0, *bnr; // build side in PARTITIONS with a list of ranges each (sampled build relation), stride bstride
    uint32_t bstride;
    uint32_t general;        // 1: general items (table side may be a list, roles may be flipped per partition): k_join_plan_gen / GEN kernels
    const JoinItem *items;   // (build partition, probe chunk) descriptors
    const uint64_t *n_items;
    uint32_t radix_bits, cap, nh, chunk;
    uint64_t *wave_counts, *wave_agg;                     // count kernel outputs [items * JOIN_WAVES]
    int32_t *out_key, *out_bpay, *out_ppay;
    uint64_t out_cap;
    unsigned long long *out_cursor; // one-probe materialisation: next free output position (zeroed by k_join_plan)
    // late materialisation: column-major extra columns gathered by row id on every match
    const int32_t *Db, *Dp;  // build side / probe side tables
    uint32_t ncb, ncp;       // columns to gather
    uint64_t sb, sp;         // column stride (elements)
    unsigned long long *stamps; // experiment builds only (-DHJ_STAMPS): per item {start, table built, end, hw id} in 100-MHz ticks
};

hipError_t launch_set_root(hipStream_t st, uint64_t *poff, uint64_t n, uint32_t *flag = nullptr);
hipError_t launch_plan(hipStream_t st, const PassArgs &pa);
hipError_t launch_hist(hipStream_t st, int mode, const PassArgs &pa);
hipError_t launch_scan_u32(hipStream_t st, uint32_t *data, const uint32_t *len_ptr, uint64_t mul, uint64_t max_len,
                           uint64_t *chunk_sums, uint64_t *chunk_prefix, uint64_t *total_out);
hipError_t launch_offsets(hipStream_t st, const PassArgs &pa, uint64_t n, uint64_t *coff);
hipError_t launch_scatter(hipStream_t st, int mode, const PassArgs &pa);
hipError_t launch_part1_fast(hipStream_t st, const FastArgs &fa);
hipError_t launch_part2_fast(hipStream_t st, const FastArgs &fa);
hipError_t launch_part1_fast2(hipStream_t st, const FastArgs &fa, const FastArgs &fb); // both relations of a join in one launch per pass
hipError_t launch_part2_fast2(hipStream_t st, const FastArgs &fa, const FastArgs &fb);
uint32_t fast_slot_cap(uint64_t expected, uint32_t P);
// The heavy-hitter bypass (round 6).  A relation known to be skewed has a DIRECT-MAPPED table of up to HOT_SLOTS candidate keys (its most
// frequent keys by a sample, slot = hot_slot(key); of two candidates with one slot the more frequent stays — a first version with 4-way
// buckets held 3 points more of config 4's S and cost pass 1 a 16-byte LDS read and four compares per tuple: slower); every step k_hot_build scans the OTHER relation for them (cnt = its tuples
// with that key, pay = the payload of one of them), and pass 1 of the skewed relation joins the tuples of candidates that are UNIQUE
// in the other relation where it first sees them — they never enter a partition.  mode 1: the matches are counted (acc[0] += 1,
// acc[1] += pay * payload); mode 2: (key, table payload, streamed payload) tuples are written at *cursor (one exact reservation per
// workgroup and round, positions >= out_cap are not written).  The reference's remedy for a hot build partition is the role flip of
// jp.cu:929-1003; this removes the hot PROBE tuples from both passes and the probe.
constexpr uint32_t HOT_SLOTS = 1024;
constexpr int32_t HOT_NEVER = INT32_MIN; // never a candidate (the sentinel of the sampling table)
struct HotArgs {
    uint32_t mode = 0;           // 0 off, 1 count, 2 emit
    const uint32_t *cand = nullptr, *cnt = nullptr; // [HOT_SLOTS] candidate keys (a filler never maps to its own slot) / tuples of the other relation per candidate
    const int32_t *pay = nullptr;                   // [HOT_SLOTS] payload of one such tuple
    unsigned long long *acc = nullptr;              // mode 1: {matches, aggregate}
    int32_t *out_key = nullptr, *out_tab = nullptr, *out_str = nullptr; // mode 2: output columns (key, the other relation's payload, this relation's payload)
    uint64_t out_cap = 0;
    unsigned long long *cursor = nullptr;
    unsigned long long *stamps = nullptr;            // experiment builds (-DHJ_STAMPS): per workgroup {ticks in phase A, B, C, waiting at the end of C, rounds, -, -, -}
};
__host__ __device__ inline uint32_t hot_slot(uint32_t key) { return (key * 0x9E3779B1u) >> 22; }
static_assert(HOT_SLOTS == 1024, "hot_slot yields 10 bits");
__host__ __device__ inline uint32_t hot_filler(uint32_t slot) { return slot == 0 ? 1u : 0u; } // a key of ANOTHER slot: hot_slot(0) = 0, hot_slot(1) = 0x278
// the sampled path of skewed relations (hj_part.hip: k_part1_var, k_part2_var)
struct VarArgs {
    const uint32_t *vbase, *vcap; // pass 1: per digit [P]; pass 2: per (parent, child) [nparents*P]
    const uint32_t *lt, *own;     // lines dealt: (lines << 16 | first line) per digit [P] (pass 2: per parent row), owner digit per LDS line [512]
    const uint32_t *heavy;        // pass 1: [1]; pass 2: per parent — one digit holds more than a quarter of the input
    const uint4 *wg;              // pass 2: per workgroup {parent d, first pass-1 span, spans, output position of its sub-slots}
    HotArgs hot;                  // pass 1: the heavy-hitter bypass
};
hipError_t launch_skew_probe(hipStream_t st, const int32_t *keys, uint64_t n, uint32_t nsamp, uint32_t b1, uint32_t b2, uint32_t *hist);
hipError_t launch_hot_sample(hipStream_t st, const int32_t *keys, uint64_t n, uint32_t nsamp, uint32_t *tkey, uint32_t *tcnt, uint32_t slots);
hipError_t launch_hot_collect(hipStream_t st, const uint32_t *tkey, const uint32_t *tcnt, uint32_t slots, uint32_t thr, uint2 *out, uint32_t *nout, uint32_t cap);
hipError_t launch_hot_build(hipStream_t st, const int32_t *keys, const int32_t *pays, uint64_t n, const uint32_t *cand, uint32_t *cnt, int32_t *pay, unsigned long long *zero_acc);
hipError_t launch_sample_joint(hipStream_t st, const int32_t *keys, uint64_t n, uint32_t bits, uint32_t stride, uint32_t *hist, uint64_t *sampled,
                               const uint32_t *hot_cand = nullptr, const uint32_t *hot_cnt = nullptr);
hipError_t launch_part1_var(hipStream_t st, const FastArgs &fa, const VarArgs &va, bool heavy);
hipError_t launch_part2_var(hipStream_t st, const FastArgs &fa, const VarArgs &va, uint32_t nwg, bool any_heavy, bool any_light);
hipError_t launch_dist_segments(hipStream_t st, const uint64_t *oend, uint32_t G, uint32_t nsp, uint32_t cap, uint32_t me, uint64_t base,
                                uint64_t *sbeg, uint64_t *send, uint32_t *flag, uint64_t *received, uint64_t *received_self = nullptr);
hipError_t launch_compact(hipStream_t st, const int32_t *k, const int32_t *p, const uint64_t *beg, const uint64_t *end,
                          uint32_t nparts, const uint64_t *off, int32_t *ok, int32_t *op);
hipError_t launch_join_plan(hipStream_t st, const JoinArgs &a, uint32_t nparts, uint32_t *items_cnt, uint64_t *zero2, uint64_t *zero_cursor);
hipError_t launch_join_plan_fused(hipStream_t st, const JoinArgs &a, uint32_t nparts, JoinItem *items, uint64_t *zero2, uint64_t *zero_cursor,
                                  uint64_t *n_items);
// plan + expand in ONE launch for any partition count: item slots reserved with one atomic per workgroup on *n_items, which the
// caller (or the pass-2 kernel before it, FastArgs.zero_items) has zeroed; the items come out in no particular order
hipError_t launch_join_plan_atomic(hipStream_t st, const JoinArgs &a, uint32_t nparts, JoinItem *items, uint64_t *zero2, uint64_t *zero_cursor,
                                   uint64_t *n_items);
hipError_t launch_sum2(hipStream_t st, const uint64_t *cnt, const uint64_t *agg, const uint32_t *len_ptr, uint64_t mul, uint64_t *out2,
                       const uint64_t *extra2 = nullptr); // extra2: two more words added in (the matches / aggregate of the heavy-hitter bypass)
hipError_t launch_join_expand(hipStream_t st, const JoinArgs &a, uint32_t nparts, const uint32_t *items_scanned,
                              const uint64_t *chunk_prefix, JoinItem *items);
size_t join_lds_bytes(uint32_t nh, uint32_t cap, bool tag16);
hipError_t join_set_lds_limit(int device, size_t bytes);
hipError_t launch_join(hipStream_t st, const JoinArgs &a, uint32_t max_items, bool tag16, int jm); // jm: 0 count + aggregate, 2 late materialisation
size_t join_mat_lds_bytes(uint32_t nh, uint32_t cap, bool tag16);
hipError_t launch_join_mat_reg(hipStream_t st, const JoinArgs &a, uint32_t max_items, bool tag16); // materialise in ONE probe, matches held in registers
hipError_t launch_np_max(hipStream_t st, const int32_t *keys, uint64_t n, uint32_t *out_max);
hipError_t launch_np_perfect(hipStream_t st, const int32_t *bk, uint64_t nb, const int32_t *bp, const int32_t *pk, const int32_t *pp,
                             uint64_t np, int32_t *lookup, uint64_t range, uint64_t *out2);
hipError_t launch_np_chained(hipStream_t st, const int32_t *bk, const int32_t *bp, uint64_t nb, const int32_t *pk, const int32_t *pp,
                             uint64_t np, uint32_t log_slots, int32_t *head, int32_t *next, uint64_t *out2);
hipError_t launch_dot(hipStream_t st, const int32_t *a, const int32_t *b, const uint64_t *n_ptr, uint64_t cap, uint64_t *out);
hipError_t launch_fill(hipStream_t st, int32_t *p, uint64_t n, int mode, uint64_t first);
hipError_t launch_gen_unique(hipStream_t st, int32_t *keys, uint64_t n, uint64_t first, uint64_t domain, uint64_t seed);
hipError_t launch_gen_zipf(hipStream_t st, int32_t *keys, uint64_t n, uint64_t first, uint64_t alphabet, double theta, uint64_t seed);
hipError_t launch_digest(hipStream_t st, const int32_t *a, const int32_t *b, const int32_t *c, uint64_t n, uint64_t *out);
hipError_t launch_verify_partitions(hipStream_t st, const int32_t *keys, const int32_t *pays, const uint64_t *beg,
                                    const uint64_t *end, uint32_t nparts, uint32_t id_shift, uint32_t id_base,
                                    uint64_t *misplaced, uint64_t *digests, uint64_t *sizes);
hipError_t launch_ubench(hipStream_t st, int kind, const int32_t *ik, const int32_t *ip, int32_t *ok, int32_t *op, uint64_t n);
hipError_t launch_shard_count(hipStream_t st, const int32_t *keys, uint64_t n, uint32_t nshards, uint64_t *counts);

} // namespace hj
#endif
