// hj_ctx.h — the context behind the C ABI (include/hj.h) and the internal helpers hj_api.hip shares with hj_dist.hip.
#ifndef HJ_CTX_H_
#define HJ_CTX_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "hj.h"
#include "hj_internal.h"

namespace hjx {

constexpr uint64_t PAD = 16; // int32 elements of slack after every column (16-byte tail loads)
constexpr uint32_t TARGET_SPANS = 1024; // 4 spans per CU: measured best (profiles/r1_spans_sweep.txt)
constexpr uint32_t DEFAULT_CAP = 4608, DEFAULT_HEADS = 4096, DEFAULT_CHUNK = 65536;
constexpr uint32_t TARGET_PART = 4096; // average build tuples per final partition
// the context's device scalars (uint64 words): the OUTPUT CURSOR of the materialising kernels sits on a 128-byte line of its own, four lines
// from the words the kernels poll (the relations' overflow flags, read by every partition workgroup every few rounds): with the cursor
// beside them (word 10, until round 6) every reservation of the writing bypass invalidated the line the flags are read from
constexpr uint32_t SC_CURSOR = 64;
constexpr size_t SC_BYTES = 1024;

struct KStat { std::string name; uint32_t launches = 0; float total_ms = 0, last_ms = 0; };
struct Stamp { int kid; hipEvent_t a, b; };

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

struct Rel {
    const int32_t *in_k = nullptr, *in_p = nullptr; // input columns (caller's or own_*)
    uint64_t n = 0;
    uint64_t n_bound = 0;     // multi-GPU sliced path: upper bound of the tuples the CURRENT partitions can hold (sizes the work-item list;
                              // 0 = n).  n itself stays the nominal size the radix bits and the build side were chosen from.
    bool bound = false;
    Buf own_k, own_p;         // hj_load_host copies
    Buf a_k, a_p, b_k, b_p;   // pass-1 / final partitioned columns
    Buf off1, off2, root;     // partition offsets (uint64) of the exact passes
    Buf beg, end;             // final partition ranges [nparts] (uint64): what the join reads, whichever path ran
    Buf s1beg, s1end;         // slot ranges written by the histogram-free pass 1 [P1 * nspans]
    Buf comp_k, comp_p, comp_off; // gap-free copy for hj_get_partitions when the layout is slotted
    const int32_t *part_k = nullptr, *part_p = nullptr;
    const uint64_t *part_beg = nullptr, *part_end = nullptr;
    const uint64_t *part_off = nullptr; // nparts+1 contiguous offsets: only valid when the exact passes ran
    uint64_t n_alloc = 0;      // elements of the partitioned columns (bounds of 16-byte tail loads)
    uint32_t nparts = 0;
    uint32_t pb1 = 0, pb2 = 0; // radix bits this relation was partitioned with
    bool partitioned = false;
    bool fast_tried = false;   // the histogram-free passes were queued: which layout holds is known on the device only
    bool probed = false;       // the look before the first attempt on this binding (skew_probe) has been taken
    bool prefer_exact = false; // the last histogram-free attempt on this binding overflowed: go straight to the exact passes
    bool flag_known_good = false; // fast_tried and the flag has been read as 0 since: the slotted ranges are valid
    // what the host knows about the relation's overflow flag ON THE DEVICE: kernels that may raise it were queued and nobody has read
    // it since (unread) / it was read raised and not reset since (maybe_set).  Neither: the flag is 0 and the histogram-free passes
    // of a steady-state step need no k_set_root in front of them (partition_both)
    bool flag_unread = true, flag_maybe_set = true;
    // the sampled path (a relation known to be skewed, on the probe side): histogram-free passes with per-digit capacities
    bool sampled = false;         // the current partitions came from it: ranges, not one range per partition
    bool sampled_failed = false;  // it overflowed on this binding: exact passes from now on
    bool force_exact = false;     // introspection calls want one gap-free range per partition
    uint32_t nranges = 0;         // ranges of the partitioned relation (== nparts unless sampled)
    const uint32_t *rpart = nullptr; // sampled: partition id of every range
    const uint32_t *pr0 = nullptr, *pnr = nullptr; // ... and per partition: first range, number of ranges (stride rstride)
    uint32_t rstride = 0;
    struct Sampled {
        bool valid = false;
        uint64_t n = 0;
        uint32_t b1 = 0, b2 = 0, span = 0, nspans = 0, nwg2 = 0, nranges = 0;
        uint64_t sizeA = 0, sizeB = 0;
        bool heavy1 = false, any_heavy2 = false, any_light2 = false;
        uint64_t sample_size = 0;
        Buf tab;                  // every table below, one allocation
        const uint32_t *vbase1 = nullptr, *vcap1 = nullptr, *lt1 = nullptr, *own1 = nullptr, *heavy1_d = nullptr;
        const uint32_t *cbase2 = nullptr, *cap2 = nullptr, *lt2 = nullptr, *own2 = nullptr, *heavy2 = nullptr, *wg2 = nullptr, *rpart = nullptr, *pr0 = nullptr, *pnr = nullptr;
        Buf rbeg, rend;           // ranges written by pass 2 [nranges]
        // the heavy-hitter bypass (only in Rel::sph): the plan's capacities leave out the tuples pass 1 joins itself
        bool hot_ready = false;   // the candidate table is on the device
        Buf hot_tab;              // cand[HOT_SLOTS] | cnt[HOT_SLOTS] | pay[HOT_SLOTS]
        uint32_t hot_keys = 0;    // candidates in the table
        double hot_share = 0;     // sampled share of the relation's tuples the bypass takes (candidates unique in the other relation)
    } sp, sph;                    // sph: the plan used when pass 1 bypasses the heavy hitters (hj_join, hj_join_and_materialize)
    int hot_mode = 0;             // the CURRENT partitions lack the tuples pass 1 joined itself: 1 = counted (scalars[13], [14]), 2 = written to the output
    bool hot_useless = false;     // the bypass was looked at for this binding and does not pay (or cannot be planned): plain sampled path
};

} // namespace hjx

struct hj_ctx {
    using Buf = hjx::Buf; using Rel = hjx::Rel; using KStat = hjx::KStat; using Stamp = hjx::Stamp;
    int device = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    hj_config cfg{};
    uint32_t bits1 = 0, bits2 = 0, cap = 0, nh = 0, chunk = 0; // effective
    int build = HJ_REL_R;
    std::string err;
    Rel rel[2];
    // workspace
    struct PassWs { Buf span_start, hist, chunk_sums, chunk_prefix; } ws[2]; // per relation (passes of one relation are serial)
    Buf items_cnt, items, wave_counts, wave_agg, jchunk_sums, jchunk_prefix;
    Buf scalars;                // device u64: [0] n_items, [1] matches, [2] agg, [3] misc, [4] misc, [5..7] baselines, [8],[9] overflow flags of R, S, [SC_CURSOR = 64] output cursor of k_join_mat (a line of its own), [12] scratch, [13],[14] matches / aggregate of the heavy-hitter bypass
    uint64_t *h_scalars = nullptr; // pinned host mirror (8 x u64) + [8],[9]: the relations' overflow flags
    bool join_planned = false;     // per-wave counts + item list of the current partitions are on the device
    hj::JoinArgs last_args{};
    bool last_tag16 = false;
    uint64_t last_matches = 0, last_agg = 0;
    uint32_t max_items = 0;
    uint32_t redo_mask = 0;         // relations whose overflow flag came back raised with the last result block
    uint32_t target_spans = 0;      // experiment knob (HJ_TARGET_SPANS)
    uint32_t fork_log2 = 40;        // inputs up to 2^fork_log2 tuples (= always): S's partition passes on a second stream beside R's (HJ_FORK_LOG2)
    uint32_t merge_log2 = 29;       // |R|+|S| up to 2^merge_log2: both relations' histogram-free passes in ONE launch per pass (HJ_MERGE_LOG2; 0 = never)
    bool plan_atomic = true;        // plain items: plan + expand in one launch with atomic slot reservation (HJ_PLAN_ATOMIC=0: plan + scan + expand)
    bool items_zeroed = false;      // the join's item counter (scalars[0]) was zeroed by the last pass-2 launch: k_join_plan_atomic may reserve on it
    int ncu = 256;                  // CUs of the device
    int force_sampled = 0;          // HJ_FORCE_SAMPLED (experiments): bit r = relation r takes the sampled path without having overflowed
    double var_guide = 2.0;         // HJ_VAR_GUIDE: pass-2 piece sizing of the sampled path (plan_sampled); 0 = pieces of one span
    bool force_build_r = false;     // streaming probe side: R builds whatever the segment size
    int fast_path = 1;              // histogram-free passes first, exact passes as the fallback (HJ_FAST_PATH=0 / hj_config.exact_only)
    // the heavy-hitter bypass: what the entry point that is running wants from pass 1 of a skewed probe side (0 nothing — every entry
    // point but hj_join (1: count) and hj_join_and_materialize (2: write to hot_out)); partition_rel reads it
    int hot_request = 0;
    int32_t *hot_out[3] = {nullptr, nullptr, nullptr}; // key, payR, payS
    uint64_t hot_cap = 0;
    unsigned long long *stamps_join = nullptr, *stamps_part2 = nullptr; // hj_debug_set_stamps: experiment builds (-DHJ_STAMPS) write per-workgroup timelines there
    bool tags_legacy = false;       // HJ_TAGS_LEGACY=1 (A/B): 16-bit tags only at >= 16 radix bits, as until round 5
    bool replan = false;            // HJ_REPLAN (experiments): re-plan the sampled geometry at every call
    bool debug = false;             // HJ_DEBUG: stderr diagnostics
    int hot_enable = 1;             // HJ_HOT=0: never bypass (A/B)
    uint32_t bits1_similar = 7;     // HJ_BITS1_SIMILAR: bits of the first pass when the two relations are of similar size (choose_bits); 0 / 9 = always 9
    bool keep_nine = false;         // a rank of the multi-GPU join: its slices, not its relations, run side by side — the first pass keeps 9 bits
    uint32_t skew_probe_log2 = 26;  // HJ_SKEW_PROBE: relations of at least 2^this tuples get a look at 2^16 keys before their first optimistic attempt (0 = never)
    Buf probe_hist;
    double hot_min_share = 0.10;    // HJ_HOT_MIN_SHARE: smallest sampled share of the relation worth the lookups
    hipStream_t copy = nullptr;     // H2D of the next probe segment
    Buf shard_root, shard_off;      // hj_shard_split: persistent (no allocation in the steady state)
    uint64_t *h_shard_off = nullptr;
    Buf seg_k[2], seg_p[2];         // double-buffered probe segments / level-0 S partitions
    Buf cop_k[2], cop_p[2];         // double-buffered level-0 R partitions (co-processing)
    int32_t *host_k[2] = {nullptr, nullptr}, *host_p[2] = {nullptr, nullptr}; // pinned staging of the host split (R, S): kept across calls
    size_t host_cap[2] = {0, 0};    // elements
    int numa_nodes = 0, numa_gpu_node = -1, numa_pinned_cpus = 0; // co-processing: host topology seen by the last call
    double host_split_gbs = 0;      // throughput of the last host level-0 split (bytes read + written per second)
    uint32_t coprocess_groups = 0;  // residency groups of the last co-processing call
    Buf out_k[2], out_p1[2], out_p2[2]; // streamed materialisation: double-buffered device output columns
    hipStream_t d2h = nullptr;      // third stream: output columns back to the host (hjcp.cu:1947-1961)
    hipEvent_t out_ready[2] = {}, out_free[2] = {};
    hipEvent_t seg_ready[2] = {};
    hipEvent_t seg_joined[2] = {};  // streaming probe, count-only: the join of the segment in staging buffer b is done
    Buf seg_res;                    // ... per segment {matches, aggregate, S's overflow flag, -}
    hipStream_t aux = nullptr;      // S's partition passes run beside R's
    hipStream_t aux_hi = nullptr;   // ... relations of very different sizes: the larger one's passes, at high priority (partition_both)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // hj_config.graph: the captured step and what it is tied to
    hipGraphExec_t graph_exec = nullptr;
    bool graph_warm = false;
    const int32_t *gkey_k[2] = {nullptr, nullptr}, *gkey_p[2] = {nullptr, nullptr};
    uint64_t gkey_n[2] = {0, 0};
    bool gkey_pe[2] = {false, false};
    hipStream_t gkey_st = nullptr;
    // host-side breakdown of the last hj_join call (hj_last_call_breakdown): what a FIRST call on a binding spends where
    struct CallProf { double t0 = 0, alloc_ms = 0, attempt_ms = 0, plan_ms = 0, total_ms = 0; uint32_t allocs = 0; } prof;
    // timing
    int events = 0;                 // 0 none (default), 1 main kernels (partition passes / join), 2 every launch: hj_enable_timings, HJ_KERNEL_EVENTS
    std::vector<KStat> kstats;
    std::vector<Stamp> stamps;
    std::vector<hipEvent_t> pool;
};


namespace hjx {

int fail(hj_ctx *c, int code, const char *fmt, ...);
int ensure(hj_ctx *c, Buf &b, size_t bytes);
void release(Buf &b);
void choose_bits(hj_ctx *c);
void invalidate(hj_ctx *c, int rel = -1);
// Geometry of the histogram-free passes for a relation of n tuples
struct FastPlan { uint32_t span, nspans, cap1, cap2; uint64_t sizeA, sizeB; };
bool plan_fast(const hj_ctx *c, uint64_t n, uint32_t P1, uint32_t P2, FastPlan &f);
int fetch_scalars(hj_ctx *c);
int kid_of(hj_ctx *c, const char *name);
hipEvent_t get_event(hj_ctx *c);
void resolve_completed(hj_ctx *c);
int hj_join_count_noretry(hj_ctx *c, uint64_t *matches, uint64_t *agg);
int hj_join_count_enqueue(hj_ctx *c);
// the one-probe materialiser without any host read (hj_dist.hip: a probe-side group joined under the exchange): items planned, kernel
// enqueued, the output cursor (scalars[10]) keeps counting across calls when keep_cursor is set
int hj_join_materialize_enqueue(hj_ctx *c, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap, bool keep_cursor);
// plan + one probe + read-back with the local redo of overflowed relations; *n_out may exceed cap (nothing beyond cap is written):
// the caller decides what that means (hj_join_materialize: HJ_ECAPACITY; hj_dist: every rank learns it first)
int materialize_local(hj_ctx *c, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap, uint64_t *n_out);
void hj_invalidate_all(hj_ctx *c);
void drop_graph(hj_ctx *c);
int whole_partitions(hj_ctx *c, int ok_mode); // partitions made under the heavy-hitter bypass are redone without it unless hot_mode == ok_mode
// (hj_stream.hip: the host-memory paths drive the same partition / plan / join steps)
// defer != nullptr: a relation that takes the plain histogram-free passes is prepared (buffers, state) but NOT launched — the two
// launches come back in *defer (used = true) and the caller enqueues them, e.g. merged with the other relation's.
// assume_clean: the caller vouches for stream order with the last read of the flags (partition_both): no k_set_root when the host
// knows the relation's device flag to be 0
struct FastPair { hj::FastArgs fa, fb; bool used = false; };
int partition_rel(hj_ctx *c, int r, FastPair *defer = nullptr, bool assume_clean = false);
int resolve_layout(hj_ctx *c, hj_ctx::Rel &R);
int plan_join(hj_ctx *c, hj::JoinArgs &a_out, bool &tag16, bool gen_ok = true, bool keep_cursor = false);

// RAII: HIP events on a stream around one kernel launch (per-kernel statistics, hj_timings)
struct Timed {
    hj_ctx *c;
    hj_ctx::Stamp s;
    bool on;
    hipStream_t st;
    static bool is_main(const char *n);
    Timed(hj_ctx *ctx, const char *name, hipStream_t stream = nullptr, bool use_given = false);
    ~Timed();
};

} // namespace hjx

#define HIPCHK(c, call)                                                                              \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess)                                                                        \
            return hjx::fail(c, e__ == hipErrorOutOfMemory ? HJ_ENOMEM : HJ_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, \
                             hipGetErrorString(e__));                                                 \
    } while (0)

#define RET(x)                                                                                        \
    do {                                                                                              \
        int r__ = (x);                                                                                \
        if (r__) return r__;                                                                          \
    } while (0)

#endif
