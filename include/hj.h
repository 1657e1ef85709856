/*
 * hj.h — C ABI of libhj.so: the MI355X-native radix-partitioned hash-join path.
 *
 * This is the drop-in boundary for the in-GPU join path of psiul/ICDE2019-GPU-Join.  The reference
 * has no FFI layer: the join sits behind one C function pointer selected by `-a HJC`
 *     unsigned int hashJoinClusteredProbe(args*, timingInfo*)       src/main.cu:31,33-36,64-66,291
 * which this library exports with the same POD argument block (hj_reference_abi.h).  Underneath it
 * the explicit entry points below replace, one for one, the host-side steps of
 * outOfGPU_Join1_payload (src/hash_join_clustered_probe.cu:802-994).  Plain pointers and sizes
 * only; every function returns 0 on success or a negative HJ_E* code (the library never exit()s —
 * the reference's CHK_ERROR print-and-exit, src/common.h:132-141, is left to the `bench` wrapper).
 *
 * Threading: one caller thread per context, one context per GPU (src/main.cu:93 cudaSetDevice).
 * All work is enqueued on the context's HIP stream; only the functions marked [sync] wait for it.
 */
#ifndef HJ_H_
#define HJ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hj_ctx hj_ctx;

enum { HJ_REL_R = 0, HJ_REL_S = 1 };

/* payload columns: the reference synthesises all-ones payloads (hjcp.cu:1994-1999); row ids are
 * what its init_payload kernel writes (jp.cu:30-33). */
enum { HJ_PAYLOAD_ONES = 0, HJ_PAYLOAD_ROWID = 1, HJ_PAYLOAD_GIVEN = 2 };

enum {
    HJ_OK = 0,
    HJ_EINVAL = -1,    /* bad argument / call order */
    HJ_EHIP = -2,      /* a HIP runtime call failed (message in hj_error) */
    HJ_ENOMEM = -3,    /* device or host allocation failed */
    HJ_ECAPACITY = -4, /* materialised output does not fit the caller's buffers */
    HJ_EIO = -5        /* relation file missing / short */
};

/* Tunables.  Zero-initialise for the defaults (radix bits derived from the relation sizes so that
 * a build partition fits the LDS hash table; replaces the compile-time log_parts1/log_parts2 of
 * src/common.h:51-52). */
typedef struct hj_config {
    uint32_t bits1;        /* radix bits of pass 1 (key bits [bits2, bits2+bits1)); 0 = auto (total from the smaller relation's size; 7 bits first
                              for two relations of similar size up to 16 bits in all, 9 otherwise) */
    uint32_t bits2;        /* radix bits of pass 2 (key bits [0, bits2)); 0 with bits1!=0 = single pass */
    uint32_t force_bits;   /* 1: take bits1/bits2 literally (bits1=bits2=0 → no partitioning) */
    uint32_t build_side;   /* 0 auto (smaller relation, ties → R), 1 = R, 2 = S — a hint: bits follow the smaller relation, and a skewed or larger build relation lets the smaller PARTITION build (jp.cu:929-1003) */
    uint32_t lds_capacity; /* build tuples per LDS hash table; 0 = default */
    uint32_t lds_heads;    /* hash-table heads (power of two); 0 = default */
    uint32_t probe_chunk;  /* probe tuples per work item (decompose_chains threshold, hjcp.cu:904); 0 = default */
    uint32_t exact_only;   /* 1: always run the histogram + scan + scatter passes (gap-free partitions).  0: two-pass
                            * partitioning first tries the histogram-free passes (fixed-capacity slots per partition, the
                            * bump-allocated buckets of jp.cu:138-192 without their atomics) and falls back to the exact
                            * passes when skew overflows a slot (the overflow flag travels with the join's result block; a
                            * flagged relation is re-partitioned and the join re-run inside the same call). */
    uint32_t reserved1;    /* was materialize_two_pass (round 2's count + scan + second probe; hj_join_materialize writes its output in
                            * ONE probe since round 3, the streaming path since round 4: removed); must be 0 */
    uint32_t reserved0;    /* was lds_stage (round 3's LDS-staged one-probe kernel, measured 20 % slower, removed); must be 0 */
    uint32_t graph;        /* 1: hj_join replays the whole step (both partition passes, plan, build+probe, result copy) from a
                            * captured hipGraph — one host call per step instead of ~14-22 launches; pays below ~2^24 tuples,
                            * where a step is bound by the host's launch rate.  The first call on a binding runs eagerly, the second
                            * captures; re-binding, hj_configure, hj_set_stream or enabling kernel timings drop the graph. */
    uint32_t reserved[5];
} hj_config;

/* Per-kernel device time of the most recent run of each kernel (HIP events on the context stream).
 * Off by default (two event records per timed launch cost ~0.1 ms per join step, 20-40 % of a 2^22-2^24 step):
 * hj_enable_timings(ctx, 1) times the data-moving kernels (partition passes, join), 2 every launch; the environment
 * variable HJ_KERNEL_EVENTS=main|all|none sets the initial level.  Finished measurements are folded into the
 * statistics at every [sync] call and whenever 256 are pending, so the number of live HIP events is bounded whether
 * or not hj_timings is ever called. */
typedef struct hj_kernel_time {
    char name[32];
    uint32_t launches; /* launches since hj_timings_reset */
    float total_ms;    /* summed over those launches */
    float last_ms;
} hj_kernel_time;

/* ---- context ---- */
int hj_create(hj_ctx **out, int device);            /* hipSetDevice(device), stream, small workspace */
int hj_destroy(hj_ctx *ctx);                        /* frees every buffer the library allocated */
const char *hj_error(const hj_ctx *ctx);            /* message of the last failing call */
/* Run on a caller-owned hipStream_t.  NULL is HIP's default (null) stream — what
 * torch.cuda.current_stream().cuda_stream is unless the caller made its own; HJ_OWN_STREAM goes back
 * to the context's private stream. */
#define HJ_OWN_STREAM ((void *)(intptr_t)-1)
int hj_set_stream(hj_ctx *ctx, void *hip_stream);
int hj_configure(hj_ctx *ctx, const hj_config *cfg);
int hj_get_config(const hj_ctx *ctx, hj_config *cfg); /* the effective values after defaults/auto */
int hj_sync(hj_ctx *ctx);                           /* [sync] */

/* ---- relations (replaces hjcp.cu:815-879: device buffers + cudaMemcpy H2D) ---- */
/* Copy host columns into library-owned HBM.  pays may be NULL unless payload_mode == HJ_PAYLOAD_GIVEN.
 * The caller's columns are never modified or freed (hjcp.cu:874-877). [sync] */
int hj_load_host(hj_ctx *ctx, int rel, const int32_t *keys, const int32_t *pays, uint64_t n,
                 int payload_mode);
/* Use columns already resident in HBM (caller-owned, e.g. torch tensors); not modified. */
int hj_bind_device(hj_ctx *ctx, int rel, const int32_t *d_keys, const int32_t *d_pays, uint64_t n);

/* ---- the path (replaces prepare_Relation_payload jp.cu:1582-1613 + decompose_chains/join launches
 *      hjcp.cu:904-913,972-974) ---- */
/* Radix-partition one relation on the low key bits (async).  A partition is a range of the partitioned columns:
 * fixed-capacity slots from the histogram-free passes or gap-free ranges from the exact passes (hj_config.exact_only);
 * the join reads either, hj_get_partitions always hands out the gap-free form. */
int hj_partition(hj_ctx *ctx, int rel);
/* hj_partition(R) and hj_partition(S) in one call: S's passes are enqueued on a second stream of the context and run beside R's
 * (every pass kernel is one or two waves of one-per-CU workgroups: the other relation's kernel fills the CUs that a kernel's tail
 * leaves idle, -1...-5 % per step); the context's stream waits for both (async).  hj_join does this itself. */
int hj_partition_both(hj_ctx *ctx);
/* Build+probe every partition pair, count only (join_partitioned_aggregate jp.cu:885-1095).
 * matches = |R ⋈ S|; agg = sum payR*payS mod 2^64 (low 32 bits = the reference's int32 aggregate,
 * jp.cu:1073,1092).  Either pointer may be NULL.  [sync] */
int hj_join_count(hj_ctx *ctx, uint64_t *matches, uint64_t *agg);
/* Build+probe and write every (key,payR,payS) output tuple to the caller's HBM columns
 * (join_partitioned_results jp.cu:1107-1416, without its 2^24 FOLD ring: every tuple is kept), in the same probe that
 * finds the matches (the reference's lead timed run, hjcp.cu:913,937-940).  The output is gap-free [0, n_out), its order
 * unspecified.  cap = capacity of each output column in tuples; *n_out = tuples produced.  HJ_ECAPACITY if
 * n_out > cap (nothing beyond cap is written).  [sync] */
int hj_join_materialize(hj_ctx *ctx, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap,
                        uint64_t *n_out);
/* One call = hj_partition(R) + hj_partition(S) + hj_join_count (the timed region hjcp.cu:881-933). [sync]
 * Because the call hands out nothing but the result, a probe side known to be skewed takes the HEAVY-HITTER BYPASS (round 6): its
 * most frequent keys (up to 1024, from a sample taken once per binding) are looked up in LDS by pass 1, and a tuple whose key the
 * other relation holds exactly once is joined right there — counted, never partitioned (config 4: 39 % of S skips 32 of its 40 bytes).
 * Keys the other relation repeats or lacks take the ordinary path.  The reference's remedy for the overflow case is the role flip of
 * jp.cu:929-1003; SURVEY §7 step 5 names heavy-hitter handling. */
int hj_join(hj_ctx *ctx, uint64_t *matches, uint64_t *agg);
/* One call = hj_partition(R) + hj_partition(S) + hj_join_materialize: the reference's lead timed run (partition both, then
 * join_partitioned_results, hjcp.cu:881-913).  Same output contract as hj_join_materialize.  With the output columns known to the
 * partition passes, the heavy-hitter bypass WRITES the (key, payR, payS) tuples of the hot keys from pass 1 (one exact reservation on
 * the output cursor per workgroup and round); the probe appends the rest.  [sync] */
int hj_join_and_materialize(hj_ctx *ctx, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap, uint64_t *n_out);
/* The bypass as the last hj_join / hj_join_and_materialize used it: *mode 0 = not used, 1 = hot tuples counted, 2 = written; keys of the
 * table the other relation held exactly once when the table was planned; the sampled share of the probe relation's tuples they cover;
 * (mode 1) the matches pass 1 counted itself.  Any pointer may be NULL. */
int hj_hot_stats(const hj_ctx *ctx, int *mode, uint32_t *keys, double *share, uint64_t *matches);

/* ---- late materialisation / wide payloads (join_partitioned_varpayload jp.cu:1420-1557,
 *      outOfGPU_Join_payload_var hjcp.cu:542-708): the relations must be partitioned with ROW-ID payloads;
 *      on every match the ncolR values Dr[payR + z*strideR] and the ncolS values Ds[payS + z*strideS]
 *      (column z of a column-major table in HBM) are gathered and added up.  *sum = that total mod 2^64
 *      (low 32 bits = the reference's int32 aggregate, jp.cu:1526-1530).  [sync] ---- */
int hj_join_late_materialize(hj_ctx *ctx, const int32_t *d_Dr, uint32_t ncolR, uint64_t strideR,
                             const int32_t *d_Ds, uint32_t ncolS, uint64_t strideS, uint64_t *matches,
                             uint64_t *sum);

/* ---- non-partitioned baselines for comparison curves, on the loaded/bound (unpartitioned) relations:
 *      kind 0 = direct-address "perfect" array (build_/probe_perfect_array jp.cu:628-668; needs unique,
 *      non-negative build keys), kind 1 = one global chained table (build_ht_chains/chains_probing
 *      jp.cu:681-742).  The smaller relation builds.  [sync] ---- */
int hj_join_nonpartitioned(hj_ctx *ctx, int kind, uint64_t *matches, uint64_t *agg);

/* ---- streaming probe side (outOfGPU_Join3_payload, hjcp.cu:1684-1984): S stays in HOST memory and is
 *      streamed through HBM in segments — H2D copy of segment i+1 on a copy stream while segment i is
 *      partitioned and joined against R, which is partitioned once.  R must be loaded/bound first and always
 *      builds.  segment_tuples = 0 picks max(|R|/4, 2^24) (the reference uses |R|/4, hjcp.cu:1697-1698).
 *      Row-id payloads count from 0 over the whole of S.  Count-only.  [sync]
 *      Side effects: any S relation bound/loaded before the call is unbound afterwards (the staging buffers
 *      take its place); R's partitions are reused across calls while the radix bits stay the same, so re-bind
 *      R after modifying a bound column in place. ---- */
int hj_join_stream_probe(hj_ctx *ctx, const int32_t *h_keys, const int32_t *h_pays, uint64_t n,
                         uint64_t segment_tuples, int payload_mode, uint64_t *matches, uint64_t *agg);

/* The same with materialisation, as the reference's Join3 does it (join_partitioned_results per segment, then a copy of
 * the segment's output to the host on a third stream, hjcp.cu:1917-1961): every segment's (key,payR,payS) tuples are
 * written to double-buffered HBM columns and copied into the caller's HOST columns (pinned memory makes the copies
 * asynchronous) while the next segment is partitioned and joined.  cap = capacity of each host column in tuples;
 * *n_out = tuples produced; HJ_ECAPACITY if n_out > cap (nothing beyond cap is written).  [sync] */
int hj_join_stream_probe_materialize(hj_ctx *ctx, const int32_t *h_keys, const int32_t *h_pays, uint64_t n,
                                     uint64_t segment_tuples, int payload_mode, int32_t *h_out_key, int32_t *h_out_payR,
                                     int32_t *h_out_payS, uint64_t cap, uint64_t *n_out, uint64_t *agg);

/* ---- CPU-GPU co-processing (outOfGPU_Join2_payload, hjcp.cu:1000-1680): BOTH relations stay in host
 *      memory.  The host splits them into level0_parts partitions on host_threads threads (16 and 16 in the
 *      reference, hjcp.cu:1256-1266, pp.cuh:38-39; 0 = those defaults / all cores up to 64); each partition
 *      pair is uploaded into double-buffered staging while the previous pair is joined on the GPU.
 *      Payload columns may be NULL (= ones).  Count-only.  [sync] ---- */
int hj_join_coprocess(hj_ctx *ctx, const int32_t *h_R, const int32_t *h_Pr, uint64_t nR, const int32_t *h_S,
                      const int32_t *h_Ps, uint64_t nS, uint32_t level0_parts, uint32_t host_threads,
                      uint64_t *matches, uint64_t *agg);

/* The host level-0 split on its own (partitions_host_omp_nontemporal_payload, partition-primitives.cu:40-125, with
 * partition_prepare/do_payload :129-232): per-thread histograms, prefix, scatter through per-thread software
 * write-combining lines flushed with non-temporal AVX2 stores.  Partition of a key = hj_shard_of(key, parts).
 * offsets[parts+1]; out_pays may be NULL (keys only); pays may be NULL (= ones).  *gbs = bytes read + written per
 * second by the whole split (the reference prints its partition throughput, pp.cu:218).  No GPU involved. */
int hj_host_split(const int32_t *keys, const int32_t *pays, uint64_t n, uint32_t parts, uint32_t threads,
                  int32_t *out_keys, int32_t *out_pays, uint64_t *offsets, double *gbs);
/* The split hj_join_coprocess itself runs (round 5): ONE pass over the input, no histogram.  A partition comes out as a list of
 * blocks of the output columns — each worker takes fixed-size blocks from its own arena as its partitions fill up, the reference's
 * bucket-chain layout (join-primitives.cu:138-192) on the host — which is all the uploads need.  out_keys / out_pays hold cap >=
 * hj_host_split_blocks_capacity(n, parts, threads) tuples; out_pays is written only when pays is given.  Alignment: with out_keys and
 * out_pays 64-byte aligned whole lines leave with non-temporal AVX2 stores; any other alignment is accepted and takes plain stores (same
 * result, the destination lines are then read for ownership first: slower, never wrong).  Block i holds block_count[i] tuples of partition block_part[i] at out_*[block_start[i]..]; blocks come sorted by
 * (partition, start).  *n_blocks = blocks produced; HJ_ECAPACITY if cap or max_blocks is too small.  No GPU involved. */
uint64_t hj_host_split_blocks_capacity(uint64_t n, uint32_t parts, uint32_t threads);
int hj_host_split_blocks(const int32_t *keys, const int32_t *pays, uint64_t n, uint32_t parts, uint32_t threads,
                         int32_t *out_keys, int32_t *out_pays, uint64_t cap, uint32_t *block_part, uint64_t *block_start,
                         uint32_t *block_count, uint64_t max_blocks, uint64_t *n_blocks, double *gbs);
/* A CPU radix join out of the library's own host code — a REPORTED BASELINE (bench.py `cpu_baseline.best_effort`, bench --cpu-baseline),
 * never a fallback of the GPU path: both relations through hj_host_split_blocks' one-pass split (up to 4096 partitions), then one
 * partition pair per thread at a time: a counting sort into cache-sized pieces and a bucket-chained table per piece, built and probed
 * like the reference's joinCpu (hash_join_clustered_probe.cu:2013-2059).  Payload columns may be NULL (= ones); matches / agg as hj_join
 * defines them; *seconds = wall time without the allocation of the staging columns.  No GPU involved. */
int hj_host_join(const int32_t *keysR, const int32_t *paysR, uint64_t nR, const int32_t *keysS, const int32_t *paysS, uint64_t nS,
                 uint32_t threads, uint64_t *matches, uint64_t *agg, double *seconds);
/* tests only: the following hj_host_split_blocks calls copy out every range the split publishes while it runs and check afterwards that
 * it was final when published (HJ_EIO otherwise; *gbs then returns the tuples published).  0 = off (the default).  Until round 5 an
 * environment variable read inside the public entry point. */
int hj_host_split_debug_progress(int on);
/* NUMA placement of the last hj_join_coprocess call (partition-primitives.cu:129-253 keeps partitions and threads per
 * socket): NUMA nodes of the host, the node closest to the context's GPU (-1 unknown: pinned staging is allocated there by
 * hipHostMalloc), and how many of that node's CPUs the split's workers were bound to (0: not bound — one node, HJ_NUMA=0). */
int hj_coprocess_numa(const hj_ctx *ctx, int *nodes, int *gpu_node, int *pinned_cpus);
/* Residency groups of the last hj_join_coprocess call: runs of consecutive level-0 pairs that were uploaded and joined together (one
 * upload per column, one join per group) because their tuples fit the device-memory budget — what the reference's knapsack over
 * PARTS_RESIDENT slots decides (groupOptimal2, partition-primitives.cu:307-468; hjcp.cu:1357-1363).  1 on a card that holds everything. */
int hj_coprocess_groups(const hj_ctx *ctx, uint32_t *groups);
/* GB/s of the host split inside the last hj_join_coprocess call of this context. */
int hj_host_split_throughput(const hj_ctx *ctx, double *gbs);

/* ---- plain device-memory helpers for callers without a HIP runtime binding of their own ---- */
int hj_device_malloc(hj_ctx *ctx, void **d_ptr, uint64_t bytes);
int hj_device_free(hj_ctx *ctx, void *d_ptr);
int hj_memcpy_d2h(hj_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes); /* [sync] */
int hj_memcpy_h2d(hj_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes); /* [sync] */

/* ---- introspection for parity tests ---- */
/* Device pointers of the partitioned columns and of the nparts+1 partition offsets (uint64): partition p =
 * [offsets[p], offsets[p+1]).  When the histogram-free passes produced the partitions (fixed-capacity slots), this
 * returns a gap-free copy owned by the context.  [sync] */
int hj_get_partitions(hj_ctx *ctx, int rel, const int32_t **d_keys, const int32_t **d_pays,
                      const uint64_t **d_offsets, uint64_t *nparts);
/* *slotted = 1 if the relation's partitions came from the histogram-free passes, 2 if from the sampled path of a skewed
 * probe-side relation (histogram-free passes with capacities from a sample; a partition is then a list of ranges), 0 if from
 * the exact passes. [sync] */
int hj_partition_layout(hj_ctx *ctx, int rel, int *slotted);
/* Host-side breakdown of the most recent hj_join call, ms of wall time: device (re)allocations (hipFree + hipMalloc inside the
 * library, and how many), the optimistic histogram-free attempt that came back with an overflow flag (0 if none did), sampling pass +
 * host planning + table upload of the sampled path (0 unless this call planned one), and the whole call.  What a FIRST call on a
 * skewed binding spends where (the reference allocates outside its timed region too, hjcp.cu:815-879; a one-join-per-process CLI
 * user still waits for it). */
int hj_last_call_breakdown(const hj_ctx *ctx, double *alloc_ms, uint32_t *allocations, double *failed_attempt_ms,
                           double *sample_plan_ms, double *total_ms);
/* experiments only: read the environment knobs again.  The library reads them ONCE, in hj_create; no entry point of the
 * path calls getenv.  tools/experiments/ switch a knob between two calls of one context with this. */
int hj_reload_knobs(hj_ctx *ctx);
/* experiments only: per-workgroup timelines.  A library built with -DHJ_STAMPS (`make -C csrc stamps` -> libhj_stamps.so) writes
 * {start, table built, end, hardware id} (s_memrealtime ticks of 10 ns; HW_ID | XCC_ID << 32) per work item of the count kernel into
 * d_join and {start, 0, end, hardware id} per parent of pass 2 into d_part2 (4 x uint64 each; NULL = off).  The shipped build ignores
 * the pointers.  tools/experiments/fixed_cost.py */
int hj_debug_set_stamps(hj_ctx *ctx, void *d_join_stamps, void *d_part2_stamps);
int hj_enable_timings(hj_ctx *ctx, int level); /* 0 off, 1 data-moving kernels, 2 every launch.  [sync] */
int hj_timings_reset(hj_ctx *ctx);
/* [sync] fills up to cap entries, returns the number of kernels known in *n. */
int hj_timings(hj_ctx *ctx, hj_kernel_time *out, uint32_t cap, uint32_t *n);

/* ---- multi-GPU: level-0 shard split (the role of the 16-way host partition of the co-processing
 *      path, hjcp.cu:1256-1266,1380-1382, moved onto the GPU).  Splits (d_keys,d_pays) into nshards
 *      contiguous runs by a hash of the key; h_counts[nshards] receives the run lengths.  The runs
 *      are what an all-to-all over xGMI exchanges.  [sync] ---- */
int hj_shard_split(hj_ctx *ctx, const int32_t *d_keys, const int32_t *d_pays, uint64_t n,
                   uint32_t nshards, int32_t *d_out_keys, int32_t *d_out_pays, uint64_t *h_counts);
/* The same with more (virtual) shards than GPUs, for size-aware assignment under skew (the reference's knapsack idea,
 * partition-primitives.cu:307-468): hj_shard_count gives the tuples per shard of one column (no data movement);
 * the caller decides which GPU owns which shard and passes h_position[v] = output position of shard v (a
 * permutation of 0..nshards-1, e.g. shards ordered by owner), so that every owner's tuples form ONE contiguous
 * run.  h_counts[i] = tuples at output position i.  NULL h_position = identity.  [sync] */
int hj_shard_count(hj_ctx *ctx, const int32_t *d_keys, uint64_t n, uint32_t nshards, uint64_t *h_counts);
int hj_shard_split_ordered(hj_ctx *ctx, const int32_t *d_keys, const int32_t *d_pays, uint64_t n, uint32_t nshards,
                           const uint32_t *h_position, int32_t *d_out_keys, int32_t *d_out_pays, uint64_t *h_counts);
/* destination shard of a key (host-side mirror of the device function, for tests/oracles) */
uint32_t hj_shard_of(int32_t key, uint32_t nshards);

/* ---- device-side input synthesis and checks (perf-run inputs too large to shuffle on the host) ---- */
/* keys[i] = pi_seed(first + i) where pi_seed is a bijection on [0, domain): a duplicate-free slice
 * of a pseudo-random permutation of 0..domain-1 when first+n <= domain; beyond the domain it wraps
 * (first + i) mod domain. (async) */
int hj_gen_unique(hj_ctx *ctx, int32_t *d_keys, uint64_t n, uint64_t first, uint64_t domain, uint64_t seed);
/* keys[i] = 1 + pi_seed(rank_i - 1), rank_i ~ Zipf(theta) over 1..alphabet (values 1..alphabet, like
 * gen_zipf src/generator_ETHZ.cu:299-348; same head probabilities, not the same random stream). (async) */
int hj_gen_zipf(hj_ctx *ctx, int32_t *d_keys, uint64_t n, uint64_t first, uint64_t alphabet, double theta,
                uint64_t seed);
int hj_fill_payload(hj_ctx *ctx, int32_t *d_pays, uint64_t n, int payload_mode, uint64_t first_rowid);
/* order-independent 64-bit digests: sum of mix(key,pay) / mix(key,payR,payS) mod 2^64.  [sync] */
int hj_digest_pairs(hj_ctx *ctx, const int32_t *d_keys, const int32_t *d_pays, uint64_t n, uint64_t *digest);
int hj_digest_triples(hj_ctx *ctx, const int32_t *d_key, const int32_t *d_payR, const int32_t *d_payS,
                      uint64_t n, uint64_t *digest);
/* number of tuples of the partitioned relation that sit in a partition other than
 * (key >> 0) & (nparts-1), plus per-partition (key,pay) digests into d_digests[nparts] if non-NULL.  [sync] */
int hj_verify_partitions(hj_ctx *ctx, int rel, uint64_t *misplaced, uint64_t *d_digests);

/* ---- measurement: on-box HBM ceilings of the two access patterns of a radix pass, no partitioning work.
 *      kind 0 = stream copy of a (key, payload) column pair, 16 bytes per lane; kind 1 = same streaming reads, every
 *      128-byte line stored at a pseudo-random aligned line position (the write pattern of the write-combining
 *      flush); kind 2 = both input columns streamed in only, kind 3 = both output columns streamed out only (what HBM gives
 *      pure reads / pure writes: a kernel reading R and writing W bytes is bounded by (R + W) / (R / read + W / write)).
 *      Kinds 4-7 (the layout gate of round 6) move the same bytes as ONE array per side: d_in_p must be d_in_k + n and d_out_p
 *      d_out_k + n, n a multiple of 32.  4 = plain one-array copy; 5 = line-interleaved pairs (a 256-byte line pair: 32 keys, then
 *      their 32 payloads); 6 = kind 5 with every line pair stored at a pseudo-random line-pair position; 7 = two columns in,
 *      scattered line pairs out.
 *      avg_ms per launch over reps launches (HIP events), bytes moved per launch (read + written). [sync] ---- */
int hj_ubench(hj_ctx *ctx, int kind, const int32_t *d_in_k, const int32_t *d_in_p, int32_t *d_out_k, int32_t *d_out_p,
              uint64_t n, uint32_t reps, double *avg_ms, uint64_t *bytes_per_launch);

/* ---- generator_ETHZ drop-in (host side; src/generator_ETHZ.cuh:11-23) ----
 * Same generators, same raw-int32 .bin cache format; the time(NULL)/rand() global state of the
 * reference is replaced by an explicit seed (0 = take time(NULL) like the reference). */
void hj_gen_set_seed(uint64_t seed);
int hj_create_relation_unique(const char *filename, int32_t *relation, uint64_t num_tuples, int64_t maxid);
int hj_create_relation_nonunique(const char *filename, int32_t *relation, uint64_t num_tuples, int64_t maxid);
int hj_create_relation_zipf(const char *filename, int32_t *relation, uint64_t num_tuples, int64_t maxid,
                            double zipf_param);
int hj_create_relation_fk_from_pk(const char *filename, int32_t *fkrel, uint64_t fk_tuples,
                                  const int32_t *pkrel, uint64_t pk_tuples);
int hj_create_relation_n(const int32_t *in_relation, int32_t *out_relation, uint64_t num_tuples, uint64_t n);
int hj_read_relation(const char *filename, int32_t *relation, uint64_t num_tuples);
int hj_write_relation(const char *filename, const int32_t *relation, uint64_t num_tuples);

const char *hj_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HJ_H_ */
