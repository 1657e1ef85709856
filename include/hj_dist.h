/*
 * hj_dist.h — C ABI of the multi-GPU join (libhj.so): level-0 shard split -> all-to-all over xGMI (RCCL) -> local
 * radix passes + build/probe on every GPU -> all-reduce of the count.
 *
 * The reference is single-GPU.  Its structural analogue is the co-processing path, which splits both relations 16 ways
 * on the host and joins every level-0 partition independently (src/hash_join_clustered_probe.cu:1256-1266, 1503-1618);
 * here the level-0 split runs on every GPU (owner = hash of the key), the shards cross the links, and each GPU joins what
 * it owns.  BASELINE.json north_star: "C++ host code ... through a thin C-ABI ... RCCL all-to-all over xGMI".
 *
 * Two ways to drive it:
 *   hj_dist       one process, G ranks: one hj context + one host thread per rank, RCCL communicators from
 *                 ncclCommInitAll (what `bench --gpus N` uses);
 *   hj_dist_rank  one process per GPU (torchrun / mpirun): rank 0 makes a 128-byte id (hj_dist_unique_id), hands it to
 *                 the others by whatever means the launcher offers, every rank calls hj_dist_rank_create
 *                 (ncclCommInitRank) (what bench.py --gpus N uses).
 *
 * The pipeline (hj_dist.hip): every relation is cut into K slices.  Slice i is split by shard with the histogram-free
 * pass (fixed-capacity slots: shard g's slots form one contiguous, fixed-size region), the G regions leave as ONE group
 * of ncclSend/ncclRecv per slice — message sizes depend only on (n, G, K), so no count ever has to come back to the
 * host — and the receiver runs its local pass 1 over the slots it got, as segments.  split(i+1) || exchange(i) ||
 * pass-1(i-1) overlap on two streams.  Skew that overflows a slot raises a flag that is all-gathered; every rank then
 * takes the exact path (exact split, counts read by the host, exact messages).
 */
#ifndef HJ_DIST_H_
#define HJ_DIST_H_

#include <stddef.h>
#include <stdint.h>

#include "hj.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hj_dist hj_dist;
typedef struct hj_dist_rank hj_dist_rank;

typedef struct hj_dist_config {
    uint32_t slices;        /* K slices per relation; 0 = 4 (fewer when a slice would be smaller than 2^16 tuples) */
    uint32_t exact_only;    /* 1: always the exact-count exchange (what skewed inputs fall back to) */
    uint32_t self_via_link; /* 1 (tests): a rank's own share also travels through ncclSend/ncclRecv instead of a device copy */
    uint32_t phantom_world; /* measurement on a one-GPU box (world size 1 only): run the sliced pipeline in the SHAPE of a
                             * phantom_world-GPU job — split fan-out, region sizes, segments per local pass-1 workgroup — with
                             * every region copied back locally instead of crossing a link (the rank then owns every shard, the
                             * result is the full join).  The stage times in hj_dist_stats are then those of one rank of such a
                             * job; bench.py models the link time beside them. */
    uint32_t single_group;  /* 1 (A/B): one pass 2 + one join per relation after its last slice, instead of joining the probe side's
                             * slices in two groups (all but the last under the exchange, the last in the tail) */
    uint32_t balance_size;  /* 1: size-aware shard assignment on the exact path even when no skew has been seen (8 virtual shards per
                             * GPU, dealt longest-first by |R|+|S|).  It is switched on by itself once a join on the same columns
                             * overflowed a slot somewhere. */
    uint32_t timeout_ms;    /* deadline of every wait on a collective (all-gather of the sizes, the drained exchange pipeline, the
                             * all-reduce of the result, ncclCommInitRank); 0 = $HJ_DIST_TIMEOUT_S seconds, or 120 s.  On expiry the call
                             * returns HJ_EHIP and hj_dist(_rank)_error names the rank, the stage, the slices whose exchange has not
                             * completed and the peers a message is owed by; the group is unusable afterwards (create a new one). */
    uint32_t reserved;      /* ignored (round 4 kept a test-only fault-injection switch here; it is the environment variable
                             * HJ_DIST_TEST_STALL_RANK now, read by the library's test hook only) */
} hj_dist_config;

typedef struct hj_dist_stats {
    uint64_t received[2];      /* tuples of R and of S this rank owned after the exchange */
    uint64_t link_bytes;       /* bytes this rank sent to OTHER ranks in the last join (padding of the fixed-size regions included) */
    uint64_t payload_bytes;    /* ... of which tuples (8 bytes each) */
    uint32_t path;             /* 0 = sliced fixed-size exchange, 1 = exact-count exchange */
    uint32_t slices, spans_per_slice, slot_capacity[2];
    /* device time of the local stages of the last join (HIP events), ms: what the links have to hide */
    float split_ms[2];         /* level-0 split of R, S: sum over the slices */
    float pass1_ms[2];         /* local pass 1 over the received slots: sum over the slices */
    float pass2_join_ms;       /* the tail: pass 2 + build/probe of the probe side's LAST group of slices (nothing left to overlap) */
    float first_split_ms;      /* split of the first slice (nothing to overlap it with) */
    float last_pass1_ms;       /* pass 1 of the last slice (the links are idle by then) */
    float wall_ms;             /* the whole hj_dist(_rank)_join call on this rank */
    float early_pass2_join_ms; /* pass 2 of the build side + pass 2 and build/probe of the probe side's earlier slices (under the exchange) */
    uint32_t probe_groups;
    uint32_t balanced;         /* exact path: 1 if the shards were assigned to GPUs by size */
    float exchange_ms;         /* sliced path: device time on the communication stream from the start of the first slice's exchange to
                                * the end of the last one's (HIP events): link_bytes / (world - 1) / exchange_ms = the rate ONE link
                                * direction sustained, to be held against its 76.8 GB/s */
    uint32_t materializing;    /* 1: the last join was a materialising one */
    uint64_t materialized;     /* ... and this rank wrote this many (key, payR, payS) tuples (more than its capacity: HJ_ECAPACITY) */
    uint32_t reserved[1];
} hj_dist_stats;

/* ---- one process, G ranks ---- */
/* devices[r] = HIP device of rank r (NULL: rank r on device r).  Distinct devices: RCCL.  Ranks that share a device (RCCL
 * refuses duplicate GPUs; test mode on a one-GPU box): the same pipeline over an in-process device-copy transport.
 * HJ_EINVAL if nranks < 1 or a device is not visible (`bench --gpus N` on a box with fewer than N GPUs fails here). */
int hj_dist_create(hj_dist **out, int nranks, const int *devices);
/* The same with the transport chosen by the caller: "rccl" (distinct devices only), "device-copy" = "copy" (ranks pull each other's
 * regions with hipMemcpyAsync / hipMemcpyPeerAsync: the copy engines instead of RCCL's kernels — no CU, no LDS taken from the
 * 155-KiB-LDS pass workgroups that run beside the exchange; peer access is enabled between distinct devices), or NULL / "" / "auto"
 * = $HJ_DIST_TRANSPORT, else rccl for distinct devices and device-copy for shared ones (what hj_dist_create does). */
int hj_dist_create_transport(hj_dist **out, int nranks, const int *devices, const char *transport);
/* The same group over another transport (contexts, bound columns and every buffer stay): a multi-GPU node times one workload over
 * RCCL's kernels and over the copy engines back to back with it.  HJ_EINVAL for a name the devices do not allow (the group keeps
 * the links it had). */
int hj_dist_set_transport(hj_dist *d, const char *transport);
int hj_dist_destroy(hj_dist *d);
const char *hj_dist_error(const hj_dist *d);
int hj_dist_world(const hj_dist *d);
const char *hj_dist_transport(const hj_dist *d);                /* "rccl" or "device-copy" */
hj_ctx *hj_dist_context(hj_dist *d, int rank);                  /* the rank's context: load / generate its slices with it */
int hj_dist_configure(hj_dist *d, const hj_dist_config *cfg);
int hj_dist_bind(hj_dist *d, int rank, int rel, const int32_t *d_keys, const int32_t *d_pays, uint64_t n);
/* The sharded join of what the ranks have bound: global match count and sum payR*payS mod 2^64.  [sync] */
int hj_dist_join(hj_dist *d, uint64_t *matches, uint64_t *agg);
/* The sharded MATERIALISING join (north_star: "join output = match count and materialised (key,payloadR,payloadS) tuples"; SURVEY
 * §8(e) "Output: stays sharded (each GPU materialises its partitions' results); global count via all-reduce"; the reference writes
 * its output per level-0 partition in the co-processing analogue, src/hash_join_clustered_probe.cu:1503-1618 with
 * join_partitioned_results src/join-primitives.cu:1107-1416).  Every rank writes the output tuples of the partitions it owns into
 * ITS OWN device columns (hj_dist_bind_output: caller-owned, on the rank's device, cap tuples each; gap-free [0, n_out[rank]),
 * order unspecified — the semantics of hj_join_materialize).  The union over the ranks is the join result; no tuple crosses a link
 * a second time.  *matches = global count (all-reduced), *agg = global sum payR*payS mod 2^64 (NULL: not computed — it costs a
 * pass over the output), n_out[world] = tuples each rank produced (all-gathered; may be NULL).  HJ_ECAPACITY when some rank's
 * output did not fit its columns (nothing beyond a capacity is written; counts and n_out are still returned; the group stays
 * usable).  [sync] */
int hj_dist_bind_output(hj_dist *d, int rank, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap);
int hj_dist_join_materialize(hj_dist *d, uint64_t *matches, uint64_t *agg, uint64_t *n_out);
int hj_dist_get_stats(hj_dist *d, int rank, hj_dist_stats *out);

/* ---- one process per GPU ---- */
#define HJ_DIST_ID_BYTES 128
int hj_dist_unique_id(void *id128);                              /* ncclGetUniqueId; rank 0 only */
int hj_dist_rank_create(hj_dist_rank **out, hj_ctx *ctx, int rank, int world, const void *id128); /* ncclCommInitRank: collective */
int hj_dist_rank_destroy(hj_dist_rank *r);
const char *hj_dist_rank_error(const hj_dist_rank *r);
int hj_dist_rank_configure(hj_dist_rank *r, const hj_dist_config *cfg);
/* Collective: every rank calls it with its local slices (device columns on the context's GPU).  [sync] */
int hj_dist_rank_join(hj_dist_rank *r, const int32_t *d_Rk, const int32_t *d_Rp, uint64_t nR, const int32_t *d_Sk,
                      const int32_t *d_Sp, uint64_t nS, uint64_t *matches, uint64_t *agg);
/* Collective, materialising (see hj_dist_join_materialize): this rank's output goes to its own columns (cap tuples each);
 * *n_out = tuples this rank produced, n_out_all[world] = every rank's (may be NULL), *matches / *agg global (agg NULL: not computed).
 * HJ_ECAPACITY on EVERY rank when some rank's output did not fit — the group stays usable (re-bind larger columns and call again); *agg
 * then covers only the tuples that fitted.  HJ_EINVAL on EVERY rank when some rank's arguments were bad (NULL or unaligned columns: the
 * verdict travels with the first all-gather, nothing is exchanged, the group stays usable).  Any other error aborts the group.  [sync] */
int hj_dist_rank_join_materialize(hj_dist_rank *r, const int32_t *d_Rk, const int32_t *d_Rp, uint64_t nR, const int32_t *d_Sk,
                                  const int32_t *d_Sp, uint64_t nS, int32_t *d_out_key, int32_t *d_out_payR, int32_t *d_out_payS,
                                  uint64_t cap, uint64_t *n_out, uint64_t *n_out_all, uint64_t *matches, uint64_t *agg);
int hj_dist_rank_get_stats(hj_dist_rank *r, hj_dist_stats *out);

/* tests only: rank `rank` (>= 0) of every group of this process stops taking part in its next probe-side exchange for 2.5 deadlines —
 * a stalled peer (the deadline tests); -1 = nobody (the default).  Until round 5 an environment variable read inside every exchange. */
int hj_dist_debug_stall_rank(int rank);

#ifdef __cplusplus
}
#endif
#endif /* HJ_DIST_H_ */
