/*
 * hj_reference_abi.h — the reference's own call boundary for the join path, re-declared so that a
 * maintainer can link the reference driver (src/main.cu) against libhj.so unchanged.
 *
 *   joinAlg table entry  {"HJC", hashJoinClusteredProbe}          src/main.cu:33-36,64-66
 *   forward declaration  unsigned int hashJoinClusteredProbe(args*, timingInfo*)   src/main.cu:31
 *   argument block       struct args                               src/common-host.h:39-52
 *   timer block          struct timingInfo (n = 5 timeval pairs + counters)        src/common.h:101-119
 *
 * Only the memory layout matters here (POD, C linkage); field meanings are the reference's.
 */
#ifndef HJ_REFERENCE_ABI_H_
#define HJ_REFERENCE_ABI_H_

#include <stddef.h>
#include <sys/time.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/common-host.h:39-52 — key columns only; payloads are synthesised by the callee (hjcp.cu:1991-1999) */
typedef struct args {
    int *S;              /* FK / probe-or-build relation keys (host) */
    size_t S_els;
    char S_filename[50];
    int *R;              /* PK relation keys (host) */
    size_t R_els;
    char R_filename[50];
    int threadsNum;         /* ignored by HJC (main.cu:265-268) */
    unsigned int sharedMem; /* ignored by HJC */
    unsigned int pivotsNum; /* ignored by HJC */
} args;

/* src/common.h:101-119 — layout only; HJC stamps start/end[n-2] (hjcp.cu:2069-2070) */
typedef struct timingInfo {
    unsigned int n; /* = 5 */
    struct timeval start[5];
    struct timeval end[5];
    double greaterTime;
    double reduce_usecs;
    double fixPositions_usecs;
    double scatter_usecs;
    double copy_usecs;
    double bitonic_usecs;
    double total_usecs;
    double greaterEventTime;
    unsigned int greaterCallsNum;
    unsigned int bitonicCallsNum;
    unsigned int reduceCallsNum;
    unsigned int fixPositionsCallsNum;
} timingInfo;

/* Drop-in for src/hash_join_clustered_probe.cu:2062-2073.  Runs the in-GPU path twice (with
 * materialisation, then count-only) on device 0 and prints the reference's transcript
 * (hjcp.cu:937-940,986-991).  Returns 0 like the reference (hjcp.cu:2010,2072). */
unsigned int hashJoinClusteredProbe(args *inputAttrs, timingInfo *time);

/* Result of the most recent hashJoinClusteredProbe call, for callers that want more than stdout. */
typedef struct hj_last_result {
    unsigned long long matches;          /* |R ⋈ S| */
    unsigned long long agg;              /* sum payR*payS mod 2^64 */
    unsigned long long materialized;     /* tuples written by the materialising run */
    double partition_ms[2], join_ms[2];  /* [0] with materialisation, [1] without */
    int status;                          /* 0 or a negative HJ_E* code */
} hj_last_result;
void hj_reference_last_result(hj_last_result *out);

#ifdef __cplusplus
}
#endif
#endif
