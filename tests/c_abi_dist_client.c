/* A plain-C client of the multi-GPU ABI (include/hj_dist.h): gcc, C99, no HIP headers.  Three ranks of one process on device 0 (the
 * device-copy transport: what a one-GPU box allows; on a node with three GPUs pass "rccl" and devices {0,1,2}), every rank holding a
 * contiguous third of R and S; the count-only join, then the MATERIALISING join into per-rank device columns, checked against closed
 * forms and, tuple by tuple, against the relations.  Exit code 0 = all good; 77 = no GPU (the library failed loudly). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "hj.h"
#include "hj_dist.h"

#define G 3
#define DCHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, d ? hj_dist_error(d) : "no group"); return 1; } } while (0)
#define CCHECK(c, x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, hj_error(c)); return 1; } } while (0)

int main(void) {
    hj_dist *d = NULL;
    const int devices[G] = {0, 0, 0};
    if (hj_dist_create_transport(&d, G, devices, "device-copy") != HJ_OK) {
        printf("no GPU: hj_dist_create_transport failed loudly, as it must\n");
        return 77;
    }
    /* R = a permutation of 0..nR-1 with payload = key - 17; S = every key of R twice (payload = row id) plus misses */
    const uint64_t nR = 90000, nS = 2 * nR + 3000;
    int32_t *R = malloc(nR * 4), *Pr = malloc(nR * 4), *S = malloc(nS * 4), *Ps = malloc(nS * 4);
    for (uint64_t i = 0; i < nR; i++) { R[i] = (int32_t)((i * 7919u) % nR); Pr[i] = R[i] - 17; }
    for (uint64_t i = 0; i < nS; i++) { S[i] = i < 2 * nR ? (int32_t)((i * 31u) % nR) : (int32_t)(nR + i); Ps[i] = (int32_t)i; }
    hj_dist_config cfg = {0};
    cfg.slices = 2;
    DCHECK(hj_dist_configure(d, &cfg));
    void *dev[G][7];
    const uint64_t cap = 2 * nR; /* a rank could hold everything */
    for (int r = 0; r < G; r++) {
        hj_ctx *c = hj_dist_context(d, r);
        CCHECK(c, hj_configure(c, &(hj_config){.bits1 = 5, .bits2 = 4})); /* two-pass radix bits at this size: the sliced path applies */
        const uint64_t r0 = nR * r / G, r1 = nR * (r + 1) / G, s0 = nS * r / G, s1 = nS * (r + 1) / G;
        const uint64_t bytes[7] = {(r1 - r0) * 4, (r1 - r0) * 4, (s1 - s0) * 4, (s1 - s0) * 4, cap * 4, cap * 4, cap * 4};
        const void *src[4] = {R + r0, Pr + r0, S + s0, Ps + s0};
        for (int j = 0; j < 7; j++) {
            CCHECK(c, hj_device_malloc(c, &dev[r][j], bytes[j] + 64));
            if (j < 4) CCHECK(c, hj_memcpy_h2d(c, dev[r][j], src[j], bytes[j]));
        }
        DCHECK(hj_dist_bind(d, r, HJ_REL_R, dev[r][0], dev[r][1], r1 - r0));
        DCHECK(hj_dist_bind(d, r, HJ_REL_S, dev[r][2], dev[r][3], s1 - s0));
        DCHECK(hj_dist_bind_output(d, r, dev[r][4], dev[r][5], dev[r][6], cap));
    }
    /* closed forms: 31 is coprime to nR, so i -> 31 i mod nR visits every key exactly twice over i < 2 nR */
    uint64_t expect_agg = 0;
    for (uint64_t i = 0; i < 2 * nR; i++) expect_agg += (uint64_t)((int64_t)(S[i] - 17) * (int64_t)Ps[i]);
    uint64_t m = 0, agg = 0, n_out[G];
    DCHECK(hj_dist_join(d, &m, &agg));
    if (m != 2 * nR || agg != expect_agg) { fprintf(stderr, "count-only: %llu matches, agg %llu\n", (unsigned long long)m, (unsigned long long)agg); return 2; }
    DCHECK(hj_dist_join_materialize(d, &m, &agg, n_out));
    if (m != 2 * nR || agg != expect_agg || n_out[0] + n_out[1] + n_out[2] != m) { fprintf(stderr, "materialising: %llu matches\n", (unsigned long long)m); return 3; }
    /* every output tuple names its two source rows: key == S[payS], payR == key - 17; every S row below 2 nR exactly once */
    unsigned char *seen = calloc(2 * nR, 1);
    for (int r = 0; r < G; r++) {
        hj_ctx *c = hj_dist_context(d, r);
        int32_t *k = malloc(n_out[r] * 4 + 4), *pr = malloc(n_out[r] * 4 + 4), *ps = malloc(n_out[r] * 4 + 4);
        CCHECK(c, hj_memcpy_d2h(c, k, dev[r][4], n_out[r] * 4));
        CCHECK(c, hj_memcpy_d2h(c, pr, dev[r][5], n_out[r] * 4));
        CCHECK(c, hj_memcpy_d2h(c, ps, dev[r][6], n_out[r] * 4));
        for (uint64_t i = 0; i < n_out[r]; i++) {
            if (ps[i] < 0 || (uint64_t)ps[i] >= 2 * nR || S[ps[i]] != k[i] || pr[i] != k[i] - 17 || seen[ps[i]]++) { fprintf(stderr, "rank %d: bad tuple %llu\n", r, (unsigned long long)i); return 4; }
            if (hj_shard_of(k[i], G) != (uint32_t)r) { fprintf(stderr, "rank %d holds a key of shard %u\n", r, hj_shard_of(k[i], G)); return 5; }
        }
        free(k); free(pr); free(ps);
    }
    hj_dist_stats st;
    DCHECK(hj_dist_get_stats(d, 0, &st));
    printf("c_abi_dist_client ok: %llu matches over %d ranks (%s transport), shares %llu + %llu + %llu\n", (unsigned long long)m, hj_dist_world(d),
           hj_dist_transport(d), (unsigned long long)n_out[0], (unsigned long long)n_out[1], (unsigned long long)n_out[2]);
    for (int r = 0; r < G; r++) for (int j = 0; j < 7; j++) (void)hj_device_free(hj_dist_context(d, r), dev[r][j]);
    DCHECK(hj_dist_destroy(d));
    return 0;
}
