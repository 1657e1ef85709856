/* A plain-C client of libhj.so: compiled with gcc against include/hj.h only (no HIP headers, no C++),
 * the way a maintainer of the reference (or any FFI) would bind the join path.  Exit code 0 = all good;
 * without a GPU it checks that the library fails loudly (hj_create != 0) and exits 77. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "hj.h"
#include "hj_reference_abi.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, hj_error(ctx)); return 1; } } while (0)

int main(void) {
    hj_ctx *ctx = NULL;
    if (hj_create(&ctx, 0) != HJ_OK) {
        printf("no GPU: hj_create failed loudly, as it must\n");
        return 77;
    }
    /* R = 0..n-1 shuffled by a fixed LCG, S = every key of R three times plus misses */
    const uint64_t nR = 50000, nS = 3 * nR + 1000;
    int32_t *R = malloc(nR * 4), *S = malloc(nS * 4), *Pr = malloc(nR * 4);
    for (uint64_t i = 0; i < nR; i++) { R[i] = (int32_t)((i * 7919u) % nR); Pr[i] = (int32_t)i - 17; }
    for (uint64_t i = 0; i < nS; i++) S[i] = i < 3 * nR ? (int32_t)(i % nR) : (int32_t)(nR + i);
    CHECK(hj_load_host(ctx, HJ_REL_R, R, Pr, nR, HJ_PAYLOAD_GIVEN));
    CHECK(hj_load_host(ctx, HJ_REL_S, S, NULL, nS, HJ_PAYLOAD_ONES));
    uint64_t matches = 0, agg = 0, expect_agg = 0;
    CHECK(hj_join(ctx, &matches, &agg));
    for (uint64_t i = 0; i < nR; i++) expect_agg += (uint64_t)(3 * (int64_t)Pr[i]); /* 7919 is coprime to nR: every key once */
    if (matches != 3 * nR || agg != expect_agg) { fprintf(stderr, "count %llu agg %llu\n", (unsigned long long)matches, (unsigned long long)agg); return 2; }
    /* materialise into library-allocated device columns and copy back */
    void *dk, *dr, *ds;
    CHECK(hj_device_malloc(ctx, &dk, matches * 4));
    CHECK(hj_device_malloc(ctx, &dr, matches * 4));
    CHECK(hj_device_malloc(ctx, &ds, matches * 4));
    uint64_t nout = 0;
    CHECK(hj_join_materialize(ctx, dk, dr, ds, matches, &nout));
    int32_t *k = malloc(nout * 4), *pr = malloc(nout * 4), *ps = malloc(nout * 4);
    CHECK(hj_memcpy_d2h(ctx, k, dk, nout * 4));
    CHECK(hj_memcpy_d2h(ctx, pr, dr, nout * 4));
    CHECK(hj_memcpy_d2h(ctx, ps, ds, nout * 4));
    if (nout != matches) return 3;
    for (uint64_t i = 0; i < nout; i++)   /* every output tuple: payR belongs to that key, payS = 1 */
        if (ps[i] != 1 || R[pr[i] + 17] != k[i]) { fprintf(stderr, "bad tuple %llu\n", (unsigned long long)i); return 4; }
    CHECK(hj_device_free(ctx, dk)); CHECK(hj_device_free(ctx, dr)); CHECK(hj_device_free(ctx, ds));
    /* the reference's own entry point, called the way main.cu does */
    args a = {0};
    a.R = R; a.R_els = nR; a.S = S; a.S_els = nS; a.threadsNum = 32; a.sharedMem = 30 << 10; a.pivotsNum = 1;
    timingInfo t = {0};
    t.n = 5;
    if (hashJoinClusteredProbe(&a, &t) != 0) return 5;
    hj_last_result res;
    hj_reference_last_result(&res);
    if (res.status != 0 || res.matches != 3 * nR) return 6;
    CHECK(hj_destroy(ctx));
    printf("c_abi_client ok: %llu matches\n", (unsigned long long)matches);
    return 0;
}
