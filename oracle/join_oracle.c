/*
 * join_oracle.c — ORACLE (test infrastructure, never shipped): CPU restatement of the join
 * semantics of the reference path and of its partition function.  See oracle.h.
 *
 * Reference semantics followed (paths relative to /root/reference/src):
 *   - partition id = (hasht(key) >> first_bit) & (parts-1), hasht = identity   common.h:45-47, jp.cu:126
 *   - inner equi-join on the int32 key, every (r,s) pair with equal keys is one output
 *     (duplicates multiply)                                                    jp.cu:1056-1079
 *   - aggregate = sum payR*payS (int32 wrap in the reference; kept mod 2^64 here, low 32 bits
 *     are the reference value)                                                 jp.cu:1073,1092
 *   - materialised record: (payR,payS) in the reference (jp.cu:1238-1239,1365-1366); the build
 *     adds the key column → (key,payR,payS)                                   SURVEY.md §8(c)
 *   - joinCpu cross-check                                                      hjcp.cu:2013-2059
 */
#include "oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef __AVX2__
#include <immintrin.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

int o_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline uint64_t o_fmix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

uint64_t o_mix_pair(int32_t key, int32_t pay) {
    return o_fmix64(((uint64_t)(uint32_t)key << 32) | (uint32_t)pay);
}

uint64_t o_mix_triple(int32_t key, int32_t pr, int32_t ps) {
    return o_fmix64(o_mix_pair(key, pr) ^ ((uint64_t)(uint32_t)ps * 0x9E3779B97F4A7C15ULL));
}

/* ---------- partition function ---------- */

void o_radix_partition(const int32_t *keys, const int32_t *pays, uint64_t n, uint32_t shift,
                       uint32_t bits, int32_t *out_keys, int32_t *out_pays, uint64_t *offsets) {
    uint64_t parts = 1ULL << bits;
    uint32_t mask = (uint32_t)(parts - 1);
    uint64_t *cur = (uint64_t *)calloc(parts + 1, sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) cur[(((uint32_t)keys[i]) >> shift) & mask]++;
    uint64_t sum = 0;
    for (uint64_t p = 0; p < parts; p++) {
        offsets[p] = sum;
        sum += cur[p];
        cur[p] = offsets[p];
    }
    offsets[parts] = sum;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t d = cur[(((uint32_t)keys[i]) >> shift) & mask]++;
        out_keys[d] = keys[i];
        if (pays) out_pays[d] = pays[i];
    }
    free(cur);
}

void o_partition_digest(const int32_t *keys, const int32_t *pays, const uint64_t *offsets,
                        uint64_t nparts, uint64_t *digest) {
    for (uint64_t p = 0; p < nparts; p++) {
        uint64_t d = 0;
        for (uint64_t i = offsets[p]; i < offsets[p + 1]; i++) d += o_mix_pair(keys[i], pays[i]);
        digest[p] = d;
    }
}

/* ---------- sort-merge join (independent of any hash table) ---------- */

/* pack (key,pay) as key<<32|pay (both as uint32 bit patterns) and LSD-radix-sort */
static uint64_t *o_sorted_pairs(const int32_t *K, const int32_t *P, uint64_t n) {
    uint64_t *a = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    uint64_t *b = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++)
        a[i] = ((uint64_t)(uint32_t)K[i] << 32) | (uint32_t)(P ? P[i] : 1);
    for (int pass = 0; pass < 8; pass++) {
        uint64_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        int sh = pass * 8;
        for (uint64_t i = 0; i < n; i++) cnt[((a[i] >> sh) & 0xFF) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (uint64_t i = 0; i < n; i++) b[cnt[(a[i] >> sh) & 0xFF]++] = a[i];
        uint64_t *t = a;
        a = b;
        b = t;
    }
    free(b);
    return a;
}

void o_join_count(const int32_t *R, const int32_t *Pr, uint64_t nR, const int32_t *S,
                  const int32_t *Ps, uint64_t nS, uint64_t *matches, uint64_t *agg,
                  uint64_t *checksum) {
    uint64_t *r = o_sorted_pairs(R, Pr, nR), *s = o_sorted_pairs(S, Ps, nS);
    uint64_t i = 0, j = 0, m = 0, a = 0, c = 0;
    while (i < nR && j < nS) {
        uint32_t kr = (uint32_t)(r[i] >> 32), ks = (uint32_t)(s[j] >> 32);
        if (kr < ks) {
            i++;
        } else if (kr > ks) {
            j++;
        } else {
            uint64_t i2 = i, j2 = j, sr = 0, ss = 0;
            while (i2 < nR && (uint32_t)(r[i2] >> 32) == kr) sr += (uint64_t)(int64_t)(int32_t)(uint32_t)r[i2++];
            while (j2 < nS && (uint32_t)(s[j2] >> 32) == kr) ss += (uint64_t)(int64_t)(int32_t)(uint32_t)s[j2++];
            m += (i2 - i) * (j2 - j);
            a += sr * ss;
            if (checksum)
                for (uint64_t x = i; x < i2; x++)
                    for (uint64_t y = j; y < j2; y++)
                        c += o_mix_triple((int32_t)kr, (int32_t)(uint32_t)r[x], (int32_t)(uint32_t)s[y]);
            i = i2;
            j = j2;
        }
    }
    free(r);
    free(s);
    if (matches) *matches = m;
    if (agg) *agg = a;
    if (checksum) *checksum = c;
}

uint64_t o_join_materialize(const int32_t *R, const int32_t *Pr, uint64_t nR, const int32_t *S,
                            const int32_t *Ps, uint64_t nS, int32_t *out_key, int32_t *out_pr,
                            int32_t *out_ps, uint64_t cap) {
    uint64_t *r = o_sorted_pairs(R, Pr, nR), *s = o_sorted_pairs(S, Ps, nS);
    uint64_t i = 0, j = 0, m = 0;
    while (i < nR && j < nS) {
        uint32_t kr = (uint32_t)(r[i] >> 32), ks = (uint32_t)(s[j] >> 32);
        if (kr < ks) {
            i++;
        } else if (kr > ks) {
            j++;
        } else {
            uint64_t i2 = i, j2 = j;
            while (i2 < nR && (uint32_t)(r[i2] >> 32) == kr) i2++;
            while (j2 < nS && (uint32_t)(s[j2] >> 32) == kr) j2++;
            /* r and s are sorted by (key,pay): emitting x-major, y-minor is (key,payR,payS) order */
            for (uint64_t x = i; x < i2; x++)
                for (uint64_t y = j; y < j2; y++) {
                    if (m < cap) {
                        out_key[m] = (int32_t)kr;
                        out_pr[m] = (int32_t)(uint32_t)r[x];
                        out_ps[m] = (int32_t)(uint32_t)s[y];
                    }
                    m++;
                }
            i = i2;
            j = j2;
        }
    }
    free(r);
    free(s);
    return m;
}

uint64_t o_triples_checksum(const int32_t *key, const int32_t *pr, const int32_t *ps, uint64_t n) {
    uint64_t c = 0;
    for (uint64_t i = 0; i < n; i++) c += o_mix_triple(key[i], pr[i], ps[i]);
    return c;
}

/* ---------- joinCpu restated: hjcp.cu:2013-2059 ---------- */

#define O_LOG_HTSIZE 20 /* hjcp.cu:2013 */

/* hjcp.cu:2016-2023 */
static inline uint32_t o_hashMurmur(uint32_t x) {
    x ^= x >> 16;
    x *= 0x85ebca6b;
    x ^= x >> 13;
    x *= 0xc2b2ae35;
    x ^= x >> 16;
    return x & ((1u << O_LOG_HTSIZE) - 1);
}

void o_joinCpu(const int32_t *R, uint64_t nR, const int32_t *S, uint64_t nS, int threads,
               uint64_t *s_out, uint32_t *g_out) {
    int32_t *ht = (int32_t *)malloc(sizeof(int32_t) << O_LOG_HTSIZE);
    int32_t *next = (int32_t *)malloc((nR ? nR : 1) * sizeof(int32_t));
    memset(ht, -1, sizeof(int32_t) << O_LOG_HTSIZE); /* hjcp.cu:2026 */
    for (uint64_t j = 0; j < nR; ++j) {               /* serial build, hjcp.cu:2035-2040 */
        uint32_t bucket = o_hashMurmur((uint32_t)R[j]);
        next[j] = ht[bucket];
        ht[bucket] = (int32_t)j;
    }
    uint64_t s = 0;
    uint32_t g = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : g) reduction(+ : s) num_threads(threads > 0 ? threads : 1)
#endif
    for (uint64_t j = 0; j < nS; ++j) {               /* probe, hjcp.cu:2044-2055 */
        uint32_t bucket = o_hashMurmur((uint32_t)S[j]);
        int32_t current = ht[bucket];
        while (current >= 0) {
            if (S[j] == R[current]) {
                g += (uint32_t)S[j];
                ++s;
            }
            current = next[current];
        }
    }
    free(ht);
    free(next);
    if (s_out) *s_out = s;
    if (g_out) *g_out = g;
}

/* ---------- CPU baseline: OpenMP two-pass radix join ("port" of the path's structure) ---------- */

typedef struct {
    int32_t *k, *p;
    uint64_t *off; /* 2^bits + 1 */
} o_parted;

/* pass over [0,n): per-thread histograms on digit (key>>shift)&mask → contiguous partitions.
 * The scatter goes through per-thread software write-combining buffers — one 64-byte line of keys and
 * one of payloads per partition — flushed as whole lines, with non-temporal AVX2 stores when the
 * destination is 32-byte aligned: the scheme of the reference's CPU partitioner
 * (partitions_host_omp_nontemporal_payload, partition-primitives.cu:40-125: per-partition buffers filled
 * tuple by tuple, _mm256_stream_si256 on flush; histogram + prefix in partition_prepare_payload :129-220). */
#define O_WC 16 /* tuples per write-combining line (64 bytes) */

static inline void o_flush_line(int32_t *dst, const int32_t *src, unsigned cnt) {
#ifdef __AVX2__
    if (cnt == O_WC && (((uintptr_t)dst) & 31) == 0) {
        _mm256_stream_si256((__m256i *)dst, _mm256_loadu_si256((const __m256i *)src));
        _mm256_stream_si256((__m256i *)(dst + 8), _mm256_loadu_si256((const __m256i *)(src + 8)));
        return;
    }
#endif
    memcpy(dst, src, cnt * sizeof(int32_t));
}

static void o_par_partition(const int32_t *K, const int32_t *P, uint64_t n, uint32_t shift,
                            uint32_t bits, int32_t *oK, int32_t *oP, uint64_t *off, int threads) {
    uint64_t parts = 1ULL << bits;
    uint32_t mask = (uint32_t)(parts - 1);
    uint64_t *hist = (uint64_t *)calloc((size_t)threads * parts, sizeof(uint64_t));
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        uint64_t lo = n * (uint64_t)t / threads, hi = n * (uint64_t)(t + 1) / threads;
        uint64_t *h = hist + (size_t)t * parts;
        for (uint64_t i = lo; i < hi; i++) h[(((uint32_t)K[i]) >> shift) & mask]++;
#ifdef _OPENMP
#pragma omp barrier
#pragma omp single
#endif
        {
            uint64_t sum = 0;
            for (uint64_t p = 0; p < parts; p++) {
                off[p] = sum;
                for (int tt = 0; tt < threads; tt++) {
                    uint64_t c = hist[(size_t)tt * parts + p];
                    hist[(size_t)tt * parts + p] = sum;
                    sum += c;
                }
            }
            off[parts] = sum;
        }
        /* write-combining buffers of this thread: [parts][O_WC] keys and payloads, and their fill */
        int32_t *bk = (int32_t *)aligned_alloc(64, parts * O_WC * sizeof(int32_t));
        int32_t *bp = (int32_t *)aligned_alloc(64, parts * O_WC * sizeof(int32_t));
        uint8_t *fill = (uint8_t *)calloc(parts, 1);
        /* the first line of a partition run is shortened so that later flushes land on 64-byte boundaries */
        for (uint64_t p = 0; p < parts; p++) fill[p] = (uint8_t)(h[p] & (O_WC - 1));
        for (uint64_t i = lo; i < hi; i++) {
            uint32_t d = (((uint32_t)K[i]) >> shift) & mask;
            unsigned f = fill[d];
            bk[d * O_WC + f] = K[i];
            bp[d * O_WC + f] = P ? P[i] : 1;
            if (++f == O_WC) {
                unsigned first = (unsigned)(h[d] & (O_WC - 1)); /* non-zero only for the run's first line */
                o_flush_line(oK + h[d], bk + d * O_WC + first, O_WC - first);
                o_flush_line(oP + h[d], bp + d * O_WC + first, O_WC - first);
                h[d] += O_WC - first;
                f = 0;
            }
            fill[d] = (uint8_t)f;
        }
        for (uint64_t p = 0; p < parts; p++) { /* tails */
            unsigned first = (unsigned)(h[p] & (O_WC - 1)), f = fill[p];
            if (f > first) {
                memcpy(oK + h[p], bk + p * O_WC + first, (f - first) * sizeof(int32_t));
                memcpy(oP + h[p], bp + p * O_WC + first, (f - first) * sizeof(int32_t));
                h[p] += f - first;
            }
        }
#ifdef __AVX2__
        _mm_sfence();
#endif
        free(bk);
        free(bp);
        free(fill);
    }
    free(hist);
}

static o_parted o_two_pass(const int32_t *K, const int32_t *P, uint64_t n, uint32_t bits1,
                           uint32_t bits2, int threads) {
    o_parted out;
    uint64_t parts1 = 1ULL << bits1, parts2 = 1ULL << bits2;
    int32_t *tK = (int32_t *)malloc((n ? n : 1) * 4), *tP = (int32_t *)malloc((n ? n : 1) * 4);
    out.k = (int32_t *)malloc((n ? n : 1) * 4);
    out.p = (int32_t *)malloc((n ? n : 1) * 4);
    out.off = (uint64_t *)malloc((parts1 * parts2 + 1) * sizeof(uint64_t));
    uint64_t *off1 = (uint64_t *)malloc((parts1 + 1) * sizeof(uint64_t));
    /* pass 1 on bits [bits2, bits2+bits1), pass 2 on bits [0,bits2): final id = low bits1+bits2 bits,
     * ordered pass-1 digit major (jp.cu:402: output partition (pid<<log_parts2)+j) */
    o_par_partition(K, P, n, bits2, bits1, tK, tP, off1, threads);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (uint64_t p1 = 0; p1 < parts1; p1++) {
        uint64_t lo = off1[p1], cnt = off1[p1 + 1] - lo;
        uint64_t *loc = (uint64_t *)malloc((parts2 + 1) * sizeof(uint64_t));
        o_radix_partition(tK + lo, tP + lo, cnt, 0, bits2, out.k + lo, out.p + lo, loc);
        for (uint64_t p2 = 0; p2 < parts2; p2++) out.off[p1 * parts2 + p2] = lo + loc[p2];
        free(loc);
    }
    out.off[parts1 * parts2] = n;
    free(tK);
    free(tP);
    free(off1);
    return out;
}

uint64_t o_radix_join_omp(const int32_t *R, const int32_t *Pr, uint64_t nR, const int32_t *S,
                          const int32_t *Ps, uint64_t nS, uint32_t bits1, uint32_t bits2,
                          int threads, uint64_t *agg_out) {
    if (threads < 1) threads = 1;
    o_parted r = o_two_pass(R, Pr, nR, bits1, bits2, threads);
    o_parted s = o_two_pass(S, Ps, nS, bits1, bits2, threads);
    uint64_t nparts = 1ULL << (bits1 + bits2), matches = 0, agg = 0;
    uint32_t rb = bits1 + bits2;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads) reduction(+ : matches) reduction(+ : agg)
#endif
    {
        uint64_t cap = 0;
        int32_t *head = NULL, *next = NULL;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (uint64_t p = 0; p < nparts; p++) {
            uint64_t r0 = r.off[p], rn = r.off[p + 1] - r0, s0 = s.off[p], sn = s.off[p + 1] - s0;
            if (!rn || !sn) continue;
            /* build on the R partition (chained table, LIFO insert as jp.cu:1021-1048) */
            uint64_t nh = 1;
            while (nh < rn) nh <<= 1;
            if (nh + rn > cap) {
                cap = 2 * (nh + rn);
                free(head);
                free(next);
                head = (int32_t *)malloc(cap * sizeof(int32_t));
                next = (int32_t *)malloc(cap * sizeof(int32_t));
            }
            for (uint64_t h = 0; h < nh; h++) head[h] = -1;
            for (uint64_t i = 0; i < rn; i++) {
                uint32_t h = (((uint32_t)r.k[r0 + i]) >> rb) & (uint32_t)(nh - 1);
                next[i] = head[h];
                head[h] = (int32_t)i;
            }
            for (uint64_t j = 0; j < sn; j++) {
                int32_t key = s.k[s0 + j];
                uint32_t h = (((uint32_t)key) >> rb) & (uint32_t)(nh - 1);
                for (int32_t c = head[h]; c >= 0; c = next[c])
                    if (r.k[r0 + c] == key) {
                        matches++;
                        agg += (uint64_t)((int64_t)r.p[r0 + c] * (int64_t)s.p[s0 + j]);
                    }
            }
        }
        free(head);
        free(next);
    }
    free(r.k); free(r.p); free(r.off);
    free(s.k); free(s.p); free(s.off);
    if (agg_out) *agg_out = agg;
    return matches;
}
