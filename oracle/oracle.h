/*
 * oracle.h — CPU ORACLE for the radix-partitioned hash-join path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (psiul/ICDE2019-GPU-Join) for the
 * north-star path: the generator_ETHZ input generators, the radix partition function, the inner
 * equi-join semantics (match count, sum(payR*payS), materialised (key,payR,payS) multiset) and the
 * reference's own (never-called) CPU cross-check joinCpu.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
 * product (icde2019-gpu-join_amd/) never links, imports or calls it.
 *
 * Pinning: the reference holds no tests or golden vectors for this path; both halves of this oracle
 * are pinned to the reference itself, compiled here from where it lies by oracle/Makefile (`ref` target, outputs
 * only into oracle/_ref/, git-ignored):
 *   - generator half: oracle/_ref/refgen = /root/reference/src/generator_ETHZ.cu (unmodified, g++ -x c++);
 *     the .bin files under tests/golden/ are its outputs (tests/golden/make_golden.py), replayed bit for bit;
 *   - join half: oracle/_ref/refjoin = the reference's own CPU join, joinCpu + h_hashMurmur
 *     (/root/reference/src/hash_join_clustered_probe.cu:2013-2059: the span is cut out of the file at build time and
 *     compiled next to the reference's common-host.h, no stand-in headers); tests/golden/join_answers.json holds the
 *     match count s and key sum g it printed for 15 pairs of golden relations (OMP_NUM_THREADS=1: `s` is missing from
 *     its reduction clause).  tests/test_oracle_join.py requires every restatement here (sort-merge o_join_*,
 *     chained-hash o_joinCpu, OpenMP radix join) to reproduce s and g, and in the build container also runs refjoin
 *     live on fresh inputs.  The closed-form counts implied by the generator construction are
 *     checked on top.  The GPU kernels of the reference cannot be built here (nvcc absent, CUDA-only) — see DESIGN.md.
 */
#ifndef HJ_ORACLE_H_
#define HJ_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- generator_ETHZ restatement (src/generator_ETHZ.cu) ---------- */

/* gen.cu:23-27  seed_generator: srand(seed) */
void o_seed_generator(unsigned int seed);
/* gen.cu:115-122 random_gen: rel[i] = RAND_RANGE(maxid), libc rand() stream */
void o_random_gen(int32_t *rel, uint64_t n, int64_t maxid);
/* gen.cu:127-149 random_unique_gen with the time(NULL) seed made explicit (time_seed) */
void o_random_unique_gen(int32_t *rel, uint64_t n, int64_t maxid, unsigned int time_seed);
/* gen.cu:194-202 knuth_shuffle (libc rand()) */
void o_knuth_shuffle(int32_t *rel, uint64_t n);
/* gen.cu:204-212 knuth_shuffle48 (nrand48 on a caller state) */
void o_knuth_shuffle48(int32_t *rel, uint64_t n, unsigned short state[3]);
/* gen.cu:162-187 create_relation_fk_from_pk, generation branch only (no file cache) */
void o_fk_from_pk(int32_t *fk, uint64_t nfk, const int32_t *pk, uint64_t npk);
/* gen.cu:299-348 gen_zipf (+ gen_alphabet :236-258, gen_zipf_lut :265-294); no "live" prints */
void o_gen_zipf(uint64_t n, unsigned int alphabet_size, double zipf_factor, int32_t *ret);
/* gen.cu:97-110 create_relation_n: n-fold concatenation */
void o_create_relation_n(const int32_t *in, int32_t *out, uint64_t n, uint64_t times);
/* gen.cu:38-72 raw little-endian int32 file format; return 0 ok, 1 cannot open, 2 short read */
int o_read_bin(const char *path, int32_t *rel, uint64_t n);
int o_write_bin(const char *path, const int32_t *rel, uint64_t n);

/* ---------- partition function (common.h:45-47, jp.cu:126) ---------- */

/* digit = (hasht(key) >> shift) & (2^bits-1), hasht = identity on the uint32 bit pattern.
 * Stable counting-sort partition; offsets has 2^bits+1 entries. */
void o_radix_partition(const int32_t *keys, const int32_t *pays, uint64_t n, uint32_t shift,
                       uint32_t bits, int32_t *out_keys, int32_t *out_pays, uint64_t *offsets);

/* Order-independent digest of the (key,payload) multiset of every partition [off[p],off[p+1]). */
void o_partition_digest(const int32_t *keys, const int32_t *pays, const uint64_t *offsets,
                        uint64_t nparts, uint64_t *digest);

/* ---------- join semantics (jp.cu:1056-1079, 1073, 1092; hjcp.cu:2044-2053) ---------- */

/* 64-bit mix used for order-independent checksums of (key,payR,payS) and (key,pay). */
uint64_t o_mix_triple(int32_t key, int32_t pr, int32_t ps);
uint64_t o_mix_pair(int32_t key, int32_t pay);

/* Inner equi-join, all pairs.  matches = number of (r,s) with R[r]==S[s];
 * agg = sum payR*payS mod 2^64 (its low 32 bits are the reference's int32 aggregate, jp.cu:1073);
 * checksum = sum o_mix_triple(key,payR,payS) mod 2^64 over all output tuples.
 * Pr/Ps may be NULL (= all ones, hjcp.cu:1994-1999). */
void o_join_count(const int32_t *R, const int32_t *Pr, uint64_t nR, const int32_t *S,
                  const int32_t *Ps, uint64_t nS, uint64_t *matches, uint64_t *agg,
                  uint64_t *checksum);

/* Materialise the join output sorted by (key,payR,payS); returns the number of output tuples
 * (writes at most cap of them). */
uint64_t o_join_materialize(const int32_t *R, const int32_t *Pr, uint64_t nR, const int32_t *S,
                            const int32_t *Ps, uint64_t nS, int32_t *out_key, int32_t *out_pr,
                            int32_t *out_ps, uint64_t cap);

/* Checksum of an already materialised output (any order). */
uint64_t o_triples_checksum(const int32_t *key, const int32_t *pr, const int32_t *ps, uint64_t n);

/* hjcp.cu:2013-2059 joinCpu restated (2^20-slot chained table, murmur3 finaliser h_hashMurmur
 * :2016-2023, serial build over R, probe with S).  s = match count, g = sum of matching S keys
 * (uint32 wrap), as the reference prints them.  The reference's OpenMP race on s (D10) is not
 * replicated.  threads<=1 → serial. */
void o_joinCpu(const int32_t *R, uint64_t nR, const int32_t *S, uint64_t nS, int threads,
               uint64_t *s, uint32_t *g);

/* CPU baseline ("port"): OpenMP two-pass radix partition on the low key bits followed by a
 * per-partition chained build/probe — the structure of the path under test (jp.cu:58-535,
 * 885-1095) on host cores.  Returns match count; *agg as in o_join_count. */
uint64_t o_radix_join_omp(const int32_t *R, const int32_t *Pr, uint64_t nR, const int32_t *S,
                          const int32_t *Ps, uint64_t nS, uint32_t bits1, uint32_t bits2,
                          int threads, uint64_t *agg);

int o_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
