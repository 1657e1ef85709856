/*
 * gen_oracle.c — ORACLE (test infrastructure, never shipped): plain-C restatement of the
 * reference's generator_ETHZ relation generators.  See oracle.h for the rules on who may call it.
 *
 * Follows /root/reference/src/generator_ETHZ.cu.  The libc PRNGs the reference uses (rand(),
 * nrand48(); not vendored, glibc) are called directly so that the streams are the reference's.
 * Validated bit-for-bit against the reference generator itself (oracle/_ref/refgen) by
 * tests/test_oracle_gen.py and against the committed outputs in tests/golden/.
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* gen.cu:16-17 */
#define O_RAND_RANGE(N) ((double)rand() / ((double)RAND_MAX + 1) * (N))
#define O_RAND_RANGE48(N, STATE) ((double)nrand48(STATE) / ((double)RAND_MAX + 1) * (N))

/* gen.cu:23-27 */
void o_seed_generator(unsigned int seed) { srand(seed); }

/* gen.cu:115-122 */
void o_random_gen(int32_t *rel, uint64_t n, int64_t maxid) {
    for (uint64_t i = 0; i < n; i++) rel[i] = (int32_t)O_RAND_RANGE(maxid);
}

/* gen.cu:204-212 */
void o_knuth_shuffle48(int32_t *rel, uint64_t n, unsigned short state[3]) {
    for (int64_t i = (int64_t)n - 1; i > 0; i--) {
        int64_t j = (int64_t)O_RAND_RANGE48(i, state);
        int32_t tmp = rel[i];
        rel[i] = rel[j];
        rel[j] = tmp;
    }
}

/* gen.cu:194-202 */
void o_knuth_shuffle(int32_t *rel, uint64_t n) {
    for (int64_t i = (int64_t)n - 1; i > 0; i--) {
        int64_t j = (int64_t)O_RAND_RANGE(i);
        int32_t tmp = rel[i];
        rel[i] = rel[j];
        rel[j] = tmp;
    }
}

/* gen.cu:127-149.  The reference seeds nrand48 with time(NULL) copied into the first 4 bytes of a
 * zeroed unsigned short[3] (little endian: state[0]=seed&0xFFFF, state[1]=seed>>16, state[2]=0).
 * Key sequence before the shuffle: 0,1,...,maxid,1,2,...,maxid,1,...  (key 0 appears once; key
 * maxid appears although a PK relation of maxid tuples holds 0..maxid-1 only). */
void o_random_unique_gen(int32_t *rel, uint64_t n, int64_t maxid, unsigned int time_seed) {
    uint64_t firstkey = 0;
    unsigned short state[3] = {0, 0, 0};
    memcpy(state, &time_seed, sizeof(time_seed));
    for (uint64_t i = 0; i < n; i++) {
        rel[i] = (int32_t)firstkey;
        if (firstkey == (uint64_t)maxid) firstkey = 0;
        firstkey++;
    }
    o_knuth_shuffle48(rel, n, state);
}

/* gen.cu:162-187 (generation branch): repeat the PK relation, then Knuth-shuffle with rand().
 * The reference does not call check_seed() here: rand() continues from whatever state it has. */
void o_fk_from_pk(int32_t *fk, uint64_t nfk, const int32_t *pk, uint64_t npk) {
    uint64_t iters = nfk / npk, i;
    for (i = 0; i < iters; i++) memcpy(fk + i * npk, pk, npk * sizeof(int32_t));
    uint64_t rem = nfk % npk;
    if (rem > 0) memcpy(fk + i * npk, pk, rem * sizeof(int32_t));
    o_knuth_shuffle(fk, nfk);
}

/* gen.cu:97-110 */
void o_create_relation_n(const int32_t *in, int32_t *out, uint64_t n, uint64_t times) {
    for (uint64_t i = 0; i < times; i++) memcpy(out + i * n, in, n * sizeof(int32_t));
}

/* gen.cu:236-258: values 1..size permuted with rand() (0 is kept out of the alphabet) */
static uint32_t *o_gen_alphabet(unsigned int size) {
    uint32_t *alphabet = (uint32_t *)malloc((size_t)size * sizeof(*alphabet));
    for (unsigned int i = 0; i < size; i++) alphabet[i] = i + 1;
    for (unsigned int i = size - 1; i > 0; i--) {
        unsigned int k = (unsigned int)((unsigned long)i * rand() / RAND_MAX);
        unsigned int tmp = alphabet[i];
        alphabet[i] = alphabet[k];
        alphabet[k] = tmp;
    }
    return alphabet;
}

/* gen.cu:265-294: cumulative Zipf density */
static double *o_gen_zipf_lut(double zipf_factor, unsigned int alphabet_size) {
    double *lut = (double *)malloc((size_t)alphabet_size * sizeof(*lut));
    double scaling_factor = 0.0, sum = 0.0;
    for (unsigned int i = 1; i <= alphabet_size; i++) scaling_factor += 1.0 / pow(i, zipf_factor);
    for (unsigned int i = 1; i <= alphabet_size; i++) {
        sum += 1.0 / pow(i, zipf_factor);
        lut[i - 1] = sum / scaling_factor;
    }
    return lut;
}

/* gen.cu:299-348 */
void o_gen_zipf(uint64_t n, unsigned int alphabet_size, double zipf_factor, int32_t *ret) {
    uint32_t *alphabet = o_gen_alphabet(alphabet_size);
    double *lut = o_gen_zipf_lut(zipf_factor, alphabet_size);
    /* gen.cu:308-311: 64 rand() draws are consumed into an unused seeds[] array */
    for (int i = 0; i < 64; i++) (void)rand();
    for (uint64_t i = 0; i < n; i++) {
        double r = ((double)(rand())) / RAND_MAX;
        unsigned int left = 0, right = alphabet_size - 1, m, pos;
        if (lut[0] >= r) {
            pos = 0;
        } else {
            while (right - left > 1) {
                m = (left + right) / 2;
                if (lut[m] < r)
                    left = m;
                else
                    right = m;
            }
            pos = right;
        }
        ret[i] = (int32_t)alphabet[pos];
    }
    free(lut);
    free(alphabet);
}

/* gen.cu:38-59: raw int32, no header.  Unlike the reference (D12) a short read is reported. */
int o_read_bin(const char *path, int32_t *rel, uint64_t n) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return 1;
    size_t got = fread(rel, sizeof(int32_t), n, fp);
    fclose(fp);
    return got == n ? 0 : 2;
}

/* gen.cu:61-72 */
int o_write_bin(const char *path, const int32_t *rel, uint64_t n) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return 1;
    size_t put = fwrite(rel, sizeof(int32_t), n, fp);
    fclose(fp);
    return put == n ? 0 : 2;
}
