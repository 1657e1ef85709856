// gen_ethz.cpp — host-side relation generators + the raw-int32 ".bin" relation cache: the drop-in
// for the reference's generator_ETHZ (src/generator_ETHZ.cu / .cuh:11-23), which main.cu calls to
// create R and S (src/main.cu:186-262).
//
// Same generators, same value streams, same file format — but the process-global libc state the
// reference leans on (rand()/srand(), nrand48() seeded from time(NULL): gen.cu:30-36,133-135) is
// replaced by generators implemented here with an explicit seed, so that inputs are reproducible and
// the library never touches the caller's rand() state:
//   * GlibcRand  — glibc's rand()/srand() TYPE_3 additive-feedback generator (r[i] = r[i-3] + r[i-31],
//                  seeded by the Lehmer sequence 16807*x mod 2^31-1, first 310 outputs dropped)
//   * Rand48     — POSIX nrand48: X' = (0x5DEECE66D * X + 0xB) mod 2^48, result = X' >> 17
// Parity with the reference's streams is checked bit-for-bit in tests/test_generator.py against
// tests/golden/ (outputs of the reference generator itself).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>

#include "hj.h"

namespace {

constexpr int32_t kRandMax = 2147483647; // glibc RAND_MAX

class GlibcRand {
public:
    void seed(unsigned int s) {
        if (s == 0) s = 1;
        int32_t word = (int32_t)s;
        r_[0] = (uint32_t)word;
        for (int i = 1; i < 31; i++) {
            // 16807 * word mod 2147483647 without overflow (Schrage)
            long hi = word / 127773, lo = word % 127773;
            word = (int32_t)(16807 * lo - 2836 * hi);
            if (word < 0) word += 2147483647;
            r_[i] = (uint32_t)word;
        }
        f_ = 3;
        b_ = 0;
        for (int i = 0; i < 310; i++) (void)step();
    }
    int next() { return (int)(step() >> 1); }

private:
    // degree 31, separation 3: front pointer starts 3 ahead of the rear pointer
    uint32_t step() {
        r_[f_] += r_[b_];
        uint32_t out = r_[f_];
        if (++f_ == 31) f_ = 0;
        if (++b_ == 31) b_ = 0;
        return out;
    }
    uint32_t r_[31];
    int f_ = 3, b_ = 0;
};

class Rand48 {
public:
    explicit Rand48(unsigned int time_seed) {
        // gen.cu:132-135: unsigned short state[3] = {0,0,0}; memcpy(state, &seed, 4) — little endian
        x_ = (uint64_t)time_seed & 0xFFFFFFFFull;
    }
    long next() {
        x_ = (x_ * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (long)(x_ >> 17);
    }

private:
    uint64_t x_;
};

struct GenState {
    GlibcRand rnd;
    bool seeded = false;       // gen.cu:20 `seeded`
    bool rnd_init = false;     // libc's own "srand was called" state (default stream = srand(1))
    unsigned int time_seed = 0; // 0 = ask time(NULL) like the reference
};
GenState g;

unsigned int now_seed() { return g.time_seed ? g.time_seed : (unsigned int)time(NULL); }

// gen.cu:30-36 check_seed
void check_seed() {
    if (!g.seeded) {
        g.rnd.seed(now_seed());
        g.seeded = g.rnd_init = true;
    }
}

inline double rand_range(int64_t n) { return (double)g.rnd.next() / ((double)kRandMax + 1) * (double)n; }

// gen.cu:194-202
void knuth_shuffle(int32_t *rel, uint64_t n) {
    for (int64_t i = (int64_t)n - 1; i > 0; i--) {
        int64_t j = (int64_t)rand_range(i);
        int32_t t = rel[i];
        rel[i] = rel[j];
        rel[j] = t;
    }
}

// gen.cu:204-212
void knuth_shuffle48(int32_t *rel, uint64_t n, Rand48 &st) {
    for (int64_t i = (int64_t)n - 1; i > 0; i--) {
        int64_t j = (int64_t)((double)st.next() / ((double)kRandMax + 1) * (double)i);
        int32_t t = rel[i];
        rel[i] = rel[j];
        rel[j] = t;
    }
}

// gen.cu:115-122
void random_gen(int32_t *rel, uint64_t n, int64_t maxid) {
    for (uint64_t i = 0; i < n; i++) rel[i] = (int32_t)rand_range(maxid);
}

// gen.cu:127-149: 0,1,..,maxid,1,..,maxid,1,.. then shuffle
void random_unique_gen(int32_t *rel, uint64_t n, int64_t maxid) {
    Rand48 st(now_seed());
    uint64_t firstkey = 0;
    for (uint64_t i = 0; i < n; i++) {
        rel[i] = (int32_t)firstkey;
        if (firstkey == (uint64_t)maxid) firstkey = 0;
        firstkey++;
    }
    knuth_shuffle48(rel, n, st);
}

// gen.cu:299-348 with gen_alphabet (:236-258) and gen_zipf_lut (:265-294)
void gen_zipf(uint64_t n, unsigned int alphabet_size, double theta, int32_t *ret) {
    std::vector<uint32_t> alphabet(alphabet_size);
    for (unsigned int i = 0; i < alphabet_size; i++) alphabet[i] = i + 1;
    for (unsigned int i = alphabet_size - 1; i > 0 && alphabet_size; i--) {
        unsigned int k = (unsigned int)((unsigned long)i * (unsigned long)g.rnd.next() / (unsigned long)kRandMax);
        uint32_t t = alphabet[i];
        alphabet[i] = alphabet[k];
        alphabet[k] = t;
    }
    std::vector<double> lut(alphabet_size);
    double scaling = 0.0, sum = 0.0;
    for (unsigned int i = 1; i <= alphabet_size; i++) scaling += 1.0 / pow((double)i, theta);
    for (unsigned int i = 1; i <= alphabet_size; i++) {
        sum += 1.0 / pow((double)i, theta);
        lut[i - 1] = sum / scaling;
    }
    for (int i = 0; i < 64; i++) (void)g.rnd.next(); // gen.cu:308-311: 64 draws into an unused array
    for (uint64_t i = 0; i < n; i++) {
        double r = (double)g.rnd.next() / kRandMax;
        unsigned int left = 0, right = alphabet_size - 1, pos;
        if (lut[0] >= r) {
            pos = 0;
        } else {
            while (right - left > 1) {
                unsigned int m = (left + right) / 2;
                if (lut[m] < r) left = m; else right = m;
            }
            pos = right;
        }
        ret[i] = (int32_t)alphabet[pos];
    }
}

bool use_cache(const char *f) { return f && f[0]; }

} // namespace

extern "C" {

// seed_generator (gen.cu:23-27) and the time(NULL) of random_unique_gen in one explicit knob
void hj_gen_set_seed(uint64_t seed) {
    g.time_seed = (unsigned int)seed;
    if (seed) {
        g.rnd.seed((unsigned int)seed);
        g.seeded = g.rnd_init = true;
    } else {
        g.seeded = g.rnd_init = false;
    }
}

// gen.cu:38-59.  0 = read, HJ_EIO = no such file or short file (the reference ignores short reads, D12)
int hj_read_relation(const char *filename, int32_t *relation, uint64_t num_tuples) {
    FILE *fp = fopen(filename, "rb");
    if (!fp) return HJ_EIO;
    size_t got = fread(relation, sizeof(int32_t), num_tuples, fp);
    fclose(fp);
    return got == num_tuples ? HJ_OK : HJ_EIO;
}

// gen.cu:61-72
int hj_write_relation(const char *filename, const int32_t *relation, uint64_t num_tuples) {
    FILE *fp = fopen(filename, "wb");
    if (!fp) return HJ_EIO;
    size_t put = fwrite(relation, sizeof(int32_t), num_tuples, fp);
    fclose(fp);
    return put == num_tuples ? HJ_OK : HJ_EIO;
}

// gen.cu:86-94: read-through cache, else generate + write.  filename NULL/"" = no cache.
int hj_create_relation_unique(const char *filename, int32_t *relation, uint64_t num_tuples, int64_t maxid) {
    if (use_cache(filename) && hj_read_relation(filename, relation, num_tuples) == HJ_OK) return HJ_OK;
    random_unique_gen(relation, num_tuples, maxid);
    return use_cache(filename) ? hj_write_relation(filename, relation, num_tuples) : HJ_OK;
}

// gen.cu:74-83
int hj_create_relation_nonunique(const char *filename, int32_t *relation, uint64_t num_tuples, int64_t maxid) {
    if (use_cache(filename) && hj_read_relation(filename, relation, num_tuples) == HJ_OK) return HJ_OK;
    check_seed();
    random_gen(relation, num_tuples, maxid);
    return use_cache(filename) ? hj_write_relation(filename, relation, num_tuples) : HJ_OK;
}

// gen.cu:214-226
int hj_create_relation_zipf(const char *filename, int32_t *relation, uint64_t num_tuples, int64_t maxid,
                            double zipf_param) {
    if (maxid <= 0 || maxid > 0xFFFFFFFFll) return HJ_EINVAL;
    if (use_cache(filename) && hj_read_relation(filename, relation, num_tuples) == HJ_OK) return HJ_OK;
    check_seed();
    gen_zipf(num_tuples, (unsigned int)maxid, zipf_param, relation);
    return use_cache(filename) ? hj_write_relation(filename, relation, num_tuples) : HJ_OK;
}

// gen.cu:162-187.  Like the reference this does not seed: the rand() stream simply continues
// (a never-seeded stream is glibc's default seed 1).
int hj_create_relation_fk_from_pk(const char *filename, int32_t *fkrel, uint64_t fk_tuples, const int32_t *pkrel,
                                  uint64_t pk_tuples) {
    if (!pk_tuples) return HJ_EINVAL;
    if (use_cache(filename) && hj_read_relation(filename, fkrel, fk_tuples) == HJ_OK) return HJ_OK;
    if (!g.rnd_init) { g.rnd.seed(1); g.rnd_init = true; }
    uint64_t iters = fk_tuples / pk_tuples, i;
    for (i = 0; i < iters; i++) memcpy(fkrel + i * pk_tuples, pkrel, pk_tuples * sizeof(int32_t));
    uint64_t rem = fk_tuples % pk_tuples;
    if (rem) memcpy(fkrel + i * pk_tuples, pkrel, rem * sizeof(int32_t));
    knuth_shuffle(fkrel, fk_tuples);
    return use_cache(filename) ? hj_write_relation(filename, fkrel, fk_tuples) : HJ_OK;
}

// gen.cu:97-110
int hj_create_relation_n(const int32_t *in_relation, int32_t *out_relation, uint64_t num_tuples, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) memcpy(out_relation + i * num_tuples, in_relation, num_tuples * sizeof(int32_t));
    return HJ_OK;
}

} // extern "C"
