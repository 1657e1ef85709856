/*
 * ref_harness.cpp — ORACLE tooling (test infrastructure): a command-line driver around the
 * REFERENCE's own generator, compiled from /root/reference/src/generator_ETHZ.cu where it lies
 * (see oracle/Makefile; the object and this binary go to oracle/_ref/, which is git-ignored).
 * Nothing here restates or copies reference code: it only calls the functions the reference
 * declares in generator_ETHZ.cuh and reports which time(NULL) seed they ran under.
 *
 * usage (JSON meta goes to stderr; stdout carries the reference's own prints):
 *   refgen unique   N MAXID OUT.bin            create_relation_unique  (time-seeded nrand48)
 *   refgen nonuniq  SEED N MAXID OUT.bin       seed_generator + create_relation_nonunique
 *   refgen zipf     SEED N ALPHABET THETA OUT  seed_generator + create_relation_zipf
 *   refgen fkpk     SEED NPK MAXID NFK PK.bin FK.bin   nonunique PK, then create_relation_fk_from_pk
 *   refgen repeat   N TIMES IN.bin OUT.bin      create_relation_n on a file
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include "generator_ETHZ.cuh"

static int dump(const char *path, const int *rel, uint64_t n) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return 1;
    fwrite(rel, sizeof(int), n, fp);
    fclose(fp);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const char *mode = argv[1];
    if (!strcmp(mode, "unique") && argc == 5) {
        uint64_t n = strtoull(argv[2], 0, 10);
        int64_t maxid = strtoll(argv[3], 0, 10);
        std::vector<int> rel(n ? n : 1);
        for (int attempt = 0; attempt < 100; attempt++) {
            remove(argv[4]); /* the reference reads the file instead of generating if it exists */
            time_t t0 = time(NULL);
            int rc = create_relation_unique(argv[4], rel.data(), n, maxid);
            time_t t1 = time(NULL);
            if (rc) return 3;
            if (t0 == t1) { /* the seed the reference took from time(NULL) is known exactly */
                fprintf(stderr, "{\"mode\":\"unique\",\"n\":%lu,\"maxid\":%ld,\"time_seed\":%lu}\n",
                        (unsigned long)n, (long)maxid, (unsigned long)t0);
                return 0;
            }
        }
        return 4;
    }
    if (!strcmp(mode, "nonuniq") && argc == 6) {
        unsigned seed = (unsigned)strtoul(argv[2], 0, 10);
        uint64_t n = strtoull(argv[3], 0, 10);
        int64_t maxid = strtoll(argv[4], 0, 10);
        std::vector<int> rel(n ? n : 1);
        remove(argv[5]);
        seed_generator(seed);
        if (create_relation_nonunique(argv[5], rel.data(), n, maxid)) return 3;
        fprintf(stderr, "{\"mode\":\"nonuniq\",\"seed\":%u,\"n\":%lu,\"maxid\":%ld}\n", seed,
                (unsigned long)n, (long)maxid);
        return 0;
    }
    if (!strcmp(mode, "zipf") && argc == 7) {
        unsigned seed = (unsigned)strtoul(argv[2], 0, 10);
        uint64_t n = strtoull(argv[3], 0, 10);
        int64_t alphabet = strtoll(argv[4], 0, 10);
        double theta = atof(argv[5]);
        std::vector<int> rel(n ? n : 1);
        remove(argv[6]);
        seed_generator(seed);
        if (create_relation_zipf(argv[6], rel.data(), n, alphabet, theta)) return 3;
        fprintf(stderr, "{\"mode\":\"zipf\",\"seed\":%u,\"n\":%lu,\"alphabet\":%ld,\"theta\":%.17g}\n",
                seed, (unsigned long)n, (long)alphabet, theta);
        return 0;
    }
    if (!strcmp(mode, "fkpk") && argc == 8) {
        unsigned seed = (unsigned)strtoul(argv[2], 0, 10);
        uint64_t npk = strtoull(argv[3], 0, 10);
        int64_t maxid = strtoll(argv[4], 0, 10);
        uint64_t nfk = strtoull(argv[5], 0, 10);
        std::vector<int> pk(npk ? npk : 1), fk(nfk ? nfk : 1);
        remove(argv[6]);
        remove(argv[7]);
        seed_generator(seed);
        if (create_relation_nonunique(argv[6], pk.data(), npk, maxid)) return 3;
        if (create_relation_fk_from_pk(argv[7], fk.data(), nfk, pk.data(), npk)) return 3;
        fprintf(stderr, "{\"mode\":\"fkpk\",\"seed\":%u,\"npk\":%lu,\"maxid\":%ld,\"nfk\":%lu}\n",
                seed, (unsigned long)npk, (long)maxid, (unsigned long)nfk);
        return 0;
    }
    if (!strcmp(mode, "repeat") && argc == 6) {
        uint64_t n = strtoull(argv[2], 0, 10), times = strtoull(argv[3], 0, 10);
        std::vector<int> in(n ? n : 1), out(n * times ? n * times : 1);
        if (readFromFile(argv[4], in.data(), n)) return 3;
        create_relation_n(in.data(), out.data(), n, times);
        if (dump(argv[5], out.data(), n * times)) return 3;
        fprintf(stderr, "{\"mode\":\"repeat\",\"n\":%lu,\"times\":%lu}\n", (unsigned long)n,
                (unsigned long)times);
        return 0;
    }
    return 2;
}
