// bench_main.cpp — the `bench` command-line driver: the reference's CLI contract (src/main.cu:80-310,
// 434-557) on top of libhj.so.
//
//   ./bench -b 7 -a HJC -R <n> -S <m> [-s theta] [--non-unique] [--full-range] [--file -k R.bin -l S.bin]
//           [-x mult] [-y mult] [-t -v -m -p -w: accepted, echoed, ignored by HJC like the reference]
//           [--seed N]   (new: reproducible generation; default = time(NULL) like the reference)
//
// Differences by design: `-a` is validated (D8: the reference walks an
// unterminated table and calls an uninitialised pointer when -a is omitted); the skew cache file
// is named unique_skew<theta>_S<m>.bin (D7: the reference's sprintf has one argument too few);
// relation files are checked for short reads (D12).
#include <getopt.h>
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hj.h"
#include "hj_reference_abi.h"

namespace {

struct JoinAlg {
    const char *name;
    unsigned int (*fn)(args *, timingInfo *);
};
// main.cu:64-66: the algorithm table has a single live entry
const JoinAlg kAlgs[] = {{"HJC", hashJoinClusteredProbe}, {nullptr, nullptr}};

struct Input {
    int option = 0;
    const JoinAlg *alg = nullptr;
    uint64_t S_n = 0, R_n = 0;
    int unique_keys = 1, full_range = 0, file_input = 0;
    double skew = 0.0;
    int threads = 32, values = 2, shared_mem = 30 << 10, one_to_many = 0;
    unsigned pivots = 1;
    int R_mult = 1, S_mult = 1;
    const char *R_file = nullptr, *S_file = nullptr;
    uint64_t seed = 0;
};

[[noreturn]] void usage_exit() { // main.cu:68-73
    printf("./bench -b <7 (generate + join), 8 (generate only)> -a HJC -R <tuples> -S <tuples> "
           "[-s skew] [--non-unique] [--full-range] [--file -k R.bin -l S.bin] [-x m] [-y m] [--seed n]\n");
    exit(1);
}

void parse(int argc, char **argv, Input *in) {
    int unique_flag = in->unique_keys, range_flag = in->full_range, file_flag = in->file_input;
    static struct option opts[] = {{"file", no_argument, nullptr, 1000},
                                   {"non-unique", no_argument, nullptr, 1001},
                                   {"full-range", no_argument, nullptr, 1002},
                                   {"seed", required_argument, nullptr, 1003},
                                   {"benchmark", required_argument, nullptr, 'b'},
                                   {"alg", required_argument, nullptr, 'a'},
                                   {"SelsNum", required_argument, nullptr, 'S'},
                                   {"RelsNum", required_argument, nullptr, 'R'},
                                   {"skew", required_argument, nullptr, 's'},
                                   {"threadsNum", required_argument, nullptr, 't'},
                                   {"values", required_argument, nullptr, 'v'},
                                   {"memory", required_argument, nullptr, 'm'},
                                   {"pivotsNum", required_argument, nullptr, 'p'},
                                   {"OneToMany", required_argument, nullptr, 'w'},
                                   {"XSelsMultiplier", required_argument, nullptr, 'x'},
                                   {"YRelsMultiplier", required_argument, nullptr, 'y'},
                                   {"R_filename", required_argument, nullptr, 'k'},
                                   {"S_filename", required_argument, nullptr, 'l'},
                                   {nullptr, 0, nullptr, 0}};
    printf("INPUT: "); // main.cu:443
    int c;
    while ((c = getopt_long(argc, argv, "b:a:S:R:s:t:v:m:p:w:x:y:k:l:", opts, nullptr)) != -1) {
        switch (c) {
        case 1000: file_flag = 1; printf("file\t"); break;
        case 1001: unique_flag = 0; printf("non-unique\t"); break;
        case 1002: range_flag = 1; printf("full-range\t"); break;
        case 1003: in->seed = strtoull(optarg, nullptr, 10); printf("seed = %lu\t", (unsigned long)in->seed); break;
        case 'b': in->option = atoi(optarg); printf("option = %d\t", in->option); break;
        case 'a':
            for (const JoinAlg *a = kAlgs; a->name; a++)
                if (!strcmp(optarg, a->name)) in->alg = a;
            if (!in->alg) { fprintf(stderr, "\nERROR: unknown join algorithm '%s' (only HJC exists)\n", optarg); exit(1); }
            printf("joinAlg = %s\t", in->alg->name);
            break;
        case 'k': in->R_file = optarg; printf("R filename = %s\t", optarg); break;
        case 'l': in->S_file = optarg; printf("S filename = %s\t", optarg); break;
        case 'S': case 'R': {
            uint64_t p = strtoull(optarg, nullptr, 10);
            if (p > ULONG_MAX / sizeof(int)) { // main.cu:491-514
                fprintf(stderr, "WARNING: %s is too big (%lu). Setting it to maximum supported value %lu\n",
                        c == 'S' ? "SelsNum" : "RelsNum", (unsigned long)p, ULONG_MAX / sizeof(int));
                p = ULONG_MAX / sizeof(int);
            }
            if (c == 'S') { in->S_n = p; printf("||S|| = %lu\t", (unsigned long)p); }
            else { in->R_n = p; printf("||R|| = %lu\t", (unsigned long)p); }
            break;
        }
        case 's': in->skew = atof(optarg); printf("skew = %f\t", in->skew); break;
        case 't': in->threads = atoi(optarg); printf("#threads = %d\t", in->threads); break;
        case 'v': in->values = atoi(optarg); printf("values per thread= %d\t", in->values); break;
        case 'm': in->shared_mem = atoi(optarg); printf("sharedMem = %d\t", in->shared_mem); break;
        case 'p': in->pivots = (unsigned)atoi(optarg); printf("pivotsNum = %u\t", in->pivots); break;
        case 'w': in->one_to_many = atoi(optarg); printf("OneToMany = %d\t", in->one_to_many); break;
        case 'x': in->S_mult = atoi(optarg); printf("SelsMultiplier = %d\t", in->S_mult); break;
        case 'y': in->R_mult = atoi(optarg); printf("RelsMultiplier = %d\t", in->R_mult); break;
        default: printf("\n"); usage_exit();
        }
    }
    in->unique_keys = unique_flag;
    in->full_range = range_flag;
    in->file_input = file_flag;
    printf("\n");
    // main.cu:556 accepts 1..9,100,101 and then rejects everything but 7/8 in main's switch
    if (in->option != 7 && in->option != 8) usage_exit();
    if (in->R_mult < 1 || in->S_mult < 1) usage_exit();
}

int32_t *alloc_col(uint64_t n, bool *pinned) {
    void *p = nullptr;
    size_t bytes = (size_t)(n ? n : 1) * sizeof(int32_t);
    // main.cu:181-183 (MEM_HOST): pinned, mapped host columns; plain malloc when no GPU is present (-b 8)
    if (hipHostMalloc(&p, bytes, hipHostMallocMapped) == hipSuccess) { *pinned = true; return (int32_t *)p; }
    *pinned = false;
    return (int32_t *)malloc(bytes);
}

int name(char *dst, const char *fmt, unsigned long a, unsigned long b = 0) {
    int n = snprintf(dst, 50, fmt, a, b);
    if (n >= 50) { fprintf(stderr, "ERROR: filename is %d characters long\n", n); return 1; }
    return 0;
}

} // namespace

int main(int argc, char **argv) {
    Input in;
    parse(argc, argv, &in);
    if (in.option == 7 && !in.alg) { fprintf(stderr, "ERROR: -a HJC is required with -b 7\n"); return 1; }
    hj_gen_set_seed(in.seed);

    // -x / -y: generate a base relation, then concatenate it (main.cu:103-109,208-248)
    const uint64_t base_R = in.R_n, base_S = in.S_n;
    args ja;
    memset(&ja, 0, sizeof ja);
    ja.R_els = base_R * (uint64_t)in.R_mult;
    ja.S_els = base_S * (uint64_t)in.S_mult;
    const unsigned long R_mb = (unsigned long)(ja.R_els * sizeof(int) / 1024 / 1024);
    const unsigned long S_mb = (unsigned long)(ja.S_els * sizeof(int) / 1024 / 1024);

    // cache file names, main.cu:116-158
    if (!in.file_input) {
        int bad = 0;
        if (in.full_range) {
            bad |= name(ja.S_filename, "fk_S%lu_pk_R%lu.bin", ja.S_els, ja.R_els);
            bad |= name(ja.R_filename, "pk_R%lu.bin", ja.R_els);
        } else if (in.unique_keys) {
            bad |= name(ja.R_filename, "unique_%lu.bin", base_R);
            if (in.skew > 0) {
                int n = snprintf(ja.S_filename, 50, "unique_skew%.2f_S%lu.bin", in.skew, (unsigned long)base_S);
                bad |= n >= 50;
            } else {
                bad |= name(ja.S_filename, "unique_%lu.bin", base_S);
            }
        } else {
            bad |= name(ja.S_filename, "nonUnique_S%lu.bin", ja.S_els);
            bad |= name(ja.R_filename, "nonUnique_R%lu.bin", ja.R_els);
        }
        if (bad) return 1;
    }

    bool pin_r = false, pin_s = false;
    ja.R = alloc_col(ja.R_els, &pin_r);
    ja.S = alloc_col(ja.S_els, &pin_s);
    if (!ja.R || !ja.S) { fprintf(stderr, "Problem allocating space for the relations\n"); return 0; }

    int rc = 0;
    if (in.file_input) { // main.cu:186-189
        printf("Reading from files\n");
        if (!in.R_file || !in.S_file) { fprintf(stderr, "ERROR: --file needs -k <R.bin> -l <S.bin>\n"); return 1; }
        rc |= hj_read_relation(in.R_file, ja.R, ja.R_els);
        rc |= hj_read_relation(in.S_file, ja.S, ja.S_els);
        if (rc) { fprintf(stderr, "ERROR: relation file missing or shorter than -R/-S\n"); return 1; }
    } else if (in.full_range) { // main.cu:190-201
        printf("Creating relation R with %lu tuples (%lu MB) using non-unique keys and full range : ", ja.R_els, R_mb);
        fflush(stdout);
        rc |= hj_create_relation_nonunique(ja.R_filename, ja.R, ja.R_els, INT_MAX);
        printf("Creating relation S with %lu tuples (%lu MB) using non-unique keys and full range : ", ja.S_els, S_mb);
        fflush(stdout);
        rc |= hj_create_relation_fk_from_pk(ja.S_filename, ja.S, ja.S_els, ja.R, ja.R_els);
    } else if (in.unique_keys) { // main.cu:203-249
        printf("Creating relation R with %lu tuples (%lu MB) using unique keys : ", ja.R_els, R_mb);
        fflush(stdout);
        if (in.R_mult == 1) {
            rc |= hj_create_relation_unique(ja.R_filename, ja.R, ja.R_els, (int64_t)ja.R_els);
        } else {
            int32_t *q = (int32_t *)malloc((base_R ? base_R : 1) * sizeof(int32_t));
            rc |= hj_create_relation_unique(ja.R_filename, q, base_R, (int64_t)base_R);
            rc |= hj_create_relation_n(q, ja.R, base_R, (uint64_t)in.R_mult);
            free(q);
        }
        int32_t *dstS = ja.S;
        uint64_t genS = ja.S_els;
        int64_t maxS = (int64_t)ja.R_els; // S draws foreign keys from R's key range (main.cu:220,229)
        int32_t *q = nullptr;
        if (in.S_mult > 1) {
            q = (int32_t *)malloc((base_S ? base_S : 1) * sizeof(int32_t));
            dstS = q; genS = base_S; maxS = (int64_t)base_S; // main.cu:236,243: the base relation is its own range
        }
        if (in.skew > 0) {
            printf("Creating relation S with %lu tuples (%lu MB) using unique keys and skew %f : ", ja.S_els, S_mb, in.skew);
            fflush(stdout);
            rc |= hj_create_relation_zipf(ja.S_filename, dstS, genS, maxS, in.skew);
        } else {
            printf("Creating relation S with %lu tuples (%lu MB) using unique keys : ", ja.S_els, S_mb);
            fflush(stdout);
            rc |= hj_create_relation_unique(ja.S_filename, dstS, genS, maxS);
        }
        if (q) { rc |= hj_create_relation_n(q, ja.S, base_S, (uint64_t)in.S_mult); free(q); }
    } else { // main.cu:250-261: uniform in [0,|R|/2) → about two tuples per value
        printf("Creating relation R with %lu tuples (%lu MB) using non-unique keys : ", ja.R_els, R_mb);
        fflush(stdout);
        rc |= hj_create_relation_nonunique(ja.R_filename, ja.R, ja.R_els, (int64_t)(ja.R_els / 2));
        printf("Creating relation S with %lu tuples (%lu MB) using non-unique keys : ", ja.S_els, S_mb);
        fflush(stdout);
        rc |= hj_create_relation_nonunique(ja.S_filename, ja.S, ja.S_els, (int64_t)(ja.R_els / 2));
    }
    printf("\n");
    fflush(stdout);
    if (rc) { fprintf(stderr, "ERROR: relation generation failed\n"); return 1; }

    int status = 0;
    if (in.option == 7) { // main.cu:264-298
        ja.sharedMem = (unsigned)in.shared_mem;
        ja.threadsNum = in.threads;
        ja.pivotsNum = in.pivots;
        printf("%s : shareMemory = %u\t#threads = %d\n", in.alg->name, ja.sharedMem, ja.threadsNum);
        fflush(stdout);
        timingInfo time;
        memset(&time, 0, sizeof time);
        time.n = 5;
        gettimeofday(&time.start[time.n - 1], nullptr);
        in.alg->fn(&ja, &time);
        gettimeofday(&time.end[time.n - 1], nullptr);
        hj_last_result res;
        hj_reference_last_result(&res);
        status = res.status ? 10 : 0; // CHK_ERROR's print-and-exit (common.h:132-141) lives here, not in the library
    }
    if (pin_r) (void)hipHostFree(ja.R); else free(ja.R);
    if (pin_s) (void)hipHostFree(ja.S); else free(ja.S);
    return status;
}
