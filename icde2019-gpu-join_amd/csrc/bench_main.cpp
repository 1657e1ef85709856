// bench_main.cpp — the `bench` command-line driver: the reference's CLI contract (src/main.cu:80-310,
// 434-557) on top of libhj.so.
//
//   ./bench -b 7 -a HJC -R <n> -S <m> [-s theta] [--non-unique] [--full-range] [--file -k R.bin -l S.bin]
//           [-x mult] [-y mult] [-t -v -m -p -w: accepted, echoed, ignored by HJC like the reference]
//           [--seed N]   (new: reproducible generation; default = time(NULL) like the reference)
//           [--gpus N]   (new: N > 1 = every GPU starts with 1/N of R and S; level-0 split on the GPUs, all-to-all over xGMI
//                         (RCCL, hj_dist.h), independent joins per GPU, counts all-reduced — the structure of
//                         hjcp.cu:1503-1618 across GPUs; refused when fewer than N GPUs are visible)
//           [--cpu-baseline]  (new: also time a CPU chained-hash join on the same columns, as a reported baseline)
//           [--json]     (new: one machine-readable line with the counts and timings at the end)
//
// Differences by design: `-a` is validated (D8: the reference walks an
// unterminated table and calls an uninitialised pointer when -a is omitted); the skew cache file
// is named unique_skew<theta>_S<m>.bin (D7: the reference's sprintf has one argument too few);
// relation files are checked for short reads (D12).
#include <getopt.h>
#ifndef HJ_HOST_ONLY
#include <hip/hip_runtime.h>
#endif
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sys/time.h>

#include <chrono>
#include <thread>
#include <vector>

#include <string>

#include "hj.h"
#include "hj_dist.h"
#include "hj_reference_abi.h"

#ifdef HJ_HOST_ONLY
// `make asan`: the driver without a GPU path (option parsing, cache-file naming, generation: -b 8), linked against the
// host-only sanitizer build of the library.  -b 7 reports that this build cannot join.
extern "C" unsigned int hashJoinClusteredProbe(args *, timingInfo *) { fprintf(stderr, "GPU Error: host-only build\n"); return 0; }
extern "C" void hj_reference_last_result(hj_last_result *out) { memset(out, 0, sizeof *out); out->status = HJ_EHIP; }
#endif

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
    int gpus = 1, json = 0, cpu_baseline = 0;
    const char *transport = nullptr; // --gpus N: "rccl" (default on distinct GPUs) or "copy" (peer copies on the copy engines)
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
                                   {"gpus", required_argument, nullptr, 1004},
                                   {"json", no_argument, nullptr, 1005},
                                   {"cpu-baseline", no_argument, nullptr, 1006},
                                   {"transport", required_argument, nullptr, 1007},
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
        case 1004: in->gpus = atoi(optarg); printf("gpus = %d\t", in->gpus); break;
        case 1005: in->json = 1; printf("json\t"); break;
        case 1006: in->cpu_baseline = 1; printf("cpu-baseline\t"); break;
        case 1007: in->transport = optarg; printf("transport = %s\t", optarg); break;
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
    if (in->R_mult < 1 || in->S_mult < 1 || in->gpus < 1 || in->gpus > 64) usage_exit();
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// --cpu-baseline: a non-partitioned chained hash join on the host, the shape of the reference's (never called)
// joinCpu (hjcp.cu:2013-2059: murmur3 finaliser, LIFO chains, serial build, parallel probe) with the table sized to
// the build side instead of a fixed 2^20 slots.  A reported baseline next to the GPU numbers, never a fallback.
struct CpuJoin { unsigned long long matches = 0; double seconds = 0; unsigned threads = 0; };
inline uint32_t murmur_fin(uint32_t x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; }
CpuJoin cpu_join(const int32_t *R, uint64_t nR, const int32_t *S, uint64_t nS) {
    CpuJoin out;
    uint32_t lg = 10;
    while (((uint64_t)1 << lg) < nR && lg < 31) lg++;
    const uint32_t mask = (uint32_t)(((uint64_t)1 << lg) - 1);
    const double t0 = now_s();
    std::vector<int64_t> head((size_t)mask + 1, -1), next(nR ? nR : 1);
    for (uint64_t j = 0; j < nR; j++) { const uint32_t b = murmur_fin((uint32_t)R[j]) & mask; next[j] = head[b]; head[b] = (int64_t)j; }
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // containers: honour the CPU quota
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            long cpus = (atol(q) + period - 1) / period;
            if (cpus >= 1 && (unsigned long)cpus < nt) nt = (unsigned)cpus;
        }
        fclose(f);
    }
    std::vector<unsigned long long> part(nt, 0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            unsigned long long m = 0;
            for (uint64_t j = nS * t / nt; j < nS * (t + 1) / nt; j++) {
                const int32_t key = S[j];
                for (int64_t cur = head[murmur_fin((uint32_t)key) & mask]; cur >= 0; cur = next[cur]) m += R[cur] == key;
            }
            part[t] = m;
        });
    for (auto &x : th) x.join();
    for (auto m : part) out.matches += m;
    out.seconds = now_s() - t0;
    out.threads = nt;
    return out;
}

#ifndef HJ_HOST_ONLY
// --gpus N (N > 1): the reference joins level-0 partitions independently (hjcp.cu:1503-1618); here every GPU starts with
// 1/N of R and of S (contiguous slices of the host columns, uploaded untimed like hjcp.cu:874-879), and hj_dist_join does
// the rest in C++: level-0 split on the GPUs, sliced all-to-all over xGMI (RCCL), local passes + build/probe, all-reduce.
// Fewer than N visible GPUs: refused (non-zero), nothing is emulated on the host.
// The reference's lead timed run writes its output (hjcp.cu:913, 937-940; per level-0 partition in the co-processing analogue,
// hjcp.cu:1503-1618): with N GPUs every GPU writes the (key, payR, payS) tuples of the partitions it owns into its own columns
// (hj_dist_join_materialize).  The columns are sized from the count run (+10 %); a rank whose share does not fit reports its size
// with HJ_ECAPACITY and the run is repeated with exact sizes.
struct MultiResult { unsigned long long matches = 0, agg = 0, materialized = 0; double seconds = 0, mat_seconds = 0; int status = 0; std::string transport, path, mat_path; hj_dist_stats st{}, mat_st{}; std::vector<uint64_t> n_out; };
MultiResult multi_gpu_join(const args &ja, int gpus, const char *transport) {
    MultiResult out;
    hj_dist *d = nullptr;
    // $HJ_BENCH_SHARE_GPU=1 (tests on a one-GPU box): every rank on device 0 over the in-process device-copy transport — the whole
    // pipeline and this transcript, no link involved; said so on stdout
    std::vector<int> share((size_t)gpus, 0);
    const bool shared = getenv("HJ_BENCH_SHARE_GPU") && atoi(getenv("HJ_BENCH_SHARE_GPU"));
    if (shared) printf("TEST MODE: %d ranks share GPU 0 (HJ_BENCH_SHARE_GPU): not a measurement\n", gpus);
    out.status = hj_dist_create_transport(&d, gpus, shared ? share.data() : nullptr, shared ? nullptr : transport);
    if (out.status) { fprintf(stderr, "GPU Error: --gpus %d%s%s: hj_dist_create_transport failed (code %d): fewer GPUs visible than ranks, no GPU, or a transport these devices do not allow\n", gpus, transport ? " --transport " : "", transport ? transport : "", out.status); return out; }
    out.transport = hj_dist_transport(d);
    std::vector<void *> bufs;
    for (int g = 0; g < gpus && !out.status; g++) {
        hj_ctx *c = hj_dist_context(d, g);
        const uint64_t r0 = ja.R_els * g / gpus, r1 = ja.R_els * (g + 1) / gpus, s0 = ja.S_els * g / gpus, s1 = ja.S_els * (g + 1) / gpus;
        void *col[4] = {nullptr, nullptr, nullptr, nullptr};
        const uint64_t n[2] = {r1 - r0, s1 - s0};
        const int32_t *src[2] = {ja.R + r0, ja.S + s0};
        for (int x = 0; x < 2 && !out.status; x++) {
            out.status = hj_device_malloc(c, &col[2 * x], (n[x] + 16) * 4);
            if (!out.status) out.status = hj_device_malloc(c, &col[2 * x + 1], (n[x] + 16) * 4);
            if (!out.status) { bufs.push_back(col[2 * x]); bufs.push_back(col[2 * x + 1]); }
            if (!out.status && n[x]) out.status = hj_memcpy_h2d(c, col[2 * x], src[x], n[x] * 4);
            if (!out.status) out.status = hj_fill_payload(c, (int32_t *)col[2 * x + 1], n[x], HJ_PAYLOAD_ONES, 0); // hjcp.cu:1994-1999
            if (!out.status) out.status = hj_sync(c);
            if (!out.status) out.status = hj_dist_bind(d, g, x, (const int32_t *)col[2 * x], (const int32_t *)col[2 * x + 1], n[x]);
        }
        if (out.status) fprintf(stderr, "GPU Error: device %d: %s (code %d)\n", g, hj_error(c), out.status);
    }
    if (!out.status) {
        uint64_t m = 0, a = 0;
        out.status = hj_dist_join(d, &m, &a); // warm-up: buffers, communicators, the learned path
        const double t0 = now_s();
        if (!out.status) out.status = hj_dist_join(d, &m, &a);
        out.seconds = now_s() - t0;
        out.matches = m; out.agg = a;
        if (out.status) fprintf(stderr, "GPU Error: %s (code %d)\n", hj_dist_error(d), out.status);
        else { hj_dist_get_stats(d, 0, &out.st); out.path = out.st.path ? "exact" : "sliced"; }
    }
    // the materialising run (timed like the count run: one warm-up, one timed call)
    std::vector<std::vector<void *>> outs((size_t)gpus);
    if (!out.status) {
        std::vector<uint64_t> cap((size_t)gpus, out.matches / (uint64_t)gpus + out.matches / (uint64_t)(10 * gpus) + 4096), n_out((size_t)gpus, 0);
        for (int attempt = 0; attempt < 2 && !out.status; attempt++) {
            for (int g = 0; g < gpus && !out.status; g++) {
                hj_ctx *c = hj_dist_context(d, g);
                for (void *p : outs[g]) hj_device_free(c, p);
                outs[g].assign(3, nullptr);
                for (int j = 0; j < 3 && !out.status; j++) out.status = hj_device_malloc(c, &outs[g][j], (cap[g] + 16) * 4);
                if (!out.status) out.status = hj_dist_bind_output(d, g, (int32_t *)outs[g][0], (int32_t *)outs[g][1], (int32_t *)outs[g][2], cap[g]);
                if (out.status) fprintf(stderr, "GPU Error: device %d: output columns of %llu tuples: %s (code %d)\n", g, (unsigned long long)cap[g], hj_error(c), out.status);
            }
            if (out.status) break;
            uint64_t m = 0;
            int rc = hj_dist_join_materialize(d, &m, nullptr, n_out.data()); // warm-up (first touch of the output columns)
            if (rc == HJ_ECAPACITY && attempt == 0) { for (int g = 0; g < gpus; g++) cap[g] = n_out[g] + 16; continue; }
            const double t0 = now_s();
            if (!rc) rc = hj_dist_join_materialize(d, &m, nullptr, n_out.data());
            out.mat_seconds = now_s() - t0;
            out.status = rc;
            if (rc) { fprintf(stderr, "GPU Error: %s (code %d)\n", hj_dist_error(d), rc); break; }
            if (m != out.matches) { fprintf(stderr, "GPU Error: the materialising join produced %llu tuples, the count join %llu\n", (unsigned long long)m, out.matches); out.status = HJ_EHIP; break; }
            out.n_out = n_out;
            for (uint64_t v : n_out) out.materialized += v;
            hj_dist_get_stats(d, 0, &out.mat_st); out.mat_path = out.mat_st.path ? "exact" : "sliced";
            break;
        }
    }
    for (int g = 0; g < gpus; g++)
        for (void *p : outs[g]) if (p) hj_device_free(hj_dist_context(d, g), p);
    for (int g = 0, i = 0; g < gpus; g++)
        for (int j = 0; j < 4 && i < (int)bufs.size(); j++, i++) hj_device_free(hj_dist_context(d, g), bufs[i]);
    hj_dist_destroy(d);
    return out;
}
#else
struct MultiStats { float exchange_ms = 0; unsigned long long link_bytes = 0; };
struct MultiResult { unsigned long long matches = 0, agg = 0, materialized = 0; double seconds = 0, mat_seconds = 0; int status = HJ_EHIP; std::string transport, path, mat_path; MultiStats st, mat_st; std::vector<uint64_t> n_out; };
MultiResult multi_gpu_join(const args &, int, const char *) { return MultiResult(); }
#endif

int32_t *alloc_col(uint64_t n, bool *pinned) {
    void *p = nullptr;
    (void)p;
    size_t bytes = (size_t)(n ? n : 1) * sizeof(int32_t);
    // main.cu:181-183 (MEM_HOST): pinned, mapped host columns; plain malloc when no GPU is present (-b 8)
#ifndef HJ_HOST_ONLY
    if (hipHostMalloc(&p, bytes, hipHostMallocMapped) == hipSuccess) { *pinned = true; return (int32_t *)p; }
#endif
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
    hj_last_result res;
    memset(&res, 0, sizeof res);
    MultiResult multi;
    if (in.option == 7 && in.gpus > 1) {
        printf("%s : %d GPUs, level-0 split + all-to-all over xGMI (RCCL)\n", in.alg->name, in.gpus);
        fflush(stdout);
        multi = multi_gpu_join(ja, in.gpus, in.transport);
        status = multi.status ? 10 : 0;
        if (!multi.status) {
            const double bytes = 2.0 * (double)(ja.R_els + ja.S_els) * sizeof(int);
            // the reference's transcript (hjcp.cu:937-940, 986-991): the materialising run first, then the count-only one
            printf("With materialization\n");
            printf("Exchange: %s, %s path\n", multi.transport.c_str(), multi.mat_path.c_str());
            printf("Total Throughput (%d GPUs) %f\n", in.gpus, bytes / multi.mat_seconds / 1000 / 1000);
            printf("Output (sharded, one share per GPU):");
            for (size_t g = 0; g < multi.n_out.size(); g++) printf(" %llu", (unsigned long long)multi.n_out[g]);
            printf(" tuples\n");
            printf("%llu results\n", multi.materialized);
            printf("Without materialization\n");
            printf("Exchange: %s, %s path\n", multi.transport.c_str(), multi.path.c_str());
            if (multi.st.exchange_ms > 0) // what ONE link direction sustained while the exchange ran, against its 76.8 GB/s
                printf("Links: %.2f GB to each peer in %.2f ms of exchange = %.1f GB/s per link direction (xGMI: 76.8)\n",
                       multi.st.link_bytes / (double)(in.gpus - 1) / 1e9, multi.st.exchange_ms, multi.st.link_bytes / (double)(in.gpus - 1) / 1e6 / multi.st.exchange_ms);
            printf("Total Throughput (%d GPUs) %f\n", in.gpus, bytes / multi.seconds / 1000 / 1000);
            printf("%llu results\n", multi.agg);
        }
        res.matches = multi.matches; res.agg = multi.agg; res.status = multi.status; res.join_ms[1] = multi.seconds * 1e3;
        res.materialized = multi.materialized; res.join_ms[0] = multi.mat_seconds * 1e3;
    } else
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
        hj_reference_last_result(&res);
        status = res.status ? 10 : 0; // CHK_ERROR's print-and-exit (common.h:132-141) lives here, not in the library
    }
    CpuJoin cpu;
    if (in.option == 7 && in.cpu_baseline) {
        cpu = cpu_join(ja.R, ja.R_els, ja.S, ja.S_els);
        printf("CPU baseline (chained hash join, %u threads): %.3f s, %.1f Mtuples/s, %llu results%s\n", cpu.threads, cpu.seconds,
               (double)(ja.R_els + ja.S_els) / cpu.seconds / 1e6, cpu.matches,
               (!status && cpu.matches != res.matches) ? "  ** differs from the GPU count **" : "");
        if (!status && cpu.matches != res.matches) status = 11;
    }
    if (in.option == 7 && in.json) {
        printf("{\"alg\": \"%s\", \"R\": %lu, \"S\": %lu, \"gpus\": %d, \"status\": %d, \"matches\": %llu, \"agg\": %llu, "
               "\"materialized\": %llu, \"partition_ms\": [%.3f, %.3f], \"join_ms\": [%.3f, %.3f]",
               in.alg->name, (unsigned long)ja.R_els, (unsigned long)ja.S_els, in.gpus, res.status, res.matches, res.agg, res.materialized,
               res.partition_ms[0], res.partition_ms[1], res.join_ms[0], res.join_ms[1]);
        if (in.cpu_baseline) printf(", \"cpu_baseline\": {\"seconds\": %.4f, \"threads\": %u, \"matches\": %llu}", cpu.seconds, cpu.threads, cpu.matches);
        printf("}\n");
    }
#ifndef HJ_HOST_ONLY
    if (pin_r) (void)hipHostFree(ja.R); else free(ja.R);
    if (pin_s) (void)hipHostFree(ja.S); else free(ja.S);
#else
    free(ja.R);
    free(ja.S);
#endif
    return status;
}
