// hj_host.cpp — the host-only half of libhj.so: the level-0 write-combining split of the co-processing path and the
// shard function.  Plain C++ (no HIP runtime): this file, gen_ethz.cpp and bench_main.cpp's option parsing are what
// `make asan` builds with -fsanitize=address,undefined for the CPU test leg (tests/test_asan.py).
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "hj.h"
#include "hj_host.h"

namespace hj {

uint32_t host_shard_of(int32_t key, uint32_t nshards) {
    uint32_t h = (uint32_t)key;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return (uint32_t)(((uint64_t)h * nshards) >> 32);
}

namespace {

// Host-side level-0 split (the role of partitions_host_omp_nontemporal_payload, partition-primitives.cu:40-125, and
// partition_prepare/do_payload :129-232): per-thread histograms over contiguous chunks, prefix, then a scatter through
// per-thread SOFTWARE WRITE-COMBINING buffers — one 64-byte line of keys and one of payloads per partition, mirroring
// the 64-byte-aligned destination line being filled — flushed with non-temporal AVX2 stores, so the output lines are
// never read into the cache (the reference's scheme; it keeps 256-tuple batches per partition, LOG_BATCH, and assumes
// aligned outputs; here a run may start anywhere: its first line is written with plain stores).
// Partition id = hj_shard_of(key, parts) (hash: balanced for dense keys).
constexpr uint32_t HWC = 16; // tuples per 64-byte line

// Streaming (non-temporal) line flush: AVX2 on x86-64 hosts that have it (checked once at run time); everywhere else the
// lines leave with plain stores — same result, the destination lines are then read for ownership first.
#if defined(__x86_64__)
__attribute__((target("avx2"))) void wc_flush_line(int32_t *dst, const int32_t *src) {
    _mm256_stream_si256(reinterpret_cast<__m256i *>(dst), _mm256_load_si256(reinterpret_cast<const __m256i *>(src)));
    _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + 8), _mm256_load_si256(reinterpret_cast<const __m256i *>(src + 8)));
}
bool host_has_streaming_stores() { static const bool ok = __builtin_cpu_supports("avx2"); return ok; }
void wc_fence() { _mm_sfence(); }
#else
void wc_flush_line(int32_t *dst, const int32_t *src) { memcpy(dst, src, HWC * 4); }
bool host_has_streaming_stores() { return false; }
void wc_fence() {}
#endif

// Partition ids of a block of keys.  The hash (murmur finaliser + multiplicative range reduction) is what both passes of the split
// spent their time on — the split is compute-bound, not memory-bound: keys only and keys + payloads run at the same rate — so it is
// computed ONCE, in the histogram pass, eight keys at a time (AVX2 where the host has it), and kept as one or two bytes per tuple
// for the scatter pass.
template <typename ID>
void shard_block_scalar(const int32_t *K, uint64_t cnt, uint32_t parts, ID *ids) {
    for (uint64_t j = 0; j < cnt; j++) ids[j] = (ID)host_shard_of(K[j], parts);
}
#if defined(__x86_64__)
template <typename ID>
__attribute__((target("avx2"))) void shard_block_avx2(const int32_t *K, uint64_t cnt, uint32_t parts, ID *ids) {
    const __m256i c1 = _mm256_set1_epi32((int)0x85ebca6bu), c2 = _mm256_set1_epi32((int)0xc2b2ae35u), pn = _mm256_set1_epi32((int)parts);
    uint64_t j = 0;
    for (; j + 8 <= cnt; j += 8) {
        __m256i h = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(K + j));
        h = _mm256_xor_si256(h, _mm256_srli_epi32(h, 16)); h = _mm256_mullo_epi32(h, c1);
        h = _mm256_xor_si256(h, _mm256_srli_epi32(h, 13)); h = _mm256_mullo_epi32(h, c2);
        h = _mm256_xor_si256(h, _mm256_srli_epi32(h, 16));
        // (h * parts) >> 32 per lane: 32 x 32 -> 64-bit products of the even and of the odd lanes
        const __m256i ev = _mm256_srli_epi64(_mm256_mul_epu32(h, pn), 32);
        const __m256i od = _mm256_mul_epu32(_mm256_srli_epi64(h, 32), pn); // the high halves are the results of the odd lanes
        const __m256i r = _mm256_blend_epi32(ev, od, 0xAA);
        alignas(32) uint32_t t[8];
        _mm256_store_si256(reinterpret_cast<__m256i *>(t), r);
        for (int e = 0; e < 8; e++) ids[j + e] = (ID)t[e];
    }
    for (; j < cnt; j++) ids[j] = (ID)host_shard_of(K[j], parts);
}
#endif
template <typename ID>
void shard_block(const int32_t *K, uint64_t cnt, uint32_t parts, ID *ids) {
#if defined(__x86_64__)
    if (host_has_streaming_stores()) { shard_block_avx2<ID>(K, cnt, parts, ids); return; } // (the same run-time AVX2 check)
#endif
    shard_block_scalar<ID>(K, cnt, parts, ids);
}

// false: the write-combining buffers could not be allocated (nothing was written)
template <typename ID>
bool wc_scatter_chunk(const int32_t *K, const int32_t *Pv, const ID *ids, uint64_t lo, uint64_t hi, uint32_t parts,
                                                      const uint64_t *start, int32_t *oK, int32_t *oP, bool stream) {
    // line[p]: 64-byte-aligned output position of the line being filled; fill[p]: next slot; first[p]: first valid slot
    std::vector<uint64_t> line(parts);
    std::vector<uint8_t> fill(parts), first(parts);
    int32_t *bufK = (int32_t *)aligned_alloc(64, (size_t)parts * HWC * 4), *bufP = oP ? (int32_t *)aligned_alloc(64, (size_t)parts * HWC * 4) : nullptr;
    if (!bufK || (oP && !bufP)) { free(bufK); free(bufP); return false; }
    for (uint32_t p = 0; p < parts; p++) {
        line[p] = start[p] & ~(uint64_t)(HWC - 1);
        fill[p] = first[p] = (uint8_t)(start[p] & (HWC - 1));
    }
    for (uint64_t i = lo; i < hi; i++) {
        const int32_t key = K[i];
        const uint32_t p = ids[i];
        uint32_t s = fill[p];
        bufK[p * HWC + s] = key;
        if (oP) bufP[p * HWC + s] = Pv ? Pv[i] : 1;
        if (++s == HWC) { // the line is complete: it leaves with streaming stores (no read-for-ownership of the destination)
            const uint64_t o = line[p];
            if (first[p] == 0 && stream) {
                wc_flush_line(oK + o, bufK + p * HWC);
                if (oP) wc_flush_line(oP + o, bufP + p * HWC);
            } else {
                for (uint32_t j = first[p]; j < HWC; j++) { oK[o + j] = bufK[p * HWC + j]; if (oP) oP[o + j] = bufP[p * HWC + j]; }
                first[p] = 0;
            }
            line[p] = o + HWC;
            s = 0;
        }
        fill[p] = (uint8_t)s;
    }
    for (uint32_t p = 0; p < parts; p++) // half-full lines
        for (uint32_t j = first[p]; j < fill[p]; j++) { oK[line[p] + j] = bufK[p * HWC + j]; if (oP) oP[line[p] + j] = bufP[p * HWC + j]; }
    wc_fence();
    free(bufK);
    free(bufP);
    return true;
}

} // namespace

int host_numa_nodes() {
    int nodes = 0;
    for (int i = 0; i < 64; i++) {
        char path[96];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", i);
        if (FILE *f = fopen(path, "r")) { nodes++; fclose(f); }
    }
    return nodes;
}

std::vector<int> host_node_cpus(int node) {
    std::vector<int> out;
    char path[96], buf[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return out;
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) return out;
    cpu_set_t mine;
    CPU_ZERO(&mine);
    const bool have_mask = sched_getaffinity(0, sizeof mine, &mine) == 0;
    for (char *p = buf; *p && *p != '\n';) { // "0-63,128-191"
        char *e;
        long a = strtol(p, &e, 10), b = a;
        if (e == p) break;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (!have_mask || CPU_ISSET((int)c, &mine)) out.push_back((int)c);
        p = (*e == ',') ? e + 1 : e;
    }
    return out;
}

// false: a host thread could not be started (pids limit of the container) or a write-combining buffer could not be allocated:
// nothing usable was written (the callers report HJ_ENOMEM)
template <typename ID>
static bool level0_split_t(const int32_t *K, const int32_t *Pv, uint64_t n, uint32_t parts, uint32_t threads,
                           int32_t *oK, int32_t *oP, std::vector<uint64_t> &off, const std::vector<int> *pin_cpus) {
    if (threads < 1) threads = 1;
    ID *ids = (ID *)malloc((size_t)(n ? n : 1) * sizeof(ID)); // partition id of every tuple: written by the histogram pass, read by the scatter
    if (!ids) return false;
    struct Free { void *p; ~Free() { free(p); } } free_ids{ids};
    std::vector<uint64_t> hist((size_t)threads * parts, 0);
    auto chunk = [&](uint32_t t, uint64_t &lo, uint64_t &hi) { lo = n * t / threads; hi = n * (t + 1) / threads; };
    // run f(t) for t = 0..threads-1 on that many host threads; thread 0's share runs on the caller
    const bool pin = pin_cpus && !pin_cpus->empty();
    auto bind_self = [&] { // the worker may run on any CPU of the staging buffers' node
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int c : *pin_cpus) CPU_SET(c, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    };
    auto parallel = [&](auto f) -> bool {
        std::vector<std::thread> th;
        bool ok = true;
        try {
            for (uint32_t t = pin ? 0 : 1; t < threads; t++) // pinned: every share on a worker (the caller's affinity stays as it is)
                th.emplace_back([&, t] { if (pin) bind_self(); f(t); });
        } catch (const std::system_error &) { ok = false; }
        if (ok && !pin) f(0);
        for (auto &x : th) x.join();
        return ok;
    };
    if (!parallel([&](uint32_t t) {
            uint64_t lo, hi; chunk(t, lo, hi);
            uint64_t *h = hist.data() + (size_t)t * parts;
            // four interleaved sets of counters: consecutive tuples of one partition do not wait for each other's increment
            std::vector<uint32_t> sub((size_t)4 * parts, 0);
            uint32_t *s0 = sub.data(), *s1 = s0 + parts, *s2 = s1 + parts, *s3 = s2 + parts;
            constexpr uint64_t BLK = 4096;
            for (uint64_t b = lo; b < hi; b += BLK) {
                const uint64_t cnt = std::min<uint64_t>(BLK, hi - b);
                shard_block<ID>(K + b, cnt, parts, ids + b);
                uint64_t j = 0;
                for (; j + 4 <= cnt; j += 4) { s0[ids[b + j]]++; s1[ids[b + j + 1]]++; s2[ids[b + j + 2]]++; s3[ids[b + j + 3]]++; }
                for (; j < cnt; j++) s0[ids[b + j]]++;
                if ((b - lo) / BLK % 65536 == 65535) { for (uint32_t p = 0; p < parts; p++) { h[p] += (uint64_t)s0[p] + s1[p] + s2[p] + s3[p]; s0[p] = s1[p] = s2[p] = s3[p] = 0; } } // (32-bit counters)
            }
            for (uint32_t p = 0; p < parts; p++) h[p] += (uint64_t)s0[p] + s1[p] + s2[p] + s3[p];
        })) return false;
    off.assign(parts + 1, 0);
    uint64_t sum = 0;
    for (uint32_t p = 0; p < parts; p++) {
        off[p] = sum;
        for (uint32_t t = 0; t < threads; t++) { uint64_t cnt = hist[(size_t)t * parts + p]; hist[(size_t)t * parts + p] = sum; sum += cnt; }
    }
    off[parts] = sum;
    // 64-byte streaming stores need 64-byte-aligned columns (pinned staging is page-aligned) and a CPU that has them; two
    // threads may share the destination line where their runs of a partition meet: both write their own slots with plain
    // stores (first/last line)
    const bool stream = host_has_streaming_stores() && (((uintptr_t)oK | (uintptr_t)oP) & 63) == 0;
    std::atomic<bool> scattered{true};
    const bool started = parallel([&](uint32_t t) {
        uint64_t lo, hi; chunk(t, lo, hi);
        if (!wc_scatter_chunk<ID>(K, Pv, ids, lo, hi, parts, hist.data() + (size_t)t * parts, oK, oP, stream)) scattered = false;
    });
    return started && scattered;
}

// ---- the same split in ONE pass over the input (the co-processing path, round 5) ----
// The histogram pass exists to make every partition one contiguous run; an upload does not need that.  Here every worker scatters its
// chunk of the input into BLOCKS of tuples that it takes from a private arena as its partitions fill up — the reference's own layout
// idea, fixed-size buckets handed out by bump allocation (join-primitives.cu:138-192), on the host and without atomics: one arena per
// worker.  A partition is then a list of blocks (every block but a worker's last one per partition is full); nothing can overflow
// whatever the key distribution, the input is read once (4 bytes per tuple instead of 4 + 4 + 1), and every flushed line is 64-byte
// aligned (blocks are), so all of them leave with streaming stores.
// one worker of the one-pass split: its chunk of the input goes, partition by partition, into blocks taken from [arena, arena_end) —
// the worker's own part of the staging columns.  Everything by value: the workers share nothing but the input.
//
// The chunk is taken in batches that stay in the cache (>= 128 tuples per partition on average): partition ids of the batch (AVX2),
// a counting sort of the batch into a local buffer (histogram on four interleaved counter sets, prefix, scatter — no data-dependent
// branch per tuple), then every partition's RUN of the batch is appended to its current block: the partly filled line of the partition is
// completed in a 64-byte staging line, whole lines go from the sorted buffer to the block with streaming stores, the rest waits in the
// staging line.  (Until round 5's last version every tuple went through its partition's staging line on its own: a counter update, an
// unpredictable "line full" branch and a 64-byte load behind sixteen 4-byte stores per line — ~10 cycles per tuple.)
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void wc_stream_lines(int32_t *dst, const int32_t *src, uint64_t lines) { // dst 64-byte aligned, src anywhere
    for (uint64_t i = 0; i < lines; i++) {
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i * 16), _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i * 16)));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i * 16 + 8), _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i * 16 + 8)));
    }
}
#else
static void wc_stream_lines(int32_t *dst, const int32_t *src, uint64_t lines) { memcpy(dst, src, lines * HWC * 4); }
#endif

static bool split_blocks_worker(const int32_t *K, const int32_t *Pv, const uint64_t lo, const uint64_t hi, uint64_t arena, const uint64_t arena_end,
                                const uint32_t parts, const uint32_t block, int32_t *oK, int32_t *oP, const bool stream, std::vector<HostBlock> *out,
                                std::atomic<uint64_t> *upto) {
    const bool pay = oP && Pv;
    const uint32_t batch = std::min<uint32_t>(1u << 18, std::max<uint32_t>(4096, parts * 128));
    std::vector<uint64_t> cur(parts, 0);        // start of the partition's current block (tuples)
    std::vector<uint32_t> cnt(parts, block);    // tuples in it (flushed + staged); block = "needs one"
    std::vector<uint8_t> opened(parts, 0);
    std::vector<uint16_t> ids(batch);
    std::vector<uint32_t> hist((size_t)4 * parts), pos(parts);
    std::vector<int32_t> sk(batch + 16), sp(pay ? batch + 16 : 0);
    int32_t *bufK = (int32_t *)aligned_alloc(64, (size_t)parts * HWC * 4), *bufP = pay ? (int32_t *)aligned_alloc(64, (size_t)parts * HWC * 4) : nullptr;
    if (!bufK || (pay && !bufP)) { free(bufK); free(bufP); return false; }
    std::vector<HostBlock> blocks;
    bool good = true;
    uint64_t low = arena; // the lowest block still open: everything below it is complete
    for (uint64_t b = lo; b < hi && good; b += batch) {
        const uint32_t m = (uint32_t)std::min<uint64_t>(batch, hi - b);
        shard_block<uint16_t>(K + b, m, parts, ids.data());
        std::fill(hist.begin(), hist.end(), 0u);
        {
            uint32_t *h0 = hist.data(), *h1 = h0 + parts, *h2 = h1 + parts, *h3 = h2 + parts;
            uint32_t j = 0;
            for (; j + 4 <= m; j += 4) { h0[ids[j]]++; h1[ids[j + 1]]++; h2[ids[j + 2]]++; h3[ids[j + 3]]++; }
            for (; j < m; j++) h0[ids[j]]++;
            uint32_t run = 0;
            for (uint32_t p = 0; p < parts; p++) { const uint32_t c = h0[p] + h1[p] + h2[p] + h3[p]; pos[p] = run; h0[p] = run; h1[p] = c; run += c; } // h0 = start, h1 = length
        }
        if (pay) for (uint32_t j = 0; j < m; j++) { const uint32_t o = pos[ids[j]]++; sk[o] = K[b + j]; sp[o] = Pv[b + j]; }
        else for (uint32_t j = 0; j < m; j++) sk[pos[ids[j]]++] = K[b + j];
        const uint32_t *start = hist.data(), *length = hist.data() + parts;
        for (uint32_t p = 0; p < parts && good; p++) {
            const int32_t *srcK = sk.data() + start[p], *srcP = pay ? sp.data() + start[p] : nullptr;
            uint32_t len = length[p];
            while (len) {
                uint32_t c = cnt[p];
                if (c == block) { // the partition needs a (new) block
                    const bool was_low = opened[p] && cur[p] == low;
                    if (opened[p]) blocks.push_back(HostBlock{p, cur[p], c});
                    if (arena + block > arena_end) { good = false; break; } // cannot happen: the arena holds every case
                    cur[p] = arena; arena += block; c = 0; opened[p] = 1;
                    if (was_low) { // the lowest open block has just been closed: publish the new complete prefix of the arena
                        low = cur[p];
                        for (uint32_t q = 0; q < parts; q++) if (opened[q] && cur[q] < low) low = cur[q];
                        if (upto) { wc_fence(); upto->store(low, std::memory_order_release); }
                    }
                }
                uint32_t take = std::min(len, block - c);
                len -= take;
                const uint32_t f = c & (HWC - 1);
                if (f) { // complete the partition's partly filled line in its staging line first
                    const uint32_t n1 = std::min(take, HWC - f);
                    memcpy(bufK + p * HWC + f, srcK, n1 * 4);
                    if (pay) memcpy(bufP + p * HWC + f, srcP, n1 * 4);
                    if (f + n1 == HWC) {
                        const uint64_t o = cur[p] + c - f;
                        if (stream) { wc_flush_line(oK + o, bufK + p * HWC); if (pay) wc_flush_line(oP + o, bufP + p * HWC); }
                        else { memcpy(oK + o, bufK + p * HWC, HWC * 4); if (pay) memcpy(oP + o, bufP + p * HWC, HWC * 4); }
                    }
                    c += n1; take -= n1; srcK += n1; if (pay) srcP += n1;
                }
                const uint32_t lines = take / HWC; // c is a whole number of lines here (or take is 0)
                if (lines) {
                    const uint64_t o = cur[p] + c;
                    if (stream) { wc_stream_lines(oK + o, srcK, lines); if (pay) wc_stream_lines(oP + o, srcP, lines); }
                    else { memcpy(oK + o, srcK, (size_t)lines * HWC * 4); if (pay) memcpy(oP + o, srcP, (size_t)lines * HWC * 4); }
                    c += lines * HWC; take -= lines * HWC; srcK += lines * HWC; if (pay) srcP += lines * HWC;
                }
                if (take) { // the rest opens the next line in the staging line
                    memcpy(bufK + p * HWC, srcK, take * 4);
                    if (pay) memcpy(bufP + p * HWC, srcP, take * 4);
                    c += take; srcK += take; if (pay) srcP += take;
                }
                cnt[p] = c;
            }
        }
    }
    for (uint32_t p = 0; p < parts && good; p++) {
        if (!opened[p]) continue;
        const uint32_t tail = cnt[p] & (HWC - 1);
        const uint64_t o = cur[p] + cnt[p] - tail;
        for (uint32_t j = 0; j < tail; j++) { oK[o + j] = bufK[p * HWC + j]; if (pay) oP[o + j] = bufP[p * HWC + j]; }
        blocks.push_back(HostBlock{p, cur[p], cnt[p]});
    }
    wc_fence();
    free(bufK); free(bufP);
    *out = std::move(blocks);
    return good;
}

uint32_t host_split_workers(uint64_t n, uint32_t threads) {
    // a worker per 2^16 tuples at least: every worker costs a partly filled block per partition, and a small input does not pay for threads
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::max(1u, threads), n >> 16));
}

uint32_t host_split_block_size(uint64_t n, uint32_t parts, uint32_t workers) {
    // partly filled blocks (one per partition and worker at most) cost capacity, small blocks cost uploads: an eighth of the mean
    // (worker, partition) share, a power of two between 2^8 and 2^20 tuples
    const uint64_t share = n / ((uint64_t)std::max(1u, workers) * std::max(1u, parts) * 8);
    uint32_t b = 256;
    while (b < (1u << 20) && (uint64_t)b * 2 <= share) b *= 2;
    return b;
}

// a worker needs at most ceil(chunk / block) + parts blocks (every partition may end on a partly filled one)
static uint64_t split_arena_tuples(uint64_t chunk, uint32_t parts, uint32_t block) { return ((chunk + block - 1) / block + parts) * (uint64_t)block; }

uint64_t host_split_blocks_capacity(uint64_t n, uint32_t parts, uint32_t threads) {
    threads = host_split_workers(n, threads);
    const uint32_t block = host_split_block_size(n, parts, threads);
    uint64_t cap = 0;
    for (uint32_t t = 0; t < threads; t++) cap += split_arena_tuples(n * (t + 1) / threads - n * t / threads, parts, block);
    return cap;
}

bool host_level0_split_blocks(const int32_t *K, const int32_t *Pv, uint64_t n, uint32_t parts, uint32_t threads, int32_t *oK, int32_t *oP,
                              std::vector<HostBlock> &blocks, std::vector<uint64_t> &part_size, const std::vector<int> *pin_cpus,
                              const std::function<void(const HostSplitProgress &)> *while_running) {
    if (parts > 4096) return false;
    threads = host_split_workers(n, threads);
    const uint32_t block = host_split_block_size(n, parts, threads);
    const bool stream = host_has_streaming_stores() && (((uintptr_t)oK | (uintptr_t)oP) & 63) == 0;
    const bool pin = pin_cpus && !pin_cpus->empty();
    std::vector<std::vector<HostBlock>> mine(threads);
    std::vector<char> okv(threads, 1);
    HostSplitProgress prog;
    std::vector<uint64_t> &astart = prog.arena;
    astart.assign(threads + 1, 0);
    for (uint32_t t = 0; t < threads; t++) astart[t + 1] = astart[t] + split_arena_tuples(n * (t + 1) / threads - n * t / threads, parts, block);
    prog.upto.reset(new std::atomic<uint64_t>[(size_t)threads * 8]);
    for (uint32_t t = 0; t < threads; t++) prog.upto[(size_t)t * 8].store(astart[t], std::memory_order_relaxed);
    std::atomic<uint32_t> finished{0};
    {
        std::vector<std::thread> th;
        bool started = true;
        auto run = [=, &mine, &okv, &astart, &prog, &finished](uint32_t t) {
            if (pin) {
                cpu_set_t set;
                CPU_ZERO(&set);
                for (int c : *pin_cpus) CPU_SET(c, &set);
                (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
            }
            try {
                okv[t] = split_blocks_worker(K, Pv, n * t / threads, n * (t + 1) / threads, astart[t], astart[t + 1], parts, block, oK, oP, stream, &mine[t],
                                             &prog.upto[(size_t)t * 8]) ? 1 : 0;
            } catch (const std::bad_alloc &) { okv[t] = 0; } // (a worker's own vectors: the split reports HJ_ENOMEM instead of terminating the process)
            finished.fetch_add(1, std::memory_order_release);
        };
        const bool own_share = !pin && !while_running; // the calling thread is worker 0
        try {
            for (uint32_t t = own_share ? 1 : 0; t < threads; t++) th.emplace_back(run, t);
        } catch (const std::system_error &) { started = false; }
        if (started && own_share) run(0);
        if (while_running) {
            while (finished.load(std::memory_order_acquire) < th.size()) {
                (*while_running)(prog);
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
        for (auto &x : th) x.join();
        if (while_running && started) (*while_running)(prog);
        if (!started) return false;
    }
    for (char c : okv) if (!c) return false;
    blocks.clear();
    part_size.assign(parts, 0);
    for (uint32_t t = 0; t < threads; t++)
        for (const HostBlock &hb : mine[t]) { blocks.push_back(hb); part_size[hb.part] += hb.count; }
    // by partition, then by address: neighbours in the staging columns stay neighbours (the caller merges them into one upload)
    std::sort(blocks.begin(), blocks.end(), [](const HostBlock &a, const HostBlock &b) { return a.part != b.part ? a.part < b.part : a.start < b.start; });
    return true;
}

bool host_level0_split(const int32_t *K, const int32_t *Pv, uint64_t n, uint32_t parts, uint32_t threads,
                       int32_t *oK, int32_t *oP, std::vector<uint64_t> &off, const std::vector<int> *pin_cpus) {
    return parts <= 256 ? level0_split_t<uint8_t>(K, Pv, n, parts, threads, oK, oP, off, pin_cpus)
                        : level0_split_t<uint16_t>(K, Pv, n, parts, threads, oK, oP, off, pin_cpus);
}


} // namespace hj

using namespace hj;

static std::atomic<int> g_split_test_progress{0};

extern "C" {

int hj_host_split(const int32_t *keys, const int32_t *pays, uint64_t n, uint32_t parts, uint32_t threads, int32_t *out_keys,
                  int32_t *out_pays, uint64_t *offsets, double *gbs) {
    if ((n && (!keys || !out_keys)) || !offsets || parts == 0 || parts > 4096) return HJ_EINVAL;
    if (threads == 0) threads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    std::vector<uint64_t> off;
    const auto t0 = std::chrono::steady_clock::now();
    if (!host_level0_split(keys, pays, n, parts, threads, out_keys, out_pays, off)) return HJ_ENOMEM;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (uint32_t p = 0; p <= parts; p++) offsets[p] = off[p];
    if (gbs) *gbs = dt > 0 ? (out_pays ? 16.0 : 8.0) * (double)n / dt / 1e9 : 0;
    return HJ_OK;
}

/* tests only: the next hj_host_split_blocks calls copy out every range the split publishes while
 * it runs and check afterwards that it was final (HJ_EIO otherwise; *gbs then returns the tuples published). */
int hj_host_split_debug_progress(int on) { g_split_test_progress.store(on); return HJ_OK; }

uint64_t hj_host_split_blocks_capacity(uint64_t n, uint32_t parts, uint32_t threads) {
    if (parts == 0 || parts > 4096) return 0;
    if (threads == 0) threads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    return host_split_blocks_capacity(n, parts, threads);
}

int hj_host_split_blocks(const int32_t *keys, const int32_t *pays, uint64_t n, uint32_t parts, uint32_t threads, int32_t *out_keys,
                         int32_t *out_pays, uint64_t cap, uint32_t *block_part, uint64_t *block_start, uint32_t *block_count,
                         uint64_t max_blocks, uint64_t *n_blocks, double *gbs) {
    if ((n && (!keys || !out_keys)) || !n_blocks || parts == 0 || parts > 4096) return HJ_EINVAL;
    if (threads == 0) threads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    if (cap < host_split_blocks_capacity(n, parts, threads)) return HJ_ECAPACITY;
    std::vector<HostBlock> blocks;
    std::vector<uint64_t> psize;
    // hj_host_split_debug_progress(1) (tests; an environment variable read here until round 5): take the complete prefixes the split publishes while it runs the way an uploader would —
    // copy them out at once — and check afterwards that what was copied was final
    struct Snap { uint64_t start; std::vector<int32_t> k; };
    std::vector<Snap> snaps;
    std::vector<uint64_t> sent, arena;
    std::function<void(const HostSplitProgress &)> snoop = [&](const HostSplitProgress &pg) {
        if (sent.empty()) { sent.assign(pg.arena.begin(), pg.arena.end() - 1); arena = pg.arena; }
        for (uint32_t t = 0; t + 1 < pg.arena.size(); t++) {
            const uint64_t d = pg.done(t);
            if (d > sent[t]) { snaps.push_back(Snap{sent[t], std::vector<int32_t>(out_keys + sent[t], out_keys + d)}); sent[t] = d; }
        }
    };
    const bool snooping = g_split_test_progress.load(std::memory_order_relaxed) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (!host_level0_split_blocks(keys, pays, n, parts, threads, out_keys, out_pays, blocks, psize, nullptr, snooping ? &snoop : nullptr)) return HJ_ENOMEM;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (snooping) {
        uint64_t covered = 0, in_full_blocks = 0;
        for (const Snap &sn : snaps) {
            if (memcmp(sn.k.data(), out_keys + sn.start, sn.k.size() * 4) != 0) return HJ_EIO; // published before it was complete
            covered += sn.k.size();
        }
        const uint32_t bs = host_split_block_size(n, parts, host_split_workers(n, threads));
        for (const HostBlock &hb : blocks) { // everything below a worker's mark is whole, FULL blocks
            const size_t t = (size_t)(std::upper_bound(arena.begin(), arena.end(), hb.start) - arena.begin()) - 1;
            if (!sent.empty() && hb.start < sent[t]) { if (hb.count != bs || hb.start + bs > sent[t]) return HJ_EIO; in_full_blocks += hb.count; }
        }
        if (covered != in_full_blocks) return HJ_EIO;
        if (gbs) *gbs = (double)covered; // (the test reads how much was published)
        gbs = nullptr;
    }
    *n_blocks = blocks.size();
    if (blocks.size() > max_blocks || (blocks.size() && (!block_part || !block_start || !block_count))) return HJ_ECAPACITY;
    for (size_t i = 0; i < blocks.size(); i++) { block_part[i] = blocks[i].part; block_start[i] = blocks[i].start; block_count[i] = (uint32_t)blocks[i].count; }
    if (gbs) *gbs = dt > 0 ? (out_pays && pays ? 16.0 : 8.0) * (double)n / dt / 1e9 : 0;
    return HJ_OK;
}

uint32_t hj_shard_of(int32_t key, uint32_t nshards) { return host_shard_of(key, nshards); }

// ---- a CPU radix join out of the library's own host code (a reported baseline, never a fallback): what these host cores do with the
// scheme of the GPU path.  Level 1: both relations through the one-pass block split above (up to 4096 partitions by hj_shard_of, software
// write-combining, streaming stores — partition-primitives.cu:40-125's idea).  Level 2, one partition pair per thread at a time: a counting
// sort of both sides by further hash bits into pieces whose build side fits the L1 cache, then a bucket-chained table per piece, built
// and probed like the reference's joinCpu (hash_join_clustered_probe.cu:2013-2059: head/next arrays, one pass over each side).  Payloads NULL = ones.
// matches and aggregate as hj_join defines them; *seconds = wall time of the join proper (allocation of the staging columns is outside it,
// as the device allocations are outside the GPU's timed region). ----
int hj_host_join(const int32_t *keysR, const int32_t *paysR, uint64_t nR, const int32_t *keysS, const int32_t *paysS, uint64_t nS,
                 uint32_t threads, uint64_t *matches, uint64_t *agg, double *seconds) {
    if ((nR && !keysR) || (nS && !keysS)) return HJ_EINVAL;
    if (threads == 0) threads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    if (matches) *matches = 0;
    if (agg) *agg = 0;
    if (seconds) *seconds = 0;
    if (nR == 0 || nS == 0) return HJ_OK;
    const uint64_t nmin = std::min(nR, nS);
    uint32_t parts = 1;
    while (parts < 4096 && nmin / parts > ((uint64_t)1 << 17)) parts <<= 1; // ~2^17 build tuples per level-1 partition (1 MiB: the per-core L2)
    struct Side { const int32_t *k, *p; uint64_t n; int32_t *ok = nullptr, *op = nullptr; std::vector<HostBlock> blocks; std::vector<uint64_t> psize; std::vector<size_t> first; };
    Side sd[2] = {{keysR, paysR, nR}, {keysS, paysS, nS}};
    int rc = HJ_OK;
    for (Side &x : sd) {
        const uint64_t cap = host_split_blocks_capacity(x.n, parts, threads);
        x.ok = (int32_t *)aligned_alloc(64, ((size_t)cap * 4 + 63) & ~(size_t)63);
        x.op = x.p ? (int32_t *)aligned_alloc(64, ((size_t)cap * 4 + 63) & ~(size_t)63) : nullptr;
        if (!x.ok || (x.p && !x.op)) rc = HJ_ENOMEM;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (Side &x : sd) {
        if (rc) break;
        if (!host_level0_split_blocks(x.k, x.p, x.n, parts, threads, x.ok, x.op, x.blocks, x.psize)) { rc = HJ_ENOMEM; break; }
        x.first.assign(parts + 1, x.blocks.size());
        for (size_t i = x.blocks.size(); i-- > 0;) x.first[x.blocks[i].part] = i; // blocks come sorted by (partition, start)
        for (uint32_t p = parts; p-- > 0;) if (x.first[p] == x.blocks.size() || x.blocks[x.first[p]].part != p) x.first[p] = x.first[p + 1];
    }
    std::atomic<uint32_t> next{0};
    std::vector<uint64_t> tm(threads, 0), ta(threads, 0);
    std::vector<char> okv(threads, 1);
    auto work = [&](uint32_t t) {
        try {
            std::vector<int32_t> bk, bp, qk, qp;       // the partition pair, sorted by piece
            std::vector<uint32_t> hb, hq, head, nxt;
            uint64_t m = 0, a = 0;
            for (uint32_t p = next.fetch_add(1); p < parts; p = next.fetch_add(1)) {
                const uint64_t nb = sd[0].psize[p], nq = sd[1].psize[p];
                if (!nb || !nq) continue;
                const bool r_builds = nb <= nq; // the smaller side of the pair builds
                const Side &B = sd[r_builds ? 0 : 1], &Q = sd[r_builds ? 1 : 0];
                const uint64_t cb = r_builds ? nb : nq, cq = r_builds ? nq : nb;
                uint32_t pieces = 1, lg = 0;
                while (pieces < 4096 && cb / pieces > 1024) { pieces <<= 1; lg++; } // ~1024 build tuples per piece: table + tuples in the L1
                auto piece_of = [&](int32_t key) { uint32_t h = (uint32_t)key * 0x9E3779B1u; return (h >> 7) & (pieces - 1); }; // (bits hj_shard_of's murmur does not share)
                auto sort_side = [&](const Side &X, uint64_t cnt, std::vector<int32_t> &ok2, std::vector<int32_t> &op2, std::vector<uint32_t> &hist) {
                    hist.assign(pieces + 1, 0);
                    ok2.resize(cnt); if (X.p) op2.resize(cnt);
                    for (size_t i = X.first[p]; i < X.first[p + 1]; i++) {
                        const HostBlock &hbk = X.blocks[i];
                        for (uint64_t j = 0; j < hbk.count; j++) hist[piece_of(X.ok[hbk.start + j]) + 1]++;
                    }
                    for (uint32_t q = 0; q < pieces; q++) hist[q + 1] += hist[q];
                    std::vector<uint32_t> cur(hist.begin(), hist.end() - 1);
                    for (size_t i = X.first[p]; i < X.first[p + 1]; i++) {
                        const HostBlock &hbk = X.blocks[i];
                        for (uint64_t j = 0; j < hbk.count; j++) {
                            const int32_t key = X.ok[hbk.start + j];
                            const uint32_t o = cur[piece_of(key)]++;
                            ok2[o] = key;
                            if (X.p) op2[o] = X.op[hbk.start + j];
                        }
                    }
                };
                sort_side(B, cb, bk, bp, hb);
                sort_side(Q, cq, qk, qp, hq);
                for (uint32_t q = 0; q < pieces; q++) {
                    const uint32_t b0 = hb[q], b1 = hb[q + 1], q0 = hq[q], q1 = hq[q + 1];
                    if (b0 == b1 || q0 == q1) continue;
                    uint32_t nh = 256;
                    while (nh < (b1 - b0)) nh <<= 1;
                    head.assign(nh, 0xFFFFFFFFu);
                    nxt.resize(b1 - b0);
                    for (uint32_t i = b0; i < b1; i++) {
                        const uint32_t h = (((uint32_t)bk[i] * 0x9E3779B1u) >> (7 + lg)) & (nh - 1);
                        nxt[i - b0] = head[h]; head[h] = i - b0;
                    }
                    for (uint32_t i = q0; i < q1; i++) {
                        const int32_t key = qk[i];
                        const int64_t pq = Q.p ? qp[i] : 1;
                        for (uint32_t e = head[(((uint32_t)key * 0x9E3779B1u) >> (7 + lg)) & (nh - 1)]; e != 0xFFFFFFFFu; e = nxt[e])
                            if (bk[b0 + e] == key) { m++; a += (uint64_t)((int64_t)(B.p ? bp[b0 + e] : 1) * pq); }
                    }
                }
            }
            tm[t] = m; ta[t] = a;
        } catch (const std::bad_alloc &) { okv[t] = 0; }
    };
    if (!rc) {
        std::vector<std::thread> th;
        try {
            for (uint32_t t = 1; t < threads; t++) th.emplace_back(work, t);
        } catch (const std::system_error &) { rc = HJ_ENOMEM; next.store(parts); }
        work(0);
        for (auto &x : th) x.join();
        for (char c : okv) if (!c) rc = HJ_ENOMEM;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (Side &x : sd) { free(x.ok); free(x.op); }
    if (rc) return rc;
    uint64_t m = 0, a = 0;
    for (uint32_t t = 0; t < threads; t++) { m += tm[t]; a += ta[t]; }
    if (matches) *matches = m;
    if (agg) *agg = a;
    if (seconds) *seconds = dt;
    return HJ_OK;
}

} // extern "C"
