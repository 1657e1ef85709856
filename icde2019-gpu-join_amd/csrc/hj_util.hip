// hj_util.hip — gfx950 device code around the hot path: payload fill (init_payload, jp.cu:30-33), input synthesis on the device
// (unique keys, Zipf), order-independent digests, the partition check, and the non-partitioned comparison baselines
// (jp.cu:628-668 perfect array, jp.cu:681-742 global chained table).
#include "hj_device.h"

namespace hj {

// ------------------------------------------------------------------------------------------------
// utilities: payload fill (init_payload jp.cu:30-33), input synthesis, digests, partition check
// ------------------------------------------------------------------------------------------------
__global__ void k_fill(int32_t *__restrict__ p, uint64_t n, int mode, uint64_t first) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = mode == 1 ? (int32_t)(uint32_t)(first + i) : 1;
}


// bijection on [0, 2^k): odd multiply, xorshift, add — four rounds keyed by the seed; cycle-walked
// down to [0, domain).
__device__ __forceinline__ uint64_t perm_round(uint64_t x, uint64_t mask, uint32_t k, uint64_t m, uint64_t c) {
    x = (x * (m | 1)) & mask;
    x ^= x >> ((k >> 1) + 1);
    x = (x + c) & mask;
    return x;
}
__global__ void k_gen_unique(int32_t *__restrict__ keys, uint64_t n, uint64_t first, uint64_t domain, uint64_t seed) {
    uint32_t k = 1;
    while (((uint64_t)1 << k) < domain) k++;
    const uint64_t mask = (((uint64_t)1 << k) - 1);
    const uint64_t s0 = fmix64(seed ^ 0x1234567ULL), s1 = fmix64(s0 + 1), s2 = fmix64(s1 + 2), s3 = fmix64(s2 + 3);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = (first + i) % domain;
        do {
            x = perm_round(x, mask, k, s0, s1 >> 7);
            x = perm_round(x, mask, k, s1, s2 >> 9);
            x = perm_round(x, mask, k, s2, s3 >> 11);
            x = perm_round(x, mask, k, s3, s0 >> 13);
        } while (x >= domain);
        keys[i] = (int32_t)(uint32_t)x;
    }
}

// Zipf(theta) ranks over an alphabet of N values, mapped through the k_gen_unique bijection and shifted
// by +1 (the reference's gen_zipf draws from an alphabet 1..N permuted at random, gen.cu:236-258,
// 299-348).  The reference builds a 2^27-entry cumulative table on the host and binary-searches it per
// tuple; here the cumulative mass H(k) = sum_{i<=k} i^-theta is exact for k <= 64 (small table in
// registers/LDS) and the Euler-Maclaurin closed form beyond, inverted by bisection — a synthetic skew
// generator with the same head probabilities (rank 1 holds 1/H(N) of the draws), not the reference's
// exact stream.
__device__ __forceinline__ double zipf_H(double k, double theta, const double *__restrict__ head, double c_tail) {
    // c_tail = head[63] - closed(64): makes the closed form continuous with the exact head at k = 64
    if (k <= 64.0) return head[(int)k - 1];
    double closed = (fabs(theta - 1.0) < 1e-9) ? log(k) : (pow(k, 1.0 - theta) - 1.0) / (1.0 - theta);
    return closed + 0.5 * pow(k, -theta) + c_tail;
}
__global__ __launch_bounds__(256) void k_gen_zipf(int32_t *__restrict__ keys, uint64_t n, uint64_t first, uint64_t alphabet,
                                                  double theta, uint64_t seed) {
    __shared__ double head[64];
    __shared__ double sh_c, sh_HN;
    if (threadIdx.x == 0) {
        double acc = 0;
        for (int i = 1; i <= 64; i++) { acc += pow((double)i, -theta); head[i - 1] = acc; }
        double closed64 = (fabs(theta - 1.0) < 1e-9) ? log(64.0) : (pow(64.0, 1.0 - theta) - 1.0) / (1.0 - theta);
        sh_c = head[63] - (closed64 + 0.5 * pow(64.0, -theta));
    }
    __syncthreads();
    if (threadIdx.x == 0) sh_HN = zipf_H((double)alphabet, theta, head, sh_c);
    __syncthreads();
    const double c_tail = sh_c, HN = sh_HN;
    uint32_t kb = 1;
    while (((uint64_t)1 << kb) < alphabet) kb++;
    const uint64_t mask = (((uint64_t)1 << kb) - 1);
    const uint64_t s0 = fmix64(seed ^ 0xABCDEF01ULL), s1 = fmix64(s0 + 1), s2 = fmix64(s1 + 2), s3 = fmix64(s2 + 3);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = fmix64((first + i) * 0x9E3779B97F4A7C15ULL ^ seed);
        const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0) * HN; // target cumulative mass
        // smallest rank k with H(k) >= u
        uint64_t lo = 1, hi = alphabet;
        while (lo < hi) {
            uint64_t mid = (lo + hi) >> 1;
            if (zipf_H((double)mid, theta, head, c_tail) >= u) hi = mid; else lo = mid + 1;
        }
        uint64_t x = lo - 1; // rank-1 in [0, alphabet) -> pseudo-random value of the alphabet
        do {
            x = perm_round(x, mask, kb, s0, s1 >> 7);
            x = perm_round(x, mask, kb, s1, s2 >> 9);
            x = perm_round(x, mask, kb, s2, s3 >> 11);
            x = perm_round(x, mask, kb, s3, s0 >> 13);
        } while (x >= alphabet);
        keys[i] = (int32_t)(uint32_t)(x + 1);
    }
}

__global__ __launch_bounds__(256) void k_digest(const int32_t *__restrict__ a, const int32_t *__restrict__ b,
                                                const int32_t *__restrict__ c, uint64_t n,
                                                unsigned long long *__restrict__ out) {
    uint64_t s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        s += c ? mix_triple(a[i], b[i], c[i]) : mix_pair(a[i], b[i]);
    s = wave_sum64(s);
    if (lane_id() == 0) atomicAdd(out, (unsigned long long)s);
}

// one workgroup per partition: tuples whose radix bits differ from the partition id, the partition's (key,pay)
// digest and its size
__global__ __launch_bounds__(256) void k_verify_partitions(const int32_t *__restrict__ keys, const int32_t *__restrict__ pays,
                                                           const uint64_t *__restrict__ beg, const uint64_t *__restrict__ end,
                                                           uint32_t nparts, unsigned long long *__restrict__ misplaced,
                                                           uint64_t *__restrict__ digests, uint64_t *__restrict__ sizes) {
    __shared__ uint64_t red[4];
    for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) {
        uint64_t bad = 0, dg = 0;
        for (uint64_t i = beg[p] + threadIdx.x; i < end[p]; i += blockDim.x) {
            if ((((uint32_t)keys[i]) & (nparts - 1)) != p) bad++;
            dg += mix_pair(keys[i], pays[i]);
        }
        bad = wave_sum64(bad);
        dg = wave_sum64(dg);
        if (lane_id() == 0 && bad) atomicAdd(misplaced, (unsigned long long)bad);
        if (sizes && threadIdx.x == 0) sizes[p] = end[p] - beg[p];
        if (digests) {
            if (lane_id() == 0) red[threadIdx.x >> 6] = dg;
            __syncthreads();
            if (threadIdx.x == 0) digests[p] = red[0] + red[1] + red[2] + red[3];
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// non-partitioned baselines (comparison curves; jp.cu:628-668 perfect array, jp.cu:681-742 global chains)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_np_max(const int32_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ out_max) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t k = (uint32_t)keys[i];
        m = k > m ? k : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if (lane_id() == 0) atomicMax(out_max, m);
}

// build_perfect_array (jp.cu:628-640): lookup[key] = row + 1 (the reference stores payload + 1; the row
// index keeps payload 0xFFFFFFFF representable)
__global__ __launch_bounds__(256) void k_np_build_perfect(const int32_t *__restrict__ keys, uint64_t n, int32_t *__restrict__ lookup) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        lookup[(uint32_t)keys[i]] = (int32_t)(uint32_t)(i + 1);
}

// probe_perfect_array (jp.cu:649-668): one dependent gather per probe tuple; out2 = {matches, agg}
__global__ __launch_bounds__(256) void k_np_probe_perfect(const int32_t *__restrict__ pk, const int32_t *__restrict__ pp, uint64_t n,
                                                          const int32_t *__restrict__ lookup, uint64_t range,
                                                          const int32_t *__restrict__ bp, unsigned long long *__restrict__ out2) {
    uint64_t m = 0, g = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t key = (uint32_t)pk[i];
        if (key < range) {
            const uint32_t res = (uint32_t)lookup[key];
            if (res) { m++; g += (uint64_t)((int64_t)pp[i] * (int64_t)bp[res - 1]); }
        }
    }
    m = wave_sum64(m); g = wave_sum64(g);
    if (lane_id() == 0) { atomicAdd(&out2[0], (unsigned long long)m); atomicAdd(&out2[1], (unsigned long long)g); }
}

// build_ht_chains (jp.cu:681-698): one global chained table, slot = key & mask, LIFO insert by atomicExch
__global__ __launch_bounds__(256) void k_np_build_chains(const int32_t *__restrict__ keys, uint64_t n, uint32_t mask,
                                                         int32_t *__restrict__ head, int32_t *__restrict__ next) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        int last = atomicExch(&head[(uint32_t)keys[i] & mask], (int32_t)(uint32_t)(i + 1));
        next[i] = last;
    }
}

// chains_probing (jp.cu:713-742)
__global__ __launch_bounds__(256) void k_np_probe_chains(const int32_t *__restrict__ pk, const int32_t *__restrict__ pp, uint64_t n,
                                                         uint32_t mask, const int32_t *__restrict__ head, const int32_t *__restrict__ next,
                                                         const int32_t *__restrict__ bk, const int32_t *__restrict__ bp,
                                                         unsigned long long *__restrict__ out2) {
    uint64_t m = 0, g = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int32_t key = pk[i], pay = pp[i];
        uint32_t nx = (uint32_t)head[(uint32_t)key & mask];
        while (nx != 0) {
            if (bk[nx - 1] == key) { m++; g += (uint64_t)((int64_t)pay * (int64_t)bp[nx - 1]); }
            nx = (uint32_t)next[nx - 1];
        }
    }
    m = wave_sum64(m); g = wave_sum64(g);
    if (lane_id() == 0) { atomicAdd(&out2[0], (unsigned long long)m); atomicAdd(&out2[1], (unsigned long long)g); }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (host)
// ------------------------------------------------------------------------------------------------

static inline uint32_t np_grid(uint64_t n) { uint64_t b = (n + 255) / 256; return (uint32_t)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

hipError_t launch_np_max(hipStream_t st, const int32_t *keys, uint64_t n, uint32_t *out_max) {
    hipLaunchKernelGGL(k_np_max, dim3(np_grid(n)), dim3(256), 0, st, keys, n, out_max);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_np_perfect(hipStream_t st, const int32_t *bk, uint64_t nb, const int32_t *bp, const int32_t *pk, const int32_t *pp,
                             uint64_t np, int32_t *lookup, uint64_t range, uint64_t *out2) {
    hipLaunchKernelGGL(k_np_build_perfect, dim3(np_grid(nb)), dim3(256), 0, st, bk, nb, lookup);
    HJ_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_np_probe_perfect, dim3(np_grid(np)), dim3(256), 0, st, pk, pp, np, lookup, range, bp,
                       reinterpret_cast<unsigned long long *>(out2));
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_np_chained(hipStream_t st, const int32_t *bk, const int32_t *bp, uint64_t nb, const int32_t *pk, const int32_t *pp,
                             uint64_t np, uint32_t log_slots, int32_t *head, int32_t *next, uint64_t *out2) {
    const uint32_t mask = (log_slots >= 32) ? 0xFFFFFFFFu : ((1u << log_slots) - 1);
    hipLaunchKernelGGL(k_np_build_chains, dim3(np_grid(nb)), dim3(256), 0, st, bk, nb, mask, head, next);
    HJ_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_np_probe_chains, dim3(np_grid(np)), dim3(256), 0, st, pk, pp, np, mask, head, next, bk, bp,
                       reinterpret_cast<unsigned long long *>(out2));
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_fill(hipStream_t st, int32_t *p, uint64_t n, int mode, uint64_t first) {
    if (!n) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_fill, dim3((uint32_t)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, p, n, mode, first);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_gen_unique(hipStream_t st, int32_t *keys, uint64_t n, uint64_t first, uint64_t domain, uint64_t seed) {
    if (!n) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_gen_unique, dim3((uint32_t)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, keys, n, first, domain, seed);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_gen_zipf(hipStream_t st, int32_t *keys, uint64_t n, uint64_t first, uint64_t alphabet, double theta, uint64_t seed) {
    if (!n) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_gen_zipf, dim3((uint32_t)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, keys, n, first, alphabet, theta, seed);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_digest(hipStream_t st, const int32_t *a, const int32_t *b, const int32_t *c, uint64_t n, uint64_t *out) {
    if (!n) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_digest, dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, a, b, c, n,
                       reinterpret_cast<unsigned long long *>(out));
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_verify_partitions(hipStream_t st, const int32_t *keys, const int32_t *pays, const uint64_t *beg,
                                    const uint64_t *end, uint32_t nparts, uint32_t, uint32_t, uint64_t *misplaced,
                                    uint64_t *digests, uint64_t *sizes) {
    hipLaunchKernelGGL(k_verify_partitions, dim3(nparts < 4096 ? nparts : 4096), dim3(256), 0, st, keys, pays, beg, end, nparts,
                       reinterpret_cast<unsigned long long *>(misplaced), digests, sizes);
    HJ_LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace hj
