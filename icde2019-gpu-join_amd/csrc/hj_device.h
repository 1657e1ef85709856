// hj_device.h — device helpers shared by the kernel files (hj_part.hip, hj_join.hip, hj_util.hip): wave64 reductions and scans,
// the workgroup exclusive scan, 16-byte loads with explicit tails, the digest mixers.  Header-only: no device code is linked across
// translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>

#include "hj_internal.h"

namespace hj {

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (device, function) pair: high-water marks are kept under this lock
// (one per kernel file: each guards the flags of its own kernels)
static std::mutex g_attr_mutex;

#define HJ_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint64_t fmix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint64_t mix_pair(int32_t key, int32_t pay) {
    return fmix64(((uint64_t)(uint32_t)key << 32) | (uint32_t)pay);
}
__device__ __forceinline__ uint64_t mix_triple(int32_t key, int32_t pr, int32_t ps) {
    return fmix64(mix_pair(key, pr) ^ ((uint64_t)(uint32_t)ps * 0x9E3779B97F4A7C15ULL));
}

#ifdef HJ_STAMPS
// experiment builds (`make stamps`, tools/experiments/fixed_cost.py): where and when a workgroup ran.  s_memrealtime counts at 100 MHz;
// HW_ID (register 4): wave, SIMD, CU, SH, SE; XCC_ID (register 20): the XCD.
__device__ __forceinline__ unsigned long long hj_now() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ unsigned long long hj_where() {
    const uint32_t hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4), xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20);
    return ((unsigned long long)xcc << 32) | hw;
}
#endif

// Partition function.  MODE 0: the reference's (hasht(key) >> first_bit) & (parts-1) with hasht =
// identity (common.h:45-47, jp.cu:126).  MODE 1: shard id for the multi-GPU level-0 split, a
// multiplicative range reduction of a murmur-finalised key (independent of the low radix bits).
// remap (MODE 1, optional): output position of each shard — the multi-GPU driver orders virtual shards by owner GPU.
template <int MODE>
__device__ __forceinline__ uint32_t digit_of(uint32_t key, uint32_t shift, uint32_t mask_or_n, const uint32_t *__restrict__ remap = nullptr) {
    if (MODE == 0) return (key >> shift) & mask_or_n;
    const uint32_t d = (uint32_t)(((uint64_t)fmix32(key) * mask_or_n) >> 32);
    return remap ? remap[d] : d;
}

// a wave-uniform 64-bit value, moved to SGPRs for good (the compiler cannot always prove uniformity across loop nests and then
// keeps such values — range bounds, stream positions — in vector registers)
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if ((int)lane_id() >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t t = __shfl_up(v, o, 64);
        if ((int)lane_id() >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Exclusive scan of one value per thread over the workgroup (blockDim.x multiple of 64, <= 1024).
// scratch: >= 17 T's of LDS.  Returns the exclusive prefix; *total = sum over the workgroup.
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T *scratch, T *total) {
    const uint32_t w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    T incl = (sizeof(T) == 8) ? (T)wave_incl_scan64((uint64_t)v) : (T)wave_incl_scan((uint32_t)v);
    if (lane_id() == 63) scratch[w] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
        for (uint32_t i = 0; i < nw; i++) { T t = scratch[i]; scratch[i] = run; run += t; }
        scratch[16] = run;
    }
    __syncthreads();
    T res = incl - v + scratch[w];
    if (total) *total = scratch[16];
    __syncthreads();
    return res;
}

// 16-byte load of 4 consecutive int32 at element index i (i % 4 == 0, base 16-B aligned); elements
// at or beyond nalloc (the true length of the array) are not touched.
__device__ __forceinline__ int4 load4(const int32_t *__restrict__ base, uint64_t i, uint64_t nalloc) {
    if (i + 4 <= nalloc) return *reinterpret_cast<const int4 *>(base + i);
    int4 v = make_int4(0, 0, 0, 0);
    if (i < nalloc) v.x = base[i];
    if (i + 1 < nalloc) v.y = base[i + 1];
    if (i + 2 < nalloc) v.z = base[i + 2];
    return v;
}
__device__ __forceinline__ int32_t elem(const int4 &v, int e) {
    return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}

// Returning "count me in" on an LDS counter, skew-aware: when at least 16 lanes of the wave carry the
// same digit as the wave's first valid lane (a heavy hitter), those lanes are served by ONE atomic add
// of their number and ranked by a ballot; everybody else adds 1 for itself.  Must be called by all 64
// lanes of the wave (valid = this lane has a tuple).  Same-address LDS atomics serialise per lane.
// few = the pass has at most 2 digits (a 2-GPU shard split; measured slower from 3 digits on): every digit present in the wave is
// served by one aggregated atomic, in turn — with 2 digits nearly every lane would otherwise queue
// on one of 2 LDS words.
__device__ __forceinline__ uint32_t rank_in_digit(uint32_t *cnt, uint32_t d, bool valid, bool few = false) {
    const uint64_t vmask = __ballot(valid);
    uint32_t r = 0;
    if (few) {
        uint64_t rem = vmask;
        while (rem) {
            const int first = __builtin_ctzll(rem);
            const uint32_t lead = (uint32_t)__shfl((int)d, first, 64);
            const bool same = valid && d == lead;
            const uint64_t smask = __ballot(same);
            uint32_t base = 0;
            if ((int)lane_id() == first) base = atomicAdd(&cnt[lead], (uint32_t)__popcll(smask));
            base = (uint32_t)__shfl((int)base, first, 64);
            if (same) r = base + (uint32_t)__popcll(smask & (((uint64_t)1 << lane_id()) - 1));
            rem &= ~smask;
        }
        return r;
    }
    if (vmask) {
        const int first = __builtin_ctzll(vmask);
        const uint32_t lead = (uint32_t)__shfl((int)d, first, 64);
        const bool same = valid && d == lead;
        const uint64_t smask = __ballot(same);
        if (__popcll(smask) >= 16) {
            uint32_t base = 0;
            if ((int)lane_id() == first) base = atomicAdd(&cnt[lead], (uint32_t)__popcll(smask));
            base = (uint32_t)__shfl((int)base, first, 64);
            if (same) r = base + (uint32_t)__popcll(smask & (((uint64_t)1 << lane_id()) - 1));
            else if (valid) r = atomicAdd(&cnt[d], 1u);
        } else if (valid) {
            r = atomicAdd(&cnt[d], 1u);
        }
    }
    return r;
}

} // namespace hj
