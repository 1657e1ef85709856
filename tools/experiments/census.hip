// tools/experiments/census.hip — how many workgroups of a given shape does one CU of this GPU hold at a time?  Every workgroup notes
// when it started and ended (s_memrealtime) and where (HW_ID, XCC_ID); the host counts the peak number resident at one time per CU.
// hipcc --offload-arch=gfx950 -O2 census.hip -o census && ./census
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <map>
#include <vector>

template <int VREGS>
__global__ void k_census(unsigned long long *st, int spin_ticks, float *sink) {
    extern __shared__ unsigned char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc[VREGS];
#pragma unroll
    for (int i = 0; i < VREGS; i++) acc[i] = (float)(threadIdx.x + i);
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
        for (int i = 0; i < VREGS; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < VREGS; i++) s += acc[i];
    if (s == 12345.678f) sink[0] = s + smem[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4), xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20);
        st[blockIdx.x * 3 + 0] = t0; st[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime(); st[blockIdx.x * 3 + 2] = ((unsigned long long)xcc << 32) | hw;
    }
}

// the same with NS wave-uniform values kept alive across the spin loop (scalar registers)
template <int NS>
__global__ void k_census_s(unsigned long long *st, int spin_ticks, float *sink, const int *__restrict__ uni) {
    extern __shared__ unsigned char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int u[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) u[i] = __builtin_amdgcn_readfirstlane(uni[i]);
    int acc = threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
        for (int i = 0; i < NS; i++) { acc = acc * 3 + u[i]; u[i] = __builtin_amdgcn_readfirstlane(u[i] ^ (u[(i + 1) % NS] >> 1)); }
    }
    if (acc == 123456789) sink[0] = (float)acc + smem[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4), xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20);
        st[blockIdx.x * 3 + 0] = t0; st[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime(); st[blockIdx.x * 3 + 2] = ((unsigned long long)xcc << 32) | hw;
    }
}

template <int NS>
static void run_s(int threads, size_t lds, int nwg) {
    unsigned long long *d;
    float *sink;
    int *uni;
    hipMalloc(&d, (size_t)nwg * 24); hipMalloc(&sink, 64); hipMalloc(&uni, 1024);
    hipMemset(d, 0, (size_t)nwg * 24); hipMemset(uni, 1, 1024);
    auto fn = k_census_s<NS>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int api = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, fn, threads, lds);
    hipLaunchKernelGGL(fn, dim3(nwg), dim3(threads), lds, 0, d, 800, sink, uni);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nwg * 3);
    hipMemcpy(h.data(), d, (size_t)nwg * 24, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    for (int i = 0; i < nwg; i++) {
        const unsigned long long cu = (h[i * 3 + 2] >> 32 << 16) | ((h[i * 3 + 2] & 0xFFFF) >> 8);
        ev[cu].push_back({h[i * 3 + 0], +1});
        ev[cu].push_back({h[i * 3 + 1], -1});
    }
    int peak = 0, total_peak = 0;
    for (auto &kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0, pk = 0;
        for (auto &x : kv.second) { cur += x.second; pk = std::max(pk, cur); }
        peak = std::max(peak, pk); total_peak += pk;
    }
    printf("SCALAR-heavy (%d uniform values)  threads %4d  lds %6zu B  : API blocks/CU %d  resident per CU (max) %d  (mean %.2f)  %s\n", NS, threads, lds, api, peak,
           (double)total_peak / ev.size(), e == hipSuccess ? "" : hipGetErrorString(e));
    hipFree(d); hipFree(sink); hipFree(uni);
}

template <int VREGS>
static void run(int threads, size_t lds, int nwg) {
    unsigned long long *d;
    float *sink;
    hipMalloc(&d, (size_t)nwg * 24);
    hipMalloc(&sink, 64);
    hipMemset(d, 0, (size_t)nwg * 24);
    auto fn = k_census<VREGS>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int api = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, fn, threads, lds);
    hipLaunchKernelGGL(fn, dim3(nwg), dim3(threads), lds, 0, d, 800 /* 8 us */, sink);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nwg * 3);
    hipMemcpy(h.data(), d, (size_t)nwg * 24, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev; // per CU: (time, +1 / -1)
    for (int i = 0; i < nwg; i++) {
        const unsigned long long cu = (h[i * 3 + 2] >> 32 << 16) | ((h[i * 3 + 2] & 0xFFFF) >> 8);
        ev[cu].push_back({h[i * 3 + 0], +1});
        ev[cu].push_back({h[i * 3 + 1], -1});
    }
    int peak = 0, total_peak = 0;
    for (auto &kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0, pk = 0;
        for (auto &x : kv.second) { cur += x.second; pk = std::max(pk, cur); }
        peak = std::max(peak, pk); total_peak += pk;
    }
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(fn));
    printf("threads %4d  lds %6zu B  vgprs %3d  sgprs %3d  : CUs %3zu  API blocks/CU %d  resident per CU (max) %d  (mean %.2f)  %s\n", threads, lds, fa.numRegs, 0, ev.size(), api, peak,
           (double)total_peak / ev.size(), e == hipSuccess ? "" : hipGetErrorString(e));
    hipFree(d); hipFree(sink);
}

int main() {
    const int nwg = 256 * 24;
    for (int threads : {256, 512, 1024}) {
        for (size_t lds : {(size_t)0, (size_t)20 * 1024, (size_t)36 * 1024, (size_t)52 * 1024, (size_t)64 * 1024}) {
            run<8>(threads, lds, nwg);
        }
    }
    run<40>(512, 0, nwg);
    run<40>(512, 52 * 1024, nwg);
    run<64>(512, 0, nwg);
    run<64>(512, 52 * 1024, nwg);
    for (size_t lds : {(size_t)0, (size_t)52 * 1024}) { run_s<40>(512, lds, nwg); run_s<60>(512, lds, nwg); run_s<80>(512, lds, nwg); run_s<96>(512, lds, nwg); run_s<80>(256, lds, nwg); run_s<96>(256, lds, nwg); }
    return 0;
}
