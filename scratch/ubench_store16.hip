// microbenchmark 2: scatter of whole 128-byte lines with 16-byte-per-lane accesses.
// Each group of 8 lanes moves one 128-B line (32 tuples) to a pseudo-random 128-B-aligned line of the
// output; two columns; reads are 16 B/lane coalesced.  Compare with scratch/ubench_store.hip (4 B/lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint64_t perm(uint64_t x, uint32_t k) {
    uint64_t mask = (1ull << k) - 1;
    x = (x * 0x9E3779B97F4A7C15ull) & mask; x ^= x >> (k/2+1); x = (x * 0xD6E8FEB86659FD93ull | 1) & mask; x ^= x >> (k/2+1);
    x = (x * 0xC2B2AE3D27D4EB4Full) & mask;
    return x;
}
// LINES_PER_SEG: how many consecutive lines form one contiguous destination segment (1 = 128 B, 2 = 256 B ...)
template<int LPS, int READ, int NT>
__global__ __launch_bounds__(512) void k(int4* __restrict__ out, int4* __restrict__ out2, const int4* __restrict__ in, const int4* __restrict__ in2, uint64_t nvec, uint32_t kbits) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        uint64_t seg = i / (8 * LPS), off = i % (8 * LPS);
        uint64_t dst = perm(seg, kbits) * (8 * LPS) + off;
        int4 a = READ ? in[i] : make_int4((int)i, 1, 2, 3);
        int4 b = READ ? in2[i] : make_int4((int)i, 4, 5, 6);
        typedef int v4i __attribute__((ext_vector_type(4)));
        if (NT) { v4i va = {a.x, a.y, a.z, a.w}, vb = {b.x, b.y, b.z, b.w};
                  __builtin_nontemporal_store(va, reinterpret_cast<v4i*>(&out[dst])); __builtin_nontemporal_store(vb, reinterpret_cast<v4i*>(&out2[dst])); }
        else { out[dst] = a; out2[dst] = b; }
    }
}
template<int LPS, int READ, int NT> float run(int4* out, int4* out2, const int4* in, const int4* in2, uint64_t nvec) {
    uint32_t kb = 0; while ((1ull << kb) < nvec / (8 * LPS)) kb++;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<LPS,READ,NT>), dim3(4096), dim3(512), 0, 0, out, out2, in, in2, nvec, kb);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}
__global__ void copyk(int4* __restrict__ out, const int4* __restrict__ in, uint64_t nvec) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) out[i] = in[i];
}
int main() {
    uint64_t n = 1ull << 30, nvec = n / 4; // 4 GiB per column
    int4 *out, *out2, *in, *in2;
    CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&out2, n * 4)); CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&in2, n * 4));
    CK(hipMemset(in, 1, n * 4)); CK(hipMemset(in2, 2, n * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 4; r++) { CK(hipEventRecord(a)); hipLaunchKernelGGL(copyk, dim3(4096), dim3(512), 0, 0, out, in, nvec); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    printf("plain int4 copy 4 GiB: %.3f ms  %.1f GB/s (read+write)\n", best, 2.0 * n * 4 / best / 1e6);
    printf("lines/seg read nt   ms   total_GB/s\n");
    float t;
    #define R(L, RD, NT) t = run<L, RD, NT>(out, out2, in, in2, nvec); printf("%d %d %d %7.3f %8.1f\n", L, RD, NT, t, (2.0 + 2.0 * RD) * n * 4 / t / 1e6);
    R(1,0,0) R(2,0,0) R(8,0,0) R(1,1,0) R(2,1,0) R(8,1,0) R(1,1,1) R(8,1,1)
    return 0;
}
