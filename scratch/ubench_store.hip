// microbenchmark: scattered segment stores. Each wave-instruction writes 64 lanes x 4 B as 64/SEG
// segments of SEG tuples; segment destinations are pseudo-random SEG-aligned (+shift) positions in a
// big buffer, every position written exactly once (bijection), so total bytes = buffer size.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ __forceinline__ uint64_t perm(uint64_t x, uint32_t k) { // bijection on k bits
    uint64_t mask = (1ull << k) - 1;
    x = (x * 0x9E3779B97F4A7C15ull) & mask; x ^= x >> (k/2+1); x = (x * 0xD6E8FEB86659FD93ull | 1) & mask; x ^= x >> (k/2+1);
    x = (x * 0xC2B2AE3D27D4EB4Full) & mask; // odd multipliers only
    return x;
}
// n tuples total, SEG tuples per segment
template<int SEG, int READ>
__global__ __launch_bounds__(512) void k(int32_t* __restrict__ out, int32_t* __restrict__ out2, const int32_t* __restrict__ in, uint64_t n, uint32_t kbits, int shift) {
    uint64_t nseg = n / SEG;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t seg = i / SEG, off = i % SEG;
        uint64_t dseg = perm(seg, kbits);        // kbits = log2(nseg)
        uint64_t dst = dseg * SEG + off + shift; // shift breaks alignment (buffer has slack)
        int32_t v = READ ? in[i] : (int32_t)i;
        out[dst] = v;
        out2[dst] = v + 1;
    }
}
template<int SEG> float run(int32_t* out, int32_t* out2, const int32_t* in, uint64_t n, int shift, bool rd) {
    uint32_t kb = 0; while ((1ull << kb) < n / SEG) kb++;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(a));
        if (rd) hipLaunchKernelGGL((k<SEG,1>), dim3(2048), dim3(512), 0, 0, out, out2, in, n, kb, shift);
        else hipLaunchKernelGGL((k<SEG,0>), dim3(2048), dim3(512), 0, 0, out, out2, in, n, kb, shift);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    uint64_t n = 1ull << 30; // 4 GiB per column, two columns written
    int32_t *out, *out2, *in;
    CK(hipMalloc(&out, (n + 64) * 4)); CK(hipMalloc(&out2, (n + 64) * 4)); CK(hipMalloc(&in, n * 4));
    CK(hipMemset(in, 1, n * 4));
    printf("seg shift read  ms   write_GB/s  total_GB/s\n");
    for (int rd = 0; rd < 2; rd++)
    for (int shift : {0, 1, 8}) {
        float t;
        #define R(S) t = run<S>(out, out2, in, n, shift % S ? shift % S : (shift?S/2:0), rd); printf("%3d %3d %d %7.3f %8.1f %8.1f\n", S, shift % S ? shift % S : (shift?S/2:0), rd, t, 2.0*n*4/t/1e6, (2.0+rd)*n*4/t/1e6);
        R(8) R(16) R(32) R(64) R(256)
    }
    return 0;
}
