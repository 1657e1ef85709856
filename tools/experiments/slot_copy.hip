// experiment: does chip-wide time-slotting of reads and writes (all CUs read for T, then all write for T) beat an unsynchronised copy?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one workgroup of 1024 threads per CU; per batch: B int4 per thread per column (B*16 KiB*2 per workgroup) held in registers
template <int B>
__global__ __launch_bounds__(1024) void k_copy_slotted(const int4 *__restrict__ ik, const int4 *__restrict__ ip, int4 *__restrict__ ok, int4 *__restrict__ op,
                                                       uint64_t n16, uint32_t slot_ticks) {
    const uint64_t per_batch = (uint64_t)1024 * B;
    for (uint64_t base = (uint64_t)blockIdx.x * per_batch; base < n16; base += (uint64_t)gridDim.x * per_batch) {
        int4 a[B], b[B];
        if (slot_ticks) { while (((wall_clock64() / slot_ticks) & 1ull) != 0ull) __builtin_amdgcn_s_sleep(2); }   // read window: even slots
#pragma unroll
        for (int j = 0; j < B; j++) {
            const uint64_t u = base + (uint64_t)j * 1024 + threadIdx.x;
            if (u < n16) { a[j] = ik[u]; b[j] = ip[u]; }
        }
        if (slot_ticks) { while (((wall_clock64() / slot_ticks) & 1ull) != 1ull) __builtin_amdgcn_s_sleep(2); }   // write window: odd slots
#pragma unroll
        for (int j = 0; j < B; j++) {
            const uint64_t u = base + (uint64_t)j * 1024 + threadIdx.x;
            if (u < n16) { ok[u] = a[j]; op[u] = b[j]; }
        }
    }
}

int main(int argc, char **argv) {
    const uint64_t n = (uint64_t)1 << 30, n16 = n / 4;
    int4 *ik, *ip, *ok, *op;
    CHK(hipMalloc(&ik, n * 4)); CHK(hipMalloc(&ip, n * 4)); CHK(hipMalloc(&ok, n * 4)); CHK(hipMalloc(&op, n * 4));
    CHK(hipMemset(ik, 1, n * 4)); CHK(hipMemset(ip, 2, n * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grids[] = {256, 512, 1024};
    const uint32_t slots[] = {0, 100, 200, 400, 800, 1600, 3200};
    for (int g : grids)
        for (uint32_t s : slots) {
            for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_copy_slotted<8>), dim3(g), dim3(1024), 0, 0, ik, ip, ok, op, n16, s);
            CHK(hipEventRecord(e0));
            for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL((k_copy_slotted<8>), dim3(g), dim3(1024), 0, 0, ik, ip, ok, op, n16, s);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            printf("B=8 grid %4d slot %5u ticks (%.1f us): %.3f ms  %.0f GB/s\n", g, s, s / 100.0, ms, 16.0 * n / (ms * 1e-3) / 1e9);
        }
    return 0;
}
