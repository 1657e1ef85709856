// experiment gate: what would pass 2 gain from writing 2-byte tags (one 128-byte line per 64 tuples) + 4-byte payloads (one 128-byte
// line per 32 tuples) instead of two 4-byte columns?  Same streaming reads (two int32 columns, 16 B per lane); every 128-byte output
// line goes to a pseudo-random aligned line position (the write pattern of the write-combining flush).  KIND 0: two int32 columns out
// (16 B moved per tuple); KIND 1: payload lines + tag lines (14 B per tuple).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(const int4 *__restrict__ ik, const int4 *__restrict__ ip, int4 *__restrict__ ok, int4 *__restrict__ op,
                                         uint64_t n16, uint64_t line_mask, uint64_t mul) {
    // a workgroup owns chunks of 512 units (2048 tuples); lane u of a chunk holds tuples 4u..4u+3
    for (uint64_t base = (uint64_t)blockIdx.x * 512; base < n16; base += (uint64_t)gridDim.x * 512) {
        int4 a[2], b[2];
#pragma unroll
        for (int j = 0; j < 2; j++) { const uint64_t u = base + (uint64_t)j * 256 + threadIdx.x; a[j] = ik[u]; b[j] = ip[u]; }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint64_t u = base + (uint64_t)j * 256 + threadIdx.x;
            const uint64_t line = u >> 3;                                   // 8 units = 32 tuples = one payload line
            const uint64_t o = (((line * mul) & line_mask) << 3) | (u & 7);
            op[o] = b[j];                                                   // payload line: 128 B scattered
            if (KIND == 0) ok[o] = a[j];                                    // key line: 128 B scattered
            else {
                // tags: 4 tuples -> 8 bytes; 16 lanes x 8 B = one 128-byte tag line per 64 tuples, scattered at tag-line granularity
                const uint64_t tline = u >> 4;
                const uint64_t to = (((tline * mul) & (line_mask >> 1)) << 4) | (u & 15);
                uint2 tg = make_uint2(((uint32_t)a[j].x & 0xFFFFu) | ((uint32_t)a[j].y << 16), ((uint32_t)a[j].z & 0xFFFFu) | ((uint32_t)a[j].w << 16));
                reinterpret_cast<uint2 *>(ok)[to] = tg;
            }
        }
    }
}

int main() {
    const uint64_t n = (uint64_t)1 << 30, n16 = n / 4, lines = n16 / 8;
    int4 *ik, *ip, *ok, *op;
    CHK(hipMalloc(&ik, n * 4)); CHK(hipMalloc(&ip, n * 4)); CHK(hipMalloc(&ok, n * 4)); CHK(hipMalloc(&op, n * 4));
    CHK(hipMemset(ik, 1, n * 4)); CHK(hipMemset(ip, 2, n * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const uint64_t mul = 0x9E3779B97F4A7C15ULL | 1;
    for (int round = 0; round < 3; round++)
        for (int kind = 0; kind < 2; kind++) {
            auto launch = [&]() {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(16384), dim3(256), 0, 0, ik, ip, ok, op, n16, lines - 1, mul);
                else hipLaunchKernelGGL(k<1>, dim3(16384), dim3(256), 0, 0, ik, ip, ok, op, n16, lines - 1, mul);
            };
            launch();
            CHK(hipEventRecord(e0));
            for (int rep = 0; rep < 5; rep++) launch();
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            const double bytes = kind == 0 ? 16.0 * n : 14.0 * n;
            printf("%s: %.3f ms per 2^30 tuples, %.0f GB/s\n", kind == 0 ? "two int32 columns out (16 B/tuple)   " : "payload lines + tag lines (14 B/tuple)", ms, bytes / (ms * 1e-3) / 1e9);
        }
    return 0;
}
