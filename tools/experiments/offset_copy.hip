// experiment: does the RELATIVE placement of the four streams of a two-column copy (keys in, payloads in, keys out, payloads out)
// decide what HBM gives it?  On one box the same copy measured 5.1 ... 6.0 TB/s in eight processes (tools/gpu_spread.sh): the buffers
// land differently.  Here the four columns are carved from ONE allocation and one of them at a time is shifted by d bytes.
// build + run on the box: hipcc --offload-arch=gfx950 -O3 -o /tmp/offset_copy tools/experiments/offset_copy.hip && /tmp/offset_copy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// the library's copy micro-benchmark shape: 16 bytes per lane per column, two loads in flight per column
__global__ __launch_bounds__(1024) void k_copy2(const int4 *__restrict__ ik, const int4 *__restrict__ ip, int4 *__restrict__ ok, int4 *__restrict__ op, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * 1024 * 2;
    for (uint64_t u = (uint64_t)blockIdx.x * 2048 + threadIdx.x; u < n16; u += stride) {
        const uint64_t v = u + 1024;
        const int4 a0 = ik[u], b0 = ip[u];
        int4 a1 = a0, b1 = b0;
        if (v < n16) { a1 = ik[v]; b1 = ip[v]; }
        ok[u] = a0; op[u] = b0;
        if (v < n16) { ok[v] = a1; op[v] = b1; }
    }
}

int main(int argc, char **argv) {
    const uint64_t n = (uint64_t)1 << 28, col = n * 4, n16 = n / 4; // 1 GiB per column
    const uint64_t slack = (uint64_t)64 << 20, pitch = col + slack;  // every column has 64 MiB of room to move in
    char *base;
    CHK(hipMalloc(&base, 4 * pitch + slack));
    base = (char *)(((uintptr_t)base + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1)); // 2-MiB aligned
    CHK(hipMemset(base, 1, 4 * pitch));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto run = [&](uint64_t d_ip, uint64_t d_ok, uint64_t d_op) {
        const int4 *ik = (const int4 *)(base), *ip = (const int4 *)(base + pitch + d_ip);
        int4 *ok = (int4 *)(base + 2 * pitch + d_ok), *op = (int4 *)(base + 3 * pitch + d_op);
        std::vector<float> t;
        for (int rep = 0; rep < 7; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_copy2, dim3(512), dim3(1024), 0, 0, ik, ip, ok, op, n16);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        return 16.0 * n / (t[t.size() / 2] * 1e-3) / 1e9; // median GB/s
    };
    const uint64_t ds[] = {0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 4194304, 8388608, 16777216, 33554432};
    printf("pitch between the columns: 1 GiB + 64 MiB; shift of ONE column at a time (GB/s, median of 5)\n");
    printf("%10s %12s %12s %12s %14s\n", "d bytes", "payload in", "keys out", "payload out", "both out");
    for (uint64_t d : ds) printf("%10llu %12.0f %12.0f %12.0f %14.0f\n", (unsigned long long)d, run(d, 0, 0), run(0, d, 0), run(0, 0, d), run(0, d, d));
    // every combination of shifts of 4 KiB ... 28 KiB for the three movable columns (keys in stays): what helps, what hurts
    {
        struct Row { double gbs; int a, b, c; };
        std::vector<Row> rows;
        for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) for (int c = 0; c < 8; c++) rows.push_back({run((uint64_t)a * 4096, (uint64_t)b * 4096, (uint64_t)c * 4096), a, b, c});
        std::sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.gbs > y.gbs; });
        if (FILE *f = fopen("gpurun_out/offset_copy_512.csv", "w")) { fprintf(f, "payload_in_4k,keys_out_4k,payload_out_4k,GBs\n"); for (const Row &r : rows) fprintf(f, "%d,%d,%d,%.0f\n", r.a, r.b, r.c, r.gbs); fclose(f); }
        printf("shifts in units of 4 KiB (payload in, keys out, payload out): best and worst of 512\n");
        for (int i = 0; i < 12; i++) printf("  best  %2d: (%d,%d,%d) %.0f\n", i, rows[i].a, rows[i].b, rows[i].c, rows[i].gbs);
        for (int i = 0; i < 8; i++) { const Row &r = rows[rows.size() - 1 - i]; printf("  worst %2d: (%d,%d,%d) %.0f\n", i, r.a, r.b, r.c, r.gbs); }
        double sum = 0, all_distinct = 0, nd = 0, some_equal = 0, ne = 0;
        for (const Row &r : rows) {
            sum += r.gbs;
            const bool distinct = r.a != 0 && r.b != 0 && r.c != 0 && r.a != r.b && r.a != r.c && r.b != r.c;
            if (distinct) { all_distinct += r.gbs; nd++; } else { some_equal += r.gbs; ne++; }
        }
        printf("  mean of all %.0f; all four columns at different 4-KiB phases (mod 32 KiB): %.0f (%d cases); some two at the same phase: %.0f (%d cases)\n",
               sum / rows.size(), all_distinct / nd, (int)nd, some_equal / ne, (int)ne);
    }
    // the pitch itself: columns exactly 1 GiB apart (what four hipMallocs of 2^28 int32 tend to give) against odd pitches
    printf("all four columns at pitch p (GB/s):\n");
    for (uint64_t extra : {(uint64_t)0, (uint64_t)4096, (uint64_t)65536, (uint64_t)(1u << 20), (uint64_t)(2u << 20), (uint64_t)(3u << 20), (uint64_t)(5u << 20), (uint64_t)(17u << 20), (uint64_t)(33u << 20)}) {
        const uint64_t p = col + extra;
        const int4 *ik = (const int4 *)(base), *ip = (const int4 *)(base + p);
        int4 *ok = (int4 *)(base + 2 * p), *op = (int4 *)(base + 3 * p);
        std::vector<float> t;
        for (int rep = 0; rep < 7; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_copy2, dim3(512), dim3(1024), 0, 0, ik, ip, ok, op, n16);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("  pitch 1 GiB + %9llu: %.0f\n", (unsigned long long)extra, 16.0 * n / (t[t.size() / 2] * 1e-3) / 1e9);
    }
    return 0;
}
