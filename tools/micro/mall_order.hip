// Does the order in which a consumer kernel sweeps its input matter when the producer kernel has just written it?
// Producer P: y[r] = x[r] + 1 for rows r ascending in time (grid-stride over rows, fp32 [M][768]: reads 158 MB,
// writes 158 MB at M = 51456).  Consumer C: LayerNorm-shaped stream (reads y fp32, writes h bf16-sized rows), sweeping
// rows ascending (same direction as P: the oldest lines first) or descending (the newest lines first).  The 256 MiB
// Infinity Cache holds the tail of P's traffic; if it serves reads, the descending consumer is faster.
// usage: mall_order [M]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define D 768

__global__ __launch_bounds__(256) void producer(const float* __restrict__ x, float* __restrict__ y, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            float4 a = *reinterpret_cast<const float4*>(x + (size_t)row * D + (v * 64 + lane) * 4);
            a.x += 1.f; a.y += 1.f; a.z += 1.f; a.w += 1.f;
            *reinterpret_cast<float4*>(y + (size_t)row * D + (v * 64 + lane) * 4) = a;
        }
    }
}

__global__ __launch_bounds__(256) void consumer(const float* __restrict__ y, unsigned short* __restrict__ h, int M, int rev) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r0 = blockIdx.x * 4 + wave; r0 < M; r0 += gridDim.x * 4) {
        const int row = rev ? M - 1 - r0 : r0;
        float4 a[3];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            a[v] = *reinterpret_cast<const float4*>(y + (size_t)row * D + (v * 64 + lane) * 4);
            s += a[v].x + a[v].y + a[v].z + a[v].w;
        }
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        s *= 1.f / D;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            uint2 p;
            p.x = (__float_as_uint(a[v].x - s) >> 16) | (__float_as_uint(a[v].y - s) & 0xffff0000u);
            p.y = (__float_as_uint(a[v].z - s) >> 16) | (__float_as_uint(a[v].w - s) & 0xffff0000u);
            *reinterpret_cast<uint2*>(h + (size_t)row * D + (v * 64 + lane) * 4) = p;
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 51456;
    float *x, *y, *junk; unsigned short* h;
    const size_t n = (size_t)M * D;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&h, n * 2));
    const size_t junk_n = (size_t)160 << 20;           // 640 MB of unrelated traffic to flush the caches
    CK(hipMalloc(&junk, junk_n * 4));
    CK(hipMemset(x, 0, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 1024;
    for (int mode = 0; mode < 3; ++mode) {             // 0: cold (flushed), 1: after P ascending, 2: after P, descending
        float best = 1e9f, sum = 0.f;
        const int reps = 10;
        for (int it = 0; it < reps + 2; ++it) {
            if (mode == 0) CK(hipMemsetAsync(junk, it, junk_n * 4, 0));
            else hipLaunchKernelGGL(producer, dim3(grid), dim3(256), 0, 0, x, y, M);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(consumer, dim3(grid), dim3(256), 0, 0, y, h, M, mode == 2);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        const double bytes = (double)n * 6;
        printf("M=%d consumer %-28s avg %.1f us  best %.1f us  (%.2f TB/s algorithmic)\n", M,
               mode == 0 ? "cold (caches flushed)" : mode == 1 ? "after producer, ascending" : "after producer, descending",
               sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) * 1e-12);
    }
    return 0;
}
