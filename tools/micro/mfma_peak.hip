// Sustained v_mfma_f32_16x16x32_bf16 rate without any memory traffic: NW waves per workgroup (one workgroup per CU),
// NACC independent accumulators per wave.  usage: mfma_peak   (prints TFLOP/s for a few configurations)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(1024) void mfma_loop(float* out, int iters) {
    f32x4 acc[NACC];
    // RANDOM operand bits (4 rotating fragment pairs): constant operands let the chip hold a higher clock
    bf16x8 av[4], bv[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; av[q][i] = (__bf16)(((int)(h >> 16) & 1023) * (1.0f / 512.0f) - 1.0f);
            h = h * 1664525u + 1013904223u; bv[q][i] = (__bf16)(((int)(h >> 16) & 1023) * (1.0f / 512.0f) - 1.0f);
        }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC>
void run(int waves, int blocks_per_cu) {
    float* out;
    const int ncu = 256, iters = 40000;
    hipMalloc(&out, (size_t)ncu * blocks_per_cu * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(ncu * blocks_per_cu), dim3(waves * 64), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(ncu * blocks_per_cu), dim3(waves * 64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)ncu * blocks_per_cu * waves * iters * NACC * 16384.0;
    printf("waves/CU %2d  acc %2d : %7.3f ms  %8.1f TFLOP/s\n", waves * blocks_per_cu, NACC, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    run<8>(4, 1); run<16>(4, 1);
    run<8>(8, 1); run<16>(8, 1); run<16>(8, 1); run<16>(8, 1);
    run<8>(16, 1); run<16>(16, 1);
    return 0;
}
