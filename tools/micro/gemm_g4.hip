// Experiment: 256 x 256 x 64 bf16 TN GEMM tile with FOUR waves per workgroup, ONE wave per SIMD (512 registers per
// lane): every wave owns a 128 x 128 block (64 accumulator fragments = 256 registers), reads 16 fragments per k-step of
// 32 and double-buffers them in registers, so LDS -> register reads for k-step s+1 run under the 64 MFMAs of k-step s
// (software pipelining inside one wave instead of a second wave per SIMD).  Per K-tile: 128 KiB of fragment reads
// (the 8-wave kernels read 192 KiB), ONE barrier (they have three), one vmcnt wait.
// C[M][N] (bf16) = A[M][K] . B[N][K]^T, M, N multiples of 256, K a multiple of 64, K >= 128.
// usage: gemm_g4            (checks a small problem against a naive kernel, then times the C2 shapes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <type_traits>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)((__attribute__((address_space(3))) const char*)p);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

#define SLOT 65536          // one K-tile: A rows 0..255 (32 KiB) then B rows 0..255 (32 KiB), 128-byte rows, XOR-swizzled

#define RD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)

struct G4Args { const bf16_t* A; const bf16_t* B; bf16_t* C; int M, N, K, lda, ldb, ldc, group_n, amask, bmask; };

__global__ __launch_bounds__(256) void gemm_g4_kernel(G4Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = p.M >> 8, tiles_n = p.N >> 8;
    // column-group-major tile order (groups of group_n N-tiles x all M-tiles), XCD-contiguous
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int per = tiles_m * p.group_n;
    const int cg = lid / per, rem = lid - cg * per;
    const int w = min(p.group_n, tiles_n - cg * p.group_n);
    const int tm = rem / w, tn = cg * p.group_n + (rem - tm * w);
    const int m0 = tm << 8, n0 = tn << 8;
    const int nk = p.K >> 6;

    // ---- staging: per K-tile and wave 8 A instructions + 8 B instructions of 1 KiB (8 rows x 128 B) ----
    // instruction q (0..31) of an operand covers rows 8q .. 8q+7; lane -> row 8q + (lane >> 3), LDS slot chunk lane & 7,
    // source chunk (lane & 7) ^ ((row >> 1) & 7) = (lane & 7) ^ (((lane >> 4) + 4 (q & 1)) & 7)
    const char* pa[2];
    const char* pb[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int sw = ((lane >> 4) + 4 * par) & 7;
        const int r = wave * 64 + (lane >> 3) + 8 * par;
        pa[par] = reinterpret_cast<const char*>(p.A + (size_t)(((tm & p.amask) << 8) + r) * p.lda) + (((lane & 7) ^ sw) << 4);
        pb[par] = reinterpret_cast<const char*>(p.B + (size_t)(((tn & p.bmask) << 8) + r) * p.ldb) + (((lane & 7) ^ sw) << 4);
    }
    const size_t stepa = (size_t)16 * p.lda * 2, stepb = (size_t)16 * p.ldb * 2;
    char* ldsw = smem + wave * 8192;           // this wave's 64 rows of the A image; + 32768 for B
#ifdef NO_GLDS
#define GLDS16(a, b) do { if (p.K < 0) glds16(a, b); } while (0)
#else
#define GLDS16(a, b) glds16(a, b)
#endif
#define ISSUE_A(kt, i)                                                                                        \
    GLDS16(pa[(i) & 1] + ((i) >> 1) * stepa + (size_t)(kt) * 128, ldsw + (((kt) & 1) * SLOT) + (i) * 1024)
#define ISSUE_B(kt, i)                                                                                        \
    GLDS16(pb[(i) & 1] + ((i) >> 1) * stepb + (size_t)(kt) * 128, ldsw + (((kt) & 1) * SLOT) + 32768 + (i) * 1024)

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fsw = (frow >> 1) & 7;
    const unsigned lds0 = lds_addr(smem);
    unsigned la[2], lb[2];     // lane address of k-step ks in slot 0: A block row 0 / B block row 0 of this wave
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned fo = frow * 128 + (((4 * ks + (lane >> 4)) ^ fsw) << 4);
        la[ks] = lds0 + wm * 16384 + fo;
        lb[ks] = lds0 + 32768 + wn * 16384 + fo;
    }
    u32x4 fa[2][8], fb[2][8];  // [register buffer][fragment]

    // one read of the 16 of a k-step: r = 0..7 -> B fragment r, r = 8..15 -> A fragment r - 8
#ifdef NO_RD
#define RD_GATE (p.K < 0)
#else
#define RD_GATE true
#endif
#define RD_ONE(buf, r, aA, aB)                                                                                \
    if (RD_GATE) do {                                                                                                    \
        if constexpr ((r) < 8) RD(fb[buf][(r) & 7], aB, ((r) & 7) * 2048);                                    \
        else RD(fa[buf][(r) & 7], aA, ((r) & 7) * 2048);                                                      \
    } while (0)
#define WAIT_FRAGS(buf)                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                       \
                 : "+v"(fa[buf][0]), "+v"(fa[buf][1]), "+v"(fa[buf][2]), "+v"(fa[buf][3]), "+v"(fa[buf][4]),  \
                   "+v"(fa[buf][5]), "+v"(fa[buf][6]), "+v"(fa[buf][7]), "+v"(fb[buf][0]), "+v"(fb[buf][1]),  \
                   "+v"(fb[buf][2]), "+v"(fb[buf][3]), "+v"(fb[buf][4]), "+v"(fb[buf][5]), "+v"(fb[buf][6]),  \
                   "+v"(fb[buf][7])                                                                           \
                 :: "memory")
    // MFMA group g (0..15) of a k-step: row block i = g >> 1, column blocks 4 (g & 1) .. +3
#define MM4(buf, g)                                                                                           \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                        \
        const int i_ = (g) >> 1, j_ = 4 * ((g) & 1) + jj;                                                     \
        /* inline asm with the accumulator TIED (dst = srcC, AGPR): as a builtin hipcc allocated dst != srcC  \
           and rotated all 256 accumulator registers through copies every iteration */                       \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                                \
                     : "+a"(acc[i_][j_]) : "v"(fb[buf][j_]), "v"(fa[buf][i_]));                               \
    }                                                                                                         \
    SB();

    // ---- prologue ----
#pragma unroll
    for (int i = 0; i < 8; ++i) { ISSUE_A(0, i); ISSUE_B(0, i); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { ISSUE_A(1, i); ISSUE_B(1, i); }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SB();
    {
        const unsigned aA = la[0], aB = lb[0];
        RD_ONE(0, 0, aA, aB); RD_ONE(0, 1, aA, aB); RD_ONE(0, 2, aA, aB); RD_ONE(0, 3, aA, aB);
        RD_ONE(0, 4, aA, aB); RD_ONE(0, 5, aA, aB); RD_ONE(0, 6, aA, aB); RD_ONE(0, 7, aA, aB);
        RD_ONE(0, 8, aA, aB); RD_ONE(0, 9, aA, aB); RD_ONE(0, 10, aA, aB); RD_ONE(0, 11, aA, aB);
        RD_ONE(0, 12, aA, aB); RD_ONE(0, 13, aA, aB); RD_ONE(0, 14, aA, aB); RD_ONE(0, 15, aA, aB);
    }

    // One K-tile.  RDN: K-tile g+1 exists (its k-step 0 is read into buffer 0); ISS: K-tile g+2 exists (issued into the
    // slot of K-tile g).  Compile-time so that the steady state is straight-line code.
    auto ktile = [&](int g, auto rdn_, auto iss_) {
        constexpr bool RDN = decltype(rdn_)::value, ISS = decltype(iss_)::value;
        const unsigned so = (unsigned)(g & 1) * SLOT, sn = SLOT - so;
        // ---- k-step 0 of K-tile g: MFMAs on register buffer 0, reads of k-step 1 into buffer 1 ----
        WAIT_FRAGS(0);
        {
            const unsigned aA = la[1] + so, aB = lb[1] + so;
            MM4(0, 0)  RD_ONE(1, 0, aA, aB);  RD_ONE(1, 1, aA, aB);  SB();
            MM4(0, 1)  RD_ONE(1, 2, aA, aB);  RD_ONE(1, 3, aA, aB);  SB();
            MM4(0, 2)  RD_ONE(1, 4, aA, aB);  RD_ONE(1, 5, aA, aB);  SB();
            MM4(0, 3)  RD_ONE(1, 6, aA, aB);  RD_ONE(1, 7, aA, aB);  SB();
            MM4(0, 4)  RD_ONE(1, 8, aA, aB);  RD_ONE(1, 9, aA, aB);  SB();
            MM4(0, 5)  RD_ONE(1, 10, aA, aB); RD_ONE(1, 11, aA, aB); SB();
            MM4(0, 6)  RD_ONE(1, 12, aA, aB); RD_ONE(1, 13, aA, aB); SB();
            MM4(0, 7)  RD_ONE(1, 14, aA, aB); RD_ONE(1, 15, aA, aB); SB();
            MM4(0, 8) MM4(0, 9) MM4(0, 10) MM4(0, 11) MM4(0, 12) MM4(0, 13) MM4(0, 14) MM4(0, 15)
        }
        // ---- k-step 1: MFMAs on buffer 1; K-tile g+1 has landed (wait + barrier), its k-step 0 -> buffer 0;
        //      slot of K-tile g is free after the barrier (every wave has its k-step-1 fragments): K-tile g+2 -> it ----
        WAIT_FRAGS(1);
        MM4(1, 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SB();
        {
            const unsigned aA = la[0] + sn, aB = lb[0] + sn;
#define STEP(m, r0, i)                                                                                        \
            MM4(1, m)                                                                                         \
            if constexpr (RDN) { RD_ONE(0, r0, aA, aB); RD_ONE(0, r0 + 1, aA, aB); }                          \
            if constexpr (ISS) { ISSUE_A(g + 2, i); ISSUE_B(g + 2, i); }                                      \
            SB();
            STEP(1, 0, 0) STEP(2, 2, 1) STEP(3, 4, 2) STEP(4, 6, 3) STEP(5, 8, 4) STEP(6, 10, 5) STEP(7, 12, 6) STEP(8, 14, 7)
#undef STEP
            MM4(1, 9) MM4(1, 10) MM4(1, 11) MM4(1, 12) MM4(1, 13) MM4(1, 14) MM4(1, 15)
        }
    };
    {
        int g = 0;
        for (; g + 2 < nk; ++g) ktile(g, std::true_type{}, std::true_type{});
        ktile(g, std::true_type{}, std::false_type{});
        ktile(g + 1, std::false_type{}, std::false_type{});
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results (hipcc does not see asm MFMAs)
    // ---- epilogue (plain): acc[i][j] holds C rows m = i*16 + (lane & 15), columns n = j*16 + (lane >> 4)*4 .. +3 ----
    const int em = m0 + wm * 128 + (lane & 15), en = n0 + wn * 128 + (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 v = acc[i][j];
            uint2 o = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            *reinterpret_cast<uint2*>(p.C + (size_t)(em + i * 16) * p.ldc + en + j * 16) = o;
        }
}

__global__ void naive_kernel(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K) {
    const int n = blockIdx.x * 16 + (threadIdx.x & 15), m = blockIdx.y * 16 + (threadIdx.x >> 4);
    float s = 0.f;
    for (int k = 0; k < K; ++k)
        s += __uint_as_float((uint32_t)A[(size_t)m * K + k] << 16) * __uint_as_float((uint32_t)B[(size_t)n * K + k] << 16);
    C[(size_t)m * N + n] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static void fill(bf16_t* h, size_t n, unsigned seed) {
    unsigned s = seed;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float f = ((int)((s >> 9) & 0xffff) - 32768) * (1.0f / 32768.0f);
        uint32_t u; memcpy(&u, &f, 4);
        h[i] = (bf16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16);
    }
}

static int g_amask = -1, g_bmask = -1;
static int run(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int group_n) {
    G4Args a{A, B, C, M, N, K, K, K, N, group_n, g_amask, g_bmask};
    const int tiles = (M >> 8) * (N >> 8);
    hipLaunchKernelGGL(gemm_g4_kernel, dim3(tiles), dim3(256), 2 * SLOT, 0, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int main() {
    CK(hipFuncSetAttribute((const void*)gemm_g4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT));
    {   // ---- correctness: 512 x 768 x 320(+) against the naive kernel ----
        const int M = 512, N = 768, K = 320;
        bf16_t *hA = (bf16_t*)malloc((size_t)M * K * 2), *hB = (bf16_t*)malloc((size_t)N * K * 2);
        fill(hA, (size_t)M * K, 1); fill(hB, (size_t)N * K, 2);
        bf16_t *A, *B, *C; float* R;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2));
        CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&R, (size_t)M * N * 4));
        CK(hipMemcpy(A, hA, (size_t)M * K * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hB, (size_t)N * K * 2, hipMemcpyHostToDevice));
        if (run(A, B, C, M, N, K, 3)) { printf("launch failed\n"); return 1; }
        hipLaunchKernelGGL(naive_kernel, dim3(N / 16, M / 16), dim3(256), 0, 0, A, B, R, M, N, K);
        CK(hipDeviceSynchronize());
        bf16_t* hC = (bf16_t*)malloc((size_t)M * N * 2); float* hR = (float*)malloc((size_t)M * N * 4);
        CK(hipMemcpy(hC, C, (size_t)M * N * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hR, R, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (size_t i = 0; i < (size_t)M * N; ++i) {
            uint32_t u = (uint32_t)hC[i] << 16; float c; memcpy(&c, &u, 4);
            maxerr = fmax(maxerr, fabs((double)c - hR[i])); maxref = fmax(maxref, fabs((double)hR[i]));
        }
        printf("check %dx%dx%d: max |err| %.4g (max |ref| %.4g) -> %s\n", M, N, K, maxerr, maxref,
               maxerr <= maxref * 0.01 ? "OK" : "WRONG");
#if !defined(NO_GLDS) && !defined(NO_RD)
        if (!(maxerr <= maxref * 0.01)) return 1;
#endif
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(C); (void)hipFree(R);
    }
    if (getenv("G4_AMASK")) g_amask = atoi(getenv("G4_AMASK"));
    if (getenv("G4_BMASK")) g_bmask = atoi(getenv("G4_BMASK"));
    printf("amask %d bmask %d\n", g_amask, g_bmask);
    const int shapes[][4] = {{51456, 2304, 768, 3}, {51456, 768, 768, 3}, {51456, 3072, 768, 4}, {51456, 768, 3072, 3},
                             {51456, 768, 2304, 3}, {8192, 8192, 8192, 4}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        bf16_t *A, *B, *C;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
        bf16_t* h = (bf16_t*)malloc((size_t)M * K * 2);
        fill(h, (size_t)M * K, 3); CK(hipMemcpy(A, h, (size_t)M * K * 2, hipMemcpyHostToDevice));
        free(h); h = (bf16_t*)malloc((size_t)N * K * 2);
        fill(h, (size_t)N * K, 4); CK(hipMemcpy(B, h, (size_t)N * K * 2, hipMemcpyHostToDevice));
        free(h);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) run(A, B, C, M, N, K, s[3]);
        CK(hipEventRecord(e0, 0));
        const int iters = 20;
        for (int i = 0; i < iters; ++i) run(A, B, C, M, N, K, s[3]);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms / iters * 1e3;
        printf("g4 M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s\n", M, N, K, us, 2.0 * M * N * K / us * 1e-6);
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(C);
    }
    return 0;
}
