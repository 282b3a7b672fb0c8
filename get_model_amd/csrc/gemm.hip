// bf16 MFMA GEMMs for gfx950 (MI355X).
//
//   gemm_tn : C[M,N] = A[M,K] * B[N,K]^T      (both operands K-contiguous)  -- forward linears and
//             dgrad (with a pre-transposed weight copy).  Fused epilogues: bias, bias+GELU(erf),
//             bias+residual+DropPath-scale (fp32 residual stream), GELU' (fc1 dgrad), region-embed
//             row remap (+mask-token blend).
//   gemm_nt_rows : C[N1,N2] = X[M,N1]^T * Y[M,N2]  (contraction over the ROW index of both operands)
//             -- wgrad.  Operand fragments come out of LDS with ds_read_b64_tr_b16 (hardware transpose);
//             split over M into fp32 slabs that reduce_slabs sums deterministically.
//
// Structure (v1): 128x128 output tile, BK=64, 4 waves (2x2, 64x64 per wave, 4x4 tiles of
// v_mfma_f32_16x16x32_bf16), operands staged global->LDS with 16-byte global_load_lds into an
// XOR-swizzled image (conflict-free ds_read_b128 / ds_read_b64_tr_b16), two LDS buffers, one barrier per
// K-step, XCD-aware tile order.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <atomic>

enum {
    EPI_BF16 = 0,         // out0(bf16) = acc (+bias)
    EPI_BIAS_GELU = 1,    // u = acc+bias ; out0(bf16)=u ; out1(bf16)=gelu(u)
    EPI_BIAS_RESID = 2,   // out0(f32) = resid + rowscale[row/T] * (acc+bias)
    EPI_DGELU = 3,        // out0(bf16) = acc * gelu'(aux)
    EPI_EMBED = 4,        // out0(f32)[b*T+1+n] = mask ? mask_token : acc+bias
    EPI_F32 = 5,          // out0(f32) = acc (+bias)
    EPI_BIAS_GELUG = 6,   // u = acc+bias ; out0(bf16)=gelu'(u) ; out1(bf16)=gelu(u)   (training forward of fc1)
    EPI_DMUL = 7,         // out0(bf16) = acc * aux   (+ column sums)                  (its backward: aux = gelu'(u))
};
// The training path stores gelu'(u) instead of u: both come out of one gelu_parts() in the forward epilogue, and the
// backward epilogue becomes a multiply (the dGELU form re-evaluates erf per element: ~40 us of a 360-us GEMM).
template <int EPI> struct EpiKind {
    static constexpr bool gelu = (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELUG);
    static constexpr bool dgelu = (EPI == EPI_DGELU || EPI == EPI_DMUL);
};
template <int EPI>
__device__ __forceinline__ float dgelu_factor(float a) { return EPI == EPI_DMUL ? a : dgelu_f(a); }
// (first output, gelu(x)) of the two-output forward epilogues
template <int EPI>
__device__ __forceinline__ void gelu_pair(float x, float& o0, float& o1) {
    float Phi, pdf;
    gelu_parts(x, Phi, pdf);
    o1 = x * Phi;
    o0 = (EPI == EPI_BIAS_GELUG) ? fmaf(x, pdf, Phi) : x;
}

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* B;
    int M, N, K, lda, ldb;
    void* out0;
    void* out1;
    int ldo;
    const float* bias;
    const float* resid;
    const float* rowscale;
    int rows_per_sample;
    const bf16_t* aux;
    int ldaux;
    const uint8_t* mask;
    const float* mask_token;
    int regions_per_sample;
    float* colsum;   // EPI_DGELU only: partial column sums of out0 (fp32, pre-rounding) [rows][N]; may be null
    int row0;        // first row of this launch within the caller's problem (split launches; only the row scale uses it)
    int prescaled;   // EPI_BIAS_RESID: A already carries the row scale (DropPath fold, functional.py): out = resid + acc +
                     // rowscale * bias instead of resid + rowscale * (acc + bias)
    int blocked;     // EPI_BIAS_GELUG: out0 / EPI_DMUL: aux is in the BLOCKED layout (getmi355.h, flags bit 1): a private
                     // hand-over format between the two, [row block of 16][column block of 64][2][64 lanes][16 B] -- the
                     // accumulator layout of a wave's 16 x 64 block, so it leaves and returns as whole 1-KiB pieces
};

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (128 * 128)          // 128 rows x 64 bf16
#define GEMM_LDS (4 * TILE_BYTES)       // 2 buffers x (A + B)

// Epilogue of one wave's MT x NT grid of 16x16 accumulator tiles.  acc[mt][nt] holds C[em + 16*mt][en + 16*nt .. +3]
// (row = lane&15, 4 consecutive columns per lane).  All global LOADS (bias, residual, pre-activation, row scale)
// are issued first from clamped addresses -- no predication, one latency exposure per tile instead of one per
// sub-tile -- then the arithmetic and the predicated stores.
template <int EPI, int MT, int NT, int MS, int NS, typename Get>
__device__ __forceinline__ void epilogue_quads(const GemmArgs& p, Get acc_get, int em, int en, int prow = 0) {
    float4 bv[NT];
    const bool has_bias = !EpiKind<EPI>::dgelu && p.bias != nullptr;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = min(en + nt * NS, p.N - 4);
        bv[nt] = has_bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (EPI == EPI_BIAS_RESID) {
        float4 rv[MT][NT];
        float sc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = min(em + mt * MS, p.M - 1);
            sc[mt] = p.rowscale ? p.rowscale[(m + p.row0) / p.rows_per_sample] : 1.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                rv[mt][nt] = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldo + min(en + nt * NS, p.N - 4));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int m = em + mt * MS, n = en + nt * NS;
                if (m < p.M && n < p.N) {
                    const f32x4 v = acc_get(mt, nt);
                    const float4 r = rv[mt][nt], b = bv[nt];
                    const float s = sc[mt], sv = p.prescaled ? 1.0f : s;
                    *reinterpret_cast<float4*>((float*)p.out0 + (size_t)m * p.ldo + n) =
                        make_float4(r.x + (sv * v[0] + s * b.x), r.y + (sv * v[1] + s * b.y), r.z + (sv * v[2] + s * b.z),
                                    r.w + (sv * v[3] + s * b.w));
                }
            }
        return;
    }
    if (EpiKind<EPI>::dgelu) {
        uint2 uv[MT][NT];
        float4 cs[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) cs[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                uv[mt][nt] = *reinterpret_cast<const uint2*>(p.aux + (size_t)min(em + mt * MS, p.M - 1) * p.ldaux +
                                                             min(en + nt * NS, p.N - 4));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int m = em + mt * MS, n = en + nt * NS;
                if (m < p.M && n < p.N) {
                    const f32x4 v = acc_get(mt, nt);
                    const uint2 u = uv[mt][nt];
                    const float d0 = v[0] * dgelu_factor<EPI>(bf2f(u.x & 0xffff)), d1 = v[1] * dgelu_factor<EPI>(bf2f(u.x >> 16));
                    const float d2 = v[2] * dgelu_factor<EPI>(bf2f(u.y & 0xffff)), d3 = v[3] * dgelu_factor<EPI>(bf2f(u.y >> 16));
                    *reinterpret_cast<uint2*>((bf16_t*)p.out0 + (size_t)m * p.ldo + n) = pack4bf(d0, d1, d2, d3);
                    cs[nt].x += d0; cs[nt].y += d1; cs[nt].z += d2; cs[nt].w += d3;
                }
            }
        if (p.colsum) {
            // bias gradient of the previous Linear = column sums of this output: reduce the wave's rows
            // (in-lane over mt, then over the 16 lanes that share a column group) and write one partial row.
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float4 c = cs[nt];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    c.x += __shfl_xor(c.x, o, 64); c.y += __shfl_xor(c.y, o, 64);
                    c.z += __shfl_xor(c.z, o, 64); c.w += __shfl_xor(c.w, o, 64);
                }
                const int n = en + nt * NS;
                if ((threadIdx.x & 15) == 0 && n < p.N)
                    *reinterpret_cast<float4*>(p.colsum + (size_t)prow * p.N + n) = c;
            }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int m = em + mt * MS, n = en + nt * NS;
            if (m >= p.M || n >= p.N) continue;
            f32x4 v = acc_get(mt, nt);
            v[0] += bv[nt].x; v[1] += bv[nt].y; v[2] += bv[nt].z; v[3] += bv[nt].w;
            if (EPI == EPI_BF16) {
                *reinterpret_cast<uint2*>((bf16_t*)p.out0 + (size_t)m * p.ldo + n) = pack4bf(v[0], v[1], v[2], v[3]);
            } else if (EPI == EPI_F32) {
                *reinterpret_cast<float4*>((float*)p.out0 + (size_t)m * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else if (EpiKind<EPI>::gelu) {
                float h0, h1, h2, h3, g0, g1, g2, g3;
                gelu_pair<EPI>(v[0], h0, g0); gelu_pair<EPI>(v[1], h1, g1);
                gelu_pair<EPI>(v[2], h2, g2); gelu_pair<EPI>(v[3], h3, g3);
                if (EPI == EPI_BIAS_GELUG && p.rowscale) {      // DropPath fold: both outputs carry the sample's factor
                    const float s_ = p.rowscale[(m + p.row0) / p.rows_per_sample];
                    h0 *= s_; h1 *= s_; h2 *= s_; h3 *= s_; g0 *= s_; g1 *= s_; g2 *= s_; g3 *= s_;
                }
                if (p.out0)      // NULL: inference, only gelu(u) is wanted
                    *reinterpret_cast<uint2*>((bf16_t*)p.out0 + (size_t)m * p.ldo + n) = pack4bf(h0, h1, h2, h3);
                *reinterpret_cast<uint2*>((bf16_t*)p.out1 + (size_t)m * p.ldo + n) = pack4bf(g0, g1, g2, g3);
            } else if (EPI == EPI_EMBED) {
                const int b = m / p.regions_per_sample, r = m - b * p.regions_per_sample;
                const size_t orow = (size_t)b * (p.regions_per_sample + 1) + 1 + r;
                if (p.mask && p.mask[m]) {
                    const float4 t = *reinterpret_cast<const float4*>(p.mask_token + n);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                }
                *reinterpret_cast<float4*>((float*)p.out0 + orow * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
}

template <int EPI, int MT, int NT>
__device__ __forceinline__ void epilogue_tiles(const GemmArgs& p, f32x4 (&acc)[MT][NT], int em, int en, int prow = 0) {
    epilogue_quads<EPI, MT, NT, 16, 16>(p, [&](int a, int b) -> f32x4 { return acc[a][b]; }, em, en, prow);
}

// ------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] * B[N,K]^T
// LDS image of one operand tile: [128 rows][8 chunks of 16 B], chunk index XORed with (row>>1)&7 so that
// the 16-lane groups of ds_read_b128 (rows r..r+15 at one K-chunk) touch 16 distinct 16-B slots.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / tiles_n, tn = lid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging: 16 wave-instructions (1 KiB = 8 rows) per operand tile, 4 per wave ----
    const char* srcA[4];
    const char* srcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int ra = min(m0 + row, p.M - 1);
        const int rb = min(n0 + row, p.N - 1);
        srcA[j] = reinterpret_cast<const char*>(p.A + (size_t)ra * p.lda) + chunk * 16;
        srcB[j] = reinterpret_cast<const char*>(p.B + (size_t)rb * p.ldb) + chunk * 16;
    }
    auto stage = [&](int buf, int kt) {
        char* a = smem + buf * (2 * TILE_BYTES);
        char* b = a + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(srcA[j] + (size_t)kt * (BK * 2), a + (wave * 4 + j) * 1024);
            glds16(srcB[j] + (size_t)kt * (BK * 2), b + (wave * 4 + j) * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane fragment offsets inside a tile (bytes): row (lane&15), K-chunk 4*ks + (lane>>4), swizzled
    const int frow = lane & 15;
    const int fsw = (frow >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * 128 + (((4 * ks + (lane >> 4)) ^ fsw) << 4);

    const int nk = p.K / BK;
    stage(0, 0);
    __syncthreads();   // (vmcnt(0) is emitted before the barrier while LDS-DMA is pending)
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* a = smem + cur * (2 * TILE_BYTES) + (wr * 64) * 128;
        const char* b = smem + cur * (2 * TILE_BYTES) + TILE_BYTES + (wc * 64) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const bf16x8*>(a + i * 16 * 128 + foff[ks]);
                fb[i] = *reinterpret_cast<const bf16x8*>(b + i * 16 * 128 + foff[ks]);
            }
            // first MFMA operand = weight rows (n), second = activation rows (m):
            // D[i][j]: i -> n = (lane>>4)*4 + reg, j -> m = lane&15  => each lane owns 4 consecutive n.
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }

    const int em = m0 + wr * 64 + (lane & 15);
    const int en = n0 + wc * 64 + (lane >> 4) * 4;
    epilogue_tiles<EPI, 4, 4>(p, acc, em, en, tm * 2 + wr);
}

// ------------------------------------------------------------------------------------------------
// gemm_tn, large-tile pipeline: 256x256 output tile, BK=64, 8 waves (2 M x 4 N, 128x64 per wave), ONE
// workgroup per CU (128 KiB LDS: two K-tile buffers of A[256][64] + B[256][64]).
//
// A K-tile is consumed in four phases, one 64x32 quadrant of the wave's output each (16 MFMAs):
//     P1 (rows 0-63 , cols 0-31 ) reads A-half0 + B-half0 fragments      P2 (rows 0-63 , cols 32-63) reads B-half1
//     P3 (rows 64-127, cols 32-63) reads A-half1                          P4 (rows 64-127, cols 0-31 ) reads nothing
// and is STAGED as four half-tiles of 16 KiB in the order they are needed (A0, B0, B1, A1).  Half-tile q
// (= 4*tile + j) is issued in phase q-7, i.e. ~1.75 K-tiles (5 half-tiles = 80 KiB per CU) stay in flight
// across barriers; each reading phase waits with a COUNTED `s_waitcnt vmcnt(10)` (never 0 in steady state)
// and one raw s_barrier.  Hazards (all waves pass the same barriers):
//   RAW: phase phi needs half-tiles <= phi+1; every wave waits for its own share (vmcnt) before the barrier.
//   WAR: half-tile q overwrites the slot of q-8, whose last read was in phase q-8 (A0,B0), q-9 (B1: read in
//        P2 = phase q-8... ) -- in all four cases at least one barrier lies between that read phase and
//        phase q-7, and a wave reaches a barrier only after its MFMAs consumed the fragments it read.
// ------------------------------------------------------------------------------------------------
#define G2_LDS 131072

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// wait until at most `halftiles` of this wave's youngest half-tile loads (2 instructions each) are pending
__device__ __forceinline__ void wait_halftiles(int halftiles) {
    if (halftiles >= 5) wait_vmcnt<10>();
    else if (halftiles == 4) wait_vmcnt<8>();
    else if (halftiles == 3) wait_vmcnt<6>();
    else if (halftiles == 2) wait_vmcnt<4>();
    else if (halftiles == 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
}
__device__ __forceinline__ void phase_barrier() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// Rows of an 8-wave output tile.  RB1 = 2 / 3 / 4: both wave rows own 64 + 16 * RB1 rows (192 / 224 / 256-row tiles).
// RB1 = 5 (round 3, persistent kernel only): the ASYMMETRIC 208-row tile -- wave row 0 owns 64 + 48 rows, wave row 1
// 64 + 32 -- for N = 768 at M = 51 456: 248 x 3 = 744 tiles = 2.9 rounds of 256 CUs, every CU busy in all three, where
// 224-row tiles (690) leave 30 % of the CUs idle in the third round.  The two waves that share a SIMD are one of each
// wave row, so the matrix pipes stay balanced.
template <int RB1> struct TileRows {
    static constexpr int rb0 = RB1 == 5 ? 3 : RB1, rb1 = RB1 == 5 ? 2 : RB1;      // real 16-row blocks of the second half
    static constexpr int hw0 = 64 + 16 * rb0, hw1 = 64 + 16 * rb1, bm = hw0 + hw1;
    static constexpr int rbmax = rb0, rbmin = rb1;
};

// (defined with the persistent kernel below; the one-tile kernel uses them for its fp32 residual epilogue)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ u32x4 make_rsrc(const void* ptr, unsigned bytes);
template <int EPI, int RB1>
__device__ __forceinline__ void epilogue_p(const GemmArgs& p, f32x4 (&acc)[8][4], int m0w, int n0w, int prow, u32x4 rs0,
                                           u32x4 rs1, u32x4 rsc, const float* bias_lds, char* stage, int row_end);

// RB1 = 16-row blocks of each wave's SECOND 64-row half that are real output (4: 256-row tile, 3: 224-row tile).
// The 224-row tile exists for tile quantisation only: M = 51 456, N = 768 is 603 tiles of 256 rows = 2.36 rounds on
// 256 CUs -> 3 rounds, but 690 tiles of 224 rows = 2.70 -> 3 rounds of 7/8 the length (-12.5 %).  Staging, LDS
// layout and phases are unchanged (the 16 unused rows per wave-row are staged from the neighbouring rows and ignored);
// phases P3 / P4 issue 12 instead of 16 MFMAs and the epilogue skips the last row block.
template <int EPI, int RB1>
__global__ __launch_bounds__(512, 2) void gemm_tn256_kernel(GemmArgs p) {
    constexpr int BMv = 128 + 32 * RB1, HW = BMv / 2;          // tile rows, rows per wave-row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int tiles_n = (p.N + 255) >> 8;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / tiles_n, tn = lid - tm * tiles_n;
    const int m0 = tm * BMv, n0 = tn << 8;
    const int nk = p.K >> 6;
    const int nq = nk << 2;   // number of half-tiles

    // ---- staging set-up: kind j = 0:A-half0 1:B-half0 2:B-half1 3:A-half1 ; 2 wave-instructions per kind ----
    const char* src[4][2];
    int dst[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int hr0 = (wave * 2 + jj) * 8;          // wave-uniform first half-row of this instruction
            const int hr = hr0 + (lane >> 3);
            int row, row0;
            if (j == 0 || j == 3) {
                const int h = (j == 3);
                row = ((hr >> 6) << 7) + h * 64 + (hr & 63);
                row0 = ((hr0 >> 6) << 7) + h * 64 + (hr0 & 63);
                const int g = min(m0 + (hr >> 6) * HW + h * 64 + (hr & 63), p.M - 1);
                src[j][jj] = reinterpret_cast<const char*>(p.A + (size_t)g * p.lda) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
                dst[j][jj] = row0 * 128;
            } else {
                const int h = j - 1;
                row = ((hr >> 5) << 6) + h * 32 + (hr & 31);
                row0 = ((hr0 >> 5) << 6) + h * 32 + (hr0 & 31);
                const int g = min(n0 + row, p.N - 1);
                src[j][jj] = reinterpret_cast<const char*>(p.B + (size_t)g * p.ldb) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
                dst[j][jj] = 32768 + row0 * 128;
            }
        }
    // issue half-tile (tile t, kind J)
#define G2_ISSUE(t, J)                                                                         \
    do {                                                                                       \
        char* b_ = smem + (((t)&1) << 16);                                                     \
        glds16(src[J][0] + (size_t)(t) * 128, b_ + dst[J][0]);                                 \
        glds16(src[J][1] + (size_t)(t) * 128, b_ + dst[J][1]);                                 \
    } while (0)

    // v_mfma_f32_16x16x32_bf16 on purpose: the 32x32x16 form of this same schedule measured ~10 % SLOWER on
    // random data (the chip holds a higher clock on the 16x16 shape, MI355X_MICROARCH.md DVFS item 7).
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fsw = (frow >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * 128 + (((4 * ks + (lane >> 4)) ^ fsw) << 4);
    const int a_base = (wm * 128) * 128;                 // + mh*64*128 + mt*16*128
    const int b_base = 32768 + (wn * 64) * 128;          // + nh*32*128 + nt*16*128

    // ---- prologue: half-tiles 0..6 ----
    G2_ISSUE(0, 0); G2_ISSUE(0, 1); G2_ISSUE(0, 2); G2_ISSUE(0, 3);
    if (nk > 1) { G2_ISSUE(1, 0); G2_ISSUE(1, 1); G2_ISSUE(1, 2); }

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#define G2_LOAD_A(MH)                                                                                      \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)      \
        fa[mt][ks] = *reinterpret_cast<const bf16x8*>(buf + a_base + ((MH)*64 + mt * 16) * 128 + foff[ks]);
#define G2_LOAD_B(FB, NH)                                                                                  \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)      \
        FB[nt][ks] = *reinterpret_cast<const bf16x8*>(buf + b_base + ((NH)*32 + nt * 16) * 128 + foff[ks]);
#define G2_MMA(MH, NH, FB)                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mt = 0; mt < ((MH) ? RB1 : 4); ++mt) \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                   \
            acc[(MH)*4 + mt][(NH)*2 + nt] =                                                                \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[nt][ks], fa[mt][ks], acc[(MH)*4 + mt][(NH)*2 + nt], 0, 0, 0);

    // STEADY (compile time): s + 2 < nk -- full-depth waits and unconditional issues, no compare / branch ladders at the
    // phase heads (see the persistent kernel: worth 5-15 % there).
    auto ktile = [&](int s, auto steady_) {
        constexpr bool STEADY = decltype(steady_)::value;
        const char* buf = smem + ((s & 1) << 16);
        const int phi = s << 2;
        // issued before phase phi: half-tiles <= min(phi+6, nq-1); needed: <= phi+1
        // ---- P1 ----
        if constexpr (STEADY) wait_vmcnt<10>(); else wait_halftiles(min(phi + 6, nq - 1) - (phi + 1));
        phase_barrier();
        G2_LOAD_A(0)
        G2_LOAD_B(fb0, 0)
        if constexpr (STEADY) G2_ISSUE(s + 1, 3); else if (s + 1 < nk) G2_ISSUE(s + 1, 3);
        G2_MMA(0, 0, fb0)
        // ---- P2 ----
        if constexpr (STEADY) wait_vmcnt<10>(); else wait_halftiles(min(phi + 7, nq - 1) - (phi + 2));
        phase_barrier();
        G2_LOAD_B(fb1, 1)
        if constexpr (STEADY) G2_ISSUE(s + 2, 0); else if (s + 2 < nk) G2_ISSUE(s + 2, 0);
        G2_MMA(0, 1, fb1)
        // ---- P3 ----
        if constexpr (STEADY) wait_vmcnt<10>(); else wait_halftiles(min(phi + 8, nq - 1) - (phi + 3));
        phase_barrier();
        G2_LOAD_A(1)
        if constexpr (STEADY) G2_ISSUE(s + 2, 1); else if (s + 2 < nk) G2_ISSUE(s + 2, 1);
        G2_MMA(1, 1, fb1)
        // ---- P4 (no new data, no barrier) ----
        if constexpr (STEADY) G2_ISSUE(s + 2, 2); else if (s + 2 < nk) G2_ISSUE(s + 2, 2);
        G2_MMA(1, 0, fb0)
    };
    {
        int s = 0;
        for (; s + 2 < nk; ++s) ktile(s, std::true_type{});
        for (; s < nk; ++s) ktile(s, std::false_type{});
    }
#undef G2_ISSUE
#undef G2_LOAD_A
#undef G2_LOAD_B
#undef G2_MMA

    if constexpr (EPI == EPI_BIAS_RESID) {
        // fp32 residual epilogue through the wave-private LDS stage of epilogue_p: every store instruction writes whole
        // 128-byte lines (8 rows x 128 B) instead of 64-byte pieces on 16 rows.  The operand ring is free here: every
        // LDS-DMA has landed (the last phases waited vmcnt(0)) and the barrier below orders all fragment reads first.
        __syncthreads();
        float* bias_img = reinterpret_cast<float*>(smem);
        if (p.bias != nullptr && tid < 64)
            *reinterpret_cast<float4*>(bias_img + tid * 4) = *reinterpret_cast<const float4*>(p.bias + min(n0 + tid * 4, p.N - 4));
        __syncthreads();
        const u32x4 rs0 = make_rsrc(p.out0, (unsigned)((size_t)p.M * p.ldo * 4));
        const float* bl = p.bias != nullptr ? bias_img + wn * 64 + (lane >> 4) * 4 : nullptr;
        epilogue_p<EPI, RB1>(p, acc, m0 + wm * HW, n0 + wn * 64, tm * 4 + wm * 2, rs0, rs0, rs0, bl, smem + 1024 + wave * 2048, p.M);
        return;
    }
    const int em = m0 + wm * HW + (lane & 15);
    const int en = n0 + wn * 64 + (lane >> 4) * 4;
    epilogue_tiles<EPI, 4, 4>(p, *reinterpret_cast<f32x4(*)[4][4]>(&acc[0]), em, en, tm * 4 + wm * 2);
    epilogue_tiles<EPI, RB1, 4>(p, *reinterpret_cast<f32x4(*)[RB1][4]>(&acc[4]), em + 64, en, tm * 4 + wm * 2 + 1);
}

// ------------------------------------------------------------------------------------------------
// gemm_tn, PERSISTENT form of the 256x256 pipeline: one workgroup per CU walks its output tiles
// (tile = iteration*grid + xcd_remap(block)), and the half-tile stream simply continues from the last K-tile of
// one output tile into the first K-tiles of the next (the LDS ring never drains).  Two things are overlapped that
// the one-tile-per-workgroup form serialises, both visible in rocprof as the gap between its in-loop rate
// (~1.3 PF) and its end-to-end rate on output-heavy shapes (0.6 PF for fc1):
//   * the next tile's operand prefetch runs under the current tile's last phases and its epilogue;
//   * the epilogue's global STORES (HBM-bound: every CU used to write at the same time, then all computed) drain
//     under the next tile's MFMAs.
// vmcnt is one in-order counter for loads, LDS-DMA and stores, so a counted wait for a half-tile issued BEFORE an
// epilogue must also allow that epilogue's stores to stay outstanding: for the five reading phases whose needed
// half-tile predates the last epilogue the wait is vmcnt(10 + E) with E = the EXACT number of store instructions
// every wave executes in the epilogue.  Exactness is why the stores are inline-asm buffer stores (never merged,
// split or skipped; out-of-range lanes are dropped by the buffer bounds check instead of by a branch).
// Under-counting E would only wait longer; over-counting would read LDS before the data landed.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define OOB_OFF 0xFFFFFFF0u

__device__ __forceinline__ u32x4 make_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    return (u32x4){(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
// Cache policy of the persistent epilogue's output stores.  Streamed (" nt"): a round of 32 output tiles per XCD writes 4-8 MiB,
// the size of the XCD's L2, and evicted the operand blocks that are reused there; streamed stores leave them alone.
// Interleaved A/B in the C2 step, all epilogues alike: "" 34.29, " nt" 33.81, " sc1" 34.19, " sc0 sc1" 34.23,
// " sc0 sc1 nt" 33.93, " sc1 nt" 33.88, " sc0 nt" 33.86 ms.  Per epilogue (1 = streamed): GETMI_NT_EPI_{BF16,RESID,GELU,DGELU};
// taking one class out of the streamed set costs: GELU pair +0.37 ms, plain bf16 +0.15 (streaming only the outputs wider than
// 1024 columns: +0.13), dGELU form +0.10, fp32 residual -0.04 (so that one is stored normally: LayerNorm reads it next);
// streaming the epilogue's own READS (residual, stored derivative) costs +0.55 (the residual reads alone: +-0); streaming only
// the derivative of the GELU pair and storing gelu(u) normally for the fc2 GEMM that follows: +0.07.
#ifndef GETMI_NT_EPI_BF16
#define GETMI_NT_EPI_BF16 1
#endif
#ifndef GETMI_NT_EPI_RESID
#define GETMI_NT_EPI_RESID 0
#endif
#ifndef GETMI_NT_EPI_GELU
#define GETMI_NT_EPI_GELU 1
#endif
#ifndef GETMI_NT_EPI_DGELU
#define GETMI_NT_EPI_DGELU 1
#endif
#ifndef GETMI_NT_EPI_LOADS
#define GETMI_NT_EPI_LOADS 1     // the epilogue's read of the stored GELU derivative (read exactly once: 32.78 -> 32.41 ms/step
#endif                           // streamed, round 3, with the blocked layout's whole-line loads)
#ifndef GETMI_NT_EPI_RESID_LOAD
#define GETMI_NT_EPI_RESID_LOAD 0    // the fp32 residual read: streamed 33.07 -> 33.07 (+-0.2), left cached
#endif
template <int EPI> struct EpiNt {
    static constexpr bool value = EPI == EPI_BIAS_RESID ? (GETMI_NT_EPI_RESID != 0)
                                : EpiKind<EPI>::gelu    ? (GETMI_NT_EPI_GELU != 0)
                                : EpiKind<EPI>::dgelu   ? (GETMI_NT_EPI_DGELU != 0)
                                : EPI == EPI_BF16       ? (GETMI_NT_EPI_BF16 != 0) : false;
};
__device__ __forceinline__ void st_b64(u32x2 v, unsigned off, u32x4 rs) {
    asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(off), "s"(rs) : "memory");
}
template <bool NT = false>
__device__ __forceinline__ void st_b128(u32x4 v, unsigned off, u32x4 rs) {
    if constexpr (NT) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" ::"v"(v), "v"(off), "s"(rs) : "memory");
    else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(off), "s"(rs) : "memory");
}
__device__ __forceinline__ u32x2 pack4bf_v(float a, float b, float c, float d) {
    return (u32x2){pack2bf(a, b), pack2bf(c, d)};
}

__device__ __forceinline__ float row16_sum(float v) {      // DPP row_ror:n = 0x120 + n
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x124, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x122, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x121, 0xf, 0xf, true));
    return v;
}

template <int EPI, int RB1>
struct EpiStores {   // buffer-store instructions per wave and output tile (EXACT: used in the counted waits)
    static constexpr int value = ((EpiKind<EPI>::gelu || EPI == EPI_BIAS_RESID) ? 4 : 2) * (4 + TileRows<RB1>::rbmax);
};

// wait until at most 2*halftiles (+ EXTRA) vector-memory operations of this wave are outstanding
template <int EXTRA>
__device__ __forceinline__ void wait_halftiles_x(int halftiles) {
    constexpr int C5 = (10 + EXTRA > 63) ? 63 : 10 + EXTRA, C4 = (8 + EXTRA > 63) ? 63 : 8 + EXTRA,
                  C3 = (6 + EXTRA > 63) ? 63 : 6 + EXTRA, C2 = (4 + EXTRA > 63) ? 63 : 4 + EXTRA,
                  C1 = (2 + EXTRA > 63) ? 63 : 2 + EXTRA, C0 = (EXTRA > 63) ? 63 : EXTRA;
    if (halftiles >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C5) : "memory");
    else if (halftiles == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C4) : "memory");
    else if (halftiles == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C3) : "memory");
    else if (halftiles == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C2) : "memory");
    else if (halftiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C0) : "memory");
}

// Wave-private LDS transposition of one 16-row x 128-byte block, entirely in inline asm: a compiler-visible access
// to LDS that LDS-DMA also writes makes hipcc drain vmcnt to 0 first.  Writes, wait, reads, wait in ONE statement
// (outputs early-clobber), so no register is reused before the LDS has consumed / produced it.
__device__ __forceinline__ void stage_w64x4_r128x2(unsigned wa, unsigned ra, u32x2 q0, u32x2 q1, u32x2 q2, u32x2 q3,
                                                   u32x4& d0, u32x4& d1) {
    asm volatile(
        "ds_write_b64 %2, %6\n\tds_write_b64 %3, %7\n\tds_write_b64 %4, %8\n\tds_write_b64 %5, %9\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:1024\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(d0), "=&v"(d1)
        : "v"(wa), "v"(wa ^ 32u), "v"(wa ^ 64u), "v"(wa ^ 96u), "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(ra)
        : "memory");
}
__device__ __forceinline__ void stage_w128x2_r128x2(unsigned wa, unsigned ra, u32x4 q0, u32x4 q1, u32x4& d0, u32x4& d1) {
    asm volatile(
        "ds_write_b128 %2, %4\n\tds_write_b128 %3, %5\n\ts_waitcnt lgkmcnt(0)\n\t"
        "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(d0), "=&v"(d1)
        : "v"(wa), "v"(wa ^ 64u), "v"(q0), "v"(q1), "v"(ra)
        : "memory");
}

// Epilogue of one wave's 128 x 64 output block (acc[a][b]: rows 16a + (lane&15), columns 16b + 4*(lane>>4) .. +3).
// Stores are re-shaped through a 2-KiB wave-private LDS stage so that every global store instruction writes whole
// 128-byte lines (16 B per lane, 8 rows x 128 B per instruction): the accumulator layout itself gives 8-byte
// pieces on 16 different rows per instruction, which measured store-ISSUE bound (the epilogue ran no faster when
// it was overlapped with the next tile's MFMAs).  XOR swizzle of the 16-B chunk with (row & 7) keeps the LDS
// writes at most 2-way conflicted.  m0w / n0w: first row / column of the wave's block.
template <int EPI, int RB1>
__device__ __forceinline__ void epilogue_p(const GemmArgs& p, f32x4 (&acc)[8][4], int m0w, int n0w, int prow,
                                           u32x4 rs0, u32x4 rs1, u32x4 rsc, const float* bias_lds, char* stage,
                                           int row_end) {
    // row_end: first row this wave must NOT store (<= p.M; the asymmetric tile's wave row 1 runs the loops of wave row 0
    // -- the store count every counted wait relies on is the same for all waves -- and drops its phantom row block here)
    constexpr int RBX = TileRows<RB1>::rbmax;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, c = lane >> 4;
    const int rrow = lane >> 3, rch = lane & 7;          // read-back: 8 rows x 8 chunks per pass
    float4 bv[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) bv[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!EpiKind<EPI>::dgelu && bias_lds != nullptr) {
        // inline asm on purpose: a compiler-visible ds_read of LDS-DMA-written memory makes hipcc emit
        // s_waitcnt vmcnt(0) (draining the operand prefetch and the previous tile's stores).  The bias DMA is
        // known complete here: it was issued >= 12 vector-memory operations before the last counted wait.
        f32x4 b0, b1, b2, b3;
        const unsigned addr = (unsigned)(unsigned long)((__attribute__((address_space(3))) const float*)bias_lds);
        asm volatile(
            "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
            "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
            : "v"(addr)
            : "memory");
        bv[0] = make_float4(b0[0], b0[1], b0[2], b0[3]);
        bv[1] = make_float4(b1[0], b1[1], b1[2], b1[3]);
        bv[2] = make_float4(b2[0], b2[1], b2[2], b2[3]);
        bv[3] = make_float4(b3[0], b3[1], b3[2], b3[3]);
    }
    // LDS stage addressing (LDS byte addresses): 16 rows x 128 B, 16-B chunk index XOR (row & 7)
    const unsigned sbase = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)stage);
    const unsigned ra = sbase + rrow * 128 + ((rch ^ (rrow & 7)) << 4);     // + 1024 for rows 8..15 (same row & 7)
    const int em = m0w + r, en = n0w + c * 4;

    if (EPI == EPI_BIAS_RESID) {
        // fp32 output: a row block of 16 rows x 32 columns (128 B per row) per stage pass
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // the residual is read in the layout the stage hands back (8 rows x 128 B per instruction, 16 B per lane): whole
            // lines, like the stores, and it is added after the transposition.  rv[a][bp][ps]: row 16a + 8ps + rrow,
            // columns 32bp + 4rch .. +3 of the wave's block.  (Issuing the loads in groups of two row blocks ahead of
            // the previous group's stores measured slower: 122.5 vs 120.5 us for attn.proj.)
            float4 rv[4][2][2];
            float sc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int ms = min(em + (half * 4 + a) * 16, p.M - 1);
                sc[a] = p.rowscale ? p.rowscale[(ms + p.row0) / p.rows_per_sample] : 1.0f;
#pragma unroll
                for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int m = min(m0w + (half * 4 + a) * 16 + ps * 8 + rrow, p.M - 1);
                        rv[a][bp][ps] = load_f4<GETMI_NT_EPI_RESID_LOAD != 0>(p.resid + (size_t)m * p.ldo +
                                                                         min(n0w + bp * 32 + rch * 4, p.N - 4));
                    }
            }
#pragma unroll
            for (int a = 0; a < (half ? RBX : 4); ++a)
#pragma unroll
                for (int bp = 0; bp < 2; ++bp) {
                    u32x4 qv[2], dd[2];
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int b = bp * 2 + bb;
                        const f32x4 v = acc[half * 4 + a][b];
                        const float4 bi = bv[b];
                        const float s_ = sc[a], sv_ = p.prescaled ? 1.0f : s_;
                        const f32x4 o = {sv_ * v[0] + s_ * bi.x, sv_ * v[1] + s_ * bi.y, sv_ * v[2] + s_ * bi.z, sv_ * v[3] + s_ * bi.w};
                        qv[bb] = __builtin_bit_cast(u32x4, o);
                    }
                    // 16-B chunk index of (bb, c) is bb*4 + c
                    stage_w128x2_r128x2(sbase + r * 128 + ((c ^ (r & 7)) << 4), ra, qv[0], qv[1], dd[0], dd[1]);
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int row = ps * 8 + rrow;
                        const int m = m0w + (half * 4 + a) * 16 + row, n = n0w + bp * 32 + rch * 4;
                        const unsigned off = (m < row_end && n < p.N) ? (unsigned)(((size_t)m * p.ldo + n) * 4) : OOB_OFF;
                        const f32x4 t = __builtin_bit_cast(f32x4, dd[ps]);
                        const float4 q = rv[a][bp][ps];
                        const f32x4 o = {q.x + t[0], q.y + t[1], q.z + t[2], q.w + t[3]};
                        st_b128<EpiNt<EPI>::value>(__builtin_bit_cast(u32x4, o), off, rs0);
                    }
                }
        }
        return;
    }

    float4 cs[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) cs[b] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint2 uv[4][4];
        if (EpiKind<EPI>::dgelu) {
            if (EPI == EPI_DMUL && p.blocked) {
                // blocked aux: this lane's four 8-byte pieces of row block (half*4 + a) are two 16-byte loads, 1 KiB per wave
                // and instruction (row blocks past the end re-read the last one: their results are never stored)
                const int nrb = (p.M + 15) >> 4, ncb = p.N >> 6;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int rb = min((m0w >> 4) + half * 4 + a, nrb - 1);
                    const bf16_t* blk = p.aux + ((size_t)rb * ncb + (n0w >> 6)) * 1024 + lane * 8;
                    const uint4 lo = load_u4<GETMI_NT_EPI_LOADS != 0>(blk), hi = load_u4<GETMI_NT_EPI_LOADS != 0>(blk + 512);
                    uv[a][0] = make_uint2(lo.x, lo.y); uv[a][1] = make_uint2(lo.z, lo.w);
                    uv[a][2] = make_uint2(hi.x, hi.y); uv[a][3] = make_uint2(hi.z, hi.w);
                }
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        uv[a][b] = load_u2<GETMI_NT_EPI_LOADS != 0>(p.aux + (size_t)min(em + (half * 4 + a) * 16, p.M - 1) * p.ldaux +
                                                                    min(en + b * 16, p.N - 4));
            }
        }
#pragma unroll
        for (int a = 0; a < (half ? RBX : 4); ++a) {
            const int ab = half * 4 + a;
            const bool mok = (em + ab * 16) < row_end;
            float rsc = 1.0f;      // DropPath fold (training forward of fc1): both outputs carry the sample's factor
            if (EPI == EPI_BIAS_GELUG && p.rowscale)
                rsc = p.rowscale[(min(em + ab * 16, p.M - 1) + p.row0) / p.rows_per_sample];
            // bf16 output(s): 16 rows x 64 columns (128 B per row) per stage pass
            u32x2 qv[4], qg[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 v = acc[ab][b];
                float d0, d1, d2, d3;
                if (EpiKind<EPI>::dgelu) {
                    const uint2 u = uv[a][b];
                    d0 = v[0] * dgelu_factor<EPI>(bf2f(u.x & 0xffff)); d1 = v[1] * dgelu_factor<EPI>(bf2f(u.x >> 16));
                    d2 = v[2] * dgelu_factor<EPI>(bf2f(u.y & 0xffff)); d3 = v[3] * dgelu_factor<EPI>(bf2f(u.y >> 16));
                    if (mok && (en + b * 16) < p.N) { cs[b].x += d0; cs[b].y += d1; cs[b].z += d2; cs[b].w += d3; }
                } else {
                    d0 = v[0] + bv[b].x; d1 = v[1] + bv[b].y; d2 = v[2] + bv[b].z; d3 = v[3] + bv[b].w;
                    if (EpiKind<EPI>::gelu) {
                        float g0, g1, g2, g3;
                        gelu_pair<EPI>(d0, d0, g0); gelu_pair<EPI>(d1, d1, g1);
                        gelu_pair<EPI>(d2, d2, g2); gelu_pair<EPI>(d3, d3, g3);
                        if (EPI == EPI_BIAS_GELUG && p.rowscale) {
                            d0 *= rsc; d1 *= rsc; d2 *= rsc; d3 *= rsc; g0 *= rsc; g1 *= rsc; g2 *= rsc; g3 *= rsc;
                        }
                        qg[b] = pack4bf_v(g0, g1, g2, g3);
                    }
                }
                qv[b] = pack4bf_v(d0, d1, d2, d3);
            }
#pragma unroll
            for (int outp = 0; outp < (EpiKind<EPI>::gelu ? 2 : 1); ++outp) {
                if (EPI == EPI_BIAS_GELUG && outp == 0 && p.blocked) {
                    // blocked out0 (the stored derivative, read back only by the DMUL epilogue of the same shape): the
                    // accumulator layout leaves as it is, two whole 1-KiB pieces per wave and row block, no LDS pass.
                    // Same number of store instructions as the staged form (the counted waits rely on it).
                    const unsigned boff = (m0w + ab * 16 < row_end)
                        ? (unsigned)((((size_t)((m0w >> 4) + ab) * (p.N >> 6) + (n0w >> 6)) * 2048) + lane * 16) : OOB_OFF;
                    st_b128<EpiNt<EPI>::value>((u32x4){qv[0][0], qv[0][1], qv[1][0], qv[1][1]}, boff, rs0);
                    st_b128<EpiNt<EPI>::value>((u32x4){qv[2][0], qv[2][1], qv[3][0], qv[3][1]},
                                               boff == OOB_OFF ? OOB_OFF : boff + 1024u, rs0);
                    continue;
                }
                u32x4 dd[2];
                // 8-byte piece of (b, c): columns b*16 + c*4 .. +3 -> 16-B chunk (b*2 + (c>>1)), half (c&1)
                // (the 8-byte half is XORed with row >> 3: a ds_write_b64 is serviced in groups of 16 consecutive lanes = the 16
                //  rows of one c over 32 banks, so (chunk, half) must take 16 different values per group; rows 8..15 then hold
                //  their two halves swapped, which the read-back undoes by renaming registers)
                const unsigned wa64 = sbase + r * 128 + ((((c >> 1) ^ (r & 7))) << 4) + (((c & 1) ^ (r >> 3)) << 3);
                if (outp == 0) stage_w64x4_r128x2(wa64, ra, qv[0], qv[1], qv[2], qv[3], dd[0], dd[1]);
                else stage_w64x4_r128x2(wa64, ra, qg[0], qg[1], qg[2], qg[3], dd[0], dd[1]);
                dd[1] = (u32x4){dd[1][2], dd[1][3], dd[1][0], dd[1][1]};
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int row = ps * 8 + rrow;
                    const int m = m0w + ab * 16 + row, n = n0w + rch * 8;
                    const unsigned off = (m < row_end && n < p.N) ? (unsigned)(((size_t)m * p.ldo + n) * 2) : OOB_OFF;
                    st_b128<EpiNt<EPI>::value>(dd[ps], off, outp == 0 ? rs0 : rs1);
                }
            }
        }
    }
    if (EpiKind<EPI>::dgelu && p.colsum) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            // sum over the 16 lanes (rows) of a DPP row by rotations 8, 4, 2, 1: every lane ends with the total, no LDS
            // traffic (the ds_bpermute form was 64 LDS operations per wave and tile)
            float4 cc = cs[b];
            cc.x = row16_sum(cc.x); cc.y = row16_sum(cc.y); cc.z = row16_sum(cc.z); cc.w = row16_sum(cc.w);
            const int n = en + b * 16;
            const unsigned off = (r == 0 && n < p.N) ? (unsigned)(((size_t)prow * p.N + n) * 4) : OOB_OFF;
            const f32x4 cv = {cc.x, cc.y, cc.z, cc.w};
            st_b128(__builtin_bit_cast(u32x4, cv), off, rsc);
            // the partial-row table has two rows per wave row (the one-tile kernel's epilogue sums its two 64-row halves
            // separately); this epilogue fills the second with zeros so that no memset of the table is needed
            st_b128((u32x4){0u, 0u, 0u, 0u}, off == OOB_OFF ? OOB_OFF : off + (unsigned)p.N * 4u, rsc);
        }
    }
}

// Which output tiles a persistent workgroup walks, and in which order.
// Logical tile order = column-group major: the N-tiles are cut into groups of `group_n` columns, and inside a group the
// order is row-block major (tm, then the group's columns).  XCD x (the workgroups with equal blockIdx % 8, made contiguous
// by xcd_remap) takes the x-th EIGHTH of that order for the whole launch, its workgroups striding through it together.
// So at any time the 32 workgroups of an XCD share a few row blocks of A and at most group_n column blocks of B, and
// those B blocks stay in the XCD's 4 MiB L2 from round to round (the host picks group_n so that they fit): with plain
// row-major rounds dealt across the XCDs every XCD re-fetched ALL of B every round (fc1: 12 blocks = 4.7 MB > L2;
// rocprofv3 FETCH_SIZE 448 MB per launch for 84 MB of operands).  group_n == 0 or a grid that is not a multiple of 8:
// the old round-robin order.
struct TileWalk {
    int first, stride, count, tiles_m, tiles_n, group_n;
    __device__ __forceinline__ void init(int bslot, int G, int ntiles, int tm_, int tn_, int gn_) {
        tiles_m = tm_; tiles_n = tn_;
        if (gn_ > 0 && (G & 7) == 0) {
            group_n = gn_;
            const int q = G >> 3, x = bslot / q, c = bslot - x * q;
            const int lo = (int)((long long)ntiles * x / 8), hi = (int)((long long)ntiles * (x + 1) / 8);
            first = lo + c; stride = q; count = (hi - lo - c + q - 1) / q;      // >= 1: hi - lo >= q because ntiles >= G
        } else {
            group_n = tn_;
            first = bslot; stride = G; count = (ntiles - bslot + G - 1) / G;
        }
    }
    // tile i of this workgroup (i clamped to the last one: look-ahead past the end re-stages the last tile, never used)
    __device__ __forceinline__ void decode(int i, int& tm, int& tn) const {
        const int lid = first + min(i, count - 1) * stride;
        const int per = tiles_m * group_n;
        const int cg = lid / per, rem = lid - cg * per;
        const int w = min(group_n, tiles_n - cg * group_n);
        tm = rem / w;
        tn = cg * group_n + (rem - tm * w);
    }
};

template <int EPI, int RB1>      // RB1: see gemm_tn256_kernel (4 = 256-row tiles, 3 = 224-row tiles)
__global__ __launch_bounds__(512, 2) void gemm_tn256p_kernel(GemmArgs p, int ntiles, int group_n, int skew) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = EpiStores<EPI, RB1>::value;
    typedef TileRows<RB1> TR;
    constexpr int BMv = TR::bm, HW0 = TR::hw0;                   // tile rows, first row of wave row 1
    constexpr int RBMIN = TR::rbmin, RBMAX = TR::rbmax;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool rb_hi = wm == 0;                                  // this wave owns RBMAX blocks in its second half (wave-uniform)
    const int G = gridDim.x;
    const int bslot = xcd_remap(blockIdx.x, G);
    const int tiles_n = (p.N + 255) >> 8;
    const int nk = p.K >> 6;                                    // >= 2 (host-checked)
    TileWalk walk;
    walk.init(bslot, G, ntiles, (p.M + BMv - 1) / BMv, tiles_n, group_n);
    const int ntl = walk.count;                                 // tiles of this workgroup (>= 1: G <= ntiles)
    const int Gk = ntl * nk;                                    // K-tiles of this workgroup's stream
    const int Q = Gk << 2;                                      // half-tiles of the stream

    const unsigned esz = (EPI == EPI_BIAS_RESID) ? 4u : 2u;
    // (a NULL out0 -- allowed for the two-output GELU epilogues -- becomes an empty buffer: its stores are dropped by the
    //  bounds check, the instruction count the waits rely on stays the same)
    // (a blocked out0 -- GemmArgs::blocked -- is padded to whole 16-row blocks)
    const int rows0 = (EPI == EPI_BIAS_GELUG && p.blocked) ? ((p.M + 15) & ~15) : p.M;
    const u32x4 rs0 = make_rsrc(p.out0 ? p.out0 : p.out1, p.out0 ? (unsigned)((size_t)rows0 * p.ldo * esz) : 0u);
    const u32x4 rs1 = make_rsrc(p.out1 ? p.out1 : p.out0, (unsigned)((size_t)p.M * p.ldo * esz));
    const u32x4 rsc = make_rsrc(p.colsum ? (const void*)p.colsum : (const void*)p.out0,
                                p.colsum ? (unsigned)((size_t)((p.M + BMv - 1) / BMv) * 4 * p.N * 4) : 0u);

    // ---- staging offsets (bytes from p.A / p.B) of the current and the next output tile ----
    unsigned cur[4][2], nxt[4][2];
    int dst[4][2];
    auto tile_offsets = [&](int it, unsigned (&o)[4][2]) {
        int tm, tn;
        walk.decode(it, tm, tn);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int hr = (wave * 2 + jj) * 8 + (lane >> 3);
                if (j == 0 || j == 3) {
                    const int row = ((hr >> 6) << 7) + (j == 3) * 64 + (hr & 63);
                    const int g = min(tm * BMv + ((hr >> 6) ? HW0 : 0) + (j == 3) * 64 + (hr & 63), p.M - 1);
                    o[j][jj] = (unsigned)((size_t)g * p.lda * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
                } else {
                    const int row = ((hr >> 5) << 6) + (j - 1) * 32 + (hr & 31);
                    const int g = min((tn << 8) + row, p.N - 1);
                    o[j][jj] = (unsigned)((size_t)g * p.ldb * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
                }
            }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int hr0 = (wave * 2 + jj) * 8;
            if (j == 0 || j == 3) dst[j][jj] = (((hr0 >> 6) << 7) + (j == 3) * 64 + (hr0 & 63)) * 128;
            else dst[j][jj] = 32768 + (((hr0 >> 5) << 6) + (j - 1) * 32 + (hr0 & 31)) * 128;
        }
    tile_offsets(0, cur);
    tile_offsets(1, nxt);
    int tile_g0 = 0;            // stream index of the current tile's first K-tile
    int it = 0;                 // tile iteration of this workgroup

    const char* baseA = reinterpret_cast<const char*>(p.A);
    const char* baseB = reinterpret_cast<const char*>(p.B);
    // issue half-tile J of stream K-tile gi (gi is in the current or in the next output tile)
#define GP_ISSUE(gi, J)                                                                                    \
    do {                                                                                                   \
        const int gi_ = (gi);                                                                              \
        const bool nx_ = gi_ >= tile_g0 + nk;                                                              \
        const unsigned koff_ = (unsigned)(gi_ - tile_g0 - (nx_ ? nk : 0)) * 128u;                          \
        char* b_ = smem + ((gi_ & 1) << 16);                                                               \
        const char* base_ = ((J) == 0 || (J) == 3) ? baseA : baseB;                                        \
        glds16(base_ + (size_t)((nx_ ? nxt[J][0] : cur[J][0]) + koff_), b_ + dst[J][0]);                    \
        glds16(base_ + (size_t)((nx_ ? nxt[J][1] : cur[J][1]) + koff_), b_ + dst[J][1]);                    \
    } while (0)

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fsw = (frow >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * 128 + (((4 * ks + (lane >> 4)) ^ fsw) << 4);
    const int a_base = (wm * 128) * 128;
    const int b_base = 32768 + (wn * 64) * 128;

    // per-wave LDS copy of the current tile's bias row segment [n0, n0+256): one 1-KiB LDS-DMA per wave and tile
    const bool has_bias = !EpiKind<EPI>::dgelu && p.bias != nullptr;    // workgroup-uniform
    char* bias_img = smem + G2_LDS;                      // 1 KiB shared image: every wave DMA-writes the same values
    char* stage = smem + G2_LDS + 1024 + wave * 2048;    // 2 KiB wave-private store stage
    auto stage_bias = [&](int iter) {
        int tm, tn;
        walk.decode(iter, tm, tn);
        glds16(p.bias + min((tn << 8) + lane * 4, p.N - 4), bias_img);
    };
    if (has_bias) stage_bias(0);
    GP_ISSUE(0, 0); GP_ISSUE(0, 1); GP_ISSUE(0, 2); GP_ISSUE(0, 3);
    GP_ISSUE(1, 0); GP_ISSUE(1, 1); GP_ISSUE(1, 2);              // Gk >= 2
    // De-synchronise the CUs (see launch_gemm_tn256p): workgroup b starts up to `skew` x 2048 cycles late, spread evenly
    // over the grid by a bit-reversal-like stride so that neighbours in an XCD differ.
    if (skew > 0) {
        const int n = (int)(((unsigned)(bslot * 97) % (unsigned)G) * (unsigned)skew / (unsigned)G);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
    }

    // Fragment reads are inline asm with COUNTED lgkmcnt waits: hipcc clusters the ds_reads of a phase and puts one
    // `s_waitcnt lgkmcnt(0)` in front of the first MFMA, which serialises the phase (all eight waves read LDS, then
    // all compute: measured 55 % MFMA-busy in the loop = 2048 MFMA cycles / (2048 + 1536 LDS cycles) per K-tile).
    // Reads are issued k-step-major (B pair, then the four A row-blocks), and every pair of MFMAs waits only for
    // the fragments it consumes, so the MFMAs start after 3 of 12 reads and run under the rest.
    // (form (ii) of cdna_hip_programming.md 5.7: a read's destination is undefined until the wait that names it.)
    u32x4 fa[4][2], fb0[2][2], fb1[2][2];
    const unsigned lds0 = lds_addr(smem);
    unsigned la[2], lb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { la[ks] = lds0 + a_base + foff[ks]; lb[ks] = lds0 + b_base + foff[ks]; }
#define GP_RD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define GP_LG1(N, x) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x) : : "memory")
#define GP_LG2(N, x, y) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x), "+v"(y) : : "memory")
#define GP_LG3(N, x, y, z) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x), "+v"(y), "+v"(z) : : "memory")
#define GP_RD_A(MH, ks, addr)                                                                              \
    GP_RD(fa[0][ks], addr, (MH)*8192); GP_RD(fa[1][ks], addr, (MH)*8192 + 2048);                           \
    GP_RD(fa[2][ks], addr, (MH)*8192 + 4096); GP_RD(fa[3][ks], addr, (MH)*8192 + 6144);
#define GP_RD_B(FB, NH, ks, addr) GP_RD(FB[0][ks], addr, (NH)*4096); GP_RD(FB[1][ks], addr, (NH)*4096 + 2048);
    // the two MFMAs of row-block mt, k-step ks of quadrant (MH, NH)
#define GP_MM2(MH, NH, FB, mt, ks)                                                                         \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                   \
            acc[(MH)*4 + mt][(NH)*2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                       \
                __builtin_bit_cast(bf16x8, FB[nt][ks]), __builtin_bit_cast(bf16x8, fa[mt][ks]),           \
                acc[(MH)*4 + mt][(NH)*2 + nt], 0, 0, 0);
#define GP_MM(MH, NH, FB, mt, ks)                                                                          \
    if constexpr ((MH) == 0 || (mt) < RBMIN) {                                                             \
        GP_MM2(MH, NH, FB, mt, ks)                                                                         \
    } else if constexpr ((mt) < RBMAX) {      /* asymmetric tile: the row block only wave row 0 owns */    \
        if (rb_hi) { GP_MM2(MH, NH, FB, mt, ks) }                                                          \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);   /* keep the pair between its wait and the next one */
#define GP_MM4(MH, NH, FB, ks) GP_MM(MH, NH, FB, 0, ks) GP_MM(MH, NH, FB, 1, ks) GP_MM(MH, NH, FB, 2, ks) GP_MM(MH, NH, FB, 3, ks)
    // counted wait of a reading phase: `win` > 0 while the needed half-tile predates the last epilogue
#define GP_WAIT(inflight)                                                                                  \
    do {                                                                                                   \
        if (win > 0) { wait_halftiles_x<E>(inflight); --win; }                                             \
        else wait_halftiles_x<0>(inflight);                                                                \
    } while (0)

    // issue half-tile J of stream K-tile gi when gi is known to lie in the CURRENT output tile (no select, no compare)
#define GP_ISSUE_CUR(gi, J)                                                                                \
    do {                                                                                                   \
        const int gi_ = (gi);                                                                              \
        const unsigned koff_ = (unsigned)(gi_ - tile_g0) * 128u;                                           \
        char* b_ = smem + ((gi_ & 1) << 16);                                                               \
        const char* base_ = ((J) == 0 || (J) == 3) ? baseA : baseB;                                        \
        glds16(base_ + (size_t)(cur[J][0] + koff_), b_ + dst[J][0]);                                       \
        glds16(base_ + (size_t)(cur[J][1] + koff_), b_ + dst[J][1]);                                       \
    } while (0)

    int win = 0;
    // One K-tile of the stream.  FAST (compile time): a K-tile in the middle of an output tile -- everything it issues
    // (K-tiles g+1, g+2) lies in the same output tile, no store of an epilogue is still counted (win == 0) and the
    // stream does not end within two K-tiles -- so its waits are the constant vmcnt(10) and its issues are
    // unconditional: the phase heads carry no compare / select / branch ladders (the wgrad loop gained 6 % from this).
    auto ktile = [&](int g, auto fast_) {
        constexpr bool FAST = decltype(fast_)::value;
        const unsigned bo = (unsigned)(g & 1) << 16;
        const unsigned a0 = la[0] + bo, a1 = la[1] + bo, b0 = lb[0] + bo, b1 = lb[1] + bo;
        const int phi = g << 2;
        // ---- P1: A-half0 x B-half0 (12 reads) ----
        if constexpr (FAST) wait_halftiles_x<0>(5); else GP_WAIT(min(phi + 6, Q - 1) - (phi + 1));
        phase_barrier();
        if constexpr (FAST) GP_ISSUE_CUR(g + 1, 3); else if (g + 1 < Gk) GP_ISSUE(g + 1, 3);
        GP_RD_B(fb0, 0, 0, b0) GP_RD_A(0, 0, a0) GP_RD_B(fb0, 0, 1, b1) GP_RD_A(0, 1, a1)
        GP_LG3(9, fb0[0][0], fb0[1][0], fa[0][0]); GP_MM(0, 0, fb0, 0, 0)
        GP_LG1(8, fa[1][0]); GP_MM(0, 0, fb0, 1, 0)
        GP_LG1(7, fa[2][0]); GP_MM(0, 0, fb0, 2, 0)
        GP_LG1(6, fa[3][0]); GP_MM(0, 0, fb0, 3, 0)
        GP_LG3(3, fb0[0][1], fb0[1][1], fa[0][1]); GP_MM(0, 0, fb0, 0, 1)
        GP_LG1(2, fa[1][1]); GP_MM(0, 0, fb0, 1, 1)
        GP_LG1(1, fa[2][1]); GP_MM(0, 0, fb0, 2, 1)
        GP_LG1(0, fa[3][1]); GP_MM(0, 0, fb0, 3, 1)
        // ---- P2: A-half0 (in registers) x B-half1 (4 reads) ----
        if constexpr (FAST) wait_halftiles_x<0>(5); else GP_WAIT(min(phi + 7, Q - 1) - (phi + 2));
        phase_barrier();
        if constexpr (FAST) GP_ISSUE_CUR(g + 2, 0); else if (g + 2 < Gk) GP_ISSUE(g + 2, 0);
        GP_RD_B(fb1, 1, 0, b0) GP_RD_B(fb1, 1, 1, b1)
        GP_LG2(2, fb1[0][0], fb1[1][0]); GP_MM4(0, 1, fb1, 0)
        GP_LG2(0, fb1[0][1], fb1[1][1]); GP_MM4(0, 1, fb1, 1)
        // ---- P3: A-half1 (8 reads) x B-half1 (in registers) ----
        if constexpr (FAST) wait_halftiles_x<0>(5); else GP_WAIT(min(phi + 8, Q - 1) - (phi + 3));
        phase_barrier();
        if constexpr (FAST) GP_ISSUE_CUR(g + 2, 1); else if (g + 2 < Gk) GP_ISSUE(g + 2, 1);
        GP_RD_A(1, 0, a0) GP_RD_A(1, 1, a1)
        GP_LG1(7, fa[0][0]); GP_MM(1, 1, fb1, 0, 0)
        GP_LG1(6, fa[1][0]); GP_MM(1, 1, fb1, 1, 0)
        GP_LG1(5, fa[2][0]); GP_MM(1, 1, fb1, 2, 0)
        GP_LG1(4, fa[3][0]); GP_MM(1, 1, fb1, 3, 0)
        GP_LG1(3, fa[0][1]); GP_MM(1, 1, fb1, 0, 1)
        GP_LG1(2, fa[1][1]); GP_MM(1, 1, fb1, 1, 1)
        GP_LG1(1, fa[2][1]); GP_MM(1, 1, fb1, 2, 1)
        GP_LG1(0, fa[3][1]); GP_MM(1, 1, fb1, 3, 1)
        // ---- P4: A-half1 x B-half0, both in registers (no barrier) ----
        if constexpr (FAST) GP_ISSUE_CUR(g + 2, 2); else if (g + 2 < Gk) GP_ISSUE(g + 2, 2);
        GP_MM4(1, 0, fb0, 0) GP_MM4(1, 0, fb0, 1)
    };

    for (int t = 0; t < ntl; ++t) {
        // K-tiles of this output tile: the first two still count the previous epilogue's stores in their waits
        // (win), the last two issue into the next output tile (or past the end of the stream): generic code
        int k = 0;
        const int kslow = (t == 0) ? 0 : 2;
        for (; k < kslow && k < nk; ++k) ktile(tile_g0 + k, std::false_type{});
        for (; k + 2 < nk; ++k) ktile(tile_g0 + k, std::true_type{});
        for (; k < nk; ++k) ktile(tile_g0 + k, std::false_type{});
        {
            // ---- tile finished: epilogue (its stores drain under the next tile), then switch tiles ----
            int tm, tn;
            walk.decode(it, tm, tn);
            const float* bl = has_bias ? reinterpret_cast<const float*>(bias_img) + wn * 64 + (lane >> 4) * 4 : nullptr;
            const int m0w = tm * BMv + (wm ? HW0 : 0);
            epilogue_p<EPI, RB1>(p, acc, m0w, (tn << 8) + wn * 64, tm * 4 + wm * 2, rs0, rs1, rsc, bl, stage,
                                 min(p.M, m0w + (wm ? TR::hw1 : TR::hw0)));
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            ++it;
            tile_g0 += nk;
#pragma unroll
            for (int j = 0; j < 4; ++j) { cur[j][0] = nxt[j][0]; cur[j][1] = nxt[j][1]; }
            tile_offsets(it + 1, nxt);
            if (has_bias && it < ntl) stage_bias(it);     // after the epilogue's LDS reads (same wave, in order)
            win = 5;
        }
    }
#undef GP_ISSUE_CUR
#undef GP_ISSUE
#undef GP_RD
#undef GP_LG1
#undef GP_LG2
#undef GP_LG3
#undef GP_RD_A
#undef GP_RD_B
#undef GP_MM
#undef GP_MM2
#undef GP_MM4
#undef GP_WAIT
}

// ------------------------------------------------------------------------------------------------
// gemm_tn, second generation ("two workgroups per CU"): 256 x 128 output tile per workgroup of FOUR waves (2 M x 2 N,
// 128 x 64 per wave -- the same per-wave block, fragment reads and MFMA order as the 256 x 256 kernels), persistent,
// 73 KiB of LDS and <= 256 VGPRs, so that TWO workgroups are resident on every CU and run unsynchronised: while one is
// in its epilogue (GELU arithmetic, the stored-derivative or fp32-residual reads: phases a single resident workgroup
// WAITS for) or at a barrier, the other one's MFMAs have the matrix pipes to themselves.  The
// price is 1.5 x the operand traffic L2 -> LDS per FLOP of the 256 x 256 tile; the gain is measured per shape
// (tools/bench_gemm.py) and the launcher keeps whichever generation is faster there.
//
// Operands per K-tile (64 deep): three 16-KiB PIECES of 128 rows x 128 B -- A0 (rows 0-63 of both wave rows), B (all
// 128 columns), A1 (rows 64-127 of both wave rows) -- streamed in that order through a ring of FOUR slots (piece q =
// 3 g + j of stream K-tile g lives in slot q & 3), one piece issued per barrier:
//     P1  wait A0(g), B(g)  | barrier | issue A0(g+1) -> slot of A1(g-1) | read A0, B-low  | 16 MFMA  A0 x B-low
//     P2                    | barrier | issue B(g+1)  -> slot of A0(g)   | read B-high     | 16 MFMA  A0 x B-high
//     P3  wait A1(g)        | barrier | issue A1(g+1) -> slot of B(g)    | read A1         | 16 MFMA  A1 x B-high
//     P4                                                                                   | 16 MFMA  A1 x B-low
// RAW: every wave waits for its own share of a piece (counted vmcnt: 4 LDS-DMA per piece and wave; pieces younger than
// the needed one stay in flight: A1(g) at P1, A0(g+1) + B(g+1) at P3) before the barrier in front of the reads.
// WAR: a slot is re-issued only after a barrier that every wave reaches after its last fragment read of the old piece
// has been consumed by MFMAs (A1(g-1): read in P3(g-1); A0(g): read in P1(g); B(g): last read in P2(g)).
// The epilogue's stores (E per wave, exact: inline asm) and the next tile's bias DMA are issued between A1(n0) and
// A0(n1) of the next output tile's first K-tiles, so that tile's two waits count them in: vmcnt(4 + EB), vmcnt(8 + EB).
// ------------------------------------------------------------------------------------------------
#define G3_RING 65536
#define G3_LDS (G3_RING + 2048 + 4 * 2048)

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tn2_kernel(GemmArgs p, int ntiles, int group_n, int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = EpiStores<EPI, 4>::value;
    constexpr int EB = E + (EpiKind<EPI>::dgelu ? 0 : 1);       // + the bias row DMA of the next tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x;
    const int bslot = xcd_remap(blockIdx.x, G);
    const int tiles_n = (p.N + 127) >> 7;
    const int nk = p.K >> 6;                                    // >= 2 (host-checked)
    TileWalk walk;
    walk.init(bslot, G, ntiles, (p.M + 255) >> 8, tiles_n, group_n);
    const int ntl = walk.count;                                 // tiles of this workgroup (>= 1: G <= ntiles)

    const unsigned esz = (EPI == EPI_BIAS_RESID) ? 4u : 2u;
    const u32x4 rs0 = make_rsrc(p.out0 ? p.out0 : p.out1, p.out0 ? (unsigned)((size_t)p.M * p.ldo * esz) : 0u);
    const u32x4 rs1 = make_rsrc(p.out1 ? p.out1 : p.out0, (unsigned)((size_t)p.M * p.ldo * esz));
    const u32x4 rsc = make_rsrc(p.colsum ? (const void*)p.colsum : (const void*)p.out0,
                                p.colsum ? (unsigned)((size_t)((p.M + 255) >> 8) * 4 * p.N * 4) : 0u);

    // ---- staging: LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds).  The per-lane byte offsets
    //      are the SAME for every output tile (row of the piece image x leading dimension + swizzled 16-B chunk); the
    //      tile enters through the descriptor (base = first row of the tile, num_records = what is left of the operand
    //      from there, so rows past M / N are dropped by the range check: no clamping arithmetic, no per-tile VGPRs) and
    //      the K-tile through the scalar offset.  Piece kind 0 = A0, 1 = B, 2 = A1; 4 wave-instructions (8 rows) each. ----
    unsigned vA0[4], vA1[4], vB[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int hr = (wave * 4 + jj) * 8 + (lane >> 3);                  // row of the piece image, 0..127
        const unsigned sw = (unsigned)(((lane & 7) ^ ((hr >> 1) & 7)) << 4);
        vA0[jj] = (unsigned)((((hr >> 6) << 7) + (hr & 63)) * p.lda * 2) + sw;
        vA1[jj] = vA0[jj] + (unsigned)(64 * p.lda * 2);
        vB[jj] = (unsigned)(hr * p.ldb * 2) + sw;
    }
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    auto tile_rsrc = [&](int iter, rsrc_t& ra, rsrc_t& rb) {
        int tm, tn;
        walk.decode(iter, tm, tn);
        const size_t rowsA = (size_t)(p.M - (tm << 8)), rowsB = (size_t)(p.N - (tn << 7));
        ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A) + (size_t)(tm << 8) * p.lda, 0,
                                               (unsigned)(rowsA * p.lda * 2), 0x00020000);
        rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B) + (size_t)(tn << 7) * p.ldb, 0,
                                               (unsigned)(rowsB * p.ldb * 2), 0x00020000);
    };
    rsrc_t curA, curB, nxtA, nxtB;
    tile_rsrc(0, curA, curB);
    tile_rsrc(1, nxtA, nxtB);
    int k_in_tile = 0;          // K-tile index inside the current output tile
    int it = 0;                 // output tile iteration of this workgroup

    // issue piece J of stream K-tile g_: K-tile kt_ of the tile whose descriptors are (RA, RB)
#define G3_ISSUE(RA, RB, J, g_, kt_)                                                                       \
    do {                                                                                                   \
        char* d_ = smem + (((3 * (g_) + (J)) & 3) << 14) + (wave << 12);                                   \
        const int koff_ = (kt_) * 128;                                                                     \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds((J) == 1 ? RB : RA, LDS_PTR(d_ + jj * 1024), 16,       \
                                                     (J) == 0 ? vA0[jj] : (J) == 1 ? vB[jj] : vA1[jj], koff_, 0, 0); \
    } while (0)

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fsw = (frow >> 1) & 7;
    const unsigned lds0 = lds_addr(smem);
    unsigned la[2], lb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned fo = (unsigned)(frow * 128 + (((4 * ks + (lane >> 4)) ^ fsw) << 4));
        la[ks] = lds0 + (unsigned)(wm * 64 * 128) + fo;
        lb[ks] = lds0 + (unsigned)(wn * 64 * 128) + fo;
    }

    const bool has_bias = !EpiKind<EPI>::dgelu && p.bias != nullptr;    // workgroup-uniform
    // bias row segment of the tile: two 1-KiB images by tile parity (every wave DMA-writes the same values; a wave that
    // is a whole epilogue ahead must not overwrite the image its neighbours still read)
    char* stage = smem + G3_RING + 2048 + wave * 2048;   // 2 KiB wave-private store stage
    auto stage_bias = [&](int iter) {
        char* bias_img = smem + G3_RING + ((iter & 1) << 10);
        // (issued unconditionally for the epilogues that may carry a bias, so that the counted waits are compile-time
        //  constants; without a bias the image is filled from the B operand and never read)
        int tm, tn;
        walk.decode(iter, tm, tn);
        const float* src = has_bias ? p.bias + min((tn << 7) + lane * 4, p.N - 4)
                                    : reinterpret_cast<const float*>(p.B) + lane * 4;
        glds16(src, bias_img);
    };
    if constexpr (!EpiKind<EPI>::dgelu) stage_bias(0);
    G3_ISSUE(curA, curB, 0, 0, 0); G3_ISSUE(curA, curB, 1, 0, 0); G3_ISSUE(curA, curB, 2, 0, 0);
    // The two workgroups of a CU start together and do identical work, so left alone they reach their epilogues at the
    // same time and nothing overlaps.  The one whose LDS allocation does not start at 0 (HW_REG_LDS_ALLOC.LDS_BASE: the
    // second resident workgroup of its CU) starts about half an output tile late, once; the offset then persists because
    // every tile takes both the same time.  A performance hint only: nothing depends on which workgroup waits.
    if (stagger > 0) {
        const unsigned lds_base = __builtin_amdgcn_s_getreg((7 << 11) | (0 << 6) | 6);
        if (lds_base != 0)
            for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    u32x4 fa[4][2], fb0[2][2], fb1[2][2];
#define G3_RD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define G3_LG1(N, x) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x) : : "memory")
#define G3_LG2(N, x, y) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x), "+v"(y) : : "memory")
#define G3_LG3(N, x, y, z) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x), "+v"(y), "+v"(z) : : "memory")
#define G3_RD_A(ks, addr)                                                                                  \
    G3_RD(fa[0][ks], addr, 0); G3_RD(fa[1][ks], addr, 2048); G3_RD(fa[2][ks], addr, 4096); G3_RD(fa[3][ks], addr, 6144);
#define G3_RD_B(FB, NH, ks, addr) G3_RD(FB[0][ks], addr, (NH)*4096); G3_RD(FB[1][ks], addr, (NH)*4096 + 2048);
#define G3_MM(MH, NH, FB, mt, ks)                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                       \
        acc[(MH)*4 + mt][(NH)*2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                           \
            __builtin_bit_cast(bf16x8, FB[nt][ks]), __builtin_bit_cast(bf16x8, fa[mt][ks]),               \
            acc[(MH)*4 + mt][(NH)*2 + nt], 0, 0, 0);                                                       \
    __builtin_amdgcn_sched_barrier(0);
#define G3_MM4(MH, NH, FB, ks) G3_MM(MH, NH, FB, 0, ks) G3_MM(MH, NH, FB, 1, ks) G3_MM(MH, NH, FB, 2, ks) G3_MM(MH, NH, FB, 3, ks)
#define G3_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")

    // One K-tile.  MODE 0: first K-tile of an output tile after the first (the previous epilogue's stores + bias DMA are
    // still counted); MODE 1: middle (everything it issues lies in the current tile); MODE 2: last K-tile of an output
    // tile (issues the next tile's first K-tile, or nothing at the end of the stream).  MODE 3: first K-tile of the FIRST
    // output tile (no stores outstanding) = MODE 1 waits.
    auto ktile = [&](int g, auto mode_) {
        constexpr int MODE = decltype(mode_)::value;
        const unsigned q0 = (unsigned)(3 * g);
        const unsigned sA0 = (q0 & 3u) << 14, sB = ((q0 + 1u) & 3u) << 14, sA1 = ((q0 + 2u) & 3u) << 14;
        const bool more = (MODE != 2) || (it + 1 < ntl);      // is there a K-tile g + 1 ?
        // ---- P1 ----
        if constexpr (MODE == 0) G3_VM(4 + EB); else G3_VM(4);
        phase_barrier();
        if constexpr (MODE == 2) { if (more) G3_ISSUE(nxtA, nxtB, 0, g + 1, 0); }
        else G3_ISSUE(curA, curB, 0, g + 1, k_in_tile + 1);
        {
            const unsigned a0 = la[0] + sA0, a1 = la[1] + sA0, b0 = lb[0] + sB, b1 = lb[1] + sB;
            G3_RD_B(fb0, 0, 0, b0) G3_RD_A(0, a0) G3_RD_B(fb0, 0, 1, b1) G3_RD_A(1, a1)
            G3_LG3(9, fb0[0][0], fb0[1][0], fa[0][0]); G3_MM(0, 0, fb0, 0, 0)
            G3_LG1(8, fa[1][0]); G3_MM(0, 0, fb0, 1, 0)
            G3_LG1(7, fa[2][0]); G3_MM(0, 0, fb0, 2, 0)
            G3_LG1(6, fa[3][0]); G3_MM(0, 0, fb0, 3, 0)
            G3_LG3(3, fb0[0][1], fb0[1][1], fa[0][1]); G3_MM(0, 0, fb0, 0, 1)
            G3_LG1(2, fa[1][1]); G3_MM(0, 0, fb0, 1, 1)
            G3_LG1(1, fa[2][1]); G3_MM(0, 0, fb0, 2, 1)
            G3_LG1(0, fa[3][1]); G3_MM(0, 0, fb0, 3, 1)
        }
        // ---- P2 (barrier for the slot of A0(g) only) ----
#ifndef G3_TWO_BARRIERS
        phase_barrier();
        if constexpr (MODE == 2) { if (more) G3_ISSUE(nxtA, nxtB, 1, g + 1, 0); }
        else G3_ISSUE(curA, curB, 1, g + 1, k_in_tile + 1);
#endif
        {
            const unsigned b0 = lb[0] + sB, b1 = lb[1] + sB;
            G3_RD_B(fb1, 1, 0, b0) G3_RD_B(fb1, 1, 1, b1)
            G3_LG2(2, fb1[0][0], fb1[1][0]); G3_MM4(0, 1, fb1, 0)
            G3_LG2(0, fb1[0][1], fb1[1][1]); G3_MM4(0, 1, fb1, 1)
        }
        // ---- P3 ----
#ifdef G3_TWO_BARRIERS
        // (B(g+1) is issued here, behind the barrier: when P3 waits, only A0(g+1) is younger than A1(g))
        if constexpr (MODE == 0) G3_VM(4 + EB);
        else if constexpr (MODE == 2) { if (more) G3_VM(4); else G3_VM(0); }
        else G3_VM(4);
        phase_barrier();
        if constexpr (MODE == 2) { if (more) G3_ISSUE(nxtA, nxtB, 1, g + 1, 0); }
        else G3_ISSUE(curA, curB, 1, g + 1, k_in_tile + 1);
#else
        if constexpr (MODE == 0) G3_VM(8 + EB);
        else if constexpr (MODE == 2) { if (more) G3_VM(8); else G3_VM(0); }
        else G3_VM(8);
        phase_barrier();
#endif
        if constexpr (MODE == 2) { if (more) G3_ISSUE(nxtA, nxtB, 2, g + 1, 0); }
        else G3_ISSUE(curA, curB, 2, g + 1, k_in_tile + 1);
        {
            const unsigned a0 = la[0] + sA1, a1 = la[1] + sA1;
            G3_RD_A(0, a0) G3_RD_A(1, a1)
            G3_LG1(7, fa[0][0]); G3_MM(1, 1, fb1, 0, 0)
            G3_LG1(6, fa[1][0]); G3_MM(1, 1, fb1, 1, 0)
            G3_LG1(5, fa[2][0]); G3_MM(1, 1, fb1, 2, 0)
            G3_LG1(4, fa[3][0]); G3_MM(1, 1, fb1, 3, 0)
            G3_LG1(3, fa[0][1]); G3_MM(1, 1, fb1, 0, 1)
            G3_LG1(2, fa[1][1]); G3_MM(1, 1, fb1, 1, 1)
            G3_LG1(1, fa[2][1]); G3_MM(1, 1, fb1, 2, 1)
            G3_LG1(0, fa[3][1]); G3_MM(1, 1, fb1, 3, 1)
        }
        // ---- P4 ----
        G3_MM4(1, 0, fb0, 0) G3_MM4(1, 0, fb0, 1)
    };
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;

    int g = 0;
    for (int t = 0; t < ntl; ++t) {
        k_in_tile = 0;
        if (t == 0) ktile(g, M1{}); else ktile(g, M0{});
        ++g;
        for (k_in_tile = 1; k_in_tile + 1 < nk; ++k_in_tile, ++g) ktile(g, M1{});
        ktile(g, M2{});
        ++g;
        {
            int tm, tn;
            walk.decode(it, tm, tn);
            const float* bl = has_bias ? reinterpret_cast<const float*>(smem + G3_RING + ((it & 1) << 10)) + wn * 64 + (lane >> 4) * 4
                                       : nullptr;
            epilogue_p<EPI, 4>(p, acc, (tm << 8) + wm * 128, (tn << 7) + wn * 64, tm * 4 + wm * 2, rs0, rs1, rsc, bl, stage, p.M);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            ++it;
            curA = nxtA; curB = nxtB;
            tile_rsrc(it + 1, nxtA, nxtB);
            if constexpr (!EpiKind<EPI>::dgelu) { if (it < ntl) stage_bias(it); }
        }
    }
#undef G3_ISSUE
#undef G3_RD
#undef G3_LG1
#undef G3_LG2
#undef G3_LG3
#undef G3_RD_A
#undef G3_RD_B
#undef G3_MM
#undef G3_MM4
#undef G3_VM
}

// ------------------------------------------------------------------------------------------------
// C[N1,N2] (+)= X[M,N1]^T * Y[M,N2]   (wgrad), split over M: block (tile, s) handles K-steps
// [s*steps_per_split, ...) and writes the fp32 slab s.
// LDS image of one operand tile: [64 rows(m)][16 chunks of 16 B] (256-B rows); chunk index XORed with
// 2*((m&3) | ((m>>3)&1)<<2) so that the 32-lane halves of ds_read_b64_tr_b16 (8 rows x 2 chunks) are
// conflict-free.  Rows >= M are sourced from a zero line.
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) const uint32_t g_zero_line[4] = {0u, 0u, 0u, 0u};

struct WgradArgs {
    const bf16_t* X;   // [M, ldx]  (dY)
    const bf16_t* Y;   // [M, ldy]  (layer input)
    int M, N1, N2, ldx, ldy;
    float* slabs;      // [nsplit][N1][N2]
    int steps_per_split;
    // optional second problem over the SAME rows (gemm_wgrad256_kernel only): blocks >= blocks_a work on it.  Two
    // weight gradients that are each too small to give every CU a long K range (proj: 9 tiles) share one launch.
    const bf16_t* Xb;
    const bf16_t* Yb;
    int N1b, N2b, ldxb, ldyb;
    float* slabsb;
    int blocks_a;      // 0: single problem
};

__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 join_tr(s16x4 lo, s16x4 hi) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// wgrad tile swizzle: XOR of the 16-B chunk index of row m (256-B rows)
__device__ __forceinline__ int wg_swz(int m) { return 2 * ((m & 3) | (((m >> 3) & 1) << 2)); }
// byte offset inside a 256-B row of element column `col` (multiple of 4) of row m
__device__ __forceinline__ int tr_col_off(int col, int m) {
    return (((col >> 3) ^ wg_swz(m)) << 4) | ((col & 4) << 1);
}

__global__ __launch_bounds__(256, 2) void gemm_wgrad_kernel(WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int tiles1 = (p.N1 + 127) / 128, tiles2 = (p.N2 + 127) / 128;
    const int ntile = tiles1 * tiles2;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid / ntile;
    const int tile = lid - split * ntile;
    const int t1 = tile / tiles2, t2 = tile - t1 * tiles2;
    const int n1_0 = t1 * 128, n2_0 = t2 * 128;

    const int nk_total = (p.M + BK - 1) / BK;
    const int k_begin = split * p.steps_per_split;
    const int k_end = min(nk_total, k_begin + p.steps_per_split);

    // staging: one wave-instruction = 1 KiB = 4 rows x 256 B; 16 per operand tile, 4 per wave.
    // lane -> row (lane>>4), physical chunk (lane&15); source chunk = phys ^ f(row).
    int srow[4], schunk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 4 + (lane >> 4);
        srow[j] = row;
        schunk[j] = (lane & 15) ^ wg_swz(row);
    }
    // clamp columns (N1/N2 tails: extra columns are computed but never stored)
    auto stage = [&](int buf, int kt) {
        char* x = smem + buf * (2 * TILE_BYTES);
        char* y = x + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = kt * BK + srow[j];
            const int c1 = min(n1_0 + schunk[j] * 8, p.N1 - 8);
            const int c2 = min(n2_0 + schunk[j] * 8, p.N2 - 8);
            const void* sx = (m < p.M) ? (const void*)(p.X + (size_t)m * p.ldx + c1) : (const void*)g_zero_line;
            const void* sy = (m < p.M) ? (const void*)(p.Y + (size_t)m * p.ldy + c2) : (const void*)g_zero_line;
            glds16(sx, x + (wave * 4 + j) * 1024);
            glds16(sy, y + (wave * 4 + j) * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposed fragment read: 16-lane group g = lane>>4 covers m = 8g..8g+7 (two 4-row blocks) of the
    // 32-deep K-step; lane (4q+p) supplies &tile[m0+q][col0 + 4p].
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int toff[2][2];   // [ks][half] byte offset of this lane's address for column-chunk 0 (col0 = 0)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = ks * 32 + 8 * g + 4 * h + q;
            toff[ks][h] = m * 256;   // chunk part added per tile below (needs XOR with f(m))
        }

    if (k_begin < k_end) {
        stage(0, k_begin);
        __syncthreads();
        for (int kt = k_begin; kt < k_end; ++kt) {
            const int cur = (kt - k_begin) & 1;
            if (kt + 1 < k_end) stage(cur ^ 1, kt + 1);
            const char* x = smem + cur * (2 * TILE_BYTES);
            const char* y = x + TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fx[4], fy[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // X columns (n1): wr*64 + i*16 ; Y columns (n2): wc*64 + i*16
                    const int m_lo = ks * 32 + 8 * g + q, m_hi = m_lo + 4;
                    const int colx = wr * 64 + i * 16 + 4 * pp;   // element column inside the X tile
                    const int coly = wc * 64 + i * 16 + 4 * pp;
                    fx[i] = join_tr(lds_tr16(x + toff[ks][0] + tr_col_off(colx, m_lo)),
                                    lds_tr16(x + toff[ks][1] + tr_col_off(colx, m_hi)));
                    fy[i] = join_tr(lds_tr16(y + toff[ks][0] + tr_col_off(coly, m_lo)),
                                    lds_tr16(y + toff[ks][1] + tr_col_off(coly, m_hi)));
                }
                // first operand = Y fragment (i -> n2), second = X fragment (j -> n1):
                // D[i][j] = sum_m Y[m][i] X[m][j]; lane owns n1 = lane&15, n2 = (lane>>4)*4 + reg.
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[b], fx[a], acc[a][b], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    float* out = p.slabs + (size_t)split * p.N1 * p.N2;
    const int e1 = n1_0 + wr * 64 + (lane & 15);
    const int e2 = n2_0 + wc * 64 + (lane >> 4) * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = e1 + a * 16, c = e2 + b * 16;
            if (r < p.N1 && c < p.N2)
                *reinterpret_cast<float4*>(out + (size_t)r * p.N2 + c) =
                    make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
        }
}

// ------------------------------------------------------------------------------------------------
// wgrad, large-tile pipeline: 256(n1) x 256(n2) output tile per workgroup, 64 rows of M per K-tile, same
// four-phase / half-tile / counted-vmcnt schedule as gemm_tn256_kernel (X plays A, Y plays B).
// A half-tile is its own [64 m][128 cols] image (256-B rows, v1 swizzle):
//     X-half h : columns {wr*128 + h*64 + 0..63 | wr = 0,1}  stored at local column wr*64 + c
//     Y-half h : columns {wc*64  + h*32 + 0..31 | wc = 0..3} stored at local column wc*32 + c
// LDS: 2 buffers x (X0 | X1 | Y0 | Y1) x 16 KiB = 128 KiB, one workgroup per CU.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_wgrad256_kernel(WgradArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // problem of this block (workgroup-uniform scalar selects)
    const bool second = pa.blocks_a > 0 && (int)blockIdx.x >= pa.blocks_a;
    struct { const bf16_t* X; const bf16_t* Y; int M, N1, N2, ldx, ldy; float* slabs; int steps_per_split; } p;
    p.X = second ? pa.Xb : pa.X; p.Y = second ? pa.Yb : pa.Y; p.M = pa.M;
    p.N1 = second ? pa.N1b : pa.N1; p.N2 = second ? pa.N2b : pa.N2;
    p.ldx = second ? pa.ldxb : pa.ldx; p.ldy = second ? pa.ldyb : pa.ldy;
    p.slabs = second ? pa.slabsb : pa.slabs; p.steps_per_split = pa.steps_per_split;
    const int bid = second ? (int)blockIdx.x - pa.blocks_a : (int)blockIdx.x;
    const int nblk = pa.blocks_a > 0 ? (second ? (int)gridDim.x - pa.blocks_a : pa.blocks_a) : (int)gridDim.x;

    // XCD-aware order: the tiles of one M-split (same X/Y rows) get consecutive logical ids = one XCD's L2
    const int tiles2 = p.N2 >> 8;
    const int ntile = (p.N1 >> 8) * tiles2;
    const int lid = xcd_remap(bid, nblk);
    const int split = lid / ntile;
    const int tile = lid - split * ntile;
    const int t1 = tile / tiles2, t2 = tile - t1 * tiles2;
    const int n1_0 = t1 << 8, n2_0 = t2 << 8;

    const int nk_total = (p.M + BK - 1) / BK;
    const int k_begin = split * p.steps_per_split;
    const int nk = max(0, min(nk_total, k_begin + p.steps_per_split) - k_begin);
    const int nq = nk << 2;

    // ---- staging: kind j = 0:X-half0 1:Y-half0 2:Y-half1 3:X-half1; 2 wave-instructions (4 rows each) per kind ----
    // 32-bit byte offsets from p.X / p.Y (host checks the operands are < 4 GiB) so that the LDS-DMA takes the
    // "SGPR base + VGPR offset" form and a K-tile step is ONE v_add per instruction (64-bit address arithmetic per
    // load made this loop instruction-issue bound).  Rows >= M exist only in the very last K-tile: slow path there.
    int srow[2];
    unsigned soff[4][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int row = (wave * 2 + jj) * 4 + (lane >> 4);
        srow[jj] = row;
        const int lc = ((lane & 15) ^ wg_swz(row)) * 8;      // local column inside the half image
        const int c0 = n1_0 + ((lc >> 6) << 7) + (lc & 63), c1 = n2_0 + ((lc >> 5) << 6) + (lc & 31);
        const size_t m0_ = (size_t)k_begin * BK + row;
        soff[0][jj] = (unsigned)((m0_ * p.ldx + c0) * 2);
        soff[3][jj] = (unsigned)((m0_ * p.ldx + c0 + 64) * 2);
        soff[1][jj] = (unsigned)((m0_ * p.ldy + c1) * 2);
        soff[2][jj] = (unsigned)((m0_ * p.ldy + c1 + 32) * 2);
    }
    const unsigned kstep_x = (unsigned)(BK * p.ldx * 2), kstep_y = (unsigned)(BK * p.ldy * 2);
    const bool ragged = (p.M % BK) != 0;
    const char* baseX = reinterpret_cast<const char*>(p.X);
    const char* baseY = reinterpret_cast<const char*>(p.Y);
    // LDS map: the two K-tile buffers of each half image sit next to each other (X0: 0 / 16 K, X1: 32 K / 48 K, Y0: 64 K /
    // 80 K, Y1: 96 K / 112 K), so that with one base register for X and one for Y (+64 K) every fragment read of either
    // buffer is "precomputed lane offset + a 16-bit instruction offset": no address arithmetic in the K loop (it was
    // one v_add per read, 54 per K-tile and wave, in a loop that already issues twice the LDS reads of gemm_tn).
    constexpr int IMG_X0 = 0, IMG_X1 = 32768, IMG_Y0 = 65536, IMG_Y1 = 98304;
#define W2_ISSUE(t, J)                                                                                        \
    do {                                                                                                      \
        char* b_ = smem + (((t)&1) << 14) + ((J) == 0 ? IMG_X0 : (J) == 3 ? IMG_X1 : (J) == 1 ? IMG_Y0 : IMG_Y1); \
        const bool isx_ = ((J) == 0 || (J) == 3);                                                             \
        const char* base_ = isx_ ? baseX : baseY;                                                             \
        const unsigned ko_ = (unsigned)(t) * (isx_ ? kstep_x : kstep_y);                                      \
        if (ragged && (k_begin + (t)) == nk_total - 1) {                                                      \
            _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                \
                const int m_ = (k_begin + (t)) * BK + srow[jj];                                               \
                const void* s_ = (m_ < p.M) ? (const void*)(base_ + (size_t)(soff[J][jj] + ko_)) : (const void*)g_zero_line; \
                glds16(s_, b_ + (wave * 2 + jj) * 1024);                                                      \
            }                                                                                                 \
        } else {                                                                                              \
            glds16(base_ + (size_t)(soff[J][0] + ko_), b_ + (wave * 2) * 1024);                               \
            glds16(base_ + (size_t)(soff[J][1] + ko_), b_ + (wave * 2 + 1) * 1024);                           \
        }                                                                                                     \
    } while (0)

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposed fragment read offsets (bytes inside a half image) for sub-tile 0; sub-tile i: XOR (i << 5)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int xoff[2][2], yoff[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = ks * 32 + 8 * g + 4 * h + q;
            xoff[ks][h] = m * 256 + tr_col_off(wr * 64 + 4 * pp, m);
            yoff[ks][h] = m * 256 + tr_col_off(wc * 32 + 4 * pp, m);
        }

    if (nk > 0) {
        W2_ISSUE(0, 0); W2_ISSUE(0, 1); W2_ISSUE(0, 2); W2_ISSUE(0, 3);
        if (nk > 1) { W2_ISSUE(1, 0); W2_ISSUE(1, 1); W2_ISSUE(1, 2); }
    }
    bf16x8 fx[4][2], fy0[2][2], fy1[2][2];
    // (inline-asm transposed reads: see common.h).  All reads of a phase are issued, then ONE lgkmcnt wait.
    // Lane addresses of the eight reads of sub-tile pair (i0, i0+1): X relative to smem, Y relative to smem + 64 K.
    const unsigned lbase = lds_addr(smem);
    unsigned px[2][8], py[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ks = (j >> 1) & 1, h = j & 1, di = j >> 2;
        px[0][j] = lbase + (unsigned)(xoff[ks][h] ^ ((0 + di) << 5));
        px[1][j] = lbase + (unsigned)(xoff[ks][h] ^ ((2 + di) << 5));
        py[j] = lbase + 65536u + (unsigned)(yoff[ks][h] ^ (di << 5));
    }
#define W2_JOIN_X(R, I0)                                                                                     \
    fx[I0][0] = join_tr64(R[0], R[1]); fx[I0][1] = join_tr64(R[2], R[3]);                                    \
    fx[(I0) + 1][0] = join_tr64(R[4], R[5]); fx[(I0) + 1][1] = join_tr64(R[6], R[7]);
#define W2_JOIN_Y(FY, R)                                                                                     \
    FY[0][0] = join_tr64(R[0], R[1]); FY[0][1] = join_tr64(R[2], R[3]);                                      \
    FY[1][0] = join_tr64(R[4], R[5]); FY[1][1] = join_tr64(R[6], R[7]);
#define W2_MMA(MH, NH, FY)                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int a = 0; a < 4; ++a)           \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                        \
            acc[(MH)*4 + a][(NH)*2 + b] =                                                                    \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(FY[b][ks], fx[a][ks], acc[(MH)*4 + a][(NH)*2 + b], 0, 0, 0);

    // one K-tile out of buffer PAR (compile time: it selects the instruction offsets)
    // STEADY (compile time): K-tiles s with s + 3 < nk -- every issue is unconditional and off the ragged last K-tile,
    // and every wait is the full-depth vmcnt(10), so the phase heads carry no compare / branch ladders.
#define W2_ISSUE_FAST(t, J)                                                                                   \
    do {                                                                                                      \
        char* b_ = smem + (((t)&1) << 14) + ((J) == 0 ? IMG_X0 : (J) == 3 ? IMG_X1 : (J) == 1 ? IMG_Y0 : IMG_Y1); \
        const bool isx_ = ((J) == 0 || (J) == 3);                                                             \
        const char* base_ = isx_ ? baseX : baseY;                                                             \
        const unsigned ko_ = (unsigned)(t) * (isx_ ? kstep_x : kstep_y);                                      \
        glds16(base_ + (size_t)(soff[J][0] + ko_), b_ + (wave * 2) * 1024);                                   \
        glds16(base_ + (size_t)(soff[J][1] + ko_), b_ + (wave * 2 + 1) * 1024);                               \
    } while (0)
    auto ktile = [&](int s, auto par_, auto steady_) {
        constexpr int PAR = decltype(par_)::value;
        constexpr bool STEADY = decltype(steady_)::value;
        const int phi = s << 2;
        tr64_t ra[8], rb[8], rc[8];
        // ---- P1: X-half0 + Y-half0 ----
        if constexpr (STEADY) wait_vmcnt<10>(); else wait_halftiles(min(phi + 6, nq - 1) - (phi + 1));
        phase_barrier();
        tr_issue8o<IMG_X0 + PAR * 16384>(px[0], ra);
        tr_issue8o<IMG_X0 + PAR * 16384>(px[1], rb);
        tr_issue8o<IMG_Y0 - 65536 + PAR * 16384>(py, rc);
        if constexpr (STEADY) W2_ISSUE_FAST(s + 1, 3); else if (s + 1 < nk) W2_ISSUE(s + 1, 3);
        tr_wait24(ra, rb, rc);
        W2_JOIN_X(ra, 0) W2_JOIN_X(rb, 2) W2_JOIN_Y(fy0, rc)
        W2_MMA(0, 0, fy0)
        // ---- P2: Y-half1 ----
        if constexpr (STEADY) wait_vmcnt<10>(); else wait_halftiles(min(phi + 7, nq - 1) - (phi + 2));
        phase_barrier();
        tr_issue8o<IMG_Y1 - 65536 + PAR * 16384>(py, rc);
        tr_wait8(rc);
        W2_JOIN_Y(fy1, rc)
        if constexpr (STEADY) W2_ISSUE_FAST(s + 2, 0); else if (s + 2 < nk) W2_ISSUE(s + 2, 0);
        W2_MMA(0, 1, fy1)
        // ---- P3: X-half1 ----
        if constexpr (STEADY) wait_vmcnt<10>(); else wait_halftiles(min(phi + 8, nq - 1) - (phi + 3));
        phase_barrier();
        tr_issue8o<IMG_X1 + PAR * 16384>(px[0], ra);
        tr_issue8o<IMG_X1 + PAR * 16384>(px[1], rb);
        if constexpr (STEADY) W2_ISSUE_FAST(s + 2, 1); else if (s + 2 < nk) W2_ISSUE(s + 2, 1);
        tr_wait16(ra, rb);
        W2_JOIN_X(ra, 0) W2_JOIN_X(rb, 2)
        W2_MMA(1, 1, fy1)
        // ---- P4 ----
        if constexpr (STEADY) W2_ISSUE_FAST(s + 2, 2); else if (s + 2 < nk) W2_ISSUE(s + 2, 2);
        W2_MMA(1, 0, fy0)
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int s = 0;
    // steady pairs: the youngest K-tile they issue is s + 3 <= nk - 2, which is never the (possibly ragged) last one
    for (; s + 4 < nk; s += 2) {
        ktile(s, P0{}, std::true_type{});
        ktile(s + 1, P1{}, std::true_type{});
    }
    for (; s < nk; s += 2) {
        ktile(s, P0{}, std::false_type{});
        if (s + 1 < nk) ktile(s + 1, P1{}, std::false_type{});
    }
#undef W2_ISSUE_FAST
#undef W2_ISSUE
#undef W2_JOIN_X
#undef W2_JOIN_Y
#undef W2_MMA

    float* out = p.slabs + (size_t)split * p.N1 * p.N2;
    const int e1 = n1_0 + wr * 128 + (lane & 15);
    const int e2 = n2_0 + wc * 64 + (lane >> 4) * 4;
    // (staging the tile through LDS for whole-line stores, as the gemm_tn epilogues do, measured no difference here:
    //  one epilogue per ~200-us workgroup)
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            store_f4<GETMI_NT_SLAB != 0>(out + (size_t)(e1 + a * 16) * p.N2 + e2 + b * 16,
                                         make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]));
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slabs[s][i]        (n % 4 == 0)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, size_t n,
                                    int nsplit, int accumulate) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        float4 s = accumulate ? *reinterpret_cast<const float4*>(out + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < nsplit; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)k * n + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + i) = s;
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers (internal C++ API used by capi.hip)
// ------------------------------------------------------------------------------------------------
// tile selection: 0 = auto, 1 = force 128x128, 2 = force 256x256 (set from GETMI_GEMM_TILE for benchmarking)
static int g_tile_mode = -1;
static bool use_big_tile(const GemmArgs& a) {
    if (g_tile_mode < 0) {
        const char* e = getenv("GETMI_GEMM_TILE");
        g_tile_mode = e ? atoi(e) : 0;
    }
    if (g_tile_mode == 1) return false;
    if (g_tile_mode == 2) return true;
    return a.M >= 2048 && a.N >= 512;
}

// CUs the one-workgroup-per-CU kernels (persistent GEMM, wgrad) may assume they get.  These workgroups take a
// whole CU (148 KiB LDS, all 512 VGPRs of every SIMD), so a CU that hosts any other resident wave -- e.g. one of
// RCCL's channel workgroups while a gradient all-reduce overlaps backward -- cannot take one, and with a grid
// equal to the CU count the displaced workgroups run as a SECOND round (2x kernel time).  getmi_set_cu_limit(n)
// shrinks the grids to n so that cannot happen; ddp.autotune measures whether it pays.
static std::atomic<int> g_num_cu{0};
static std::atomic<int> g_cu_limit{0};
// One C-ABI call = one value of the CU budget: an entry point takes a snapshot when it starts (cu_scope_enter) and every
// grid size, split count and workspace layout inside that call uses the snapshot, so a getmi_set_cu_limit from another
// thread (ddp.GradBuckets flips it from the autograd thread while the main thread may be in forward) can never make the
// query, the check and the launch of one call disagree.  The *_workspace_bytes queries always answer for ALL CUs, which
// is an upper bound for every limit (fewer CUs -> fewer M-splits), so a buffer sized before a change still fits after it.
static thread_local int t_cu_snapshot = 0;
static int total_cu() {
    int n = g_num_cu.load(std::memory_order_relaxed);
    if (n == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        g_num_cu.store(n, std::memory_order_relaxed);
    }
    return n;
}
static int limited_cu() {
    const int n = total_cu(), lim = g_cu_limit.load(std::memory_order_relaxed);
    return (lim > 0 && lim < n) ? lim : n;
}
static int num_cu() { return t_cu_snapshot > 0 ? t_cu_snapshot : limited_cu(); }
// -> token for cu_scope_leave.  all_cus: snapshot the unlimited count (workspace queries).  Nested scopes keep the outer one.
int getmi_cu_scope_enter(int all_cus) {
    if (t_cu_snapshot > 0) return -1;
    t_cu_snapshot = all_cus ? total_cu() : limited_cu();
    return 0;
}
void getmi_cu_scope_leave(int token) {
    if (token == 0) t_cu_snapshot = 0;
}
int getmi_set_cu_limit_impl(int n) { return g_cu_limit.exchange(n > 0 ? n : 0, std::memory_order_relaxed); }
int getmi_cu_count_impl() { return limited_cu(); }

// Tile height of the 8-wave kernels for an M x N problem: 224 rows (RB1 = 3) when the rounds the CUs need, weighted by
// the tile height, come out shorter than with 256 rows (RB1 = 4) even after charging the short tile 6 % (same
// staging and barriers for 7/8 of the MFMAs; measured: N = 768 at M = 51 456 gains 3-6 %, N = 3072 would lose 3 %
// although its rounds x height is 4 % smaller).  GETMI_GEMM_BM=256|224 forces one.
static double tile_cost(int rb1);
static int tile_rows_of(int rb1) { return rb1 == 5 ? 208 : 128 + 32 * rb1; }
// allow208: the caller's epilogue has the asymmetric 208-row instantiation (plain bf16 and fp32 residual)
static int pick_rb1(int M, int N, bool allow208 = false) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("GETMI_GEMM_BM");
        forced = e ? atoi(e) : 0;
    }
    if (forced == 256) return 4;
    if (forced == 224) return 3;
    const int tiles_n = (N + 255) / 256, cus = num_cu();
    auto span = [&](int rb1) {      // rounds x relative time of one round of such tiles (tile_cost)
        const int bm = tile_rows_of(rb1);
        const long tiles = (long)((M + bm - 1) / bm) * tiles_n;
        return (double)((tiles + cus - 1) / cus) * tile_cost(rb1);
    };
    int best = 4;
    double c = span(4);
    if (span(3) < c) { best = 3; c = span(3); }
    // The 208-row tile is OFF by default: measured in the C2 step (three interleaved rounds, profiles/r03_ab_208.txt) 34.05-
    // 34.29 ms with it vs 33.99-34.06 without; per shape it moves proj dgrad 72 -> 69 us and fc1 dgrad 217 -> 210 us but proj /
    // fc2 forward +-1 us: the third round's idle CUs were not what these GEMMs wait for.  GETMI_GEMM_BM=208 selects it where
    // the round model prefers it (kept under test).
    if (allow208 && forced == 208 && span(5) < c - 1e-9) { best = 5; c = span(5); }
    return best;
}

template <int EPI, int RB1>
static int launch_gemm_tn256(const GemmArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256_kernel<EPI, RB1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
        attr_set = true;
    }
    static_assert(RB1 != 5, "the asymmetric tile exists in the persistent kernel only");
    constexpr int BMv = 128 + 32 * RB1;
    const int grid = ((a.M + BMv - 1) / BMv) * ((a.N + 255) / 256);
    hipLaunchKernelGGL((gemm_tn256_kernel<EPI, RB1>), dim3(grid), dim3(512), G2_LDS, st, a);
    GETMI_CHECK_LAUNCH("gemm_tn256");
    return GETMI_OK;
}

// Width (in N-tiles of `tile_n` columns) of the column groups of TileWalk: the divisor-free choice that minimises the
// bytes an XCD has to fetch from beyond its L2 under a simple model -- A is fetched once per column group; the B blocks of
// a group are fetched once per XCD when the group fits in about 2.5 MB of the 4 MiB L2 (they stay resident from round to
// round), and once per round otherwise.  GETMI_GEMM_GROUP=n forces a width (0 = old round-robin order).
static int pick_group_n(int M, int N, int K, int tile_m, int tile_n, int grid) {
    static int forced = -2;
    if (forced == -2) {
        const char* e = getenv("GETMI_GEMM_GROUP");
        forced = e ? atoi(e) : -1;
    }
    const int tiles_n = (N + tile_n - 1) / tile_n, tiles_m = (M + tile_m - 1) / tile_m;
    if (forced >= 0) return forced > tiles_n ? tiles_n : forced;
    if (grid % 8 != 0) return 0;
    const double a_bytes = 2.0 * M * K, blk = 2.0 * tile_n * K;
    const double rounds = (double)tiles_m * tiles_n / grid;
    double best = 0;
    int best_g = tiles_n;
    for (int g = 1; g <= tiles_n; ++g) {
        const int ncg = (tiles_n + g - 1) / g;
        const bool resident = g * blk <= 2.5e6;
        const double cost = a_bytes * ncg + (resident ? 8.0 : 8.0 * (rounds < 1 ? 1 : rounds)) * g * blk;
        if (best == 0 || cost < best - 1.0) { best = cost; best_g = g; }
    }
    return best_g;
}

template <int EPI, int RB1>
static int launch_gemm_tn256p(const GemmArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256p_kernel<EPI, RB1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS + 1024 + 16384);
        attr_set = true;
    }
    constexpr int BMv = TileRows<RB1>::bm;
    const int ntiles = ((a.M + BMv - 1) / BMv) * ((a.N + 255) / 256);
    const int grid = ntiles < num_cu() ? ntiles : num_cu();
    static int skew_env = -2;
    if (skew_env == -2) {
        const char* e = getenv("GETMI_GEMM_SKEW");
        skew_env = e ? atoi(e) : -1;
    }
    const int skew = skew_env >= 0 ? skew_env : 0;
    hipLaunchKernelGGL((gemm_tn256p_kernel<EPI, RB1>), dim3(grid), dim3(512), G2_LDS + 1024 + 16384, st, a, ntiles,
                       pick_group_n(a.M, a.N, a.K, BMv, 256, grid), skew);
    GETMI_CHECK_LAUNCH("gemm_tn256p");
    return GETMI_OK;
}

// second generation (gemm_tn2_kernel): which epilogues / shapes take it.  GETMI_GEMM_GEN: 1 = never, 2 = every epilogue it
// implements, unset = the per-epilogue choice measured on MI355X at the bench shapes.
static int gemm_gen_mode() {
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("GETMI_GEMM_GEN");
        mode = e ? atoi(e) : 0;
    }
    return mode;
}
template <int EPI>
static bool use_gen2(const GemmArgs& a) {
    constexpr bool impl = (EPI == EPI_BF16 || EpiKind<EPI>::gelu || EpiKind<EPI>::dgelu || EPI == EPI_BIAS_RESID);
    if (!impl || gemm_gen_mode() == 1) return false;
    const size_t esz = (EPI == EPI_BIAS_RESID) ? 4 : 2;
    const bool ok = a.K >= 128 && a.N % 8 == 0 && (size_t)a.M * a.lda * 2 < 0xF0000000ull &&
                    (size_t)a.N * a.ldb * 2 < 0xF0000000ull && (size_t)a.M * a.ldo * esz < 0xF0000000ull;
    if (!ok) return false;
    if (gemm_gen_mode() == 2) return true;
    // Default: OFF.  Measured on MI355X at M = 51 456 (profiles/r02_gemm_generations.txt): back to back on warm caches the
    // two-workgroup kernel wins some shapes (qkv forward 207 -> 194 us, proj dgrad 72 -> 66, dGELU-form 334 -> 321) and
    // loses others (fc1 forward 310 -> 333, plain N = 3072 226 -> 239); INSIDE the training step, where operands come
    // from HBM / the Infinity Cache rather than from a warm L2, it loses everywhere (plain bf16 GEMMs 145 -> 165 us
    // average, dGELU-form 308 -> 315; 36.1 -> 36.8 ms per step): its 1.5 x operand traffic per FLOP through the CU's one
    // vector-memory path costs more than the unsynchronised epilogues give back.  Staggering the two workgroups of a CU
    // (GETMI_GEMM_STAGGER) or the CUs of the chip (GETMI_GEMM_SKEW) only added their delay.  Kept as GETMI_GEMM_GEN=2.
    return false;
}

template <int EPI>
static int launch_gemm_tn2(const GemmArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn2_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
        attr_set = true;
    }
    const int ntiles = ((a.M + 255) / 256) * ((a.N + 127) / 128);
    const int slots = 2 * num_cu();
    const int grid = ntiles < slots ? ntiles : slots;
    // stagger of the second workgroup of each CU, in units of s_sleep 127 (~8 k cycles): half of a tile's main loop at
    // ~4.5 k cycles per K-tile and workgroup.  GETMI_GEMM_STAGGER overrides (0 = none).
    static int forced = -2;
    if (forced == -2) {
        const char* e = getenv("GETMI_GEMM_STAGGER");
        forced = e ? atoi(e) : -1;
    }
    const int nk = a.K / 64;
    const int stagger = forced >= 0 ? forced : (ntiles > slots ? (nk * 9 + 31) / 32 : 0);
    hipLaunchKernelGGL((gemm_tn2_kernel<EPI>), dim3(grid), dim3(256), G3_LDS, st, a, ntiles,
                       pick_group_n(a.M, a.N, a.K, 256, 128, grid), stagger);
    GETMI_CHECK_LAUNCH("gemm_tn2");
    return GETMI_OK;
}

// ---- tile-height plan ------------------------------------------------------------------------------------------
// Relative time of one round of tiles of height 256 / 224 / 192 rows (RB1 = 4 / 3 / 2): a K-tile costs the same staging,
// barriers and fragment reads at every height and 64 / 56 / 48 MFMAs per wave (measured for 224; linear model for 192).
static double tile_cost(int rb1) { return rb1 == 4 ? 1.0 : rb1 == 3 ? 0.9275 : rb1 == 5 ? 0.895 : 0.87; }
struct RowSeg { int rows, rb1; };
// A problem whose tile count is not close to a multiple of the CU count loses a partial round.  For the epilogues that
// only index by row (plain bf16, fp32 residual) the rows may be cut in two launches with different tile heights so that
// each launch is close to whole rounds: M = 51 456, N = 768 is 3 rounds of 224-row tiles, or ONE round of 224 (85 row
// blocks) + two of 192.  GETMI_GEMM_MIXED=0 turns this off.
static int plan_rows(int M, int N, bool may_split, RowSeg seg[2]) {
    static int mixed = -1;
    if (mixed < 0) {
        const char* e = getenv("GETMI_GEMM_MIXED");
        mixed = e ? atoi(e) : 1;
    }
    const int tiles_n = (N + 255) / 256, cus = num_cu();
    const int rb_u = pick_rb1(M, N, may_split), h_u = tile_rows_of(rb_u);
    const long tiles_u = (long)((M + h_u - 1) / h_u) * tiles_n;
    double best = (double)((tiles_u + cus - 1) / cus) * tile_cost(rb_u);
    seg[0] = RowSeg{M, rb_u};
    int nseg = 1;
    if (!may_split || !mixed || tiles_u <= cus) return nseg;
    static const int pairs[3][2] = {{4, 3}, {4, 2}, {3, 2}};
    for (int pi = 0; pi < 3; ++pi) {
        const int rb1 = pairs[pi][0], rb2 = pairs[pi][1], h1 = 128 + 32 * rb1, h2 = 128 + 32 * rb2;
        for (int r1 = 1; r1 <= 16; ++r1) {
            const long nb1 = (long)r1 * cus / tiles_n;          // whole row blocks of r1 full rounds
            const long rows1 = nb1 * h1;
            if (nb1 == 0 || rows1 >= M) break;
            const long nb2 = (M - rows1 + h2 - 1) / h2;
            const long r2 = (nb2 * tiles_n + cus - 1) / cus;
            // + the second launch: the drain of one grid and the fill of the next cost ~4 us, a fifth of a K = 768 round.
            // Measured at M = 51 456: N = 2304 (1 round of 256 + 7 of 224 instead of 8 of 256) 236 -> 228 us; N = 768
            // (1 of 224 + 2 of 192 instead of 3 of 224) +1 ... +5 %, which this charge excludes.
            const double c = r1 * tile_cost(rb1) + r2 * tile_cost(rb2) + 0.2;
            if (c < best - 1e-9) {
                best = c;
                seg[0] = RowSeg{(int)rows1, rb1};
                seg[1] = RowSeg{(int)(M - rows1), rb2};
                nseg = 2;
            }
        }
    }
    return nseg;
}

template <int EPI, int RB1>
static int launch_gemm_big(const GemmArgs& a, hipStream_t st) {
    // (BIAS_RESID used to stay on the one-tile-per-workgroup kernel -- persistent measured 302 vs 278 us for fc2 when its
    //  epilogue read the residual as 64-byte pieces behind the stores; with the row-major whole-line residual reads and
    //  the lean K loop the persistent form is ahead: attn.proj 122 -> 116 us, fc2 257 -> 256 us.)
    constexpr bool can_persist = (EPI == EPI_BF16 || EpiKind<EPI>::gelu || EpiKind<EPI>::dgelu || EPI == EPI_BIAS_RESID);
    const size_t esz = (EPI == EPI_BIAS_RESID) ? 4 : 2;
    const bool fits32 = (size_t)a.M * a.lda * 2 < 0xF0000000ull && (size_t)a.N * a.ldb * 2 < 0xF0000000ull &&
                        (size_t)a.M * a.ldo * esz < 0xF0000000ull;
    if constexpr (can_persist) {
        if (a.K >= 128 && a.N % 8 == 0 && fits32 && g_tile_mode != 3) return launch_gemm_tn256p<EPI, RB1>(a, st);
    }
    if constexpr (RB1 == 5) return launch_gemm_tn256<EPI, 3>(a, st);      // (not persistent-eligible: the 224-row one-tile kernel)
    else return launch_gemm_tn256<EPI, RB1>(a, st);
}

template <int EPI>
static int launch_gemm_tn(const GemmArgs& a, hipStream_t st) {
    if (use_big_tile(a)) {
        if constexpr (EPI == EPI_BF16 || EpiKind<EPI>::gelu || EpiKind<EPI>::dgelu || EPI == EPI_BIAS_RESID) {
            if (use_gen2<EPI>(a)) return launch_gemm_tn2<EPI>(a, st);
        }
        constexpr bool splittable = (EPI == EPI_BF16 || EPI == EPI_BIAS_RESID);
        RowSeg seg[2];
        const int nseg = plan_rows(a.M, a.N, splittable, seg);
        int row0 = 0;
        for (int i = 0; i < nseg; ++i) {
            GemmArgs s = a;
            const size_t esz = (EPI == EPI_BIAS_RESID) ? 4 : 2;
            s.M = seg[i].rows;
            s.row0 = a.row0 + row0;
            s.A = a.A + (size_t)row0 * a.lda;
            s.out0 = reinterpret_cast<char*>(a.out0) + (size_t)row0 * a.ldo * esz;
            if (a.resid) s.resid = a.resid + (size_t)row0 * a.ldo;
            int rc;
            if constexpr (splittable) {
                rc = seg[i].rb1 == 4 ? launch_gemm_big<EPI, 4>(s, st)
                   : seg[i].rb1 == 3 ? launch_gemm_big<EPI, 3>(s, st)
                   : seg[i].rb1 == 5 ? launch_gemm_big<EPI, 5>(s, st) : launch_gemm_big<EPI, 2>(s, st);
            } else {
                rc = seg[i].rb1 == 4 ? launch_gemm_big<EPI, 4>(s, st) : launch_gemm_big<EPI, 3>(s, st);
            }
            if (rc) return rc;
            row0 += seg[i].rows;
        }
        return GETMI_OK;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr_set = true;
    }
    const int grid = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm_tn_kernel<EPI>, dim3(grid), dim3(256), GEMM_LDS, st, a);
    GETMI_CHECK_LAUNCH("gemm_tn");
    return GETMI_OK;
}

// number of partial rows the DGELU column-sum epilogue writes for this problem (depends on the tile choice, which
// depends on the CU budget of the running call)
int getmi_gemm_tn_colsum_rows(int M, int N) {
    GemmArgs a{};
    a.M = M; a.N = N;
    if (!use_big_tile(a)) return ((M + 127) / 128) * 2;
    const int bm = pick_rb1(M, N) == 3 ? 224 : 256;
    return ((M + bm - 1) / bm) * 4;
}
// 1 when the launch this call would make writes EVERY entry of its getmi_gemm_tn_colsum_rows(M, N) partial rows (no clear
// needed); the diagnostic two-workgroup generation (GETMI_GEMM_GEN=2) uses its own tile grid and leaves rows untouched
int getmi_gemm_tn_colsum_dense() { return gemm_gen_mode() == 2 ? 0 : 1; }
// ... and its upper bound over every tile choice (what a workspace must hold whatever the CU limit is at launch time)
int getmi_gemm_tn_colsum_rows_max(int M, int N) {
    GemmArgs a{};
    a.M = M; a.N = N;
    if (!use_big_tile(a)) return ((M + 127) / 128) * 2;
    return ((M + 223) / 224) * 4;
}

// Can a GEMM of this shape hand gelu'(u) from the BIAS_GELUG epilogue to the DMUL epilogue in the blocked layout
// (GemmArgs::blocked)?  Only the persistent 256-row-tile kernel implements it; the answer depends on the shape and on
// process-wide diagnostic switches, never on the CU limit, so the forward and the backward call of one layer agree.
int getmi_gemm_tn_blocked_ok_impl(int M, int N, int K) {
    GemmArgs a{};
    a.M = M; a.N = N; a.K = K;
    if (M <= 0 || N <= 0 || K <= 0 || !use_big_tile(a) || g_tile_mode == 3 || gemm_gen_mode() != 0) return 0;
    const size_t lim = 0xF0000000ull;
    return (K >= 128 && K % BK == 0 && N % 64 == 0 && (size_t)M * K * 2 < lim && (size_t)N * K * 2 < lim &&
            (size_t)((M + 15) & ~15) * N * 2 < lim) ? 1 : 0;
}

int getmi_gemm_tn_impl(int epi, const GemmArgs& a, hipStream_t st) {
    GETMI_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "gemm_tn: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    if (a.blocked) {
        GETMI_CHECK_ARG(epi == EPI_BIAS_GELUG || epi == EPI_DMUL, "gemm_tn: the blocked layout (flags bit 1) exists for epilogues 6 (out0) and 7 (aux) only");
        GETMI_CHECK_ARG(getmi_gemm_tn_blocked_ok_impl(a.M, a.N, a.K) && a.lda == a.K && a.ldb == a.K &&
                        (epi == EPI_BIAS_GELUG ? a.ldo == a.N : (a.ldaux == a.N && a.ldo == a.N)),
                        "gemm_tn: blocked layout not available for M=%d N=%d K=%d (ask getmi_gemm_tn_blocked_ok; dense operands)", a.M, a.N, a.K);
    }
    GETMI_CHECK_ARG(a.K % BK == 0, "gemm_tn: K=%d must be a multiple of %d (pad the operands)", a.K, BK);
    GETMI_CHECK_ARG(a.N % 4 == 0 && a.ldo % 4 == 0, "gemm_tn: N=%d and ldo=%d must be multiples of 4", a.N, a.ldo);
    GETMI_CHECK_ARG(a.lda % 8 == 0 && a.ldb % 8 == 0, "gemm_tn: lda/ldb must be multiples of 8 (16-byte rows)");
    GETMI_CHECK_ARG(a.lda >= a.K && a.ldb >= a.K && a.ldo >= a.N, "gemm_tn: leading dimension too small");
    GETMI_CHECK_ARG(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.B % 16 == 0) && ((uintptr_t)a.out0 % 16 == 0),
                    "gemm_tn: pointers must be 16-byte aligned");
    GETMI_CHECK_ARG(a.out0 != nullptr || epi == EPI_BIAS_GELU || epi == EPI_BIAS_GELUG,
                    "gemm_tn: out0 may be NULL only for the two-output GELU epilogues");
    switch (epi) {
        case EPI_BF16: return launch_gemm_tn<EPI_BF16>(a, st);
        case EPI_BIAS_GELU:
            GETMI_CHECK_ARG(a.out1 && a.bias, "gemm_tn: bias_gelu needs out1 and bias");
            return launch_gemm_tn<EPI_BIAS_GELU>(a, st);
        case EPI_BIAS_RESID:
            GETMI_CHECK_ARG(a.resid && (!a.rowscale || a.rows_per_sample > 0), "gemm_tn: bias_resid needs resid");
            return launch_gemm_tn<EPI_BIAS_RESID>(a, st);
        case EPI_DGELU:
            GETMI_CHECK_ARG(a.aux && a.ldaux % 4 == 0, "gemm_tn: dgelu needs aux");
            return launch_gemm_tn<EPI_DGELU>(a, st);
        case EPI_EMBED:
            GETMI_CHECK_ARG(a.regions_per_sample > 0 && a.M % a.regions_per_sample == 0,
                            "gemm_tn: embed needs regions_per_sample dividing M");
            GETMI_CHECK_ARG(!a.mask || a.mask_token, "gemm_tn: embed mask needs mask_token");
            return launch_gemm_tn<EPI_EMBED>(a, st);
        case EPI_F32: return launch_gemm_tn<EPI_F32>(a, st);
        case EPI_BIAS_GELUG:
            GETMI_CHECK_ARG(a.out1 && a.bias, "gemm_tn: bias_gelu_grad needs out1 and bias");
            GETMI_CHECK_ARG(!a.rowscale || a.rows_per_sample > 0, "gemm_tn: bias_gelu_grad row scale needs rows_per_sample");
            return launch_gemm_tn<EPI_BIAS_GELUG>(a, st);
        case EPI_DMUL:
            GETMI_CHECK_ARG(a.aux && a.ldaux % 4 == 0, "gemm_tn: dmul needs aux");
            return launch_gemm_tn<EPI_DMUL>(a, st);
    }
    getmi_set_error("gemm_tn: unknown epilogue %d", epi);
    return GETMI_ERR_ARG;
}

// choose the number of M-splits so that tiles*splits fills the chip (~2 blocks per CU on 256 CUs)
static bool wgrad_big(int M, int N1, int N2) {
    if (g_tile_mode < 0) {
        const char* e = getenv("GETMI_GEMM_TILE");
        g_tile_mode = e ? atoi(e) : 0;
    }
    if (g_tile_mode == 1) return false;
    return (N1 % 256 == 0) && (N2 % 256 == 0) && M >= 4096 && (size_t)M * (N1 > N2 ? N1 : N2) * 2 < 0xF0000000ull;
}

int getmi_wgrad_nsplit(int M, int N1, int N2) {
    const bool big = wgrad_big(M, N1, N2);
    const int ntile = big ? (N1 / 256) * (N2 / 256) : ((N1 + 127) / 128) * ((N2 + 127) / 128);
    const int nk = (M + BK - 1) / BK;
    // one full wave of blocks (big tile: 1 block/CU -> 256 slots; 128x128: 2 blocks/CU -> 512 slots): never
    // spill a few blocks into a second round
    int want = (big ? num_cu() : 2 * num_cu()) / ntile;
    if (want < 1) want = 1;
    if (want > nk) want = nk;
    if (want > 64) want = 64;
    const int steps = (nk + want - 1) / want;
    return (nk + steps - 1) / steps;
}

int getmi_gemm_wgrad_impl(const bf16_t* X, const bf16_t* Y, int M, int N1, int N2, int ldx, int ldy, float* out,
                          int accumulate, float* slabs, size_t slab_bytes, hipStream_t st) {
    GETMI_CHECK_ARG(M > 0 && N1 >= 8 && N2 >= 8, "gemm_wgrad: bad shape M=%d N1=%d N2=%d", M, N1, N2);
    GETMI_CHECK_ARG(N1 % 8 == 0 && N2 % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0,
                    "gemm_wgrad: N1, N2, ldx, ldy must be multiples of 8");
    GETMI_CHECK_ARG(((uintptr_t)X % 16 == 0) && ((uintptr_t)Y % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                        ((uintptr_t)slabs % 16 == 0),
                    "gemm_wgrad: pointers must be 16-byte aligned");
    const int nsplit = getmi_wgrad_nsplit(M, N1, N2);
    GETMI_CHECK_ARG(slab_bytes >= (size_t)nsplit * N1 * N2 * sizeof(float),
                    "gemm_wgrad: workspace too small (%zu < %zu)", slab_bytes, (size_t)nsplit * N1 * N2 * sizeof(float));
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr_set = true;
    }
    WgradArgs a{};
    a.X = X; a.Y = Y; a.M = M; a.N1 = N1; a.N2 = N2; a.ldx = ldx; a.ldy = ldy; a.slabs = slabs;
    const int nk = (M + BK - 1) / BK;
    a.steps_per_split = (nk + nsplit - 1) / nsplit;
    if (wgrad_big(M, N1, N2)) {
        static bool attr_set2 = false;
        if (!attr_set2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad256_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
            attr_set2 = true;
        }
        hipLaunchKernelGGL(gemm_wgrad256_kernel, dim3((N1 / 256) * (N2 / 256) * nsplit), dim3(512), G2_LDS, st, a);
    } else {
        const int ntile = ((N1 + 127) / 128) * ((N2 + 127) / 128);
        hipLaunchKernelGGL(gemm_wgrad_kernel, dim3(ntile * nsplit), dim3(256), GEMM_LDS, st, a);
    }
    GETMI_CHECK_LAUNCH("gemm_wgrad");
    const size_t n = (size_t)N1 * N2;
    int rgrid = (int)((n / 4 + 255) / 256);
    if (rgrid > 2048) rgrid = 2048;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(rgrid), dim3(256), 0, st, slabs, out, n, nsplit, accumulate);
    GETMI_CHECK_LAUNCH("reduce_slabs");
    return GETMI_OK;
}

// ---- two weight gradients over the same rows in ONE launch (both on the 256x256 kernel) -----------------------
static int wgrad_pair_nsplit(int M, int ta, int tb) {
    const int nk = (M + BK - 1) / BK;
    int want = num_cu() / (ta + tb);
    if (want < 1) want = 1;
    if (want > nk) want = nk;
    if (want > 64) want = 64;
    const int steps = (nk + want - 1) / want;
    return (nk + steps - 1) / steps;
}
static bool wgrad_pair_ok(int M, int N1a, int N2a, int N1b, int N2b) {
    return wgrad_big(M, N1a, N2a) && wgrad_big(M, N1b, N2b);
}
size_t getmi_wgrad_pair_workspace_bytes(int M, int N1a, int N2a, int N1b, int N2b) {
    if (!wgrad_pair_ok(M, N1a, N2a, N1b, N2b))     // falls back to two launches sharing the workspace
        return sizeof(float) * ((size_t)getmi_wgrad_nsplit(M, N1a, N2a) * N1a * N2a > (size_t)getmi_wgrad_nsplit(M, N1b, N2b) * N1b * N2b
                                    ? (size_t)getmi_wgrad_nsplit(M, N1a, N2a) * N1a * N2a
                                    : (size_t)getmi_wgrad_nsplit(M, N1b, N2b) * N1b * N2b);
    const int ns = wgrad_pair_nsplit(M, (N1a / 256) * (N2a / 256), (N1b / 256) * (N2b / 256));
    return (size_t)ns * ((size_t)N1a * N2a + (size_t)N1b * N2b) * sizeof(float);
}
int getmi_gemm_wgrad_pair_impl(const bf16_t* Xa, const bf16_t* Ya, int N1a, int N2a, int ldxa, int ldya, float* outa,
                               int acca, const bf16_t* Xb, const bf16_t* Yb, int N1b, int N2b, int ldxb, int ldyb,
                               float* outb, int accb, int M, float* slabs, size_t slab_bytes, hipStream_t st) {
    GETMI_CHECK_ARG(slab_bytes >= getmi_wgrad_pair_workspace_bytes(M, N1a, N2a, N1b, N2b), "gemm_wgrad_pair: workspace too small");
    if (!wgrad_pair_ok(M, N1a, N2a, N1b, N2b)) {
        if (int rc = getmi_gemm_wgrad_impl(Xa, Ya, M, N1a, N2a, ldxa, ldya, outa, acca, slabs, slab_bytes, st)) return rc;
        return getmi_gemm_wgrad_impl(Xb, Yb, M, N1b, N2b, ldxb, ldyb, outb, accb, slabs, slab_bytes, st);
    }
    GETMI_CHECK_ARG(ldxa % 8 == 0 && ldya % 8 == 0 && ldxb % 8 == 0 && ldyb % 8 == 0, "gemm_wgrad_pair: ld must be multiples of 8");
    GETMI_CHECK_ARG(((uintptr_t)Xa % 16 == 0) && ((uintptr_t)Ya % 16 == 0) && ((uintptr_t)Xb % 16 == 0) &&
                        ((uintptr_t)Yb % 16 == 0) && ((uintptr_t)outa % 16 == 0) && ((uintptr_t)outb % 16 == 0) &&
                        ((uintptr_t)slabs % 16 == 0), "gemm_wgrad_pair: pointers must be 16-byte aligned");
    const int ta = (N1a / 256) * (N2a / 256), tb = (N1b / 256) * (N2b / 256);
    const int ns = wgrad_pair_nsplit(M, ta, tb);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad256_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
        attr_set = true;
    }
    WgradArgs a{};
    a.X = Xa; a.Y = Ya; a.M = M; a.N1 = N1a; a.N2 = N2a; a.ldx = ldxa; a.ldy = ldya; a.slabs = slabs;
    a.Xb = Xb; a.Yb = Yb; a.N1b = N1b; a.N2b = N2b; a.ldxb = ldxb; a.ldyb = ldyb;
    a.slabsb = slabs + (size_t)ns * N1a * N2a;
    a.blocks_a = ta * ns;
    const int nk = (M + BK - 1) / BK;
    a.steps_per_split = (nk + ns - 1) / ns;
    hipLaunchKernelGGL(gemm_wgrad256_kernel, dim3((ta + tb) * ns), dim3(512), G2_LDS, st, a);
    GETMI_CHECK_LAUNCH("gemm_wgrad_pair");
    const size_t na = (size_t)N1a * N2a, nb = (size_t)N1b * N2b;
    int ga = (int)((na / 4 + 255) / 256), gb = (int)((nb / 4 + 255) / 256);
    if (ga > 2048) ga = 2048;
    if (gb > 2048) gb = 2048;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ga), dim3(256), 0, st, a.slabs, outa, na, ns, acca);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gb), dim3(256), 0, st, a.slabsb, outb, nb, ns, accb);
    GETMI_CHECK_LAUNCH("reduce_slabs");
    return GETMI_OK;
}
