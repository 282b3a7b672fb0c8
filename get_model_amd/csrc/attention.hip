// Fused (flash-style) multi-head attention for the GET region encoder on gfx950, head_dim 64.
//
// Replaces model/transformer.py:58-96 of the reference (reshape/permute, q*scale, q@k^T, softmax, attn@v,
// transpose) and its autograd backward.  Input is the packed projection qkv[B,T,3,H,64] (bf16) exactly as
// the QKV GEMM writes it; output is o[B,T,H*64] (bf16) plus the per-row logsumexp [B,H,T] (fp32, natural
// log) that the backward uses to recompute probabilities.  The [B,H,T,T] score tensor never exists.
//
// forward  : one workgroup = 4 waves = 128 query rows of one (b,h); K/V stream through LDS in 64-key chunks
//            (16-B global_load_lds, swizzled image, double buffered).  "Query on the lane": S^T = K.Q^T so
//            each lane owns one query column -> online softmax is lane-local (+1 exchange with lane^32),
//            and the P^T accumulator feeds the PV MFMA as its B operand without leaving registers.
// backward : one workgroup = 8 waves = 256 keys of one (b,h) (wave w owns 32 keys; dK^T, dV^T stay in its
//            accumulators for the whole sweep over 32-row query slices).  "Key on the lane": S = Q.K^T and
//            dP = dO.V^T accumulators are the B operands of dV^T += dO^T.P and dK^T += Q^T.dS; only dS
//            crosses LDS (transposed image) for dQ = dS.K, which the 8 waves compute as eight 16x16 tiles
//            over all 256 keys -> no cross-wave or cross-workgroup reduction when T <= 256 (per-key-block fp32
//            slabs summed by a small kernel otherwise).
#include "common.h"
static constexpr bool kNtAttn = GETMI_NT_ATTN != 0;    // outputs are streamed (see common.h)
#include <stdlib.h>

#ifndef GETMI_NT_ATTN_DQ
#define GETMI_NT_ATTN_DQ 0    // the dQ tiles leave as 32-byte pieces (four waves share a 128-byte row): streamed they reach HBM
                              // as partial writes (WRITE_SIZE 239 -> 319 MB per launch); cached 32.86 -> 32.69 ms/step (4 rounds)
#endif
#ifndef GETMI_ATTN_V_NT
#define GETMI_ATTN_V_NT 0     // cache policy of the streamed backward's V loads (asm)
#endif
#if GETMI_ATTN_V_NT
#define ATTN_V_POL " nt"
#else
#define ATTN_V_POL ""
#endif
#define HD 64
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// chunk swizzle for images with 128-byte rows (64 bf16): serves both ds_read_b128 row reads (16-lane groups
// see 8 distinct values per row parity) and ds_read_b64_tr_b16 transposed reads (rows r, r+2 land in
// different 64-B halves).
__device__ __forceinline__ int swz128(int row) {
    const int x = (row >> 1) & 7;
    return ((x & 1) << 2) | (x >> 1);
}
// byte offset of element column `col` (multiple of 4) of row `row` in a 128-B-row swizzled image
__device__ __forceinline__ int off128(int row, int col) {
    return row * 128 + ((((col >> 3) ^ swz128(row)) << 4) | ((col & 4) << 1));
}
__device__ __forceinline__ s16x4 lds_tr16a(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 join_tr_a(s16x4 lo, s16x4 hi) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                        float a7) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 v = {pack2bf(a0, a1), pack2bf(a2, a3), pack2bf(a4, a5), pack2bf(a6, a7)};
    return __builtin_bit_cast(bf16x8, v);
}
// Row-per-lane accumulators (lane = row r of 32, half h2; registers = 4 consecutive d at 8*g4 + 4*h2 of each 32-wide
// d block) leave a kernel as 8-byte pieces on 64 different rows per store instruction, which is store-ISSUE bound
// (measured with s_memtime stamps: 4-7 k cycles for the dK / dV rows of one backward workgroup).  Staged through a
// wave-private LDS block [rows][128 B] (16-B chunk index XOR (row & 7)) they leave as whole 128-byte rows instead:
// 16 bytes per lane, 8 rows per instruction.  An 8-byte LDS write is serviced in groups of 16 consecutive lanes (16
// consecutive rows here) over 32 banks = 128 B, so the (chunk, half) slot must differ across 16 rows: rows with bit 3 set
// hold the two 8-byte halves of every chunk swapped, and the read-back swaps them again (2-way conflicts before).
__device__ __forceinline__ void stage_row_piece(char* blk, int row, int d0, uint2 v) {
    *reinterpret_cast<uint2*>(blk + row * 128 + ((((d0 >> 3) ^ (row & 7)) << 4) | ((((d0 >> 2) ^ (row >> 3)) & 1) << 3))) = v;
}
__device__ __forceinline__ uint4 stage_row_chunk(const char* blk, int row, int chunk) {
    const uint4 v = *reinterpret_cast<const uint4*>(blk + row * 128 + ((chunk ^ (row & 7)) << 4));
    return ((row >> 3) & 1) ? make_uint4(v.z, v.w, v.x, v.y) : v;
}
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

struct AttnFwdArgs {
    const bf16_t* qkv;
    bf16_t* out;
    bf16_t* out_lo;   // optional [B,T,H*64]: bf16(O - float(out)), the rounding residual of `out` (see attn_bwd_kernel: delta)
    float* lse;
    int B, T, H;
    float scale;
    const float* sample_scale;   // optional [B]: out = sample_scale[b] * softmax(QK^T)V (the DropPath factor of the attention
                                 // branch folded into the saved output, functional.py); samples with factor 0 are skipped
};

__global__ __launch_bounds__(512) void attn_fwd_kernel(AttnFwdArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 8192];   // [buf][K|V][64 keys][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, H = p.H;
    const int nw = blockDim.x >> 6;                 // waves per workgroup (4..8, chosen by the launcher): 32 queries each
    const int nqb = (T + 32 * nw - 1) / (32 * nw);
    const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
    const int b = bh / H, h = bh - b * H;
    const size_t rs = (size_t)3 * H * HD;   // elements per token row of qkv
    const bf16_t* base = p.qkv + (size_t)b * T * rs + h * HD;
    const int q0 = qb * 32 * nw + wave * 32;
    const int r = lane & 31, h2 = lane >> 5;
    const float sb = p.sample_scale ? p.sample_scale[b] : 1.0f;      // workgroup-uniform
    if (sb == 0.0f) {
        // dropped sample (stochastic depth): its branch contributes nothing -- zero rows, no arithmetic, no barrier taken
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 8 + (lane >> 3), ch = lane & 7;
            if (q0 + row < T) {
                const size_t o = ((size_t)b * T + q0 + row) * (H * HD) + h * HD + ch * 8;
                store_u4<kNtAttn>(p.out + o, z);
                if (p.out_lo != nullptr) store_u4<kNtAttn>(p.out_lo + o, z);
            }
        }
        if (h2 == 0 && q0 + r < T) p.lse[((size_t)b * H + h) * T + q0 + r] = 0.f;
        return;
    }

    bf16x8 qf[4];
    {
        const int qrow = min(q0 + r, T - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            qf[kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * rs + kk * 16 + h2 * 8);
    }

    const int nchunk = (T + 63) >> 6;
    // staging: K chunk = 8 wave-instructions (8 rows x 128 B each) spread over the waves, same for V
    auto stage = [&](int buf, int c) {
        char* kb = smem + buf * 16384;
        char* vb = kb + 8192;
        for (int I = wave; I < 8; I += nw) {
            const int row = I * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz128(row);
            const int key = min(c * 64 + row, T - 1);
            const bf16_t* src = base + (size_t)key * rs + chunk * 8;
            glds16_attn(src + H * HD, kb + I * 1024);
            glds16_attn(src + 2 * H * HD, vb + I * 1024);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const float c2 = p.scale * LOG2E;
    const int gi = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;

    stage(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunk) stage(cur ^ 1, c + 1);
        const char* kb = smem + cur * 16384;
        const char* vb = kb + 8192;
        if (q0 < T) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int key0 = c * 64 + sub * 32;
                if (key0 < T) {
                    f32x16 s;
#pragma unroll
                    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int row = sub * 32 + r;
                        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(
                            kb + row * 128 + (((2 * kk + h2) ^ swz128(row)) << 4));
                        s = MFMA32(kf, qf[kk], s);
                    }
                    // (this loop is VALU-issue bound, not MFMA bound: keep the per-element work at max / fma / exp / add)
                    if (key0 + 32 > T) {   // only the last sub-block has padded keys (wave-uniform)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * h2;
                            s[i] = (key < T) ? s[i] : -INFINITY;
                        }
                    }
                    float mloc = s[0];
#pragma unroll
                    for (int i = 1; i < 16; ++i) mloc = fmaxf(mloc, s[i]);
                    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * c2;       // c2 > 0: max commutes with the scale
                    const float mnew = fmaxf(m, mloc);
                    float lsum = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        s[i] = __builtin_amdgcn_exp2f(fmaf(s[i], c2, -mnew));
                        lsum += s[i];
                    }
                    if (__builtin_amdgcn_ballot_w64(mnew != m) != 0) {   // running max grew somewhere: rescale (exact)
                        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                        l *= alpha;
#pragma unroll
                        for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
                        m = mnew;
                    }
                    l += lsum;
                    bf16x8 pf[2];
                    pf[0] = pack8(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
                    pf[1] = pack8(s[8], s[9], s[10], s[11], s[12], s[13], s[14], s[15]);
                    // V^T fragments by inline-asm transposed reads: in front of the ds_read_tr intrinsic hipcc waits
                    // vmcnt(0) while an LDS-DMA is in flight, i.e. for the NEXT chunk's staging, which defeats the
                    // double buffer.  This chunk's image is complete: the barrier that ended the previous iteration
                    // came after every wave's vmcnt(0).
                    {
                        const unsigned vbase = lds_addr(vb);
                        bf16x8 vf[2][2];   // [db][ss]
                        unsigned a[8];
#pragma unroll
                        for (int db = 0; db < 2; ++db)
#pragma unroll
                            for (int ss = 0; ss < 2; ++ss) {
                                const int klo = sub * 32 + 16 * ss + 4 * h2 + qq;
                                const int col = db * 32 + 16 * (gi & 1) + 4 * pp;
                                a[(db * 2 + ss) * 2] = vbase + off128(klo, col);
                                a[(db * 2 + ss) * 2 + 1] = vbase + off128(klo + 8, col);
                            }
                        tr_read8(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
#pragma unroll
                        for (int db = 0; db < 2; ++db)
#pragma unroll
                            for (int ss = 0; ss < 2; ++ss) o[db] = MFMA32(vf[db][ss], pf[ss], o[db]);
                    }
                }
            }
        }
        __syncthreads();
    }

    l += __shfl_xor(l, 32, 64);
    const int q = q0 + r;
    // (the chunk loop ended with a barrier: the K/V buffers are free; each wave takes 4 KiB of them as its stage)
    if (q0 < T) {
        const float inv = sb / l;
        char* blk = smem + wave * 4096;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = db * 32 + 8 * g4 + 4 * h2;
                stage_row_piece(blk, r, d0, pack4bf(o[db][4 * g4] * inv, o[db][4 * g4 + 1] * inv,
                                                    o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv));
            }
        if (h2 == 0 && q < T) p.lse[((size_t)b * H + h) * T + q] = (m + __builtin_amdgcn_logf(l)) * LN2;
        // (same wave wrote and reads: DS operations of a wave execute in order)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 8 + (lane >> 3), ch = lane & 7;
            const uint4 v = stage_row_chunk(blk, row, ch);
            if (q0 + row < T)
                store_u4<kNtAttn>(p.out + ((size_t)b * T + q0 + row) * (H * HD) + h * HD + ch * 8, v);
        }
        if (p.out_lo != nullptr) {
            // what the bf16 rounding of O dropped: the backward forms delta = rowsum(dO * O) from out + out_lo
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = db * 32 + 8 * g4 + 4 * h2;
                    const float v0 = o[db][4 * g4] * inv, v1 = o[db][4 * g4 + 1] * inv, v2 = o[db][4 * g4 + 2] * inv,
                                v3 = o[db][4 * g4 + 3] * inv;
                    const uint2 hi = pack4bf(v0, v1, v2, v3);
                    stage_row_piece(blk, r, d0, pack4bf(v0 - bf2f(hi.x & 0xffff), v1 - bf2f(hi.x >> 16),
                                                        v2 - bf2f(hi.y & 0xffff), v3 - bf2f(hi.y >> 16)));
                }
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3), ch = lane & 7;
                const uint4 v = stage_row_chunk(blk, row, ch);
                if (q0 + row < T)
                    store_u4<kNtAttn>(p.out_lo + ((size_t)b * T + q0 + row) * (H * HD) + h * HD + ch * 8, v);
            }
        }
    }
}

// Diagnostic build only (-DGETMI_ATTN_STAMPS, tools/attn_stamps.py): s_memtime stamps of every wave of the backward
// kernel at its phase boundaries, read back through getmi_debug_attn_stamps.  The product library has none of this.
#ifdef GETMI_ATTN_STAMPS
#define STAMP_WG 4096
__device__ unsigned long long g_attn_stamps[STAMP_WG * 8 * 32];
#define STAMP(i)                                                                                          \
    do {                                                                                                  \
        if (lane == 0 && blockIdx.x < STAMP_WG)                                                           \
            g_attn_stamps[((size_t)blockIdx.x * 8 + wave) * 32 + (i)] = __builtin_amdgcn_s_memtime();     \
    } while (0)
extern "C" int getmi_debug_attn_stamps(void* dst, size_t bytes) {
    if (bytes > sizeof(g_attn_stamps)) bytes = sizeof(g_attn_stamps);
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define STAMP(i) do {} while (0)
#endif

// One level of a reduce-scatter butterfly over lane bit M (M = 32, 16, 8, 4, 2, 1): lanes with the bit clear end with
// lo + lo(partner), lanes with it set with hi + hi(partner); partner = lane ^ M.  v_permlane16_swap exchanges the odd
// 16-lane rows of its first operand with the even rows of its second, which is exactly the M = 16 exchange; below that the
// partner is a DPP row rotation / quad permutation.  (ds_bpermute shuffles made this reduction 2 k cycles per workgroup.)
template <int M>
__device__ __forceinline__ float bfly_level(float lo, float hi, bool up) {
    if constexpr (M == 32) {         // v_permlane32_swap: upper half of the first operand <-> lower half of the second
        typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
        const u2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (M == 16) {
        typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
        const u2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        // lanes with the bit clear read lane + M, lanes with it set read lane - M (row_ror:n reads lane - n mod 16)
        constexpr int CLR = M == 8 ? 0x128 : M == 4 ? 0x12C : M == 2 ? 0x4E : 0xB1;
        constexpr int SET = M == 8 ? 0x128 : M == 4 ? 0x124 : M == 2 ? 0x4E : 0xB1;
        const float a = lo + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(lo), CLR, 0xf, 0xf, true));
        const float b = hi + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(hi), SET, 0xf, 0xf, true));
        return up ? b : a;
    }
}
template <int M>
__device__ __forceinline__ void bfly_reduce(float (&v)[32], int lane) {
    const bool up = (lane & M) != 0;
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = bfly_level<M>(v[j], v[j + M], up);
}

struct AttnBwdArgs {
    const bf16_t* qkv;
    const bf16_t* out;    // [B,T,H*64] forward output (for delta = rowsum(dO * O))
    const bf16_t* out_lo; // optional: the forward's rounding residual of `out`.  delta from the bf16 output alone breaks
                          // sum_key dS[q][key] = 0 by dO.(O - bf16(O)) per row, which is harmless everywhere except in the
                          // q_bias gradient (a pure cancellation sum): 8-23 % error there in a 12-layer model vs 0.6 %
                          // with the residual (profiles/r02_qbias_error.txt)
    const bf16_t* dout;   // [B,T,H*64]
    const float* lse;     // [B,H,T]
    bf16_t* dqkv;         // [B,T,3,H,64]
    float* dq_acc;        // [nkb][B,T,H,64] fp32 partial dQ per 256-key block, only when T > 256
    float* bias_partial;  // [2][B*nkb][H*64] fp32: per-workgroup column sums of dQ (k=0) and dV (k=1), or null
    int B, T, H;
    float scale;
    const float* sample_scale;   // optional [B]: `out` (and out_lo) are sample_scale[b] * O and dout is the gradient w.r.t. THAT
                                 // tensor (see AttnFwdArgs).  With s = sample_scale[b]: dO = s * dout, so delta = rowsum(dout * out)
                                 // needs no factor, dP = s * dout.V^T; the kernels form P o (dout.V^T - delta / s) and multiply
                                 // dQ, dK, dV and the bias sums by s at the end.  s == 0: the sample is skipped (zeros).
    int skew;             // diagnostic (GETMI_ATTN_SKEW): first-round workgroups start up to skew x 2048 cycles late.
                          // Measured: no effect at any spread (profiles/r02_attn_bwd_skew.txt), so the prologue is not an HBM
                          // burst of synchronised workgroups but the per-CU fetch limit (~10 B/clk from HBM)
};

// A sample whose DropPath factor is 0 (AttnBwdArgs::sample_scale): every gradient of this (b, h, key block) is zero.
__device__ __forceinline__ void attn_bwd_zero(const AttnBwdArgs& p, int b, int h, int kbi, int nkb, int key_base, int key_count,
                                              int tid) {
    const int T = p.T, HH = p.H * HD;
    const size_t rs = (size_t)3 * HH;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    const float4 zf = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < T * 8; i += 512) {
        const int q = i >> 3, ch = i & 7;
        if (p.dq_acc == nullptr) {
            store_u4<kNtAttn>(p.dqkv + ((size_t)b * T + q) * rs + h * HD + ch * 8, z);
        } else {
            float* d = p.dq_acc + ((size_t)kbi * p.B * T + (size_t)b * T + q) * HH + h * HD + ch * 8;
            store_f4<kNtAttn>(d, zf);
            store_f4<kNtAttn>(d + 4, zf);
        }
    }
    const int nkeys = max(0, min(key_count, T - key_base));
    for (int i = tid; i < nkeys * 16; i += 512) {
        const int which = i / (nkeys * 8), j = i - which * nkeys * 8;
        const int key = key_base + (j >> 3), ch = j & 7;
        store_u4<kNtAttn>(p.dqkv + ((size_t)b * T + key) * rs + (size_t)(1 + which) * HH + h * HD + ch * 8, z);
    }
    if (p.bias_partial != nullptr && tid < 128)
        p.bias_partial[((size_t)(tid >> 6) * p.B * nkb + (size_t)b * nkb + kbi) * HH + h * HD + (tid & 63)] = 0.f;
}

// LDS map (bytes).  Everything a workgroup reads is staged ONCE per 256-query chunk (for T <= 256: once), so no
// LDS-DMA is in flight inside the slice loop: hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS read that may
// alias an in-flight global_load_lds, which in the earlier per-slice double-buffered version exposed the full
// global latency seven times per workgroup (measured 299 us -> see DESIGN.md).
#define BW_KIMG 0                 // [256 keys][128 B]
#define BW_QIMG 32768             // [256 q][128 B]
#define BW_DOIMG 65536            // [256 q][128 B]
#define BW_DS 98304               // 2 x [256 keys][32 q] bf16 (64-B rows, halves swapped by (key>>3)&1)
#define BW_ST 131072              // lse'[256] | delta[256] fp32
#define BW_LDS (131072 + 2048)

__global__ __launch_bounds__(512) void attn_bwd_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, H = p.H;
    const int nkb = (T + 255) >> 8;
    const int bh = blockIdx.x / nkb, kbi = blockIdx.x - bh * nkb;
    const int b = bh / H, h = bh - b * H;
    const size_t rs = (size_t)3 * H * HD;
    const bf16_t* base = p.qkv + (size_t)b * T * rs + h * HD;
    const bf16_t* dobase = p.dout + (size_t)b * T * (H * HD) + h * HD;
    const bf16_t* obase = p.out + (size_t)b * T * (H * HD) + h * HD;
    const float* lse = p.lse + ((size_t)b * H + h) * T;
    const int key_base = kbi * 256;
    const int r = lane & 31, h2 = lane >> 5;
    const int gi = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int mykey = key_base + wave * 32 + r;          // this lane's key (column of S, dP)
    const bool wave_active = (key_base + wave * 32) < T;  // wave-uniform
    const bool wave_ragged = (key_base + wave * 32 + 32) > T;
    const float c2 = p.scale * LOG2E;
    const float inv_scale = 1.0f / p.scale;
    const int nks = min(8, (T - key_base + 31) >> 5);     // 32-key groups of this workgroup that hold a valid key
    const float sb = p.sample_scale ? p.sample_scale[b] : 1.0f;      // workgroup-uniform (see AttnBwdArgs)
    if (sb == 0.0f) {
        attn_bwd_zero(p, b, h, kbi, nkb, key_base, 256, tid);
        return;
    }
    const float osc = p.scale * sb, inv_sb = 1.0f / sb;
    STAMP(0);
    if (p.skew > 0 && blockIdx.x < 256) {
        const int n = (int)(((blockIdx.x * 97u) & 255u) * (unsigned)p.skew >> 8);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
    }

    // ---- K image: 256 rows = 32 wave-instructions, 4 per wave (rows past T clamp to T-1: finite, and their dS is 0) ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int I = wave * 4 + j;
        if (I * 8 < nks * 32) {
            const int row = I * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz128(row);
            const int key = min(key_base + row, T - 1);
            glds16_attn(base + (size_t)key * rs + H * HD + chunk * 8, smem + BW_KIMG + I * 1024);
        }
    }
    // ---- V fragments of this wave's 32 keys (second operand of dP = dO.V^T): V[key r][16kk + 8h2 ..] ----
    bf16x8 vf[4];
    {
        const int key = min(mykey, T - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            vf[kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)key * rs + 2 * H * HD + kk * 16 + h2 * 8);
    }

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
    bf16x8 kfr[4];   // this wave's K rows as the B operand of S = Q.K^T           (slice-invariant, kept in VGPRs)
    bf16x8 ktf[8];   // K^T fragments of all 256 keys for this wave's dQ d-block   (slice-invariant, kept in VGPRs)
    float dssum = 0.f;   // running sum over queries of this lane's fp32 dS / scale column (q_bias gradient, see the end)
    float* stf = reinterpret_cast<float*>(smem + BW_ST);
    int dsbuf = 0;
    STAMP(1);

    for (int qc = 0; qc < T; qc += 256) {
        if (qc) __syncthreads();   // every wave has left the previous chunk's images
        const int nrow = min(256, ((T - qc + 31) >> 5) << 5);   // rows of this chunk the slices will read
        // ---- Q chunk: waves 0-3, dO chunk: waves 4-7; 8 rows (1 KiB) per wave-instruction, 8 per wave ----
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int II = (wave & 3) * 8 + j;
            if (II * 8 < nrow) {
                const int row = II * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ swz128(row);
                const int q = min(qc + row, T - 1);
                const bf16_t* src = (wave < 4) ? (base + (size_t)q * rs + chunk * 8)
                                               : (dobase + (size_t)q * (H * HD) + chunk * 8);
                glds16_attn(src, smem + (wave < 4 ? BW_QIMG : BW_DOIMG) + II * 1024);
            }
        }
        // ---- lse' = lse / scale (+inf past T -> P = 0) and delta = rowsum(dO * O): two threads per query ----
        {
            const int qi = qc + (tid >> 1);
            float acc = 0.f;
            if (qi < T) {
                const uint4* po = reinterpret_cast<const uint4*>(obase + (size_t)qi * (H * HD) + (tid & 1) * 32);
                const uint4* pd = reinterpret_cast<const uint4*>(dobase + (size_t)qi * (H * HD) + (tid & 1) * 32);
                if (p.out_lo != nullptr) {       // workgroup-uniform
                    const uint4* pl = reinterpret_cast<const uint4*>(p.out_lo + ((size_t)b * T + qi) * (H * HD) + h * HD +
                                                                      (tid & 1) * 32);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint4 a = po[k], d = pd[k], e = pl[k];
                        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, dw[4] = {d.x, d.y, d.z, d.w}, ew[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
                        for (int w = 0; w < 4; ++w)
                            acc += (bf2f(aw[w] & 0xffff) + bf2f(ew[w] & 0xffff)) * bf2f(dw[w] & 0xffff) +
                                   (bf2f(aw[w] >> 16) + bf2f(ew[w] >> 16)) * bf2f(dw[w] >> 16);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint4 a = po[k], d = pd[k];
                        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                        for (int w = 0; w < 4; ++w)
                            acc += bf2f(aw[w] & 0xffff) * bf2f(dw[w] & 0xffff) + bf2f(aw[w] >> 16) * bf2f(dw[w] >> 16);
                    }
                }
            }
            acc += __shfl_xor(acc, 1, 64);
            // stored negated: they are the initial values of the S' / dP' accumulators
            if (tid & 1) stf[tid >> 1] = (qi < T) ? -lse[qi] * inv_scale : -INFINITY;
            else stf[256 + (tid >> 1)] = -acc * inv_sb;
        }
        STAMP(2);
        __syncthreads();
        STAMP(3);
        if (qc == 0) {
            const int krow = wave * 32 + r;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                kfr[kk] = *reinterpret_cast<const bf16x8*>(smem + BW_KIMG + krow * 128 +
                                                           (((2 * kk + h2) ^ swz128(krow)) << 4));
            const int kcol = (wave >> 1) * 16 + 4 * pp;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int klo = ks * 32 + 8 * gi + qq;
                ktf[ks] = join_tr_a(lds_tr16a(smem + BW_KIMG + off128(klo, kcol)),
                                    lds_tr16a(smem + BW_KIMG + off128(klo + 4, kcol)));
            }
        }
        STAMP(4);
        const int nsl = nrow >> 5;
        for (int sl = 0; sl < nsl; ++sl) {
            const char* qimg = smem + BW_QIMG + sl * 4096;
            const char* doimg = smem + BW_DOIMG + sl * 4096;
            const float* st = stf + sl * 32;
            char* dsimg = smem + BW_DS + dsbuf * 16384;
            if (wave_active) {
                // accumulator rows are queries: q(reg) = (reg&3) + 8*(reg>>2) + 4*h2
                f32x16 s, dp;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 a = *reinterpret_cast<const float4*>(st + 8 * g4 + 4 * h2);
                    const float4 d = *reinterpret_cast<const float4*>(st + 256 + 8 * g4 + 4 * h2);
                    s[4 * g4] = a.x; s[4 * g4 + 1] = a.y; s[4 * g4 + 2] = a.z; s[4 * g4 + 3] = a.w;
                    dp[4 * g4] = d.x; dp[4 * g4 + 1] = d.y; dp[4 * g4 + 2] = d.z; dp[4 * g4 + 3] = d.w;
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int ch = ((2 * kk + h2) ^ swz128(r)) << 4;
                    const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(qimg + r * 128 + ch);
                    const bf16x8 dofr = *reinterpret_cast<const bf16x8*>(doimg + r * 128 + ch);
                    s = MFMA32(qfr, kfr[kk], s);      // S'[q][key] = Q.K^T - lse/scale
                    dp = MFMA32(dofr, vf[kk], dp);    // dP'[q][key] = dO.V^T - delta
                }
                // (VALU-lean: the softmax scale of dS is applied once to the dK / dQ results, not per element)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s[i] = __builtin_amdgcn_exp2f(s[i] * c2);   // P
                    dp[i] = s[i] * dp[i];                       // dS / scale
                }
                if (wave_ragged) {   // wave-uniform: only the wave that straddles T has padded keys
                    const bool kvalid = mykey < T;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        s[i] = kvalid ? s[i] : 0.f;
                        dp[i] = kvalid ? dp[i] : 0.f;
                    }
                }
                // column sums of dS for the q_bias gradient, taken HERE from the fp32 values: d(q_bias) = sum_q dQ[q] =
                // scale * sum_key (sum_q dS[q][key]) K[key] is a cancellation sum (every row of dS sums to ~0), and summing
                // the dQ tiles instead, which are built from the bf16-rounded dS image, leaves it at that rounding's floor
                dssum += (((dp[0] + dp[1]) + (dp[2] + dp[3])) + ((dp[4] + dp[5]) + (dp[6] + dp[7]))) +
                         (((dp[8] + dp[9]) + (dp[10] + dp[11])) + ((dp[12] + dp[13]) + (dp[14] + dp[15])));
                bf16x8 pf[2], dsf[2];
                pf[0] = pack8(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
                pf[1] = pack8(s[8], s[9], s[10], s[11], s[12], s[13], s[14], s[15]);
                dsf[0] = pack8(dp[0], dp[1], dp[2], dp[3], dp[4], dp[5], dp[6], dp[7]);
                dsf[1] = pack8(dp[8], dp[9], dp[10], dp[11], dp[12], dp[13], dp[14], dp[15]);
                // dS^T image: row = key (wave*32 + r), 32 q per row (64 B); 32-B halves swapped when (key>>3)&1
                {
                    const int krow = wave * 32 + r;
                    char* row = dsimg + krow * 64;
                    const int hs = (krow >> 3) & 1;
                    const uint2* d2 = reinterpret_cast<const uint2*>(dsf);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int q = 8 * g4 + 4 * h2;   // first of 4 consecutive query rows
                        const int byte = ((((q >> 4) ^ hs) << 5) | ((q & 15) << 1));
                        *reinterpret_cast<uint2*>(row + byte) = d2[g4];
                    }
                }
                // dV^T[d][key] += dO^T[d][q].P[q][key] ; dK^T[d][key] += Q^T[d][q].dS[q][key]
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int ss = 0; ss < 2; ++ss) {
                        const int qlo = 16 * ss + 4 * h2 + qq;
                        const int col = db * 32 + 16 * (gi & 1) + 4 * pp;
                        const bf16x8 dot = join_tr_a(lds_tr16a(doimg + off128(qlo, col)),
                                                     lds_tr16a(doimg + off128(qlo + 8, col)));
                        const bf16x8 qt = join_tr_a(lds_tr16a(qimg + off128(qlo, col)),
                                                    lds_tr16a(qimg + off128(qlo + 8, col)));
                        dv[db] = MFMA32(dot, pf[ss], dv[db]);
                        dk[db] = MFMA32(qt, dsf[ss], dk[db]);
                    }
            }
            if (qc == 0 && sl < 7) STAMP(5 + 3 * sl);
            // one barrier per slice: the dS image is double buffered, so a wave that runs ahead into slice sl+1
            // writes the other buffer, and cannot reach slice sl+2 before everyone has passed slice sl+1's barrier.
            __syncthreads();
            if (qc == 0 && sl < 7) STAMP(6 + 3 * sl);
            // ---- dQ^T tile [16 d][16 q] per wave: d-block = wave>>1, q-half = wave&1; sum over all valid keys ----
            {
                const int qh = wave & 1, dblk = wave >> 1;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks < nks) {
                        const int klo = ks * 32 + 8 * gi + qq;
                        const int khi = klo + 4;
                        const int qcol = 4 * pp;   // within the 16-query half
                        const int blo = klo * 64 + ((((qh ^ ((klo >> 3) & 1)) << 5)) | (qcol << 1));
                        const int bhi = khi * 64 + ((((qh ^ ((khi >> 3) & 1)) << 5)) | (qcol << 1));
                        const bf16x8 dst = join_tr_a(lds_tr16a(dsimg + blo), lds_tr16a(dsimg + bhi));
                        acc = MFMA16(ktf[ks], dst, acc);   // D[d][q] = sum_key K[key][d] dS[q][key]
                    }
                }
                const int q = qc + sl * 32 + qh * 16 + (lane & 15);
                const int d0 = dblk * 16 + (lane >> 4) * 4;
                if (q < T) {
                    if (p.dq_acc == nullptr) {
                        store_u2<kNtAttn && (GETMI_NT_ATTN_DQ != 0)>(p.dqkv + ((size_t)b * T + q) * rs + h * HD + d0, pack4bf(acc[0] * osc, acc[1] * osc, acc[2] * osc, acc[3] * osc));
                    } else {
                        // long context: this key block's partial dQ goes to its own fp32 slab (plain 16-byte
                        // stores; fp32 atomics measured 2.5 ms per call at T = 2001), summed by attn_dq_convert
                        float* dst = p.dq_acc + ((size_t)kbi * p.B * T + (size_t)b * T + q) * (H * HD) + h * HD + d0;
                        store_f4<kNtAttn>(dst, make_float4(acc[0] * osc, acc[1] * osc, acc[2] * osc, acc[3] * osc));
                    }
                }
            }
            if (qc == 0 && sl < 7) STAMP(7 + 3 * sl);
            dsbuf ^= 1;
        }
    }

    STAMP(26);
    if (p.bias_partial != nullptr) {
        // ---- q_bias / v_bias gradients = column sums of dQ / dV over all tokens: this workgroup's share, from the
        //      fp32 accumulators (the q_bias gradient is a cancellation sum; summing the bf16-rounded dqkv loses it).
        //      dV: lane = key, registers = d -> reduce over the 32 lanes of each half wave; columns of padded keys
        //      are exactly 0.  dQ: lane&15 = query -> reduce over 16 lanes.  Then across waves through LDS.
        __syncthreads();                                   // every wave is done with the dS images: reuse them
        float* red = reinterpret_cast<float*>(smem + BW_DS);          // [8 waves][64] dV | [8 waves][64] dQ at +512
        {
            // butterfly reduce-scatter over the 32 lanes of a half wave: 16+8+4+2+1 exchanges instead of 32 x 5, each a
            // lane-row swap or DPP add (bfly_level);
            // lane r ends with the sum over keys of accumulator register r (= db*16 + i)
            float v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = dv[k >> 4][k & 15];
            bfly_reduce<16>(v, lane);
            bfly_reduce<8>(v, lane);
            bfly_reduce<4>(v, lane);
            bfly_reduce<2>(v, lane);
            bfly_reduce<1>(v, lane);
            const int i = r & 15;
            red[wave * 64 + (r >> 4) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2] = v[0];
        }
        {
            // dQ: partial[d] = sum over this wave's 32 keys of (sum_q dS[q][key]) * K[key][d], d = lane.  The K image is
            // still whole (the dK / dV staging below overwrites it only after the next barrier); rows of waves past T
            // were never staged, hence the wave_active guard (their column sums are 0 anyway).
            // lane = (key quad kg = lane>>3, 8-column chunk ch = lane&7): four 16-byte K pieces times the column sums of
            // their keys, then a reduce-scatter over the eight key quads (lane bits 5, 4, 3) leaves one column per lane.
            // (One key per iteration with a 2-byte LDS read each was 1.5 k cycles per workgroup.)
            const float cs = dssum + __shfl_xor(dssum, 32, 64);      // both query halves of key r, in every lane r, r + 32
            float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (wave_active) {
                const int kg = lane >> 3, ch = lane & 7;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float c = __shfl(cs, kg * 4 + i, 64);
                    const int krow = wave * 32 + kg * 4 + i;
                    const uint4 kv = *reinterpret_cast<const uint4*>(smem + BW_KIMG + krow * 128 + ((ch ^ swz128(krow)) << 4));
                    const uint32_t kw[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        a8[2 * w] = fmaf(c, bf2f(kw[w] & 0xffff), a8[2 * w]);
                        a8[2 * w + 1] = fmaf(c, bf2f(kw[w] >> 16), a8[2 * w + 1]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) a8[j] = bfly_level<32>(a8[j], a8[j + 4], (lane & 32) != 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) a8[j] = bfly_level<16>(a8[j], a8[j + 2], (lane & 16) != 0);
            a8[0] = bfly_level<8>(a8[0], a8[1], (lane & 8) != 0);
            red[512 + wave * 64 + (lane & 7) * 8 + ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1)] = a8[0];
        }
        __syncthreads();
        if (tid < 128) {
            const int d = tid & 63;
            float v;
            if (tid < 64) {
                v = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += red[512 + w * 64 + d];
                v *= osc;
            } else {
                v = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += red[w * 64 + d];
                v *= sb;
            }
            const size_t R = (size_t)p.B * nkb;
            p.bias_partial[((size_t)(tid >> 6) * R + (size_t)b * nkb + kbi) * (H * HD) + h * HD + d] = v;
        }
    }
    STAMP(27);
    if (wave_active) {
        // every wave is past the last slice's barrier, after which nobody reads the K / Q / dO images: wave w takes
        // 8 KiB of them as its stage: rows 0-31 = dK of its keys, rows 32-63 = dV
        char* blk = smem + BW_KIMG + wave * 8192;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = db * 32 + 8 * g4 + 4 * h2;
                stage_row_piece(blk, r, d0, pack4bf(dk[db][4 * g4] * osc, dk[db][4 * g4 + 1] * osc,
                                                    dk[db][4 * g4 + 2] * osc, dk[db][4 * g4 + 3] * osc));
                stage_row_piece(blk, 32 + r, d0, pack4bf(dv[db][4 * g4] * sb, dv[db][4 * g4 + 1] * sb, dv[db][4 * g4 + 2] * sb,
                                                         dv[db][4 * g4 + 3] * sb));
            }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 8 + (lane >> 3), ch = lane & 7;
            const uint4 v = stage_row_chunk(blk, row, ch);
            const int key = key_base + wave * 32 + (row & 31);
            if (key < T)
                store_u4<kNtAttn>(p.dqkv + ((size_t)b * T + key) * rs + (size_t)(1 + (row >> 5)) * H * HD + h * HD +
                                          ch * 8, v);
        }
    }
    STAMP(28);
}

// ------------------------------------------------------------------------------------------------------------------
// Streamed backward (one key block per (b, h) for T <= 224, key blocks of 224 keys above).  Same arithmetic, accumulator
// layouts and images as attn_bwd_kernel; what changes is WHEN the operands arrive and WHO fetches them.  attn_bwd_kernel
// stages everything (K, V, all of Q, dO, and the O / O_lo rows for delta: ~155 KB per workgroup) before its first slice,
// and at the ~10 B/clk one CU gets while its single resident workgroup waits, that prologue was 22 k of the kernel's 54 k
// cycles.  Here only K, V and the first slice are waited for.  Wave 7 owns no keys and is the WORKER: it requests the
// Q / dO / O / O_lo / lse rows of slice X (16 + 1 LDS-DMA instructions) three slices ahead and forms the slice's
// constants (-lse/scale, -delta) one slice ahead; waves 0-6 own 32 keys each and run nothing but slice arithmetic:
//     worker, slice s    : request slice s+3 (ring slots: Q / dO / constants / lse = slice & 7, O / O_lo = slice % 3);
//                          constants of slice s+1 from the LDS copies (visible since barrier s-1);
//                          s_waitcnt vmcnt(17) -> slice s+2 has landed (loads return in order); barrier s publishes it
//     key owners, slice s: S', dP' (the constants enter through one extra MFMA K step each), P, dS, dV^T, dK^T; barrier s
//     all eight waves    : one 16 x 16 dQ^T tile of slice s each
// Every LDS access while LDS-DMA is in flight is inline asm (hipcc guards plain LDS reads AND writes with vmcnt(0)), the
// barriers are raw s_barrier behind lgkmcnt(0), V comes by asm global loads fused with their wait, and the two code paths
// share no lane-derived registers (notes at the respective places: each of them was a measured stall, a wrong result or
// a fault).  Cycle stamps: profiles/r02_attn_bwd_stream_stamps.txt.
#define BS_STAT 131072            // [256 q][16 B]: -lse/scale and -delta of query q as three bf16 terms each (see bs_split3)
#define BS_OSTG (131072 + 4096)   // 3 slots x (O rows 4 KiB | O_lo rows 4 KiB) of one slice
#define BS_LSE (BS_OSTG + 3 * 8192)   // 8 slots x 64 fp32: the lse rows of a slice (+ the next 32, unused), by dword LDS-DMA
#define BS_LDS (BS_LSE + 2048)

// LDS reads of the streamed kernel in inline asm (see above).  b128 / b64 reads with a compile-time byte offset.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ void bs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// (hipcc guards plain LDS WRITES the same way, so the few stores inside the streamed section are asm as well)
__device__ __forceinline__ void bs_write_b64(unsigned addr, u32x2_t v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void bs_write_b128(unsigned addr, u32x4_t v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// x as the sum of three bf16 numbers (truncations, so every remainder is exact: 24 mantissa bits in all), packed as the
// first three elements of an MFMA operand row.  The per-query constants -lse/scale and -delta enter the S' and dP'
// accumulators through ONE extra MFMA each against a row of ones, instead of eight 16-byte LDS reads per wave and slice
// that initialised the accumulators (a third of the kernel's ds_read_b128 traffic; the LDS pipe is its busiest unit).
__device__ __forceinline__ u32x2_t bs_split3(float x) {
    const unsigned h = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(h);
    const unsigned l = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(l);
    return (u32x2_t){(h >> 16) | l, __float_as_uint(r2) >> 16};
}
__global__ __launch_bounds__(512) void attn_bwd_stream_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, H = p.H;
    // key blocks of 224 keys (seven key-owning waves + the worker) when T > 224; every block sweeps all query slices
    const int nkb = T <= 224 ? 1 : (T + 223) / 224;
    const int bh = blockIdx.x / nkb, kbi = blockIdx.x - bh * nkb;
    const int b = bh / H, h = bh - b * H;
    const int key_base = kbi * 224;
    const size_t rs = (size_t)3 * H * HD;
    const int HH = H * HD;
    const bf16_t* base = p.qkv + (size_t)b * T * rs + h * HD;
    const bool has_lo = p.out_lo != nullptr;
    const float* lse = p.lse + ((size_t)b * H + h) * T;
    const int r = lane & 31, h2 = lane >> 5;
    const int gi = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int mykey = key_base + wave * 32 + r;
    const bool wave_active = wave < 7 && (key_base + wave * 32) < T;       // wave 7 is the worker: it never owns keys
    const bool wave_ragged = (key_base + wave * 32 + 32) > T;
    const float c2 = p.scale * LOG2E;
    const float inv_scale = 1.0f / p.scale;
    const int nks = min(7, (T - key_base + 31) >> 5);      // 32-key groups of this block with a valid key
    const int nsl = (T + 31) >> 5;                           // 32-query slices
    const float sb = p.sample_scale ? p.sample_scale[b] : 1.0f;      // workgroup-uniform (see AttnBwdArgs)
    if (sb == 0.0f) {
        attn_bwd_zero(p, b, h, kbi, nkb, key_base, 224, tid);
        return;
    }
    const float osc = p.scale * sb, inv_sb = 1.0f / sb;
    // Q / dO images and the constants table are rings of 8 slices (slot = slice & 7; for T <= 224 every slice has its own
    // slot), the O / O_lo stage a ring of 3: slice X is requested during slice X-3 and last read during slice X.
    const unsigned smem0 = lds_addr(smem);
    STAMP(0);

    // ---- K image (rows past T clamp to T-1) ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int I = wave * 4 + j;
        if (I * 8 < nks * 32) {
            const int row = I * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz128(row);
            const int key = min(key_base + row, T - 1);
            glds16_attn(base + (size_t)key * rs + HH + chunk * 8, smem + BW_KIMG + I * 1024);
        }
    }
    STAMP(1);

    f32x16 dk[2], dv[2];     // zeroed at the START of the key-owners' path and at the END of the worker's (which has none)
    float dssum;

    // K^T fragments of all keys for this wave's dQ d-block (slice-invariant, kept in VGPRs; every wave incl. the worker
    // computes a dQ tile).  Key rows ks*32 + 8gi + qq (+4): the swizzle of a row depends on (row >> 1) & 7 only, which
    // ks*32 does not change -> one lane address each for the lower / upper quad, ks * 4096 as the instruction offset.
    auto load_ktf = [&](bf16x8 (&ktf)[8], int ln) {      // ln: the caller's copy of the lane id (see the worker path)
        const int gi = ln >> 4, qq = (ln >> 2) & 3, pp = ln & 3;
        const int kcol = (wave >> 1) * 16 + 4 * pp;
        const unsigned kl = smem0 + BW_KIMG + off128(8 * gi + qq, kcol), kh = smem0 + BW_KIMG + off128(8 * gi + qq + 4, kcol);
        tr64_t t0[8], t1[8];
        tr_issue8ab<0, 0, 4096, 4096, 8192, 8192, 12288, 12288>(kl, kh, t0);
        tr_issue8ab<16384, 16384, 20480, 20480, 24576, 24576, 28672, 28672>(kl, kh, t1);
        tr_wait16(t0, t1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            ktf[ks] = join_tr64(t0[2 * ks], t0[2 * ks + 1]);
            ktf[4 + ks] = join_tr64(t1[2 * ks], t1[2 * ks + 1]);
        }
    };
    // ---- dQ^T tile [16 d][16 q] per wave: d-block = wave>>1, q-half = wave&1; sum over all valid keys.  dS^T reads: all
    //      16 of a slice are one lane address + ks * 2048 (+256 for the upper key quad) ----
    auto dq_tile = [&](const bf16x8 (&ktf)[8], int sl, int dsbuf, int ln) {
        const int gi = ln >> 4, qq = (ln >> 2) & 3, pp = ln & 3;
        const int qh = wave & 1, dblk = wave >> 1, klo = 8 * gi + qq;
        // this lane's piece = 8-byte slot (qh*4 + pp) of key row klo (and klo + 4), slots XORed with (key >> 1) & 7 (see the
        // writer); (klo + 4) >> 1 differs from klo >> 1 by bit 1 only (klo & 7 <= 3)
        const unsigned dsx = (unsigned)(((qh * 4 + pp) ^ ((klo >> 1) & 7)) << 3);
        const unsigned dsa = smem0 + BW_DS + dsbuf * 16384 + (unsigned)(klo * 64) + dsx;
        const unsigned dsb = smem0 + BW_DS + dsbuf * 16384 + (unsigned)(klo * 64) + (dsx ^ 16u);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        tr64_t ta[8], tb[8];
        tr_issue8ab<0, 256, 2048, 2048 + 256, 4096, 4096 + 256, 6144, 6144 + 256>(dsa, dsb, ta);
        tr_issue8ab<8192, 8192 + 256, 10240, 10240 + 256, 12288, 12288 + 256, 14336, 14336 + 256>(dsa, dsb, tb);
        tr_wait16(ta, tb);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            if (ks < nks) acc = MFMA16(ktf[ks], join_tr64(ta[2 * ks], ta[2 * ks + 1]), acc);
#pragma unroll
        for (int ks = 4; ks < 8; ++ks)
            if (ks < nks) acc = MFMA16(ktf[ks], join_tr64(tb[2 * (ks - 4)], tb[2 * (ks - 4) + 1]), acc);
        const int q = sl * 32 + qh * 16 + (ln & 15);
        const int d0 = dblk * 16 + (ln >> 4) * 4;
        if (q < T) {
            if (p.dq_acc == nullptr)
                store_u2<kNtAttn && (GETMI_NT_ATTN_DQ != 0)>(p.dqkv + ((size_t)b * T + q) * rs + h * HD + d0, pack4bf(acc[0] * osc, acc[1] * osc, acc[2] * osc, acc[3] * osc));
            else      // several key blocks: this block's partial dQ goes to its own fp32 slab, summed by attn_dq_convert
                store_f4<kNtAttn>(p.dq_acc + ((size_t)kbi * p.B * T + (size_t)b * T + q) * HH + h * HD + d0, make_float4(acc[0] * osc, acc[1] * osc, acc[2] * osc, acc[3] * osc));
        }
    };

    if (wave == 7) {
        // ======== worker wave.  Wave 7 never owns keys (T <= 224, or key blocks of 224); it requests every slice (Q, dO, O, O_lo
        // rows: 16 LDS-DMA instructions of 8 rows each, 12 without O_lo, + 1 for the lse rows) and forms delta, so that the seven key-owning waves
        // run nothing but the slice arithmetic.  (Spread over all waves, the requests and delta cost each of them ~800
        // cycles per slice -- profiles/r02_attn_bwd_stream_stamps.txt -- which ate the prologue's gain.)  It is a code
        // path of its own so that its addresses and the key-owners' accumulators never share the register file.
        // An opaque copy of the lane id: everything lane-derived in this path is recomputed from it, so that no value of the
        // common code (which hipcc had spilled around the key-owners' loop) is reloaded here -- a scratch reload while
        // LDS-DMA is in flight costs a vmcnt(0).
        int wl;
        asm volatile("v_mov_b32 %0, %1" : "=v"(wl) : "v"(lane));
        const char* w_base[4] = {reinterpret_cast<const char*>(base),
                                 reinterpret_cast<const char*>(p.dout + (size_t)b * T * HH + h * HD),
                                 reinterpret_cast<const char*>(p.out + (size_t)b * T * HH + h * HD),
                                 reinterpret_cast<const char*>((has_lo ? p.out_lo : p.out) + (size_t)b * T * HH + h * HD)};
        const int w_row = wl >> 3;                               // row within an 8-row instruction
        const unsigned w_ch[2] = {(unsigned)(((wl & 7) ^ swz128(w_row)) * 16), (unsigned)(((wl & 7) ^ swz128(w_row + 8)) * 16)};
        const int n_arr = has_lo ? 4 : 3;
        auto issue_slice = [&](int X, int slot) {
    #pragma unroll
            for (int part = 0; part < 4; ++part) {
                const unsigned q = (unsigned)min(X * 32 + part * 8 + w_row, T - 1);
                const unsigned oq = q * (unsigned)(rs * 2) + w_ch[part & 1], oo = q * (unsigned)(HH * 2) + w_ch[part & 1];
    #pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a < n_arr) {
                        char* dst = smem + (a == 0 ? BW_QIMG + (X & 7) * 4096 : a == 1 ? BW_DOIMG + (X & 7) * 4096
                                                                                       : BS_OSTG + slot * 8192 + (a - 2) * 4096);
                        glds16_attn(w_base[a] + (a == 0 ? oq : oo), dst + part * 1024);
                    }
                }
            }
            // + the lse rows of the slice, one dword per lane (lanes 32-63 fetch the next slice's: unused)
            __builtin_amdgcn_global_load_lds(GLB_PTR(lse + min(X * 32 + wl, T - 1)), LDS_PTR(smem + BS_LSE + (X & 7) * 256), 4, 0, 0);
        };
        // constants of slice X: -lse/scale and -delta.  delta: wl = (query wl>>1, half wl&1), 32 elements each
        const int d_lq = wl >> 1;
        auto delta_slice = [&](int X, int slot) {
            const unsigned row = smem0 + d_lq * 128, od = BW_DOIMG + (X & 7) * 4096, oo = BS_OSTG + slot * 8192;
            unsigned ad[4];
    #pragma unroll
            for (int k = 0; k < 4; ++k) ad[k] = row + ((((wl & 1) * 4 + k) ^ swz128(d_lq)) << 4);
            u32x4_t d[4], a[4], e[4];
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
                         "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"
                         : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
                         : "v"(ad[0] + od), "v"(ad[1] + od), "v"(ad[2] + od), "v"(ad[3] + od),
                           "v"(ad[0] + oo), "v"(ad[1] + oo), "v"(ad[2] + oo), "v"(ad[3] + oo)
                         : "memory");
            // (the O_lo pieces are read unconditionally -- stale LDS without O_lo -- and dropped after the wait: defining them in
            //  two arms of an if would let hipcc merge the arms with moves in front of the wait, see the note at the V fragments)
            float lv;
            asm volatile("ds_read_b128 %0, %5 offset:4096\n\tds_read_b128 %1, %6 offset:4096\n\t"
                         "ds_read_b128 %2, %7 offset:4096\n\tds_read_b128 %3, %8 offset:4096\n\tds_read_b32 %4, %9"
                         : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "=&v"(lv)
                         : "v"(ad[0] + oo), "v"(ad[1] + oo), "v"(ad[2] + oo), "v"(ad[3] + oo),
                           "v"(smem0 + BS_LSE + (X & 7) * 256 + d_lq * 4)
                         : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]),
                           "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(lv)
                         :: "memory");
            if (!has_lo) {
    #pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = (u32x4_t){0u, 0u, 0u, 0u};
            }
            float acc = 0.f;
    #pragma unroll
            for (int k = 0; k < 4; ++k)
    #pragma unroll
                for (int w = 0; w < 4; ++w)
                    acc += (bf2f(a[k][w] & 0xffff) + bf2f(e[k][w] & 0xffff)) * bf2f(d[k][w] & 0xffff) +
                           (bf2f(a[k][w] >> 16) + bf2f(e[k][w] >> 16)) * bf2f(d[k][w] >> 16);
            acc += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc), 0xB1, 0xf, 0xf, true));   // + the other half
            if ((wl & 1) == 0) {      // negated: they enter S' and dP'.  Rows past T: a finite stand-in for -inf, so that the
                                      // split stays finite and P underflows to exactly 0
                const u32x2_t cl = bs_split3(X * 32 + d_lq < T ? -lv * inv_scale : -60000.f), cd = bs_split3(-acc * inv_sb);
                bs_write_b128(smem0 + BS_STAT + ((X & 7) * 32 + d_lq) * 16, (u32x4_t){cl[0], cl[1], cd[0], cd[1]});
            }
        };
        auto wait_all_but_one_slice = [&]() {      // loads return in order: at most one slice's requests outstanding
            if (has_lo) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        };
        issue_slice(0, 0);      // right behind this wave's K share
        if (nsl > 1) issue_slice(1, 1);
        if (nsl > 2) issue_slice(2, 2);
        STAMP(2);
        if (nsl > 2) {          // slice 0 landed: slices 1, 2 may be outstanding
            if (has_lo) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        delta_slice(0, 0);      // from the worker's own requests: no barrier needed, and it runs while the others wait for K / V
        bs_barrier();           // P1: K image and slice 0 visible
        STAMP(3);
        bf16x8 ktf[8];
        load_ktf(ktf, wl);
        if (nsl > 2) wait_all_but_one_slice();      // slice 1 landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bs_barrier();           // P2: constants of slice 0, slice 1 visible
        STAMP(4);
        int dsbuf = 0, slot_next = 0;      // ring slot of slice sl + 3 (== slot of slice sl); slice sl + 1 is slot_next + 1
        for (int sl = 0; sl < nsl; ++sl) {
            if (sl + 3 < nsl) issue_slice(sl + 3, slot_next);
            if (sl + 1 < nsl) delta_slice(sl + 1, slot_next == 2 ? 0 : slot_next + 1);
            if (sl < 7) STAMP(5 + 3 * sl);
            // slice sl+2 has landed when only slice sl+3's requests may be outstanding (this wave's dQ stores can only
            // lower the count)
            if (sl + 3 < nsl) wait_all_but_one_slice();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bs_barrier();
            if (sl < 7) STAMP(6 + 3 * sl);
            dq_tile(ktf, sl, dsbuf, wl);
            if (sl < 7) STAMP(7 + 3 * sl);
            dsbuf ^= 1;
            slot_next = slot_next == 2 ? 0 : slot_next + 1;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
        dssum = 0.f;
    } else {
        // ======== key-owning waves (and idle ones when T <= 192: they only take part in the barriers and dQ tiles) ========
        // V fragments of this wave's 32 keys: plain global loads, but in asm together with their wait --
        // hipcc waits vmcnt(0) at the first use of a compiler-visible load while LDS-DMA requests are outstanding, and a
        // register with an asm load in flight must not be visible to it as free (an earlier version issued these loads in
        // the common code: the worker never named the registers, hipcc reused them for its address arithmetic, the landing
        // loads overwrote those, and the kernel faulted).  The wait also covers this wave's K share.
        u32x4_t vraw[4];
        {
            const bf16_t* vp = base + (size_t)min(mykey, T - 1) * rs + 2 * HH + h2 * 8;
            asm volatile("global_load_dwordx4 %0, %4, off" ATTN_V_POL "\n\tglobal_load_dwordx4 %1, %4, off offset:32" ATTN_V_POL "\n\t"
                         "global_load_dwordx4 %2, %4, off offset:64" ATTN_V_POL "\n\tglobal_load_dwordx4 %3, %4, off offset:96" ATTN_V_POL "\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(vraw[0]), "=&v"(vraw[1]), "=&v"(vraw[2]), "=&v"(vraw[3])
                         : "v"(vp)
                         : "memory");
        }
        STAMP(2);
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
        dssum = 0.f;
        bf16x8 vf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) vf[kk] = __builtin_bit_cast(bf16x8, vraw[kk]);
        bs_barrier();           // P1
        STAMP(3);
        bf16x8 ktf[8];
        load_ktf(ktf, lane);
        bs_barrier();           // P2
        // lane addresses of the in-loop reads.  Transposed Q / dO: rows qlo (lo) and qlo + 8 (hi) of d block db; the second
        // 16-query half is +2048 bytes and the dO image +32768 bytes from the Q image, both as instruction offsets.  Row
        // fragments of K / Q / dO: one address per 16-element K step (the chunk swizzle is lane dependent).
        unsigned tq[2][2], fk[4];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int qlo = 4 * h2 + qq, col = db * 32 + 16 * (gi & 1) + 4 * pp;
            tq[db][0] = smem0 + BW_QIMG + (unsigned)off128(qlo, col);
            tq[db][1] = smem0 + BW_QIMG + (unsigned)off128(qlo + 8, col);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)      // this lane's K row piece; the Q piece of slice sl is (32768 + (sl - wave) * 4096) above it
            fk[kk] = smem0 + BW_KIMG + (wave * 32 + r) * 128 + (((2 * kk + h2) ^ swz128(r)) << 4);
        bf16x8 kfr[4];      // this wave's K rows as the B operand of S = Q.K^T (slice-invariant)
        {
            u32x4_t k0, k1, k2, k3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3) : "v"(fk[0]), "v"(fk[1]), "v"(fk[2]), "v"(fk[3]) : "memory");
            kfr[0] = __builtin_bit_cast(bf16x8, k0); kfr[1] = __builtin_bit_cast(bf16x8, k1);
            kfr[2] = __builtin_bit_cast(bf16x8, k2); kfr[3] = __builtin_bit_cast(bf16x8, k3);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fk[kk] += 32768 - wave * 4096;      // now the Q piece of slice 0
        const unsigned st_a = smem0 + BS_STAT + r * 16;
        const unsigned hmask = h2 ? 0u : 0xffffffffu;      // the constants ride on k = 0..2 of the extra K step: lanes of half 0
        // dS^T image of the streamed kernel: row = key (64 B = eight 8-byte slots of 4 queries), slot index XOR (key >> 1) & 7.
        // A ds_write_b64 is serviced in groups of 16 consecutive lanes (= 16 consecutive keys here) over 32 banks = two
        // 64-byte rows: the 8 keys of one parity must hit 8 different slots.  (Halves swapped by (key >> 3) & 1, the staged
        // kernel's layout, was a 4-way conflict on every write: 29 % of this kernel's LDS cycles.)
        const unsigned ds_s0 = (unsigned)(h2 ^ (((wave * 32 + r) >> 1) & 7));
        const unsigned ds_wa = smem0 + BW_DS + (wave * 32 + r) * 64 + (ds_s0 << 3);            // queries 4h2..      (slot h2)
        const unsigned ds_wb = smem0 + BW_DS + (wave * 32 + r) * 64 + ((ds_s0 ^ 4u) << 3);     // queries 16 + 4h2.. (slot 4 + h2)
        const int ds_d2 = (ds_s0 & 2u) ? -16 : 16;                                             // to slot ^ 2: queries + 8
        STAMP(4);
        int dsbuf = 0;
        for (int sl = 0; sl < nsl; ++sl) {
            if (wave_active) {
                // accumulator rows are queries: q(reg) = (reg&3) + 8*(reg>>2) + 4*h2.  This lane's query constants and the
                // row fragments of Q / dO: 9 reads in issue order, consumed under counted lgkmcnt waits
                u32x4_t stq, fa[4], fd[4];
                {
                    const unsigned sa = st_a + (sl & 7) * 512, so = (sl & 7) * 4096;
                    asm volatile(
                        "ds_read_b128 %0, %9\n\t"
                        "ds_read_b128 %1, %10\n\tds_read_b128 %2, %10 offset:32768\n\tds_read_b128 %3, %11\n\t"
                        "ds_read_b128 %4, %11 offset:32768\n\tds_read_b128 %5, %12\n\tds_read_b128 %6, %12 offset:32768\n\t"
                        "ds_read_b128 %7, %13\n\tds_read_b128 %8, %13 offset:32768"
                        : "=&v"(stq), "=&v"(fa[0]), "=&v"(fd[0]), "=&v"(fa[1]), "=&v"(fd[1]), "=&v"(fa[2]), "=&v"(fd[2]), "=&v"(fa[3]),
                          "=&v"(fd[3])
                        : "v"(sa), "v"(fk[0] + so), "v"(fk[1] + so), "v"(fk[2] + so), "v"(fk[3] + so)
                        : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(stq) :: "memory");
                f32x16 s, dp;
                {
                    const u32x4_t al = {stq[0] & hmask, stq[1] & hmask, 0u, 0u}, ad = {stq[2] & hmask, stq[3] & hmask, 0u, 0u};
                    const u32x4_t ones = {0x3F803F80u & hmask, 0x00003F80u & hmask, 0u, 0u};
                    f32x16 zero;
#pragma unroll
                    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
                    s = MFMA32(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, ones), zero);    // -lse/scale
                    dp = MFMA32(__builtin_bit_cast(bf16x8, ad), __builtin_bit_cast(bf16x8, ones), zero);   // -delta
                }
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fa[0]), "+v"(fd[0]) :: "memory");
                s = MFMA32(__builtin_bit_cast(bf16x8, fa[0]), kfr[0], s);      // S' = Q.K^T - lse/scale
                dp = MFMA32(__builtin_bit_cast(bf16x8, fd[0]), vf[0], dp);     // dP' = dO.V^T - delta
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[1]), "+v"(fd[1]) :: "memory");
                s = MFMA32(__builtin_bit_cast(bf16x8, fa[1]), kfr[1], s);
                dp = MFMA32(__builtin_bit_cast(bf16x8, fd[1]), vf[1], dp);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[2]), "+v"(fd[2]) :: "memory");
                s = MFMA32(__builtin_bit_cast(bf16x8, fa[2]), kfr[2], s);
                dp = MFMA32(__builtin_bit_cast(bf16x8, fd[2]), vf[2], dp);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[3]), "+v"(fd[3]) :: "memory");
                s = MFMA32(__builtin_bit_cast(bf16x8, fa[3]), kfr[3], s);
                dp = MFMA32(__builtin_bit_cast(bf16x8, fd[3]), vf[3], dp);
                // transposed Q / dO fragments for dK / dV: requested here, consumed after the softmax arithmetic
                tr64_t ta[8], tb[8];
                {
                    const unsigned so = (sl & 7) * 4096;
                    // per d block, order: dO lo, dO hi, Q lo, Q hi of query half 0, then the same of half 1 (+2048)
                    tr_issue8ab<32768, 32768, 0, 0, 32768 + 2048, 32768 + 2048, 2048, 2048>(tq[0][0] + so, tq[0][1] + so, ta);
                    tr_issue8ab<32768, 32768, 0, 0, 32768 + 2048, 32768 + 2048, 2048, 2048>(tq[1][0] + so, tq[1][1] + so, tb);
                }
                // (VALU-lean: the softmax scale of dS is applied once to the dK / dQ results, not per element)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s[i] = __builtin_amdgcn_exp2f(s[i] * c2);   // P
                    dp[i] = s[i] * dp[i];                       // dS / scale
                }
                if (wave_ragged) {
                    const bool kvalid = mykey < T;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        s[i] = kvalid ? s[i] : 0.f;
                        dp[i] = kvalid ? dp[i] : 0.f;
                    }
                }
                dssum += (((dp[0] + dp[1]) + (dp[2] + dp[3])) + ((dp[4] + dp[5]) + (dp[6] + dp[7]))) +
                         (((dp[8] + dp[9]) + (dp[10] + dp[11])) + ((dp[12] + dp[13]) + (dp[14] + dp[15])));
                bf16x8 pf[2], dsf[2];
                pf[0] = pack8(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
                pf[1] = pack8(s[8], s[9], s[10], s[11], s[12], s[13], s[14], s[15]);
                dsf[0] = pack8(dp[0], dp[1], dp[2], dp[3], dp[4], dp[5], dp[6], dp[7]);
                dsf[1] = pack8(dp[8], dp[9], dp[10], dp[11], dp[12], dp[13], dp[14], dp[15]);
                {
                    // dS^T image (layout at ds_wa): this lane's 4 pieces of 4 queries (4h2.., 8 + 4h2.., 16 + 4h2.., 24 + 4h2..)
                    const u32x2_t* d2 = reinterpret_cast<const u32x2_t*>(dsf);
                    const unsigned o = dsbuf * 16384;
                    bs_write_b64(ds_wa + o, d2[0]);
                    bs_write_b64(ds_wa + o + ds_d2, d2[1]);
                    bs_write_b64(ds_wb + o, d2[2]);
                    bs_write_b64(ds_wb + o + ds_d2, d2[3]);
                }
                tr_wait16(ta, tb);
                // dV^T[d][key] += dO^T[d][q].P[q][key] ; dK^T[d][key] += Q^T[d][q].dS[q][key]
                dv[0] = MFMA32(join_tr64(ta[0], ta[1]), pf[0], dv[0]);
                dk[0] = MFMA32(join_tr64(ta[2], ta[3]), dsf[0], dk[0]);
                dv[0] = MFMA32(join_tr64(ta[4], ta[5]), pf[1], dv[0]);
                dk[0] = MFMA32(join_tr64(ta[6], ta[7]), dsf[1], dk[0]);
                dv[1] = MFMA32(join_tr64(tb[0], tb[1]), pf[0], dv[1]);
                dk[1] = MFMA32(join_tr64(tb[2], tb[3]), dsf[0], dk[1]);
                dv[1] = MFMA32(join_tr64(tb[4], tb[5]), pf[1], dv[1]);
                dk[1] = MFMA32(join_tr64(tb[6], tb[7]), dsf[1], dk[1]);
            }
            if (sl < 7) STAMP(5 + 3 * sl);
            bs_barrier();
            if (sl < 7) STAMP(6 + 3 * sl);
            dq_tile(ktf, sl, dsbuf, lane);
            if (sl < 7) STAMP(7 + 3 * sl);
            dsbuf ^= 1;
        }
    }

    STAMP(26);
    if (wave_active) {
        // dK / dV rows FIRST (round 3): their stores drain while the bias sums below go through two barriers.  Staged in the Q / dO
        // ring area (free since the last slice's barrier; 7 x 8 KiB), because the bias sums still read the K image.
        char* blk = smem + BW_QIMG + wave * 8192;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = db * 32 + 8 * g4 + 4 * h2;
                stage_row_piece(blk, r, d0, pack4bf(dk[db][4 * g4] * osc, dk[db][4 * g4 + 1] * osc,
                                                    dk[db][4 * g4 + 2] * osc, dk[db][4 * g4 + 3] * osc));
                stage_row_piece(blk, 32 + r, d0, pack4bf(dv[db][4 * g4] * sb, dv[db][4 * g4 + 1] * sb, dv[db][4 * g4 + 2] * sb,
                                                         dv[db][4 * g4 + 3] * sb));
            }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 8 + (lane >> 3), ch = lane & 7;
            const uint4 v = stage_row_chunk(blk, row, ch);
            const int key = key_base + wave * 32 + (row & 31);
            if (key < T)
                store_u4<kNtAttn>(p.dqkv + ((size_t)b * T + key) * rs + (size_t)(1 + (row >> 5)) * HH + h * HD + ch * 8, v);
        }
    }
    STAMP(27);
    if (p.bias_partial != nullptr) {
        // as in attn_bwd_kernel (no LDS-DMA is in flight any more: plain LDS accesses)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem + BW_DS);
        {
            float v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = dv[k >> 4][k & 15];
            bfly_reduce<16>(v, lane);
            bfly_reduce<8>(v, lane);
            bfly_reduce<4>(v, lane);
            bfly_reduce<2>(v, lane);
            bfly_reduce<1>(v, lane);
            const int i = r & 15;
            red[wave * 64 + (r >> 4) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2] = v[0];
        }
        {
            // lane = (key quad kg = lane>>3, 8-column chunk ch = lane&7): four 16-byte K pieces times the column sums of
            // their keys, then a reduce-scatter over the eight key quads (lane bits 5, 4, 3) leaves one column per lane.
            // (One key per iteration with a 2-byte LDS read each was 1.5 k cycles per workgroup.)
            const float cs = dssum + __shfl_xor(dssum, 32, 64);      // both query halves of key r, in every lane r, r + 32
            float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (wave_active) {
                const int kg = lane >> 3, ch = lane & 7;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float c = __shfl(cs, kg * 4 + i, 64);
                    const int krow = wave * 32 + kg * 4 + i;
                    const uint4 kv = *reinterpret_cast<const uint4*>(smem + BW_KIMG + krow * 128 + ((ch ^ swz128(krow)) << 4));
                    const uint32_t kw[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        a8[2 * w] = fmaf(c, bf2f(kw[w] & 0xffff), a8[2 * w]);
                        a8[2 * w + 1] = fmaf(c, bf2f(kw[w] >> 16), a8[2 * w + 1]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) a8[j] = bfly_level<32>(a8[j], a8[j + 4], (lane & 32) != 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) a8[j] = bfly_level<16>(a8[j], a8[j + 2], (lane & 16) != 0);
            a8[0] = bfly_level<8>(a8[0], a8[1], (lane & 8) != 0);
            red[512 + wave * 64 + (lane & 7) * 8 + ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1)] = a8[0];
        }
        __syncthreads();
        if (tid < 128) {
            const int d = tid & 63;
            float v = 0.f;
            if (tid < 64) {
#pragma unroll
                for (int w = 0; w < 8; ++w) v += red[512 + w * 64 + d];
                v *= osc;
            } else {
#pragma unroll
                for (int w = 0; w < 8; ++w) v += red[w * 64 + d];
                v *= sb;
            }
            p.bias_partial[((size_t)(tid >> 6) * p.B * nkb + (size_t)b * nkb + kbi) * HH + h * HD + d] = v;
        }
    }
    STAMP(28);
}

// dqkv[b,t,0,h,:] = bf16(sum over key blocks of dq_acc[kb][b,t,h,:])   (long-context path only)
__global__ void attn_dq_convert_kernel(const float* __restrict__ acc, bf16_t* __restrict__ dqkv, size_t ntok,
                                       int H, int nkb) {
    const size_t n4 = ntok * H * (HD / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t tok = i / (H * (HD / 4));
        const int rem = (int)(i - tok * (H * (HD / 4)));
        float4 v = *reinterpret_cast<const float4*>(acc + i * 4);
        for (int kb = 1; kb < nkb; ++kb) {
            const float4 w = *reinterpret_cast<const float4*>(acc + ((size_t)kb * n4 + i) * 4);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        store_u2<kNtAttn>(dqkv + tok * (size_t)(3 * H * HD) + rem * 4, pack4bf(v.x, v.y, v.z, v.w));
    }
}

// ------------------------------------------------------------------------------------------------
int getmi_attn_fwd_impl(const bf16_t* qkv, bf16_t* out, bf16_t* out_lo, float* lse, int B, int T, int H, float scale,
                        const float* sample_scale, hipStream_t st) {
    GETMI_CHECK_ARG(B > 0 && T > 0 && H > 0, "attn_fwd: bad shape B=%d T=%d H=%d", B, T, H);
    GETMI_CHECK_ARG(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0), "attn_fwd: pointers must be 16-byte aligned");
    GETMI_CHECK_ARG((uintptr_t)out_lo % 16 == 0, "attn_fwd: out_lo must be 16-byte aligned");
    AttnFwdArgs a{qkv, out, out_lo, lse, B, T, H, scale, sample_scale};
    // waves (= 32-query groups) per workgroup: 8 (256 queries: K/V of a (b, h) staged once for T <= 256, one staging
    // instruction per wave and chunk) unless 4 already cover T.  Measured at T = 201: 104 us vs 111 (4 waves) and 118
    // (7 waves: the staging no longer divides evenly); T = 2001: 366 vs 375.  GETMI_ATTN_FWD_WAVES overrides (4..8)
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("GETMI_ATTN_FWD_WAVES");
        forced = e ? atoi(e) : 0;
    }
    int nw = T > 128 ? 8 : 4;
    if (forced >= 4 && forced <= 8) nw = forced;
    if (nw < 4) nw = 4;
    if (nw > 8) nw = 8;
    const int nqb = (T + 32 * nw - 1) / (32 * nw);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H * nqb), dim3(64 * nw), 0, st, a);
    GETMI_CHECK_LAUNCH("attn_fwd");
    return GETMI_OK;
}

// GETMI_ATTN_STREAM=0 keeps the staged kernel (key blocks of 256); default: the streamed one (one block for T <= 224, else
// blocks of 224 keys: its eighth wave is the staging worker)
static bool attn_bwd_streamed() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GETMI_ATTN_STREAM");
        v = e ? (atoi(e) != 0) : 1;
    }
    return v != 0;
}
static int attn_bwd_nkb(int T) { return attn_bwd_streamed() ? (T <= 224 ? 1 : (T + 223) / 224) : (T + 255) / 256; }
static size_t attn_bwd_slab_bytes(int B, int T, int H) {
    const int nkb = attn_bwd_nkb(T);
    const size_t n = nkb > 1 ? (size_t)nkb * B * T * H * HD * sizeof(float) : 0;   // per-key-block dQ slabs
    return (n + 255) & ~(size_t)255;
}
size_t getmi_attn_bwd_workspace_bytes(int B, int T, int H) {
    // + per-workgroup column sums of dQ and dV (bias gradients): [2][B * key blocks][H*64] fp32
    return attn_bwd_slab_bytes(B, T, H) + (size_t)2 * B * attn_bwd_nkb(T) * H * HD * sizeof(float);
}
int getmi_colreduce_impl(const float* partial, float* out0, float* out1, float* out2, int R, int C, int accumulate,
                         hipStream_t st);

int getmi_attn_bwd_impl(const bf16_t* qkv, const bf16_t* out, const bf16_t* out_lo, const bf16_t* dout, const float* lse, bf16_t* dqkv,
                        float* dq_bias, float* dv_bias, int B, int T, int H, float scale, const float* sample_scale,
                        void* workspace, size_t workspace_bytes, hipStream_t st) {
    GETMI_CHECK_ARG(B > 0 && T > 0 && H > 0, "attn_bwd: bad shape B=%d T=%d H=%d", B, T, H);
    GETMI_CHECK_ARG(workspace_bytes >= getmi_attn_bwd_workspace_bytes(B, T, H), "attn_bwd: workspace too small");
    GETMI_CHECK_ARG(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)dqkv % 16 == 0) &&
                        ((uintptr_t)out % 16 == 0) && ((uintptr_t)out_lo % 16 == 0) && ((uintptr_t)workspace % 256 == 0),
                    "attn_bwd: pointers must be 16-byte aligned (workspace 256)");
    const int nkb = attn_bwd_nkb(T);
    float* dq_acc = nkb > 1 ? reinterpret_cast<float*>(workspace) : nullptr;   // every slab element is written by its key block
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_stream_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BS_LDS);
        attr_set = true;
    }
    const bool want_bias = dq_bias != nullptr || dv_bias != nullptr;
    float* bias_partial = want_bias ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + attn_bwd_slab_bytes(B, T, H))
                                    : nullptr;
    static int skew_env = -2;
    if (skew_env == -2) {
        const char* e = getenv("GETMI_ATTN_SKEW");
        skew_env = e ? atoi(e) : -1;
    }
    AttnBwdArgs a{qkv, out, out_lo, dout, lse, dqkv, dq_acc, bias_partial, B, T, H, scale, sample_scale,
                  skew_env > 0 ? skew_env : 0};
    if (attn_bwd_streamed())
        hipLaunchKernelGGL(attn_bwd_stream_kernel, dim3(B * H * nkb), dim3(512), BS_LDS, st, a);
    else
        hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * H * nkb), dim3(512), BW_LDS, st, a);
    GETMI_CHECK_LAUNCH("attn_bwd");
    if (dq_acc) {
        const size_t n4 = (size_t)B * T * H * (HD / 4);
        int grid = (int)((n4 + 255) / 256);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(attn_dq_convert_kernel, dim3(grid), dim3(256), 0, st, dq_acc, dqkv, (size_t)B * T, H, nkb);
        GETMI_CHECK_LAUNCH("attn_dq_convert");
    }
    if (want_bias) {
        // every (row, column) of both partial matrices was written by exactly one workgroup
        const int R = B * nkb, C = H * HD;
        const float* pq = bias_partial;
        if (dq_bias && dv_bias) return getmi_colreduce_impl(pq, dq_bias, dv_bias, nullptr, R, C, 0, st);
        if (dq_bias) return getmi_colreduce_impl(pq, dq_bias, nullptr, nullptr, R, C, 0, st);
        return getmi_colreduce_impl(pq + (size_t)R * C, dv_bias, nullptr, nullptr, R, C, 0, st);
    }
    return GETMI_OK;
}
