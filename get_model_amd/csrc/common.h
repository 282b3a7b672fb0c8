// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the GET region hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef uint16_t bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __bfloat16_as_ushort(__float2bfloat16(f)); }
// two floats -> packed bf16 pair (RNE) in ONE v_cvt_pk_bf16_f32; converting them one at a time costs four
// instructions per pair (two half-used v_cvt_pk + shift + or), which showed in every bf16 epilogue.
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint2 pack4bf(float a, float b, float c, float d) {
    return make_uint2(pack2bf(a, b), pack2bf(c, d));
}

// 16-byte asynchronous global -> LDS copy (global_load_lds_dwordx4). LDS destination = lds_wave_base +
// lane*16 (wave-uniform base), global source is per lane.
#ifndef GETMI_GLDS_AUX
#define GETMI_GLDS_AUX 0        // cache-policy bits of the LDS-DMA loads (gfx940+: 1 = sc0, 2 = nt, 16 = sc1); experiment switch
#endif
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, GETMI_GLDS_AUX);
}
// ... and of the attention kernels' LDS-DMA loads (every byte is read once per kernel there)
#ifndef GETMI_GLDS_AUX_ATTN
#define GETMI_GLDS_AUX_ATTN 2   // nt: 32.77 -> 32.66 ms/step (three interleaved rounds, one box); sc1 (16): no change
#endif
__device__ __forceinline__ void glds16_attn(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, GETMI_GLDS_AUX_ATTN);
}

// Streaming ("nt") stores / loads for data the storing kernel does not read again: the line is not kept in the XCD's L2
// ahead of the operands that ARE reused there (measured on the persistent GEMM epilogue: -0.45 ms per C2 step).
// Elsewhere it LOSES, because the next kernel finds the data on chip when it was stored normally (same A/B, ms per step):
// GEMM epilogue only 35.01; + LayerNorm loads/stores + attention outputs + wgrad slabs 35.50; without the LayerNorm part
// 35.32, without the attention part 35.41, without the slab part 35.11.  So these three stayed OFF in round 2.
// Round 3 (bf16 gradient stream, folded DropPath: different neighbours for every tensor), same kind of A/B on one box:
// attention outputs streamed 33.07 -> 32.86 (ON now); LayerNorm 32.78 -> 32.75 (noise, off); slabs 33.07 -> 33.27 (off).
#ifndef GETMI_NT_LN
#define GETMI_NT_LN 0
#endif
#ifndef GETMI_NT_ATTN
#define GETMI_NT_ATTN 1
#endif
#ifndef GETMI_NT_SLAB
#define GETMI_NT_SLAB 0
#endif
typedef __attribute__((ext_vector_type(4))) float nt_f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned nt_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned nt_u32x4;
template <bool NT> __device__ __forceinline__ void store_f4(float* p, float4 v) {
    if constexpr (NT) __builtin_nontemporal_store((nt_f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f32x4*>(p));
    else *reinterpret_cast<float4*>(p) = v;
}
template <bool NT> __device__ __forceinline__ void store_u2(void* p, uint2 v) {
    if constexpr (NT) __builtin_nontemporal_store((nt_u32x2){v.x, v.y}, reinterpret_cast<nt_u32x2*>(p));
    else *reinterpret_cast<uint2*>(p) = v;
}
template <bool NT> __device__ __forceinline__ void store_u4(void* p, uint4 v) {
    if constexpr (NT) __builtin_nontemporal_store((nt_u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u32x4*>(p));
    else *reinterpret_cast<uint4*>(p) = v;
}
template <bool NT> __device__ __forceinline__ float4 load_f4(const float* p) {
    if constexpr (NT) {
        const nt_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    } else return *reinterpret_cast<const float4*>(p);
}
template <bool NT> __device__ __forceinline__ uint4 load_u4(const void* p) {
    if constexpr (NT) {
        const nt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(p));
        return make_uint4(v[0], v[1], v[2], v[3]);
    } else return *reinterpret_cast<const uint4*>(p);
}
template <bool NT> __device__ __forceinline__ uint2 load_u2(const void* p) {
    if constexpr (NT) {
        const nt_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x2*>(p));
        return make_uint2(v[0], v[1]);
    } else return *reinterpret_cast<const uint2*>(p);
}

// Bijective XCD-aware remap of a 1-D block id: blocks that share an XCD (bid % 8 equal) get a
// contiguous range of logical ids, so neighbouring tiles hit the same per-XCD L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact (erf) GELU and its derivative, fp32.  Phi(x) = 0.5*erfc(-x/sqrt2) through Abramowitz-Stegun 7.1.26
// (|abs error of erf| <= 1.5e-7, far below the bf16 rounding of the stored result): one v_rcp + one v_exp.
__device__ __forceinline__ void gelu_parts(float x, float& Phi, float& pdf) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
    // pdf(x) = exp(-x^2/2)/sqrt(2 pi) straight out of the exponential (log2 of the constant rides in the fma); the
    // polynomial's coefficients carry 0.5 (of Phi) * sqrt(2 pi) (to undo that constant): tail = poly * t * pdf
    pdf = __builtin_amdgcn_exp2f(fmaf(x * x, -0.72134752044448170f, -1.3257480647361593f));
    constexpr float C = 0.5f * 2.5066282746310002f;
    float poly = fmaf(C * 1.061405429f, t, C * -1.453152027f);
    poly = fmaf(poly, t, C * 1.421413741f);
    poly = fmaf(poly, t, C * -0.284496736f);
    poly = fmaf(poly, t, C * 0.254829592f);
    // 0.5*erfc(|x|/sqrt2) = poly*t*pdf =: tail in (0, 0.5];  Phi = x >= 0 ? 1 - tail : tail = 0.5 + copysign(0.5 - tail, x):
    // one v_bfi instead of compare + select + subtract per element (the fc1 epilogue is VALU-issue bound: round 3)
    const float h = fmaf(-(poly * t), pdf, 0.5f);
    Phi = 0.5f + __builtin_copysignf(h, x);
}
__device__ __forceinline__ float gelu_f(float x) {
    float Phi, pdf;
    gelu_parts(x, Phi, pdf);
    return x * Phi;
}
__device__ __forceinline__ float dgelu_f(float x) {
    float Phi, pdf;
    gelu_parts(x, Phi, pdf);
    return fmaf(x, pdf, Phi);
}

// ---- transposed LDS reads (ds_read_b64_tr_b16) in inline asm ------------------------------------------------
// hipcc emits `s_waitcnt vmcnt(0)` in front of the __builtin_amdgcn_ds_read_tr16_b64 intrinsic whenever an
// LDS-DMA (global_load_lds) is in flight -- it cannot prove the read does not alias the DMA target -- which
// drains the whole operand prefetch every phase (measured: wgrad waves 50 % in s_waitcnt).  The asm form is
// invisible to that pass; ordering against the DMA is done by the kernels' own counted vmcnt + barrier.
// Eight reads (= four 8-element MFMA fragments) and their lgkmcnt wait in ONE statement, early-clobber outputs.
typedef __attribute__((ext_vector_type(2))) unsigned tr64_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)((__attribute__((address_space(3))) const char*)p);
}
__device__ __forceinline__ bf16x8 join_tr64(tr64_t lo, tr64_t hi) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
    const u4_t v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void tr_read8(unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned a4, unsigned a5,
                                         unsigned a6, unsigned a7, bf16x8& f0, bf16x8& f1, bf16x8& f2, bf16x8& f3) {
    tr64_t r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\t"
        "ds_read_b64_tr_b16 %3, %11\n\tds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\t"
        "ds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)
        : "memory");
    f0 = join_tr64(r0, r1); f1 = join_tr64(r2, r3); f2 = join_tr64(r4, r5); f3 = join_tr64(r6, r7);
}
// issue-only form: 8 reads, NO wait; the destinations are undefined until tr_wait16 / tr_wait24 (which take them
// as read-write operands, so the compiler cannot touch them earlier: cdna_hip_programming.md 5.7 form (ii)).
__device__ __forceinline__ void tr_issue8(unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned a4, unsigned a5,
                                          unsigned a6, unsigned a7, tr64_t (&r)[8]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\t"
        "ds_read_b64_tr_b16 %3, %11\n\tds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\t"
        "ds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)
        : "memory");
}
// the same with a compile-time byte offset in the instruction (no address arithmetic per read)
template <int OFF>
__device__ __forceinline__ void tr_issue8o(const unsigned (&a)[8], tr64_t (&r)[8]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%16\n\tds_read_b64_tr_b16 %1, %9 offset:%16\n\t"
        "ds_read_b64_tr_b16 %2, %10 offset:%16\n\tds_read_b64_tr_b16 %3, %11 offset:%16\n\t"
        "ds_read_b64_tr_b16 %4, %12 offset:%16\n\tds_read_b64_tr_b16 %5, %13 offset:%16\n\t"
        "ds_read_b64_tr_b16 %6, %14 offset:%16\n\tds_read_b64_tr_b16 %7, %15 offset:%16"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "n"(OFF)
        : "memory");
}
// eight reads from TWO lane addresses (even reads from a, odd reads from b) with a compile-time byte offset each
template <int O0, int O1, int O2, int O3, int O4, int O5, int O6, int O7>
__device__ __forceinline__ void tr_issue8ab(unsigned a, unsigned b, tr64_t (&r)[8]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%10\n\tds_read_b64_tr_b16 %1, %9 offset:%11\n\t"
        "ds_read_b64_tr_b16 %2, %8 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:%14\n\tds_read_b64_tr_b16 %5, %9 offset:%15\n\t"
        "ds_read_b64_tr_b16 %6, %8 offset:%16\n\tds_read_b64_tr_b16 %7, %9 offset:%17"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(a), "v"(b), "n"(O0), "n"(O1), "n"(O2), "n"(O3), "n"(O4), "n"(O5), "n"(O6), "n"(O7)
        : "memory");
}
__device__ __forceinline__ void tr_wait8(tr64_t (&a)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                 :
                 : "memory");
}
__device__ __forceinline__ void tr_wait16(tr64_t (&a)[8], tr64_t (&b)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
                 :
                 : "memory");
}
__device__ __forceinline__ void tr_wait24(tr64_t (&a)[8], tr64_t (&b)[8], tr64_t (&c)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                 :
                 : "memory");
}

// two fragments (4 reads)
__device__ __forceinline__ void tr_read4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, bf16x8& f0, bf16x8& f1) {
    tr64_t r0, r1, r2, r3;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\tds_read_b64_tr_b16 %2, %6\n\t"
        "ds_read_b64_tr_b16 %3, %7\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
        : "memory");
    f0 = join_tr64(r0, r1); f1 = join_tr64(r2, r3);
}

// ---- host side error channel (thread-local message, C-ABI returns an int code) ----
void getmi_set_error(const char* fmt, ...);
#define GETMI_OK 0
#define GETMI_ERR_ARG 1
#define GETMI_ERR_LAUNCH 2

#define GETMI_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            getmi_set_error(__VA_ARGS__);          \
            return GETMI_ERR_ARG;                  \
        }                                          \
    } while (0)

#define GETMI_CHECK_LAUNCH(name)                                                         \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            getmi_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
            return GETMI_ERR_LAUNCH;                                                     \
        }                                                                                \
    } while (0)
