// Fused multi-tensor AdamW for the GET region models: ONE launch updates every parameter of the
// model and, for the GEMM weights, also emits the bf16 operand copies the next forward/backward consume
// ([N,K] and the transposed [K,N]) -- the reference runs torch.optim.AdamW per parameter group (optim.py:129-130)
// and autocast re-casts the weights in every forward.
//
// HBM-bound: 16 B read (p, g, m, v) + 12 B written (+ 4 B of bf16 copies) per element.  Work is described by a
// device-resident job table (built once by the host side, 64 B per job):
//   flat job : `count` consecutive elements from `start` of one tensor (chunks of 4096; count % 4 == 0 except the
//              tensor's last chunk), optional [N,K] bf16 copy;
//   tile job : one 64 x 64 tile of a 2-D weight whose dimensions are multiples of 64, [N,K] and [K,N] bf16 copies
//              (the transposed one staged through LDS so both are written in full 128-byte rows).
// Per-group hyper-parameters travel by value in the kernel arguments (<= 64 groups per launch).
#include "common.h"
#ifndef GETMI_NT_ADAM
#define GETMI_NT_ADAM 0      // stream (nt) the optimizer state and the gradient: each is touched once per step
#endif
static constexpr bool kNtAdam = GETMI_NT_ADAM != 0;

struct AdamGroup {   // mirrors getmi_adamw_group (include/getmi355.h)
    float lr, weight_decay, beta1, beta2, eps, step_size, inv_sqrt_bc2, pad;
};
struct AdamGroups {
    AdamGroup g[64];
};
struct AdamJob {     // 8 x int64
    float* p;
    const float* g;
    float* m;
    float* v;
    bf16_t* wn;      // [rows, cols] bf16 copy or null
    bf16_t* wt;      // [cols, rows] bf16 copy or null (tile jobs only)
    long long a;     // flat: start element ; tile: r0 | c0 << 32
    long long b;     // flat: count (bits 0-31) ; tile: cols (0-19) | rows (20-39) ; group (40-47)
};

__device__ __forceinline__ float adam_one(float p, float g, float& m, float& v, const AdamGroup& h) {
    // torch.optim.AdamW (decoupled decay, no amsgrad): same operation order as its single-tensor reference
    p *= 1.0f - h.lr * h.weight_decay;
    m = m + (g - m) * (1.0f - h.beta1);                 // lerp form, as torch
    v = h.beta2 * v + (1.0f - h.beta2) * g * g;
    const float denom = sqrtf(v) * h.inv_sqrt_bc2 + h.eps;
    return p - h.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamJob* __restrict__ jobs, int nflat, AdamGroups hy) {
    __shared__ bf16_t tile[64][66];
    const AdamJob j = jobs[blockIdx.x];
    const int grp = (int)((j.b >> 40) & 0xff);
    const AdamGroup h = hy.g[grp];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < nflat) {
        const size_t start = (size_t)j.a;
        const int count = (int)(j.b & 0xffffffffll);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = (it * 256 + tid) * 4;
            if (e + 3 < count) {
                const float4 p = *reinterpret_cast<const float4*>(j.p + start + e);
                const float4 g = load_f4<kNtAdam>(j.g + start + e);
                float4 m = load_f4<kNtAdam>(j.m + start + e);
                float4 v = load_f4<kNtAdam>(j.v + start + e);
                float4 o;
                o.x = adam_one(p.x, g.x, m.x, v.x, h); o.y = adam_one(p.y, g.y, m.y, v.y, h);
                o.z = adam_one(p.z, g.z, m.z, v.z, h); o.w = adam_one(p.w, g.w, m.w, v.w, h);
                *reinterpret_cast<float4*>(j.p + start + e) = o;
                store_f4<kNtAdam>(j.m + start + e, m);
                store_f4<kNtAdam>(j.v + start + e, v);
                if (j.wn) *reinterpret_cast<uint2*>(j.wn + start + e) = pack4bf(o.x, o.y, o.z, o.w);
            } else {
                for (int k = e; k < count && k < e + 4; ++k) {      // ragged tail of a tensor
                    float m = j.m[start + k], v = j.v[start + k];
                    const float o = adam_one(j.p[start + k], j.g[start + k], m, v, h);
                    j.p[start + k] = o; j.m[start + k] = m; j.v[start + k] = v;
                    if (j.wn) j.wn[start + k] = f2bf(o);
                }
            }
        }
        return;
    }
    // ---- 64 x 64 tile of a [rows, cols] weight: thread = 4 consecutive columns of rows r, r+16, r+32, r+48 ----
    const int r0 = (int)(j.a & 0xffffffffll), c0 = (int)(j.a >> 32);
    const int cols = (int)(j.b & 0xfffff), rows = (int)((j.b >> 20) & 0xfffff);
    const int tr = tid >> 4, tc = (tid & 15) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = tr + it * 16;
        const size_t off = (size_t)(r0 + r) * cols + c0 + tc;
        const float4 p = *reinterpret_cast<const float4*>(j.p + off);
        const float4 g = load_f4<kNtAdam>(j.g + off);
        float4 m = load_f4<kNtAdam>(j.m + off);
        float4 v = load_f4<kNtAdam>(j.v + off);
        float4 o;
        o.x = adam_one(p.x, g.x, m.x, v.x, h); o.y = adam_one(p.y, g.y, m.y, v.y, h);
        o.z = adam_one(p.z, g.z, m.z, v.z, h); o.w = adam_one(p.w, g.w, m.w, v.w, h);
        *reinterpret_cast<float4*>(j.p + off) = o;
        store_f4<kNtAdam>(j.m + off, m);
        store_f4<kNtAdam>(j.v + off, v);
        const uint2 q = pack4bf(o.x, o.y, o.z, o.w);
        if (j.wn) *reinterpret_cast<uint2*>(j.wn + off) = q;
        tile[r][tc] = (bf16_t)(q.x & 0xffff); tile[r][tc + 1] = (bf16_t)(q.x >> 16);
        tile[r][tc + 2] = (bf16_t)(q.y & 0xffff); tile[r][tc + 3] = (bf16_t)(q.y >> 16);
    }
    if (j.wt == nullptr) return;   // block-uniform
    __syncthreads();
    // transposed copy: row (c0 + c) of wt, elements r0 .. r0+63: thread = 4 consecutive r of columns c, c+16, ...
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = tr + it * 16;
        const uint32_t lo = (uint32_t)tile[tc][c] | ((uint32_t)tile[tc + 1][c] << 16);
        const uint32_t hi = (uint32_t)tile[tc + 2][c] | ((uint32_t)tile[tc + 3][c] << 16);
        *reinterpret_cast<uint2*>(j.wt + (size_t)(c0 + c) * rows + r0 + tc) = make_uint2(lo, hi);
    }
}

int getmi_adamw_step_impl(const void* jobs, int nflat, int ntile, const AdamGroup* groups, int ngroups, hipStream_t st) {
    GETMI_CHECK_ARG(jobs != nullptr && nflat >= 0 && ntile >= 0, "adamw_step: bad job table");
    GETMI_CHECK_ARG(ngroups >= 1 && ngroups <= 64, "adamw_step: %d parameter groups (1..64 per call)", ngroups);
    GETMI_CHECK_ARG(((uintptr_t)jobs % 8) == 0, "adamw_step: job table must be 8-byte aligned");
    if (nflat + ntile == 0) return GETMI_OK;
    AdamGroups hy;
    memset(&hy, 0, sizeof(hy));
    for (int i = 0; i < ngroups; ++i) hy.g[i] = groups[i];
    hipLaunchKernelGGL(adamw_kernel, dim3(nflat + ntile), dim3(256), 0, st, reinterpret_cast<const AdamJob*>(jobs), nflat, hy);
    GETMI_CHECK_LAUNCH("adamw_step");
    return GETMI_OK;
}
