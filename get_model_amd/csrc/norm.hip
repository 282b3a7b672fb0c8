// LayerNorm forward / backward for the fp32 residual stream (HBM-bound; one wave per token row).
//
// Replaces norm1 / norm2 / encoder.norm (model/transformer.py:198,221,344; eps 1e-6 from :315) and, in the
// backward, also folds in the residual-gradient add, the per-sample DropPath scale of the *previous*
// branch and the bf16 cast that the preceding GEMMs' dgrad/wgrad consume, plus the column sums that are
// the previous Linear's bias gradient:
//     dx  = dres + LN'(dy)                       (fp32, the gradient of the residual stream)
//     dz  = bf16(rowscale[row/T] * dx)           (gradient w.r.t. the previous branch's Linear output)
//     partial[blk] = column sums over the block's rows of {dy*xhat, dy, dz}
// Algorithmic bytes per row (D=768): fwd 4*D read + 2*D write; bwd (2+4+4)*D read + (4+2)*D write.
#include "common.h"
#include <stdlib.h>
static constexpr bool kNtLn = GETMI_NT_LN != 0;    // activations are streamed: read once, written once
#ifndef GETMI_NT_LN_BWD_LOADS
#define GETMI_NT_LN_BWD_LOADS GETMI_NT_LN      // the backward's three input streams alone (their last use in the step)
#endif
static constexpr bool kNtLnBwdLd = GETMI_NT_LN_BWD_LOADS != 0;

template <int NV>  // D = NV * 256
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                     float eps) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 g[NV], bt[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        g[v] = *reinterpret_cast<const float4*>(gamma + (v * 64 + lane) * 4);
        bt[v] = *reinterpret_cast<const float4*>(beta + (v * 64 + lane) * 4);
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float* xr = x + (size_t)row * D;
        float4 a[NV];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            a[v] = load_f4<kNtLn>(xr + (v * 64 + lane) * 4);
            s += a[v].x + a[v].y + a[v].z + a[v].w;
        }
        const float mu = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            a[v].x -= mu; a[v].y -= mu; a[v].z -= mu; a[v].w -= mu;
            q += a[v].x * a[v].x + a[v].y * a[v].y + a[v].z * a[v].z + a[v].w * a[v].w;
        }
        const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
        bf16_t* yr = y + (size_t)row * D;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            store_u2<kNtLn>(yr + (v * 64 + lane) * 4,
                            pack4bf(a[v].x * rs * g[v].x + bt[v].x, a[v].y * rs * g[v].y + bt[v].y,
                                    a[v].z * rs * g[v].z + bt[v].z, a[v].w * rs * g[v].w + bt[v].w));
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// RB / XB: the residual-gradient stream comes in (dres) / goes out (dx) as bf16 instead of fp32 (the "bf16 gradient
// stream", GETMI_BF16_GRAD in functional.py: 12 instead of 16 bytes per element).  zsum: accumulate the column sums of
// bf16(rowscale * dx) even when dz is null (with rowscale == null and XB the caller uses dx itself as dz).
template <int NV, bool RB, bool XB>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const void* __restrict__ dres_,
                                                     const float* __restrict__ rowscale, int rows_per_sample,
                                                     void* __restrict__ dx_, bf16_t* __restrict__ dz,
                                                     float* __restrict__ partial, int M, int zsum) {
    constexpr int D = NV * 256;
    __shared__ __attribute__((aligned(16))) float red[3][4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* dres = reinterpret_cast<const float*>(dres_);
    const bf16_t* dresb = reinterpret_cast<const bf16_t*>(dres_);
    float* dx = reinterpret_cast<float*>(dx_);
    bf16_t* dxb = reinterpret_cast<bf16_t*>(dx_);
    float4 g[NV], ag[NV], ab[NV], az[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        g[v] = *reinterpret_cast<const float4*>(gamma + (v * 64 + lane) * 4);
        ag[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[v] = ag[v];
        az[v] = ag[v];
    }
    // software pipeline over this wave's rows: the next row's x / dy / dres loads are issued before the current
    // row's two wave reductions, so the reduction latency is hidden behind HBM traffic.
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    float4 nx[NV], nr[NV];
    uint2 nd[NV], nrb[NV];
    float nmu = 0.f, nrs = 0.f;
    auto issue = [&](int r) {
        nmu = mean[r];
        nrs = rstd[r];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            nx[v] = load_f4<kNtLnBwdLd>(x + (size_t)r * D + (v * 64 + lane) * 4);
            nd[v] = load_u2<kNtLnBwdLd>(dy + (size_t)r * D + (v * 64 + lane) * 4);
            if (dres_) {
                if constexpr (RB) nrb[v] = load_u2<kNtLnBwdLd>(dresb + (size_t)r * D + (v * 64 + lane) * 4);
                else nr[v] = load_f4<kNtLnBwdLd>(dres + (size_t)r * D + (v * 64 + lane) * 4);
            }
        }
    };
    if (row < M) issue(row);
    for (; row < M; row += stride) {
        const float mu = nmu, rs = nrs;
        const float sc = rowscale ? rowscale[row / rows_per_sample] : 1.0f;
        float4 xh[NV], d[NV], rv[NV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 xv = nx[v];
            const uint2 dv = nd[v];
            if (!dres_) rv[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            else if constexpr (RB)
                rv[v] = make_float4(bf2f(nrb[v].x & 0xffff), bf2f(nrb[v].x >> 16), bf2f(nrb[v].y & 0xffff), bf2f(nrb[v].y >> 16));
            else rv[v] = nr[v];
            xh[v] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            d[v] = make_float4(bf2f(dv.x & 0xffff), bf2f(dv.x >> 16), bf2f(dv.y & 0xffff), bf2f(dv.y >> 16));
        }
        if (row + stride < M) issue(row + stride);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            ag[v].x += d[v].x * xh[v].x; ag[v].y += d[v].y * xh[v].y;
            ag[v].z += d[v].z * xh[v].z; ag[v].w += d[v].w * xh[v].w;
            ab[v].x += d[v].x; ab[v].y += d[v].y; ab[v].z += d[v].z; ab[v].w += d[v].w;
            d[v].x *= g[v].x; d[v].y *= g[v].y; d[v].z *= g[v].z; d[v].w *= g[v].w;
            c1 += d[v].x + d[v].y + d[v].z + d[v].w;
            c2 += d[v].x * xh[v].x + d[v].y * xh[v].y + d[v].z * xh[v].z + d[v].w * xh[v].w;
        }
        c1 = wave_sum(c1) * (1.0f / D);
        c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float4 o = make_float4(rs * (d[v].x - c1 - xh[v].x * c2), rs * (d[v].y - c1 - xh[v].y * c2),
                                   rs * (d[v].z - c1 - xh[v].z * c2), rs * (d[v].w - c1 - xh[v].w * c2));
            o.x += rv[v].x; o.y += rv[v].y; o.z += rv[v].z; o.w += rv[v].w;
            if constexpr (XB) store_u2<kNtLn>(dxb + (size_t)row * D + (v * 64 + lane) * 4, pack4bf(o.x, o.y, o.z, o.w));
            else store_f4<kNtLn>(dx + (size_t)row * D + (v * 64 + lane) * 4, o);
            if (dz || zsum) {
                const float4 z = make_float4(o.x * sc, o.y * sc, o.z * sc, o.w * sc);
                if (dz) store_u2<kNtLn>(dz + (size_t)row * D + (v * 64 + lane) * 4, pack4bf(z.x, z.y, z.z, z.w));
                az[v].x += z.x; az[v].y += z.y; az[v].z += z.z; az[v].w += z.w;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        *reinterpret_cast<float4*>(&red[0][wave][(v * 64 + lane) * 4]) = ag[v];
        *reinterpret_cast<float4*>(&red[1][wave][(v * 64 + lane) * 4]) = ab[v];
        *reinterpret_cast<float4*>(&red[2][wave][(v * 64 + lane) * 4]) = az[v];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * D; i += 256) {
        const int k = i / D, c = i - k * D;
        partial[((size_t)k * gridDim.x + blockIdx.x) * D + c] = red[k][0][c] + red[k][1][c] + red[k][2][c] + red[k][3][c];
    }
}

// out_k[c] (+)= sum_r partial[k][r][c]   for k = blockIdx.y; 32 columns x 32 row groups per block, 4 loads in flight
__global__ __launch_bounds__(1024) void colreduce_kernel(const float* __restrict__ partial, float* out0, float* out1,
                                                         float* out2, int R, int C, int accumulate) {
    __shared__ float red[32][33];
    const int k = blockIdx.y;
    float* out = k == 0 ? out0 : (k == 1 ? out1 : out2);
    if (out == nullptr) return;   // block-uniform
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    float s = 0.f;
    if (c < C) {
        const float* p = partial + (size_t)k * R * C + c;
        int r = ty;
        for (; r + 96 < R; r += 128) {
            const float a0 = p[(size_t)r * C], a1 = p[(size_t)(r + 32) * C];
            const float a2 = p[(size_t)(r + 64) * C], a3 = p[(size_t)(r + 96) * C];
            s += (a0 + a1) + (a2 + a3);
        }
        for (; r < R; r += 32) s += p[(size_t)r * C];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty < 16) red[ty][tx] += red[ty + 16][tx];
    __syncthreads();
    if (ty == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][tx];
        out[c] = accumulate ? out[c] + t : t;
    }
}

int getmi_colreduce_impl(const float* partial, float* out0, float* out1, float* out2, int R, int C, int accumulate,
                         hipStream_t st) {
    hipLaunchKernelGGL(colreduce_kernel, dim3((C + 31) / 32, 3), dim3(1024), 0, st, partial, out0, out1, out2, R, C,
                       accumulate);
    GETMI_CHECK_LAUNCH("colreduce");
    return GETMI_OK;
}

static int ln_grid_cap(const char* name, int dflt) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : dflt;
    return v < 1 ? dflt : v;
}
static int ln_grid(int M) {      // forward
    static int cap = 0;
    if (cap == 0) cap = ln_grid_cap("GETMI_LN_GRID", 1024);
    int g = (M + 3) / 4;
    if (g > cap) g = cap;
    return g < 1 ? 1 : g;
}
// Backward: TWO workgroups per CU.  Measured in the training step at M = 51 456 (tools/exp_ln_grid.sh, MI355X, us per launch):
// 256 workgroups 75.8, 512 76.8, 768 79.8, 1024 86.9, 384 83.1, 640 82.0 -- whole multiples of the CU count win and fewer
// resident rows win (the band of rows in flight stays narrow: four streams per row); 512 keeps two workgroups per CU
// for the case where the CU limit of an overlapped gradient exchange takes some CUs away.
static int ln_bwd_grid(int M) {
    static int cap = 0;
    if (cap == 0) cap = ln_grid_cap("GETMI_LN_BWD_GRID", 512);      // 2 x 256 CUs (MI355X)
    int g = (M + 3) / 4;
    if (g > cap) g = cap;
    return g < 1 ? 1 : g;
}

int getmi_ln_fwd_impl(const float* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd,
                      int M, int D, float eps, hipStream_t st) {
    GETMI_CHECK_ARG(M > 0, "ln_fwd: M=%d", M);
    GETMI_CHECK_ARG(D == 256 || D == 512 || D == 768 || D == 1024, "ln_fwd: D=%d unsupported (256,512,768,1024)", D);
    const int g = ln_grid(M);
    switch (D / 256) {
        case 1: hipLaunchKernelGGL(ln_fwd_kernel<1>, dim3(g), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, eps); break;
        case 2: hipLaunchKernelGGL(ln_fwd_kernel<2>, dim3(g), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, eps); break;
        case 3: hipLaunchKernelGGL(ln_fwd_kernel<3>, dim3(g), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, eps); break;
        case 4: hipLaunchKernelGGL(ln_fwd_kernel<4>, dim3(g), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, eps); break;
    }
    GETMI_CHECK_LAUNCH("ln_fwd");
    return GETMI_OK;
}

size_t getmi_ln_bwd_workspace_bytes(int M, int D) { return (size_t)3 * ln_bwd_grid(M) * D * sizeof(float); }

// dgamma/dbeta/dbias_prev: [D] each (dbias_prev may be null); accumulate: add into existing values.
// dres_bf16 / dx_bf16: element type of the residual-gradient stream on the way in / out (0 = fp32, 1 = bf16).
// dz may be null with dbias_prev set only when rowscale is null and dx is bf16 (then dz == dx bit for bit).
int getmi_ln_bwd_impl(const bf16_t* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      const void* dres, int dres_bf16, const float* rowscale, int rows_per_sample, void* dx, int dx_bf16,
                      bf16_t* dz, float* dgamma, float* dbeta, float* dbias_prev, int accumulate, int M, int D,
                      void* workspace, size_t workspace_bytes, hipStream_t st) {
    GETMI_CHECK_ARG(M > 0, "ln_bwd: M=%d", M);
    GETMI_CHECK_ARG(D == 256 || D == 512 || D == 768 || D == 1024, "ln_bwd: D=%d unsupported", D);
    GETMI_CHECK_ARG(workspace_bytes >= getmi_ln_bwd_workspace_bytes(M, D), "ln_bwd: workspace too small");
    GETMI_CHECK_ARG(!rowscale || rows_per_sample > 0, "ln_bwd: rows_per_sample");
    // (dbias_prev without dz: the column sums of rowscale * dx are formed without storing that tensor -- the DropPath fold of
    //  functional.py, whose GEMMs consume the unscaled bf16 dx)
    const int g = ln_bwd_grid(M);
    const int zsum = (dbias_prev != nullptr && dz == nullptr) ? 1 : 0;
    float* partial = reinterpret_cast<float*>(workspace);
#define LN_BWD_LAUNCH2(NV, RB, XB)                                                                                \
    hipLaunchKernelGGL((ln_bwd_kernel<NV, RB, XB>), dim3(g), dim3(256), 0, st, dy, x, mean, rstd, gamma, dres,   \
                       rowscale, rows_per_sample, dx, dz, partial, M, zsum)
#define LN_BWD_LAUNCH(NV)                                                                                         \
    do {                                                                                                          \
        if (dres_bf16 && dx_bf16) LN_BWD_LAUNCH2(NV, true, true);                                                 \
        else if (dres_bf16) LN_BWD_LAUNCH2(NV, true, false);                                                      \
        else if (dx_bf16) LN_BWD_LAUNCH2(NV, false, true);                                                        \
        else LN_BWD_LAUNCH2(NV, false, false);                                                                    \
    } while (0)
    switch (D / 256) {
        case 1: LN_BWD_LAUNCH(1); break;
        case 2: LN_BWD_LAUNCH(2); break;
        case 3: LN_BWD_LAUNCH(3); break;
        case 4: LN_BWD_LAUNCH(4); break;
    }
#undef LN_BWD_LAUNCH
#undef LN_BWD_LAUNCH2
    GETMI_CHECK_LAUNCH("ln_bwd");
    if (int rc = getmi_colreduce_impl(partial, dgamma, dbeta, dbias_prev, g, D, accumulate, st)) return rc;
    return GETMI_OK;
}
