// HBM-bound helper kernels of the GET region hot path: casts/packing, region-embed glue, expression head,
// losses, row gather/scatter, column sums.  All are simple streaming kernels with 8-16 B per-lane accesses.
#include "common.h"
#include <mutex>

#define GRID_CAP 2048

static int grid_for(size_t n, int per_block) {
    size_t g = (n + per_block - 1) / per_block;
    if (g > GRID_CAP) g = GRID_CAP;
    return g < 1 ? 1 : (int)g;
}

// ---- fp32 [R, C] -> bf16 [R, Cp] with zero padding of columns C..Cp-1 (Cp % 4 == 0) ----
__global__ void cast_pad_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t R, int C, int Cp) {
    const int q = Cp >> 2;
    const size_t n = R * q;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / q;
        const int c = (int)(i - r * q) * 4;
        const float* s = src + r * C + c;
        const float a = c < C ? s[0] : 0.f, b = c + 1 < C ? s[1] : 0.f, d = c + 2 < C ? s[2] : 0.f,
                    e = c + 3 < C ? s[3] : 0.f;
        *reinterpret_cast<uint2*>(dst + r * Cp + c) = pack4bf(a, b, d, e);
    }
}

// ---- fp32 [R, C] -> bf16 [C, R] (transposed copy, for dgrad); 32x32 tiles through LDS ----
__global__ void cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * R + r] = f2bf(tile[tx][i]);
    }
}

// ---- qkv bias = [q_bias, 0, v_bias]  (model/transformer.py:50-56) ----
__global__ void qkv_bias_kernel(const float* __restrict__ qb, const float* __restrict__ vb, float* __restrict__ out,
                                int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D) { out[i] = qb[i]; out[D + i] = 0.f; out[2 * D + i] = vb[i]; }
}

// ---- x0[b, 0, :] = cls_token  (model/model.py:572-573) ----
__global__ void fill_cls_kernel(float* __restrict__ x0, const float* __restrict__ cls, int B, int T, int D) {
    const size_t n = (size_t)B * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / D;
        const int d = (int)(i - b * D);
        x0[b * T * (size_t)D + d] = cls[d];
    }
}

// ---- backward of cls-concat + mask-token blend + embed bias (model/model.py:520-525) ----
// dz[b*N+n, :] = mask ? 0 : dx0[b*T+1+n, :] (bf16); partial[0]=sum_b dx0[b*T] (cls), partial[1]=sum over masked
// region rows (mask_token), partial[2]=sum over unmasked region rows (embed bias).  One wave per row.
template <int NV>
__global__ __launch_bounds__(256) void embed_bwd_prep_kernel(const float* __restrict__ dx0,
                                                             const uint8_t* __restrict__ mask, bf16_t* __restrict__ dz,
                                                             float* __restrict__ partial, int B, int N) {
    constexpr int D = NV * 256;
    __shared__ __attribute__((aligned(16))) float red[3][4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = N + 1;
    float4 acc[3][NV];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[k][v] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int rows = B * T;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int b = row / T, t = row - b * T;
        const int kind = (t == 0) ? 0 : ((mask && mask[b * N + t - 1]) ? 1 : 2);   // wave-uniform
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 g = *reinterpret_cast<const float4*>(dx0 + (size_t)row * D + (v * 64 + lane) * 4);
            if (kind == 0) { acc[0][v].x += g.x; acc[0][v].y += g.y; acc[0][v].z += g.z; acc[0][v].w += g.w; }
            else if (kind == 1) { acc[1][v].x += g.x; acc[1][v].y += g.y; acc[1][v].z += g.z; acc[1][v].w += g.w; }
            else { acc[2][v].x += g.x; acc[2][v].y += g.y; acc[2][v].z += g.z; acc[2][v].w += g.w; }
            if (t > 0) {
                const float4 z = (kind == 2) ? g : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<uint2*>(dz + ((size_t)b * N + t - 1) * D + (v * 64 + lane) * 4) =
                    pack4bf(z.x, z.y, z.z, z.w);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int v = 0; v < NV; ++v) *reinterpret_cast<float4*>(&red[k][wave][(v * 64 + lane) * 4]) = acc[k][v];
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * D; i += 256) {
        const int k = i / D, c = i - k * D;
        partial[((size_t)k * gridDim.x + blockIdx.x) * D + c] = red[k][0][c] + red[k][1][c] + red[k][2][c] + red[k][3][c];
    }
}

// ---- dz = bf16(rowscale[row/T] * dx) and its column sums (the gradient entering a residual branch:
//      DropPath backward of model/transformer.py:247-248 + bias gradient of the branch's last Linear) ----
template <int NV>
__global__ __launch_bounds__(256) void scale_cast_kernel(const float* __restrict__ dx, const float* __restrict__ rowscale,
                                                         int rows_per_sample, bf16_t* __restrict__ dz,
                                                         float* __restrict__ partial, int M) {
    constexpr int D = NV * 256;
    __shared__ __attribute__((aligned(16))) float red[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float sc = rowscale ? rowscale[row / rows_per_sample] : 1.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float4 g = *reinterpret_cast<const float4*>(dx + (size_t)row * D + (v * 64 + lane) * 4);
            g.x *= sc; g.y *= sc; g.z *= sc; g.w *= sc;
            *reinterpret_cast<uint2*>(dz + (size_t)row * D + (v * 64 + lane) * 4) = pack4bf(g.x, g.y, g.z, g.w);
            acc[v].x += g.x; acc[v].y += g.y; acc[v].z += g.z; acc[v].w += g.w;
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) *reinterpret_cast<float4*>(&red[wave][(v * 64 + lane) * 4]) = acc[v];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256)
        partial[(size_t)blockIdx.x * D + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// ---- expression head: logits = hf[b, 1+n, :] . Wh^T + bh ; pred = softplus(logits)
//      (model/model.py:575-576, model/modules.py:255-258; Softplus beta=1, threshold=20) ----
template <int NV, int OUT>
__global__ __launch_bounds__(256) void head_fwd_kernel(const bf16_t* __restrict__ hf, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ logits,
                                                       float* __restrict__ pred, int B, int N) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 w[OUT][NV];
#pragma unroll
    for (int o = 0; o < OUT; ++o)
#pragma unroll
        for (int v = 0; v < NV; ++v) w[o][v] = *reinterpret_cast<const float4*>(W + (size_t)o * D + (v * 64 + lane) * 4);
    const int rows = B * N;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int b = row / N, n = row - b * N;
        const bf16_t* hr = hf + ((size_t)b * (N + 1) + 1 + n) * D;
        float s[OUT];
#pragma unroll
        for (int o = 0; o < OUT; ++o) s[o] = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const uint2 hv = *reinterpret_cast<const uint2*>(hr + (v * 64 + lane) * 4);
            const float h0 = bf2f(hv.x & 0xffff), h1 = bf2f(hv.x >> 16), h2 = bf2f(hv.y & 0xffff), h3 = bf2f(hv.y >> 16);
#pragma unroll
            for (int o = 0; o < OUT; ++o) s[o] += h0 * w[o][v].x + h1 * w[o][v].y + h2 * w[o][v].z + h3 * w[o][v].w;
        }
#pragma unroll
        for (int o = 0; o < OUT; ++o) {
            const float z = wave_sum(s[o]) + bias[o];
            if (lane == 0) {
                logits[(size_t)row * OUT + o] = z;
                pred[(size_t)row * OUT + o] = z > 20.f ? z : log1pf(__expf(z));
            }
        }
    }
}

// dh[b, 1+n, :] = sum_o dlogit[o] * Wh[o, :] (bf16; cls rows zero); partial[blk][OUT*D + OUT] = {dWh, dbh} sums
template <int NV, int OUT>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ logits,
                                                       const bf16_t* __restrict__ hf, const float* __restrict__ W,
                                                       bf16_t* __restrict__ dh, float* __restrict__ partial, int B, int N) {
    constexpr int D = NV * 256;
    constexpr int PC = OUT * D + OUT;
    constexpr int PCP = (PC + 3) & ~3;   // keep every wave's row 16-byte aligned for the float4 stores
    __shared__ __attribute__((aligned(16))) float red[4][PCP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = N + 1;
    float4 w[OUT][NV], aw[OUT][NV];
    float abias[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        abias[o] = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            w[o][v] = *reinterpret_cast<const float4*>(W + (size_t)o * D + (v * 64 + lane) * 4);
            aw[o][v] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const int rows = B * T;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int b = row / T, t = row - b * T;
        bf16_t* dr = dh + (size_t)row * D;
        if (t == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v) *reinterpret_cast<uint2*>(dr + (v * 64 + lane) * 4) = make_uint2(0u, 0u);
            continue;
        }
        const size_t rr = (size_t)b * N + t - 1;
        float dl[OUT];
#pragma unroll
        for (int o = 0; o < OUT; ++o) {
            const float z = logits[rr * OUT + o];
            const float sg = z > 20.f ? 1.0f : 1.0f / (1.0f + __expf(-z));
            dl[o] = dpred[rr * OUT + o] * sg;
            abias[o] += dl[o];
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const uint2 hv = *reinterpret_cast<const uint2*>(hf + (size_t)row * D + (v * 64 + lane) * 4);
            const float h0 = bf2f(hv.x & 0xffff), h1 = bf2f(hv.x >> 16), h2 = bf2f(hv.y & 0xffff), h3 = bf2f(hv.y >> 16);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int o = 0; o < OUT; ++o) {
                g.x += dl[o] * w[o][v].x; g.y += dl[o] * w[o][v].y; g.z += dl[o] * w[o][v].z; g.w += dl[o] * w[o][v].w;
                aw[o][v].x += dl[o] * h0; aw[o][v].y += dl[o] * h1; aw[o][v].z += dl[o] * h2; aw[o][v].w += dl[o] * h3;
            }
            *reinterpret_cast<uint2*>(dr + (v * 64 + lane) * 4) = pack4bf(g.x, g.y, g.z, g.w);
        }
    }
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
#pragma unroll
        for (int v = 0; v < NV; ++v) *reinterpret_cast<float4*>(&red[wave][o * D + (v * 64 + lane) * 4]) = aw[o][v];
        if (lane == 0) red[wave][OUT * D + o] = abias[o];   // every lane holds the same per-row dl sums
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PC; i += 256)
        partial[(size_t)blockIdx.x * PC + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// ---- losses (two-stage deterministic reduction); grad_unit = d loss / d pred for an upstream gradient of 1 ----
// kind 0: PoissonNLL(log_input=False, full=False, mean, eps):  pred - obs*log(pred+eps)
// kind 1: MSE(mean): (pred-obs)^2
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ obs,
                                                           float* __restrict__ grad_unit, float* __restrict__ partial,
                                                           size_t R, int C, int ldp, int ldo, int ldg, int kind,
                                                           float eps) {
    __shared__ float red[4];
    const size_t n = R * C;
    const float inv_n = 1.0f / (float)n;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C;
        const int c = (int)(i - r * C);
        const float p = pred[r * ldp + c], o = obs[r * ldo + c];
        float g;
        if (kind == 0) { s += p - o * logf(p + eps); g = 1.0f - o / (p + eps); }
        else { const float d = p - o; s += d * d; g = 2.0f * d; }
        grad_unit[r * ldg + c] = g * inv_n;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void loss_final_kernel(const float* __restrict__ partial, int nblk, float inv_n, float weight,
                                  float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) * inv_n * weight;
}
// out[i] = in[i] * scalar[0] * weight   (upstream gradient is a device scalar: no host sync)
__global__ void scale_by_scalar_kernel(const float* __restrict__ in, const float* __restrict__ scalar, float weight,
                                       float* __restrict__ out, size_t n) {
    const float s = scalar[0] * weight;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i] * s;
}

// ---- bf16 row gather / scatter (pretrain masked rows, model/model.py:527); one wave per row, D % 8 == 0 ----
__global__ void gather_rows_kernel(const bf16_t* __restrict__ src, const int64_t* __restrict__ idx,
                                   bf16_t* __restrict__ dst, int n, int D, int scatter) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
        const size_t s = scatter ? (size_t)i : (size_t)idx[i];
        const size_t d = scatter ? (size_t)idx[i] : (size_t)i;
        for (int c = lane * 8; c < D; c += 512)
            *reinterpret_cast<uint4*>(dst + d * D + c) = *reinterpret_cast<const uint4*>(src + s * D + c);
    }
}

// ---- masked-row compaction (pretrain: x[:, 1:][mask], model/model.py:527,535) without a host round trip -------
// One workgroup of 1024 threads: thread t counts the set bytes of its contiguous chunk, a block-wide exclusive scan
// turns the counts into output offsets, and the chunk is walked again to write, in ascending order (= the row order
// of torch boolean indexing), the token row b*(N+1)+1+r and the region row b*N+r of every masked region.
__global__ __launch_bounds__(1024) void mask_compact_kernel(const uint8_t* __restrict__ mask, int total, int N,
                                                            int64_t* __restrict__ token_rows,
                                                            int64_t* __restrict__ region_rows, int* __restrict__ count) {
    __shared__ int wave_total[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (total + 1023) / 1024;
    const int lo = min(tid * per, total), hi = min(lo + per, total);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += mask[i] != 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_total[w];
    int off = before + incl - c;
    for (int i = lo; i < hi; ++i)
        if (mask[i]) {
            const int b = i / N, r = i - b * N;
            token_rows[off] = (int64_t)b * (N + 1) + 1 + r;
            region_rows[off] = i;
            ++off;
        }
    if (tid == 1023) count[0] = before + incl;
}

int getmi_mask_compact_impl(const uint8_t* mask, int B, int N, int64_t* token_rows, int64_t* region_rows, int* count,
                            hipStream_t st) {
    GETMI_CHECK_ARG(B > 0 && N > 0 && (long long)B * N < (1ll << 31), "mask_compact: bad shape B=%d N=%d", B, N);
    hipLaunchKernelGGL(mask_compact_kernel, dim3(1), dim3(1024), 0, st, mask, B * N, N, token_rows, region_rows, count);
    GETMI_CHECK_LAUNCH("mask_compact");
    return GETMI_OK;
}

// ---- column sums of a bf16 matrix [M, N] -> partial[gridDim.y][N] (then colreduce); N % 4 == 0 ----
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ X, float* __restrict__ partial,
                                                          int M, int N, int ld) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= N) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = blockIdx.y; r < M; r += gridDim.y) {
        const uint2 v = *reinterpret_cast<const uint2*>(X + (size_t)r * ld + c);
        s.x += bf2f(v.x & 0xffff); s.y += bf2f(v.x >> 16); s.z += bf2f(v.y & 0xffff); s.w += bf2f(v.y >> 16);
    }
    *reinterpret_cast<float4*>(partial + (size_t)blockIdx.y * N + c) = s;
}

// Streaming form for N % 8 == 0, ld % 8 == 0, 16-byte aligned X: a block is 64 column chunks (16 B each) x 4 row
// lanes, every thread keeps 8 row loads in flight (HBM-bound: 8 waves/CU x 8 x 1 KiB outstanding).
__global__ __launch_bounds__(256) void colsum_bf16_wide_kernel(const bf16_t* __restrict__ X, float* __restrict__ partial,
                                                               int M, int N, int ld) {
    __shared__ __attribute__((aligned(16))) float red[4][64][8];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + tx) * 8;
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
    auto add = [&](const uint4& v) {
        s[0] += bf2f(v.x & 0xffff); s[1] += bf2f(v.x >> 16); s[2] += bf2f(v.y & 0xffff); s[3] += bf2f(v.y >> 16);
        s[4] += bf2f(v.z & 0xffff); s[5] += bf2f(v.z >> 16); s[6] += bf2f(v.w & 0xffff); s[7] += bf2f(v.w >> 16);
    };
    if (c < N) {
        const int stride = gridDim.y * 4;
        const bf16_t* px = X + c;
        int r = blockIdx.y * 4 + ty;
        for (; r + 7 * stride < M; r += 8 * stride) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4*>(px + (size_t)(r + u * stride) * ld);
#pragma unroll
            for (int u = 0; u < 8; ++u) add(v[u]);
        }
        for (; r < M; r += stride) add(*reinterpret_cast<const uint4*>(px + (size_t)r * ld));
    }
    *reinterpret_cast<float4*>(&red[ty][tx][0]) = make_float4(s[0], s[1], s[2], s[3]);
    *reinterpret_cast<float4*>(&red[ty][tx][4]) = make_float4(s[4], s[5], s[6], s[7]);
    __syncthreads();
    if (ty == 0 && c < N) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = (red[0][tx][i] + red[1][tx][i]) + (red[2][tx][i] + red[3][tx][i]);
        float* o = partial + (size_t)blockIdx.y * N + c;
        *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(s[4], s[5], s[6], s[7]);
    }
}

// ================================================================================================
int getmi_colreduce_impl(const float* partial, float* out0, float* out1, float* out2, int R, int C, int accumulate,
                         hipStream_t st);

int getmi_cast_pad_impl(const float* src, bf16_t* dst, size_t R, int C, int Cp, hipStream_t st) {
    GETMI_CHECK_ARG(R > 0 && C > 0 && Cp >= C && Cp % 4 == 0, "cast_pad: bad shape R=%zu C=%d Cp=%d", R, C, Cp);
    hipLaunchKernelGGL(cast_pad_kernel, dim3(grid_for(R * (Cp / 4), 256)), dim3(256), 0, st, src, dst, R, C, Cp);
    GETMI_CHECK_LAUNCH("cast_pad");
    return GETMI_OK;
}

int getmi_cast_transpose_impl(const float* src, bf16_t* dst, int R, int C, hipStream_t st) {
    GETMI_CHECK_ARG(R > 0 && C > 0, "cast_transpose: bad shape");
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, st, src, dst, R, C);
    GETMI_CHECK_LAUNCH("cast_transpose");
    return GETMI_OK;
}

int getmi_qkv_bias_impl(const float* qb, const float* vb, float* out, int D, hipStream_t st) {
    hipLaunchKernelGGL(qkv_bias_kernel, dim3((D + 255) / 256), dim3(256), 0, st, qb, vb, out, D);
    GETMI_CHECK_LAUNCH("qkv_bias");
    return GETMI_OK;
}

int getmi_fill_cls_impl(float* x0, const float* cls, int B, int T, int D, hipStream_t st) {
    hipLaunchKernelGGL(fill_cls_kernel, dim3(grid_for((size_t)B * D, 256)), dim3(256), 0, st, x0, cls, B, T, D);
    GETMI_CHECK_LAUNCH("fill_cls");
    return GETMI_OK;
}

static int rows_grid(int rows) {
    int g = (rows + 3) / 4;
    if (g > 1024) g = 1024;
    return g < 1 ? 1 : g;
}

size_t getmi_embed_bwd_workspace_bytes_impl(int B, int N, int D) { return (size_t)3 * rows_grid(B * (N + 1)) * D * sizeof(float); }

int getmi_embed_bwd_prep_impl(const float* dx0, const uint8_t* mask, bf16_t* dz, float* dcls, float* dmask_token,
                              float* dbias, int accumulate, int B, int N, int D, void* ws, size_t ws_bytes,
                              hipStream_t st) {
    GETMI_CHECK_ARG(D == 768 || D == 256 || D == 512 || D == 1024, "embed_bwd_prep: D=%d unsupported", D);
    GETMI_CHECK_ARG(ws_bytes >= getmi_embed_bwd_workspace_bytes_impl(B, N, D), "embed_bwd_prep: workspace too small");
    const int g = rows_grid(B * (N + 1));
    float* partial = reinterpret_cast<float*>(ws);
    switch (D / 256) {
        case 1: hipLaunchKernelGGL(embed_bwd_prep_kernel<1>, dim3(g), dim3(256), 0, st, dx0, mask, dz, partial, B, N); break;
        case 2: hipLaunchKernelGGL(embed_bwd_prep_kernel<2>, dim3(g), dim3(256), 0, st, dx0, mask, dz, partial, B, N); break;
        case 3: hipLaunchKernelGGL(embed_bwd_prep_kernel<3>, dim3(g), dim3(256), 0, st, dx0, mask, dz, partial, B, N); break;
        case 4: hipLaunchKernelGGL(embed_bwd_prep_kernel<4>, dim3(g), dim3(256), 0, st, dx0, mask, dz, partial, B, N); break;
    }
    GETMI_CHECK_LAUNCH("embed_bwd_prep");
    return getmi_colreduce_impl(partial, dcls, dmask_token, dbias, g, D, accumulate, st);
}

size_t getmi_scale_cast_workspace_bytes_impl(int M, int D) { return (size_t)rows_grid(M) * D * sizeof(float); }

int getmi_scale_cast_impl(const float* dx, const float* rowscale, int rows_per_sample, bf16_t* dz, float* dbias,
                          int accumulate, int M, int D, void* ws, size_t ws_bytes, hipStream_t st) {
    GETMI_CHECK_ARG(D == 768 || D == 256 || D == 512 || D == 1024, "scale_cast: D=%d unsupported", D);
    GETMI_CHECK_ARG(M > 0 && (!rowscale || rows_per_sample > 0), "scale_cast: bad shape");
    GETMI_CHECK_ARG(ws_bytes >= getmi_scale_cast_workspace_bytes_impl(M, D), "scale_cast: workspace too small");
    const int g = rows_grid(M);
    float* partial = reinterpret_cast<float*>(ws);
    switch (D / 256) {
        case 1: hipLaunchKernelGGL(scale_cast_kernel<1>, dim3(g), dim3(256), 0, st, dx, rowscale, rows_per_sample, dz, partial, M); break;
        case 2: hipLaunchKernelGGL(scale_cast_kernel<2>, dim3(g), dim3(256), 0, st, dx, rowscale, rows_per_sample, dz, partial, M); break;
        case 3: hipLaunchKernelGGL(scale_cast_kernel<3>, dim3(g), dim3(256), 0, st, dx, rowscale, rows_per_sample, dz, partial, M); break;
        case 4: hipLaunchKernelGGL(scale_cast_kernel<4>, dim3(g), dim3(256), 0, st, dx, rowscale, rows_per_sample, dz, partial, M); break;
    }
    GETMI_CHECK_LAUNCH("scale_cast");
    return getmi_colreduce_impl(partial, dbias, nullptr, nullptr, g, D, accumulate, st);
}

// ---- getmi_scale_cast_noalloc: the same two kernels with the column-sum partials in a built-in device buffer -------------
// For callers that cannot allocate or pass a workspace (HIP graph capture, serving loops, harnesses that forbid
// hipMalloc).  `g_scale_cast_scratch` is a file-scope device array: in HOST code its name is only a handle, never a
// device address, so the address every kernel launch and every other host-side use takes is the one
// hipGetSymbolAddress returns, looked up once per device (each GPU holds its own copy at its own address).
#define SCALE_CAST_SCRATCH_BYTES (3u << 20)      // 3 MiB: 1024 partial rows x 768 columns x 4 B (the largest grid at D = 768)
static __device__ __align__(256) float g_scale_cast_scratch[SCALE_CAST_SCRATCH_BYTES / sizeof(float)];
#define SCRATCH_MAX_DEVICES 64
static float* g_scale_cast_scratch_ptr[SCRATCH_MAX_DEVICES];
static std::mutex g_scale_cast_scratch_lock;

static int scale_cast_scratch(float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCRATCH_MAX_DEVICES) {
        getmi_set_error("scale_cast_noalloc: no current HIP device (or device index %d >= %d)", dev, SCRATCH_MAX_DEVICES);
        return GETMI_ERR_ARG;
    }
    std::lock_guard<std::mutex> guard(g_scale_cast_scratch_lock);
    if (g_scale_cast_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_scale_cast_scratch));
        if (e != hipSuccess || p == nullptr) {
            getmi_set_error("scale_cast_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return GETMI_ERR_LAUNCH;
        }
        g_scale_cast_scratch_ptr[dev] = reinterpret_cast<float*>(p);
    }
    *out = g_scale_cast_scratch_ptr[dev];
    return GETMI_OK;
}

// Also the entry point's SET-UP call: it resolves the buffer address for the current device.  hipGetSymbolAddress may
// load the code object and is not a stream operation, so it must not happen for the first time inside a HIP graph
// capture; callers query this limit once per device at start-up (outside capture), after which every
// getmi_scale_cast_noalloc call on that device is two kernel launches and nothing else.
size_t getmi_scale_cast_noalloc_max_bytes_impl(void) {
    float* unused = nullptr;
    (void)scale_cast_scratch(&unused);
    return SCALE_CAST_SCRATCH_BYTES;
}

int getmi_scale_cast_noalloc_impl(const float* dx, const float* rowscale, int rows_per_sample, bf16_t* dz, float* dbias,
                                  int accumulate, int M, int D, hipStream_t st) {
    GETMI_CHECK_ARG(D == 768 || D == 256 || D == 512 || D == 1024, "scale_cast_noalloc: D=%d unsupported", D);
    GETMI_CHECK_ARG(M > 0, "scale_cast_noalloc: M=%d", M);
    const size_t need = getmi_scale_cast_workspace_bytes_impl(M, D);
    GETMI_CHECK_ARG(need <= SCALE_CAST_SCRATCH_BYTES,
                    "scale_cast_noalloc: M=%d x D=%d needs %zu bytes of scratch, the built-in buffer holds %u "
                    "(use getmi_scale_cast with a caller workspace)", M, D, need, SCALE_CAST_SCRATCH_BYTES);
    float* scratch = nullptr;
    if (int rc = scale_cast_scratch(&scratch)) return rc;
    // every partial row the reduction reads is written by this call's first kernel: nothing has to start at zero
    return getmi_scale_cast_impl(dx, rowscale, rows_per_sample, dz, dbias, accumulate, M, D, scratch, SCALE_CAST_SCRATCH_BYTES, st);
}

int getmi_head_fwd_impl(const bf16_t* hf, const float* W, const float* bias, float* logits, float* pred, int B, int N,
                        int D, int OUT, hipStream_t st) {
    GETMI_CHECK_ARG(D == 256 || D == 512 || D == 768 || D == 1024, "head_fwd: D=%d unsupported (256, 512, 768, 1024)", D);
    GETMI_CHECK_ARG(OUT == 1 || OUT == 2, "head_fwd: output_dim=%d unsupported (1 or 2)", OUT);
    const int g = rows_grid(B * N);
#define HEAD_FWD(NV)                                                                                                      \
    do {                                                                                                                  \
        if (OUT == 1) hipLaunchKernelGGL((head_fwd_kernel<NV, 1>), dim3(g), dim3(256), 0, st, hf, W, bias, logits, pred, B, N); \
        else hipLaunchKernelGGL((head_fwd_kernel<NV, 2>), dim3(g), dim3(256), 0, st, hf, W, bias, logits, pred, B, N);    \
    } while (0)
    switch (D / 256) {
        case 1: HEAD_FWD(1); break;
        case 2: HEAD_FWD(2); break;
        case 3: HEAD_FWD(3); break;
        default: HEAD_FWD(4); break;
    }
#undef HEAD_FWD
    GETMI_CHECK_LAUNCH("head_fwd");
    return GETMI_OK;
}

size_t getmi_head_bwd_workspace_bytes_impl(int B, int N, int D, int OUT) {
    return (size_t)rows_grid(B * (N + 1)) * (OUT * D + OUT) * sizeof(float);
}

// dwb: [OUT*D + OUT] = {dWh (row-major [OUT, D]), dbh}
int getmi_head_bwd_impl(const float* dpred, const float* logits, const bf16_t* hf, const float* W, bf16_t* dh,
                        float* dwb, int accumulate, int B, int N, int D, int OUT, void* ws, size_t ws_bytes,
                        hipStream_t st) {
    GETMI_CHECK_ARG(D == 256 || D == 512 || D == 768 || D == 1024, "head_bwd: D=%d unsupported (256, 512, 768, 1024)", D);
    GETMI_CHECK_ARG(OUT == 1 || OUT == 2, "head_bwd: output_dim=%d unsupported (1 or 2)", OUT);
    GETMI_CHECK_ARG(ws_bytes >= getmi_head_bwd_workspace_bytes_impl(B, N, D, OUT), "head_bwd: workspace too small");
    const int g = rows_grid(B * (N + 1));
    float* partial = reinterpret_cast<float*>(ws);
#define HEAD_BWD(NV)                                                                                                      \
    do {                                                                                                                  \
        if (OUT == 1) hipLaunchKernelGGL((head_bwd_kernel<NV, 1>), dim3(g), dim3(256), 0, st, dpred, logits, hf, W, dh, partial, B, N); \
        else hipLaunchKernelGGL((head_bwd_kernel<NV, 2>), dim3(g), dim3(256), 0, st, dpred, logits, hf, W, dh, partial, B, N); \
    } while (0)
    switch (D / 256) {
        case 1: HEAD_BWD(1); break;
        case 2: HEAD_BWD(2); break;
        case 3: HEAD_BWD(3); break;
        default: HEAD_BWD(4); break;
    }
#undef HEAD_BWD
    GETMI_CHECK_LAUNCH("head_bwd");
    return getmi_colreduce_impl(partial, dwb, nullptr, nullptr, g, OUT * D + OUT, accumulate, st);
}

size_t getmi_loss_workspace_bytes_impl(void) { return 256 * sizeof(float); }

int getmi_loss_fwd_impl(int kind, const float* pred, const float* obs, float* loss, float* grad_unit, size_t R, int C,
                        int ldp, int ldo, int ldg, float weight, float eps, void* ws, size_t ws_bytes, hipStream_t st) {
    GETMI_CHECK_ARG(kind == 0 || kind == 1, "loss: kind=%d", kind);
    GETMI_CHECK_ARG(R > 0 && C > 0 && ldp >= C && ldo >= C && ldg >= C, "loss: bad shape");
    GETMI_CHECK_ARG(ws_bytes >= getmi_loss_workspace_bytes_impl(), "loss: workspace too small");
    const size_t n = R * C;
    int g = (int)((n + 1023) / 1024);
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    float* partial = reinterpret_cast<float*>(ws);
    hipLaunchKernelGGL(loss_partial_kernel, dim3(g), dim3(256), 0, st, pred, obs, grad_unit, partial, R, C, ldp, ldo, ldg,
                       kind, eps);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, partial, g, 1.0f / (float)n, weight, loss);
    GETMI_CHECK_LAUNCH("loss");
    return GETMI_OK;
}

int getmi_scale_by_scalar_impl(const float* in, const float* scalar, float weight, float* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, in, scalar, weight, out, n);
    GETMI_CHECK_LAUNCH("scale_by_scalar");
    return GETMI_OK;
}

int getmi_gather_rows_impl(const bf16_t* src, const int64_t* idx, bf16_t* dst, int n, int D, int scatter, hipStream_t st) {
    GETMI_CHECK_ARG(D % 8 == 0, "gather_rows: D=%d must be a multiple of 8", D);
    if (n <= 0) return GETMI_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows_grid(n)), dim3(256), 0, st, src, idx, dst, n, D, scatter);
    GETMI_CHECK_LAUNCH("gather_rows");
    return GETMI_OK;
}

size_t getmi_colsum_workspace_bytes_impl(int N) { return (size_t)256 * N * sizeof(float); }

int getmi_colsum_bf16_impl(const bf16_t* X, float* out, int accumulate, int M, int N, int ld, void* ws, size_t ws_bytes,
                           hipStream_t st) {
    GETMI_CHECK_ARG(N % 4 == 0 && ld % 4 == 0, "colsum: N and ld must be multiples of 4");
    GETMI_CHECK_ARG(ws_bytes >= getmi_colsum_workspace_bytes_impl(N), "colsum: workspace too small");
    int gy = M < 256 ? M : 256;
    float* partial = reinterpret_cast<float*>(ws);
    if (N % 8 == 0 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && M >= 1024) {
        gy = 256;   // 4 rows per block pass
        hipLaunchKernelGGL(colsum_bf16_wide_kernel, dim3((N + 511) / 512, gy), dim3(256), 0, st, X, partial, M, N, ld);
    } else {
        hipLaunchKernelGGL(colsum_bf16_kernel, dim3((N + 1023) / 1024, gy), dim3(256), 0, st, X, partial, M, N, ld);
    }
    GETMI_CHECK_LAUNCH("colsum");
    return getmi_colreduce_impl(partial, out, nullptr, nullptr, gy, N, accumulate, st);
}
