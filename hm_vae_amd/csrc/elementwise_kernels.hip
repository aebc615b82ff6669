// elementwise_kernels.hip — HBM-bound SoA kernels of the hm-vae hot path (gfx950).
//
// Reference code replaced (paths relative to the reference root):
//   skeleton.py:228-231   SkeletonPool.forward   (dense matmul with a constant 1/len matrix)
//   skeleton.py:258-261   SkeletonUnpool.forward (dense matmul with a constant 0/1 matrix)
//   seq_two_hier_sa_vae.py:129,256  nn.LeakyReLU(0.2);  :235-240  nn.Upsample(x2, linear)
//   seq_two_hier_sa_vae.py:419-428  reparametrize, kl_loss;  :408-411 total loss
//   trainer_motion_vae.py:29-31,92  torch.optim.Adam step
// All activations are [row][Bp] float with the batch innermost; every thread moves 16 bytes.
#include <hip/hip_runtime.h>
#include "hmvae.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int grid_for(long long n_items, int per_block = 256, int cap = 8192) {
    long long nb = (n_items + per_block - 1) / per_block;
    if (nb < 1) nb = 1;
    if (nb > cap) nb = cap;
    return (int)nb;
}
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH)

__device__ __forceinline__ f32x4 leaky4(f32x4 v, float slope) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
    return v;
}

// ---------------- pool (+LeakyReLU) ----------------
__global__ __launch_bounds__(256) void pool_leaky_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              const int* __restrict__ e1, const int* __restrict__ e2,
                                                              int n_out_edges, int cht, int bq, float slope) {
    const long long total = (long long)n_out_edges * cht * bq;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int q = (int)(idx % bq);
        const long long row = idx / bq;
        const int e = (int)(row / cht);
        const int rem = (int)(row - (long long)e * cht);
        const int a = e1[e], b = e2[e];
        f32x4 v = *(const f32x4*)(in + ((size_t)a * cht + rem) * (bq * 4) + q * 4);
        if (b >= 0) {
            f32x4 w = *(const f32x4*)(in + ((size_t)b * cht + rem) * (bq * 4) + q * 4);
            v = (v + w) * 0.5f;
        }
        *(f32x4*)(out + (size_t)row * (bq * 4) + q * 4) = leaky4(v, slope);
    }
}

extern "C" int hmvae_pool_leaky_fwd(const float* in, float* out, const int32_t* src_e1, const int32_t* src_e2,
                                    int32_t n_out_edges, int32_t ch, int32_t T, int32_t bp, float slope,
                                    hmvae_stream_t stream) {
    if (!in || !out || bp <= 0 || (bp & 3)) return HMVAE_ERR_ARG;
    const long long total = (long long)n_out_edges * ch * T * (bp / 4);
    hipLaunchKernelGGL(pool_leaky_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, src_e1,
                       src_e2, n_out_edges, ch * T, bp / 4, slope);
    return LAUNCH_OK();
}

__global__ __launch_bounds__(256) void pool_leaky_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ outv,
                                                              float* __restrict__ din, const int* __restrict__ dst_e,
                                                              const float* __restrict__ w_e, int n_in_edges, int cht,
                                                              int bq, float slope) {
    const long long total = (long long)n_in_edges * cht * bq;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int q = (int)(idx % bq);
        const long long row = idx / bq;
        const int e = (int)(row / cht);
        const int rem = (int)(row - (long long)e * cht);
        const size_t o = ((size_t)dst_e[e] * cht + rem) * (bq * 4) + q * 4;
        const f32x4 g = *(const f32x4*)(dout + o);
        const f32x4 y = *(const f32x4*)(outv + o);
        const float w = w_e[e];
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = g[k] * (y[k] > 0.f ? 1.f : slope) * w;
        *(f32x4*)(din + (size_t)row * (bq * 4) + q * 4) = r;
    }
}

extern "C" int hmvae_pool_leaky_bwd(const float* dout, const float* out, float* din, const int32_t* dst_e,
                                    const float* w_e, int32_t n_in_edges, int32_t ch, int32_t T, int32_t bp, float slope,
                                    hmvae_stream_t stream) {
    if (!dout || !out || !din || bp <= 0 || (bp & 3)) return HMVAE_ERR_ARG;
    const long long total = (long long)n_in_edges * ch * T * (bp / 4);
    hipLaunchKernelGGL(pool_leaky_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, out, din,
                       dst_e, w_e, n_in_edges, ch * T, bp / 4, slope);
    return LAUNCH_OK();
}

// ---------------- upsample (x2 linear, align_corners=False) + unpool ----------------
__global__ __launch_bounds__(256) void up_unpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             const int* __restrict__ src_e, int n_out_edges, int ch,
                                                             int t_in, int up, int bq) {
    const int t_out = up ? 2 * t_in : t_in;
    const long long total = (long long)n_out_edges * ch * t_out * bq;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int q = (int)(idx % bq);
        const long long row = idx / bq;
        const int t = (int)(row % t_out);
        const long long ec = row / t_out;
        const int c = (int)(ec % ch);
        const int e = (int)(ec / ch);
        const float* base = in + ((size_t)src_e[e] * ch + c) * t_in * (bq * 4) + q * 4;
        f32x4 v;
        if (!up) {
            v = *(const f32x4*)(base + (size_t)t * (bq * 4));
        } else {
            const int i = t >> 1;
            int i0, i1;
            float w0, w1;
            if (t & 1) {
                i0 = i;
                i1 = i + 1 < t_in ? i + 1 : t_in - 1;
                w0 = 0.75f;
                w1 = 0.25f;
            } else {
                i0 = i > 0 ? i - 1 : 0;
                i1 = i;
                w0 = 0.25f;
                w1 = 0.75f;
            }
            const f32x4 a = *(const f32x4*)(base + (size_t)i0 * (bq * 4));
            const f32x4 b = *(const f32x4*)(base + (size_t)i1 * (bq * 4));
            v = a * w0 + b * w1;
        }
        *(f32x4*)(out + (size_t)row * (bq * 4) + q * 4) = v;
    }
}

extern "C" int hmvae_upsample_unpool_fwd(const float* in, float* out, const int32_t* src_e, int32_t n_out_edges,
                                         int32_t ch, int32_t T_in, int32_t up, int32_t bp, hmvae_stream_t stream) {
    if (!in || !out || bp <= 0 || (bp & 3)) return HMVAE_ERR_ARG;
    const long long total = (long long)n_out_edges * ch * (up ? 2 * T_in : T_in) * (bp / 4);
    hipLaunchKernelGGL(up_unpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, src_e,
                       n_out_edges, ch, T_in, up, bp / 4);
    return LAUNCH_OK();
}

__global__ __launch_bounds__(256) void up_unpool_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                             const int* __restrict__ fine_off,
                                                             const int* __restrict__ fine_list, int n_in_edges, int ch,
                                                             int t_in, int up, int bq) {
    const int t_out = up ? 2 * t_in : t_in;
    const long long total = (long long)n_in_edges * ch * t_in * bq;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int q = (int)(idx % bq);
        const long long row = idx / bq;
        const int i = (int)(row % t_in);
        const long long ec = row / t_in;
        const int c = (int)(ec % ch);
        const int e = (int)(ec / ch);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = fine_off[e]; k < fine_off[e + 1]; ++k) {
            const float* base = dout + ((size_t)fine_list[k] * ch + c) * t_out * (bq * 4) + q * 4;
            if (!up) {
                acc += *(const f32x4*)(base + (size_t)i * (bq * 4));
            } else {
                // out[2i] = .25 in[i-1] + .75 in[i] (in[-1]:=in[0]);  out[2i+1] = .75 in[i] + .25 in[i+1] (in[T]:=in[T-1])
                const f32x4 e0 = *(const f32x4*)(base + (size_t)(2 * i) * (bq * 4));
                const f32x4 o0 = *(const f32x4*)(base + (size_t)(2 * i + 1) * (bq * 4));
                f32x4 s = (e0 + o0) * 0.75f;
                if (i + 1 < t_in) s += *(const f32x4*)(base + (size_t)(2 * i + 2) * (bq * 4)) * 0.25f;
                else s += o0 * 0.25f;
                if (i > 0) s += *(const f32x4*)(base + (size_t)(2 * i - 1) * (bq * 4)) * 0.25f;
                else s += e0 * 0.25f;
                acc += s;
            }
        }
        *(f32x4*)(din + (size_t)row * (bq * 4) + q * 4) = acc;
    }
}

extern "C" int hmvae_upsample_unpool_bwd(const float* dout, float* din, const int32_t* fine_off, const int32_t* fine_list,
                                         int32_t n_in_edges, int32_t ch, int32_t T_in, int32_t up, int32_t bp,
                                         hmvae_stream_t stream) {
    if (!dout || !din || bp <= 0 || (bp & 3)) return HMVAE_ERR_ARG;
    const long long total = (long long)n_in_edges * ch * T_in * (bp / 4);
    hipLaunchKernelGGL(up_unpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, din,
                       fine_off, fine_list, n_in_edges, ch, T_in, up, bp / 4);
    return LAUNCH_OK();
}

// ---------------- LeakyReLU backward from the output ----------------
__global__ __launch_bounds__(256) void leaky_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ outv,
                                                         float* __restrict__ dy, long long len4, long long n_blocks,
                                                         long long stride4, float slope) {
    const long long total = len4 * n_blocks;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long blk = idx / len4;
        const long long o = blk * stride4 + (idx - blk * len4);
        const f32x4 g = ((const f32x4*)dout)[o];
        const f32x4 y = ((const f32x4*)outv)[o];
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = g[k] * (y[k] > 0.f ? 1.f : slope);
        ((f32x4*)dy)[o] = r;
    }
}

extern "C" int hmvae_leaky_bwd(const float* dout, const float* out, float* dy, int64_t block_len, int64_t n_blocks,
                               int64_t block_stride, float slope, hmvae_stream_t stream) {
    if (!dout || !out || !dy || block_len <= 0 || n_blocks <= 0 || (block_len & 3) || (block_stride & 3)) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(leaky_bwd_kernel, dim3(grid_for(block_len / 4 * n_blocks)), dim3(256), 0, (hipStream_t)stream, dout,
                       out, dy, (long long)(block_len / 4), (long long)n_blocks, (long long)(block_stride / 4), slope);
    return LAUNCH_OK();
}

// ---------------- block reduction helper ----------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// returns the block total in thread 0 (256 threads)
__device__ __forceinline__ float block_sum256(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------- row sums (bias gradients) ----------------
__global__ __launch_bounds__(256) void row_sums_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        const int* __restrict__ rows, int rows_per_out, int row_step,
                                                        int n_cols, int accumulate) {
    __shared__ float sh[4];
    const int o = blockIdx.x;
    const int base = rows[o];
    float s = 0.f;
    const int nq = n_cols >> 2;
    // the 4 waves take different rows (independent loads in flight), lanes take 16-byte pieces of a row
    for (int k = threadIdx.x >> 6; k < rows_per_out; k += 4) {
        const f32x4* p = (const f32x4*)(src + (size_t)(base + k * row_step) * n_cols);
        for (int q = threadIdx.x & 63; q < nq; q += 64) {
            const f32x4 v = p[q];
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    const float tot = block_sum256(s, sh);
    if (threadIdx.x == 0) dst[o] = accumulate ? dst[o] + tot : tot;
}

// rows[i] = first row of output i; rows_per_out rows spaced `row_step` apart are summed.
extern "C" int hmvae_row_sums(const float* src, float* dst, const int32_t* rows, int32_t n_out, int32_t rows_per_out,
                              int32_t row_step, int32_t n_cols, int32_t accumulate, hmvae_stream_t stream) {
    if (!src || !dst || !rows || n_out <= 0 || rows_per_out <= 0 || (n_cols & 3)) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(row_sums_kernel, dim3(n_out), dim3(256), 0, (hipStream_t)stream, src, dst, rows, rows_per_out,
                       row_step, n_cols, accumulate);
    return LAUNCH_OK();
}

__global__ __launch_bounds__(256) void row_sums_scaled_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               const int* __restrict__ rows, const float* __restrict__ scale,
                                                               int rows_per_out, int row_step, int n_cols) {
    __shared__ float sh[4];
    const int o = blockIdx.x;
    const int base = rows[o];
    float s = 0.f;
    const int nq = n_cols >> 2;
    for (int k = threadIdx.x >> 6; k < rows_per_out; k += 4) {
        const f32x4* p = (const f32x4*)(src + (size_t)(base + k * row_step) * n_cols);
        for (int q = threadIdx.x & 63; q < nq; q += 64) {
            const f32x4 v = p[q];
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    const float tot = block_sum256(s, sh);
    if (threadIdx.x == 0) dst[o] = scale[o] * tot;
}

extern "C" int hmvae_row_sums_scaled(const float* src, float* dst, const int32_t* rows, const float* scale, int32_t n_out,
                                     int32_t rows_per_out, int32_t row_step, int32_t n_cols, hmvae_stream_t stream) {
    if (!src || !dst || !rows || !scale || n_out <= 0 || rows_per_out <= 0 || (n_cols & 3)) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(row_sums_scaled_kernel, dim3(n_out), dim3(256), 0, (hipStream_t)stream, src, dst, rows, scale,
                       rows_per_out, row_step, n_cols);
    return LAUNCH_OK();
}

__global__ __launch_bounds__(256) void row_sums_multi_kernel(const hmvae_rowsum_op* __restrict__ ops, int n_ops) {
    __shared__ float sh[4];
    int oi = 0;
    for (int i = 1; i < n_ops; ++i)
        if ((int)blockIdx.x >= ops[i].block_start) oi = i;
    const hmvae_rowsum_op op = ops[oi];
    const int o = blockIdx.x - op.block_start;
    if (o >= op.n_out) return;
    const int base = op.rows[o];
    float s = 0.f;
    const int nq = op.n_cols >> 2;
    for (int k = threadIdx.x >> 6; k < op.rows_per_out; k += 4) {
        const f32x4* p = (const f32x4*)(op.src + (size_t)(base + k * op.row_step) * op.n_cols);
        for (int q = threadIdx.x & 63; q < nq; q += 64) {
            const f32x4 v = p[q];
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    const float tot = block_sum256(s, sh);
    if (threadIdx.x == 0) op.dst[o] = op.accumulate ? op.dst[o] + tot : tot;
}

extern "C" int hmvae_row_sums_multi(const hmvae_rowsum_op* ops_dev, int32_t n_ops, int32_t total_blocks,
                                    hmvae_stream_t stream) {
    if (!ops_dev || n_ops <= 0 || total_blocks <= 0) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(row_sums_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, ops_dev, n_ops);
    return LAUNCH_OK();
}

// ---------------- AoS (B,T,C) <-> SoA (C,T,Bp) ----------------
__global__ __launch_bounds__(256) void aos_to_soa_kernel(const float* __restrict__ aos, float* __restrict__ soa, int B,
                                                          int T, int C, int bp) {
    __shared__ float tile[64][65];
    const int t = blockIdx.z;
    const int b0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int b = b0 + r, c = c0 + tx;
        tile[r][tx] = (b < B && c < C) ? aos[((size_t)b * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, b = b0 + tx;
        if (c < C && b < bp) soa[((size_t)c * T + t) * bp + b] = tile[tx][r];
    }
}

extern "C" int hmvae_aos_to_soa(const float* aos, float* soa, int32_t B, int32_t T, int32_t C, int32_t bp,
                                hmvae_stream_t stream) {
    if (!aos || !soa || B <= 0 || bp < B || (bp & 63) || T <= 0 || T > 65535) return HMVAE_ERR_ARG;
    dim3 grid(bp / 64, (C + 63) / 64, T);
    hipLaunchKernelGGL(aos_to_soa_kernel, grid, dim3(256), 0, (hipStream_t)stream, aos, soa, B, T, C, bp);
    return LAUNCH_OK();
}

__global__ __launch_bounds__(256) void soa_to_aos_kernel(const float* __restrict__ soa, float* __restrict__ aos, int B,
                                                          int T, int C, int bp) {
    __shared__ float tile[64][65];
    const int t = blockIdx.z;
    const int b0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, b = b0 + tx;
        tile[r][tx] = (c < C && b < bp) ? soa[((size_t)c * T + t) * bp + b] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int b = b0 + r, c = c0 + tx;
        if (b < B && c < C) aos[((size_t)b * T + t) * C + c] = tile[tx][r];
    }
}

extern "C" int hmvae_soa_to_aos(const float* soa, float* aos, int32_t B, int32_t T, int32_t C, int32_t bp,
                                hmvae_stream_t stream) {
    if (!aos || !soa || B <= 0 || bp < B || (bp & 63) || T <= 0 || T > 65535) return HMVAE_ERR_ARG;
    dim3 grid(bp / 64, (C + 63) / 64, T);
    hipLaunchKernelGGL(soa_to_aos_kernel, grid, dim3(256), 0, (hipStream_t)stream, soa, aos, B, T, C, bp);
    return LAUNCH_OK();
}

// ---------------- reparametrize + KL ----------------
// grid: (bp/256 rounded up, k_edges, d); thread = one batch column of one latent component of one edge.  (One thread
// walking the d components of its column was a rolled loop of d dependent memory round trips: 12-17 us for a few
// hundred KB, on the critical path between encoder and decoder.)
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(const float* __restrict__ dist, const float* __restrict__ eps,
                                                              float* __restrict__ z, float* __restrict__ kl_partial,
                                                              int k_edges, int dl, int B, int bp, int sample) {
    __shared__ float sh[4];
    const int b = blockIdx.x * 256 + threadIdx.x;
    const int e = blockIdx.y;
    const int j = blockIdx.z;
    float kl = 0.f;
    if (b < bp) {
        const bool live = b < B;
        const float mu = dist[(size_t)(e * 2 * dl + j) * bp + b];
        const float lv = dist[(size_t)(e * 2 * dl + dl + j) * bp + b];
        float zz = mu;
        if (sample && live) zz = eps[((size_t)b * k_edges + e) * dl + j] * expf(0.5f * lv) + mu;
        z[(size_t)(e * dl + j) * bp + b] = live ? zz : 0.f;
        if (live) kl = -0.5f * (1.f + lv - mu * mu - expf(lv));
    }
    const float tot = block_sum256(kl, sh);
    if (threadIdx.x == 0) kl_partial[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
}

extern "C" int hmvae_reparam_kl_fwd(const float* dist, const float* eps, float* z, float* kl_partial, int32_t k_edges,
                                    int32_t d, int32_t B, int32_t bp, int32_t sample, hmvae_stream_t stream) {
    if (!dist || !z || !kl_partial || (sample && !eps) || bp < B || d <= 0 || d > 65535) return HMVAE_ERR_ARG;
    dim3 grid((bp + 255) / 256, k_edges, d);
    hipLaunchKernelGGL(reparam_kl_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dist, eps, z, kl_partial, k_edges,
                       d, B, bp, sample);
    return LAUNCH_OK();
}

__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(const float* __restrict__ dist, const float* __restrict__ eps,
                                                              const float* __restrict__ dz, float* __restrict__ ddist,
                                                              int k_edges, int dl, int B, int bp, int use_dz,
                                                              float kl_scale) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    const int e = blockIdx.y;
    const int j = blockIdx.z;
    if (b >= bp) return;
    const bool live = b < B;
    const size_t rmu = (size_t)(e * 2 * dl + j) * bp + b;
    const size_t rlv = (size_t)(e * 2 * dl + dl + j) * bp + b;
    float gmu = 0.f, glv = 0.f;
    if (live) {
        const float mu = dist[rmu], lv = dist[rlv];
        gmu = kl_scale * mu;
        glv = kl_scale * 0.5f * (expf(lv) - 1.f);
        if (use_dz) {
            const float g = dz[(size_t)(e * dl + j) * bp + b];
            gmu += g;
            if (eps) glv += g * eps[((size_t)b * k_edges + e) * dl + j] * 0.5f * expf(0.5f * lv);
        }
    }
    ddist[rmu] = gmu;
    ddist[rlv] = glv;
}

extern "C" int hmvae_reparam_kl_bwd(const float* dist, const float* eps, const float* dz, float* ddist, int32_t k_edges,
                                    int32_t d, int32_t B, int32_t bp, int32_t use_dz, float kl_scale,
                                    hmvae_stream_t stream) {
    if (!dist || !ddist || (use_dz && !dz) || bp < B || d <= 0 || d > 65535) return HMVAE_ERR_ARG;
    dim3 grid((bp + 255) / 256, k_edges, d);
    hipLaunchKernelGGL(reparam_kl_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dist, eps, dz, ddist, k_edges, d, B,
                       bp, use_dz, kl_scale);
    return LAUNCH_OK();
}

// ---------------- standard-normal noise for the reparametrisation (seq_two_hier_sa_vae.py:359-366: torch.randn_like) ----------------
// Philox4x32-10 (counter = element index / 4 and the launch count, key = seed) + Box-Muller, four values per thread.  The
// launch count lives on the device (state[0]) and is advanced by the LAST workgroup to finish -- every workgroup reads it
// first, so a captured graph draws fresh noise at every replay without a host-side seed update in front of it (torch's
// generator costs two scalar fill launches before every replay of a graph that holds torch.randn, and one launch per level).
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, long long n, unsigned long long seed,
                                                    unsigned long long* __restrict__ state) {
    __shared__ unsigned long long s_launch;
    __shared__ int s_last;
    if (threadIdx.x == 0) s_launch = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned long long launch = s_launch;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; 4 * q < n; q += (long long)gridDim.x * 256) {
        unsigned c[4] = {(unsigned)q, (unsigned)((unsigned long long)q >> 32), (unsigned)launch, (unsigned)(launch >> 32)};
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = (float)(c[2 * h] >> 8) * 5.9604645e-8f + 2.9802322e-8f;       // (0, 1)
            const float u2 = (float)(c[2 * h + 1] >> 8) * 5.9604645e-8f;                   // [0, 1)
            const float rad = sqrtf(-2.f * logf(u1));
            float sn, cs;
            sincosf(6.2831853071795865f * u2, &sn, &cs);
            z[2 * h] = rad * cs;
            z[2 * h + 1] = rad * sn;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < n) out[4 * q + e] = z[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long done = __hip_atomic_fetch_add(state + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = done == (unsigned long long)gridDim.x - 1;
        if (s_last) {
            __hip_atomic_store(state + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(state, launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" int hmvae_randn(float* out, int64_t n, uint64_t seed, uint64_t* state, hmvae_stream_t stream) {
    if (!out || !state || n <= 0) return HMVAE_ERR_ARG;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, (long long)n,
                       (unsigned long long)seed, (unsigned long long*)state);
    return LAUNCH_OK();
}

// ---------------- fixed-order final sums ----------------
__global__ __launch_bounds__(256) void finalize_sums_kernel(const float* __restrict__ partial, int n_partial, int stride,
                                                             int k, float scale, float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[(size_t)i * stride + k];
    const float tot = block_sum256(s, sh);
    if (threadIdx.x == 0) out[0] = tot * scale;
}

extern "C" int hmvae_finalize_sums(const float* partial, int32_t n_partial, int32_t stride, int32_t k, float scale,
                                   float* out, hmvae_stream_t stream) {
    if (!partial || !out || n_partial <= 0) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_partial, stride, k,
                       scale, out);
    return LAUNCH_OK();
}

// ---------------- test-time optimisation: l2_criterion against an anchor (seq_two_hier_sa_vae.py:1366-1373) ----------------
// mean((p - p0)^2) over one tensor (p0 NULL: the anchor is zero, the latent regulariser :1366) and its gradient ADDED to g:
//     g[i] += gscale * (p[i] - p0[i]),   partial[block] = pscale * sum (p - p0)^2 over the block's grid-stride share
// A fixed grid (L2A_BLOCKS) keeps the partial layout independent of n: several tensors append their blocks to one buffer
// and one finalize launch sums them in a fixed order.  (16 blocks took 73 us on the decoder's 1.8 M-float tensors and were
// half of an iteration's GPU time; 128 blocks stream them at HBM/L2 speed.)
constexpr int L2A_BLOCKS = 128;
__global__ __launch_bounds__(256) void l2_anchor_kernel(const float* __restrict__ p, const float* __restrict__ p0, long long n,
                                                         float gscale, float pscale, float* __restrict__ g,
                                                         float* __restrict__ partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)L2A_BLOCKS * 256) {
        const float e = p[i] - (p0 ? p0[i] : 0.f);
        acc += e * e;
        if (g) g[i] += gscale * e;
    }
    const float tot = block_sum256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot * pscale;
}

extern "C" int hmvae_l2_anchor(const float* p, const float* p0, int64_t n, float grad_scale, float partial_scale, float* g,
                               float* partial, int32_t* n_blocks_out, hmvae_stream_t stream) {
    if (n_blocks_out) *n_blocks_out = L2A_BLOCKS;
    if (!p && !p0 && !g && !partial) return HMVAE_OK;                        // size query
    if (!p || !partial || n <= 0) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(l2_anchor_kernel, dim3(L2A_BLOCKS), dim3(256), 0, (hipStream_t)stream, p, p0, (long long)n, grad_scale,
                       partial_scale, g, partial);
    return LAUNCH_OK();
}

__global__ __launch_bounds__(256) void vae_loss_finalize_kernel(const float* __restrict__ mse, int n_mse,
                                                                 const float* __restrict__ kl0, int n_kl0,
                                                                 const float* __restrict__ kld, int n_kld, float inv_n6,
                                                                 float inv_nR, float inv_nP, float inv_r0, float inv_rd,
                                                                 float w6, float wR, float wP, float wkl, float wskl,
                                                                 float* __restrict__ out) {
    __shared__ float sh[4];
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < n_mse; i += 256) {
        s[0] += mse[(size_t)i * 3];
        s[1] += mse[(size_t)i * 3 + 1];
        s[2] += mse[(size_t)i * 3 + 2];
    }
    for (int i = threadIdx.x; i < n_kl0; i += 256) s[3] += kl0[i];
    for (int i = threadIdx.x; i < n_kld; i += 256) s[4] += kld[i];
    float r[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) r[k] = block_sum256(s[k], sh);
    if (threadIdx.x == 0) {
        const float l6 = r[0] * inv_n6, lR = r[1] * inv_nR, lP = r[2] * inv_nP;
        const float k0 = r[3] * inv_r0, kd = r[4] * inv_rd;
        const float lkl = wkl * kd + wskl * k0;
        out[0] = w6 * l6 + wR * lR + wP * lP + wkl * kd + wskl * k0;
        out[1] = lkl;
        out[2] = l6;
        out[3] = lR;
        out[4] = lP;
        out[5] = k0;
        out[6] = kd;
        out[7] = 0.f;
    }
}

extern "C" int hmvae_vae_loss_finalize(const float* mse_partial, int32_t n_mse, const float* kl0_partial, int32_t n_kl0,
                                       const float* kld_partial, int32_t n_kld, float inv_n6, float inv_nR, float inv_nP,
                                       float inv_rows0, float inv_rowsd, float w6, float wR, float wP, float w_kl,
                                       float w_skl, float* out8, hmvae_stream_t stream) {
    if (!mse_partial || !kl0_partial || !kld_partial || !out8 || n_mse <= 0 || n_kl0 <= 0 || n_kld <= 0) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(vae_loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mse_partial, n_mse, kl0_partial,
                       n_kl0, kld_partial, n_kld, inv_n6, inv_nR, inv_nP, inv_rows0, inv_rowsd, w6, wR, wP, w_kl, w_skl, out8);
    return LAUNCH_OK();
}

// ---------------- TrajectoryModel losses (trajectory_pred_model.py:237-262, 292-305) ----------------
// One thread per (component c, sequence b): d[t] = v[t] - g[t] (normalised root velocity); the trajectory term compares
// running sums of the de-standardised velocities from frame 1 on (the mean and the pose cancel in pred - gt; all 24
// joints carry the same offset, so the mean over (T,B,24,3) equals the mean over (T,B,3)):
//     D[t] = std_c * sum_{1<=s<=t} d[s],  D[0] = 0
//     L_v = sum d^2 / (3TB),  L_t = sum D^2 / (3TB)
//     dL/dv[t] = gv * d[t] + (t >= 1) * gt * std_c * sum_{t'>=t} D[t']        gv = 2 w_v/(3TB), gt = 2 w_t/(3TB)
// Pass 1 parks D in dv, pass 2 walks back with the suffix sum.  Rows: c*T + t, batch innermost (SoA).
__global__ __launch_bounds__(64) void traj_loss_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                        const float* __restrict__ std3, int T, int B, int bp, float gv,
                                                        float gt, float* __restrict__ dv, float* __restrict__ partial) {
    const int chunks = bp >> 6;
    const int c = blockIdx.x / chunks;
    const int b = (blockIdx.x - c * chunks) * 64 + threadIdx.x;
    const float sd = std3[c];
    const size_t base = (size_t)c * T * bp + b;
    float sv = 0.f, st = 0.f;
    if (b < B) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) {
            const size_t o = base + (size_t)t * bp;
            const float d = v[o] - g[o];
            sv += d * d;
            if (t >= 1) acc += sd * d;
            st += acc * acc;
            dv[o] = acc;
        }
        float suf = 0.f;
        for (int t = T - 1; t >= 0; --t) {
            const size_t o = base + (size_t)t * bp;
            const float d = v[o] - g[o];
            suf += dv[o];
            dv[o] = gv * d + (t >= 1 ? gt * sd * suf : 0.f);
        }
    } else {
        for (int t = 0; t < T; ++t) dv[base + (size_t)t * bp] = 0.f;
    }
    sv = wave_sum(sv);
    st = wave_sum(st);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sv;
        partial[2 * blockIdx.x + 1] = st;
    }
}

extern "C" int hmvae_traj_loss(const float* v, const float* g, const float* std3, int32_t T, int32_t B, int32_t bp,
                               float grad_v, float grad_t, float* dv, float* partial, int32_t* n_blocks_out,
                               hmvae_stream_t stream) {
    if (bp <= 0 || (bp & 63) || B <= 0 || B > bp || T <= 0) return HMVAE_ERR_ARG;
    const int n_blocks = 3 * (bp >> 6);
    if (n_blocks_out) *n_blocks_out = n_blocks;
    if (!v && !g && !dv && !partial) return HMVAE_OK;          // size query
    if (!v || !g || !std3 || !dv || !partial) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(traj_loss_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, v, g, std3, T, B, bp, grad_v,
                       grad_t, dv, partial);
    return LAUNCH_OK();
}

// out = the nine values TrajectoryModel.forward returns (:265): [total, 0, 0, 0, 0, 0, rec_root_v, 0, rec_root_trans]
__global__ __launch_bounds__(256) void traj_loss_finalize_kernel(const float* __restrict__ partial, int n_blocks,
                                                                  float inv_n, float wv, float wt, float* __restrict__ out) {
    __shared__ float sh[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n_blocks; i += 256) {
        a += partial[2 * i];
        b += partial[2 * i + 1];
    }
    const float sv = block_sum256(a, sh);
    const float st = block_sum256(b, sh);
    if (threadIdx.x < 9) {
        const float lv = sv * inv_n, lt = st * inv_n;
        float r = 0.f;
        if (threadIdx.x == 0) r = wv * lv + wt * lt;
        if (threadIdx.x == 6) r = lv;
        if (threadIdx.x == 8) r = lt;
        out[threadIdx.x] = r;
    }
}

extern "C" int hmvae_traj_loss_finalize(const float* partial, int32_t n_blocks, float inv_n, float w_v, float w_t,
                                        float* out9, hmvae_stream_t stream) {
    if (!partial || !out9 || n_blocks <= 0) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(traj_loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_blocks, inv_n, w_v,
                       w_t, out9);
    return LAUNCH_OK();
}

// ---------------- key-frame root loss of the test-time optimisation loop (seq_two_hier_sa_vae.py:1031-1045, :1376-1389) ----------------
// cal_key_frame_root_loss as released: the "key frames" are the indices j at which row 0 of the window's (T, 24) mask is
// non-zero -- i.e. the JOINTS observed in frame 0, used as FRAME indices 0..23 (the comment there says otherwise; the code
// is what is restated).  pred_root_trans[t] = root pose + sum_{s=1..t} (mean_c + std_c * v[s]) (gen_motion_w_trajectory
// :486-499; the root pose is the constant root offset and cancels), so for consecutive selected j' < j
//     e = sum_{s=j'+1..j} (mean_c + std_c * v[s]) - (gt[j] - gt[j']),   L = sum e^2 / N,   dL/dv[s] = g * std_c * e(s)
// with e(s) the term of the interval that holds s (0 outside every interval), g = 2 w / N, N = 3 * sum_b (K_b - 1) (host).
// One thread per (component c, window b); rows c*T + t (v, gt), j*T (mask: joint j at frame 0), batch innermost.
constexpr int KF_MAX = 24;
__global__ __launch_bounds__(64) void keyframe_root_loss_kernel(const float* __restrict__ v, const float* __restrict__ gt,
                                                                 const float* __restrict__ mask,
                                                                 const float* __restrict__ mean3,
                                                                 const float* __restrict__ std3, int T, int B, int bp,
                                                                 float g, float* __restrict__ dv, float* __restrict__ partial) {
    const int chunks = bp >> 6;
    const int c = blockIdx.x / chunks;
    const int b = (blockIdx.x - c * chunks) * 64 + threadIdx.x;
    const float sd = std3[c], mu = mean3[c];
    const size_t base = (size_t)c * T * bp + b;
    const int nj = T < KF_MAX ? T : KF_MAX;
    float sum = 0.f;
    for (int t = 0; t < T; ++t) dv[base + (size_t)t * bp] = 0.f;
    if (b < B) {
        float acc = 0.f;                 // de-standardised running sum up to frame j
        int jp = -1;                     // previous selected index
        float accp = 0.f, gtp = 0.f;
        for (int j = 0; j < nj; ++j) {
            if (j >= 1) acc += mu + sd * v[base + (size_t)j * bp];
            if (mask[(size_t)j * T * bp + b] != 0.f) {
                const float gj = gt[base + (size_t)j * bp];
                if (jp >= 0) {
                    const float e = (acc - accp) - (gj - gtp);
                    sum += e * e;
                    const float ge = g * sd * e;
                    for (int s = jp + 1; s <= j; ++s) dv[base + (size_t)s * bp] = ge;
                }
                jp = j;
                accp = acc;
                gtp = gj;
            }
        }
    }
    sum = wave_sum(sum);
    if (threadIdx.x == 0) partial[blockIdx.x] = sum;
}

extern "C" int hmvae_keyframe_root_loss(const float* v, const float* gt, const float* mask, const float* mean3,
                                        const float* std3, int32_t T, int32_t B, int32_t bp, float grad_scale, float* dv,
                                        float* partial, int32_t* n_blocks_out, hmvae_stream_t stream) {
    if (bp <= 0 || (bp & 63) || B <= 0 || B > bp || T <= 0) return HMVAE_ERR_ARG;
    const int n_blocks = 3 * (bp >> 6);
    if (n_blocks_out) *n_blocks_out = n_blocks;
    if (!v && !gt && !mask && !dv && !partial) return HMVAE_OK;          // size query
    if (!v || !gt || !mask || !mean3 || !std3 || !dv || !partial) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(keyframe_root_loss_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, v, gt, mask, mean3, std3, T,
                       B, bp, grad_scale, dv, partial);
    return LAUNCH_OK();
}

// ---------------- batch assembly from a device-resident dataset (utils_motion_vae.py:124-187) ----------------
// frames: the packed T_i x 579 sequences, back to back.  One workgroup builds MB_FRAMES consecutive frames of one clip:
// every source column is read once (coalesced) and lands in the 7 output tensors of the reference's batch tuple.
//   [0:144) 6D, [144:360) rotation matrices, [360:432) FK positions (raw -> seq_rot_pos, standardised -> seq_joint_pos),
//   [432:504) / [504:576) velocities (standardised), [576:579) root velocity (standardised)
// Standardisation is done in float64 and rounded once, like the reference (numpy float64 table, then .float()).
// rot_flag (random_root_rot_flag, :161-185): root matrix <- M * root, every joint's 6D re-read from the matrices
// ([col0 | col1]), root velocity <- standardise(M * raw root velocity).
constexpr int MB_FRAMES = 4;
constexpr int MB_COLS = 579;
__global__ __launch_bounds__(256) void motion_batch_kernel(const float* __restrict__ frames, long long n_rows,
                                                           const hmvae_clip* __restrict__ clips,
                                                           const double* __restrict__ mean, const double* __restrict__ stdv,
                                                           int T, int rot_flag, float* __restrict__ r6,
                                                           float* __restrict__ rmat, float* __restrict__ rot_pos,
                                                           float* __restrict__ joint_pos, float* __restrict__ lin_v,
                                                           float* __restrict__ ang_v, float* __restrict__ root_v) {
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * MB_FRAMES;
    const hmvae_clip cl = clips[b];
    for (int i = threadIdx.x; i < MB_FRAMES * MB_COLS; i += 256) {
        const int f = i / MB_COLS, c = i - f * MB_COLS;
        const int t = t0 + f;
        if (t >= T) break;
        long long row = cl.row0 + (long long)t * cl.stride;
        row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);          // descriptors are validated on the host
        const float* src = frames + row * MB_COLS;
        const size_t fo = (size_t)b * T + t;
        const float x = src[c];
        if (c < 144) {
            float v = x;
            if (rot_flag) {
                const int j = c / 6, r = c - j * 6, a = r / 3, ii = r - a * 3;   // [col a][row ii] of joint j
                if (j == 0) v = (cl.rot[ii * 3] * src[144 + a] + cl.rot[ii * 3 + 1] * src[147 + a]) + cl.rot[ii * 3 + 2] * src[150 + a];
                else v = src[144 + j * 9 + ii * 3 + a];
            }
            r6[fo * 144 + c] = v;
        } else if (c < 360) {
            const int k = c - 144;
            float v = x;
            if (rot_flag && k < 9) {
                const int ii = k / 3, jj = k - ii * 3;
                v = (cl.rot[ii * 3] * src[144 + jj] + cl.rot[ii * 3 + 1] * src[147 + jj]) + cl.rot[ii * 3 + 2] * src[150 + jj];
            }
            rmat[fo * 216 + k] = v;
        } else {
            const float n = (float)(((double)x - mean[c]) / stdv[c]);
            if (c < 432) {
                rot_pos[fo * 72 + (c - 360)] = x;
                joint_pos[fo * 72 + (c - 360)] = n;
            } else if (c < 504) {
                lin_v[fo * 72 + (c - 432)] = n;
            } else if (c < 576) {
                ang_v[fo * 72 + (c - 504)] = n;
            } else {
                const int ii = c - 576;
                float v = n;
                if (rot_flag) {
                    const float av = (cl.rot[ii * 3] * src[576] + cl.rot[ii * 3 + 1] * src[577]) + cl.rot[ii * 3 + 2] * src[578];
                    v = (float)(((double)av - mean[c]) / stdv[c]);
                }
                root_v[fo * 3 + ii] = v;
            }
        }
    }
}

extern "C" int hmvae_motion_batch(const float* frames, int64_t n_rows, const hmvae_clip* clips_dev, const double* mean,
                                  const double* stdv, int32_t B, int32_t T, int32_t rot_flag, float* r6, float* rmat,
                                  float* rot_pos, float* joint_pos, float* lin_v, float* ang_v, float* root_v,
                                  hmvae_stream_t stream) {
    if (!frames || !clips_dev || !mean || !stdv || !r6 || !rmat || !rot_pos || !joint_pos || !lin_v || !ang_v || !root_v)
        return HMVAE_ERR_ARG;
    if (n_rows <= 0 || B <= 0 || B > 65535 || T <= 0) return HMVAE_ERR_ARG;
    dim3 grid((unsigned)((T + MB_FRAMES - 1) / MB_FRAMES), (unsigned)B);
    hipLaunchKernelGGL(motion_batch_kernel, grid, dim3(256), 0, (hipStream_t)stream, frames, (long long)n_rows, clips_dev, mean,
                       stdv, T, rot_flag, r6, rmat, rot_pos, joint_pos, lin_v, ang_v, root_v);
    return LAUNCH_OK();
}

// ---------------- exact three-way bf16 split (operand planes of the bf16x3 weight-gradient GEMM) ----------------
// p0 = top 16 bits of x; r = x - p0 (exact); p1 = top 16 bits of r; p2 = top 16 bits of r - p1 (exact: <= 8 bits left)
typedef __attribute__((ext_vector_type(2))) __bf16 split_bf16x2;
typedef float split_f32x2 __attribute__((ext_vector_type(2)));

// x = p0 + p1 + p2 exactly, round-to-nearest-even at every level (v_cvt_pk_bf16_f32): |p1| <= 2^-9 |x|, |p2| <= 2^-17 |x|;
// each remainder is exactly representable, and the last one has at most 8 significant bits, so nothing is lost.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes,
                                                           long long n8, long long plane_stride) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float4 a = ((const float4*)x)[2 * i], b = ((const float4*)x)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const split_f32x2 p = {v[2 * k], v[2 * k + 1]};
        h[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, split_bf16x2));
        const split_f32x2 r = {p[0] - __uint_as_float(h[k] << 16), p[1] - __uint_as_float(h[k] & 0xffff0000u)};
        m[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, split_bf16x2));
        const split_f32x2 q = {r[0] - __uint_as_float(m[k] << 16), r[1] - __uint_as_float(m[k] & 0xffff0000u)};
        l[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(q, split_bf16x2));
    }
    *(uint4*)(planes + 8 * i) = make_uint4(h[0], h[1], h[2], h[3]);
    *(uint4*)(planes + plane_stride + 8 * i) = make_uint4(m[0], m[1], m[2], m[3]);
    *(uint4*)(planes + 2 * plane_stride + 8 * i) = make_uint4(l[0], l[1], l[2], l[3]);
}

extern "C" int hmvae_split_bf16x3(const float* x, uint16_t* planes, int64_t n, int64_t plane_stride, hmvae_stream_t stream) {
    if (!x || !planes || n <= 0 || (n & 7) || plane_stride < n || (plane_stride & 7)) return HMVAE_ERR_ARG;
    const long long n8 = n >> 3;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, planes,
                       n8, (long long)plane_stride);
    return LAUNCH_OK();
}

// ---------------- Adam ----------------
__global__ __launch_bounds__(256) void adam_kernel(const hmvae_adam_tensor* __restrict__ tab,
                                                    const float* __restrict__ lr_dev, const int* __restrict__ step_dev,
                                                    float b1, float b2, float eps, float wd) {
    const hmvae_adam_tensor t = tab[blockIdx.y];
    const int step = step_dev[0];
    // torch computes the bias corrections in double on the host; keep them in double here
    const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
    const float step_size = lr_dev[0] / bc1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < t.n; i += (long long)gridDim.x * 256) {
        const float p = t.p[i];
        float g = t.g[i];
        g = g + wd * p;                                   // torch: grad.add(param, alpha=weight_decay)
        float m = t.m[i];
        m = m + (g - m) * (1.f - b1);                     // torch: exp_avg.lerp_(grad, 1 - beta1)
        float v = t.v[i];
        v = v * b2 + (1.f - b2) * g * g;                  // torch: exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        t.m[i] = m;
        t.v[i] = v;
        t.p[i] = p - step_size * (m / denom);             // torch: param.addcdiv_(exp_avg, denom, value=-step_size)
    }
}

extern "C" int hmvae_adam_step(const hmvae_adam_tensor* table_dev, int32_t n_tensors, int64_t max_numel,
                               const float* lr_dev, const int32_t* step_dev, float beta1, float beta2, float eps,
                               float weight_decay, hmvae_stream_t stream) {
    if (!table_dev || !lr_dev || !step_dev || n_tensors <= 0 || max_numel <= 0) return HMVAE_ERR_ARG;
    long long nb = (max_numel + 255) / 256;
    if (nb > 1024) nb = 1024;
    dim3 grid((unsigned)nb, (unsigned)n_tensors);
    hipLaunchKernelGGL(adam_kernel, grid, dim3(256), 0, (hipStream_t)stream, table_dev, lr_dev, step_dev, beta1, beta2, eps,
                       weight_decay);
    return LAUNCH_OK();
}

extern "C" int hmvae_version(void) { return 1; }

extern "C" const char* hmvae_error_string(int code) {
    switch (code) {
        case HMVAE_OK: return "ok";
        case HMVAE_ERR_ARG: return "invalid argument (null pointer, bad size or alignment)";
        case HMVAE_ERR_LAUNCH: return "HIP kernel launch failed";
        case HMVAE_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}
