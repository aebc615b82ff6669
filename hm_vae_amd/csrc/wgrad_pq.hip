// Weight gradient of a convolution over FEW positions as one plain GEMM over (row, position) x (channel, position) pairs
// followed, in the same launch, by the contraction with the layer's position/tap relation.  gfx950 only.
//
// The reference computes dW[r, c, k] = sum_p sum_b dy[r, p, b] * xin[c, pos(p, k), b]  (skeleton.py:95-105, backward) where
// xin is what the layer in front of the convolution made of its source tensor `a` -- nn.Upsample(x2, linear) and
// SkeletonUnpool in the Decoder (seq_two_hier_sa_vae.py:233-242), reflection padding everywhere.  All of that is linear:
//     xin[c, pos(p, k)] = sum_q M[p, q, k] * a[c, q]            (planner.compose_conv gives M as virtual taps)
// so      dW[r, c, k] = sum_{p, q} M[p, q, k] * G[r, p, c, q],     G[r, p, c, q] = sum_b dy[r, p, b] * a[c, q, b].
// With T output and S source positions there are only T * S different G per (r, c), against T * kt (position, tap) products
// in the direct form: the first decoder layer (T = 8, S = 4, kt = 15) needs 32 instead of 120 products per weight, the
// second (16, 8) 128 instead of 240 -- and G is a dense GEMM with NO gather, no windows and K = the batch.  The direct
// kernels (wgrad_bf3.hip) spend their time on operand delivery (0.9 MFMAs per staged row, 740 MB through L2 per launch on the
// first decoder layer); here a 128 x 128 tile stages 256 rows per 96 MFMAs of every wave, once.
//
// Launch: one workgroup of 4 waves per tile of (128 / T rows) x (128 / S channels); the tile's G is 128 x 128:
//   rows i = r_local * T + p  <- dy row  dz_base[r] + p * dz_ts
//   cols j = c_local * S + q  <- a  row  x_base[c]  + q * x_ts
// main loop over the batch in stages of 32 columns, exactly the staging of wgrad_bf3.hip: 16-byte pieces global -> registers
// (next stage in flight under the MFMA block), split on the way into LDS into three exact bf16 planes (x = p0 + p1 + p2), six
// cross terms per product on v_mfma_f32_16x16x32_bf16 (everything above 2^-24 of a product, fp32 accumulation): fp32-grade.
// Epilogue (per half of the tile's rows): the wave tiles go to LDS as G'[pq][rc] (pq = p * S + q, rc = r_local * CT + c_local), and
// S[k][rc] = sum_pq M[pq][k] G'[pq][rc] runs as fp32 MFMAs (v_mfma_f32_16x16x4_f32: A = M^T from LDS, B = G' -- conflict-free
// reads both), 16 (row, channel) pairs per instruction; lane (n, kk) then holds taps 4kk..4kk+3 of pair n and writes them to
// every dense child the mask byte names (SkeletonUnpool copies / SkeletonPool means, as hmvae_wreduce_desc) -- no slab, no
// reduce launch, no up/unpool pass in front.
#include <hip/hip_runtime.h>
#include "hmvae.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte store at any 4-byte boundary

namespace {

constexpr int ROWB = 64;              // LDS row: 32 bf16
constexpr int TILE = 128;             // rows and columns of G per workgroup
constexpr unsigned PL = TILE * ROWB;  // one plane of one operand

// x = p0 + p1 + p2 exactly, round-to-nearest-even at each level; returns the three packed pairs of (x0, x1)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h0, unsigned& h1, unsigned& h2) {
    f32x2 v = {x0, x1};
    h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r = {x0 - __uint_as_float(h0 << 16), x1 - __uint_as_float(h0 & 0xffff0000u)};
    h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    f32x2 s = {r[0] - __uint_as_float(h1 << 16), r[1] - __uint_as_float(h1 & 0xffff0000u)};
    h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(s, bf16x2));
}

// the 16-byte slot of a 64-byte row, XORed so that 16 consecutive rows x 4 slots are conflict-free for ds_read_b128
__device__ __forceinline__ unsigned swz(unsigned row, unsigned slot) { return (slot ^ ((row >> 1) & 3u)) << 4; }

__global__ __launch_bounds__(256, 2) void wgrad_pq_kernel(hmvae_wgrad_pq_desc d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // the epilogue's look-ups, fetched while the first stage is in flight: mask bytes of the tile's (row, channel) pairs and
    // the dense ids of every packed id's children (read at the tile's end they were chains of dependent global loads: 12 of
    // the launch's 42 us with the scalar stores below them)
    __shared__ unsigned char s_ok[128 * 128 / 16];          // RT * CT <= 1024 pairs
    __shared__ int s_child[128 * 4];                        // child ids of the tile's channels (child_side 0) or rows (1)
    __shared__ float s_scale[128];
    char* bufA = lds;
    char* bufB = lds + 3 * PL;

    // consecutive tiles (same rows, next channels) on one XCD: they share the gradient rows in its L2
    const int per = (d.n_tiles + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || tile >= d.n_tiles) return;
    const int r0 = d.tile_r0[tile], c0 = d.tile_c0[tile];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;   // this wave's 64 x 64 quarter of G
    const int kk = lane >> 4, n = lane & 15;
    const int q8 = threadIdx.x & 7;            // 16-byte piece of a 128-byte fp32 row = 8-byte piece of a 64-byte plane row
    const int rr = threadIdx.x >> 3;           // row inside a load pass (32 rows per pass)
    const int lt = d.log_t, ls = d.log_s;
    const int T = 1 << lt, S = 1 << ls;
    const unsigned rowb = (unsigned)d.bp * 4u;

    unsigned a_src[4], b_src[4], st[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int idx = ps * 32 + rr;
        int r = r0 + (idx >> lt);
        r = r < d.n_rows ? r : d.n_rows - 1;                    // rows past the end: loaded, never written
        a_src[ps] = (unsigned)(d.dz_base[r] + (idx & (T - 1)) * d.dz_ts) * rowb + (unsigned)q8 * 16u;
        int c = c0 + (idx >> ls);
        c = c < d.n_chan ? c : d.n_chan - 1;
        b_src[ps] = (unsigned)(d.x_base[c] + (idx & (S - 1)) * d.x_ts) * rowb + (unsigned)q8 * 16u;
        st[ps] = (unsigned)idx * ROWB + swz((unsigned)idx, (unsigned)q8 >> 1) + ((unsigned)q8 & 1u) * 8u;
    }
    unsigned fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned ra = (unsigned)(wm * 64 + t * 16 + n), rb = (unsigned)(wn * 64 + t * 16 + n);
        fa[t] = ra * ROWB + swz(ra, (unsigned)kk);
        fb[t] = rb * ROWB + swz(rb, (unsigned)kk);
    }
    const char* __restrict__ dzb = (const char*)d.dz;
    const char* __restrict__ xb = (const char*)d.x;

    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[4], rb[4];
    auto fetch = [&](int s) {
        const char* __restrict__ dzs = dzb + (size_t)(s << 7);
        const char* __restrict__ xs = xb + (size_t)(s << 7);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) ra[ps] = *(const f32x4*)(dzs + a_src[ps]);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) rb[ps] = *(const f32x4*)(xs + b_src[ps]);
    };
    auto put = [&](char* base, unsigned off, const f32x4& v) {   // split + three 8-byte stores
        unsigned h0a, h1a, h2a, h0b, h1b, h2b;
        split2(v[0], v[1], h0a, h1a, h2a);
        split2(v[2], v[3], h0b, h1b, h2b);
        *(u32x2*)(base + off) = (u32x2){h0a, h0b};
        *(u32x2*)(base + PL + off) = (u32x2){h1a, h1b};
        *(u32x2*)(base + 2 * PL + off) = (u32x2){h2a, h2b};
    };
    const int n_stage = d.bp >> 5;
    fetch(0);
    {
        const int lct0 = 7 - ls, CT0 = 1 << lct0, RT0 = TILE >> lt;
        for (int i = threadIdx.x; i < RT0 * CT0; i += 256) {
            const int r = r0 + (i >> lct0), c = c0 + (i & (CT0 - 1));
            s_ok[i] = (r < d.n_rows && c < d.n_chan) ? d.ok[(size_t)r * d.n_chan + c] : (unsigned char)0;
        }
        if (d.child_tab != nullptr) {
            const int n_id = d.child_side ? RT0 : CT0, id0 = d.child_side ? r0 : c0, id_end = d.child_side ? d.n_rows : d.n_chan;
            for (int i = threadIdx.x; i < n_id * 4; i += 256) s_child[i] = id0 + (i >> 2) < id_end ? d.child_tab[(id0 + (i >> 2)) * 4 + (i & 3)] : -1;
            for (int i = threadIdx.x; i < n_id; i += 256)
                s_scale[i] = (d.child_scale && id0 + i < id_end) ? d.child_scale[id0 + i] : 1.f;
        }
    }
    for (int s = 0; s < n_stage; ++s) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) put(bufA, st[ps], ra[ps]);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) put(bufB, st[ps], rb[ps]);
        __syncthreads();
        if (s + 1 < n_stage) fetch(s + 1);                        // in flight during the MFMA block below
        __builtin_amdgcn_sched_barrier(0);
        {
            bf16x8 a[4][3], b[4][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t][pl] = *(const bf16x8*)(bufA + pl * PL + fa[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t][pl] = *(const bf16x8*)(bufB + pl * PL + fb[t]);
            }
            // smallest cross terms first: (0,2) (2,0) (1,1) | (0,1) (1,0) | (0,0)
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][PA[q]], b[nt][PB[q]], acc[mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                          // every wave has read its fragments: the image is free
    }

    // ---- epilogue: G'[pq][rc] and M^T[pq][16] in LDS, contraction on the fp32 matrix pipe, dense stores.  Two passes, one
    // per half of the tile's rows (the waves with wm == pass hold them), so that the image fits into the staging buffers'
    // 48 KB: with a 66-74 KB image two workgroups filled a CU's LDS and the data-gradient chain's launches, which run beside
    // the weight gradients, found no room (measured: the step got SLOWER with 220 us less work in it).
    const int lct = 7 - ls;                                       // CT = 128 / S channels per tile
    const int CT = 1 << lct;
    const int RCH = (64 >> lt) << lct;                            // (row, channel) pairs of one pass
    const int PQ = T * S;
    // pitch of G': + 16 puts the four pq rows of a contraction step on disjoint banks; where that does not fit, + 1
    const int GST = RCH + ((size_t)PQ * (RCH + 16 + 16) * sizeof(float) <= 6 * PL ? 16 : 1);
    float* G = (float*)lds;
    float* Mt = G + PQ * GST;
    for (int i = threadIdx.x; i < PQ * 16; i += 256) Mt[i] = d.m[i];
    const int n_blk = RCH >> 4;
    for (int pass = 0; pass < 2; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = mt * 16 + kk * 4 + r;             // row inside this half
                    const int rowoff = (i & (T - 1)) * S * GST + ((i >> lt) << lct);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int j = wn * 64 + nt * 16 + n;
                        G[rowoff + (j & (S - 1)) * GST + (j >> ls)] = acc[mt][nt][r];
                    }
                }
        }
        __syncthreads();
        for (int blk = wave; blk < n_blk; blk += 4) {
            const int rc = blk * 16 + n;
            const int r_l = pass * (64 >> lt) + (rc >> lct), c_l = rc & (CT - 1);
            const int r = r0 + r_l, c = c0 + c_l;
            const unsigned bits = s_ok[(r_l << lct) + c_l];
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
            const float* gp = G + kk * GST + rc;
            const float* mp = Mt + kk * 16 + n;
#pragma unroll 4
            for (int pq = 0; pq < PQ; pq += 4)
                s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(mp[pq * 16], gp[pq * GST], s4, 0, 0, 0);
            if (!bits) continue;
            // s4[i] = dW of tap 4 kk + i for the packed (row, channel) pair rc: the four lanes (kk) of a pair write its kt
            // consecutive floats, 16 bytes each -- neighbouring pairs' runs are adjacent in the dense tensor
            const int nk = d.kt - 4 * kk;                         // taps this lane holds: >= 4, 3, 2, 1 or <= 0
            if (nk <= 0) continue;
            const bool has_tab = d.child_tab != nullptr;
            const int cid = d.child_side ? r_l : c_l;
            const float sc = has_tab ? s_scale[cid] : 1.f;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                if (!((bits >> ci) & 1u)) continue;
                int dr = r, dc = c;
                if (has_tab) {
                    const int ch = s_child[cid * 4 + ci];
                    if (d.child_side) dr = ch; else dc = ch;
                } else if (ci > 0) {
                    break;
                }
                float* o = d.dense + (size_t)dr * d.row_stride + (size_t)dc * d.chan_stride + 4 * kk;
                f32x4 v = s4 * sc;
                if (d.accumulate) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < nk) v[i] += o[i];
                }
                if (nk >= 4) {
                    *(f32x4u*)o = v;
                } else {
                    o[0] = v[0];
                    if (nk > 1) o[1] = v[1];
                    if (nk > 2) o[2] = v[2];
                }
            }
        }
        __syncthreads();                                          // the image is free for the other half
    }
}

}  // namespace

// LDS bytes of a launch (the host mirrors this in ops.MaskedGemm.pq_lds), or 0 if the problem does not fit
static size_t pq_lds_bytes(int log_t, int log_s) {
    if (log_t < 0 || log_s < 0 || log_t > 6 || log_s > 7) return 0;        // a half tile holds whole rows: t_dst <= 64
    const size_t PQ = (size_t)1 << (log_t + log_s);
    const size_t RCH = ((size_t)64 >> log_t) * ((size_t)TILE >> log_s);
    if (RCH < 16 || PQ < 16) return 0;                       // (the epilogue's mask table holds 16384 / PQ <= 1024 pairs)
    const size_t gst = RCH + (PQ * (RCH + 16 + 16) * sizeof(float) <= 6 * (size_t)PL ? 16 : 1);
    size_t lds = (PQ * gst + PQ * 16) * sizeof(float);
    if (lds < 6 * (size_t)PL) lds = 6 * PL;
    return lds <= 160 * 1024 - 4096 ? lds : 0;               // (the kernel's static tables sit beside the dynamic block)
}

extern "C" int hmvae_wgrad_pq(const hmvae_wgrad_pq_desc* d, hmvae_stream_t stream) {
    if (!d) return HMVAE_ERR_ARG;
    if (!d->dz || !d->x || !d->dense || !d->dz_base || !d->x_base || !d->tile_r0 || !d->tile_c0 || !d->ok || !d->m)
        return HMVAE_ERR_ARG;
    if (d->n_tiles <= 0 || d->n_rows <= 0 || d->n_chan <= 0 || d->kt < 1 || d->kt > 16) return HMVAE_ERR_ARG;
    if (d->bp <= 0 || (d->bp & 31)) return HMVAE_ERR_ARG;
    if (d->child_scale && !d->child_tab) return HMVAE_ERR_ARG;
    const size_t lds = pq_lds_bytes(d->log_t, d->log_s);
    if (!lds) return HMVAE_ERR_UNSUPPORTED;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)wgrad_pq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096) != hipSuccess)
            return HMVAE_ERR_LAUNCH;
        attr = true;
    }
    const unsigned grid = (unsigned)(((d->n_tiles + 7) / 8) * 8);
    hipLaunchKernelGGL(wgrad_pq_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, *d);
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}
