// gemm_kernels.hip — fp32-MFMA gather-GEMM, weight-gradient GEMM, weight packing (gfx950).
//
// Replaces, for the hm-vae hot path (reference paths):
//   skeleton.py:95-105   SkeletonConv.forward  (weight*mask, F.pad reflect, F.conv1d)  and its autograd
//   seq_two_hier_sa_vae.py:162,267  nn.Linear latent heads and their autograd
//
// Data layout: activations are [row][Bp] with batch innermost (Bp % 64 == 0), row = channel*T + t.
// MFMA: v_mfma_f32_16x16x4_f32 — A: lane l holds A[m=l&15][k=l>>4]; B: lane l holds B[k=l>>4][n=l&15];
// D: lane l, reg r holds D[row=(l>>4)*4+r][col=l&15]  (cdna_hip_programming.md §3).
//
// gather-GEMM work split: one workgroup = one (row-group, output position, 64-column batch tile);
// its 4 waves split the reduction over 4-channel steps and are summed through LDS.  The 64 columns of a
// tile are permuted (MFMA tile j, column n  <->  batch column 4n+j) so that one 16-byte load per lane
// feeds all four 16-column MFMA tiles and the epilogue stores 16 bytes per lane.
#include <hip/hip_runtime.h>
#include "pack_taps_inl.h"
#include <type_traits>
#include "hmvae.h"
#include "wreduce_inl.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int MT>
__global__ __launch_bounds__(256) void gather_gemm_kernel(hmvae_gemm_desc d) {
    __shared__ __attribute__((aligned(16))) float red[4 * MT * 16 * 64];
    const int n_ct = d.bp >> 6;
    int bid = blockIdx.x;
    const int g = bid % d.n_groups;
    bid /= d.n_groups;
    const int ct = bid % n_ct;
    const int p = bid / n_ct;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: keeps the K loop uniform
    const int kk = lane >> 4;
    const int n = lane & 15;

    const int nc4 = d.g_nc4[g];
    const int choff = d.g_choff[g];
    const int poff = d.p_off[p];
    const int plen = d.p_len[p];
    const float* __restrict__ wg = d.wp + d.g_woff[g] + lane;
    const float* __restrict__ srcc = d.src + (ct << 6) + (n << 2);
    const int* __restrict__ pairs = d.pairs + poff;
    const int bp = d.bp;
    const int src_ts = d.src_ts;
    const int kt = d.kt;

    // two-level accumulation: `acc` holds the running sum of one 4-channel step (<= kt MFMAs, i.e. a short fp32
    // fma chain); it is folded into `tot` at every step boundary.  Keeps the rounding error of the K ~ 10^4
    // contractions at the level of a blocked CPU reduction instead of one long sequential chain.
    f32x4 acc[MT][4], tot[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            tot[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    if (plen > 0 && wave < nc4) {
        int c4 = wave;
        int i = 0;
        int rowbase = d.chan_rows[choff + c4 * 4 + kk];
        int rowbase_nx = (c4 + 4 < nc4) ? d.chan_rows[choff + (c4 + 4) * 4 + kk] : 0;
        float a_cur[MT];
        f32x4 b_cur;
        {
            const int pr = pairs[0];
            const int q = pr & 0xffff, tap = pr >> 16;
            const float* wc = wg + (size_t)(c4 * kt + tap) * (MT * 64);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a_cur[mt] = wc[mt * 64];
            b_cur = *(const f32x4*)(srcc + (size_t)(rowbase + q * src_ts) * bp);
        }
        bool more = true;
        while (more) {
            int ni = i + 1, nc = c4;
            int rb = rowbase;
            if (ni == plen) {
                ni = 0;
                nc += 4;
                rb = rowbase_nx;
            }
            const bool nvalid = nc < nc4;
            float a_nx[MT];
            f32x4 b_nx;
            if (nvalid) {
                const int pr = pairs[ni];
                const int q = pr & 0xffff, tap = pr >> 16;
                const float* wc = wg + (size_t)(nc * kt + tap) * (MT * 64);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_nx[mt] = wc[mt * 64];
                b_nx = *(const f32x4*)(srcc + (size_t)(rb + q * src_ts) * bp);
                if (nc != c4) {
                    rowbase = rb;
                    rowbase_nx = (nc + 4 < nc4) ? d.chan_rows[choff + (nc + 4) * 4 + kk] : 0;
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mt][j] = MFMA16(a_cur[mt], b_cur[j], acc[mt][j]);
            }
            if (nvalid) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nx[mt];
                b_cur = b_nx;
            }
            if (nc != c4) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        tot[mt][j] += acc[mt][j];
                        acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
            }
            c4 = nc;
            i = ni;
            more = nvalid;
        }
    }

    // cross-wave reduction through LDS; acc[mt][j][r] is row mt*16+kk*4+r, batch column 4n+j
    float* mine = red + wave * (MT * 16 * 64);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 v = {tot[mt][0][r], tot[mt][1][r], tot[mt][2][r], tot[mt][3][r]};
            *(f32x4*)(mine + (mt * 16 + kk * 4 + r) * 64 + (n << 2)) = v;
        }
    }
    __syncthreads();
    const int rl = threadIdx.x >> 4;   // 0..15
    const int cq = threadIdx.x & 15;   // column quad
#pragma unroll
    for (int ps = 0; ps < MT; ++ps) {
        const int row = ps * 16 + rl;
        const int drow = d.dst_rows[g * (MT * 16) + row];
        if (drow < 0) continue;
        f32x4 s = *(const f32x4*)(red + row * 64 + (cq << 2));
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            f32x4 t = *(const f32x4*)(red + w * (MT * 16 * 64) + row * 64 + (cq << 2));
            s += t;
        }
        if (d.bias) {
            const int bi = d.bias_idx[g * (MT * 16) + row];
            const float bv = bi >= 0 ? d.bias[bi] : 0.f;
            s += (f32x4){bv, bv, bv, bv};
        }
        if (d.flags & HMVAE_FLAG_LEAKY) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = s[e] > 0.f ? s[e] : s[e] * d.slope;
        }
        float* dp = d.dst + (size_t)(drow + p * d.dst_ts) * bp + (ct << 6) + (cq << 2);
        if (d.flags & HMVAE_FLAG_ACCUM) {
            f32x4 o = *(const f32x4*)dp;
            s += o;
        }
        *(f32x4*)dp = s;
    }
}

extern "C" int hmvae_gather_gemm(const hmvae_gemm_desc* desc, hmvae_stream_t stream) {
    if (!desc || !desc->src || !desc->dst || !desc->wp || (desc->bias && !desc->bias_idx)) return HMVAE_ERR_ARG;
    if (desc->bp <= 0 || (desc->bp & 63) || desc->n_groups <= 0 || desc->n_pos <= 0 || desc->kt <= 0) return HMVAE_ERR_ARG;
    const long long nblk = (long long)desc->n_groups * desc->n_pos * (desc->bp >> 6);
    if (nblk > 0x7fffffffLL) return HMVAE_ERR_ARG;
    dim3 grid((unsigned)nblk), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (desc->mt) {
        case 1: hipLaunchKernelGGL(gather_gemm_kernel<1>, grid, block, 0, s, *desc); break;
        case 2: hipLaunchKernelGGL(gather_gemm_kernel<2>, grid, block, 0, s, *desc); break;
        case 3: hipLaunchKernelGGL(gather_gemm_kernel<3>, grid, block, 0, s, *desc); break;
        case 4: hipLaunchKernelGGL(gather_gemm_kernel<4>, grid, block, 0, s, *desc); break;
        default: return HMVAE_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// weight gradient: rows = group rows (dz), columns = (listed channel, tap), reduction over (t, batch).
// K permutation: MFMA k-step j of a 16-batch block uses batch elements {4*kk + j}, so each lane feeds
// four k-steps from one 16-byte load on both operands.
// ------------------------------------------------------------------------------------------------
// BF: the operands are read as THREE bf16 planes per tensor (hmvae_split_bf16x3: x = p0 + p1 + p2 exactly) and a chunk's
// 32 batch columns go through v_mfma_f32_16x16x32_bf16 -- six products per tile (the cross terms down to 2^-16; what is
// dropped is below 2^-23 of the product, one fp32 rounding) in 6 x 16 issue cycles where the fp32 instruction needs
// 8 x 32 for the same K.  Same lane maps on both sides (lane (n, kk) holds 8 consecutive batch elements of row / column
// n; the C/D layout does not depend on the input type), so everything outside load / fma is shared.
template <int MT, bool ZP, bool BF>
__global__ __launch_bounds__(256) void wgrad_kernel(hmvae_wgrad_desc d) {
    // LDS: first the per-position activation offset table (during the reduction loop), then the cross-wave sums
    extern __shared__ __attribute__((aligned(16))) float red[];
    unsigned* xtab = (unsigned*)red;                  // [t_dst][64]: byte offset of column c's activation row at position t
    // Work items = (tile, range of reduction chunks, slab part).  Classic launch: one item per workgroup, item =
    // (tile, split), XCD x (blockIdx % 8) taking a contiguous tile-major range -- the tiles of one input block (same
    // activation rows, different output-row pieces) are neighbours, so an XCD's L2 sees ~1/8 of the activations.
    // Persistent launch (n_workers > 0): exactly (CUs x resident workgroups) workers, each walking a host-made list of
    // items with the same number of chunks (planner.build_stream_schedule): no tail, fewer tile prologues.
    int it0, it1;
    if (d.n_workers > 0) {
        const int perw = d.n_workers >> 3;
        const int worker = (blockIdx.x & 7) * perw + (blockIdx.x >> 3);
        it0 = d.worker_items[2 * worker];
        it1 = it0 + d.worker_items[2 * worker + 1];
    } else {
        const int n_items = d.n_tiles * d.n_split;
        const int per = (n_items + 7) >> 3;
        it0 = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= per || it0 >= n_items) return;
        it1 = it0 + 1;
    }
  for (int item = it0; item < it1; ++item) {
    int tile, lo, hi;
    size_t part;
    {
        const bool TRq = d.transposed != 0;
        const int Tq = TRq ? d.t_src + 2 * d.pad : d.t_dst;
        if (d.n_workers > 0) {
            tile = d.items[4 * item];
            lo = d.items[4 * item + 1];
            hi = d.items[4 * item + 2];
            part = (size_t)d.items[4 * item + 3];
        } else {
            tile = item / d.n_split;
            const int sp = item - tile * d.n_split;
            const long long total = (long long)Tq * (d.bp >> 5);
            lo = (int)(total * sp / d.n_split);
            hi = (int)(total * (sp + 1) / d.n_split);
            part = (size_t)sp * d.n_tiles + tile;
        }
    }
    const int g = d.tile_g[tile];
    const int col0 = d.tile_col0[tile];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: keeps the K loop uniform
    const int kk = lane >> 4;
    const int n = lane & 15;
    const int bp = d.bp;
    const int ncols = d.g_ncols[g];
    const int choff = d.g_choff[g];
    const bool TR = d.transposed != 0;
    const int T = TR ? d.t_src + 2 * d.pad : d.t_dst;        // positions the reduction walks over

    // table: column c = threadIdx & 63 of this tile, positions t = wave, wave+4, ...   (all address arithmetic that
    // depends on (t, tap) happens here, once; the loop only adds a scalar base — fp32 MFMA shares the SIMD's fp32
    // datapath with ordinary vector instructions, so loop VALU work is time taken from the matrix pipe)
    {
        const int c = threadIdx.x & 63;
        const int col = col0 + c;
        int xr, tp;
        if (col < ncols) {
            const int ch = col / d.kt;
            tp = col - ch * d.kt;
            xr = d.chan_rows[choff + ch];
        } else {
            tp = 0;
            xr = d.chan_rows[choff];
        }
        for (int t = threadIdx.x >> 6; t < T; t += 4) {
            int u = t * d.stride + tp - d.pad;
            unsigned v;
            if (TR) {
                // column = (output channel, tap) of dz: padded input position t feeds output position t - tap
                const int td = t - tp;
                v = (td >= 0 && td < d.t_dst) ? (unsigned)(xr + td * d.dz_ts) * (unsigned)bp * 4u : 0xffffffffu;
            } else if (d.pad_mode == HMVAE_PAD_REFLECT) {
                if (u < 0) u = -u;
                if (u >= d.t_src) u = 2 * (d.t_src - 1) - u;
                v = (unsigned)(xr + u * d.x_ts) * (unsigned)bp * 4u;
            } else {
                v = (u >= 0 && u < d.t_src) ? (unsigned)(xr + u * d.x_ts) * (unsigned)bp * 4u : 0xffffffffu;
            }
            xtab[t * 64 + c] = v;
        }
    }
    unsigned rowbyte[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = d.dz_rows[g * (MT * 16) + mt * 16 + n];
        rowbyte[mt] = (unsigned)(r < 0 ? 0 : r) * (unsigned)bp * 4u + (unsigned)kk * 32u;
    }
    const unsigned kk32 = (unsigned)kk * 32u;
    __syncthreads();

    f32x4 acc[MT][4], tot[MT][4];          // two-level accumulation, see gather_gemm_kernel
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            tot[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    // Reduction chunks: one position t x 32 batch columns.  Lane (n, kk) loads the 32 contiguous bytes
    // [b0 + 8kk, b0 + 8kk + 8) of its row, so the 4 kk-lanes of a row cover one full 128-byte line, and MFMA k-step j
    // (0..7) uses batch element 8kk + j on both operands.  Chunks are numbered with t fastest, so the four waves of
    // a workgroup work on neighbouring positions of the same batch block at the same time.
    const char* __restrict__ dzb = (const char*)(TR ? d.x : d.dz);       // row operand
    const char* __restrict__ xb = (const char*)(TR ? d.dz : d.x);        // column operand
    const unsigned dz_tstep = (unsigned)(TR ? d.x_ts : d.dz_ts) * (unsigned)bp * 4u;
    const int pad = d.pad, tsrc = d.t_src;
    struct OpsF {
        f32x4 a[MT][2];
        f32x4 b[4][2];
    };
    struct OpsB {
        bf16x8 a[MT][3];
        bf16x8 b[4][3];
    };
    using Ops = std::conditional_t<BF, OpsB, OpsF>;
    // plane addressing: element offsets are the fp32 ones, so byte offsets halve
    const char* __restrict__ dzp = (const char*)(TR ? d.x_planes : d.dz_planes);
    const char* __restrict__ xp = (const char*)(TR ? d.dz_planes : d.x_planes);
    const size_t dz_ps = (size_t)(TR ? d.x_plane_stride : d.dz_plane_stride) * 2, x_ps = (size_t)(TR ? d.dz_plane_stride : d.x_plane_stride) * 2;
    auto load = [&](Ops& o, int chunk) {
        const int bb = chunk / T;            // scalar
        const int t = chunk - bb * T;
        int tr = t;
        if (TR) {                            // row operand sits at the reflected source position of padded position t
            tr = t - pad;
            tr = tr < 0 ? -tr : tr;
            tr = tr >= tsrc ? 2 * (tsrc - 1) - tr : tr;
        }
        if constexpr (BF) {
            const char* dzs = dzp + (((size_t)tr * dz_tstep) >> 1) + (size_t)(bb << 6);
            const char* xs = xp + (size_t)(bb << 6);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) o.a[mt][pl] = *(const bf16x8*)(dzs + pl * dz_ps + (rowbyte[mt] >> 1));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const unsigned e = xtab[t * 64 + nt * 16 + n];
                const bool ok = !ZP || e != 0xffffffffu;
                const unsigned off = ((ok ? e : 0u) + kk32) >> 1;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const bf16x8 v = *(const bf16x8*)(xs + pl * x_ps + off);
                    if (ZP) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        o.b[nt][pl] = ok ? v : __builtin_bit_cast(bf16x8, z);
                    } else {
                        o.b[nt][pl] = v;
                    }
                }
            }
        } else {
        const char* dzs = dzb + (size_t)tr * dz_tstep + (size_t)(bb << 7);     // uniform base for this chunk
        const char* xs = xb + (size_t)(bb << 7);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            o.a[mt][0] = *(const f32x4*)(dzs + rowbyte[mt]);
            o.a[mt][1] = *(const f32x4*)(dzs + rowbyte[mt] + 16);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const unsigned e = xtab[t * 64 + nt * 16 + n];
            if (ZP) {
                const bool ok = e != 0xffffffffu;
                const unsigned off = (ok ? e : 0u) + kk32;
                const f32x4 v0 = *(const f32x4*)(xs + off), v1 = *(const f32x4*)(xs + off + 16);
                o.b[nt][0] = ok ? v0 : (f32x4){0.f, 0.f, 0.f, 0.f};
                o.b[nt][1] = ok ? v1 : (f32x4){0.f, 0.f, 0.f, 0.f};
            } else {
                const unsigned off = e + kk32;
                o.b[nt][0] = *(const f32x4*)(xs + off);
                o.b[nt][1] = *(const f32x4*)(xs + off + 16);
            }
        }
        }
    };
    auto fma = [&](const Ops& o) {
        if constexpr (BF) {
            // smallest cross terms first: (0,2) (2,0) (1,1) | (0,1) (1,0) | (0,0)
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o.a[mt][PA[q]], o.b[nt][PB[q]], acc[mt][nt], 0, 0, 0);
        } else {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MFMA16(o.a[mt][h][j], o.b[nt][h][j], acc[mt][nt]);
        }
    };
    auto flush = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                tot[mt][nt] += acc[mt][nt];
                acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
    };
    // this wave's chunks: lo + wave, lo + wave + 4, ...   (software pipelined two deep over two static register sets)
    const int first = lo + wave;
    const int n_mine = first < hi ? (hi - first + 3) >> 2 : 0;
    if (n_mine > 0) {
        Ops o0, o1;
        load(o0, first);
        const int last = first + 4 * (n_mine - 1);
        const int pairs = n_mine >> 1;
        constexpr int FLUSH_PAIRS = 4;           // 8 chunks = 64 k-steps = a chain of 256 fp32 fmas
        int c = first;
        for (int pb = 0; pb < pairs; pb += FLUSH_PAIRS) {
            const int pe = pb + FLUSH_PAIRS < pairs ? pb + FLUSH_PAIRS : pairs;
            for (int pi = pb; pi < pe; ++pi) {
                // The scheduling barriers keep each prefetch AHEAD of the MFMA block it hides behind.  Left alone the
                // compiler sinks the loads into the second half of that block: the next block then waits for loads
                // issued ~50 MFMAs (1 500 cycles) earlier, less than a loaded memory round trip, and with one wave per
                // SIMD (MT = 3: 268 registers) nothing else fills the stall.
                load(o1, c + 4);
                __builtin_amdgcn_sched_barrier(0);
                fma(o0);
                __builtin_amdgcn_sched_barrier(0);
                const int nxt = c + 8 <= last ? c + 8 : last;      // clamped over-fetch at the tail
                load(o0, nxt);
                __builtin_amdgcn_sched_barrier(0);
                fma(o1);
                __builtin_amdgcn_sched_barrier(0);
                c += 8;
            }
            flush();
        }
        if (n_mine & 1) fma(o0);
        flush();
    }
    __syncthreads();                        // everyone is done with xtab: the LDS becomes the reduction buffer
    // tot[mt][nt][r] is row mt*16 + kk*4 + r, column nt*16 + n
    float* mine = red + wave * (MT * 16 * 64);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(mt * 16 + kk * 4 + r) * 64 + nt * 16 + n] = tot[mt][nt][r];
    __syncthreads();
    const int rl = threadIdx.x >> 4;
    const int cq = threadIdx.x & 15;
    float* out = d.slab + part * (MT * 16 * 64);
#pragma unroll
    for (int ps = 0; ps < MT; ++ps) {
        const int row = ps * 16 + rl;
        f32x4 s = *(const f32x4*)(red + row * 64 + (cq << 2));
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *(const f32x4*)(red + w * (MT * 16 * 64) + row * 64 + (cq << 2));
        *(f32x4*)(out + row * 64 + (cq << 2)) = s;
    }
    __syncthreads();                        // the next item rebuilds the position table in the same LDS
  }
}

template <int MT>
static void launch_wgrad(const hmvae_wgrad_desc* d, dim3 grid, hipStream_t s) {
    size_t lds = (size_t)4 * MT * 16 * 64 * sizeof(float);
    const size_t tab = (size_t)d->t_dst * 64 * sizeof(unsigned);
    if (tab > lds) lds = tab;
    const int npos = d->transposed ? d->t_src + 2 * d->pad : d->t_dst;
    const size_t tab2 = (size_t)npos * 64 * sizeof(unsigned);
    if (tab2 > lds) lds = tab2;
    const bool zp = d->pad_mode == HMVAE_PAD_ZEROS || d->transposed;
    if (d->dz_planes) {
        if (zp) hipLaunchKernelGGL((wgrad_kernel<MT, true, true>), grid, dim3(256), lds, s, *d);
        else hipLaunchKernelGGL((wgrad_kernel<MT, false, true>), grid, dim3(256), lds, s, *d);
    } else {
        if (zp) hipLaunchKernelGGL((wgrad_kernel<MT, true, false>), grid, dim3(256), lds, s, *d);
        else hipLaunchKernelGGL((wgrad_kernel<MT, false, false>), grid, dim3(256), lds, s, *d);
    }
}

int hmvae_wgrad_lds_launch(const hmvae_wgrad_desc* d, unsigned grid, hipStream_t s);      // wgrad_lds.hip
int hmvae_wgrad_bf3_launch(const hmvae_wgrad_desc* d, unsigned grid, hipStream_t s);      // wgrad_bf3.hip

extern "C" int hmvae_wgrad_gemm(const hmvae_wgrad_desc* desc, hmvae_stream_t stream) {
    if (!desc || !desc->dz || !desc->x || !desc->slab) return HMVAE_ERR_ARG;
    if ((desc->dz_planes == nullptr) != (desc->x_planes == nullptr)) return HMVAE_ERR_ARG;
    if (desc->bp <= 0 || (desc->bp & 63) || desc->n_tiles <= 0 || desc->n_split <= 0 || desc->kt <= 0) return HMVAE_ERR_ARG;
    if (desc->pad_mode == HMVAE_PAD_REFLECT && desc->pad >= desc->t_src && desc->pad > 0) return HMVAE_ERR_ARG;
    const long long nblk = desc->n_workers > 0 ? desc->n_workers : (long long)desc->n_tiles * desc->n_split;
    if (nblk > 0x7fffffffLL) return HMVAE_ERR_ARG;
    if (desc->n_workers < 0 || (desc->n_workers & 7) || (desc->n_workers > 0 && (!desc->items || !desc->worker_items)))
        return HMVAE_ERR_ARG;
    if (desc->transposed && (desc->stride != 1 || desc->pad_mode != HMVAE_PAD_REFLECT)) return HMVAE_ERR_UNSUPPORTED;
    {
        const int npos = desc->transposed ? desc->t_src + 2 * desc->pad : desc->t_dst;
        if ((size_t)npos * 64 * sizeof(unsigned) > 64 * 1024) return HMVAE_ERR_UNSUPPORTED;       // position table in LDS
    }
    dim3 grid((unsigned)(((nblk + 7) / 8) * 8));
    hipStream_t s = (hipStream_t)stream;
    if (desc->mode == 1) {
        if (desc->t_dst < 4) return HMVAE_ERR_UNSUPPORTED;
        return hmvae_wgrad_lds_launch(desc, grid.x, s);
    }
    if (desc->fused_reduce && desc->mode != 2) return HMVAE_ERR_UNSUPPORTED;
    if (desc->mode == 2) {
        if (desc->t_dst < 4) return HMVAE_ERR_UNSUPPORTED;
        return hmvae_wgrad_bf3_launch(desc, grid.x, s);
    }
    if (desc->mode != 0) return HMVAE_ERR_ARG;
    switch (desc->mt) {
        case 1: launch_wgrad<1>(desc, grid, s); break;
        case 2: launch_wgrad<2>(desc, grid, s); break;
        case 3: launch_wgrad<3>(desc, grid, s); break;
        case 4: launch_wgrad<4>(desc, grid, s); break;
        default: return HMVAE_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}

// One workgroup per (tile, 16-row piece), strided when the grid is smaller: the tile's group and first column are
// uniform, a thread keeps one of the 64 columns -- its (channel, tap) split is done once -- and 4 of the 16 rows, so per
// element there is one mask byte, n_split coalesced slab reads and one store (no per-element divisions or table walks).
__device__ __forceinline__ void wreduce_body(const hmvae_wreduce_desc& d, int blk, int nblk) {
    const int rows = d.mt * 16;
    const int per_tile = rows * 64;
    const int col = threadIdx.x & 63;
    const int r0 = threadIdx.x >> 6;
    const size_t split_stride = (size_t)d.n_tiles * per_tile;
    for (int work = blk; work < d.n_tiles * d.mt; work += nblk) {
        const int tile = work / d.mt, piece = work - tile * d.mt;
        const float* sl = d.slab + (size_t)tile * per_tile + col;
        // the slab sums first: their addresses need none of the tables, so these loads are in flight while the row / mask
        // lookups resolve (behind the `continue` they were one more level of a dependent chain of memory round trips)
        float sums[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = piece * 16 + r0 + 4 * i;
            float s0 = 0.f, s1 = 0.f;
            if (d.tile_part0) {                        // persistent weight-gradient launch: the tile's parts are listed
                const float* pp = d.slab + (size_t)d.tile_part0[tile] * per_tile + col + row * 64;
                const int np = d.tile_nparts[tile];
                for (int sp = 0; sp < np; ++sp) s0 += pp[(size_t)sp * per_tile];
            } else {
                int sp = 0;
                for (; sp + 1 < d.n_split; sp += 2) {
                    s0 += sl[sp * split_stride + row * 64];
                    s1 += sl[(sp + 1) * split_stride + row * 64];
                }
                if (sp < d.n_split) s0 += sl[sp * split_stride + row * 64];
            }
            sums[i] = s0 + s1;
        }
        const WreduceColR wc = wreduce_col_resolve(d, tile, col);
        if (!wc.live) continue;
        const int g = d.tile_g[tile];
        int rid[4];
        unsigned bits[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = piece * 16 + r0 + 4 * i;
            rid[i] = d.row_id[g * rows + row];
            bits[i] = wc.okp[row * wc.ok_stride];
        }
        wreduce_store_rows<4>(d, wc, rid, bits, sums);
    }
}

__global__ __launch_bounds__(256) void wreduce_kernel(hmvae_wreduce_desc d) { wreduce_body(d, blockIdx.x, gridDim.x); }

__global__ __launch_bounds__(256) void wreduce_multi_kernel(const hmvae_wreduce_desc* __restrict__ descs,
                                                            const int* __restrict__ block_start, int n_ops) {
    int oi = 0;
    for (int i = 1; i < n_ops; ++i)
        if ((int)blockIdx.x >= block_start[i]) oi = i;
    const hmvae_wreduce_desc d = descs[oi];
    wreduce_body(d, blockIdx.x - block_start[oi], block_start[oi + 1] - block_start[oi]);
}

extern "C" int hmvae_wgrad_reduce_multi(const hmvae_wreduce_desc* descs_dev, const int32_t* block_start_dev, int32_t n_ops,
                                        int32_t total_blocks, hmvae_stream_t stream) {
    if (!descs_dev || !block_start_dev || n_ops <= 0 || total_blocks <= 0) return HMVAE_ERR_ARG;
    hipLaunchKernelGGL(wreduce_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev,
                       block_start_dev, n_ops);
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}

extern "C" int hmvae_wgrad_reduce(const hmvae_wreduce_desc* desc, hmvae_stream_t stream) {
    if (!desc || !desc->slab || !desc->dense || !desc->pair_ok || desc->n_tiles <= 0) return HMVAE_ERR_ARG;
    const int nw = desc->n_tiles * desc->mt;
    const int nb = nw < 16384 ? nw : 16384;
    hipLaunchKernelGGL(wreduce_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, *desc);
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}

// One workgroup per (group, 4-channel step).  Reads walk the dense tensor in its own memory order (taps fastest, then
// whichever of row/channel has the smaller stride), land in an LDS image of the packed chunk, and leave as whole
// 256-byte MFMA fragments: both sides of the transposition are coalesced.
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ dense, float* __restrict__ packed,
                                                   const int* __restrict__ g_nc4, const int* __restrict__ g_woff,
                                                   const int* __restrict__ g_choff, const int* __restrict__ chan_id,
                                                   const int* __restrict__ row_id, const int* __restrict__ g_okoff,
                                                   const unsigned char* __restrict__ pair_ok, int mt, int kt, int row_stride,
                                                   int chan_stride, int tap_stride, int tap_flip) {
    extern __shared__ float img[];                    // [kt][mt][64]
    const int g = blockIdx.y;
    const int c4 = blockIdx.x;
    const int nc4 = g_nc4[g];
    if (c4 >= nc4) return;
    const int rows = mt * 16;
    const bool chan_inner = chan_stride < row_stride;  // forward layout: channels of a row are adjacent
    // one (row, channel) pair per thread (rows*4 <= 256), its kt taps read back to back: no index arithmetic per element
    const int ab = threadIdx.x;
    if (ab < rows * 4) {
        int r, kk;
        if (chan_inner) {
            kk = ab & 3;
            r = ab >> 2;
        } else {
            r = ab % rows;
            kk = ab / rows;
        }
        const int rid = row_id[g * rows + r];
        const int cid = chan_id[g_choff[g] + c4 * 4 + kk];
        const bool ok = rid >= 0 && cid >= 0 && pair_ok[g_okoff[g] + r * (nc4 * 4) + c4 * 4 + kk];
        const float* src = dense + (ok ? (size_t)rid * row_stride + (size_t)cid * chan_stride : 0);
        float* dsti = img + (r >> 4) * 64 + kk * 16 + (r & 15);
        for (int tap = 0; tap < kt; ++tap) {
            const int td = tap_flip ? (kt - 1 - tap) : tap;
            dsti[tap * mt * 64] = ok ? src[(size_t)td * tap_stride] : 0.f;
        }
    }
    __syncthreads();
    float* out = packed + g_woff[g] + (size_t)c4 * kt * mt * 64;
    const int n = kt * mt * 64;
    for (int i = threadIdx.x; i < n; i += 256) out[i] = img[i];
}

extern "C" int hmvae_pack_weights(const float* dense, float* packed, const int32_t* g_nc4, const int32_t* g_woff,
                                  const int32_t* g_choff, const int32_t* chan_id, const int32_t* row_id,
                                  const int32_t* g_okoff, const uint8_t* pair_ok, int32_t n_groups, int32_t max_nc4,
                                  int32_t mt, int32_t kt, int32_t row_stride, int32_t chan_stride, int32_t tap_stride,
                                  int32_t tap_flip, hmvae_stream_t stream) {
    if (!dense || !packed || !g_okoff || !pair_ok || n_groups <= 0 || max_nc4 <= 0 || mt < 1 || mt > 4 || kt <= 0)
        return HMVAE_ERR_ARG;
    dim3 grid((unsigned)max_nc4, (unsigned)n_groups), block(256);
    const size_t lds = (size_t)kt * mt * 64 * sizeof(float);
    if (lds > 64 * 1024) return HMVAE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pack_kernel, grid, block, lds, (hipStream_t)stream, dense, packed, g_nc4, g_woff, g_choff, chan_id,
                       row_id, g_okoff, pair_ok, mt, kt, row_stride, chan_stride, tap_stride, tap_flip);
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}


// ---- every pack of the model in one launch (23 launches of 4-15 us each were mostly launch floor) ----
// One workgroup packs `cb` consecutive 4-channel steps of one group (4 for plain ops: with 16-row groups that is 256
// (row, channel) pairs, one per thread; their life is a chain of dependent table reads, so fewer and fuller
// workgroups finish sooner).  Composite ops (hmvae.h: virtual taps, merged channels) stage the real taps of every pair
// once -- summed over the dense pairs merged into it -- and build all virtual taps from that.
__global__ __launch_bounds__(256) void pack_multi_kernel(const hmvae_pack_op* __restrict__ ops, int n_ops) {
    extern __shared__ float img[];                    // [cb][kt][mt][64]
    // which op owns this block: one lane-indexed load of the (<= 64) block_start fields + a ballot, instead of a serial
    // scan whose n_ops dependent loads were most of a small block's life
    int oi;
    {
        const int lane = threadIdx.x & 63;
        const int st = lane < n_ops ? ops[lane].block_start : 0x7fffffff;
        const unsigned long long m = __ballot((int)blockIdx.x >= st);
        oi = __popcll(m) - 1;
    }
    const hmvae_pack_op o = ops[oi];
    const int local = blockIdx.x - o.block_start;
    const int cb = o.cb;
    const int cblocks = (o.max_nc4 + cb - 1) / cb;
    const int g = local / cblocks;
    const int c40 = (local - g * cblocks) * cb;
    const int nc4 = o.g_nc4[g];
    if (c40 >= nc4) return;
    const int ncb = nc4 - c40 < cb ? nc4 - c40 : cb;               // steps of this block
    const int mt = o.mt, kt = o.kt;
    const int rows = mt * 16;
    const int ppc = rows * 4;                                       // pairs per step
    const int step_floats = kt * mt * 64;
    const bool chan_inner = o.chan_stride < o.row_stride;
    const int tid = threadIdx.x;
    auto pair_rk = [&](int pair, int& r, int& kk) {
        if (chan_inner) {
            kk = pair & 3;
            r = pair >> 2;
        } else {
            r = pair % rows;
            kk = pair / rows;
        }
    };
    if (o.fold_taps == nullptr) {
        // one (row, channel) pair at a time per thread, its kt taps read back to back
        for (int pi = tid; pi < ncb * ppc; pi += 256) {
            const int st = pi / ppc, pair = pi - st * ppc, c4 = c40 + st;
            int r, kk;
            pair_rk(pair, r, kk);
            const int rid = o.row_id[g * rows + r];
            const int cid = o.chan_id[o.g_choff[g] + c4 * 4 + kk];
            const bool ok = rid >= 0 && cid >= 0 && o.pair_ok[o.g_okoff[g] + r * (nc4 * 4) + c4 * 4 + kk];
            const float* src = o.dense + (ok ? (size_t)rid * o.row_stride + (size_t)cid * o.chan_stride : 0);
            float* dsti = img + st * step_floats + (r >> 4) * 64 + kk * 16 + (r & 15);
            if (HMVAE_PACK_WIDE && kt == 15 && o.tap_stride == 1) {
                float t[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) t[k] = 0.f;
                if (ok) hmvae_add_taps15(src, t);
#pragma unroll
                for (int tap = 0; tap < 15; ++tap) dsti[tap * mt * 64] = t[tap];
            } else {
                for (int tap = 0; tap < kt; ++tap) dsti[tap * mt * 64] = ok ? src[(size_t)tap * o.tap_stride] : 0.f;
            }
        }
    } else {
        // Composite op.  A thread keeps the (<= 16) real taps of one (row, channel) pair in registers -- summed over the
        // dense pairs merged into it -- and produces its virtual taps as a small dense product with the coefficient
        // matrix C[vt][k] (built in LDS from the sparse tables, read back as broadcast float4s): no index chasing.
        // (Chasing table -> tap -> staged value per term was a serial chain of LDS round trips: 3/4 of the kernel.)
        const int kr = o.fold_kt;                              // taps of the dense tensor (host checks kr <= 16)
        const int nf = o.fold_nf;
        float* ct = img + cb * step_floats;                    // [kt][16]
        for (int i = tid; i < kt * 16; i += 256) ct[i] = 0.f;
        __syncthreads();
        for (int i = tid; i < kt * nf; i += 256) {
            const int k = o.fold_taps[i];
            if (k >= 0) ct[(i / nf) * 16 + k] = o.fold_coef[i];          // a tap appears once per virtual tap
        }
        __syncthreads();
        const int np = ncb * ppc;                              // pairs of this block: 64..256
        const int rep = np <= 128 ? (np <= 64 ? 4 : 2) : 1;    // threads per pair (they split the virtual taps)
        const int pi = tid % np, part = tid / np;
        if (part < rep) {
            const int st = pi / ppc, pair = pi - st * ppc, c4 = c40 + st;
            int r, kk;
            pair_rk(pair, r, kk);
            const int rid = o.row_id[g * rows + r];
            const int cid = o.chan_id[o.g_choff[g] + c4 * 4 + kk];
            const unsigned bits = (rid >= 0 && cid >= 0) ? o.pair_ok[o.g_okoff[g] + r * (nc4 * 4) + c4 * 4 + kk] : 0u;
            float t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = 0.f;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {                   // dense (row, channel) pairs merged into this one
                if (!((bits >> ci) & 1u)) continue;
                int dr = rid, dc = cid;
                if (o.child_tab) {
                    if (o.child_side) dr = o.child_tab[rid * 4 + ci];
                    else dc = o.child_tab[cid * 4 + ci];
                }
                const float* src = o.dense + (size_t)dr * o.row_stride + (size_t)dc * o.chan_stride;
                if (HMVAE_PACK_WIDE && kr == 15 && o.tap_stride == 1) {
                    hmvae_add_taps15(src, t);
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < kr) t[k] += src[(size_t)k * o.tap_stride];
                }
            }
            if (bits && o.child_scale) {                       // e.g. 1/2: the packed row is the MEAN of its children
                const float sc = o.child_scale[o.child_side ? rid : cid];
#pragma unroll
                for (int k = 0; k < 16; ++k) t[k] *= sc;
            }
            float* dsti = img + st * step_floats + (r >> 4) * 64 + kk * 16 + (r & 15);
            const int v0 = part * kt / rep, v1 = (part + 1) * kt / rep;
            for (int vt = v0; vt < v1; ++vt) {
                const f32x4* c = (const f32x4*)(ct + vt * 16);
                const f32x4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) v += t[k] * c0[k] + t[4 + k] * c1[k] + t[8 + k] * c2[k] + t[12 + k] * c3[k];
                dsti[vt * mt * 64] = v;
            }
        }
    }
    __syncthreads();
    float* out = o.packed + o.g_woff[g] + (size_t)c40 * step_floats;     // the steps of a group are contiguous
    const int n = ncb * step_floats;
    for (int i = tid; i < n; i += 256) out[i] = img[i];
}

extern "C" int hmvae_pack_weights_multi(const hmvae_pack_op* ops_dev, int32_t n_ops, int32_t total_blocks,
                                        int32_t max_lds_floats, hmvae_stream_t stream) {
    if (!ops_dev || n_ops <= 0 || n_ops > 64 || total_blocks <= 0 || max_lds_floats <= 0) return HMVAE_ERR_ARG;
    const size_t lds = (size_t)max_lds_floats * sizeof(float);
    if (lds > 64 * 1024) return HMVAE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), lds, (hipStream_t)stream, ops_dev, n_ops);
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}
