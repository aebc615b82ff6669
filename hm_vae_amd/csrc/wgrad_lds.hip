// Weight-gradient GEMM, LDS-staged variant (hmvae_wgrad_desc.mode == 1).  gfx950 only.
//
// Same contract as wgrad_kernel (gemm_kernels.hip): slab[part][mt*16][64] = sum over the item's chunks (one position t x
// 32 batch columns) of dz[row, t, b] * x[chan(col), t*stride + tap(col) - pad, b].  What differs is how the operands
// reach the matrix pipe.  The direct kernel loads every fragment from global memory in MFMA layout: lane (n, kk) takes
// 32 bytes of row n, so the 16 lanes of a quarter-wave touch 16 different cache lines per pass -- rocprofv3 counted 64
// L1 accesses per 1 KB load instruction (6.6e7 per launch of the 1127-tile layer, 94 % hits), the matrix pipe 59 % busy
// and the four waves of a workgroup each re-reading nearly the same activation rows (neighbouring positions differ by
// one tap).  Here a workgroup stages, per 4 positions of one 32-column batch block,
//     A: the mt*16 gradient rows at each of the 4 positions              (whole 128-byte lines, 8 lanes per line)
//     B: per input channel of the tile the window of 3*stride + kt rows  (instead of 4 * kt gathered rows)
// in LDS (rows 144 bytes apart: the fragment reads of 16 consecutive rows hit 16 different bank groups) and every wave
// reads the fragments of its position from there.  One buffer per workgroup and TWO workgroups per CU (<= 256 registers,
// <= 80 KB LDS): a workgroup alternates a load phase (global -> registers -> LDS) and an MFMA phase, and with one wave
// per SIMD nothing of the load phase hides behind its own MFMAs (measured: double buffering inside one workgroup left
// the matrix pipe 39 % busy) -- the second workgroup's MFMA phase does.
#include <hip/hip_runtime.h>
#include "hmvae.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int TB = 4;                 // positions per stage = waves per workgroup
constexpr int ROWB = 144;             // LDS row pitch in bytes (128 data + 16 pad)
// BP: passes of 32 rows that cover the activation window of a stage (nch_max * W <= BP * 32); ZP: zero padding (rows
// outside the sequence are zeros instead of reflections).  Every load and LDS write of a stage is unconditional --
// dummy rows, positions past the end of a partial stage and rows past the window read a valid address and are simply
// never used -- so fetch / stash compile to straight-line code (the first version branched around each of them).
template <int MT, int BP, bool ZP>
__global__ __launch_bounds__(256, 2) void wgrad_lds_kernel(hmvae_wgrad_desc d, int nch_max, int W, int WT) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // layout: [utab: nch_max*WT u32][buf: A | B]; after the loop the front is reused as the reduction buffer
    constexpr int AROWS = MT * 16 * TB;
    constexpr int brows = BP * 32;
    unsigned* utab = (unsigned*)lds;
    const int utab_bytes = ((nch_max * WT * 4 + 15) >> 4) << 4;
    char* buf0 = lds + utab_bytes;

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
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4;
    const int n = lane & 15;
    const int bp = d.bp;
    const int T = d.t_dst;
    const int q8 = threadIdx.x & 7;            // 16-byte piece of a 128-byte row
    const int rr = threadIdx.x >> 3;           // row inside a pass of 32 rows

    for (int item = it0; item < it1; ++item) {
        int tile, lo, hi;
        size_t part;
        if (d.n_workers > 0) {
            tile = d.items[4 * item];
            lo = d.items[4 * item + 1];
            hi = d.items[4 * item + 2];
            part = (size_t)d.items[4 * item + 3];
        } else {
            tile = item / d.n_split;
            const int sp = item - tile * d.n_split;
            const long long total = (long long)T * (bp >> 5);
            lo = (int)(total * sp / d.n_split);
            hi = (int)(total * (sp + 1) / d.n_split);
            part = (size_t)sp * d.n_tiles + tile;
        }
        const int g = d.tile_g[tile];
        const int col0 = d.tile_col0[tile];
        const int ncols = d.g_ncols[g];
        const int choff = d.g_choff[g];
        const int kt = d.kt;
        const int ch_lo = col0 / kt;
        const int last_col = (col0 + 63 < ncols ? col0 + 63 : ncols - 1);
        const int nch = last_col / kt - ch_lo + 1;            // <= nch_max (host)

        // ---- tables of this item ----
        // utab[chl][i]: byte offset of channel chl's row at padded position i (source position i - pad), or ~0 = zero row
        for (int i = threadIdx.x; i < nch * WT; i += 256) {
            const int chl = i / WT, v = i - chl * WT;
            const int xr = d.chan_rows[choff + ch_lo + chl];
            int u = v - d.pad;
            unsigned e;
            if (d.pad_mode == HMVAE_PAD_REFLECT) {
                if (u < 0) u = -u;
                if (u >= d.t_src) u = 2 * (d.t_src - 1) - u;
                e = (unsigned)(xr + u * d.x_ts) * (unsigned)bp * 4u;
            } else {
                e = (u >= 0 && u < d.t_src) ? (unsigned)(xr + u * d.x_ts) * (unsigned)bp * 4u : 0xffffffffu;
            }
            utab[i] = e;
        }
        // this thread's rows of a stage: A passes (row index = pass*32 + rr over [p][r]), B passes (over [chl][v])
        constexpr int APASS = AROWS / 32;                         // MT * 64 rows: a whole number of passes
        unsigned a_src[APASS];          // byte offset of row r at position 0, + piece (dummy rows read row 0: never used)
        int a_p[APASS];                 // position inside the stage
        unsigned a_lds[APASS];
#pragma unroll
        for (int ps = 0; ps < APASS; ++ps) {
            const int idx = ps * 32 + rr;
            const int p = idx / (MT * 16), r = idx - p * (MT * 16);
            a_p[ps] = p;
            const int row = d.dz_rows[g * (MT * 16) + r];
            a_src[ps] = (unsigned)(row < 0 ? 0 : row) * (unsigned)bp * 4u + (unsigned)q8 * 16u;
            a_lds[ps] = (unsigned)idx * ROWB + (unsigned)q8 * 16u;
        }
        int b_u[BP];                    // index into utab of (chl, v) at stage position 0 (rows past the window: entry 0)
        unsigned b_lds[BP];
        const int utab_last = nch * WT - 1;
#pragma unroll
        for (int ps = 0; ps < BP; ++ps) {
            const int idx = ps * 32 + rr;
            const int chl = idx / W, v = idx - chl * W;
            b_u[ps] = idx < nch * W ? chl * WT + v : 0;
            b_lds[ps] = (unsigned)(AROWS * ROWB) + (unsigned)idx * ROWB + (unsigned)q8 * 16u;
        }
        // fragment addresses: A row mt*16 + n of position `wave`; B column nt*16 + n -> (chl, tap)
        unsigned fa[MT], fb[4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[mt] = (unsigned)(wave * (MT * 16) + mt * 16 + n) * ROWB + (unsigned)kk * 32u;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int col = col0 + nt * 16 + n;
            col = col < ncols ? col : ncols - 1;                 // columns past the group's end are never read back
            const int ch = col / kt, tp = col - ch * kt;
            fb[nt] = (unsigned)(AROWS * ROWB) + (unsigned)((ch - ch_lo) * W + wave * d.stride + tp) * ROWB + (unsigned)kk * 32u;
        }
        const unsigned dz_tstep = (unsigned)d.dz_ts * (unsigned)bp * 4u;
        const char* __restrict__ dzb = (const char*)d.dz;
        const char* __restrict__ xb = (const char*)d.x;
        __syncthreads();                                          // utab visible

        f32x4 acc[MT][4], tot[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                tot[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }

        // stages: (bb, tb) with tb a multiple of TB; a stage holds the chunks bb*T + tb .. + TB-1 that lie in [lo, hi)
        const int spb = (T + TB - 1) / TB;                        // stages per batch block
        const int s_lo = (lo / T) * spb + (lo % T) / TB;
        const int s_hi = ((hi - 1) / T) * spb + ((hi - 1) % T) / TB + 1;
        f32x4 ra[APASS], rb[BP];
        auto fetch = [&](int s) {                                 // global -> registers (rows of stage s)
            const int bb = s / spb, tb = (s - bb * spb) * TB;
            const size_t boff = (size_t)(bb << 7);
            unsigned e[BP];
#pragma unroll
            for (int ps = 0; ps < BP; ++ps) {
                int i = b_u[ps] + tb * d.stride;
                i = i < utab_last ? i : utab_last;
                e[ps] = utab[i];
            }
#pragma unroll
            for (int ps = 0; ps < APASS; ++ps) {
                int t = tb + a_p[ps];
                t = t < T ? t : T - 1;
                ra[ps] = *(const f32x4*)(dzb + (size_t)t * dz_tstep + boff + a_src[ps]);
            }
#pragma unroll
            for (int ps = 0; ps < BP; ++ps) {
                if (ZP) {
                    const bool ok = e[ps] != 0xffffffffu;
                    const f32x4 v = *(const f32x4*)(xb + boff + (ok ? e[ps] : 0u) + (unsigned)q8 * 16u);
                    rb[ps] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
                } else {
                    rb[ps] = *(const f32x4*)(xb + boff + e[ps] + (unsigned)q8 * 16u);
                }
            }
        };
        auto stash = [&](char* buf) {                             // registers -> LDS
#pragma unroll
            for (int ps = 0; ps < APASS; ++ps) *(f32x4*)(buf + a_lds[ps]) = ra[ps];
#pragma unroll
            for (int ps = 0; ps < BP; ++ps) *(f32x4*)(buf + b_lds[ps]) = rb[ps];
        };
        int n_since = 0;
        for (int s = s_lo; s < s_hi; ++s) {
            char* cur = buf0;
            fetch(s);
            stash(cur);
            __syncthreads();
            {
                const int bb = s / spb, tb = (s - bb * spb) * TB;
                const int t = tb + wave;
                const int chunk = bb * T + t;
                if (t < T && chunk >= lo && chunk < hi) {         // wave-uniform
                    f32x4 a[MT][2], b[4][2];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        a[mt][0] = *(const f32x4*)(cur + fa[mt]);
                        a[mt][1] = *(const f32x4*)(cur + fa[mt] + 16);
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        b[nt][0] = *(const f32x4*)(cur + fb[nt]);
                        b[nt][1] = *(const f32x4*)(cur + fb[nt] + 16);
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MFMA16(a[mt][h][j], b[nt][h][j], acc[mt][nt]);
                    if (++n_since == 8) {                         // two-level accumulation, as in wgrad_kernel
                        n_since = 0;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) {
                                tot[mt][nt] += acc[mt][nt];
                                acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            }
                    }
                }
            }
            __syncthreads();                                      // every wave has read its fragments: the buffer is free
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) tot[mt][nt] += acc[mt][nt];
        // cross-wave sum through LDS and the slab write: as in wgrad_kernel (the loop's last barrier has passed)
        float* red = (float*)lds;
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
            f32x4 sacc = *(const f32x4*)(red + row * 64 + (cq << 2));
#pragma unroll
            for (int w = 1; w < 4; ++w) sacc += *(const f32x4*)(red + w * (MT * 16 * 64) + row * 64 + (cq << 2));
            *(f32x4*)(out + row * 64 + (cq << 2)) = sacc;
        }
        __syncthreads();
    }
}

}  // namespace

// called by hmvae_wgrad_gemm (gemm_kernels.hip) when desc->mode == 1; argument checks are done there
int hmvae_wgrad_lds_launch(const hmvae_wgrad_desc* d, unsigned grid, hipStream_t s) {
    if (d->transposed || d->dz_planes || d->kt < 2) return HMVAE_ERR_UNSUPPORTED;
    const int nch_max = 63 / d->kt + 2;
    const int W = (TB - 1) * d->stride + d->kt;
    const int WT = (d->t_dst - 1) * d->stride + d->kt;
    const int bpass = (nch_max * W + 31) / 32;
    const int BPs = bpass <= 4 ? 4 : (bpass <= 6 ? 6 : 8);
    if (bpass > 8) return HMVAE_ERR_UNSUPPORTED;
    const size_t utab = (((size_t)nch_max * WT * 4 + 15) >> 4) << 4;
    const size_t stage = ((size_t)d->mt * 16 * TB + (size_t)BPs * 32) * ROWB;
    size_t lds = utab + stage;
    const size_t red = (size_t)4 * d->mt * 16 * 64 * sizeof(float);
    if (red > lds) lds = red;
    if (lds > 80 * 1024) return HMVAE_ERR_UNSUPPORTED;       // two workgroups per CU
    const bool zp = d->pad_mode == HMVAE_PAD_ZEROS;
    static bool attr_set[5][9][2] = {};                    // dynamic LDS above 64 KB must be allowed once per kernel
#define WL(MT_, BP_, ZP_)                                                                                               \
    {                                                                                                                   \
        if (!attr_set[MT_][BP_][ZP_]) {                                                                                 \
            if (hipFuncSetAttribute((const void*)wgrad_lds_kernel<MT_, BP_, ZP_>,                                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)              \
                return HMVAE_ERR_LAUNCH;                                                                                \
            attr_set[MT_][BP_][ZP_] = true;                                                                             \
        }                                                                                                               \
        hipLaunchKernelGGL((wgrad_lds_kernel<MT_, BP_, ZP_>), dim3(grid), dim3(256), lds, s, *d, nch_max, W, WT);       \
    }
#define WL_Z(MT_, BP_)      \
    if (zp) WL(MT_, BP_, 1) \
    else WL(MT_, BP_, 0)
#define WL_B(MT_)                      \
    if (BPs == 4) { WL_Z(MT_, 4) }     \
    else if (BPs == 6) { WL_Z(MT_, 6) } \
    else { WL_Z(MT_, 8) }
    switch (d->mt) {
        case 1: WL_B(1) break;
        case 2: WL_B(2) break;
        case 3: WL_B(3) break;
        case 4: WL_B(4) break;
        default: return HMVAE_ERR_UNSUPPORTED;
    }
#undef WL_B
#undef WL_Z
#undef WL
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}
