// Weight-gradient GEMM on the bf16 matrix pipe with fp32-grade results (hmvae_wgrad_desc.mode == 2).  gfx950 only.
//
// Same contract as wgrad_kernel / wgrad_lds_kernel: slab[part][mt*16][64] = sum over the item's chunks (one position t x
// 32 batch columns) of dz[row, t, b] * x[chan(col), t*stride + tap(col) - pad, b]   (skeleton.py:95-105, backward).
//
// Why: the fp32 instruction v_mfma_f32_16x16x4_f32 issues in 32 cycles for K = 4 -- 256 cycles per 32 batch columns and
// tile -- and the fp32 kernels sit at 55-60 % of that.  A float is, exactly, the sum of three bf16 numbers (round to
// nearest at every level: x = p0 + p1 + p2 with |p1| <= 2^-8 |x|, |p2| <= 2^-16 |x|), and of the nine cross products of
// two such triples the six of weight >= 2^-16 carry everything above 2^-24 of the product -- one fp32 rounding.  Those six
// run as v_mfma_f32_16x16x32_bf16 (K = 32 in 16 cycles): 96 cycles for the same K, with fp32 accumulation of whole
// 32-term dot products (the fp32 form rounds after every term).
// The first attempt at this (round 1: planes made by a separate pass, fragments straight from global memory) was 1.4x
// SLOWER: 1.5x the bytes through a load path that was already the limit.  Here the split happens ONCE per element inside
// the kernel, on the way from global memory to LDS, and LDS holds the planes:
//   stage = 4 positions x 32 batch columns of one tile (as wgrad_lds_kernel)
//     A: the mt*16 gradient rows at each of the 4 positions          -> 3 planes of 64-byte rows
//     B: per input channel of the tile its window of 3*stride + kt source rows -> 3 planes of 64-byte rows
//   every thread loads 16-byte pieces (8 lanes cover one 128-byte line), splits its four floats with v_cvt_pk_bf16_f32
//   (22 vector instructions) and writes three 8-byte pieces; wave w then reads the fragments of position w
//   (ds_read_b128: lane (n, kk) takes the 8 consecutive batch columns 8kk.. of row / column n) and issues 72 MFMAs (mt 3).
// LDS rows are 64 bytes; the 16-byte slot of a row is XORed with (row >> 1) & 3, which makes the four 16-lane groups of a
// ds_read_b128 conflict-free for 16 consecutive rows, and a channel's window starts kt + 8m rows after the previous one,
// so the column tiles that straddle two channels (all of them at kt = 15) see consecutive rows modulo 8 as well.
// Two workgroups per CU (<= 80 KB LDS, <= 256 registers): one converts while the other multiplies, and the next stage's
// global loads are issued before the MFMA block and land under it.
// What the counters and two rebuilt variants said about this kernel (round 2, dec0 problem, 1127 tiles, B = 256):
//  * the matrix pipe is 23 % busy, the LDS array 18 %, the texture path 14 %, L2 hit rate 91 %: no unit is saturated; the
//    SIMD's ONE instruction-issue port is -- ~400 vector instructions (the split: 18 per four floats, plus addressing), 72
//    MFMAs (each holds the port for 8 of its 16 cycles) and ~70 LDS / memory instructions per wave and stage;
//  * wave specialisation (4 converting + 4 multiplying waves, two LDS images, one barrier per stage) was 10-20 % SLOWER:
//    the roles still share each SIMD's issue port, and one 8-wave workgroup per CU rounds 4.4 tiles per CU up to 5;
//  * planes split beforehand (hmvae_split_bf16x3) and moved global -> LDS by DMA (global_load_lds_dwordx4, swizzle on the
//    source address) removed every vector instruction of the split and gained 5-10 % on the kernel -- less than the
//    15-20 us the extra pass costs per layer: 1.5x the bytes meet the CU's L2 -> LDS rate (~30 B/cycle), and each DMA
//    piece costs the issuing wave 100+ cycles.
// What is left on this structure is arithmetic intensity: a 48 x 128 tile halves the gradient rows' bytes and split work
// per MFMA.
#include <hip/hip_runtime.h>
#include "hmvae.h"
#include "wreduce_inl.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TB = 4;                 // positions per stage = waves per workgroup
constexpr int ROWB = 64;              // LDS row: 32 bf16

// x = p0 + p1 + p2 exactly, round-to-nearest-even at each level; returns the three packed pairs of (x0, x1)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h0, unsigned& h1, unsigned& h2) {
    f32x2 v = {x0, x1};
    h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r = {x0 - __uint_as_float(h0 << 16), x1 - __uint_as_float(h0 & 0xffff0000u)};
    h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    f32x2 s = {r[0] - __uint_as_float(h1 << 16), r[1] - __uint_as_float(h1 & 0xffff0000u)};
    h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(s, bf16x2));
}

__device__ __forceinline__ unsigned swz(unsigned row, unsigned slot) { return (slot ^ ((row >> 1) & 3u)) << 4; }

// NH = 1: a workgroup (4 waves, two per CU) owns one 64-column tile.  NH = 2: a workgroup of 8 waves (one per CU) owns a
// PAIR of neighbouring column tiles of one row group (desc.pair_tab): waves 0-3 multiply the first tile, waves 4-7 the
// second, all 512 threads stage -- the gradient rows (the larger operand: every position needs its own) are loaded and
// split once for 128 columns, i.e. half the bytes and half the vector instructions per MFMA for them.
template <int MT, int BP, bool ZP, int NH>
__global__ __launch_bounds__(256 * NH, NH == 1 ? 2 : 1) void wgrad_bf3_kernel(hmvae_wgrad_desc d, int nch_max, int W, int WP,
                                                                              int WT, int brows) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // layout: [utab: nch_max*WT u32][A planes 3 x AROWS x 64][B planes 3 x brows x 64]; afterwards the reduction buffer.
    // BP counts the LOAD passes (THREADS / 8 rows each) over the compact (channel, window row) index space; the image
    // itself spaces the channels' windows WP rows apart (brows = nch_max * WP + one spare row that takes the slots past the
    // tile's windows).
    constexpr int THREADS = 256 * NH;
    constexpr int PROWS = THREADS / 8;                 // rows per load pass
    constexpr int AROWS = MT * 16 * TB;
    constexpr unsigned APL = AROWS * ROWB;
    const unsigned BPL = (unsigned)brows * ROWB;
    unsigned* utab = (unsigned*)lds;
    const int utab_bytes = ((nch_max * WT * 4 + 15) >> 4) << 4;
    char* bufA = lds + utab_bytes;
    char* bufB = bufA + 3 * APL;

    const int n_units = NH == 2 ? d.n_pairs : d.n_tiles;          // what an item's first index counts
    int it0, it1;
    if (NH == 1 && d.n_workers > 0) {
        const int perw = d.n_workers >> 3;
        const int worker = (blockIdx.x & 7) * perw + (blockIdx.x >> 3);
        it0 = d.worker_items[2 * worker];
        it1 = it0 + d.worker_items[2 * worker + 1];
    } else {
        const int n_items = n_units * d.n_split;
        const int per = (n_items + 7) >> 3;
        it0 = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= per || it0 >= n_items) return;
        it1 = it0 + 1;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pos = wave & 3;                  // this wave's position inside a stage
    const int half = wave >> 2;                // ... and its column tile of the pair
    const int kk = lane >> 4;
    const int n = lane & 15;
    const int bp = d.bp;
    const int T = d.t_dst;
    const int q8 = threadIdx.x & 7;            // 16-byte piece of a 128-byte fp32 row = 8-byte piece of a 64-byte plane row
    const int rr = threadIdx.x >> 3;           // row inside a load pass

    for (int item = it0; item < it1; ++item) {
        int tile, tile2 = -1, lo, hi, sp = 0;
        size_t part;
        if (NH == 1 && d.n_workers > 0) {
            tile = d.items[4 * item];
            lo = d.items[4 * item + 1];
            hi = d.items[4 * item + 2];
            part = (size_t)d.items[4 * item + 3];
        } else {
            const int unit = item / d.n_split;
            sp = item - unit * d.n_split;
            const long long total = (long long)T * (bp >> 5);
            lo = (int)(total * sp / d.n_split);
            hi = (int)(total * (sp + 1) / d.n_split);
            if (NH == 2) {
                tile = d.pair_tab[2 * unit];
                tile2 = d.pair_tab[2 * unit + 1];              // -1: the group's last, unpaired tile
            } else {
                tile = unit;
            }
            part = (size_t)sp * d.n_tiles + tile;
        }
        const int my_tile = half ? tile2 : tile;               // wave-uniform; -1: nothing to multiply
        const int g = d.tile_g[tile];
        const int col0 = d.tile_col0[tile];                    // the pair's second tile starts 64 columns later
        const int ncols = d.g_ncols[g];
        const int choff = d.g_choff[g];
        const int kt = d.kt;
        const int ch_lo = col0 / kt;
        const int span = (NH == 2 && tile2 >= 0) ? 128 : 64;
        const int last_col = (col0 + span - 1 < ncols ? col0 + span - 1 : ncols - 1);
        const int nch = last_col / kt - ch_lo + 1;            // <= nch_max (host)

        // utab[chl][i]: byte offset of channel chl's row at padded position i (source position i - pad), or ~0 = zero row
        for (int i = threadIdx.x; i < nch * WT; i += THREADS) {
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
        // this thread's rows of a stage: A passes (row index = pass*PROWS + rr over [p][r]), B passes (over [chl][v]).
        // Every load, split and store of a stage is unconditional (dummy rows read row 0, slots past the windows go to the
        // spare row): a version that skipped them under per-lane conditions lost the overlap of the prefetch with the
        // MFMA block (the compiler waits for everything at the first branch) and ran 20 % slower.
        constexpr int APASS = AROWS / PROWS;
        unsigned a_src[APASS];          // byte offset of row r at position 0, + piece: 32-bit, added to a uniform base
        int a_p[APASS];                 // position inside the stage
        unsigned a_lds[APASS];          // plane-0 byte address of this thread's 8-byte piece
#pragma unroll
        for (int ps = 0; ps < APASS; ++ps) {
            const int idx = ps * PROWS + rr;
            const int p = idx / (MT * 16), r = idx - p * (MT * 16);
            a_p[ps] = p;
            const int row = d.dz_rows[g * (MT * 16) + r];
            a_src[ps] = (unsigned)(row < 0 ? 0 : row) * (unsigned)bp * 4u + (unsigned)q8 * 16u;
            a_lds[ps] = (unsigned)idx * ROWB + swz((unsigned)idx, (unsigned)q8 >> 1) + ((unsigned)q8 & 1u) * 8u;
        }
        int b_u[BP];                    // index into utab of (chl, v) at stage position 0
        unsigned b_lds[BP];
        const int utab_last = nch * WT - 1;
#pragma unroll
        for (int ps = 0; ps < BP; ++ps) {
            const int idx = ps * PROWS + rr;
            const int chl = idx / W, v = idx - chl * W;
            const bool in = idx < nch * W;
            b_u[ps] = in ? chl * WT + v : 0;
            const unsigned lrow = in ? (unsigned)(chl * WP + v) : (unsigned)(brows - 1);
            b_lds[ps] = lrow * ROWB + swz(lrow, (unsigned)q8 >> 1) + ((unsigned)q8 & 1u) * 8u;
        }
        // fragment addresses: A row mt*16 + n of position `pos`; B column nt*16 + n of this wave's tile -> (chl, tap)
        unsigned fa[MT], fb[4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned ra = (unsigned)(pos * (MT * 16) + mt * 16 + n);
            fa[mt] = ra * ROWB + swz(ra, (unsigned)kk);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int col = col0 + half * 64 + nt * 16 + n;
            col = col < ncols ? col : ncols - 1;                 // columns past the group's end are never read back
            const int ch = col / kt, tp = col - ch * kt;
            const unsigned rb = (unsigned)((ch - ch_lo) * WP + pos * d.stride + tp);
            fb[nt] = rb * ROWB + swz(rb, (unsigned)kk);
        }
        const unsigned dz_tstep = (unsigned)d.dz_ts * (unsigned)bp * 4u;
        const char* __restrict__ dzb = (const char*)d.dz;
        const char* __restrict__ xb = (const char*)d.x;
        // fused epilogue: this thread's column of its tile (channel, tap, mask pointer) and the ids / mask bytes of its
        // rows are looked up NOW -- chains of dependent global loads that would otherwise sit exposed at the tile's end
        const int e_tile = (threadIdx.x >> 8) ? tile2 : tile;  // the tile this thread writes in the epilogue
        WreduceColR wc = {};
        int e_rid[MT * 4];
        unsigned e_bits[MT * 4];
        if (d.fused_reduce != nullptr && e_tile >= 0) {
            const hmvae_wreduce_desc rd = *d.fused_reduce;     // local copy: stores cannot alias it
            wc = wreduce_col_resolve(rd, e_tile, threadIdx.x & 63);
#pragma unroll
            for (int i = 0; i < MT * 4; ++i) {
                const int row = ((threadIdx.x >> 6) & 3) + 4 * i;
                e_rid[i] = rd.row_id[g * (MT * 16) + row];
                e_bits[i] = wc.live ? wc.okp[row * wc.ok_stride] : 0u;
            }
        }
        __syncthreads();                                          // utab visible

        f32x4 acc[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // stages: (bb, tb) with tb a multiple of TB; a stage holds the chunks bb*T + tb .. + TB-1 that lie in [lo, hi)
        const int spb = (T + TB - 1) / TB;                        // stages per batch block
        const int s_lo = (lo / T) * spb + (lo % T) / TB;
        const int s_hi = ((hi - 1) / T) * spb + ((hi - 1) % T) / TB + 1;
        f32x4 ra[APASS], rb[BP];
        auto fetch = [&](int bb, int tb) {                        // global -> registers (rows of stage (bb, tb))
            const char* __restrict__ dzs = dzb + (size_t)(bb << 7);       // uniform bases, 32-bit per-lane offsets
            const char* __restrict__ xs = xb + (size_t)(bb << 7);
#pragma unroll
            for (int ps = 0; ps < APASS; ++ps) {
                int t = tb + a_p[ps];
                t = t < T ? t : T - 1;
                ra[ps] = *(const f32x4*)(dzs + ((unsigned)t * dz_tstep + a_src[ps]));
            }
#pragma unroll
            for (int ps = 0; ps < BP; ++ps) {
                int i = b_u[ps] + tb * d.stride;
                i = i < utab_last ? i : utab_last;
                const unsigned e = utab[i];
                if (ZP) {
                    const bool ok = e != 0xffffffffu;
                    const f32x4 v = *(const f32x4*)(xs + ((ok ? e : 0u) + (unsigned)q8 * 16u));
                    rb[ps] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
                } else {
                    rb[ps] = *(const f32x4*)(xs + (e + (unsigned)q8 * 16u));
                }
            }
        };
        auto put = [&](char* base, unsigned plane_bytes, unsigned off, const f32x4& v) {   // split + three 8-byte stores
            unsigned h0a, h1a, h2a, h0b, h1b, h2b;
            split2(v[0], v[1], h0a, h1a, h2a);
            split2(v[2], v[3], h0b, h1b, h2b);
            *(u32x2*)(base + off) = (u32x2){h0a, h0b};
            *(u32x2*)(base + plane_bytes + off) = (u32x2){h1a, h1b};
            *(u32x2*)(base + 2 * plane_bytes + off) = (u32x2){h2a, h2b};
        };
        auto stash = [&]() {                                      // registers -> planes in LDS
#pragma unroll
            for (int ps = 0; ps < APASS; ++ps) put(bufA, APL, a_lds[ps], ra[ps]);
#pragma unroll
            for (int ps = 0; ps < BP; ++ps) put(bufB, BPL, b_lds[ps], rb[ps]);
        };
        int bb = s_lo / spb, tb = (s_lo - bb * spb) * TB;         // walked incrementally: no division in the loop
        fetch(bb, tb);
        for (int s = s_lo; s < s_hi; ++s) {
            stash();
            __syncthreads();
            int nbb = bb, ntb = tb + TB;
            if (ntb >= spb * TB) { ntb = 0; ++nbb; }
            if (s + 1 < s_hi) fetch(nbb, ntb);                    // in flight during the MFMA block below
            __builtin_amdgcn_sched_barrier(0);
            {
                const int t = tb + pos;
                const int chunk = bb * T + t;
                if (my_tile >= 0 && t < T && chunk >= lo && chunk < hi) {         // wave-uniform
                    bf16x8 a[MT][3], b[4][3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) a[mt][pl] = *(const bf16x8*)(bufA + pl * APL + fa[mt]);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) b[nt][pl] = *(const bf16x8*)(bufB + pl * BPL + fb[nt]);
                    }
                    // smallest cross terms first: (0,2) (2,0) (1,1) | (0,1) (1,0) | (0,0)
                    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][PA[q]], b[nt][PB[q]], acc[mt][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            bb = nbb;
            tb = ntb;
            __syncthreads();                                      // every wave has read its fragments: the image is free
        }
        // cross-wave sum through LDS and the write-out (the loop's last barrier has passed): waves 0-3 hold the first
        // tile's partial sums, waves 4-7 the second's; threads 0-255 / 256-511 finish the first / second tile
        float* red = (float*)lds;
        float* mine = red + wave * (MT * 16 * 64);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(mt * 16 + kk * 4 + r) * 64 + nt * 16 + n] = acc[mt][nt][r];
        __syncthreads();
        const float* redh = red + (threadIdx.x >> 8) * (4 * MT * 16 * 64);
        const int t256 = threadIdx.x & 255;
        if (e_tile >= 0) {
            if (d.fused_reduce != nullptr) {
                // the tile goes straight into the dense gradient: thread = one column x 4 of every 16 rows (as
                // wreduce_kernel, so a column's (channel, tap) split and mask pointer are computed once)
                const hmvae_wreduce_desc rd = *d.fused_reduce;
                const int col = t256 & 63, r0 = t256 >> 6;
#pragma unroll
                for (int i0 = 0; i0 < MT * 4; i0 += 4) {
                    int rid4[4];
                    unsigned bits4[4];
                    float s4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = r0 + 4 * (i0 + i);
                        float sacc = redh[row * 64 + col];
#pragma unroll
                        for (int w = 1; w < 4; ++w) sacc += redh[w * (MT * 16 * 64) + row * 64 + col];
                        s4[i] = sacc;
                        rid4[i] = e_rid[i0 + i];
                        bits4[i] = e_bits[i0 + i];
                    }
                    wreduce_store_rows<4>(rd, wc, rid4, bits4, s4);
                }
            } else {
                const int rl = t256 >> 4;
                const int cq = t256 & 15;
                float* out = d.slab + ((size_t)sp * d.n_tiles + e_tile) * (MT * 16 * 64);
                if (NH == 1) out = d.slab + part * (MT * 16 * 64);               // (persistent launches number their parts)
#pragma unroll
                for (int ps = 0; ps < MT; ++ps) {
                    const int row = ps * 16 + rl;
                    f32x4 sacc = *(const f32x4*)(redh + row * 64 + (cq << 2));
#pragma unroll
                    for (int w = 1; w < 4; ++w) sacc += *(const f32x4*)(redh + w * (MT * 16 * 64) + row * 64 + (cq << 2));
                    *(f32x4*)(out + row * 64 + (cq << 2)) = sacc;
                }
            }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Wide tiles (round 3): one workgroup of 8 waves per CU owns up to 112 rows x 128 columns (MT = 5..7 row tiles, a PAIR of
// column tiles).  Why: the 48 x 64 tile above re-reads its operands from L2 once per tile -- (48 + ~32 window rows) x K x 4
// bytes for 48 x 64 x K products: on the 336 x 10080 x 2048 problem of the first decoder layer that is 740 MB per launch at
// the ~6-8 TB/s the L2 fabric delivers, i.e. the launch's whole time -- and splits every staged element for only 0.9 MFMAs.
// A 112 x 128 tile stages 0.5 rows per MFMA instead of 1.1 (both the bytes and the split's vector instructions), and 252
// equal tiles of 16 stages fill the 256 CUs in ONE round with the reduction unsplit, so the epilogue writes the dense
// gradient and no slab / reduce launch is needed.  Differences to the kernel above:
//  * wave w multiplies position (w & 3) of the stage for column tile (w >> 2) over ALL row tiles; the A fragments of a row
//    tile are read when that row tile's 24 MFMAs are due (three registers sets, not 3 x MT), the B fragments once per stage;
//  * the image holds only the group's LIVE rows (groups are cut evenly, the last 16-row tiles of a group may be empty): row
//    p * (16 * nlive) + r, nlive <= MT passes of 64 rows for the gradient operand, dead row tiles cost neither loads nor
//    MFMAs;
//  * the cross-position sum and the write-out run row tile by row tile through a 32 KB buffer (the whole 112 x 128 tile of
//    eight waves would need 229 KB);
//  * the fused epilogue looks its row ids / mask bytes up when it needs them (7 x 4 x 2 registers held across the main loop
//    would cost the second wave per SIMD).
template <int MT, int BP, bool ZP>
__global__ __launch_bounds__(512, 1) void wgrad_bf3w_kernel(hmvae_wgrad_desc d, int nch_max, int W, int WP, int WT, int brows) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ int s_nlive;
    constexpr int THREADS = 512;
    constexpr int PROWS = THREADS / 8;                 // 64 rows per load pass
    constexpr int AROWS = MT * 16 * TB;
    constexpr unsigned APL = AROWS * ROWB;
    const unsigned BPL = (unsigned)brows * ROWB;
    unsigned* utab = (unsigned*)lds;
    const int utab_bytes = ((nch_max * WT * 4 + 15) >> 4) << 4;
    char* bufA = lds + utab_bytes;
    char* bufB = bufA + 3 * APL;

    const int n_items = d.n_pairs * d.n_split;
    const int per = (n_items + 7) >> 3;
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || item >= n_items) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pos = wave & 3;
    const int half = wave >> 2;
    const int kk = lane >> 4;
    const int n = lane & 15;
    const int bp = d.bp;
    const int T = d.t_dst;
    const int q8 = threadIdx.x & 7;
    const int rr = threadIdx.x >> 3;

    const int unit = item / d.n_split;
    const int sp = item - unit * d.n_split;
    const long long total = (long long)T * (bp >> 5);
    const int lo = (int)(total * sp / d.n_split);
    const int hi = (int)(total * (sp + 1) / d.n_split);
    const int tile = d.pair_tab[2 * unit];
    const int tile2 = d.pair_tab[2 * unit + 1];                // -1: the group's last, unpaired tile
    const int my_tile = half ? tile2 : tile;                   // wave-uniform; -1: nothing to multiply
    const int g = d.tile_g[tile];
    const int col0 = d.tile_col0[tile];
    const int ncols = d.g_ncols[g];
    const int choff = d.g_choff[g];
    const int kt = d.kt;
    const int ch_lo = col0 / kt;
    const int span = tile2 >= 0 ? 128 : 64;
    const int last_col = (col0 + span - 1 < ncols ? col0 + span - 1 : ncols - 1);
    const int nch = last_col / kt - ch_lo + 1;

    if (threadIdx.x == 0) s_nlive = 1;
    __syncthreads();
    if (threadIdx.x < MT * 16 && d.dz_rows[g * (MT * 16) + threadIdx.x] >= 0) atomicMax(&s_nlive, ((int)threadIdx.x >> 4) + 1);
    for (int i = threadIdx.x; i < nch * WT; i += THREADS) {
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
    __syncthreads();
    const int nlive = __builtin_amdgcn_readfirstlane(s_nlive);      // live 16-row tiles of this group (uniform)
    const int arows = nlive * 16;                                   // image rows per position

    // A passes over the compact index space [p][r < arows]: exactly nlive passes of 64 rows
    unsigned a_src[MT];
    int a_p[MT];
    unsigned a_lds[MT];
#pragma unroll
    for (int ps = 0; ps < MT; ++ps) {
        const int idx = ps * PROWS + rr;
        const int p = idx / arows, r = idx - p * arows;
        a_p[ps] = p;
        const int row = ps < nlive ? d.dz_rows[g * (MT * 16) + r] : 0;
        a_src[ps] = (unsigned)(row < 0 ? 0 : row) * (unsigned)bp * 4u + (unsigned)q8 * 16u;
        a_lds[ps] = (unsigned)idx * ROWB + swz((unsigned)idx, (unsigned)q8 >> 1) + ((unsigned)q8 & 1u) * 8u;
    }
    int b_u[BP];
    unsigned b_lds[BP];
    const int utab_last = nch * WT - 1;
#pragma unroll
    for (int ps = 0; ps < BP; ++ps) {
        const int idx = ps * PROWS + rr;
        const int chl = idx / W, v = idx - chl * W;
        const bool in = idx < nch * W;
        b_u[ps] = in ? chl * WT + v : 0;
        const unsigned lrow = in ? (unsigned)(chl * WP + v) : (unsigned)(brows - 1);
        b_lds[ps] = lrow * ROWB + swz(lrow, (unsigned)q8 >> 1) + ((unsigned)q8 & 1u) * 8u;
    }
    const int bpass_live = (nch * W + PROWS - 1) / PROWS;           // uniform: passes that hold at least one window row
    unsigned fa[MT], fb[4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const unsigned ra = (unsigned)(pos * arows + mt * 16 + n);
        fa[mt] = ra * ROWB + swz(ra, (unsigned)kk);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        int col = col0 + half * 64 + nt * 16 + n;
        col = col < ncols ? col : ncols - 1;
        const int ch = col / kt, tp = col - ch * kt;
        const unsigned rb = (unsigned)((ch - ch_lo) * WP + pos * d.stride + tp);
        fb[nt] = rb * ROWB + swz(rb, (unsigned)kk);
    }
    const unsigned dz_tstep = (unsigned)d.dz_ts * (unsigned)bp * 4u;
    const char* __restrict__ dzb = (const char*)d.dz;
    const char* __restrict__ xb = (const char*)d.x;

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int spb = (T + TB - 1) / TB;
    const int s_lo = (lo / T) * spb + (lo % T) / TB;
    const int s_hi = ((hi - 1) / T) * spb + ((hi - 1) % T) / TB + 1;
    f32x4 ra[MT], rb[BP];
    auto fetch = [&](int bb, int tb) {
        const char* __restrict__ dzs = dzb + (size_t)(bb << 7);
        const char* __restrict__ xs = xb + (size_t)(bb << 7);
#pragma unroll
        for (int ps = 0; ps < MT; ++ps) {
            if (ps < nlive) {
                int t = tb + a_p[ps];
                t = t < T ? t : T - 1;
                ra[ps] = *(const f32x4*)(dzs + ((unsigned)t * dz_tstep + a_src[ps]));
            }
        }
#pragma unroll
        for (int ps = 0; ps < BP; ++ps) {
            if (ps < bpass_live) {
                int i = b_u[ps] + tb * d.stride;
                i = i < utab_last ? i : utab_last;
                const unsigned e = utab[i];
                if (ZP) {
                    const bool ok = e != 0xffffffffu;
                    const f32x4 v = *(const f32x4*)(xs + ((ok ? e : 0u) + (unsigned)q8 * 16u));
                    rb[ps] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
                } else {
                    rb[ps] = *(const f32x4*)(xs + (e + (unsigned)q8 * 16u));
                }
            }
        }
    };
    auto put = [&](char* base, unsigned plane_bytes, unsigned off, const f32x4& v) {
        unsigned h0a, h1a, h2a, h0b, h1b, h2b;
        split2(v[0], v[1], h0a, h1a, h2a);
        split2(v[2], v[3], h0b, h1b, h2b);
        *(u32x2*)(base + off) = (u32x2){h0a, h0b};
        *(u32x2*)(base + plane_bytes + off) = (u32x2){h1a, h1b};
        *(u32x2*)(base + 2 * plane_bytes + off) = (u32x2){h2a, h2b};
    };
    auto stash = [&]() {
#pragma unroll
        for (int ps = 0; ps < MT; ++ps)
            if (ps < nlive) put(bufA, APL, a_lds[ps], ra[ps]);
#pragma unroll
        for (int ps = 0; ps < BP; ++ps)
            if (ps < bpass_live) put(bufB, BPL, b_lds[ps], rb[ps]);
    };
    int bb = s_lo / spb, tb = (s_lo - bb * spb) * TB;
    fetch(bb, tb);
    for (int s = s_lo; s < s_hi; ++s) {
        stash();
        __syncthreads();
        int nbb = bb, ntb = tb + TB;
        if (ntb >= spb * TB) { ntb = 0; ++nbb; }
        if (s + 1 < s_hi) fetch(nbb, ntb);
        __builtin_amdgcn_sched_barrier(0);
        {
            const int t = tb + pos;
            const int chunk = bb * T + t;
            if (my_tile >= 0 && t < T && chunk >= lo && chunk < hi) {
                bf16x8 b[4][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) b[nt][pl] = *(const bf16x8*)(bufB + pl * BPL + fb[nt]);
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (mt < nlive) {
                        bf16x8 a[3];
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) a[pl] = *(const bf16x8*)(bufA + pl * APL + fa[mt]);
#pragma unroll
                        for (int q = 0; q < 6; ++q)
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], b[nt][PB[q]], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        bb = nbb;
        tb = ntb;
        __syncthreads();
    }
    // epilogue, one 16-row tile at a time: the four position waves of a column tile leave their partial sums in LDS, the 256
    // threads of that half add them and write the rows out
    float* red = (float*)lds;
    float* mine = red + wave * (16 * 64);
    const int e_tile = (threadIdx.x >> 8) ? tile2 : tile;
    const float* redh = red + (threadIdx.x >> 8) * (4 * 16 * 64);
    const int t256 = threadIdx.x & 255;
    WreduceColR wc = {};
    hmvae_wreduce_desc rd = {};
    if (d.fused_reduce != nullptr) {
        rd = *d.fused_reduce;                                   // local copy: stores cannot alias it
        if (e_tile >= 0) wc = wreduce_col_resolve(rd, e_tile, t256 & 63);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt < nlive) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(kk * 4 + r) * 64 + nt * 16 + n] = acc[mt][nt][r];
            __syncthreads();
            if (e_tile >= 0) {
                if (d.fused_reduce != nullptr) {
                    const int col = t256 & 63, r0 = t256 >> 6;
                    int rid[4];
                    unsigned bits[4];
                    float s4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = mt * 16 + r0 + 4 * i;
                        rid[i] = rd.row_id[g * (MT * 16) + row];
                        bits[i] = wc.live ? wc.okp[row * wc.ok_stride] : 0u;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = r0 + 4 * i;
                        float sacc = redh[row * 64 + col];
#pragma unroll
                        for (int w = 1; w < 4; ++w) sacc += redh[w * (16 * 64) + row * 64 + col];
                        s4[i] = sacc;
                    }
                    wreduce_store_rows<4>(rd, wc, rid, bits, s4);
                } else {
                    const int rl = t256 >> 4, cq = t256 & 15;
                    float* out = d.slab + ((size_t)sp * d.n_tiles + e_tile) * (MT * 16 * 64) + (mt * 16 + rl) * 64 + (cq << 2);
                    f32x4 sacc = *(const f32x4*)(redh + rl * 64 + (cq << 2));
#pragma unroll
                    for (int w = 1; w < 4; ++w) sacc += *(const f32x4*)(redh + w * (16 * 64) + rl * 64 + (cq << 2));
                    *(f32x4*)out = sacc;
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace

// Geometry shared with the host-side eligibility check (ops.py mirrors it): returns the LDS bytes, or 0 if unsupported.
static size_t bf3_geometry(const hmvae_wgrad_desc* d, int* nch_max, int* W, int* WP, int* WT, int* BPs, int* brows) {
    const bool pair = d->pair_tab != nullptr;
    if (d->transposed || d->dz_planes || d->kt < 2 || d->t_dst < TB || d->mt < 1 || d->mt > (pair ? 7 : 4)) return 0;
    if (d->fused_reduce && (d->n_split != 1 || d->n_workers != 0)) return 0;
    if (pair && (d->n_pairs <= 0 || d->n_workers != 0)) return 0;
    *nch_max = (pair ? 127 : 63) / d->kt + 2;
    *W = (TB - 1) * d->stride + d->kt;
    *WP = d->kt;
    while (*WP < *W) *WP += 8;                                   // window pitch = kt (mod 8): see the header comment
    *WT = (d->t_dst - 1) * d->stride + d->kt;
    *brows = *nch_max * *WP + 1;                                 // + the spare row
    const int prows = pair ? 64 : 32;
    const int bpass = (*nch_max * *W + prows - 1) / prows;       // load passes over the compact (chl, v) index space
    if (bpass > 8) return 0;
    *BPs = bpass <= 4 ? 4 : (bpass <= 5 ? 5 : (bpass <= 6 ? 6 : 8));
    const size_t utab = (((size_t)*nch_max * *WT * 4 + 15) >> 4) << 4;
    const size_t stage = 3 * ((size_t)d->mt * 16 * TB + (size_t)*brows) * ROWB;
    size_t lds = utab + stage;
    const size_t red = d->mt > 4 ? (size_t)8 * 16 * 64 * sizeof(float)          // wide tiles: one 16-row tile at a time
                                 : (size_t)(pair ? 8 : 4) * d->mt * 16 * 64 * sizeof(float);
    if (red > lds) lds = red;
    if (lds > (pair ? 160 : 80) * 1024) return 0;                // two 4-wave workgroups per CU, or one of 8 waves
    return lds;
}

// called by hmvae_wgrad_gemm (gemm_kernels.hip) when desc->mode == 2; argument checks are done there
int hmvae_wgrad_bf3_launch(const hmvae_wgrad_desc* d, unsigned grid, hipStream_t s) {
    int nch_max, W, WP, WT, BPs, brows;
    const size_t lds = bf3_geometry(d, &nch_max, &W, &WP, &WT, &BPs, &brows);
    if (!lds) return HMVAE_ERR_UNSUPPORTED;
    const bool zp = d->pad_mode == HMVAE_PAD_ZEROS;
    const bool pair = d->pair_tab != nullptr;
    if (pair) {
        const long long n_items = (long long)d->n_pairs * d->n_split;
        grid = (unsigned)(((n_items + 7) / 8) * 8);
    }
    if (pair && d->mt > 4) {
        static bool wattr[8][9][2] = {};
#define WW(MT_, BP_, ZP_)                                                                                              \
    {                                                                                                                  \
        if (!wattr[MT_][BP_][ZP_]) {                                                                                   \
            if (hipFuncSetAttribute((const void*)wgrad_bf3w_kernel<MT_, BP_, ZP_>,                                     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess)        \
                return HMVAE_ERR_LAUNCH;                                                                               \
            wattr[MT_][BP_][ZP_] = true;                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL((wgrad_bf3w_kernel<MT_, BP_, ZP_>), dim3(grid), dim3(512), lds, s, *d, nch_max, W, WP, WT,  \
                           brows);                                                                                     \
    }
#define WW_Z(MT_, BP_)      \
    if (zp) WW(MT_, BP_, 1) \
    else WW(MT_, BP_, 0)
#define WW_B(MT_) WW_Z(MT_, 4)
        // (at most four load passes over the window rows: every 15-tap layer fits; eight would spill at MT = 7.  The kernel's
        // static s_nlive sits beside the dynamic block.)
        if (BPs > 4 || lds + 64 > 160 * 1024) return HMVAE_ERR_UNSUPPORTED;
        switch (d->mt) {
            case 5: WW_B(5) break;
            case 6: WW_B(6) break;
            case 7: WW_B(7) break;
            default: return HMVAE_ERR_UNSUPPORTED;
        }
#undef WW_B
#undef WW_Z
#undef WW
        return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
    }
    static bool attr_set[5][9][2][3] = {};                 // dynamic LDS above 64 KB must be allowed once per kernel
#define WL(MT_, BP_, ZP_, NH_)                                                                                          \
    {                                                                                                                   \
        if (!attr_set[MT_][BP_][ZP_][NH_]) {                                                                            \
            if (hipFuncSetAttribute((const void*)wgrad_bf3_kernel<MT_, BP_, ZP_, NH_>,                                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)              \
                return HMVAE_ERR_LAUNCH;                                                                                \
            attr_set[MT_][BP_][ZP_][NH_] = true;                                                                        \
        }                                                                                                               \
        hipLaunchKernelGGL((wgrad_bf3_kernel<MT_, BP_, ZP_, NH_>), dim3(grid), dim3(256 * NH_), lds, s, *d, nch_max, W, \
                           WP, WT, brows);                                                                              \
    }
#define WL_N(MT_, BP_, ZP_)        \
    if (pair) WL(MT_, BP_, ZP_, 2) \
    else WL(MT_, BP_, ZP_, 1)
#define WL_Z(MT_, BP_)        \
    if (zp) WL_N(MT_, BP_, 1) \
    else WL_N(MT_, BP_, 0)
#define WL_B(MT_)                       \
    if (BPs == 4) { WL_Z(MT_, 4) }      \
    else if (BPs == 5) { WL_Z(MT_, 5) } \
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
#undef WL_N
#undef WL
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}
