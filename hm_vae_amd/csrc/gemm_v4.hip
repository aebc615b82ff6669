// gemm_v4.hip — gather-GEMM, fourth generation: fp32-grade results on the bf16 matrix pipe (hmvae_gather_gemm_v4,
// hmvae_pack_weights_v4; contract in include/hmvae.h).  gfx950 only.
//
// Same maths as gemm_v2.hip (SkeletonConv forward / backward-data with its mask, padding, stride and the linear layers
// around it folded into the weights: reference skeleton.py:95-105, seq_two_hier_sa_vae.py:126-129, 233-242).
// Why a fourth kernel: v_mfma_f32_16x16x4_f32 issues at the fp32 vector rate and the v2 kernel keeps that pipe ~46 % busy,
// because a 16-row group gives every loaded activation ONE row tile to feed (the CU's load path, ~20 B/cycle, is the
// limit).  Here
//   * a float is the exact sum of three bf16 numbers; six of the nine cross products carry everything above 2^-24 of
//     a product and run as v_mfma_f32_16x16x32_bf16 (fp32 accumulation): 96 cycles for K = 32 where fp32 needs 256;
//   * the weights are packed once per step straight into A-operand order as fp32 (32 bytes per lane and row tile, read
//     from L2 by the one wave that owns the tile) and split into their planes in registers: 4 instead of 6 bytes per
//     weight through the load path -- the weight fragments are the larger half of a K step's bytes;
//   * the activations are split once per workgroup on their way into LDS (row-major bf16 planes, 16-byte global loads,
//     8-byte LDS stores) and shared by the 4 waves, i.e. by 4*mtw row tiles; the B operands come out of LDS already
//     transposed (ds_read_b64_tr_b16: lane (n, kk) gets rows 8kk .. 8kk+7 of column n), so the batch-innermost layout
//     of the activations needs no shuffle;
//   * groups hold up to 64 / 128 rows (planner.make_groups_v4: rows that share a channel list are pooled before they
//     are cut), so one staged K step feeds 8-16 row tiles instead of 1.
// LDS image of a K step (32 rows x nc columns, three planes, double buffered): row pitch nc*2 bytes, the 32-byte chunk
// (16 columns) of a row XORed with a function of the row such that the 8 rows a 32-lane half of the transposing read
// touches fall into 8 different bank groups (see chunk_swz).
// Split K: the K steps of a unit are cut into n_split parts (one workgroup each); the last workgroup of a unit to
// finish sums the parts in part order (threadfence + counter, no spinning) and runs the epilogue.
#include <hip/hip_runtime.h>
#include "pack_taps_inl.h"
#include <type_traits>
#include "hmvae.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef HMVAE_STAMP
#define STAMP4(i)                                                                                  \
    do {                                                                                           \
        if (d.debug && threadIdx.x == 0) {                                                         \
            unsigned long long t_;                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            d.debug[(size_t)blockIdx.x * 8 + (i)] = t_;                                            \
        }                                                                                          \
    } while (0)
#else
#define STAMP4(i)
#endif

#include <cstdlib>

// Issue priority of the data path's waves over whatever runs beside them (the weight gradients on their side streams): the
// launches of the forward / data-gradient chain wait for one another, the side work does not.  HMVAE_CHAIN_PRIO=0..3 (default 3), read once.
// Measured on the graph step at B = 256, three interleaved rounds: 1.337 (0) -> 1.311 ms (1 or 3).
static int chain_prio() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("HMVAE_CHAIN_PRIO");
        v = e ? atoi(e) : 3;
        if (v < 0 || v > 3) v = 3;
    }
    return v;
}
#define CHAIN_SETPRIO(p)                                   \
    do {                                                   \
        if ((p) == 1) __builtin_amdgcn_s_setprio(1);       \
        else if ((p) == 2) __builtin_amdgcn_s_setprio(2);  \
        else if ((p) == 3) __builtin_amdgcn_s_setprio(3);  \
    } while (0)

namespace {

#ifndef V4_R14
#define V4_R14 6
#endif
constexpr int V4_MAX_ITEMS = 256;
constexpr int V4_MAX_NC32 = 32;

// x = p0 + p1 + p2 exactly, round-to-nearest-even at each level; returns the three packed pairs of (x0, x1)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h0, unsigned& h1, unsigned& h2) {
#ifdef V4_ABL_NOSPLIT            // timing ablation (wrong results): no arithmetic behind the planes
    h0 = __float_as_uint(x0); h1 = __float_as_uint(x1); h2 = h0 ^ h1;
    return;
#endif
    f32x2 v = {x0, x1};
    h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r = {x0 - __uint_as_float(h0 << 16), x1 - __uint_as_float(h0 & 0xffff0000u)};
    h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    f32x2 s = {r[0] - __uint_as_float(h1 << 16), r[1] - __uint_as_float(h1 & 0xffff0000u)};
    h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(s, bf16x2));
}

// chunk XOR of row k: nc = 128 (8 chunks per row, 256-byte pitch = one bank period): rows 8a + q and 8(a^1) + q, q < 4,
// must differ -> (k & 3) | bit 3 of k;  nc = 64 (4 chunks, 128-byte pitch: odd rows sit half a period further):
// (k >> 1) & 1 | bit 3 of k.
template <int NC>
__device__ __forceinline__ unsigned chunk_swz(unsigned k) {
    if (NC == 128) return (k & 3u) | (((k >> 3) & 1u) << 2);
    return ((k >> 1) & 1u) | (((k >> 3) & 1u) << 1);
}

// (Position blocks -- units of 2 / 4 output positions sharing the weight fragment of every step and a scheduled set of
// activation images -- were built in round 2, measured never faster than this plain form at 256 and 1 024 sequences per GPU
// on any layer side, with four- and eight-wave workgroups alike, and removed in round 3: DESIGN.md 4.6 / EXPERIMENTS.md.)
template <int MTW, int NT, int R>
__global__ __launch_bounds__(256, 2) void gather_gemm4_kernel(hmvae_gemm4_desc d, int prio) {
    CHAIN_SETPRIO(prio);
    constexpr int NC = NT * 16;                  // columns of the unit
    constexpr int PITCH = NC * 2;                // bytes per image row
    constexpr unsigned PLB = 32 * PITCH;         // bytes per plane
    constexpr unsigned BUFB = 3 * PLB;           // bytes per image
    constexpr int CQ = NC / 4;                   // 16-byte pieces per fp32 row
    constexpr int RPP = 256 / CQ;                // rows per load pass
    constexpr int NPASS = 32 / RPP;
    constexpr int MTG = 4 * MTW;                 // row tiles per group
    constexpr int RP = MTG * 16;
    constexpr int D = R - 1;                     // K steps in flight ahead of the one being multiplied
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // layout: [image 0][image 1][rows table: max_nc32*32 u32][items: max_items*2 int][flag]
    unsigned* lds_rows = (unsigned*)(lds + 2u * BUFB);
    int* lds_items = (int*)(lds_rows + V4_MAX_NC32 * 32);
    int* lds_flag = lds_items + 2 * V4_MAX_ITEMS;

    const int S = d.n_split;
    const int n_wg = d.n_units * S;
    const int per = (n_wg + 7) >> 3;
    const int wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || wg >= n_wg) return;
    const int unit = wg / S;
    const int sp = wg - unit * S;
    const int n_cb = d.bp / NC;
    const int cb = unit % n_cb;
    const int r0 = unit / n_cb;
    const int pb = r0 % d.n_pos;                 // output position
    const int g = r0 / d.n_pos;

    STAMP4(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nc32 = d.g_nc32[g];
    const int choff = d.g_choff[g];
    const int nitems = d.pb_len[pb];
    const int bp = d.bp;
    {
        const int* gi = d.items + 2 * d.pb_off[pb];
        for (int i = tid; i < 2 * nitems; i += 256) lds_items[i] = (i & 1) ? (gi[i] & 0xff) : gi[i] * d.src_ts * bp * 4;
        const int* gr = d.chan_rows + choff;
        for (int i = tid; i < 32 * nc32; i += 256) lds_rows[i] = (unsigned)gr[i] * (unsigned)bp * 4u;
    }
    const int nks = nc32 * nitems;
    const int k0 = (int)((long long)nks * sp / S);                  // this part's K steps
    const int k1 = (int)((long long)nks * (sp + 1) / S);
    const int ntiles = d.g_ntiles[g];
    const bool mine = wave * MTW < ntiles;                      // this wave owns at least one live row tile (wave-uniform)

    // B staging: this thread's pieces of a K step
    const int col4 = tid % CQ;
    const int kr = tid / CQ;
    unsigned st_off[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const unsigned k = (unsigned)(kr + RPP * i);
        st_off[i] = k * PITCH + ((((unsigned)col4 >> 2) ^ chunk_swz<NC>(k)) << 5) + (((unsigned)col4 & 3u) << 3);
    }
    const unsigned col_b = (unsigned)(cb * NC + col4 * 4) * 4u;
    const char* __restrict__ sbase = (const char*)d.src;
    // B fragments: lane (i, kk) of the transposing read supplies row 8kk + q (+ 4 for the second half), piece p
    const unsigned kk = (unsigned)lane >> 4, li = (unsigned)lane & 15u, q = li >> 2, p = li & 3u;
    const unsigned rd_base = ((8u * kk + q) * PITCH + (p << 3)) | (chunk_swz<NC>(8u * kk + q) << 5);
    // A fragments
    const char* __restrict__ wbase = (const char*)d.wp4 + (size_t)d.g_woff4[g] * 1024;
    const unsigned lane_a = (unsigned)(wave * MTW) * 2048u + (unsigned)lane * 16u;
    const unsigned kt = (unsigned)d.kt;

    f32x4 acc[MTW][NT];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    __syncthreads();                                             // tables visible
    STAMP4(1);

    // Iterator over K steps (c32, item), item fastest; clamped at the part's last step (the over-fetch is never used).
    // Operands run D K steps ahead of the MFMAs in a ring of R = D + 1 statically named register sets (the trip body is
    // fully unrolled): set u is split into LDS and multiplied in sub-step u while set (u + D) % R -- drained in the
    // previous sub-step -- is being refilled.  The loads of a K step take 1-2 us under load; its MFMAs 0.2-0.7 us.
    int c32 = k0 / (nitems > 0 ? nitems : 1), it = k0 - c32 * nitems, ks = k0;
    f32x4 rb[R][NPASS];
    f32x4 ra[R][MTW][2];                  // a row tile's A fragment: 8 consecutive channels of row m per lane, fp32
    auto advance = [&]() {
        const bool more = ks + 1 < k1;
        const int ni = it + 1;
        const bool wrap = ni == nitems;
        it = more ? (wrap ? 0 : ni) : it;
        c32 = more ? (wrap ? c32 + 1 : c32) : c32;
        ks = more ? ks + 1 : ks;
    };
    auto fetch = [&](f32x4 (&b)[NPASS], f32x4 (&a)[MTW][2], bool with_a) {
        const int2 item = *(const int2*)(lds_items + 2 * it);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const unsigned ro = lds_rows[c32 * 32 + kr + RPP * i];
            b[i] = *(const f32x4*)(sbase + (ro + (unsigned)item.x + col_b));
        }
        if (with_a) {
            const unsigned fo = ((unsigned)c32 * kt + (unsigned)item.y) * (unsigned)(MTG * 2048) + lane_a;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int h = 0; h < 2; ++h) a[mt][h] = *(const f32x4*)(wbase + (fo + (unsigned)(mt * 2048 + h * 1024)));
        }
        advance();
    };
    auto stash = [&](char* img, const f32x4 (&b)[NPASS]) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            unsigned h0a, h1a, h2a, h0b, h1b, h2b;
            split2(b[i][0], b[i][1], h0a, h1a, h2a);
            split2(b[i][2], b[i][3], h0b, h1b, h2b);
            *(u32x2*)(img + st_off[i]) = (u32x2){h0a, h0b};
            *(u32x2*)(img + PLB + st_off[i]) = (u32x2){h1a, h1b};
            *(u32x2*)(img + 2 * PLB + st_off[i]) = (u32x2){h2a, h2b};
        }
    };
    auto compute = [&](const char* img, const f32x4 (&af)[MTW][2]) {
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
        // the weights' three planes are made here, in registers (44 vector instructions per tile and K step, beside
        // 24-96 MFMAs): 4 instead of 6 bytes per weight through the load path, and nothing for the pack to split
        bf16x8 a[MTW][3];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            u32x4 w0, w1, w2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned p0, p1, p2, q0, q1, q2;
                split2(af[mt][h][0], af[mt][h][1], p0, p1, p2);
                split2(af[mt][h][2], af[mt][h][3], q0, q1, q2);
                w0[2 * h] = p0; w0[2 * h + 1] = q0;
                w1[2 * h] = p1; w1[2 * h + 1] = q1;
                w2[2 * h] = p2; w2[2 * h + 1] = q2;
            }
            a[mt][0] = __builtin_bit_cast(bf16x8, w0);
            a[mt][1] = __builtin_bit_cast(bf16x8, w1);
            a[mt][2] = __builtin_bit_cast(bf16x8, w2);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bf16x8 b[3];
            const unsigned ro = rd_base ^ ((unsigned)nt << 5);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(img + pl * PLB + ro));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(img + pl * PLB + 4 * PITCH + ro));
                const s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                b[pl] = __builtin_bit_cast(bf16x8, t);
            }
            // smallest cross terms first: (0,2) (2,0) (1,1) | (0,1) (1,0) | (0,0)
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
#ifdef V4_ABL_NOMFMA             // timing ablation (wrong results): the operands are consumed by one cheap instruction
                    acc[mt][nt][0] += __builtin_bit_cast(f32x4, a[mt][PA[t]])[0] + __builtin_bit_cast(f32x4, b[PB[t]])[0];
#else
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][PA[t]], b[PB[t]], acc[mt][nt], 0, 0, 0);
#endif
                }
        }
    };
    const int n = k1 - k0;
    int cnt = 0;
    auto run = [&](auto MINE) {
        constexpr bool M = decltype(MINE)::value;
#pragma unroll
        for (int u = 0; u < D; ++u) fetch(rb[u], ra[u], M);
        const int trips = n / R;
        STAMP4(2);
        for (int t = 0; t < trips; ++t) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                char* img = lds + ((cnt & 1) ? BUFB : 0u);
                stash(img, rb[u]);
#ifndef V4_ABL_NOBAR             // timing ablation (wrong results): no workgroup barrier per K step
                __syncthreads();
#endif
                fetch(rb[(u + D) % R], ra[(u + D) % R], M);
                if (M) compute(img, ra[u]);
                ++cnt;
            }
        }
        const int rem = n - trips * R;
#pragma unroll
        for (int u = 0; u < R - 1; ++u) {
            if (u < rem) {
                char* img = lds + ((cnt & 1) ? BUFB : 0u);
                stash(img, rb[u]);
                __syncthreads();
                if (M) compute(img, ra[u]);
                ++cnt;
            }
        }
    };
    if (n > 0) {
        if (mine) run(std::integral_constant<bool, true>{});
        else run(std::integral_constant<bool, false>{});
    }

    // ---- epilogue ----
    STAMP4(3);
    const int n_ = (int)li, rq = (int)kk;
    auto finish = [&](f32x4 s, int row_in_group, int colbase, int opos) {     // bias / activation / accumulate / 16-byte store
        const int drow = d.dst_rows[g * RP + row_in_group];
        if (drow < 0) return;
        if (d.bias) {
            const int bi = d.bias_idx[g * RP + row_in_group];
            float bv = bi >= 0 ? d.bias[bi] : 0.f;
            if (d.bias_idx2) {
                const int b2 = d.bias_idx2[g * RP + row_in_group];
                if (b2 >= 0) bv = 0.5f * (bv + d.bias[b2]);
            }
            s += (f32x4){bv, bv, bv, bv};
        }
        if (d.flags & HMVAE_FLAG_LEAKY) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = s[e] > 0.f ? s[e] : s[e] * d.slope;
        }
        float* dp = d.dst + (size_t)(drow + opos * d.dst_ts) * bp + cb * NC + colbase;
        if (d.flags & HMVAE_FLAG_ACCUM) s += *(const f32x4*)dp;
        if (d.flags & HMVAE_FLAG_LEAKY_GRAD) {
            const int arow = d.act_rows ? d.act_rows[g * RP + row_in_group] : drow;
            if (arow >= 0) {
                const f32x4 a = *(const f32x4*)(d.act + (size_t)(arow + opos * d.dst_ts) * bp + cb * NC + colbase);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = a[e] > 0.f ? s[e] : s[e] * d.slope;
            }
        }
        *(f32x4*)dp = s;
    };
    // The tile leaves the registers through LDS (the images are free now: every wave is past its last read after the
    // barrier) so that rows go out as 16-byte pieces: [row < RP][NC] floats.
    // Split K, hand-off between workgroups inside the launch (the per-XCD L2s are not coherent with each other and a CU's
    // L1 is never refreshed): every byte of a part is stored write-through (sc1, 16 bytes per lane), every storing wave
    // drains its stores, the workgroup meets at a barrier, ONE lane adds to the unit's counter (agent scope); the workgroup
    // whose add came last reads all parts with sc1 loads (they bypass its L1) in part order.  Nobody waits for anybody.
    float* tile = (float*)lds;
    const int live_rows = ntiles * 16;
    constexpr unsigned TILE_B = RP * NC * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(d.scratch + (size_t)unit * S * (RP * NC)), 0, S * (int)TILE_B, 0x00020000);
    const unsigned part_b = (unsigned)sp * TILE_B;
    __syncthreads();
    if (mine) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    tile[((wave * MTW + mt) * 16 + rq * 4 + r) * NC + nt * 16 + n_] = acc[mt][nt][r];
    }
    __syncthreads();
    for (int idx = tid; idx < live_rows * CQ; idx += 256) {
        const int row = idx / CQ, c4 = idx - row * CQ;
        const f32x4 v = *(const f32x4*)(tile + row * NC + c4 * 4);
        if (S == 1) finish(v, row, c4 * 4, pb);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                    part_b + (unsigned)(row * NC + c4 * 4) * 4u, 0, 16);
    }
    if (S == 1) {
        STAMP4(6);
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP4(4);
    if (tid == 0) *lds_flag = __hip_atomic_fetch_add(d.counters + unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    STAMP4(5);
    if (*lds_flag != S - 1) return;
    for (int idx = tid; idx < live_rows * CQ; idx += 256) {
        const int row = idx / CQ, c4 = idx - row * CQ;
        const unsigned off = (unsigned)(row * NC + c4 * 4) * 4u;
        f32x4 s = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
        for (int i = 1; i < S; ++i)
            s += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)i * TILE_B, 0, 16));
        finish(s, row, c4 * 4, pb);
    }
    if (tid == 0) __hip_atomic_store(d.counters + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left as found
    STAMP4(6);
}

// ---- weight pack: effective matrix -> fp32 fragments in A-operand order ----
// block = (group, row tile, 32 channels): 512 (row, channel) pairs, two neighbouring channels per thread -- thread (m,
// kk, jh) owns row m and channels 8kk + 2jh, +1 of the block, i.e. 8 bytes of lane (m, kk)'s 32-byte piece per virtual
// tap.  The (<= 16) real taps of both pairs stay in registers; virtual taps are produced 8 at a time into a 16 KB LDS
// image ([vt][channel half][lane][16 bytes] = the output order: a wave reads a fragment as two whole 1 KB rows) and leave as
// whole 2 KB fragments.
constexpr int P4_VC = 8;                       // virtual taps per LDS image


__global__ __launch_bounds__(256) void pack4_kernel(const hmvae_pack_op* __restrict__ ops, int n_ops) {
    extern __shared__ __attribute__((aligned(16))) char simg[];   // [P4_VC][64 lanes][8 floats], then C[kt][16] floats
    int oi;
    {
        const int lane = threadIdx.x & 63;
        const int st = lane < n_ops ? ops[lane].block_start : 0x7fffffff;
        const unsigned long long m = __ballot((int)blockIdx.x >= st);
        oi = __popcll(m) - 1;
    }
    const hmvae_pack_op o = ops[oi];
    const int local = blockIdx.x - o.block_start;
    const int mtg = o.mt, kt = o.kt;
    const int rows = mtg * 16;
    const int nc32_max = o.max_nc4 >> 3;
    const int c32 = local % nc32_max;
    const int r1 = local / nc32_max;
    const int tg = r1 % mtg;
    const int g = r1 / mtg;
    const int nc4 = o.g_nc4[g];
    if (c32 * 8 >= nc4) return;
    const int tid = threadIdx.x;
    const int m = tid & 15, kk = (tid >> 4) & 3, jh = tid >> 6;
    const int r = tg * 16 + m;
    const int c0 = c32 * 32 + kk * 8 + jh * 2;                     // this thread's two channels of the group's list
    const int rid = o.row_id[g * rows + r];
    int cid[2];
    unsigned bits[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        cid[e] = o.chan_id[o.g_choff[g] + c0 + e];
        bits[e] = (rid >= 0 && cid[e] >= 0) ? o.pair_ok[o.g_okoff[g] + r * (nc4 * 4) + c0 + e] : 0u;
    }
    f32x2* img = (f32x2*)simg;
    // 8-byte slot of this thread inside a 2 KB fragment: [half of the 8 channels][lane (kk, m)][4 floats]
    const unsigned my_pair = (unsigned)((jh >> 1) * 128 + (kk * 16 + m) * 2 + (jh & 1));
    char* outg = (char*)o.packed + (size_t)o.g_woff[g] * 1024;
    const bool folded = o.fold_taps != nullptr;
    const int kr = folded ? o.fold_kt : 0;
    float t[2][16];
    float* ct = (float*)(simg + P4_VC * 2048);                      // [kt][16]
    const float* psrc[2] = {o.dense, o.dense};
    if (folded) {
        const int nf = o.fold_nf;
        for (int i = tid; i < kt * 16; i += 256) ct[i] = 0.f;
        __syncthreads();
        for (int i = tid; i < kt * nf; i += 256) {
            const int k = o.fold_taps[i];
            if (k >= 0) ct[(i / nf) * 16 + k] = o.fold_coef[i];
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t[e][k] = 0.f;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {                        // dense (row, channel) pairs merged into this one
                if (!((bits[e] >> ci) & 1u)) continue;
                int dr = rid, dc = cid[e];
                if (o.child_tab) {
                    if (o.child_side) dr = o.child_tab[rid * 4 + ci];
                    else dc = o.child_tab[cid[e] * 4 + ci];
                }
                const float* src = o.dense + (size_t)dr * o.row_stride + (size_t)dc * o.chan_stride;
                if (HMVAE_PACK_WIDE && kr == 15 && o.tap_stride == 1) {
                    hmvae_add_taps15(src, t[e]);
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < kr) t[e][k] += src[(size_t)k * o.tap_stride];
                }
            }
            if (bits[e] && o.child_scale) {
                const float sc = o.child_scale[o.child_side ? rid : cid[e]];
#pragma unroll
                for (int k = 0; k < 16; ++k) t[e][k] *= sc;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e)
            psrc[e] = o.dense + (bits[e] ? (size_t)rid * o.row_stride + (size_t)cid[e] * o.chan_stride : 0);
    }
    // fragment (c32*kt + vt)*mtg + tg, 2 KB each: the image of nv virtual taps from v0 leaves as whole fragments
    auto copy_out = [&](int v0, int nv) {
        for (int i = tid; i < nv * 128; i += 256) {
            const int vl = i >> 7, w16 = i & 127;
            const size_t frag = (size_t)(c32 * kt + v0 + vl) * mtg + tg;
            *(u32x4*)(outg + frag * 2048 + (size_t)w16 * 16) = *(const u32x4*)(simg + (size_t)i * 16);
        }
    };
    if (HMVAE_PACK_WIDE && !folded && kt == 15 && o.tap_stride == 1) {               // plain 15-tap convolution: the taps in registers
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t[e][k] = 0.f;
            if (bits[e]) hmvae_add_taps15(psrc[e], t[e]);
        }
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (ch) __syncthreads();
#pragma unroll
            for (int vl = 0; vl < P4_VC; ++vl)
                if (ch * P4_VC + vl < 15) img[vl * 256 + my_pair] = f32x2{t[0][ch * P4_VC + vl], t[1][ch * P4_VC + vl]};
            __syncthreads();
            copy_out(ch * P4_VC, ch ? 15 - P4_VC : P4_VC);
        }
        return;
    }
    for (int v0 = 0; v0 < kt; v0 += P4_VC) {
        const int nv = kt - v0 < P4_VC ? kt - v0 : P4_VC;
        __syncthreads();                                           // C visible (first pass) / previous image copied out
        for (int vl = 0; vl < nv; ++vl) {
            const int vt = v0 + vl;
            f32x2 v;
            if (folded) {
                const f32x4* c = (const f32x4*)(ct + vt * 16);
                const f32x4 q0 = c[0], q1 = c[1], q2 = c[2], q3 = c[3];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        a += t[e][k] * q0[k] + t[e][4 + k] * q1[k] + t[e][8 + k] * q2[k] + t[e][12 + k] * q3[k];
                    v[e] = a;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) v[e] = bits[e] ? psrc[e][(size_t)vt * o.tap_stride] : 0.f;
            }
            img[vl * 256 + my_pair] = v;
        }
        __syncthreads();
        copy_out(v0, nv);
    }
}

template <int MTW, int NT>
int launch4(const hmvae_gemm4_desc* d, hipStream_t s) {
    // K steps in flight per wave: enough to cover ~2 us of load latency with the MFMA time of a step (96 cycles per tile
    // pair), within the registers the accumulators (MTW * NT tiles) leave
    constexpr int W = MTW * NT;
    constexpr int R = W <= 4 ? V4_R14 : (W <= 8 ? 4 : 3);
    constexpr int NC = NT * 16;
    size_t lds = (size_t)2 * 3 * 32 * NC * 2 + V4_MAX_NC32 * 32 * 4 + 2 * V4_MAX_ITEMS * 4 + 16;
    const size_t tile = (size_t)MTW * 64 * NC * 4;
    if (tile > lds) lds = tile;
    if (lds > 160 * 1024) return HMVAE_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gather_gemm4_kernel<MTW, NT, R>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return HMVAE_ERR_LAUNCH;
        attr_set = true;
    }
    const int n_wg = d->n_units * d->n_split;
    const int per = (n_wg + 7) / 8;
    hipLaunchKernelGGL((gather_gemm4_kernel<MTW, NT, R>), dim3((unsigned)(per * 8)), dim3(256), lds, s, *d, chain_prio());
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}

}  // namespace

extern "C" int hmvae_gather_gemm_v4(const hmvae_gemm4_desc* d, hmvae_stream_t stream) {
    if (!d || !d->src || !d->dst || !d->wp4 || !d->items || !d->g_nc32 || !d->g_woff4 || !d->g_choff || !d->g_ntiles ||
        !d->chan_rows || !d->dst_rows || !d->pb_off || !d->pb_len || (d->bias && !d->bias_idx))
        return HMVAE_ERR_ARG;
    if (d->n_groups <= 0 || d->n_pos <= 0 || d->kt <= 0 || d->bp <= 0 || (d->nc != 64 && d->nc != 128) || d->bp % d->nc)
        return HMVAE_ERR_ARG;
    if ((d->flags & HMVAE_FLAG_LEAKY_GRAD) && (!d->act || (d->flags & HMVAE_FLAG_LEAKY))) return HMVAE_ERR_ARG;
    if (d->mtw != 1 && d->mtw != 2) return HMVAE_ERR_UNSUPPORTED;
    if (d->n_split < 1 || (d->n_split > 1 && (!d->scratch || !d->counters))) return HMVAE_ERR_ARG;
    if (d->max_items <= 0 || d->max_items > V4_MAX_ITEMS || d->max_nc32 <= 0 || d->max_nc32 > V4_MAX_NC32)
        return HMVAE_ERR_UNSUPPORTED;
    if ((long long)d->n_groups * d->n_pos * (d->bp / d->nc) != d->n_units) return HMVAE_ERR_ARG;
    if (d->pp > 1) return HMVAE_ERR_UNSUPPORTED;         // position blocks: removed in round 3 (never faster), field reserved
    hipStream_t s = (hipStream_t)stream;
    switch (d->mtw * 1000 + d->nc) {
        case 1064: return launch4<1, 4>(d, s);
        case 1128: return launch4<1, 8>(d, s);
        case 2064: return launch4<2, 4>(d, s);
        case 2128: return launch4<2, 8>(d, s);
        default: return HMVAE_ERR_UNSUPPORTED;
    }
}

extern "C" int hmvae_pack_weights_v4(const hmvae_pack_op* ops_dev, int32_t n_ops, int32_t total_blocks, int32_t max_kt,
                                     hmvae_stream_t stream) {
    if (!ops_dev || n_ops <= 0 || n_ops > 64 || total_blocks <= 0 || max_kt <= 0) return HMVAE_ERR_ARG;
    const size_t lds = (size_t)P4_VC * 2048 + (size_t)max_kt * 64;
    if (lds > 64 * 1024) return HMVAE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pack4_kernel, dim3((unsigned)total_blocks), dim3(256), lds, (hipStream_t)stream, ops_dev, n_ops);
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}
