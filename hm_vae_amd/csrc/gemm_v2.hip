// gemm_v2.hip — gather-GEMM, second generation (the one the engine launches).
//
// Same contract as hmvae_gather_gemm (SkeletonConv forward / backward-data, per-edge Linear; reference
// skeleton.py:95-105, seq_two_hier_sa_vae.py:162,267).  What the rocprof/PMC passes on MI355X taught (profiles/):
//   * v_mfma_f32_16x16x4_f32 runs on the SIMD's fp32 datapath: every other vector instruction a wave issues is
//     time taken from the matrix pipe (an ablation without MFMAs ran in ~half the time, the two halves add up).
//     So the loop is on an instruction diet: all table lookups come from LDS copies holding precomputed BYTE
//     offsets, operand addresses are one 32-bit add on top of a uniform base (saddr loads), the iterator is scalar.
//   * a wave owns MT*16 rows x CW*64 batch columns of one output position: one weight fragment feeds CW*4 MFMAs and
//     one 16-byte activation load per 64 columns feeds MT*4 of them.
//   * loads run DEPTH items ahead of the MFMAs in a ring of statically named register sets (a register that is the
//     destination of an in-flight load must never be copied; a branch in the steady-state loop makes the compiler
//     fall back to vmcnt(0)); accumulators are folded into a second level every 96 items, outside the hot loop.
//   * work units are dealt to the 8 XCDs in group-major contiguous ranges (blockIdx % 8 picks the range), so one
//     XCD's L2 only sees the packed weights of its own few row groups.
// K is split over the 4 waves of a workgroup (summed through LDS) unless a group is too short for that.
#include <hip/hip_runtime.h>
#include "hmvae.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MAX_ITEMS = 256;      // items per position block (host checks)
constexpr int MAX_C4 = 512;         // 4-channel steps per group (host checks)
constexpr int TABLE_INTS = 2 * MAX_ITEMS + 4 * MAX_C4;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#ifdef HMVAE_STAMP
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        if (d.debug && lane == 0) {                                                                \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            d.debug[((size_t)blockIdx.x * nwaves + wid) * 4 + (i)] = t_;                           \
        }                                                                                          \
    } while (0)
#else
#define STAMP(i)
#endif

template <int MT, int CW>
struct VStage {          // operands of one item, in flight between its loads and its MFMAs
    f32x4 b[CW];         // CW units of 64 batch columns
    float a[MT];
};
struct SStage {          // the item's table entries, read from the LDS copies of the tables (byte offsets)
    unsigned boff;       // activation: (row base of this lane's channel + q*src_ts) * bp * 4
    unsigned aoff;       // weights:    (c4*kt + tap) * MT*64*4
    int taps;
};

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

template <int MT, int CW, int DEPTH>
__global__ __launch_bounds__(512) void gather_gemm2_kernel(hmvae_gemm2_desc d, int prio) {
    CHAIN_SETPRIO(prio);
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int nwaves = blockDim.x >> 6;
    const int wpos = d.wpos;                         // 1: waves split K;  4: waves take 4 neighbouring position blocks
    const int nw = wpos > 1 ? 1 : nwaves;            // K-split factor
    const int per = (d.n_units + 7) >> 3;
    const int u = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || u >= d.n_units) return;
    const int n_ct = d.bp / (64 * CW);
    const int n_pq = (d.n_pblk + wpos - 1) / wpos;
    int ct, pq, g;
    if (d.xcd_mode == 0) {
        // group-major ranges: an XCD's L2 only sees the packed weights of its own few groups (weight-heavy layers)
        ct = u % n_ct;
        const int r0 = u / n_ct;
        pq = r0 % n_pq;
        g = r0 / n_pq;
    } else {
        // position-major ranges, group fastest: every XCD gets the same mix of (unequally long) groups, and only a
        // slice of the positions, i.e. of the activation rows (layers whose weights fit in every L2)
        g = u % d.n_groups;
        const int r0 = u / d.n_groups;
        ct = r0 % n_ct;
        pq = r0 / n_ct;
    }
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // scalar: keeps the K loop uniform
    const int wave = wpos > 1 ? 0 : wid;             // index inside the K split
    const int pb = wpos > 1 ? pq * wpos + wid : pq;  // this wave's position block
    const bool live = pb < d.n_pblk;
    const int kk = lane >> 4;
    const int n = lane & 15;

    STAMP(0);
    const int nc4 = d.g_nc4[g];
    const int choff = d.g_choff[g];
    const int nitems = live ? d.pb_len[pb] : 0;
    // tables -> LDS (uniform/broadcast ds_reads in the loop instead of scalar-memory loads with their waits)
    int* lds_rows = (int*)red;                                                   // [nc4*4], shared by the workgroup
    int* lds_items = lds_rows + 4 * MAX_C4 + (wpos > 1 ? wid * 2 * MAX_ITEMS : 0);   // [nitems][2] (per wave if wpos>1)
    {
        // items become {q*src_ts*bp*4, taps}; rows become row*bp*4: the loop then needs one add per operand address
        if (wpos > 1) {
            if (live) {
                const int* gi = d.items + 2 * d.pb_off[pb];
                for (int i = lane; i < 2 * nitems; i += 64) lds_items[i] = (i & 1) ? gi[i] : gi[i] * d.src_ts * d.bp * 4;
            }
        } else {
            const int* gi = d.items + 2 * d.pb_off[pb];
            for (int i = threadIdx.x; i < 2 * nitems; i += blockDim.x)
                lds_items[i] = (i & 1) ? gi[i] : gi[i] * d.src_ts * d.bp * 4;
        }
        const int* gr = d.chan_rows + choff;
        for (int i = threadIdx.x; i < 4 * nc4; i += blockDim.x) lds_rows[i] = gr[i] * d.bp * 4;
    }
    __syncthreads();
    const char* __restrict__ wbase = (const char*)(d.wp + d.g_woff[g]);        // uniform
    const char* __restrict__ sbase = (const char*)d.src;                          // uniform
    const unsigned lane_a = (unsigned)lane * 4u;                                  // this lane's byte offset in an A fragment
    const unsigned lane_b = (unsigned)(ct * (64 * CW) + (n << 2)) * 4u;           // ... in the wave's batch tile
    const int bp = d.bp;
    const int kt = d.kt;

    f32x4 acc[CW][MT][4], tot[CW][MT][4];
#pragma unroll
    for (int pl = 0; pl < CW; ++pl)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[pl][mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                tot[pl][mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }

    const int n_mine = wave < nc4 ? (nc4 - wave + nw - 1) / nw : 0;
    const int total = n_mine * nitems;
    STAMP(1);
    if (total > 0) {
        // Software pipeline, three stages per item:  S (scalar table reads) -> V (vector operand loads) -> M (MFMAs).
        // S runs 3 items ahead of M, V runs 2 items ahead.  The steady-state loop body is branch-free straight-line
        // code over three statically named register sets, so the compiler's s_waitcnt insertion sees a regular
        // stream and emits counted vmcnt waits (a conditional or a register copy of an in-flight load makes it
        // fall back to vmcnt(0), which serialises everything).  The iterator is clamped at the last item, so the
        // pipeline's over-fetch at the tail re-reads valid addresses and is simply never consumed.
        int lc4 = wave, lit = 0;
        auto s_fetch = [&](SStage& s) {
            const int2 it = *(const int2*)(lds_items + 2 * lit);
            s.taps = it.y;
            s.boff = (unsigned)lds_rows[lc4 * 4 + kk] + (unsigned)it.x + lane_b;
            s.aoff = (unsigned)(lc4 * kt + (it.y & 0xff)) * (unsigned)(MT * 256) + lane_a;
            const int nl = lit + 1;
            const bool wrap = nl == nitems;
            const int nc = wrap ? lc4 + nw : lc4;
            const bool ok = nc < nc4;
            lit = ok ? (wrap ? 0 : nl) : lit;
            lc4 = ok ? nc : lc4;
        };
        auto v_issue = [&](const SStage& s, VStage<MT, CW>& v) {
#pragma unroll
            for (int cu = 0; cu < CW; ++cu) v.b[cu] = *(const f32x4*)(sbase + s.boff + cu * 256);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) v.a[mt] = *(const float*)(wbase + s.aoff + mt * 256);
        };
        auto mfma = [&](const VStage<MT, CW>& v, int) {
#pragma unroll
            for (int cu = 0; cu < CW; ++cu)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[cu][mt][j] = MFMA16(v.a[mt], v.b[cu][j], acc[cu][mt][j]);
        };
        auto flush = [&]() {     // fold the short fp32 chain into the running total (two-level accumulation)
#pragma unroll
            for (int pl = 0; pl < CW; ++pl)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        tot[pl][mt][j] += acc[pl][mt][j];
                        acc[pl][mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
        };
        // Ring of DEPTH+1 operand sets with STATIC roles (the trip body is fully unrolled): set u is consumed by the
        // MFMAs of sub-step u while set (u+DEPTH)%(DEPTH+1) is being filled for the item DEPTH ahead.  The table
        // entries of a whole trip are read from LDS one trip ahead into their own registers.
        constexpr int R = DEPTH + 1;
        VStage<MT, CW> v[R];
        SStage c[R], nx[R];
        int tp[R];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            s_fetch(c[u]);
            v_issue(c[u], v[u]);
            tp[u] = c[u].taps;
        }
#pragma unroll
        for (int u = 0; u < R; ++u) s_fetch(nx[u]);
        const int trips = total / R;
        constexpr int FLUSH_ITEMS = 96;          // a chain of <= 384 fp32 fmas per accumulator between folds
        constexpr int FLUSH_TRIPS = (FLUSH_ITEMS + R - 1) / R;
        for (int tb = 0; tb < trips; tb += FLUSH_TRIPS) {
            const int te = tb + FLUSH_TRIPS < trips ? tb + FLUSH_TRIPS : trips;
            for (int tr = tb; tr < te; ++tr) {
#pragma unroll
                for (int u = 0; u < R; ++u) c[u] = nx[u];
#pragma unroll
                for (int u = 0; u < R; ++u) s_fetch(nx[u]);
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const int w = (u + DEPTH) % R;      // == u-1 (mod R): the set drained by the previous sub-step
                    v_issue(c[u], v[w]);
                    tp[w] = c[u].taps;
                    mfma(v[u], tp[u]);
                }
            }
            flush();
        }
        // tail: the next (total % R) items are already in flight in sets 0..
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            if (trips * R + u < total) mfma(v[u], tp[u]);
        flush();
    }

    STAMP(2);
    const int p = pb;
    if (!live) {
        if (nw > 1) {          // keep the barriers of the K-split epilogue matched
            for (int cu = 0; cu < CW; ++cu) { if (cu > 0) __syncthreads(); __syncthreads(); }
        }
        return;
    }
    const int col0 = ct * (64 * CW);
    if (nw == 1) {
        // no K split: straight from registers, 16 bytes per lane
#pragma unroll
        for (int cu = 0; cu < CW; ++cu)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mt * 16 + kk * 4 + r;
                    const int drow = d.dst_rows[g * (MT * 16) + row];
                    if (drow < 0) continue;
                    f32x4 s = {tot[cu][mt][0][r], tot[cu][mt][1][r], tot[cu][mt][2][r], tot[cu][mt][3][r]};
                    if (d.bias) {
                        const int bi = d.bias_idx[g * (MT * 16) + row];
                        float bv = bi >= 0 ? d.bias[bi] : 0.f;
                        if (d.bias_idx2) {               // row = mean of two dense rows (SkeletonPool folded in)
                            const int b2 = d.bias_idx2[g * (MT * 16) + row];
                            if (b2 >= 0) bv = 0.5f * (bv + d.bias[b2]);
                        }
                        s += (f32x4){bv, bv, bv, bv};
                    }
                    if (d.flags & HMVAE_FLAG_LEAKY) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) s[e] = s[e] > 0.f ? s[e] : s[e] * d.slope;
                    }
                    float* dp = d.dst + (size_t)(drow + p * d.dst_ts) * bp + col0 + cu * 64 + (n << 2);
                    if (d.flags & HMVAE_FLAG_ACCUM) s += *(const f32x4*)dp;
                    if (d.flags & HMVAE_FLAG_LEAKY_GRAD) {
                        const int arow = d.act_rows ? d.act_rows[g * (MT * 16) + row] : drow;
                        if (arow >= 0) {
                            const f32x4 a = *(const f32x4*)(d.act + (size_t)(arow + p * d.dst_ts) * bp + col0 + cu * 64 + (n << 2));
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[e] = a[e] > 0.f ? s[e] : s[e] * d.slope;
                        }
                    }
                    *(f32x4*)dp = s;
                }
        STAMP(3);
        return;
    }
    // K split over the waves: sum through LDS, one 64-column unit at a time
    float* redf = red + TABLE_INTS + 3 * 2 * MAX_ITEMS;
    float* mine = redf + wave * (MT * 16 * 64);
    const int rl = threadIdx.x >> 4;
    const int cq = threadIdx.x & 15;
#pragma unroll
    for (int cu = 0; cu < CW; ++cu) {
        if (cu > 0) __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 v = {tot[cu][mt][0][r], tot[cu][mt][1][r], tot[cu][mt][2][r], tot[cu][mt][3][r]};
                *(f32x4*)(mine + (mt * 16 + kk * 4 + r) * 64 + (n << 2)) = v;
            }
        __syncthreads();
        for (int row = rl; row < MT * 16; row += (blockDim.x >> 4)) {
            const int drow = d.dst_rows[g * (MT * 16) + row];
            if (drow < 0) continue;
            f32x4 s = *(const f32x4*)(redf + row * 64 + (cq << 2));
            for (int w = 1; w < nw; ++w) s += *(const f32x4*)(redf + w * (MT * 16 * 64) + row * 64 + (cq << 2));
            if (d.bias) {
                const int bi = d.bias_idx[g * (MT * 16) + row];
                float bv = bi >= 0 ? d.bias[bi] : 0.f;
                if (d.bias_idx2) {
                    const int b2 = d.bias_idx2[g * (MT * 16) + row];
                    if (b2 >= 0) bv = 0.5f * (bv + d.bias[b2]);
                }
                s += (f32x4){bv, bv, bv, bv};
            }
            if (d.flags & HMVAE_FLAG_LEAKY) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = s[e] > 0.f ? s[e] : s[e] * d.slope;
            }
            float* dp = d.dst + (size_t)(drow + p * d.dst_ts) * bp + col0 + cu * 64 + (cq << 2);
            if (d.flags & HMVAE_FLAG_ACCUM) s += *(const f32x4*)dp;
            if (d.flags & HMVAE_FLAG_LEAKY_GRAD) {
                const int arow = d.act_rows ? d.act_rows[g * (MT * 16) + row] : drow;
                if (arow >= 0) {
                    const f32x4 a = *(const f32x4*)(d.act + (size_t)(arow + p * d.dst_ts) * bp + col0 + cu * 64 + (cq << 2));
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] = a[e] > 0.f ? s[e] : s[e] * d.slope;
                }
            }
            *(f32x4*)dp = s;
        }
    }
    STAMP(3);
}

template <int MT, int CW>
static int launch2(const hmvae_gemm2_desc* d, hipStream_t s) {
    const int per = (d->n_units + 7) / 8;
    dim3 grid((unsigned)(per * 8)), block(64 * (d->wpos > 1 ? d->wpos : d->ksplit));
    const size_t lds = (TABLE_INTS + 3 * 2 * MAX_ITEMS) * sizeof(int) +
                       (d->ksplit > 1 ? (size_t)d->ksplit * MT * 16 * 64 * sizeof(float) : 0);
    // prefetch distance in items: short MFMA groups need more loads in flight per wave
#ifndef G2_DEPTH_BONUS
#define G2_DEPTH_BONUS 0
#endif
    constexpr int DEPTH = ((MT * CW == 1) ? 4 : (MT * CW == 2 ? 3 : 2)) + G2_DEPTH_BONUS;
    hipLaunchKernelGGL((gather_gemm2_kernel<MT, CW, DEPTH>), grid, block, lds, s, *d, chain_prio());
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}

extern "C" int hmvae_gather_gemm_v2(const hmvae_gemm2_desc* desc, hmvae_stream_t stream) {
    if (!desc || !desc->src || !desc->dst || !desc->wp || !desc->items || (desc->bias && !desc->bias_idx)) return HMVAE_ERR_ARG;
    if ((desc->flags & HMVAE_FLAG_LEAKY_GRAD) && (!desc->act || (desc->flags & HMVAE_FLAG_LEAKY))) return HMVAE_ERR_ARG;
    if (desc->bp <= 0 || (desc->bp & 63) || desc->n_groups <= 0 || desc->n_pos <= 0 || desc->kt <= 0 || desc->kt > 254)
        return HMVAE_ERR_ARG;
    if (desc->ksplit != 1 && desc->ksplit != 2 && desc->ksplit != 4 && desc->ksplit != 8) return HMVAE_ERR_ARG;
    if (desc->max_items <= 0 || desc->max_items > MAX_ITEMS || desc->max_nc4 <= 0 || desc->max_nc4 > MAX_C4) return HMVAE_ERR_UNSUPPORTED;
    const int cw = desc->np;            // 64-column units per wave (1, 2 or 4)
    if (cw != 1 && cw != 2 && cw != 4) return HMVAE_ERR_ARG;
    if (desc->bp % (64 * cw)) return HMVAE_ERR_ARG;
    if (desc->n_pblk != desc->n_pos) return HMVAE_ERR_ARG;
    if (desc->wpos != 1 && desc->wpos != 4) return HMVAE_ERR_ARG;
    if (desc->wpos > 1 && desc->ksplit != 1) return HMVAE_ERR_ARG;
    if ((long long)desc->n_groups * ((desc->n_pblk + desc->wpos - 1) / desc->wpos) * (desc->bp / (64 * cw)) != desc->n_units)
        return HMVAE_ERR_ARG;
    // 32-bit byte offsets inside the kernel
    hipStream_t s = (hipStream_t)stream;
    const int key = desc->mt * 10 + cw;
    switch (key) {
        case 11: return launch2<1, 1>(desc, s);
        case 12: return launch2<1, 2>(desc, s);
        case 14: return launch2<1, 4>(desc, s);
        case 21: return launch2<2, 1>(desc, s);
        case 22: return launch2<2, 2>(desc, s);
        case 31: return launch2<3, 1>(desc, s);
        case 41: return launch2<4, 1>(desc, s);
        default: return HMVAE_ERR_UNSUPPORTED;
    }
}
