// gemm_v3.hip — window gather-GEMM for the temporal convolutions (third generation), persistent and K-streamed.
//
// Same contract as hmvae_gather_gemm_v2 for SkeletonConv forward / backward-data (reference skeleton.py:95-105).
// What the v2 profiles and the ablations of the first v3 draft showed (round-1 ablation builds, -DHMVAE_ABLATE):
//   * v2 spends ~6 non-matrix instructions per MFMA on table look-ups and address arithmetic for loads that go straight
//     from L2 to registers, one load per (position, tap) item, nothing shared between the 8 waves of a workgroup;
//   * with those gone, what is left is LOAD IMBALANCE: the row groups of a skeleton layer have K lengths between 18 and
//     60 four-channel steps, a layer is only 100..800 workgroup-sized units, and a unit is 0.3..0.6 of a CU's fair
//     share, so a pure-MFMA skeleton of the kernel already took 2x the ideal time.
// So here
//   * a UNIT is one 16-row group x a CHUNK of neighbouring output positions x 64 batch columns;  per 4-channel K step
//     the union of the source positions the chunk reads (a contiguous WINDOW of rows) and the step's kt weight
//     fragments are staged in LDS with 16-byte loads, double buffered, one barrier per step;
//   * a wave owns one output position of the chunk (or one of `ts` slices of its taps); its items are (window row,
//     tap) pairs, wave-uniform and the same for every K step, held in SGPRs; the steady state per item is two adds,
//     one ds_read_b128 (activations, 4 MFMA column tiles) + one ds_read_b32 (weights), four MFMAs, in fully unrolled
//     chunks of UN items whose reads are issued one chunk ahead of their MFMAs;
//   * the launch is PERSISTENT: exactly (CUs x resident workgroups) workers, each walking a host-made list of
//     SEGMENTS = (unit, K-step range).  The host cuts the layer's total K steps into equal shares (planner.py,
//     "stream-K"), so every worker issues the same number of MFMAs whatever the group sizes are.  A unit cut between
//     workers leaves raw partial tiles in a scratch buffer; the fix-up kernel sums them in K order and applies the
//     epilogue (deterministic, no atomics, no spinning).
// Reflection padding and the transposed (backward-data) relation are just different item lists (planner.py).
// Accumulation is two-level like v2 (short fp32 chains folded into a running total).
#include <hip/hip_runtime.h>
#include "hmvae.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {
constexpr int V3_MAX_C4 = 512;       // 4-channel steps per group (host checks)
constexpr int V3_MAX_ITEMS = 16;     // items per wave, before padding to a multiple of UN (host checks)
constexpr int V3_WAVES = 8;

template <int UN>
struct Opnd {            // operands of one chunk of items
    f32x4 b[UN];
    float a[UN];
};

// bias + LeakyReLU + accumulate + store of one 16-byte piece of an output row (shared with the fix-up kernel)
__device__ __forceinline__ void emit(const hmvae_conv3_desc& d, int g, int row, int out_pos, int col, f32x4 s) {
    const int drow = d.dst_rows[g * 16 + row];
    if (drow < 0) return;
    if (d.bias) {
        const int bi = d.bias_idx[g * 16 + row];
        float bv = bi >= 0 ? d.bias[bi] : 0.f;
        if (d.bias_idx2) {                   // row = mean of two dense rows (SkeletonPool folded in)
            const int b2 = d.bias_idx2[g * 16 + row];
            if (b2 >= 0) bv = 0.5f * (bv + d.bias[b2]);
        }
        s += (f32x4){bv, bv, bv, bv};
    }
    if (d.flags & HMVAE_FLAG_LEAKY) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = s[e] > 0.f ? s[e] : s[e] * d.slope;
    }
    float* dp = d.dst + (size_t)(drow + out_pos * d.dst_ts) * d.bp + col;
    if (d.flags & HMVAE_FLAG_ACCUM) s += *(const f32x4*)dp;
    *(f32x4*)dp = s;
}

template <int NS, int UN, int NA>
__global__ __launch_bounds__(512, 4) void conv_window_kernel(hmvae_conv3_desc d) {
    constexpr int MAXCH = (V3_MAX_ITEMS + UN - 1) / UN;
    constexpr int MAXI = MAXCH * UN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // worker -> range of the schedule; workers of one XCD (blockIdx % 8) get neighbouring ranges (shared L2)
    const int per = d.n_workers >> 3;
    const int worker = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 4;
    const int n = lane & 15;
    const int W = d.win;
    const int kt = d.kt;
    const int bp = d.bp;
    const int ts = d.ts;

    // LDS: [rows table 4*MAX_C4 ints][buf0][buf1];  buf = B window + A fragments (+ one all-zero fragment, tap == kt,
    // that the padding items of a chunk point at)
    int* lds_rows = (int*)smem;
    const int b_bytes = 4 * W * 256;                 // [kk][window row][64 floats]
    const int buf_bytes = b_bytes + (kt + 1) * 256;  // + [tap][64 floats]
    char* buf0 = (char*)(lds_rows + 4 * V3_MAX_C4);
    float* redf = (float*)buf0;                      // epilogue: [wave][16 rows][64], aliases the staging buffers
    const int lane_b = kk * W * 256 + n * 16;
    const int lane_a = b_bytes + lane * 4;
    const char* __restrict__ sbase = (const char*)d.src;

    const int seg0 = d.worker_seg[2 * worker], seg1 = seg0 + d.worker_seg[2 * worker + 1];     // before any store
    for (int sg = seg0; sg < seg1; ++sg) {
        // One vector load per table (lane i fetches entry i) + readlane: the kernel stores to global memory inside this
        // loop, so the compiler may not use scalar loads for tables it cannot prove read-only, and per-lane copies of
        // uniform values would cost VGPRs and vector address arithmetic in the K loop.
        const int sv = d.segs[8 * sg + (lane & 7)];   // {g, pc, ct, k0, k1, part or -1, channel-list offset, weight offset}
        const int g = __builtin_amdgcn_readlane(sv, 0), pc = __builtin_amdgcn_readlane(sv, 1);
        const int ct = __builtin_amdgcn_readlane(sv, 2), k0 = __builtin_amdgcn_readlane(sv, 3);
        const int k1 = __builtin_amdgcn_readlane(sv, 4), part = __builtin_amdgcn_readlane(sv, 5);
        const int choff = __builtin_amdgcn_readlane(sv, 6), woff = __builtin_amdgcn_readlane(sv, 7);
        const int wv = d.wave_tab[(pc * V3_WAVES + wid) * 4 + (lane & 3)];   // {out position, pattern, n items, 0}
        const int pat = __builtin_amdgcn_readlane(wv, 1);
        const int n_it = __builtin_amdgcn_readlane(wv, 2);                   // a multiple of UN (host pads)
        // the wave's items, uniform, one SGPR each: row*256 | tap*256 << 16  (32 table entries per wave)
        unsigned iw[MAXI];
        {
            const int iv = d.items[(pc * V3_WAVES + wid) * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MAXI; ++i) iw[i] = (unsigned)__builtin_amdgcn_readlane(iv, i) << 8;
            const int* gr = d.chan_rows + choff + 4 * k0;
            for (int i = tid; i < 4 * (k1 - k0); i += 512) lds_rows[i] = gr[i] * bp * 4;
            if (tid < 128) *(float*)(buf0 + (tid >> 6) * buf_bytes + b_bytes + kt * 256 + lane * 4) = 0.f;
        }
        // loader slots: 16 bytes each; slot s -> (kk_s, window row j, 16-column quad); 64*W slots for B, kt*16 for A
        // (slots past the end of the window wrap around: a second copy of the same bytes to the same place, no branch)
        unsigned goff[NS], lrow[NS], loff[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            int s = i * 512 + tid;
            s = s < 64 * W ? s : s - 64 * W;
            s = s < 64 * W ? s : s - 64 * W;          // NS*512 < 3*64*W for every window the host allows (W >= 4)
            const int ks = s / (16 * W);
            const int rem = s - ks * 16 * W;
            const int j = rem >> 4;
            lrow[i] = (unsigned)ks * 4u;
            loff[i] = (unsigned)s * 16u;
            goff[i] = (unsigned)(d.win_rows[pc * W + j] * d.src_ts) * (unsigned)bp * 4u +
                      (unsigned)(ct * 64 + (rem & 15) * 4) * 4u;
        }
        // weight fragments of a 4-channel step: kt * 16 slots of 16 bytes; NA slots per thread (1 up to 32 taps, 2 up to 64:
        // the folded decoder layers have 61 virtual taps); slots past the end wrap (a second copy of the same bytes)
        unsigned aslot[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int sl = tid + i * 512;
            aslot[i] = (unsigned)(sl < kt * 16 ? sl : sl % (kt * 16)) * 16u;
        }
        const char* __restrict__ wbase = (const char*)(d.wp + woff);
        __syncthreads();

        f32x4 acc[4], tot[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            tot[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        f32x4 rb[NS], ra[NA];
        auto gload = [&](int c4) {                   // c4 absolute; the rows table holds steps k0..k1
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const unsigned ro = *(const unsigned*)((const char*)lds_rows + (unsigned)(c4 - k0) * 16u + lrow[i]);
                rb[i] = *(const f32x4*)(sbase + (ro + goff[i]));
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = *(const f32x4*)(wbase + ((unsigned)(c4 * kt) * 256u + aslot[i]));
        };
        auto lstore = [&](char* buf) {
#pragma unroll
            for (int i = 0; i < NS; ++i) *(f32x4*)(buf + loff[i]) = rb[i];
#pragma unroll
            for (int i = 0; i < NA; ++i) *(f32x4*)(buf + b_bytes + aslot[i]) = ra[i];
        };
        auto compute = [&](const char* buf) {
            Opnd<UN> r[2];
            auto reads = [&](Opnd<UN>& o, int ch) {
#pragma unroll
                for (int k = 0; k < UN; ++k) {
                    o.b[k] = *(const f32x4*)(buf + (iw[ch * UN + k] & 0xffffu) + lane_b);
                    o.a[k] = *(const float*)(buf + (iw[ch * UN + k] >> 16) + lane_a);
                }
            };
            auto mfmas = [&](const Opnd<UN>& o) {
#pragma unroll
                for (int k = 0; k < UN; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = MFMA16(o.a[k], o.b[k][j], acc[j]);
            };
            if constexpr (UN == 5) {
                // the regular waves of a 15-tap convolution: rows r0..r0+14 with taps 0..14 (pattern 1, forward) or
                // 14..0 (pattern 2, interior of a transposed stride-1 convolution): every LDS offset is an immediate
                // on top of two per-step base addresses, no address arithmetic between the MFMAs.  The reads run 3
                // items ahead of their MFMAs in a ring of 4 register sets; they and their waits are inline asm (with
                // the loaded registers as operands of the wait, so no MFMA can move above it) because the compiler's
                // scheduler otherwise sinks every read next to its use and waits lgkmcnt(0) per item.
                if (pat != 0) {
                    const unsigned pb = (unsigned)(size_t)(buf - smem) + (iw[0] & 0xffffu) + lane_b;
                    const unsigned pa = (unsigned)(size_t)(buf - smem) + lane_a;
                    f32x4 sb[4];
                    float sa[4];
#define V3_ISSUE(I, AOFF)                                                                                   \
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b32 %1, %4 offset:%5"                            \
                 : "=&v"(sb[(I) & 3]), "=&v"(sa[(I) & 3])                                                    \
                 : "v"(pb), "n"((I) * 256), "v"(pa), "n"((AOFF) * 256))
#define V3_WAIT(I, N) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(sb[(I) & 3]), "+v"(sa[(I) & 3]) : "n"(N))
#define V3_MFMA(I)                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j] = MFMA16(sa[(I) & 3], sb[(I) & 3][j], acc[j])
#define V3_STEP(I, AOFF3)                                                                                   \
    if ((I) + 3 < 15) V3_ISSUE((I) + 3, AOFF3);                                                             \
    V3_WAIT(I, 2 * (((I) + 3 < 15 ? (I) + 3 : 14) - (I)));                                                   \
    V3_MFMA(I)
                    if (pat == 1) {
                        V3_ISSUE(0, 0); V3_ISSUE(1, 1); V3_ISSUE(2, 2);
                        V3_STEP(0, 3); V3_STEP(1, 4); V3_STEP(2, 5); V3_STEP(3, 6); V3_STEP(4, 7);
                        V3_STEP(5, 8); V3_STEP(6, 9); V3_STEP(7, 10); V3_STEP(8, 11); V3_STEP(9, 12);
                        V3_STEP(10, 13); V3_STEP(11, 14); V3_STEP(12, 0); V3_STEP(13, 0); V3_STEP(14, 0);
                    } else {
                        V3_ISSUE(0, 14); V3_ISSUE(1, 13); V3_ISSUE(2, 12);
                        V3_STEP(0, 11); V3_STEP(1, 10); V3_STEP(2, 9); V3_STEP(3, 8); V3_STEP(4, 7);
                        V3_STEP(5, 6); V3_STEP(6, 5); V3_STEP(7, 4); V3_STEP(8, 3); V3_STEP(9, 2);
                        V3_STEP(10, 1); V3_STEP(11, 0); V3_STEP(12, 0); V3_STEP(13, 0); V3_STEP(14, 0);
                    }
#undef V3_ISSUE
#undef V3_WAIT
#undef V3_MFMA
#undef V3_STEP
                    return;
                }
            }
            if (n_it > 0) reads(r[0], 0);
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) {
                if (ch * UN < n_it) {
                    if (ch + 1 < MAXCH && (ch + 1) * UN < n_it) reads(r[(ch + 1) & 1], ch + 1);
                    mfmas(r[ch & 1]);
                }
            }
        };

        // fold the short chains every `fold` steps: <= 96 items -> <= 384 fmas per accumulator, as in v2
        const int fold = n_it > 0 ? (96 / n_it > 0 ? 96 / n_it : 1) : 1;
        gload(k0);
        lstore(buf0);
        __syncthreads();
        int since = 0;
        for (int c4 = k0; c4 < k1; ++c4) {
            char* cur = buf0 + ((c4 - k0) & 1) * buf_bytes;
            char* nxt = buf0 + ((c4 - k0 + 1) & 1) * buf_bytes;
            const bool more = c4 + 1 < k1;
            if (more) gload(c4 + 1);
            compute(cur);
            if (++since == fold) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tot[j] += acc[j];
                    acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                since = 0;
            }
            if (more) lstore(nxt);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tot[j] += acc[j];

        // epilogue through LDS (the staging buffers are free after the last barrier): `ts` neighbouring waves hold tap
        // slices of one output position and are summed here; whole units go to dst, cut ones to the scratch tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 v = {tot[0][r], tot[1][r], tot[2][r], tot[3][r]};
            *(f32x4*)(redf + wid * 1024 + (kk * 4 + r) * 64 + (n << 2)) = v;
        }
        __syncthreads();
        const int n_slots = V3_WAVES / ts;
        for (int e = tid; e < n_slots * 256; e += 512) {
            const int sl = e >> 8;
            const int row = (e >> 4) & 15;
            const int cq = e & 15;
            f32x4 s = *(const f32x4*)(redf + (sl * ts) * 1024 + row * 64 + (cq << 2));
            for (int w = 1; w < ts; ++w) s += *(const f32x4*)(redf + (sl * ts + w) * 1024 + row * 64 + (cq << 2));
            if (part >= 0) {
                *(f32x4*)(d.scratch + ((size_t)part * n_slots + sl) * 1024 + row * 64 + (cq << 2)) = s;
            } else {
                const int op = d.wave_tab[((size_t)pc * V3_WAVES + sl * ts) * 4];
                if (op >= 0) emit(d, g, row, op, ct * 64 + (cq << 2), s);
            }
        }
        __syncthreads();                              // the next segment reuses the tables and the buffers
    }
}

// one block per cut unit: sum its partial tiles in K order, then the usual epilogue
__global__ __launch_bounds__(256) void conv3_fixup_kernel(hmvae_conv3_desc d) {
    const int* fx = d.fixups + 8 * blockIdx.x;        // {g, pc, ct, first part, n parts, 0, 0, 0}
    const int g = fx[0], pc = fx[1], ct = fx[2], p0 = fx[3], np = fx[4];
    const int n_slots = V3_WAVES / d.ts;
    for (int e = threadIdx.x; e < n_slots * 256; e += 256) {
        const int sl = e >> 8;
        const int row = (e >> 4) & 15;
        const int cq = e & 15;
        const int op = d.wave_tab[((size_t)pc * V3_WAVES + sl * d.ts) * 4];
        if (op < 0) continue;
        f32x4 s = *(const f32x4*)(d.scratch + ((size_t)p0 * n_slots + sl) * 1024 + row * 64 + (cq << 2));
        for (int i = 1; i < np; ++i)
            s += *(const f32x4*)(d.scratch + ((size_t)(p0 + i) * n_slots + sl) * 1024 + row * 64 + (cq << 2));
        emit(d, g, row, op, ct * 64 + (cq << 2), s);
    }
}

template <int NS, int UN, int NA>
int launch3(const hmvae_conv3_desc* d, hipStream_t s) {
    const size_t buf = (size_t)4 * d->win * 256 + (size_t)(d->kt + 1) * 256;
    size_t lds = (size_t)4 * V3_MAX_C4 * 4 + 2 * buf;
    const size_t red = (size_t)4 * V3_MAX_C4 * 4 + (size_t)V3_WAVES * 4096;
    if (red > lds) lds = red;
    if (d->lds_floor > 0 && lds < (size_t)d->lds_floor) lds = (size_t)d->lds_floor;     // pins workgroups per CU
    if (lds > 160 * 1024) return HMVAE_ERR_UNSUPPORTED;
    auto kern = conv_window_kernel<NS, UN, NA>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return HMVAE_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)d->n_workers), dim3(512), lds, s, *d);
    if (hipGetLastError() != hipSuccess) return HMVAE_ERR_LAUNCH;
    if (d->n_fixups > 0) {
        hipLaunchKernelGGL(conv3_fixup_kernel, dim3((unsigned)d->n_fixups), dim3(256), 0, s, *d);
        if (hipGetLastError() != hipSuccess) return HMVAE_ERR_LAUNCH;
    }
    return HMVAE_OK;
}
}  // namespace

extern "C" int hmvae_gather_gemm_v3(const hmvae_conv3_desc* d, hmvae_stream_t stream) {
    if (!d || !d->src || !d->dst || !d->wp || !d->items || !d->wave_tab || !d->win_rows || !d->segs || !d->worker_seg ||
        (d->bias && !d->bias_idx))
        return HMVAE_ERR_ARG;
    if (d->bp <= 0 || (d->bp & 63) || d->n_groups <= 0 || d->n_chunks <= 0 || d->kt <= 0 || d->kt > 64) return HMVAE_ERR_ARG;
    if (d->ts != 1 && d->ts != 2 && d->ts != 4 && d->ts != 8) return HMVAE_ERR_ARG;
    if (d->win < 4 || d->win > 32) return HMVAE_ERR_UNSUPPORTED;
    if (d->un < 3 || d->un > 5 || d->max_items <= 0 || d->max_items % d->un) return HMVAE_ERR_ARG;
    if (d->max_nc4 <= 0 || d->max_nc4 > V3_MAX_C4 || d->max_items > (V3_MAX_ITEMS + d->un - 1) / d->un * d->un) return HMVAE_ERR_UNSUPPORTED;
    if (d->n_workers <= 0 || (d->n_workers & 7) || d->n_fixups < 0 || (d->n_fixups > 0 && (!d->fixups || !d->scratch)))
        return HMVAE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int ns = (64 * d->win + 511) / 512;
#define V3_CASE(NS_, UN_) \
    case NS_ * 10 + UN_: return d->kt <= 32 ? launch3<NS_, UN_, 1>(d, s) : launch3<NS_, UN_, 2>(d, s)
    switch (ns * 10 + d->un) {
        V3_CASE(1, 3); V3_CASE(1, 4); V3_CASE(1, 5);
        V3_CASE(2, 3); V3_CASE(2, 4); V3_CASE(2, 5);
        V3_CASE(3, 3); V3_CASE(3, 4); V3_CASE(3, 5);
        V3_CASE(4, 3); V3_CASE(4, 4); V3_CASE(4, 5);
        default: return HMVAE_ERR_UNSUPPORTED;
    }
#undef V3_CASE
}
