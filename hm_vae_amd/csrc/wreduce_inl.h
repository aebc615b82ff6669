// One element of the slab -> dense weight-gradient reduction (hmvae_wreduce_desc) for the fused epilogue of the
// weight-gradient GEMM (wgrad_bf3.hip): the same walk as wreduce_body (gemm_kernels.hip) -- column -> (channel, tap),
// mask byte, deduplicated children, accumulate -- on a sum the GEMM already holds.  Internal header.
#pragma once
#include "hmvae.h"

struct WreduceCol {          // what a thread keeps for one of the tile's 64 columns
    const unsigned char* okp;
    float* ocol;
    int cid, tap, ok_stride;
    bool live;
};

__device__ __forceinline__ WreduceCol wreduce_col(const hmvae_wreduce_desc& d, int tile, int col) {
    WreduceCol c;
    const int g = d.tile_g[tile];
    const int gc = d.tile_col0[tile] + col;
    c.live = gc < d.g_ncols[g];
    const int ch = c.live ? gc / d.kt : 0;
    c.tap = c.live ? gc - ch * d.kt : 0;
    c.cid = c.live ? d.chan_id[d.g_choff[g] + ch] : -1;
    c.live = c.live && c.cid >= 0;
    c.okp = d.pair_ok + d.g_okoff[g] + ch;
    c.ok_stride = d.g_nc4[g] * 4;
    c.ocol = d.dense + (size_t)(c.cid < 0 ? 0 : c.cid) * d.chan_stride + c.tap;
    return c;
}

// s = the finished sum of one tile element: column c, row with dense id `rid` (row_id[g * rows + row]) and mask byte `bits`
// (c.okp[row * c.ok_stride])
__device__ __forceinline__ void wreduce_store(const hmvae_wreduce_desc& d, const WreduceCol& c, int rid, unsigned bits, float s) {
    if (!c.live || rid < 0 || !bits) return;
    if (d.child_tab == nullptr) {
        float* o = c.ocol + (size_t)rid * d.row_stride;
        if (d.accumulate) s += *o;
        *o = s;
    } else {                                   // one computed gradient, several dense (row, channel) owners
        if (d.child_scale) s *= d.child_scale[d.child_side ? rid : c.cid];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            if (!((bits >> ci) & 1u)) continue;
            const int dr = d.child_side ? d.child_tab[rid * 4 + ci] : rid;
            const int dc = d.child_side ? c.cid : d.child_tab[c.cid * 4 + ci];
            float* o = d.dense + (size_t)dr * d.row_stride + (size_t)dc * d.chan_stride + c.tap;
            *o = d.accumulate ? *o + s : s;
        }
    }
}


// ---- round 3: the same walk with every table look-up that does not depend on the row resolved ONCE per column, and the
// per-row look-ups of a batch of rows issued before the first store.  The first form above chased, per element, descriptor
// field -> child table -> dense pointer between two stores (the descriptor lives in device memory and a float store may
// alias it, so the compiler re-read it after every store): three dependent memory round trips per element, 28 elements
// per thread of a 112 x 128 tile -- 20 us of a 100 us launch (rocprofv3, first decoder layer, wide tiles).
struct WreduceColR {
    float* o[4];             // child_side 0: this column in each child channel (nullptr: no such child); else o[0] = the column
    const unsigned char* okp;
    int ok_stride, cid;
    float scale;             // child_side 0: child_scale[cid] (or 1)
    bool live;
};

// d: a LOCAL copy of the descriptor (registers / scalar registers: nothing a store can alias)
__device__ __forceinline__ WreduceColR wreduce_col_resolve(const hmvae_wreduce_desc& d, int tile, int col) {
    WreduceColR c;
    const int g = d.tile_g[tile];
    const int gc = d.tile_col0[tile] + col;
    c.live = gc < d.g_ncols[g];
    const int ch = c.live ? gc / d.kt : 0;
    const int tap = c.live ? gc - ch * d.kt : 0;
    c.cid = c.live ? d.chan_id[d.g_choff[g] + ch] : -1;
    c.live = c.live && c.cid >= 0;
    c.okp = d.pair_ok + d.g_okoff[g] + ch;
    c.ok_stride = d.g_nc4[g] * 4;
    const int cid = c.cid < 0 ? 0 : c.cid;
    c.scale = 1.f;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) c.o[ci] = nullptr;
    if (d.child_tab != nullptr && d.child_side == 0) {
        if (c.live) {
            if (d.child_scale) c.scale = d.child_scale[cid];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                const int dc = d.child_tab[cid * 4 + ci];
                if (dc >= 0) c.o[ci] = d.dense + (size_t)dc * d.chan_stride + tap;
            }
        }
    } else {
        c.o[0] = d.dense + (size_t)cid * d.chan_stride + tap;
    }
    return c;
}

// NR finished sums of one column: rows with dense ids rid[i] (row_id[g * rows + row]) and mask bytes bits[i]
template <int NR>
__device__ __forceinline__ void wreduce_store_rows(const hmvae_wreduce_desc& d, const WreduceColR& c, const int (&rid)[NR],
                                                   const unsigned (&bits)[NR], const float (&s)[NR]) {
    if (!c.live) return;
    if (d.child_tab == nullptr) {
        float old[NR];
        if (d.accumulate) {
#pragma unroll
            for (int i = 0; i < NR; ++i) old[i] = (rid[i] >= 0 && bits[i]) ? c.o[0][(size_t)rid[i] * d.row_stride] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (rid[i] >= 0 && bits[i]) c.o[0][(size_t)rid[i] * d.row_stride] = d.accumulate ? old[i] + s[i] : s[i];
    } else if (d.child_side == 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (rid[i] < 0 || !bits[i]) continue;
            const float v = s[i] * c.scale;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                if (!((bits[i] >> ci) & 1u) || c.o[ci] == nullptr) continue;
                float* o = c.o[ci] + (size_t)rid[i] * d.row_stride;
                *o = d.accumulate ? *o + v : v;
            }
        }
    } else {
        int dr[NR][4];
        float sc[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const bool ok = rid[i] >= 0 && bits[i];
            sc[i] = (ok && d.child_scale) ? d.child_scale[rid[i]] : 1.f;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) dr[i][ci] = (ok && ((bits[i] >> ci) & 1u)) ? d.child_tab[rid[i] * 4 + ci] : -1;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float v = s[i] * sc[i];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                if (dr[i][ci] < 0) continue;
                float* o = c.o[0] + (size_t)dr[i][ci] * d.row_stride;
                *o = d.accumulate ? *o + v : v;
            }
        }
    }
}
