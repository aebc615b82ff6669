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
