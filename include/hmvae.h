/*
 * hmvae.h — C ABI of libhmvae_hip.so, the MI355X (gfx950) kernels under the hm-vae training hot path.
 *
 * The reference (lijiaman/hm-vae) has no FFI: its "operator interface" is PyTorch modules.  Each entry
 * point below names the reference code it replaces (paths relative to the reference root).  The
 * Python side (hm_vae_amd/_lib.py) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it is called `desc` or `*_host`;
 *   - the caller owns all memory; nothing here allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t; 0 = default stream) so calls are graph-capturable;
 *   - return value: HMVAE_OK or a negative HMVAE_ERR_* (hmvae_error_string() gives text); never aborts;
 *   - activations inside the engine are "SoA": float [rows][Bp], row = channel*T + t (or the strided
 *     forms described per call), Bp = batch padded to a multiple of 64, batch innermost;
 *   - "AoS" is the reference's layout: (N, J, 6|9|3) frame-major float.
 */
#ifndef HMVAE_H
#define HMVAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMVAE_OK 0
#define HMVAE_ERR_ARG (-1)
#define HMVAE_ERR_LAUNCH (-2)
#define HMVAE_ERR_UNSUPPORTED (-3)

#define HMVAE_FLAG_LEAKY 1      /* dst = leaky_relu(acc + bias, slope) */
#define HMVAE_FLAG_ACCUM 2      /* dst += result (read-modify-write) */
#define HMVAE_FLAG_LEAKY_GRAD 4 /* v2 / v4 gather-GEMMs: dst = (result [+ dst]) * (act > 0 ? 1 : slope) with act read at
                                   act[act_rows[g,r] + p*dst_ts][b] (act_rows NULL: dst_rows; an entry of -1: factor 1) --
                                   the LeakyReLU backward of the layer that produced `act`, fused into the data-gradient
                                   GEMM that writes its output gradient (seq_two_hier_sa_vae.py:129, :250 + autograd) */

#define HMVAE_PAD_REFLECT 0
#define HMVAE_PAD_ZEROS 1

typedef void* hmvae_stream_t;   /* hipStream_t */

int hmvae_version(void);
const char* hmvae_error_string(int code);

/* ------------------------------------------------------------------------------------------------
 * Gather-GEMM on fp32 MFMA (v_mfma_f32_16x16x4_f32).  One kernel covers
 *   SkeletonConv forward            skeleton.py:95-105   (masked weight, reflect pad, stride)
 *   SkeletonConv backward-data      (autograd of the above)
 *   nn.Linear latent heads fwd/bwd  seq_two_hier_sa_vae.py:162, 267
 * dst[dst_rows[g,r] + p*dst_ts][b] (+)= act( bias[g,r] + sum_{(q,tap) in pairs(p)} sum_{c in list(g)}
 *                                          W[g][c][tap][r] * src[chan_rows[c] + q*src_ts][b] )
 * Packed weight layout per group: [c4 < nc4][tap < kt][mt][kk(4)*16 + m] (64 floats per MFMA A operand).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* src;        /* [src_rows][bp] */
    float* dst;              /* [dst_rows][bp] */
    const float* wp;         /* packed weights */
    const float* bias;       /* dense bias vector or NULL */
    const int32_t* bias_idx; /* [n_groups*mt*16] index into bias per packed row (-1: none) */
    const int32_t* g_nc4;    /* [n_groups] number of 4-channel steps */
    const int32_t* g_woff;   /* [n_groups] float offset of the group's weights in wp */
    const int32_t* g_choff;  /* [n_groups] offset of the group's channel list in chan_rows */
    const int32_t* chan_rows;/* src row base per listed channel (dummy entries point at row 0, weight 0) */
    const int32_t* dst_rows; /* [n_groups*mt*16] dst row base, or -1 for padding rows */
    const int32_t* p_off;    /* [n_pos] */
    const int32_t* p_len;    /* [n_pos] */
    const int32_t* pairs;    /* q | (tap << 16) */
    int32_t n_groups, mt, kt, n_pos, bp;
    int32_t src_ts, dst_ts;  /* row stride per position step on each side */
    int32_t flags;
    float slope;
} hmvae_gemm_desc;

int hmvae_gather_gemm(const hmvae_gemm_desc* desc, hmvae_stream_t stream);

/* Second-generation gather-GEMM (the one the engine uses): same maths and weight layout as hmvae_gather_gemm; a
 * wave owns mt*16 rows x np*64 batch columns of ONE output position and walks that position's list of "items"
 * (source position q, tap) with its operand loads software-pipelined.  Work unit = (group, position, np*64-column
 * tile); units are dealt to XCDs in contiguous group-major ranges.  `np` = 64-column units per wave (1, 2 or 4). */
typedef struct {
    const float* src;
    float* dst;
    const float* wp;
    const float* bias;
    const int32_t* bias_idx;
    const int32_t* g_nc4;
    const int32_t* g_woff;
    const int32_t* g_choff;
    const int32_t* chan_rows;
    const int32_t* dst_rows;
    const int32_t* pb_off;   /* [n_pblk] first item of the block */
    const int32_t* pb_len;   /* [n_pblk] */
    const int32_t* items;    /* pairs of int32: q, taps packed 8 bits per owned position */
    int32_t n_groups, mt, kt, n_pos, bp;
    int32_t np, n_pblk, n_units, ksplit;
    int32_t xcd_mode;             /* 0: units dealt to XCDs in group-major ranges; 1: position-major ranges, group fastest */
    int32_t wpos;                 /* 1: the workgroup's waves split K (ksplit); 4: they take 4 neighbouring position blocks */
    int32_t max_items, max_nc4;   /* largest pb_len / g_nc4 (the kernel stages both tables in LDS: <= 256 / <= 512) */
    int32_t src_ts, dst_ts;
    int32_t flags;
    float slope;
    uint64_t* debug;         /* NULL in production; a diagnostic build (-DHMVAE_STAMP) writes 4 s_memtime stamps per wave */
    const int32_t* bias_idx2;/* NULL, or a second bias index per packed row (-1: none): bias = mean of the two (rows that
                                are the mean of two dense rows: SkeletonPool folded into the convolution) */
    const float* act;        /* HMVAE_FLAG_LEAKY_GRAD: the activation tensor (laid out like dst) */
    const int32_t* act_rows; /* [n_groups*mt*16] act row base per packed row, -1: no factor; NULL: dst_rows */
} hmvae_gemm2_desc;

int hmvae_gather_gemm_v2(const hmvae_gemm2_desc* desc, hmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Window gather-GEMM (third generation) for the temporal convolutions: SkeletonConv forward and
 * backward-data (skeleton.py:95-105), 16-row groups only.  Same operands and epilogue as v2.
 * A UNIT is (row group, chunk of neighbouring output positions, 64 batch columns).  The contiguous
 * window of source positions [chunk_q0, chunk_q0 + win) a chunk reads is staged in LDS once per
 * 4-channel K step and shared by the 8 waves of the workgroup; each wave owns one output position
 * of the chunk (ts == 1) or one of `ts` slices of its items, summed through LDS.
 *   items       = [n_chunks][8][32] int32, (window row) | (tap << 16); every wave's item count
 *                 (<= 16) is padded to a multiple of `un` with items whose tap == kt (an all-zero
 *                 weight fragment)
 *   wave_tab    = [n_chunks][8][4] int32: {output position or -1, pattern, item count, 0}; waves
 *                 w*ts .. w*ts+ts-1 share an output position.  pattern 1: the 15 items are (r0+i,
 *                 tap i); pattern 2: (r0+i, tap 14-i); 0: anything (only with un == 5 may it be != 0)
 *   win_rows    = [n_chunks][win] int32: source position staged in each window row
 * The launch is persistent: n_workers workgroups (a multiple of 8), worker w runs the segments
 * worker_seg[2w] .. +worker_seg[2w+1];  segs[s] = {group, chunk, column tile, k0, k1, part,
 * g_choff[group], g_woff[group]} is the K-step range [k0, k1) of a unit.  part < 0: the segment covers the whole unit and stores
 * to dst.  part >= 0: the unit was cut between workers; the raw tile goes to scratch[part] and
 * fixups[f] = {group, chunk, column tile, first part, n parts, 0, 0, 0} lists the cut units, whose
 * parts are summed in K order by a second launch.  planner.py builds the schedule so that every
 * worker issues the same number of MFMAs.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* src;
    float* dst;
    const float* wp;         /* packed weights of 16-row groups, as for v2 (hmvae_pack_weights) */
    const float* bias;
    const int32_t* bias_idx;
    const int32_t* chan_rows;
    const int32_t* dst_rows;
    const int32_t* wave_tab; /* [n_chunks][8][4] */
    const int32_t* win_rows; /* [n_chunks][win] */
    const int32_t* items;    /* [n_chunks][8][32] */
    const int32_t* segs;     /* [n_segs][8] */
    const int32_t* worker_seg;/* [n_workers][2] */
    const int32_t* fixups;   /* [n_fixups][8] */
    float* scratch;          /* [n_parts][8/ts][16][64] partial tiles of cut units */
    int32_t n_groups, kt, bp, n_chunks, ts, win, max_nc4, max_items, un;
    int32_t n_workers, n_fixups;
    int32_t lds_floor;       /* request at least this much LDS per workgroup (pins the workgroups per CU the schedule
                                was made for); 0 = only what the kernel needs */
    int32_t src_ts, dst_ts, flags;
    float slope;
    const int32_t* bias_idx2;/* as in hmvae_gemm2_desc */
} hmvae_conv3_desc;

int hmvae_gather_gemm_v3(const hmvae_conv3_desc* desc, hmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fourth-generation gather-GEMM: the same contract as hmvae_gather_gemm_v2 (SkeletonConv forward /
 * backward-data, skeleton.py:95-105, with everything folded into its weights), computed on the bf16
 * matrix pipe with fp32-grade results.  Every fp32 operand is the exact sum of three bf16 planes
 * (round to nearest at each level); of the nine cross products the six of weight >= 2^-16 carry
 * everything above 2^-24 of a product and run as v_mfma_f32_16x16x32_bf16 with fp32 accumulation --
 * 96 issue cycles where v_mfma_f32_16x16x4_f32 needs 256 for the same K.
 *   weights     : packed once per step by hmvae_pack_weights_v4 into MFMA A-operand order, fp32:
 *                 wp4[group][c32 < nc32][tap < kt][tile < mtg][h < 2][lane (m, kk)][4 floats] (2 KB per fragment),
 *                 channel of (h, lane, j) = c32*32 + kk*8 + 4h + j of the group's list (lists are padded to 32);
 *                 the wave that owns a row tile splits its fragment into the three planes in registers;
 *   activations : stay fp32 in HBM ([rows][bp]); a workgroup (4 waves) stages the 32 rows of a K step
 *                 (32 channels of one source position) for its 64 / 128 batch columns into LDS as three
 *                 row-major bf16 planes (one split per element per workgroup) and every wave takes
 *                 its B operands from them with the transposing read ds_read_b64_tr_b16.
 * A UNIT is (group, output position, block of nc = 64|128 columns); wave w of the workgroup owns row
 * tiles w*mtw .. w*mtw + mtw - 1 of the group (mtg = 4*mtw tiles of 16 rows per group).  The K steps
 * (c32, item) of a unit are cut into n_split equal parts, one workgroup each; for n_split > 1 the raw
 * tiles go to scratch[unit][part] and the LAST workgroup of a unit to finish (counters[unit], an
 * agent-scope atomic after a release fence) sums the parts in part order and runs the epilogue --
 * deterministic, no second launch, nobody waits.  counters must be zero before the first launch and
 * are left zero by every launch.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* src;
    float* dst;
    const uint16_t* wp4;     /* packed fragments (hmvae_pack_weights_v4) */
    const float* bias;
    const int32_t* bias_idx; /* [n_groups*mtg*16], as v2 */
    const int32_t* bias_idx2;/* NULL or as v2 */
    const int32_t* g_nc32;   /* [n_groups] 32-channel steps of the group */
    const int32_t* g_woff4;  /* [n_groups] offset of the group's weights in 1 KB fragments */
    const int32_t* g_choff;  /* [n_groups] offset of the group's channel list in chan_rows */
    const int32_t* g_ntiles; /* [n_groups] row tiles that hold at least one valid row */
    const int32_t* chan_rows;/* src row base per listed channel (dummy entries: a valid row, weight 0) */
    const int32_t* dst_rows; /* [n_groups*mtg*16] dst row base or -1 */
    const int32_t* pb_off;   /* [n_pos] first item of the output position */
    const int32_t* pb_len;   /* [n_pos] */
    const int32_t* items;    /* pairs of int32: source position q, (virtual) tap */
    float* scratch;          /* [n_units][n_split][mtg*16][nc] (n_split > 1) */
    int32_t* counters;       /* [n_units] (n_split > 1) */
    int32_t n_groups, mtw, nc, kt, n_pos, bp;
    int32_t n_split, n_units;            /* n_units = n_groups * n_pos * (bp / nc) */
    int32_t max_items, max_nc32;         /* largest pb_len (<= 256) / g_nc32 (<= 32) */
    int32_t src_ts, dst_ts, flags;
    float slope;
    uint64_t* debug;         /* NULL in production; a diagnostic build (-DHMVAE_STAMP) writes 8 s_memtime stamps per workgroup */
    const float* act;        /* HMVAE_FLAG_LEAKY_GRAD, as in hmvae_gemm2_desc */
    const int32_t* act_rows;
    /* Reserved (round 2's position blocks -- units of 2 / 4 output positions sharing weight fragments and scheduled activation
     * images -- were measured never faster than the plain launch and removed in round 3): NULL, 0, 0.  pp > 1 is refused with
     * HMVAE_ERR_UNSUPPORTED. */
    const int32_t* blk_pos;
    int32_t pp, n_img;
} hmvae_gemm4_desc;

int hmvae_gather_gemm_v4(const hmvae_gemm4_desc* desc, hmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Weight-gradient GEMM (autograd of skeleton.py:95-105 / nn.Linear w.r.t. weight), split over the
 * (position, batch) reduction into `n_split` slabs, then a deterministic reduce into the dense grad.
 * slab[s][tile][mt*16][64] = sum_{t,b in split s} dz[dz_rows[g,r] + t*dz_ts][b] *
 *                                               x[chan_rows[c(col)] + q(t,tap(col))*x_ts][b]
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* dz;         /* [rows][bp] gradient w.r.t. conv output */
    const float* x;          /* [rows][bp] conv input */
    float* slab;             /* [n_split][n_tiles][mt*16][64] */
    const int32_t* tile_g;   /* [n_tiles] group of the tile */
    const int32_t* tile_col0;/* [n_tiles] first (channel,tap) column of the tile inside its group */
    const int32_t* g_ncols;  /* [n_groups] number of valid columns = listed channels * kt */
    const int32_t* g_choff;  /* [n_groups] */
    const int32_t* chan_rows;/* x row base per listed channel */
    const int32_t* dz_rows;  /* [n_groups*mt*16] dz row base or -1 */
    int32_t n_tiles, n_groups, mt, kt, n_split;
    int32_t t_dst, t_src, stride, pad, pad_mode, bp;
    int32_t dz_ts, x_ts;
    int32_t transposed;      /* 0: rows = output channels (dz), columns = (input channel, tap) read from x.
                                1 (stride 1, reflect only): rows = input channels (x, walked over the PADDED positions),
                                columns = (output channel, tap) read from dz at t' - tap, zero outside [0, t_dst);
                                then dz_rows holds x row bases and chan_rows holds dz row bases.  Same FLOPs per useful
                                weight, but the 16-row MFMA tiles are filled by the many input channels instead of the
                                few (6..12) output channels of a joint. */
    /* Persistent launch (n_workers > 0, a multiple of 8): worker w runs the items worker_items[2w] .. +worker_items[2w+1];
     * items[i] = {tile, first chunk, end chunk, slab part}: chunks are (position, 32 batch columns) steps of the tile's
     * reduction, numbered position-fastest; the tile's result is the sum of its parts (hmvae_wreduce_desc.tile_part0 /
     * tile_nparts).  planner.build_stream_schedule gives every worker the same number of chunks.  n_workers == 0: one
     * workgroup per (tile, split) as described above. */
    const int32_t* items;
    const int32_t* worker_items;
    int32_t n_workers;
    int32_t mode;            /* 0: operands straight from global memory in MFMA layout.  1: staged through LDS per 4 positions
                                (gradient rows as whole lines, the activation operand as a window of rows per channel); not
                                for transposed / bf16-plane / single-tap problems or fewer than 4 positions
                                (HMVAE_ERR_UNSUPPORTED).  2: staged through LDS like mode 1, but every element is split on
                                the way into THREE bf16 planes (x = p0 + p1 + p2 exactly, round to nearest) and the products
                                run on the bf16 matrix pipe as the six cross terms of weight >= 2^-16 -- fp32-grade results
                                (everything above 2^-24 of a product is kept, accumulation in fp32) at 3/8 of the fp32
                                instruction's issue cycles; same restrictions as mode 1 */
    /* Optional: both operands as three bf16 planes each (hmvae_split_bf16x3: value = plane0 + plane1 + plane2 exactly),
     * plane p of element i at planes[p * plane_stride + i], same element indexing as dz / x.  When dz_planes is given the
     * products run on the bf16 matrix pipe as six cross terms per fp32 product (everything above 2^-23 of it), fp32
     * accumulation: fp32-grade results at 3/8 of the fp32 MFMA issue cycles.  NULL: plain fp32 MFMA on dz / x. */
    const uint16_t* dz_planes;
    const uint16_t* x_planes;
    int64_t dz_plane_stride, x_plane_stride;
    /* Optional, mode 2 with n_split == 1 and n_workers == 0 only: the reduce descriptor of this weight tensor IN DEVICE
     * MEMORY.  The GEMM's epilogue then writes every tile straight into the dense gradient (mask, deduplicated children,
     * accumulate: exactly what hmvae_wgrad_reduce does) -- no slab traffic, no reduce launch.  NULL: write the slab. */
    const struct hmvae_wreduce_desc_s* fused_reduce;
    /* Optional, mode 2, n_workers == 0: the tiles taken two at a time.  pair_tab[2i], pair_tab[2i+1] = a tile and the next
     * 64-column tile of the SAME row group (or -1 for a group's last, unpaired tile), n_pairs entries; a workgroup of 8
     * waves then owns the pair and stages the gradient rows once for both.  Slab layout and reduce are unchanged.
     * With pair_tab, mt may be 5..7 ("wide tiles", round 3: up to 112 rows x 128 columns per workgroup, one workgroup per CU;
     * row groups cut evenly, dead 16-row tiles of a group cost nothing; at most four load passes over the activation
     * windows, i.e. every 15-tap layer -- otherwise HMVAE_ERR_UNSUPPORTED); without it mt <= 4. */
    const int32_t* pair_tab;
    int32_t n_pairs;
    int32_t pad2_;
} hmvae_wgrad_desc;

int hmvae_wgrad_gemm(const hmvae_wgrad_desc* desc, hmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Weight gradient of a convolution over few positions through the position-pair products (round 3).
 * The layer's input is a LINEAR image of a source tensor a (reflection padding; in the Decoder also nn.Upsample(x2, linear)
 * and SkeletonUnpool, seq_two_hier_sa_vae.py:233-242):  xin[c, pos(p, k)] = sum_q M[p, q, k] a[c, q], hence
 *     dW[row, chan, k] = sum_{p, q} M[p, q, k] * G[row, p, chan, q],   G = sum_b dz[row, p, b] * a[chan, q, b]
 * (backward of skeleton.py:95-105).  G is one dense GEMM over the batch -- t_dst * t_src products per weight instead of
 * t_dst * kt -- and the contraction with M runs in the same launch, which writes the dense gradient (mask, children of a
 * packed channel / row as in hmvae_wreduce_desc).  fp32-grade: three exact bf16 planes per operand, six cross terms, fp32
 * accumulation (as hmvae_wgrad_desc.mode 2); the contraction in fp32.
 *   dz row of (row r, position p) = dz_base[r] + p * dz_ts;  a row of (channel c, position q) = x_base[c] + q * x_ts; bp floats
 *   each.  t_dst = 1 << log_t (<= 64), t_src = 1 << log_s (<= 128), powers of two.  A tile = (128 >> log_t) rows x (128 >> log_s)
 *   channels starting at (tile_r0, tile_c0); tiles whose mask bytes are all zero need not be listed.
 *   ok[r * n_chan + c]: bit i = child i of the packed id on the child side takes the value (bit 0 alone without child_tab).
 *   m[(p * t_src + q) * 16 + k] = M[p, q, k], zero for k >= kt.
 * HMVAE_ERR_UNSUPPORTED when t_dst * t_src < 16 or the epilogue's image (t_dst * t_src * (8192 / (t_dst * t_src) + 17) floats)
 * exceeds the LDS.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* dz;
    const float* x;
    float* dense;
    const int32_t* dz_base;    /* [n_rows] */
    const int32_t* x_base;     /* [n_chan] */
    const int32_t* tile_r0;    /* [n_tiles] */
    const int32_t* tile_c0;    /* [n_tiles] */
    const uint8_t* ok;         /* [n_rows][n_chan] */
    const int32_t* child_tab;  /* [n][4] dense ids of the children (-1 unused) or NULL */
    const float* child_scale;  /* NULL, or a factor per packed id on the child side */
    const float* m;            /* [t_dst * t_src][16] */
    int32_t n_tiles, n_rows, n_chan, kt;
    int32_t log_t, log_s, bp, dz_ts, x_ts;
    int32_t row_stride, chan_stride;   /* of the dense tensor, in floats (tap stride 1) */
    int32_t child_side;        /* 0: children are channels, 1: rows */
    int32_t accumulate;
    int32_t pad_;
} hmvae_wgrad_pq_desc;

int hmvae_wgrad_pq(const hmvae_wgrad_pq_desc* desc, hmvae_stream_t stream);

/* x[i] -> planes[p * plane_stride + i], p = 0..2: the exact three-way bf16 split of an fp32 tensor (each plane is the
 * round-to-nearest bf16 of the running remainder: x = p0 + p1 + p2 exactly), n a multiple of 8.  Elements n .. plane_stride
 * of a plane are not touched: hmvae_wgrad_desc.mode 2 reads the last 32 of them as its zero row (plane_stride >= n + 32,
 * zero-filled by the caller once). */
int hmvae_split_bf16x3(const float* x, uint16_t* planes, int64_t n, int64_t plane_stride, hmvae_stream_t stream);

/* dense[row_out[g,r]*row_stride + chan_id[c]*chan_stride + tap] (+)= sum_s slab[s][tile][r][col] */
typedef struct hmvae_wreduce_desc_s {
    const float* slab;
    float* dense;
    const int32_t* tile_g;
    const int32_t* tile_col0;
    const int32_t* g_ncols;
    const int32_t* g_choff;
    const int32_t* chan_id;  /* dense channel index per listed channel, -1 for dummies */
    const int32_t* row_id;   /* [n_groups*mt*16] dense row index or -1 */
    const int32_t* g_nc4;    /* [n_groups] */
    const int32_t* g_okoff;  /* [n_groups] offset into pair_ok */
    const uint8_t* pair_ok;  /* per group [mt*16][nc4*4]: 1 when (row, channel) is inside the neighbour mask */
    int32_t n_tiles, mt, kt, n_split;
    int32_t row_stride, chan_stride;
    int32_t accumulate;
    /* Deduplicated weight gradients (composite ops, see hmvae_pack_op): copies of one coarse channel (SkeletonUnpool) see
     * the same activations, rows averaged by SkeletonPool the same output gradient, so their weight gradients are
     * computed once.  With child_tab, an element is written to every child i (child_tab[id*4 + i], replacing the
     * channel index for child_side 0, the row index for child_side 1) whose bit is set in pair_ok, times
     * child_scale[id] (the mean's 1/n) when given. */
    const int32_t* child_tab;
    const float* child_scale;
    int32_t child_side, pad_;
    const int32_t* tile_part0;  /* NULL: slab[s][tile], s < n_split.  Else the parts of tile t are slab[tile_part0[t] + i], */
    const int32_t* tile_nparts; /* i < tile_nparts[t] (persistent weight-gradient launch) */
} hmvae_wreduce_desc;

int hmvae_wgrad_reduce(const hmvae_wreduce_desc* desc, hmvae_stream_t stream);

/* Several reductions in one launch: descs_dev[i] as above, block_start_dev[i] = first workgroup of reduction i
 * (exclusive prefix sum, n_ops+1 entries; give each one ~ceil(tiles*mt*16*64/256) workgroups, capped as you like). */
int hmvae_wgrad_reduce_multi(const hmvae_wreduce_desc* descs_dev, const int32_t* block_start_dev, int32_t n_ops,
                             int32_t total_blocks, hmvae_stream_t stream);

/* packed[g][c4][tap][mt][kk*16+m] = dense[row_id*row_stride + chan_id*chan_stride + tap*tap_stride]
 * (0 for -1 ids and for (row, channel) pairs outside the mask: this is skeleton.py:96 `weight * mask`) */
int hmvae_pack_weights(const float* dense, float* packed, const int32_t* g_nc4, const int32_t* g_woff,
                       const int32_t* g_choff, const int32_t* chan_id, const int32_t* row_id,
                       const int32_t* g_okoff, const uint8_t* pair_ok, int32_t n_groups, int32_t max_nc4,
                       int32_t mt, int32_t kt, int32_t row_stride, int32_t chan_stride, int32_t tap_stride,
                       int32_t tap_flip, hmvae_stream_t stream);

/* All of a model's weight packs in ONE launch: `ops_dev` is a device array of n_ops descriptors whose block_start
 * fields are the exclusive prefix sum of n_groups*max_nc4 (one workgroup per (group, 4-channel step)). */
typedef struct {
    const float* dense;
    float* packed;
    const int32_t* g_nc4;
    const int32_t* g_woff;
    const int32_t* g_choff;
    const int32_t* chan_id;
    const int32_t* row_id;
    const int32_t* g_okoff;
    const uint8_t* pair_ok;
    int32_t n_groups, max_nc4, mt, kt;
    int32_t row_stride, chan_stride, tap_stride, block_start;
    int32_t cb;              /* 4-channel steps per workgroup (1..4) */
    /* Composite ops: the packed matrix is  Weff[row, chan, vt] = sum_j fold_coef[vt][j] * sum_i W[row_i, chan_i,
     * fold_taps[vt][j]]  with `kt` VIRTUAL taps vt over the fold_kt taps of the dense tensor (taps < 0 unused), and the
     * inner sum over the dense (row, channel) pairs merged into the packed pair: bit i of pair_ok selects child i,
     * child_tab[id*4 + i], of the packed channel (child_side 0) or of the packed row (child_side 1); without child_tab
     * the pair is itself (bit 0).  This is how a convolution over few positions (taps folded per position pair), the x2
     * linear upsample in front of a decoder convolution (a polyphase filter over the coarse positions) and
     * SkeletonUnpool (copies of a coarse channel read through summed weights) run as plain gather-GEMMs over fewer
     * (position, tap) items and channels; planner.compose_conv / merge_channels build the tables.  fold_taps == NULL:
     * plain op, pair_ok is 0/1. */
    const int32_t* fold_taps;   /* [kt][fold_nf] */
    const float* fold_coef;     /* [kt][fold_nf] */
    const int32_t* child_tab;   /* [n][4] or NULL */
    int32_t fold_nf, fold_kt, child_side, fold_pad;
    const float* child_scale;   /* NULL, or a factor per packed id on the child side (1/2: the mean of two children) */
} hmvae_pack_op;

/* One workgroup per (group, cb consecutive 4-channel steps): op i owns the blocks block_start ..
 * block_start + n_groups*ceil(max_nc4/cb).  max_lds_floats: the largest per-op LDS need, cb*kt*mt*64 floats, plus
 * kt*16 for a composite op (fold_kt <= 16, 16-row groups) */
int hmvae_pack_weights_multi(const hmvae_pack_op* ops_dev, int32_t n_ops, int32_t total_blocks, int32_t max_lds_floats,
                             hmvae_stream_t stream);

/* Weight packs for hmvae_gather_gemm_v4: the effective matrix of a hmvae_pack_op -- mask, virtual taps, merged
 * rows / channels: same tables, lists padded to 32 channels, g_nc4 = 8 * nc32 -- as fp32 fragments in the
 * layout above.  One workgroup per (group, row tile, 32 channels); op i owns the blocks block_start .. block_start +
 * n_groups * mt * max_nc4 / 8 (mt = tiles per group, g_woff in 1 KB fragments, packed = the uint16 buffer). */
int hmvae_pack_weights_v4(const hmvae_pack_op* ops_dev, int32_t n_ops, int32_t total_blocks, int32_t max_kt,
                          hmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The latent bridge of one hierarchy level as one launch per direction: the per-edge nn.Linear head of the Encoder
 * (seq_two_hier_sa_vae.py:162-170: features -> (mu, logvar)), reparametrize + kl_loss (:359-391) and the per-edge head
 * of the Decoder (:267-272: z -> features), with the arithmetic of hmvae_reparam_kl_fwd / _bwd in the middle.  The
 * heads' weights are the dense nn.Linear tensors ([out][in], shared by the edges); rows are SoA ([rows][bp]):
 *   x    [e * x_es + i]          i < n_in     the encoder's activations of edge e
 *   dist [e * 2d + r]            r < 2d       (mu | logvar), kept for the backward pass
 *   z    [e * d + j]             j < d
 *   y    [e * y_es + y_off + o]  o < n_out    the decoder's input features of edge e
 * A workgroup owns (edge, 16 batch columns); kl_partial gets one sum per workgroup: [k_edges * bp / 16].
 * d <= 24 and a multiple of 4, n_in and n_out multiples of 64 (the waves take quarters in 16-row tiles), bp a multiple of 64:
 * HMVAE_ERR_UNSUPPORTED otherwise (the caller runs the three launches).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* x;
    const float* w1;         /* [2d][n_in] */
    const float* b1;         /* [2d] */
    const float* eps;        /* [B * k_edges][d] noise (sample != 0) */
    const float* w2;         /* [n_out][d] */
    const float* b2;         /* [n_out] */
    float* dist;
    float* z;
    float* y;
    float* kl_partial;
    int32_t k_edges, n_in, d, n_out, x_es, y_es, y_off, B, bp, sample;
} hmvae_bridge_fwd_desc;

int hmvae_latent_bridge_fwd(const hmvae_bridge_fwd_desc* desc, hmvae_stream_t stream);

/* Backward: dz = W2^T dy, ddist = kl_scale * d(KL)/d(dist) + the reparametrisation's backward (written: the heads' weight
 * gradients read it), dx = W1^T ddist -- with HMVAE_FLAG_ACCUM added to what dx holds, with HMVAE_FLAG_LEAKY_GRAD then
 * multiplied by (act > 0 ? 1 : slope), act laid out like x (the LeakyReLU backward of the layer that produced x).
 * dx NULL: stop after ddist.  eps NULL: no sampling (z = mu). */
typedef struct {
    const float* dy;         /* gradient w.r.t. y, laid out like y */
    const float* w2;
    const float* dist;
    const float* eps;
    const float* w1;
    const float* act;
    float* ddist;            /* laid out like dist */
    float* dx;               /* laid out like x, or NULL */
    int32_t k_edges, n_in, d, n_out, x_es, y_es, y_off, B, bp, flags;
    float kl_scale, slope;
} hmvae_bridge_bwd_desc;

int hmvae_latent_bridge_bwd(const hmvae_bridge_bwd_desc* desc, hmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise / reduction kernels on SoA activations
 * ------------------------------------------------------------------------------------------------ */
/* SkeletonPool (mean of <=2 edges) + LeakyReLU: skeleton.py:228-231 + seq_two_hier_sa_vae.py:129.
 * out[(e'*ch + c)*T + t] = leaky( sum_k in[(src_e[e',k]*ch + c)*T + t] * w[e'] ); src_e2 = -1 for single. */
int hmvae_pool_leaky_fwd(const float* in, float* out, const int32_t* src_e1, const int32_t* src_e2, int32_t n_out_edges,
                         int32_t ch, int32_t T, int32_t bp, float slope, hmvae_stream_t stream);
/* din[(e*ch+c)*T+t] = dout[(dst_e[e]*ch+c)*T+t] * leaky'(out[...]) * w[e] */
int hmvae_pool_leaky_bwd(const float* dout, const float* out, float* din, const int32_t* dst_e, const float* w_e,
                         int32_t n_in_edges, int32_t ch, int32_t T, int32_t bp, float slope, hmvae_stream_t stream);
/* nn.Upsample(x2, linear, align_corners=False) + SkeletonUnpool: seq_two_hier_sa_vae.py:233-242, skeleton.py:258-261.
 * out[(e*ch+c)*T_out + t] = lerp(in[(src_e[e]*ch+c)*T_in + ...]); up=0 copies (T_out == T_in). */
int hmvae_upsample_unpool_fwd(const float* in, float* out, const int32_t* src_e, int32_t n_out_edges, int32_t ch,
                              int32_t T_in, int32_t up, int32_t bp, hmvae_stream_t stream);
/* transpose of the above: din[(e'*ch+c)*T_in + t] = sum over fine edges e of e' and output positions */
int hmvae_upsample_unpool_bwd(const float* dout, float* din, const int32_t* fine_off, const int32_t* fine_list,
                              int32_t n_in_edges, int32_t ch, int32_t T_in, int32_t up, int32_t bp,
                              hmvae_stream_t stream);
/* dy = dout * (out > 0 ? 1 : slope)  (LeakyReLU backward from the activation's output) on n_blocks blocks of
 * block_len floats that start `block_stride` floats apart (same addressing for dout, out and dy) */
int hmvae_leaky_bwd(const float* dout, const float* out, float* dy, int64_t block_len, int64_t n_blocks,
                    int64_t block_stride, float slope, hmvae_stream_t stream);
/* dst[i] (+)= sum_{k < rows_per_out} sum_cols src[rows[i] + k*row_step][:]   (bias gradients; padded batch
 * columns are zero by construction) */
int hmvae_row_sums(const float* src, float* dst, const int32_t* rows, int32_t n_out, int32_t rows_per_out,
                   int32_t row_step, int32_t n_cols, int32_t accumulate, hmvae_stream_t stream);
/* the same with a factor per output (scale[o] * sum): bias gradients of rows that were averaged by SkeletonPool */
int hmvae_row_sums_scaled(const float* src, float* dst, const int32_t* rows, const float* scale, int32_t n_out,
                          int32_t rows_per_out, int32_t row_step, int32_t n_cols, hmvae_stream_t stream);
/* Several row-sum jobs (bias gradients) in one launch; one workgroup per output element. */
typedef struct {
    const float* src;
    float* dst;
    const int32_t* rows;
    int32_t n_out, rows_per_out, row_step, n_cols, accumulate, block_start;
} hmvae_rowsum_op;

int hmvae_row_sums_multi(const hmvae_rowsum_op* ops_dev, int32_t n_ops, int32_t total_blocks, hmvae_stream_t stream);

/* AoS (B, T, C) <-> SoA (C, T, Bp) */
int hmvae_aos_to_soa(const float* aos, float* soa, int32_t B, int32_t T, int32_t C, int32_t bp, hmvae_stream_t stream);
int hmvae_soa_to_aos(const float* soa, float* aos, int32_t B, int32_t T, int32_t C, int32_t bp, hmvae_stream_t stream);

/* n standard-normal floats (the reparametrisation's noise, seq_two_hier_sa_vae.py:359-366 torch.randn_like): Philox4x32-10
 * keyed by `seed`, counter = (element / 4, launch count) + Box-Muller.  state[0] = launch count, state[1] = scratch (both
 * zero at first; the launch advances state[0] itself, so a captured graph draws fresh noise at every replay). */
int hmvae_randn(float* out, int64_t n, uint64_t seed, uint64_t* state, hmvae_stream_t stream);

/* reparametrize + KL: seq_two_hier_sa_vae.py:419-428.
 * dist rows: (e*2d + j) mu, (e*2d + d + j) logvar, [rows][bp]; eps AoS (B*k, d); z rows (e*d + j).
 * kl_partial[(j * k_edges + e) * ceil(bp/256) + block] = sum over the block's columns of -0.5*(1 + lv - mu^2 - exp(lv))
 * for component j of edge e (b < B only): d * k_edges * ceil(bp/256) entries */
int hmvae_reparam_kl_fwd(const float* dist, const float* eps, float* z, float* kl_partial, int32_t k_edges, int32_t d,
                         int32_t B, int32_t bp, int32_t sample, hmvae_stream_t stream);
/* ddist = dz through z = mu + eps*exp(lv/2) (if use_dz) + kl_scale * dKL/d(mu,lv) */
int hmvae_reparam_kl_bwd(const float* dist, const float* eps, const float* dz, float* ddist, int32_t k_edges, int32_t d,
                         int32_t B, int32_t bp, int32_t use_dz, float kl_scale, hmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Rotation lift + forward kinematics
 * ------------------------------------------------------------------------------------------------ */
/* my_tools.py:19-39 + fk_layer.py:47-93, AoS.  rot: (n, J, rot_dim) with rot_dim 6 or 9; offsets (J,3) shared
 * (offset_stride 0) or per frame (offset_stride J*3).  R_out (n,J,9) and pos_out (n,J,3) may be NULL.
 * parents_host: J ints on the HOST (fk_layer.py:13-34 takes any tree).  The SMPL-24 tree runs the compile-time fast path
 * (subtree per wave, registers + LDS tiles); any other tree with joint 0 as root, parents[j] < j and J <= 64 runs the
 * table-driven kernels (same results, one thread per (frame, joint)).  Invalid trees: HMVAE_ERR_ARG; J > 64 with a
 * non-SMPL tree: HMVAE_ERR_UNSUPPORTED. */
int hmvae_fk_fwd(const float* rot, int32_t rot_dim, const float* offsets, int64_t offset_stride,
                 const int32_t* parents_host, int32_t n_joints, int64_t n_frames, float* R_out, float* pos_out,
                 hmvae_stream_t stream);
/* backward of the above: d_rot (n,J,rot_dim) from dR (n,J,9; may be NULL) and dpos (n,J,3; may be NULL) */
int hmvae_fk_bwd(const float* rot, int32_t rot_dim, const float* offsets, int64_t offset_stride,
                 const int32_t* parents_host, int32_t n_joints, int64_t n_frames, const float* dR, const float* dpos,
                 float* d_rot, hmvae_stream_t stream);

/* seq_two_hier_sa_vae.py:436-474 (decode tail) fused with the three MSEs of :395-397/:430-434 and their
 * backward: SoA decoder output (144 rows x T) and SoA targets -> per-block loss partials + dL/d(dec_out).
 * scale_* = weight / numel of each MSE.  partial: [n_blocks][3] sums of squared errors (6d, rot, pos); with every
 * buffer NULL only *n_blocks_out is set (size query). */
int hmvae_decode_tail_loss(const float* dec_out, const float* gt6d, const float* gtR, const float* gtP,
                           const float* offsets, int32_t T, int32_t B, int32_t bp, float scale_6d, float scale_rot,
                           float scale_pos, float* d_dec_out, float* partial, int32_t* n_blocks_out,
                           hmvae_stream_t stream);
/* The same with l2_masked_criterion (seq_two_hier_sa_vae.py:717-735, the test-time optimisation loops :1363-1365): joint j
 * of frame (b, t) weighs its squared errors -- 6D, matrix and position alike -- by mask[(j*T + t)*bp + b] (SoA, 24*T
 * rows); the means keep the full element counts, as `loss.mean()` over the masked product does. */
int hmvae_decode_tail_loss_masked(const float* dec_out, const float* gt6d, const float* gtR, const float* gtP,
                                  const float* mask, const float* offsets, int32_t T, int32_t B, int32_t bp,
                                  float scale_6d, float scale_rot, float scale_pos, float* d_dec_out, float* partial,
                                  int32_t* n_blocks_out, hmvae_stream_t stream);
/* l2_criterion of a tensor against an anchor (seq_two_hier_sa_vae.py:1366-1373: the latent regulariser, p0 = NULL = zeros,
 * and the L2-to-original regulariser of the deep-copied decoder): g[i] += grad_scale * (p[i] - p0[i]) (g may be NULL),
 * partial[0..*n_blocks_out) = partial_scale * block sums of (p - p0)^2 in a fixed order.  All-NULL: size query. */
int hmvae_l2_anchor(const float* p, const float* p0, int64_t n, float grad_scale, float partial_scale, float* g,
                    float* partial, int32_t* n_blocks_out, hmvae_stream_t stream);
/* decode tail, inference: SoA decoder output -> AoS cont6d (B,T,24,6), R (B,T,24,3,3), pos (B,T,24,3).
 * c6 may be NULL (the frame-major copy of the 6D input is then not written). */
int hmvae_decode_tail_fwd(const float* dec_out, const float* offsets, int32_t T, int32_t B, int32_t bp, float* c6,
                          float* R, float* pos, hmvae_stream_t stream);

/* seq_two_hier_sa_vae.py:408-411 + the means of :425-434 in one launch: fixed-order sums of the MSE / KL block
 * partials, then out = [total, kl, rec_6d, rec_rot, rec_pose, kl_shallow, kl_deep, 0] */
int hmvae_vae_loss_finalize(const float* mse_partial, int32_t n_mse, const float* kl0_partial, int32_t n_kl0,
                            const float* kld_partial, int32_t n_kld, float inv_n6, float inv_nR, float inv_nP,
                            float inv_rows0, float inv_rowsd, float w6, float wR, float wP, float w_kl, float w_skl,
                            float* out8, hmvae_stream_t stream);

/* TrajectoryModel losses and their gradient (trajectory_pred_model.py:237-262 with gen_motion_w_trajectory :292-305
 * and l2_criterion :267-271).  v, g: SoA (3*T rows x bp) predicted / ground-truth normalised root velocity; std3: the
 * three de-standardisation factors std_vals[576:579] (device).  dv = dL/dv with grad_v = 2*rec_root_v_w/(3*T*B),
 * grad_t = 2*rec_root_trans_w/(3*T*B) (grad_t = 0 when use_accumulation_root_v is off).  partial: [n_blocks][2]
 * block sums (squared velocity error, squared trajectory error).  All-NULL buffers: only *n_blocks_out is set. */
int hmvae_traj_loss(const float* v, const float* g, const float* std3, int32_t T, int32_t B, int32_t bp, float grad_v,
                    float grad_t, float* dv, float* partial, int32_t* n_blocks_out, hmvae_stream_t stream);
/* out9 = [total, 0, 0, 0, 0, 0, rec_root_v, 0, rec_root_trans] (the return order of trajectory_pred_model.py:265) */
int hmvae_traj_loss_finalize(const float* partial, int32_t n_blocks, float inv_n, float w_v, float w_t, float* out9,
                             hmvae_stream_t stream);

/* The trajectory term of the test-time optimisation loop (seq_two_hier_sa_vae.py:1376-1389 with cal_key_frame_root_loss
 * :1031-1045 and gen_motion_w_trajectory :486-499).  v: SoA (3*T rows x bp) normalised root velocity the trajectory model
 * predicts for the decoded window; gt: SoA (3*T x bp) ground-truth root translation; mask: SoA (24*T x bp) of
 * l2_masked_criterion (row j*T + t) -- as released, the selected "key frames" are the indices j whose mask entry at FRAME 0
 * is non-zero; mean3 / std3: mean_vals / std_vals[576:579] (device).  For consecutive selected j' < j the loss term is
 * e^2 with e = sum_{s=j'+1..j} (mean + std*v[s]) - (gt[j] - gt[j']), and dv[s] = grad_scale * std * e for the s of that
 * interval (0 elsewhere): with grad_scale = 2*reg_w_trajectory/N, N = 3 * sum over windows of (selected - 1) counted by the
 * caller, dv is the gradient of reg_w_trajectory * mean(e^2).  partial: [n_blocks] block sums of e^2.  All-NULL buffers:
 * only *n_blocks_out is set. */
int hmvae_keyframe_root_loss(const float* v, const float* gt, const float* mask, const float* mean3, const float* std3,
                             int32_t T, int32_t B, int32_t bp, float grad_scale, float* dv, float* partial,
                             int32_t* n_blocks_out, hmvae_stream_t stream);

/* out[0..n_out) : fixed-order sums of partial[i*stride + k], scaled */
int hmvae_finalize_sums(const float* partial, int32_t n_partial, int32_t stride, int32_t k, float scale, float* out,
                        hmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Batch assembly from a device-resident dataset — MotionSeqData.__getitem__ (utils_motion_vae.py:124-187) for a whole
 * batch in one launch.  frames: the `T_i x 579` sequences of the on-disk format (utils/process_all_data_motion.py:155-158)
 * packed back to back, n_rows in total.  One descriptor per clip, built by the host sampler: first source row, frame
 * stride (change_fps, :66-83), and the root rotation of rand_rotation_matrix (:16-57) used when rot_flag != 0 (:161-185).
 * mean / stdv: the 579-entry float64 statistics (zero std already replaced by 1).  Outputs: the reference's 7-tuple,
 * each (B, T, n) contiguous fp32.  Source rows are clamped to [0, n_rows): validate descriptors on the host.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t row0;
    int32_t stride;
    int32_t reserved;
    float rot[9];      /* row-major */
    float pad_[3];
} hmvae_clip;          /* 64 bytes */
int hmvae_motion_batch(const float* frames, int64_t n_rows, const hmvae_clip* clips_dev, const double* mean,
                       const double* stdv, int32_t B, int32_t T, int32_t rot_flag, float* r6, float* rmat, float* rot_pos,
                       float* joint_pos, float* lin_v, float* ang_v, float* root_v, hmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Adam (torch.optim.Adam semantics: L2 weight decay folded into the gradient) — trainer_motion_vae.py:29-31, 92
 * One launch over a table of tensors; tensors whose gradient is absent are simply not listed.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
} hmvae_adam_tensor;

/* lr_dev: 1 float, step_dev: 1 int (the step being taken, >= 1) — both on the device so a captured
 * hipGraph replays with the current values. */
int hmvae_adam_step(const hmvae_adam_tensor* table_dev, int32_t n_tensors, int64_t max_numel, const float* lr_dev,
                    const int32_t* step_dev, float beta1, float beta2, float eps, float weight_decay,
                    hmvae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HMVAE_H */
