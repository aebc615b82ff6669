// latent_bridge.hip — the latent bridge of one hierarchy level as ONE launch per direction (hmvae_latent_bridge_fwd / _bwd,
// contract in include/hmvae.h).  gfx950.
//
// Reference: the per-edge nn.Linear heads around the reparametrisation, seq_two_hier_sa_vae.py:162-170 (Encoder: features
// -> (mu, logvar)), :359-391 (reparametrize + kl_loss), :267-272 (Decoder: z -> features), and their autograd.
//
// Between the encoder's last and the decoder's first convolution the step ran three launches (head, reparametrisation,
// head) of a few microseconds each, and six in the backward pass -- 66 MFLOP in all, but serial and with a queue hand-over
// between every two: replaying the step without them is 45-70 us (3-5 %) shorter (tools/abl_step.py).  Every edge is
// independent (the heads' weights are shared by the edges, each edge's feature vector maps to its own latent), so a
// workgroup owns (edge, 16 batch columns) and walks the whole chain with the intermediate tiles in LDS:
//   forward   dist = W1 x + b1 (2d x 16)  ->  z = mu + eps * exp(logvar / 2), KL partial  ->  y = W2 z + b2
//   backward  dz = W2^T dy  ->  ddist = d(KL) + reparametrisation backward  ->  dx = W1^T ddist (+= / * leaky'(act))
// The three small products (48 x 384, 384 x 24 and their transposes) run on the fp32 matrix pipe; how the kernels got from
// 160-270 us per launch (fp32 FMAs, scalar weight loads) to 8-9 us is told in front of the kernels and in DESIGN.md 7.
#include <hip/hip_runtime.h>
#include "hmvae.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

constexpr int LB_MAX_D = 24;          // latent size per edge (host checks)
constexpr int LB_NT = 1;              // 16-column tiles per workgroup: 16 columns -> 4x the workgroups of a 64-column tile,
constexpr int LB_NC = 16 * LB_NT;     // a quarter of the serial MFMA chain each (28 workgroups left 228 CUs idle)
constexpr int LB_RS = 256 / LB_NC;    // row slots of the element-wise phases (thread = column c, rows rs, rs + LB_RS, ..)
#define LB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float lb_block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// Every phase is a small matrix product on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: lane (m, kk) supplies A[m][kk] and
// B[kk][m], and holds C[4kk .. 4kk+3][m]).  The reduction index is ours to order: lane group kk takes a CONTIGUOUS run of the
// reduction range, so a lane's A values are consecutive floats of one weight row where the weights are the row-major operand.
// What bounds these kernels is latency, not arithmetic (two FMA versions -- rounds of dependent row loads, then 64-byte
// scalar weight loads the compiler had to wait for one by one -- took 160-270 us per launch): all operands of a phase are
// independent loads issued up front.
//
// forward: dist (2d x 64) = W1 (2d x n_in) x (n_in x 64): the four waves split n_in, partial tiles summed through LDS;
//          y (n_out x 64) = W2 (n_out x d) z (d x 64): the waves split the rows.
__global__ __launch_bounds__(256) void latent_bridge_fwd_kernel(hmvae_bridge_fwd_desc p, int prio) {
    CHAIN_SETPRIO(prio);
    __shared__ float sP[4 * 48 * LB_NC];           // the waves' partial sums of dist (3 row tiles)
    __shared__ float sD[48 * LB_NC];               // (mu | logvar) of the tile, then z in its first d rows
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63;
    const int c = threadIdx.x % LB_NC, rs = threadIdx.x / LB_NC;       // element-wise phases: column, row slot
    const int m = lane & 15, kk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int e = blockIdx.y;
    const int c0 = blockIdx.x * LB_NC;
    const int col = c0 + c;
    const int bp = p.bp, d = p.d, d2 = 2 * p.d, n_in = p.n_in;
    const int dq = d >> 2;                         // d % 4 == 0 (host), dq <= 6
    const int oq = p.n_out >> 2;                   // rows of y per wave: a multiple of 16 (host), at most six tiles per pass
    constexpr int NJ = (LB_MAX_D + LB_RS - 1) / LB_RS;                 // latent components per thread
    float epsv[NJ], w2a[6][6], b2v[6][4];
    {
        const int kq = n_in >> 2;                  // reduction range of this wave; lane group kk takes kq / 4 of it
        const int ks = kq >> 2;                    // a multiple of 4 (host: n_in % 64 == 0)
        const int kb = wv * kq + kk * ks;
        f32x4 acc[3][LB_NT];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < LB_NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* xb = p.x + ((size_t)e * p.x_es + kb) * bp + c0 + m;
        const float* wb[3];
        bool wok[3];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const int r = mt * 16 + m;
            wok[mt] = r < d2;
            wb[mt] = p.w1 + (size_t)(wok[mt] ? r : 0) * n_in + kb;
        }
        // six rounds of four reduction steps per lane group at a time, every load of the six issued before the first MFMA
        // (unconditional: a round past the end re-reads round 0 and multiplies zeros)
        for (int sb = 0; sb < ks; sb += 24) {
            f32x4 a[6][3];
            float b[6][4][LB_NT];
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int s0 = sb + 4 * it < ks ? sb + 4 * it : 0;
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) a[it][mt] = *(const f32x4*)(wb[mt] + s0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nt = 0; nt < LB_NT; ++nt) b[it][i][nt] = xb[(size_t)(s0 + i) * bp + nt * 16];
            }
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const bool on = sb + 4 * it < ks;                      // (uniform)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) {
                        const float av = (on && wok[mt]) ? a[it][mt][i] : 0.f;
#pragma unroll
                        for (int nt = 0; nt < LB_NT; ++nt) acc[mt][nt] = LB_MFMA(av, b[it][i][nt], acc[mt][nt]);
                    }
            }
        }
        // Everything else the kernel reads from memory is requested NOW, behind the first phase's operands (in the step a
        // dependent round trip costs 3-5 us -- the weight-gradient GEMMs keep the memory system busy): the noise of this
        // thread's components and the decoder head's weights and biases of this wave's six row tiles.
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int j = rs + LB_RS * i;
            epsv[i] = (p.sample && j < d && col < p.B) ? p.eps[((size_t)col * p.k_edges + e) * d + j] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int o0 = wv * oq + 16 * t < (wv + 1) * oq ? wv * oq + 16 * t : wv * oq;
            const float* wr = p.w2 + (size_t)(o0 + m) * d + kk * dq;
#pragma unroll
            for (int s2 = 0; s2 < 6; ++s2) w2a[t][s2] = wr[s2 < dq ? s2 : 0];
#pragma unroll
            for (int i = 0; i < 4; ++i) b2v[t][i] = p.b2[o0 + kk * 4 + i];
        }
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < LB_NT; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) sP[(wv * 48 + mt * 16 + kk * 4 + i) * LB_NC + nt * 16 + m] = acc[mt][nt][i];
    }
    __syncthreads();
    for (int r = rs; r < d2; r += LB_RS) {
        const float v = p.b1[r] + ((sP[r * LB_NC + c] + sP[(48 + r) * LB_NC + c]) + (sP[(96 + r) * LB_NC + c] + sP[(144 + r) * LB_NC + c]));
        sD[r * LB_NC + c] = v;
        p.dist[(size_t)(e * d2 + r) * bp + col] = v;
    }
    __syncthreads();
    // ---- reparametrisation + KL (hmvae_reparam_kl_fwd's arithmetic): component j = rs, rs + LB_RS, .. ----
    const bool live = col < p.B;
    float kl = 0.f;
    float zj[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = rs + LB_RS * i;
        zj[i] = 0.f;
        if (j < d) {
            const float mu = sD[j * LB_NC + c], lv = sD[(d + j) * LB_NC + c];
            float zz = mu;
            if (p.sample && live) zz = epsv[i] * expf(0.5f * lv) + mu;
            zj[i] = live ? zz : 0.f;
            if (live) kl += -0.5f * (1.f + lv - mu * mu - expf(lv));
            p.z[(size_t)(e * d + j) * bp + col] = zj[i];
        }
    }
    const float tot = lb_block_sum(kl, sh);            // (two barriers: every thread has read its mu / logvar)
    if (threadIdx.x == 0) p.kl_partial[(size_t)e * gridDim.x + blockIdx.x] = tot;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = rs + LB_RS * i;
        if (j < d) sD[j * LB_NC + c] = zj[i];
    }
    __syncthreads();
    // ---- y = W2 z + b2: wave wv owns a quarter of the rows; reduction over d = 4 * dq: lane group kk takes dq components ----
    float zb[6][LB_NT];
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int nt = 0; nt < LB_NT; ++nt) zb[s][nt] = s < dq ? sD[(kk * dq + s) * LB_NC + nt * 16 + m] : 0.f;
    const int o_end = (wv + 1) * oq;
    for (int ob = wv * oq; ob < o_end; ob += 96) {                     // six row tiles per pass; the first pass's weights are in
        if (ob != wv * oq) {                                           // registers already (oq <= 96 in the shipped shapes)
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int o0 = ob + 16 * t < o_end ? ob + 16 * t : ob;
                const float* wr = p.w2 + (size_t)(o0 + m) * d + kk * dq;
#pragma unroll
                for (int s2 = 0; s2 < 6; ++s2) w2a[t][s2] = wr[s2 < dq ? s2 : 0];
#pragma unroll
                for (int i = 0; i < 4; ++i) b2v[t][i] = p.b2[o0 + kk * 4 + i];
            }
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int o0 = ob + 16 * t;
            if (o0 >= o_end) break;                                    // (uniform)
            f32x4 acc[LB_NT];
#pragma unroll
            for (int nt = 0; nt < LB_NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                if (s < dq) {                                          // (uniform)
#pragma unroll
                    for (int nt = 0; nt < LB_NT; ++nt) acc[nt] = LB_MFMA(w2a[t][s], zb[s][nt], acc[nt]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* yo = p.y + ((size_t)e * p.y_es + p.y_off + o0 + kk * 4 + i) * bp + c0 + m;
#pragma unroll
                for (int nt = 0; nt < LB_NT; ++nt) yo[nt * 16] = acc[nt][i] + b2v[t][i];
            }
        }
    }
}

// backward: dz (d x 64) = W2^T (d x n_out) dy (n_out x 64): the waves split n_out, partial tiles through LDS;
//           dx (n_in x 64) = W1^T (n_in x 2d) ddist (2d x 64): the waves split the rows.
__global__ __launch_bounds__(256) void latent_bridge_bwd_kernel(hmvae_bridge_bwd_desc p, int prio) {
    CHAIN_SETPRIO(prio);
    __shared__ float sP[4 * 32 * LB_NC];           // the four waves' partial dz (2 row tiles)
    __shared__ float sD[48 * LB_NC];               // ddist of the tile
    const int lane = threadIdx.x & 63;
    const int c = threadIdx.x % LB_NC, rs = threadIdx.x / LB_NC;
    const int m = lane & 15, kk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int e = blockIdx.y;
    const int c0 = blockIdx.x * LB_NC;
    const int col = c0 + c;
    const int bp = p.bp, d = p.d, d2 = 2 * p.d, n_in = p.n_in;
    const int rq = d2 >> 2;                        // <= 12
    const int kq = n_in >> 2;                      // rows of dx per wave: a multiple of 16 (host)
    constexpr int NJ = (LB_MAX_D + LB_RS - 1) / LB_RS;
    float muv[NJ], lvv[NJ], epv[NJ], w1a[6][12];
    {
        const int oq = p.n_out >> 2;               // reduction range of this wave; lane group kk takes oq / 4 of it
        const int os = oq >> 2;
        const int ob = wv * oq + kk * os;
        f32x4 acc[2][LB_NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < LB_NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* gb = p.dy + ((size_t)e * p.y_es + p.y_off + ob) * bp + c0 + m;
        const float* wb = p.w2 + (size_t)ob * d;
        const bool ok0 = m < d, ok1 = 16 + m < d;
        for (int sb = 0; sb < os; sb += 24) {                          // six rounds at a time, as in the forward kernel
            float a[6][4][2], b[6][4][LB_NT];
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int s0 = sb + 4 * it < os ? sb + 4 * it : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[it][i][0] = wb[(size_t)(s0 + i) * d + (ok0 ? m : 0)];          // A[j][o] = W2[o][j]
                    a[it][i][1] = wb[(size_t)(s0 + i) * d + (ok1 ? 16 + m : 0)];
#pragma unroll
                    for (int nt = 0; nt < LB_NT; ++nt) b[it][i][nt] = gb[(size_t)(s0 + i) * bp + nt * 16];
                }
            }
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const bool on = sb + 4 * it < os;                      // (uniform)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float a0 = (on && ok0) ? a[it][i][0] : 0.f, a1 = (on && ok1) ? a[it][i][1] : 0.f;
#pragma unroll
                    for (int nt = 0; nt < LB_NT; ++nt) {
                        acc[0][nt] = LB_MFMA(a0, b[it][i][nt], acc[0][nt]);
                        acc[1][nt] = LB_MFMA(a1, b[it][i][nt], acc[1][nt]);
                    }
                }
            }
        }
        // everything else the kernel reads from memory is requested now (see the forward kernel): this thread's (mu, logvar,
        // noise) and the encoder head's weights of this wave's first six row tiles
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int j = rs + LB_RS * i;
            const bool on = j < d && col < p.B;
            muv[i] = on ? p.dist[(size_t)(e * d2 + j) * bp + col] : 0.f;
            lvv[i] = on ? p.dist[(size_t)(e * d2 + d + j) * bp + col] : 0.f;
            epv[i] = (on && p.eps) ? p.eps[((size_t)col * p.k_edges + e) * d + j] : 0.f;
        }
        if (p.dx != nullptr) {
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int k0 = wv * kq + 16 * t < (wv + 1) * kq ? wv * kq + 16 * t : wv * kq;
#pragma unroll
                for (int s2 = 0; s2 < 12; ++s2) w1a[t][s2] = p.w1[(size_t)(kk * rq + (s2 < rq ? s2 : 0)) * n_in + k0 + m];
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < LB_NT; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) sP[(wv * 32 + mt * 16 + kk * 4 + i) * LB_NC + nt * 16 + m] = acc[mt][nt][i];
    }
    __syncthreads();
    // ---- ddist: KL gradient + reparametrisation backward (hmvae_reparam_kl_bwd's arithmetic), component j = rs, rs + LB_RS, ..
    const bool live = col < p.B;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = rs + LB_RS * i;
        if (j >= d) break;
        const float g = (sP[j * LB_NC + c] + sP[(32 + j) * LB_NC + c]) + (sP[(64 + j) * LB_NC + c] + sP[(96 + j) * LB_NC + c]);
        float gm = 0.f, gl = 0.f;
        if (live) {
            gm = p.kl_scale * muv[i] + g;
            gl = p.kl_scale * 0.5f * (expf(lvv[i]) - 1.f);
            if (p.eps) gl += g * epv[i] * 0.5f * expf(0.5f * lvv[i]);
        }
        p.ddist[(size_t)(e * d2 + j) * bp + col] = gm;
        p.ddist[(size_t)(e * d2 + d + j) * bp + col] = gl;
        sD[j * LB_NC + c] = gm;
        sD[(d + j) * LB_NC + c] = gl;
    }
    if (p.dx == nullptr) return;                                       // the caller only wants ddist
    __syncthreads();
    // ---- dx = W1^T ddist: wave wv owns a quarter of the input features; reduction over 2d = 4 * rq ----
    float gbv[12][LB_NT];
#pragma unroll
    for (int s = 0; s < 12; ++s)
#pragma unroll
        for (int nt = 0; nt < LB_NT; ++nt) gbv[s][nt] = s < rq ? sD[(kk * rq + s) * LB_NC + nt * 16 + m] : 0.f;
    const int k_end = (wv + 1) * kq;
    for (int kb = wv * kq; kb < k_end; kb += 96) {                     // six row tiles per pass; the first pass's weights are in
        if (kb != wv * kq) {                                           // registers already
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int k0 = kb + 16 * t < k_end ? kb + 16 * t : kb;
#pragma unroll
                for (int s2 = 0; s2 < 12; ++s2) w1a[t][s2] = p.w1[(size_t)(kk * rq + (s2 < rq ? s2 : 0)) * n_in + k0 + m];
            }
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int k0 = kb + 16 * t;
            if (k0 >= k_end) break;                                    // (uniform)
            f32x4 acc[LB_NT];
#pragma unroll
            for (int nt = 0; nt < LB_NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                if (s < rq) {                                          // (uniform)
#pragma unroll
                    for (int nt = 0; nt < LB_NT; ++nt) acc[nt] = LB_MFMA(w1a[t][s], gbv[s][nt], acc[nt]);   // A[k][r] = W1[r][k]
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t row = (size_t)e * p.x_es + k0 + kk * 4 + i;
#pragma unroll
                for (int nt = 0; nt < LB_NT; ++nt) {
                    float* o = p.dx + row * bp + c0 + nt * 16 + m;
                    float v = acc[nt][i];
                    if (p.flags & HMVAE_FLAG_ACCUM) v += *o;
                    if (p.flags & HMVAE_FLAG_LEAKY_GRAD) v = p.act[row * bp + c0 + nt * 16 + m] > 0.f ? v : v * p.slope;
                    *o = v;
                }
            }
        }
    }
}

}  // namespace

extern "C" int hmvae_latent_bridge_fwd(const hmvae_bridge_fwd_desc* d, hmvae_stream_t stream) {
    if (!d || !d->x || !d->w1 || !d->b1 || !d->w2 || !d->b2 || !d->dist || !d->z || !d->y || !d->kl_partial ||
        (d->sample && !d->eps))
        return HMVAE_ERR_ARG;
    if (d->k_edges <= 0 || d->bp <= 0 || (d->bp & 63) || d->B <= 0 || d->B > d->bp || d->n_out <= 0) return HMVAE_ERR_ARG;
    if (d->d <= 0 || d->d > LB_MAX_D || d->n_in <= 0 || (d->n_in & 63) || (d->n_out & 63) || (d->d & 3)) return HMVAE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(latent_bridge_fwd_kernel, dim3((unsigned)(d->bp / LB_NC), (unsigned)d->k_edges), dim3(256), 0,
                       (hipStream_t)stream, *d, chain_prio());
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}

extern "C" int hmvae_latent_bridge_bwd(const hmvae_bridge_bwd_desc* d, hmvae_stream_t stream) {
    if (!d || !d->dy || !d->w2 || !d->dist || !d->ddist || (d->dx && !d->w1)) return HMVAE_ERR_ARG;
    if (d->k_edges <= 0 || d->bp <= 0 || (d->bp & 63) || d->B <= 0 || d->B > d->bp || d->n_out <= 0) return HMVAE_ERR_ARG;
    if ((d->flags & HMVAE_FLAG_LEAKY_GRAD) && !d->act) return HMVAE_ERR_ARG;
    if (d->d <= 0 || d->d > LB_MAX_D || d->n_in <= 0 || (d->n_in & 63) || (d->n_out & 63) || (d->d & 3)) return HMVAE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(latent_bridge_bwd_kernel, dim3((unsigned)(d->bp / LB_NC), (unsigned)d->k_edges), dim3(256), 0,
                       (hipStream_t)stream, *d, chain_prio());
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}
