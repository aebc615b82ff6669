// fk_kernels.hip — rotation-6D lift, forward kinematics and the fused decode tail (gfx950).
//
// Reference code replaced (paths relative to the reference root):
//   my_tools.py:6-39        normalize_vector / cross_product / rotation_matrix_from_ortho6d
//   fk_layer.py:47-93       ForwardKinematicsLayer (4x4 homogeneous chain, 23 gather+bmm+copy_ steps)
//   seq_two_hier_sa_vae.py:436-474 _decode tail, :395-397/:430-434 the three MSE losses (+ autograd)
//
// The SMPL-24 kinematic tree is compile-time: the traversal is a template recursion (depth-first), so the global
// rotation of a joint lives in registers only while its subtree is being visited; nothing 4x4 or (N,24,4,4) is ever
// materialised.  A frame's walk is split BY SUBTREE over the four waves of a workgroup (fk_chain_role, fk_bwd_by_role:
// legs | left arm | right arm | head + trunk), frame-major tensors are staged through LDS in 32-frame tiles; any other
// tree (<= 64 joints, parents before children) goes through the table-driven kernels at the end of this file.
// The backward pass is fused in the same traversal: a node returns S = sum of dP over its subtree and H·R^T, the
// gradient it hands to its parent's global rotation ("FK backward recursion").
//   p_j = p_par + G_par·off_j,  G_j = G_par·R_j
//   H_j = sum_{c child of j} ( H_c·R_c^T + S_c ⊗ off_c ),   dR_j = dR_j(direct) + G_par^T·H_j
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "hmvae.h"

namespace {

constexpr int NJ = 24;
// utils/data/joint24_parents.json
__host__ __device__ constexpr int smpl_parent(int j) {
    constexpr int P[NJ] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};
    return P[j];
}
__host__ __device__ constexpr bool has_children(int j) {
    for (int c = 0; c < NJ; ++c)
        if (smpl_parent(c) == j) return true;
    return false;
}

struct V3 {
    float x, y, z;
};
struct M3 {
    float a[9];  // row-major
};

__device__ __forceinline__ V3 operator+(V3 u, V3 v) { return {u.x + v.x, u.y + v.y, u.z + v.z}; }
__device__ __forceinline__ V3 mulv(const M3& A, V3 v) {
    return {A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z, A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z,
            A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z};
}
__device__ __forceinline__ M3 mul(const M3& A, const M3& B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.a[i * 3 + j] = A.a[i * 3] * B.a[j] + A.a[i * 3 + 1] * B.a[3 + j] + A.a[i * 3 + 2] * B.a[6 + j];
    return C;
}
// A^T * B
__device__ __forceinline__ M3 mul_tn(const M3& A, const M3& B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.a[i * 3 + j] = A.a[i] * B.a[j] + A.a[3 + i] * B.a[3 + j] + A.a[6 + i] * B.a[6 + j];
    return C;
}
// A * B^T
__device__ __forceinline__ M3 mul_nt(const M3& A, const M3& B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C.a[i * 3 + j] = A.a[i * 3] * B.a[j * 3] + A.a[i * 3 + 1] * B.a[j * 3 + 1] + A.a[i * 3 + 2] * B.a[j * 3 + 2];
    return C;
}
__device__ __forceinline__ void add_outer(M3& H, V3 s, V3 o) {
    H.a[0] += s.x * o.x; H.a[1] += s.x * o.y; H.a[2] += s.x * o.z;
    H.a[3] += s.y * o.x; H.a[4] += s.y * o.y; H.a[5] += s.y * o.z;
    H.a[6] += s.z * o.x; H.a[7] += s.z * o.y; H.a[8] += s.z * o.z;
}
__device__ __forceinline__ void add_m(M3& A, const M3& B) {
#pragma unroll
    for (int i = 0; i < 9; ++i) A.a[i] += B.a[i];
}

// my_tools.py:10-17 — products rounded separately (no fma), like the reference's tensor ops
__device__ __forceinline__ V3 cross_exact(V3 u, V3 v) {
#pragma clang fp contract(off)
    V3 r;
    r.x = u.y * v.z - u.z * v.y;
    r.y = u.z * v.x - u.x * v.z;
    r.z = u.x * v.y - u.y * v.x;
    return r;
}
__device__ __forceinline__ float norm_exact(V3 v) {
#pragma clang fp contract(off)
    return sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
}

constexpr float kEps = 1e-6f;  // F.normalize(eps=1e-6), my_tools.py:8

// my_tools.py:19-39: x = n(a), z = n(x × b), y = z × x, columns of R are (x, y, z)
__device__ __forceinline__ void lift6d(const float* in, M3& R) {
    const V3 a = {in[0], in[1], in[2]}, b = {in[3], in[4], in[5]};
    const float na = fmaxf(norm_exact(a), kEps);
    const V3 x = {a.x / na, a.y / na, a.z / na};
    const V3 zr = cross_exact(x, b);
    const float nz = fmaxf(norm_exact(zr), kEps);
    const V3 z = {zr.x / nz, zr.y / nz, zr.z / nz};
    const V3 y = cross_exact(z, x);
    R.a[0] = x.x; R.a[1] = y.x; R.a[2] = z.x;
    R.a[3] = x.y; R.a[4] = y.y; R.a[5] = z.y;
    R.a[6] = x.z; R.a[7] = y.z; R.a[8] = z.z;
}

__device__ __forceinline__ V3 cross_f(V3 u, V3 v) {
    return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
__device__ __forceinline__ float dot_f(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }

// backward of lift6d: given the 6D input and dL/dR, return dL/d(in).
// (round 3) The two projections are SCALED by one reciprocal per vector instead of divided component by component (an IEEE
// fp32 division is ~11 instructions, the backward kernels run this function twice per joint and are bound by vector-
// instruction issue): a relative ulp on an output.  The unit vectors x and z themselves keep the reference's divisions:
// they feed the cancelling differences dz - z (z . dz), and on the ill-conditioned rows of the reference fixture (nearly
// parallel 6D vectors) a reciprocal-multiplied z moved the gradient by 2e-3 of its largest entry.
__device__ __forceinline__ void lift6d_bwd(const float* in, const M3& dR, float* din) {
    const V3 a = {in[0], in[1], in[2]}, b = {in[3], in[4], in[5]};
    const float na_raw = norm_exact(a);
    const float na = fmaxf(na_raw, kEps);
    const float ina = 1.f / na;
    const V3 x = {a.x / na, a.y / na, a.z / na};
    const V3 zr = cross_exact(x, b);
    const float nz_raw = norm_exact(zr);
    const float nz = fmaxf(nz_raw, kEps);
    const float inz = 1.f / nz;
    const V3 z = {zr.x / nz, zr.y / nz, zr.z / nz};
    V3 dx = {dR.a[0], dR.a[3], dR.a[6]};
    const V3 dy = {dR.a[1], dR.a[4], dR.a[7]};
    V3 dz = {dR.a[2], dR.a[5], dR.a[8]};
    // y = z × x
    dz = dz + cross_f(x, dy);
    dx = dx + cross_f(dy, z);
    // z = zr / max(|zr|, eps)
    V3 dzr;
    if (nz_raw >= kEps) {
        const float s = dot_f(z, dz);
        dzr = {(dz.x - z.x * s) * inz, (dz.y - z.y * s) * inz, (dz.z - z.z * s) * inz};
    } else {
        dzr = {dz.x * inz, dz.y * inz, dz.z * inz};
    }
    // zr = x × b
    dx = dx + cross_f(b, dzr);
    const V3 db = cross_f(dzr, x);
    // x = a / max(|a|, eps)
    V3 da;
    if (na_raw >= kEps) {
        const float s = dot_f(x, dx);
        da = {(dx.x - x.x * s) * ina, (dx.y - x.y * s) * ina, (dx.z - x.z * s) * ina};
    } else {
        da = {dx.x * ina, dx.y * ina, dx.z * ina};
    }
    din[0] = da.x; din[1] = da.y; din[2] = da.z;
    din[3] = db.x; din[4] = db.y; din[5] = db.z;
}

// ------------------------------------------------------------------------------------------------
// forward traversal
// IO policy: void rot(int j, M3&), V3 off(int j), void put_R(int j, const M3&), void put_P(int j, V3)
// ------------------------------------------------------------------------------------------------
template <int J, class IO>
__device__ __forceinline__ void fk_fwd_node(const M3& Gp, V3 Pp, IO& io);

template <int J, int C, class IO>
__device__ __forceinline__ void fk_fwd_children(const M3& G, V3 P, IO& io) {
    if constexpr (C < NJ) {
        if constexpr (smpl_parent(C) == J) fk_fwd_node<C>(G, P, io);
        fk_fwd_children<J, C + 1>(G, P, io);
    }
}

template <int J, class IO>
__device__ __forceinline__ void fk_fwd_node(const M3& Gp, V3 Pp, IO& io) {
    M3 R;
    io.rot(J, R);
    io.put_R(J, R);
    const V3 P = Pp + mulv(Gp, io.off(J));
    io.put_P(J, P);
    if constexpr (has_children(J)) {
        const M3 G = mul(Gp, R);
        fk_fwd_children<J, J + 1>(G, P, io);
    }
}

template <class IO>
__device__ __forceinline__ void fk_fwd_root(IO& io) {
    M3 R;
    io.rot(0, R);
    io.put_R(0, R);
    const V3 P = io.off(0);            // fk_layer.py:69 — the root keeps its own offset as position
    io.put_P(0, P);
    fk_fwd_children<0, 1>(R, P, io);
}

// ------------------------------------------------------------------------------------------------
// fused forward + backward traversal
// IO policy adds: V3 grad_P(int j, V3 P), void grad_R(int j, const M3& R, M3& dR), void emit(int j, const M3& dR)
// ------------------------------------------------------------------------------------------------
template <int J, class IO>
__device__ __forceinline__ void fk_bwd_node(const M3& Gp, V3 Pp, IO& io, V3& S_out, M3& HRt_out);

template <int J, int C, class IO>
__device__ __forceinline__ void fk_bwd_children(const M3& G, V3 P, IO& io, V3& S, M3& H) {
    if constexpr (C < NJ) {
        if constexpr (smpl_parent(C) == J) {
            V3 Sc;
            M3 Hc;
            fk_bwd_node<C>(G, P, io, Sc, Hc);
            S = S + Sc;
            add_m(H, Hc);
            add_outer(H, Sc, io.off(C));
        }
        fk_bwd_children<J, C + 1>(G, P, io, S, H);
    }
}

template <int J, class IO>
__device__ __forceinline__ void fk_bwd_node(const M3& Gp, V3 Pp, IO& io, V3& S_out, M3& HRt_out) {
    M3 R;
    io.rot(J, R);
    const V3 P = Pp + mulv(Gp, io.off(J));
    V3 S = io.grad_P(J, P);
    M3 dR;
    if constexpr (has_children(J)) {
        M3 H;
#pragma unroll
        for (int i = 0; i < 9; ++i) H.a[i] = 0.f;
        {
            const M3 G = mul(Gp, R);
            fk_bwd_children<J, J + 1>(G, P, io, S, H);
        }
        io.rot(J, R);                  // re-read instead of keeping 9 registers live across the subtree
        io.grad_R(J, R, dR);
        add_m(dR, mul_tn(Gp, H));
        HRt_out = mul_nt(H, R);
    } else {
        io.grad_R(J, R, dR);
#pragma unroll
        for (int i = 0; i < 9; ++i) HRt_out.a[i] = 0.f;
    }
    io.emit(J, dR);
    S_out = S;
}

template <class IO>
__device__ __forceinline__ void fk_bwd_root(IO& io) {
    M3 R;
    io.rot(0, R);
    const V3 P = io.off(0);
    V3 S = io.grad_P(0, P);            // position of the root is a constant: S is only a by-product
    M3 H;
#pragma unroll
    for (int i = 0; i < 9; ++i) H.a[i] = 0.f;
    fk_bwd_children<0, 1>(R, P, io, S, H);
    io.rot(0, R);
    M3 dR;
    io.grad_R(0, R, dR);
    add_m(dR, H);
    io.emit(0, dR);
}

// ------------------------------------------------------------------------------------------------
// AoS policies (reference layout: frame-major (N, J, 6|9|3))
// ------------------------------------------------------------------------------------------------
template <int RD>
struct AosFwd {
    const float* rin;      // this frame: J*RD floats
    const float* offs;     // J*3
    float* R_out;          // or nullptr
    float* P_out;          // or nullptr
    __device__ __forceinline__ void rot(int j, M3& R) const {
        if constexpr (RD == 6) {
            float in[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) in[k] = rin[j * 6 + k];
            lift6d(in, R);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) R.a[k] = rin[j * 9 + k];
        }
    }
    __device__ __forceinline__ V3 off(int j) const { return {offs[j * 3], offs[j * 3 + 1], offs[j * 3 + 2]}; }
    __device__ __forceinline__ void put_R(int j, const M3& R) const {
        if (R_out) {
#pragma unroll
            for (int k = 0; k < 9; ++k) R_out[j * 9 + k] = R.a[k];
        }
    }
    __device__ __forceinline__ void put_P(int j, V3 P) const {
        if (P_out) {
            P_out[j * 3] = P.x;
            P_out[j * 3 + 1] = P.y;
            P_out[j * 3 + 2] = P.z;
        }
    }
};

// ForwardKinematicsLayer on frame-major input (fk_layer.py:47-93).  A thread walking its own frame reads and writes
// 4 bytes per 64-byte line per instruction and serialises the 24-joint chain; instead 256 threads own 32 consecutive
// frames -- one contiguous piece of memory -- staged in LDS (rows padded to odd lengths: conflict-free per-frame access):
//   A  stream the rotations in with 16-byte loads (RD == 6: then lift them, three joints per thread);
//   B  the kinematic chain by subtree, one role per wave: 0 trunk + head, 1 legs, 2 / 3 arms (lanes 0..31 = frames);
//   C  stream positions (and, if asked for, the 3x3 rotations) out with 16-byte stores.
struct ChainAos {
    const float* offs;       // this frame's offsets (or the shared ones)
    const float* R;          // LDS row of this frame
    float* P;
    __device__ __forceinline__ void rot(int j, M3& Rm) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rm.a[k] = R[j * 9 + k];
    }
    __device__ __forceinline__ V3 off(int j) const { return {offs[j * 3], offs[j * 3 + 1], offs[j * 3 + 2]}; }
    __device__ __forceinline__ void put_R(int, const M3&) const {}
    __device__ __forceinline__ void put_P(int j, V3 p) const {
        P[j * 3] = p.x;
        P[j * 3 + 1] = p.y;
        P[j * 3 + 2] = p.z;
    }
};

template <class IO>
__device__ __forceinline__ void fk_chain_role(int role, IO& io) {
    M3 G0;
    io.rot(0, G0);
    const V3 P0 = io.off(0);                // fk_layer.py:69 -- the root keeps its own offset as position
    if (role == 1) {
        fk_fwd_node<1>(G0, P0, io);
        fk_fwd_node<2>(G0, P0, io);
        return;
    }
    M3 Rm;
    io.rot(3, Rm);
    const V3 P3 = P0 + mulv(G0, io.off(3));
    const M3 G3 = mul(G0, Rm);
    io.rot(6, Rm);
    const V3 P6 = P3 + mulv(G3, io.off(6));
    const M3 G6 = mul(G3, Rm);
    io.rot(9, Rm);
    const V3 P9 = P6 + mulv(G6, io.off(9));
    const M3 G9 = mul(G6, Rm);
    if (role == 0) {
        io.put_P(0, P0);
        io.put_P(3, P3);
        io.put_P(6, P6);
        io.put_P(9, P9);
        fk_fwd_node<12>(G9, P9, io);
    } else if (role == 2) {
        fk_fwd_node<13>(G9, P9, io);
    } else {
        fk_fwd_node<14>(G9, P9, io);
    }
}

constexpr int FK_FRAMES = 32;
#ifndef FK_BWD_OCC6
#define FK_BWD_OCC6 3
#endif
template <int N>   // floats per frame, a multiple of 4; LDS rows are N + 1 apart
__device__ __forceinline__ void fk_stream_in(const float* __restrict__ src, float* lds, int nf) {
    // every 16-byte load of the tile is issued before the first LDS write: a rolled loop keeps ONE load per thread in
    // flight (load, wait, 4 ds_writes, repeat), i.e. 4 KB per workgroup against the ~50 KB per CU the HBM latency needs
    constexpr int IT = (FK_FRAMES * (N / 4) + 255) / 256;
    float4 v[IT];
    const int tot = nf * (N / 4);
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < tot) v[k] = *(const float4*)(src + 4 * (size_t)i);
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < tot) {
            const int f = i / (N / 4), q = i - f * (N / 4);
            float* l = lds + f * (N + 1) + 4 * q;
            l[0] = v[k].x; l[1] = v[k].y; l[2] = v[k].z; l[3] = v[k].w;
        }
    }
}
template <int N>
__device__ __forceinline__ void fk_stream_out(const float* lds, float* __restrict__ dst, int nf) {
    for (int i = threadIdx.x; i < nf * (N / 4); i += 256) {
        const int f = i / (N / 4), q = i - f * (N / 4);
        const float* l = lds + f * (N + 1) + 4 * q;
        *(float4*)(dst + (size_t)f * N + 4 * q) = float4{l[0], l[1], l[2], l[3]};
    }
}

// 6D input of a 32-frame tile straight from global memory into the image of lifted matrices (rows LR apart); KEEP: the
// thread's inputs are also returned (the backward's phase C lifts them again), pair k of thread t = t + 256 k.
struct __attribute__((packed, aligned(4))) F3u { float x, y, z; };
template <bool WRITE>
__device__ __forceinline__ void fk_lift_in(const float* __restrict__ src, int nf, float* sR, float (*keep)[6]) {
    constexpr int LR = NJ * 9 + 1;
    constexpr int PAIRS = NJ * FK_FRAMES / 256;
    float in[PAIRS][6];
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
        const int p = threadIdx.x + 256 * k;
        if (p < nf * NJ) {
            const float2* q = (const float2*)(src + 6 * (size_t)p);
            const float2 a = q[0], b = q[1], c = q[2];
            in[k][0] = a.x; in[k][1] = a.y; in[k][2] = b.x; in[k][3] = b.y; in[k][4] = c.x; in[k][5] = c.y;
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) in[k][i] = 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
        const int p = threadIdx.x + 256 * k;
        if (keep) {
#pragma unroll
            for (int i = 0; i < 6; ++i) keep[k][i] = in[k][i];
        }
        if (WRITE && p < nf * NJ) {
            const int f = p / NJ, j = p - f * NJ;
            M3 Rm;
            lift6d(in[k], Rm);
#pragma unroll
            for (int i = 0; i < 9; ++i) sR[f * LR + j * 9 + i] = Rm.a[i];
        }
    }
}

// PF: per-frame offsets (off_stride = 72) instead of one shared set.  Either way the offsets are staged in LDS with the
// tile: read from global inside the chain they cost five dependent L2 round trips per workgroup.
template <int RD, bool PF>
__global__ __launch_bounds__(256) void fk_fwd_aos_kernel(const float* __restrict__ rot, const float* __restrict__ offs,
                                                         long long n, float* __restrict__ R_out,
                                                         float* __restrict__ P_out) {
    constexpr int LR = NJ * 9 + 1, LP = NJ * 3 + 1;
    __shared__ float sR[FK_FRAMES * LR];
    __shared__ float sP[FK_FRAMES * LP];
    __shared__ float sO[PF ? FK_FRAMES * LP : NJ * 3];
    const long long f0 = (long long)blockIdx.x * FK_FRAMES;
    const int nf = n - f0 < FK_FRAMES ? (int)(n - f0) : FK_FRAMES;
    float o_sh = 0.f;                   // issued with the tile's loads, parked in LDS after them
    if constexpr (!PF) {
        if (threadIdx.x < NJ * 3) o_sh = offs[threadIdx.x];
    }
    if constexpr (RD == 9) {
        fk_stream_in<NJ * 9>(rot + f0 * (NJ * 9), sR, nf);
    } else {
        // (round 3) A (frame, joint) pair's six floats are 24 contiguous, 8-byte aligned bytes of the frame-major input, and the
        // pairs of a workgroup's 32 frames are ONE contiguous run: thread t takes pairs t, t + 256, t + 512 with three 8-byte
        // loads each and lifts them in registers straight into the matrix image.  The round-2 form staged a 6D image in LDS
        // first (16-byte loads, four scalar LDS writes each, barrier, 18 scalar LDS reads, barrier): 57-61 us at 262 144
        // frames against 48-51 for the matrix-input variant that moves a third more bytes.
        fk_lift_in<true>(rot + f0 * (NJ * 6), nf, sR, nullptr);
    }
    if constexpr (PF) fk_stream_in<NJ * 3>(offs + f0 * (NJ * 3), sO, nf);
    else if (threadIdx.x < NJ * 3) sO[threadIdx.x] = o_sh;
    __syncthreads();
    {
        const int lane = threadIdx.x & 63;
        if (lane < nf) {
            ChainAos io{PF ? sO + lane * LP : sO, sR + lane * LR, sP + lane * LP};
            fk_chain_role(threadIdx.x >> 6, io);
        }
    }
    __syncthreads();
    if (P_out) fk_stream_out<NJ * 3>(sP, P_out + f0 * (NJ * 3), nf);
    if (R_out) fk_stream_out<NJ * 9>(sR, R_out + f0 * (NJ * 9), nf);
}

template <int RD>
struct AosBwd {
    const float* rin;
    const float* offs;
    const float* dR_up;    // J*9 or nullptr
    const float* dP_up;    // J*3 or nullptr
    float* d_rot;          // J*RD
    __device__ __forceinline__ void rot_in(int j, float* in) const {
#pragma unroll
        for (int k = 0; k < RD; ++k) in[k] = rin[j * RD + k];
    }
    __device__ __forceinline__ void rot(int j, M3& R) const {
        float in[RD];
        rot_in(j, in);
        if constexpr (RD == 6) lift6d(in, R);
        else {
#pragma unroll
            for (int k = 0; k < 9; ++k) R.a[k] = in[k];
        }
    }
    __device__ __forceinline__ V3 off(int j) const { return {offs[j * 3], offs[j * 3 + 1], offs[j * 3 + 2]}; }
    __device__ __forceinline__ V3 grad_P(int j, V3) const {
        if (dP_up) return {dP_up[j * 3], dP_up[j * 3 + 1], dP_up[j * 3 + 2]};
        return {0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void grad_R(int j, const M3&, M3& dR) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) dR.a[k] = dR_up ? dR_up[j * 9 + k] : 0.f;
    }
    __device__ __forceinline__ void emit(int j, const M3& dR) const {
        if constexpr (RD == 6) {
            float in[6], din[6];
            rot_in(j, in);
            lift6d_bwd(in, dR, din);
#pragma unroll
            for (int k = 0; k < 6; ++k) d_rot[j * 6 + k] = din[k];
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) d_rot[j * 9 + k] = dR.a[k];
        }
    }
};

// Fused forward + backward of the chain split by subtree over the four waves of a workgroup (one role per wave: no
// divergence), shared by the training tail (SoA operands, decode_tail_loss_kernel) and the frame-major backward below:
//   wave 1: the legs (subtrees of joints 1 and 2, hanging off the root)
//   wave 2 / 3: the left / right arm (subtrees of 13 / 14, hanging off joint 9)
//   wave 0: the head (subtree of 12), then -- after one barrier -- the trunk 9 -> 6 -> 3 -> 0 with the limbs' sums
// Every wave recomputes the trunk's forward chain it needs (<= 4 lifts + 3 products); a limb hands its parent
// S = sum of dP over the subtree and H = H_c R_c^T + S (x) off_c through LDS (12 floats per frame).
// Contains one __syncthreads(): every thread of the workgroup must call it (`live` masks the work, not the call).
template <int FR, class IO>
__device__ __forceinline__ void fk_bwd_by_role(int role, int lane, bool live, IO& io, float (*xch)[12][FR]) {
    auto hand_over = [&](int slot, const V3& S, const M3& H) {
        xch[slot][0][lane] = S.x;
        xch[slot][1][lane] = S.y;
        xch[slot][2][lane] = S.z;
#pragma unroll
        for (int i = 0; i < 9; ++i) xch[slot][3 + i][lane] = H.a[i];
    };
    auto limb = [&](auto child, const M3& G, V3 P, V3& S, M3& H) {      // add subtree `child` to its parent's (S, H)
        constexpr int C = decltype(child)::value;
        V3 Sc;
        M3 Hc;
        fk_bwd_node<C>(G, P, io, Sc, Hc);
        S = S + Sc;
        add_m(H, Hc);
        add_outer(H, Sc, io.off(C));
    };
    M3 G0, G3, G6, G9;
    V3 P0, P3, P6, P9;
    if (live) {
        // forward along the trunk (role 1 only needs the root)
        io.rot(0, G0);
        P0 = io.off(0);
        if (role != 1) {
            M3 R;
            io.rot(3, R);
            P3 = P0 + mulv(G0, io.off(3));
            G3 = mul(G0, R);
            io.rot(6, R);
            P6 = P3 + mulv(G3, io.off(6));
            G6 = mul(G3, R);
            io.rot(9, R);
            P9 = P6 + mulv(G6, io.off(9));
            G9 = mul(G6, R);
        }
        V3 S = {0.f, 0.f, 0.f};
        M3 H;
#pragma unroll
        for (int i = 0; i < 9; ++i) H.a[i] = 0.f;
        if (role == 1) {
            limb(std::integral_constant<int, 1>{}, G0, P0, S, H);
            limb(std::integral_constant<int, 2>{}, G0, P0, S, H);
        } else if (role == 2) {
            limb(std::integral_constant<int, 13>{}, G9, P9, S, H);
        } else if (role == 3) {
            limb(std::integral_constant<int, 14>{}, G9, P9, S, H);
        } else {
            limb(std::integral_constant<int, 12>{}, G9, P9, S, H);
        }
        hand_over(role, S, H);
    }
    __syncthreads();
    if (role == 0 && live) {
        // trunk, leaves to root: node J with the sums (S, H) over its children
        auto node = [&](int J, const M3& Gp, const M3& R, V3 P, V3& S, M3& H, bool root) {
            const V3 gp = io.grad_P(J, P);
            S = S + gp;
            M3 dR;
            io.grad_R(J, R, dR);
            add_m(dR, root ? H : mul_tn(Gp, H));
            io.emit(J, dR);
            return mul_nt(H, R);                       // H R^T: what the parent's H receives (plus S (x) off_J)
        };
        auto take = [&](int slot, V3& S, M3& H) {
            S = S + V3{xch[slot][0][lane], xch[slot][1][lane], xch[slot][2][lane]};
#pragma unroll
            for (int i = 0; i < 9; ++i) H.a[i] += xch[slot][3 + i][lane];
        };
        V3 S = {0.f, 0.f, 0.f};
        M3 H, R;
#pragma unroll
        for (int i = 0; i < 9; ++i) H.a[i] = 0.f;
        take(0, S, H);                                 // head, left arm, right arm hang off joint 9
        take(2, S, H);
        take(3, S, H);
        io.rot(9, R);
        M3 Hu = node(9, G6, R, P9, S, H, false);
        H = Hu;
        add_outer(H, S, io.off(9));
        io.rot(6, R);
        Hu = node(6, G3, R, P6, S, H, false);
        H = Hu;
        add_outer(H, S, io.off(6));
        io.rot(3, R);
        Hu = node(3, G0, R, P3, S, H, false);
        H = Hu;
        add_outer(H, S, io.off(3));
        take(1, S, H);                                 // the legs hang off the root
        io.rot(0, R);
        node(0, R, R, P0, S, H, true);
    }
}

// Backward of ForwardKinematicsLayer (+ the 6D lift) on frame-major tensors.  History of this kernel (262 144 frames,
// 6D input, 2 304 algorithmic bytes per frame):
//   1. a thread per frame straight on global memory: every load / store instruction touched 64 different 576-byte frames
//      and the chain was a 24-joint serial walk that lifted every rotation three times -- 491 us;
//   2. 32-frame LDS tiles + the subtree split of fk_bwd_by_role, lifts still inside the chain: 565 us -- not the memory
//      access but the chain's instruction count and its 273 registers (one wave per SIMD) were the limit; the same code
//      compiled for two waves per SIMD (30 spills) ran 336 us;
//   3. (this version) three phases over the LDS tile, so that the serial part is 3x3 algebra only:
//      A  stream the rotations (and dL/dR) in; RD == 6: lift them, three joints per thread, inputs kept in registers;
//      B  the chain by subtree (fk_bwd_by_role) on the LIFTED rotations; a joint's total dL/dR replaces its dL/dR row;
//      C  RD == 6: the lift's backward for three joints per thread, in parallel; stream the gradient out.
//      dL/dp is read from global memory inside the chain (three floats per joint: it does not fit the tile twice over).
// The same subtree split WITHOUT the template recursion (round 3): every limb of the SMPL-24 tree is a path, so a wave walks
// its path down once -- keeping the parents' global rotations (9 registers per level, <= 5 levels) and the position
// gradients -- and up once with the running sums (S, H): ~110 registers where the recursive walk above holds 270 (it keeps
// every level's whole frame alive), which is what lets the single-image kernels below keep their phase-A values in
// registers at two / three waves per SIMD.  Same arithmetic, same order of the sums per joint.  The policy's emit() receives
// what the chain ADDS to a joint's dL/dR (G_parent^T H_j); direct dL/dR terms are the caller's.
template <int N, class IO>
__device__ __forceinline__ void fk_path_bwd(const int (&J)[N], const M3& Gp0, V3 Pp0, IO& io, V3& S, M3& H) {
    M3 Gp[N];
    V3 dP[N];
    Gp[0] = Gp0;
    V3 P = Pp0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        P = P + mulv(Gp[i], io.off(J[i]));
        dP[i] = io.grad_P(J[i], P);
        if (i + 1 < N) {
            M3 R;
            io.rot(J[i], R);
            Gp[i + 1] = mul(Gp[i], R);
        }
        __builtin_amdgcn_sched_barrier(0);          // one joint at a time: hoisting every level's loads costs 60+ registers
    }
    V3 Sc = {0.f, 0.f, 0.f};
    M3 Hc;
#pragma unroll
    for (int k = 0; k < 9; ++k) Hc.a[k] = 0.f;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        Sc = Sc + dP[i];
        M3 R;
        io.rot(J[i], R);
        io.emit(J[i], mul_tn(Gp[i], Hc));
        M3 up = mul_nt(Hc, R);                      // what the parent's H receives: H R^T + S (x) off
        add_outer(up, Sc, io.off(J[i]));
        Hc = up;
        __builtin_amdgcn_sched_barrier(0);
    }
    S = S + Sc;
    add_m(H, Hc);
}

template <int FR, class IO>
__device__ __forceinline__ void fk_bwd_by_role2(int role, int lane, bool live, IO& io, float (*xch)[12][FR]) {
    constexpr int LEG_L[4] = {1, 4, 7, 10}, LEG_R[4] = {2, 5, 8, 11}, HEAD[2] = {12, 15};
    constexpr int ARM_L[5] = {13, 16, 18, 20, 22}, ARM_R[5] = {14, 17, 19, 21, 23};
    // the trunk's forward chain is three 3x3 products: every wave computes what its limb hangs off, and wave 0 computes it
    // AGAIN after the barrier instead of holding four matrices and four positions across its head path
    auto trunk_to_9 = [&](M3& G6o, M3& G9o, V3& P9o) {
        M3 G0, G3, R;
        io.rot(0, G0);
        const V3 P0 = io.off(0);
        io.rot(3, R);
        const V3 P3 = P0 + mulv(G0, io.off(3));
        G3 = mul(G0, R);
        io.rot(6, R);
        const V3 P6 = P3 + mulv(G3, io.off(6));
        G6o = mul(G3, R);
        io.rot(9, R);
        P9o = P6 + mulv(G6o, io.off(9));
        G9o = mul(G6o, R);
    };
    if (live) {
        V3 S = {0.f, 0.f, 0.f};
        M3 H;
#pragma unroll
        for (int i = 0; i < 9; ++i) H.a[i] = 0.f;
        if (role == 1) {
            M3 G0;
            io.rot(0, G0);
            const V3 P0 = io.off(0);
            fk_path_bwd(LEG_L, G0, P0, io, S, H);
            fk_path_bwd(LEG_R, G0, P0, io, S, H);
        } else {
            M3 G6, G9;
            V3 P9;
            trunk_to_9(G6, G9, P9);
            if (role == 2) fk_path_bwd(ARM_L, G9, P9, io, S, H);
            else if (role == 3) fk_path_bwd(ARM_R, G9, P9, io, S, H);
            else fk_path_bwd(HEAD, G9, P9, io, S, H);
        }
        xch[role][0][lane] = S.x;
        xch[role][1][lane] = S.y;
        xch[role][2][lane] = S.z;
#pragma unroll
        for (int i = 0; i < 9; ++i) xch[role][3 + i][lane] = H.a[i];
    }
    __syncthreads();
    if (role == 0 && live) {
        auto take = [&](int slot, V3& S, M3& H) {
            S = S + V3{xch[slot][0][lane], xch[slot][1][lane], xch[slot][2][lane]};
#pragma unroll
            for (int i = 0; i < 9; ++i) H.a[i] += xch[slot][3 + i][lane];
        };
        // trunk joint J (parent rotation Gp, position P): H holds the sums over its children
        auto node = [&](int J, const M3& Gp, V3 P, V3& S, M3& H, bool root) {
            S = S + io.grad_P(J, P);
            M3 R;
            io.rot(J, R);
            io.emit(J, root ? H : mul_tn(Gp, H));
            M3 up = mul_nt(H, R);
            add_outer(up, S, io.off(J));
            H = up;
        };
        M3 G0, G3, G6, R;
        io.rot(0, G0);
        const V3 P0 = io.off(0);
        io.rot(3, R);
        const V3 P3 = P0 + mulv(G0, io.off(3));
        G3 = mul(G0, R);
        io.rot(6, R);
        const V3 P6 = P3 + mulv(G3, io.off(6));
        G6 = mul(G3, R);
        const V3 P9 = P6 + mulv(G6, io.off(9));
        V3 S = {0.f, 0.f, 0.f};
        M3 H;
#pragma unroll
        for (int i = 0; i < 9; ++i) H.a[i] = 0.f;
        take(0, S, H);                                 // head, left arm, right arm hang off joint 9
        take(2, S, H);
        take(3, S, H);
        node(9, G6, P9, S, H, false);
        node(6, G3, P6, S, H, false);
        node(3, G0, P3, S, H, false);
        take(1, S, H);                                 // the legs hang off the root
        node(0, G0, P0, S, H, true);
    }
}

// Phase B policy of the single-image backward kernels (round 3): rotations come from the LDS image; what the chain ADDS to a
// joint's dL/dR (G_parent^T H_j) replaces the joint's rotation in the image once the joint is done with it -- a limb joint is
// read by its own wave only, and the trunk's rotations are read by every wave BEFORE fk_bwd_by_role's barrier and replaced by
// wave 0 after it.  The direct dL/dR stays with the threads of phases A / C.  One image instead of two: twice the workgroups
// per CU, and these kernels are chains of dependent LDS round trips that only other workgroups can hide.
template <bool PF>
struct LdsBwd1 {
    float* R;              // LDS row of this frame: lifted rotations in, induced dL/dR out (J*9)
    const float* offs;     // LDS: J*3 (shared or this frame's)
    const float* dP;       // LDS row J*3: dL/dp (zeros if absent)
    __device__ __forceinline__ void rot(int j, M3& Rm) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rm.a[k] = R[j * 9 + k];
    }
    __device__ __forceinline__ V3 off(int j) const { return {offs[j * 3], offs[j * 3 + 1], offs[j * 3 + 2]}; }
    __device__ __forceinline__ V3 grad_P(int j, V3) const { return {dP[j * 3], dP[j * 3 + 1], dP[j * 3 + 2]}; }
    __device__ __forceinline__ void grad_R(int, const M3&, M3& g) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) g.a[k] = 0.f;
    }
    __device__ __forceinline__ void emit(int j, const M3& g) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[j * 9 + k] = g.a[k];
    }
};

// Backward of ForwardKinematicsLayer (+ the 6D lift) on frame-major tensors.  History of this kernel (262 144 frames,
// 6D input, 2 304 algorithmic bytes per frame):
//   1. a thread per frame straight on global memory: every load / store instruction touched 64 different 576-byte frames
//      and the chain was a 24-joint serial walk that lifted every rotation three times -- 491 us;
//   2. 32-frame LDS tiles + the subtree split of fk_bwd_by_role, lifts still inside the chain: 565 us -- not the memory
//      access but the chain's instruction count and its 273 registers (one wave per SIMD) were the limit; the same code
//      compiled for two waves per SIMD (30 spills) ran 336 us;
//   3. three phases over the LDS tile, so that the serial part is 3x3 algebra only (round 2: 238-252 us):
//      A  the rotations in; RD == 6: lift them, three joints per thread, inputs kept in registers;
//      B  the chain by subtree (fk_bwd_by_role) on the LIFTED rotations;
//      C  RD == 6: the lift's backward for three joints per thread, in parallel; the gradient out;
//   4. (this version, round 3) ONE image: the (frame, joint) pairs of a tile are one contiguous run of the frame-major
//      tensors, so thread t reads pairs t, t + 256, t + 512 straight from global memory (24 bytes of 6D, 36 of dL/dR: kept
//      in registers, never staged), the chain leaves the induced dL/dR in the rotations' place (LdsBwd1), dL/dp is a
//      third, small image instead of stride-288 reads inside the chain, and phase C stores its 24 bytes per pair itself.
// OCC: waves per SIMD the kernel is compiled for (3: 168 registers, 16 of them spilled at RD == 6; 2: 256, none) -- both are
// instantiated for the 6D input and picked at run time (HMVAE_FK_BWD_OCC), the matrix-input variant needs 126 either way
template <int RD, bool PF, int OCC>
__global__ __launch_bounds__(256, OCC) void fk_bwd_aos_kernel(const float* __restrict__ rot, const float* __restrict__ offs,
                                                         long long n, const float* __restrict__ dR_up,
                                                         const float* __restrict__ dP_up, float* __restrict__ d_rot) {
    constexpr int LR = NJ * 9 + 1, LP = NJ * 3 + 1;
    constexpr int PAIRS = NJ * FK_FRAMES / 256;          // (frame, joint) pairs per thread in the parallel phases
    extern __shared__ __attribute__((aligned(16))) float fk_lds[];
    float* sR = fk_lds;                                  // rotations, then the induced dL/dR
    float* sP = sR + FK_FRAMES * LR;                     // dL/dp
    float* sO = sP + FK_FRAMES * LP;                     // offsets
    float (*xch)[12][FK_FRAMES] = (float (*)[12][FK_FRAMES])(sO + (PF ? FK_FRAMES * LP : NJ * 3 + 1));
    const long long f0 = (long long)blockIdx.x * FK_FRAMES;
    const int nf = n - f0 < FK_FRAMES ? (int)(n - f0) : FK_FRAMES;
    float o_sh = 0.f;
    if constexpr (!PF) {
        if (threadIdx.x < NJ * 3) o_sh = offs[threadIdx.x];
    }
    // ---- A ----
    float in[RD == 6 ? PAIRS : 1][6];
    float dd[RD == 6 ? PAIRS : 1][6];                    // the DIRECT dL/dR of this thread's pairs, already through the lift's
                                                         // backward (it is linear in dL/dR): 6 instead of 9 values across the chain
    if constexpr (RD == 6) {
        float dRd[PAIRS][9];
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) {
            const int p = threadIdx.x + 256 * k;
            if (dR_up && p < nf * NJ) {
                const F3u* q = (const F3u*)(dR_up + f0 * (NJ * 9) + 9 * (size_t)p);
                const F3u a = q[0], b = q[1], c = q[2];
                dRd[k][0] = a.x; dRd[k][1] = a.y; dRd[k][2] = a.z; dRd[k][3] = b.x; dRd[k][4] = b.y; dRd[k][5] = b.z;
                dRd[k][6] = c.x; dRd[k][7] = c.y; dRd[k][8] = c.z;
            } else {
#pragma unroll
                for (int i = 0; i < 9; ++i) dRd[k][i] = 0.f;
            }
        }
        fk_lift_in<true>(rot + f0 * (NJ * 6), nf, sR, in);
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) {
            M3 g;
#pragma unroll
            for (int i = 0; i < 9; ++i) g.a[i] = dRd[k][i];
            lift6d_bwd(in[k], g, dd[k]);
        }
    } else {
        fk_stream_in<NJ * 9>(rot + f0 * (NJ * 9), sR, nf);
    }
    if (dP_up) {
        fk_stream_in<NJ * 3>(dP_up + f0 * (NJ * 3), sP, nf);
    } else {
        for (int i = threadIdx.x; i < FK_FRAMES * LP; i += 256) sP[i] = 0.f;
    }
    if constexpr (PF) fk_stream_in<NJ * 3>(offs + f0 * (NJ * 3), sO, nf);
    else if (threadIdx.x < NJ * 3) sO[threadIdx.x] = o_sh;
    __syncthreads();
    // ---- B ----
    {
        const int lane = threadIdx.x & 63;
        const bool live = lane < nf;
        const int fl = live ? lane : 0;
        LdsBwd1<PF> io{sR + fl * LR, PF ? sO + fl * LP : sO, sP + fl * LP};
        fk_bwd_by_role2<FK_FRAMES>(threadIdx.x >> 6, fl, live, io, xch);
    }
    __syncthreads();
    // ---- C ----
    if constexpr (RD == 6) {
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) {
            const int p = threadIdx.x + 256 * k;
            if (p < nf * NJ) {
                const int f = p / NJ, j = p - f * NJ;
                M3 g;
#pragma unroll
                for (int i = 0; i < 9; ++i) g.a[i] = sR[f * LR + j * 9 + i];
                float din[6];
                lift6d_bwd(in[k], g, din);
                float2* q = (float2*)(d_rot + f0 * (NJ * 6) + 6 * (size_t)p);
                q[0] = float2{din[0] + dd[k][0], din[1] + dd[k][1]};
                q[1] = float2{din[2] + dd[k][2], din[3] + dd[k][3]};
                q[2] = float2{din[4] + dd[k][4], din[5] + dd[k][5]};
            }
        }
    } else {
        // total = upstream + induced, 16-byte pieces in the tensor's own order
        const float* up = dR_up ? dR_up + f0 * (NJ * 9) : nullptr;
        float* dst = d_rot + f0 * (NJ * 9);
        for (int i = threadIdx.x; i < nf * (NJ * 9 / 4); i += 256) {
            const int f = i / (NJ * 9 / 4), q = i - f * (NJ * 9 / 4);
            const float* l = sR + f * LR + 4 * q;
            float4 v = float4{l[0], l[1], l[2], l[3]};
            if (up) {
                const float4 u = *(const float4*)(up + (size_t)f * (NJ * 9) + 4 * q);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            *(float4*)(dst + (size_t)f * (NJ * 9) + 4 * q) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Any other kinematic tree (fk_layer.py:13-34 takes arbitrary `parents`): table-driven kernels.
// Requirements (what the reference's loop `for i in 1..J-1: G[i] = G[parents[i]] @ L[i]` needs as well): joint 0 is the
// root, parents[j] < j, at most 64 joints.  The host turns `parents` into one ancestor bit mask per joint, passed by
// value; ancestors have smaller indices than descendants and indices grow along a chain, so "bits in ascending order" is
// the chain root -> parent and "descending" is parent -> root -- no lists, no recursion, no cross-thread sums:
//   forward   p_j = off_0 + sum_k G_par(k) off_k over the chain = Horner from the leaf upward:
//             acc = off_j; for a = ancestors of j, nearest first: acc = R_a acc + off_a
//   backward  dL/dR_k = dL/dR_k(direct) + G_par(k)^T sum_{j below k} dL/dp_j (x) w_kj,   p_j - p_k = G_k w_kj,
//             w_kj = Horner from j up to (excluding) k;  j is below k  <=>  bit k of anc[j]
// One thread per (frame, joint); a workgroup stages the (lifted) rotations of 16 frames in LDS first.
// ------------------------------------------------------------------------------------------------
constexpr int GEN_MAXJ = 64, GEN_FRAMES = 16;
struct GenTree {
    int n;
    unsigned long long anc[GEN_MAXJ];
};

__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void ldm(const float* p, M3& R) {
#pragma unroll
    for (int k = 0; k < 9; ++k) R.a[k] = p[k];
}

// stage rotations (lifted if RD == 6) of the workgroup's frames: sR[f][J*9 + 1]
template <int RD>
__device__ __forceinline__ void gen_stage_R(const float* __restrict__ rot, long long f0, int nf, int J, float* sR) {
    const int LR = J * 9 + 1;
    if constexpr (RD == 9) {
        for (int i = threadIdx.x; i < nf * J * 9; i += 256) {
            const int f = i / (J * 9), r = i - f * (J * 9);
            sR[f * LR + r] = rot[(size_t)f0 * (J * 9) + i];
        }
    } else {
        for (int i = threadIdx.x; i < nf * J; i += 256) {
            const int f = i / J, j = i - f * J;
            float in[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) in[k] = rot[((size_t)(f0 + f) * J + j) * 6 + k];
            M3 R;
            lift6d(in, R);
#pragma unroll
            for (int k = 0; k < 9; ++k) sR[f * LR + j * 9 + k] = R.a[k];
        }
    }
}

template <int RD, bool PF>
__global__ __launch_bounds__(256) void fk_fwd_generic_kernel(const float* __restrict__ rot, const float* __restrict__ offs,
                                                             long long n, float* __restrict__ R_out,
                                                             float* __restrict__ P_out, GenTree tr) {
    extern __shared__ __attribute__((aligned(16))) float fk_lds[];
    const int J = tr.n, LR = J * 9 + 1;
    float* sR = fk_lds;
    const long long f0 = (long long)blockIdx.x * GEN_FRAMES;
    const int nf = n - f0 < GEN_FRAMES ? (int)(n - f0) : GEN_FRAMES;
    gen_stage_R<RD>(rot, f0, nf, J, sR);
    __syncthreads();
    for (int i = threadIdx.x; i < nf * J; i += 256) {
        const int f = i / J, j = i - f * J;
        const float* Rf = sR + f * LR;
        const float* of = offs + (PF ? (size_t)(f0 + f) * (J * 3) : 0);
        if (R_out) {
#pragma unroll
            for (int k = 0; k < 9; ++k) R_out[((size_t)(f0 + f) * J + j) * 9 + k] = Rf[j * 9 + k];
        }
        if (P_out) {
            V3 acc = ld3(of + j * 3);
            unsigned long long m = tr.anc[j];
            while (m) {                                  // nearest ancestor first
                const int a = 63 - __builtin_clzll(m);
                m &= ~(1ull << a);
                M3 R;
                ldm(Rf + a * 9, R);
                acc = mulv(R, acc) + ld3(of + a * 3);
            }
            float* po = P_out + ((size_t)(f0 + f) * J + j) * 3;
            po[0] = acc.x; po[1] = acc.y; po[2] = acc.z;
        }
    }
}

template <int RD, bool PF>
__global__ __launch_bounds__(256) void fk_bwd_generic_kernel(const float* __restrict__ rot, const float* __restrict__ offs,
                                                             long long n, const float* __restrict__ dR_up,
                                                             const float* __restrict__ dP_up, float* __restrict__ d_rot,
                                                             GenTree tr) {
    extern __shared__ __attribute__((aligned(16))) float fk_lds[];
    const int J = tr.n, LR = J * 9 + 1, LP = J * 3 + 1;
    float* sR = fk_lds;
    float* sDP = sR + GEN_FRAMES * LR;
    const long long f0 = (long long)blockIdx.x * GEN_FRAMES;
    const int nf = n - f0 < GEN_FRAMES ? (int)(n - f0) : GEN_FRAMES;
    gen_stage_R<RD>(rot, f0, nf, J, sR);
    if (dP_up) {
        for (int i = threadIdx.x; i < nf * J * 3; i += 256) {
            const int f = i / (J * 3), r = i - f * (J * 3);
            sDP[f * LP + r] = dP_up[(size_t)f0 * (J * 3) + i];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nf * J; i += 256) {
        const int f = i / J, k = i - f * J;
        const float* Rf = sR + f * LR;
        const float* of = offs + (PF ? (size_t)(f0 + f) * (J * 3) : 0);
        M3 M;
#pragma unroll
        for (int e = 0; e < 9; ++e) M.a[e] = 0.f;
        if (dP_up) {
            const unsigned long long kbit = 1ull << k, above = ~((kbit << 1) - 1ull);
            for (int j = k + 1; j < J; ++j) {
                if (!(tr.anc[j] & kbit)) continue;       // j is not below k
                V3 w = ld3(of + j * 3);
                unsigned long long m = tr.anc[j] & above;    // the ancestors of j strictly between k and j, nearest first
                while (m) {
                    const int a = 63 - __builtin_clzll(m);
                    m &= ~(1ull << a);
                    M3 R;
                    ldm(Rf + a * 9, R);
                    w = mulv(R, w) + ld3(of + a * 3);
                }
                add_outer(M, ld3(sDP + f * LP + j * 3), w);
            }
        }
        M3 dR;
#pragma unroll
        for (int e = 0; e < 9; ++e) dR.a[e] = dR_up ? dR_up[((size_t)(f0 + f) * J + k) * 9 + e] : 0.f;
        if (dP_up) {
            // G_par(k) = product of the ancestors' rotations, root first (identity for the root)
            unsigned long long m = tr.anc[k];
            if (m) {
                M3 G;
                bool first = true;
                while (m) {
                    const int a = __builtin_ctzll(m);
                    m &= m - 1;
                    M3 R;
                    ldm(Rf + a * 9, R);
                    G = first ? R : mul(G, R);
                    first = false;
                }
                add_m(dR, mul_tn(G, M));
            } else {
                add_m(dR, M);
            }
        }
        float* out = d_rot + ((size_t)(f0 + f) * J + k) * RD;
        if constexpr (RD == 6) {
            float in[6], din[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) in[e] = rot[((size_t)(f0 + f) * J + k) * 6 + e];
            lift6d_bwd(in, dR, din);
#pragma unroll
            for (int e = 0; e < 6; ++e) out[e] = din[e];
        } else {
#pragma unroll
            for (int e = 0; e < 9; ++e) out[e] = dR.a[e];
        }
    }
}

// 0 = the compile-time SMPL-24 tree, 1 = a valid generic tree (tr filled), < 0 = error code
int classify_tree(const int32_t* parents, int n_joints, GenTree* tr) {
    if (!parents || n_joints <= 0) return HMVAE_ERR_ARG;
    bool smpl = n_joints == NJ;
    for (int j = 0; smpl && j < NJ; ++j) smpl = parents[j] == smpl_parent(j);
    if (smpl) return 0;
    if (n_joints > GEN_MAXJ) return HMVAE_ERR_UNSUPPORTED;
    if (parents[0] >= 0) return HMVAE_ERR_ARG;                       // joint 0 is the root (fk_layer.py:69)
    tr->n = n_joints;
    tr->anc[0] = 0;
    for (int j = 1; j < n_joints; ++j) {
        if (parents[j] < 0 || parents[j] >= j) return HMVAE_ERR_ARG; // parents come before their children
        tr->anc[j] = tr->anc[parents[j]] | (1ull << parents[j]);
    }
    for (int j = n_joints; j < GEN_MAXJ; ++j) tr->anc[j] = 0;
    return 1;
}

}  // namespace

extern "C" int hmvae_fk_fwd(const float* rot, int32_t rot_dim, const float* offsets, int64_t offset_stride,
                            const int32_t* parents_host, int32_t n_joints, int64_t n_frames, float* R_out,
                            float* pos_out, hmvae_stream_t stream) {
    if (!rot || !offsets || n_frames <= 0 || (rot_dim != 6 && rot_dim != 9)) return HMVAE_ERR_ARG;
    if (offset_stride != 0 && offset_stride != (int64_t)n_joints * 3) return HMVAE_ERR_ARG;
    GenTree tr;
    const int kind = classify_tree(parents_host, n_joints, &tr);
    if (kind < 0) return kind;
    hipStream_t s = (hipStream_t)stream;
    const bool pf = offset_stride != 0;
    if (kind == 1) {
        dim3 ggrid((unsigned)((n_frames + GEN_FRAMES - 1) / GEN_FRAMES));
        const size_t lds = sizeof(float) * GEN_FRAMES * (n_joints * 9 + 1);
#define FK_GEN_FWD(RD_, PF_)                                                                                            \
        hipLaunchKernelGGL((fk_fwd_generic_kernel<RD_, PF_>), ggrid, dim3(256), lds, s, rot, offsets, (long long)n_frames, \
                           R_out, pos_out, tr)
        if (rot_dim == 6) { if (pf) FK_GEN_FWD(6, true); else FK_GEN_FWD(6, false); }
        else { if (pf) FK_GEN_FWD(9, true); else FK_GEN_FWD(9, false); }
#undef FK_GEN_FWD
        return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
    }
    dim3 grid((unsigned)((n_frames + FK_FRAMES - 1) / FK_FRAMES)), block(256);
#define FK_FWD_LAUNCH(RD_, PF_) \
    hipLaunchKernelGGL((fk_fwd_aos_kernel<RD_, PF_>), grid, block, 0, s, rot, offsets, (long long)n_frames, R_out, pos_out)
    if (rot_dim == 6) {
        if (pf) FK_FWD_LAUNCH(6, true);
        else FK_FWD_LAUNCH(6, false);
    } else {
        if (pf) FK_FWD_LAUNCH(9, true);
        else FK_FWD_LAUNCH(9, false);
    }
#undef FK_FWD_LAUNCH
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}

extern "C" int hmvae_fk_bwd(const float* rot, int32_t rot_dim, const float* offsets, int64_t offset_stride,
                            const int32_t* parents_host, int32_t n_joints, int64_t n_frames, const float* dR,
                            const float* dpos, float* d_rot, hmvae_stream_t stream) {
    if (!rot || !offsets || !d_rot || n_frames <= 0 || (rot_dim != 6 && rot_dim != 9)) return HMVAE_ERR_ARG;
    if (offset_stride != 0 && offset_stride != (int64_t)n_joints * 3) return HMVAE_ERR_ARG;
    GenTree tr;
    const int kind = classify_tree(parents_host, n_joints, &tr);
    if (kind < 0) return kind;
    hipStream_t s = (hipStream_t)stream;
    const bool pf = offset_stride != 0;
    if (kind == 1) {
        dim3 ggrid((unsigned)((n_frames + GEN_FRAMES - 1) / GEN_FRAMES));
        const size_t glds = sizeof(float) * GEN_FRAMES * ((n_joints * 9 + 1) + (n_joints * 3 + 1));
#define FK_GEN_BWD(RD_, PF_)                                                                                            \
        hipLaunchKernelGGL((fk_bwd_generic_kernel<RD_, PF_>), ggrid, dim3(256), glds, s, rot, offsets, (long long)n_frames, \
                           dR, dpos, d_rot, tr)
        if (rot_dim == 6) { if (pf) FK_GEN_BWD(6, true); else FK_GEN_BWD(6, false); }
        else { if (pf) FK_GEN_BWD(9, true); else FK_GEN_BWD(9, false); }
#undef FK_GEN_BWD
        return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
    }
    dim3 grid((unsigned)((n_frames + FK_FRAMES - 1) / FK_FRAMES)), block(256);
    const size_t lds = sizeof(float) * (FK_FRAMES * ((NJ * 9 + 1) + (NJ * 3 + 1)) + (pf ? FK_FRAMES * (NJ * 3 + 1) : NJ * 3 + 1) +
                                        4 * 12 * FK_FRAMES);
    static int occ6 = 0;
    if (!occ6) {
        const char* e = getenv("HMVAE_FK_BWD_OCC");
        occ6 = (e && atoi(e) == 2) ? 2 : FK_BWD_OCC6;
    }
#define FK_BWD_LAUNCH(RD_, PF_, OCC_)                                                                                  \
    {                                                                                                                  \
        static bool attr = false;                                                                                      \
        if (!attr) {                                                                                                   \
            if (hipFuncSetAttribute((const void*)fk_bwd_aos_kernel<RD_, PF_, OCC_>,                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)             \
                return HMVAE_ERR_LAUNCH;                                                                               \
            attr = true;                                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL((fk_bwd_aos_kernel<RD_, PF_, OCC_>), grid, block, lds, s, rot, offsets, (long long)n_frames, \
                           dR, dpos, d_rot);                                                                           \
    }
    if (rot_dim == 6) {
        if (occ6 == 2) {
            if (pf) FK_BWD_LAUNCH(6, true, 2)
            else FK_BWD_LAUNCH(6, false, 2)
        } else {
            if (pf) FK_BWD_LAUNCH(6, true, 3)
            else FK_BWD_LAUNCH(6, false, 3)
        }
    } else {
        if (pf) FK_BWD_LAUNCH(9, true, 3)
        else FK_BWD_LAUNCH(9, false, 3)
    }
#undef FK_BWD_LAUNCH
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// decode tail on SoA decoder output: rows (j*6+c), position t, batch column b -> ((j*6+c)*T + t)*bp + b
// ------------------------------------------------------------------------------------------------
namespace {

// MK: every joint's squared errors are weighted by mask[(j*T + t)*bp + b] (l2_masked_criterion, seq_two_hier_sa_vae.py:717-735:
// the means still divide by the full element counts).
template <bool MK>
struct TailLoss {
    const float* x;        // dec_out + t*bp + b
    const float* g6;       // gt 6d, same addressing
    const float* gR;       // gt R rows (j*9+k)
    const float* gP;       // gt pos rows (j*3+k)
    const float* mk;       // mask rows j (MK only)
    const float* offs;
    float* dx;             // d dec_out
    size_t rs;             // row stride between consecutive channels = T*bp
    float s6, sR, sP;      // sums of squared errors
    float c6, cR, cP;      // 2*weight/numel
    __device__ __forceinline__ float m(int j) const {
        if constexpr (MK) return mk[(size_t)j * rs];
        return 1.f;
    }
    __device__ __forceinline__ void rot_in(int j, float* in) const {
#pragma unroll
        for (int k = 0; k < 6; ++k) in[k] = x[(size_t)(j * 6 + k) * rs];
    }
    __device__ __forceinline__ void rot(int j, M3& R) const {
        float in[6];
        rot_in(j, in);
        lift6d(in, R);
    }
    __device__ __forceinline__ V3 off(int j) const { return {offs[j * 3], offs[j * 3 + 1], offs[j * 3 + 2]}; }
    __device__ __forceinline__ V3 grad_P(int j, V3 P) {
        const float ex = P.x - gP[(size_t)(j * 3) * rs], ey = P.y - gP[(size_t)(j * 3 + 1) * rs],
                    ez = P.z - gP[(size_t)(j * 3 + 2) * rs];
        if constexpr (MK) {
            const float w = m(j), c = cP * w;
            sP += w * (ex * ex + ey * ey + ez * ez);
            return {c * ex, c * ey, c * ez};
        }
        sP += ex * ex + ey * ey + ez * ez;
        return {cP * ex, cP * ey, cP * ez};
    }
    __device__ __forceinline__ void grad_R(int j, const M3& R, M3& dR) {
        const float w = m(j), c = MK ? cR * w : cR;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float e = R.a[k] - gR[(size_t)(j * 9 + k) * rs];
            if constexpr (MK) acc += e * e; else sR += e * e;
            dR.a[k] = c * e;
        }
        if constexpr (MK) sR += w * acc;
    }
    __device__ __forceinline__ void emit(int j, const M3& dR) {
        float in[6], din[6];
        rot_in(j, in);
        lift6d_bwd(in, dR, din);
        const float w = m(j), c = MK ? c6 * w : c6;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float e = in[k] - g6[(size_t)(j * 6 + k) * rs];
            if constexpr (MK) acc += e * e; else s6 += e * e;
            dx[(size_t)(j * 6 + k) * rs] = din[k] + c * e;
        }
        if constexpr (MK) s6 += w * acc;
    }
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// One workgroup = TL_FRAMES frames x 4 waves.  TL_FRAMES = 32 (half-empty waves, two workgroups per CU at the 16 384
// frames of the training step) was measured at 62 us against 35 us for 64: the kernel is bound by vector-instruction
// issue, not by the latency of its loads, and half-empty waves double the instructions per frame.
// A frame's work is split by SUBTREE over the waves (a wave runs one role, so
// there is no divergence and every load/store stays a 256-byte row segment): one thread per frame left 3 of 4 SIMDs
// without a wave at the 16 384 frames of the training step and ran ~120 us on the critical path.
//   wave 1: the legs (subtrees of joints 1 and 2, hanging off the root)
//   wave 2 / 3: the left / right arm (subtrees of 13 / 14, hanging off joint 9)
//   wave 0: the head (subtree of 12), then -- after one barrier -- the trunk 9 -> 6 -> 3 -> 0 with the limbs' sums
// Every wave recomputes the trunk's forward chain it needs (<= 4 lifts + 3 products); a limb hands its parent
// S = sum of dP over the subtree and H = H_c R_c^T + S (x) off_c through LDS (12 floats per frame).
constexpr int TL_FRAMES = 64;
// (348 registers, one wave per SIMD: compiled for two waves per SIMD it spills 133 of them and runs 51 instead of 31 us
// at the training step's 16 384 frames)
template <bool MK>
__global__ __launch_bounds__(256) void decode_tail_loss_kernel(const float* __restrict__ dec, const float* __restrict__ g6,
                                                               const float* __restrict__ gR, const float* __restrict__ gP,
                                                               const float* __restrict__ mask,
                                                               const float* __restrict__ offs, int T, int B, int bp,
                                                               float c6, float cR, float cP, float* __restrict__ ddec,
                                                               float* __restrict__ partial) {
    __shared__ float xch[4][12][TL_FRAMES];           // [limb][S(3) + H(9)][frame]
    __shared__ float psum[4][3];
    const int lane = threadIdx.x & 63;
    const int role = threadIdx.x >> 6;                // wave-uniform
    const int b = blockIdx.x * TL_FRAMES + (lane & (TL_FRAMES - 1));
    const int t = blockIdx.y;
    const size_t rs = (size_t)T * bp;
    const size_t o = (size_t)t * bp + b;
    const bool mine = lane < TL_FRAMES;
    const bool live = mine && b < B;
    TailLoss<MK> io{dec + o, g6 + o, gR + o, gP + o, MK ? mask + o : nullptr, offs, ddec + o, rs, 0.f, 0.f, 0.f, c6, cR, cP};
    if (!live && mine && b < bp && role == 0) {
        for (int r = 0; r < NJ * 6; ++r) ddec[o + (size_t)r * rs] = 0.f;   // padded batch columns carry zero gradient
    }
    fk_bwd_by_role<TL_FRAMES>(role, lane, live, io, xch);
    const float s6 = wave_sum_f(live ? io.s6 : 0.f), sR = wave_sum_f(live ? io.sR : 0.f), sP = wave_sum_f(live ? io.sP : 0.f);
    if (lane == 0) {
        psum[role][0] = s6;
        psum[role][1] = sR;
        psum[role][2] = sP;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float* p = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 3;
        p[threadIdx.x] = (psum[0][threadIdx.x] + psum[1][threadIdx.x]) + (psum[2][threadIdx.x] + psum[3][threadIdx.x]);
    }
}

// ---- the same tail in three phases over an LDS tile (round 2): the serial part is 3x3 algebra only ----
// decode_tail_loss_kernel above walks each subtree joint by joint and does EVERYTHING of a joint inside the walk: six row
// loads, the lift, 18 target loads, the losses, later the lift's backward -- a dependent chain per wave with one wave per SIMD
// (348 registers), 36-40 us on the training step's critical path and 0.17 of the HBM rate at 262 144 frames.  Here, as in
// fk_bwd_aos_kernel: (A) every thread lifts six joints of its frame in parallel and books the 6D / matrix errors (their sums,
// the direct dL/dR into the tile, the direct dL/d6D kept in registers); (B) the chain by subtree (fk_bwd_by_role) reads lifted
// rotations from the tile, adds the position loss and leaves every joint's TOTAL dL/dR in the tile; (C) the lift's backward
// for the same six joints in parallel, plus the direct 6D term, straight to global memory.  All global accesses are 256-byte
// row segments (lanes = consecutive batch columns).
template <bool MK>
struct TailChain {
    const float* R;        // LDS row of this frame: lifted rotations, J*9
    float* dR;             // LDS row J*9: direct dL/dR in, total dL/dR out
    const float* gP;       // gt positions + t*bp + b
    const float* mk;       // mask + t*bp + b (MK only)
    const float* offs;
    size_t rs;
    float sP, cP;
    __device__ __forceinline__ void rot(int j, M3& Rm) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rm.a[k] = R[j * 9 + k];
    }
    __device__ __forceinline__ V3 off(int j) const { return {offs[j * 3], offs[j * 3 + 1], offs[j * 3 + 2]}; }
    __device__ __forceinline__ V3 grad_P(int j, V3 P) {
        const float ex = P.x - gP[(size_t)(j * 3) * rs], ey = P.y - gP[(size_t)(j * 3 + 1) * rs],
                    ez = P.z - gP[(size_t)(j * 3 + 2) * rs];
        if constexpr (MK) {
            const float w = mk[(size_t)j * rs], c = cP * w;
            sP += w * (ex * ex + ey * ey + ez * ez);
            return {c * ex, c * ey, c * ez};
        }
        sP += ex * ex + ey * ey + ez * ez;
        return {cP * ex, cP * ey, cP * ez};
    }
    // (round 3) dR is the rotations' own row: the chain leaves what it ADDS to a joint's dL/dR where the joint's rotation was
    // (see LdsBwd1); the direct dL/dR stays in the registers of phases A / C
    __device__ __forceinline__ void grad_R(int, const M3&, M3& g) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) g.a[k] = 0.f;
    }
    __device__ __forceinline__ void emit(int j, const M3& g) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) dR[j * 9 + k] = g.a[k];
    }
};

constexpr int TL3_LR = NJ * 9 + 1;                   // odd row length: conflict-free per-frame accesses
#ifndef TL3_FR_DEFAULT
#define TL3_FR_DEFAULT 64        // measured at 262 144 frames: 239 us (64) against 307 us (32)
#endif
// FR: frames (batch columns) per workgroup.  64: full waves in the chain phase, six joints per thread in the parallel phases;
// 32: half-empty waves in the chain but three joints per thread (63 instead of 126 values held across the chain) and 34 KB of
// LDS -- three workgroups per CU whose chain phases hide behind each other's parallel phases.
template <bool MK, int FR>
__global__ __launch_bounds__(256, FR == 64 ? 2 : 3) void decode_tail_loss3_kernel(const float* __restrict__ dec, const float* __restrict__ g6,
                                                                const float* __restrict__ gR, const float* __restrict__ gP,
                                                                const float* __restrict__ mask,
                                                                const float* __restrict__ offs, int T, int B, int bp,
                                                                float c6, float cR, float cP, float* __restrict__ ddec,
                                                                float* __restrict__ partial) {
    constexpr int TL3_JT = NJ * FR / 256;             // joints per thread in the parallel phases
    constexpr int JS = 256 / FR;                      // joint stride between a thread's joints
    extern __shared__ __attribute__((aligned(16))) float tl3[];
    float* sR = tl3;                                             // [FR][TL3_LR] lifted rotations, then the induced dL/dR
    float (*xch)[12][FR] = (float (*)[12][FR])(sR + FR * TL3_LR);
    float (*psum)[3] = (float (*)[3])(&xch[4][0][0]);
    const int f = threadIdx.x & (FR - 1);
    const int jg = threadIdx.x / FR;                  // joints jg, jg + JS, ... in the parallel phases
    const int b = blockIdx.x * FR + f;
    const int t = blockIdx.y;
    const size_t rs = (size_t)T * bp;
    const size_t o = (size_t)t * bp + b;
    const bool live = b < B;
    // inputs and the DIRECT part of dL/d(6D) of this thread's joints: the 6D error plus the matrix error taken through the
    // lift's backward right here (it is linear in dL/dR) -- 12 instead of 21 values per joint held across the chain
    float in[TL3_JT][6], d6[TL3_JT][6];
    float s6 = 0.f, sRs = 0.f;
    // ---- A ----
    if (live) {
#pragma unroll
        for (int i = 0; i < TL3_JT; ++i) {
            const int j = jg + JS * i;
            M3 dRd;
            const float w = MK ? mask[o + (size_t)j * rs] : 1.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) in[i][k] = dec[o + (size_t)(j * 6 + k) * rs];
            M3 Rm;
            lift6d(in[i], Rm);
            float a6 = 0.f, aR = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float e = in[i][k] - g6[o + (size_t)(j * 6 + k) * rs];
                a6 += e * e;
                d6[i][k] = (MK ? c6 * w : c6) * e;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float e = Rm.a[k] - gR[o + (size_t)(j * 9 + k) * rs];
                aR += e * e;
                sR[f * TL3_LR + j * 9 + k] = Rm.a[k];
                dRd.a[k] = (MK ? cR * w : cR) * e;
            }
            {
                float dl[6];
                lift6d_bwd(in[i], dRd, dl);
#pragma unroll
                for (int k = 0; k < 6; ++k) d6[i][k] += dl[k];
            }
            s6 += MK ? w * a6 : a6;
            sRs += MK ? w * aR : aR;
        }
    }
    __syncthreads();
    // ---- B ---- (lanes = frames; with FR == 32 the upper half of every wave idles)
    float sPs = 0.f;
    {
        const int lane = threadIdx.x & 63;
        const int fb = lane & (FR - 1);
        const int bb = blockIdx.x * FR + fb;
        const bool lv = lane < FR && bb < B;
        const size_t ob = (size_t)t * bp + bb;
        TailChain<MK> io{sR + fb * TL3_LR, sR + fb * TL3_LR, gP + ob, MK ? mask + ob : nullptr, offs, rs, 0.f, cP};
        fk_bwd_by_role2<FR>(threadIdx.x >> 6, fb, lv, io, xch);
        sPs = lv ? io.sP : 0.f;
    }
    __syncthreads();
    // ---- C ----
    if (live) {
#pragma unroll
        for (int i = 0; i < TL3_JT; ++i) {
            const int j = jg + JS * i;
            M3 g;
#pragma unroll
            for (int k = 0; k < 9; ++k) g.a[k] = sR[f * TL3_LR + j * 9 + k];
            float din[6];
            lift6d_bwd(in[i], g, din);
#pragma unroll
            for (int k = 0; k < 6; ++k) ddec[o + (size_t)(j * 6 + k) * rs] = din[k] + d6[i][k];
        }
    } else if (b < bp) {
#pragma unroll
        for (int i = 0; i < TL3_JT; ++i)
#pragma unroll
            for (int k = 0; k < 6; ++k) ddec[o + (size_t)((jg + JS * i) * 6 + k) * rs] = 0.f;   // padded columns: zero gradient
    }
    const float w6 = wave_sum_f(s6), wR = wave_sum_f(sRs), wP = wave_sum_f(sPs);
    if ((threadIdx.x & 63) == 0) {
        psum[threadIdx.x >> 6][0] = w6;
        psum[threadIdx.x >> 6][1] = wR;
        psum[threadIdx.x >> 6][2] = wP;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float* p = partial + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 3;
        p[threadIdx.x] = (psum[0][threadIdx.x] + psum[1][threadIdx.x]) + (psum[2][threadIdx.x] + psum[3][threadIdx.x]);
    }
}

struct TailFwd {
    const float* x;
    const float* offs;
    size_t rs;
    float* c6;   // this frame: 144
    float* R;    // 216
    float* P;    // 72
    __device__ __forceinline__ void rot(int j, M3& Rm) const {
        float in[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            in[k] = x[(size_t)(j * 6 + k) * rs];
            c6[j * 6 + k] = in[k];
        }
        lift6d(in, Rm);
    }
    __device__ __forceinline__ V3 off(int j) const { return {offs[j * 3], offs[j * 3 + 1], offs[j * 3 + 2]}; }
    __device__ __forceinline__ void put_R(int j, const M3& Rm) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[j * 9 + k] = Rm.a[k];
    }
    __device__ __forceinline__ void put_P(int j, V3 p) const {
        P[j * 3] = p.x;
        P[j * 3 + 1] = p.y;
        P[j * 3 + 2] = p.z;
    }
};

// The (B, T, 24, 6|3x3|3) outputs are frame-major: a frame's 144 + 216 + 72 floats are contiguous and the frames of a
// workgroup (same t, consecutive b) lie T frames apart, so a thread storing its own frame would write 4 bytes per
// 64-byte line per instruction, and one thread per frame leaves the kinematic chain a 24-joint serial walk.  Here a
// workgroup of 256 threads owns 32 frames and works in three phases over an LDS image of them (rows padded to odd
// lengths: conflict-free per-frame accesses; 56 KB, so two workgroups share a CU):
//   A  the 32 x 24 rotation-6D lifts, three per thread, no dependencies (coalesced 128-byte row reads);
//   B  the kinematic chain by subtree: wave 0 trunk + head, 1 legs, 2 / 3 arms, rotations read back from LDS;
//   C  all threads stream the image out as whole 256-byte pieces.
struct ChainLds {                        // phase B policy: rotations come from the image, positions go to it
    const float* offs;
    const float* R;
    float* P;
    __device__ __forceinline__ void rot(int j, M3& Rm) const {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rm.a[k] = R[j * 9 + k];
    }
    __device__ __forceinline__ V3 off(int j) const { return {offs[j * 3], offs[j * 3 + 1], offs[j * 3 + 2]}; }
    __device__ __forceinline__ void put_R(int, const M3&) const {}
    __device__ __forceinline__ void put_P(int j, V3 p) const {
        P[j * 3] = p.x;
        P[j * 3 + 1] = p.y;
        P[j * 3 + 2] = p.z;
    }
};

constexpr int TF_FRAMES = 32;
// TF_GROUPS: time steps (groups of 32 frames) a workgroup walks
// C6: also write the frame-major copy of the 6D input the decode API returns (576 B/frame more to store and 18 KB more
// LDS: two instead of four workgroups per CU); callers that only want R / positions pass c6 = NULL.
template <bool C6, int TF_GROUPS, bool ALIAS = true>
__global__ __launch_bounds__(256) void decode_tail_fwd_kernel(const float* __restrict__ dec, const float* __restrict__ offs,
                                                              int T, int B, int bp, float* __restrict__ c6,
                                                              float* __restrict__ R, float* __restrict__ P) {
    constexpr int L6 = NJ * 6 + 1, LR = NJ * 9 + 1, LP = NJ * 3 + 1;
    constexpr int LIFTS = NJ * TF_FRAMES / 256;      // lifts per thread
    // (round 3) the frame-major 6D copy is streamed out of the FRONT of sR's space before the lifted matrices are written
    // there -- two more barriers per group, but 37 instead of 56 KB of LDS: four workgroups per CU like the variant without
    // the copy (72 instead of 36 KB of loads in flight per CU; bandwidth x latency asks for ~55)
    // (ALIAS = false: round 2's form with an image of its own for the copy, kept for the A/B: HMVAE_TF_ALIAS=0)
    __shared__ float sR[TF_FRAMES * LR];
    __shared__ float s6own[(C6 && !ALIAS) ? TF_FRAMES * L6 : 1];
    float* const s6 = ALIAS ? sR : s6own;
    __shared__ float sP[TF_FRAMES * LP];
    __shared__ float sO[NJ * 3];                     // offsets: LDS broadcast reads inside the chain, no global round trips
    const size_t rs = (size_t)T * bp;
    const int f = threadIdx.x & (TF_FRAMES - 1);
    const int j0 = threadIdx.x / TF_FRAMES;
    // A workgroup owns 32 sequences (b0..b0+31) and TF_GROUPS consecutive time steps.  ALL its inputs are requested up
    // front: with one group per workgroup in flight a CU holds 4 x 18 KB of loads half of the time, about half of what
    // the HBM latency needs (bandwidth x latency = ~55 KB per CU); the registers hold the later groups while the first
    // one is lifted, chained and streamed out.
    float in[TF_GROUPS][LIFTS][6];
    const int b0 = blockIdx.x * TF_FRAMES;
    const int nf = B - b0 < TF_FRAMES ? B - b0 : TF_FRAMES;          // frames of a group
    const int t0 = blockIdx.y * TF_GROUPS;
#pragma unroll
    for (int g = 0; g < TF_GROUPS; ++g) {
        if (t0 + g < T) {                            // coalesced 128-byte row pieces; b0 + f < bp always (bp % 64 == 0)
            const float* x = dec + (size_t)(t0 + g) * bp + b0 + f;
#pragma unroll
            for (int i = 0; i < LIFTS; ++i) {
                const int j = j0 + i * (256 / TF_FRAMES);
#pragma unroll
                // nontemporal: the decoder output is read exactly once (measured on one box, interleaved builds: 96-103 us
                // against 104.4-104.8 us with plain loads at 262 144 frames; nontemporal STORES of R / positions made it
                // 116 us together with these loads and 100 / 144 us (without / with the 6D copy) alone; nontemporal
                // loads in the frame-major FK kernels cost 2-5 %)
                for (int k = 0; k < 6; ++k) in[g][i][k] = __builtin_nontemporal_load(&x[(size_t)(j * 6 + k) * rs]);
            }
        }
    }
    if (threadIdx.x < NJ * 3) sO[threadIdx.x] = offs[threadIdx.x];
#pragma unroll
    for (int g = 0; g < TF_GROUPS; ++g) {
        const int t = t0 + g;
        if (t >= T) break;
        // C (first part): out[(b0 + f) * T + t][0..n): n floats per frame, consecutive threads -> consecutive 16-byte pieces
        auto stream_out = [&](auto n_c, const float* lds, float* out) {
            constexpr int n = decltype(n_c)::value;        // compile-time: the division below is a multiply
            // 16-byte stores (a quarter of the store instructions; the four LDS reads per lane are 4-way bank conflicted,
            // which is cheaper than the extra global stores: 165 vs 182 us at 262 144 frames).  Unrolled: all LDS reads of
            // a thread are issued before its first store instead of one read-wait-store round per iteration.
            constexpr int IT = (TF_FRAMES * (n / 4) + 255) / 256;
            float4 v[IT];
            const int tot = nf * (n / 4);
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int i = threadIdx.x + k * 256;
                if (i < tot) {
                    const int fr = i / (n / 4), q = i - fr * (n / 4);
                    const float* l = lds + fr * (n + 1) + 4 * q;
                    v[k] = float4{l[0], l[1], l[2], l[3]};
                }
            }
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int i = threadIdx.x + k * 256;
                if (i < tot) {
                    const int fr = i / (n / 4), q = i - fr * (n / 4);
                    *(float4*)(out + ((size_t)(b0 + fr) * T + t) * n + 4 * q) = v[k];
                }
            }
        };
        if constexpr (C6) {
#pragma unroll
            for (int i = 0; i < LIFTS; ++i) {
                const int j = j0 + i * (256 / TF_FRAMES);
#pragma unroll
                for (int k = 0; k < 6; ++k) s6[f * L6 + j * 6 + k] = in[g][i][k];
            }
            if constexpr (ALIAS) {
                __syncthreads();
                stream_out(std::integral_constant<int, NJ * 6>{}, s6, c6);
                __syncthreads();                           // the 6D image is overwritten by the matrices
            }
        }
        // A: lifts
#pragma unroll
        for (int i = 0; i < LIFTS; ++i) {
            const int j = j0 + i * (256 / TF_FRAMES);
            M3 Rm;
            lift6d(in[g][i], Rm);
#pragma unroll
            for (int k = 0; k < 9; ++k) sR[f * LR + j * 9 + k] = Rm.a[k];
        }
        __syncthreads();
        // B: chain, one subtree role per wave, lanes 0..31 = frames
        {
            const int lane = threadIdx.x & 63;
            if (lane < nf) {
                ChainLds io{sO, sR + lane * LR, sP + lane * LP};
                fk_chain_role(threadIdx.x >> 6, io);
            }
        }
        __syncthreads();
        // C: the matrices and the positions
        if constexpr (C6 && !ALIAS) stream_out(std::integral_constant<int, NJ * 6>{}, s6, c6);
        stream_out(std::integral_constant<int, NJ * 9>{}, sR, R);
        stream_out(std::integral_constant<int, NJ * 3>{}, sP, P);
        __syncthreads();                                   // the image is rewritten by the next group
    }
}

}  // namespace

static int tail_loss_launch(const float* dec_out, const float* gt6d, const float* gtR, const float* gtP, const float* mask,
                            bool masked, const float* offsets, int32_t T, int32_t B, int32_t bp, float scale_6d,
                            float scale_rot, float scale_pos, float* d_dec_out, float* partial, int32_t* n_blocks_out,
                            hmvae_stream_t stream) {
    if (T <= 0 || T > 65535 || B <= 0 || bp < B || (bp & 63)) return HMVAE_ERR_ARG;
    static int three_phase = -1, fr3 = 0;
    if (three_phase < 0) {
        const char* e = getenv("HMVAE_TAIL3");       // "0": the joint-by-joint walk (round 1), kept for the A/B
        three_phase = (e && e[0] == '0') ? 0 : 1;
        const char* f = getenv("HMVAE_TAIL3_FR");    // frames per workgroup of the three-phase kernel: 64 or 32
        fr3 = (f && (atoi(f) == 64 || atoi(f) == 32)) ? atoi(f) : TL3_FR_DEFAULT;
    }
    const int fr = three_phase ? fr3 : TL_FRAMES;
    dim3 grid(bp / fr, T), block(256);
    if (n_blocks_out) *n_blocks_out = (int32_t)(grid.x * grid.y);
    if (!dec_out && !gt6d && !gtR && !gtP && !mask && !d_dec_out && !partial) return HMVAE_OK;      // size query
    if (!dec_out || !gt6d || !gtR || !gtP || !offsets || !d_dec_out || !partial || (masked && !mask)) return HMVAE_ERR_ARG;
    if (three_phase) {
        const size_t lds = sizeof(float) * ((size_t)fr * TL3_LR + 4 * 12 * fr + 16);      // 68 KB (64 frames) / 34 KB (32)
        static bool attr[2][2] = {{false, false}, {false, false}};
#define TL3_GO(MK_, FR_)                                                                                                   \
    {                                                                                                                      \
        if (!attr[MK_][FR_ == 64]) {                                                                                       \
            if (hipFuncSetAttribute((const void*)decode_tail_loss3_kernel<MK_, FR_>,                                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)                  \
                return HMVAE_ERR_LAUNCH;                                                                                   \
            attr[MK_][FR_ == 64] = true;                                                                                   \
        }                                                                                                                  \
        hipLaunchKernelGGL((decode_tail_loss3_kernel<MK_, FR_>), grid, block, lds, (hipStream_t)stream, dec_out, gt6d, gtR, \
                           gtP, mask, offsets, T, B, bp, 2.f * scale_6d, 2.f * scale_rot, 2.f * scale_pos, d_dec_out,      \
                           partial);                                                                                       \
    }
        if (masked) {
            if (fr == 64) TL3_GO(true, 64) else TL3_GO(true, 32)
        } else {
            if (fr == 64) TL3_GO(false, 64) else TL3_GO(false, 32)
        }
#undef TL3_GO
        return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
    }
    if (masked)
        hipLaunchKernelGGL(decode_tail_loss_kernel<true>, grid, block, 0, (hipStream_t)stream, dec_out, gt6d, gtR, gtP, mask,
                           offsets, T, B, bp, 2.f * scale_6d, 2.f * scale_rot, 2.f * scale_pos, d_dec_out, partial);
    else
        hipLaunchKernelGGL(decode_tail_loss_kernel<false>, grid, block, 0, (hipStream_t)stream, dec_out, gt6d, gtR, gtP, mask,
                           offsets, T, B, bp, 2.f * scale_6d, 2.f * scale_rot, 2.f * scale_pos, d_dec_out, partial);
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}

extern "C" int hmvae_decode_tail_loss(const float* dec_out, const float* gt6d, const float* gtR, const float* gtP,
                                      const float* offsets, int32_t T, int32_t B, int32_t bp, float scale_6d,
                                      float scale_rot, float scale_pos, float* d_dec_out, float* partial,
                                      int32_t* n_blocks_out, hmvae_stream_t stream) {
    return tail_loss_launch(dec_out, gt6d, gtR, gtP, nullptr, false, offsets, T, B, bp, scale_6d, scale_rot, scale_pos,
                            d_dec_out, partial, n_blocks_out, stream);
}

extern "C" int hmvae_decode_tail_loss_masked(const float* dec_out, const float* gt6d, const float* gtR, const float* gtP,
                                             const float* mask, const float* offsets, int32_t T, int32_t B, int32_t bp,
                                             float scale_6d, float scale_rot, float scale_pos, float* d_dec_out,
                                             float* partial, int32_t* n_blocks_out, hmvae_stream_t stream) {
    return tail_loss_launch(dec_out, gt6d, gtR, gtP, mask, true, offsets, T, B, bp, scale_6d, scale_rot, scale_pos, d_dec_out,
                            partial, n_blocks_out, stream);
}

extern "C" int hmvae_decode_tail_fwd(const float* dec_out, const float* offsets, int32_t T, int32_t B, int32_t bp,
                                     float* c6, float* R, float* pos, hmvae_stream_t stream) {
    if (!dec_out || !offsets || !R || !pos) return HMVAE_ERR_ARG;
    if (T <= 0 || T > 65535 || B <= 0 || bp < B || (bp & 63)) return HMVAE_ERR_ARG;
    static int groups = 0;
    if (!groups) {
        const char* e = getenv("HMVAE_TF_GROUPS");
        groups = e ? atoi(e) : 1;        // measured at 262 144 frames: 104 / 107 / 113 / 123 us for 1 / 2 / 4 / 8
        if (groups != 1 && groups != 2 && groups != 4) groups = 1;
    }
    dim3 block(256);
#define TF_LAUNCH(C6_, G_)                                                                                             \
    hipLaunchKernelGGL((decode_tail_fwd_kernel<C6_, G_>), dim3((B + TF_FRAMES - 1) / TF_FRAMES, (T + G_ - 1) / G_), block, \
                       0, (hipStream_t)stream, dec_out, offsets, T, B, bp, c6, R, pos)
#define TF_PICK(C6_)                              \
    switch (groups) {                             \
        case 1: TF_LAUNCH(C6_, 1); break;         \
        case 2: TF_LAUNCH(C6_, 2); break;         \
        case 4: TF_LAUNCH(C6_, 4); break;         \
        default: TF_LAUNCH(C6_, 1); break;        \
    }
    static int alias = -1;
    if (alias < 0) {
        const char* e = getenv("HMVAE_TF_ALIAS");
        alias = (e && e[0] == '0') ? 0 : 1;
    }
    if (c6 && !alias && groups == 1) {
        hipLaunchKernelGGL((decode_tail_fwd_kernel<true, 1, false>), dim3((B + TF_FRAMES - 1) / TF_FRAMES, T), block, 0,
                           (hipStream_t)stream, dec_out, offsets, T, B, bp, c6, R, pos);
    } else if (c6) {
        TF_PICK(true)
    } else {
        TF_PICK(false)
    }
#undef TF_PICK
#undef TF_LAUNCH
    return hipGetLastError() == hipSuccess ? HMVAE_OK : HMVAE_ERR_LAUNCH;
}
