// Dev tool (GPU box): the v3 compute loop in isolation -- LDS-fed fp32 MFMAs, with/without a barrier per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
template <int UN, int MODE>   // MODE bit0: barrier per step, bit1: global loads + LDS stores per step (3 x 16 B per thread)
__global__ __launch_bounds__(512) void loop(float* out, const float* src, int steps, int W, int nchunk, const int* items) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, kk = lane >> 4, n = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 512) ((float*)sm)[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    const int b_bytes = 4 * W * 256;
    const int buf_bytes = b_bytes + 16 * 256;
    const int lane_b = kk * W * 256 + n * 16, lane_a = b_bytes + lane * 4;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 rb[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float* gp = src + ((size_t)blockIdx.x * 512 + tid) * 4;
    unsigned iw[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) iw[i] = (unsigned)items[wid * 32 + i] << 8;
    for (int st = 0; st < steps; ++st) {
        const char* buf = sm + (st & 1) * buf_bytes;
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) rb[i] = *(const f32x4*)(gp + (size_t)((st * 3 + i) & 63) * 512 * 1024);
        }
        f32x4 b[2][UN]; float a[2][UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            if (MODE & 4) {
                b[0][k] = *(const f32x4*)(buf + (iw[k] & 0xffffu) + lane_b);
                a[0][k] = *(const float*)(buf + (iw[k] >> 16) + lane_a);
            } else {
                b[0][k] = *(const f32x4*)(buf + (wid + k) * 256 + lane_b);
                a[0][k] = *(const float*)(buf + k * 256 + lane_a);
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            if (ch < nchunk) {
                if (ch + 1 < nchunk) {
#pragma unroll
                    for (int k = 0; k < UN; ++k) {
                        if (MODE & 4) {
                            b[(ch + 1) & 1][k] = *(const f32x4*)(buf + (iw[(ch + 1) * UN + k] & 0xffffu) + lane_b);
                            a[(ch + 1) & 1][k] = *(const float*)(buf + (iw[(ch + 1) * UN + k] >> 16) + lane_a);
                        } else {
                            b[(ch + 1) & 1][k] = *(const f32x4*)(buf + (wid + (ch + 1) * UN + k) * 256 + lane_b);
                            a[(ch + 1) & 1][k] = *(const float*)(buf + ((ch + 1) * UN + k) * 256 + lane_a);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < UN; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = MFMA16(a[ch & 1][k], b[ch & 1][k][j], acc[j]);
            }
        }
        if (MODE & 2) {
            char* nb = sm + ((st + 1) & 1) * buf_bytes;
#pragma unroll
            for (int i = 0; i < 3; ++i) *(f32x4*)(nb + (i * 512 + tid) * 16) = rb[i] + 1.0f;
        }
        if (MODE & 1) __syncthreads();
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}
template <int MODE>
void run(const char* name, float* out, const float* src, int nb, const int* items) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 2000, W = 22;
    const size_t lds = 2 * (4 * W * 256 + 16 * 256) + 8192;
    hipFuncSetAttribute((const void*)loop<5, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<5, MODE>), dim3(nb), dim3(512), lds, 0, out, src, steps, W, 3, items);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)nb * 8 * steps * 60 * 2048.0;
        if (rep == 2) printf("%-40s blocks %d: %.3f ms  %.1f TF\n", name, nb, ms, flops / ms / 1e9);
    }
}
int main() {
    float *out, *src;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&src, (size_t)64 * 512 * 1024 * 4 + (size_t)1024 * 512 * 16);
    hipMemset(src, 0, (size_t)64 * 512 * 1024 * 4 + (size_t)1024 * 512 * 16);
    int h[8 * 32]; int* items;
    for (int w = 0; w < 8; ++w) for (int k = 0; k < 32; ++k) h[w * 32 + k] = (w + (k % 15)) | ((k % 15) << 16);
    hipMalloc(&items, sizeof(h)); hipMemcpy(items, h, sizeof(h), hipMemcpyHostToDevice);
    for (int nb : {512}) {
        run<0>("lds-fed mfma (imm offsets)", out, src, nb, items);
        run<3>("imm + loader + barrier", out, src, nb, items);
        run<4>("sgpr offsets", out, src, nb, items);
        run<7>("sgpr offsets + loader + barrier", out, src, nb, items);
    }
    return 0;
}
