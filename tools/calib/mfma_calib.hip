// Dev tool (GPU box): bare fp32-MFMA rate and s_memtime tick rate, to tell clock effects from kernel effects.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mfma_loop(float* out, unsigned long long* stamps, int iters, int lds_reads) {
    extern __shared__ float sm[];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) sm[i] = i * 0.5f;
    __syncthreads();
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (lds_reads) {
            f32x4 v = *(const f32x4*)(sm + ((threadIdx.x * 4 + it * 64) & 8191 & ~3));
            b = v[0];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}
int main() {
    float* out; unsigned long long* st;
    const int nb = 512;
    hipMalloc(&out, nb * 512 * 4); hipMalloc(&st, nb * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int iters : {2000, 20000, 200000}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_loop, dim3(nb), dim3(512), 32768, 0, out, st, iters, mode);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
                double flops = (double)nb * 8 * iters * 16 * 2048.0;
                printf("lds_reads=%d iters=%d: %.3f ms  %.1f TF  s_memtime ticks %.0f -> %.1f MHz tick rate (if the wave spans the kernel)\n",
                       mode, iters, ms, flops / ms / 1e9, (double)(h[1] - h[0]), (double)(h[1] - h[0]) / ms / 1e3);
            }
        }
    return 0;
}
