// microbenchmark: achievable v_mfma_f32_32x32x2_f32 rate vs #independent accumulators and waves/SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int waves_per_simd) {
    float* d; hipMalloc(&d, 256 * 1024 * 4 * 16);
    int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    int blocks_per_cu = (256 * waves_per_simd) / threads;
    int iters = 2000;
    dim3 g(256 * blocks_per_cu), b(threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, g, b, 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, g, b, 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)g.x * (threads / 64) * iters * 16.0 * NACC * 4096.0;
    printf("nacc=%d waves/SIMD=%d : %.1f TF/s\n", NACC, waves_per_simd, flops / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) { run<1>(w); run<2>(w); run<4>(w); }
    return 0;
}
