// microbenchmark: where does a 64x64x32 fp32-MFMA k-tile loop lose time?  Adds the GEMM's ingredients one at a time:
//   F_LDSR  ds_read_b128 fragments (8 per wave per tile)      F_BAR   one barrier per tile
//   F_LDSW  2 ds_write_b128 per thread per tile               F_GLD   2 global float4 loads per thread per tile (L2 resident)
#include <hip/hip_runtime.h>
#include <stdio.h>
// build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_lds.hip -o tools/mfma_lds.bin ; run: ./tools/mfma_lds.bin
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { F_LDSR = 1, F_BAR = 2, F_LDSW = 4, F_GLD = 8, F_VALU = 16, F_SALU = 32 };   // F_VALU: +80 VALU ops / tile, F_SALU: +40 SALU ops / tile
template <int FLAGS, int NACC>
__global__ __launch_bounds__(256) void k(const float* __restrict__ g, float* out, int iters) {
    __shared__ f32x4 lds[2][2][64 * 9];      // [buf][A|B][row*9 + kq]  (36-float row stride)
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = (lane & 31) + 32 * (wave & 1), kh = lane >> 5;
    for (int i = threadIdx.x; i < 2 * 2 * 64 * 9; i += 256) (&lds[0][0][0])[i] = f32x4{1e-3f, 2e-3f, 1e-3f, 0.f};
    __syncthreads();
    const f32x4* gp = reinterpret_cast<const f32x4*>(g) + blockIdx.x * 512 + threadIdx.x;
    f32x4 ga{0, 0, 0, 0}, gb{0, 0, 0, 0};
    int buf = 0, dummy = lane, sdummy = iters;
    for (int it = 0; it < iters; ++it) {
        if (FLAGS & F_GLD) { ga = gp[(it & 63) * 8192]; gb = gp[(it & 63) * 8192 + 256]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a, b;
            if (FLAGS & F_LDSR) {
                a = lds[buf][0][row * 9 + q * 2 + kh];
                b = lds[buf][1][((lane & 31) + 32 * (wave >> 1)) * 9 + q * 2 + kh];
            } else {
                a = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f}; b = a;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[i], 0, 0, 0);
            if (FLAGS & F_VALU) {
#pragma unroll
                for (int v = 0; v < 20; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(dummy) : "v"(lane));
            }
            if (FLAGS & F_SALU) {
#pragma unroll
                for (int v = 0; v < 10; ++v) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sdummy) : : "scc");
            }
        }
        if (FLAGS & F_LDSW) {
            lds[buf ^ 1][0][(threadIdx.x >> 2) * 9 + (threadIdx.x & 3) * 2] = ga;
            lds[buf ^ 1][1][(threadIdx.x >> 2) * 9 + (threadIdx.x & 3) * 2 + 1] = gb;
        }
        if (FLAGS & F_BAR) __syncthreads();
        if (FLAGS & (F_LDSW | F_BAR)) buf ^= 1;
    }
    float s = ga[0] + gb[1] + dummy + sdummy;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int FLAGS, int NACC> void run(int blocks_per_cu, const float* g, float* d) {
    int iters = 1000;
    dim3 grid(256 * blocks_per_cu), b(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FLAGS, NACC>), grid, b, 0, 0, g, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FLAGS, NACC>), grid, b, 0, 0, g, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid.x * 4 * iters * 16.0 * NACC * 4096.0;
    printf("flags=%2d nacc=%d blocks/CU=%d : %.1f TF/s\n", FLAGS, NACC, blocks_per_cu, flops / (ms * 1e-3) / 1e12);
}
template <int NACC> void sweep(const float* g, float* d) {
    for (int bpc : {4}) {
        run<0, NACC>(bpc, g, d);
        run<F_LDSR, NACC>(bpc, g, d);
        run<F_LDSR | F_BAR, NACC>(bpc, g, d);
        run<F_LDSR | F_BAR | F_LDSW, NACC>(bpc, g, d);
        run<F_LDSR | F_BAR | F_LDSW | F_GLD, NACC>(bpc, g, d);
        run<F_LDSR | F_BAR | F_LDSW | F_GLD | F_VALU, NACC>(bpc, g, d);
        run<F_LDSR | F_BAR | F_LDSW | F_GLD | F_VALU | F_SALU, NACC>(bpc, g, d);
        run<F_VALU, NACC>(bpc, g, d);
    }
}
int main() {
    float *g, *d;
    hipMalloc(&g, (size_t)64 * 8192 * 16 + 2048 * 512 * 16 + (1 << 20)); hipMalloc(&d, 256 * 8 * 256 * 4);
    hipMemset(g, 0, (size_t)64 * 8192 * 16 + 2048 * 512 * 16);
    sweep<1>(g, d);
    sweep<4>(g, d);
    return 0;
}
