// probe of ds_read_b64_tr_b16 lane semantics on gfx950: LDS holds u16 value = its own element index; every lane passes an address
// and prints what it got.  hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // image: [k rows][64 cols] u16 (row stride 128 B).  lane i of a 16-lane group: row i/4 (+ 4*mode), cols 4*(i%4) .. +3, group g: cols += 16 g
    const int g = l / 16, i = l % 16;
    const int row = i / 4, col = 4 * (i % 4) + 16 * g;
    const unsigned addr = (unsigned)((row * 64 + col) * 2);
    typedef __attribute__((address_space(3))) v4s* lp;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)((__attribute__((address_space(3))) char*)lds + addr));
    out[l * 4 + 0] = (unsigned short)r[0]; out[l * 4 + 1] = (unsigned short)r[1];
    out[l * 4 + 2] = (unsigned short)r[2]; out[l * 4 + 3] = (unsigned short)r[3];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 4 * 4);
    k<<<1, 64>>>(d, 0);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    unsigned h[256]; e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("copy: %s\n", hipGetErrorString(e));
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf(" (r%u,c%u)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
        printf("\n");
    }
    return 0;
}
