// Stand-alone development harness for the fp16-pair plane GEMM (NT form): operands are two fp16 planes [2][rows][ld]
// (x = hi + lo, hi = fp16(x), lo = fp16(x - hi), values pre-scaled by a power of two so that lo stays a normal fp16);
// a product is lo*hi + hi*lo + hi*hi (smallest first) as v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 pair_gemm_probe.hip -o pair_gemm_probe ; run on the GPU box.
// Prints per shape and tile: us, TF/s fp32-equivalent (2 M N K / t), and for the load-only / mma-only variants what each side costs.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned pg_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pg_f16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned PG_INVALID = 0x80000000u;

struct desc {
    const unsigned short* A; const unsigned short* B; float* C;
    long long a_plane, b_plane, lda, ldb, ldc;
    int M, N, K;
    float alpha;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pg_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
}

// MODE bit 0: do the DMA loads, bit 1: do the fragment reads + MFMAs (3 = the real kernel)
template <int BM, int BN, int WGM, int WGN, int NST, int MODE>
__global__ __launch_bounds__(WGM * WGN * 64) void pair_gemm_kernel(const desc p) {
    constexpr bool LOADS = MODE & 1, MMAS = MODE & 2;
    constexpr int NW = WGM * WGN, NT = NW * 64;
    constexpr int BK = 32, NP = 2;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM * 64, PB = BN * 64;
    constexpr int STAGE = NP * (PA + PB);
    constexpr int NIA = NP * BM / 16 / NW, NIB = NP * BN / 16 / NW;
    static_assert((NP * BM / 16) % NW == 0 && (NP * BN / 16) % NW == 0, "DMA units must divide over the waves");
    static_assert(NST == 2 || NST == 3, "two or three LDS stages");
    constexpr int CTS = BN + 4;
    constexpr int EROWS = (BM * CTS * 4 <= 160 * 1024 - 1024) ? BM : 64;          // rows staged per epilogue pass
    constexpr int SMEM = (NST * STAGE > EROWS * CTS * 4) ? NST * STAGE : EROWS * CTS * 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = p.M, N = p.N;
    const long long a_plane = p.a_plane, b_plane = p.b_plane, lda = p.lda, ldb = p.ldb;
    constexpr unsigned XCDS = 8, XCD_GROUP = 8;
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned total = gx * gy;
    const unsigned xcd = lin % XCDS, local = lin / XCDS;
    const unsigned per_xcd = (total + XCDS - 1) / XCDS, tall = (total % XCDS) ? (total % XCDS) : XCDS;
    const unsigned rem = xcd < tall ? xcd * per_xcd + local : tall * per_xcd + (xcd - tall) * (per_xcd - 1) + local;
    const unsigned band = XCD_GROUP * gy, bid = rem / band, first = bid * XCD_GROUP;
    const unsigned bm = min(gx - first, XCD_GROUP), inb = rem - bid * band;
    const unsigned tile_m = first + inb % bm, tile_n = inb / bm;
    const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
    if (m0 >= M || n0 >= N) return;
    const int ntiles = p.K / BK;

    unsigned avo[NIA], bvo[NIB];
    int alds[NIA], blds[NIB];
    {
        const int lrow = lane >> 2;
        const int lchunk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int u = wave + NW * i, q = u / (BM / 16), rb = u % (BM / 16);
            const int r = rb * 16 + lrow;
            avo[i] = (m0 + r < M) ? (unsigned)(((long long)q * a_plane + (long long)r * lda) * 2 + lchunk * 16) : PG_INVALID;
            alds[i] = __builtin_amdgcn_readfirstlane(q * PA + rb * 1024);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const int u = wave + NW * i, q = u / (BN / 16), rb = u % (BN / 16);
            const int r = rb * 16 + lrow;
            bvo[i] = (n0 + r < N) ? (unsigned)(((long long)q * b_plane + (long long)r * ldb) * 2 + lchunk * 16) : PG_INVALID;
            blds[i] = __builtin_amdgcn_readfirstlane(NP * PA + q * PB + rb * 1024);
        }
    }
    const unsigned short* abase = p.A + (long long)m0 * lda;
    const unsigned short* bbase = p.B + (long long)n0 * ldb;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue = [&](int stage, unsigned inv) {
        if constexpr (LOADS) {
            unsigned char* sb = smem + stage * STAGE;
            const __amdgpu_buffer_rsrc_t ra = pg_rsrc(abase), rb = pg_rsrc(bbase);
#pragma unroll
            for (int i = 0; i < NIA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb + alds[i]), 16, (int)(avo[i] | inv), 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NIB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + blds[i]), 16, (int)(bvo[i] | inv), 0, 0, 0);
            abase += BK;
            bbase += BK;
        }
    };

    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 31, lk = lane >> 5;
    const int sw = (lr >> 2) & 3;
    const int fo0 = lr * 64 + (((0 + lk) ^ sw) << 4), fo1 = lr * 64 + (((2 + lk) ^ sw) << 4);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    pg_u32x4 fa0[NP][TM], fb0[NP][TN], fa1[NP][TM], fb1[NP][TN];
    auto read_frags = [&](int stage, int fo, pg_u32x4 (&fa)[NP][TM], pg_u32x4 (&fb)[NP][TN]) {
        if constexpr (MMAS) {
            const unsigned char* as = smem + stage * STAGE + (wm * WM) * 64 + fo;
            const unsigned char* bs = smem + stage * STAGE + NP * PA + (wn * WN) * 64 + fo;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[q][i] = *reinterpret_cast<const pg_u32x4*>(as + q * PA + i * 32 * 64);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[q][j] = *reinterpret_cast<const pg_u32x4*>(bs + q * PB + j * 32 * 64);
            }
        }
    };
    // piece products, smallest first: (lo,hi) (hi,lo) (hi,hi)
    auto mma = [&](const pg_u32x4 (&fa)[NP][TM], const pg_u32x4 (&fb)[NP][TN]) {
        if constexpr (MMAS) {
            constexpr int qa[3] = {1, 0, 0}, qb[3] = {0, 1, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pg_f16x8, fa[qa[t]][i]),
                                                                           __builtin_bit_cast(pg_f16x8, fb[qb[t]][j]), acc[i][j], 0, 0, 0);
        }
    };

    constexpr int NIW = LOADS ? NIA + NIB : 0;
    constexpr int NMMA = MMAS ? 3 * TM * TN : 0, NRD = MMAS ? NP * (TM + TN) : 0;
    issue(0, 0u);
    if constexpr (NST == 3) {
        issue(1, ntiles > 1 ? 0u : PG_INVALID);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    auto settle0 = [&]() {
        if constexpr (MMAS) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa0[q][i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb0[q][j]));
            }
        }
    };
    read_frags(0, fo0, fa0, fb0);
    settle0();
    int cur = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int nxt = (cur + 1 == NST) ? 0 : cur + 1;
        const int nn = (nxt + 1 == NST) ? 0 : nxt + 1;
        __builtin_amdgcn_sched_barrier(0);
        issue(NST == 3 ? nn : nxt, t + NST - 1 < ntiles ? 0u : PG_INVALID);
        read_frags(cur, fo1, fa1, fb1);
        mma(fa0, fb0);
#pragma unroll
        for (int g = 0; g < NMMA; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (g < NIW) { __builtin_amdgcn_sched_group_barrier(0x004, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NST == 3) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NIW) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nxt, fo0, fa0, fb0);
            mma(fa1, fb1);
#pragma unroll
            for (int g = 0; g < NMMA; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
            mma(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nxt, fo0, fa0, fb0);
        }
        __builtin_amdgcn_sched_barrier(0);
        settle0();
        cur = nxt;
    }
    __syncthreads();

    // ---- epilogue: passes of EROWS rows through LDS, float4 row pieces ----
    const float alpha = p.alpha;
    const long long ldc = p.ldc;
    float* const C = p.C;
    float* const Ct = reinterpret_cast<float*>(smem);
    constexpr int QN = BN / 4;
#pragma unroll
    for (int ps = 0; ps < BM / EROWS; ++ps) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rb = wm * WM + i * 32 - ps * EROWS;          // first row of the block inside this pass
            if (rb >= 0 && rb < EROWS) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Ct[(rb + (r & 3) + 8 * (r >> 2) + 4 * lk) * CTS + wn * WN + j * 32 + lr] = acc[i][j][r] * alpha;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EROWS * QN / NT; ++q) {
            const int idx = tid + q * NT;
            const int row = idx / QN, c = (idx % QN) * 4;
            const int gm = m0 + ps * EROWS + row, gn = n0 + c;
            if (gm >= M || gn >= N) continue;
            const float4 v = *reinterpret_cast<const float4*>(&Ct[row * CTS + c]);
            *reinterpret_cast<float4*>(C + (long long)gm * ldc + gn) = v;
        }
        if (ps + 1 < BM / EROWS) __syncthreads();
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static unsigned short f2h(float x) { _Float16 h = (_Float16)x; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; __builtin_memcpy(&h, &u, 2); return (float)h; }

struct Problem { const char* name; int M, N, K; };

template <int BM, int BN, int WGM, int WGN, int NST, int MODE>
static float run(const desc& d, int iters) {
    dim3 g((d.M + BM - 1) / BM, (d.N + BN - 1) / BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pair_gemm_kernel<BM, BN, WGM, WGN, NST, MODE>), g, dim3(WGM * WGN * 64), 0, 0, d);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((pair_gemm_kernel<BM, BN, WGM, WGN, NST, MODE>), g, dim3(WGM * WGN * 64), 0, 0, d);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    const Problem probs[] = {
        {"QKV fwd      ", 4128, 2304, 768}, {"attn-out fwd ", 4128, 768, 768}, {"FFN1 fwd     ", 4128, 3072, 768},
        {"FFN2 fwd     ", 4128, 768, 3072}, {"QKV dgrad    ", 4128, 768, 2304}, {"QKV fwd 4096 ", 4096, 2304, 768},
        {"FFN1 4096    ", 4096, 3072, 768}, {"out 4096     ", 4096, 768, 768}, {"big 8192^2x1k", 8192, 8192, 1024},
    };
    const int iters = 20;
    for (const Problem& pr : probs) {
        const int M = pr.M, N = pr.N, K = pr.K;
        std::vector<unsigned short> ha((size_t)2 * M * K), hb((size_t)2 * N * K);
        srand(1234);
        auto fill = [&](std::vector<unsigned short>& v, int rows, float scale) {
            for (size_t i = 0; i < (size_t)rows * K; ++i) {
                const float x = scale * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
                const unsigned short h = f2h(x);
                v[i] = h;
                v[(size_t)rows * K + i] = f2h(x - h2f(h));
            }
        };
        fill(ha, M, 16.f);
        fill(hb, N, 8.f);
        unsigned short *dA, *dB; float* dC;
        CK(hipMalloc(&dA, ha.size() * 2)); CK(hipMalloc(&dB, hb.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
        desc d{dA, dB, dC, (long long)M * K, (long long)N * K, K, K, N, M, N, K, 1.0f};
        std::vector<float> hc((size_t)M * N);
        auto check = [&](const char* tag) {
            CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0;
            for (int s = 0; s < 64; ++s) {
                const int m = (s * 977 + (s % 3 == 0 ? M - 1 - s : 0)) % M, n = (s * 613 + 5) % N;
                double ref = 0, mag = 0;
                for (int k = 0; k < K; ++k) {
                    const double ah = h2f(ha[(size_t)m * K + k]), al = h2f(ha[(size_t)M * K + (size_t)m * K + k]);
                    const double bh = h2f(hb[(size_t)n * K + k]), bl = h2f(hb[(size_t)N * K + (size_t)n * K + k]);
                    ref += ah * bh + ah * bl + al * bh;
                    mag += fabs(ah * bh);
                }
                const double e = fabs(ref - hc[(size_t)m * N + n]) / mag;
                if (e > worst) worst = e;
            }
            if (worst > 2e-6) printf("   !! %s mismatch: err/mag %.3g\n", tag, worst);
            return worst;
        };
        const double fl = 2.0 * M * N * K;
        printf("%s %dx%dx%d\n", pr.name, M, N, K);
#define RUNCFG(BM, BN, WGM, WGN, NST)                                                                                              \
        {                                                                                                                          \
            CK(hipMemset(dC, 0, (size_t)M * N * 4));                                                                               \
            const float t3 = run<BM, BN, WGM, WGN, NST, 3>(d, iters);                                                              \
            const double err = check(#BM "x" #BN);                                                                                 \
            const float t1 = run<BM, BN, WGM, WGN, NST, 1>(d, iters);                                                              \
            const float t2 = run<BM, BN, WGM, WGN, NST, 2>(d, iters);                                                              \
            const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);                                                    \
            const double bytes = (double)tiles * (K / 32) * 2 * (BM + BN) * 64;                                                    \
            const double rounds = (double)((tiles + 255) / 256);                                                                   \
            printf("  %3dx%3d w%dx%d st%d: %7.1f us %6.1f TF (err %.1e) | load-only %7.1f us (%.1f B/clk/CU on the critical path @2.4GHz) | mma-only %7.1f us (%.0f TF) | tiles %ld\n", \
                   BM, BN, WGM, WGN, NST, t3, fl / t3 * 1e-6, err, t1, bytes / tiles * rounds / (t1 * 2400.0), t2, fl / t2 * 1e-6, tiles); \
            fflush(stdout);                                                                                                        \
        }
        RUNCFG(128, 128, 4, 2, 3)
        RUNCFG(128, 128, 2, 4, 3)
        RUNCFG(128, 128, 2, 2, 3)
        RUNCFG(256, 128, 4, 2, 3)
        RUNCFG(256, 128, 4, 2, 2)
        RUNCFG(256, 192, 4, 2, 2)
        RUNCFG(256, 256, 2, 4, 2)
        RUNCFG(256, 256, 4, 2, 2)
        RUNCFG(128, 96, 4, 1, 3)
        RUNCFG(128, 192, 4, 2, 3)
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
