// Stand-alone development harness: schedules of the bf16x3 plane GEMM k-loop (NT form, operands = three bf16 planes [3][rows][ld]).
// SCHED 0 = the shipped schedule of gemm_planes.hip (8 waves in lockstep, one barrier per 32-deep k-tile, 3-stage LDS ring);
// SCHED 1 = ping-pong by k-tile: the workgroup's waves form two groups (the two waves of every SIMD in different groups), offset by
//           one barrier interval: while one group runs the 24 MFMAs of a k-tile, the other issues its DMA and reads its fragments;
// SCHED 2 = ping-pong by 16-deep k-step (one fragment set, four barriers per k-tile).
// All three run the same MFMAs in the same order per accumulator: results are bit-identical.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 pingpong_gemm_probe.hip -o pingpong_gemm_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned pg_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pg_bf16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned PG_INVALID = 0x80000000u;

struct desc {
    const unsigned short* A; const unsigned short* B; float* C;
    long long a_plane, b_plane, lda, ldb, ldc;
    int M, N, K;
    float alpha;
    unsigned* simd_map;          // [blocks][16]: HW_ID of every wave (diagnostic)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pg_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
}

// WIDE = 1: what-if (results are garbage): every DMA instruction reads 8 rows x 128 B instead of 16 rows x 64 B -- same LDS bytes and
// instruction count, half the L2 requests
template <int BM, int BN, int WGM, int WGN, int SCHED, int GSEL, int WIDE = 0>
__global__ __launch_bounds__(WGM * WGN * 64) void pp_gemm_kernel(const desc p) {
    constexpr int NW = WGM * WGN, NT = NW * 64, NST = (SCHED == 3) ? 2 : 3;
    constexpr int BK = 32, NP = 3;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM * 64, PB = BN * 64;
    constexpr int STAGE = NP * (PA + PB);
    constexpr int NIA = NP * BM / 16 / NW, NIB = NP * BN / 16 / NW;
    static_assert((NP * BM / 16) % NW == 0 && (NP * BN / 16) % NW == 0, "DMA units must divide over the waves");
    constexpr int CTS = BN + 4;
    constexpr int SMEM = (NST * STAGE > BM * CTS * 4) ? NST * STAGE : BM * CTS * 4;
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
    if (p.simd_map && lin < 64 && lane == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        p.simd_map[lin * 16 + wave] = hw;
    }
    if (m0 >= M || n0 >= N) return;
    const int ntiles = p.K / BK;

    unsigned avo[NIA], bvo[NIB];
    int alds[NIA], blds[NIB];
    {
        const int lrow = WIDE ? (lane >> 3) : (lane >> 2);
        const int lchunk = WIDE ? (lane & 7) : ((lane & 3) ^ ((lane >> 4) & 3));
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
    auto issue_a = [&](int stage, unsigned inv) {
        unsigned char* sb = smem + stage * STAGE;
        const __amdgpu_buffer_rsrc_t ra = pg_rsrc(abase);
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb + alds[i]), 16, (int)(avo[i] | inv), 0, 0, 0);
        abase += BK;
    };
    auto issue_b = [&](int stage, unsigned inv) {
        unsigned char* sb = smem + stage * STAGE;
        const __amdgpu_buffer_rsrc_t rb = pg_rsrc(bbase);
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + blds[i]), 16, (int)(bvo[i] | inv), 0, 0, 0);
        bbase += BK;
    };
    auto issue = [&](int stage, unsigned inv) { issue_a(stage, inv); issue_b(stage, inv); };

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
        const unsigned char* as = smem + stage * STAGE + (wm * WM) * 64 + fo;
        const unsigned char* bs = smem + stage * STAGE + NP * PA + (wn * WN) * 64 + fo;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[q][i] = *reinterpret_cast<const pg_u32x4*>(as + q * PA + i * 32 * 64);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[q][j] = *reinterpret_cast<const pg_u32x4*>(bs + q * PB + j * 32 * 64);
        }
    };
    auto mma = [&](const pg_u32x4 (&fa)[NP][TM], const pg_u32x4 (&fb)[NP][TN]) {
        constexpr int qa[6] = {2, 0, 1, 1, 0, 0}, qb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pg_bf16x8, fa[qa[t]][i]),
                                                                        __builtin_bit_cast(pg_bf16x8, fb[qb[t]][j]), acc[i][j], 0, 0, 0);
    };
    constexpr int NIW = NIA + NIB;
    constexpr int NMMA = 6 * TM * TN, NRD = NP * (TM + TN);

    if constexpr (SCHED == 0) {
        issue(0, 0u);
        issue(1, ntiles > 1 ? 0u : PG_INVALID);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
        __builtin_amdgcn_s_barrier();
        auto settle0 = [&]() {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa0[q][i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb0[q][j]));
            }
        };
        read_frags(0, fo0, fa0, fb0);
        settle0();
        int cur = 0;
        for (int t = 0; t < ntiles; ++t) {
            const int nxt = (cur + 1 == NST) ? 0 : cur + 1;
            const int nn = (nxt + 1 == NST) ? 0 : nxt + 1;
            __builtin_amdgcn_sched_barrier(0);
            issue(nn, t + NST - 1 < ntiles ? 0u : PG_INVALID);
            read_frags(cur, fo1, fa1, fb1);
            mma(fa0, fb0);
#pragma unroll
            for (int g = 0; g < NMMA; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (g < NIW) { __builtin_amdgcn_sched_group_barrier(0x004, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
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
            __builtin_amdgcn_sched_barrier(0);
            settle0();
            cur = nxt;
        }
    } else {
        // ---- ping-pong: group 1 runs one barrier interval behind group 0 ----------------------------------------------------------
        const int grp = (GSEL == 0) ? (wave / (NW / 2)) : (wave & 1);
        issue(0, 0u);
        if constexpr (SCHED == 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            issue(1, ntiles > 1 ? 0u : PG_INVALID);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
        }
        __builtin_amdgcn_s_barrier();                       // tile 0 landed (every wave's share)
        if (grp) __builtin_amdgcn_s_barrier();              // the followers start one interval late
        int cur = 0;
        for (int t = 0; t < ntiles; ++t) {
            const int nxt = (cur + 1 == NST) ? 0 : cur + 1;
            const int nn = (nxt + 1 == NST) ? 0 : nxt + 1;
            const unsigned inv = t + 2 < ntiles ? 0u : PG_INVALID;
            if constexpr (SCHED == 1) {
                // LOAD interval: DMA of tile t + 2 (into the stage of tile t - 1, which both groups have read), fragments of tile t
                __builtin_amdgcn_sched_barrier(0);
                issue(nn, inv);
                read_frags(cur, fo0, fa0, fb0);
                read_frags(cur, fo1, fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                // this wave's share of tile t + 1 has landed (tile t + 2 stays in flight); its fragment reads are done
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NIW) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // MFMA interval
                __builtin_amdgcn_s_setprio(1);
                mma(fa0, fb0);
                mma(fa1, fb1);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            } else if constexpr (SCHED == 3) {
                // two stages: the whole of tile t + 1 is issued in the first LOAD interval of tile t (its stage held tile t - 1, whose
                // last readers finished an interval ago) and waited for at the end of the second one
                __builtin_amdgcn_sched_barrier(0);
                issue(nxt, t + 1 < ntiles ? 0u : PG_INVALID);
                read_frags(cur, fo0, fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                mma(fa0, fb0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                read_frags(cur, fo1, fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                mma(fa0, fb0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            } else {
                __builtin_amdgcn_sched_barrier(0);
                issue_a(nn, inv);
                read_frags(cur, fo0, fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                mma(fa0, fb0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                issue_b(nn, inv);
                read_frags(cur, fo1, fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NIW) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                mma(fa0, fb0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            cur = nxt;
        }
        if (!grp) __builtin_amdgcn_s_barrier();             // the leaders wait for the followers' last interval
    }
    __syncthreads();

    const float alpha = p.alpha;
    const long long ldc = p.ldc;
    float* const C = p.C;
    float* const Ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ct[(wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CTS + wn * WN + j * 32 + lr] = acc[i][j][r] * alpha;
    __syncthreads();
    constexpr int QN = BN / 4;
#pragma unroll
    for (int q = 0; q < BM * QN / NT; ++q) {
        const int idx = tid + q * NT;
        const int row = idx / QN, c = (idx % QN) * 4;
        const int gm = m0 + row, gn = n0 + c;
        if (gm >= M || gn >= N) continue;
        const float4 v = *reinterpret_cast<const float4*>(&Ct[row * CTS + c]);
        *reinterpret_cast<float4*>(C + (long long)gm * ldc + gn) = v;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static unsigned short bf_hi(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
static float bf_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Problem { const char* name; int M, N, K; };

template <int BM, int BN, int WGM, int WGN, int SCHED, int GSEL, int WIDE = 0>
static float run(const desc& d, int iters) {
    dim3 g((d.M + BM - 1) / BM, (d.N + BN - 1) / BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pp_gemm_kernel<BM, BN, WGM, WGN, SCHED, GSEL, WIDE>), g, dim3(WGM * WGN * 64), 0, 0, d);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((pp_gemm_kernel<BM, BN, WGM, WGN, SCHED, GSEL, WIDE>), g, dim3(WGM * WGN * 64), 0, 0, d);
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
        {"FFN2 fwd     ", 4128, 768, 3072}, {"FFN1 4096    ", 4096, 3072, 768}, {"big 8192^2x1k", 8192, 8192, 1024},
    };
    const int iters = 20;
    unsigned* dmap;
    CK(hipMalloc(&dmap, 64 * 16 * 4));
    CK(hipMemset(dmap, 0xff, 64 * 16 * 4));
    bool printed_map = false;
    for (const Problem& pr : probs) {
        const int M = pr.M, N = pr.N, K = pr.K;
        std::vector<unsigned short> ha((size_t)3 * M * K), hb((size_t)3 * N * K);
        srand(1234);
        auto fill = [&](std::vector<unsigned short>& v, int rows, float scale) {
            const size_t pl = (size_t)rows * K;
            for (size_t i = 0; i < pl; ++i) {
                const float x = scale * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
                const unsigned short h = bf_hi(x);
                const float r1 = x - bf_f(h);
                const unsigned short m = bf_hi(r1);
                const float r2 = r1 - bf_f(m);
                v[i] = h; v[pl + i] = m; v[2 * pl + i] = bf_hi(r2);
            }
        };
        fill(ha, M, 1.f);
        fill(hb, N, 1.f);
        unsigned short *dA, *dB; float *dC, *dC0;
        CK(hipMalloc(&dA, ha.size() * 2)); CK(hipMalloc(&dB, hb.size() * 2));
        CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dC0, (size_t)M * N * 4));
        CK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
        desc d{dA, dB, dC, (long long)M * K, (long long)N * K, K, K, N, M, N, K, 1.0f, nullptr};
        std::vector<float> hc((size_t)M * N), hc0((size_t)M * N);
        auto check = [&](bool ref) {
            CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0;
            for (int s = 0; s < 48; ++s) {
                const int m = (s * 977 + (s % 3 == 0 ? M - 1 - s : 0)) % M, n = (s * 613 + 5) % N;
                double r = 0, mag = 0;
                for (int k = 0; k < K; ++k) {
                    double a = 0, b = 0;
                    for (int q = 0; q < 3; ++q) { a += bf_f(ha[(size_t)q * M * K + (size_t)m * K + k]); b += bf_f(hb[(size_t)q * N * K + (size_t)n * K + k]); }
                    r += a * b; mag += fabs(a * b);
                }
                const double e = fabs(r - hc[(size_t)m * N + n]) / mag;
                if (e > worst) worst = e;
            }
            size_t diff = 0;
            if (ref) hc0 = hc;
            else for (size_t i = 0; i < hc.size(); ++i) diff += (memcmp(&hc[i], &hc0[i], 4) != 0);
            if (worst > 2e-6 || diff) printf("   !! err/mag %.3g, %zu elements differ from SCHED 0\n", worst, diff);
            return worst;
        };
        const double fl = 2.0 * M * N * K;
        printf("%s %dx%dx%d\n", pr.name, M, N, K);
#define RUNCFG(BM, BN, WGM, WGN, SCHED, GSEL, REF, WIDE)                                                                                \
        {                                                                                                                          \
            CK(hipMemset(dC, 0, (size_t)M * N * 4));                                                                               \
            d.simd_map = printed_map ? nullptr : dmap;                                                                            \
            const float t3 = run<BM, BN, WGM, WGN, SCHED, GSEL, WIDE>(d, iters);                                                        \
            const double err = WIDE ? -1.0 : check(REF);                                                                                        \
            const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);                                                    \
            const double rounds = (double)((tiles + 255) / 256);                                                                   \
            printf("  %3dx%3d w%dx%d sched %d gsel %d wide %d: %7.1f us %6.1f TF (err %.1e) | per full round of tiles: MFMA-busy %.2f @2.4GHz | tiles %ld\n", \
                   BM, BN, WGM, WGN, SCHED, GSEL, WIDE, t3, fl / t3 * 1e-6, err,                                                         \
                   (double)(K / 16) * 6 * (BM / 32) * (BN / 32) * 32 / 4 * rounds / (t3 * 2400.0), tiles);                         \
            fflush(stdout);                                                                                                        \
            if (!printed_map) {                                                                                                    \
                unsigned hm[64 * 16];                                                                                              \
                CK(hipMemcpy(hm, dmap, sizeof hm, hipMemcpyDeviceToHost));                                                         \
                for (int b = 0; b < 4; ++b) {                                                                                      \
                    printf("    block %d: wave -> (simd, cu, se)", b);                                                             \
                    for (int w = 0; w < WGM * WGN; ++w) printf(" %d:(%u,%u,%u)", w, (hm[b * 16 + w] >> 4) & 3, (hm[b * 16 + w] >> 8) & 15, (hm[b * 16 + w] >> 13) & 7); \
                    printf("\n");                                                                                                  \
                }                                                                                                                  \
                printed_map = true;                                                                                                \
            }                                                                                                                      \
        }
        RUNCFG(128, 128, 2, 4, 0, 0, true, 0)
        RUNCFG(128, 128, 2, 4, 1, 0, false, 0)
        RUNCFG(128, 128, 2, 4, 2, 0, false, 0)
        RUNCFG(128, 128, 2, 4, 0, 0, false, 1)
        RUNCFG(128, 128, 2, 4, 1, 0, false, 1)
        RUNCFG(128, 128, 2, 4, 2, 0, false, 1)
        RUNCFG(256, 128, 4, 2, 3, 0, false, 0)
        RUNCFG(256, 128, 4, 2, 3, 0, false, 1)
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dC0));
    }
    return 0;
}
