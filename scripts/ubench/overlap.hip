// Micro-benchmark (developer tool): how do MFMA, VALU, transcendental and LDS instructions of the waves of one SIMD
// share issue cycles on gfx950?  One workgroup per CU (100 KB of LDS), wave w runs on SIMD w % 4; every wave runs
// `iters` iterations of its role.  Reported: shader-clock cycles of the whole workgroup per iteration (s_memtime of
// the last wave to finish minus the first to start), and the average shader clock over that time.
// Build: hipcc -O3 --offload-arch=gfx950 overlap.hip -o overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum Role { IDLE = 0, MFMA1, MFMA2, MFMA4, FMA8, PK8, EXP4, MFMA1_F7, MFMA2_F7, LDS2, MFMA16_1, MFMA16_2, MFMA1_F4 };

// CHAINS independent accumulators, one MFMA per iteration (round robin), NV independent FMAs behind each MFMA
template <int CHAINS, int NV>
__device__ __forceinline__ void mfma_loop(int iters, float *out, int lane)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = 0.0f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane + i;
    const float k = 1.0000001f, cc = 1e-9f;
    for (int it = 0; it < iters; it += CHAINS) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(k), "v"(cc));
        }
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456f) out[lane] = s;
}

template <int CHAINS>
__device__ __forceinline__ void mfma16_loop(int iters, float *out, int lane)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = 0.0f;
    for (int it = 0; it < iters; it += CHAINS) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 4; ++i) s += acc[c][i];
    if (s == 123.456f) out[lane] = s;
}

__device__ __forceinline__ void valu_loop(int iters, float *out, int lane)
{
    float x0 = lane * 0.5f, x1 = lane * 0.25f, x2 = lane, x3 = 1.0f, x4 = 2.0f, x5 = 3.f, x6 = 4.f, x7 = 5.f;
    const float k = 1.0000001f, c = 1e-9f;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                     "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(k), "v"(c));
    }
    const float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (s == 123.456f) out[lane] = s;
}

__device__ __forceinline__ void pk_loop(int iters, float *out, int lane)
{
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{(float)lane + i, (float)lane - i};
    const f32x2 k = {1.0000001f, 1.0000002f}, c = {1e-9f, 2e-9f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(k), "v"(c));
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    if (s == 123.456f) out[lane] = s;
}

__device__ __forceinline__ void exp_loop(int iters, float *out, int lane)
{
    float x0 = lane * 0.001f, x1 = lane * 0.002f, x2 = 0.1f, x3 = 0.2f;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                     "v_add_f32 %0, -1.0, %0\n\tv_add_f32 %1, -1.0, %1\n\tv_add_f32 %2, -1.0, %2\n\tv_add_f32 %3, -1.0, %3"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    }
    const float s = x0 + x1 + x2 + x3;
    if (s == 123.456f) out[lane] = s;
}

__device__ __forceinline__ void lds_loop(int iters, float *out, int lane, const f16x8 *lds)
{
    f16x8 s = 0;
    for (int it = 0; it < iters; ++it) {
        f16x8 v0, v1;
        const unsigned a0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)&lds[lane + 64 * (it & 7)];
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0) : "memory");
        s += v0;
        s += v1;
    }
    if ((float)s[0] == 123.456f) out[lane] = (float)s[1];
}

struct Roles { int r[16]; };

__global__ void __launch_bounds__(1024) mix_kernel(Roles roles, int iters, float *out, unsigned long long *clk)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 640) ((f16x8 *)smem)[threadIdx.x] = 0;
    __syncthreads();
    const int role = roles.r[wave];
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    switch (role) {
    case MFMA1: mfma_loop<1, 0>(iters, out, lane); break;
    case MFMA2: mfma_loop<2, 0>(iters, out, lane); break;
    case MFMA4: mfma_loop<4, 0>(iters, out, lane); break;
    case FMA8: valu_loop(iters, out, lane); break;
    case PK8: pk_loop(iters, out, lane); break;
    case EXP4: exp_loop(iters, out, lane); break;
    case MFMA1_F4: mfma_loop<1, 4>(iters, out, lane); break;
    case MFMA1_F7: mfma_loop<1, 7>(iters, out, lane); break;
    case MFMA2_F7: mfma_loop<2, 7>(iters, out, lane); break;
    case LDS2: lds_loop(iters, out, lane, (const f16x8 *)smem); break;
    case MFMA16_1: mfma16_loop<1>(iters, out, lane); break;
    case MFMA16_2: mfma16_loop<2>(iters, out, lane); break;
    default: break;
    }
    if (blockIdx.x == 0 && lane == 0 && role != IDLE) {
        atomicMin(&clk[0], c0);
        atomicMax(&clk[1], (unsigned long long)__builtin_readcyclecounter());
        atomicMin(&clk[2], w0);
        atomicMax(&clk[3], (unsigned long long)wall_clock64());
    }
}

static int run(const char *name, std::vector<int> per_simd, int iters, float *out, unsigned long long *clk)
{
    // per_simd: the roles of the waves of ONE SIMD; replicated on the four SIMDs (wave w -> SIMD w % 4)
    Roles roles;
    const int waves = 4 * (int)per_simd.size();
    for (int i = 0; i < 16; ++i) roles.r[i] = i < waves ? per_simd[i / 4] : 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 100 * 1024;  // one workgroup per CU
    CK(hipFuncSetAttribute((const void *)mix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned long long h[4];
    for (int rep = 0; rep < 2; ++rep) {
        h[0] = h[2] = ~0ull; h[1] = h[3] = 0;
        CK(hipMemcpy(clk, h, 32, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(mix_kernel, dim3(256), dim3(64 * waves), lds, 0, roles, iters, out, clk);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost));
    const double cyc = (double)(h[1] - h[0]), wall = (double)(h[3] - h[2]) * 10.0;  // wall clock: 100 MHz
    printf("%-62s %7.3f ms %7.1f cyc/iter  %.2f GHz\n", name, ms, cyc / iters, cyc / wall);
    return 0;
}

int main()
{
    float *out;
    unsigned long long *clk;
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&clk, 64));
    const int it = 200000;
    printf("# roles per SIMD; cycles per iteration of the whole SIMD (all its waves run `iters` iterations)\n");
    run("1x [MFMA dependent chain]", {MFMA1}, it, out, clk);
    run("1x [MFMA 2 chains]", {MFMA2}, it, out, clk);
    run("1x [MFMA 4 chains]", {MFMA4}, it, out, clk);
    run("2x [MFMA dependent chain]", {MFMA1, MFMA1}, it, out, clk);
    run("4x [MFMA dependent chain]", {MFMA1, MFMA1, MFMA1, MFMA1}, it, out, clk);
    run("2x [MFMA 2 chains]", {MFMA2, MFMA2}, it, out, clk);
    run("4x [MFMA 2 chains]", {MFMA2, MFMA2, MFMA2, MFMA2}, it, out, clk);
    run("1x [8 FMA]", {FMA8}, it, out, clk);
    run("2x [8 FMA]", {FMA8, FMA8}, it, out, clk);
    run("4x [8 FMA]", {FMA8, FMA8, FMA8, FMA8}, it, out, clk);
    run("1x [8 PK_FMA]", {PK8}, it, out, clk);
    run("2x [8 PK_FMA]", {PK8, PK8}, it, out, clk);
    run("4x [8 PK_FMA]", {PK8, PK8, PK8, PK8}, it, out, clk);
    run("1x [4 EXP + 4 ADD]", {EXP4}, it, out, clk);
    run("2x [4 EXP + 4 ADD]", {EXP4, EXP4}, it, out, clk);
    run("4x [4 EXP + 4 ADD]", {EXP4, EXP4, EXP4, EXP4}, it, out, clk);
    run("[MFMA dep] + [8 FMA]", {MFMA1, FMA8}, it, out, clk);
    run("[MFMA dep] + 2x [8 FMA]", {MFMA1, FMA8, FMA8}, it, out, clk);
    run("2x [MFMA dep] + 2x [8 FMA]", {MFMA1, MFMA1, FMA8, FMA8}, it, out, clk);
    run("[MFMA 2 chains] + [8 FMA]", {MFMA2, FMA8}, it, out, clk);
    run("[MFMA 2 chains] + 2x [8 FMA]", {MFMA2, FMA8, FMA8}, it, out, clk);
    run("2x [MFMA 2 chains] + 2x [8 FMA]", {MFMA2, MFMA2, FMA8, FMA8}, it, out, clk);
    run("[MFMA 2 chains] + 2x [4 EXP + 4 ADD]", {MFMA2, EXP4, EXP4}, it, out, clk);
    run("1x [MFMA dep + 4 FMA]", {MFMA1_F4}, it, out, clk);
    run("1x [MFMA dep + 7 FMA]", {MFMA1_F7}, it, out, clk);
    run("1x [MFMA 2 chains + 7 FMA each]", {MFMA2_F7}, it, out, clk);
    run("4x [MFMA dep + 7 FMA]", {MFMA1_F7, MFMA1_F7, MFMA1_F7, MFMA1_F7}, it, out, clk);
    run("4x [MFMA 2 chains + 7 FMA each]", {MFMA2_F7, MFMA2_F7, MFMA2_F7, MFMA2_F7}, it, out, clk);
    run("1x [2 ds_read_b128]", {LDS2}, it, out, clk);
    run("4x [2 ds_read_b128]", {LDS2, LDS2, LDS2, LDS2}, it, out, clk);
    run("[MFMA 2 chains] + 3x [2 ds_read_b128]", {MFMA2, LDS2, LDS2, LDS2}, it, out, clk);
    run("1x [MFMA16 dependent chain]", {MFMA16_1}, it, out, clk);
    run("1x [MFMA16 2 chains]", {MFMA16_2}, it, out, clk);
    run("4x [MFMA16 dependent chain]", {MFMA16_1, MFMA16_1, MFMA16_1, MFMA16_1}, it, out, clk);
    run("2x [MFMA16 2 chains] + 2x [8 FMA]", {MFMA16_2, MFMA16_2, FMA8, FMA8}, it, out, clk);
    return 0;
}
