// Micro-benchmark (developer tool): does it pay to interleave a wave's vector work with its own MFMAs?
// The matrix-core spreading kernel's owner waves alternate "build the A fragment (16-24 vector instructions)" and
// "3 dependent MFMAs"; three such waves and one builder wave (vector work only) share a SIMD.  Roles:
//   PHASED24  : 8 v_mul_f32 + 8 v_cvt_pk_f16_f32 + 8 v_fma_mix_f32, then 3 dependent v_mfma_f32_32x32x16_f16 (round-3 kernel)
//   PHASED16  : 16 packed-f16 instructions (4 chains of mul, fma, fma, fma), then the 3 MFMAs
//   INTER16   : MFMA, 5 packed, MFMA, 5 packed, MFMA, 6 packed (the same 16 + 3, interleaved in program order)
//   INTER24   : MFMA, 8 of the 24, MFMA, 8, MFMA, 8
//   BUILDER   : 24 vector instructions per iteration (v_fma_f32 / v_exp_f32 mix), no MFMA
// Reported per role: cycles per iteration of one wave (its own s_memtime span / iterations) and the shader clock.
// Build: hipcc -O3 --offload-arch=gfx950 phase_interleave.hip -o phase_interleave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum Role { IDLE = 0, PHASED24, PHASED16, INTER16, INTER24, BUILDER, MFMA_ONLY };

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define PKMUL(d, x, y) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define PKFMA(d, x, y) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(d) : "v"(x), "v"(y))
#define VMUL(d, x, y) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define VCVT(d, x, y) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define VMIX(d, x, y, z) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(x), "v"(y), "v"(z))

struct Roles { int r[16]; };

__global__ void __launch_bounds__(1024) mix_kernel(Roles roles, int iters, float *out, unsigned long long *clk)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = roles.r[wave];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc = 0.0f;
    unsigned p0 = 0x3c003c00u + lane, p1 = 0x38003800u + lane, p2 = 0x3a003a00u, p3 = 0x39003900u + 2 * lane;
    unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    float x0 = lane * 0.5f, x1 = 1.0f + lane, x2 = 0.25f, x3 = 3.0f, y0, y1, y2, y3, r0, r1, r2, r3;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if (role == PHASED24 || role == INTER24) {
#define S8(k) VMUL(y0, x0, x1); VMUL(y1, x2, x3); VCVT(d##k, y0, y1); VMIX(r0, x0, x1, d##k); VMIX(r1, x2, x3, d##k); VCVT(p##k, r0, r1); \
              VMUL(y2, x1, x2); VMUL(y3, x3, x0);
#define S8B(k) VCVT(d##k, y2, y3); VMIX(r2, x1, x2, d##k); VMIX(r3, x3, x0, d##k); VCVT(p##k, r2, r3); \
               VMUL(y0, x0, x3); VMUL(y1, x2, x1); VCVT(d0, y0, y1); VMIX(r0, x0, x3, d0);
#define S8C VMIX(r1, x2, x1, d0); VCVT(p0, r0, r1); VMUL(y2, x1, x3); VMUL(y3, x2, x0); VCVT(d1, y2, y3); VMIX(r2, x1, x3, d1); VMIX(r3, x2, x0, d1); VCVT(p1, r2, r3);
        if (role == PHASED24) {
            for (int it = 0; it < iters; ++it) {
                S8(2) S8B(3) S8C
                MFMA(acc, a, b); MFMA(acc, a, b); MFMA(acc, a, b);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                MFMA(acc, a, b); S8(2) MFMA(acc, a, b); S8B(3) MFMA(acc, a, b); S8C
            }
        }
    } else if (role == PHASED16 || role == INTER16) {
        if (role == PHASED16) {
            for (int it = 0; it < iters; ++it) {
                PKMUL(d0, p0, p1); PKMUL(d1, p1, p2); PKMUL(d2, p2, p3); PKMUL(d3, p3, p0);
                PKFMA(d0, p0, p1); PKFMA(d1, p1, p2); PKFMA(d2, p2, p3); PKFMA(d3, p3, p0);
                PKFMA(d0, p2, p1); PKFMA(d1, p3, p2); PKFMA(d2, p0, p3); PKFMA(d3, p1, p0);
                PKFMA(d0, p0, p3); PKFMA(d1, p1, p0); PKFMA(d2, p2, p1); PKFMA(d3, p3, p2);
                MFMA(acc, a, b); MFMA(acc, a, b); MFMA(acc, a, b);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                MFMA(acc, a, b);
                PKMUL(d0, p0, p1); PKMUL(d1, p1, p2); PKMUL(d2, p2, p3); PKMUL(d3, p3, p0); PKFMA(d0, p0, p1);
                MFMA(acc, a, b);
                PKFMA(d1, p1, p2); PKFMA(d2, p2, p3); PKFMA(d3, p3, p0); PKFMA(d0, p2, p1); PKFMA(d1, p3, p2);
                MFMA(acc, a, b);
                PKFMA(d2, p0, p3); PKFMA(d3, p1, p0); PKFMA(d0, p0, p3); PKFMA(d1, p1, p0); PKFMA(d2, p2, p1); PKFMA(d3, p3, p2);
            }
        }
    } else if (role == BUILDER) {
        float z0 = lane * 0.001f, z1 = 0.5f, z2 = 0.25f, z3 = 0.125f;
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %1\n\tv_fma_f32 %3, %3, %1, %2\n\t"
                         "v_add_f32 %0, -1.0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %1\n\tv_fma_f32 %3, %3, %1, %2\n\t"
                         "v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %1\n\tv_fma_f32 %3, %3, %1, %2\n\t"
                         "v_add_f32 %0, -1.0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %1\n\tv_fma_f32 %3, %3, %1, %2\n\t"
                         "v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %1\n\tv_fma_f32 %3, %3, %1, %2\n\t"
                         "v_add_f32 %0, -1.0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %1\n\tv_fma_f32 %3, %3, %1, %2"
                         : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3));
        }
        x0 += z0 + z1 + z2 + z3;
    } else if (role == MFMA_ONLY) {
        for (int it = 0; it < iters; ++it) { MFMA(acc, a, b); MFMA(acc, a, b); MFMA(acc, a, b); }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = x0 + x1 + (float)(d0 + d1 + d2 + d3) + (float)(p0 + p1 + p2 + p3);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 123.456f) out[lane] = s;
    if (blockIdx.x == 0 && lane == 0 && role != IDLE) {
        atomicAdd(&clk[role * 4 + 0], c1 - c0);
        atomicAdd(&clk[role * 4 + 1], (unsigned long long)(w1 - w0));
        atomicAdd(&clk[role * 4 + 2], 1ull);
    }
}

static const char *kNames[] = {"idle", "PHASED24", "PHASED16", "INTER16", "INTER24", "BUILDER", "MFMA_ONLY"};

static int run(const char *name, std::vector<int> per_simd, int iters, float *out, unsigned long long *clk)
{
    Roles roles;
    const int waves = 4 * (int)per_simd.size();
    for (int i = 0; i < 16; ++i) roles.r[i] = i < waves ? per_simd[i / 4] : 0;
    unsigned long long h[32];
    float ms = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(clk, 0, sizeof(h)));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(mix_kernel, dim3(256), dim3(64 * waves), 0, 0, roles, iters, out, clk);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-44s %7.3f ms |", name, ms);
    for (int r = 1; r < 7; ++r)
        if (h[r * 4 + 2])
            printf("  %s %.1f cyc/iter/wave (%.2f GHz)", kNames[r], (double)h[r * 4] / h[r * 4 + 2] / iters,
                   (double)h[r * 4] / ((double)h[r * 4 + 1] * 10.0));
    printf("\n");
    return 0;
}

int main()
{
    float *out;
    unsigned long long *clk;
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&clk, 256));
    const int it = 100000;
    printf("# waves per SIMD and their roles; per role: cycles per iteration of one wave (3 MFMAs + its vector work)\n");
    run("1x MFMA_ONLY (3 dependent MFMAs)", {MFMA_ONLY}, it, out, clk);
    run("3x MFMA_ONLY", {MFMA_ONLY, MFMA_ONLY, MFMA_ONLY}, it, out, clk);
    run("1x PHASED24", {PHASED24}, it, out, clk);
    run("1x INTER24", {INTER24}, it, out, clk);
    run("1x PHASED16", {PHASED16}, it, out, clk);
    run("1x INTER16", {INTER16}, it, out, clk);
    run("3x PHASED24", {PHASED24, PHASED24, PHASED24}, it, out, clk);
    run("3x INTER24", {INTER24, INTER24, INTER24}, it, out, clk);
    run("3x PHASED16", {PHASED16, PHASED16, PHASED16}, it, out, clk);
    run("3x INTER16", {INTER16, INTER16, INTER16}, it, out, clk);
    run("1x BUILDER", {BUILDER}, it, out, clk);
    run("3x PHASED24 + BUILDER", {PHASED24, PHASED24, PHASED24, BUILDER}, it, out, clk);
    run("3x INTER24 + BUILDER", {INTER24, INTER24, INTER24, BUILDER}, it, out, clk);
    run("3x PHASED16 + BUILDER", {PHASED16, PHASED16, PHASED16, BUILDER}, it, out, clk);
    run("3x INTER16 + BUILDER", {INTER16, INTER16, INTER16, BUILDER}, it, out, clk);
    return 0;
}
