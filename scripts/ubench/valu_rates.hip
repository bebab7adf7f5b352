// Micro-benchmark (developer tool): issue cost of the VALU instructions the matrix-core kernels are made of, on gfx950.
// One workgroup per CU, 4 waves per SIMD, every wave runs `iters` x 16 independent instructions of one kind.
// Reported: ns per wave instruction and SIMD, relative to v_fma_f32.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define REP16(S) S S S S S S S S S S S S S S S S

template <int KIND>
__global__ void __launch_bounds__(1024) rate_kernel(int iters, float *out)
{
    float x0 = threadIdx.x * 0.5f + 1.0f, x1 = threadIdx.x * 0.25f + 2.0f, x2 = 1.5f, x3 = 0.75f;
    float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
    unsigned u0 = threadIdx.x, u1 = threadIdx.x * 3u, u2 = 7u, u3 = 11u;
    unsigned long long w0 = threadIdx.x, w1 = 12345ull;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {1.0000001f, 0.9999999f};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y0) : "v"(x0), "v"(x1));) }
        if (KIND == 1) { REP16(asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x1));) }
        if (KIND == 2) { REP16(asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(y0) : "v"(x0), "v"(x1));) }
        if (KIND == 3) { REP16(asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(y0) : "v"(x0), "v"(x1));) }
        if (KIND == 4) { REP16(asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(y1) : "v"(x0), "v"(x1), "v"(y0));) }
        if (KIND == 5) { REP16(asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(y1) : "v"(x0), "v"(x1), "v"(y0));) }
        if (KIND == 6) { REP16(asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(p1), "v"(p2));) }
        if (KIND == 7) { REP16(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p2));) }
        if (KIND == 8) { REP16(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x1));) }
        if (KIND == 9) { REP16(asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x1));) }
        if (KIND == 10) { REP16(asm volatile("v_exp_f32 %0, %1" : "=v"(y0) : "v"(x0));) }
        if (KIND == 11) { REP16(asm volatile("v_lshl_add_u64 %0, %1, 2, %0" : "+v"(w0) : "v"(w1));) }
        if (KIND == 12) { REP16(asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(w0) : "v"(u0), "v"(u1) : "s2", "s3");) }
        if (KIND == 13) { REP16(asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u2) : "v"(u0), "v"(u1));) }
        if (KIND == 14) { REP16(asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(u2) : "v"(u0), "v"(u1));) }
        if (KIND == 15) { REP16(asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(y0) : "v"(x0), "v"(x1));) }
        if (KIND == 16) { REP16(asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x1));) }
        if (KIND == 17) { REP16(asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(y0) : "v"(x0));) }
        if (KIND == 18) { REP16(asm volatile("v_and_b32 %0, %1, %2" : "=v"(u2) : "v"(u0), "v"(u1));) }
        if (KIND == 19) { REP16(asm volatile("v_readlane_b32 s2, %0, 3" : : "v"(u0) : "s2");) }
    }
    const float s = y0 + y1 + y2 + y3 + p0.x + p0.y + (float)u2 + (float)u3 + (float)w0;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int KIND>
static int run(const char *name, int waves_per_simd, float *out, double &base)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(256 * waves_per_simd), 0, 0, 100, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(256 * waves_per_simd), 0, 0, iters, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / ((double)iters * 16 * waves_per_simd);  // per wave instruction on one SIMD
    if (KIND == 0) base = ns;
    printf("%-34s %d waves/SIMD  %6.2f ns per instruction  (%.2fx v_fma_f32; %.1f cycles at 2.4 GHz)\n", name, waves_per_simd, ns,
           ns / base, ns * 2.4);
    return 0;
}

int main()
{
    float *out;
    CK(hipMalloc(&out, 4096));
    double base = 1.0;
    for (int w : {4, 1}) {
        run<0>("v_fma_f32", w, out, base);
        run<1>("v_mul_f32", w, out, base);
        run<2>("v_fma_mixlo_f16 (f32,f32 -> f16)", w, out, base);
        run<3>("v_fma_mixhi_f16", w, out, base);
        run<4>("v_fma_mixlo_f16 with an f16 addend", w, out, base);
        run<5>("v_fma_mix_f32 with an f16 addend", w, out, base);
        run<6>("v_pk_mul_f32", w, out, base);
        run<7>("v_pk_fma_f32", w, out, base);
        run<8>("v_cvt_pk_f16_f32", w, out, base);
        run<9>("v_cvt_pkrtz_f16_f32", w, out, base);
        run<10>("v_exp_f32", w, out, base);
        run<11>("v_lshl_add_u64", w, out, base);
        run<12>("v_mad_u64_u32", w, out, base);
        run<13>("v_mul_lo_u32", w, out, base);
        run<14>("v_add3_u32", w, out, base);
        run<15>("v_pk_fma_f16", w, out, base);
        run<16>("v_pk_mul_f16", w, out, base);
        run<17>("v_cvt_f32_f16", w, out, base);
        run<18>("v_and_b32", w, out, base);
        run<19>("v_readlane_b32", w, out, base);
    }
    return 0;
}
