// Microbenchmark: throughput of the inner step of an MFMA-based spreading kernel (developer tool).
// Per "K-block" a wave: reads 8 psi1 values + 8 scalings from LDS, builds a 2-way f16-split A fragment
// (8 mul, 4+4 cvt_pk, 8 cvt back, 8 sub), reads 4 B fragments (ds_read_b128) and issues 6 v_mfma_f32_32x32x16_f16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, int iters)
{
    __shared__ float4 lds[4096];  // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 1024) lds[i] = make_float4(1e-3f * (i & 63), 0.5f, 0.25f, 0.125f);
    __syncthreads();
    f32x16 acc0 = 0.f, acc1 = 0.f;
    const f32x4 *l4 = (const f32x4 *)lds;
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 7 + wave) & 255) * 8;
        // psi1 rows (8 floats) and scalings (8 floats)
        const f32x4 p0 = l4[base + (lane & 31) % 8], p1 = l4[base + 1 + (lane & 7)];
        const f32x4 s0 = l4[(base + 64) & 4095], s1 = l4[(base + 65) & 4095];
        f16x8 ah, al;
        if (MODE & 1) {
            float v[8] = {p0.x * s0.x, p0.y * s0.y, p0.z * s0.z, p0.w * s0.w, p1.x * s1.x, p1.y * s1.y, p1.z * s1.z, p1.w * s1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const _Float16 h = (_Float16)v[j];
                ah[j] = h;
                al[j] = (_Float16)(v[j] - (float)h);
            }
        } else {
            ah = __builtin_bit_cast(f16x8, p0);
            al = __builtin_bit_cast(f16x8, p1);
        }
        // B fragments: hi/lo for two column tiles
        const f16x8 b0h = __builtin_bit_cast(f16x8, l4[(base + 128 + lane) & 4095]);
        const f16x8 b0l = __builtin_bit_cast(f16x8, l4[(base + 192 + lane) & 4095]);
        const f16x8 b1h = __builtin_bit_cast(f16x8, l4[(base + 256 + lane) & 4095]);
        const f16x8 b1l = __builtin_bit_cast(f16x8, l4[(base + 320 + lane) & 4095]);
        if (MODE & 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0h, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0l, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0h, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h, acc1, 0, 0, 0);
        } else {
            acc0[0] += (float)ah[0] + (float)al[1] + (float)b0h[0] + (float)b0l[1];
            acc1[0] += (float)b1h[0] + (float)b1l[1];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * 1024 + tid] = s;
}

template <int MODE>
void run(const char *name)
{
    float *out;
    const int blocks = 256, iters = 4000;
    CHECK(hipMalloc(&out, blocks * 1024 * 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<MODE><<<blocks, 1024>>>(out, iters); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a)); k<MODE><<<blocks, 1024>>>(out, iters); CHECK(hipEventRecord(b)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    // per CU: 16 waves x iters K-block steps
    const double steps_per_cu = 16.0 * iters;
    printf("%-34s %8.3f ms -> %7.1f CU-cycles per (wave, K-block) step @2.1GHz; %6.1f ns\n", name, ms,
           ms * 1e-3 * 2.1e9 / steps_per_cu, ms * 1e6 / steps_per_cu);
    CHECK(hipFree(out));
}

int main()
{
    run<0>("LDS reads only");
    run<1>("LDS + A-fragment VALU");
    run<2>("LDS + 6 MFMA");
    run<3>("LDS + VALU + 6 MFMA");
    return 0;
}
