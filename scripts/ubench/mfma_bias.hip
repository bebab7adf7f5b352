// Developer microbenchmark: signed rounding bias of v_mfma_f32_32x32x16_f16 accumulation.
// D = sum over 4 k-steps of A_ks B_ks (exact f16 inputs, so the only error is the fp32 accumulation inside / between
// the MFMAs); compared with the exact double result, in units of ulp(D).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16 *A, const _Float16 *B, float *D, int ksteps)
{
    const int lane = threadIdx.x, r32 = lane & 31, h = lane >> 5;
    f32x16 acc = 0.0f;
    for (int ks = 0; ks < ksteps; ++ks) {
        f16x8 a, b;
        for (int jj = 0; jj < 8; ++jj) {
            a[jj] = A[r32 * 16 * ksteps + ks * 16 + 8 * h + jj];
            b[jj] = B[(ks * 16 + 8 * h + jj) * 32 + r32];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int reg = 0; reg < 16; ++reg) D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r32] = acc[reg];
}

int main()
{
    const int ks = 4, K = 16 * ks;
    std::vector<_Float16> A(32 * K), B(K * 32);
    std::vector<float> D(1024);
    _Float16 *dA, *dB; float *dD;
    (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dD, 4096);
    srand(7);
    for (int sign = 0; sign < 3; ++sign) {
        double sum_err_ulp = 0, sum_abs_ulp = 0; long cnt = 0;
        for (int trial = 0; trial < 300; ++trial) {
            for (auto &v : A) { double u = rand() / (double)RAND_MAX; double s = sign == 0 ? 1 : (sign == 1 ? -1 : (rand() & 1 ? 1 : -1)); v = (_Float16)(s * 2048.0 * u); }
            for (auto &v : B) { double u = rand() / (double)RAND_MAX; v = (_Float16)(2048.0 * u * u * u); }
            (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
            (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
            k<<<1, 64>>>(dA, dB, dD, ks);
            (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
            for (int r = 0; r < 32; ++r)
                for (int c = 0; c < 32; ++c) {
                    double ref = 0;
                    for (int kk = 0; kk < K; ++kk) ref += (double)(float)A[r * K + kk] * (double)(float)B[kk * 32 + c];
                    const double ulp = ldexp(1.0, ilogb(fabs(ref) + 1e-300) - 23);
                    const double e = ((double)D[r * 32 + c] - ref) / ulp;
                    sum_err_ulp += e; sum_abs_ulp += fabs(e); ++cnt;
                }
        }
        printf("A sign %s: mean signed error %.4f ulp, mean |error| %.4f ulp\n", sign == 0 ? "+" : (sign == 1 ? "-" : "mixed"),
               sum_err_ulp / cnt, sum_abs_ulp / cnt);
    }
    return 0;
}
