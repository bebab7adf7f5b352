// Developer microbenchmark: (1) error of the f16 hi/lo split as compiled for the MFMA spreading kernel,
// (2) exactness of v_mfma_f32_32x32x16_f16 on split operands against a float64 dot product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void split_kernel(const float *p, const float *a, float *hi, float *lo, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = p[i] * a[i];
    const _Float16 vh = (_Float16)v;
    const _Float16 vl = (_Float16)(v - (float)vh);
    hi[i] = (float)vh;
    lo[i] = (float)vl;
}

// one wave: D = A (32 x 16) * B (16 x 32), operands given as float, split in-kernel
__global__ void mfma_kernel(const float *A, const float *B, float *D)
{
    const int lane = threadIdx.x, r32 = lane & 31, h = lane >> 5;
    f16x8 ah, al, bh, bl;
    for (int jj = 0; jj < 8; ++jj) {
        const float va = A[r32 * 16 + 8 * h + jj];
        const _Float16 x = (_Float16)va;
        ah[jj] = x; al[jj] = (_Float16)(va - (float)x);
        const float vb = B[(8 * h + jj) * 32 + r32];
        const _Float16 y = (_Float16)vb;
        bh[jj] = y; bl[jj] = (_Float16)(vb - (float)y);
    }
    f32x16 acc = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        D[row * 32 + r32] = acc[reg];
    }
}

int main()
{
    const int n = 1 << 22;
    std::vector<float> p(n), a(n), hi(n), lo(n);
    srand(1);
    for (int i = 0; i < n; ++i) { p[i] = (float)(rand() / (double)RAND_MAX); a[i] = 2048.0f * (float)(rand() / (double)RAND_MAX); }
    float *dp, *da, *dh, *dl;
    (void)hipMalloc(&dp, n * 4); (void)hipMalloc(&da, n * 4); (void)hipMalloc(&dh, n * 4); (void)hipMalloc(&dl, n * 4);
    (void)hipMemcpy(dp, p.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    split_kernel<<<n / 256, 256>>>(dp, da, dh, dl, n);
    (void)hipMemcpy(hi.data(), dh, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(lo.data(), dl, n * 4, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    for (int i = 0; i < n; ++i) {
        const double v = (double)p[i] * a[i];
        const double e = fabs(((double)hi[i] + lo[i]) - v) / (fabs(v) + 1e-30);
        if (e > worst) worst = e;
        if (e > 1e-6 && fabs(v) > 1e-2) ++bad;
    }
    printf("split: worst rel err %.3e, count(rel err > 1e-6) = %d of %d\n", worst, bad, n);

    // MFMA exactness, many random trials; A entries mimic the kernel (a few large, many small/zero)
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, 512 * 4); (void)hipMalloc(&dB, 512 * 4); (void)hipMalloc(&dD, 1024 * 4);
    std::vector<float> A(512), B(512), D(1024);
    double worst_m = 0; int bad_m = 0, total = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        for (int i = 0; i < 512; ++i) {
            const double u = (double)rand() / RAND_MAX, w = (double)rand() / RAND_MAX;
            A[i] = (float)(2048.0 * exp(-20.0 * u * u) * ((double)rand() / RAND_MAX));
            B[i] = (float)(2048.0 * exp(-20.0 * w * w));
        }
        (void)hipMemcpy(dA, A.data(), 512 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 512 * 4, hipMemcpyHostToDevice);
        mfma_kernel<<<1, 64>>>(dA, dB, dD);
        (void)hipMemcpy(D.data(), dD, 1024 * 4, hipMemcpyDeviceToHost);
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double ref = 0, mag = 0;
                for (int k = 0; k < 16; ++k) { ref += (double)A[r * 16 + k] * B[k * 32 + c]; mag += fabs((double)A[r * 16 + k] * B[k * 32 + c]); }
                const double e = fabs(D[r * 32 + c] - ref) / (mag + 1e-30);
                if (e > worst_m) worst_m = e;
                if (e > 2e-6) {
                    if (bad_m < 3) {
                        printf("bad: D %.9g ref %.9g mag %.9g\n", D[r * 32 + c], ref, mag);
                        double hh = 0, hl = 0, lh = 0;
                        for (int k = 0; k < 16; ++k) {
                            const float va = A[r * 16 + k], vb = B[k * 32 + c];
                            const _Float16 ah = (_Float16)va, bh = (_Float16)vb;
                            const _Float16 al = (_Float16)(va - (float)ah), bl = (_Float16)(vb - (float)bh);
                            hh += (double)(float)ah * (float)bh; hl += (double)(float)ah * (float)bl; lh += (double)(float)al * (float)bh;
                            printf("   k %2d  a %.9g (%.9g + %.9g)  b %.9g (%.9g + %.9g)\n", k, va, (float)ah, (float)al, vb, (float)bh, (float)bl);
                        }
                        printf("   host split sums: hh %.9g hl %.9g lh %.9g total %.9g\n", hh, hl, lh, hh + hl + lh);
                    }
                    ++bad_m;
                }
                ++total;
            }
    }
    printf("mfma 3-term split: worst rel err %.3e, count(> 2e-6) = %d of %d\n", worst_m, bad_m, total);
    return 0;
}
