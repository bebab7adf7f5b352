// Test entry for the error-free transformations the kernels are built on (tests/test_gpu_eft.py compares them bit for
// bit with a float64 emulation): split_cell (common.h), split_pair and split_product_f16x4 (mfma_split.h).  The library
// is built with -ffp-contract=fast; twice a compiler-fused residual broke one of these silently,
// so they are written in inline asm and pinned here.  Not part of the C ABI of include/nfft_hip.h (no reference
// counterpart); exported as nfft_dbg_eft for the test.
#include "common.h"
#include "mfma_split.h"

namespace nfft {
namespace {

__global__ void __launch_bounds__(256) eft_kernel(int kind, int64_t n, int M, const unsigned *__restrict__ a, const unsigned *__restrict__ b,
                                                 const unsigned *__restrict__ c, const unsigned *__restrict__ d,
                                                 unsigned *__restrict__ out0, unsigned *__restrict__ out1)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (kind == 0) {  // split_cell(pos = a[i], M) -> cell, frac
            int cell;
            float frac;
            split_cell(__uint_as_float(a[i]), M, cell, frac);
            out0[i] = (unsigned)cell;
            out1[i] = __float_as_uint(frac);
        } else if (kind == 1) {  // split_pair(a[i], b[i]) -> hi (two f16), lo (two f16)
            unsigned hi, lo;
            split_pair(__uint_as_float(a[i]), __uint_as_float(b[i]), hi, lo);
            out0[i] = hi;
            out1[i] = lo;
        } else {  // split_product_f16x4 on four identical pairs {ph = a, pl = b, ah = c, al = d} (packed f16 pairs)
            const u32x4 ph = {a[i], a[i], a[i], a[i]}, pl = {b[i], b[i], b[i], b[i]}, xh = {c[i], c[i], c[i], c[i]},
                        xl = {d[i], d[i], d[i], d[i]};
            u32x4 hi, lo;
            split_product_f16x4(ph, pl, xh, xl, hi, lo);
            // the four chains of the statement must agree
            out0[i] = (hi.x == hi.y && hi.x == hi.z && hi.x == hi.w) ? hi.x : 0xffffffffu;
            out1[i] = (lo.x == lo.y && lo.x == lo.z && lo.x == lo.w) ? lo.x : 0xffffffffu;
        }
    }
}

} // namespace
} // namespace nfft

extern "C" int nfft_dbg_eft(int kind, int64_t n, int M, const void *a, const void *b, const void *c, const void *d, void *out0,
                            void *out1, void *stream)
{
    if (n <= 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(nfft::eft_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, kind, n, M, (const unsigned *)a,
                       (const unsigned *)b, (const unsigned *)c, (const unsigned *)d, (unsigned *)out0, (unsigned *)out1);
    return (int)hipGetLastError();
}
