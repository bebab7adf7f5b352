// Shared pieces of the matrix-core kernels (spread_mfma.hip, interp_mfma.hip): fp32 values enter
// v_mfma_f32_32x32x16_f16 as two-way f16 splits (v = hi + lo); hi*hi + hi*lo + lo*hi keeps ~22 bits.
#pragma once
#include <hip/hip_runtime.h>

namespace nfft {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// four floats at an address that is only dword-aligned (grid rows read at a tile offset): the type says so, the
// instruction stays global_load_dwordx4
typedef float f32x4_dw __attribute__((ext_vector_type(4), aligned(4)));

// Operands are scaled to at most 2^11 before the split (values <= 2048 fit f16): the lo parts, ~2^-11 of the
// value, are then normal f16 numbers instead of subnormals; the consumer multiplies by 2^-11 per operand.
constexpr float kOpScale = 2048.0f;

// Two-way f16 split of fp32 values in three VALU instructions per pair: hi = RN16(RN32(v)), lo = RN16(v - hi) with
// the subtraction done by the mixed-precision FMA against the hi that is actually used (letting the compiler fuse
// the residual under -ffp-contract=fast pairs it with a differently rounded hi: one f16 ulp off near ties).
__device__ __forceinline__ void split_pair(const float v0, const float v1, unsigned &hi, unsigned &lo)
{
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(v0), "v"(v1));
}

// The same for products p * a: hi = RN16(p a), lo = RN16(p a - hi), both from the exact product (4 instructions).
__device__ __forceinline__ void split_product_pair(const float p0, const float a0, const float p1, const float a1,
                                                   unsigned &hi, unsigned &lo)
{
    asm("v_fma_mixlo_f16 %0, %2, %3, 0\n\t"
        "v_fma_mixhi_f16 %0, %4, %5, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %3, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %4, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(p0), "v"(a0), "v"(p1), "v"(a1));
}

// One dword per active lane, global -> LDS without a register in between (LDS-DMA): lane l of the wave lands at
// lds_wave_base + 4 l.  Issued through asm so that the compiler does not drain it (vmcnt(0)) at the next LDS read or
// barrier; the consumer waits with wait_lds_dma() one pipeline step later.
__device__ __forceinline__ void lds_dma_dword(const float *gsrc, const float *lds_wave_base)
{
    const unsigned lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds)
                 : "memory");
}
// Sixteen bytes per active lane (a plan record {p0, p1, p2, x}): lane l lands at lds_wave_base + 16 l.
__device__ __forceinline__ void lds_dma_dwordx4(const float *gsrc, const void *lds_wave_base)
{
    const unsigned lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds)
                 : "memory");
}
__device__ __forceinline__ void wait_lds_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// all but the newest request (2 LDS-DMA instructions: the point record and the coefficient) have landed
__device__ __forceinline__ void wait_lds_dma_but_newest() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
// workgroup barrier that leaves global traffic (LDS-DMA, flush atomics) in flight
__device__ __forceinline__ void barrier_lds_only() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


} // namespace nfft
