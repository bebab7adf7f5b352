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

// Two-way f16 split of a pair of fp32 values: hi = RN16(v), lo = RN16(v - hi) with the residual formed in fp32 by the
// mixed-precision FMA against the hi that is actually used (letting the compiler fuse the residual under
// -ffp-contract=fast pairs it with a differently rounded hi: one f16 ulp off near ties).
// Instruction choice (scripts/ubench/valu_rates.hip, cycles per wave instruction and SIMD with 4 waves resident):
// v_fma_mixlo/mixhi_f16 -- an FMA with an f16 HALF-register result -- cost 8.5-8.9 (13 with one wave: the partial
// register write chains them), v_fma_mix_f32 4.6, v_cvt_pk_f16_f32 4.5, v_pk_mul_f32 4.7, v_mul_f32 3.1.  So the f16
// results are formed by the packed convert and only full-register FMAs are used: 4 instructions (18 cycles) instead of
// 3 (21.5), and for products 5 (23) instead of 4 (35) -- the products are 10 x 16 instructions per K-block of the
// spreading kernel.  Separate asm statements: the compiler interleaves the chains of neighbouring pairs.
__device__ __forceinline__ void split_pair(const float v0, const float v1, unsigned &hi, unsigned &lo)
{
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v0), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v1), "v"(hi));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
}

// The same for products p * a: hi = RN16(RN32(p a)), lo = RN16(p a - hi) with the residual taken from the EXACT product
// (one FMA), so hi + lo carries p a to ~2^-22 whichever way hi was rounded.
__device__ __forceinline__ void split_product_pair(const float p0, const float a0, const float p1, const float a1,
                                                   unsigned &hi, unsigned &lo)
{
    float v0, v1, r0, r1;
    asm("v_mul_f32 %0, %1, %2" : "=v"(v0) : "v"(p0), "v"(a0));
    asm("v_mul_f32 %0, %1, %2" : "=v"(v1) : "v"(p1), "v"(a1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(p0), "v"(a0), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(p1), "v"(a1), "v"(hi));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
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
