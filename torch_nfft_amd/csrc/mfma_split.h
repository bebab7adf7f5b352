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
// 3 (21.5).  Separate asm statements: the compiler interleaves the chains of neighbouring pairs.
__device__ __forceinline__ void split_pair(const float v0, const float v1, unsigned &hi, unsigned &lo)
{
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v0), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v1), "v"(hi));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
}

// Split of a PRODUCT p a from the f16 splits of its factors (p = ph + pl, a = ah + al, all four normal f16 numbers), in
// packed f16 arithmetic: hi = RN16(ph ah); e = ph ah - hi EXACTLY (the rounding error of an f16 product is an f16 number:
// one FMA); lo = RN16(pl ah + RN16(ph al + e)).  hi + lo = p a (1 + O(2^-21)): the dropped term pl al is 2^-22 of the
// product and the two roundings of lo are 2^-11 of a 2^-10 part.  4 VOP3P instructions per PAIR of elements and no fp32
// operand: the spreading kernel's plane-owner waves form 8 such pairs per K-block and plane, which was 6 instructions per
// pair (v_mul x2, v_cvt_pk, v_fma_mix x2, v_cvt_pk) on fp32 factors (scripts/ubench/phase_interleave.hip: three owner waves
// per SIMD, 16 packed instead of 24 mixed instructions beside 3 MFMAs, -9.5 % time).  Written in asm: under
// -ffp-contract=fast nothing may re-associate an error-free transformation (tests/test_gpu_eft.py checks it bit for bit).
// (four pairs per statement: the compiler pads every boundary between dependent asm statements with an s_nop, and one
// statement lets the four independent chains alternate)
__device__ __forceinline__ void split_product_f16x4(const u32x4 ph, const u32x4 pl, const u32x4 ah, const u32x4 al, u32x4 &hi,
                                                    u32x4 &lo)
{
    unsigned h0, h1, h2, h3, q0, q1, q2, q3;
    asm("v_pk_mul_f16 %0, %8, %16\n\t"
        "v_pk_mul_f16 %1, %9, %17\n\t"
        "v_pk_mul_f16 %2, %10, %18\n\t"
        "v_pk_mul_f16 %3, %11, %19\n\t"
        "v_pk_fma_f16 %4, %8, %16, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
        "v_pk_fma_f16 %5, %9, %17, %1 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
        "v_pk_fma_f16 %6, %10, %18, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
        "v_pk_fma_f16 %7, %11, %19, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
        "v_pk_fma_f16 %4, %8, %20, %4\n\t"
        "v_pk_fma_f16 %5, %9, %21, %5\n\t"
        "v_pk_fma_f16 %6, %10, %22, %6\n\t"
        "v_pk_fma_f16 %7, %11, %23, %7\n\t"
        "v_pk_fma_f16 %4, %12, %16, %4\n\t"
        "v_pk_fma_f16 %5, %13, %17, %5\n\t"
        "v_pk_fma_f16 %6, %14, %18, %6\n\t"
        "v_pk_fma_f16 %7, %15, %19, %7"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
        : "v"(ph.x), "v"(ph.y), "v"(ph.z), "v"(ph.w), "v"(pl.x), "v"(pl.y), "v"(pl.z), "v"(pl.w), "v"(ah.x), "v"(ah.y),
          "v"(ah.z), "v"(ah.w), "v"(al.x), "v"(al.y), "v"(al.z), "v"(al.w));
    hi = u32x4{h0, h1, h2, h3};
    lo = u32x4{q0, q1, q2, q3};
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
// ... with two coefficient columns per sweep: 3 instructions per step
__device__ __forceinline__ void wait_lds_dma_but_newest3() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
// workgroup barrier that leaves global traffic (LDS-DMA, flush atomics) in flight
__device__ __forceinline__ void barrier_lds_only() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


} // namespace nfft
