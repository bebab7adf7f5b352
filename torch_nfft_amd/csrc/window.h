// Tile/ring configuration and per-lane tap bookkeeping shared by the spreading and
// interpolation kernels (device only).
#pragma once
#include "common.h"

namespace nfft {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DIM, int W>
struct TapCfg {
    static constexpr TileCfg tc = tile_cfg(DIM, W);
    static constexpr int T1 = tc.T1, T2 = tc.T2, NP = tc.NP, TC = tc.TC;
    static constexpr int W0 = DIM == 3 ? W : 1;   // taps along axis 0
    static constexpr int W1 = DIM >= 2 ? W : 1;   // taps along axis 1
    static constexpr int M0OFF = DIM == 3 ? (W / 2 - 1) : 0;  // cutoff m on axis 0 (0 when degenerate)
    static constexpr int P1 = T1 + W1 - 1;        // padded rows of a plane
    static constexpr int P2 = T2 + W - 1;         // padded columns of a plane
    // Row stride: T2 + W.  With T2 a multiple of 32 the stride is == W (mod 32), so the W1 x W taps of
    // one point, enumerated row-major over the lanes, hit 32 consecutive LDS banks per half-wave.
    static constexpr int S2 = T2 + W;
    static constexpr int S0 = P1 * S2;            // plane stride (floats)
    static constexpr int CELLS = NP * S0;         // resident cells
    static constexpr int TAPS12 = W1 * W;         // taps of one point inside a plane = lanes used
    static constexpr int PASSES = (TAPS12 + 63) / 64;
    static_assert(TC >= 1, "chunk must hold at least one plane");
    static_assert(NP == TC + W0 - 1, "resident planes = chunk + halo");
    static_assert(CELLS * 8 <= 160 * 1024, "LDS budget (8-byte cells)");
};

// Per-lane description of the in-plane taps (l1, l2) a lane evaluates in pass p: t = lane + 64 p.
template <int DIM, int W>
struct LaneTaps {
    using C = TapCfg<DIM, W>;
    float c1[C::PASSES];   // m - l1  (0 when axis 1 is degenerate)
    float c2[C::PASSES];   // m - l2
    int off[C::PASSES];    // l1 * S2 + l2
    bool valid[C::PASSES];
    __device__ __forceinline__ void init(int lane, int m)
    {
#pragma unroll
        for (int p = 0; p < C::PASSES; ++p) {
            const int t = lane + 64 * p;
            const int l1 = t / W, l2 = t - l1 * W;
            valid[p] = t < C::TAPS12;
            c1[p] = DIM >= 2 ? (float)(m - l1) : 0.0f;
            c2[p] = (float)(m - l2);
            off[p] = l1 * C::S2 + l2;
        }
    }
};

__device__ __forceinline__ float readlane_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// Per-point quantities computed with one lane per point (up to 64 points at a time).
template <int DIM, int W>
struct PointPrep {
    using C = TapCfg<DIM, W>;
    float f0, f1, f2;  // fractional offsets per internal axis
    int base;          // cell index of the window origin inside the resident planes:
                       // ((cell0 - chunk_base0) * P1 + (cell1 - tile_base1)) * S2 + (cell2 - tile_base2)
    __device__ __forceinline__ void load(const Geom &g, const float *__restrict__ spos, int64_t j, int tb0, int tb1,
                                         int tb2)
    {
        int c0 = 0, c1 = 0, c2 = 0;
        f0 = f1 = f2 = 0.0f;
        if (DIM == 3) {
            const f32x4 rec = *(const f32x4 *)(spos + j * 4);  // {p0, p1, p2, x}
            split_cell(rec.x, g.M, c0, f0);
            split_cell(rec.y, g.M, c1, f1);
            split_cell(rec.z, g.M, c2, f2);
        } else if (DIM == 2) {
            split_cell(spos[j * 2 + 0], g.M, c1, f1);
            split_cell(spos[j * 2 + 1], g.M, c2, f2);
        } else {
            split_cell(spos[j], g.M, c2, f2);
        }
        base = (c0 - tb0) * C::S0 + (c1 - tb1) * C::S2 + (c2 - tb2);
    }
    __device__ __forceinline__ void clear() { f0 = f1 = f2 = 0.0f; base = 0; }
};

} // namespace nfft
