// Internal launch interface between the C-ABI layer (api.hip) and the kernel files.
#pragma once
#include "../../include/nfft_hip.h"
#include "common.h"

namespace nfft {

// binning.hip
int launch_plan_points(const Geom &g, const PlanLayout &L, const float *pos, const int64_t *batch, int64_t n, int64_t B,
                       void *plan, hipStream_t stream);
// Seal of a point plan: a 64-bit checksum of the points (and the batch vector) it was built from, left in the plan's seal
// block (common.h: PlanLayout::off_seal) by the pass that counts the points -- no pass of its own.  verify: recompute the
// checksum from the arrays as they are now into accumulator `slot` and raise kFaultStalePlan in the device's status
// block when it differs.
int launch_points_verify(const float *pos, const int64_t *batch, int64_t n, int dim, void *seal, int slot, hipStream_t stream);
// xs[c * n + slot] = xr[perm[slot] * cols + c]
int launch_gather_rows(const Geom &g, const PlanLayout &L, const void *plan, int64_t n, const float *xr, int64_t cols,
                       float *xs, hipStream_t stream);

// spread.hip: grid[p, :] += ... for local planes p in [0, nplanes); global plane plane0 + p = b * Cr + cr
int launch_spread(const Geom &g, const PlanLayout &L, const void *plan, const float *xr, const float *xs, int64_t n, int64_t Cr,
                  int64_t plane0, int64_t nplanes, float *grid, hipStream_t stream);

// spread_reg.hip: register-tile spreading for 3-D grids (no atomics; writes every cell of the planes, so the
// grid needs no zero-fill).  Same arguments as launch_spread.
bool spread_reg_supported(const Geom &g);
int launch_spread_reg(const Geom &g, const PlanLayout &L, const void *plan, const float *xs, int64_t n, int64_t Cr,
                      int64_t plane0, int64_t nplanes, float *grid, hipStream_t stream);

// spread_mfma.hip: matrix-core spreading for the wide 3-D tiling (g.wide)
bool spread_mfma_supported(const Geom &g);
// largest |x| per (point set, real column) plane -> xmax[B * Cr] (bit patterns): the kernel's operand scales.  Reads the
// caller's row-major [point][Cr] array; set boundaries come from the halo plan (the batch vector is sorted)
int launch_plane_absmax(const Geom &g_halo, const PlanLayout &L_halo, const void *plan_halo, const float *xr, int64_t n,
                        int64_t B, int64_t Cr, unsigned *xmax, hipStream_t stream);
// coefficients: xr (row-major [point][Cr], read through the index in the plan records) or, when xr == nullptr, xs (copy in
// plan order, planar, stride L.cap: gather_rows)
int launch_spread_mfma(const Geom &g, const PlanLayout &L, const void *plan, const float *xr, const float *xs,
                       const unsigned *xmax, int64_t n, int64_t Cr, int64_t plane0, int64_t nplanes, float *grid,
                       hipStream_t stream);

// interp.hip: yr[perm[slot] * Cr + cr] = sum over taps of grid[p, ...]
int launch_interp(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                  int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream);
// matrix-core gather for the wide 3-D tiling (interp_mfma.hip); NFFT_HIP_GATHER=lds keeps launch_interp
bool interp_mfma_supported(const Geom &g);
int launch_interp_mfma(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                       int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream);

// streamed variant (interp_stream.hip): producer waves feed the plane ring, consumer waves pull blocks from a queue
bool interp_stream_supported(const Geom &g);
bool interp_stream_pays(const Geom &g, const PlanLayout &L, int64_t n);  // big work items only
int launch_interp_stream(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                         int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream);
// several coefficient columns per workgroup, one wave per column (interp_cols.hip): the point-side operands are built
// once for 8 columns, each wave streams its own column's planes from global memory
bool interp_cols_supported(const Geom &g, int64_t Cr);
int launch_interp_cols(const Geom &g, const PlanLayout &L, const void *plan, const float *grid, int64_t n, int64_t Cr,
                       int64_t plane0, int64_t nplanes, float *yr, hipStream_t stream);

// spectral.hip
// adjoint roll-off: spec = R2C(grid) per real plane [nplanes, M^(d-1) * (M/2+1)] complex -> y [B, N^d, C]
// mult / mult_kind: optional per-frequency factor [N^d] multiplied in on the way out (fastsum; 0 none, 1 float, 2 float2)
int launch_deconv_adjoint(const Geom &g, const float2 *spec, int64_t C, int x_is_complex, int real_output,
                          int64_t plane0, int64_t nplanes, void *y, const void *mult, int mult_kind, hipStream_t stream);
// forward roll-off: xhat [B, N^d, C] -> Hermitian half-spectra of the real planes of g
int launch_deconv_forward(const Geom &g, const void *xhat, int64_t C, int x_is_complex, int real_output,
                          int64_t plane0, int64_t nplanes, float2 *spec, hipStream_t stream);

// fft.cpp (rocFFT, plans cached per (kind, dim, M, batch) and device)
// kR2C / kC2R: full dim-dimensional real transforms of every plane; k*Rows: 1-D transforms of every grid row
// (last axis only), used together with the pruned column passes of colfft.hip
// kC2CForward: in-place complex forward transform of an N^dim array (coefficient set-up, coeffs.hip)
enum FftKind { kR2C = 0, kC2R = 1, kR2CRows = 2, kC2RRows = 3, kC2CForward = 4 };
int64_t fft_work_bytes(FftKind kind, int dim, int M, int64_t nplanes);
int fft_execute(FftKind kind, int dim, int M, int64_t nplanes, void *in, void *out, void *work, int64_t work_bytes,
                hipStream_t stream);

// colfft.hip: pruned strided passes over axes 1 and 0 fused with the roll-off (3-D, power-of-two M)
bool colfft_supported(const Geom &g);
int64_t colfft_scratch_bytes(const Geom &g, int64_t nplanes);
// `compact`: the axis-2 half spectrum holds only the N/2+1 kept columns per row (own row passes below) instead of
// rocFFT's M/2+1
int launch_colfft_adjoint(const Geom &g, const float2 *spec, bool compact, void *scratch, int64_t scratch_planes,
                          int64_t C, int x_is_complex, int real_output, int64_t plane0, int64_t nplanes, void *y,
                          const void *mult, int mult_kind, hipStream_t stream);
int launch_colfft_forward(const Geom &g, const void *xhat, void *scratch, int64_t scratch_planes, int64_t C,
                          int x_is_complex, int real_output, int64_t plane0, int64_t nplanes, float2 *spec, bool compact,
                          hipStream_t stream);
// Column-innermost pipeline for several coefficient columns (3-D, M = 128 .. 1024): the planes of a chunk travel in groups
// of 16 with the plane index innermost in both intermediate arrays, so the last adjoint pass writes (the first forward pass
// reads) the reference's [B, N^3, C] layout in place -- no planar copy, no transposes.  The buffers hold
// colfft_ci_planes(nplanes) planes (whole groups).  Row passes included.
bool colfft_ci_supported(const Geom &g);
int64_t colfft_ci_planes(int64_t nplanes);
int launch_row_r2c_ci(const Geom &g, const float *grid, int64_t nplanes, float2 *spec, hipStream_t stream);
int launch_row_c2r_ci(const Geom &g, const float2 *spec, int64_t nplanes, float *grid, hipStream_t stream);
int launch_colfft_adjoint_ci(const Geom &g, const float2 *spec, void *scratch, int64_t C, int x_is_complex,
                             int real_output, int64_t plane0, int64_t nplanes, void *y, const void *mult, int mult_kind,
                             hipStream_t stream);
int launch_colfft_forward_ci(const Geom &g, const void *xhat, float2 *spec, void *scratch, int64_t C, int x_is_complex,
                             int real_output, int64_t plane0, int64_t nplanes, hipStream_t stream);
// planar [ncols][N^d] <-> column-interleaved [B, N^d, C] copies (tiled transposes) for the column passes with C > 1
int launch_column_layout(bool to_interleaved, const void *src, void *dst, int64_t K, int64_t C, int64_t col0,
                         int64_t ncols, int elem_bytes, hipStream_t stream);
// pruned real <-> half-complex row passes (axis 2), one wave per row; M in {128 .. 1024}
bool rowfft_supported(const Geom &g);
int launch_row_r2c(const Geom &g, const float *grid, void *scratch, int64_t scratch_planes, int64_t nplanes,
                   float2 *spec, hipStream_t stream);
int launch_row_c2r(const Geom &g, const float2 *spec, void *scratch, int64_t scratch_planes, int64_t nplanes,
                   float *grid, hipStream_t stream);

// smallgrid.hip: transforms whose oversampled grid (<= 4096 cells) fits one workgroup's LDS -- one kernel per direction,
// no point plan
bool small_grid_supported(const nfft_hip_problem *p);
int launch_small_grid_adjoint(const nfft_hip_problem *p, const float *pos, const int64_t *batch, const void *x, int x_is_complex,
                           int real_output, void *y, const void *mult, int mult_kind, hipStream_t stream);
int launch_small_grid_forward(const nfft_hip_problem *p, const float *pos, const int64_t *batch, const void *xhat, int x_is_complex,
                           int real_output, void *y, hipStream_t stream);

// api.hip: optional per-stage GPU timing with HIP events on the caller's stream (nfft_hip_profile_*)
enum Stage { kStagePlan = 0, kStageGather, kStageZero, kStageSpread, kStageFft, kStageDeconv, kStageInterp, kNumStages };
struct StageTimer {
    StageTimer(Stage stage, hipStream_t stream);
    ~StageTimer();
    int slot;
    hipStream_t stream;
};

} // namespace nfft
