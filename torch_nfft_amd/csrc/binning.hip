// Point plan: counting sort of the points by (batch, pencil, chunk) tile.
//
// Replaces the reference's per-point HBM temporaries (shifts: spatial_window_operations.cu:38-61,
// psi: :68-97; allocated at core_cuda.cu:188-211 and re-read 2m+2 times per axis by the spreading
// kernel).  Here the points themselves are reordered once (three streaming passes over n points) and
// every consumer re-derives cell index and window weights in registers.
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "kernels.h"

namespace nfft {

PlanLayout plan_layout(const Geom &g, int64_t n, int64_t B)
{
    PlanLayout L;
    L.ntiles = (int64_t)g.tiles_per_batch * B;
    size_t scan_bytes = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int *)nullptr, (int *)nullptr, (int)(L.ntiles + 1));
    L.scan_bytes = (int64_t)scan_bytes;
    int64_t o = 0;
    L.off_offsets = o; o = align_up(o + (L.ntiles + 1) * 4, 256);
    L.off_cursor = o;  o = align_up(o + (L.ntiles + 1) * 4, 256);
    L.off_perm = o;    o = align_up(o + n * 4, 256);
    L.off_spos = o;    o = align_up(o + n * g.dim * 4, 256);
    L.off_scan = o;    o = align_up(o + L.scan_bytes, 256);
    L.total = o;
    return L;
}

__device__ __forceinline__ int point_tile(const Geom &g, const float *__restrict__ pos, const int64_t *__restrict__ batch,
                                          int64_t i, int64_t B)
{
    int cell[3] = {0, 0, 0};
    for (int u = 0; u < g.dim; ++u) {
        float fr;
        split_cell(pos[i * g.dim + u], g.M, cell[u + 3 - g.dim], fr);
    }
    int64_t b = batch ? batch[i] : 0;
    b = b < 0 ? 0 : (b >= B ? B - 1 : b);
    return (int)b * g.tiles_per_batch + tile_of_cells(g, cell);
}

__global__ void __launch_bounds__(256) bin_count_kernel(Geom g, const float *__restrict__ pos,
                                                       const int64_t *__restrict__ batch, int64_t n, int64_t B,
                                                       int *__restrict__ count)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&count[point_tile(g, pos, batch, i, B)], 1);
}

__global__ void __launch_bounds__(256) bin_fill_kernel(Geom g, const float *__restrict__ pos,
                                                      const int64_t *__restrict__ batch, int64_t n, int64_t B,
                                                      const int *__restrict__ offsets, int *__restrict__ cursor,
                                                      int *__restrict__ perm, float *__restrict__ spos)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = point_tile(g, pos, batch, i, B);
        const int slot = offsets[t] + atomicAdd(&cursor[t], 1);
        perm[slot] = (int)i;
        for (int u = 0; u < g.dim; ++u) spos[(int64_t)slot * g.dim + u] = pos[i * g.dim + u];
    }
}

// xs[c, slot] = xr[perm[slot], c]  (tile-ordered, column-major copy of the real coefficient columns)
__global__ void __launch_bounds__(256) gather_rows_kernel(const int *__restrict__ perm, const float *__restrict__ xr,
                                                         float *__restrict__ xs, int64_t n, int64_t cols)
{
    const int64_t total = n * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = e / cols, c = e - slot * cols;
        xs[c * n + slot] = xr[(int64_t)perm[slot] * cols + c];
    }
}

static inline int grid_for(int64_t work, int block)
{
    int64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 32) g = 256 * 32;
    return (int)g;
}

int launch_plan_points(const Geom &g, const PlanLayout &L, const float *pos, const int64_t *batch, int64_t n, int64_t B,
                       void *plan, hipStream_t stream)
{
    char *base = (char *)plan;
    int *offsets = (int *)(base + L.off_offsets);
    int *cursor = (int *)(base + L.off_cursor);
    int *perm = (int *)(base + L.off_perm);
    float *spos = (float *)(base + L.off_spos);
    // counts are accumulated in `cursor`, scanned into `offsets`, then `cursor` restarts at zero
    NFFT_HIP_CHECK(hipMemsetAsync(cursor, 0, (L.ntiles + 1) * 4, stream));
    if (n > 0) {
        hipLaunchKernelGGL(bin_count_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, g, pos, batch, n, B, cursor);
    }
    size_t scan_bytes = (size_t)L.scan_bytes;
    NFFT_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(base + L.off_scan, scan_bytes, cursor, offsets,
                                                    (int)(L.ntiles + 1), stream));
    NFFT_HIP_CHECK(hipMemsetAsync(cursor, 0, (L.ntiles + 1) * 4, stream));
    if (n > 0) {
        hipLaunchKernelGGL(bin_fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, g, pos, batch, n, B, offsets,
                           cursor, perm, spos);
    }
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gather_rows(const Geom &g, const PlanLayout &L, const void *plan, int64_t n, const float *xr, int64_t cols,
                       float *xs, hipStream_t stream)
{
    const int *perm = (const int *)((const char *)plan + L.off_perm);
    if (n * cols > 0)
        hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * cols, 256)), dim3(256), 0, stream, perm, xr, xs, n, cols);
    NFFT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace nfft
