// Calibration of rocprofv3's FETCH_SIZE for the access patterns of the NFFT kernels (MI355X_MICROARCH.md, HBM: the
// counter reports half the bytes of wide coalesced reads on gfx950; "other access widths are uncalibrated").
// Every kernel reads a KNOWN number of bytes out of buffers far larger than the 256 MiB Infinity Cache; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- scripts/ubench/fetch_calib
// and compare the counter (KiB) with the byte counts this program prints.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));

// 16 bytes per lane, coalesced
__global__ void __launch_bounds__(256) stream_dwordx4(const f32x4 *__restrict__ a, size_t n4, float *out)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
// 4 bytes per lane, coalesced
__global__ void __launch_bounds__(256) stream_dword(const float *__restrict__ a, size_t n, float *out)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 123.456f) out[0] = s;
}
// 12 bytes per lane, coalesced (the plan's tile-ordered coordinates)
__global__ void __launch_bounds__(256) stream_12byte(const float *__restrict__ a, size_t n3, float *out)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (size_t)gridDim.x * blockDim.x) {
        f32x3 v;
        __builtin_memcpy(&v, a + 3 * i, 12);
        s += v.x + v.y + v.z;
    }
    if (s == 123.456f) out[0] = s;
}
// one dword per lane through LDS-DMA, coalesced (the spreading kernel's staging)
__global__ void __launch_bounds__(256) stream_lds_dma(const float *__restrict__ a, size_t n, float *out)
{
    __shared__ float land[256];
    float s = 0.f;
    const unsigned lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)&land[(threadIdx.x >> 6) * 64];
    const unsigned ldsu = __builtin_amdgcn_readfirstlane(lds);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "v"(a + i), "s"(ldsu) : "memory");
        s += land[threadIdx.x];
    }
    if (s == 123.456f) out[0] = s;
}
// 4-byte reads at random places of a big table through an index stream (the coefficient permutation): eight in flight
__global__ void __launch_bounds__(256) random_dword(const int *__restrict__ idx, const float *__restrict__ table, size_t n,
                                                    float *out)
{
    float s = 0.f;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 8 * step) {
        int j[8];
        for (int q = 0; q < 8; ++q) j[q] = i0 + q * step < n ? idx[i0 + q * step] : -1;
        for (int q = 0; q < 8; ++q) s += j[q] >= 0 ? table[j[q]] : 0.f;
    }
    if (s == 123.456f) out[0] = s;
}
// the same with TWO reads per index, 64 bytes apart inside one aligned 128-byte line: if the counter does not grow against
// random_dword, a random read fetches (and the counter under-reports) a whole 128-byte line; if it doubles, the memory-side
// request is 64 bytes and the raw counter is exact for this pattern
__global__ void __launch_bounds__(256) random_pair64(const int *__restrict__ idx, const float *__restrict__ table, size_t n,
                                                     float *out)
{
    float s = 0.f;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 8 * step) {
        int j[8];
        for (int q = 0; q < 8; ++q) j[q] = i0 + q * step < n ? (idx[i0 + q * step] & ~31) : -1;  // 128-byte aligned
        for (int q = 0; q < 8; ++q) s += j[q] >= 0 ? table[j[q]] + table[j[q] + 16] : 0.f;
    }
    if (s == 123.456f) out[0] = s;
}
// ... and 32 bytes apart inside one aligned 64-byte half line
__global__ void __launch_bounds__(256) random_pair32(const int *__restrict__ idx, const float *__restrict__ table, size_t n,
                                                     float *out)
{
    float s = 0.f;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 8 * step) {
        int j[8];
        for (int q = 0; q < 8; ++q) j[q] = i0 + q * step < n ? (idx[i0 + q * step] & ~31) : -1;
        for (int q = 0; q < 8; ++q) s += j[q] >= 0 ? table[j[q]] + table[j[q] + 8] : 0.f;
    }
    if (s == 123.456f) out[0] = s;
}
// the index stream alone (to subtract)
__global__ void __launch_bounds__(256) index_stream(const int *__restrict__ idx, size_t n, float *out)
{
    int s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += idx[i];
    if (s == 123456789) out[0] = (float)s;
}
// float atomics, one dword per lane, coalesced rows (the spreading kernel's flush) -- for WRITE_SIZE
__global__ void __launch_bounds__(256) atomic_rows(float *__restrict__ a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        atomicAdd(a + i, 1.0f);
}
// scattered 4-byte stores through an index stream (the gather's y[perm[i]]) -- for WRITE_SIZE
__global__ void __launch_bounds__(256) random_store(const int *__restrict__ idx, float *__restrict__ table, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        table[idx[i]] = 1.0f;
}

int main()
{
    const size_t big = (size_t)1 << 28;     // floats: 1 GiB
    const size_t nrand = 10000000;          // random reads (config C3's point count)
    const size_t table = (size_t)1 << 27;   // floats: 512 MiB table for the random reads
    float *a = nullptr, *out = nullptr;
    int *idx = nullptr;
    CHECK(hipMalloc(&a, big * 4));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMalloc(&idx, nrand * 4));
    CHECK(hipMemset(a, 0, big * 4));
    std::vector<int> h(nrand);
    unsigned long long st = 88172645463325252ull;
    for (size_t i = 0; i < nrand; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        h[i] = (int)(st % table);
    }
    CHECK(hipMemcpy(idx, h.data(), nrand * 4, hipMemcpyHostToDevice));
    const int blocks = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_dwordx4, dim3(blocks), dim3(256), 0, 0, (const f32x4 *)a, big / 4, out);
        hipLaunchKernelGGL(stream_dword, dim3(blocks), dim3(256), 0, 0, a, big, out);
        hipLaunchKernelGGL(stream_12byte, dim3(blocks), dim3(256), 0, 0, a, big / 3, out);
        hipLaunchKernelGGL(stream_lds_dma, dim3(blocks), dim3(256), 0, 0, a, big / 4, out);
        hipLaunchKernelGGL(index_stream, dim3(blocks), dim3(256), 0, 0, idx, nrand, out);
        hipLaunchKernelGGL(random_dword, dim3(blocks), dim3(256), 0, 0, idx, a, nrand, out);
        hipLaunchKernelGGL(random_pair64, dim3(blocks), dim3(256), 0, 0, idx, a, nrand, out);
        hipLaunchKernelGGL(random_pair32, dim3(blocks), dim3(256), 0, 0, idx, a, nrand, out);
        hipLaunchKernelGGL(atomic_rows, dim3(blocks), dim3(256), 0, 0, a, big / 4);
        hipLaunchKernelGGL(random_store, dim3(blocks), dim3(256), 0, 0, idx, a, nrand);
        CHECK(hipDeviceSynchronize());
    }
    printf("bytes requested per launch (KiB):\n");
    printf("  stream_dwordx4  %zu\n  stream_dword    %zu\n  stream_12byte   %zu\n  stream_lds_dma  %zu\n", big * 4 / 1024,
           big * 4 / 1024, (big / 3) * 12 / 1024, big / 4 * 4 / 1024);
    printf("  index_stream    %zu\n  random_dword    %zu index stream + %zu reads x {4 B payload, 32 B, 64 B, 128 B} = %zu / %zu / %zu / %zu\n",
           nrand * 4 / 1024, nrand * 4 / 1024, nrand, nrand * 4 / 1024, nrand * 32 / 1024, nrand * 64 / 1024, nrand * 128 / 1024);
    printf("  random_pair64 / random_pair32: as random_dword with a second read 64 / 32 bytes away in the same 128-byte line\n");
    printf("  atomic_rows     %zu written\n  random_store    %zu index stream read, %zu stores x {4, 32, 64 B} = %zu / %zu / %zu written\n",
           big / 4 * 4 / 1024, nrand * 4 / 1024, nrand, nrand * 4 / 1024, nrand * 32 / 1024, nrand * 64 / 1024);
    return 0;
}
