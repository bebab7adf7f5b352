// Micro-benchmark (developer tool): throughput of global float atomic adds by memory scope on gfx950.  Every workgroup
// adds rows of 64 floats (one 256-byte run per wave instruction, as the spreading kernel's flush does) into its OWN region
// of a large buffer, so that any scope is correct; reported: GB/s of atomic payload.
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics atomic_scope.hip -o atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int SCOPE>  // 0 agent, 1 workgroup, 2 plain store (reference), 3 wavefront
__global__ void __launch_bounds__(1024) flush_kernel(float *buf, int rows_per_wg, int reps)
{
    float *mine = buf + (size_t)blockIdx.x * rows_per_wg * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < reps; ++r)
        for (int row = wave; row < rows_per_wg; row += 16) {
            float *p = mine + (size_t)row * 64 + lane;
            const float v = 1.0f + lane;
            if (SCOPE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (SCOPE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (SCOPE == 3) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (SCOPE == 2) *p = v;
        }
}

template <int SCOPE>
static int run(const char *name, float *buf, int wgs, int rows, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(flush_kernel<SCOPE>, dim3(wgs), dim3(1024), 0, 0, buf, rows, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(flush_kernel<SCOPE>, dim3(wgs), dim3(1024), 0, 0, buf, rows, reps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * rows * 64 * 4 * reps;
    printf("%-28s %5d workgroups x %6d rows x %d reps: %8.3f ms  %8.1f GB/s\n", name, wgs, rows, reps, ms, bytes / ms / 1e6);
    return 0;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;  // 2 GiB
    float *buf;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    // (a) every cell touched once per launch: 2 GiB streamed; (b) a 64 MiB region revisited 16 times (L2 / MALL resident)
    for (int pass = 0; pass < 2; ++pass) {
        const int wgs = 2048;
        const int rows = pass == 0 ? (int)(bytes / 256 / wgs) : 128;
        const int reps = pass == 0 ? 1 : 32;
        run<0>("atomic add, agent scope", buf, wgs, rows, reps);
        run<1>("atomic add, workgroup scope", buf, wgs, rows, reps);
        run<3>("atomic add, wavefront scope", buf, wgs, rows, reps);
        run<2>("plain store", buf, wgs, rows, reps);
    }
    return 0;
}
