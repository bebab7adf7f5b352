// Microbenchmark: LDS instruction throughput on gfx950 (developer tool; not part of the product).
// Each wave issues a long unrolled stream of one DS instruction on conflict-free, lane-linear addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 16;

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int stride_mode)
{
    __shared__ float lds[16384];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 256) lds[i] = 0.f;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    // wave-private 4 KiB region (1024 floats); lane-linear addresses
    unsigned base = (wave * 2048 + (stride_mode == 0 ? lane : stride_mode == 1 ? (lane * 2) % 64 + lane / 32 : lane * 32 % 1024)) * 4;
    if (OP == 3 || OP == 4) base = (wave * 2048 + lane * 2) * 4;  // 8-byte ops
    float v = 1.0f + lane;
    float acc = 0.f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned a = base + ((u & 7) * 256);
            if (OP == 0) asm volatile("ds_add_f32 %0, %1" ::"v"(a), "v"(v) : "memory");
            if (OP == 1) asm volatile("ds_add_u32 %0, %1" ::"v"(a), "v"(v) : "memory");
            if (OP == 2) asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
            if (OP == 3) { double dv = v; asm volatile("ds_add_u64 %0, %1" ::"v"(a), "v"(dv) : "memory"); }
            if (OP == 4) { double dv = v; asm volatile("ds_add_f64 %0, %1" ::"v"(a), "v"(dv) : "memory"); }
            if (OP == 5) { float r; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory"); acc += r; }
            if (OP == 6) { float r; asm volatile("ds_add_rtn_f32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a), "v"(v) : "memory"); acc += r; }
            if (OP == 7) asm volatile("ds_max_f32 %0, %1" ::"v"(a), "v"(v) : "memory");
            if (OP == 8) asm volatile("ds_pk_add_f16 %0, %1" ::"v"(a), "v"(v) : "memory");
            if (OP == 9) asm volatile("ds_pk_add_bf16 %0, %1" ::"v"(a), "v"(v) : "memory");
            if (OP == 10) asm volatile("ds_max_u32 %0, %1" ::"v"(a), "v"(v) : "memory");
            if (OP == 11) { float r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(a) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1\n ds_write_b32 %2, %0" : "+v"(r) : "v"(v), "v"(a) : "memory"); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + tid] = lds[tid] + acc;
}

template <int OP>
void run(const char *name, int stride_mode = 0)
{
    float *out;
    const int blocks = 256 * 4;
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    k<OP><<<blocks, 256>>>(out, stride_mode);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    k<OP><<<blocks, 256>>>(out, stride_mode);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    // per CU: blocks/256 sequential-ish rounds (4 blocks per CU could co-reside: LDS 64 KB each -> 2 per CU)
    const double wave_instr = (double)blocks * 4 * ITERS * UNROLL;
    const double per_cu_per_s = wave_instr / 256.0 / (ms * 1e-3);
    printf("%-28s stride_mode %d: %8.3f ms  -> %.2f cycles per wave-instruction per CU @2.4GHz (%.1f lanes/clk/CU)\n", name,
           stride_mode, ms, 2.4e9 / per_cu_per_s, 64.0 * per_cu_per_s / 2.4e9);
    CHECK(hipFree(out));
}

int main()
{
    run<2>("ds_write_b32");
    run<5>("ds_read_b32 (+wait each)");
    run<0>("ds_add_f32");
    run<0>("ds_add_f32", 1);
    run<0>("ds_add_f32", 2);
    run<1>("ds_add_u32");
    run<1>("ds_add_u32", 2);
    run<3>("ds_add_u64");
    run<4>("ds_add_f64");
    run<6>("ds_add_rtn_f32 (+wait each)");
    run<7>("ds_max_f32");
    run<10>("ds_max_u32");
    run<8>("ds_pk_add_f16");
    run<9>("ds_pk_add_bf16");
    run<11>("read+add+write (RMW)");
    return 0;
}
