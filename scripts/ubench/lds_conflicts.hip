// Microbenchmark: cost of ds_add_f64 / ds_read_b128 / ds_read_b32 under different per-lane address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 2000, UNROLL = 8;

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, const int *lane_off)
{
    __shared__ double lds[8192];  // 64 KiB
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) lds[i] = 0.0;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    // byte address inside a wave-private 16 KiB region
    const unsigned base = wave * 16384 + lane_off[lane];
    double dv = 1.0 + lane;
    float acc = 0.f;
    for (int it = 0; it < ITERS; ++it) {
        float4 r4[UNROLL]; float2 r2[UNROLL]; float r1[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned a = base + u * 1024;
            if (OP == 0) asm volatile("ds_add_f64 %0, %1" ::"v"(a), "v"(dv) : "memory");
            if (OP == 1) { float4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(a) : "memory"); r4[u] = r; }
            if (OP == 2) { float r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(a) : "memory"); r1[u] = r; }
            if (OP == 3) { float2 r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(a) : "memory"); r2[u] = r; }
        }
        if (OP != 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (OP == 1) acc += r4[u].x + r4[u].y + r4[u].z + r4[u].w;
                if (OP == 2) acc += r1[u];
                if (OP == 3) acc += r2[u].x + r2[u].y;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + tid] = (float)lds[tid] + acc;
}

template <int OP>
void run(const char *name, const char *pat, const std::vector<int> &off)
{
    float *out; int *d_off;
    const int blocks = 2048;
    CHECK(hipMalloc(&out, blocks * 256 * 4)); CHECK(hipMalloc(&d_off, 64 * 4));
    CHECK(hipMemcpy(d_off, off.data(), 64 * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<OP><<<blocks, 256>>>(out, d_off); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a)); k<OP><<<blocks, 256>>>(out, d_off); CHECK(hipEventRecord(b)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double per_cu_per_s = (double)blocks * 4 * ITERS * UNROLL / 256.0 / (ms * 1e-3);
    printf("%-16s %-34s %8.3f ms -> %6.2f cycles per wave-instruction per CU @2.4GHz\n", name, pat, ms, 2.4e9 / per_cu_per_s);
    CHECK(hipFree(out)); CHECK(hipFree(d_off));
}

int main()
{
    std::mt19937 rng(1);
    auto lin = [](int stride) { std::vector<int> v(64); for (int l = 0; l < 64; ++l) v[l] = l * stride; return v; };
    auto rnd = [&](int gran, int range) { std::vector<int> v(64); for (int l = 0; l < 64; ++l) v[l] = (int)(rng() % range) * gran; return v; };
    // distinct residues inside every group of G lanes, but otherwise random rows
    auto distinct = [&](int gran, int G, int nslots) { std::vector<int> v(64); for (int l = 0; l < 64; ++l) v[l] = ((l % G) + nslots * (int)(rng() % 4)) * gran; return v; };
    run<0>("ds_add_f64", "lane-linear 8 B", lin(8));
    run<0>("ds_add_f64", "random 8-B slots (of 128)", rnd(8, 128));
    run<0>("ds_add_f64", "distinct mod 16 per 16 lanes", distinct(8, 16, 16));
    run<0>("ds_add_f64", "distinct mod 32 per 32 lanes", distinct(8, 32, 32));
    run<0>("ds_add_f64", "distinct mod 16 per 32 lanes(2x)", distinct(8, 16, 16));
    run<1>("ds_read_b128", "lane-linear 16 B", lin(16));
    run<1>("ds_read_b128", "random 16-B slots (of 64)", rnd(16, 64));
    run<1>("ds_read_b128", "distinct mod 16 per 16 lanes", distinct(16, 16, 16));
    run<3>("ds_read_b64", "lane-linear 8 B", lin(8));
    run<3>("ds_read_b64", "random 8-B slots (of 128)", rnd(8, 128));
    run<2>("ds_read_b32", "lane-linear 4 B", lin(4));
    run<2>("ds_read_b32", "random 4-B slots (of 256)", rnd(4, 256));
    return 0;
}
