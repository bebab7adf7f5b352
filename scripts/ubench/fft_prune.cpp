// Experiment: full 3-D R2C vs (1-D R2C rows + pruned strided 2-D C2C) with rocFFT (developer tool).
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define RC(x) do { rocfft_status s = (x); if (s != rocfft_status_success) { printf("%s: status %d\n", #x, (int)s); exit(1); } } while (0)

struct Plan { rocfft_plan p; rocfft_execution_info info; void *work; };
Plan finish(rocfft_plan p) {
    Plan P{p, nullptr, nullptr};
    size_t wb = 0; RC(rocfft_plan_get_work_buffer_size(p, &wb));
    RC(rocfft_execution_info_create(&P.info));
    if (wb) { CHECK(hipMalloc(&P.work, wb)); RC(rocfft_execution_info_set_work_buffer(P.info, P.work, wb)); }
    printf("   work buffer %.1f MB\n", wb / 1e6);
    return P;
}
float time_exec(Plan &P, void *in, void *out, int reps = 10) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    void *ins[1] = {in}, *outs[1] = {out};
    RC(rocfft_execute(P.p, ins, outs, P.info)); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) RC(rocfft_execute(P.p, ins, outs, P.info));
    CHECK(hipEventRecord(b)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main(int argc, char **argv) {
    const size_t M = argc > 1 ? atoi(argv[1]) : 512, Mh = M / 2 + 1, KEEP = M / 4 + 1;
    rocfft_setup();
    float *grid; float2 *spec;
    CHECK(hipMalloc(&grid, M * M * M * 4)); CHECK(hipMalloc(&spec, M * M * Mh * 8));
    CHECK(hipMemset(grid, 0, M * M * M * 4));
    // A: full 3-D R2C
    { size_t len[3] = {M, M, M}; rocfft_plan p;
      RC(rocfft_plan_create(&p, rocfft_placement_notinplace, rocfft_transform_type_real_forward, rocfft_precision_single, 3, len, 1, nullptr));
      Plan P = finish(p); printf("A full 3-D R2C: %.3f ms\n", time_exec(P, grid, spec)); }
    // B1: batched 1-D R2C along the last axis
    { size_t len[1] = {M}; rocfft_plan p;
      RC(rocfft_plan_create(&p, rocfft_placement_notinplace, rocfft_transform_type_real_forward, rocfft_precision_single, 1, len, M * M, nullptr));
      Plan P = finish(p); printf("B1 1-D R2C x M^2 rows: %.3f ms\n", time_exec(P, grid, spec)); }
    // B2: in-place 2-D C2C over (axis0, axis1), strides (M*Mh, Mh), batch over the kept kappa2 columns (distance 1)
    for (size_t keep : {KEEP, Mh}) {
      size_t len[2] = {M, M}; size_t str[2] = {Mh, M * Mh}; size_t off[1] = {0};
      rocfft_plan_description d; RC(rocfft_plan_description_create(&d));
      RC(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved,
                                                 off, off, 2, str, 1, 2, str, 1));
      rocfft_plan p;
      RC(rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_single, 2, len, keep, d));
      Plan P = finish(p); printf("B2 strided 2-D C2C, %zu columns: %.3f ms\n", keep, time_exec(P, spec, spec)); }
    // C: two 1-D strided passes (axis 1 then axis 0) batched over kept columns, looped over the other axis is not expressible;
    //    instead: 1-D C2C along axis 0 (stride M*Mh) batched over (axis1, kept kappa2) is not a single-distance batch either.
    // D: full 3-D C2R for comparison
    { size_t len[3] = {M, M, M}; rocfft_plan p;
      RC(rocfft_plan_create(&p, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, rocfft_precision_single, 3, len, 1, nullptr));
      Plan P = finish(p); printf("D full 3-D C2R: %.3f ms\n", time_exec(P, spec, grid)); }
    return 0;
}
