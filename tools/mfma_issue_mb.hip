// hipcc -O3 --offload-arch=gfx950 tools/mfma_issue_mb.hip -o mfma_mb && ./mfma_mb   (results: profiles/r2_mfma_issue_mb.log)
// microbenchmark: fp32 MFMA 32x32x2 issue rate vs waves per SIMD and dependency pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int PAT, int NT>
__global__ __launch_bounds__(NT, 1) void k(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;
    float av[4] = {a, a + 1, a + 2, a + 3}, bv[4] = {b, b + 1, b + 2, b + 3};
    for (int it = 0; it < iters; ++it) {
        if (PAT == 0) {          // 4 dependent MFMAs per accumulator, accumulators in turn
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[(q + u) & 3], acc[i], 0, 0, 0);
        } else {                 // accumulators interleaved
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[(q + u) & 3], acc[i], 0, 0, 0);
        }
        if (PAT == 2) __builtin_amdgcn_s_barrier();
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int g = 0; g < 16; ++g) s += acc[i][g];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int PAT, int NT>
void run(const char* what, int grid) {
    float* out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 3000;
    k<PAT, NT><<<grid, NT>>>(out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<PAT, NT><<<grid, NT>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 32 * (NT / 64) / 4.0 * (grid / 256.0);
    printf("%-44s grid %4d x %4d: %.1f us, %.1f cycles/MFMA/SIMD @2.4GHz, %.1f TFLOP/s\n", what, grid, NT, ms * 1e3,
           ms * 1e-3 * 2.4e9 / mfma_per_simd, (double)iters * 32 * (NT / 64) * grid * 4096.0 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<0, 256>("dependent x4, 1 wave/SIMD", 256);
    run<1, 256>("interleaved, 1 wave/SIMD", 256);
    run<0, 512>("dependent x4, 2 waves/SIMD", 256);
    run<1, 512>("interleaved, 2 waves/SIMD", 256);
    run<2, 512>("interleaved + barrier/32, 2 waves/SIMD", 256);
    run<0, 768>("dependent x4, 3 waves/SIMD", 256);
    run<1, 1024>("interleaved, 4 waves/SIMD", 256);
    run<0, 256>("dependent x4, 1 wave/SIMD, 2 WG/CU", 512);
    run<0, 256>("dependent x4, 1 wave/SIMD, 3 WG/CU", 768);
    return 0;
}
