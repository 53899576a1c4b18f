// hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_mb.hip -o tools/_tmp/mfma_shape_mb && tools/_tmp/mfma_shape_mb
// Which fp32 MFMA shape does the chip clock higher on?  (MI355X_MICROARCH.md, DVFS give-back item 7: on bf16 the 16x16
// shape ran 1.12-1.15x the FLOP/s of the 32x32 one at equal cycles.)  One 64 x 64 output tile per wave, random operands,
// fragments re-read from LDS by ds_read_b128 every K step (or kept in registers), two waves per SIMD; reports wall
// TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// BAR: a workgroup barrier every 4 iterations (64 MFMAs of 32x32x2 per wave = one 32-float K stage of the GEMM); NT: 512 =
// one 8-wave workgroup per CU, 256 = two 4-wave workgroups per CU (both: two waves per SIMD)
template <int SHAPE, bool LDS, int NT = 512, bool BAR = false>
__global__ __launch_bounds__(NT, 2) void k(const float* rnd, float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[16384];      // 64 KB of operands
    for (int i = threadIdx.x; i < 16384; i += NT) smem[i] = rnd[(blockIdx.x * 977 + i) & 65535];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
        f32x4 a[2], b[2];
        const f32x4* s4 = reinterpret_cast<const f32x4*>(smem);
        const int base = (wave & 3) * 256 + lane;
        for (int i = 0; i < 2; ++i) { a[i] = s4[base + i * 64]; b[i] = s4[base + 128 + i * 64]; }
        for (int it = 0; it < iters; ++it) {
            if (LDS) {
                const int o = base + ((it & 3) << 10);
                for (int i = 0; i < 2; ++i) { a[i] = s4[o + i * 64]; b[i] = s4[o + 128 + i * 64]; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
            if (BAR && (it & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int g = 0; g < 16; ++g) s += acc[i][j][g];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 a[4], b[4];
        const f32x4* s4 = reinterpret_cast<const f32x4*>(smem);
        const int base = (wave & 3) * 256 + lane;
        for (int i = 0; i < 4; ++i) { a[i] = s4[(base + i * 64) & 4095]; b[i] = s4[(base + 512 + i * 64) & 4095]; }
        // one iteration = the same 64 x 64 x 8 update as the 32x32 loop's: 16 blocks x 2 of the 4 k-quads per read ...
        // (4 reads of 4 k-values per row block cover 16 k: so an iteration of 64 MFMAs = TWO iterations of the loop above)
        for (int it = 0; it < iters; it += 2) {
            if (LDS) {
                const int o = base + ((it & 2) << 10);
                for (int i = 0; i < 4; ++i) { a[i] = s4[(o + i * 64) & 4095]; b[i] = s4[(o + 512 + i * 64) & 4095]; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int g = 0; g < 4; ++g) s += acc[i][j][g];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (threadIdx.x == 0) { atomicAdd(clk, t1 - t0); atomicAdd(clk + 1, r1 - r0); }
}

template <int SHAPE, bool LDS, int NT = 512, bool BAR = false>
void run(const char* what, const float* rnd, float* out, unsigned long long* clk) {
    const int iters = 40000, grid = 256 * 512 / NT;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {                      // the third run is reported (clocks settled)
        hipMemset(clk, 0, 16);
        hipEventRecord(e0);
        k<SHAPE, LDS, NT, BAR><<<grid, NT>>>(rnd, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)iters * 16 * 4096.0 * (NT / 64) * grid;    // per wave and iteration: 16 MFMAs of 32x32x2 (or 32 of 16x16x4)
    printf("%-40s %8.1f us  %6.1f TFLOP/s  clock %.3f GHz\n", what, ms * 1e3, flops / (ms * 1e-3) / 1e12, 0.1 * (double)h[0] / (double)h[1]);
}

int main() {
    float* rnd; float* out; unsigned long long* clk;
    hipMalloc(&rnd, 65536 * 4); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&clk, 16);
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(rnd, h, 65536 * 4, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<32, false>("32x32x2, operands in registers", rnd, out, clk);
        run<16, false>("16x16x4, operands in registers", rnd, out, clk);
        run<32, true>("32x32x2, fragments from LDS each step", rnd, out, clk);
        run<16, true>("16x16x4, fragments from LDS each step", rnd, out, clk);
        run<32, true, 512, true>("32x32x2, LDS, 8-wave WG, barrier / 64 MFMA", rnd, out, clk);
        run<32, true, 256, false>("32x32x2, LDS, 2 x 4-wave WG, no barrier", rnd, out, clk);
        run<32, true, 256, true>("32x32x2, LDS, 2 x 4-wave WG, barrier / 64", rnd, out, clk);
    }
    return 0;
}
