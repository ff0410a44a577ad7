// The two fp64 MFMA shapes of gfx950 side by side: v_mfma_f64_16x16x4_f64 (2 048 flop) and v_mfma_f64_4x4x4_4b_f64
// (four 4x4x4 products, 512 flop) -- issue rate with independent accumulators, a chain through the C operand, and a chain
// through the B operand (the recurrence of k_rsos's chain wave).  One question: is the small shape's flop rate at least the
// large one's (then a lower-triangular T costs 10 tile products instead of 16 and dependent steps become short)?
//   hipcc --offload-arch=gfx950 -O3 mfma64_shapes.hip -o mfma64_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

// MODE 0: 16x16x4, NACC independent accumulators    1: 16x16x4, chain through B (acc[0] of the last is B of the next)
// MODE 2: 4x4x4_4b, NACC independent accumulators   3: 4x4x4_4b, chain through B
template <int MODE, int NACC>
__global__ void k(long long* out, double* sink, int iters) {
    v4d acc[NACC];
    double acc1[NACC];
    for (int i = 0; i < NACC; ++i) {
        acc[i] = v4d{0, 0, 0, 0};
        acc1[i] = 0.0;
    }
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[(i + NACC - 1) % NACC][0], acc[i], 0, 0, 0);
            if (MODE == 2) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
            if (MODE == 3) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, acc1[(i + NACC - 1) % NACC], acc1[i], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc1[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE, int NACC>
void run(int waves, int grid) {
    long long* d;
    double* s;
    (void)hipMalloc(&d, 8 * 4096);
    (void)hipMalloc(&s, 8 * 1024 * 1024);
    const int iters = 2000;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(64 * waves), 0, 0, d, s, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(64 * waves), 0, 0, d, s, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    long long h[64];
    (void)hipMemcpy(h, d, 8 * waves, hipMemcpyDeviceToHost);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (MODE < 2 ? 2048.0 : 512.0);
    const double per_wave = (double)h[0] / (iters * NACC);
    static const char* names[] = {"16x16x4 independent", "16x16x4 chain via B", "4x4x4_4b independent", "4x4x4_4b chain via B"};
    printf("%-22s acc %d, waves/CU %2d, grid %4d: %6.1f clock64 ticks per MFMA per wave; kernel %.3f ms -> %6.2f TFLOP/s (%.1f flop per tick per wave)\n",
           names[MODE], NACC, waves, grid, per_wave, ms, flop * iters * NACC * waves * (double)grid / (ms * 1e-3) / 1e12, flop / per_wave);
    (void)hipFree(d);
    (void)hipFree(s);
}

int main() {
    for (int grid : {1, 256}) {
        run<0, 1>(4, grid);
        run<0, 4>(4, grid);
        run<0, 2>(8, grid);
        run<2, 1>(4, grid);
        run<2, 4>(4, grid);
        run<2, 8>(4, grid);
        run<2, 4>(8, grid);
        run<2, 4>(12, grid);
        run<1, 1>(1, grid);
        run<1, 3>(1, grid);
        run<3, 1>(1, grid);
        run<3, 3>(1, grid);
        run<3, 3>(4, grid);
    }
    return 0;
}
