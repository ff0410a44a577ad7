// v_mfma_f64_16x16x4_f64 issue cost on gfx950: cycles per MFMA for one wave per SIMD / two waves per SIMD, with one
// accumulator chain or several independent ones, operands in registers.  hipcc --offload-arch=gfx950 -O3 mfma64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(long long* out, double* sink, int iters) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
void run(int waves, int grid) {
    long long* d;
    double* s;
    (void)hipMalloc(&d, 8 * 4096);
    (void)hipMalloc(&s, 8 * 1024 * 1024);
    const int iters = 2000;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(64 * waves), 0, 0, d, s, iters);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(64 * waves), 0, 0, d, s, iters);
    (void)hipDeviceSynchronize();
    long long h[64];
    (void)hipMemcpy(h, d, 8 * waves, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(64 * waves), 0, 0, d, s, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double per_wave = (double)h[0] / (iters * NACC);
    double per_simd = per_wave / ((waves + 3) / 4);
    printf("acc chains %d, waves/CU %2d, grid %4d: %.1f cycles per MFMA per wave, %.1f per SIMD; kernel %.3f ms -> %.1f TFLOP/s\n", NACC,
           waves, grid, per_wave, per_simd, ms, 2.0 * 1024 * iters * NACC * waves * (double)grid / (ms * 1e-3) / 1e12);
    (void)hipFree(d);
    (void)hipFree(s);
}

int main() {
    for (int grid : {1, 256}) {
        run<1>(1, grid);
        run<1>(4, grid);
        run<1>(8, grid);
        run<1>(12, grid);
        run<2>(4, grid);
        run<4>(4, grid);
        run<2>(8, grid);
    }
    return 0;
}
