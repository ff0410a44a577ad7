// HBM copy / read / write rates of a few access shapes on gfx950 (measurement aid for DESIGN.md's
// "copy ceiling"): hipcc --offload-arch=gfx950 -O3 tools/micro/membw.hip -o /tmp/membw && /tmp/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

// A: grid-stride, 16 B per lane, one access per iteration
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy_gs(const v2d* __restrict__ x, v2d* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        v2d v = NTL ? __builtin_nontemporal_load(x + i) : x[i];
        if (NTS) __builtin_nontemporal_store(v, y + i); else y[i] = v;
    }
}
// B: each workgroup owns a contiguous span, U loads in flight per lane
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy_span(const v2d* __restrict__ x, v2d* __restrict__ y, size_t n) {
    const size_t per = (size_t)256 * U;
    for (size_t b = (size_t)blockIdx.x * per; b < n; b += (size_t)gridDim.x * per) {
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = b + (size_t)u * 256 + threadIdx.x;
            if (i < n) v[u] = NTL ? __builtin_nontemporal_load(x + i) : x[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = b + (size_t)u * 256 + threadIdx.x;
            if (i < n) { if (NTS) __builtin_nontemporal_store(v[u], y + i); else y[i] = v[u]; }
        }
    }
}
// C: 8 B per lane (what fp64 per-sample kernels issue)
__global__ __launch_bounds__(256) void k_copy8(const double* __restrict__ x, double* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i];
}
// D: read only (sum), write only
template <int U>
__global__ __launch_bounds__(256) void k_read(const v2d* __restrict__ x, double* __restrict__ out, size_t n) {
    double acc = 0;
    const size_t per = (size_t)256 * U;
    for (size_t b = (size_t)blockIdx.x * per; b < n; b += (size_t)gridDim.x * per) {
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = b + (size_t)u * 256 + threadIdx.x; v[u] = i < n ? x[i] : v2d{0, 0}; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 1.2345e-300) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(v2d* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = v2d{1.0, 2.0};
}
// E: 64 rows x 128 B per wave-tile (K2's shape: rows 7 KB apart), 8 B per lane
__global__ __launch_bounds__(256) void k_rows(const double* __restrict__ x, double* __restrict__ y, size_t n, int chunk, int seg) {
    // sequence s = global lane's row; each row walks its chunk in segments of `seg` frames
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int rpi = 64 / seg;            // rows per instruction
    const int col = lane % seg, rsub = lane / seg;
    const size_t nseq = n / chunk;
    for (int t0 = 0; t0 < chunk; t0 += seg) {
        for (int j = 0; j < 64 / rpi; ++j) {
            size_t row = wave * 64 + (size_t)j * rpi + rsub;
            if (row < nseq) { size_t i = row * chunk + t0 + col; y[i] = x[i]; }
        }
    }
}

int main() {
    const size_t bytes = (size_t)1843200000;  // the IIR's signal: 28.8 M x 8 x 8 B
    const size_t n16 = bytes / 16, n8 = bytes / 8;
    void *x, *y; double* o;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&o, 64));
    CK(hipMemset(x, 1, bytes)); CK(hipMemset(y, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double traffic, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        const int R = 20;
        for (int i = 0; i < R; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, traffic / ms / 1e6);
    };
    const double rw = 2.0 * bytes;
    timeit("hipMemcpyDtoD", rw, [&] { CK(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0)); });
    for (int g : {1024, 2048, 4096, 8192, 65536}) {
        char nm[96];
        snprintf(nm, 96, "grid-stride 16B grid=%d", g); timeit(nm, rw, [&] { hipLaunchKernelGGL((k_copy_gs<false, false>), dim3(g), dim3(256), 0, 0, (const v2d*)x, (v2d*)y, n16); });
    }
    timeit("grid-stride 16B nt-store grid=4096", rw, [&] { hipLaunchKernelGGL((k_copy_gs<false, true>), dim3(4096), dim3(256), 0, 0, (const v2d*)x, (v2d*)y, n16); });
    timeit("grid-stride 16B nt-load+store grid=4096", rw, [&] { hipLaunchKernelGGL((k_copy_gs<true, true>), dim3(4096), dim3(256), 0, 0, (const v2d*)x, (v2d*)y, n16); });
    timeit("one block per 4 KB, 16B (grid = n/256)", rw, [&] { hipLaunchKernelGGL((k_copy_gs<false, false>), dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, (const v2d*)x, (v2d*)y, n16); });
    timeit("span U=4 grid=2048", rw, [&] { hipLaunchKernelGGL((k_copy_span<4, false, false>), dim3(2048), dim3(256), 0, 0, (const v2d*)x, (v2d*)y, n16); });
    timeit("span U=8 grid=2048", rw, [&] { hipLaunchKernelGGL((k_copy_span<8, false, false>), dim3(2048), dim3(256), 0, 0, (const v2d*)x, (v2d*)y, n16); });
    timeit("span U=8 nt-store grid=2048", rw, [&] { hipLaunchKernelGGL((k_copy_span<8, false, true>), dim3(2048), dim3(256), 0, 0, (const v2d*)x, (v2d*)y, n16); });
    timeit("span U=8 nt both grid=2048", rw, [&] { hipLaunchKernelGGL((k_copy_span<8, true, true>), dim3(2048), dim3(256), 0, 0, (const v2d*)x, (v2d*)y, n16); });
    timeit("span U=8 grid=n/(256*8)", rw, [&] { hipLaunchKernelGGL((k_copy_span<8, false, false>), dim3((unsigned)(n16 / 2048)), dim3(256), 0, 0, (const v2d*)x, (v2d*)y, n16); });
    timeit("span U=16 grid=1024", rw, [&] { hipLaunchKernelGGL((k_copy_span<16, false, false>), dim3(1024), dim3(256), 0, 0, (const v2d*)x, (v2d*)y, n16); });
    timeit("8B per lane grid=8192", rw, [&] { hipLaunchKernelGGL(k_copy8, dim3(8192), dim3(256), 0, 0, (const double*)x, (double*)y, n8); });
    timeit("read only U=8 grid=2048", (double)bytes, [&] { hipLaunchKernelGGL((k_read<8>), dim3(2048), dim3(256), 0, 0, (const v2d*)x, o, n16); });
    timeit("read only U=8 grid=n/2048", (double)bytes, [&] { hipLaunchKernelGGL((k_read<8>), dim3((unsigned)(n16 / 2048)), dim3(256), 0, 0, (const v2d*)x, o, n16); });
    timeit("write only grid=4096", (double)bytes, [&] { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, (v2d*)y, n16); });
    for (int seg : {16, 32, 64}) {
        char nm[96]; snprintf(nm, 96, "rows: 64 rows/wave, %d B segments, chunk 896", seg * 8);
        const int chunk = 896; const size_t nseq = n8 / chunk; const unsigned g = (unsigned)((nseq + 255) / 256);
        timeit(nm, rw, [&] { hipLaunchKernelGGL(k_rows, dim3(g), dim3(256), 0, 0, (const double*)x, (double*)y, n8, chunk, seg); });
    }
    return 0;
}
