// Does the 256 MiB Infinity Cache serve a slab that one kernel has just written to the next kernel?
// write(slab) ; read(slab) over the slabs of a 2 GiB buffer, for several slab sizes, against the same two
// kernels over the whole buffer.  hipcc --offload-arch=gfx950 -O3 tools/micro/slab_cache.hip -o /tmp/slab_cache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_write(double2* p, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u, i += 256)
        if (i < n) p[i] = make_double2(v, v + u);
}
__global__ __launch_bounds__(256) void k_read(const double2* p, size_t n, double* sink) {
    size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x;
    double a = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u, i += 256)
        if (i < n) { double2 q = p[i]; a += q.x + q.y; }
    if (a == 12345.678) sink[0] = a;
}
__global__ __launch_bounds__(256) void k_copy(const double2* p, double2* q, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x;
    double2 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < n) r[u] = p[i + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < n) q[i + u * 256] = r[u];
}

int main() {
    const size_t total = (size_t)2 << 30;
    double2 *a, *b, *c; double* sink;
    CK(hipMalloc(&a, total)); CK(hipMalloc(&b, total)); CK(hipMalloc(&c, total)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(a, 0, total)); CK(hipMemset(b, 0, total)); CK(hipMemset(c, 0, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes[] = {(size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20, (size_t)512 << 20, total};
    for (int mode = 0; mode < 3; ++mode)
        for (size_t s : sizes) {
            const size_t n = s / 16;
            const unsigned blocks = (unsigned)((n + 1023) / 1024);
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                for (size_t off = 0; off < total; off += s) {
                    double2* pa = a + off / 16; double2* pb = b + off / 16; double2* pc = c + off / 16;
                    if (mode == 0) {            // write a slab, read it back
                        k_write<<<blocks, 256>>>(pa, n, 1.0);
                        k_read<<<blocks, 256>>>(pa, n, sink);
                    } else if (mode == 1) {     // the pipeline's shape: a -> b (K3), read b (P1), b -> c (P3)
                        k_copy<<<blocks, 256>>>(pa, pb, n);
                        k_read<<<blocks, 256>>>(pb, n, sink);
                        k_copy<<<blocks, 256>>>(pb, pc, n);
                    } else {                    // read the same slab twice (is a just-READ slab served on-die?)
                        k_read<<<blocks, 256>>>(pa, n, sink);
                        k_read<<<blocks, 256>>>(pa, n, sink);
                    }
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double moved = mode == 0 ? 2.0 * total : mode == 1 ? 5.0 * total : 2.0 * total;
            std::printf("mode %d slab %5zu MiB: %.3f ms for the 2 GiB buffer, %.2f TB/s of kernel-level bytes\n", mode, s >> 20, best, moved / best / 1e9);
        }
    return 0;
}
