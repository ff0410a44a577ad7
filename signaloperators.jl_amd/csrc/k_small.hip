// Hand-written HIP kernels for gfx950 (MI355X, CDNA4; wave64).  No CUDA shims, no dual paths.
// The small kernels next to the filters -- the reset of their "no non-finite chunk yet" words and the NaN fill behind a
// non-finite chunk -- in a translation unit of their own: the runtime loads a code object the first time one of ITS
// kernels is launched (~1.1 ms per MB), and the fused resampler + IIR plan (k_rsos) launches nothing of k_sos.hip but these.
#include "kcommon.h"

namespace so {

// "no non-finite chunk yet" into a filter's per-channel words, as a kernel of our own: hipMemsetAsync becomes a memset NODE
// when the launch sequence is captured into a HIP graph, and a replayed graph of a single-stream plan whose stages write
// windows of the result left those words in a state that made k_sos_poison fill whole windows with NaN (ROCm 7.0; direct
// launches and multi-stream graphs were fine: tests/test_gpu_window_alias.py under SIGOPS_SINGLE_STREAM=1).  A kernel node
// carries its arguments by value.
__global__ void k_fill_u32(uint32_t* __restrict__ p, int n, uint32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
int launch_fill_u32(void* p, size_t n, uint32_t v, hipStream_t st) {  // returns the launch's hipError_t
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint32_t*)p, (int)n, v);
    // (the call this replaces was checked: a launch that did not happen leaves the words stale -- whole windows NaN, or a
    //  one-pass filter waiting for a ticket that never comes)
    return (int)hipGetLastError();
}

// NaN over the frames behind a channel's first non-finite chunk (SosGeom::bad): the reference's sequential recurrence
// never recovers from a NaN or Inf (reference src/filters.jl:252-255 -> DSP.jl filt!: the state carries it on), the
// chunked form does after K chunks.  A few workgroups per channel; channels without a bad chunk return at once.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sos_poison(T* __restrict__ y, SosGeom g) {
    const int ch = blockIdx.y;
    const int kb = g.bad[ch];
    if (kb >= g.nchunks - 1) return;  // (nothing behind the last chunk; the usual case: kb is the large initial value)
    int64_t f0 = (int64_t)(kb + 1) * g.chunk;
    if (f0 < g.store_lo) f0 = g.store_lo;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int64_t f = f0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; f < g.n; f += (int64_t)gridDim.x * kBlock) {
        const int64_t o = (int64_t)ch * g.out_pitch + f;
        if (sizeof(T) == 8 && g.out_dtype == SO_F32) reinterpret_cast<float*>(y)[o] = (float)nan;
        else y[o] = (T)nan;
    }
}

int launch_sos_poison(void* y, const SosGeom& g, hipStream_t st) {
    if (g.bad == nullptr || g.nchunks <= 1) return 0;
    const dim3 grid(32, (unsigned)g.nch);
    if (g.in_dtype == SO_F32) hipLaunchKernelGGL((k_sos_poison<float>), grid, dim3(kBlock), 0, st, (float*)y, g);
    else hipLaunchKernelGGL((k_sos_poison<double>), grid, dim3(kBlock), 0, st, (double*)y, g);
    return 1;
}

}  // namespace so
