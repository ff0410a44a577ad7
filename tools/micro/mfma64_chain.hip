// The recurrence of k_rsos's chain wave in isolation: three v_mfma_f64_16x16x4_f64 per step whose result (registers
// 0..2 of the accumulator) is the B operand of the next step's three -- bare, and with the LDS traffic of the real
// loop (3 operand reads + 1 counter read requested a step ahead, 3 state writes + 1 counter write) placed in
// different spots.  Cycles per step for one wave on a CU.   hipcc --offload-arch=gfx950 -O3 mfma64_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define SO_LDS __attribute__((address_space(3)))

template <int MODE>
__global__ void k(long long* out, double* sink, int iters) {
    __shared__ double buf[64 * 64];
    __shared__ int flags[64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 64; i += 64) buf[i] = 1e-6 * i;
    for (int i = lane; i < 64; i += 64) flags[i] = 1 << 30;
    __syncthreads();
    volatile SO_LDS double* xs = (volatile SO_LDS double*)buf;
    volatile SO_LDS double* ss = (volatile SO_LDS double*)buf + 32 * 64;
    volatile SO_LDS int* fl = (volatile SO_LDS int*)flags;
    double a0 = 1e-3 * threadIdx.x, a1 = 2e-3, a2 = 3e-3;
    v4d st = v4d{1e-3, 2e-3, 3e-3, 0};
    double cur[3] = {0.5, 0.25, 0.125}, nxt[3] = {0, 0, 0};
    int f1 = 1 << 30;
    long long t0 = clock64();
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
        v4d acc = v4d{cur[0], cur[1], cur[2], 0};
        int f2 = 0;
        if (MODE == 2) {  // requests at the top, before the MFMAs
            if (f1 > it) {
                for (int v = 0; v < 3; ++v) nxt[v] = xs[slot * 192 + v * 64 + lane];
                f2 = fl[slot];
            }
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, st[0], acc, 0, 0, 0);
        if (MODE == 1 || MODE == 3) {  // everything under the first MFMA
            __builtin_amdgcn_sched_barrier(0);
            if (f1 > it) {
                for (int v = 0; v < 3; ++v) nxt[v] = xs[slot * 192 + v * 64 + lane];
                f2 = fl[slot];
            }
            for (int v = 0; v < 3; ++v) ss[slot * 192 + v * 64 + lane] = st[v];
            fl[32 + (slot & 7)] = it;
            __builtin_amdgcn_sched_barrier(0);
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, st[1], acc, 0, 0, 0);
        if (MODE == 2) {
            __builtin_amdgcn_sched_barrier(0);
            for (int v = 0; v < 3; ++v) ss[slot * 192 + v * 64 + lane] = st[v];
            fl[32 + (slot & 7)] = it;
            __builtin_amdgcn_sched_barrier(0);
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, st[2], acc, 0, 0, 0);
        if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
        st = acc;
        if (MODE != 0) {
            for (int v = 0; v < 3; ++v) cur[v] = nxt[v];
            f1 = __builtin_amdgcn_readfirstlane(f2);
        }
        slot = slot + 1 == 21 ? 0 : slot + 1;
    }
    long long t1 = clock64();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = st[0] + st[1] + st[2] + cur[0];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x] = t1 - t0;
}

// The form the chain wave uses: two steps per loop iteration with the roles of the register sets swapped (no copies of
// the state or of the operands), every LDS request under a step's FIRST MFMA, and the counter read requested two steps
// before it is looked at (so that no wait ever stands between two MFMAs).
__global__ void k2(long long* out, double* sink, int iters) {
    __shared__ double buf[64 * 64];
    __shared__ int flags[64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 64; i += 64) buf[i] = 1e-6 * i;
    for (int i = lane; i < 64; i += 64) flags[i] = 1 << 30;
    __syncthreads();
    volatile SO_LDS double* xs = (volatile SO_LDS double*)buf;
    volatile SO_LDS double* ss = (volatile SO_LDS double*)buf + 32 * 64;
    volatile SO_LDS int* fl = (volatile SO_LDS int*)flags;
    double a0 = 1e-3 * threadIdx.x, a1 = 2e-3, a2 = 3e-3;
    v4d sA = v4d{1e-3, 2e-3, 3e-3, 0}, sB = v4d{0, 0, 0, 0};
    double dA[3] = {0.5, 0.25, 0.125}, dB[3] = {0.1, 0.2, 0.3};
    int fA = 1 << 30, fB = 1 << 30;  // counter values requested in earlier steps (pending LDS loads)
    long long t0 = clock64();
    int slot = 0;
    auto step = [&](int it, v4d& sin, v4d& sout, double (&din)[3], int& fold, int& fnew) __attribute__((always_inline)) {
        v4d acc = v4d{din[0], din[1], din[2], 0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, sin[0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int f = __builtin_amdgcn_readfirstlane(fold);  // requested two steps ago
        if (f > it) {
            for (int v = 0; v < 3; ++v) din[v] = xs[slot * 192 + v * 64 + lane];  // operands of step it + 2 into the set just read
            fnew = fl[slot];
        }
        for (int v = 0; v < 3; ++v) ss[slot * 192 + v * 64 + lane] = sin[v];
        fl[32 + (slot & 7)] = it;
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, sin[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, sin[2], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        sout = acc;
        slot = slot + 1 == 21 ? 0 : slot + 1;
    };
    for (int it = 0; it < iters; it += 2) {
        step(it, sA, sB, dA, fA, fA);
        step(it + 1, sB, sA, dB, fB, fB);
    }
    long long t1 = clock64();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = sA[0] + sA[1] + sA[2] + dA[0] + sB[0] + dB[1];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x] = t1 - t0;
}

// ... and with every VECTOR instruction (addresses, the counter's readfirstlane, the counter value to store) moved out
// from under the MFMAs into the gap before a step's first one: an fp64 MFMA keeps the vector ALU for its 64 cycles, so
// a v_mov "under" it waits for its end and pushes the next MFMA of the chain back by that much.
__global__ void k3(long long* out, double* sink, int iters) {
    __shared__ double buf[64 * 64];
    __shared__ int flags[64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 64; i += 64) buf[i] = 1e-6 * i;
    for (int i = lane; i < 64; i += 64) flags[i] = 1 << 30;
    __syncthreads();
    const uint32_t xs0 = (uint32_t)(uintptr_t)(SO_LDS double*)buf + lane * 8, ss0 = xs0 + 32 * 64 * 8;
    const uint32_t fl0 = (uint32_t)(uintptr_t)(SO_LDS int*)flags;
    double a0 = 1e-3 * threadIdx.x, a1 = 2e-3, a2 = 3e-3;
    v4d sA = v4d{1e-3, 2e-3, 3e-3, 0}, sB = v4d{0, 0, 0, 0};
    double dA[3] = {0.5, 0.25, 0.125}, dB[3] = {0.1, 0.2, 0.3};
    int fA = 1 << 30, fB = 1 << 30;
    long long t0 = clock64();
    int slot = 0;
    auto step = [&](int it, v4d& sin, v4d& sout, double (&din)[3], int& fold) __attribute__((always_inline)) {
        // ---- the gap: vector instructions ----
        const int f = __builtin_amdgcn_readfirstlane(fold);
        uint32_t ax = xs0 + slot * 1536, as = ss0 + slot * 1536;
        uint32_t af = fl0 + slot * 4, ag = fl0 + 128 + (slot & 7) * 4;
        int itv = it;
        asm volatile("" : "+v"(ax), "+v"(as), "+v"(af), "+v"(ag), "+v"(itv));  // (in vector registers NOW)
        v4d acc = v4d{din[0], din[1], din[2], 0};
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, sin[0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- under the first MFMA: scalar and LDS instructions only ----
        if (f > it) {
            for (int v = 0; v < 3; ++v) din[v] = *(volatile SO_LDS double*)(uintptr_t)(ax + v * 512);
            fold = *(volatile SO_LDS int*)(uintptr_t)af;
        }
        for (int v = 0; v < 3; ++v) *(volatile SO_LDS double*)(uintptr_t)(as + v * 512) = sin[v];
        *(volatile SO_LDS int*)(uintptr_t)ag = itv;
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, sin[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, sin[2], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        sout = acc;
        slot = slot + 1 == 21 ? 0 : slot + 1;
    };
    for (int it = 0; it < iters; it += 2) {
        step(it, sA, sB, dA, fA);
        step(it + 1, sB, sA, dB, fB);
    }
    long long t1 = clock64();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = sA[0] + sA[1] + sA[2] + dA[0] + sB[0] + dB[1];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    long long* d;
    double* s;
    (void)hipMalloc(&d, 8 * 4096);
    (void)hipMalloc(&s, 8 * 1024 * 1024);
    const int iters = 4000;
    const char* names[] = {"bare recurrence", "LDS traffic under the first MFMA", "reads before, writes under the second MFMA",
                           "under the first MFMA, fences around the third"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, s, iters); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, s, iters); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, s, iters); break;
            default: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, s, iters); break;
            }
        }
        (void)hipDeviceSynchronize();
        long long h;
        (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %.1f cycles per step  (%s)\n", mode, (double)h / iters, names[mode]);
    }
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d, s, iters);
    (void)hipDeviceSynchronize();
    long long h;
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("k2    : %.1f cycles per step  (two steps per iteration, register sets swapped, counter two steps ahead)\n", (double)h / iters);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k3, dim3(1), dim3(64), 0, 0, d, s, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("k3    : %.1f cycles per step  (... and no vector instruction under an MFMA)\n", (double)h / iters);
    return 0;
}
