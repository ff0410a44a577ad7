// Hand-written HIP kernel for gfx950 (MI355X, CDNA4; wave64).  No CUDA shims, no dual paths.
// K3a k_resample_arb: the resampler for rates WITHOUT a period (irrational ratios such as the reference benchmark's
// "resampling-irrational" x pi, non-integer frame rates; reference src/reformatting.jl:92-98 -> DSP.jl FIRArbitrary,
// src/filters.jl:248-255), persistent and pipelined.
//
// k_resample_tiled2 (kernels2.hip) stages a tile, evaluates it and stores it, one phase after the other, and the two
// workgroups of a CU stay in step: 0.14 + 0.26 + 0.10 ms on a quarter of config 3 at x pi/3, 20 % of the HBM roofline.
// Here a workgroup walks ONE contiguous range of outputs for its CT channels:
//   * a loader wave streams the range's input, 128 frames x CT channel rows per chunk, into an LDS ring by LDS-DMA
//     (global_load_lds_dwordx4: no registers, no vector instructions; planar rows, so the 16-byte reads of lanes that
//     own consecutive output pairs fall on consecutive bank groups), a few chunks ahead of the compute waves;
//   * NC compute waves take batches of 128 (256) outputs, each the next one nobody has taken.  A lane owns outputs m and m + 1 for all CT channels
//     and walks the union of their input windows two frames (one 16-byte read per channel) at a time.  The taps of a
//     frame are formed ONCE per output -- c = h[p + 32 k] + alpha dh[p + 32 k], DSP.jl's interpolated polyphase tap --
//     and applied to the CT channels: 2 + 2 CT multiply-adds per frame and output instead of the 4 CT of the two
//     separate dot products (yLower + alpha yUpper, the reference's association; the results differ by one rounding
//     of the taps, ~1e-16 relative, and k_resample_periodic bakes its taps the same way).  Both tables sit in LDS phase by phase, (h, dh) side by side, taps ascending, with Z zero taps on
//     either side (a tap index outside an output's window needs no clamp, only its address) and an ODD pitch in
//     16-byte units: lanes at different phases read different bank groups, lanes at the same phase the same address;
//   * the waves meet through single-writer counters in LDS (frames staged; first frame each compute wave still
//     needs), never through a barrier, and the result leaves in 16-byte stores, 1 KB per channel and wave.
// Inputs outside the signal are zeros (Pad(x.signal, zero), reference src/filters.jl:240): chunks that touch an end
// of the signal are staged by guarded loads instead of the DMA.  A NaN or Inf next to a window meets a zero tap, as in
// k_resample_tiled2: it reaches the one or two outputs beside those the reference puts it in.
#include "kcommon.h"
#include "krespos.h"
#include "kstage.h"

namespace so {

#define SO_LDS __attribute__((address_space(3)))
#define SO_GLB __attribute__((address_space(1)))

namespace {

constexpr int kArbMaxDepth = 8;  // chunks in flight, at most
constexpr int kArbMaxNC = 11;  // + the loader: twelve waves, 168 registers each

struct ArbShared {
    int ldp;              // frames of the range staged so far (relative to its first staged frame)
    int rd[kArbMaxNC];    // first frame each compute wave still needs
    int next;             // next batch of the range nobody has taken yet
    int pad_[3];          // (64 bytes: the dynamic block behind it stays 16-byte aligned)
};

__device__ __forceinline__ int a_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t a_uni64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int a_flag_ld(const SO_LDS int* p) { return *(const volatile SO_LDS int*)p; }
__device__ __forceinline__ void a_flag_st(SO_LDS int* p, int v) { *(volatile SO_LDS int*)p = v; }
// (a wait for another wave of the workgroup that does not end is a bug of the protocol: said in the plan's host-mapped error
//  word, the wave ends, the host reports it -- k_rsos.hip SO_SPIN_PAUSE; without the word a trap)
__device__ __forceinline__ void a_pause(int& spins, uint32_t* err) {
    if (++spins > (1 << 22)) {
        if (err == nullptr) __builtin_trap();
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_endpgm();
    }
    __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ int a_wave_max(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return a_uni(v);
}
__device__ __forceinline__ int a_wave_min(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return a_uni(v);
}

// s_waitcnt vmcnt(n), wave-uniform runtime n up to 60 (rounded down: a stricter wait)
__device__ __forceinline__ void wait_vmcnt_le60a(int n) {
#define SO_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 0 ? 0 : (n > 56 ? 56 : n) & ~(n >= 16 ? 7 : 0)) {
        SO_W(0) SO_W(1) SO_W(2) SO_W(3) SO_W(4) SO_W(5) SO_W(6) SO_W(7) SO_W(8) SO_W(9) SO_W(10) SO_W(11) SO_W(12) SO_W(13) SO_W(14) SO_W(15)
        SO_W(16) SO_W(24) SO_W(32) SO_W(40) SO_W(48) SO_W(56)
    }
#undef SO_W
}

}  // namespace

// ring: CT rows of RINGF frames (a power of two), row c at ring + c RINGF; taps: 32 phases x ((taps + 2 Z) | 1) x (h, dh)
template <int CT, int NO, typename T>
__global__ __launch_bounds__(NO == 4 ? 512 : 768) void k_resample_arb(const T* __restrict__ x, T* __restrict__ y,
                                                       const double* __restrict__ pfbt, const double* __restrict__ dpfbt,
                                                       RsArb a) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];  // (16-byte reads of the ring: an 8-byte aligned base splits every one)
    __shared__ __attribute__((aligned(16))) ArbShared sh;
    const RsGeom& g = a.g;
    const int taps = g.taps, Z = a.zrows;
    const int trows = (taps + 2 * Z) | 1;  // granules (h, dh) per phase: odd
    const int RINGF = a.ringf;
    const int NC = a.nc;
    SO_LDS double* const tab = (SO_LDS double*)lds_raw;  // tab[(p * trows + k + Z) * 2 + {0: h, 1: dh}]
    SO_LDS T* const ring = (SO_LDS T*)(tab + (size_t)trows * 64);  // (Float32 signals: a ring of floats, widened where they are read)
    constexpr int ESZ = (int)sizeof(T);
    constexpr int CHF = 1024 / ESZ;  // frames per loader chunk: one LDS-DMA instruction (1 KB) per channel row
    const int tid = threadIdx.x, lane = tid & 63, wave = a_uni(tid >> 6);
    const bool arb = g.arbitrary != 0;
    // ---- set-up, all waves: tables with their zero rows, a zeroed ring (stale LDS may hold NaN patterns, and a zero
    //      tap does not silence those), counters ----
    for (int i = tid; i < trows * 32; i += blockDim.x) {
        const int p = i / trows, k = i - p * trows - Z;
        const bool real = k >= 0 && k < taps;
        tab[2 * i] = real ? pfbt[k * 32 + p] : 0.0;
        tab[2 * i + 1] = (real && arb) ? dpfbt[k * 32 + p] : 0.0;
    }
    for (int i = tid; i < CT * RINGF; i += blockDim.x) ring[i] = (T)0;
    if (tid == 0) sh.ldp = sh.next = 0;
    if (tid < kArbMaxNC) sh.rd[tid] = 0;
    // ---- this workgroup's range ----
    const int64_t ncg = g.nch / CT;
    const int64_t cg = (int64_t)blockIdx.x % ncg, rg = (int64_t)blockIdx.x / ncg;
    const int c0 = (int)cg * CT;
    const int64_t b0 = rg * a.bpr, b1 = min(a.nbatches, b0 + a.bpr);  // batches of 64 NO outputs
    const int64_t mlo = b0 * (64 * NO), mhi = min(g.n_out, b1 * (64 * NO));
    // first and last frame the range reads: positions do not decrease with m
    int64_t jf, jl;
    {
        int p;
        double al;
        rs_pos(g, g.m0 + mlo, jf, p, al);
        rs_pos(g, g.m0 + (mhi > mlo ? mhi - 1 : mlo), jl, p, al);
    }
    const int64_t F0 = ((jf - (taps - 1)) & ~(int64_t)1) & ~(int64_t)15;  // first staged frame: 128-byte aligned
    const int E = (int)(jl + 4 - F0);                                      // frames to stage (a pair beyond the last window)
    const int NK = (E + CHF - 1) / CHF;
    __syncthreads();
    if (mhi <= mlo) return;
    SO_LDS int* const f_ldp = (SO_LDS int*)&sh.ldp;
    SO_LDS int* const f_rd = (SO_LDS int*)sh.rd;

    if (wave == NC) {
        // =========================== loader ===========================
        const uint32_t ring_b = (uint32_t)(uintptr_t)ring;
        const uint32_t row_bytes = (uint32_t)RINGF * (uint32_t)ESZ;
        const char* const xb = (const char*)(x + (int64_t)c0 * g.in_pitch);
        const uint32_t lane16 = (uint32_t)lane * 16u;
        int minrd = 0, spins = 0;
        __builtin_amdgcn_s_setprio(3);  // (a handful of scalar instructions and DMAs per chunk: never behind the compute waves' arithmetic)
        const int depth = a.depth;
        const int debug = a.debug;
        int cnt[kArbMaxDepth] = {0};  // DMA instructions of the chunks in flight, youngest first
        if (debug & 4) {
            a_flag_st(f_ldp, E);
            return;
        }
        for (int k = 0; k < NK + depth - 1; ++k) {
            if (k < NK) {
                // ring space: chunk k overwrites frames [128 (k + 1) - RINGF - 128, ...)
                const int needrd = (k + 1) * CHF - RINGF;
                while (needrd > 0 && minrd < needrd && !(debug & 1)) {
                    int v = lane < NC ? a_flag_ld(f_rd + lane) : 0x7fffffff;
                    minrd = a_wave_min(v);
                    if (minrd < needrd) a_pause(spins, a.err);
                }
                spins = 0;
                const int64_t fa = F0 + (int64_t)k * CHF;  // absolute first frame of the chunk
                const uint32_t rho = (uint32_t)((k * CHF) & (RINGF - 1));
                int n = 0;
                if (fa >= 0 && fa + CHF <= g.n_in && a.dma_ok) {
                    dma_rows<CT>(~0ull, lane16, xb + fa * ESZ, g.in_pitch * ESZ, ring_b + rho * (uint32_t)ESZ, row_bytes);
                    n = CT;
                } else {  // an end of the signal: guarded loads, 16 bytes' worth of frames per lane
                    constexpr int FPL = 16 / ESZ;
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const T* row = x + (int64_t)(c0 + c) * g.in_pitch;
#pragma unroll
                        for (int e = 0; e < FPL; ++e) {
                            const int64_t f = fa + FPL * lane + e;
                            ring[(size_t)c * RINGF + rho + FPL * lane + e] = (f >= 0 && f < g.n_in) ? row[f] : (T)0;
                        }
                    }
                }
#pragma unroll
                for (int i = kArbMaxDepth - 1; i > 0; --i) cnt[i] = cnt[i - 1];
                cnt[0] = n;
            } else {
#pragma unroll
                for (int i = kArbMaxDepth - 1; i > 0; --i) cnt[i] = cnt[i - 1];
                cnt[0] = 0;
            }
            // chunk k - (depth - 1) has landed once at most the younger chunks' instructions are outstanding
            const int done = k - (depth - 1);
            if (done >= 0) {
                int younger = 0;
#pragma unroll
                for (int i = 0; i < kArbMaxDepth - 1; ++i)
                    if (i < depth - 1) younger += cnt[i];
                wait_vmcnt_le60a(younger);
                a_flag_st(f_ldp, min(E, (done + 1) * CHF));
            }
        }
        return;
    }
    if (wave > NC) return;

    // =========================== compute waves ===========================
    const bool vec_out = a.vec_out != 0;
    const uint32_t tab_b = (uint32_t)(uintptr_t)tab;
    const uint32_t ring_mask = (uint32_t)RINGF * (uint32_t)ESZ - 1u;
    constexpr int BO = 64 * NO;  // outputs per batch
    int spins = 0;
    // Batches are handed out in order to whichever wave is free (an LDS counter): the waves of a SIMD that also runs
    // the loader, or three compute waves where the others run two, take fewer -- with a fixed round-robin the slowest
    // SIMD set the pace (five compute waves were slower than four).
    for (;;) {
        int bi = 0;
        if (lane == 0) bi = atomicAdd((int*)&sh.next, 1);
        const int64_t b = b0 + a_uni(bi);
        if (b >= b1) break;
        const int64_t m_ = b * BO + NO * lane;
        int nv = (int)min<int64_t>(NO, mhi - m_);  // outputs of this lane inside the range (<= 0: none)
        const int64_t ml = nv > 0 ? m_ : mhi - 1;
        int r[NO], pp[NO];
        double al[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            int64_t j;
            rs_pos(g, g.m0 + (o < nv ? ml + o : (nv > 0 ? ml + nv - 1 : ml)), j, pp[o], al[o]);
            r[o] = (int)(j - F0);
        }
        const int sA = (r[0] - (taps - 1)) & ~1;  // first frame of the lane's walk (even; F0 is), relative
        const int npairs = a_wave_max(((r[NO - 1] - sA) >> 1) + 1);
        const int smin = a_wave_min(sA);
        const int emax = a_wave_max(sA + 2 * npairs);
        if (lane == 0) a_flag_st(f_rd + wave, smin);
        while (a_uni(a_flag_ld(f_ldp)) < min(emax, E) && !(a.debug & 1)) a_pause(spins, a.err);
        spins = 0;
        asm volatile("" ::: "memory");
        // frame f of the walk is tap k = r - f of an output whose newest input is r: granule k + Z of its phase's row
        uint32_t q[NO];  // the tap of the pair's SECOND frame; the first frame's is the next granule
#pragma unroll
        for (int o = 0; o < NO; ++o) q[o] = tab_b + (uint32_t)((pp[o] * trows + r[o] - sA - 1 + Z) * 16);
        uint32_t xo = ((uint32_t)sA * (uint32_t)ESZ) & ring_mask;
        double acc[NO][CT];
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[o][c] = 0.0;
        for (int it = 0; it < ((a.debug & 2) ? 1 : npairs); ++it) {
            double c0v[NO], c1v[NO];
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                // (h, dh): the interpolated tap h + alpha dh (dh == 0 for a rate that needs no interpolation)
                const v2d t1 = *(const SO_LDS v2d*)(uintptr_t)q[o], t0 = *(const SO_LDS v2d*)(uintptr_t)(q[o] + 16u);
                c0v[o] = fma(al[o], t0[1], t0[0]);
                c1v[o] = fma(al[o], t1[1], t1[0]);
                q[o] -= 32u;
            }
            v2d xv[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if constexpr (ESZ == 8) xv[c] = *(const SO_LDS v2d*)((const SO_LDS char*)ring + (size_t)c * RINGF * 8 + xo);
                else {
                    typedef float v2f __attribute__((ext_vector_type(2)));
                    const v2f w2 = *(const SO_LDS v2f*)((const SO_LDS char*)ring + (size_t)c * RINGF * 4 + xo);
                    xv[c] = v2d{(double)w2[0], (double)w2[1]};
                }
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
#pragma unroll
                for (int o = 0; o < NO; ++o) acc[o][c] = fma(c0v[o], xv[c][0], acc[o][c]);
#pragma unroll
                for (int o = 0; o < NO; ++o) acc[o][c] = fma(c1v[o], xv[c][1], acc[o][c]);
            }
            xo = (xo + 2u * (uint32_t)ESZ) & ring_mask;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            T SO_GLB* op = (T SO_GLB*)(y + (int64_t)(c0 + c) * g.out_pitch + m_);
            if (nv >= NO && vec_out) {
                typedef T v2t __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int o = 0; o < NO; o += 2) *(v2t SO_GLB*)(op + o) = v2t{(T)acc[o][c], (T)acc[o + 1][c]};
            } else {
#pragma unroll
                for (int o = 0; o < NO; ++o)
                    if (o < nv) op[o] = (T)acc[o][c];
            }
        }
    }
    if (lane == 0) a_flag_st(f_rd + wave, 0x7fffffff);
}

// LDS the kernel needs besides its small static block (esz: bytes per ring element)
size_t resample_arb_lds_bytes(int taps, int zrows, int ct, int ringf, int esz) {
    return (size_t)2 * ((taps + 2 * zrows) | 1) * 32 * 8 + (size_t)ct * ringf * esz;
}

template <typename T>
static int launch_resample_arb_t(const void* x, void* y, const double* pfbt, const double* dpfbt, RsArb a, hipStream_t st) {
    const RsGeom& g = a.g;
    constexpr int esz = (int)sizeof(T);
    a.dma_ok = 1;  // (any element-aligned address: executor.cpp carrier_vec_ok)
    a.vec_out = ((uintptr_t)y & (2 * esz - 1)) == 0 && (g.out_pitch & 1) == 0;
    a.depth = a.depth < 2 ? 2 : (a.depth > kArbMaxDepth ? kArbMaxDepth : a.depth);
    const size_t lds = resample_arb_lds_bytes(g.taps, a.zrows, a.ct, a.ringf, esz);
    if (lds + 1024 > 160 * 1024) return -1;
    const unsigned grid = (unsigned)((int64_t)a.nranges * (g.nch / a.ct));
    const unsigned threads = (unsigned)(a.nc + 1) * 64u;
#define SO_ARB(CTV, NOV)                                                                                                                      \
    {                                                                                                                                          \
        static bool seen[64];                                                                                                                  \
        if (first_use_on_device(seen))                                                                                                         \
            (void)hipFuncSetAttribute((const void*)k_resample_arb<CTV, NOV, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); \
        hipLaunchKernelGGL((k_resample_arb<CTV, NOV, T>), dim3(grid), dim3(threads), lds, st, (const T*)x, (T*)y, pfbt, dpfbt, a);             \
    }
    if (a.no == 4) {
        if (a.ct != 8 || a.nc > 7) return -1;
        SO_ARB(8, 4)
        return 0;
    }
    switch (a.ct) {
    case 8: SO_ARB(8, 2) break;
    case 4: SO_ARB(4, 2) break;
    case 2: SO_ARB(2, 2) break;
    case 1: SO_ARB(1, 2) break;
    default: return -1;
    }
#undef SO_ARB
    return 0;
}

// returns 0 when launched, -1 if this geometry is not covered (the caller falls back to k_resample_tiled2)
int launch_resample_arb(const void* x, void* y, const double* pfbt, const double* dpfbt, const RsArb& a, hipStream_t st) {
    const RsGeom& g = a.g;
    if (g.n_out <= 0) return 0;
    if ((a.no != 2 && a.no != 4) || g.in_dtype != g.out_dtype || g.nphi != 32 || a.nc < 1 || a.nc > kArbMaxNC || (a.ringf & (a.ringf - 1)) ||
        a.ringf < 4 * 256)
        return -1;
    if (g.in_dtype == SO_F64) return launch_resample_arb_t<double>(x, y, pfbt, dpfbt, a, st);
    if (g.in_dtype == SO_F32) return launch_resample_arb_t<float>(x, y, pfbt, dpfbt, a, st);
    return -1;
}

}  // namespace so
