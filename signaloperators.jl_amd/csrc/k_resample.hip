// Hand-written HIP kernels for gfx950 (MI355X, CDNA4; wave64).  No CUDA shims, no dual paths.
// K3 k_resample*: polyphase FIR resampler (reference src/reformatting.jl:92-98, src/filters.jl:248-255)
#include "kcommon.h"
#include "krespos.h"
#include "kstage.h"

// One translation unit per (tile type, channels per tile) of the periodic kernel (build.py: -DSO_RP_UNIT=1 .. 8 = Float64 tiles
// of 8 / 4 / 2 / 1 channels, Float32 tiles of 8 / 4 / 2 / 1), and -DSO_RP_UNIT=0 for everything else in this file (the other
// resampler kernels, the fix-up kernel, the dispatcher): the runtime loads a code object the first time one of ITS kernels is
// launched (~1.1 ms per MB; all 150 instantiations in one object were 5.4 MB in front of a first resampling sink), and the
// units compile side by side.  -1 (the default, tools/build_variant.sh): everything in this one.
#ifndef SO_RP_UNIT
#define SO_RP_UNIT -1
#endif

namespace so {

typedef double v4d __attribute__((ext_vector_type(4)));

#if SO_RP_UNIT <= 0

template <typename T>
__global__ __launch_bounds__(kBlock) void k_resample(const T* __restrict__ x,
                                                     const double* __restrict__ pfb,
                                                     const double* __restrict__ dpfb, RsGeom g,
                                                     T* __restrict__ y) {
    const int64_t mi = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (mi >= g.n_out) return;
    const int ch = blockIdx.y;
    const int64_t m = g.m0 + mi;
    int64_t j;
    int p;
    double alpha;
    rs_pos(g, m, j, p, alpha);
    const T* xp = x + (int64_t)ch * g.in_pitch;
    const double* pf = pfb + (int64_t)p * g.taps;
    const double* df = dpfb + (int64_t)p * g.taps;
    double lo = 0.0, hi = 0.0;
    for (int k = g.taps - 1; k >= 0; --k) {  // oldest input first, like DSP.jl's dot
        const int64_t i = j - k;
        const double xv = (i >= 0 && i < g.n_in) ? (double)xp[i] : 0.0;
        lo += pf[k] * xv;
        if (g.arbitrary) hi += df[k] * xv;
    }
    const double r = g.arbitrary ? lo + hi * alpha : lo;
    y[(int64_t)ch * g.out_pitch + mi] = (T)r;
}

void launch_resample(const void* x, void* y, const double* pfb, const double* dpfb,
                     const RsGeom& g, hipStream_t st) {
    if (g.n_out <= 0) return;
    dim3 grid((unsigned)((g.n_out + kBlock - 1) / kBlock), (unsigned)g.nch);
    if (g.in_dtype == SO_F32)
        hipLaunchKernelGGL((k_resample<float>), grid, dim3(kBlock), 0, st, (const float*)x, pfb,
                           dpfb, g, (float*)y);
    else
        hipLaunchKernelGGL((k_resample<double>), grid, dim3(kBlock), 0, st, (const double*)x, pfb,
                           dpfb, g, (double*)y);
}

// ---------------------------------------------------------------------------
// K3t: tiled polyphase resampler for rates WITHOUT a usable period (irrational ratios such as the
// reference benchmark's "resampling-irrational" x pi, non-integer frame rates, very long periods).
// The thread-per-output kernel above streams a lane-private row of taps and inputs from L2 per
// output (2 % of the roofline on a config-3-sized signal); here a workgroup stages ct channels x
// ~1000 input frames once (coalesced, zero padded outside the signal) together with BOTH polyphase
// tables, transposed to [tap][phase] so that the lanes' different phases fall on different LDS
// banks.  A lane owns an output for all ct channels: per tap two table reads serve ct inputs, and
// the two inner products stay separate and oldest-first (DSP.jl FIRArbitrary: yLower + alpha*yUpper).
template <typename T, int CT>
__global__ __launch_bounds__(kBlock) void k_resample_tiled(const T* __restrict__ x, T* __restrict__ y,
                                                           const double* __restrict__ pfbt,
                                                           const double* __restrict__ dpfbt, RsTiled g) {
    extern __shared__ double lds_raw[];
    const int taps = g.g.taps, nphi = g.g.nphi;
    double* const tp = lds_raw;
    double* const td = tp + (size_t)taps * nphi;
    T* const xs = reinterpret_cast<T*>(td + (size_t)taps * nphi);
    const int tid = threadIdx.x;
    for (int i = tid; i < taps * nphi; i += kBlock) {
        tp[i] = pfbt[i];
        td[i] = dpfbt[i];
    }
    const int64_t tx = (int64_t)blockIdx.x % g.ntiles, tc = (int64_t)blockIdx.x / g.ntiles;
    const int c0 = (int)tc * CT;
    const int64_t m0 = tx * g.tile_out;
    const int64_t m1 = m0 + g.tile_out < g.g.n_out ? m0 + g.tile_out : g.g.n_out;
    int64_t j0, j1;
    int p;
    double alpha;
    rs_pos(g.g, g.g.m0 + m0, j0, p, alpha);
    rs_pos(g.g, g.g.m0 + m1 - 1, j1, p, alpha);
    const int64_t xlo = j0 - (taps - 1);  // global input frame of LDS element 0
    const int nfr = (int)(j1 - xlo + 1);  // <= tile_in by the planner's choice of tile_out
    for (int c = 0; c < CT; ++c) {
        const T* row = x + (int64_t)(c0 + c) * g.g.in_pitch;
        for (int i = tid; i < nfr; i += kBlock) {
            const int64_t n = xlo + i;
            xs[c * g.pitch + i] = (n >= 0 && n < g.g.n_in) ? row[n] : (T)0;  // Pad(x.signal, zero), src/filters.jl:240
        }
    }
    __syncthreads();
    for (int64_t mi = m0 + tid; mi < m1; mi += kBlock) {
        int64_t j;
        rs_pos(g.g, g.g.m0 + mi, j, p, alpha);
        const T* __restrict__ xb = xs + (int)(j - xlo);
        double lo[CT], hi[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) lo[c] = hi[c] = 0.0;
        // oldest input first, like DSP.jl's dot; four taps per round so that the LDS reads of a
        // round are in flight together (one wave per SIMD is latency-bound on a read-use-read chain)
        int k = taps - 1;
        for (; k >= 3; k -= 4) {
            double pf[4], df[4], xv[4][CT];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pf[u] = tp[(k - u) * nphi + p];
                df[u] = td[(k - u) * nphi + p];
#pragma unroll
                for (int c = 0; c < CT; ++c) xv[u][c] = (double)xb[c * g.pitch - (k - u)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    lo[c] += pf[u] * xv[u][c];
                    hi[c] += df[u] * xv[u][c];
                }
        }
        for (; k >= 0; --k) {
            const double pf = tp[k * nphi + p];
            const double df = td[k * nphi + p];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const double xv = (double)xb[c * g.pitch - k];
                lo[c] += pf * xv;
                hi[c] += df * xv;
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const double r = g.g.arbitrary ? lo[c] + hi[c] * alpha : lo[c];
            y[(int64_t)(c0 + c) * g.g.out_pitch + mi] = (T)r;
        }
    }
}


template <typename T>
static void launch_resample_tiled_t(const void* x, void* y, const double* pfbt, const double* dpfbt, const RsTiled& g,
                                    hipStream_t st) {
    const size_t ldsb = ((size_t)2 * g.g.taps * g.g.nphi * 8 + (size_t)g.ct * g.pitch * sizeof(T) + 15) / 16 * 16;
    const unsigned grid = (unsigned)(g.ntiles * (g.g.nch / g.ct));
#define SO_RT(CTV)                                                                                                         \
    {                                                                                                                      \
        static bool seen[64];                                                                                              \
        if (first_use_on_device(seen))                                                                                     \
            (void)hipFuncSetAttribute((const void*)k_resample_tiled<T, CTV>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                      160 * 1024);                                                                         \
        hipLaunchKernelGGL((k_resample_tiled<T, CTV>), dim3(grid), dim3(kBlock), ldsb, st, (const T*)x, (T*)y, pfbt, dpfbt, g); \
    }
    switch (g.ct) {
    case 8: SO_RT(8) break;
    case 4: SO_RT(4) break;
    case 2: SO_RT(2) break;
    default: SO_RT(1) break;
    }
#undef SO_RT
}

void launch_resample_tiled(const void* x, void* y, const double* pfbt, const double* dpfbt, const RsTiled& g,
                           hipStream_t st) {
    if (g.g.n_out <= 0) return;
    if (g.g.in_dtype == SO_F32) launch_resample_tiled_t<float>(x, y, pfbt, dpfbt, g, st);
    else launch_resample_tiled_t<double>(x, y, pfbt, dpfbt, g, st);
}


// ---------------------------------------------------------------------------
// K3r: row-tiled polyphase resampler for rational rates with LONG periods / filters (e.g.
// 44.1 kHz -> 16 kHz: 441 inputs and 160 outputs per period, 148 taps per output), where one
// period of the MFMA kernel's tile no longer fits LDS twice and the taps no longer fit registers.
//
// A workgroup stages pb periods x ct channels of input (one LDS tile, coalesced loads, zero
// padded outside the signal) and produces all L outputs of those rows.  A wave covers
// `rows = ct*pb` rows x `64/rows` consecutive output phases: lanes of one phase read the SAME
// combined tap (h + alpha*dh, host-built table ctab[phase][age]) -- a handful of distinct
// addresses per load instruction instead of 64 -- and their own row's input from LDS at a row
// stride of M elements.  (The thread-per-output fallback streams a lane-private row of taps per
// output: 1.2 KB of L2 traffic per 8-byte result on the slab config.)
template <typename T>
__global__ __launch_bounds__(1024) void k_resample_rows(const T* __restrict__ x, T* __restrict__ y,
                                                        const double* __restrict__ ctab,
                                                        const int* __restrict__ jr,
                                                        const double* __restrict__ mtab,
                                                        const int* __restrict__ jend, RsRows g) {
    extern __shared__ double lds_raw[];
    T* const lds = reinterpret_cast<T*>(lds_raw);
    const int rows = g.ct * g.pb, ph = 64 / rows;
    const int64_t ntx = (g.nperiods + g.pb - 1) / g.pb;
    const int64_t tx = (int64_t)blockIdx.x % ntx, tc = (int64_t)blockIdx.x / ntx;
    const int c0 = (int)tc * g.ct;
    const int64_t P0 = tx * g.pb;
    const int64_t xbase = P0 * g.M + g.jlo;  // global input frame of LDS element 0
    // ---- stage: ct rows of tile_len frames ----
    for (int c = 0; c < g.ct && !(g.debug & 2); ++c) {
        const T* row = x + (int64_t)(c0 + c) * g.in_pitch;
        for (int i = threadIdx.x; i < g.tile_len; i += blockDim.x) {
            const int64_t n = xbase + i;
            lds[c * g.pitch + i] = (n >= 0 && n < g.n_in) ? row[n] : (T)0;
        }
    }
    __syncthreads();
    if (g.debug & 1) return;
    // ---- compute ----
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if (g.kw > 0) {
        // MFMA path: unit = (group of 16 phases, 16-row tile q).  Y[16 x 16] = X[16 x kw] * Tap[kw x 16]
        // with v_mfma_f64_16x16x4_f64; A from LDS (one sample per lane), B from the L2-resident
        // tap block (one tap per lane, 512 contiguous bytes per wave load), both fetched four
        // k-steps ahead.  Operand / result maps as in k_resample_periodic.
        const int kq = lane >> 4, n16 = lane & 15;
        const int nq = rows >> 4, nunits = g.ngroups * nq, ksteps = g.kw >> 2;
        const int pbmask = g.pb - 1;
        for (int u = wave; u < nunits; u += nwaves) {
            const int gi = u / nq, q = u % nq;
            const int rho_a = 16 * q + n16;  // A operand row of this lane
            const T* __restrict__ ap = lds + (rho_a >> g.pbshift) * g.pitch + (rho_a & pbmask) * (int)g.M - g.jlo +
                                       (jend[gi] - (g.kw - 1)) + kq;
            const double* __restrict__ bp = mtab + ((size_t)gi * g.kw + kq) * 16 + n16;
            v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
            double a0[4], b0[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sn = j < ksteps ? j : ksteps - 1;  // (never read past the tap block)
                a0[j] = (double)ap[4 * sn];
                b0[j] = bp[(size_t)(4 * sn) * 16];
            }
            for (int s4 = 0; s4 < ksteps; s4 += 4) {
                double a1[4], b1[4];
                const bool more = s4 + 4 < ksteps;
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // next four k-steps (clamped on the last round)
                    int sn = more ? s4 + 4 + j : s4 + j;
                    sn = sn < ksteps ? sn : ksteps - 1;
                    a1[j] = (double)ap[4 * sn];
                    b1[j] = bp[(size_t)(4 * sn) * 16];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (s4 + j < ksteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[j], b0[j], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a0[j] = a1[j];
                    b0[j] = b1[j];
                }
            }
            const int r = gi * 16 + n16;
            // A non-finite accumulator: the group's window -- wider than any single output's, padded with zero taps -- held a
            // non-finite sample (0 * NaN is NaN).  The reference multiplies each output's own taps only (src/filters.jl:252-255
            // -> DSP.jl's polyphase kernels): the unit's outputs once more, output by output from the combined-tap table the
            // scalar path below uses -- the tile is still in LDS.  (Rounds 4 - 5 stated the superset instead.)
            {
                const bool bad = !(isfinite(acc[0]) && isfinite(acc[1]) && isfinite(acc[2]) && isfinite(acc[3]));
                if (__builtin_amdgcn_ballot_w64(bad) != 0 && r < (int)g.L && !(g.debug & 8)) {
                    const double* __restrict__ tp = ctab + (size_t)r * g.taps;
                    const int jn = jr[r];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rho = 16 * q + kq + 4 * i;
                        const T* __restrict__ xp = lds + (rho >> g.pbshift) * g.pitch + (rho & pbmask) * (int)g.M - g.jlo + jn;
                        double e = 0.0;
                        for (int k = 0; k < g.taps; ++k) e = fma(tp[k], (double)xp[-k], e);
                        acc[i] = e;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rho = 16 * q + kq + 4 * i;  // D: row = (lane>>4) + 4*reg
                const int64_t period = P0 + (rho & pbmask);
                const int64_t m = period * g.L + r;
                if (period < g.nperiods && r < (int)g.L && m < g.n_out)
                    y[(int64_t)(c0 + (rho >> g.pbshift)) * g.out_pitch + m] = (T)acc[i];
            }
        }
        return;
    }
    const int row = lane % rows, pl = lane / rows;  // this lane's row and phase slot
    const int cl = row / g.pb, p = row % g.pb;
    const int64_t period = P0 + p;
    const T* __restrict__ xin = lds + cl * g.pitch + p * (int)g.M - g.jlo;
    T* __restrict__ yrow = y + (int64_t)(c0 + cl) * g.out_pitch + period * g.L;
    for (int rb = wave * ph; rb < (int)g.L; rb += nwaves * ph) {
        const int r = rb + pl;
        const bool live = r < (int)g.L;
        const int rr = live ? r : (int)g.L - 1;
        const double* __restrict__ tp = ctab + (size_t)rr * g.taps;
        const T* __restrict__ xp = xin + jr[rr];  // newest input of this output
        // Taps come from global memory (L1/L2 hits, but ~500 cycles each): fetch them eight at a
        // time so that eight loads are in flight per lane, then do the eight LDS reads + FMAs.
        // Accumulation order is k = 0,1,2,... in ONE chain per output, as in the oracle's loop.
        double acc = 0.0;
        int k = 0;
        for (; k + 8 <= g.taps; k += 8) {
            double tk[8], xk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tk[u] = tp[k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) xk[u] = (double)xp[-(k + u)];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma(tk[u], xk[u], acc);
        }
        for (; k < g.taps; ++k) acc = fma(tp[k], (double)xp[-k], acc);
        const int64_t m = period * g.L + r;
        if (live && period < g.nperiods && m < g.n_out) yrow[r] = (T)acc;
    }
}


int launch_resample_rows(const void* x, void* y, const double* ctab, const int* jr, const double* mtab,
                         const int* jend, const RsRows& g, int dtype, hipStream_t st) {
    if (g.n_out <= 0) return 0;
    const int64_t ntiles = ((g.nperiods + g.pb - 1) / g.pb) * (g.nch / g.ct);
    const size_t esz = dtype == SO_F32 ? 4 : 8;
    const size_t ldsb = ((size_t)g.ct * g.pitch * esz + 7) / 8 * 8;
    if (dtype == SO_F32) {
        static bool seen[64];
        if (first_use_on_device(seen))
            (void)hipFuncSetAttribute((const void*)k_resample_rows<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_resample_rows<float>), dim3((unsigned)ntiles), dim3(g.threads), ldsb, st, (const float*)x, (float*)y,
                           ctab, jr, mtab, jend, g);
    } else {
        static bool seen[64];
        if (first_use_on_device(seen))
            (void)hipFuncSetAttribute((const void*)k_resample_rows<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_resample_rows<double>), dim3((unsigned)ntiles), dim3(g.threads), ldsb, st, (const double*)x,
                           (double*)y, ctab, jr, mtab, jend, g);
    }
    return 0;
}

// ---------------------------------------------------------------------------
// K3p: periodic polyphase resampler (rational L/M), persistent and software-pipelined.
//
// Tile = pt periods x ct channels = 32 rows, staged in LDS as fp64 from a 128-byte aligned
// global start (whole cache lines per LDS-DMA instruction, ~5 % halo).  One workgroup per CU
// loops over tiles with a RING of nslots (3-4) LDS slots and wave specialisation: waves
// [0,ncompute) compute tile i while the loader waves already have tiles i+1 .. i+nslots-2
// in flight; the loaders retire a tile with a COUNTED s_waitcnt vmcnt(N) (N = the LDS-DMA
// instructions this wave issued for younger tiles), so HBM reads never drain at a tile
// boundary; one raw s_barrier per tile (cdna_hip_programming.md, 3-buffer glds span).
//
// Source: a list of *carriers* (sorted frame ranges).  A carrier is a planar array
// x[c*cstride + n + df] plus up to 4 steps  v = v (op) F_k[n]  /  F_k[n] (op) v  whose F_k
// come from the piece's per-frame program (ramps, generators, constants: reference
// src/ramps.jl:60-72, src/functions.jl:53-60, src/mapsignal.jl:249-255).  This covers
// Amplify/Mix/Ramp chains over one array without ever materialising them; anything more
// general is materialised by K1 first and arrives as a 0-step carrier.
//
// Compute: for a group of 16 consecutive outputs of the period, the 32 rows x 16 outputs
// block is the product  Y[32 x 16] = X[32 x kw] * Tap[kw x 16]  (X = the rows' input
// windows, Tap = the group's combined taps h + alpha*dh, zero outside each output's
// support).  It is evaluated with v_mfma_f64_16x16x4_f64 used purely as a register-blocking
#endif  // SO_RP_UNIT <= 0

#if SO_RP_UNIT != 0
// device: each lane supplies ONE input sample (one ds_read_b64) and ONE tap (a register,
// loaded once per kernel) per 1024 multiply-adds, so neither LDS bandwidth nor tap
// delivery limits the kernel (a scalar-operand VALU formulation measured ~700 clk per
// 64-byte tap line on the scalar cache).  The kernel stays HBM-bound, which is the
// roofline it is reported against.
//   tab  [ngroups][KS*4][16] taps, oldest input first, zero padded to KS k-steps
//   jend [ngroups]           newest input of the group's window, relative to the period base
// Operand maps (cdna_hip_programming.md §3): A[l&15][k=l>>4], B[k=l>>4][l&15],
// D: col = l&15, row = (l>>4) + 4*reg.

// cycle stamp of workgroup 0 (tuning aid, SIGOPS_RS_TRACE)
__device__ __forceinline__ void rs_stamp(const RsPeriodic& g, int wave, int it, int k) {
    if (g.trace != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && it < kRsTraceIters)
        g.trace[(wave * kRsTraceIters + it) * kRsTraceStamps + k] = clock64();
}


// the few geometry fields the general staging path needs (passed by value to its out-of-line
// copy: taking the address of the kernel-argument struct would move it to scratch)
struct RsStageGeom {
    int64_t n_in;
    int32_t lds_pitch, pad;
};

// Is the wave's chunk of 64 consecutive 16-byte vectors [gf,gl) a plain copy out of one fp64
// array carrier (so it can go by LDS-DMA)?  `cu` returns the carrier.  Pure function of the
// chunk position: the issue pass and the modify pass must agree on it.
__device__ __forceinline__ bool rs_dma_chunk(const RsStageGeom& g, const RsCtl& ctl, int64_t gf,
                                             int64_t gl, int& cu) {
    const DCarrier* car = ctl.car;
    const int ncar = ctl.ncar;
    cu = 0;
    while (cu + 1 < ncar && car[cu].b <= gf) ++cu;  // carriers are sorted
    const DCarrier& C = car[cu];
    // (a carrier whose step takes a second array: the general path loads both)
    return !(g.pad & 8) && C.base != nullptr && C.vec_ok && C.dtype == SO_F64 && gf >= C.a &&
           gl <= C.b && gf >= 0 && gl <= g.n_in && (((gf + C.df) & 1) == 0) && !(C.nsteps > 0 && (C.arg[0] & kCarArr2));
}

// Stage one input tile (CT channels x nfr frames from global frame xbase) into an LDS slot
// as fp64, zero-padded outside [0,n_in) (Pad(x.signal,zero), reference src/filters.jl:240).
//   PASS 0 (issue):  per wave-chunk of 64 vectors either fire CT LDS-DMA instructions (fp64
//                    array carrier; returns how many were issued) or, for anything else
//                    (edges, f32, generated pieces), load -> steps -> LDS store right away.
//   PASS 1 (modify): after the wave's own DMA of this tile has landed (`allowed` = DMA
//                    instructions it issued for younger tiles), apply the carrier steps in
//                    place to exactly the chunks it copied.
template <typename T, int CT, int PASS, bool A2 = false>
__device__ __forceinline__ int stage_tile(const RsStageGeom& g, int64_t xbase, int nfr, int c0,
                                          T* __restrict__ buf, const RsCtl& ctl,
                                          const RsGlobalTables& gsrc, int tid, int nthr, int allowed) {
    const DCarrier* car = ctl.car;
    const DLeaf* leaves = ctl.leaves;
    constexpr int V = 16 / sizeof(T);
    const int nvec = (nfr + V - 1) / V;  // lds_pitch leaves room for the round-up
    int ndma = 0;
    bool waited = false;
    const int ci = 0;  // (<= kCtlCar carriers: the slow path scans from the first)
    // wave-uniform loop over chunks of 64 vectors; lane l owns vector ivb + l
    const int lane = tid & 63;
    for (int ivb = __builtin_amdgcn_readfirstlane(tid - lane); ivb < nvec; ivb += nthr) {
        const int iv = ivb + lane;
        const bool act = iv < nvec;
        const int64_t gi = xbase + (int64_t)iv * V;
        if constexpr (sizeof(T) == 8) {
            // fp64 fast path: a wave's 64 consecutive vectors (128 frames; fewer in the
            // exec-masked last chunk -- inactive lanes of a global_load_lds write nothing)
            // lie inside one array carrier -> asynchronous DMA of all CT channel rows.
            const int nact = nvec - ivb < 64 ? nvec - ivb : 64;
            const int64_t gf = xbase + (int64_t)ivb * V, gl = gf + (int64_t)nact * V;
            int cu;
            if (rs_dma_chunk(g, ctl, gf, gl, cu)) {
                const DCarrier& C = car[cu];
                if constexpr (PASS == 0) {
                    const int64_t cs = C.cstride;
                    const double* src = (const double*)C.base + ((int64_t)c0 * cs + C.df) + gi;
                    if (act) {
                        const uint32_t la = __builtin_amdgcn_readfirstlane(lds_addr(buf + ivb * V));
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            dma16(src + (int64_t)c * cs,
                                  __builtin_amdgcn_readfirstlane(la + (uint32_t)(c * g.lds_pitch) * 8u));
                    }
                    ndma += CT;
                } else if (C.nsteps > 0 && !(g.pad & 32)) {
                    double F[kMaxFrameSlots][V];
#pragma unroll
                    for (int k = 0; k < kMaxFrameSlots; ++k)
#pragma unroll
                        for (int e = 0; e < V; ++e) F[k][e] = 0.0;
                    if (!(g.pad & 64)) {
                        for (int k = 0; k < C.nslots; ++k) {
                            const DLeaf& L = leaves[C.slot_leaf[k]];
                            const int kind = C.slot_kind[k];
#pragma unroll
                            for (int e = 0; e < V; ++e) {
                                const double v = slot_eval(kind, L, gi + e);
                                switch (k) {
                                case 0: F[0][e] = v; break;
                                case 1: F[1][e] = v; break;
                                case 2: F[2][e] = v; break;
                                default: F[3][e] = v; break;
                                }
                            }
                        }
                    }
                    if (!waited) {  // this tile's DMA landed in LDS
                        wait_vmcnt_le(allowed);
                        waited = true;
                    }
                    if (act) rmw_chunk<CT, true>(lds_addr(buf + iv * V), g.lds_pitch, C, F);
                }
                continue;
            }
        }
        if constexpr (PASS == 0)
            if (act) stage_generic_impl<T, CT, A2>(g.n_in, g.lds_pitch, gsrc.car, ctl.ncar, gsrc.ops, gsrc.leaves, gi, iv, ci, c0, buf);
    }
    if constexpr (PASS == 1) {
        if (!waited) wait_vmcnt_le(allowed);
    }
    return ndma;
}

// The general staging path out of line: it carries the frame interpreter and CT x V register
// blocks, and inlined next to the loader's fast path it pushes that past the 128-register
// budget -- spills there are scratch reloads with s_waitcnt vmcnt(0) in the middle of the
// LDS-DMA ring.  Only tiles at a signal/carrier edge and non-fp64 sources come here.
template <typename T, int CT, int PASS, bool A2 = false>
__device__ __attribute__((noinline)) int stage_tile_ool(int64_t n_in, int lds_pitch, int pad, int64_t xbase,
                                                        int nfr, int c0, T* __restrict__ buf,
                                                        const RsCtl* ctl, const DCarrier* gcar,
                                                        const DOp* gops, const DLeaf* gleaves, int tid,
                                                        int nthr, int allowed) {
    const RsStageGeom g{n_in, lds_pitch, pad};
    const RsGlobalTables gsrc{nullptr, gcar, gops, gleaves};
    return stage_tile<T, CT, PASS, A2>(g, xbase, nfr, c0, buf, *ctl, gsrc, tid, nthr, allowed);
}

// GA (gain at the A operand): a Float32 array times ONE Float64 per-frame gain (`Amplify(x32,
// Signal(sin))`: the product is a Float64 signal, so it cannot be formed in the Float32 tile).  The
// tile ring holds the raw Float32 samples (LDS-DMA, like a plain Float32 source), the gain ring is
// three deep and the compute waves multiply while they fetch the A operand: (double)x * F[frame].
// ST (state waves): the stage's only consumer is an SOS filter and the last two loader waves
// compute its chunk states from the staged tiles (see RsPeriodic::nstate).  A separate
// instantiation: the state waves' 24 tap registers must not raise the register budget (and with
// it, spills) of the kernels that do not use them.
// GADD (with GA): the one fused step is an ADD (`Mix(x32, Signal(sin))`) instead of a multiply: the gain
// ring then holds zeros outside the fused pieces (the zero extension of the stage's input is 0, not 0 + g).
// Q: 16-row MFMA tiles per workgroup tile (rows = 16 Q = periods x channels).  Q = 1 (round 3) halves the tile for
// rates whose period is long (44.1 -> 16 kHz: 441 inputs per period, 36 k-steps): two 32-row slots of it do not fit
// LDS, two 16-row slots do, and the ring, the edge handling and the fused sources of this kernel then serve
// what used to go to the row-tiled kernel without any overlap of loads and MFMAs (config 5).
// A2: carrier 0's one step takes a SECOND Float64 array as its operand (`Mix(x, y)` / `Amplify(x, y)` of two arrays: DCarrier::base2,
// arg bit kCarArr2).  Both arrays go by LDS-DMA: the first into the tile ring as ever, the loader wave's own chunk of the second
// (CT rows x 1 KB) into a staging area of that wave behind a ring of TWO tiles (the planner's choice for this instantiation: one
// tile of look-ahead for either array); retiring the tile, the wave applies the step in place -- two LDS reads, one operation,
// one LDS write per 16 bytes, the operations K1 would have done on the way to a materialised sum.  (The chunk in registers
// instead -- 4 CT of them, fetched an iteration ahead or at retire time -- was spilled right behind its loads in a kernel that
// has 128: each spill a wait for its load.)  A separate instantiation: nothing of this in the other kernels' loader loops.
// With F32M: two Float32 arrays of a Float32 signal -- Float32 tile and staging rows, the step in Float32 (Julia's arithmetic on
// Float32 operands, what K1's materialised map computes), the products on the Float32 MFMA.
// F32M: a Float32 signal all the way (T = TO = float, plain source): operands, taps and accumulators in Float32 on
// v_mfma_f32_16x16x4_f32 -- 32 cycles per instruction and SIMD where the Float64 one takes 64 (this kernel is bound by its
// MFMAs on Float32 data: half the bytes, the same matrix cycles).  Same A / B operand maps; the RESULT map differs: row =
// 4 (lane >> 4) + register, not (lane >> 4) + 4 register.  A k-ordered chain of Float32 fmas per output: within the
// reference's 1e-6 for Float32 results (measured: profiles/r05/relerr_maxima_f32mfma.json), not bit-equal to the Float64
// products rounded once -- SIGOPS_RS_NO_F32MFMA keeps those.
typedef float v4f __attribute__((ext_vector_type(4)));
template <typename T, int CT, int KS, int G, bool TWO = false, typename TO = T, bool GA = false, bool ST = false, bool GADD = false, int Q = 2, bool F32M = false, bool A2 = false>
__global__ __launch_bounds__(G >= 3 ? 512 : 1024) void k_resample_periodic(
    const double* __restrict__ tab, const int* __restrict__ jend, RsPeriodic g, TO* __restrict__ y,
    RsGlobalTables gsrc) {
    // LDS: ring of nslots tiles in the SAMPLE type (fp32 tiles are converted at the A-operand
    // read, so fp32 sources go by LDS-DMA too), then the fp64 gain ring
    extern __shared__ double lds_raw[];
    T* const lds = reinterpret_cast<T*>(lds_raw);
    constexpr int V = 16 / (int)sizeof(T);       // frames per 16-byte vector
    constexpr int kAlign = 128 / (int)sizeof(T);  // frames per 128-byte line
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int nc = g.ncompute;
    const int S = g.nslots;
    const int bufsz = CT * g.lds_pitch;
    const int64_t ntx = (g.nperiods + g.pt - 1) / g.pt;
    const int64_t stride = gridDim.x;
    // Control block (carriers, slot leaves) -> LDS once per workgroup: the loader waves must not
    // depend on global / kernarg loads inside the tile loop (they queue behind the HBM stream;
    // measured ~1000 cycles per dependent scalar load while the chip streams).
    // (A separate __shared__ object, not part of the dynamic ring: hipcc orders every LDS read
    //  that may alias an LDS-DMA destination behind s_waitcnt vmcnt(0), which would drain the
    //  ring at each control read.)
    __shared__ RsCtl sctl;
    {
        const int* src = reinterpret_cast<const int*>(gsrc.ctl);
        int* dst = reinterpret_cast<int*>(&sctl);
        for (int i = threadIdx.x; i < (int)(sizeof(RsCtl) / 4); i += blockDim.x) dst[i] = src[i];
    }
    if (ST && g.nstate > 0) {
        // the state waves' window can reach frames of a slot no tile has written yet: zero the ring
        // once (whatever bits LDS holds at kernel start must not meet a zero tap as NaN)
        const int nq = (int)(((size_t)S * bufsz * sizeof(T)) / 8);
        for (int i = threadIdx.x; i < nq; i += blockDim.x) lds_raw[i] = 0.0;
        // ... and their taps live in LDS, [4*ksw][10] behind the gain ring (registers would be 48 per
        // lane across the whole loader loop -- spilled, measured 3.5 ms; L2 costs a dependent
        // round trip per k-step group while the chip streams, 1.56 ms)
        double* wl = lds_raw + ((size_t)S * bufsz * sizeof(T) + 7) / 8 + (size_t)(GA ? 3 : 2) * g.fslots * g.fpitch +
                     (g.ftwo ? kRsTwoDoubles : 0);
        for (int i = threadIdx.x; i < 4 * g.ksw * 10; i += blockDim.x) wl[i] = g.wtab[(size_t)(i / 10) * 16 + (i % 10)];
    }
    __syncthreads();
    const RsCtl& ctl = sctl;
    // Tile `it` of this workgroup is tile  t = blockIdx.x + it*gridDim.x  = (tc, tx): channel group
    // tc = t / ntx, x-tile tx = t % ntx.  Both roles step (tc, tx) and the tile's first input
    // frame xb incrementally -- one 64-bit division per kernel, not per tile (the loader's
    // per-tile control code is on the critical path: a single wave retires roughly one
    // instruction per 5 cycles).  The tile is staged from the 128-byte aligned frame below xb.
    const int64_t ptM = (int64_t)g.pt * g.M;
    const int64_t dq = stride / ntx, dr = stride % ntx;
    const int64_t ngrp = g.nch / CT;
    struct TilePos {
        int64_t tc, tx, xb;
    };
    auto tile_first = [&]() {
        TilePos p;
        p.tc = (int64_t)blockIdx.x / ntx;
        p.tx = (int64_t)blockIdx.x % ntx;
        p.xb = p.tx * ptM + g.jlo;
        return p;
    };
    auto tile_next = [&](TilePos& p) {
        p.tc += dq;
        p.tx += dr;
        p.xb += dr * ptM;
        if (p.tx >= ntx) {
            p.tx -= ntx;
            p.xb -= ntx * ptM;
            ++p.tc;
        }
    };
    // fast tiles: one fp64 array carrier covers the whole staged range -> straight-line DMA
    // issue with the carrier's facts in scalar registers (read from the LDS control block
    // once), no per-chunk carrier logic
    const DCarrier& C0 = ctl.car[0];
    const int64_t a0 = rfl64(C0.a), b0 = rfl64(C0.b), cs0 = rfl64(C0.cstride), df0 = rfl64(C0.df);
    const double* base0 = (const double*)rfl64((int64_t)(uintptr_t)C0.base);
    const int nsteps0 = __builtin_amdgcn_readfirstlane(C0.nsteps);
    // (tiles inside carrier 0 -- normally all but the signal's edges -- see nothing of the
    //  other carriers, e.g. the generated tail of an infinite Amplify)
    const bool single = __builtin_amdgcn_readfirstlane((int)(C0.base != nullptr && C0.vec_ok &&
                                                             C0.dtype == (sizeof(T) == 8 ? SO_F64 : SO_F32))) &&
                        !(df0 & (V - 1)) && !(g.pad & 8);
    bool nodiv0 = true;  // (division steps take the general in-place path: code size)
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < nsteps0 && __builtin_amdgcn_readfirstlane(C0.op[k]) == OP_DIV) nodiv0 = false;
    StepTab st0;  // carrier 0's steps and slot recipes, in registers
    st0.nsteps = nsteps0;
    const int nslots0 = __builtin_amdgcn_readfirstlane(C0.nslots);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        st0.op[k] = __builtin_amdgcn_readfirstlane(C0.op[k]);
        st0.arg[k] = __builtin_amdgcn_readfirstlane(C0.arg[k]);
    }
    const int64_t lo_ok = a0 > 0 ? a0 : 0;
    const int64_t hi_ok = b0 < g.n_in ? b0 : g.n_in;
    // A2: the second array of carrier 0's one step; one2: 0 v*m, 1 v+m, 2 v-m, 3 m-v
    int one2 = -1;
    const char* base2 = nullptr;
    int64_t cs2 = 0, df2 = 0;
    int a2_lanes = 0;  // vectors per row the active loader lanes take in one round (a fast tile: one round)
    if constexpr (A2) {
        // (Float64 arrays, or -- a Float32 signal all the way, F32M -- two Float32 arrays whose step rounds to Float32: the
        //  tile's own arithmetic)
        if (nsteps0 == 1 && (st0.arg[0] & kCarArr2) && (sizeof(T) == 8 ? !(st0.arg[0] & 0x200) : (F32M && (st0.arg[0] & 0x200) != 0)) &&
            __builtin_amdgcn_readfirstlane((int)(C0.base2 != nullptr && C0.vec_ok2 && C0.dtype2 == (sizeof(T) == 8 ? SO_F64 : SO_F32)))) {
            one2 = st0.op[0] == OP_MUL ? 0 : st0.op[0] == OP_ADD ? 1 : st0.op[0] == OP_SUB ? ((st0.arg[0] & 0x100) ? 3 : 2) : -1;
            base2 = (const char*)rfl64((int64_t)(uintptr_t)C0.base2);
            cs2 = rfl64(C0.cstride2);
            df2 = rfl64(C0.df2);
            if (df2 & (V - 1)) one2 = -1;
        }
        const int nldr_ = nwaves - (ST ? g.nstate : 0);
        a2_lanes = 64 * (g.nload > 0 && g.nload < nldr_ - nc ? g.nload : nldr_ - nc);
    }
    // GADD: frames the fused pieces cover (carriers are sorted and adjacent): the gain is added there only
    int64_t ga_lo = 0, ga_hi = 0;
    if constexpr (GADD) {
        const int64_t bl = rfl64(ctl.car[max(0, min(kCtlCar - 1, __builtin_amdgcn_readfirstlane(ctl.ncar) - 1))].b);
        ga_lo = lo_ok;
        ga_hi = bl < g.n_in ? bl : g.n_in;
    }
    auto is_fast = [&](const TilePos& p, int64_t& xa, int& nfr) __attribute__((always_inline)) {
        const int sh = (int)(p.xb & (kAlign - 1));
        xa = p.xb - sh;
        nfr = g.tile_len + sh;
        // (in-place steps of the fast path are fp64-only; fused fp32 sources take the general path)
        bool ok = single && nodiv0 && (sizeof(T) == 8 || nsteps0 == 0 || (A2 && one2 >= 0)) && xa >= lo_ok &&
                  xa + ((nfr + V - 1) & ~(V - 1)) <= hi_ok;
        if constexpr (A2)  // (a step on a second array: the fast form, or the general path for the whole tile)
            if (nsteps0 > 0 && (st0.arg[0] & kCarArr2)) ok = ok && one2 >= 0 && S == 2 && (nfr + V - 1) / V <= a2_lanes;
        return ok;
    };
    // Gain ring (g.fslots > 0): the per-frame slot values of a fused source -- sin generators,
    // ramps: ~150 fp64 instructions per frame -- are evaluated by ALL sixteen waves, two tiles
    // ahead, into one of two LDS arrays F[slot][frame]; the loader's in-place step then only
    // reads them.  Left to the six loader waves alone the evaluation sits on their critical
    // path (issue -> wait -> modify -> barrier) and the kernel runs 25 % below its plain-copy
    // speed.
    const bool fused0 = (nsteps0 > 0 || GA) && !(g.pad & 32);
    const bool fring = (sizeof(T) == 8 || GA) && g.fslots > 0 && fused0 && nslots0 <= g.fslots;
    constexpr int kFDepth = GA ? 3 : 2;  // gain arrays (GA: the compute waves still read tile it's while it+2's are written)
    double* const fbase = lds_raw + ((size_t)S * bufsz * sizeof(T) + 7) / 8;
    // Who evaluates which frames.  fp64 MFMA and fp64 VALU run on the same ALUs here (matrix and
    // vector fp64 peak are equal on MI355X): a loader wave's gain arithmetic only gets issue
    // slots once the compute waves on its SIMD have finished their MFMA burst (measured: 4-6k
    // cycles for 1.4k of work), so the evaluation belongs at the END of the compute waves' own
    // iteration.  Shares of 64 frames are dealt so that MFMA + gain work per SIMD comes out even
    // (10 compute waves on 4 SIMDs: 3+2, 3+2, 2+3, 2+3 units).
    int share0, share1 = -1, nshares;
    {
        auto ncomp_on = [&](int w) { return (nc - (w & 3) + 3) >> 2; };
        int minc = 1 << 30, maxc = 0;
        for (int sd = 0; sd < 4 && sd < nc; ++sd) {
            minc = min(minc, ncomp_on(sd));
            maxc = max(maxc, ncomp_on(sd));
        }
        const bool uneven = minc < maxc;
        // class A: compute waves on the least loaded SIMDs (one share each, the first wave of
        // such a SIMD a second one); class B: the other compute waves; then the loader waves
        int nA = 0, nA2 = 0, nB = 0, iA = 0, iA2 = 0, iB = 0;
        for (int w = 0; w < nc; ++w) {
            const bool a = uneven && ncomp_on(w) == minc;
            if (a) {
                if (w < wave) ++iA;
                ++nA;
                if (w < 4) {
                    if (w < wave) ++iA2;
                    ++nA2;
                }
            } else {
                if (w < wave) ++iB;
                ++nB;
            }
        }
        nshares = nA + nA2 + nB + (nwaves - nc);
        if (wave >= nc) share0 = nA + nA2 + nB + (wave - nc);
        else if (uneven && ncomp_on(wave) == minc) {
            share0 = iA;
            if (wave < 4) share1 = nA + iA2;
        } else share0 = nA + nA2 + iB;
    }
    const int kind0 = __builtin_amdgcn_readfirstlane(ctl.car[0].slot_kind[0]);
    const DLeaf leaf0 = leaf_uniform(ctl.leaves[min(kCtlLeaves - 1, max(0, __builtin_amdgcn_readfirstlane(ctl.car[0].slot_leaf[0])))]);
    // TWO: two-level evaluation of a sine generator in slot 0 (`Amplify(x, Signal(sin, ω=...))`; the
    // planner picks this instantiation when slot 0 is exactly that and the only slot):
    //   sin(θ(nb) + l·δ) = sin θ(nb) · cos(l·δ) + cos θ(nb) · sin(l·δ)
    // with nb the first frame of a share of 64 and l the lane.  (sin, cos)(l·δ) is a per-lane
    // constant (dtab, written once); the share bases of a whole tile are one full-precision
    // evaluation by ONE wave, lane u -> share u (b_duty, three tiles ahead, published by the
    // tile barriers); a share is then one multiply and one fma per frame instead of the ~100
    // vector instructions of division + reduction + two polynomials, all of which compete with
    // the fp64 MFMAs for the same ALUs.  The phase of frame nb + l is the reference's phase of nb
    // (src/functions.jl:57-60, every operation rounded separately) plus fl(fl(l/fs)·ω): it differs
    // from the reference's own rounding of the phase of nb + l by a few ulp of the phase (~1e-11
    // in the gain at 1e4 cycles, the size of the reference's own rounding error there; the
    // parity bound is 1e-6).
    auto dtab = [&]() __attribute__((always_inline)) { return fbase + (size_t)kFDepth * g.fslots * g.fpitch; };
    auto btab = [&]() __attribute__((always_inline)) { return fbase + (size_t)kFDepth * g.fslots * g.fpitch + 128; };
    // gain array written during iteration `it` (tile it+2) / read for tile `it`
    auto fbw = [&](int it) __attribute__((always_inline)) { return GA ? (it + 2) % 3 : (it & 1); };
    auto fbr = [&](int it) __attribute__((always_inline)) { return GA ? it % 3 : (it & 1); };
    const bool twolvl = TWO && fring && g.ftwo;
    // the wave that evaluates the share bases: the last loader in the loaders' own order (see
    // lidx below), i.e. one that sits on a SIMD with the most compute waves and, with g.nload
    // set, has no copy / modify work of its own
    int bwave = nc;
    if constexpr (TWO) {
        auto ncomp_on = [&](int w) { return (nc - (w & 3) + 3) >> 2; };
        for (int w = nc + 1; w < nwaves; ++w)
            if (ncomp_on(w) >= ncomp_on(bwave)) bwave = w;
    }
    auto b_duty = [&](const TilePos& p, int bb) __attribute__((always_inline)) {
        int64_t xa;
        int nfr;
        if (!twolvl || wave != bwave || !(p.tc < ngrp) || (!is_fast(p, xa, nfr) && !GA) || (g.pad & 64)) return;
        const int nb = (g.tile_len + 63) >> 6;  // <= kRsTwoBases (planner)
        sine_table(xa + (nfr - g.tile_len) + leaf0.df + 1, 64, nb, leaf0.v0, leaf0.v1, leaf0.v2, leaf0.flag,
                   btab() + bb * 2 * kRsTwoBases);
        if (nb > 64)
            sine_table(xa + (nfr - g.tile_len) + leaf0.df + 1 + 64 * 64, 64, nb - 64, leaf0.v0, leaf0.v1, leaf0.v2,
                       leaf0.flag, btab() + bb * 2 * kRsTwoBases + 128);
    };
    auto f_duty = [&](const TilePos& p, int fb, int bb) __attribute__((always_inline)) {
        int64_t xa;
        int nfr;
        if (!fring || share0 * 64 >= g.tile_len || !(p.tc < ngrp) || (!is_fast(p, xa, nfr) && !GA) || (g.pad & 64)) return;
        double* Fb = fbase + (size_t)fb * g.fslots * g.fpitch;
        if constexpr (TWO) {
            if (!twolvl) return;  // (no LDS reserved: the loaders evaluate in place)
            const double2 d = *reinterpret_cast<const double2*>(dtab() + 2 * lane);
#pragma unroll 1
            for (int sp = 0; sp < 2; ++sp) {
                const int shr = sp ? share1 : share0;
                if (shr < 0) break;
                for (int u = shr; u * 64 < g.tile_len; u += nshares) {
                    const int f = (nfr - g.tile_len) + u * 64 + lane;
                    const double2 b = *reinterpret_cast<const double2*>(btab() + bb * 2 * kRsTwoBases + 2 * u);
                    double v = fma(b.x, d.y, b.y * d.x);
                    if (kind0 & 0x100) v = (double)(float)v;
                    if constexpr (GADD) v = (xa + f >= ga_lo && xa + f < ga_hi) ? v : 0.0;
                    if (f < nfr) Fb[f] = v;
                }
            }
        } else {
#pragma unroll 1
            for (int k = 0; k < nslots0; ++k) {
                // (slot 0's recipe is kept in scalar registers for the whole kernel; further slots
                //  are re-read from the LDS control block)
                const int kind = k == 0 ? kind0 : __builtin_amdgcn_readfirstlane(ctl.car[0].slot_kind[k]);
                const DLeaf L = k == 0 ? leaf0 : leaf_uniform(ctl.leaves[__builtin_amdgcn_readfirstlane(ctl.car[0].slot_leaf[k])]);
#pragma unroll 1
                for (int sp = 0; sp < 2; ++sp) {
                    const int shr = sp ? share1 : share0;
                    if (shr < 0) break;
                    for (int f = (nfr - g.tile_len) + shr * 64 + lane; f < nfr; f += nshares * 64) {
                        double v = slot_eval(kind, L, xa + f);
                        if constexpr (GADD) v = (xa + f >= ga_lo && xa + f < ga_hi) ? v : 0.0;
                        Fb[k * g.fpitch + f] = v;
                    }
                }
            }
        }
    };
    // prologue of the two-level evaluation: lane constants, share bases of tiles 0 and 1
    if constexpr (TWO) {
        if (twolvl) {
            if (wave == 0) sine_table(0, 1, 64, leaf0.v0, 0.0, leaf0.v2, leaf0.flag, dtab());
            TilePos pb0 = tile_first();
            b_duty(pb0, 0);
            tile_next(pb0);
            b_duty(pb0, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    // prologue: gains of tiles 0 and 1, published by one extra barrier (both roles)
    TilePos pf = tile_first();  // next tile whose gains are due
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
        f_duty(pf, k, k);
        tile_next(pf);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (TWO) {
        if (twolvl) {  // (tile 2's bases share a buffer with tile 0's, which the gains above just read)
            b_duty(pf, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    // The two roles run separate loops with the same number of workgroup barriers (whole
    // waves take one branch), so their register live ranges do not overlap.
    if (wave >= nc) {
        // ---- loader waves ----
        // Loader waves are numbered so that those on the SIMDs with the fewest compute waves come
        // first (waves go to SIMD wave&3): the first chunks of a tile are the full ones, and a
        // loader wave next to three MFMA-issuing waves runs ~30 % slower than one next to two.
        // (g.nstate: the last nstate loader waves are the IIR state waves; they copy nothing)
        const int nldr = nwaves - (ST ? g.nstate : 0);  // waves [nc, nldr) load
        const bool swave = wave >= nldr;
        int lidx = 0;
        {
            auto ncomp_on = [&](int w) { return (nc - (w & 3) + 3) >> 2; };  // compute waves on w's SIMD
            const int mine = ncomp_on(wave);
            for (int w = nc; w < nldr; ++w) {
                const int other = ncomp_on(w);
                if (other < mine || (other == mine && w < wave)) ++lidx;
            }
            if (swave) lidx = 1 << 20;
        }
        // g.nload > 0: only the first nload loader waves (those on the least loaded SIMDs) copy and
        // modify; the others just keep the barrier count
        const int nactive = g.nload > 0 && g.nload < nldr - nc ? g.nload : nldr - nc;
        const int lthr = nactive * 64;
        const int ltid = (lidx < nactive ? lidx * 64 : (1 << 30)) + lane;
        const int llane = lane;
        const int lw64 = __builtin_amdgcn_readfirstlane(ltid - llane);  // first vector of this wave
        // one step `v (op) slot0`, no Float32 rounding: 0 mul, 1 add, 2 v-m, 3 m-v; -1: step interpreter
        int one0 = -1;
        if (nsteps0 == 1 && (st0.arg[0] & (0x2ff | kCarArr2)) == 0 && !(g.pad & 128))
            one0 = st0.op[0] == OP_MUL ? 0 : st0.op[0] == OP_ADD ? 1 : st0.op[0] == OP_SUB ? ((st0.arg[0] & 0x100) ? 3 : 2) : -1;
        // Loader waves issue a handful of instructions and then sleep on memory; without a
        // raised priority the MFMA-issuing compute waves on the same SIMD win arbitration
        // and the loads only go out once the arithmetic is over (measured: phases add up).
        if (!(g.pad & 16)) __builtin_amdgcn_s_setprio(3);
        const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(lds));  // byte address of the ring
        const uint32_t lane16 = (uint32_t)llane * 16u;                            // per-lane byte offset in a chunk
        // A2: this wave's staging rows of the second array (CT x 1 KB, behind the tile ring -- there is no gain ring here)
        [[maybe_unused]] const uint32_t a2_stage =
            __builtin_amdgcn_readfirstlane(lds_addr(fbase) + (uint32_t)(lidx < nactive ? lidx : 0) * (uint32_t)CT * 1024u);
        // state waves: row offsets of the A operands, first slot of this wave's half of the window
        constexpr int kSwK = 24;  // k-steps per state wave (planner: 2 * kSwK * 4 >= staged span of a row)
        int srow[Q];
        int swk0 = 0;
        if constexpr (ST) {
            swk0 = swave ? (wave - nldr) * kSwK * 4 : 0;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int rho = 16 * q + (lane & 15);
                srow[q] = (rho >> g.ptshift) * g.lds_pitch + (rho & (g.pt - 1)) * (int)g.M + (lane >> 4);
            }
        }
        const int A = S - 2;  // tiles in flight beyond the one being retired
        int cnt0 = 0, cnt1 = 0;  // DMA instructions of the youngest / second youngest issued tile
        auto issue = [&](const TilePos& p, int slot, int tr_it) {
            int n = 0;
            if (p.tc < ngrp && !((g.pad & 2) && tr_it != kRsTraceIters)) {
                int64_t xa;
                int nfr;
                const int c0 = (int)p.tc * CT;
                rs_stamp(g, wave, tr_it, 4);
                if (is_fast(p, xa, nfr)) {
                    const int nvec = (nfr + V - 1) / V;
                    const char* row = (const char*)base0 + ((int64_t)c0 * cs0 + df0 + xa) * (int64_t)sizeof(T);
                    const uint32_t lbase = lds_base + (uint32_t)(slot * bufsz) * (uint32_t)sizeof(T);
                    for (int ivb = lw64; ivb < nvec; ivb += lthr) {
                        const int nact = nvec - ivb;  // lanes of this chunk (scalar)
                        const uint64_t mask = nact >= 64 ? ~0ull : ((1ull << nact) - 1ull);
                        dma_rows<CT>(mask, lane16, row + (int64_t)ivb * 16, cs0 * (int64_t)sizeof(T),
                                     lbase + (uint32_t)ivb * 16u, (uint32_t)g.lds_pitch * (uint32_t)sizeof(T));
                        n += CT;
                        if constexpr (A2) {
                            if (one2 >= 0) {  // (one round per tile: is_fast) the same chunk of the second array -> this wave's staging rows
                                const char* rowb = base2 + ((int64_t)c0 * cs2 + df2 + xa) * (int64_t)sizeof(T);
                                dma_rows<CT>(mask, lane16, rowb + (int64_t)ivb * 16, cs2 * (int64_t)sizeof(T), a2_stage, 1024u);
                                n += CT;
                            }
                        }
                    }
                } else {
                    n = stage_tile_ool<T, CT, 0, A2>(g.n_in, g.lds_pitch, g.pad, xa, nfr, c0, lds + slot * bufsz, &sctl,
                                                 gsrc.car, gsrc.ops, gsrc.leaves, ltid, lthr, 0);
                }
            }
            cnt1 = cnt0;
            cnt0 = n;
        };
        TilePos pn = tile_first();  // next tile to issue
        int sn = 0;                 // ... and its slot
        for (int k = 0; k <= A; ++k) {
            issue(pn, sn, kRsTraceIters);
            tile_next(pn);
            sn = sn + 1 == S ? 0 : sn + 1;
        }
        TilePos pr = tile_first();  // tile being retired
        int sr = 0;                 // ... and its slot
        for (int it = 0;; ++it) {
            const int allowed = A >= 2 ? cnt0 + cnt1 : (A == 1 ? cnt0 : 0);
            const bool live = pr.tc < ngrp;
            rs_stamp(g, wave, it, 0);
            int64_t xa;
            int nfr;
            const bool fast = is_fast(pr, xa, nfr);
            // (a fused source without a gain ring -- it did not fit in LDS -- takes the general
            //  in-place path: same chunk ownership as the fast issue)
            bool a2tile = false;
            if constexpr (A2) a2tile = live && fast && one2 >= 0 && !((g.pad & 2) && it > 0);
            if (a2tile) {
                if constexpr (A2) {
                    const int nvec = (nfr + V - 1) / V;
                    const uint32_t lbase = lds_addr(lds + sr * bufsz);
                    rs_stamp(g, wave, it, 5);
                    wait_vmcnt_le(allowed);  // this tile's DMAs -- both arrays' -- have landed
                    rs_stamp(g, wave, it, 6);
                    const int iv = lw64 + llane;
                    if (iv < nvec) {
                        const uint32_t la = lbase + (uint32_t)iv * 16u, sa = a2_stage + lane16;
                        switch (one2) {
                        case 0: rmw_arr2<T, CT, 0>(la, g.lds_pitch, sa); break;
                        case 1: rmw_arr2<T, CT, 1>(la, g.lds_pitch, sa); break;
                        case 2: rmw_arr2<T, CT, 2>(la, g.lds_pitch, sa); break;
                        default: rmw_arr2<T, CT, 3>(la, g.lds_pitch, sa); break;
                        }
                    }
                    rs_stamp(g, wave, it, 7);
                }
            } else if (live && !((g.pad & 2) && it > 0) && (!fast || (fused0 && !fring))) {
                stage_tile_ool<T, CT, 1, A2>(g.n_in, g.lds_pitch, g.pad, xa, nfr, (int)pr.tc * CT, lds + sr * bufsz,
                                         &sctl, gsrc.car, gsrc.ops, gsrc.leaves, ltid, lthr, allowed);
            } else if (live && fast && fring && !GA) {
                // carrier 0's steps in place on the chunks this wave copied, gains from the ring
                const int nvec = (nfr + 1) >> 1;
                const uint32_t lbase = lds_addr(lds + sr * bufsz);
                const double* Fb = fbase + (size_t)(it & 1) * g.fslots * g.fpitch;
                rs_stamp(g, wave, it, 5);
                wait_vmcnt_le(allowed);  // this tile's DMA landed in LDS
                rs_stamp(g, wave, it, 6);
                for (int ivb = lw64; ivb < nvec; ivb += lthr) {
                    const int iv = ivb + llane;
                    if (one0 >= 0) {
                        if (iv < nvec) {
                            const uint32_t la = lbase + (uint32_t)iv * 16u, fa = lds_addr(Fb) + (uint32_t)iv * 16u;
                            switch (one0) {
                            case 0: rmw_one_chunk<CT, 0>(la, g.lds_pitch, fa); break;
                            case 1: rmw_one_chunk<CT, 1>(la, g.lds_pitch, fa); break;
                            case 2: rmw_one_chunk<CT, 2>(la, g.lds_pitch, fa); break;
                            default: rmw_one_chunk<CT, 3>(la, g.lds_pitch, fa); break;
                            }
                        }
                    } else if (iv < nvec) {
                        double F[kMaxFrameSlots][2];
#pragma unroll
                        for (int k = 0; k < kMaxFrameSlots; ++k) {
                            F[k][0] = F[k][1] = 0.0;
                            if (k < nslots0) {
                                const double2 v = *reinterpret_cast<const double2*>(Fb + k * g.fpitch + 2 * iv);
                                F[k][0] = v.x;
                                F[k][1] = v.y;
                            }
                        }
                        rmw_chunk<CT, false>(lbase + (uint32_t)iv * 16u, g.lds_pitch, st0, F);
                    }
                }
                rs_stamp(g, wave, it, 7);
            } else {
                wait_vmcnt_le(allowed);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS stores / in-place steps done
            rs_stamp(g, wave, it, 1);
            __builtin_amdgcn_s_barrier();  // publishes this tile; the slot of the previous one is free
            rs_stamp(g, wave, it, 2);
            if (!live) break;  // (the compute waves' last barrier)
            issue(pn, sn, it);
            rs_stamp(g, wave, it, 3);
            if constexpr (ST) if (swave) {
                // ---- IIR state pass on tile `it` (published by the barrier above, slot sr) ----
                // Y[32 rows x 16] = X[32 x 4*ksh] * wtab[4*ksh x 16] over this wave's HALF of the staged
                // span (the other state wave takes the other half; k_sos_combine adds the two): the
                // taps sit in registers for the whole kernel like the compute waves' (streamed from
                // L2 they cost ~1000 cycles per dependent round while the chip streams: measured
                // 1.56 ms for the kernel), the A operands come from LDS one k-step ahead.
                const int sh = (int)(pr.xb & (kAlign - 1));
                const T* __restrict__ cur = lds + sr * bufsz + sh + swk0;
                const double* __restrict__ wl = fbase + (size_t)kFDepth * g.fslots * g.fpitch + (g.ftwo ? kRsTwoDoubles : 0) +
                                                (size_t)(swk0 + (lane >> 4)) * 10 + ((lane & 15) < 10 ? (lane & 15) : 0);
                v4d acc[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
                double abuf[2][Q], bbuf[2];
                bbuf[0] = wl[0];
#pragma unroll
                for (int q = 0; q < Q; ++q) abuf[0][q] = (double)cur[srow[q]];
#pragma unroll
                for (int s = 0; s < kSwK; ++s) {
                    if (s + 1 < kSwK) {
#pragma unroll
                        for (int q = 0; q < Q; ++q) abuf[(s + 1) & 1][q] = (double)cur[srow[q] + 4 * (s + 1)];
                        bbuf[(s + 1) & 1] = wl[40 * (s + 1)];
                    }
#pragma unroll
                    for (int q = 0; q < Q; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[s & 1][q], bbuf[s & 1], acc[q], 0, 0, 0);
                }
                // Eight stores, no waits between them: every address is ONE per-lane base plus a
                // wave-uniform offset (an address computed per store spilled and reloaded from scratch
                // here, and a scratch reload waits vmcnt(0) -- i.e. for the previous STORE: 12 000 cycles
                // per tile, measured).  Rows (lane>>4)+4i+16q are (period (lane>>4)&(pt-1)..., channel ...).
                // Row (lane>>4) + 4i + 16q of the result is (period kq + (4i mod pt), channel (16q + 4i) / pt),
                // kq = lane >> 4 (no carry: kq < 4 <= pt): the per-lane part of the address is ONE pointer.
                constexpr int PT = (16 * Q) / CT;
                const int kq = lane >> 4;
                const int64_t nper = g.nperiods;
                double* vl = g.vper + ((size_t)(wave - nldr) * (size_t)g.nch + (size_t)((int)pr.tc * CT)) * (size_t)nper * 16 +
                             (size_t)(pr.tx * PT + kq) * 16 + (lane & 15);
                const int plim = (int)(nper - pr.tx * PT < PT ? nper - pr.tx * PT : PT);  // periods of this tile inside the signal
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        constexpr int kDummy = 0;
                        (void)kDummy;
                        const int ppi = (4 * i) % PT, cci = (16 * q + 4 * i) / PT;  // compile-time after unrolling
                        if (kq + ppi < plim) vl[((size_t)cci * (size_t)nper + (size_t)ppi) * 16] = acc[q][i];
                    }
            }
            f_duty(pf, fbw(it), it & 1);  // gains of tile it+2
            rs_stamp(g, wave, it, 4);
            tile_next(pf);
            if constexpr (TWO) b_duty(pf, (it + 1) & 1);  // share bases of tile it+3
            tile_next(pn);
            tile_next(pr);
            sn = sn + 1 == S ? 0 : sn + 1;
            sr = sr + 1 == S ? 0 : sr + 1;
        }
        return;
    }
    // ---- compute waves ----
    const int kq = lane >> 4, n16 = lane & 15;
    const int ptmask = g.pt - 1, ptshift = g.ptshift;  // pt is a power of two
    const int gbeg = wave * G;
    using AT = typename std::conditional<F32M, float, double>::type;  // MFMA operand / accumulator element
    using ACC = typename std::conditional<F32M, v4f, v4d>::type;
    AT breg[G][KS];  // taps of this wave's G groups: registers for the whole kernel
    int rowoff[Q];
    int goff[Q];  // GA: the same offset without the channel row (gains depend on the frame only)
#pragma unroll
    for (int gg = 0; gg < G; ++gg)
#pragma unroll
        for (int s = 0; s < KS; ++s)
            breg[gg][s] = gbeg + gg < g.ngroups ? (AT)tab[((size_t)(gbeg + gg) * KS + s) * 64 + lane] : (AT)0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int rho = 16 * q + n16;  // A operand: row m = lane & 15 of row-tile q
        rowoff[q] = (rho >> ptshift) * g.lds_pitch + (rho & ptmask) * (int)g.M - g.jlo - (KS * 4 - 1) + kq;
        goff[q] = (rho & ptmask) * (int)g.M - g.jlo - (KS * 4 - 1) + kq;
    }
    int jeg[G];  // newest input of each group's window
#pragma unroll
    for (int gg = 0; gg < G; ++gg) jeg[gg] = gbeg + gg < g.ngroups ? jend[gbeg + gg] : 0;
    // output offsets of this lane's 4 accumulator rows per row-tile, relative to the tile's
    // (channel c0, period P0) origin
    int64_t yoff[Q][4];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rho = F32M ? 16 * q + 4 * kq + i : 16 * q + kq + 4 * i;  // D: row = (lane>>4) + 4*reg (Float32 MFMA: 4*(lane>>4) + reg)
            yoff[q][i] = (int64_t)(rho >> ptshift) * g.out_pitch + (int64_t)(rho & ptmask) * g.L + n16;
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // taps loaded
    __builtin_amdgcn_s_barrier();  // tile 0 staged
    int slot = 0, it = 0;
    for (TilePos p = tile_first(); p.tc < ngrp; tile_next(p), ++it) {
        rs_stamp(g, wave, it, 0);
        const int sh = (int)(p.xb & (kAlign - 1));
        const T* __restrict__ cur = lds + slot * bufsz + sh;
        const int64_t P0 = p.tx * g.pt;
        const int c0 = (int)p.tc * CT;
        TO* __restrict__ ytile = y + ((int64_t)c0 * g.out_pitch + P0 * g.L);
        // interior tile: every row's period is complete -> no per-element bounds checks
        const bool interior = P0 + g.pt <= g.nperiods && (P0 + g.pt) * g.L <= g.n_out;
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            const int gi = gbeg + gg;
            if (gi < g.ngroups && !(g.pad & 1)) {
                const int je = jeg[gg];
                ACC acc[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] = ACC{0, 0, 0, 0};
                // A operands are software-pipelined one k-step ahead (double-buffered
                // registers) so the LDS latency hides under the previous step's MFMAs; the
                // per-step address is an immediate offset from fixed row pointers.
                const T* __restrict__ ap[Q];
                const double* __restrict__ gp[Q];  // GA: this tile's gains at the A operand's frames
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    ap[q] = cur + (rowoff[q] + je);
                    gp[q] = fbase + (size_t)fbr(it) * g.fslots * g.fpitch + (sh + goff[q] + je);
                }
                AT abuf[2][Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    abuf[0][q] = (AT)ap[q][0];
                    if constexpr (GA && GADD) abuf[0][q] += gp[q][0];
                    else if constexpr (GA) abuf[0][q] *= gp[q][0];
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s + 1 < KS) {
#pragma unroll
                        for (int q = 0; q < Q; ++q) {
                            abuf[(s + 1) & 1][q] = (AT)ap[q][4 * (s + 1)];
                            if constexpr (GA && GADD) abuf[(s + 1) & 1][q] += gp[q][4 * (s + 1)];
                            else if constexpr (GA) abuf[(s + 1) & 1][q] *= gp[q][4 * (s + 1)];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if constexpr (F32M) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[s & 1][q], breg[gg][s], acc[q], 0, 0, 0);
                        else acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[s & 1][q], breg[gg][s], acc[q], 0, 0, 0);
                    }
                }
                const int r = gi * 16 + n16;  // output index inside the period
                // A non-finite accumulator: the group's window -- wider than any single output's, and padded with zero taps --
                // held a non-finite sample (0 * NaN is NaN).  The reference multiplies each output's own taps only: the tile
                // and group go onto a list, and the launch behind this one recomputes them output by output (k_rs_fixup).
                if (g.nf != nullptr) {
                    bool bad = false;
#pragma unroll
                    for (int q = 0; q < Q; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) bad |= !isfinite(acc[q][i]);
                    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) {
                        const uint32_t idx = atomicAdd(g.nf, 1u);
                        if (idx < kRsNfCap) {
                            uint32_t* e = g.nf + 4 + 4 * (size_t)idx;
                            e[0] = (uint32_t)(uint64_t)P0;
                            e[1] = (uint32_t)((uint64_t)P0 >> 32);
                            e[2] = (uint32_t)c0;
                            e[3] = (uint32_t)gi;
                        }
                    }
                }
                if (g.pad & 4) continue;
                if (interior && gi * 16 + 16 <= g.L) {
#pragma unroll
                    for (int q = 0; q < Q; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (g.pad & 256) __builtin_nontemporal_store((TO)acc[q][i], &ytile[yoff[q][i] + gi * 16]);
                            else ytile[yoff[q][i] + gi * 16] = (TO)acc[q][i];
                        }
                } else {
#pragma unroll
                    for (int q = 0; q < Q; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int rho = F32M ? 16 * q + 4 * kq + i : 16 * q + kq + 4 * i;
                            const int64_t period = P0 + (rho & ptmask);
                            const int64_t m = period * g.L + r;
                            if (period < g.nperiods && r < g.L && m < g.n_out)
                                ytile[yoff[q][i] + gi * 16] = (TO)acc[q][i];
                        }
                }
            }
        }
        // Raw barrier: __syncthreads() would also drain vmcnt(0), i.e. make the compute waves
        // wait for their output stores every tile.  The LDS reads of this tile were consumed
        // by the MFMAs above, so only lgkmcnt matters here; stores stay in flight.
        rs_stamp(g, wave, it, 3);
        f_duty(pf, fbw(it), it & 1);  // this wave's share of the gains of tile it+2
        tile_next(pf);
        rs_stamp(g, wave, it, 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rs_stamp(g, wave, it, 1);
        __builtin_amdgcn_s_barrier();  // next tile published; all finished reading this one
        rs_stamp(g, wave, it, 2);
        slot = slot + 1 == S ? 0 : slot + 1;
    }
}

// loader waves that copy (RsPeriodic::nload, or all of them): the A2 instantiation's staging areas
static int rs_a2_loaders(const RsPeriodic& g) {
    const int nl = g.nwaves - g.ncompute;
    return g.nload > 0 && g.nload < nl ? g.nload : nl;
}
template <typename T, int CT, int KS, int G, bool TWO = false, typename TO = T, bool GA = false, bool ST = false, bool GADD = false, int Q = 2, bool F32M = false, bool A2 = false>
static void launch_rp_k(void* y, const double* tab, const int* jend, const RsPeriodic& g,
                        const RsGlobalTables& gsrc, hipStream_t st) {
    const int64_t ntiles = ((g.nperiods + g.pt - 1) / g.pt) * (g.nch / CT);
    dim3 grid((unsigned)std::min<int64_t>(ntiles, g.grid));
    size_t lds = (((size_t)g.nslots * CT * g.lds_pitch * sizeof(T) + 7) / 8 + (size_t)(GA ? 3 : 2) * g.fslots * g.fpitch + (g.ftwo ? kRsTwoDoubles : 0) +
                  (ST ? (size_t)4 * g.ksw * 10 : 0)) * 8;  // + static RsCtl
    if constexpr (A2) lds += (size_t)rs_a2_loaders(g) * CT * 1024;  // the loader waves' staging rows of the second array
    static bool seen[64];
    if (first_use_on_device(seen))
        (void)hipFuncSetAttribute((const void*)k_resample_periodic<T, CT, KS, G, TWO, TO, GA, ST, GADD, Q, F32M, A2>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - (int)sizeof(RsCtl) - 64);  // static: the control block
    hipLaunchKernelGGL((k_resample_periodic<T, CT, KS, G, TWO, TO, GA, ST, GADD, Q, F32M, A2>), grid, dim3(64 * g.nwaves), lds, st, tab,
                       jend, g, (TO*)y, gsrc);
}

// ST instantiations: Float64, 14 k-steps, one group per compute wave, state waves on
template <int CT>
static int launch_rp_st(void* y, const double* tab, const int* jend, const RsPeriodic& g,
                        const RsGlobalTables& gsrc, hipStream_t st) {
    const int gper = (g.ngroups + g.ncompute - 1) / g.ncompute;
    if (g.kw != 4 * 14 || gper != 1 || g.out_f32 || g.ga) return -1;
    if (g.ftwo) launch_rp_k<double, CT, 14, 1, true, double, false, true>(y, tab, jend, g, gsrc, st);
    else launch_rp_k<double, CT, 14, 1, false, double, false, true>(y, tab, jend, g, gsrc, st);
    return 0;
}

// GA instantiations: Float32 tiles, Float64 arithmetic and (unless the sink buffer is Float32) result
template <int CT>
static int launch_rp_ga(void* y, const double* tab, const int* jend, const RsPeriodic& g,
                        const RsGlobalTables& gsrc, hipStream_t st) {
    const int gper = (g.ngroups + g.ncompute - 1) / g.ncompute;
    if (g.kw != 4 * 14 || gper != 1) return -1;
    if (g.ga == 2) {  // the fused step is an add (`Mix`)
        if (g.out_f32) {
            if (g.ftwo) launch_rp_k<float, CT, 14, 1, true, float, true, false, true>(y, tab, jend, g, gsrc, st);
            else launch_rp_k<float, CT, 14, 1, false, float, true, false, true>(y, tab, jend, g, gsrc, st);
        } else {
            if (g.ftwo) launch_rp_k<float, CT, 14, 1, true, double, true, false, true>(y, tab, jend, g, gsrc, st);
            else launch_rp_k<float, CT, 14, 1, false, double, true, false, true>(y, tab, jend, g, gsrc, st);
        }
        return 0;
    }
    if (g.out_f32) {
        if (g.ftwo) launch_rp_k<float, CT, 14, 1, true, float, true>(y, tab, jend, g, gsrc, st);
        else launch_rp_k<float, CT, 14, 1, false, float, true>(y, tab, jend, g, gsrc, st);
    } else {
        if (g.ftwo) launch_rp_k<float, CT, 14, 1, true, double, true>(y, tab, jend, g, gsrc, st);
        else launch_rp_k<float, CT, 14, 1, false, double, true>(y, tab, jend, g, gsrc, st);
    }
    return 0;
}

template <typename T, int CT>
static int launch_rp_ct(void* y, const double* tab, const int* jend, const RsPeriodic& g,
                        const RsGlobalTables& gsrc, hipStream_t st) {
    const int gper = (g.ngroups + g.ncompute - 1) / g.ncompute;
#define SO_RP(KS_, G_)                                                                   \
    if (g.kw == 4 * KS_ && gper == G_) {                                                  \
        if constexpr (sizeof(T) == 8 && G_ == 1) {                                        \
            if constexpr (CT >= 4) {                                                      \
                if (g.out_f32) { /* Float64 arithmetic, Float32 result */                 \
                    if (g.ftwo) launch_rp_k<T, CT, KS_, G_, true, float>(y, tab, jend, g, gsrc, st); \
                    else launch_rp_k<T, CT, KS_, G_, false, float>(y, tab, jend, g, gsrc, st); \
                    return 0;                                                             \
                }                                                                         \
            }                                                                             \
            if (g.ftwo && !g.out_f32) {                                                   \
                launch_rp_k<T, CT, KS_, G_, true>(y, tab, jend, g, gsrc, st);             \
                return 0;                                                                 \
            }                                                                             \
        }                                                                                 \
        if (g.out_f32) return -1;                                                         \
        if constexpr (sizeof(T) == 4 && G_ == 1 && CT >= 4 && KS_ <= 20) {                \
            if (g.f32m) { /* Float32 all the way: the Float32 MFMA */                     \
                launch_rp_k<T, CT, KS_, G_, false, T, false, false, false, 2, true>(y, tab, jend, g, gsrc, st); \
                return 0;                                                                 \
            }                                                                             \
        }                                                                                 \
        launch_rp_k<T, CT, KS_, G_>(y, tab, jend, g, gsrc, st);                           \
        return 0;                                                                         \
    }
    SO_RP(12, 1) SO_RP(14, 1) SO_RP(16, 1) SO_RP(20, 1) SO_RP(28, 1) SO_RP(14, 2) SO_RP(14, 3)
#undef SO_RP
    return -1;
}

// 16-row tiles (RsPeriodic::rows == 16): one group per compute wave, long windows (Float32 signals too since the end of
// round 4: a 101-tap FIR or 44.1 -> 16 kHz on Float32 data went to the row-tiled kernel, 1.04 ms where Float64 took 0.66)
template <typename T, int CT>
static int launch_rp_q1(void* y, const double* tab, const int* jend, const RsPeriodic& g, const RsGlobalTables& gsrc,
                        hipStream_t st) {
    const int gper = (g.ngroups + g.ncompute - 1) / g.ncompute;
    if (gper != 1 || g.out_f32 || g.ga || g.nstate) return -1;
    if (sizeof(T) == 4 && g.ftwo) return -1;
#define SO_RQ(KS_)                                                                                                  \
    if (g.kw == 4 * KS_) {                                                                                          \
        if constexpr (sizeof(T) == 8) {                                                                             \
            if (g.ftwo) {                                                                                           \
                launch_rp_k<T, CT, KS_, 1, true, T, false, false, false, 1>(y, tab, jend, g, gsrc, st);             \
                return 0;                                                                                           \
            }                                                                                                       \
        }                                                                                                           \
        launch_rp_k<T, CT, KS_, 1, false, T, false, false, false, 1>(y, tab, jend, g, gsrc, st);                    \
        return 0;                                                                                                   \
    }
    SO_RQ(28) SO_RQ(36)
#undef SO_RQ
    return -1;
}

// everything the periodic kernel has for one tile type and channel count: this unit's instantiations
template <typename T, int CT>
static int launch_rp_unit(void* y, const double* tab, const int* jend, const RsPeriodic& g, const RsGlobalTables& gsrc, hipStream_t st) {
    constexpr bool F = sizeof(T) == 4;
    constexpr bool wide = CT == 8 || CT == 4;
    if (g.arr2) {  // a step on a second array: the A2 instantiations (14 k-steps, one group per compute wave; Float64, or -- two
                   // Float32 arrays of a Float32 signal -- the Float32 MFMA's)
        const int gper = (g.ngroups + g.ncompute - 1) / g.ncompute;
        if (g.rows != 32 || g.kw != 4 * 14 || gper != 1 || g.out_f32 || g.ga || g.nstate) return -1;
        if constexpr (wide && F) {
            if (!g.f32m) return -1;
            launch_rp_k<float, CT, 14, 1, false, float, false, false, false, 2, true, true>(y, tab, jend, g, gsrc, st);
            return 0;
        } else if constexpr (wide) {
            launch_rp_k<double, CT, 14, 1, false, double, false, false, false, 2, false, true>(y, tab, jend, g, gsrc, st);
            return 0;
        }
        return -1;
    }
    if (g.rows == 16) {
        if constexpr (wide) return launch_rp_q1<T, CT>(y, tab, jend, g, gsrc, st);
        return -1;
    }
    if (g.nstate > 0) {
        if constexpr (wide && !F) return launch_rp_st<CT>(y, tab, jend, g, gsrc, st);
        return -1;
    }
    if (g.ga) {
        if constexpr (wide && F) return launch_rp_ga<CT>(y, tab, jend, g, gsrc, st);
        return -1;
    }
    return launch_rp_ct<T, CT>(y, tab, jend, g, gsrc, st);
}
#endif  // SO_RP_UNIT != 0

#define SO_RP_ARGS void *y, const double *tab, const int *jend, const RsPeriodic &g, const RsGlobalTables &gsrc, hipStream_t st
#define SO_RP_UNITS(X) X(1, d8, double, 8) X(2, d4, double, 4) X(3, d2, double, 2) X(4, d1, double, 1) X(5, f8, float, 8) X(6, f4, float, 4) X(7, f2, float, 2) X(8, f1, float, 1)
#if SO_RP_UNIT == 0
#define SO_RP_X(ID, TAG, T, CT) int launch_rp_unit_##TAG(SO_RP_ARGS);
#elif SO_RP_UNIT < 0
#define SO_RP_X(ID, TAG, T, CT) \
    int launch_rp_unit_##TAG(SO_RP_ARGS) { return launch_rp_unit<T, CT>(y, tab, jend, g, gsrc, st); }
#else
#define SO_RP_X(ID, TAG, T, CT) SO_RP_X2(ID, TAG, T, CT)
#define SO_RP_X2(ID, TAG, T, CT) SO_RP_IF_##ID(int launch_rp_unit_##TAG(SO_RP_ARGS) { return launch_rp_unit<T, CT>(y, tab, jend, g, gsrc, st); })
#endif
// (SO_RP_IF_<id>(code...): `code` in the unit with that id, nothing elsewhere)
#define SO_RP_IF_1(...)
#define SO_RP_IF_2(...)
#define SO_RP_IF_3(...)
#define SO_RP_IF_4(...)
#define SO_RP_IF_5(...)
#define SO_RP_IF_6(...)
#define SO_RP_IF_7(...)
#define SO_RP_IF_8(...)
#if SO_RP_UNIT == 1
#undef SO_RP_IF_1
#define SO_RP_IF_1(...) __VA_ARGS__
#elif SO_RP_UNIT == 2
#undef SO_RP_IF_2
#define SO_RP_IF_2(...) __VA_ARGS__
#elif SO_RP_UNIT == 3
#undef SO_RP_IF_3
#define SO_RP_IF_3(...) __VA_ARGS__
#elif SO_RP_UNIT == 4
#undef SO_RP_IF_4
#define SO_RP_IF_4(...) __VA_ARGS__
#elif SO_RP_UNIT == 5
#undef SO_RP_IF_5
#define SO_RP_IF_5(...) __VA_ARGS__
#elif SO_RP_UNIT == 6
#undef SO_RP_IF_6
#define SO_RP_IF_6(...) __VA_ARGS__
#elif SO_RP_UNIT == 7
#undef SO_RP_IF_7
#define SO_RP_IF_7(...) __VA_ARGS__
#elif SO_RP_UNIT == 8
#undef SO_RP_IF_8
#define SO_RP_IF_8(...) __VA_ARGS__
#endif
SO_RP_UNITS(SO_RP_X)
#undef SO_RP_X

#if SO_RP_UNIT <= 0
// returns 0 when launched, -1 if no instantiation fits (caller falls back to k_resample)
int launch_resample_periodic(void* y, const double* tab, const int* jend, const RsPeriodic& g,
                             int dtype, const RsGlobalTables& gsrc, hipStream_t st) {
    if (g.n_out <= 0) return 0;
    // the tile type: Float32 tiles for Float32 signals and for the GA instantiations (a Float32 array under a Float64 step)
    const bool f32 = g.ga ? true : g.nstate > 0 ? false : dtype == SO_F32;
    if (!f32 && !g.nstate && dtype != SO_F64) return -1;
    if (f32) {
        switch (g.ct) {
        case 8: return launch_rp_unit_f8(y, tab, jend, g, gsrc, st);
        case 4: return launch_rp_unit_f4(y, tab, jend, g, gsrc, st);
        case 2: return launch_rp_unit_f2(y, tab, jend, g, gsrc, st);
        default: return g.arr2 || g.rows == 16 || g.ga ? -1 : launch_rp_unit_f1(y, tab, jend, g, gsrc, st);
        }
    }
    switch (g.ct) {
    case 8: return launch_rp_unit_d8(y, tab, jend, g, gsrc, st);
    case 4: return launch_rp_unit_d4(y, tab, jend, g, gsrc, st);
    case 2: return launch_rp_unit_d2(y, tab, jend, g, gsrc, st);
    default: return g.arr2 || g.rows == 16 || g.nstate ? -1 : launch_rp_unit_d1(y, tab, jend, g, gsrc, st);
    }
}

// ---------------------------------------------------------------------------
// K3f: sparse fix-up of the outputs DSP.jl's floating-point phase accumulator positions
// differently from the closed form (RsFix list built by the planner, see sigops_internal.h).
// One wave per listed output: lane k owns tap ages k, k+64, ...; the per-frame values of a
// fused source are evaluated once per tap and shared by all channels; the two inner products
// (pfb and dpfb rows, DSP.jl FIRArbitrary: yLower + alpha*yUpper) are reduced across the wave.
// Runs after the main resampler kernel on the same stream and overwrites y[m].
__global__ __launch_bounds__(kBlock) void k_resample_fix(RsFixArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (e >= a.nfix) return;
    const RsFix fx = a.fix[e];
    const double* pf = a.pfb + (int64_t)fx.p * a.taps;
    const double* df = a.dpfb + (int64_t)fx.p * a.taps;
    const bool st32 = a.stage_dtype == SO_F32;
    for (int c = 0; c < a.nch; ++c) {
        double lo = 0.0, hi = 0.0;
        for (int k = lane; k < a.taps; k += 64) {
            const int64_t n = fx.j - k;
            double xv = 0.0;
            if (n >= 0 && n < a.n_in) {
                if (a.ncar > 0) {
                    int ck = 0;
                    while (ck + 1 < a.ncar && a.car[ck].b <= n) ++ck;
                    const DCarrier& C = a.car[ck];
                    if (n >= C.a && n < C.b) {
                        double F[kMaxFrameSlots][1];
#pragma unroll
                        for (int s = 0; s < kMaxFrameSlots; ++s) F[s][0] = 0.0;
                        if ((C.nsteps > 0 || C.pad_) && C.frame_len > 0) {
                            const int64_t nn[1] = {n};
                            double fo[1];
                            run_program<1, false, 2, true>(a.ops, C.frame_pc, C.frame_len, a.leaves, nn, c, F, fo);
                        }
                        double val[1][1] = {{C.pad_ == 2 ? 1.0 : 0.0}};
                        if (C.base != nullptr) {
                            const int64_t off = (int64_t)c * C.cstride + n + C.df;
                            val[0][0] = C.dtype == SO_F32 ? (double)SO_GLOBAL_PTR(float, C.base)[off]
                                                          : SO_GLOBAL_PTR(double, C.base)[off];
                        }
                        if (C.nsteps == 1 && (C.arg[0] & kCarArr2) && C.base != nullptr) carrier_arr2<1, 1>(C, c, n, false, val);
                        else if (C.nsteps > 0) carrier_apply<1, 1>(C, F, val, st32);
                        if (C.pad_ >= 3) val[0][0] += F[0][0];     // GA carrier, add: Float32 sample plus its Float64 operand
                        else if (C.pad_) val[0][0] *= F[0][0];  // GA carrier: Float32 sample times its Float64 gain
                        xv = st32 ? (double)(float)val[0][0] : val[0][0];
                    }
                } else {
                    const int64_t off = (int64_t)c * a.in_pitch + n;
                    xv = a.in_dtype == SO_F32 ? (double)((const float*)a.x)[off] : ((const double*)a.x)[off];
                }
            }
            lo = fma(pf[k], xv, lo);
            hi = fma(df[k], xv, hi);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo += __shfl_xor(lo, off, 64);
            hi += __shfl_xor(hi, off, 64);
        }
        if (lane == 0) {
            double r = lo + hi * fx.alpha;
            if (st32) r = (double)(float)r;
            const int64_t o = (int64_t)c * a.out_pitch + fx.m;
            if (a.out_dtype == SO_F32) ((float*)a.y)[o] = (float)r;
            else ((double*)a.y)[o] = r;
        }
    }
}

void launch_resample_fix(const RsFixArgs& a, hipStream_t st) {
    if (a.nfix <= 0) return;
    const int per = kBlock / 64;
    hipLaunchKernelGGL(k_resample_fix, dim3((unsigned)((a.nfix + per - 1) / per)), dim3(kBlock), 0, st, a);
}

#endif  // SO_RP_UNIT <= 0

}  // namespace so
