// Round-3 kernels (a translation unit of their own).
//
//   k_sos_exact   K2x  SOS IIR in DSP.jl's own order of operations -- one sequence per channel from the
//                      first frame to the last, every product and sum rounded on its own (Julia does
//                      not contract a*b + c; reference src/filters.jl:252-255 -> DSP.jl `filt!` for
//                      SecondOrderSections).  The planner selects it for ill-conditioned cascades
//                      (SosGeom::exact, stages.cpp sos_rounding_sensitivity), where the chunked scan's
//                      and a fused multiply-add's different rounding is amplified to 1e-7 ... 1e-3.
#include <hip/hip_runtime.h>

#include "../../include/sigops.h"
#include "kernels.h"
#include "krespos.h"
#include "sigops_internal.h"

namespace so {

template <int NS>
__device__ __forceinline__ double sos_step_exact(double x, double (&s)[2 * NS], const SosCoefs& cf) {
#pragma clang fp contract(off)
    double y = x;
#pragma unroll
    for (int f = 0; f < NS; ++f) {
        const double xi = y;
        const double p0 = cf.b0[f] * xi;
        y = s[2 * f] + p0;
        const double p1 = cf.b1[f] * xi, p2 = cf.a1[f] * y;
        const double q1 = s[2 * f + 1] + p1;
        s[2 * f] = q1 - p2;
        const double p3 = cf.b2[f] * xi, p4 = cf.a2[f] * y;
        s[2 * f + 1] = p3 - p4;
    }
    return y;
}

// One wave = up to 64 channels; a tile is 64 rows x 16 frames parked in LDS with an odd pitch (global
// accesses are 128-byte row segments, every lane walks its own row), the next tile's loads are in
// flight during the arithmetic.  NS2 > 0: a second group of sections follows the first on the same
// sample (cascades of 9 ... 16 sections: no rounding to the sample type between the groups).
constexpr int kXT = 16;
template <int NS, int NS2, typename T>
__global__ __launch_bounds__(64) void k_sos_exact(const T* __restrict__ x, T* __restrict__ y, SosGeom g, SosCoefs cf,
                                                  SosCoefs cf2) {
#pragma clang fp contract(off)
    __shared__ double tile[64 * (kXT + 1)];
    const int lane = threadIdx.x;
    const int ch0 = blockIdx.x * 64;
    const int nrow = g.nch - ch0 < 64 ? g.nch - ch0 : 64;
    double s[2 * NS], s2[2 * (NS2 > 0 ? NS2 : 1)];
#pragma unroll
    for (int d = 0; d < 2 * NS; ++d) s[d] = 0.0;
#pragma unroll
    for (int d = 0; d < 2 * (NS2 > 0 ? NS2 : 1); ++d) s2[d] = 0.0;
    const int rsub = lane >> 4, col = lane & 15;
    const T* xrow[16];
    bool rok[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = j * 4 + rsub;
        rok[j] = r < nrow;
        xrow[j] = x + ((int64_t)(ch0 + (rok[j] ? r : 0)) * g.in_pitch + col);
    }
    double xv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) xv[j] = rok[j] && col < g.n ? (double)xrow[j][0] : 0.0;
    const double gain = NS2 > 0 ? cf2.gain : cf.gain;
    for (int64_t t0 = 0; t0 < g.n; t0 += kXT) {
#pragma unroll
        for (int j = 0; j < 16; ++j) tile[(j * 4 + rsub) * (kXT + 1) + col] = xv[j];
        __builtin_amdgcn_wave_barrier();
        if (t0 + kXT < g.n) {
#pragma unroll
            for (int j = 0; j < 16; ++j) xv[j] = rok[j] && t0 + kXT + col < g.n ? (double)xrow[j][t0 + kXT] : 0.0;
        }
        double* row = tile + lane * (kXT + 1);
#pragma unroll
        for (int t = 0; t < kXT; ++t) {
            double yv = sos_step_exact<NS>(row[t], s, cf);
            if constexpr (NS2 > 0) yv = sos_step_exact<NS2>(yv, s2, cf2);
            row[t] = yv * gain;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const int r = j * 4 + rsub;
            if (r < nrow && t0 + col < g.n && t0 + col >= g.store_lo) {
                const int64_t o = (int64_t)(ch0 + r) * g.out_pitch + t0 + col;
                if (sizeof(T) == 8 && g.out_dtype == SO_F32) reinterpret_cast<float*>(y)[o] = (float)tile[r * (kXT + 1) + col];
                else y[o] = (T)tile[r * (kXT + 1) + col];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NS, int NS2>
static void launch_exact_t(const void* x, void* y, const SosGeom& g, const SosCoefs& a, const SosCoefs& b, hipStream_t st) {
    const dim3 grid((unsigned)((g.nch + 63) / 64));
    if (g.in_dtype == SO_F32)
        hipLaunchKernelGGL((k_sos_exact<NS, NS2, float>), grid, dim3(64), 0, st, (const float*)x, (float*)y, g, a, b);
    else
        hipLaunchKernelGGL((k_sos_exact<NS, NS2, double>), grid, dim3(64), 0, st, (const double*)x, (double*)y, g, a, b);
}

// cascades of up to 16 sections in one launch (a: sections 1..8, b: sections 9..16 or nsec == 0);
// returns 0 when launched, -1 for a shape it has no instantiation for
int launch_sos_exact(const void* x, void* y, const SosGeom& g, const SosCoefs& a, const SosCoefs& b, hipStream_t st) {
    if (g.n <= 0) return 0;
#define SO_X1(N_) case N_: launch_exact_t<N_, 0>(x, y, g, a, b, st); return 0;
#define SO_X2(N_) case N_: launch_exact_t<8, N_>(x, y, g, a, b, st); return 0;
    if (b.nsec == 0) {
        switch (a.nsec) { SO_X1(1) SO_X1(2) SO_X1(3) SO_X1(4) SO_X1(5) SO_X1(6) SO_X1(7) SO_X1(8) default: return -1; }
    }
    if (a.nsec != 8) return -1;
    switch (b.nsec) { SO_X2(1) SO_X2(2) SO_X2(3) SO_X2(4) SO_X2(5) SO_X2(6) SO_X2(7) SO_X2(8) default: return -1; }
#undef SO_X1
#undef SO_X2
}

// ---------------------------------------------------------------------------
// K2 exact scan.  The default pass 2 (k_sos_scan) sums the K nearest chunks only: what a chunk's state
// contributes has decayed below 2^-70 of ITS OWN size after K chunks.  That is below rounding for every
// sample that is not itself 2^70 times smaller than what came before -- but the tail of a filter long
// after its input went silent is exactly that, and a `Normpower` of such a tail (reference
// src/filters.jl:296-309 divides by its rms) makes the cut visible (round 2 soak: 5e-4).  When a
// Normpower consumes the filter the planner asks for this scan instead: the full recurrence
//     s0[k+1] = M s0[k] + v[k],  M = A^L,
// in blocks of kXsBlock chunks -- block totals from zero state, a sequential scan of the totals with
// M^kXsBlock per channel, then the recurrence again inside every block from its true start state.
// Three small launches; every product keeps the relative accuracy of the decaying state.
template <int NS>
__device__ __forceinline__ void xs_step(const double* __restrict__ m, double (&s)[2 * NS], const double* __restrict__ add) {
    constexpr int D = 2 * NS;
    double t[D];
#pragma unroll
    for (int r = 0; r < D; ++r) {
        double a = add ? add[r] : 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) a = fma(m[r * D + d], s[d], a);
        t[r] = a;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = t[d];
}

// PHASE 0: T_b (zero-state end state of block b) -> sblk;  PHASE 2: s0 of every chunk of block b from S_b in sblk
template <int NS, int PHASE>
__global__ __launch_bounds__(kBlock) void k_sos_xs_block(const double* __restrict__ v, double* __restrict__ s0,
                                                         const double* __restrict__ mats, double* __restrict__ sblk,
                                                         SosGeom g, int nblk) {
    constexpr int D = 2 * NS;
    __shared__ double m1[D * D];
    if ((int)threadIdx.x < D * D) m1[threadIdx.x] = mats[threadIdx.x];
    __syncthreads();
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (tid >= (int64_t)nblk * g.nch) return;
    const int b = (int)(tid % nblk), ch = (int)(tid / nblk);
    const int k0 = b * kXsBlock, k1 = min(g.nchunks, k0 + kXsBlock);
    double s[D];
    double* sb = sblk + ((int64_t)ch * nblk + b) * 16;
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = PHASE == 2 ? sb[d] : 0.0;
    const double* vp = v + ((int64_t)ch * g.nchunks + k0) * D;
    double* sp = s0 + ((int64_t)ch * g.nchunks + k0) * D;
    for (int k = k0; k < k1; ++k, vp += D, sp += D) {
        if (PHASE == 2) {
#pragma unroll
            for (int d = 0; d < D; ++d) sp[d] = s[d];
            if (k + 1 == g.nchunks) break;  // (the last chunk's end state is never computed, nor needed)
        }
        if (PHASE == 0 && k + 1 == g.nchunks) break;  // (a channel's last chunk has no v: nothing follows it)
        xs_step<NS>(m1, s, vp);
    }
    if (PHASE == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) sb[d] = s[d];
    }
}

// S_(b+1) = MB S_b + T_b, sequentially per channel, in place (sblk: T_b in, S_b out)
template <int NS>
__global__ __launch_bounds__(64) void k_sos_xs_scan(const double* __restrict__ mats, double* __restrict__ sblk, int nch,
                                                    int nblk) {
    constexpr int D = 2 * NS;
    __shared__ double mb[D * D];
    for (int i = threadIdx.x; i < D * D; i += 64) mb[i] = mats[D * D + i];
    __syncthreads();
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= nch) return;
    double s[D], t[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = 0.0;
    double* sb = sblk + (int64_t)ch * nblk * 16;
    for (int b = 0; b < nblk; ++b, sb += 16) {
#pragma unroll
        for (int d = 0; d < D; ++d) t[d] = sb[d];  // T_b
#pragma unroll
        for (int d = 0; d < D; ++d) sb[d] = s[d];  // S_b
        xs_step<NS>(mb, s, t);
    }
}

template <int NS>
static void launch_xs_t(const double* v, double* s0, const double* mats, double* sblk, const SosGeom& g, hipStream_t st) {
    const int nblk = (g.nchunks + kXsBlock - 1) / kXsBlock;
    const int64_t nthr = (int64_t)nblk * g.nch;
    const dim3 grid((unsigned)((nthr + kBlock - 1) / kBlock));
    hipLaunchKernelGGL((k_sos_xs_block<NS, 0>), grid, dim3(kBlock), 0, st, v, s0, mats, sblk, g, nblk);
    hipLaunchKernelGGL((k_sos_xs_scan<NS>), dim3((unsigned)((g.nch + 63) / 64)), dim3(64), 0, st, mats, sblk, g.nch, nblk);
    hipLaunchKernelGGL((k_sos_xs_block<NS, 2>), grid, dim3(kBlock), 0, st, v, s0, mats, sblk, g, nblk);
}

int launch_sos_xscan(const double* v, double* s0, const double* mats, double* sblk, const SosGeom& g, int nsec, hipStream_t st) {
    if (g.nchunks <= 1) return 0;
    switch (nsec) {
    case 1: launch_xs_t<1>(v, s0, mats, sblk, g, st); break;
    case 2: launch_xs_t<2>(v, s0, mats, sblk, g, st); break;
    case 3: launch_xs_t<3>(v, s0, mats, sblk, g, st); break;
    case 4: launch_xs_t<4>(v, s0, mats, sblk, g, st); break;
    case 5: launch_xs_t<5>(v, s0, mats, sblk, g, st); break;
    case 6: launch_xs_t<6>(v, s0, mats, sblk, g, st); break;
    case 7: launch_xs_t<7>(v, s0, mats, sblk, g, st); break;
    default: launch_xs_t<8>(v, s0, mats, sblk, g, st); break;
    }
    return 3;
}


// ---------------------------------------------------------------------------
// K3t2: the tiled resampler for rates without a usable period (k_resample_tiled, k_resample.hip), two outputs per
// lane.  The one-output form reads two table values and ct inputs from LDS per tap for 2 ct multiply-adds and is
// bound by those reads (38 taps x 10 reads per output of 8 channels: 0.27 ms of LDS time in its 0.57 ms on a quarter
// of config 3 at x pi/3).  A lane that owns outputs m and m + 1 walks the union of their input windows once: an input
// (ct LDS reads) serves both outputs, each with the tap it has in THAT output's filter (four table reads), 12 reads
// for 4 ct multiply-adds.  Taps outside an output's window are read from a row of zeros on either side of the table
// (index clamped, not selected), so every output still adds its own taps oldest first: the same sums as the
// one-output form, bit for bit, on finite inputs (a NaN or Inf next to a window meets a zero weight: it reaches the
// one or two outputs beside those the reference puts it in).
// Measured (26 460 000 / 4 x 8 frames at x pi/3): 0.569 -> 0.487 ms, of which staging 0.14 ms, the result's stores
// 0.10 ms and the tap loop 0.26 ms (its multiply-adds alone: 0.11) -- the phases of the two workgroups of a CU do
// not overlap (they start together and stay in step).  A persistent form that prefetches the next tile into registers
// was built and is slower: 12 more doubles per thread push the 8-channel kernel from 108 to 155 VGPRs -- one
// workgroup per CU instead of two (0.59 ms) -- or, capped at 128, spill the prefetch itself (0.68 ms).
// (The matrix cores are no help here: v_mfma_f64_16x16x4 runs at the vector rate on this chip, 64 cycles, and a
//  banded weight matrix over 8 of 16 rows wastes four fifths of it -- built and measured: 0.65 ms.)
constexpr int kT2Threads = 512;  // eight waves share a staged tile (two workgroups per CU: four waves per SIMD)

// The staged inputs are kept as doubles, frame by frame: the CT channels of a frame side by side (16-byte reads with
// immediate offsets, no address arithmetic per channel) in rows of CT + 2 doubles -- with 16 bytes per lane a
// quarter wave reads at a time, and 80-byte (48-byte) rows put its 16 lanes on 16 different bank groups.
template <int CT>
struct T2Row {
    static constexpr int pitch = CT >= 2 ? CT + 2 : 1;
};

template <typename T, int CT>
__global__ __launch_bounds__(kT2Threads) void k_resample_tiled2(const T* __restrict__ x, T* __restrict__ y,
                                                                const double* __restrict__ pfbt,
                                                                const double* __restrict__ dpfbt, RsTiled g) {
    constexpr int FP = T2Row<CT>::pitch;
    extern __shared__ double lds_raw[];
    const int taps = g.g.taps, nphi = g.g.nphi;
    // tables as [tap + 1][phase], a row of zeros before tap 0 and one after the last
    double* const tp = lds_raw;
    double* const td = tp + (size_t)(taps + 2) * nphi;
    double* const xs = td + (size_t)(taps + 2) * nphi;
    const int tid = threadIdx.x;
    const bool arb = g.g.arbitrary != 0;
    for (int i0 = tid; i0 < (taps + 2) * nphi; i0 += 4 * kT2Threads) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kT2Threads - nphi;  // row -1 and row taps: zeros
            const bool real = i >= 0 && i < taps * nphi;
            a[u] = real ? pfbt[i] : 0.0;
            b[u] = (real && arb) ? dpfbt[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kT2Threads;
            if (i < (taps + 2) * nphi) {
                tp[i] = a[u];
                td[i] = b[u];
            }
        }
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
    constexpr int kLeft = 8;  // frames staged before the first window: the union window of a pair starts up to
                              // (wave maximum of j_B - j_A) - (its own) frames before its first output's
    const int64_t xlo = j0 - (taps - 1) - kLeft;  // global input frame of LDS frame 0
    const int nfr = (int)(j1 - xlo + 1);          // <= tile_in by the planner's choice of tile_out
    // staging: eight loads in flight per thread (consecutive lanes read consecutive frames of a channel)
    {
        const int total = CT * nfr;
        for (int e0 = tid; e0 < total; e0 += 8 * kT2Threads) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * kT2Threads;
                const int c = e / nfr, i = e - c * nfr;
                const int64_t n = xlo + i;
                v[u] = (e < total && n >= 0 && n < g.g.n_in) ? (double)x[(int64_t)(c0 + c) * g.g.in_pitch + n] : 0.0;  // Pad(x.signal, zero), src/filters.jl:240
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * kT2Threads;
                const int c = e / nfr, i = e - c * nfr;
                if (e < total) xs[i * FP + c] = v[u];
            }
        }
    }
    __syncthreads();
    const int tdo = (taps + 2) * nphi;
    // (whole waves stay in the loop -- the wave maximum below reads every lane --; lanes past the tile's end work on
    //  its last output and store nothing)
    for (int64_t mw = m0 + 2 * (tid & ~63); mw < m1; mw += 2 * kT2Threads) {
        const int64_t ma_ = mw + 2 * (tid & 63);
        const bool one = ma_ < m1, two = ma_ + 1 < m1;
        const int64_t ma = one ? ma_ : m1 - 1;
        int64_t jA, jB;
        int pA, pB;
        double alA, alB;
        rs_pos(g.g, g.g.m0 + ma, jA, pA, alA);
        rs_pos(g.g, g.g.m0 + (two ? ma + 1 : ma), jB, pB, alB);
        const int d = (int)(jB - jA);  // 0, 1, 2 ... : how much later output B's window ends
        int dmax = d;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, __shfl_xor(dmax, off, 64));
        const double* const tA = tp + nphi + pA;  // tA[k * nphi]: tap k of output A's phase, k = -1 .. taps
        const double* const tB = tp + nphi + pB;
        double loA[CT], hiA[CT], loB[CT], hiB[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) loA[c] = hiA[c] = loB[c] = hiB[c] = 0.0;
        // q = distance from the pair's newest input, oldest first; tap of that input: q in B's filter, q - d in A's
        const int q0 = taps - 1 + dmax;
        const double* __restrict__ xf = xs + ((int)(jB - xlo) - q0) * FP;  // the oldest frame of the union window
        for (int q = q0; q >= 0; --q, xf += FP) {
            const int kA = min(max(q - d, -1), taps) * nphi, kB = min(q, taps) * nphi;
            const double fA = tA[kA], fB = tB[kB];
            const double gA = arb ? tA[kA + tdo] : 0.0, gB = arb ? tB[kB + tdo] : 0.0;
            double xv[CT];
            if constexpr (CT >= 2) {
#pragma unroll
                for (int c = 0; c < CT; c += 2) {
                    const double2 w2 = *reinterpret_cast<const double2*>(xf + c);
                    xv[c] = w2.x;
                    xv[c + 1] = w2.y;
                }
            } else
                xv[0] = xf[0];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                loA[c] += fA * xv[c];
                hiA[c] += gA * xv[c];
                loB[c] += fB * xv[c];
                hiB[c] += gB * xv[c];
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const double rA = arb ? loA[c] + hiA[c] * alA : loA[c];
            const double rB = arb ? loB[c] + hiB[c] * alB : loB[c];
            T* o = y + (int64_t)(c0 + c) * g.g.out_pitch + ma;
            if (one) o[0] = (T)rA;
            if (two) o[1] = (T)rB;
        }
    }
}

template <typename T>
static void launch_resample_tiled2_t(const void* x, void* y, const double* pfbt, const double* dpfbt, const RsTiled& g,
                                     hipStream_t st) {
    const size_t ldsb = ((size_t)2 * (g.g.taps + 2) * g.g.nphi * 8 + (size_t)(g.ct >= 2 ? g.ct + 2 : 1) * g.tile_in * 8 + 15) / 16 * 16;
    const unsigned grid = (unsigned)(g.ntiles * (g.g.nch / g.ct));
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 ? 0 : (dev > 63 ? 63 : dev);
#define SO_RT2(CTV)                                                                                                       \
    {                                                                                                                     \
        static bool seen[64];                                                                                             \
        if (!seen[dev]) { /* (per device: a process may drive several GPUs) */                                            \
            (void)hipFuncSetAttribute((const void*)k_resample_tiled2<T, CTV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                        \
            seen[dev] = true;                                                                                             \
        }                                                                                                                 \
        hipLaunchKernelGGL((k_resample_tiled2<T, CTV>), dim3(grid), dim3(kT2Threads), ldsb, st, (const T*)x, (T*)y, pfbt, \
                           dpfbt, g);                                                                                     \
    }
    switch (g.ct) {
    case 8: SO_RT2(8) break;
    case 4: SO_RT2(4) break;
    case 2: SO_RT2(2) break;
    default: SO_RT2(1) break;
    }
#undef SO_RT2
}

void launch_resample_tiled2(const void* x, void* y, const double* pfbt, const double* dpfbt, const RsTiled& g, hipStream_t st) {
    if (g.g.n_out <= 0) return;
    if (g.g.in_dtype == SO_F32) launch_resample_tiled2_t<float>(x, y, pfbt, dpfbt, g, st);
    else launch_resample_tiled2_t<double>(x, y, pfbt, dpfbt, g, st);
}

}  // namespace so
