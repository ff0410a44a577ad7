// Hand-written HIP kernel for gfx950 (MI355X, CDNA4; wave64).  No CUDA shims, no dual paths.
// k_rsos_fixup: the launch behind k_rsos (the fused resampler + IIR, and the plain IIR in one pass) that makes the set of
// a channel's non-finite outputs the REFERENCE's.
//
// Reference: the IIR filters sample by sample (src/filters.jl:252-255 -> DSP.jl filt!, DF2T): its output is non-finite from
// the FIRST sample whose resampled value is non-finite, and that value is non-finite exactly where one of the taps-per-phase
// input samples the polyphase kernel multiplies is (its own zero padding included: 0 * NaN is NaN).  k_rsos runs the
// recurrence in blocks of 16 outputs on the matrix cores -- a block that holds a non-finite value is non-finite from its
// first output on -- and resamples a group of 16 outputs from one window that is wider than any single output's: its set is
// a superset, up to a block and a few outputs early.  Rounds 4 and 5 stated and pinned that superset; this kernel removes
// it.  The chain wave of k_rsos notes, per channel, the first time range whose walk ended in a non-finite state
// (RsSos::bad).  Here, per such channel (there is none in a launch over finite data: every workgroup reads one word and
// returns): the first non-finite stored output of that range is found, the outputs of its block -- and of the blocks
// behind it as far as the reference is still finite -- are recomputed the reference's way (each output's own taps; the
// cascade as the per-sample DF2T recurrence, from rest a warm-up ahead of the block, as every range of k_rsos starts), and
// written over the block's NaNs up to the reference's first non-finite output.  Everything behind the range becomes NaN
// as before (k_sos_poison's part, done by the same launch).
#include "kcommon.h"
#include "kstage.h"

namespace so {

constexpr int kFixThreads = 256;
constexpr int kFixSeg = 1024;   // outputs per segment of the recomputation
constexpr int kFixIn = 6144;    // input frames staged per segment (a segment is shortened where it would need more)
constexpr int kFixAhead = 64;   // outputs behind the block's first that are looked at

template <typename TO>
__device__ __forceinline__ void rsos_fixup_body(const RsFixup& fx) {
    const RsSos& g = fx.g;
    const int ch = blockIdx.y;
    if (ch >= g.nch) return;
    const int r = g.bad[ch];
    if (r < 0 || r >= g.nranges) return;  // (the usual case: the large initial value)
    TO* const y = (TO*)fx.y + (int64_t)ch * g.out_pitch;
    const int64_t prL = g.pr * g.L;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    // ---- behind the range: NaN (the next range started from rest and would be finite again) ----
    {
        int64_t f0 = (int64_t)(r + 1) * prL;
        if (f0 < g.store_lo) f0 = g.store_lo;
        for (int64_t f = f0 + (int64_t)blockIdx.x * kFixThreads + threadIdx.x; f < g.n_out; f += (int64_t)gridDim.x * kFixThreads) y[f] = (TO)nan;
    }
    if (blockIdx.x != 0) return;
    // ---- inside the range ----
    __shared__ long long s_first;
    __shared__ double xin[kFixIn];
    __shared__ double xs[kFixSeg];
    __shared__ double ys[kFixAhead + 16];
    const int tid = threadIdx.x;
    const int64_t lo = max((int64_t)r * prL, g.store_lo), hi = min((int64_t)(r + 1) * prL, g.n_out);
    if (tid == 0) s_first = hi;
    __syncthreads();
    for (int64_t m = lo + tid; m < hi; m += kFixThreads)
        if (!isfinite((double)y[m])) {
            atomicMin(&s_first, (long long)m);
            break;
        }
    __syncthreads();
    const int64_t m0 = s_first;
    if (m0 >= hi) return;
    const int64_t L = g.L, M = g.M;
    const int kw = 4 * g.ks, taps = fx.taps;
    const int64_t bs = m0 & ~(int64_t)15;                         // the block (ranges begin at multiples of L, L % 16 == 0)
    const int64_t ws = max((int64_t)0, bs - (int64_t)g.wp * L);   // from rest, a warm-up ahead
    const int64_t me = min(bs + kFixAhead, g.n_out);
    const int ncar = fx.gsrc.ctl->ncar;
    auto newest = [&](int64_t m) {  // newest input frame of output m
        const int64_t P = m / L, rr = m % L;
        return P * M + fx.jend[rr / 16] + fx.jrel[rr];
    };
    double s1[kMaxSec], s2[kMaxSec];
#pragma unroll
    for (int k = 0; k < kMaxSec; ++k) s1[k] = s2[k] = 0.0;
    __shared__ long long s_bad;
    if (tid == 0) s_bad = -1;
    __syncthreads();
    for (int64_t seg = ws; seg < me;) {
        // a segment of outputs whose inputs fit the staging buffer
        int64_t se = min(seg + kFixSeg, me);
        const int64_t jlo = newest(seg) - (taps - 1);
        while (se > seg + 1 && newest(se - 1) - jlo + 1 > kFixIn) se = seg + (se - seg) / 2;
        const int64_t nin = newest(se - 1) - jlo + 1;
        if (nin > kFixIn) return;  // (one output's own window does not fit: not a geometry k_rsos runs)
        // ---- the source's frames [jlo, jlo + nin): two per call, zero outside the signal ----
        for (int t = tid; 2 * t < nin; t += kFixThreads)
            stage_generic_impl<double, 1, true>(g.n_in, 0, fx.gsrc.car, ncar, fx.gsrc.ops, fx.gsrc.leaves, jlo + 2 * t, t, 0, ch, xin);
        __syncthreads();
        // ---- resample: each output from its own taps (oldest first, one accumulator: the reference's dot product) ----
        for (int64_t m = seg + tid; m < se; m += kFixThreads) {
            const int64_t P = m / L, rr = m % L;
            const int gi = (int)(rr / 16), t16 = (int)(rr % 16);
            const int jr = fx.jrel[rr];
            const int64_t jm = P * M + fx.jend[gi] + jr;
            double acc = 0.0;
            for (int a = taps - 1; a >= 0; --a) {
                const int kk = kw - 1 + jr - a;  // the tap's slot in the group's window (outside it: a tap the table does not hold, 0)
                const double tap = kk >= 0 && kk < kw ? fx.tab[((size_t)gi * kw + kk) * 16 + t16] : 0.0;
                acc = fma(tap, xin[jm - a - jlo], acc);
            }
            if (g.x32) acc = (double)(float)acc;  // (a Float32 signal: the resampler hands the filter Float32 samples)
            xs[m - seg] = acc;
        }
        __syncthreads();
        // ---- the cascade, sample by sample (DF2T, src/filters.jl:252-255 -> DSP.jl filt!) ----
        if (tid == 0) {
            const SosCoefs& cf = fx.cf;
            for (int64_t m = seg; m < se; ++m) {
                double v = xs[m - seg];
#pragma unroll
                for (int k = 0; k < kMaxSec; ++k)
                    if (k < cf.nsec) {
                        const double xi = v;
                        v = s1[k] + cf.b0[k] * xi;
                        s1[k] = s2[k] + cf.b1[k] * xi - cf.a1[k] * v;
                        s2[k] = cf.b2[k] * xi - cf.a2[k] * v;
                    }
                v *= cf.gain;
                if (m >= bs) {
                    if (!isfinite(v) || (sizeof(TO) == 4 && !isfinite((double)(float)v))) {
                        s_bad = m;
                        break;
                    }
                    ys[m - bs] = v;
                }
            }
        }
        __syncthreads();
        if (s_bad >= 0) break;
        seg = se;
    }
    const int64_t mb = s_bad;
    if (mb < 0) return;  // (nothing non-finite the reference's way within reach: left as the kernel wrote it)
    // ---- the reference is finite up to mb: its values over the block's NaNs ----
    for (int64_t m = max(bs, lo) + tid; m < mb; m += kFixThreads) y[m] = (TO)ys[m - bs];
}

template <typename TO>
__global__ __launch_bounds__(kFixThreads) void k_rsos_fixup(RsFixup fx) {
    rsos_fixup_body<TO>(fx);
}
// ... behind a batched launch: blockIdx.z = the member (its own words say whether there is anything to do)
template <typename TO>
__global__ __launch_bounds__(kFixThreads) void k_rsos_fixup_batch(const RsFixup* __restrict__ items) {
    const RsFixup* fx = items + blockIdx.z;
    const int ch = blockIdx.y;
    if (ch >= fx->g.nch) return;
    const int r = fx->g.bad[ch];
    if (r < 0 || r >= fx->g.nranges) return;  // (the usual case: nothing is copied from the table)
    const RsFixup F = *fx;
    rsos_fixup_body<TO>(F);
}

// k_rs_fixup: the same for the periodic resampler ALONE (k_resample_periodic, no filter behind it).  Its compute waves list the
// (tile, group)s whose accumulators held a non-finite value (RsPeriodic::nf); ONE workgroup -- the list is empty in a launch
// over finite data: it reads a word and returns -- recomputes every listed group's rows x 16 outputs from each output's own
// taps (the reference's dot product: src/filters.jl:252-255 -> DSP.jl's polyphase kernels, 0 * NaN included) and stores
// them, finite or not: the group's NaNs that were only the window's become the reference's values.  It leaves the list empty
// for the next launch.  (Before k_resample_fix: the outputs DSP.jl's accumulator places differently keep their own taps.)
constexpr int kRsFixRows = 32, kRsFixKw = 160;
template <typename TO>
__global__ __launch_bounds__(kFixThreads) void k_rs_fixup(RsPerFixup fx) {
    const RsPeriodic& g = fx.g;
    const uint32_t count = g.nf[0];
    if (count == 0) return;
    __shared__ double win[kRsFixRows][kRsFixKw + 2];
    const int tid = threadIdx.x;
    const int rows = g.rows, kw = g.kw, taps = fx.taps;
    const int ptmask = g.pt - 1, ptshift = g.ptshift;
    const int ncar = fx.gsrc.ctl->ncar;
    const uint32_t n = count < kRsNfCap ? count : kRsNfCap;
    if (rows <= kRsFixRows && kw <= kRsFixKw) {
        for (uint32_t e = 0; e < n; ++e) {
            const uint32_t* en = g.nf + 4 + 4 * (size_t)e;
            const int64_t P0 = (int64_t)(((uint64_t)en[1] << 32) | en[0]);
            const int c0 = (int)en[2], gi = (int)en[3];
            const int je = fx.jend[gi];
            // ---- every row's window of the group: source frames [P M + je - (kw - 1), P M + je], two per call ----
            __syncthreads();
            for (int t = tid; t < rows * (kw / 2); t += kFixThreads) {
                const int rho = t / (kw / 2), h = t - rho * (kw / 2);
                const int64_t P = P0 + (rho & ptmask);
                stage_generic_impl<double, 1, true>(g.n_in, 0, fx.gsrc.car, ncar, fx.gsrc.ops, fx.gsrc.leaves, P * g.M + je - (kw - 1) + 2 * h, h, 0,
                                              c0 + (rho >> ptshift), &win[rho][0]);
            }
            __syncthreads();
            // ---- outputs ----
            for (int t = tid; t < rows * 16; t += kFixThreads) {
                const int rho = t >> 4, n16 = t & 15;
                const int64_t P = P0 + (rho & ptmask), r = (int64_t)gi * 16 + n16, m = P * g.L + r;
                const int ch = c0 + (rho >> ptshift);
                if (P >= g.nperiods || r >= g.L || m >= g.n_out || ch >= g.nch) continue;
                const int jr = fx.jrel[r];
                double acc = 0.0;
                for (int a = taps - 1; a >= 0; --a) {
                    const int kk = kw - 1 + jr - a;
                    if (kk < 0 || kk >= kw) continue;  // (a tap the table does not hold: zero, and outside the staged window)
                    acc = fma(fx.tab[((size_t)gi * kw + kk) * 16 + n16], win[rho][kk], acc);
                }
                ((TO*)fx.y)[(int64_t)ch * g.out_pitch + m] = (TO)acc;
            }
        }
    }
    __syncthreads();
    if (tid == 0) g.nf[0] = 0;
}

int launch_rs_fixup(const RsPerFixup& fx, hipStream_t st) {
    if (fx.g.nf == nullptr) return 0;
    if (fx.out_f32) hipLaunchKernelGGL((k_rs_fixup<float>), dim3(1), dim3(kFixThreads), 0, st, fx);
    else hipLaunchKernelGGL((k_rs_fixup<double>), dim3(1), dim3(kFixThreads), 0, st, fx);
    return 1;
}

int launch_rsos_fixup_batch(const RsFixup* items, int nitems, int nch, int out_f32, hipStream_t st) {
    if (nitems <= 0) return 0;
    const dim3 grid(32, (unsigned)nch, (unsigned)nitems);
    if (out_f32) hipLaunchKernelGGL((k_rsos_fixup_batch<float>), grid, dim3(kFixThreads), 0, st, items);
    else hipLaunchKernelGGL((k_rsos_fixup_batch<double>), grid, dim3(kFixThreads), 0, st, items);
    return 1;
}

int launch_rsos_fixup(const RsFixup& fx, hipStream_t st) {
    if (fx.g.bad == nullptr || fx.g.nranges < 1) return 0;
    const dim3 grid(32, (unsigned)fx.g.nch);
    if (fx.g.out_f32) hipLaunchKernelGGL((k_rsos_fixup<float>), grid, dim3(kFixThreads), 0, st, fx);
    else hipLaunchKernelGGL((k_rsos_fixup<double>), grid, dim3(kFixThreads), 0, st, fx);
    return 1;
}

}  // namespace so
