// Hand-written HIP kernels for gfx950 (MI355X, CDNA4; wave64).  No CUDA shims, no dual paths.
// K2 k_sos_*: second-order-sections IIR, time-parallel by exact chunked state propagation
// (reference src/filters.jl:252-255 + DSP.jl DF2T)
#include <type_traits>

#include "kcommon.h"

namespace so {

// ---------------------------------------------------------------------------
// K2: SOS IIR.  The recurrence is linear, so a chunk's output is the zero-state
// response to its own samples plus the zero-input response to the state at its start.
//   pass 1  k_sos_tiled<.,.,false> : v_k = state at the END of chunk k from zero state (only the
//           last min(L,W) frames matter: older frames have decayed below 2^-70)
//   pass 2  k_sos_scan  : s0_k = sum_{j=1..K} M^(j-1) v_{k-j},  M = A^L (host-computed
//           powers of the cascade's state matrix); K terms until ||M^K|| < 2^-70
//   pass 3  k_sos_tiled<.,.,true>  : run DF2T on chunk k from s0_k and write the output
// DF2T per section (DSP.jl filt!):
//   y = s1 + b0 x ; s1 = s2 + b1 x - a1 y ; s2 = b2 x - a2 y ; out = y*g after the cascade
template <int NS>
__device__ __forceinline__ double sos_step(double x, double (&s)[2 * NS], const SosCoefs& cf) {
    double y = x;
#pragma unroll
    for (int f = 0; f < NS; ++f) {
        const double xi = y;
        y = s[2 * f] + cf.b0[f] * xi;
        s[2 * f] = s[2 * f + 1] + cf.b1[f] * xi - cf.a1[f] * y;
        s[2 * f + 1] = cf.b2[f] * xi - cf.a2[f] * y;
    }
    return y;
}

// Tiled streaming of 64 sequences per wave (sequence = one chunk of one channel).  A tile is
// 64 rows x kTT frames: the wave loads it with coalesced 128-byte row segments (4 rows per
// load instruction), parks it in LDS with an odd row pitch, then every lane walks ITS row
// (conflict-free, stride kTT+1) through the DF2T cascade with the section states in registers
// and, for the apply pass, writes the outputs back through LDS the same coalesced way.
// APPLY == false: pass 1 (final state of the last min(L,W) frames from zero state)
// APPLY == true : pass 3 (outputs from the propagated initial state)
constexpr int kTT = 16;

// One tile step of a lane's row through the cascade: kTT frames from LDS, outputs back in place (APPLY).
// HEAD: the row's first `head` columns lie before its chunk and are left alone (see sos_tiled_body).
template <int NS, bool APPLY, bool HEAD, int TT, typename E>
__device__ __forceinline__ void sos_row_steps(E* __restrict__ row, double (&s)[2 * NS], const SosGeom& g, const SosCoefs& cf,
                                              double& gs, double& gc, const int head) {
    if (g.src_op == 0) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if (HEAD && t < head) continue;
            const double yv = sos_step<NS>((double)row[t], s, cf);
            if (APPLY) row[t] = (E)(yv * cf.gain);
            // (frames past a short row's end are zeros and never stored; their effect on
            //  the state is irrelevant: only full chunks feed pass 1)
        }
    } else {
        // fused source (SosGeom::src_op): the array sample plus / times a sine generator, the value
        // of `Mix(Signal(sin), x)` / `Amplify(x, Signal(sin))` (reference src/mapsignal.jl:249-272,
        // src/functions.jl:57-60) formed here instead of by a K1 pass through HBM.  The lane walks
        // its row in time, so the sine advances by one rotation per frame from the exact value at
        // the row's first frame (gs, gc).
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if (HEAD && t < head) continue;
            const double xin = g.src_op == 1 ? (double)row[t] + gs : (double)row[t] * gs;
            const double ns_ = fma(gs, g.src_cd, gc * g.src_sd), nc_ = fma(gc, g.src_cd, -(gs * g.src_sd));
            gs = ns_;
            gc = nc_;
            const double yv = sos_step<NS>(xin, s, cf);
            if (APPLY) row[t] = (E)(yv * cf.gain);
        }
    }
}

// (`blk`: the block's index among the blocks of THIS filter -- blockIdx.x for a launch of its own, the offset
//  into its share of a batched launch, k_sos_tiled_batch below)
template <int NS, typename T, bool APPLY>
__device__ __forceinline__ void sos_tiled_body(const T* __restrict__ x, T* __restrict__ y,
                                               const double* __restrict__ s0, double* __restrict__ v,
                                               const SosGeom& g, const SosCoefs& cf, const int64_t blk) {
    // Float32 signals: steps of 32 frames, two per lane and load / store instruction -- 128 bytes per row segment as for
    // Float64 (with 16-frame steps a Float32 row segment was 64 bytes: twice the memory requests per byte, and the
    // Float32 filter took 1.0 ms where the Float64 one took 0.55: tools/channel_matrix.py) -- in a tile of floats (the
    // cascade computes in Float64 either way; a value is rounded once, where it used to be rounded by the store)
    constexpr int TT = sizeof(T) == 4 ? 2 * kTT : kTT;
    constexpr int VE = TT / 16;  // elements per lane and row of a load / store instruction
    typedef typename std::conditional<sizeof(T) == 4, float, double>::type E;
    __shared__ E tile[kBlock / 64][64 * (TT + 1)];
    __shared__ int64_t rowbase[kBlock / 64][64];  // frame of column 0 of the row's first tile step
    __shared__ int64_t rowmin[kBlock / 64][64];   // first frame the row stores
    __shared__ int rowlen[kBlock / 64][64];       // columns the row spans (head + frames to process)
    __shared__ int rowhead[kBlock / 64][64];      // its first column inside the chunk
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nck = APPLY ? g.nchunks : g.nchunks - 1;  // pass 1 skips every channel's last chunk
    const int64_t nseq = (int64_t)nck * g.nch;
    const int64_t seq = (blk * (kBlock / 64) + w) * 64 + lane;
    const bool live = seq < nseq;
    const int k = live ? (int)(seq % nck) : 0;
    const int ch = live ? (int)(seq / nck) : 0;
    int64_t beg = (int64_t)k * g.chunk;
    int64_t end = beg + g.chunk < g.n ? beg + g.chunk : g.n;
    // pass 1: the last min(L, W) frames of the chunk (full chunks only: end = beg + L), W rounded up to whole tile
    // steps -- a row shorter than a step would be followed by zeros INSIDE the step, and the state keeps turning over
    // them: a filter that forgets within 8 frames (W = 1 ... 8; a first-order Butterworth at fs/4 has its pole at 0)
    // handed on the state of 16 - W frames later, i.e. nothing (tools/soak_kernels.py seed 10102: 7 % off)
    if (!APPLY) {
        const int64_t w16 = (g.warm + (TT - 1)) / TT * TT;
        beg = end - (w16 < g.chunk ? w16 : g.chunk);
    }
    // Pass 3 stores whole cache lines whatever the alignment of the result's channel rows (SosGeom::align_rows): the
    // tile steps of a row start `head` frames BEFORE its chunk, where the result's 128-byte line starts (64-byte half
    // line of a Float32 result: a step is 16 frames wide); those columns are neither filtered nor stored, the chunk
    // borders -- and with them every rounding -- stay where they are.  A row off its lines writes every line in two
    // halves, 64 rows x 12 waves x 256 CUs at a time, more than L2 holds until the second half arrives: config 5,
    // rows of 3 628 118 frames (what an n x 128 Array has): pass 3 2.6 ms against 1.65 ms with aligned rows.
    int head = 0;
    if (APPLY && g.align_rows && live)
        head = (int)(((uintptr_t)y / (g.out_dtype == SO_F32 ? 4 : 8) + (uint64_t)ch * (uint64_t)g.out_pitch + (uint64_t)beg) % TT);
    const int len = live ? head + (int)(end - beg) : 0;
    rowbase[w][lane] = beg - head;  // (the row's channel is kept in rowch)
    rowmin[w][lane] = beg > g.store_lo ? beg : g.store_lo;
    rowlen[w][lane] = len;
    rowhead[w][lane] = head;
    double s[2 * NS];
#pragma unroll
    for (int d = 0; d < 2 * NS; ++d) s[d] = 0.0;
    if (APPLY && live && k > 0 && s0 != nullptr) {
        const double* sp = s0 + ((int64_t)ch * g.nchunks + k) * (2 * NS);
#pragma unroll
        for (int d = 0; d < 2 * NS; ++d) s[d] = sp[d];
    }
    // every row of this wave belongs to a (chunk, channel); rows are consecutive chunks of one
    // channel except where the wave straddles a channel boundary, so the per-row channel is
    // kept alongside the frame offset
    __shared__ int rowch[kBlock / 64][64];
    rowch[w][lane] = ch;
    __builtin_amdgcn_wave_barrier();
    int maxlen = len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
    const bool anyhead = APPLY && __any(head != 0);
    E* tl = tile[w];
    const int rsub = lane >> 4, col = (lane & 15) * VE;  // load/store role: 4 rows x 16 lanes x VE columns
    // The passes are latency-bound (a few waves per CU, each a chain of tile round trips): all 16
    // loads of a tile are issued together and the NEXT tile's loads are issued before this tile's
    // arithmetic, so a wave always has one tile (8 KB) in flight.
    const T* xrow[16];  // this lane's 16 load rows: element address of (row, column col)
    int xlen[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = j * 4 + rsub;
        xlen[j] = rowlen[w][r];
        xrow[j] = x + ((int64_t)rowch[w][r] * g.in_pitch + rowbase[w][r] + col);
    }
    E xv[16][VE];
    // (lo: first column of the row this lane may read; hi: one past the last)
    auto load_cols = [&](int j, int t0, int lo) __attribute__((always_inline)) {
        if constexpr (VE == 2) {
            if (t0 + col >= lo && t0 + col + 1 < xlen[j]) {  // both columns: one 8-byte load (any 4-byte address)
                typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
                const f2 w2 = *reinterpret_cast<const f2*>(xrow[j] + t0);
                xv[j][0] = w2[0];
                xv[j][1] = w2[1];
                return;
            }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) xv[j][e] = (t0 + col + e < xlen[j] && t0 + col + e >= lo) ? (E)xrow[j][t0 + e] : (E)0;
    };
#pragma unroll
    for (int j = 0; j < 16; ++j)  // (head columns hold frames of the chunk before, or of nothing: not this row's to read)
        load_cols(j, 0, APPLY ? rowhead[w][j * 4 + rsub] : 0);
    // fused sine source: (sin, cos) of the generator's phase at this lane's first frame, evaluated like
    // func_eval (every operation rounded on its own, first frame t = 1/fs)
    double gs = 0.0, gc = 1.0;
    if (g.src_op != 0) {
        const double tt = __ddiv_rn((double)(beg + g.src_df + 1), g.src_fs);
        const double ph = g.src_has_omega ? __dadd_rn(__dmul_rn(tt, g.src_omega), g.src_phi) : __dadd_rn(tt, g.src_phi);
        sincospi_c(2.0 * ph, gs, gc);
    }
    for (int t0 = 0; t0 < maxlen; t0 += TT) {
        // ---- tile t0 (loaded one iteration ago: 16 instructions x (4 rows x 128 B)) -> LDS ----
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int e = 0; e < VE; ++e) tl[(j * 4 + rsub) * (TT + 1) + col + e] = xv[j][e];
        __builtin_amdgcn_wave_barrier();
        if (t0 + TT < maxlen) {
#pragma unroll
            for (int j = 0; j < 16; ++j) load_cols(j, t0 + TT, 0);
        }
        // ---- every lane: its own row through the cascade ----
        E* row = tl + lane * (TT + 1);
        if (t0 == 0 && anyhead) sos_row_steps<NS, APPLY, true, TT, E>(row, s, g, cf, gs, gc, head);
        else sos_row_steps<NS, APPLY, false, TT, E>(row, s, g, cf, gs, gc, 0);
        __builtin_amdgcn_wave_barrier();
        if (APPLY) {
            // ---- coalesced store ----
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int r = j * 4 + rsub;
                const int64_t f0 = rowbase[w][r] + t0 + col;  // frame of this lane's first column
                const int64_t o = (int64_t)rowch[w][r] * g.out_pitch + f0;
                if constexpr (VE == 2) {
                    if (t0 + col + 1 < rowlen[w][r] && f0 >= rowmin[w][r]) {  // both columns: one 8-byte store
                        typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
                        f2 w2;
                        w2[0] = tl[r * (TT + 1) + col];
                        w2[1] = tl[r * (TT + 1) + col + 1];
                        *reinterpret_cast<f2*>(y + o) = w2;
                        continue;
                    }
                }
#pragma unroll
                for (int e = 0; e < VE; ++e)
                    if (t0 + col + e < rowlen[w][r] && f0 + e >= rowmin[w][r]) {
                        // (a Float64 filter writing a Float32 result itself: `convert` on store, src/sink.jl:262-266)
                        if (sizeof(T) == 8 && g.out_dtype == SO_F32) reinterpret_cast<float*>(y)[o + e] = (float)tl[r * (TT + 1) + col + e];
                        else y[o + e] = (T)tl[r * (TT + 1) + col + e];
                    }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!APPLY && live) {
        double* vp = v + ((int64_t)ch * g.nchunks + k) * (2 * NS);
#pragma unroll
        for (int d = 0; d < 2 * NS; ++d) vp[d] = s[d];
    }
    if (APPLY && live && g.bad != nullptr) {  // a non-finite state at the end of the row: everything behind it is NaN
        bool nf = false;
#pragma unroll
        for (int d = 0; d < 2 * NS; ++d) nf = nf || !(fabs(s[d]) <= 1.7976931348623157e308);
        if (nf) atomicMin(g.bad + ch, k);
    }
}

// (k_fill_u32 and k_sos_poison -- the two small kernels every filtered sink launches -- live in k_small.hip: a code
//  object of their own, so that a first sink that only needs them does not load this file's 2.7 MB)

template <int NS, typename T, bool APPLY>
__global__ __launch_bounds__(kBlock) void k_sos_tiled(const T* __restrict__ x, T* __restrict__ y,
                                                      const double* __restrict__ s0,
                                                      double* __restrict__ v, SosGeom g,
                                                      SosCoefs cf) {
    sos_tiled_body<NS, T, APPLY>(x, y, s0, v, g, cf, (int64_t)blockIdx.x);
}

// mpow: [kterms][D][D] row-major powers of M = A^L built on the host (mpow[0] = I, mpow[1] = M);
// the host uses them to choose the truncation K, the kernel only needs M itself.
template <int NS>
__device__ __forceinline__ void sos_scan_body(const double* __restrict__ v, const double* __restrict__ mpow,
                                              const SosGeom& g, double* __restrict__ s0, const int64_t blk) {
    constexpr int D = 2 * NS;
    // Horner form of  s0_k = sum_{j=1..K} M^(j-1) v_(k-j):  s <- M s + v_(k-j), oldest term first.
    // One matrix (M = A^L, second entry of the host's power table) in LDS, read as broadcasts; no
    // barrier and no matrix fetch per term -- the per-term global round trips of the previous
    // power-table form made this the longest of the three passes.
    __shared__ double m1[D * D];
    if ((int)threadIdx.x < D * D) m1[threadIdx.x] = g.kterms >= 2 ? mpow[(int64_t)D * D + threadIdx.x] : 0.0;
    __syncthreads();
    const int64_t tid = blk * kBlock + threadIdx.x;
    const int64_t nseq = (int64_t)g.nchunks * g.nch;
    const bool live = tid < nseq;
    const int k = live ? (int)(tid % g.nchunks) : 0;
    const int ch = live ? (int)(tid / g.nchunks) : 0;
    double acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.0;
    const int jmax = live ? (k < g.kterms ? k : g.kterms) : 0;
    const double* vp = v + ((int64_t)ch * g.nchunks + (k - jmax)) * D;  // oldest term, then forward
    for (int j = jmax; j >= 1; --j, vp += D) {
        double vv[D], t[D];
#pragma unroll
        for (int d = 0; d < D; ++d) vv[d] = vp[d];
#pragma unroll
        for (int r = 0; r < D; ++r) {
            double a = vv[r];
#pragma unroll
            for (int d = 0; d < D; ++d) a = fma(m1[r * D + d], acc[d], a);
            t[r] = a;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = t[d];
    }
    if (live) {
        double* sp = s0 + ((int64_t)ch * g.nchunks + k) * D;
#pragma unroll
        for (int d = 0; d < D; ++d) sp[d] = acc[d];
    }
}

template <int NS>
__global__ __launch_bounds__(kBlock) void k_sos_scan(const double* __restrict__ v,
                                                     const double* __restrict__ mpow, SosGeom g,
                                                     double* __restrict__ s0) {
    sos_scan_body<NS>(v, mpow, g, s0, (int64_t)blockIdx.x);
}

// ---------------------------------------------------------------------------
// Batched launches: many independent filters of one shape (the scenes under an `Append`, reference
// src/appending.jl:59-76: every child is its own signal with its own filter state) share ONE launch per
// pass.  A filter whose launch would not fill the chip -- a few hundred workgroups that start and drain
// together -- otherwise pays that ramp three times per scene.  desc[m] describes filter m; `first[PASS]`
// is the index of its first workgroup in the batched grid (ascending, desc[n] holds the totals).
typedef const SosDesc __attribute__((address_space(4))) * SosDescPtr;  // constant address space: scalar loads

// (a struct out of the constant address space, eight bytes at a time; what the kernel never reads is never loaded)
template <typename S>
__device__ __forceinline__ void load_const(S& dst, const S __attribute__((address_space(4))) * src) {
    static_assert(sizeof(S) % 8 == 0, "eight-byte granules");
    typedef const uint64_t __attribute__((address_space(4))) * P;
    P p = (P)src;
    uint64_t* q = reinterpret_cast<uint64_t*>(&dst);
#pragma unroll
    for (size_t i = 0; i < sizeof(S) / 8; ++i) q[i] = p[i];
}

template <int PASS>
__device__ __forceinline__ int sos_batch_member(const SosDesc* __restrict__ desc, int n, int64_t blk) {
    SosDescPtr d = (SosDescPtr)desc;
    int lo = 0, hi = n;  // largest m with first[m] <= blk
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (d[mid].first[PASS] <= blk) lo = mid;
        else hi = mid;
    }
    return lo;
}

template <int NS, typename T, bool APPLY>
__global__ __launch_bounds__(kBlock) void k_sos_tiled_batch(const SosDesc* __restrict__ desc, int n) {
    const int m = sos_batch_member<APPLY ? 2 : 0>(desc, n, (int64_t)blockIdx.x);
    SosDescPtr d = (SosDescPtr)desc + m;
    SosGeom g;
    SosCoefs cf;
    load_const(g, &d->g);
    load_const(cf, &d->cf);
    sos_tiled_body<NS, T, APPLY>((const T*)d->x, (T*)d->y, APPLY ? (const double*)d->s0 : nullptr, APPLY ? nullptr : d->v, g, cf,
                                 (int64_t)blockIdx.x - d->first[APPLY ? 2 : 0]);
}

template <int NS>
__global__ __launch_bounds__(kBlock) void k_sos_scan_batch(const SosDesc* __restrict__ desc, int n) {
    const int m = sos_batch_member<1>(desc, n, (int64_t)blockIdx.x);
    SosDescPtr d = (SosDescPtr)desc + m;
    SosGeom g;
    load_const(g, &d->g);
    sos_scan_body<NS>(d->v, d->mpow, g, d->s0, (int64_t)blockIdx.x - d->first[1]);
}

template <int NS, typename T>
static void launch_sos_t(const void* x, void* y, double* v, double* s0, const double* mpow,
                         const SosGeom& g, const SosCoefs& cf, hipStream_t st) {
    const int64_t nseq = (int64_t)g.nchunks * g.nch;
    if (g.nchunks > 1) {
        const int64_t n1 = (int64_t)(g.nchunks - 1) * g.nch;
        hipLaunchKernelGGL((k_sos_tiled<NS, T, false>), dim3((unsigned)((n1 + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, st, (const T*)x, (T*)y, (const double*)nullptr, v, g, cf);
        hipLaunchKernelGGL((k_sos_scan<NS>), dim3((unsigned)((nseq + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, st, v, mpow, g, s0);
    }
    hipLaunchKernelGGL((k_sos_tiled<NS, T, true>), dim3((unsigned)((nseq + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, st, (const T*)x, (T*)y, g.nchunks > 1 ? (const double*)s0 : nullptr,
                       (double*)nullptr, g, cf);
}

template <typename T>
static void launch_sos_ns(const void* x, void* y, double* v, double* s0, const double* mpow,
                          const SosGeom& g, const SosCoefs& cf, hipStream_t st) {
    switch (cf.nsec) {
    case 1: launch_sos_t<1, T>(x, y, v, s0, mpow, g, cf, st); break;
    case 2: launch_sos_t<2, T>(x, y, v, s0, mpow, g, cf, st); break;
    case 3: launch_sos_t<3, T>(x, y, v, s0, mpow, g, cf, st); break;
    case 4: launch_sos_t<4, T>(x, y, v, s0, mpow, g, cf, st); break;
    case 5: launch_sos_t<5, T>(x, y, v, s0, mpow, g, cf, st); break;
    case 6: launch_sos_t<6, T>(x, y, v, s0, mpow, g, cf, st); break;
    case 7: launch_sos_t<7, T>(x, y, v, s0, mpow, g, cf, st); break;
    default: launch_sos_t<8, T>(x, y, v, s0, mpow, g, cf, st); break;
    }
}

// The state pass already done by the resampler in front (k_resample_periodic's state waves):
// vper[ch][period][16] holds the zero-state end-of-period states; a chunk is pt periods:
//   v_chunk = sum_{p<pt} Q^(pt-1-p) v_p,  Q = A^Ls   (Horner, Q in LDS)
template <int NS>
__global__ __launch_bounds__(kBlock) void k_sos_combine(const double* __restrict__ vper, int64_t nper,
                                                        const double* __restrict__ qmat, int pt, SosGeom g,
                                                        double* __restrict__ v) {
    constexpr int D = 2 * NS;
    __shared__ double q[D * D];
    if ((int)threadIdx.x < D * D) q[threadIdx.x] = qmat[threadIdx.x];
    __syncthreads();
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t nseq = (int64_t)g.nchunks * g.nch;
    if (tid >= nseq) return;
    const int k = (int)(tid % g.nchunks), ch = (int)(tid / g.nchunks);
    if ((int64_t)(k + 1) * pt > nper) return;  // (an incomplete last chunk: its end state is never used)
    double acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.0;
    const double* vp = vper + ((int64_t)ch * nper + (int64_t)k * pt) * 16;
    const int64_t half = (int64_t)g.nch * nper * 16;  // the second state wave's partial sums
    for (int p = 0; p < pt; ++p, vp += 16) {
        double t[D];
#pragma unroll
        for (int r = 0; r < D; ++r) {
            double a = vp[r] + vp[half + r];
#pragma unroll
            for (int d = 0; d < D; ++d) a = fma(q[r * D + d], acc[d], a);
            t[r] = a;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = t[d];
    }
    double* o = v + ((int64_t)ch * g.nchunks + k) * D;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = acc[d];
}

template <int NS, typename T>
static void launch_sos_pre_t(const void* x, void* y, const double* vper, int64_t nper, const double* qmat, int pt,
                             double* v, double* s0, const double* mpow, const SosGeom& g, const SosCoefs& cf,
                             hipStream_t st) {
    const int64_t nseq = (int64_t)g.nchunks * g.nch;
    hipLaunchKernelGGL((k_sos_combine<NS>), dim3((unsigned)((nseq + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, vper, nper,
                       qmat, pt, g, v);
    hipLaunchKernelGGL((k_sos_scan<NS>), dim3((unsigned)((nseq + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, v, mpow, g, s0);
    hipLaunchKernelGGL((k_sos_tiled<NS, T, true>), dim3((unsigned)((nseq + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       (const T*)x, (T*)y, (const double*)s0, (double*)nullptr, g, cf);
}

// IIR with the state pass precomputed (returns the number of launches)
int launch_sos_prestate(const void* x, void* y, const double* vper, int64_t nper, const double* qmat, int pt,
                        double* v, double* s0, const double* mpow, const SosGeom& g, const SosCoefs& cf,
                        hipStream_t st) {
    if (g.n <= 0) return 0;
#define SO_PRE(NS_)                                                                                             \
    case NS_:                                                                                                   \
        if (g.in_dtype == SO_F32) launch_sos_pre_t<NS_, float>(x, y, vper, nper, qmat, pt, v, s0, mpow, g, cf, st); \
        else launch_sos_pre_t<NS_, double>(x, y, vper, nper, qmat, pt, v, s0, mpow, g, cf, st);                    \
        break;
    switch (cf.nsec) {
        SO_PRE(1) SO_PRE(2) SO_PRE(3) SO_PRE(4) SO_PRE(5) SO_PRE(6) SO_PRE(7)
    default:
        if (g.in_dtype == SO_F32) launch_sos_pre_t<8, float>(x, y, vper, nper, qmat, pt, v, s0, mpow, g, cf, st);
        else launch_sos_pre_t<8, double>(x, y, vper, nper, qmat, pt, v, s0, mpow, g, cf, st);
    }
#undef SO_PRE
    return 3;
}

// One pass of the three-pass form on its own (phase 1: chunk end states v from zero state; phase 3:
// outputs from the chunk start states s0) -- for callers that put a different scan in between
// (kernels2.hip launch_sos_xscan).  Returns the number of launches.
template <int NS, typename T>
static void launch_sos_phase_t(const void* x, void* y, double* v, const double* s0, const SosGeom& g, const SosCoefs& cf,
                               int phase, hipStream_t st) {
    const int64_t nseq = (int64_t)g.nchunks * g.nch;
    if (phase == 1) {
        const int64_t n1 = (int64_t)(g.nchunks - 1) * g.nch;
        hipLaunchKernelGGL((k_sos_tiled<NS, T, false>), dim3((unsigned)((n1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           (const T*)x, (T*)y, (const double*)nullptr, v, g, cf);
    } else {
        hipLaunchKernelGGL((k_sos_tiled<NS, T, true>), dim3((unsigned)((nseq + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           (const T*)x, (T*)y, s0, (double*)nullptr, g, cf);
    }
}
int launch_sos_phase(const void* x, void* y, double* v, const double* s0, const SosGeom& g, const SosCoefs& cf, int phase,
                     hipStream_t st) {
    if (g.n <= 0 || (phase == 1 && g.nchunks <= 1)) return 0;
#define SO_PH(NS_)                                                                                     \
    case NS_:                                                                                          \
        if (g.in_dtype == SO_F32) launch_sos_phase_t<NS_, float>(x, y, v, s0, g, cf, phase, st);          \
        else launch_sos_phase_t<NS_, double>(x, y, v, s0, g, cf, phase, st);                              \
        break;
    switch (cf.nsec) {
        SO_PH(1) SO_PH(2) SO_PH(3) SO_PH(4) SO_PH(5) SO_PH(6) SO_PH(7)
    default:
        if (g.in_dtype == SO_F32) launch_sos_phase_t<8, float>(x, y, v, s0, g, cf, phase, st);
        else launch_sos_phase_t<8, double>(x, y, v, s0, g, cf, phase, st);
    }
#undef SO_PH
    return 1;
}

// ... and for the members of a batch (one launch: blockIdx.y = member, its channels in turn)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sos_poison_batch(const SosDesc* __restrict__ desc) {
    SosDescPtr d = (SosDescPtr)desc + blockIdx.y;
    SosGeom g;
    load_const(g, &d->g);
    if (g.bad == nullptr || g.nchunks <= 1) return;
    T* y = (T*)d->y;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int ch = 0; ch < g.nch; ++ch) {
        const int kb = g.bad[ch];
        if (kb >= g.nchunks - 1) continue;
        int64_t f0 = (int64_t)(kb + 1) * g.chunk;
        if (f0 < g.store_lo) f0 = g.store_lo;
        for (int64_t f = f0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; f < g.n; f += (int64_t)gridDim.x * kBlock) {
            const int64_t o = (int64_t)ch * g.out_pitch + f;
            if (sizeof(T) == 8 && g.out_dtype == SO_F32) reinterpret_cast<float*>(y)[o] = (float)nan;
            else y[o] = (T)nan;
        }
    }
}

int launch_sos_poison_batch(const SosDesc* desc, int n, int dtype, hipStream_t st) {
    const dim3 grid(8, (unsigned)n);
    if (dtype == SO_F32) hipLaunchKernelGGL((k_sos_poison_batch<float>), grid, dim3(kBlock), 0, st, desc);
    else hipLaunchKernelGGL((k_sos_poison_batch<double>), grid, dim3(kBlock), 0, st, desc);
    return 1;
}

template <int NS, typename T>
static void launch_sos_batch_t(const SosDesc* desc, int n, const int64_t* total, hipStream_t st) {
    if (total[0] > 0)
        hipLaunchKernelGGL((k_sos_tiled_batch<NS, T, false>), dim3((unsigned)total[0]), dim3(kBlock), 0, st, desc, n);
    if (total[1] > 0) hipLaunchKernelGGL((k_sos_scan_batch<NS>), dim3((unsigned)total[1]), dim3(kBlock), 0, st, desc, n);
    hipLaunchKernelGGL((k_sos_tiled_batch<NS, T, true>), dim3((unsigned)total[2]), dim3(kBlock), 0, st, desc, n);
}

// desc: device array of n + 1 descriptors (the last one carries the grid sizes in `first`); total = those sizes
int launch_sos_batch(const SosDesc* desc, int n, int nsec, int dtype, const int64_t* total, hipStream_t st) {
    switch (nsec) {
#define SOS_BATCH_CASE(NS_)                                                        \
    case NS_:                                                                      \
        if (dtype == SO_F32) launch_sos_batch_t<NS_, float>(desc, n, total, st);   \
        else launch_sos_batch_t<NS_, double>(desc, n, total, st);                  \
        break;
        SOS_BATCH_CASE(1) SOS_BATCH_CASE(2) SOS_BATCH_CASE(3) SOS_BATCH_CASE(4)
        SOS_BATCH_CASE(5) SOS_BATCH_CASE(6) SOS_BATCH_CASE(7)
#undef SOS_BATCH_CASE
    default:
        if (dtype == SO_F32) launch_sos_batch_t<8, float>(desc, n, total, st);
        else launch_sos_batch_t<8, double>(desc, n, total, st);
    }
    return (total[0] > 0) + (total[1] > 0) + 1;
}

int launch_sos(const void* x, void* y, double* v, double* s0, const double* mpow,
               const SosGeom& g, const SosCoefs& cf, hipStream_t st) {
    if (g.n <= 0) return 0;
    if (g.in_dtype == SO_F32) launch_sos_ns<float>(x, y, v, s0, mpow, g, cf, st);
    else launch_sos_ns<double>(x, y, v, s0, mpow, g, cf, st);
    return g.nchunks > 1 ? 3 : 1;
}

// ---------------------------------------------------------------------------
// K2 single pass: one read and one write of the signal (the three-pass form above reads it twice).
//
// Every WAVE works on its own: it takes the next tile in TIME ORDER (atomic ticket; a tile is
// 2048 frames of one channel), loads it with coalesced 16-byte accesses and transposes it
// through a small LDS buffer so that lane k holds sub-chunk k (kSosLc consecutive frames) in
// REGISTERS:
//   1. zero-state DF2T over the lane's sub-chunk                    -> v_k  (state at its end)
//   2. inclusive scan over the 64 lanes with powers of M = A^lc (Kogge-Stone, ds_bpermute):
//                                                    P_k = sum_{i<=k} M^(k-i) v_i ; V = P_63
//   3. V (the tile's zero-state end state, a function of the tile's own samples only) is
//      published; the state entering the tile is
//          sigma = sum_{j>=0} (A^tf)^j V_(t-1-j),  truncated after kt terms (||(A^tf)^kt|| < 2^-70)
//      -- a look-back over kt earlier tiles of the same channel that are all in flight or done
//      (lower tickets) and whose V never waits for anything: no serial chain through the tiles.
//      V slots are pre-set to an all-ones bit pattern (a NaN no arithmetic produces) and written
//      with agent-scope atomic stores, so "is it there yet" and the value are ONE memory round
//      trip, with no flag, no fence and no cache-wide writeback/invalidate.
//   4. s0_k = P_(k-1) + M^k sigma (M^k by the binary expansion of k), DF2T from s0_k on the
//      registers, transpose back, coalesced store.
// No workgroup barrier after the matrices are staged: loads, arithmetic, look-back latency and
// stores of the ~12 waves of a CU overlap on their own.  State matrices are lower
// block-triangular (cascade), so only those entries are multiplied.  The arithmetic that
// produces the outputs is the same DF2T recurrence as DSP.jl's filt! from a start state that
// differs from the sequential one by rounding (~1e-16 relative).
template <int D>
__device__ __forceinline__ void matvec_tri(const double* __restrict__ m, const double (&v)[D], double (&out)[D]) {
#pragma unroll
    for (int r = 0; r < D; ++r) {
        double a = 0.0;
#pragma unroll
        for (int c = 0; c <= (r | 1); ++c) a = fma(m[r * D + c], v[c], a);
        out[r] = a;
    }
}
__device__ __forceinline__ double bperm_f64(int byte_addr, double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(byte_addr, lo);
    hi = __builtin_amdgcn_ds_bpermute(byte_addr, hi);
    return __hiloint2double(hi, lo);
}
constexpr int kSosTf = 64 * kSosLc;            // frames per wave tile
constexpr unsigned long long kSosEmpty = ~0ull;  // "not published yet"

template <int NS, typename T>
__global__ __launch_bounds__(kBlock, 2) void k_sos_onepass(const T* __restrict__ x, T* __restrict__ y, SosOne g,
                                                        SosCoefs cf, const double* __restrict__ tabs,
                                                        int* __restrict__ sync, double* __restrict__ vpub) {
    constexpr int D = 2 * NS;
    constexpr int LC = kSosLc, LP = kSosLc + 1;  // odd pitch: the 16 rows of a round fall on different banks
    constexpr int V = 16 / (int)sizeof(T);       // elements per 16-byte vector
    constexpr int RV = 512 / V / 64;             // vectors per lane and round (a round = 512 frames = 16 rows)
    typedef T vecT __attribute__((ext_vector_type(V)));
    extern __shared__ double lds_raw[];
    double* const ksm = lds_raw;  // [nlev][D*D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T* const buf = reinterpret_cast<T*>(lds_raw + g.nlev * D * D) + wave * (16 * LP);  // this wave's transposer
    for (int i = tid; i < g.nlev * D * D; i += blockDim.x) ksm[i] = tabs[i];
    __syncthreads();  // (the only workgroup barrier)
    // One ticket per workgroup and up to kSosBatch tiles per wave: same-address atomics run at
    // ~10-30 ns each on this chip, so a ticket per tile would by itself cost more than the whole
    // kernel (measured: 112 500 tickets = 1.36 ms with all arithmetic removed).  A wave's tiles are
    // the SAME time tile of g.bt different channels: tile (t, c_i) waits for (t-1, c_i), which is
    // iteration i of a wave with a lower slot number -- the chain of "publish after the previous
    // iteration's wait" steps down one iteration per link, so it is at most g.bt long.  (Time-
    // consecutive tiles in one wave would chain through ALL running workgroups: measured 267 ms.)
    __shared__ int s_ticket;
    if (tid == 0) s_ticket = atomicAdd(&sync[0], 1);
    __syncthreads();
    const int ncs = (g.nch + g.bt - 1) / g.bt;             // channel slots per time tile
    const int64_t slot = (int64_t)s_ticket * (kBlock / 64) + wave;  // time-major: (t-1, cs) is a lower slot
    if (slot >= (int64_t)g.ntiles * ncs) return;
    const int tt = (int)(slot / ncs), cs = (int)(slot - (int64_t)tt * ncs);
    const int64_t f0 = (int64_t)tt * kSosTf;
    const int64_t left = g.n - f0;  // frames of this time tile inside the signal (>= 1)
    const int ch0 = cs * g.bt;
    const int nit = g.nch - ch0 < g.bt ? g.nch - ch0 : g.bt;
    const int grp = lane >> 4, rrow = (lane & 15) * LP;
    const int k = lane;
    auto issue_loads = [&](int ch, vecT (&ld)[4][RV]) {
        const T* __restrict__ xin = x + (int64_t)ch * g.in_pitch + f0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < RV; ++j) {
                const int f = (r * (512 / V) + j * 64 + lane) * V;
#pragma unroll
                for (int e = 0; e < V; ++e) ld[r][j][e] = (T)0;
                if (g.vec_in && f + V <= left) ld[r][j] = *reinterpret_cast<const vecT*>(xin + f);
                else {
#pragma unroll
                    for (int e = 0; e < V; ++e)
                        if (f + e < left) ld[r][j][e] = xin[f + e];
                }
            }
    };
    vecT ld[4][RV];
    issue_loads(ch0, ld);
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
        const int ch = ch0 + it;
        T* __restrict__ yout = y + (int64_t)ch * g.out_pitch + f0;
        // ---- four transposition rounds: lane k gets sub-chunk k in registers ----
        T xr[LC];
#pragma unroll
        for (int n = 0; n < LC; ++n) xr[n] = (T)0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < RV; ++j) {
                const int fl = (j * 64 + lane) * V;  // frame inside the round
                T* dst = buf + (fl / LC) * LP + (fl % LC);
#pragma unroll
                for (int e = 0; e < V; ++e) dst[e] = ld[r][j][e];
            }
            __builtin_amdgcn_wave_barrier();
            if (grp == r) {
#pragma unroll
                for (int n = 0; n < LC; ++n) xr[n] = buf[rrow + n];
            }
            __builtin_amdgcn_wave_barrier();
        }
        // the next tile's loads fly during this tile's arithmetic
        if (it + 1 < nit) issue_loads(ch + 1, ld);
        // ---- 1: zero-state pass over the lane's sub-chunk ----
        double s[D];
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = 0.0;
        if (!(g.debug & 1)) {
#pragma unroll
            for (int n = 0; n < LC; ++n) (void)sos_step<NS>((double)xr[n], s, cf);
        }
        // ---- 2: inclusive scan over the lanes ----
#pragma unroll 1
        for (int lev = 0; lev < 6 && !(g.debug & 2); ++lev) {
            const int d = 1 << lev;
            const int addr = (lane - d) << 2;
            double p[D], q[D];
#pragma unroll
            for (int i = 0; i < D; ++i) p[i] = bperm_f64(addr, s[i]);
            matvec_tri<D>(ksm + lev * D * D, p, q);
            if (k >= d) {
#pragma unroll
                for (int i = 0; i < D; ++i) s[i] += q[i];
            }
        }
        // ---- 3: publish V, look back ----
        if (k == 63) {
            double* vp = vpub + ((int64_t)tt * g.nch + ch) * D;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double pv = s[i];
                if ((unsigned long long)__double_as_longlong(pv) == kSosEmpty) pv = __longlong_as_double(0x7ff8000000000000ll);
                __hip_atomic_store(vp + i, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        double e0[D];  // P_(k-1)
        {
            const int addr = (lane - 1) << 2;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double up = bperm_f64(addr, s[i]);
                e0[i] = k > 0 ? up : 0.0;
            }
        }
        // ---- 4a: the outputs from P_(k-1) alone, on the registers.  The part that needs sigma is
        //      added afterwards (superposition): the look-back's memory round trip -- the earlier
        //      tile publishes its V at about the time this one asks for it -- hides behind this pass
        //      instead of stalling every wave of the CU at the same point (measured: 0.5 ms).
        // (Float64: the outputs replace the samples in their registers; Float32 samples keep the
        //  sum in Float64 until the final rounding)
        typedef typename std::conditional<sizeof(T) == 8, T, double>::type YT;
        YT yloc[sizeof(T) == 8 ? 1 : LC];
        YT* const yv = sizeof(T) == 8 ? reinterpret_cast<YT*>(xr) : yloc;
        if (!(g.debug & 16)) {
#pragma unroll
            for (int n = 0; n < LC; ++n) yv[n] = sos_step<NS>((double)xr[n], e0, cf);
        }
        const int nb = tt < g.kt ? tt : g.kt;  // earlier tiles that still matter
        if (nb > 0 && !(g.debug & 4)) {        // (wave-uniform)
            double w[D];
#pragma unroll
            for (int i = 0; i < D; ++i) w[i] = 0.0;
            if (k < nb) {
                const double* vp = vpub + ((int64_t)(tt - 1 - k) * g.nch + ch) * D;
                double vv[D];
                for (;;) {
                    if (g.debug & 64) {
#pragma unroll
                        for (int i = 0; i < D; ++i) vv[i] = 0.0;
                        break;
                    }
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        vv[i] = __hip_atomic_load(vp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && (unsigned long long)__double_as_longlong(vv[i]) != kSosEmpty;
                    }
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                matvec_tri<D>(tabs + (size_t)(g.nlev + k) * D * D, vv, w);  // (A^tf)^k V_(t-1-k)
            }
            // sum of the first nb lanes into lane 0 (the others hold zeros), then to every lane
            for (int off = 1; off < nb; off <<= 1) {
                const int addr = (lane ^ off) << 2;
#pragma unroll
                for (int i = 0; i < D; ++i) w[i] += bperm_f64(addr, w[i]);
            }
#pragma unroll
            for (int i = 0; i < D; ++i) w[i] = rfl_f64(w[i]);
            // M^k sigma by the binary expansion of k
#pragma unroll 1
            for (int lev = 0; lev < 6 && !(g.debug & 8); ++lev) {
                double q[D];
                matvec_tri<D>(ksm + lev * D * D, w, q);
                if ((k >> lev) & 1) {
#pragma unroll
                    for (int i = 0; i < D; ++i) w[i] = q[i];
                }
            }
            // ---- 4b: zero-input response of the sub-chunk to M^k sigma ----
            if (!(g.debug & (16 | 32))) {
#pragma unroll
                for (int n = 0; n < LC; ++n) yv[n] += sos_step<NS>(0.0, w, cf);
            }
        }
#pragma unroll
        for (int n = 0; n < LC; ++n) xr[n] = (T)(yv[n] * cf.gain);
        // ---- transpose back and store ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (grp == r) {
#pragma unroll
                for (int n = 0; n < LC; ++n) buf[rrow + n] = xr[n];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < RV; ++j) {
                const int fl = (j * 64 + lane) * V;
                const T* src = buf + (fl / LC) * LP + (fl % LC);
                const int f = r * 512 + fl;
                if (g.vec_out && f + V <= left) {
                    vecT o;
#pragma unroll
                    for (int e = 0; e < V; ++e) o[e] = src[e];
                    *reinterpret_cast<vecT*>(yout + f) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < V; ++e)
                        if (f + e < left) yout[f + e] = src[e];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}


template <int NS, typename T>
static void launch_sos_one_t(const void* x, void* y, const SosOne& g, const SosCoefs& cf, const double* tabs,
                             int* sync, double* vpub, hipStream_t st) {
    const size_t ldsb = (size_t)g.nlev * 4 * NS * NS * 8 + (size_t)(kBlock / 64) * 16 * (kSosLc + 1) * sizeof(T);
    const int64_t nslots = (int64_t)g.ntiles * ((g.nch + g.bt - 1) / g.bt);  // one per wave
    const int64_t per = kBlock / 64;
    hipLaunchKernelGGL((k_sos_onepass<NS, T>), dim3((unsigned)((nslots + per - 1) / per)), dim3(kBlock), ldsb, st,
                       (const T*)x, (T*)y, g, cf, tabs, sync, vpub);
}

template <typename T>
static void launch_sos_one_ns(const void* x, void* y, const SosOne& g, const SosCoefs& cf, const double* tabs,
                              int* sync, double* vpub, hipStream_t st) {
    switch (cf.nsec) {
    case 1: launch_sos_one_t<1, T>(x, y, g, cf, tabs, sync, vpub, st); break;
    case 2: launch_sos_one_t<2, T>(x, y, g, cf, tabs, sync, vpub, st); break;
    case 3: launch_sos_one_t<3, T>(x, y, g, cf, tabs, sync, vpub, st); break;
    case 4: launch_sos_one_t<4, T>(x, y, g, cf, tabs, sync, vpub, st); break;
    case 5: launch_sos_one_t<5, T>(x, y, g, cf, tabs, sync, vpub, st); break;
    case 6: launch_sos_one_t<6, T>(x, y, g, cf, tabs, sync, vpub, st); break;
    case 7: launch_sos_one_t<7, T>(x, y, g, cf, tabs, sync, vpub, st); break;
    default: launch_sos_one_t<8, T>(x, y, g, cf, tabs, sync, vpub, st); break;
    }
}

void launch_sos_onepass(const void* x, void* y, const SosOne& g, const SosCoefs& cf, const double* tabs,
                        int* sync, double* vpub, int dtype, hipStream_t st) {
    if (g.n <= 0) return;
    if (dtype == SO_F32) launch_sos_one_ns<float>(x, y, g, cf, tabs, sync, vpub, st);
    else launch_sos_one_ns<double>(x, y, g, cf, tabs, sync, vpub, st);
}

}  // namespace so
