// Hand-written HIP kernels for gfx950 (MI355X, CDNA4; wave64).  No CUDA shims, no
// dual paths.  Every kernel here is HBM-bound integer/fp64 streaming work: the
// levers are coalesced loads/stores, LDS staging and enough workgroups for 256 CUs.
//
//   k_pointwise   K1  fused generator / map / ramp / index kernel
//                     (replaces frame() recursion + sink_helper!, reference
//                      src/sink.jl:256-260, src/mapsignal.jl:249-272)
//   k_sos_*       K2  second-order-sections IIR, time-parallel by exact chunked
//                     state propagation (reference src/filters.jl:252-255 + DSP.jl DF2T)
//   k_resample    K3  polyphase FIR resampler (reference src/reformatting.jl:92-98)
//   k_sumsq_*     K4  Normpower reduction (reference src/filters.jl:296-309)
#include <hip/hip_runtime.h>

#include "../../include/sigops.h"
#include "kernels.h"
#include "sigops_internal.h"

namespace so {

// ---------------------------------------------------------------------------
// leaf evaluators
__device__ __forceinline__ double leaf_load(const DLeaf& L, int64_t n, int c) {
    int64_t f = (int64_t)L.sf * n + L.df;
    if (L.mode == LM_CYCLE) {  // x[(i-1)%end+1]  reference src/padding.jl:132
        f = f % L.modn;
    } else if (L.mode == LM_MIRROR) {  // reference src/padding.jl:142-148
        int64_t cnt = f / L.modn, rem = f % L.modn;
        f = (cnt & 1) ? L.modn - rem - 1 : rem;
    }
    int64_t ch = (int64_t)L.sc * c + L.dc;
    int64_t off = f * L.fstride + ch * L.cstride;
    if (L.dtype == SO_F32) return (double)((const float*)L.base)[off];
    return ((const double*)L.base)[off];
}

// reference src/functions.jl:53-60 — every operation separately rounded (Julia does
// not contract), frame index is 1-based so the first sample is t = 1/fs
__device__ __forceinline__ double func_eval(const DLeaf& L, int64_t n) {
    double i1 = (double)((int64_t)L.sf * n + L.df + 1);
    double t = __ddiv_rn(i1, L.v2);
    if (L.flag) {
        double ph = __dadd_rn(__dmul_rn(t, L.v0), L.v1);
        if (L.mode == SO_FN_SIN) return sinpi(2.0 * ph);
        double a = __dmul_rn(6.283185307179586, fmod(ph, 1.0));
        return L.mode == SO_FN_COS ? cos(a) : a;
    }
    double tt = __dadd_rn(t, L.v1);
    if (L.mode == SO_FN_SIN) return sinpi(2.0 * tt);
    return L.mode == SO_FN_COS ? cos(tt) : tt;
}

// reference src/ramps.jl:60-72
__device__ __forceinline__ double ramp_eval(const DLeaf& L, int64_t n) {
    int64_t n0 = (int64_t)L.sf * n + L.df;
    double x;
    if (L.flag == 0)
        x = __ddiv_rn((double)n0, L.v0);
    else
        x = __dsub_rn(1.0, __ddiv_rn((double)(n0 + 1 - L.modn), L.v0));
    return L.mode == SO_RAMP_SINRAMP ? sinpi(0.5 * x) : x;
}

// ---------------------------------------------------------------------------
// 4-deep register stack machine over E frames per thread.  Program words are
// wave-uniform (scalar loads); the stack lives in VGPRs (static indexing only).
template <int E>
struct Frames {
    int64_t n[E];
};

#define SO_PUSH(expr)                 \
    _Pragma("unroll") for (int e = 0; e < E; ++e) { \
        s3[e] = s2[e];                \
        s2[e] = s1[e];                \
        s1[e] = s0[e];                \
        s0[e] = (expr);               \
    }
#define SO_BIN(opr)                   \
    _Pragma("unroll") for (int e = 0; e < E; ++e) { \
        s0[e] = s1[e] opr s0[e];      \
        s1[e] = s2[e];                \
        s2[e] = s3[e];                \
    }

template <int E>
__device__ __forceinline__ void run_program(const DOp* __restrict__ ops, int pc, int len,
                                            const DLeaf* __restrict__ leaves,
                                            const int64_t (&n)[E], int c,
                                            double (&F)[kMaxFrameSlots][E], double (&out)[E]) {
    double s0[E], s1[E], s2[E], s3[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s0[e] = s1[e] = s2[e] = s3[e] = 0.0;
    for (int i = 0; i < len; ++i) {
        const DOp op = ops[pc + i];
        switch (op.code) {
        case OP_CONST: {
            const double v = leaves[op.arg].v0;
            SO_PUSH(v);
            break;
        }
        case OP_LOAD: {
            const DLeaf& L = leaves[op.arg];
            SO_PUSH(leaf_load(L, n[e], c));
            break;
        }
        case OP_SCALAR: {
            const double v = *(const double*)leaves[op.arg].base;
            SO_PUSH(v);
            break;
        }
        case OP_FUNC: {
            const DLeaf& L = leaves[op.arg];
            SO_PUSH(func_eval(L, n[e]));
            break;
        }
        case OP_RAMP: {
            const DLeaf& L = leaves[op.arg];
            SO_PUSH(ramp_eval(L, n[e]));
            break;
        }
        case OP_ADD: SO_BIN(+); break;
        case OP_SUB: SO_BIN(-); break;
        case OP_MUL: SO_BIN(*); break;
        case OP_DIV: SO_BIN(/); break;
        case OP_NEG:
#pragma unroll
            for (int e = 0; e < E; ++e) s0[e] = -s0[e];
            break;
        case OP_ROUND32:
#pragma unroll
            for (int e = 0; e < E; ++e) s0[e] = (double)(float)s0[e];
            break;
        case OP_STOREF:
#pragma unroll
            for (int e = 0; e < E; ++e) {
                switch (op.arg) {
                case 0: F[0][e] = s0[e]; break;
                case 1: F[1][e] = s0[e]; break;
                case 2: F[2][e] = s0[e]; break;
                default: F[3][e] = s0[e]; break;
                }
                s0[e] = s1[e];
                s1[e] = s2[e];
                s2[e] = s3[e];
            }
            break;
        case OP_LOADF:
            switch (op.arg) {
            case 0: SO_PUSH(F[0][e]); break;
            case 1: SO_PUSH(F[1][e]); break;
            case 2: SO_PUSH(F[2][e]); break;
            default: SO_PUSH(F[3][e]); break;
            }
            break;
        default: break;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] = s0[e];
}

// K1: one workgroup = kBlock*E consecutive frames x a channel chunk of one piece.
// Lane l handles frames base + l + e*kBlock, so every load/store instruction is a
// fully coalesced run of 64 consecutive elements per wave.
template <int E>
__global__ __launch_bounds__(kBlock) void k_pointwise(const DPiece* __restrict__ pieces,
                                                      int npieces, const DOp* __restrict__ ops,
                                                      const DLeaf* __restrict__ leaves,
                                                      OutView out) {
    const int64_t bid = blockIdx.x;
    int lo = 0, hi = npieces - 1;
    while (lo < hi) {  // wave-uniform binary search: piece owning this workgroup
        int mid = (lo + hi + 1) >> 1;
        if (pieces[mid].block0 <= bid) lo = mid;
        else hi = mid - 1;
    }
    const DPiece P = pieces[lo];
    const int64_t rel = bid - P.block0;
    const int64_t bf = rel % P.nblk_f;
    const int bc = (int)(rel / P.nblk_f);
    const int cbeg = P.c0 + bc * P.chc;
    const int cend = min(P.c1, cbeg + P.chc);
    int64_t n[E], ns[E];
    bool valid[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        ns[e] = P.a + bf * (int64_t)(kBlock * E) + threadIdx.x + (int64_t)e * kBlock;
        valid[e] = ns[e] < P.b;
        n[e] = valid[e] ? ns[e] : P.b - 1;  // clamp: loads stay in range, store is skipped
    }
    double F[kMaxFrameSlots][E];
#pragma unroll
    for (int k = 0; k < kMaxFrameSlots; ++k)
#pragma unroll
        for (int e = 0; e < E; ++e) F[k][e] = 0.0;
    double v[E];
    if (P.frame_len > 0) run_program<E>(ops, P.frame_pc, P.frame_len, leaves, n, cbeg, F, v);
    for (int c = cbeg; c < cend; ++c) {
        run_program<E>(ops, P.samp_pc, P.samp_len, leaves, n, c, F, v);
        if (out.dtype == SO_F32) {
            float* o = (float*)out.base + (int64_t)c * out.cstride;
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (valid[e]) o[ns[e] * out.fstride] = (float)v[e];
        } else {
            double* o = (double*)out.base + (int64_t)c * out.cstride;
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (valid[e]) o[ns[e] * out.fstride] = v[e];
        }
    }
}

void launch_pointwise(const DPiece* d_pieces, int npieces, int64_t nblocks, const DOp* d_ops,
                      const DLeaf* d_leaves, OutView out, hipStream_t st) {
    if (nblocks <= 0) return;
    hipLaunchKernelGGL((k_pointwise<2>), dim3((unsigned)nblocks), dim3(kBlock), 0, st, d_pieces,
                       npieces, d_ops, d_leaves, out);
}

// ---------------------------------------------------------------------------
// K2: SOS IIR.  The recurrence is linear, so a chunk's output is the zero-state
// response to its own samples plus the zero-input response to the state at its start.
//   pass 1  k_sos_state : v_k = state at the END of chunk k from zero state (only the
//           last min(L,W) frames matter: older frames have decayed below 2^-70)
//   pass 2  k_sos_scan  : s0_k = sum_{j=1..K} M^(j-1) v_{k-j},  M = A^L (host-computed
//           powers of the cascade's state matrix); K terms until ||M^K|| < 2^-70
//   pass 3  k_sos_apply : run DF2T on chunk k from s0_k and write the output
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

template <int NS, typename T>
__global__ __launch_bounds__(kBlock) void k_sos_state(const T* __restrict__ x, SosGeom g,
                                                      SosCoefs cf, double* __restrict__ v) {
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t nseq = (int64_t)(g.nchunks - 1) * g.nch;
    if (tid >= nseq) return;
    const int k = (int)(tid % (g.nchunks - 1));
    const int ch = (int)(tid / (g.nchunks - 1));
    const int64_t end = (int64_t)(k + 1) * g.chunk;
    const int64_t beg = end - (g.warm < g.chunk ? g.warm : g.chunk);
    const T* xp = x + (int64_t)ch * g.in_pitch;
    double s[2 * NS];
#pragma unroll
    for (int d = 0; d < 2 * NS; ++d) s[d] = 0.0;
    for (int64_t i = beg; i < end; ++i) (void)sos_step<NS>((double)xp[i], s, cf);
    double* vp = v + ((int64_t)ch * g.nchunks + k) * (2 * NS);
#pragma unroll
    for (int d = 0; d < 2 * NS; ++d) vp[d] = s[d];
}

// mpow: [kterms][D][D] row-major, mpow[0] = I
template <int NS>
__global__ __launch_bounds__(kBlock) void k_sos_scan(const double* __restrict__ v,
                                                     const double* __restrict__ mpow, SosGeom g,
                                                     double* __restrict__ s0) {
    constexpr int D = 2 * NS;
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t nseq = (int64_t)g.nchunks * g.nch;
    if (tid >= nseq) return;
    const int k = (int)(tid % g.nchunks);
    const int ch = (int)(tid / g.nchunks);
    double acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.0;
    const int K = k < g.kterms ? k : g.kterms;
    for (int j = K; j >= 1; --j) {  // smallest terms first
        const double* vp = v + ((int64_t)ch * g.nchunks + (k - j)) * D;
        const double* m = mpow + (int64_t)(j - 1) * D * D;
        double vv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) vv[d] = vp[d];
#pragma unroll
        for (int r = 0; r < D; ++r) {
            double a = acc[r];
#pragma unroll
            for (int d = 0; d < D; ++d) a += m[r * D + d] * vv[d];
            acc[r] = a;
        }
    }
    double* sp = s0 + ((int64_t)ch * g.nchunks + k) * D;
#pragma unroll
    for (int d = 0; d < D; ++d) sp[d] = acc[d];
}

template <int NS, typename T>
__global__ __launch_bounds__(kBlock) void k_sos_apply(const T* __restrict__ x,
                                                      const double* __restrict__ s0, SosGeom g,
                                                      SosCoefs cf, T* __restrict__ y) {
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t nseq = (int64_t)g.nchunks * g.nch;
    if (tid >= nseq) return;
    const int k = (int)(tid % g.nchunks);
    const int ch = (int)(tid / g.nchunks);
    const int64_t beg = (int64_t)k * g.chunk;
    const int64_t end = beg + g.chunk < g.n ? beg + g.chunk : g.n;
    const T* xp = x + (int64_t)ch * g.in_pitch;
    T* yp = y + (int64_t)ch * g.out_pitch;
    double s[2 * NS];
    const double* sp = s0 + ((int64_t)ch * g.nchunks + k) * (2 * NS);
#pragma unroll
    for (int d = 0; d < 2 * NS; ++d) s[d] = (s0 != nullptr && k > 0) ? sp[d] : 0.0;
    for (int64_t i = beg; i < end; ++i) {
        const double yv = sos_step<NS>((double)xp[i], s, cf);
        yp[i] = (T)(yv * cf.gain);
    }
}

template <int NS, typename T>
static void launch_sos_t(const void* x, void* y, double* v, double* s0, const double* mpow,
                         const SosGeom& g, const SosCoefs& cf, hipStream_t st) {
    const int64_t nseq = (int64_t)g.nchunks * g.nch;
    if (g.nchunks > 1) {
        const int64_t n1 = (int64_t)(g.nchunks - 1) * g.nch;
        hipLaunchKernelGGL((k_sos_state<NS, T>), dim3((unsigned)((n1 + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, st, (const T*)x, g, cf, v);
        hipLaunchKernelGGL((k_sos_scan<NS>), dim3((unsigned)((nseq + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, st, v, mpow, g, s0);
    }
    hipLaunchKernelGGL((k_sos_apply<NS, T>), dim3((unsigned)((nseq + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, st, (const T*)x, g.nchunks > 1 ? s0 : nullptr, g, cf,
                       (T*)y);
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

int launch_sos(const void* x, void* y, double* v, double* s0, const double* mpow,
               const SosGeom& g, const SosCoefs& cf, hipStream_t st) {
    if (g.n <= 0) return 0;
    if (g.in_dtype == SO_F32) launch_sos_ns<float>(x, y, v, s0, mpow, g, cf, st);
    else launch_sos_ns<double>(x, y, v, s0, mpow, g, cf, st);
    return g.nchunks > 1 ? 3 : 1;
}

// ---------------------------------------------------------------------------
// K3: polyphase resampler.  Output m sits at fine-grid position q_m (SURVEY.md
// Appendix A): j = newest input, p = phase, alpha = fractional phase;
//   y[m] = sum_k pfb[p][k] x[j-k]  +  alpha * sum_k dpfb[p][k] x[j-k]
// (DSP.jl FIRArbitrary: yLower + yUpper*alpha; rational kernels have alpha == 0).
// Position arithmetic is bit-exact with the oracle: two separately rounded fp64
// operations (no FMA contraction) or pure int64.
__device__ __forceinline__ void rs_pos(const RsGeom& g, int64_t m, int64_t& j, int& p,
                                       double& alpha) {
    if (g.arbitrary && g.exact) {
        const int64_t N = m * ((int64_t)g.nphi * g.M);
        const int64_t qi = g.c0i + N / g.L;
        alpha = __ddiv_rn((double)(N % g.L), (double)g.L);
        j = qi / g.nphi;
        p = (int)(qi % g.nphi);
    } else if (g.arbitrary) {
        const double t = __dmul_rn((double)m, g.delta);
        const double q = __dadd_rn(g.c0, t);
        const double fl = floor(q);
        const int64_t qi = (int64_t)fl;
        alpha = q - fl;
        j = qi / g.nphi;
        p = (int)(qi % g.nphi);
    } else {
        const int64_t qi = g.c0i + m * g.M;
        alpha = 0.0;
        j = qi / g.L;
        p = (int)(qi % g.L);
    }
}

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

// K3p: periodic polyphase resampler (rational L/M).  Workgroup = 256 threads = 4 waves;
// tile = pt periods x ct channels = 64 rows, staged once in LDS as fp64 (coalesced
// global reads, 3 % halo).  Lane = row.  Each wave walks its share of the period's output
// groups; per group the RM x kw tap table is wave-uniform (scalar loads -> SGPR FMA
// operands) and every ds_read_b64 of an input sample feeds RM fp64 FMAs, so neither LDS
// bandwidth nor tap traffic limits the kernel: HBM streaming does.
//   tab  [ngroups][kw][RM]  combined taps h + alpha*dh, oldest input first, zero padded
//   jend [ngroups]          newest input of the group's window, relative to the period base
template <typename T, int RM>
__global__ __launch_bounds__(1024) void k_resample_periodic(const T* __restrict__ x,
                                                              const double* __restrict__ tab,
                                                              const int* __restrict__ jend,
                                                              RsPeriodic g, T* __restrict__ y,
                                                              const DPiece* __restrict__ pieces,
                                                              int npieces,
                                                              const DOp* __restrict__ ops,
                                                              const DLeaf* __restrict__ leaves) {
    extern __shared__ double lds[];
    const int64_t P0 = (int64_t)blockIdx.x * g.pt;
    const int c0 = blockIdx.y * g.ct;
    const int64_t xbase = P0 * g.M + g.jlo;  // global input index of LDS slot 0 (can be < 0)
    if (npieces == 0) {
        // plain planar source
        for (int c = 0; c < g.ct; ++c) {
            const T* xp = x + (int64_t)(c0 + c) * g.in_pitch;
            double* lp = lds + c * g.lds_pitch;
            for (int i = threadIdx.x; i < g.tile_len; i += blockDim.x) {
                const int64_t gi = xbase + i;
                lp[i] = (gi >= 0 && gi < g.n_in) ? (double)xp[gi] : 0.0;  // zero padding
            }
        }
    } else {
        // fused source: the child's pointwise program (mapsignal / ramps / cuts /
        // generators) is evaluated straight into the LDS tile; the intermediate never
        // touches HBM.  Pieces are walked one at a time so control flow stays wave-uniform.
        const int64_t lo = xbase > 0 ? xbase : 0;
        const int64_t hi = xbase + g.tile_len < g.n_in ? xbase + g.tile_len : g.n_in;
        for (int i = threadIdx.x; i < g.tile_len; i += blockDim.x) {
            const int64_t gi = xbase + i;
            if (gi < 0 || gi >= g.n_in)
                for (int c = 0; c < g.ct; ++c) lds[c * g.lds_pitch + i] = 0.0;
        }
        int a = 0, b = npieces - 1;
        while (a < b) {  // first piece whose end is beyond lo (pieces are sorted by frame)
            int mid = (a + b) >> 1;
            if (pieces[mid].b > lo) b = mid;
            else a = mid + 1;
        }
        for (int pi = a; pi < npieces && pieces[pi].a < hi; ++pi) {
            const DPiece P = pieces[pi];
            const int64_t fa = P.a > lo ? P.a : lo;
            const int64_t fb = P.b < hi ? P.b : hi;
            for (int64_t gi = fa + threadIdx.x; gi < fb; gi += blockDim.x) {
                int64_t n[1] = {gi};
                double F[kMaxFrameSlots][1];
                double v[1];
#pragma unroll
                for (int k = 0; k < kMaxFrameSlots; ++k) F[k][0] = 0.0;
                if (P.frame_len > 0) run_program<1>(ops, P.frame_pc, P.frame_len, leaves, n, c0, F, v);
                const int slot = (int)(gi - xbase);
                for (int c = 0; c < g.ct; ++c) {
                    run_program<1>(ops, P.samp_pc, P.samp_len, leaves, n, c0 + c, F, v);
                    // the reference stores the child into a buffer of the child's sample
                    // type before filtering (src/filters.jl:207,244)
                    lds[c * g.lds_pitch + slot] = sizeof(T) == 4 ? (double)(float)v[0] : v[0];
                }
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pl = lane % g.pt, cl = lane / g.pt;
    const int64_t period = P0 + pl;
    const int rowbase = cl * g.lds_pitch + pl * (int)g.M - g.jlo - (g.kw - 1);
    T* yp = y + (int64_t)(c0 + cl) * g.out_pitch + period * g.L;
    const int64_t mbase = period * g.L;
    const int nwaves = blockDim.x >> 6;
    const int gper = (g.ngroups + nwaves - 1) / nwaves;
    const int gbeg = wave * gper;
    const int gend = min(g.ngroups, gbeg + gper);
    for (int gi = gbeg; gi < gend; ++gi) {
        const double* __restrict__ tg = tab + (size_t)gi * g.kw * RM;
        const double* __restrict__ xr = lds + (rowbase + jend[gi]);
        double acc[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) acc[r] = 0.0;
#pragma unroll 4
        for (int k = 0; k < g.kw; ++k) {
            const double xv = xr[k];
#pragma unroll
            for (int r = 0; r < RM; ++r) acc[r] = fma(tg[k * RM + r], xv, acc[r]);
        }
        const int r0 = gi * RM;
        if (period < g.nperiods) {
            if (g.vec_ok && r0 + RM <= g.L && mbase + r0 + RM <= g.n_out) {
                if constexpr (sizeof(T) == 8) {
#pragma unroll
                    for (int r = 0; r < RM; r += 2) {
                        double2 v;
                        v.x = acc[r];
                        v.y = acc[r + 1];
                        *reinterpret_cast<double2*>(yp + r0 + r) = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < RM; r += 4) {
                        float4 v;
                        v.x = (float)acc[r];
                        v.y = (float)acc[r + 1];
                        v.z = (float)acc[r + 2];
                        v.w = (float)acc[r + 3];
                        *reinterpret_cast<float4*>(yp + r0 + r) = v;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < RM; ++r)
                    if (r0 + r < g.L && mbase + r0 + r < g.n_out) yp[r0 + r] = (T)acc[r];
            }
        }
    }
}

void launch_resample_periodic(const void* x, void* y, const double* tab, const int* jend,
                              const RsPeriodic& g, int dtype, const DPiece* pieces, int npieces,
                              const DOp* ops, const DLeaf* leaves, hipStream_t st) {
    if (g.n_out <= 0) return;
    constexpr int RM = 8;
    dim3 grid((unsigned)((g.nperiods + g.pt - 1) / g.pt), (unsigned)(g.nch / g.ct));
    size_t lds = (size_t)g.ct * g.lds_pitch * sizeof(double);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_resample_periodic<double, RM>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_resample_periodic<float, RM>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (dtype == SO_F32)
        hipLaunchKernelGGL((k_resample_periodic<float, RM>), grid, dim3(64 * g.nwaves), lds, st,
                           (const float*)x, tab, jend, g, (float*)y, pieces, npieces, ops, leaves);
    else
        hipLaunchKernelGGL((k_resample_periodic<double, RM>), grid, dim3(64 * g.nwaves), lds, st,
                           (const double*)x, tab, jend, g, (double*)y, pieces, npieces, ops, leaves);
}

// ---------------------------------------------------------------------------
// K4: sum of squares over a planar [nch][pitch] buffer with n valid frames per
// channel; deterministic two-stage tree (no atomics), fp64 accumulation.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sumsq_partial(const T* __restrict__ x, int64_t n,
                                                          int nch, int64_t pitch,
                                                          double* __restrict__ partial) {
    __shared__ double red[kBlock / 64];
    const int64_t total = n * nch;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * kBlock) {
        const int64_t ch = i / n, f = i - ch * n;
        const double v = (double)x[ch * pitch + f];
        acc += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) s += red[w];
        partial[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(kBlock) void k_sumsq_final(const double* __restrict__ partial,
                                                        int nparts, double count,
                                                        double* __restrict__ rms) {
    __shared__ double red[kBlock];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += kBlock) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) rms[0] = sqrt(red[0] / count);
}

void launch_rms(const void* x, int dtype, int64_t n, int nch, int64_t pitch, double* partial,
                int nparts, double* rms, hipStream_t st) {
    if (dtype == SO_F32)
        hipLaunchKernelGGL((k_sumsq_partial<float>), dim3(nparts), dim3(kBlock), 0, st,
                           (const float*)x, n, nch, pitch, partial);
    else
        hipLaunchKernelGGL((k_sumsq_partial<double>), dim3(nparts), dim3(kBlock), 0, st,
                           (const double*)x, n, nch, pitch, partial);
    hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(kBlock), 0, st, partial, nparts,
                       (double)n * (double)nch, rms);
}

}  // namespace so
