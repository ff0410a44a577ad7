// Leaf evaluators of the fused pointwise programs (array loads, the compact sin/cos kernels, generator and
// ramp formulas): shared by the ahead-of-time kernels (kcommon.h) and -- as TEXT, embedded by build.py into
// rtc_embed.inc -- by the kernels hipRTC specialises at plan time (rtc.cpp).  Self-contained: needs only
// DLeaf (sigops_internal.h) and the SO_* constants; no standard headers.
#pragma once

namespace so {

// ---------------------------------------------------------------------------
// leaf evaluators
// All leaf parameters are wave-uniform (scalar registers); only the frame index n (and,
// in channel-vectorised evaluation, nothing else) lives per lane.  sf is -1/0/+1, so the
// per-lane address math is one 64-bit add in the common planar case (no 64-bit multiplies,
// which are quarter-rate on CDNA).
// Array leaves live in global memory: reading them through address-space-1 pointers gives global_load_* instead of
// flat_load_* (DLeaf::base is a `const void*`).  A FLAT load counts on lgkmcnt as well as vmcnt, so every wait for a
// scalar load -- the interpreter's next program word, the next leaf's descriptor -- also waited for the samples just
// requested: one memory round trip per operand, one after the other.
#define SO_GLOBAL_PTR(T, p) ((const T __attribute__((address_space(1)))*)(p))

// A device scalar (OP_SCALAR: the rms a Normpower's sum of squares left in its buffer one launch earlier): the same word for
// every lane and every element of the launch -- read through the SCALAR cache from a wave-uniform address in the constant
// address space (s_load_dwordx2), not as a flat load per element and lane (what `*(const double*)L.base` compiles to: a
// vector-memory instruction in front of every division, counted on both wait counters -- the dividing pass of `Normpower`
// ran at 3.2 TB/s where a multiplication by a constant runs at 5.0; tools/norm_probe.py).
__device__ __forceinline__ double scalar_leaf(const void* base) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    // (the builtin returns a signed int: without the casts a low word with its top bit set sign-extends over the high word)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const uint64_t u = ((uint64_t)hi << 32) | lo;
    return *(const double __attribute__((address_space(4)))*)(uintptr_t)u;
}

typedef double so_v2d __attribute__((ext_vector_type(2)));
typedef float so_v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double leaf_load(const DLeaf& L, int64_t n, int c) {
    int64_t f = L.df;
    if (L.sf > 0) f += n;
    else if (L.sf < 0) f -= n;
    // (cycle / mirror padding, reference src/padding.jl:132-148, is resolved on the host
    //  into one piece per wrap with sf = +1 / -1: no integer division on the device)
    const int64_t choff = ((int64_t)L.sc * c + L.dc) * L.cstride;  // uniform
    const int64_t off = (L.fstride == 1 ? f : f * L.fstride) + choff;
    if (L.dtype == SO_F32) return (double)SO_GLOBAL_PTR(float, L.base)[off];
    return SO_GLOBAL_PTR(double, L.base)[off];
}

// Compact fp64 sin/cos kernels (Taylor on |t| <= 1/4 after exact octant reduction).  The
// device library's sinpi/cos carry large-argument paths that cost ~40 VGPRs of pressure in
// every kernel that inlines the interpreter; these need ~12 and are accurate to ~1 ulp.
// fma with a CONSTANT operand held in a scalar register pair.  hipcc otherwise materialises
// every fp64 polynomial coefficient with two v_mov_b32 into the accumulator of a v_fmac (35 of
// the ~110 instructions of one sinpi evaluation, all on the vector ALU that the fp64 MFMAs of
// the resampler also need); s_mov_b32 is free by comparison.  Same operands, same rounding.
__device__ __forceinline__ double fma_addc(double a, double b, double c_const) {  // a*b + C
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_const));
    return r;
}
__device__ __forceinline__ double fma_mulc(double a, double b_const, double c) {  // a*C + c
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_const), "v"(c));
    return r;
}
__device__ __forceinline__ void sincospi_quarter(double t, double& s, double& c) {
    const double t2 = t * t;
    double ps = 7.952054001475513e-07;
    ps = fma_addc(ps, t2, -2.1915353447830217e-05);
    ps = fma_addc(ps, t2, 0.00046630280576761255);
    ps = fma_addc(ps, t2, -0.0073704309457143504);
    ps = fma_addc(ps, t2, 0.08214588661112823);
    ps = fma_addc(ps, t2, -0.5992645293207921);
    ps = fma_addc(ps, t2, 2.5501640398773455);
    ps = fma_addc(ps, t2, -5.16771278004997);
    const double t3 = t2 * t;
    s = fma_mulc(t, 3.141592653589793, fma_mulc(t, 1.2246467991473532e-16, t3 * ps));
    double pc = -1.3878952462213771e-07;
    pc = fma_addc(pc, t2, 4.303069587032947e-06);
    pc = fma_addc(pc, t2, -0.0001046381049248457);
    pc = fma_addc(pc, t2, 0.0019295743094039231);
    pc = fma_addc(pc, t2, -0.02580689139001406);
    pc = fma_addc(pc, t2, 0.2353306303588932);
    pc = fma_addc(pc, t2, -1.3352627688545895);
    pc = fma_addc(pc, t2, 4.0587121264167685);
    pc = fma_addc(pc, t2, -4.934802200544679);
    c = fma(pc, t2, 1.0);
}
// sinpi(x) with Julia's semantics: exact at integers and half-integers (src/functions.jl:57-60)
__device__ __forceinline__ double sinpi_c(double x) {
    const double k = rint(2.0 * x);
    const double t = fma(-0.5, k, x);  // exact, |t| <= 1/4
    double s, c;
    sincospi_quarter(t, s, c);
    const int q = (int)((long long)k & 3);
    const double r = (q & 1) ? c : s;
    return (q & 2) ? -r : r;
}
// sin(pi x) and cos(pi x) together (same reduction and kernels as sinpi_c)
__device__ __forceinline__ void sincospi_c(double x, double& so, double& co) {
    const double k = rint(2.0 * x);
    const double t = fma(-0.5, k, x);
    double s, c;
    sincospi_quarter(t, s, c);
    const int q = (int)((long long)k & 3);
    const double rs = (q & 1) ? c : s, rc = (q & 1) ? s : c;
    so = (q & 2) ? -rs : rs;
    co = (q == 1 || q == 2) ? -rc : rc;
}
// (sin, cos)(2 pi phase) of frames i0 + stride*lane, lane < count, of a sine generator (phase as
// in func_eval, i0 already 1-based); out of line so that its ~40 live registers do not add to
// the resampler's main loops
__device__ __attribute__((noinline)) void sine_table(int64_t i0, int stride, int count, double omega, double phi,
                                                     double fs, int has_omega, double* dst) {
    const int lane = threadIdx.x & 63;
    if (lane < count) {
        const double t = __ddiv_rn((double)(i0 + (int64_t)stride * lane), fs);
        const double ph = has_omega ? __dadd_rn(__dmul_rn(t, omega), phi) : __dadd_rn(t, phi);
        double sb, cb;
        sincospi_c(2.0 * ph, sb, cb);
        dst[2 * lane] = sb;
        dst[2 * lane + 1] = cb;
    }
}
// cos(x), x in radians, |x| < 2^20: two-term Cody-Waite reduction to x = k*pi/2 + r
__device__ __forceinline__ double cos_c(double x) {
    const double k = rint(x * 0.6366197723675814);
    double r = fma(-k, 1.5707963267948966, x);
    r = fma(-k, 6.123233995736766e-17, r);
    double s, c;
    sincospi_quarter(r * 0.3183098861837907, s, c);  // r/pi in [-1/4, 1/4]
    const int q = (int)((long long)k & 3);
    const double v = (q & 1) ? s : c;  // cos(r + k pi/2): c, -s, -c, s
    return (q == 1 || q == 2) ? -v : v;
}

// reference src/functions.jl:53-60 — every operation separately rounded (Julia does
// not contract), frame index is 1-based so the first sample is t = 1/fs
__device__ __forceinline__ double func_eval(const DLeaf& L, int64_t n) {
    double i1 = (double)((L.sf ? n : 0) + L.df + 1);
    double t = __ddiv_rn(i1, L.v2);
    if (L.flag) {
        double ph = __dadd_rn(__dmul_rn(t, L.v0), L.v1);
        if (L.mode == SO_FN_SIN) return sinpi_c(2.0 * ph);
        double a = __dmul_rn(6.283185307179586, ph - trunc(ph));  // 2π*(ph % 1.0)
        return L.mode == SO_FN_COS ? cos_c(a) : a;
    }
    double tt = __dadd_rn(t, L.v1);
    if (L.mode == SO_FN_SIN) return sinpi_c(2.0 * tt);
    return L.mode == SO_FN_COS ? cos_c(tt) : tt;
}

// reference src/ramps.jl:60-72
__device__ __forceinline__ double ramp_eval(const DLeaf& L, int64_t n) {
    int64_t n0 = (L.sf ? n : 0) + L.df;
    double x;
    if (L.flag == 0)
        x = __ddiv_rn((double)n0, L.v0);
    else
        x = __dsub_rn(1.0, __ddiv_rn((double)(n0 + 1 - L.modn), L.v0));
    return L.mode == SO_RAMP_SINRAMP ? sinpi_c(0.5 * x) : x;
}

}  // namespace so
