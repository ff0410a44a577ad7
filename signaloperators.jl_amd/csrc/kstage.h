// Staging helpers shared by the resampler kernels (k_resample.hip) and the fused resampler -> IIR kernel
// (k_rsos.hip): carrier steps on register blocks, LDS accesses the compiler must not see, LDS-DMA, counted
// vmcnt waits, in-place read-modify-write of staged chunks, the general (slow) staging path.
#pragma once
#include "kcommon.h"

namespace so {

// v = v (op) F_k  chain of a carrier on a CT x V register block (wave-uniform control flow)
// (CarT: DCarrier in memory, or StepTab in registers -- hence the fully unrolled, guarded loop:
//  register arrays must be indexed statically)
struct StepTab {
    int nsteps;
    int op[4], arg[4];
};
template <int CT, int V, bool DIV = true, typename CarT>
__device__ __forceinline__ void carrier_apply(const CarT& C, const double (&F)[kMaxFrameSlots][V],
                                              double (&val)[CT][V], bool to_f32) {
    // The opcode switch is OUTSIDE the element loops (wave-uniform branch, then CT*V straight
    // operations); the other way round the code grows by the number of cases per element --
    // with fp64 division among them, ~9000 instructions for CT=8 -- and the loader thrashes
    // the instruction cache.  DIV == false drops the division cases (fast path of the loader).
#define SO_STEP(EXPR)                                  \
    _Pragma("unroll") for (int e = 0; e < V; ++e) {    \
        const double m = mm[e];                        \
        (void)m;                                       \
        _Pragma("unroll") for (int c = 0; c < CT; ++c) { \
            const double v = val[c][e];                \
            (void)v;                                   \
            val[c][e] = (EXPR);                        \
        }                                              \
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= C.nsteps) break;
        const int op = C.op[i], arg = C.arg[i], slot = arg & 0xff;
        const bool flip = arg & 0x100, r32 = arg & 0x200;
        double mm[V];
#pragma unroll
        for (int e = 0; e < V; ++e) mm[e] = slot == 0 ? F[0][e] : slot == 1 ? F[1][e] : slot == 2 ? F[2][e] : F[3][e];
        switch (op) {
        case OP_ADD: SO_STEP(v + m) break;
        case OP_SUB:
            if (flip) { SO_STEP(m - v) } else { SO_STEP(v - m) }
            break;
        case OP_MUL: SO_STEP(v * m) break;
        case OP_DIV:
            if constexpr (DIV) {
                if (flip) { SO_STEP(m / v) } else { SO_STEP(v / m) }
            }
            break;
        case OP_NEG: SO_STEP(-v) break;
        case OP_LOADF: SO_STEP(m) break;  // generated piece: the value IS the slot
        default: break;                            // OP_ROUND32: only the rounding below
        }
        if (r32) {  // Julia Float32 arithmetic
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int e = 0; e < V; ++e) val[c][e] = (double)(float)val[c][e];
        }
    }
#undef SO_STEP
    if (to_f32) {  // the reference stores the child in the child's sample type before
                   // filtering (src/filters.jl:207,244)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < V; ++e) val[c][e] = (double)(float)val[c][e];
    }
}

// LDS accesses the compiler must not see (it cannot tell them from the LDS-DMA destinations
// in flight and would wait vmcnt(0)); the caller orders them with explicit waits.
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ v2d lds_ld16(uint32_t a) {
    v2d v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
}
template <int OFF>  // ... at a + OFF (the instruction's own offset field: no address register per row)
__device__ __forceinline__ v2d lds_ld16_off(uint32_t a) {
    v2d v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st16(uint32_t a, v2d v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory");
}
// All of v[] live in distinct registers here.  Put between the arithmetic and the ds_write_b128s
// of a read-modify-write: a vector-ALU write to the data registers of a 16-byte LDS store right
// after it is a hardware hazard the compiler only pads for instructions it can see, so the
// products must not be computed into a register pair an earlier store just read.
template <int N>
__device__ __forceinline__ void lds_pin(v2d (&v)[N]) {
    if constexpr (N == 1) asm volatile("" : "+v"(v[0])::"memory");
    else if constexpr (N == 2) asm volatile("" : "+v"(v[0]), "+v"(v[1])::"memory");
    else if constexpr (N == 4) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
    else asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(v2d (&v)[N]) {  // results of lds_ld16 are valid after this
    if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0])::"memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1])::"memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
}

// (cache-policy bits of the staging loads, e.g. -DSO_LD_NT='" nt"'.  Measured with the same hint on the fused kernel's result
//  stores, k_rsos.hip SO_ST_NT: the headline's 3.5 GB stream 0.961 -> 0.947 ms, but the same kernel on 12.5 M x 8 frames 0.457
//  -> 0.59 and K3's two-array instantiation 0.545 -> 0.60 -- what a loop of executes over 1 - 2 GB keeps in the 256 MB
//  memory-side cache is worth more than the hint.  Not set.)
#ifndef SO_LD_NT
#define SO_LD_NT ""
#endif
// 16-byte-per-lane asynchronous global -> LDS copy (global_load_lds_dwordx4): the wave
// writes 1 KiB contiguously at the wave-uniform LDS address `l`; no VGPR staging, so a few
// loader waves keep whole tiles in flight.
// Issued from inline asm on purpose: with the builtin, hipcc knows an LDS write is pending on
// the VM counter and puts s_waitcnt vmcnt(0) in front of later LDS reads it cannot prove
// disjoint (even reads of an unrelated __shared__ object were hit), which drains the ring of
// tiles in flight.  Hidden from the compiler, the only waits are the counted ones below; its
// own vmcnt(N) for ordinary loads only get stricter (in-order return), never wrong.
__device__ __forceinline__ void dma16(const void* g, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" SO_LD_NT
                 :
                 : "s"(lds_byte_addr), "v"(g)
                 : "memory");  // (m0 is reserved: hipcc keeps nothing live in it, and warns if it is listed)
}

// One chunk (up to 64 lanes x 16 bytes) of CT channel rows by LDS-DMA with NO vector-ALU
// instruction: scalar row bases + one per-lane byte offset register (global saddr form), exec
// mask built by scalar code.  While the compute waves of the SIMD run their MFMA burst a loader
// wave gets a vector-ALU issue slot only every ~64+ cycles (fp64 MFMA and VALU share the ALUs),
// so per-DMA address arithmetic on the VALU costs more than the copy itself.
#define SO_DMA_ROW(I) "s_mov_b32 m0, %[l" #I "]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[voff], %[b" #I "]" SO_LD_NT "\n\t"
template <int CT>
__device__ __forceinline__ void dma_rows(uint64_t mask, uint32_t voff, const char* base, int64_t row_stride,
                                         uint32_t lds, uint32_t lds_stride) {  // strides in bytes
    uint64_t sv;
    const char* b0 = base;
    const char* b1 = base + row_stride;
    const char* b2 = base + 2 * row_stride;
    const char* b3 = base + 3 * row_stride;
    if constexpr (CT == 1) {
        asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[mask]\n\t" SO_DMA_ROW(0) "s_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(sv)
                     : [mask] "s"(mask), [voff] "v"(voff), [b0] "s"(b0), [l0] "s"(lds)
                     : "memory");  // (m0 is reserved: hipcc keeps nothing live in it, and warns if it is listed)
    } else if constexpr (CT == 2) {
        asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[mask]\n\t" SO_DMA_ROW(0) SO_DMA_ROW(1) "s_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(sv)
                     : [mask] "s"(mask), [voff] "v"(voff), [b0] "s"(b0), [l0] "s"(lds), [b1] "s"(b1), [l1] "s"(lds + lds_stride)
                     : "memory");  // (m0 is reserved: hipcc keeps nothing live in it, and warns if it is listed)
    } else {
        asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[mask]\n\t" SO_DMA_ROW(0) SO_DMA_ROW(1) SO_DMA_ROW(2) SO_DMA_ROW(3) "s_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(sv)
                     : [mask] "s"(mask), [voff] "v"(voff), [b0] "s"(b0), [l0] "s"(lds), [b1] "s"(b1), [l1] "s"(lds + lds_stride),
                       [b2] "s"(b2), [l2] "s"(lds + 2 * lds_stride), [b3] "s"(b3), [l3] "s"(lds + 3 * lds_stride)
                     : "memory");  // (m0 is reserved: hipcc keeps nothing live in it, and warns if it is listed)
        if constexpr (CT == 8)
            dma_rows<4>(mask, voff, base + 4 * row_stride, row_stride, lds + 4 * lds_stride, lds_stride);
    }
}
#undef SO_DMA_ROW

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate).
// Rounding n DOWN is always safe (a stricter wait).
__device__ __forceinline__ void wait_vmcnt_le(int n) {
#define SO_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 0 ? 0 : (n > 24 ? 24 : n)) {
        SO_W(0) SO_W(1) SO_W(2) SO_W(3) SO_W(4) SO_W(5) SO_W(6) SO_W(7) SO_W(8) SO_W(9) SO_W(10)
        SO_W(11) SO_W(12) SO_W(13) SO_W(14) SO_W(15) SO_W(16) SO_W(17) SO_W(18) SO_W(19) SO_W(20)
        SO_W(21) SO_W(22) SO_W(23) SO_W(24)
    }
#undef SO_W
}

// In-place carrier steps on one lane's 16-byte vector of each of the CT channel rows of a
// staged fp64 tile.  LDS access in asm: a compiler-visible ds_read of an LDS-DMA destination
// could be ordered behind vmcnt(0); the caller's counted wait is the real ordering.
template <int CT, bool DIV, typename CarT>
__device__ __forceinline__ void rmw_chunk(uint32_t la, int lds_pitch, const CarT& C,
                                          const double (&F)[kMaxFrameSlots][2]) {
    if constexpr (CT > 4) {  // four rows at a time: 8 x 16 bytes in flight is 32 registers twice over
        rmw_chunk<4, DIV>(la, lds_pitch, C, F);
        rmw_chunk<CT - 4, DIV>(la + (uint32_t)(4 * lds_pitch) * 8u, lds_pitch, C, F);
        return;
    }
    v2d raw[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) raw[c] = lds_ld16(la + (uint32_t)(c * lds_pitch) * 8u);
    lds_wait(raw);
    double val[CT][2];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        val[c][0] = raw[c][0];
        val[c][1] = raw[c][1];
    }
    carrier_apply<CT, 2, DIV>(C, F, val, false);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        raw[c][0] = val[c][0];
        raw[c][1] = val[c][1];
    }
    lds_pin(raw);
#pragma unroll
    for (int c = 0; c < CT; ++c) lds_st16(la + (uint32_t)(c * lds_pitch) * 8u, raw[c]);
}

// The commonest in-place steps, ONE multiply / add / subtract with frame slot 0 (`Amplify` or `Mix`
// with a generator, a ramp or a number), without the step interpreter: per chunk of CT rows one
// slot read, CT reads, 2*CT arithmetic instructions, CT writes.  (The general rmw_chunk spends ~4x
// the vector instructions on slot selection and step dispatch, and every one of them waits for
// a gap between the MFMAs.)  OP: 0 v*m, 1 v+m, 2 v-m, 3 m-v.
template <int CT, int OP>
__device__ __forceinline__ void rmw_one_chunk(uint32_t la, int lds_pitch, uint32_t fa) {
    v2d f[1];
    f[0] = lds_ld16(fa);
    v2d raw[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) raw[c] = lds_ld16(la + (uint32_t)(c * lds_pitch) * 8u);
    lds_wait(f);
    lds_wait(raw);
#pragma unroll
    for (int c = 0; c < CT; ++c)
        raw[c] = OP == 0 ? raw[c] * f[0] : OP == 1 ? raw[c] + f[0] : OP == 2 ? raw[c] - f[0] : f[0] - raw[c];
    lds_pin(raw);
#pragma unroll
    for (int c = 0; c < CT; ++c) lds_st16(la + (uint32_t)(c * lds_pitch) * 8u, raw[c]);
}

// ... and the in-place step whose operand is a second ARRAY's samples, staged by LDS-DMA as rows of 1 KB at `sa` (this lane's
// 16 bytes of row c at sa + 1024 c: k_resample_periodic A2).  OP: 0 v*m, 1 v+m, 2 v-m, 3 m-v -- the operations K1 would have
// done on the way to a materialised sum, on the same values.
// (T = float: the tile and the staging rows hold Float32 samples, four to the 16 bytes, and the operation is Float32's --
//  Julia's `+` / `*` on Float32 operands, what K1's materialised map computes)
typedef float v4f_ __attribute__((ext_vector_type(4)));
template <typename T, int CT, int OP, int R0 = 0>
__device__ __forceinline__ void rmw_arr2(uint32_t la, int lds_pitch, uint32_t sa) {
    if constexpr (CT > 4) {  // four rows at a time (registers)
        rmw_arr2<T, 4, OP, R0>(la, lds_pitch, sa);
        rmw_arr2<T, CT - 4, OP, R0 + 4>(la + (uint32_t)(4 * lds_pitch) * (uint32_t)sizeof(T), lds_pitch, sa);
        return;
    } else {
        v2d raw[CT], m[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) raw[c] = lds_ld16(la + (uint32_t)(c * lds_pitch) * (uint32_t)sizeof(T));
        m[0] = lds_ld16_off<R0 * 1024>(sa);
        if constexpr (CT > 1) m[1] = lds_ld16_off<(R0 + 1) * 1024>(sa);
        if constexpr (CT > 2) m[2] = lds_ld16_off<(R0 + 2) * 1024>(sa);
        if constexpr (CT > 3) m[3] = lds_ld16_off<(R0 + 3) * 1024>(sa);
        lds_wait(raw);
        lds_wait(m);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if constexpr (sizeof(T) == 8) {
                raw[c] = OP == 0 ? raw[c] * m[c] : OP == 1 ? raw[c] + m[c] : OP == 2 ? raw[c] - m[c] : m[c] - raw[c];
            } else {
                const v4f_ a = __builtin_bit_cast(v4f_, raw[c]), b = __builtin_bit_cast(v4f_, m[c]);
                const v4f_ r = OP == 0 ? a * b : OP == 1 ? a + b : OP == 2 ? a - b : b - a;
                raw[c] = __builtin_bit_cast(v2d, r);
            }
        }
        lds_pin(raw);
#pragma unroll
        for (int c = 0; c < CT; ++c) lds_st16(la + (uint32_t)(c * lds_pitch) * (uint32_t)sizeof(T), raw[c]);
    }
}

// The one step of a carrier whose operand is a SECOND array (DCarrier::base2, arg bit kCarArr2): `v (op) y[c][n]` on a
// CT x V register block of frames g0 (+ e where `vec`) -- the general path's form (scalar loads, row by row: the block of
// the first array's samples is already in registers, a second one next to it would not fit the 128 the kernel has).
template <int CT, int V, typename CarT>
__device__ __forceinline__ void carrier_arr2(const CarT& C, int c0, int64_t g0, bool vec, double (&val)[CT][V]) {
    const int op = C.op[0];
    const bool flip = C.arg[0] & 0x100, r32 = C.arg[0] & 0x200;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        double b[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int64_t off = (int64_t)(c0 + c) * C.cstride2 + g0 + (vec ? e : 0) + C.df2;
            b[e] = C.dtype2 == SO_F32 ? (double)SO_GLOBAL_PTR(float, C.base2)[off] : SO_GLOBAL_PTR(double, C.base2)[off];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const double v = val[c][e], m = b[e];
            double r = op == OP_ADD ? v + m : op == OP_SUB ? (flip ? m - v : v - m) : v * m;
            if (r32) r = (double)(float)r;
            val[c][e] = r;
        }
    }
}

// Slow path of the staging (tile edges, f32 sources, generated pieces, unaligned arrays): one
// 16-byte vector per lane, load -> carrier steps -> LDS store, synchronously.
template <typename T, int CT, bool A2 = false>
__device__ __forceinline__ void stage_generic_impl(int64_t n_in, int lds_pitch,
                                                        const DCarrier* __restrict__ car, int ncar,
                                                        const DOp* __restrict__ ops,
                                                        const DLeaf* __restrict__ leaves,
                                                        int64_t gi, int iv, int ci, int c0,
                                                        T* __restrict__ buf) {
    struct { int64_t n_in; int lds_pitch; } g{n_in, lds_pitch};
    constexpr int V = 16 / sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(V)));
    int cj = ci;
    while (cj + 1 < ncar && car[cj].b <= gi) ++cj;  // mostly 0 iterations
    // (a carrier without an array -- base == nullptr -- is a purely generated piece)
    const bool vec = gi >= car[cj].a && gi + V <= car[cj].b && gi >= 0 && gi + V <= g.n_in &&
                     (car[cj].base == nullptr ||
                      (car[cj].vec_ok && (((gi + car[cj].df) % V) == 0) &&
                       car[cj].dtype == (sizeof(T) == 4 ? SO_F32 : SO_F64)));
    const int nsub = vec ? 1 : V;
#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub) {
        const int64_t g0 = gi + sub;  // first (vec) or only (scalar) frame of this pass
        int ck = cj;
        while (ck + 1 < ncar && car[ck].b <= g0) ++ck;
        const DCarrier& C = car[ck];
        const bool ok = vec || (g0 >= 0 && g0 < g.n_in && g0 >= C.a && g0 < C.b);
        // ---- per-frame values first (keeps the interpreter's registers dead while
        //      the CT loads are in flight) ----
        const bool steps = ok && C.nsteps > 0;
        double F[kMaxFrameSlots][V];
#pragma unroll
        for (int k = 0; k < kMaxFrameSlots; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) F[k][e] = 0.0;
        if (steps && C.frame_len > 0) {
            int64_t nn[V];
            double fo[V];
#pragma unroll
            for (int e = 0; e < V; ++e) nn[e] = vec ? g0 + e : g0;
            run_program<V, false, 2, true>(ops, C.frame_pc, C.frame_len, leaves, nn, c0, F, fo);
        }
        // ---- loads (CT independent loads in flight) ----
        double val[CT][V];
        if (C.base == nullptr) {
            const double fill = C.pad_ == 2 ? 1.0 : 0.0;  // GA: a generated piece that is the gain itself
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int e = 0; e < V; ++e) val[c][e] = ok ? fill : 0.0;
        } else if (vec) {
            // (global_load, not flat_load: a FLAT load also counts on lgkmcnt, which the LDS stores below and the LDS-DMA
            //  ring wait on -- kleaf.h SO_GLOBAL_PTR)
            const T __attribute__((address_space(1)))* xp = SO_GLOBAL_PTR(T, C.base) + (int64_t)c0 * C.cstride + g0 + C.df;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const vecT v = *(const vecT __attribute__((address_space(1)))*)(xp + (int64_t)c * C.cstride);
#pragma unroll
                for (int e = 0; e < V; ++e) val[c][e] = (double)v[e];
            }
        } else {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                double xv = 0.0;
                if (ok) {
                    const int64_t off = (int64_t)(c0 + c) * C.cstride + g0 + C.df;
                    xv = C.dtype == SO_F32 ? (double)SO_GLOBAL_PTR(float, C.base)[off]
                                           : SO_GLOBAL_PTR(double, C.base)[off];
                }
#pragma unroll
                for (int e = 0; e < V; ++e) val[c][e] = xv;
            }
        }
        // ---- steps ----
        bool two_arrays = false;
        if constexpr (A2) two_arrays = steps && C.nsteps == 1 && (C.arg[0] & kCarArr2) && C.base != nullptr;
        if (two_arrays) carrier_arr2<CT, V>(C, c0, g0, vec, val);
        else if (steps) carrier_apply<CT, V>(C, F, val, sizeof(T) == 4);
        // ---- LDS stores ----
        if (vec) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int e = 0; e < V; ++e) buf[c * g.lds_pitch + iv * V + e] = (T)val[c][e];
        } else {
#pragma unroll
            for (int c = 0; c < CT; ++c) buf[c * g.lds_pitch + iv * V + sub] = (T)val[c][0];
        }
    }
}

// fp32 sources (V = 4 elements per vector) need more registers than the 128 the 16-wave kernel
// has: out of line for them, so that the spills stay inside the callee -- a scratch reload in
// the loader's loop waits vmcnt(0) and drains every LDS-DMA in flight.  The fp64 kernels inline
// it and use no scratch at all (a kernel with a scratch frame costs ~0.1 ms per launch here).
template <typename T, int CT>
__device__ __attribute__((noinline)) void stage_generic_ool(int64_t n_in, int lds_pitch,
                                                            const DCarrier* __restrict__ car, int ncar,
                                                            const DOp* __restrict__ ops,
                                                            const DLeaf* __restrict__ leaves,
                                                            int64_t gi, int iv, int ci, int c0,
                                                            T* __restrict__ buf) {
    stage_generic_impl<T, CT>(n_in, lds_pitch, car, ncar, ops, leaves, gi, iv, ci, c0, buf);
}
template <typename T, int CT>
__device__ __forceinline__ void stage_generic(int64_t n_in, int lds_pitch,
                                              const DCarrier* __restrict__ car, int ncar,
                                              const DOp* __restrict__ ops,
                                              const DLeaf* __restrict__ leaves, int64_t gi, int iv,
                                              int ci, int c0, T* __restrict__ buf) {
    if constexpr (sizeof(T) == 8) stage_generic_impl<T, CT>(n_in, lds_pitch, car, ncar, ops, leaves, gi, iv, ci, c0, buf);
    else stage_generic_ool<T, CT>(n_in, lds_pitch, car, ncar, ops, leaves, gi, iv, ci, c0, buf);
}

}  // namespace so
