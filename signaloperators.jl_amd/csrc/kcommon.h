// Device helpers shared by the kernel translation units (k_pointwise.hip, k_sos.hip, k_resample.hip):
// leaf evaluators, the compact sin/cos kernels, the register stack machine of the fused pointwise
// programs.  Split out of kernels.hip in round 3: one 3 000-line translation unit took 2.5 minutes to
// compile and its 6 MB code object was loaded whole by a process's first launch (8 ms before a
// one-shot sink of a 5 s sine could start); three units compile side by side and load on demand.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "../../include/sigops.h"
#include "kernels.h"
#include "sigops_internal.h"

#include "kleaf.h"

namespace so {

// One per-frame slot in closed form (DCarrier::slot_kind): the same arithmetic as the
// interpreter's OP_CONST / OP_SCALAR / OP_FUNC / OP_RAMP, without the stack machine.
__device__ __forceinline__ double slot_eval(int kind, const DLeaf& L, int64_t n) {
    double v;
    switch (kind & 0xff) {
    case OP_CONST: v = L.v0; break;
    case OP_SCALAR: v = scalar_leaf(L.base); break;
    case OP_FUNC: v = func_eval(L, n); break;
    default: v = ramp_eval(L, n); break;
    }
    return (kind & 0x100) ? (double)(float)v : v;
}

// A leaf read from LDS sits in vector registers; its mode/flag fields then look lane-varying
// to the compiler and every `if (L.flag)` / `L.mode == ...` becomes compute-both-and-select
// (all three trig kernels per frame).  Passing the fields through readfirstlane makes the
// branches scalar again.
__device__ __forceinline__ int64_t rfl64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double rfl_f64(double v) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ DLeaf leaf_uniform(const DLeaf& L) {
    DLeaf U = L;
    const uint64_t b = (uint64_t)(uintptr_t)L.base;
    U.base = (const void*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b));  // (the builtin returns a signed int)
    U.v0 = rfl_f64(L.v0);
    U.v1 = rfl_f64(L.v1);
    U.v2 = rfl_f64(L.v2);
    U.df = (int64_t)__builtin_bit_cast(uint64_t, rfl_f64(__builtin_bit_cast(double, (uint64_t)L.df)));
    U.modn = (int64_t)__builtin_bit_cast(uint64_t, rfl_f64(__builtin_bit_cast(double, (uint64_t)L.modn)));
    U.sf = __builtin_amdgcn_readfirstlane(L.sf);
    U.mode = __builtin_amdgcn_readfirstlane(L.mode);
    U.flag = __builtin_amdgcn_readfirstlane(L.flag);
    return U;
}

// frames n and n+1 at once
__device__ __forceinline__ void slot_eval2(int kind, const DLeaf& L, int64_t n, double& o0, double& o1) {
    double v0, v1;
    switch (kind & 0xff) {
    case OP_CONST: v0 = v1 = L.v0; break;
    case OP_SCALAR: v0 = v1 = scalar_leaf(L.base); break;
    case OP_FUNC:
        v0 = func_eval(L, n);
        v1 = func_eval(L, n + 1);
        break;
    default:
        v0 = ramp_eval(L, n);
        v1 = ramp_eval(L, n + 1);
        break;
    }
    if (kind & 0x100) {
        v0 = (double)(float)v0;
        v1 = (double)(float)v1;
    }
    o0 = v0;
    o1 = v1;
}

// ---------------------------------------------------------------------------
// D-deep register stack machine over E elements per thread.  Program words are
// wave-uniform (scalar loads); the stack lives in VGPRs (static indexing only).
// D is 2 for left-fold chains (almost every tree) and kStackDepth otherwise, which keeps
// the register footprint of the common case small enough for high occupancy.
//   CV == false: element e is frame n[e] of channel c (K1: E frames per thread).
//   CV == true : element e is channel c+e of the single frame n[0] (stage-kernel tile
//                staging: all channels of a frame at once -> E independent loads in flight).
#define SO_PUSH(expr)                                   \
    _Pragma("unroll") for (int e = 0; e < E; ++e) {     \
        _Pragma("unroll") for (int d = D - 1; d > 0; --d) st[d][e] = st[d - 1][e]; \
        st[0][e] = (expr);                              \
    }
#define SO_POP1()                                       \
    _Pragma("unroll") for (int d = 1; d < D - 1; ++d) st[d][e] = st[d + 1][e];
#define SO_BIN(opr)                                     \
    _Pragma("unroll") for (int e = 0; e < E; ++e) {     \
        st[0][e] = st[1][e] opr st[0][e];               \
        SO_POP1()                                       \
    }

// HEAVY == false drops the generator/ramp opcodes (the planner always hoists them into the
// per-frame program), so the per-sample interpreter carries no transcendental code.
// PAIR (E == 2, n[1] == n[0] + 1 for every lane, same parity of n[0] across the wave): array
// leaves with unit frame stride are read with one 16-byte (fp64) / 8-byte (fp32) load per lane
// when the pair is naturally aligned -- the widest, best-coalesced form of a streaming read.
template <int E, bool CV, int D, bool HEAVY, bool PAIR = false>
__device__ __forceinline__ void run_program(const DOp* __restrict__ ops, int pc, int len,
                                            const DLeaf* __restrict__ leaves,
                                            const int64_t (&n)[CV ? 1 : E], int c,
                                            double (&F)[kMaxFrameSlots][CV ? 1 : E],
                                            double (&out)[E], bool pair_rt = true) {
    double st[D][E];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int e = 0; e < E; ++e) st[d][e] = 0.0;
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
            if constexpr (PAIR && E == 2 && !CV) {
                if (pair_rt && L.sf > 0 && L.fstride == 1) {  // wave-uniform
                    const int64_t choff = ((int64_t)L.sc * c + L.dc) * L.cstride;
                    const int64_t off = n[0] + L.df + choff;  // element index of the pair's first frame
                    const int par = __builtin_amdgcn_readfirstlane((int)off) & 1;
                    double v0, v1;
                    // (global_load_*, and no flag between the load and its push: the samples are waited for where the
                    //  next operation needs them, not here)
                    if (L.dtype == SO_F64) {
                        const double __attribute__((address_space(1)))* pa = SO_GLOBAL_PTR(double, L.base) + off;
                        if (((((uintptr_t)L.base) >> 3) & 1) == (uintptr_t)par && ((uintptr_t)L.base & 7) == 0) {
                            const so_v2d v = *(const so_v2d __attribute__((address_space(1)))*)pa;
                            v0 = v.x;
                            v1 = v.y;
                        } else {
                            v0 = pa[0];
                            v1 = pa[1];
                        }
                    } else {
                        const float __attribute__((address_space(1)))* pa = SO_GLOBAL_PTR(float, L.base) + off;
                        if (((((uintptr_t)L.base) >> 2) & 1) == (uintptr_t)par && ((uintptr_t)L.base & 3) == 0) {
                            const so_v2f v = *(const so_v2f __attribute__((address_space(1)))*)pa;
                            v0 = (double)v.x;
                            v1 = (double)v.y;
                        } else {
                            v0 = (double)pa[0];
                            v1 = (double)pa[1];
                        }
                    }
#pragma unroll
                    for (int d = D - 1; d > 0; --d) {
                        st[d][0] = st[d - 1][0];
                        st[d][1] = st[d - 1][1];
                    }
                    st[0][0] = v0;
                    st[0][1] = v1;
                    break;
                }
            }
            SO_PUSH(leaf_load(L, n[CV ? 0 : e], CV ? c + e : c));
            break;
        }
        case OP_SCALAR: {
            // (flag: the launch that computed the scalar also left it in this leaf's v0 -- RmsPatch; control-block copies of a
            //  leaf never carry the flag, Plan::make_ctl)
            const DLeaf& L = leaves[op.arg];
            const double v = L.flag ? L.v0 : scalar_leaf(L.base);
            SO_PUSH(v);
            break;
        }
        case OP_FUNC:
            if constexpr (HEAVY) {
                const DLeaf& L = leaves[op.arg];
                SO_PUSH(func_eval(L, n[CV ? 0 : e]));
            }
            break;
        case OP_RAMP:
            if constexpr (HEAVY) {
                const DLeaf& L = leaves[op.arg];
                SO_PUSH(ramp_eval(L, n[CV ? 0 : e]));
            }
            break;
        case OP_ADD: SO_BIN(+); break;
        case OP_SUB: SO_BIN(-); break;
        case OP_MUL: SO_BIN(*); break;
        case OP_DIV: SO_BIN(/); break;
        case OP_NEG:
#pragma unroll
            for (int e = 0; e < E; ++e) st[0][e] = -st[0][e];
            break;
        case OP_ROUND32:
#pragma unroll
            for (int e = 0; e < E; ++e) st[0][e] = (double)(float)st[0][e];
            break;
        case OP_STOREF:
#pragma unroll
            for (int e = 0; e < E; ++e) {
                switch (op.arg) {
                case 0: F[0][CV ? 0 : e] = st[0][e]; break;
                case 1: F[1][CV ? 0 : e] = st[0][e]; break;
                case 2: F[2][CV ? 0 : e] = st[0][e]; break;
                default: F[3][CV ? 0 : e] = st[0][e]; break;
                }
                st[0][e] = st[1][e];
                SO_POP1()
            }
            break;
        case OP_LOADF:
            switch (op.arg) {
            case 0: SO_PUSH(F[0][CV ? 0 : e]); break;
            case 1: SO_PUSH(F[1][CV ? 0 : e]); break;
            case 2: SO_PUSH(F[2][CV ? 0 : e]); break;
            default: SO_PUSH(F[3][CV ? 0 : e]); break;
            }
            break;
        default: break;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] = st[0][e];
}

// hipFuncSetAttribute is per device: remember it per (kernel, device) -- a process that drives
// several GPUs must raise the dynamic-LDS limit on each of them.
static inline bool first_use_on_device(bool (&seen)[64]) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 ? 0 : (dev > 63 ? 63 : dev);
    const bool first = !seen[dev];
    seen[dev] = true;
    return first;
}

}  // namespace so
