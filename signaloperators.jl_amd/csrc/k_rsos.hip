// Hand-written HIP kernel for gfx950 (MI355X, CDNA4; wave64).  No CUDA shims, no dual paths.
// K5 k_rsos: periodic polyphase resampler and the SOS IIR that consumes it, in ONE pass over HBM.
//
// Reference: `ToFramerate(Filt(x))` is rewritten to `Filt(ToFramerate(x))` (src/filters.jl:143-148) and the IIR's
// nextblock pulls its resampling child block by block and filters the block in place (src/filters.jl:240-255): the
// resampled signal never exists as a whole.  K3 + K2 wrote it to HBM, read it twice and wrote the result (9.2 GB
// for 3.5 GB on the north-star pipeline); here it lives in MFMA accumulators and a few KB of LDS.
//
// Geometry.  A workgroup (one per CU, persistent) owns a *sequence group*: 16 rows = rgs time ranges x ct channels
// of the signal, and walks all of them together, block by block (16 outputs per row and block).  The recurrence
// only needs 16 independent sequences because it runs in block state-space form on the matrix cores:
//     [ y ]   [ T   C    ] [ x ]      x: the block's 16 resampled samples, s: the cascade's 2*nsec DF2T states,
//     [ s']   [ D   A^16 ] [ s ]      T: 16 x 16 lower-triangular Toeplitz of the impulse response
// (entries from the DF2T recurrence itself, host-built).  Per block and 16 rows:
//     y waves   X  = Tap_g^T . Win        ks MFMAs    (A: taps of phase group g from LDS, B: the rows' input windows
//                                                      from the LDS ring; D[t][row] IS the B operand of what follows)
//     chain     S' = D . X + A^16 . S     4 + 3 MFMAs (ONE wave carries the state of all 16 rows in registers; the
//                                                      only serial dependency of the kernel is its 3 MFMAs per block)
//     y waves   Y  = X^T . T^T + S^T . C^T  4 + 3 MFMAs  (D[row][t]: 16 lanes store one 128-byte line)
// with v_mfma_f64_16x16x4_f64: A[l&15][k=l>>4], B[k=l>>4][l&15], D: col = l&15, row = (l>>4) + 4*reg -- so an
// accumulator register v is at once k-step v of a B operand (k = time or state, col = row of the group) and, read
// as an A operand, k-step v of the transposed product.  No transposes, no shuffles.
// Waves: 0 chain | 4, 8, 12 loaders (LDS-DMA of 128-frame row chunks into a ring, fused `Mix`/`Amplify` source
// applied in place) | the others y waves, blocks dealt round robin.  Waves meet through LDS sequence counters
// (single writer each), never through barriers: the chain wave must not wait for anybody but its x block.
// A range starts wp periods before its first output from zero state (||A^(wp L)|| < 2^-70: the planner's warm
// start) and stores nothing there.
#include "kcommon.h"
#include "kstage.h"

// One translation unit per window length, result type and workgroup size (build.py: -DSO_RSOS_ONLY_KS=4 | 12 | 13 | 14 |
// 16 | 20 with -DSO_RSOS_ONLY_F32=0 | 1 and -DSO_RSOS_ONLY_NW=12 | 16 | 8, and -DSO_RSOS_ONLY_KS=0 for the dispatcher):
// the runtime loads a code object the first time one of ITS kernels is launched, and all instantiations in one object
// were 8 ms of a first sink's 9 ms execute; the units also compile side by side.  -1 (the default, tools/build_variant.sh):
// everything in this one.
#ifndef SO_RSOS_ONLY_KS
#define SO_RSOS_ONLY_KS -1
#endif
#ifndef SO_RSOS_ONLY_F32
#define SO_RSOS_ONLY_F32 0
#endif
#ifndef SO_RSOS_ONLY_NW
#define SO_RSOS_ONLY_NW 0  // (0: every workgroup size in this unit)
#endif

namespace so {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define SO_LDS __attribute__((address_space(3)))
// LDS sequence counters (single writer each) and the data they announce: VOLATILE accesses -- the compiler keeps them in
// program order among themselves and never caches or speculates them, and the LDS executes a wave's instructions in the
// order they were issued, so a counter written after its data (or after the reads whose completion it announces) needs
// no wait in front of it.  Not inline asm: behind an asm LDS instruction the compiler no longer knows how many
// operations are outstanding and every later wait becomes lgkmcnt(0) -- the chain wave then waited for its own
// just-issued stores in every block.
__device__ __forceinline__ int flag_ld(uint32_t a) { return *(const volatile SO_LDS int*)(uintptr_t)a; }
__device__ __forceinline__ void flag_st(uint32_t a, int v) { *(volatile SO_LDS int*)(uintptr_t)a = v; }
// every wait of this kernel is for another wave of the same workgroup and lasts microseconds: a wait that does not end
// is a bug of the protocol.  The wave that finds one says so in the plan's host-mapped error word (RsSos::err) and ENDS; the
// waves that wait for it in turn do the same, the launch finishes with a wrong result, and the host reports
// "k_rsos: a wait between its waves did not end" with the next call on the plan (executor.cpp) -- an error return, not the
// dead process a trap is (the ROCm runtime aborts on one) and not a hung device.  Without the word: a trap.
// (SIGOPS_RSOS_DEBUG bit 32768 lowers the limit of the y waves' wait for a state to 4 096 polls: with bit 64 -- no chain wave --
//  that wait never ends, and the test of the host's report need not hold a GPU for seconds)
// (inline and without a call: as an out-of-line routine -- even a noreturn one -- the cold path cost the headline 1 %)
#define SO_SPIN_PAUSE(spins, sleep, limit)                         \
    do {                                                           \
        if (++(spins) > (limit)) rsos_wait_failed(sh);             \
        if ((sleep) == 1) __builtin_amdgcn_s_sleep(1);             \
        else __builtin_amdgcn_s_sleep(2);                          \
    } while (0)
__device__ __forceinline__ int wave_min(int v, int n) {  // min of lanes [0, n), wave-uniform result
    int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
    for (int i = 1; i < 12; ++i)
        if (i < n) m = min(m, __builtin_amdgcn_readlane(v, i));
    return m;
}
// s_waitcnt vmcnt(n), wave-uniform runtime n up to 60 (rounded down: a stricter wait)
__device__ __forceinline__ void wait_vmcnt_le60(int n) {
#define SO_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    if (n < 8) {
        switch (n < 0 ? 0 : n) { SO_W(0) SO_W(1) SO_W(2) SO_W(3) SO_W(4) SO_W(5) SO_W(6) SO_W(7) }
    } else {
        switch (n > 60 ? 15 : n >> 2) {
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
            case 13: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
            case 14: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
        }
    }
#undef SO_W
}

// (sin, cos)(2 pi phase) of a sine generator at 1-based frame i1 (phase as in func_eval; every operation rounded
// on its own); out of line: ~40 live registers that must not add to the loader's loop
__device__ __attribute__((noinline)) double2 rsos_sine_at(int64_t i1, double omega, double phi, double fs, int has_omega) {
    const double t = __ddiv_rn((double)i1, fs);
    const double ph = has_omega ? __dadd_rn(__dmul_rn(t, omega), phi) : __dadd_rn(t, phi);
    double sb, cb;
    sincospi_c(2.0 * ph, sb, cb);
    return double2{sb, cb};
}

// general staging of one chunk of one unit (RU rows): edges of the signal and of carrier 0, Float32 arrays,
// generated pieces, carriers with several steps -- everything the LDS-DMA fast path does not take
template <int RU>
__device__ __attribute__((noinline)) void rsos_stage_slow(int64_t n_in, int rpitch, const DCarrier* gcar, int ncar, const DOp* gops,
                                                          const DLeaf* gleaves, int64_t n0, int lanes, int ch0, double* buf) {
    const int lane = threadIdx.x & 63;
    if (lane < lanes) stage_generic_impl<double, RU, true>(n_in, rpitch, gcar, ncar, gops, gleaves, n0 + 2 * lane, lane, 0, ch0, buf);  // (true: a second array's step too)
}
// ... into a ring that keeps Float32 samples (RsSos::ring32): four frames per lane, rows 2 rpitch floats apart
template <int RU>
__device__ __attribute__((noinline)) void rsos_stage_slow32(int64_t n_in, int rpitch, const DCarrier* gcar, int ncar, const DOp* gops,
                                                            const DLeaf* gleaves, int64_t n0, int lanes, int ch0, float* buf) {
    const int lane = threadIdx.x & 63;
    if (lane < lanes) stage_generic_impl<float, RU>(n_in, 2 * rpitch, gcar, ncar, gops, gleaves, n0 + 4 * lane, lane, 0, ch0, buf);
}

// in-place step of carrier 0 on one lane's 16 bytes of each of the RU rows of a landed chunk; OP: 0 v*m, 1 v+m, 2 v-m, 3 m-v
template <int RU, int OP>
__device__ __forceinline__ void rsos_rmw(uint32_t la, uint32_t row_bytes, v2d gn) {
    if constexpr (RU > 4) {
        rsos_rmw<4, OP>(la, row_bytes, gn);
        rsos_rmw<RU - 4, OP>(la + 4 * row_bytes, row_bytes, gn);
        return;
    } else {
        v2d raw[RU];
#pragma unroll
        for (int c = 0; c < RU; ++c) raw[c] = lds_ld16(la + (uint32_t)c * row_bytes);
        lds_wait(raw);
#pragma unroll
        for (int c = 0; c < RU; ++c) raw[c] = OP == 0 ? raw[c] * gn : OP == 1 ? raw[c] + gn : OP == 2 ? raw[c] - gn : gn - raw[c];
        lds_pin(raw);
#pragma unroll
        for (int c = 0; c < RU; ++c) lds_st16(la + (uint32_t)c * row_bytes, raw[c]);
    }
}

// v += m on one lane's two frames of each of the RU rows of a landed chunk with the LDS's own adder (ds_add_f64: an
// IEEE double add like v_add_f64): nothing for the vector ALU, which the chain wave's MFMAs keep busy
template <int RU>
__device__ __forceinline__ void rsos_add(uint32_t la, uint32_t row_bytes, v2d gn) {
    const double g0 = gn[0], g1 = gn[1];
#pragma unroll
    for (int c = 0; c < RU; ++c) {
        const uint32_t a = la + (uint32_t)c * row_bytes;
        asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:8" ::"v"(a), "v"(g0), "v"(g1) : "memory");
    }
}

// ... with the rows' distance known at compile time (RB bytes; the ring geometries of the common shapes): one address register
// and immediate offsets -- no vector instruction per row (every one of them waits for a gap in the MFMA stream of the
// chain wave and the y wave this wave shares its SIMD with: ~30 of them per chunk were 0.13 ms of the headline's 1.08)
template <int RB, int C, int RU>
__device__ __forceinline__ void rsos_add_imm(uint32_t la, double g0, double g1) {
    if constexpr (C < RU) {
        asm volatile("ds_add_f64 %0, %1 offset:%3\n\tds_add_f64 %0, %2 offset:%4" ::"v"(la), "v"(g0), "v"(g1), "n"(C * RB), "n"(C * RB + 8) : "memory");
        rsos_add_imm<RB, C + 1, RU>(la, g0, g1);
    }
}

// ... for the groups of fewer than eight channels (four, eight or sixteen units per chunk): rows [R, R + N) of the group from
// two address registers (rows 0 .. 7 and 8 .. 15: the immediate holds 16 bits) ...
template <int RB, int R, int N>
__device__ __forceinline__ void rsos_add_rows(uint32_t la0, uint32_t la8, double g0, double g1) {
    if constexpr (N > 0) {
        asm volatile("ds_add_f64 %0, %1 offset:%3\n\tds_add_f64 %0, %2 offset:%4" ::"v"(R < 8 ? la0 : la8), "v"(g0), "v"(g1), "n"((R & 7) * RB), "n"((R & 7) * RB + 8) : "memory");
        rsos_add_rows<RB, R + 1, N - 1>(la0, la8, g0, g1);
    }
}
// ... and ALL units of this loader wave (u = q + J NL; q RU rows are in la0 / la8), every one of them on the fast path: the
// operand from the unit's share base and the lane's own (sin, cos) pairs, the adds -- four vector instructions per unit, none
// per row (unit by unit with its own address arithmetic and per-row adds the step cost a four-channel group 0.31 ms where an
// eight-channel one pays 0.02)
template <int RB, int RU, int NL, int J, int MUC>
__device__ __forceinline__ void rsos_add_units(uint32_t la0, uint32_t la8, const double2 (&bs)[MUC], double2 d0, double2 d1, bool sine, double gc, bool neg) {
    if constexpr (J < MUC) {
        double g0 = gc, g1 = gc;
        if (sine) {
            g0 = fma(bs[J].x, d0.y, bs[J].y * d0.x);
            g1 = fma(bs[J].x, d1.y, bs[J].y * d1.x);
        }
        if (neg) {
            g0 = -g0;
            g1 = -g1;
        }
        rsos_add_rows<RB, J * NL * RU, RU>(la0, la8, g0, g1);
        rsos_add_units<RB, RU, NL, J + 1, MUC>(la0, la8, bs, d0, d1, sine, gc, neg);
    }
}

// The fused step on a ring that KEEPS Float32 samples (RsSos::f32m: a Float32 result resampled on the Float32 MFMA).  Lane l owns
// frames 2 l and 2 l + 1 of every row of the landed chunk (four bytes a frame: one 8-byte LDS access per lane and row): read,
// one Float32 operation per frame, written back.  (The LDS's own Float32 adder -- ds_add_f32, as the Float64 path uses
// ds_add_f64 -- made the step cost 1.07 ms of 1.88, with either lane map: measured, not used.)  The operand is the step's
// Float64 value rounded to Float32 once: the samples that enter the resampler are Float32 on this path.
// OP: 0 v*m, 1 v+m (v-m: with -m), 3 m-v
template <int RU, int OP>
__device__ __forceinline__ void rsos_step32(uint32_t la, uint32_t row_bytes, float g0, float g1) {
    if constexpr (RU > 4) {
        rsos_step32<4, OP>(la, row_bytes, g0, g1);
        rsos_step32<RU - 4, OP>(la + 4 * row_bytes, row_bytes, g0, g1);
    } else {
        float2 raw[RU];
#pragma unroll
        for (int c = 0; c < RU; ++c) asm volatile("ds_read_b64 %0, %1" : "=v"(raw[c]) : "v"(la + (uint32_t)c * row_bytes) : "memory");
        if constexpr (RU == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3])::"memory");
        else if constexpr (RU == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0])::"memory");
#pragma unroll
        for (int c = 0; c < RU; ++c) {
            raw[c].x = OP == 0 ? raw[c].x * g0 : OP == 1 ? raw[c].x + g0 : g0 - raw[c].x;
            raw[c].y = OP == 0 ? raw[c].y * g1 : OP == 1 ? raw[c].y + g1 : g1 - raw[c].y;
        }
#pragma unroll
        for (int c = 0; c < RU; ++c) asm volatile("ds_write_b64 %0, %1" ::"v"(la + (uint32_t)c * row_bytes), "v"(raw[c]) : "memory");
    }
}

// ... with the rows' distance known at compile time (RB bytes): one address register, immediate offsets, all RU rows' reads in
// flight behind ONE wait (per row an address add and per four rows a wait were half of the step's instructions)
template <int RB, int RU, int OP>
__device__ __forceinline__ void rsos_step32_imm(uint32_t la, float g0, float g1) {
    float2 raw[RU];
#define SO_RD(C) if constexpr (C < RU) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(raw[C < RU ? C : 0]) : "v"(la), "n"((C < RU ? C : 0) * RB) : "memory");
    SO_RD(0) SO_RD(1) SO_RD(2) SO_RD(3) SO_RD(4) SO_RD(5) SO_RD(6) SO_RD(7)
#undef SO_RD
    if constexpr (RU == 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7])::"memory");
    else if constexpr (RU == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3])::"memory");
    else if constexpr (RU == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1])::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0])::"memory");
#pragma unroll
    for (int c = 0; c < RU; ++c) {
        raw[c].x = OP == 0 ? raw[c].x * g0 : OP == 1 ? raw[c].x + g0 : g0 - raw[c].x;
        raw[c].y = OP == 0 ? raw[c].y * g1 : OP == 1 ? raw[c].y + g1 : g1 - raw[c].y;
    }
#define SO_WR(C) if constexpr (C < RU) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(la), "v"(raw[C < RU ? C : 0]), "n"((C < RU ? C : 0) * RB) : "memory");
    SO_WR(0) SO_WR(1) SO_WR(2) SO_WR(3) SO_WR(4) SO_WR(5) SO_WR(6) SO_WR(7)
#undef SO_WR
}

// A landed chunk of a Float32 array: 128 floats in the upper half of each row's 1 KB slot.  Widened in place -- lane l
// owns frames l and 64 + l of every row; all reads (the upper half) are back before the first write, so the lower
// half's doubles may overwrite what the upper half's were read from -- with the fused step applied on the way
// (OP: -1 none, 0 v*m, 1 v+m, 2 v-m, 3 m-v; g0 / g1: its operand at the lane's two frames).
template <int RU, int OP>
__device__ __forceinline__ void rsos_widen(uint32_t slot, uint32_t row_bytes, int lane, double g0, double g1) {
    if constexpr (RU > 4) {
        rsos_widen<4, OP>(slot, row_bytes, lane, g0, g1);
        rsos_widen<RU - 4, OP>(slot + 4 * row_bytes, row_bytes, lane, g0, g1);
        return;
    } else {
        float a[RU], b[RU];
        const uint32_t ra = slot + 512u + (uint32_t)lane * 4u;
#pragma unroll
        for (int c = 0; c < RU; ++c) {
            asm volatile("ds_read_b32 %0, %1" : "=v"(a[c]) : "v"(ra + (uint32_t)c * row_bytes) : "memory");
            asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(b[c]) : "v"(ra + (uint32_t)c * row_bytes) : "memory");
        }
        if constexpr (RU == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])::"memory");
        else if constexpr (RU == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0])::"memory");
        // (the LDS adder for v + m, as the Float64 path uses it, is slower here: 1.36 ms against 1.33 with the vector add)
        double v0[RU], v1[RU];
#pragma unroll
        for (int c = 0; c < RU; ++c) {
            const double x0 = (double)a[c], x1 = (double)b[c];
            v0[c] = OP == 0 ? x0 * g0 : OP == 1 ? x0 + g0 : OP == 2 ? x0 - g0 : OP == 3 ? g0 - x0 : x0;
            v1[c] = OP == 0 ? x1 * g1 : OP == 1 ? x1 + g1 : OP == 2 ? x1 - g1 : OP == 3 ? g1 - x1 : x1;
        }
        const uint32_t wa = slot + (uint32_t)lane * 8u;
#pragma unroll
        for (int c = 0; c < RU; ++c) {
            asm volatile("ds_write_b64 %0, %1" ::"v"(wa + (uint32_t)c * row_bytes), "v"(v0[c]) : "memory");
            asm volatile("ds_write_b64 %0, %1 offset:512" ::"v"(wa + (uint32_t)c * row_bytes), "v"(v1[c]) : "memory");
        }
    }
}

constexpr int kRsosFlagLdp = 0, kRsosFlagYrd = 4, kRsosFlagXseq = 16, kRsosFlagSseq = 48, kRsosFlagXh = 49, kRsosFlagLnd = 52, kRsosFlags = 56;
// Helper geometry (RsSos::help; 12 waves, taps in registers): the y waves of residues kRsosHres[0..2] -- one on each of SIMDs 1 .. 3 --
// hand the y wave that shares the chain's SIMD (residue kRsosHelper) their X block instead of computing D . X themselves:
// 68 / 68 / 68 / 66 MFMAs per round of ten blocks on the four SIMDs instead of 72 / 72 / 72 / 54.  (Not the chain wave: its
// steps are the kernel's serial dependency.  Not the back parts -- state-dependent, needed at once: round 5 moved those and
// made that wave the pace of everybody --: D . X of a block is needed ~8 blocks after its X exists.)
constexpr int kRsosHres0 = 1, kRsosHres1 = 5, kRsosHres2 = 7, kRsosHelper = 6, kRsosHslots = 3;  // (SIMDs 2, 3, 1)
constexpr int kRsosMaxGroups = 256;

// What the three roles share besides the dynamic LDS (taps, ring, exchange slots): a copy of the kernel's arguments
// (the roles are out-of-line functions with register allocations of their own -- one kernel body for all three spilled
// a hundred scalars and paid for it in every loop -- and read their geometry from here), the sequence counters, the
// window ends of the period's blocks, the fused source's control block.
struct RsosShared {
    RsSos g;
    const DCarrier* gcar;
    const DOp* gops;
    const DLeaf* gleaves;
    void* y;
    const double* tab;  // [ngroups][KS][64] taps in global memory (y waves that keep their phases' taps in registers)
    int flags[kRsosFlags];
    int jend[kRsosMaxGroups];
    RsCtl ctl;
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }


// Global-memory pointers rebuilt from values in LDS are GENERIC to the compiler: their loads and stores would be flat_
// instructions, which count on lgkmcnt as well as vmcnt -- every wait for an LDS read would then also wait for the result
// stores of the block before (a trip to HBM).  Typed as address space 1 they are global_ instructions.
#define SO_GLB __attribute__((address_space(1)))

// (see SO_SPIN_PAUSE above; never taken in a launch that works)
__device__ __forceinline__ void rsos_wait_failed(SO_LDS RsosShared* sh) {
    uint32_t SO_GLB* const e = (uint32_t SO_GLB*)rfl64((int64_t)(uintptr_t)sh->g.err);
    if (e == nullptr) __builtin_trap();
    if ((threadIdx.x & 63) == 0) __hip_atomic_store((uint32_t*)e, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_endpgm();
}

// cycle stamp of workgroup 0 (tuning aid, SIGOPS_RSOS_TRACE); iterations [it0, it0 + kRsosTraceIters) are recorded
#ifndef SO_RSOS_TRACE
#define SO_RSOS_TRACE 0  // (build with -DSO_RSOS_TRACE=1: the stamps cost the y waves ~70 instructions per block even when off)
#endif
__device__ __forceinline__ void rsos_stamp(long long* trace, int wave, int it, int k, int it0 = 400) {
    if constexpr (!SO_RSOS_TRACE) return;
    if (trace != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && it >= it0 && it < it0 + kRsosTraceIters)
        ((long long SO_GLB*)trace)[(wave * kRsosTraceIters + (it - it0)) * 8 + k] = clock64();
}

// who waits for whom (tuning aid, a -DSO_RSOS_COUNT=1 build with SIGOPS_RSOS_TRACE=2): every wave of workgroup 0 counts the
// polls of each of its waits -- in the spin paths only, nothing in the common path -- and leaves them in its first trace row
#ifndef SO_RSOS_COUNT
#define SO_RSOS_COUNT 0
#endif
__device__ __forceinline__ void rsos_count_out(long long* trace, int wave, int row, int k, long long v) {
    if constexpr (!SO_RSOS_COUNT) return;
    if (trace != nullptr && blockIdx.x == 0) ((long long SO_GLB*)trace)[((wave * kRsosTraceIters) + row) * 8 + k] = v + 1;
}

// dynamic LDS: [ngroups][KS][64] taps | [16][rpitch] ring | [NX][3][64] D.x blocks | [NS][3][64] states | [16][16][2] sine bases
struct RsosLds {
    SO_LDS double *taps, *ring, *gtab;
    volatile SO_LDS double *xs, *ss;  // what the waves hand each other
    volatile SO_LDS double* xh;       // helper geometry: the fourth register of an X block ([3 waves][kRsosHslots][64]; 0..2 go to its xs slot)
};
// The roles are called with generic pointers (in vector registers, as the calling convention has it): made wave-uniform
// 32-bit LDS pointers again here, so that every access below is a ds_ instruction with a scalar base.
__device__ __forceinline__ SO_LDS double* rsos_lds_ptr(const void* p) {
    return (SO_LDS double*)(uintptr_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const SO_LDS void*)p);
}
// state slots: the state entering block b is written at the chain's step b and read by the back part of block b, which its
// y wave runs BEHIND the front part of its block b + NY -- i.e. after it has handed over that block's D . x, when the chain is
// free to run on.  What holds the chain back for good is the same wave's NEXT hand-over: D . x of block b + 2 NY comes behind
// the back part of block b in program order, and the chain reads D . x two steps ahead, so it cannot begin step b + 2 NY - 1
// before the state of block b has been read: 2 NY - 1 slots are safe by construction, fewer are a race that a faster chain
// wins (NY + 4 slots, tried for a larger input ring: one bad block in a million, the re-read of a low-priority wave whose
// vector instructions wait behind its neighbours' MFMAs while the chain runs six steps; waiting for the state BEFORE the
// hand-over instead makes the y waves and the chain wait for each other: 0.955 -> 1.007 ms).
#ifndef SO_NSS_EXTRA
#define SO_NSS_EXTRA (ny + 1)  // (-DSO_NSS_EXTRA=4: the racy NY + 4 of the measurement above)
#endif
__host__ __device__ constexpr int rsos_nss(int ny) { return ny + SO_NSS_EXTRA; }
__device__ __forceinline__ RsosLds rsos_carve(double* dyn_, int tapd, int rpitch, int nx, int ns) {  // tapd: doubles of the tap table in LDS
    RsosLds l;
    SO_LDS double* dyn = rsos_lds_ptr(dyn_);
    l.taps = dyn;
    l.ring = l.taps + tapd;
    l.xs = l.ring + (size_t)16 * rpitch;
    l.ss = l.xs + (size_t)nx * 192;
    l.gtab = (SO_LDS double*)l.ss + (size_t)ns * 192;
    l.xh = l.gtab + 16 * 16 * 2;
    return l;
}

// geometry of a sequence group that all roles derive the same way (wave-uniform)
struct RsosGroup {
    int cg;         // channel group
    int64_t pb0;    // first input frame of the first period range 0 of the group walks (may be negative)
    int64_t ob0;    // ... and its first output
    int64_t prM, prL;  // inputs / outputs from one range to the next
    int64_t r0;     // first range of the group
    int e0m, prMm;  // alignment of range ri's first staged frame: (e0m + ri prMm) & 15 frames above a 128-byte line
};
__device__ __forceinline__ RsosGroup rsos_group(const SO_LDS RsosShared* sh, int64_t G, bool single, int64_t base0_8, int64_t cs0, int64_t df0) {
    const SO_LDS RsSos& g = sh->g;
    const int ct = uni(g.ct), rgs = uni(g.rgs);
    const int64_t ncg = uni(g.nch) / ct;
    const int64_t pr = rfl64(g.pr), M = rfl64(g.M), L = rfl64(g.L);
    const int wp = uni(g.wp), ulo = uni(g.ulo);
    RsosGroup q;
    q.cg = (int)(G % ncg);
    const int64_t r0 = (G / ncg) * rgs;
    q.r0 = r0;
    q.pb0 = (r0 * pr - wp) * M;
    q.ob0 = (r0 * pr - wp) * L;
    q.prM = pr * M;
    q.prL = pr * L;
    if (single) {
        const int64_t e = base0_8 + (int64_t)(q.cg * ct) * cs0 + df0 + q.pb0 + ulo;
        q.e0m = (int)(((e % 16) + 16) % 16);
        q.prMm = (int)(q.prM % 16);
    } else
        q.e0m = q.prMm = 0;
    return q;
}

// =========================== chain wave ===========================
// NK: k-steps of the state, 4 states each (a cascade of 1 - 2 sections: 1, 3 - 4: 2, 5 - 6: 3) -- the MFMAs of this wave's
// recurrence, 64 cycles each; the y waves' S^T C^T takes as many
template <int NY, int NK>
__device__ __attribute__((noinline)) void rsos_chain(RsosShared* sh_, double* dyn, int64_t G_) {
    constexpr int NX = 2 * NY + 1;
    const int lane = threadIdx.x & 63;
    SO_LDS RsosShared* const sh = (SO_LDS RsosShared*)rsos_lds_ptr(sh_);
    const SO_LDS RsSos& g = sh->g;
    const int ngroups = uni(g.ngroups);
    const RsosLds l = rsos_carve(dyn, uni(g.cyc) > 0 ? 0 : ngroups * uni(g.ks) * 64, uni(g.rpitch), NX, rsos_nss(NY));
    const int NB = (uni(g.wp) + (int)rfl64(g.pr)) * ngroups;
    const double SO_GLB* mats = (const double SO_GLB*)rfl64((int64_t)(uintptr_t)g.mats);
    long long* trace = (long long*)rfl64((int64_t)(uintptr_t)g.trace);
    const uint32_t fl_base = (uint32_t)(uintptr_t)sh->flags;
    const int debug = uni(g.debug);
    double Ak[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) Ak[v] = mats[(4 + v) * 64 + lane];
    // this wave IS the critical path: its three MFMAs per block must not queue behind the 25 of the y wave that shares
    // the SIMD (measured: 500 cycles for the three without, next to a y wave in its MFMA phase)
    __builtin_amdgcn_s_setprio(3);
    // The only serial dependency of the kernel: S' = (D . X) + A^16 . S, three MFMAs per block on the state this wave
    // carries in registers for all 16 rows.  D . X arrives from the y waves (they have X in registers); the block after
    // the next is already on its way from LDS, its counter polled behind the MFMAs.
    int spins = 0;
    [[maybe_unused]] int cnt_wait = 0, cnt_blocks = 0;  // (SO_RSOS_COUNT: per lane = the y wave whose D.x was late)
    [[maybe_unused]] const long long cyc0 = SO_RSOS_COUNT ? clock64() : 0;
    while (uni(flag_ld(fl_base + 4 * (kRsosFlagXseq + 0))) < 1 && !(debug & 4)) SO_SPIN_PAUSE(spins, 1, 1 << 22);
    // Two steps per loop iteration with the roles of the register sets swapped: the state a step leaves is the next
    // step's B operand where it is, the operand set a step has consumed is refilled (under its first MFMA) with the
    // block two steps on, and the counter that guards that refill was requested two steps earlier still.
    // What the loop looks like is dictated by three measurements (tools/micro/mfma64_chain.hip, one wave on a CU):
    //   240 cycles per block  the bare recurrence: 3 x 64 for MFMAs that wait for each other + 48 before the next
    //                         block's first one may read this block's result;
    //   432                   the same with the block's LDS traffic in the natural places -- an in-order wave does
    //                         nothing while it waits, so every round trip adds its latency to the kernel's only serial
    //                         chain;
    //   294                   with every LDS request UNDER a block's first MFMA, no wait between two MFMAs, and no
    //                         VECTOR instruction under an MFMA either: an fp64 MFMA keeps the vector ALU for its 64
    //                         cycles, a v_mov or v_readfirstlane "under" it waits for its end and pushes the chain back.
    //                         Addresses, the counter's readfirstlane and the counter value therefore go into the gap
    //                         in front of the first MFMA; under it are scalar and LDS instructions only.
    v4d sA = v4d{0.0, 0.0, 0.0, 0.0}, sB = v4d{0.0, 0.0, 0.0, 0.0};  // state entering the even / odd block
    // D . x of the even / odd block, each in the four register pairs an MFMA takes as its C operand (rows 12 .. 15 of the
    // state do not exist: the fourth pair is never read back and may hold anything -- it is never written either)
    v4d dA = v4d{0.0, 0.0, 0.0, 0.0}, dB = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int v = 0; v < NK; ++v) dA[v] = l.xs[v * 64 + lane];
    if (NB > 1) {
        spins = 0;
        while (uni(flag_ld(fl_base + 4 * (kRsosFlagXseq + 1))) < 2 && !(debug & 4)) SO_SPIN_PAUSE(spins, 1, 1 << 22);
#pragma unroll
        for (int v = 0; v < NK; ++v) dB[v] = l.xs[192 + v * 64 + lane];
    }
    const uint32_t xs0 = (uint32_t)(uintptr_t)l.xs + (uint32_t)lane * 8u, ss0 = (uint32_t)(uintptr_t)l.ss + (uint32_t)lane * 8u;
    int fA = NB > 2 ? flag_ld(fl_base + 4 * (kRsosFlagXseq + 2 % NX)) : 0x7fffffff;  // counters of blocks 2 and 3
    int fB = NB > 3 ? flag_ld(fl_base + 4 * (kRsosFlagXseq + 3 % NX)) : 0x7fffffff;
    int slot = 0;   // of block b (D . x slots: b modulo NX)
    int sslot = 0;  // ... and its state slot (b modulo NS)
    constexpr int NS = rsos_nss(NY);
    auto step = [&](int b, v4d& sin, v4d& sout, v4d& din, int& fpend) __attribute__((always_inline)) {
        // ---- the gap in front of the first MFMA: vector instructions ----
        int f = uni(fpend);  // counter of block b + 2's slot, requested two steps ago
        const int s2 = slot + 2 >= NX ? slot + 2 - NX : slot + 2, s4 = s2 + 2 >= NX ? s2 + 2 - NX : s2 + 2;
        uint32_t ax = xs0 + (uint32_t)s2 * 1536u, as = ss0 + (uint32_t)sslot * 1536u;
        uint32_t af = fl_base + 4u * (uint32_t)(kRsosFlagXseq + s4), ag = fl_base + 4u * (uint32_t)kRsosFlagSseq;
        int bv = b;
        asm volatile("" : "+v"(ax), "+v"(as), "+v"(af), "+v"(ag), "+v"(bv));  // (in vector registers NOW)
        __builtin_amdgcn_sched_barrier(0);
        v4d acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[0], sin[0], din, 0, 0, 0);  // (C = D . x where the LDS reads left it: no copy)
        __builtin_amdgcn_sched_barrier(0);
        // ---- under the MFMAs: scalar and LDS instructions, a few per MFMA (eight LDS instructions under the first one took
        //      longer to issue than its 64 cycles and pushed the second back: 0.74 ms for the chain alone, 0.53 without them).
        //      Under the first: the state entering block b (the registers the MFMAs are reading) for its y wave -- the
        //      earliest it can go.  Under the second: D . x of block b + 2 into the registers the FIRST MFMA took its C
        //      operand from (issued 64 cycles behind it: that read is over).  ... and the counter of block b + 4.  Nothing under the third: the next step's first wait would count its requests.
#ifndef SO_CHAIN_ORDER
#define SO_CHAIN_ORDER 1
#endif
        auto put_state = [&]() __attribute__((always_inline)) {
            if (!(debug & 1024)) {
#pragma unroll
                for (int v = 0; v < NK; ++v) *(volatile SO_LDS double*)(uintptr_t)(as + (uint32_t)v * 512u) = sin[v];
                *(volatile SO_LDS int*)(uintptr_t)ag = bv;
            }
        };
        const bool more = b + 2 < NB && !(debug & 512);
        auto wait_dx = [&]() __attribute__((always_inline)) {
            if (more && f < b + 3 && !(debug & 4)) {  // (the y waves are behind: wait here -- everybody waits for this wave anyway)
                spins = 0;
                if constexpr (SO_RSOS_COUNT) cnt_blocks += lane == (b + 2) % NY ? 1 : 0;
                do {
                    SO_SPIN_PAUSE(spins, 1, 1 << 22);
                    if constexpr (SO_RSOS_COUNT) cnt_wait += lane == (b + 2) % NY ? 1 : 0;
                    f = uni(flag_ld(fl_base + 4 * (kRsosFlagXseq + s2)));
                } while (f < b + 3);
            }
        };
        auto get_dx = [&]() __attribute__((always_inline)) {
            if (more) {
#pragma unroll
                for (int v = 0; v < NK; ++v) din[v] = *(volatile SO_LDS double*)(uintptr_t)(ax + (uint32_t)v * 512u);
            }
        };
        auto get_flag = [&]() __attribute__((always_inline)) {
            if (more && b + 4 < NB) fpend = *(volatile SO_LDS int*)(uintptr_t)af;
        };
        if constexpr (NK == 1) {
            // (one MFMA: the LDS traffic behind it -- the requests are out before its 64 cycles are)
            put_state();
            wait_dx();
            get_dx();
            get_flag();
        } else if constexpr (NK == 2) {
            // (two: everything under the first -- five LDS instructions --, nothing under the last, as below)
            put_state();
            wait_dx();
            get_dx();
            get_flag();
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[1], sin[1], acc, 0, 0, 0);
        } else if constexpr (SO_CHAIN_ORDER == 0) {
            wait_dx();
            get_dx();
            get_flag();
            put_state();
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[1], sin[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[2], sin[2], acc, 0, 0, 0);
        } else {
            put_state();
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[1], sin[1], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            wait_dx();
            get_dx();
            get_flag();
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[2], sin[2], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        sout = acc;
        slot = slot + 1 == NX ? 0 : slot + 1;
        sslot = sslot + 1 == NS ? 0 : sslot + 1;
    };
    for (int b = 0; b < NB; b += 2) {
        rsos_stamp(trace, 0, b >> 1, 0, 200);
        step(b, sA, sB, dA, fA);
        if (b + 1 < NB) step(b + 1, sB, sA, dB, fB);
        rsos_stamp(trace, 0, b >> 1, 3, 200);
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (SO_RSOS_COUNT) {
        if (lane < 8) rsos_count_out(trace, 0, 0, lane, cnt_wait);
        else if (lane < 16) rsos_count_out(trace, 0, 1, lane - 8, cnt_wait);
        if (lane < 8) rsos_count_out(trace, 0, 2, lane, cnt_blocks);
        else if (lane < 16) rsos_count_out(trace, 0, 3, lane - 8, cnt_blocks);
        if (lane == 0) rsos_count_out(trace, 0, 4, 0, clock64() - cyc0);
    }
    // A filter never recovers from a non-finite sample: the reference's recurrence carries a NaN or Inf on in its state
    // to the end of the channel (k_sos_poison).  Here the next range starts from rest wp periods early and
    // would be finite again: a range whose walk ends in a non-finite state is noted per channel (the smallest such
    // range), and the launch behind this kernel fills NaN into everything that follows it.
    int32_t* bad = (int32_t*)rfl64((int64_t)(uintptr_t)g.bad);
    if (bad != nullptr) {
        const v4d sE = (NB & 1) ? sB : sA;  // the state the last block left
        const bool nf = !(isfinite(sE[0]) && (NK < 2 || isfinite(sE[1])) && (NK < 3 || isfinite(sE[2])));
        const uint64_t m = __ballot(nf);
        const uint32_t rows = (uint32_t)((m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xffffu);  // row = lane & 15
        if (rows != 0 && lane < 16 && ((rows >> lane) & 1u)) {
            const int ct = uni(g.ct), rgs = uni(g.rgs);
            const int64_t G = rfl64(G_);
            const int64_t ncg = uni(g.nch) / ct;
            const int64_t r = (G / ncg) * rgs + lane / ct;
            if (r < uni(g.nranges)) atomicMin(bad + ((int)(G % ncg) * ct + lane % ct), (int32_t)r);
        }
    }
}

// Two arrays (RsSos::arr2): the second one's rows can be read 16 bytes per lane at the frames the first one's 128-byte lines
// start at -- both arrays' rows equally aligned.  Loaders, step waves and y waves agree on it (it is part of `single`).
__device__ __forceinline__ bool rsos_two_ok(const SO_LDS DCarrier& C0, int chunk) {
    const uint64_t b0 = (uint64_t)rfl64((int64_t)(uintptr_t)C0.base), b2 = (uint64_t)rfl64((int64_t)(uintptr_t)C0.base2);
    const int64_t df0 = rfl64(C0.df), df2 = rfl64(C0.df2), cs0 = rfl64(C0.cstride), cs2 = rfl64(C0.cstride2);
    return uni((int)(C0.base2 != nullptr && C0.vec_ok2 && C0.dtype2 == SO_F64)) && chunk == 128 && !((((b2 >> 3) + (uint64_t)df2) ^ ((b0 >> 3) + (uint64_t)df0)) & 1) &&
           !((cs2 ^ cs0) & 1);
}

// =========================== loader waves ===========================
// vmcnt of this wave, read without waiting (HW_REG_IB_STS: VM_CNT in bits 3:0 and 23:22)
__device__ __forceinline__ int vmcnt_now() {
    const uint32_t v = __builtin_amdgcn_s_getreg(7 | (0 << 6) | (31 << 11));
    return (int)((v & 0xfu) | ((v >> 18) & 0x30u));
}

// MODE (RsSos::gsplit: the 16-wave geometry of two-channel groups with a fused step): 0 the whole loader; 1 the loader without
// the step -- it issues, waits for a chunk to land and says so (kRsosFlagLnd) --; 2 the STEP wave (one of the otherwise idle
// waves 13 / 14, on SIMDs 1 / 2) that applies the step to a landed chunk and publishes it.  Issue and step of a stereo chunk
// are ~2 600 - 3 100 and ~2 000 - 2 500 cycles of ONE wave's time (every vector instruction next to fp64 MFMAs waits for a
// gap between them) against a chunk period of ~6 000 at the plain pipeline's pace: as two stages of a pipeline they fit.
// A2 (RsSos::arr2; sixteen waves, groups of eight channels): carrier 0's one step takes a SECOND Float64 array as its operand --
// `Mix(x, y)` / `Amplify(x, y)` of two arrays in front of the resampler (reference: src/mapsignal.jl:54-57 evaluated block by
// block inside the resampler's pull, src/filters.jl:240-244).  The first array goes into the ring by LDS-DMA as ever (the two
// loader waves, MODE 1: issue, wait, report); the second one's samples go through the REGISTERS of the two step waves (MODE 2,
// one unit of eight rows each): 16 bytes per lane and row, THREE chunks in flight per wave (96 registers: 48 KB per workgroup --
// under load a global load takes ~5 us to return; one chunk in flight next to the loader's own DMA made the second array's
// 0.8 GB take 0.95 ms), applied to the landed chunk by the LDS's own adder like the sine of the fused `Mix(sine, x)`.  No
// staging buffer (none fits next to the ring), no vector instruction per row.
template <int NY, int NL, int RU, bool SRC32, int RB = 0, int MODE = 0, bool A2 = false>
__device__ __attribute__((noinline)) void rsos_loader(RsosShared* sh_, double* dyn, int64_t G_, int q_) {
    static_assert(!A2 || ((RU == 8 || RU == 4 || RU == 2) && !SRC32 && MODE == 2 && NL == 2), "the second array's waves: groups of two, four or eight channels, sixteen waves");
    constexpr int NX = 2 * NY + 1;
    const int lane = threadIdx.x & 63;
    const int wave = uni(threadIdx.x >> 6);
    SO_LDS RsosShared* const sh = (SO_LDS RsosShared*)rsos_lds_ptr(sh_);
    const int64_t G = rfl64(G_);
    const int q = uni(q_);
    const SO_LDS RsSos& g = sh->g;
    const int ngroups = uni(g.ngroups), rpitch = uni(g.rpitch), RING = uni(g.ring), CH = uni(g.chunk);
    const RsosLds l = rsos_carve(dyn, uni(g.cyc) > 0 ? 0 : ngroups * uni(g.ks) * 64, rpitch, NX, rsos_nss(NY));
    const int ct = uni(g.ct);
    const int M = (int)rfl64(g.M);
    const int NP = uni(g.wp) + (int)rfl64(g.pr);
    const int ulo = uni(g.ulo);
    const int64_t n_in = rfl64(g.n_in);
    long long* trace = (long long*)rfl64((int64_t)(uintptr_t)g.trace);
    const uint32_t fl_base = (uint32_t)(uintptr_t)sh->flags;
    constexpr int nunits = 16 / RU;
    const int lanes = CH >> 1;
    const uint64_t dmask = lanes >= 64 ? ~0ull : ((1ull << lanes) - 1ull);
    const int total_rho = (NP - 1) * M + uni(sh->jend[ngroups - 1]) - ulo + 1 + 16;
    const int NK = (total_rho + CH - 1) / CH;
    const uint32_t ring_b = (uint32_t)(uintptr_t)l.ring;
    const uint32_t row_bytes = (uint32_t)rpitch * 8u;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    __builtin_amdgcn_s_setprio(2);
    // carrier 0: the fast path's only source
    const SO_LDS DCarrier& C0 = sh->ctl.car[0];
    const int64_t a0 = rfl64(C0.a), b0 = rfl64(C0.b), cs0 = rfl64(C0.cstride), df0 = rfl64(C0.df);
    const char* const base0 = (const char*)rfl64((int64_t)(uintptr_t)C0.base);
    const int fuse = uni(g.fuse), fuse_sine = uni(g.fuse_sine), debug = uni(g.debug);
    const bool ring32 = SRC32 && uni(g.ring32) != 0;  // (the chunks stay Float32, at four bytes a frame: the y waves widen their operands)
    const bool sring = ring32 && uni(g.sring) != 0;   // (... and the fused v + m / v - m is added by the y waves: the summand ring)
    const uint32_t half_bytes = (uint32_t)uni(g.rpitch) * 4u;
    constexpr bool src32 = SRC32;  // (a Float32 array: 4-byte elements, 128 of them per chunk and LDS-DMA instruction; its own
                                   //  instantiation: the Float64 loader's per-chunk path stays what it was)
    constexpr int esh = src32 ? 2 : 3;
    const bool single = uni((int)(C0.base != nullptr && C0.vec_ok && C0.dtype == (src32 ? SO_F32 : SO_F64))) && !(df0 & 1) && fuse >= -1 &&
                        (!src32 || CH == 128) && (!uni(g.arr2) || rsos_two_ok(C0, CH));
    const int64_t lo_ok = a0 > 0 ? a0 : 0, hi_ok = b0 < n_in ? b0 : n_in;
    const RsosGroup grp = rsos_group(sh, G, single, (int64_t)((uintptr_t)base0 >> esh), cs0, df0);
    // the fused step's gain: a constant, or a sine generator evaluated in two levels (share bases per chunk in gtab,
    // per-lane (sin, cos) of the lane's two frame offsets)
    const int kind0 = uni(C0.slot_kind[0]);
    const DLeaf leaf0 = leaf_uniform(sh_->ctl.leaves[min(kCtlLeaves - 1, max(0, uni(C0.slot_leaf[0])))]);
    double2 d0 = double2{0.0, 1.0}, d1 = double2{0.0, 1.0};
    double2 d16 = double2{0.0, 1.0};  // (sin, cos) of the phase 16 chunks add: the share bases' step from refresh to refresh
    double gconst = 0.0;
    if (fuse >= 0) {
        if (fuse_sine) {  // (the lane's two frames of a chunk: 2 l and 2 l + 1, or l and 64 + l where it widens Float32)
            // (... or, where the ring keeps the Float32 samples and the step is done on them -- RsSos::f32m --, 2 l and 2 l + 1 again)
            d0 = rsos_sine_at(src32 && !ring32 ? lane : 2 * lane, leaf0.v0, 0.0, leaf0.v2, leaf0.flag);
            d1 = rsos_sine_at(src32 && !ring32 ? 64 + lane : 2 * lane + 1, leaf0.v0, 0.0, leaf0.v2, leaf0.flag);
            d16 = rsos_sine_at(16 * CH, leaf0.v0, 0.0, leaf0.v2, leaf0.flag);
        } else
            gconst = slot_eval(kind0, leaf0, 0);
    }
    // share bases of this lane's (unit slot, chunk of sixteen): evaluated in full every eighth refresh, turned by d16 in between
    // (4 multiply-adds instead of the ~150 vector instructions of a sincospi next to the chain wave's MFMAs; seven
    // rotations add < 1e-15 to a base)
    double2 rbase[4] = {double2{0.0, 1.0}, double2{0.0, 1.0}, double2{0.0, 1.0}, double2{0.0, 1.0}};
    // unit u: rows [u RU, (u + 1) RU) of the group = RU channels of range ri.  What the loop needs of this wave's units
    // (u = q + j NL) sits in lane j: first staged frame (ring position 0), its address in channel ch0's row, the unit's
    // first ring row in LDS, and the chunks [klo, khi) that lie inside carrier 0 and the signal -- the LDS-DMA ones.
    const int shift = CH == 128 ? 7 : 6;
    int64_t Au_l = 0, rowb_l = 0;
    int klo_l = 1, khi_l = 0, ch0_l = 0;
    int kzl_l = 0, kzh_l = 0x3fffffff;  // chunks below kzl / from kzh on lie wholly outside the signal: zeros
    uint32_t lds_l = 0;
    {
        const int u = q + lane * NL;
        if (u < nunits) {
            const int ri = (u * RU) / ct, c0u = (u * RU) % ct;
            ch0_l = grp.cg * ct + c0u;
            Au_l = grp.pb0 + ri * grp.prM + ulo - ((grp.e0m + ri * grp.prMm) & 15);
            rowb_l = (int64_t)(uintptr_t)base0 + (((int64_t)ch0_l * cs0 + df0 + Au_l) << esh);
            lds_l = ring_b + (uint32_t)(u * RU) * row_bytes;
            {
                const int64_t zl = (-Au_l) >> shift, zh = (n_in - Au_l + CH - 1) >> shift;  // (k + 1) CH <= -Au ; k CH >= n_in - Au
                kzl_l = (int)(zl < 0 ? 0 : (zl > 0x3fffffff ? 0x3fffffff : zl));
                kzh_l = (int)(zh < 0 ? 0 : (zh > 0x3fffffff ? 0x3fffffff : zh));
            }
            if (single) {
                const int64_t lo = (lo_ok - Au_l + CH - 1) >> shift, hi = (hi_ok - Au_l) >> shift;
                klo_l = (int)(lo < 0 ? 0 : (lo > 0x3fffffff ? 0x3fffffff : lo));
                khi_l = (int)(hi < 0 ? 0 : (hi > 0x3fffffff ? 0x3fffffff : hi));
            }
        }
    }
    auto lane64 = [&](int64_t v, int j) __attribute__((always_inline)) {
        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, j), hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)v >> 32), j);
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    const int MU = (nunits - q + NL - 1) / NL;  // units of this wave
    // The first two units' facts in scalar registers (the tables above stay for the others): next to the chain wave's
    // MFMA stream every VECTOR instruction of this wave -- a v_readlane as much as an fp64 add -- waits for a gap
    // between two MFMAs (~64 cycles each, measured: 16 LDS-DMA instructions took 2 300 cycles to issue), so the
    // per-chunk path of the common shapes (8 channels: two units, 16 channels: one) is scalar code + LDS-DMA only.
    const int64_t rowb_s0 = lane64(rowb_l, 0), rowb_s1 = lane64(rowb_l, 1), Au_s0 = lane64(Au_l, 0), Au_s1 = lane64(Au_l, 1);
    const int klo_s0 = __builtin_amdgcn_readlane(klo_l, 0), khi_s0 = __builtin_amdgcn_readlane(khi_l, 0);
    const int klo_s1 = __builtin_amdgcn_readlane(klo_l, 1), khi_s1 = __builtin_amdgcn_readlane(khi_l, 1);
    const int ch0_s0 = __builtin_amdgcn_readlane(ch0_l, 0), ch0_s1 = __builtin_amdgcn_readlane(ch0_l, 1);
    const uint32_t lds_s0 = (uint32_t)__builtin_amdgcn_readlane((int)lds_l, 0), lds_s1 = (uint32_t)__builtin_amdgcn_readlane((int)lds_l, 1);
    auto u_klo = [&](int j) __attribute__((always_inline)) { return j == 0 ? klo_s0 : j == 1 ? klo_s1 : __builtin_amdgcn_readlane(klo_l, j); };
    auto u_khi = [&](int j) __attribute__((always_inline)) { return j == 0 ? khi_s0 : j == 1 ? khi_s1 : __builtin_amdgcn_readlane(khi_l, j); };
    auto u_rowb = [&](int j) __attribute__((always_inline)) { return j == 0 ? rowb_s0 : j == 1 ? rowb_s1 : lane64(rowb_l, j); };
    auto u_Au = [&](int j) __attribute__((always_inline)) { return j == 0 ? Au_s0 : j == 1 ? Au_s1 : lane64(Au_l, j); };
    auto u_ch0 = [&](int j) __attribute__((always_inline)) { return j == 0 ? ch0_s0 : j == 1 ? ch0_s1 : __builtin_amdgcn_readlane(ch0_l, j); };
    auto u_lds = [&](int j) __attribute__((always_inline)) { return j == 0 ? lds_s0 : j == 1 ? lds_s1 : (uint32_t)__builtin_amdgcn_readlane((int)lds_l, j); };
    auto issue = [&](int k, int rho0) __attribute__((always_inline)) -> int {
        int n = 0;
        if constexpr (RU < 8) {
            // More than two units per chunk (groups of 4 / 2 / 1 channels): the units' state for chunk k from the lane table in
            // one go -- lane j classifies unit j with three vector compares --, the DMA units by scanning a ballot mask, three
            // readlanes (address, LDS row) each.  (Unit by unit -- six table reads and a chain of scalar selects per unit --
            // issuing a stereo chunk's eight units took 3 700 cycles where an eight-channel chunk's two take 800: the loader
            // alone 0.89 ms against 0.38 for the same samples.)
            const bool mine = lane < MU;
            const bool fast_l = mine && k >= klo_l && k < khi_l && !(debug & 8192);
            const bool zero_l = mine && !(k >= klo_l && k < khi_l) && (k < kzl_l || k >= kzh_l);
            const bool slow_l = mine && !(k >= klo_l && k < khi_l) && !zero_l && !(debug & 4096);
            const uint64_t fast_m = __ballot(fast_l), zero_m = __ballot(zero_l), slow_m = __ballot(slow_l);
            const int64_t row_l = rowb_l + ((int64_t)k << (shift + esh));
            for (uint64_t m = fast_m; m; m &= m - 1) {
                const int j = __builtin_ctzll(m);
                const char* row = (const char*)(uintptr_t)lane64(row_l, j);
                const uint32_t ld = (uint32_t)__builtin_amdgcn_readlane((int)lds_l, j);
                if constexpr (src32) dma_rows<RU>(0xffffffffull, lane16, row, cs0 * 4, ld + (ring32 ? (uint32_t)rho0 * 4u : (uint32_t)rho0 * 8u + 512u), row_bytes);
                else dma_rows<RU>(dmask, lane16, row, cs0 * 8, ld + (uint32_t)rho0 * 8u, row_bytes);
                n += RU;
            }
            for (uint64_t m = zero_m; m; m &= m - 1) {
                const int j = __builtin_ctzll(m);
                if (lane < (ring32 ? lanes >> 1 : lanes)) {
                    const uint32_t la = (uint32_t)__builtin_amdgcn_readlane((int)lds_l, j) + (uint32_t)rho0 * (ring32 ? 4u : 8u) + lane16;
                    const v2d z = v2d{0.0, 0.0};
#pragma unroll
                    for (int c = 0; c < RU; ++c) lds_st16(la + (uint32_t)c * row_bytes, z);
                }
            }
            for (uint64_t m = slow_m; m; m &= m - 1) {
                const int j = __builtin_ctzll(m);
                const int u = q + j * NL;
                if (ring32) {
                    if constexpr (src32)
                        rsos_stage_slow32<RU>(n_in, rpitch, (const DCarrier*)rfl64((int64_t)(uintptr_t)sh->gcar), uni(sh->ctl.ncar),
                                              (const DOp*)rfl64((int64_t)(uintptr_t)sh->gops), (const DLeaf*)rfl64((int64_t)(uintptr_t)sh->gleaves),
                                              lane64(Au_l, j) + ((int64_t)k << shift), lanes >> 1, __builtin_amdgcn_readlane(ch0_l, j),
                                              (float*)(l.ring + (size_t)(u * RU) * rpitch) + rho0);
                } else
                    rsos_stage_slow<RU>(n_in, rpitch, (const DCarrier*)rfl64((int64_t)(uintptr_t)sh->gcar), uni(sh->ctl.ncar),
                                        (const DOp*)rfl64((int64_t)(uintptr_t)sh->gops), (const DLeaf*)rfl64((int64_t)(uintptr_t)sh->gleaves),
                                        lane64(Au_l, j) + ((int64_t)k << shift), lanes, __builtin_amdgcn_readlane(ch0_l, j),
                                        (double*)(l.ring + (size_t)(u * RU) * rpitch + rho0));
            }
            return n;
        }
        for (int j = 0; j < MU; ++j) {
            const bool fast = k >= u_klo(j) && k < u_khi(j);
            if (fast && (debug & 8192)) continue;
            if (fast) {
                const char* row = (const char*)(uintptr_t)(u_rowb(j) + ((int64_t)k << (shift + esh)));
                if constexpr (src32) dma_rows<RU>(0xffffffffull, lane16, row, cs0 * 4, u_lds(j) + (ring32 ? (uint32_t)rho0 * 4u : (uint32_t)rho0 * 8u + 512u), row_bytes);
                else dma_rows<RU>(dmask, lane16, row, cs0 * 8, u_lds(j) + (uint32_t)rho0 * 8u, row_bytes);
                n += RU;
            } else if (k < __builtin_amdgcn_readlane(kzl_l, j) || k >= __builtin_amdgcn_readlane(kzh_l, j)) {
                // wholly before the signal's first frame (the warm-up of the first range) or behind its last: zeros
                // (Pad(x.signal, zero), reference src/filters.jl:240) -- the general path below costs ~30 000 cycles
                // per chunk, and the workgroup that walks range 0 held the whole kernel up by 0.46 ms with it
                if (lane < (ring32 ? lanes >> 1 : lanes)) {
                    const uint32_t la = u_lds(j) + (uint32_t)rho0 * (ring32 ? 4u : 8u) + lane16;
                    const v2d z = v2d{0.0, 0.0};
#pragma unroll
                    for (int c = 0; c < RU; ++c) lds_st16(la + (uint32_t)c * row_bytes, z);
                }
            } else if (!(debug & 4096) && ring32) {
                if constexpr (src32) {
                    const int u = q + j * NL;
                    rsos_stage_slow32<RU>(n_in, rpitch, (const DCarrier*)rfl64((int64_t)(uintptr_t)sh->gcar), uni(sh->ctl.ncar),
                                          (const DOp*)rfl64((int64_t)(uintptr_t)sh->gops), (const DLeaf*)rfl64((int64_t)(uintptr_t)sh->gleaves),
                                          u_Au(j) + ((int64_t)k << shift), lanes >> 1, u_ch0(j),
                                          (float*)(l.ring + (size_t)(u * RU) * rpitch) + rho0);
                }
            } else if (!(debug & 4096)) {
                const int u = q + j * NL;
                rsos_stage_slow<RU>(n_in, rpitch, (const DCarrier*)rfl64((int64_t)(uintptr_t)sh->gcar), uni(sh->ctl.ncar),
                                    (const DOp*)rfl64((int64_t)(uintptr_t)sh->gops), (const DLeaf*)rfl64((int64_t)(uintptr_t)sh->gleaves),
                                    u_Au(j) + ((int64_t)k << shift), lanes, u_ch0(j),
                                    (double*)(l.ring + (size_t)(u * RU) * rpitch + rho0));
            }
        }
        return n;
    };
    auto retire = [&](int k, int rho0, int allowed) __attribute__((always_inline)) {
        rsos_stamp(trace, wave, k, 2, 40);
        if constexpr (MODE == 1) {  // (the step wave takes it from here)
            wait_vmcnt_le60(allowed);
            flag_st(fl_base + 4 * (kRsosFlagLnd + q), k + 1);
            rsos_stamp(trace, wave, k, 4, 40);
            return;
        }
        const bool gain = (fuse >= 0 || (src32 && !ring32)) && !(debug & 2);
        // what does not depend on the chunk's samples comes BEFORE the wait for them: the share bases of the next sixteen
        // chunks (every sixteenth chunk) and this chunk's own, read back from LDS
        if (gain && fuse >= 0 && fuse_sine && (k & 15) == 0 && !(debug & 2048)) {  // lane = (unit slot, chunk)
#pragma unroll
            for (int j4 = 0; j4 < 16; j4 += 4) {
                if (j4 >= MU) break;
                const int j = j4 + (lane >> 4);
                double2 bs;
                if ((k & 127) == 0 || (debug & 65536)) {
                    const int64_t Au = (int64_t)(((uint64_t)(uint32_t)__shfl((int)((uint64_t)Au_l >> 32), j, 64) << 32) |
                                                 (uint32_t)__shfl((int)(uint32_t)Au_l, j, 64));
                    bs = rsos_sine_at(Au + ((int64_t)(k + (lane & 15)) << shift) + leaf0.df + 1, leaf0.v0, leaf0.v1, leaf0.v2, leaf0.flag);
                } else {
                    const double2 o = rbase[j4 >> 2];
                    bs.x = fma(o.x, d16.y, o.y * d16.x);
                    bs.y = fma(o.y, d16.y, -(o.x * d16.x));
                }
                rbase[j4 >> 2] = bs;
                if (j < MU) {
                    const int u = q + j * NL;
                    l.gtab[(u * 16 + (lane & 15)) * 2] = bs.x;
                    l.gtab[(u * 16 + (lane & 15)) * 2 + 1] = bs.y;
                }
            }
        }
        double2 bs0 = double2{0.0, 0.0}, bs1 = double2{0.0, 0.0};  // the first two units' bases (the others read theirs below)
        if (gain && fuse >= 0 && fuse_sine) {
            const int u0 = q, u1 = q + NL;
            bs0 = double2{l.gtab[(u0 * 16 + (k & 15)) * 2], l.gtab[(u0 * 16 + (k & 15)) * 2 + 1]};
            if (MU > 1) bs1 = double2{l.gtab[(u1 * 16 + (k & 15)) * 2], l.gtab[(u1 * 16 + (k & 15)) * 2 + 1]};
        }
        if constexpr (MODE == 0) wait_vmcnt_le60(allowed);  // chunk k's DMA has landed
        rsos_stamp(trace, wave, k, 3, 40);
        bool gain_done = false;
        if constexpr (RU < 8 && !SRC32) {
            if (gain) {
                // more than two units per chunk: which of them take the step from one vector compare and a ballot, their share
                // bases four units at a time (ONE wait for four LDS reads), the units' ring rows from scalars -- unit by unit
                // (two table reads, an LDS round trip for the base, one more table read each) the step cost a stereo
                // chunk's eight units a third of the kernel (Mix(sin, x) of two channels: 1.75 ms against 1.39 without it)
                const uint64_t fm = __ballot(lane < MU && k >= klo_l && k < khi_l);
                constexpr int MUC = nunits / NL;  // (units per loader wave: nunits is a power of two, NL 1 or 2)
                bool all_done = false;
                if constexpr (RB > 0 && MUC * NL == nunits) {
                    // the common chunk: every unit of the wave inside the signal, v + m or v - m, whole chunks
                    if ((fuse == 1 || fuse == 2) && MU == MUC && fm == ((1ull << MUC) - 1ull) && lanes == 64) {
                        double2 bs[MUC];
#pragma unroll
                        for (int j = 0; j < MUC; ++j) {
                            bs[j] = double2{0.0, 0.0};
                            if (fuse_sine) {
                                const int u = q + j * NL;
                                bs[j] = double2{l.gtab[(u * 16 + (k & 15)) * 2], l.gtab[(u * 16 + (k & 15)) * 2 + 1]};
                            }
                        }
                        const uint32_t la0 = ring_b + (uint32_t)(q * RU) * (uint32_t)RB + (uint32_t)rho0 * 8u + lane16;
                        rsos_add_units<RB, RU, NL, 0, MUC>(la0, la0 + 8u * (uint32_t)RB, bs, d0, d1, fuse_sine != 0, gconst, fuse == 2);
                        all_done = true;
                    }
                }
#pragma unroll 1
                for (int jb = 0; jb < MU && !all_done; jb += 4) {
                    double2 bsr[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        bsr[i] = double2{0.0, 0.0};
                        if (fuse_sine && ((fm >> (jb + i)) & 1)) {
                            const int u = q + (jb + i) * NL;
                            bsr[i] = double2{l.gtab[(u * 16 + (k & 15)) * 2], l.gtab[(u * 16 + (k & 15)) * 2 + 1]};
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!((fm >> (jb + i)) & 1)) continue;
                        const int u = q + (jb + i) * NL;
                        v2d gn;
                        if (fuse_sine) {
                            gn[0] = fma(bsr[i].x, d0.y, bsr[i].y * d0.x);
                            gn[1] = fma(bsr[i].x, d1.y, bsr[i].y * d1.x);
                        } else
                            gn[0] = gn[1] = gconst;
                        if (lane < lanes) {
                            const uint32_t la = ring_b + (uint32_t)(u * RU) * row_bytes + (uint32_t)rho0 * 8u + lane16;
                            switch (fuse) {
                            case 0: rsos_rmw<RU, 0>(la, row_bytes, gn); break;
                            case 1: rsos_add<RU>(la, row_bytes, gn); break;   // v + m: the LDS adds (no fp64 vector instruction)
                            case 2: rsos_add<RU>(la, row_bytes, -gn); break;  // v - m
                            default: rsos_rmw<RU, 3>(la, row_bytes, gn); break;
                            }
                        }
                    }
                }
                gain_done = true;
            }
        }
        if (gain && !gain_done) {
            for (int j = 0; j < MU; ++j) {
                if (!(k >= u_klo(j) && k < u_khi(j))) {
                    // (the summand ring -- see below -- of a chunk the general path staged with the step applied, or of zeros: zero)
                    if (SRC32 && sring && lane < lanes)
                        asm volatile("ds_write_b64 %0, %1" ::"v"(u_lds(j) + half_bytes + (uint32_t)rho0 * 4u + (uint32_t)lane * 8u), "v"(float2{0.f, 0.f}) : "memory");
                    continue;
                }
                const int u = q + j * NL;
                v2d gn;
                if (fuse_sine) {
                    const double2 bs = j == 0 ? bs0 : j == 1 ? bs1 : double2{l.gtab[(u * 16 + (k & 15)) * 2], l.gtab[(u * 16 + (k & 15)) * 2 + 1]};
                    gn[0] = fma(bs.x, d0.y, bs.y * d0.x);
                    gn[1] = fma(bs.x, d1.y, bs.y * d1.x);
                } else
                    gn[0] = gn[1] = gconst;
                if (src32 && ring32 && sring) {
                    // The summand ring (RsSos::sring: v + m / v - m on a ring of Float32 samples): the operand's two frames of this
                    // lane go into the FREE half of the unit's first ring row (Float32 samples fill half a row), and the y waves
                    // add them to their window operands -- one LDS write per unit here instead of a read, an add and a write per
                    // row and lane next to the chain wave's MFMAs (the step in place cost 0.16 ms of the Float32 headline's 0.97).
                    if (lane < lanes) {
                        const float g0 = fuse == 2 ? -(float)gn[0] : (float)gn[0], g1 = fuse == 2 ? -(float)gn[1] : (float)gn[1];
                        asm volatile("ds_write_b64 %0, %1" ::"v"(u_lds(j) + half_bytes + (uint32_t)rho0 * 4u + (uint32_t)lane * 8u), "v"(float2{g0, g1}) : "memory");
                    }
                } else if (src32 && ring32) {  // (the step on the Float32 samples where they lie: RsSos::f32m)
                    if (lane < lanes) {
                        const uint32_t la = u_lds(j) + (uint32_t)rho0 * 4u + (uint32_t)lane * 8u;
                        const float g0 = (float)gn[0], g1 = (float)gn[1];
                        if constexpr (RB > 0) {
                            switch (fuse) {
                            case 0: rsos_step32_imm<RB, RU, 0>(la, g0, g1); break;
                            case 1: rsos_step32_imm<RB, RU, 1>(la, g0, g1); break;
                            case 2: rsos_step32_imm<RB, RU, 1>(la, -g0, -g1); break;
                            default: rsos_step32_imm<RB, RU, 3>(la, g0, g1); break;
                            }
                        } else {
                            switch (fuse) {
                            case 0: rsos_step32<RU, 0>(la, row_bytes, g0, g1); break;
                            case 1: rsos_step32<RU, 1>(la, row_bytes, g0, g1); break;
                            case 2: rsos_step32<RU, 1>(la, row_bytes, -g0, -g1); break;
                            default: rsos_step32<RU, 3>(la, row_bytes, g0, g1); break;
                            }
                        }
                    }
                } else if constexpr (src32) {
                    const uint32_t slot = u_lds(j) + (uint32_t)rho0 * 8u;
                    switch (fuse) {
                    case 0: rsos_widen<RU, 0>(slot, row_bytes, lane, gn[0], gn[1]); break;
                    case 1: rsos_widen<RU, 1>(slot, row_bytes, lane, gn[0], gn[1]); break;
                    case 2: rsos_widen<RU, 2>(slot, row_bytes, lane, gn[0], gn[1]); break;
                    case 3: rsos_widen<RU, 3>(slot, row_bytes, lane, gn[0], gn[1]); break;
                    default: rsos_widen<RU, -1>(slot, row_bytes, lane, 0.0, 0.0); break;
                    }
                } else if (lane < lanes) {
                    const uint32_t la = u_lds(j) + (uint32_t)rho0 * 8u + lane16;
                    switch (fuse) {
                    case 0: rsos_rmw<RU, 0>(la, row_bytes, gn); break;
                    case 1:  // v + m: the LDS adds (no fp64 vector instruction)
                        if constexpr (RB > 0) rsos_add_imm<RB, 0, RU>(la, gn[0], gn[1]);
                        else rsos_add<RU>(la, row_bytes, gn);
                        break;
                    case 2:  // v - m
                        if constexpr (RB > 0) rsos_add_imm<RB, 0, RU>(la, -gn[0], -gn[1]);
                        else rsos_add<RU>(la, row_bytes, -gn);
                        break;
                    default: rsos_rmw<RU, 3>(la, row_bytes, gn); break;
                    }
                }
            }
        }
        flag_st(fl_base + 4 * (kRsosFlagLdp + q), (k + 1) * CH);
        rsos_stamp(trace, wave, k, 4, 40);
    };
    if constexpr (MODE == 2 && A2) {
        // the second array's wave of the units q, q + 2, ... (eight rows in all: one unit of eight channels, two of four, four of
        // two): three register sets, each a chunk of those rows (16 bytes per lane and row), loaded three chunks ahead; a set is
        // applied to its chunk once the loader has reported the chunk landed and no more of this wave's loads are outstanding
        // than were issued behind the set's
        constexpr int MUA = (16 / RU) / NL;
        const char* const base2 = (const char*)rfl64((int64_t)(uintptr_t)C0.base2);
        const int64_t cs2 = rfl64(C0.cstride2), df2 = rfl64(C0.df2);
        v2d b0[8], b1[8], b2[8];
        auto ld = [&](int k, v2d (&b)[8]) __attribute__((always_inline)) -> int {
            int n = 0;
            if (k >= NK) return 0;
#pragma unroll
            for (int j = 0; j < MUA; ++j) {
                if (!(k >= u_klo(j) && k < u_khi(j))) continue;
                const char* row = base2 + ((((int64_t)u_ch0(j) * cs2 + df2 + u_Au(j)) << 3) + ((int64_t)k << 10));
#pragma unroll
                for (int c = 0; c < RU; ++c) {
                    // (the row's address complete in its scalar pair BEFORE the load: seen without this, the high half's
                    //  add-with-carry scheduled behind the load that reads the pair)
                    uint64_t rc = (uint64_t)(uintptr_t)(row + (int64_t)c * cs2 * 8);
                    asm volatile("" : "+s"(rc));
                    asm volatile("global_load_dwordx4 %0, %1, %2" SO_LD_NT : "=v"(b[j * RU + c]) : "v"(lane16), "s"(rc) : "memory");
                }
                n += RU;
            }
            return n;
        };
        auto ap = [&](int k, int rho0, v2d (&b)[8], int nself, int younger) __attribute__((always_inline)) {
            int sp = 0;
            while (uni(flag_ld(fl_base + 4 * (kRsosFlagLnd + q))) < k + 1 && !(debug & 32)) SO_SPIN_PAUSE(sp, 2, 1 << 22);
            if (nself && !(debug & 2)) {
                wait_vmcnt_le60(younger);
#pragma unroll
                for (int j = 0; j < MUA; ++j) {
                    if (!(k >= u_klo(j) && k < u_khi(j))) continue;  // (staged by the general path, the step applied)
                    const uint32_t la = u_lds(j) + (uint32_t)rho0 * 8u + lane16;
                    if (fuse == 1) {  // v + m
#pragma unroll
                        for (int c = 0; c < RU; ++c)
                            asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:8" ::"v"(la + (uint32_t)c * row_bytes), "v"(b[j * RU + c][0]), "v"(b[j * RU + c][1]) : "memory");
                    } else if (fuse == 2) {  // v - m
#pragma unroll
                        for (int c = 0; c < RU; ++c)
                            asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:8" ::"v"(la + (uint32_t)c * row_bytes), "v"(-b[j * RU + c][0]), "v"(-b[j * RU + c][1]) : "memory");
                    } else {  // v * m, m - v: read, one operation, written back
#pragma unroll
                        for (int c = 0; c < RU; ++c) {
                            const uint32_t a = la + (uint32_t)c * row_bytes;
                            v2d vv[1] = {lds_ld16(a)};
                            lds_wait(vv);
                            vv[0] = fuse == 0 ? vv[0] * b[j * RU + c] : b[j * RU + c] - vv[0];
                            lds_pin(vv);
                            lds_st16(a, vv[0]);
                        }
                    }
                }
            }
            flag_st(fl_base + 4 * (kRsosFlagLdp + q), (k + 1) * CH);
        };
        int n0 = ld(0, b0), n1 = ld(1, b1), n2 = ld(2, b2);
        int rho = 0;
        for (int k = 0; k < NK; k += 3) {
            ap(k, rho, b0, n0, n1 + n2);
            n0 = ld(k + 3, b0);
            rho = rho + CH == RING ? 0 : rho + CH;
            if (k + 1 < NK) {
                ap(k + 1, rho, b1, n1, n2 + n0);
                n1 = ld(k + 4, b1);
                rho = rho + CH == RING ? 0 : rho + CH;
            }
            if (k + 2 < NK) {
                ap(k + 2, rho, b2, n2, n0 + n1);
                n2 = ld(k + 5, b2);
                rho = rho + CH == RING ? 0 : rho + CH;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing of this wave in flight when it leaves: the registers are the next group's)
        __builtin_amdgcn_s_setprio(0);
        return;
    }
    if constexpr (MODE == 2) {  // the step wave: chunk after chunk as the loader reports them landed
        int rho = 0, spins2 = 0;
        for (int k = 0; k < NK; ++k) {
            spins2 = 0;
            while (uni(flag_ld(fl_base + 4 * (kRsosFlagLnd + q))) < k + 1 && !(debug & 32)) SO_SPIN_PAUSE(spins2, 2, 1 << 22);
            retire(k, rho, 0);
            rho = rho + CH == RING ? 0 : rho + CH;
        }
        __builtin_amdgcn_s_setprio(0);
        return;
    }
    int issued = 0, retired = 0;
    int rho_i = 0, rho_r = 0;       // ring positions of the next chunk to issue / to retire
    int c0n = 0, c1n = 0, c2n = 0;  // DMA instructions of the youngest, second and third youngest chunk in flight
    const int depth = uni(g.depth);
    int minrd = 0;  // what the y waves no longer need (cached)
    int spins = 0;
    [[maybe_unused]] int cnt_idle = 0;
    [[maybe_unused]] const long long cyc0 = SO_RSOS_COUNT ? clock64() : 0;
    for (;;) {
        // a chunk that has landed is published before anything else is issued (vmcnt read without waiting): with "issue
        // while there is room, then retire" a landed chunk waited for up to `depth` chunk issues of 16 LDS-DMA instructions
        // each -- and the y waves for it (SIGOPS_RSOS_DEPTH=2 was 7 % faster than 4)
        if (retired < issued && !(debug & 16384)) {
            const int inflight = issued - retired;
            const int allowed = (inflight > 1 ? c0n : 0) + (inflight > 2 ? c1n : 0) + (inflight > 3 ? c2n : 0);
            if (vmcnt_now() <= allowed) {
                retire(retired, rho_r, allowed);
                rho_r = rho_r + CH == RING ? 0 : rho_r + CH;
                ++retired;
                spins = 0;
                continue;
            }
        }
        bool can = issued < NK && issued - retired < depth;
        if (can && (issued + 1) * CH > RING) {  // ring space: the chunk overwrites positions rho - ring
            const int needrd = (issued + 1) * CH - RING;
            if (minrd < needrd) {
                const int v = lane < NY ? flag_ld(fl_base + 4 * (kRsosFlagYrd + lane)) : 0x7fffffff;
                minrd = wave_min(v, NY);
            }
            can = minrd >= needrd || (debug & 32);
        }
        if (can) {
            spins = 0;
            rsos_stamp(trace, wave, issued, 0, 40);
            const int n = issue(issued, rho_i);
            rsos_stamp(trace, wave, issued, 1, 40);
            rho_i = rho_i + CH == RING ? 0 : rho_i + CH;
            c2n = c1n;
            c1n = c0n;
            c0n = n;
            ++issued;
            continue;
        }
        if (retired < issued) {
            const int inflight = issued - retired;
            const int allowed = (inflight > 1 ? c0n : 0) + (inflight > 2 ? c1n : 0) + (inflight > 3 ? c2n : 0);
            retire(retired, rho_r, allowed);
            rho_r = rho_r + CH == RING ? 0 : rho_r + CH;
            ++retired;
            spins = 0;
            continue;
        }
        if (issued >= NK) break;
        SO_SPIN_PAUSE(spins, 2, 1 << 22);
        if constexpr (SO_RSOS_COUNT) ++cnt_idle;
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (SO_RSOS_COUNT) {
        if (lane == 0) {
            rsos_count_out(trace, wave, 0, 0, cnt_idle);
            rsos_count_out(trace, wave, 0, 4, clock64() - cyc0);
        }
    }
}

// =========================== helper wave ===========================
// (16-wave geometry with RsSos::help: wave 12, the fourth wave of the chain's SIMD.)  D . X of the blocks whose y waves
// hand over X itself -- residues kRsosHres0/1/2 of every round of NY blocks, in block order: four MFMAs that depend on
// nothing but X, on the SIMD whose matrix pipe the chain's dependent steps leave idle two thirds of the time.  The result
// replaces X in the block's exchange slot and the slot's counter tells the chain: what the y wave would have done itself.
template <int NY>
__device__ __attribute__((noinline)) void rsos_helper(RsosShared* sh_, double* dyn) {
    constexpr int NX = 2 * NY + 1;
    const int lane = threadIdx.x & 63;
    SO_LDS RsosShared* const sh = (SO_LDS RsosShared*)rsos_lds_ptr(sh_);
    const SO_LDS RsSos& g = sh->g;
    const RsosLds l = rsos_carve(dyn, 0, uni(g.rpitch), NX, rsos_nss(NY));
    const int NB = (uni(g.wp) + (int)rfl64(g.pr)) * uni(g.ngroups);
    const double SO_GLB* mats = (const double SO_GLB*)rfl64((int64_t)(uintptr_t)g.mats);
    const uint32_t fl_base = (uint32_t)(uintptr_t)sh->flags;
    const int debug = uni(g.debug);
    const bool f32m = uni(g.f32m) != 0 && uni(g.out_f32) != 0;
    double Dk[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) Dk[v] = f32m ? mats[(lane >> 4) * 64 + v * 16 + (lane & 15)] : mats[v * 64 + lane];
    __builtin_amdgcn_s_setprio(1);
    constexpr int res[3] = {kRsosHres0, kRsosHres1, kRsosHres2};
    int hslot[3] = {kRsosHres0 % NX, kRsosHres1 % NX, kRsosHres2 % NX}, hring = 0;
    for (int b0 = 0; b0 < NB; b0 += NY) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int bh = b0 + res[j];
            if (bh < NB) {
                int spins = 0;
                while (uni(flag_ld(fl_base + 4 * (kRsosFlagXh + j))) < bh + 1 && !(debug & 16)) SO_SPIN_PAUSE(spins, 2, 1 << 22);
                double xr[4];
#pragma unroll
                for (int v = 0; v < 3; ++v) xr[v] = l.xs[hslot[j] * 192 + v * 64 + lane];
                xr[3] = l.xh[(j * kRsosHslots + hring) * 64 + lane];
                v4d dh = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int v = 0; v < 4; ++v) dh = __builtin_amdgcn_mfma_f64_16x16x4f64(Dk[v], xr[v], dh, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 3; ++v) l.xs[hslot[j] * 192 + v * 64 + lane] = dh[v];
                flag_st(fl_base + 4 * (kRsosFlagXseq + hslot[j]), bh + 1);
            }
            hslot[j] += NY;
            if (hslot[j] >= NX) hslot[j] -= NX;
        }
        hring = hring + 1 == kRsosHslots ? 0 : hring + 1;
    }
    __builtin_amdgcn_s_setprio(0);
}

// =========================== y waves ===========================
// CYC > 0: the wave's blocks cycle through CYC phase groups of the period (yi, yi + NY, ... modulo ngroups) and their
// taps stay in registers for the whole kernel; CYC == 0: taps of any phase from the LDS table.
// F32M (Float32 results over a ring of Float32 samples, RsSos::f32m): the resampling product X = Tap^T . Win on
// v_mfma_f32_16x16x4_f32 -- taps rounded to Float32 once, Float32 window operands straight from the ring, Float32
// accumulators: 32 cycles per instruction where the Float64 one takes 64, and no widening of the window.  Its RESULT map is
// row = 4 (lane >> 4) + reg where the Float64 instruction's is (lane >> 4) + 4 reg: register v of X, widened, is still k-step v of
// the products that follow (D . X, X^T T^T: all Float64, like the cascade) -- with the k index of THEIR other operand
// permuted the same way (Dk, Tk below are read from the matrices' table in that order).
template <int KS, int NY, int NL, typename TO, int CYC, bool F32M = false, bool HW = false>
__device__ __attribute__((noinline)) void rsos_ywave(RsosShared* sh_, double* dyn, int64_t G_, int yi_) {
    static_assert(!F32M || (sizeof(TO) == 4 && CYC > 0), "the Float32 MFMA form: Float32 results, taps in registers");
    constexpr int NX = 2 * NY + 1;
    const int lane = threadIdx.x & 63;
    const int wave = uni(threadIdx.x >> 6);
    SO_LDS RsosShared* const sh = (SO_LDS RsosShared*)rsos_lds_ptr(sh_);
    const int64_t G = rfl64(G_);
    const int yi = uni(yi_);
    const SO_LDS RsSos& g = sh->g;
    const int ngroups = uni(g.ngroups), rpitch = uni(g.rpitch), RING = uni(g.ring);
    const RsosLds l = rsos_carve(dyn, CYC > 0 ? 0 : ngroups * KS * 64, rpitch, NX, rsos_nss(NY));
    const int ct = uni(g.ct), wp = uni(g.wp), debug = uni(g.debug);
    const int M = (int)rfl64(g.M);
    const int NB = (wp + (int)rfl64(g.pr)) * ngroups;
    constexpr int kw = 4 * KS;
    const int ulo_kw = uni(g.ulo) + (kw - 1);
    const int64_t n_out = rfl64(g.n_out), out_pitch = rfl64(g.out_pitch);
    const double SO_GLB* mats = (const double SO_GLB*)rfl64((int64_t)(uintptr_t)g.mats);
    long long* trace = (long long*)rfl64((int64_t)(uintptr_t)g.trace);
    TO SO_GLB* const y = (TO SO_GLB*)rfl64((int64_t)(uintptr_t)sh->y);
    const uint32_t fl_base = (uint32_t)(uintptr_t)sh->flags;
    const SO_LDS DCarrier& C0 = sh->ctl.car[0];
    const int64_t cs0 = rfl64(C0.cstride), df0 = rfl64(C0.df);
    const bool src32 = uni(g.src32) != 0;
    const bool x32 = sizeof(TO) == 4 && uni(g.x32) != 0;
    // (the ring keeps Float32 samples, four bytes a frame: RsSos::ring32 -- Float32-result instantiations only)
    [[maybe_unused]] const bool ring32 = sizeof(TO) == 4 && uni(g.ring32) != 0;
    [[maybe_unused]] const bool sring = F32M && uni(g.sring) != 0;
    [[maybe_unused]] const int srow = ((lane & 15) / (ct < 8 ? ct : 8)) * (ct < 8 ? ct : 8);  // (first ring row of this lane's unit)
    const bool single = uni((int)(C0.base != nullptr && C0.vec_ok && C0.dtype == (src32 ? SO_F32 : SO_F64))) && !(df0 & 1) && uni(g.fuse) >= -1 &&
                        (!src32 || uni(g.chunk) == 128) && (!uni(g.arr2) || rsos_two_ok(C0, uni(g.chunk)));
    const RsosGroup grp = rsos_group(sh, G, single, (int64_t)(rfl64((int64_t)(uintptr_t)C0.base) >> (src32 ? 2 : 3)), cs0, df0);
    const int gq = lane >> 4, n16 = lane & 15;
    double Dk[4], Tk[4], Ck[3];
    const int nkc = (uni(g.debug) & 131072) ? 3 : (2 * uni(g.nsec) + 3) / 4;  // k-steps of the state (rsos_chain's NK)
    // (k-step v, lane (gq, n16) holds index 4 v + gq of the contracted dimension; F32M: 4 gq + v -- see above)
#pragma unroll
    for (int v = 0; v < 4; ++v) Dk[v] = F32M ? mats[(lane >> 4) * 64 + v * 16 + (lane & 15)] : mats[v * 64 + lane];
#pragma unroll
    for (int v = 0; v < 4; ++v) Tk[v] = F32M ? mats[(7 + (lane >> 4)) * 64 + v * 16 + (lane & 15)] : mats[(7 + v) * 64 + lane];
#pragma unroll
    for (int v = 0; v < 3; ++v) Ck[v] = mats[(11 + v) * 64 + lane];
    // this lane's B-operand row ...
    int cl;   // its alignment shift + the lane's k offset
    int nb0;  // blocks of this row before the signal's first output: their resampled samples do not exist (the cascade
              // starts from rest at output 0, whatever the taps of "earlier outputs" would reach)
    {
        const int ri = n16 / ct;
        cl = ((grp.e0m + ri * grp.prMm) & 15) + gq;
        const int64_t ob = grp.ob0 + ri * grp.prL;
        nb0 = ob < 0 ? (int)((-ob) / 16) : 0;
    }
    const SO_LDS char* const ringc = (const SO_LDS char*)l.ring + (uint32_t)n16 * (uint32_t)rpitch * 8u;
    // ... and the four result rows it stores: row gq + 4v, time n16
    // ... as byte offsets from the group's first result element (row 0, output 0 of its first range): the stores take a
    // scalar base, advanced per block by scalar adds, and a 32-bit lane offset -- no 64-bit vector add per store.  (A group
    // whose rows lie more than 4 GB apart -- long results of many channels -- rebuilds the address per store.)
    // ONE lane offset (row gq, time n16) and three wave-uniform steps: row gq + 4 v is (channel, range) = ((gq + 4 v) % ct,
    // (gq + 4 v) / ct), and for every ct in {1, 2, 4, 8, 16} its distance from row gq is the same for all gq -- it goes onto the
    // stores' SCALAR base.  (Four lane offsets and four per-lane block limits in registers were what the 128-register
    // instantiation -- 16 waves: two-channel groups -- spilled: every store was preceded by a scratch reload and an
    // `s_waitcnt vmcnt(0)`, i.e. by a wait for the PREVIOUS store's acknowledgement: the 0.3 ms "store cost" of stereo signals.)
    uint32_t yo0;
    int64_t ystep[4];
    const int64_t ybase_e = (int64_t)(grp.cg * ct) * out_pitch + grp.ob0;
#ifndef SO_Y_EARLY
#define SO_Y_EARLY 0  // (see block(): the pending block's state-dependent MFMAs in front of D . X -- measured, slower)
#endif
#ifndef SO_Y_SADDR
#define SO_Y_SADDR 1
#endif
#ifndef SO_ST_NT
#define SO_ST_NT ""  // (cache-policy bits of the result stores: measured and not set, see kstage.h SO_LD_NT)
#endif
    const bool yfits = SO_Y_SADDR && uni((int)((((int64_t)(ct - 1) * out_pitch + (int64_t)(16 / ct) * grp.prL + 16) * (int64_t)sizeof(TO)) < ((int64_t)1 << 32)));
    int nbs_all;     // block b is stored by every lane of the wave while b < nbs_all (no predicates then)
    {
        int m = 0x7fffffff;
        int64_t off0 = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = gq + 4 * v;
            const int ri = row / ct, cv = row % ct;
            const int64_t ob = grp.ob0 + ri * grp.prL;
            const int64_t off = ((int64_t)cv * out_pitch + (int64_t)ri * grp.prL + n16) * (int64_t)sizeof(TO);
            if (v == 0) off0 = off;
            ystep[v] = rfl64(off - off0);
            const int64_t tl = n_out - ob - n16;  // 16 b < tl
            const int64_t nb = tl <= 0 ? 0 : (tl + 15) / 16;
            m = min(m, (int)(nb > 0x3fffffff ? 0x3fffffff : nb));
        }
        yo0 = (uint32_t)off0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off, 64));
        nbs_all = uni(m);
    }
    // a window's warm-up (RsSos::store_lo): outputs below it are not stored.  Only the first blocks of the first ranges
    // can lie below it: from block nbl_max on every row of the group stores -- the common path
    const int64_t slo = rfl64(g.store_lo);
    int nbl_max = 0;
    if (slo > grp.ob0) {
        const int64_t nb = (slo - grp.ob0 + 15) / 16;
        nbl_max = (int)(nb > 0x3fffffff ? 0x3fffffff : nb);
    }
    int nb0_any;  // some lane's row starts before the signal: blocks below this need the zeroing
    {
        int m = nb0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
        nb0_any = uni(m);
    }
    // helper geometry: this wave hands its X blocks over (hj: which of the three), or is the one that takes them
    constexpr bool HELPABLE = HW && NY == 10 && CYC == 1;  // (its own instantiation: the 12-wave kernel's block is what it was)
    const bool help_on = HELPABLE && uni(g.help) != 0;
    const int hj = !help_on ? -1 : yi == kRsosHres0 ? 0 : yi == kRsosHres1 ? 1 : yi == kRsosHres2 ? 2 : -1;
    int hr = 0;  // (owner: this block's slot of the fourth-register ring; helper: per owner, below)
    int pi = 0, gi = yi;
    while (gi >= ngroups) {
        gi -= ngroups;
        ++pi;
    }
    int slot = yi % NX;
    int avail = 0;
    int wbm = -1;  // the block's window start modulo the ring
    if (yi >= NB) flag_st(fl_base + 4 * (kRsosFlagYrd + yi), 0x7fffffff);  // (no block for this wave: it needs nothing of the ring)
    // (two LDS addresses that never change, in vector registers the compiler cannot rematerialise from their scalars: a
    //  v_mov per use is a vector instruction that waits for a gap between the SIMD's MFMAs)
    uint32_t f_sseq = fl_base + 4 * kRsosFlagSseq, f_yrd = fl_base + 4 * (kRsosFlagYrd + yi);
    asm volatile("" : "+v"(f_sseq), "+v"(f_yrd));
    // Per block of this wave, in this order:
    //   front(b)    one batch of LDS reads (the previous block's state counter and -- speculatively -- its state, then
    //               the 4 KS window samples), KS MFMAs -> X, X to the chain wave;
    //   back(prev)  Y(prev) = X^T T^T (kept since the previous round) + S^T C^T, stored;
    //   T part(b)   X^T T^T of this block, kept for the next round.
    // The front part of a block thus runs one block of this wave (NY blocks of the group) ahead of its back part: the
    // x blocks reach the chain wave a whole round of the y waves before their states are needed.  (With front and back
    // of one block back to back all y waves ended up waiting for the chain together, and then the chain for all of
    // them: a convoy.)  An in-order wave does nothing while it waits, so what a block costs besides its 21 MFMAs is
    // one LDS round trip, the hazard gaps behind three MFMA groups and ~100 other instructions.
    constexpr int NS = rsos_nss(NY);
    int sslot = yi % NS;                 // state slot of the block in hand (b modulo NS)
    int pb_ = -1, ppi_ = 0, pslot_ = 0;  // the block whose back part is due: index, period, STATE slot
    // ... and the address of its first result element in the group's row 0 (bytes; a running scalar: blocks are NY apart)
    uint64_t ycur = (uint64_t)rfl64((int64_t)(uintptr_t)(y + (ybase_e + (int64_t)16 * yi))), ypend = 0;
    [[maybe_unused]] int cnt_in = 0, cnt_in_b = 0, cnt_st = 0, cnt_st_b = 0;
    [[maybe_unused]] const long long cyc0 = SO_RSOS_COUNT ? clock64() : 0;
    // ... and its X^T T^T: two register sets that swap roles from block to block (the set a block fills is the one the
    // next block's back part accumulates into and stores from: no copy between them)
    v4d payA = v4d{0.0, 0.0, 0.0, 0.0}, payB = v4d{0.0, 0.0, 0.0, 0.0};
    // The back part of the pending block pb_ in three pieces: the wait for its state (also what keeps this wave from running
    // away from the chain wave: the exchange slots are reused on the strength of it, so it stays for warm-up blocks, whose
    // result is not computed), Y = X^T T^T + S^T C^T, the stores.
    auto back_wait = [&](int sq, double (&sv)[3]) __attribute__((always_inline)) {
        int spins = 0;
        sq = uni(sq);
        if (sq < pb_ && !(debug & 8)) {  // (the speculative read was early: wait, read again)
            if constexpr (SO_RSOS_COUNT) ++cnt_st_b;
            do {
                SO_SPIN_PAUSE(spins, 1, (debug & 32768) ? 4096 : (1 << 22));
                if constexpr (SO_RSOS_COUNT) ++cnt_st;
                sq = uni(flag_ld(f_sseq));
            } while (sq < pb_);
#pragma unroll
            for (int v = 0; v < 3; ++v) sv[v] = l.ss[pslot_ * 192 + v * 64 + lane];
        }
    };
    // (as many k-steps as the cascade has states / 4: the chain wave writes no more of a state slot -- RsosChain's NK)
    auto back_mfma = [&](const double (&sv)[3], v4d ay) __attribute__((always_inline)) {
        ay = __builtin_amdgcn_mfma_f64_16x16x4f64(sv[0], Ck[0], ay, 0, 0, 0);
        if (nkc > 1) ay = __builtin_amdgcn_mfma_f64_16x16x4f64(sv[1], Ck[1], ay, 0, 0, 0);
        if (nkc > 2) ay = __builtin_amdgcn_mfma_f64_16x16x4f64(sv[2], Ck[2], ay, 0, 0, 0);
        return ay;
    };
    // (fresh: the MFMAs that wrote ay are the last instructions in front of this -- the 18 wait states a Float64 store needs
    //  behind them are inserted here, tied to the accumulators; not fresh: at least eight MFMAs lie in between)
    auto back_store = [&](v4d ay, bool fresh) __attribute__((always_inline)) {
        if (debug & 1) return;
        const int64_t t0 = (int64_t)16 * pb_;
        // (a scalar register pair + the lanes' 32-bit offsets: the store's own address form -- written out, the compiler
        //  widens the offsets once and adds 64-bit lane pointers per store again.  It does not see the MFMA result a
        //  Float64 store reads behind the asm either: hence `fresh`)
        const uint64_t yb = (uint64_t)rfl64((int64_t)ypend);
        if (fresh && yfits && sizeof(TO) == 8) asm volatile("s_nop 15\n\ts_nop 1" : "+v"(ay[0]), "+v"(ay[1]), "+v"(ay[2]), "+v"(ay[3])::"memory");
        // (the output this lane's register v holds: its row's first output + the block's offset + the lane's time)
        auto out_index = [&](int v) __attribute__((always_inline)) { return grp.ob0 + (int64_t)((gq + 4 * v) / ct) * grp.prL + t0 + n16; };
        auto put = [&](int v, TO val) __attribute__((always_inline)) {
            if (yfits) {
                const uint64_t ybv = yb + (uint64_t)ystep[v];  // (scalar)
                if constexpr (sizeof(TO) == 8) asm volatile("global_store_dwordx2 %0, %1, %2" SO_ST_NT ::"v"(yo0), "v"(val), "s"(ybv) : "memory");
                else asm volatile("global_store_dword %0, %1, %2" ::"v"(yo0), "v"(val), "s"(ybv) : "memory");
            } else {
                // (results whose rows lie more than 4 GB apart: the address per store, from a row index the compiler cannot see
                //  through -- hoisted out of the loop, four 64-bit lane pointers are what gets spilled)
                int row = gq + 4 * v;
                asm volatile("" : "+v"(row));
                y[(int64_t)(grp.cg * ct + row % ct) * out_pitch + grp.ob0 + (int64_t)(row / ct) * grp.prL + n16 + t0] = val;
            }
        };
        if (pb_ >= nbl_max && pb_ < nbs_all) {  // (the common path: every lane stores, no predicates)
#pragma unroll
            for (int v = 0; v < 4; ++v) put(v, (TO)ay[v]);
        } else {  // (the first blocks of a window's first ranges, the last blocks of the signal: output by output)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t m = out_index(v);
                if (m >= slo && m < n_out) put(v, (TO)ay[v]);
            }
        }
    };
    auto back = [&](int sq, double (&sv)[3], v4d& pay_) __attribute__((always_inline)) {
        back_wait(sq, sv);
        if (ppi_ < wp) return;  // (a warm-up block: nothing to store)
        rsos_stamp(trace, wave, pb_ / NY, 4, 40);
        back_store(back_mfma(sv, pay_), true);
        rsos_stamp(trace, wave, pb_ / NY, 5, 40);
    };
    // (je / jn: the window ends of this block's phase group and of the wave's next block's -- loop constants of a wave whose
    //  blocks cycle through CYC groups, read from LDS otherwise)
    using TapT = std::conditional_t<F32M, float, double>;
    auto block = [&](int b, const TapT (&at)[KS], v4d& pin, v4d& pout, int je, int jn) __attribute__((always_inline)) {
        const int wb = pi * M + je - ulo_kw;
        const int need = wb + kw + 16;
        rsos_stamp(trace, wave, b / NY, 0, 40);
        int spins = 0;
        if constexpr (SO_RSOS_COUNT) cnt_in_b += avail < need ? 1 : 0;
        while (avail < need && !(debug & 16)) {
            const int v = lane < NL ? flag_ld(fl_base + 4 * (kRsosFlagLdp + lane)) : 0x7fffffff;
            avail = wave_min(v, NL);
            if (avail < need) {
                SO_SPIN_PAUSE(spins, 2, 1 << 22);
                if constexpr (SO_RSOS_COUNT) ++cnt_in;
            }
        }
        rsos_stamp(trace, wave, b / NY, 1, 40);
        // ---- the block's LDS reads, all at once ----
        int sq = 0;
        double sv[3] = {0.0, 0.0, 0.0};
        if (pb_ >= 0) {
            sq = flag_ld(f_sseq);
#pragma unroll
            for (int v = 0; v < 3; ++v) sv[v] = l.ss[pslot_ * 192 + v * 64 + lane];
        }
        if (wbm < 0) wbm = wb % RING;
        const int pos = wbm + cl;  // < RING + 20: one conditional subtraction per read wraps it
        double bx[KS];
        bool have_bx = false;
        [[maybe_unused]] float bf32[KS];
        if constexpr (F32M) {  // (the ring keeps Float32 samples: they ARE the B operands)
            if (wbm + 20 + kw <= RING) {
                const SO_LDS float* p0 = (const SO_LDS float*)(ringc + (uint32_t)pos * 4u);
#pragma unroll
                for (int s = 0; s < KS; ++s) bf32[s] = p0[4 * s];
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int p_ = pos + 4 * s;
                    bf32[s] = *(const SO_LDS float*)(ringc + (uint32_t)(p_ >= RING ? p_ - RING : p_) * 4u);
                }
            }
            if (sring) {  // (the fused v + m: the loader left m -- Float32, per range -- in the free half of the unit's first row)
                float sn[KS];
                const SO_LDS char* const sbase = (const SO_LDS char*)l.ring + (uint32_t)srow * (uint32_t)rpitch * 8u + (uint32_t)rpitch * 4u;
                if (wbm + 20 + kw <= RING) {
                    const SO_LDS float* p0 = (const SO_LDS float*)(sbase + (uint32_t)pos * 4u);
#pragma unroll
                    for (int s = 0; s < KS; ++s) sn[s] = p0[4 * s];
                } else {
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const int p_ = pos + 4 * s;
                        sn[s] = *(const SO_LDS float*)(sbase + (uint32_t)(p_ >= RING ? p_ - RING : p_) * 4u);
                    }
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) bf32[s] += sn[s];
            }
            have_bx = true;
        } else if constexpr (sizeof(TO) == 4) {
            if (ring32) {  // Float32 samples in the ring: widened here (KS conversions per block; the loader's widening pass of
                           // every chunk, one wave next to the chain's MFMAs, was the slower place for them)
                float bf[KS];
                if (wbm + 20 + kw <= RING) {
                    const SO_LDS float* p0 = (const SO_LDS float*)(ringc + (uint32_t)pos * 4u);
#pragma unroll
                    for (int s = 0; s < KS; ++s) bf[s] = p0[4 * s];
                } else {
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const int p_ = pos + 4 * s;
                        bf[s] = *(const SO_LDS float*)(ringc + (uint32_t)(p_ >= RING ? p_ - RING : p_) * 4u);
                    }
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) bx[s] = (double)bf[s];
                have_bx = true;
            }
        }
        if (have_bx) {
        } else if (wbm + 20 + kw <= RING) {  // (no lane's window wraps: one address, immediate offsets)
            const SO_LDS double* p0 = (const SO_LDS double*)(ringc + (uint32_t)pos * 8u);
#pragma unroll
            for (int s = 0; s < KS; ++s) bx[s] = p0[4 * s];
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int p_ = pos + 4 * s;
                bx[s] = *(const SO_LDS double*)(ringc + (uint32_t)(p_ >= RING ? p_ - RING : p_) * 8u);
            }
        }
        // ---- resample: X[t][row] = sum_k Tap[t][k] Win[k][row] (LDS returns in order: counted lgkmcnt waits) ----
        v4d ax = v4d{0.0, 0.0, 0.0, 0.0};
        if constexpr (F32M) {
            // two accumulator chains (the instruction's dependent latency is 40 cycles against 32 of issue), added at the end
            v4f a0 = v4f{0.f, 0.f, 0.f, 0.f}, a1 = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s & 1) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[s], bf32[s], a1, 0, 0, 0);
                else a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[s], bf32[s], a0, 0, 0, 0);
            }
            const v4f af = a0 + a1;
#pragma unroll
            for (int v = 0; v < 4; ++v) ax[v] = (double)af[v];
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) ax = __builtin_amdgcn_mfma_f64_16x16x4f64(at[s], bx[s], ax, 0, 0, 0);
        }
        if constexpr (sizeof(TO) == 4 && !F32M) {
            // a Float32 signal all the way: the reference's resampler hands the filter Float32 samples (K3's store rounds the
            // same accumulators the same way); the Float64-result instantiations -- the headline's -- do not carry the test
            if (x32) {
#pragma unroll
                for (int v = 0; v < 4; ++v) ax[v] = (double)(float)ax[v];
            }
        }
        // ---- (-DSO_Y_EARLY=1) the pending block's state-dependent product HERE where its state has arrived: D . X and
        //      X^T T^T below read the resampler's accumulators as A / B operands and cannot issue until the last of its MFMAs
        //      has left the pipe; these three depend on nothing in flight, and the stores then need no wait states.  Measured:
        //      the y waves alone 0.899 -> 0.896 ms, the whole kernel 0.964 -> 0.978 -- D . x reaches the chain three MFMAs later.
        bool early = false;
        v4d ay_early = pin;
        if (SO_Y_EARLY && pb_ >= 0 && ppi_ >= wp && (uni(sq) >= pb_ || (debug & 8))) {
            ay_early = back_mfma(sv, pin);
            early = true;
        }
        // next block of this wave: its window start is what this wave still needs of the ring
        int pi2 = pi, gi2 = gi + NY;
        while (gi2 >= ngroups) {
            gi2 -= ngroups;
            ++pi2;
        }
        const int wb2 = b + NY < NB ? pi2 * M + jn - ulo_kw : 0x7fffffff;
        flag_st(f_yrd, wb2);
        if (b + NY < NB) {
            wbm += wb2 - wb;
            while (wbm >= RING) wbm -= RING;
        }
        if (b < nb0_any) {  // (a real branch -- the first blocks of the signal's first range only: if-converted it is eight
                            //  selects behind an MFMA-result hazard in every block of the kernel)
            asm volatile("" ::: "memory");
            if (b < nb0) ax = v4d{0.0, 0.0, 0.0, 0.0};
            asm volatile("" : "+v"(ax));
        }
        rsos_stamp(trace, wave, b / NY, 2, 40);
        // ---- D . X for the chain wave (its share of the recurrence that does not depend on the state) ----
        // (helper geometry, a wave of the three: X itself goes to the exchange slot below, the helper wave forms D . X)
        v4d dx = v4d{0.0, 0.0, 0.0, 0.0};
        if (!HELPABLE || hj < 0) {
#pragma unroll
            for (int v = 0; v < 4; ++v) dx = __builtin_amdgcn_mfma_f64_16x16x4f64(Dk[v], ax[v], dx, 0, 0, 0);
        }
        // ---- X^T T^T of this block (kept for the next round) ----
        // (also for a warm-up block, whose result is not stored: 7 % of the blocks; skipping the four products there makes
        //  the set a conditional value, and the register allocator pays for that with four copies behind an MFMA-result
        //  hazard in every block)
        pout = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[0], Tk[0], v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
#pragma unroll
        for (int v = 1; v < 4; ++v) pout = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[v], Tk[v], pout, 0, 0, 0);
        if (!HELPABLE || hj < 0) {
#pragma unroll
            for (int v = 0; v < 3; ++v) l.xs[slot * 192 + v * 64 + lane] = dx[v];
            flag_st(fl_base + 4 * (kRsosFlagXseq + slot), b + 1);
        } else {
#pragma unroll
            for (int v = 0; v < 3; ++v) l.xs[slot * 192 + v * 64 + lane] = ax[v];
            l.xh[(hj * kRsosHslots + hr) * 64 + lane] = ax[3];
            flag_st(fl_base + 4 * (kRsosFlagXh + hj), b + 1);
            hr = hr + 1 == kRsosHslots ? 0 : hr + 1;
        }
        // ---- the previous block's result ----
        if (early) back_store(ay_early, false);
        else if (pb_ >= 0) back(sq, sv, pin);
        rsos_stamp(trace, wave, b / NY, 3, 40);
        pb_ = b;
        ppi_ = pi;
        pslot_ = sslot;
        ypend = ycur;
        ycur += (uint64_t)(16 * NY) * sizeof(TO);
        sslot += NY % NS;
        if (sslot >= NS) sslot -= NS;
        pi = pi2;
        gi = gi2;
        slot += NY;  // (b + NY) mod (2 NY + 1)
        if (slot >= NX) slot -= NX;
    };
    int cur = 0;  // the set the pending block's T part is in: 0 payA, 1 payB
    auto last_back = [&]() __attribute__((always_inline)) {
        if (pb_ < 0) return;
        double sv[3];
        const int sq = flag_ld(f_sseq);
#pragma unroll
        for (int v = 0; v < 3; ++v) sv[v] = l.ss[pslot_ * 192 + v * 64 + lane];
        if (cur) back(sq, sv, payB);
        else back(sq, sv, payA);
    };
    if constexpr (CYC > 0) {
        const double SO_GLB* tab = (const double SO_GLB*)rfl64((int64_t)(uintptr_t)sh->tab);
        TapT treg[CYC][KS];
        int jec[CYC];
#pragma unroll
        for (int c = 0; c < CYC; ++c) {
            const int gc = (yi + c * NY) % ngroups;
            jec[c] = uni(sh->jend[gc]);
#pragma unroll
            for (int s = 0; s < KS; ++s) treg[c][s] = (TapT)tab[((size_t)gc * KS + s) * 64 + lane];
        }
        for (int b = yi; b < NB;) {
#pragma unroll
            for (int k = 0; k < 2 * CYC; ++k) {
                if (b < NB) {
                    if ((k & 1) == 0) block(b, treg[k % CYC], payA, payB, jec[k % CYC], jec[(k + 1) % CYC]);
                    else block(b, treg[k % CYC], payB, payA, jec[k % CYC], jec[(k + 1) % CYC]);
                    cur = (k & 1) ^ 1;
                }
                b += NY;
            }
        }
        last_back();
        if constexpr (SO_RSOS_COUNT) {
            if (lane == 0) {
                rsos_count_out(trace, wave, 0, 0, cnt_in);
                rsos_count_out(trace, wave, 0, 1, cnt_in_b);
                rsos_count_out(trace, wave, 0, 2, cnt_st);
                rsos_count_out(trace, wave, 0, 3, cnt_st_b);
                rsos_count_out(trace, wave, 0, 4, clock64() - cyc0);
            }
        }
    } else if constexpr (!F32M) {
        for (int b = yi; b < NB; b += NY) {
            double at[KS];
            const SO_LDS double* tp = l.taps + gi * KS * 64 + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) at[s] = tp[s * 64];
            int gn = gi + NY;
            while (gn >= ngroups) gn -= ngroups;
            const int je = uni(sh->jend[gi]), jn = uni(sh->jend[gn]);
            if (cur == 0) block(b, at, payA, payB, je, jn);
            else block(b, at, payB, payA, je, jn);
            cur ^= 1;
        }
        last_back();
    }
}


// NW: 8 / 12 / 16 waves, or 17 = the HELPER geometry: 16 waves of 128 registers with ONE loader (wave 4), ten y waves (1 - 3,
// 5 - 7, 8 - 11: wave 8 shares the chain's SIMD) and wave 12 -- the chain's SIMD again -- forming D . X for three of them
// (rsos_helper): what VERDICT round 5 asked for, a second wave on SIMD 0 that uses the matrix cycles the chain leaves.
constexpr int rsos_nthreads(int nw) { return (nw == 17 ? 16 : nw) * 64; }
// (the kernel's body: sequence groups G0, G0 + Gstep, ... of ONE filter -- the whole grid's for k_rsos, a share of it for a member
//  of k_rsos_batch)
template <int KS, int NW, typename TO, int CYC>
__device__ __forceinline__ void rsos_body(const double* __restrict__ tab, const int* __restrict__ jend_g, const RsSos& g, TO* __restrict__ y,
                                          const RsGlobalTables& gsrc, int64_t G0, int64_t Gstep) {
    constexpr bool HW = NW == 17;
    constexpr int NT = rsos_nthreads(NW);
    constexpr int NY = NW >= 16 ? 10 : NW - 2, NL = NW == 16 ? 2 : 1, NX = 2 * NY + 1;
    // wave 0: chain; wave 4 (and 8 at 16 waves): loaders; the others: y waves (at 16 waves: ten of them -- one more on the
    // chain's SIMD, three on each of the others --, waves 13..15 have nothing to do)
    extern __shared__ double lds_raw[];
    __shared__ RsosShared sh;
    const int wave = uni(threadIdx.x >> 6);
    // ---- once per workgroup: arguments, taps, window ends, control block ----
    {
        if (threadIdx.x == 0) {
            sh.g = g;  // (member-wise from scalar registers: taking the argument's address would move it to scratch)
            sh.gcar = gsrc.car;
            sh.gops = gsrc.ops;
            sh.gleaves = gsrc.leaves;
            sh.y = (void*)y;
            sh.tab = tab;
        }
    }
    if constexpr (CYC == 0)
        for (int i = threadIdx.x; i < g.ngroups * KS * 64; i += NT) lds_raw[i] = tab[i];
    for (int i = threadIdx.x; i < g.ngroups; i += NT) sh.jend[i] = jend_g[i];
    {
        const int* src = reinterpret_cast<const int*>(gsrc.ctl);
        int* dst = reinterpret_cast<int*>(&sh.ctl);
        for (int i = threadIdx.x; i < (int)(sizeof(RsCtl) / 4); i += NT) dst[i] = src[i];
    }
    const int64_t ncg = g.nch / g.ct;
    const int64_t ngrp = ncg * ((g.nranges + g.rgs - 1) / g.rgs);
    double* const ss = lds_raw + (CYC > 0 ? 0 : (size_t)g.ngroups * KS * 64) + (size_t)16 * g.rpitch + (size_t)NX * 192;  // (state slot 0)
    const int ru = g.ct < 8 ? g.ct : 8;
    for (int64_t G = G0; G < ngrp; G += Gstep) {
        __syncthreads();  // (the previous group's LDS traffic is over; the first time: the tables are in place)
        if (threadIdx.x < kRsosFlags) sh.flags[threadIdx.x] = 0;
        if (threadIdx.x < 192) ss[threadIdx.x] = 0.0;  // s_0 = 0
        __syncthreads();
        if (wave == 0) {
            if (!(g.debug & 64)) {
                const int nk = (g.debug & 131072) ? 3 : (2 * g.nsec + 3) / 4;
                if (nk <= 1) rsos_chain<NY, 1>(&sh, lds_raw, G);
                else if (nk == 2) rsos_chain<NY, 2>(&sh, lds_raw, G);
                else rsos_chain<NY, 3>(&sh, lds_raw, G);
            }
        } else if (NW == 16 && (wave == 13 || wave == 14)) {
            // (the step waves of two-channel / four-channel groups: RsSos::gsplit)
            if (!g.gsplit || (g.debug & 256) || g.src32) continue;
            const int q = wave - 13;
            if (g.arr2) {  // (the second array's waves)
                if constexpr (NL == 2) {
                    if (ru == 8) rsos_loader<NY, NL, 8, false, 0, 2, true>(&sh, lds_raw, G, q);
                    else if (ru == 4) rsos_loader<NY, NL, 4, false, 0, 2, true>(&sh, lds_raw, G, q);
                    else if (ru == 2) rsos_loader<NY, NL, 2, false, 0, 2, true>(&sh, lds_raw, G, q);
                }
            } else if (ru == 2) {
                if (g.rpitch == 770) rsos_loader<NY, NL, 2, false, 770 * 8, 2>(&sh, lds_raw, G, q);
                else if (g.rpitch == 642) rsos_loader<NY, NL, 2, false, 642 * 8, 2>(&sh, lds_raw, G, q);
                else rsos_loader<NY, NL, 2, false, 0, 2>(&sh, lds_raw, G, q);
            } else if (ru == 4) {
                if (g.rpitch == 770) rsos_loader<NY, NL, 4, false, 770 * 8, 2>(&sh, lds_raw, G, q);
                else if (g.rpitch == 642) rsos_loader<NY, NL, 4, false, 642 * 8, 2>(&sh, lds_raw, G, q);
                else rsos_loader<NY, NL, 4, false, 0, 2>(&sh, lds_raw, G, q);
            }
        } else if (wave == 4 || (NW == 16 && wave == 8)) {
            const int q = wave == 4 ? 0 : 1;
            if (g.debug & 256) continue;
            if (NW == 16 && g.gsplit && !g.src32 && ru == 8 && g.arr2) {
                rsos_loader<NY, NL, 8, false, 0, 1>(&sh, lds_raw, G, q);
                continue;
            }
            if (NW == 16 && g.gsplit && !g.src32 && (ru == 2 || ru == 4)) {
                if (ru == 2) {
                    if (g.rpitch == 770) rsos_loader<NY, NL, 2, false, 770 * 8, 1>(&sh, lds_raw, G, q);
                    else if (g.rpitch == 642) rsos_loader<NY, NL, 2, false, 642 * 8, 1>(&sh, lds_raw, G, q);
                    else rsos_loader<NY, NL, 2, false, 0, 1>(&sh, lds_raw, G, q);
                } else {
                    if (g.rpitch == 770) rsos_loader<NY, NL, 4, false, 770 * 8, 1>(&sh, lds_raw, G, q);
                    else if (g.rpitch == 642) rsos_loader<NY, NL, 4, false, 642 * 8, 1>(&sh, lds_raw, G, q);
                    else rsos_loader<NY, NL, 4, false, 0, 1>(&sh, lds_raw, G, q);
                }
                continue;
            }
            switch (ru) {
            case 8:
                if (g.src32 && g.rpitch == 770) rsos_loader<NY, NL, 8, true, 770 * 8>(&sh, lds_raw, G, q);
                else if (g.src32 && g.rpitch == 642) rsos_loader<NY, NL, 8, true, 642 * 8>(&sh, lds_raw, G, q);
                else if (g.src32) rsos_loader<NY, NL, 8, true>(&sh, lds_raw, G, q);
                else if (g.rpitch == 770) rsos_loader<NY, NL, 8, false, 770 * 8>(&sh, lds_raw, G, q);  // (rings of 768 / 640 frames:
                else if (g.rpitch == 642) rsos_loader<NY, NL, 8, false, 642 * 8>(&sh, lds_raw, G, q);  //  row offsets as immediates)
                else rsos_loader<NY, NL, 8, false>(&sh, lds_raw, G, q);
                break;
            case 4:
                if (g.src32) rsos_loader<NY, NL, 4, true>(&sh, lds_raw, G, q);
                else if (g.rpitch == 770) rsos_loader<NY, NL, 4, false, 770 * 8>(&sh, lds_raw, G, q);
                else if (g.rpitch == 642) rsos_loader<NY, NL, 4, false, 642 * 8>(&sh, lds_raw, G, q);
                else rsos_loader<NY, NL, 4, false>(&sh, lds_raw, G, q);
                break;
            case 2:
                if (g.src32) rsos_loader<NY, NL, 2, true>(&sh, lds_raw, G, q);
                else if (g.rpitch == 770) rsos_loader<NY, NL, 2, false, 770 * 8>(&sh, lds_raw, G, q);
                else if (g.rpitch == 642) rsos_loader<NY, NL, 2, false, 642 * 8>(&sh, lds_raw, G, q);
                else rsos_loader<NY, NL, 2, false>(&sh, lds_raw, G, q);
                break;
            default:
                if (g.src32) rsos_loader<NY, NL, 1, true>(&sh, lds_raw, G, q);
                else rsos_loader<NY, NL, 1, false>(&sh, lds_raw, G, q);
                break;
            }
        } else if (HW && wave == 12) {
            if (g.help && !(g.debug & 128)) rsos_helper<NY>(&sh, lds_raw);
        } else if (!(g.debug & 128))
            {
            if (NW >= 16 && wave > 12) continue;
            // (12 waves, and the helper geometry's sixteen: wave w sits on SIMD w % 4 and consecutive blocks go to consecutive
            //  SIMDs -- residues 0 3 7 | 1 4 8 | 2 5 9 on SIMDs 1 | 2 | 3, 6 next to the chain.  Three consecutive residues on one
            //  SIMD, tried for the helper geometry, cost the plain 12-wave form 4 %: the chain takes the blocks in order.)
            const int yi = NW == 16 ? (wave < 4 ? wave - 1 : wave < 8 ? wave - 2 : wave < 12 ? wave - 3 : 9) : wave - (wave > 4 ? 2 : 1);
            if constexpr (sizeof(TO) == 4 && CYC > 0) {
                if (g.f32m) {
                    rsos_ywave<KS, NY, NL, TO, CYC, true, HW>(&sh, lds_raw, G, yi);
                    continue;
                }
            }
            rsos_ywave<KS, NY, NL, TO, CYC, false, HW>(&sh, lds_raw, G, yi);
        }
    }
}

template <int KS, int NW, typename TO, int CYC>
__global__ __launch_bounds__(rsos_nthreads(NW)) void k_rsos(const double* __restrict__ tab, const int* __restrict__ jend_g, RsSos g,
                                                  TO* __restrict__ y, RsGlobalTables gsrc) {
    rsos_body<KS, NW, TO, CYC>(tab, jend_g, g, y, gsrc, (int64_t)blockIdx.x, (int64_t)gridDim.x);
}

// Several filters in ONE launch (config 4: the 64 scenes of an Append, two channels each -- as launches of their own every one
// of them would fill a tenth of the chip): workgroup b walks the groups b % gpm, b % gpm + gpm, ... of member b / gpm.  The
// members' geometries, sources and results come from a table in global memory; everything else is k_rsos.
template <int KS, int NW, typename TO, int CYC>
__global__ __launch_bounds__(rsos_nthreads(NW)) void k_rsos_batch(const RsosItem* __restrict__ items, int gpm) {
    const int it = (int)(blockIdx.x / (unsigned)gpm);
    const RsosItem I = items[it];
    rsos_body<KS, NW, TO, CYC>(I.tab, I.jend, I.g, (TO*)I.y, I.gsrc, (int64_t)(blockIdx.x % (unsigned)gpm), (int64_t)gpm);
}

constexpr size_t kRsosStaticLds = sizeof(RsosShared) + 64;

#if SO_RSOS_ONLY_KS != 0
template <int KS, int NW, typename TO, int CYC>
static void launch_rsos_k(const double* tab, const int* jend, const RsSos& g, void* y, const RsGlobalTables& gsrc, int grid, hipStream_t st) {
    const size_t lds = rsos_lds_bytes(g.ngroups, KS, g.rpitch, NW, CYC);
    static bool seen[64];
    if (first_use_on_device(seen))
        (void)hipFuncSetAttribute((const void*)k_rsos<KS, NW, TO, CYC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rsos_lds_budget());
    hipLaunchKernelGGL((k_rsos<KS, NW, TO, CYC>), dim3((unsigned)grid), dim3(rsos_nthreads(NW)), lds, st, tab, jend, g, (TO*)y, gsrc);
}

template <int KS, int NW, typename TO>
static void launch_rsos_batch_k(const RsosItem* items, int nitems, int gpm, const RsSos& g0, hipStream_t st) {
    const size_t lds = rsos_lds_bytes(g0.ngroups, KS, g0.rpitch, NW, 1);
    static bool seen[64];
    if (first_use_on_device(seen))
        (void)hipFuncSetAttribute((const void*)k_rsos_batch<KS, NW, TO, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rsos_lds_budget());
    hipLaunchKernelGGL((k_rsos_batch<KS, NW, TO, 1>), dim3((unsigned)(nitems * gpm)), dim3(rsos_nthreads(NW)), lds, st, items, gpm);
}

// returns 0 when launched, -1 if no instantiation fits
template <int KS, typename TO>
static int launch_rsos_t(const double* tab, const int* jend, const RsSos& g, void* y, const RsGlobalTables& gsrc, int grid, hipStream_t st) {
    // (register taps: cyc * KS doubles per y wave -- up to 32 of them at 12 waves per workgroup (168 registers each), up
    //  to 80 at 8 waves (256 registers))
    if constexpr (SO_RSOS_ONLY_NW == 0 || SO_RSOS_ONLY_NW == 12) if (g.nwaves == 12) {
        switch (g.cyc) {
        case 0: launch_rsos_k<KS, 12, TO, 0>(tab, jend, g, y, gsrc, grid, st); return 0;
        case 1: launch_rsos_k<KS, 12, TO, 1>(tab, jend, g, y, gsrc, grid, st); return 0;
        case 2:
            if constexpr (KS <= 16) {
                launch_rsos_k<KS, 12, TO, 2>(tab, jend, g, y, gsrc, grid, st);
                return 0;
            }
            return -1;
        default: return -1;
        }
    }
    if constexpr (SO_RSOS_ONLY_NW == 0 || SO_RSOS_ONLY_NW == 16) if (g.nwaves == 16) {
        if (g.cyc != 1) return -1;
        launch_rsos_k<KS, 16, TO, 1>(tab, jend, g, y, gsrc, grid, st);
        return 0;
    }
    if constexpr (SO_RSOS_ONLY_NW == 0 || SO_RSOS_ONLY_NW == 17) if (g.nwaves == 17) {
        if (g.cyc != 1) return -1;
        launch_rsos_k<KS, 17, TO, 1>(tab, jend, g, y, gsrc, grid, st);
        return 0;
    }
    if constexpr (SO_RSOS_ONLY_NW == 0 || SO_RSOS_ONLY_NW == 8) if (g.nwaves == 8) {
        switch (g.cyc) {
        case 0: launch_rsos_k<KS, 8, TO, 0>(tab, jend, g, y, gsrc, grid, st); return 0;
        case 1: launch_rsos_k<KS, 8, TO, 1>(tab, jend, g, y, gsrc, grid, st); return 0;
        case 2: launch_rsos_k<KS, 8, TO, 2>(tab, jend, g, y, gsrc, grid, st); return 0;
        case 3:
            if constexpr (KS <= 20) {
                launch_rsos_k<KS, 8, TO, 3>(tab, jend, g, y, gsrc, grid, st);
                return 0;
            }
            return -1;
        case 5:
            if constexpr (KS <= 16) {
                launch_rsos_k<KS, 8, TO, 5>(tab, jend, g, y, gsrc, grid, st);
                return 0;
            }
            return -1;
        default: return -1;
        }
    }
    return -1;
}
#endif  // SO_RSOS_ONLY_KS != 0

#define SO_RSOS_ARGS const double *tab, const int *jend, const RsSos &g, void *y, const RsGlobalTables &gsrc, int grid, hipStream_t st
#if SO_RSOS_ONLY_KS > 0
// this unit's window length: launch_rsos_ks<N>, called by the dispatcher's unit
#define SO_RSOS_CAT2(a, b) a##b
#define SO_RSOS_CAT(a, b) SO_RSOS_CAT2(a, b)
// (... and result type: -DSO_RSOS_ONLY_F32=0 the Float64 results of this window length, =1 the Float32 ones)
// (... and workgroup size: launch_rsos_ks<N><d|f><12|16|8>)
#if SO_RSOS_ONLY_F32
#define SO_RSOS_TAG f
#define SO_RSOS_TO float
#else
#define SO_RSOS_TAG d
#define SO_RSOS_TO double
#endif
int SO_RSOS_CAT(SO_RSOS_CAT(SO_RSOS_CAT(launch_rsos_ks, SO_RSOS_ONLY_KS), SO_RSOS_TAG), SO_RSOS_ONLY_NW)(SO_RSOS_ARGS) {
    return launch_rsos_t<SO_RSOS_ONLY_KS, SO_RSOS_TO>(tab, jend, g, y, gsrc, grid, st);
}
#if SO_RSOS_ONLY_KS == 4 && (SO_RSOS_ONLY_NW == 12 || SO_RSOS_ONLY_NW == 16)
// (the batched launch exists for the plain filter's identity window: 12 waves, or 16 with the step waves)
int SO_RSOS_CAT(SO_RSOS_CAT(launch_rsos_batch4, SO_RSOS_TAG), SO_RSOS_ONLY_NW)(const RsosItem* items, int nitems, int gpm, const RsSos& g0, hipStream_t st) {
    launch_rsos_batch_k<4, SO_RSOS_ONLY_NW, SO_RSOS_TO>(items, nitems, gpm, g0, st);
    return 0;
}
#endif
#else
// LDS the kernel needs besides its static block (the planner sizes the ring with this); cyc > 0: no tap table
size_t rsos_lds_bytes(int ngroups, int ks, int rpitch, int nwaves, int cyc) {
    const int ny = nwaves >= 16 ? 10 : nwaves - 2, nx = 2 * ny + 1;
    return ((cyc > 0 ? 0 : (size_t)ngroups * ks * 64) + (size_t)16 * rpitch + (size_t)nx * 192 + (size_t)rsos_nss(ny) * 192 + 16 * 16 * 2 +
            (nwaves == 17 ? 3 * kRsosHslots * 64 : 0)) * 8;
}
size_t rsos_lds_budget() { return 160 * 1024 - kRsosStaticLds; }

#if SO_RSOS_ONLY_KS == 0
#define SO_RS(KS_) \
    int launch_rsos_ks##KS_##d12(SO_RSOS_ARGS); int launch_rsos_ks##KS_##d16(SO_RSOS_ARGS); int launch_rsos_ks##KS_##d8(SO_RSOS_ARGS); \
    int launch_rsos_ks##KS_##f12(SO_RSOS_ARGS); int launch_rsos_ks##KS_##f16(SO_RSOS_ARGS); int launch_rsos_ks##KS_##f8(SO_RSOS_ARGS); \
    int launch_rsos_ks##KS_##d17(SO_RSOS_ARGS); int launch_rsos_ks##KS_##f17(SO_RSOS_ARGS);
SO_RS(4) SO_RS(12) SO_RS(13) SO_RS(14) SO_RS(16) SO_RS(20)
#undef SO_RS
#endif
#if SO_RSOS_ONLY_KS == 0
int launch_rsos_batch4d12(const RsosItem*, int, int, const RsSos&, hipStream_t);
int launch_rsos_batch4f12(const RsosItem*, int, int, const RsSos&, hipStream_t);
int launch_rsos_batch4d16(const RsosItem*, int, int, const RsSos&, hipStream_t);
int launch_rsos_batch4f16(const RsosItem*, int, int, const RsSos&, hipStream_t);
#endif
// the members of a batch share window length, waves, ring and result type (g0: any member's geometry); 0 when launched
int launch_rsos_batch(const RsosItem* items, int nitems, int gpm, const RsSos& g0, hipStream_t st) {
    if (nitems <= 0 || gpm <= 0) return 0;
    if (g0.ks != 4 || g0.cyc != 1 || (g0.nwaves != 12 && g0.nwaves != 16)) return -1;
#if SO_RSOS_ONLY_KS == 0
    if (g0.nwaves == 12) return g0.out_f32 ? launch_rsos_batch4f12(items, nitems, gpm, g0, st) : launch_rsos_batch4d12(items, nitems, gpm, g0, st);
    return g0.out_f32 ? launch_rsos_batch4f16(items, nitems, gpm, g0, st) : launch_rsos_batch4d16(items, nitems, gpm, g0, st);
#else
    if (g0.nwaves == 12) {
        if (g0.out_f32) launch_rsos_batch_k<4, 12, float>(items, nitems, gpm, g0, st);
        else launch_rsos_batch_k<4, 12, double>(items, nitems, gpm, g0, st);
    } else {
        if (g0.out_f32) launch_rsos_batch_k<4, 16, float>(items, nitems, gpm, g0, st);
        else launch_rsos_batch_k<4, 16, double>(items, nitems, gpm, g0, st);
    }
    return 0;
#endif
}

int launch_rsos(SO_RSOS_ARGS) {
    if (g.n_out <= 0) return 0;
    if (g.ngroups > kRsosMaxGroups) return -1;
#if SO_RSOS_ONLY_KS == 0
#define SO_RS(KS_) \
    if (g.ks == KS_) {                                                                                                             \
        if (g.nwaves == 12) return g.out_f32 ? launch_rsos_ks##KS_##f12(tab, jend, g, y, gsrc, grid, st) : launch_rsos_ks##KS_##d12(tab, jend, g, y, gsrc, grid, st); \
        if (g.nwaves == 16) return g.out_f32 ? launch_rsos_ks##KS_##f16(tab, jend, g, y, gsrc, grid, st) : launch_rsos_ks##KS_##d16(tab, jend, g, y, gsrc, grid, st); \
        if (g.nwaves == 8) return g.out_f32 ? launch_rsos_ks##KS_##f8(tab, jend, g, y, gsrc, grid, st) : launch_rsos_ks##KS_##d8(tab, jend, g, y, gsrc, grid, st);    \
        if (g.nwaves == 17) return g.out_f32 ? launch_rsos_ks##KS_##f17(tab, jend, g, y, gsrc, grid, st) : launch_rsos_ks##KS_##d17(tab, jend, g, y, gsrc, grid, st); \
        return -1;                                                                                                                 \
    }
#else
#define SO_RS(KS_) \
    if (g.ks == KS_) return g.out_f32 ? launch_rsos_t<KS_, float>(tab, jend, g, y, gsrc, grid, st) : launch_rsos_t<KS_, double>(tab, jend, g, y, gsrc, grid, st);
#endif
    SO_RS(4) SO_RS(12) SO_RS(13) SO_RS(14) SO_RS(16) SO_RS(20)
#undef SO_RS
    return -1;
}
#endif

}  // namespace so
