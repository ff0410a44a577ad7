// Internal structures shared by the planner (host) and the HIP kernels (device).
// Not part of the C-ABI (include/sigops.h).
#pragma once
#include <cstdint>

namespace so {

// ---------------------------------------------------------------------------
// Fused pointwise programs.  A *piece* is a rectangle [a,b) x [c0,c1) of one
// stage's output inside which the value is a straight-line expression (every
// Until/After/Pad/Append/Ramp boundary has been resolved on the host into piece
// boundaries, reference src/cutting.jl, src/padding.jl, src/appending.jl,
// src/ramps.jl state machines).  The expression is compiled to postfix code for
// a 4-deep register stack machine; channel-independent sub-expressions
// (generators, ramps: reference src/functions.jl:53-60, src/ramps.jl:60-72) are
// hoisted into a per-frame program and evaluated once per frame.
enum OpCode : int32_t {
    OP_CONST = 0,   // push leaf.v0
    OP_LOAD = 1,    // push array element
    OP_SCALAR = 2,  // push *(const double*)leaf.base (device scalar, e.g. rms)
    OP_FUNC = 3,    // push SignalFunction value
    OP_RAMP = 4,    // push ramp gain
    OP_ADD = 5,
    OP_SUB = 6,
    OP_MUL = 7,
    OP_DIV = 8,
    OP_NEG = 9,
    OP_ROUND32 = 10,  // round top of stack to Float32 (Julia Float32 arithmetic)
    OP_STOREF = 11,   // pop -> frame slot arg
    OP_LOADF = 12     // push frame slot arg
};

enum LeafMode : int32_t { LM_PLAIN = 0, LM_CYCLE = 1, LM_MIRROR = 2 };

struct DOp {
    int32_t code;
    int32_t arg;
};

struct DLeaf {
    const void* base;
    int64_t fstride, cstride;  // elements
    int64_t df, dc;            // idx_f = sf*n + df ; idx_c = sc*c + dc
    int64_t modn;              // CYCLE / MIRROR length ; RAMP: M (start of :off ramp)
    double v0, v1, v2;         // CONST: value | FUNC: omega, phi, fs | RAMP: R
    int32_t sf, sc;
    int32_t dtype;  // SO_F32 / SO_F64 of `base`
    int32_t mode;   // LeafMode | FUNC: so_fn_t | RAMP: so_rampfn_t
    int32_t flag;   // FUNC: has_omega | RAMP: direction (0 on, 1 off)
    int32_t buf;    // >=0: `base` is patched from plan buffer `buf` after allocation
};

constexpr int kBlock = 256;
constexpr int kPointwiseE = 2;  // frames per thread of k_pointwise (independent loads in flight)
constexpr int kMaxFrameSlots = 4;
constexpr int kStackDepth = 4;

struct DPiece {
    int64_t a, b;  // frames [a,b) in stage-output coordinates
    int32_t c0, c1;
    int32_t frame_pc, frame_len;
    int32_t samp_pc, samp_len;
    int64_t block0;  // first workgroup of this piece
    int64_t nblk_f;  // workgroups along frames
    int32_t chc;     // channels per workgroup
    int32_t depth;   // stack depth the programs need (2 -> small-register interpreter)
    int32_t chain;   // the per-sample program is `LOAD (LOADF s, binop)*` over a planar unit-stride leaf:
                     // k_pointwise's chain path (eight channel loads in flight per lane, no interpreter)
    int32_t sub;     // consecutive blocks of kBlock*E frames per workgroup (>= 1)
};

// A carrier: one planar array read at (frame n, channel c) -> base[c*cstride + n + df],
// followed by up to 4 steps against per-frame values (see k_resample_periodic).
struct DCarrier {
    int64_t a, b;  // frames [a,b) of the stage input this carrier covers
    const void* base;
    int64_t cstride, df;
    int32_t dtype;
    int32_t buf;         // >=0: base patched from plan buffer (like DLeaf)
    int32_t array_node;  // ARRAY node the base comes from, or -1
    int32_t vec_ok;      // base/cstride allow 16-byte vector loads
    int32_t frame_pc, frame_len, depth, nsteps;
    int32_t op[4];   // OpCode
    int32_t arg[4];  // slot | flip<<8 | round32<<9
    // closed form of the frame program (what the resampler's loader evaluates in its hot
    // loop): slot k = leaf slot_leaf[k] evaluated as slot_kind[k] (OP_CONST / OP_SCALAR /
    // OP_FUNC / OP_RAMP, | 0x100: rounded to Float32)
    int32_t nslots;
    int32_t slot_leaf[4];
    int32_t slot_kind[4];
    int32_t pad_;  // 1: GA carrier (k_resample_periodic GA): Float32 array, its single multiply by slot 0
                   //    is applied by the compute waves (nsteps is 0 for the staging code); 2: a generated
                   //    piece of such a source (staged as 1.0f); 3 / 4: the same with an ADD (staged as 0.0f)
    // A SECOND planar array as the operand of one step (arg bit 0x400 instead of a slot): `v (op) base2[c*cstride2 + n + df2]`
    // -- `Mix(x, y)` / `Amplify(x, y)` of two arrays in front of a resampler, which the reference evaluates block by block
    // inside the resampler's pull (src/mapsignal.jl:54-57, src/filters.jl:240-244) and K1 used to materialise.  Patched
    // like `base`.  array_node2 < 0 and buf2 < 0: none.
    const void* base2;
    int64_t cstride2, df2;
    int32_t dtype2, buf2, array_node2, vec_ok2;
};
constexpr int kCarArr2 = 0x400;  // DCarrier::arg bit: the step's operand is the second array's sample
inline bool car_has_arr2(const DCarrier& c) {  // (host code)
    for (int i = 0; i < c.nsteps && i < 4; ++i)
        if (c.arg[i] & kCarArr2) return true;
    return false;
}

// Control block of the periodic resampler's fused source.  Every workgroup copies it into LDS
// at kernel start, so the loader waves read carriers / slot leaves with ds_reads instead of
// dependent global or kernarg loads that queue behind the HBM stream they are trying to keep
// busy.  Plans that do not fit are materialised by K1 and arrive as one 0-step carrier.
constexpr int kCtlCar = 4, kCtlOps = 32, kCtlLeaves = 6;
struct RsCtl {
    int32_t ncar, nops, nleaves, pad;
    DCarrier car[kCtlCar];      // frame_pc indexes `ops` below
    DOp ops[kCtlOps];           // leaf operands index `leaves` below
    DLeaf leaves[kCtlLeaves];
};

// The same tables in global memory (carriers index the plan-wide ops / leaves): used only by
// the out-of-line slow path of the staging, which must not take the address of kernel
// arguments (that would move them to scratch).
struct RsGlobalTables {
    const RsCtl* ctl;  // copied to LDS at kernel start
    const DCarrier* car;
    const DOp* ops;
    const DLeaf* leaves;
};

struct OutView {
    void* base;
    int64_t fstride, cstride;  // elements
    int32_t dtype;
    int32_t pad;
};

// ---------------------------------------------------------------------------
// Second-order-sections IIR (DSP.jl DF2T `filt!`; call site
// reference src/filters.jl:252-255).  Up to kMaxSec sections per launch.
constexpr int kMaxSec = 8;
// Where a Normpower's sum-of-squares launch leaves the rms besides its own buffer: the `v0` of the scalar leaves that read it
// (DLeaf::flag = 1 marks such a leaf: the pointwise programs then take the value from the leaf they fetch anyway -- one
// dependent scalar load less per program run than through the leaf's pointer: the dividing pass 0.50 -> 0.34 ms for 12.5 M x 8)
struct RmsPatch {
    double* dst[8];
    int32_t n;
    int32_t pad;
};

struct SosCoefs {
    double b0[kMaxSec], b1[kMaxSec], b2[kMaxSec], a1[kMaxSec], a2[kMaxSec];
    double gain;
    int32_t nsec;
    int32_t pad;
};

struct SosGeom {
    int64_t n;       // frames to produce
    int64_t chunk;   // L: frames per chunk
    int64_t warm;    // W: pass-1 only filters the last min(L,W) frames of a chunk
    int32_t nchunks;
    int32_t nch;
    int32_t kterms;  // K: terms of the truncated power sum in pass 2
    int32_t in_dtype, out_dtype;
    int32_t exact;   // 1: one chunk, DSP.jl's order of operations without fused multiply-adds (ill-conditioned cascades)
    int64_t in_pitch, out_pitch;  // elements between channels
    int64_t store_lo;             // pass 3 stores frames >= store_lo only (warm-up frames of a windowed result)
    // pass 3's tile steps start on the RESULT's cache lines (a row's first step begins up to 15 frames before its
    // chunk; those columns are skipped): whole-line stores for results whose channel rows sit at any multiple of
    // 8 bytes, as the columns of an n x c Array do.  Chunk borders, and so every rounding, are unaffected.  0: off.
    int32_t align_rows;
    int32_t pad_;
    // bad[channel]: the first chunk whose output pass ended in a non-finite state (set to a large value before the
    // passes).  The reference's recurrence stays NaN from the first non-finite sample of a channel to its end; the
    // scan only looks back K chunks, so k_sos_poison writes NaN over everything behind that chunk.  nullptr: off.
    int32_t* bad;
    // fused sine source (k_sos_tiled): the input is x[n] (+|*) sinpi(2((n + src_df + 1)/fs * omega + phi)) --
    // `Mix` / `Amplify` of an array with `Signal(sin)` formed in the filter's own loads
    int32_t src_op;               // 0 none, 1 add, 2 multiply
    int32_t src_has_omega;
    int64_t src_df;
    double src_omega, src_phi, src_fs;
    double src_cd, src_sd;        // cos / sin of the phase step per frame (2 pi omega / fs)
};

// One filter of a batched launch (k_sos_tiled_batch / k_sos_scan_batch): what launch_sos would have been
// given for it, plus the index of its first workgroup in each of the three batched grids.
struct SosDesc {
    const void* x;
    void* y;
    double* v;
    double* s0;
    const double* mpow;
    int64_t first[3];  // pass 1, scan, pass 3
    SosGeom g;
    SosCoefs cf;
};

// Single-pass variant (k_sos_onepass): one read and one write of the signal.  A WAVE owns a tile of
// 64 * kSosLc frames of one channel; lane k owns sub-chunk k (kSosLc frames) in registers.
//   tabs = [nlev][D*D] powers M^(2^s) of M = A^lc (scan over the lanes) followed by
//          [kt][D*D]   powers (A^tf)^j, j = 0..kt-1 (look-back over the kt previous tiles)
//   sync = [0] ticket counter (zeroed before every launch)
//   vpub = [ntiles][nch][D] zero-state end states of the tiles (set to all-ones bytes = "not
//          published yet" before every launch)
constexpr int kSosLc = 32;  // frames per lane
struct SosOne {
    int64_t n;       // frames to produce
    int64_t in_pitch, out_pitch;
    int32_t nch;
    int32_t ntiles;  // tiles along time = ceil(n / (64*kSosLc))
    int32_t nlev;    // 6 = log2(64 lanes)
    int32_t kt;      // look-back terms (<= 64)
    int32_t vec_in, vec_out;  // 16-byte vector loads / stores are legal
    int32_t bt;               // tiles (channels of one time tile) per wave, 1..4
    int32_t debug;            // ablation bits (SIGOPS_SOS_DEBUG): 1 skip pass 1, 2 skip scan, 4 skip look-back, 8 skip M^k sigma, 16 skip pass 3
};

// ---------------------------------------------------------------------------
// Polyphase FIR resampler (DSP.jl FIRRational/FIRInterpolator/FIRDecimator/
// FIRArbitrary kernels; call site reference
// src/filters.jl:252-255 with ResamplerFn, src/reformatting.jl:92-98).
struct RsGeom {
    int64_t n_in, n_out;
    int64_t m0;          // first output index produced (outputs m0 .. m0+n_out)
    int64_t j0;          // first input frame staged: rs_pos returns j relative to it (warm start of a rate without a period)
    int64_t L, M;        // rational
    double delta, c0;    // arbitrary: q_m = c0 + m*delta  (two roundings)
    int64_t c0i;         // rational: q_m = c0i + m*M
    int32_t arbitrary;
    int32_t exact;       // arbitrary kernel, integer frame rates: q_m = c0i + m*nphi*M/L exactly
    int32_t nphi, taps, nch;
    int32_t in_dtype, out_dtype;
    int64_t in_pitch, out_pitch;
};

// Tiled variant for rates without a usable period (irrational / non-integer frame rates, huge L):
// a workgroup stages ct channels x tile_in input frames and both polyphase tables in LDS and
// evaluates tile_out consecutive outputs with per-output closed-form positions (k_resample_tiled).
struct RsTiled {
    RsGeom g;
    int32_t ct;        // channels per workgroup tile
    int32_t tile_out;  // outputs per tile
    int32_t tile_in;   // input frames per channel the LDS tile can hold
    int32_t pitch;     // LDS elements between channel rows
    int64_t ntiles;    // tiles along time
    int32_t pair;      // 1: k_resample_tiled2 (two outputs per lane)
    int32_t pad_;
};

// Persistent form of the same (k_resample_arb.hip): a workgroup walks one contiguous range of batches of 128 outputs for
// ct channels, a loader wave stages the input through an LDS ring, nc compute waves evaluate.
struct RsArb {
    RsGeom g;
    int32_t ct;       // channels per workgroup
    int32_t nc;       // compute waves
    int32_t ringf;    // ring frames per channel row (a power of two)
    int32_t zrows;    // rows of zeros on either side of the tap tables
    int32_t nranges;  // ranges along time (grid = nranges * nch / ct)
    int32_t dma_ok, vec_out, depth, debug;  // depth: loader chunks in flight; debug: ablation bits (SIGOPS_ARB_DEBUG)
    int32_t no, pad_;  // outputs per lane (2 or 4); a batch is 64 * no outputs
    int64_t bpr;       // batches per range
    int64_t nbatches;  // ceil(n_out / (64 no))
    uint32_t* err;     // the plan's host-mapped error word (see RsSos::err), or null: a wait that does not end traps
};

// DSP.jl's FIRArbitrary positions its outputs with a floating-point phase accumulator
// (ϕAccumulator += Δ once per output; reference call site
// src/reformatting.jl:92-98 + src/filters.jl:248-255).  The kernels position outputs in closed
// form; the planner replays the accumulator on the host (it is data independent) and lists the
// outputs whose (newest input, phase) it resolves differently in a way that matters -- the
// wrap-around ties, where the accumulator sits a rounding error below Nϕ+1 and the first tap
// h[0] is dropped.  k_resample_fix recomputes exactly those outputs with the accumulator's
// (j, p, alpha) after the main kernel.
struct RsFix {
    int64_t m;      // output frame
    int64_t j;      // newest input (0-based)
    int32_t p;      // phase (0-based)
    int32_t pad;
    double alpha;
};
struct RsFixArgs {
    const RsFix* fix;
    int64_t nfix;
    const double* pfb;   // [nphi][taps]
    const double* dpfb;
    int64_t n_in;
    int32_t taps, nch;
    int32_t stage_dtype;  // sample type of the resampler stage (inputs are rounded to it)
    int32_t out_dtype;    // element type of y (Float32 result of a Float64 stage: rounded on store)
    // source: carriers (ncar > 0) or a plain planar array
    const DCarrier* car;
    const DOp* ops;
    const DLeaf* leaves;
    int32_t ncar, in_dtype;
    const void* x;
    int64_t in_pitch;
    void* y;
    int64_t out_pitch;
};

// Periodic variant: for a rational rate L/M the (phase, alpha) pattern repeats every
// L outputs / M inputs: the tap pattern of a group of 16 consecutive outputs is the same
// for every (period, channel) row, so a tile of 32 rows x 16 outputs is one small matrix
// product against a per-group tap matrix (see k_resample_periodic).
struct RsPeriodic {
    int64_t n_in, n_out;
    int64_t L, M;       // outputs / inputs per (super-)period
    int64_t nperiods;
    int32_t pt, ct;     // periods x channels per workgroup tile, pt*ct == rows
    int32_t rows;       // 32 (two 16-row MFMA tiles) or 16 (one: long periods, see k_resample_periodic's Q)
    int32_t f32m;       // a Float32 signal all the way (Float32 tile, Float32 result): the products on v_mfma_f32_16x16x4_f32 with
                        // Float32-rounded taps -- half the matrix cycles of the Float64 instruction (k_resample_periodic F32M)
    int32_t ngroups;    // groups of 16 consecutive outputs per period
    int32_t kw;         // inputs in a group's window (multiple of 4)
    int32_t tile_len;   // inputs per channel staged in LDS
    int32_t lds_pitch;  // doubles between channels in LDS
    int32_t jlo;        // input index (relative to the tile's first period base) of LDS slot 0
    int32_t nch;
    int32_t vec_ok;     // 16-byte aligned vector stores are legal
    int32_t nwaves;     // waves per workgroup (groups are dealt to waves in contiguous blocks)
    int32_t ptshift;    // log2(pt)
    int32_t ncompute;   // waves [0,ncompute) compute, the rest stage the next tile
    int32_t grid;       // persistent workgroups (one per CU)
    int32_t pad;        // ablation bits (SIGOPS_RS_DEBUG)
    int32_t nslots;     // LDS ring depth (tiles resident per workgroup)
    int32_t fslots;     // gain ring: per-frame slots held in LDS (0: none), two arrays of fslots*fpitch
    int32_t fpitch;     // doubles per slot row of the gain ring
    int32_t nload;      // loader waves that copy / modify (0: all of them)
    int32_t ftwo;       // gain ring: LDS reserved for the two-level sin evaluation (kRsTwoDoubles more doubles)
    int32_t out_f32;    // fp64 kernel storing into a Float32 result (`sink` of a Float64 signal into Float32)
    int32_t ga;         // GA instantiation: Float32 tiles, the gain multiplied (1) or added (2) at the A operand; lds_pitch in floats
    int32_t arr2;       // A2 instantiation: some carrier's step takes a second array (DCarrier::base2)
    // Fused IIR state pass (the stage's only consumer is an SOS filter): the last nstate (= 2) loader
    // waves multiply every row's staged window [jlo, jlo + 4*ksw) by wtab = (G . Tap), the filter's
    // zero-state end-of-period state as a linear function of the resampler's INPUT, and write
    // vper[ch][period][16] (components 0..D-1 valid) -- what k_sos_tiled<.,.,false> would have
    // computed by reading the resampled signal once more.
    int32_t nstate, ksw;
    const double* wtab;  // [4*ksw][16]
    double* vper;
    int64_t in_pitch, out_pitch;
    long long* trace;   // SIGOPS_RS_TRACE: [16 waves][kRsTraceIters][kRsTraceStamps] cycle stamps of workgroup 0, or null
    // (tile, group)s whose accumulators held a non-finite value: nf[0] counts them, entries of four words from nf[4] on (first
    // period of the tile, low and high word; first channel; group) -- what k_rs_fixup recomputes output by output.  null: off.
    uint32_t* nf;
};
constexpr int kRsTraceIters = 48, kRsTraceStamps = 8;
constexpr uint32_t kRsNfCap = 2047;  // (a launch that lists more keeps its own, wider set for the rest)

// k_rs_fixup (k_exact.hip): the launch behind k_resample_periodic that makes its set of non-finite outputs the REFERENCE's
struct RsPerFixup {
    RsPeriodic g;       // the launch's geometry (nf, rows, pt, ct, L, M, kw, ngroups, out_pitch, n_in, n_out as launched)
    const double* tab;  // [ngroups][kw][16]
    const int* jend;    // [ngroups]
    const int* jrel;    // [L] newest input of output r, relative to its group's window end
    int32_t taps, out_f32;
    RsGlobalTables gsrc;
    void* y;
};
constexpr int kRsTwoBases = 128;                          // share bases per tile (tiles of up to 8192 frames)
constexpr int kRsTwoDoubles = 128 + 2 * 2 * kRsTwoBases;  // per-lane (sin,cos) of the lane's phase offset + two buffers of share bases

// ---------------------------------------------------------------------------
// Fused periodic resampler -> SOS IIR (k_rsos.hip; reference src/filters.jl:143-148 puts the resampler under the
// filter, and src/filters.jl:240-255 filters every block of the resampled child in place: the resampled signal
// never exists as a whole).  One persistent workgroup walks a *sequence group* of 16 rows = rgs time ranges x ct
// channels block by block (16 outputs per row and block):
//   resampled block  X[16 t x 16 rows] = Tap_g^T[16 x 4 ks] . Win[4 ks x 16 rows]          (y waves, MFMA)
//   next state       S'[12 x 16 rows]  = D[12 x 16] . X + A^16[12 x 12] . S                (ONE chain wave, MFMA)
//   result block     Y[16 rows x 16 t] = X^T . T^T + S^T . C^T                              (y waves, MFMA)
// (block state-space form of the DF2T cascade: T lower-triangular Toeplitz of its impulse response, C / D / A^16
// from the same recurrence).  A range starts `wp` periods early from zero state (what earlier frames leave in the
// state has decayed below 2^-70 by then) and stores nothing for those periods.
struct RsSos {
    int64_t n_in, n_out;  // stage input frames (zero outside); result frames
    int64_t L, M;         // outputs / inputs per (super-)period, L % 16 == 0
    int64_t pr;           // periods per range
    int32_t wp;           // warm-up periods in front of every range
    int32_t nranges;
    int32_t ngroups;      // L / 16 blocks per period
    int32_t ks;           // MFMA k-steps of a block's input window (kw / 4)
    int32_t ulo;          // input frame (relative to a period's first input) of ring position 0, before the row's alignment shift
    int32_t ct, rgs;      // channels x ranges per sequence group, ct * rgs == 16
    int32_t nch;
    int32_t nwaves;       // 8, 12 or 16: wave 0 chain, waves 4, 8, 12 loaders, the others y waves
    int32_t ring;         // input ring per row in LDS, frames (power of two, multiple of chunk)
    int32_t rpitch;       // doubles between ring rows (ring + 2: rows on different banks)
    int32_t chunk;        // frames per row and LDS-DMA instruction: 128 or 64
    int32_t depth;        // chunks in flight per loader wave (<= 4)
    int32_t nsec;         // sections of the cascade (<= 6)
    int32_t fuse;         // fast-path step of carrier 0 with frame slot 0: -1 none, 0 v*m, 1 v+m, 2 v-m, 3 m-v
    int32_t fuse_sine;    // ... slot 0 is a sine generator (two-level evaluation); else a constant
    int32_t out_f32;
    int32_t src32;        // carrier 0 is a Float32 array: its chunks land as Float32 in the upper half of their ring slots and the
                          // loader widens them in place (together with the fused step)
    int32_t ring32;       // src32 without a step (a Float32 signal all the way): the ring KEEPS Float32 samples, four bytes a frame
                          // in the first half of each row, and the y waves widen their window operands (no widening pass by the loader)
    int32_t x32;          // a Float32 pipeline: the resampled values are rounded to Float32 before the cascade reads them (the
                          // reference's Float32 resampler output; Float32 result instantiations only)
    int32_t f32m;         // ring32 with or without a step, a Float32 result, taps in registers: the resampling product on the Float32
                          // MFMA (taps rounded once, Float32 accumulators; the cascade stays Float64) and the step on the Float32 samples
    int32_t sring;        // f32m with the fast path's step v + m / v - m: the loader writes m (Float32, once per unit and chunk) into the
                          // free half of the unit's first ring row and the y waves add it to their window operands
    int32_t gsplit;       // 16 waves, groups of 2 / 4 channels with a fused step: the step is applied by waves 13 / 14 (k_rsos.hip, MODE)
    int32_t help;         // 12 waves, taps in registers: three y waves hand their X blocks to the y wave on the chain's SIMD, which
                          // forms D . X for them (k_rsos.hip, kRsosHres*: the SIMDs' MFMA counts evened out)
    int32_t debug;        // ablation bits (SIGOPS_RSOS_DEBUG): 1 no stores, 2 no gain, 4 chain does not wait for x, 8 y waves not for states, 16 nor for input, 32 loader not for ring space
    int32_t cyc;          // > 0: a y wave's blocks cycle through cyc phase groups whose taps it keeps in registers; 0: tap table in LDS
    int32_t arr2;         // the fast path's step takes a SECOND array as its operand (DCarrier::base2): the loader's A2 instantiation
    int32_t pad_;
    int64_t out_pitch;
    int64_t store_lo;     // outputs below this one are not stored (a window's warm-up: the kernel's output 0 is where the
                          // resampler stage's warm start begins, the result where the window does)
    const double* mats;   // [14][64] MFMA operands: D k-steps 0..3, A^16 k-steps 0..2, T^T k-steps 0..3, C^T k-steps 0..2
    int32_t* bad;         // per channel: first range that ended in a non-finite state, or null
    long long* trace;     // SIGOPS_RSOS_TRACE: [16 waves][kRsosTraceIters][8] cycle stamps of workgroup 0, or null
    uint32_t* err;        // host-mapped word of the plan: a wait between the kernel's waves that did not end writes 1 here and the
                          // wave ends (the host reports it with the next call on the plan); null: such a wait traps
};
constexpr int kRsosTraceIters = 96;

// One member of a batched k_rsos launch (k_rsos_batch): what launch_rsos would have been given for it
struct RsosItem {
    RsSos g;
    const double* tab;
    const int* jend;
    void* y;
    RsGlobalTables gsrc;
};

// k_rsos_fixup (k_exact.hip): the launch behind k_rsos that makes a channel's non-finite outputs the REFERENCE's set
struct RsFixup {
    RsSos g;            // the fused launch's geometry (bad, store_lo, out_pitch, out_f32, x32 as launched)
    const double* tab;  // [ngroups][4 ks][16] taps of output rr of group gi at window slot kk
    const int* jend;    // [ngroups] window end (newest input of the group's last output), relative to the period's first input
    const int* jrel;    // [L] newest input of output r of the period, relative to its group's window end (<= 0)
    int32_t taps;       // taps per output as the reference multiplies them
    int32_t pad;
    SosCoefs cf;        // the cascade's sections (the per-sample DF2T form)
    RsGlobalTables gsrc;  // the kernel's source, for single frames
    void* y;            // result, in the kernel's output coordinates
};

// Row-tiled variant for rational rates whose period does not fit the MFMA kernel's LDS ring or
// tap registers (strong downsampling: many inputs per period, long filters), see k_resample_rows.
struct RsRows {
    int64_t n_in, n_out;
    int64_t L, M;        // outputs / inputs per (super-)period
    int64_t nperiods;
    int32_t taps;        // taps per output (columns of ctab)
    int32_t ct, pb;      // channels x periods per workgroup tile (rows = ct*pb, a divisor of 64)
    int32_t jlo;         // input index (relative to the tile's first period base) of LDS element 0
    int32_t tile_len;    // inputs per channel row staged in LDS
    int32_t pitch;       // LDS elements between channel rows
    int32_t nch;
    int32_t debug;       // ablation bits (SIGOPS_RS_DEBUG): 1 skip compute, 2 skip staging
    // MFMA path (rows = 16 or 32): groups of 16 phases against a [kw x 16] tap block streamed
    // from L2; kw == 0 selects the scalar path
    int32_t kw, ngroups, pbshift;
    int32_t threads;     // workgroup size
    int64_t in_pitch, out_pitch;
};

}  // namespace so
