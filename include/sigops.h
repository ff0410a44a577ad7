/*
 * sigops.h — C-ABI of libsigops, the MI355X (gfx950) sink engine for
 * SignalOperators.jl operator trees.
 *
 * The reference (haberdashPI/SignalOperators.jl v0.5.1) has NO foreign-function
 * boundary on this path: `sink!` is a Julia block-pull loop
 * (src/sink.jl:225-241, inner loop src/sink.jl:256-260).  This header is the
 * boundary a Julia `ccall` (or any other FFI) binds to replace that loop for a
 * device sink: the host keeps the lazy operator tree exactly as the reference
 * builds it (layers L3-L5 of SURVEY.md), flattens it into the node table below
 * and hands it to `so_plan_create` / `so_plan_execute`.
 *
 * Each entry point cites the reference interface it replaces.  Plain pointers
 * and sizes only; no C++/torch types.  All structs are POD with generic slots so
 * that a Julia `struct` / Python `ctypes.Structure` mirror is trivial.
 *
 * Conventions
 *   - frames are 0-based here (Julia is 1-based: julia frame i == frame i-1).
 *   - sample layout is described by (frame_stride, chan_stride) in ELEMENTS;
 *     Julia's `Array{T,2}(undef,nframes,nch)` (src/sink.jl:116) is
 *     frame_stride=1, chan_stride=nframes ("planar, time fastest").
 *   - every function returns SO_OK (0) or a negative so_status_t; the message is
 *     available from so_last_error() (thread-local), mirroring the reference's
 *     `error(msg)` -> ErrorException convention (src/sink.jl:96-97,162).
 */
#ifndef SIGOPS_H
#define SIGOPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The entry points below are the library's whole dynamic symbol table: libsigops.so is built with
 * -fvisibility=hidden, and everything declared between here and the matching pop is exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define SO_ABI_VERSION 1

/* ---- status codes ------------------------------------------------------- */
typedef enum so_status {
    SO_OK = 0,
    SO_ERR_INVALID = -1,     /* malformed node table / argument                          */
    SO_ERR_LENGTH = -2,      /* reference `error(...)` on lengths: unknown/infinite length
                                (src/sink.jl:96-97), buffer longer than signal
                                (src/sink.jl:161-163), After past the end
                                (src/cutting.jl:174-181), Append after infinite
                                (src/appending.jl:61-63)                                 */
    SO_ERR_UNSUPPORTED = -3, /* tree is valid for the reference but not lowerable here
                                (host must fall back to the stock CPU sink!)             */
    SO_ERR_RUNTIME = -4,     /* HIP runtime failure                                      */
    SO_ERR_CHANNELS = -5,    /* channel-count mismatch (src/reformatting.jl:167-168)     */
    SO_ERR_NODEVICE = -6     /* no HIP device available: the engine never computes on the
                                CPU                                                      */
} so_status_t;

/* ---- sample types ------------------------------------------------------- */
typedef enum so_dtype {
    SO_F32 = 0,
    SO_F64 = 1,
    SO_I64 = 2 /* only for CONST literals (NumberSignal{Int}, src/numbers.jl:1-11) */
} so_dtype_t;

/* ---- lengths (src/inflen.jl, src/signal.jl:28-37, src/numbers.jl:5-9) ---- */
#define SO_LEN_INF (-1)       /* inflen                                      */
#define SO_LEN_MISSING (-2)   /* missing                                     */
#define SO_LEN_UNCHECKED (-3) /* host did not compute it; planner infers     */

/* ---- node kinds (one per reference node type on the hot path) ----------- */
typedef enum so_kind {
    SO_NODE_ARRAY = 0,    /* arrays / (array,fs) tuples      src/arrays.jl:35-132          */
    SO_NODE_CONST = 1,    /* NumberSignal                    src/numbers.jl:1-64           */
    SO_NODE_FUNC = 2,     /* SignalFunction                  src/functions.jl:11-60        */
    SO_NODE_UNTIL = 3,    /* CutApply{..,Val{:Until}}        src/cutting.jl:130,199-210    */
    SO_NODE_AFTER = 4,    /* CutApply{..,Val{:After}}        src/cutting.jl:134,160-191    */
    SO_NODE_PAD = 5,      /* PaddedSignal (Pad / Extend)     src/padding.jl:3-19,150-235   */
    SO_NODE_APPEND = 6,   /* AppendSignals                   src/appending.jl:59-110       */
    SO_NODE_RAMP = 7,     /* RampSignal{:on/:off} (a GAIN)   src/ramps.jl:6-119            */
    SO_NODE_MAP = 8,      /* MapSignal / OperateOn           src/mapsignal.jl:8-30,131-272 */
    SO_NODE_FILT_SOS = 9, /* FilteredSignal, IIR DF2T SOS    src/filters.jl:98-262         */
    SO_NODE_RESAMPLE = 10,/* FilteredSignal{..ResamplerFn}   src/reformatting.jl:92-122    */
    SO_NODE_NORMPOWER = 11/* NormedSignal                    src/filters.jl:266-314        */
} so_kind_t;

/* FUNC opcodes: whitelisted `fn` of Signal(fn;ω,ϕ) (src/functions.jl:53-60) */
typedef enum so_fn {
    SO_FN_SIN = 0,     /* specialised through sinpi, src/functions.jl:57-60 */
    SO_FN_COS = 1,
    SO_FN_IDENTITY = 2
} so_fn_t;

/* RAMP shaping functions (src/ramps.jl:4 `sinramp`, tests use `identity`) */
typedef enum so_rampfn {
    SO_RAMP_SINRAMP = 0, /* sinpi(0.5x) */
    SO_RAMP_IDENTITY = 1
} so_rampfn_t;

/* MAP functions (src/mapsignal.jl:308,333,360,389; src/reformatting.jl:148-184) */
typedef enum so_mapfn {
    SO_MAP_ADD = 0,        /* Mix      (+, left fold)                               */
    SO_MAP_MUL = 1,        /* Amplify  (*, left fold)                               */
    SO_MAP_SUB = 2,        /* -  (binary) / unary negate when one child             */
    SO_MAP_DIV = 3,        /* /                                                     */
    SO_MAP_TUPLECAT = 4,   /* AddChannel        (bychannel=false)                   */
    SO_MAP_GETCHAN = 5,    /* SelectChannel(n)  (bychannel=false), i3 = n (1-based) */
    SO_MAP_AS1CHANNEL = 6, /* ToChannels(x,1) = sum over channels                   */
    SO_MAP_ASNCHANNELS = 7,/* ToChannels(x,n) = replicate channel 1, i3 = n         */
    SO_MAP_TOELTYPE = 8,   /* ToEltype(x,T), i3 = so_dtype_t                        */
    SO_MAP_REVERSECH = 9   /* OperateOn(reverse,x,bychannel=false) (runtests.jl:273) */
} so_mapfn_t;

/* PAD kinds (src/padding.jl:150-192) */
typedef enum so_padkind {
    SO_PAD_VALUE = 0,    /* number: convert(T,p) on all channels; d0 = value     */
    SO_PAD_VECTOR = 1,   /* tuple/vector: per-channel values; p0 = double[nch]   */
    SO_PAD_ZERO = 2,     /* `zero` type function                                  */
    SO_PAD_ONE = 3,      /* `one`  type function                                  */
    SO_PAD_LASTFRAME = 4,/* value function `lastframe`                            */
    SO_PAD_CYCLE = 5,    /* indexing function x[(i-1)%end+1,j]  padding.jl:132    */
    SO_PAD_MIRROR = 6    /* indexing function `mirror`          padding.jl:142-148*/
} so_padkind_t;

/* RESAMPLE kernel kinds (DSP.jl FIRFilter constructors) */
typedef enum so_rskind {
    SO_RS_RATIONAL = 0, /* ratio l0//l1 exact (FIRInterpolator/FIRDecimator/FIRRational);
                           Nphi = l0                                                   */
    SO_RS_ARBITRARY = 1, /* ratio d0::Float64 (FIRArbitrary), Nphi = i1 (32)           */
    SO_RS_FIR = 2       /* Filt(x,h) with FIR coefficients h (reference src/filters.jl:96-97
                           RawFilterFn -> DF2TFilter(PolynomialRatio(h,[1]))): ratio 1, no
                           delay compensation, y[n] = sum_k h[k] x[n-k]; p0 = h, i2 = len  */
} so_rskind_t;

/*
 * One node of the flattened operator tree.  Children are indices into the same
 * array and MUST be smaller than the node's own index (post-order).  The same
 * child index may be referenced by several parents (shared sub-trees).
 *
 * Slot usage per kind (unused slots must be 0 / NULL):
 *
 *  ARRAY     p0=data  l0=nframes  i0=is_device(0 host,1 HIP device ptr)
 *            s0=frame_stride s1=chan_stride (elements)   dtype=element type
 *            l1=first resident frame (device arrays; 0 = all of it): p0 is the address frame 0 WOULD
 *            have, frames [l1,l0) are in memory, reading an earlier one is SO_ERR_LENGTH
 *  CONST     d0=value  i0=literal type (so_dtype_t; SO_I64 promotes like Julia Int)
 *  FUNC      i0=so_fn_t  i1=has_omega  d0=omega(Hz)  d1=phi (cycles if has_omega
 *            else seconds, src/functions.jl:92-95)       fs = frame rate (required)
 *  UNTIL     l0 = resolvelen (frames, may be <0: src/cutting.jl:32,130)
 *  AFTER     l0 = resolvelen (frames)
 *  PAD       i0=so_padkind_t  i1=extend(0 Pad / 1 Extend)  d0=value  p0=vector
 *  APPEND    children = signals in order
 *  RAMP      i0=direction(0 :on, 1 :off)  i1=so_rampfn_t  l0=R=resolvelen
 *            (max(1,frames), src/ramps.jl:26); child 0 = the signal being ramped
 *            (gives length/nch/dtype); the node's VALUE is the gain
 *  MAP       i0=so_mapfn_t  i1=bychannel  i2=so_padkind_t of `padding`  d0=pad value
 *            i3=extra (see so_mapfn_t); children = x.signals (un-extended)
 *  FILT_SOS  i0=nsections  p0=double[6*nsec] rows (b0,b1,b2,a0,a1,a2), a0==1
 *            d0=gain  i1=blocksize (reference `blocksize`; results are invariant)
 *  RESAMPLE  i0=so_rskind_t  i1=Nphi  l0=num l1=den (RATIONAL)  d0=rate (ARBITRARY)
 *            p0=double[hlen] = resample_filter(ratio) taps   i2=hlen  i3=blocksize
 *            fs = NEW frame rate; child fs = old frame rate
 *  NORMPOWER child 0
 */
typedef struct so_node {
    int32_t kind;            /* so_kind_t                                              */
    int32_t dtype;           /* so_dtype_t: sampletype(x) of this node                  */
    int32_t nch;             /* nchannels(x)                                            */
    int32_t n_children;
    const int32_t* children; /* [n_children] indices                                    */
    int64_t nframes;         /* nframes(x) as the host computed it (>=0, SO_LEN_INF,
                                SO_LEN_MISSING) or SO_LEN_UNCHECKED.  The planner
                                re-derives every length from the reference's length
                                algebra and fails with SO_ERR_INVALID on disagreement.  */
    double fs;               /* framerate(x) in Hz, NaN = missing                       */
    int32_t i0, i1, i2, i3;
    int64_t l0, l1;
    double d0, d1, d2, d3;
    const void* p0;
    const void* p1;
    int64_t s0, s1;
} so_node_t;

/* Description of the sink buffer (`result` of sink!(result,x), src/sink.jl:158-168) */
typedef struct so_out_desc {
    int32_t dtype;        /* element type of result                         */
    int32_t nch;          /* size(result,2)                                 */
    int64_t nframes;      /* size(result,1): frames to write (<= nframes(x)) */
    int64_t frame_stride; /* elements                                       */
    int64_t chan_stride;  /* elements                                       */
    int32_t is_device;    /* 1: `out` of so_plan_execute is a HIP device ptr */
    int32_t reserved;
} so_out_desc_t;

/* Per-plan statistics filled by the last so_plan_execute. */
typedef struct so_stats {
    int32_t n_stages;            /* materialising stages (pointwise/IIR/resample/reduce) */
    int32_t n_launches;          /* kernel launches of one execute                       */
    int64_t algorithmic_bytes;   /* leaf bytes read + result bytes written               */
    int64_t scratch_bytes;       /* device scratch owned by the plan                     */
    int64_t h2d_bytes;           /* host leaves copied to the device per execute         */
    int64_t d2h_bytes;
    double last_exec_ms;         /* hipEvent time of the last execute (kernels only)     */
    double dominant_kernel_ms;   /* hipEvent time of the dominant kernel of the last run */
    int64_t dominant_kernel_bytes; /* algorithmic bytes attributed to that kernel        */
    char dominant_kernel[64];    /* its name                                             */
} so_stats_t;

typedef struct so_plan so_plan_t; /* opaque */

/* ABI version check (no reference counterpart). */
int32_t so_abi_version(void);

/* Thread-local message of the last failure on this thread
 * (replaces the text of the reference's ErrorException). */
const char* so_last_error(void);

/* Number of visible HIP devices (0 => every compute entry point returns
 * SO_ERR_NODEVICE; there is no CPU fallback). */
int32_t so_device_count(void);

/*
 * Plan creation = everything `sink!` does before its first block:
 * length validation (process_sink_params src/sink.jl:94-99, sink! check
 * src/sink.jl:161-163), ToChannels to the buffer's channel count is the HOST's job
 * (src/sink.jl:164) and only verified here, FilterBlock construction
 * (src/filters.jl:204-211).  Walks the node table once, infers lengths/types, splits
 * it into materialising stages and compiles every pointwise chain into a fused
 * program.  Host array leaves are copied to the device at execute time; device
 * leaves are used in place.
 */
int32_t so_plan_create(const so_node_t* nodes, int32_t n_nodes, int32_t root,
                       const so_out_desc_t* out, int32_t device, so_plan_t** plan);

/* nframes(x) of the root as inferred by the planner (SO_LEN_INF / SO_LEN_MISSING
 * possible only if plan creation failed). */
int64_t so_plan_nframes(const so_plan_t* plan);

/*
 * Execute = the block loop `sink!(result,x,::IsSignal)` (src/sink.jl:225-241):
 * writes out_desc.nframes frames into `out`.  `hip_stream` is a hipStream_t or NULL
 * (default stream).  Asynchronous w.r.t. the host when `out` and all leaves are
 * device pointers; otherwise returns after the D2H copy.
 */
int32_t so_plan_execute(so_plan_t* plan, void* out, void* hip_stream);

/*
 * Waits for what the plan has launched so far (`hip_stream`: the stream of its last execute) and returns what only
 * shows once the kernels have run: SO_ERR_RUNTIME with "k_rsos: a wait between its waves did not end" if a kernel gave
 * up on a wait between its waves (its result is invalid).  The call to make behind an execute into a DEVICE result
 * before trusting it -- the reference's `sink!` (src/sink.jl:225-241) returns when the result is complete, and so does
 * execute + check; an execute into a host result synchronises and reports by itself.  so_plan_destroy without it
 * prints the failure to stderr (it has nobody to return it to).
 */
int32_t so_plan_check(so_plan_t* plan, void* hip_stream);

/* Rebind the data pointer of ARRAY node `node_index` (same shape/strides/residency)
 * so a plan can be reused for a new input without re-planning. */
int32_t so_plan_set_array(so_plan_t* plan, int32_t node_index, const void* data);

int32_t so_plan_stats(const so_plan_t* plan, so_stats_t* stats);

/* Per-step view of the same statistics: step `index` of the plan's launch sequence (a fused
 * pointwise launch, or one stage = the launches of one IIR / resampler / Normpower node).
 * Returns the number of steps; fills *info when 0 <= index < that number.  ms is valid after an
 * execute with profiling enabled. */
typedef struct so_step_info {
    char name[64];
    int64_t algorithmic_bytes;  /* bytes the step must read + write (its own inputs and outputs) */
    double ms;                  /* hipEvent time of the step in the last profiled execute         */
    int32_t launches;
    int32_t pad;
} so_step_info_t;
int32_t so_plan_step_info(const so_plan_t* plan, int32_t index, so_step_info_t* info);

/* How the executes of this plan were issued so far (no reference counterpart; lets tests and
 * benchmarks tell a captured HIP-graph replay from direct launches).  -1 for an unknown counter. */
typedef enum so_counter {
    SO_COUNTER_GRAPH_REPLAYS = 0,   /* executes replayed from the captured launch graph            */
    SO_COUNTER_GRAPH_CAPTURES = 1,  /* times the launch sequence was captured                      */
    SO_COUNTER_DIRECT_EXECUTES = 2, /* executes issued launch by launch                            */
    SO_COUNTER_FUSED_MFMAS_PER_BLOCK = 3 /* fp64 MFMAs the fused resampler + IIR kernel issues per block of 16 outputs x 16
                                          * rows (0: the plan has no such launch): what a matrix-pipe roofline is priced with */
} so_counter_t;
int64_t so_plan_counter(const so_plan_t* plan, int32_t which);

/* Diagnostic: compile `body` (HIP source of a pointwise step as the planner generates it: per-piece device
 * functions + `extern "C" __global__ void k_rtc(...)`) with hipRTC for gfx950 against the library's own
 * embedded definitions.  Needs no device.  SO_OK, or SO_ERR_UNSUPPORTED with the compiler's log (also
 * copied to `log`).  Pointwise steps the ahead-of-time interpreter kernel could only run after
 * materialising sub-expressions are specialised this way at plan time (SIGOPS_RTC=0 disables, =1 forces
 * it for every pointwise step); reference shape: one loop per map nest, src/mapsignal.jl:249-272. */
int32_t so_rtc_compile_check(const char* body, char* log, int32_t log_capacity);

/* Large pointwise steps the interpreter CAN run are specialised too, but never at the caller's expense: the plan
 * uses the specialised kernel if this process has it -- or, where the host sets SIGOPS_CACHE_DIR, that directory of
 * code objects (the library writes nowhere on its own) --, otherwise it
 * keeps the interpreter -- same values, operation for operation -- while a background thread compiles the kernel
 * for the plans to come.  so_rtc_wait_idle returns when every compile queued so far has finished (a service
 * warming up; tests).  SIGOPS_RTC_NOASYNC=1 switches the background path off. */
int32_t so_rtc_wait_idle(void);
/* Ends the background compiles in an orderly way: queued ones are dropped, the one in flight is waited for, the thread is
 * joined (a later plan starts a new one).  For a host about to tear the GPU runtime down -- interpreter shutdown hooks:
 * `atexit` in the Python mirror and in the Julia glue --; the library also runs it from the C atexit chain. */
int32_t so_rtc_shutdown(void);

/* When enabled, so_plan_execute brackets every kernel with hipEvents (on the stream
 * the kernels are launched on) and fills so_stats_t.*_ms.  Off by default.
 *   enable = 1: every execute synchronises the stream and reads its own events;
 *   enable = 2: deferred -- executes only record (a set of events per execute, the most recent 256
 *               kept) and never synchronise; after the caller has synchronised the stream,
 *               so_plan_step_info reports each step's MEAN over the recorded executes.  Lets a
 *               benchmark time the kernels inside its own timed region. */
int32_t so_plan_set_profiling(so_plan_t* plan, int32_t enable);

void so_plan_destroy(so_plan_t* plan);

/* ---- the exchange step of a sharded sink (no reference counterpart: the
 *      reference is single-process).  One process per GPU; every rank evaluates its share of the
 *      result with so_plan_execute -- the frames of its Append children (reference
 *      src/appending.jl:59-76: children are independent) or its slab of channels -- and the shares are
 *      gathered into the full planar buffer of every rank by grouped RCCL send / recv (xGMI inside a
 *      node).  RCCL is bound at run time; without it these return SO_ERR_UNSUPPORTED. ------------- */
typedef struct so_comm so_comm_t; /* opaque */
typedef struct so_slab {   /* what ONE rank contributes: `rows` runs of `row_elems` elements ...          */
    int64_t rows, row_elems;
    int64_t dst_offset;     /* ... and where they go in every rank's full buffer: first element,       */
    int64_t dst_row_stride; /*     elements between runs (planar result: rows = channels, stride =      */
} so_slab_t;                /*     chan_stride; a time range: dst_offset = its first frame)             */
/* rank 0 fills 128 bytes that the host passes to every rank (MPI, a file, a socket ...) */
int32_t so_comm_unique_id(void* id128);
int32_t so_comm_create(const void* id128, int32_t world, int32_t rank, int32_t device, so_comm_t** comm);
/* mine: this rank's share (device pointer; runs src_row_stride elements apart; may already sit at its place in
 * `full`: then nothing is copied locally); slabs[world]: every rank's share; full: device buffer */
int32_t so_comm_allgather(so_comm_t* comm, const void* mine, int64_t src_row_stride, void* full,
                          const so_slab_t* slabs, int32_t dtype, void* stream);
/* The operands of a root Mix(xs...) = OperateOn(+, xs...) (reference src/mapsignal.jl:307-308) evaluated on
 * different ranks: every rank holds a partial sum in a buffer of the result's shape -- `rows` runs of `row_elems`
 * elements, `row_stride` elements apart -- and one grouped reduction adds the buffers up IN PLACE: into every rank's
 * (root < 0) or into rank `root`'s (the other buffers are then unspecified). */
int32_t so_comm_reduce_sum(so_comm_t* comm, void* buf, int64_t rows, int64_t row_elems, int64_t row_stride,
                           int32_t dtype, int32_t root, void* stream);
const char* so_comm_last_error(void);
void so_comm_destroy(so_comm_t* comm);

/* ---- filter design (replaces the DSP.jl calls the reference makes at sink time:
 *      src/filters.jl:10-11,94 and src/reformatting.jl:93-96).  Host-side fp64; a
 *      Julia host would normally pass DSP.jl's own coefficients instead. ---------- */
typedef enum so_filt_type {
    SO_FILT_LOWPASS = 0,
    SO_FILT_HIGHPASS = 1,
    SO_FILT_BANDPASS = 2,
    SO_FILT_BANDSTOP = 3
} so_filt_type_t;

typedef enum so_filt_method {
    SO_METHOD_BUTTERWORTH = 0, /* Butterworth(order)            */
    SO_METHOD_CHEBYSHEV1 = 1   /* Chebyshev1(order, ripple dB)  */
} so_filt_method_t;

/* digitalfilter(Type(f1[,f2];fs=fs), method) |> SecondOrderSections.
 * sos must hold 6*(2*order) doubles at most; *nsections receives the count. */
int32_t so_design_iir(int32_t type, double f1, double f2, double fs, int32_t method,
                      int32_t order, double ripple_db, double* sos, int32_t sos_capacity,
                      int32_t* nsections, double* gain);

/* digitalfilter(Type(f1[,f2];fs=fs), method) as the ZeroPoleGain object itself: zeros / poles as
 * interleaved (re,im) pairs, `capacity` complex numbers each.  With so_zpk_to_sos this is the raw
 * filter object path `Filt(x, digitalfilter(...))` of the reference (src/filters.jl:89-97;
 * test/runtests.jl:365-368 expects it bit-equal to the named form). */
int32_t so_design_iir_zpk(int32_t type, double f1, double f2, double fs, int32_t method, int32_t order,
                          double ripple_db, double* z, int32_t* nz, double* p, int32_t* np, int32_t capacity,
                          double* k);
/* DF2TFilter(::ZeroPoleGain) -> SecondOrderSections + gain (resolve_filter, src/filters.jl:94). */
int32_t so_zpk_to_sos(const double* z, int32_t nz, const double* p, int32_t np, double k, double* sos,
                      int32_t sos_capacity, int32_t* nsections, double* gain);

/* DF2TFilter(::PolynomialRatio) for the engine: `filt(b, a, x[, si])` and `Filt(x, PolynomialRatio(b, a))` of any order
 * (reference src/filters.jl:68-95 hand DSP.jl's direct-form recurrence the coefficients; the device runs cascades of
 * second-order sections).  Both polynomials are factored in Float64 (Aberth-Ehrlich; multiple roots by clustering), conjugates made
 * exact and paired as so_zpk_to_sos pairs them; leading zeros of `b` become delay sections, surplus zeros FIR sections.
 * `*residual` (may be NULL) = relative l2 distance between the impulse responses of the direct form and of the cascade
 * over the filter's memory: the caller's gate for ill-conditioned polynomials.  sos must hold 6 * (max(nb, na) + 1) / 2
 * doubles at least. */
int32_t so_tf_to_sos(const double* b, int32_t nb, const double* a, int32_t na, double* sos, int32_t sos_capacity,
                     int32_t* nsections, double* gain, double* residual);
/* The zero-input response of the direct form from initial state `si` (max(nb, na) - 1 entries, DSP.jl's layout): what
 * `filt(b, a, x, si)` adds to the response from rest.  Writes up to `capacity` frames; `*nframes` = the frames after
 * which the state is below 2^-80 of the response's peak (capacity where it has not decayed). */
int32_t so_tf_zero_input(const double* b, int32_t nb, const double* a, int32_t na, const double* si, int32_t nsi,
                         double* out, int64_t capacity, int64_t* nframes);

/* resample_filter(ratio): rational (num/den, Nphi=num) or arbitrary (rate, nphi).
 * Call with h==NULL to get the length in *hlen. */
int32_t so_design_resample_rational(int64_t num, int64_t den, double* h, int32_t capacity,
                                    int32_t* hlen);
int32_t so_design_resample_arbitrary(double rate, int32_t nphi, double* h, int32_t capacity,
                                     int32_t* hlen);

/* Diagnostics, host only (no device needed): the position (newest input j, 0-based; phase p,
 * 0-based; alpha) the arbitrary-rate resampler uses for each of the outputs 0..n_out-1.
 * Replaces nothing in the reference: it makes visible how the engine follows DSP.jl's
 * FIRArbitrary phase accumulator (`update`: ϕAccumulator += Δ per output; reference call site
 * src/reformatting.jl:92-98, src/filters.jl:252-255) -- the closed form of the kernels, the
 * period positions whose tap tables are built from the accumulator's wrap-around ties
 * (*nbaked of them) and the sparse fix-up list (*nfix entries) combined.  h = resample_filter
 * taps (so_design_resample_arbitrary).  SIGOPS_RS_EXACT=1 in the environment disables the
 * accumulator emulation (closed-form positions only, a measurement aid). */
int32_t so_resample_positions(double fs_in, double fs_out, double rate, int32_t nphi, const double* h,
                              int32_t hlen, int64_t n_out, int64_t* j, int32_t* p, double* alpha,
                              int64_t* nfix, int64_t* nbaked);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SIGOPS_H */
