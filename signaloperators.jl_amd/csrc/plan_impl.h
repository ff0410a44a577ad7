// Shared declarations of the planner / stage setup / executor translation units (planner.cpp,
// stages.cpp, accumulator.cpp, executor.cpp): the plan, its nodes, expressions, pieces, buffers, stages.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "kernels.h"
#include "plan.h"
#include "sigops_internal.h"

namespace so {


struct PlanError {
    int status;
    std::string msg;
};
[[noreturn]] inline void fail(int status, const std::string& msg) { throw PlanError{status, msg}; }

#define HIPCHECK(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            fail(SO_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

// ---- lengths (reference src/inflen.jl, src/signal.jl:28-37, src/numbers.jl:5-9) ----
enum { LK_FIN, LK_INF, LK_EXT, LK_NUMEXT };
struct Len {
    int k;
    int64_t n;
};
constexpr int64_t BIG = (int64_t)1 << 62;
inline bool isinf_(Len l) { return l.k != LK_FIN; }
inline int64_t clean(Len l) { return l.k == LK_FIN ? l.n : BIG; }

inline int promote(int a, int b) {
    if (a == SO_F64 || b == SO_F64) return SO_F64;
    if (a == SO_F32 || b == SO_F32) return SO_F32;
    return SO_I64;
}
inline int float_of(int t) { return t == SO_I64 ? SO_F64 : t; }
inline double roundto(int t, double v) { return t == SO_F32 ? (double)(float)v : v; }
inline size_t dsize(int t) { return t == SO_F32 ? 4 : 8; }

struct Node {
    so_node_t nd;
    std::vector<int> kids;
    Len len;
    int dtype, nch;
    double fs;
    int64_t short_skip = 0;  // > 0: evaluating this node skips more frames than a child below has
    int64_t checked_upto = 0;  // frames [0, checked_upto) have been lowered for their errors (check_frames)
};

// ---- expressions -------------------------------------------------------------
enum { E_CONST, E_LOAD, E_SCALAR, E_FUNC, E_RAMP, E_ADD, E_SUB, E_MUL, E_DIV, E_NEG, E_ROUND32, E_RETYPE };
struct Expr {
    int op;
    int dtype;
    int a = -1, b = -1;
    DLeaf leaf{};
    int array_node = -1;  // E_LOAD of an ARRAY node (for so_plan_set_array)
    bool mono = true, heavy = false;
};
struct Map {
    int sf;
    int64_t df;
    int sc;
    int64_t dc;
};
struct Rect {
    int64_t a, b;
    int c0, c1;
};
struct Piece {
    Rect r;
    int e;
};

struct Buf {
    int64_t frames = 0, pitch = 0;
    int nch = 0, dtype = SO_F64;
    size_t bytes = 0;
    void* d = nullptr;
    bool external = false;  // aliases a user/device leaf pointer
    int64_t frame0 = 0;     // stage buffers: node frame stored at position 0 (Stage::base)
};

enum { ST_SOS, ST_RESAMPLE, ST_NORM };
struct Stage {
    int kind, node;
    int64_t need = 0;  // output frames [0,need)
    int64_t lo = (int64_t)1 << 62;  // first frame anybody reads
    int64_t base = 0;  // first frame the stage computes (warm start, see process_stage): its buffer holds [base, need)
    int64_t in_base = 0;  // first frame of the child the stage consumes
    bool processed = false;
    bool norm_alias = false;  // Normpower whose `vals` is its child stage's output buffer (no copy)
    bool norm_direct = false; // Normpower of a plain array leaf: the rms is taken over the array where it lies, readers divide its loads
    bool under_norm = false;  // a Normpower consumes this stage (directly or through further stages)
    int64_t norm_df = 0;      // ... the largest frame offset it is read at on such a path (0: every Normpower's region starts at this stage's first frame)
    int out_buf = -1, in_buf = -1, aux_buf = -1;
    int64_t win_off = -1;  // >= 0: the stage writes the RESULT's frames [win_off, win_off + need) itself (window aliasing)
    // input source (after processing): either a materialised buffer or a direct view
    const void* in_ptr = nullptr;  // direct device pointer (nullptr -> in_buf)
    int in_array_node = -1;
    int64_t in_offset = 0;  // elements (direct)
    int64_t in_pitch = 0, in_frames = 0;
    int pw_step = -1;  // pointwise step materialising the input
    std::vector<DCarrier> carriers;  // periodic resampler: input expressed as carriers
    int car_buf = -1;
    int ctl_buf = -1;  // device copy of the RsCtl control block
    // SOS
    std::vector<SosCoefs> groups;
    SosGeom sg{};
    int mpow_buf = -1, v_buf = -1, s0_buf = -1;
    int src_op = 0;      // fused sine source of the first group's input (SosGeom::src_op): 1 add, 2 multiply
    DLeaf src_fn{};      // ... the generator (E_FUNC leaf: v0 omega, v1 phi, v2 fs, flag has_omega, df)
    bool xscan = false;  // pass 2 is the exact block scan (launch_sos_xscan): no 2^-70 cut anywhere
    int batch = -1;      // member of Plan::batches[batch]: its three passes run inside that batch's launches
    int xs_mats_buf = -1, sblk_buf = -1;
    int bad_buf = -1;    // first non-finite chunk per channel (SosGeom::bad)
    std::vector<std::vector<double>> xs_mats_host;  // per group: [M][M^kXsBlock]
    std::vector<std::vector<double>> mpow_host;  // per group
    // single-pass kernel (k_sos_onepass)
    bool onepass = false;
    SosOne so1{};
    int one_tabs_buf = -1, one_sync_buf = -1, one_vpub_buf = -1;
    std::vector<double> one_tabs_host;    // per group: [nlev + kt][D*D]
    std::vector<size_t> one_tabs_off;     // doubles
    // resample
    RsGeom rg{};
    int pfb_buf = -1, dpfb_buf = -1;
    std::vector<double> pfb_host, dpfb_host;
    bool periodic = false;
    RsPeriodic rp{};
    bool tiled = false;  // tiled resampler without a period (k_resample_tiled)
    RsTiled rt{};
    std::vector<double> rsos_tab_host;  // fused kernel: tap operands of a super-period of its own (small rational ratios)
    std::vector<int> rsos_jend_host;
    int rsos_tab_buf = -1, rsos_jend_buf = -1;
    // ... for the exact recomputation behind a non-finite sample (k_rsos_fixup): newest input of every output of the period
    // relative to its group's window end, and the taps per output the REFERENCE multiplies (its own zero padding included)
    int rsb = -1;  // member of Plan::rsbatches[rsb]: launched with the others (k_rsos_batch), not on its own
    std::vector<DCarrier> alt_carriers;  // a resampler whose input K1 materialises as `x (op) y` of two arrays: that map as ONE two-array
                                         // carrier -- what the fused resampler + IIR kernel reads instead, if it takes the stage (fuse_resample_sos)
    std::vector<int> rs_jrel_host;  // the periodic resampler's own (k_rs_fixup): as rsos_jrel_host, from per_j and jend_host
    int rs_jrel_buf = -1, rs_nf_buf = -1;
    std::vector<int> rsos_jrel_host;
    int rsos_jrel_buf = -1, rsos_taps = 0;
    bool arbk = false;   // ... its persistent form (k_resample_arb), geometry in ra
    RsArb ra{};
    int pfbt_buf = -1, dpfbt_buf = -1;
    std::vector<double> pfbt_host, dpfbt_host;
    bool rows = false;  // row-tiled resampler (k_resample_rows)
    RsRows rr{};
    int mtab_buf = -1, mjend_buf = -1;
    std::vector<double> mtab_host;
    std::vector<int> mjend_host;
    int tab_buf = -1, jend_buf = -1;
    std::vector<double> tab_host;
    std::vector<int> jend_host;
    // periodic variant: per-period positions (kept for the fused IIR state pass)
    std::vector<int64_t> per_j;
    std::vector<int> per_p;
    std::vector<double> per_a;
    int jend_last = 0;
    // fused IIR state pass (this resampler computes its SOS consumer's chunk states)
    int wtab_buf = -1, vper_buf = -1;
    std::vector<double> wtab_host;
    // ... and on the SOS side: the resampler stage that provides vper, Q = A^Ls
    int pre_stage = -1;
    int qmat_buf = -1;
    std::vector<double> qmat_host;
    // fused resampler -> IIR (k_rsos): on the IIR stage, the periodic resampler stage it has absorbed; on that
    // resampler stage, `fused_away` (no launch, no output buffer: its tables and carriers serve the fused kernel)
    int rsos_src = -1;
    bool fused_away = false;
    RsSos rs{};
    int rsos_grid = 0;
    int rsos_mats_buf = -1;
    std::vector<double> rsos_mats_host;
    // outputs DSP.jl's phase accumulator positions differently (recomputed by k_resample_fix)
    std::vector<RsFix> fix_host;
    int fix_buf = -1;
    // norm
    int partial_buf = -1, rms_buf = -1;
    std::vector<int> rms_leaves;  // scalar leaves that read rms_buf: the sum-of-squares launch writes the value into their v0 too (RmsPatch)
    int nparts = 0;
};

struct PwStep {
    int piece0 = 0, npieces = 0;
    int64_t nblocks = 0;
    int out_buf = -1;  // -1: final output
    int64_t bytes = 0;
    bool deep = false;  // some piece needs the 4-deep interpreter
    bool chain = false;  // some piece takes k_pointwise's chain path
    bool il = false;     // ... with an interleaved leaf (the LDS-transposing instantiation)
    std::vector<int> pre;  // pointwise steps that materialise sub-expressions this one reads (run first)
    const void* rtc = nullptr;  // hipRTC-specialised kernel of this step (rtc.cpp), or null: the interpreter (k_pointwise)
    std::vector<int> rtc_leaves;  // ... and the leaves its source refers to (dependencies, plan_lanes)
};

struct Step {
    int kind;  // 0 pointwise, 1 stage kernel, 2 batch of IIR stages (idx into Plan::batches), 3 batch of one-pass IIR stages (Plan::rsbatches)
    int idx;
    std::string name;
    int64_t bytes = 0;
    double ms = 0;
    int launches = 0;
};

// Independent IIR stages of one shape that share their launches (k_sos_tiled_batch, k_sos_scan_batch)
struct SosBatch {
    std::vector<int> members;    // stage ids in step order
    int nsec = 0, dtype = 0;
    int desc_buf = -1;           // device SosDesc[members + 1]
    int bad_buf = -1;            // the members' SosGeom::bad arrays, back to back (one memset per execute)
    std::vector<int> bad_off;    // member m's first channel in it
    std::vector<SosDesc> host;   // the descriptors as last uploaded
    int64_t total[3] = {0, 0, 0};
};

// Several plain filters -- the scenes of an Append -- through the one-pass kernel in ONE launch (k_rsos_batch): every member has
// its own geometry (Stage::rs), source and result; they share window length, waves, ring and result type, and every member
// gets `gpm` sequence groups of the grid.
struct RsBatch {
    std::vector<int> members;  // stage ids in step order
    int gpm = 1;               // workgroups (sequence groups) per member
    int items_buf = -1, fix_buf = -1, bad_buf = -1;
    std::vector<int> bad_off;           // member m's first channel in bad_buf
    std::vector<RsosItem> host;         // the launch table as last uploaded
    std::vector<RsFixup> fhost;         // ... and the fix-up launch's
};

constexpr int kProfExecs = 256;  // executes a deferred-profiling plan keeps events for

struct HostLeaf {
    int node;
    const void* src;
    size_t bytes;
    int buf;
};


struct Plan {
    int device = 0;
    std::vector<Node> nodes;
    int root = -1;
    so_out_desc_t out{};
    std::vector<Expr> exprs;
    std::vector<Buf> bufs;
    std::map<int, int> stage_of_node;
    std::vector<Stage> stages;
    std::vector<PwStep> pw;
    std::vector<Step> steps;
    std::vector<SosBatch> batches;
    std::vector<RsBatch> rsbatches;  // plain filters through k_rsos_batch (fuse_plain_sos)
    std::vector<DPiece> pieces;
    std::vector<DOp> ops;
    std::vector<DLeaf> leaves;
    std::vector<int> leaf_array_node;  // per leaf: ARRAY node or -1
    std::vector<HostLeaf> host_leaves;
    std::map<int, int> array_buf;  // ARRAY node -> buf id (host arrays: device copy)
    std::map<int, const void*> array_ptr;  // current data pointer per ARRAY node
    DPiece* d_pieces = nullptr;
    DOp* d_ops = nullptr;
    DLeaf* d_leaves = nullptr;
    uint32_t* kerr = nullptr;  // host-mapped word a kernel writes when it gives up on a wait between its waves (k_rsos: RsSos::err)
    int out_stage_buf = -1;  // device staging for a host result
    int out_alias_buf = -1;  // pseudo buffer standing for the result (leaves of in-place root pieces point at it)
    void try_window_alias(std::vector<Piece>& rootp);
    int alias_stage = -1;    // stage whose kernel writes the final output directly
    bool alias_narrow = false;  // ... rounding its Float64 values to the Float32 result
    int64_t alias_skip = 0;     // ... from its local frame alias_skip on (an IIR's warm-up frames are not stored)
    bool interleaved_host = false;  // host result with frame_stride = nch, chan_stride = 1
    std::vector<char> host_tmp;
    int profiling = 0;  // 0 off, 1 per-execute (synchronising), 2 deferred (see plan_execute_direct)
    int64_t prof_execs = 0;
    std::vector<hipEvent_t> events;
    // independent step chains (Append children, Mix operands with their own filters ...) run on
    // separate HIP streams: the small latency-bound kernels of different chains overlap
    std::vector<std::vector<int>> step_deps;  // per step: earlier steps it must wait for
    std::vector<int> step_lane;               // per step: 0 = the caller's stream
    std::vector<char> step_signals;           // per step: a later step on another lane waits for it
    int nlanes = 1;
    std::vector<hipStream_t> lane_streams;  // [1..nlanes)
    std::vector<hipEvent_t> step_done;
    hipEvent_t ev_start = nullptr;
    void plan_lanes();
    // captured launch sequence (see plan_execute)
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t capture_stream = nullptr;
    const void* graph_out = nullptr;
    const void* last_out = nullptr;
    int64_t array_epoch = 0, graph_epoch = -1, last_epoch = -1;
    bool graph_failed = false;
    int64_t n_replays = 0, n_captures = 0, n_direct = 0;  // so_plan_counter
    so_stats_t stats{};
    int64_t algo_bytes = 0;
    std::map<int, bool> array_counted;

    // ---- helpers ---------------------------------------------------------
    int add_expr(const Expr& e) {
        exprs.push_back(e);
        return (int)exprs.size() - 1;
    }
    int new_buf(int64_t frames, int nch, int dtype) {
        Buf b;
        b.frames = frames;
        b.pitch = (frames + 63) / 64 * 64;
        if (b.pitch == 0) b.pitch = 64;
        b.nch = nch;
        b.dtype = dtype;
        b.bytes = (size_t)b.pitch * (size_t)std::max(nch, 1) * dsize(dtype);
        bufs.push_back(b);
        return (int)bufs.size() - 1;
    }
    int raw_buf(size_t bytes) {
        Buf b;
        b.bytes = std::max<size_t>(bytes, 8);
        b.dtype = SO_F64;
        bufs.push_back(b);
        return (int)bufs.size() - 1;
    }

    int mk_const(double v, int dtype) {
        Expr e;
        e.op = E_CONST;
        e.dtype = dtype;
        e.leaf.v0 = v;
        e.leaf.buf = -1;
        return add_expr(e);
    }
    int mk_un(int op, int a, int dtype) {
        Expr e;
        e.op = op;
        e.dtype = dtype;
        e.a = a;
        e.mono = exprs[a].mono;
        e.heavy = exprs[a].heavy;
        return add_expr(e);
    }
    bool is_const(int e, double v) const { return exprs[e].op == E_CONST && exprs[e].leaf.v0 == v; }
    int mk_bin(int op, int a, int b) {
        int ta = exprs[a].dtype, tb = exprs[b].dtype;
        int t = promote(ta, tb);
        if (op == E_DIV && t == SO_I64) t = SO_F64;
        // x*1 == x exactly (ramp flat regions, reference src/ramps.jl:56-59)
        if (op == E_MUL && is_const(b, 1.0)) return t == ta ? a : mk_un(E_RETYPE, a, t);
        if (op == E_MUL && is_const(a, 1.0)) return t == tb ? b : mk_un(E_RETYPE, b, t);
        if (exprs[a].op == E_CONST && exprs[b].op == E_CONST) {
            double x = exprs[a].leaf.v0, y = exprs[b].leaf.v0, r;
            switch (op) {
            case E_ADD: r = x + y; break;
            case E_SUB: r = x - y; break;
            case E_MUL: r = x * y; break;
            default: r = x / y;
            }
            return mk_const(roundto(t, r), t);
        }
        Expr e;
        e.op = op;
        e.dtype = t;
        e.a = a;
        e.b = b;
        e.mono = exprs[a].mono && exprs[b].mono;
        e.heavy = exprs[a].heavy || exprs[b].heavy;
        return add_expr(e);
    }

    // ---- model -----------------------------------------------------------
    void build_nodes(const so_node_t* in, int n);
    Len map_maxlen(Len x, Len y) const {
        if (x.k == LK_NUMEXT && y.k == LK_NUMEXT) return x;
        if (x.k == LK_INF || y.k == LK_INF) return Len{LK_INF, 0};
        int64_t a = x.k == LK_NUMEXT ? 0 : x.n, b = y.k == LK_NUMEXT ? 0 : y.n;
        return Len{LK_FIN, std::max(a, b)};
    }

    // ---- lowering --------------------------------------------------------
    std::vector<Piece> lower(int ni, Rect r, Map m);
    std::vector<Piece> lower_padded(int ni, int padkind, double padvalue, const double* padvec,
                                    Rect r, Map m, bool always_pad);
    std::vector<Piece> pad_pieces(int child, int padkind, double padvalue, const double* padvec,
                                  Rect r, Map m);
    std::vector<Piece> combine(const std::vector<std::vector<Piece>>& kids, Rect r, int op,
                               int force_dtype);
    int stage_for(int ni, int kind);
    void use_stage(Stage& S, const Rect& r, const Map& m);
    int in_norm = 0;  // > 0: lowering below a Normpower (stages created or used here are marked Stage::under_norm)
    int dry = 0;  // > 0: lower() only looks for the errors evaluating those frames raises (no stages, no buffers)
    void check_frames(int ni, int64_t upto);
    void process_stage(int sid);
    int emit_pointwise(const std::vector<Piece>& ps, int out_buf, int out_dtype);
    std::string rtc_expr(int e, std::vector<int>& monos, bool in_mono);
    std::vector<int>* rtc_loads = nullptr;  // rtc_source: the array leaves of the piece being written, read as frame pairs
    std::string rtc_source(const std::vector<Piece>& ps);
    bool match_carrier(int ei, DCarrier& C, std::vector<int>& monos);
    bool build_carriers(const std::vector<Piece>& ps, int nch, std::vector<DCarrier>& out, bool allow_ga = false, bool allow_arr2 = false);
    bool carrier_arr2_ok = false;  // (match_carrier: a step on a second array may be formed -- set by build_carriers for its matches)
    RsCtl make_ctl(const Stage& S) const;
    void gen(int e, std::vector<DOp>& code, std::map<int, int>& hoisted, std::vector<DOp>& fcode,
             bool allow_hoist);
    int depth(int e) const;
    int frame_slots(int e) const;
    int shift_expr(int e, int64_t a, int c0);
    int materialise(int e, const Rect& r, std::vector<int>& pre);
    int legalise(int e, const Rect& r, std::vector<int>& pre);
    void push_pw_step(int idx);
    int add_leaf(const Expr& e);
    void count_array(int ni);
    void fuse_state_passes();
    void fuse_resample_sos();
    void fuse_plain_sos();
    void batch_sos_stages();
    void sos_chunking(int sid, int64_t need, int nch, int dtype, const std::vector<SosCoefs>& groups, bool exact, int64_t target);
    void finalize();
    void release();
};

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};


// ---- stages.cpp ----
using Mat = std::vector<double>;
Mat matmul(const Mat& a, const Mat& b, int D);
double maxabs(const Mat& a);
Mat ident(int D);
Mat sos_state_matrix(const SosCoefs& cf);
Mat matpow(Mat A, int64_t e, int D);

// ---- rtc.cpp ----
const void* rtc_kernel(const std::string& body, int device, std::string& err);
const void* rtc_kernel_if_ready(const std::string& body, int device);
void rtc_wait_idle();
int rtc_compile_check(const std::string& body, std::string& err);
int rtc_launch(const void* fn, int64_t nblocks, const DPiece* d_pieces, int npieces, const DLeaf* d_leaves, OutView out,
               hipStream_t st);

// ---- accumulator.cpp ----
// One double per key between processes (SIGOPS_CACHE_DIR; accumulator.cpp): host-side analyses that depend on nothing but
// their key -- a cascade's rounding sensitivity -- and cost a first sink a millisecond each.  The file carries the key in
// full; false / no-op without a cache directory.
bool disk_value_get(const char* kind, const std::vector<double>& key, double& val);
void disk_value_put(const char* kind, const std::vector<double>& key, double val);
void replay_phase_accumulator(const RsGeom& g, const double* h, int hlen, int64_t need, bool bake,
                              std::vector<uint8_t>& prev, std::vector<RsFix>& fix, int64_t from = 0);
void rs_detect_exact(RsGeom& g, double fo, double fi, double rate);

}  // namespace so
