/*
 * sigops_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's block-pull sink engine
 * (haberdashPI/SignalOperators.jl v0.5.1) for the node kinds of include/sigops.h.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object; the product (libsigops) never links or calls it.
 *
 * Structure follows the reference one-to-one: every signal node has
 *     nextblock(x,maxlen,skip[,block])   and   frame(x,block,i)
 * and `sink!` pulls blocks and writes frames one by one
 * (src/sink.jl:225-241,256-267).  Citations are given per function.
 *
 * Third-party numerics (DSP.jl 0.6.10, docs/Manifest.toml:48-52; source not in
 * /root/reference) are restated from the published algorithm as recorded in
 * DF2T second-order sections `filt!`, polyphase FIR kernels
 * (FIRInterpolator / FIRDecimator / FIRRational / FIRArbitrary), `setphase!`,
 * `timedelay`.
 *
 * PARITY PINNING: the structural operators are pinned by the reference's own
 * exact known-answer tests (tests/test_oracle_golden.py <- test/runtests.jl, see
 * SURVEY.md Appendix E).  For Filt / ToFramerate the reference holds NO golden
 * vectors ("parity unpinned" for those values): they are pinned instead against
 * SciPy (sosfilt, upfirdn/firwin) and analytic sines in tests/test_oracle_dsp.py.
 *
 * Resampler positions: the arbitrary-rate kernel (every ratio with max(num,den) > 3,
 * src/reformatting.jl:103-111) follows DSP.jl's FIRArbitrary: the phase is ACCUMULATED in
 * Float64, one addition of Delta per output (fir_block below).  That is the default.
 *
 * Documented divergences from the reference:
 *   C-1  a block emits exactly the outputs whose newest input lies in it, i.e. what DSP.jl's
 *        filt! writes (DSP.outputlength, which the reference uses for the block's length, can
 *        over-count by 1-2 and would expose stale rows of the output buffer).
 *        Measurement aid, NOT the default: SO_ORACLE_EXACT_POSITIONS=1 (or
 *        so_oracle_set_positions(1)) positions every output with the closed form
 *        q_m = c0 + m*Delta -- exact integer arithmetic q_m = c0 + m*Nphi*M/L when both frame
 *        rates are integers (e.g. 44100 -> 48000).  The two differ at the wrap-around ties
 *        (alpha == 0 at phase 0), which accumulated rounding error resolves to
 *        (previous input, last phase, alpha ~ 1): DSP.jl then drops the tap h[0].
 *   C-2  NormedSignal honours its block offset (intended semantics).
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off; `make native` = -march=native for timing).
 */
#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sigops.h"

/* ------------------------------------------------------------------------- */
/* error handling: reference `error(msg)` -> longjmp to the entry point       */
static __thread char g_err[512];
static __thread jmp_buf g_jmp;
static __thread int g_status;

static void fail(int status, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    g_status = status;
    longjmp(g_jmp, 1);
}

const char* so_oracle_last_error(void) { return g_err; }

/* simple arena so that longjmp cannot leak */
typedef struct Chunk { struct Chunk* next; } Chunk;
static __thread Chunk* g_arena;
static void* xalloc(size_t n) {
    Chunk* c = (Chunk*)calloc(1, sizeof(Chunk) + n + 16);
    if (!c) fail(SO_ERR_RUNTIME, "oracle: out of memory");
    c->next = g_arena;
    g_arena = c;
    return (void*)(c + 1);
}
static void arena_free(void) {
    while (g_arena) {
        Chunk* n = g_arena->next;
        free(g_arena);
        g_arena = n;
    }
}

/* ------------------------------------------------------------------------- */
/* lengths: src/inflen.jl, src/signal.jl:28-37 (Extended), src/numbers.jl:5-9  */
enum { L_FIN = 0, L_INF = 1, L_EXT = 2, L_NUMEXT = 3 };
typedef struct { int k; int64_t n; } Len;
static Len len_fin(int64_t n) { Len l = {L_FIN, n}; return l; }
static Len len_inf(void) { Len l = {L_INF, 0}; return l; }
static int len_isinf(Len l) { return l.k != L_FIN; }
#define BIG ((int64_t)1 << 62)
/* nframes(x) = cleanextend(nframes_helper(x)) as an int64 (BIG == inflen) */
static int64_t len_clean(Len l) { return l.k == L_FIN ? l.n : BIG; }
static int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; }
static int64_t imax(int64_t a, int64_t b) { return a > b ? a : b; }

/* value type promotion (Julia promote_type on Float32/Float64/Int64) */
static int promote(int a, int b) {
    if (a == SO_F64 || b == SO_F64) return SO_F64;
    if (a == SO_F32 || b == SO_F32) return SO_F32;
    return SO_I64;
}
static int float_of(int t) { return t == SO_I64 ? SO_F64 : t; }
static double roundto(int t, double v) {
    if (t == SO_F32) return (double)(float)v;
    return v;
}

/* ------------------------------------------------------------------------- */
typedef struct OSig OSig;
struct OSig {
    const so_node_t* nd;
    int kind, dtype, nch;
    double fs;
    Len len; /* nframes_helper */
    int nkids;
    OSig** kids;
    /* PAD synthesized by MAP (Extend.(signals,padding), src/mapsignal.jl:26) or by
       FILT (Pad(x.signal,zero), src/filters.jl:240) */
    int pad_kind, pad_extend;
    double pad_value;
    const double* pad_vec;
    so_node_t synth; /* backing node for synthesized PADs */
};

static int g_blocksize_override; /* 0 = use the node's */
static int g_phase_accumulate = 1;
static int g_position_mode = -1; /* -1: environment decides; 0: accumulator; 1: closed form */
static int g_intended; /* 0: the reference's behaviour, quirks included; 1: intended semantics */

/* Resampler positions of the arbitrary-rate kernel.  0 (default) = DSP.jl's floating-point
   phase accumulator, the reference's algorithm; 1 = closed form (exact rational positions for
   integer frame rates), kept for divergence measurements; -1 = SO_ORACLE_EXACT_POSITIONS=1 in
   the environment selects the closed form. */
void so_oracle_set_positions(int mode) { g_position_mode = mode; }

/* 0 (default) = the reference as it behaves, including quirk C-7: a FilteredSignal's end test
   compares a buffer-local index with the global length (src/filters.jl:224-227), so a filtered or
   resampled child longer than one block never reports its end and a parent Append / Pad / Mix /
   After keeps pulling the filter's zero-input tail.  1 = intended semantics: the child ends after
   nframes(x) frames (what the documented meaning of those operators implies, and what the
   engine implements).  Used by the multi-rate fuzz tests. */
void so_oracle_set_semantics(int mode) { g_intended = mode; }

/* DSP.filt(b, a, x::AbstractSignal, si) (reference src/filters.jl:68-87): `filt!(data, b, a, sink(x, Array), si)`,
   i.e. DSP.jl 0.6.10's direct-form II transposed recurrence of order max(|b|, |a|) - 1 on the sunk samples, one
   channel after the other (`_filt_iir!`: coefficients divided by a[1] first; per sample
       y = si[1] + b[1] x;  si[j] = si[j+1] + b[j+1] x - a[j+1] y;  si[end] = b[end] x - a[end] y).
   x, y: n frames x nch channels, channel-major (chan_stride frames apart); si: `ord` entries per channel
   (si_stride apart; 0 = the same vector for every channel; NULL = rest).  Pinned against scipy.signal.lfilter in
   tests/test_oracle_dsp.py (the same recurrence). */
int so_oracle_filt_direct(const double* b, int nb, const double* a, int na, const double* x, double* y, int64_t n,
                          int nch, int64_t chan_stride, const double* si, int64_t si_stride) {
    int sz = nb > na ? nb : na, ord = sz - 1;
    if (nb < 1 || na < 1 || a[0] == 0.0) return -1;
    double* bn = (double*)calloc((size_t)sz * 3, sizeof(double));
    double *an = bn + sz, *st = an + sz;
    for (int i = 0; i < nb; ++i) bn[i] = b[i] / a[0];
    for (int i = 0; i < na; ++i) an[i] = a[i] / a[0];
    for (int ch = 0; ch < nch; ++ch) {
        for (int j = 0; j < ord; ++j) st[j] = si ? si[(int64_t)ch * si_stride + j] : 0.0;
        const double* xc = x + (int64_t)ch * chan_stride;
        double* yc = y + (int64_t)ch * chan_stride;
        for (int64_t i = 0; i < n; ++i) {
            double xi = xc[i];
            double yi = (ord ? st[0] : 0.0) + bn[0] * xi;
            for (int j = 0; j + 1 < ord; ++j) st[j] = st[j + 1] + bn[j + 1] * xi - an[j + 1] * yi;
            if (ord) st[ord - 1] = bn[ord] * xi - an[ord] * yi;
            yc[i] = yi;
        }
    }
    free(bn);
    return 0;
}

static double sinpi_(double x) {
    /* Julia sinpi: exact argument reduction, src/functions.jl:57-60 use it */
    double r = fmod(x, 2.0);
    double n = nearbyint(2.0 * r);
    double t = r - 0.5 * n; /* exact, |t| <= 0.25 */
    int q = ((int)n % 4 + 4) % 4;
    switch (q) {
    case 0: return sin(M_PI * t);
    case 1: return cos(M_PI * t);
    case 2: return -sin(M_PI * t);
    default: return -cos(M_PI * t);
    }
}

static OSig* build(const so_node_t* nodes, int32_t n_nodes, int32_t idx);

static OSig* make_pad(OSig* child, int kind, int extend, double value, const double* vec) {
    OSig* s = (OSig*)xalloc(sizeof(OSig));
    s->synth.kind = SO_NODE_PAD;
    s->nd = &s->synth;
    s->kind = SO_NODE_PAD;
    s->dtype = child->dtype;
    s->nch = child->nch;
    s->fs = child->fs;
    s->nkids = 1;
    s->kids = (OSig**)xalloc(sizeof(OSig*));
    s->kids[0] = child;
    s->pad_kind = kind;
    s->pad_extend = extend;
    s->pad_value = value;
    s->pad_vec = vec;
    if (extend) {
        s->len.k = L_EXT;
        s->len.n = len_clean(child->len);
    } else {
        s->len = len_inf();
    }
    return s;
}

/* Extend(x,p) = isknowninf(nframes(x)) ? x : PaddedSignal(x,p,true)  src/padding.jl:97-101 */
static OSig* extend_sig(OSig* x, int kind, double value) {
    if (len_isinf(x->len)) return x;
    return make_pad(x, kind, 1, value, NULL);
}

/* maxlen / tolen: src/mapsignal.jl:147-154 */
static Len map_maxlen(Len x, Len y) {
    if (x.k == L_NUMEXT && y.k == L_NUMEXT) return x;
    int xi = (x.k == L_INF), yi = (y.k == L_INF);
    if (xi || yi) return len_inf();
    int64_t a = (x.k == L_NUMEXT) ? 0 : x.n;
    int64_t b = (y.k == L_NUMEXT) ? 0 : y.n;
    return len_fin(imax(a, b));
}

static OSig* build(const so_node_t* nodes, int32_t n_nodes, int32_t idx) {
    if (idx < 0 || idx >= n_nodes) fail(SO_ERR_INVALID, "oracle: bad node index %d", idx);
    const so_node_t* nd = &nodes[idx];
    OSig* s = (OSig*)xalloc(sizeof(OSig));
    s->nd = nd;
    s->kind = nd->kind;
    s->dtype = nd->dtype;
    s->nch = nd->nch;
    s->fs = nd->fs;
    s->nkids = nd->n_children;
    s->kids = (OSig**)xalloc(sizeof(OSig*) * (size_t)(nd->n_children + 1));
    for (int j = 0; j < nd->n_children; ++j) {
        if (nd->children[j] >= idx) fail(SO_ERR_INVALID, "oracle: node table not in post-order");
        s->kids[j] = build(nodes, n_nodes, nd->children[j]);
    }
    OSig* c0 = s->nkids ? s->kids[0] : NULL;
    switch (nd->kind) {
    case SO_NODE_ARRAY: s->len = len_fin(nd->l0); break;
    case SO_NODE_CONST: s->len.k = L_NUMEXT; s->len.n = 0; s->nch = 1; break;
    case SO_NODE_FUNC: s->len = len_inf(); break;
    case SO_NODE_UNTIL: { /* src/cutting.jl:130 */
        int64_t L = imax(0, nd->l0);
        s->len = len_isinf(c0->len) ? len_fin(L) : len_fin(imin(c0->len.n, L));
        break;
    }
    case SO_NODE_AFTER: { /* src/cutting.jl:134 */
        if (len_isinf(c0->len)) s->len = c0->len;
        else {
            int64_t v = c0->len.n - nd->l0;
            if (v < 0) v = 0;
            if (v > c0->len.n) v = c0->len.n;
            s->len = len_fin(v);
        }
        break;
    }
    case SO_NODE_PAD: /* src/padding.jl:13-14 */
        s->pad_kind = nd->i0;
        s->pad_extend = nd->i1;
        s->pad_value = nd->d0;
        s->pad_vec = (const double*)nd->p0;
        if (nd->i1) {
            s->len.k = L_EXT;
            s->len.n = len_clean(c0->len);
        } else s->len = len_inf();
        break;
    case SO_NODE_APPEND: { /* src/appending.jl:59-76 */
        int64_t tot = 0;
        int inf = 0;
        for (int j = 0; j < s->nkids; ++j) {
            if (len_isinf(s->kids[j]->len)) {
                if (j < s->nkids - 1) fail(SO_ERR_LENGTH, "Cannot Append to the end of an infinite signal");
                inf = 1;
            } else tot += s->kids[j]->len.n;
        }
        s->len = inf ? len_inf() : len_fin(tot);
        break;
    }
    case SO_NODE_RAMP: s->len = c0->len; break; /* WrappedSignal src/wrapping.jl:16 */
    case SO_NODE_MAP: {
        Len l = s->kids[0]->len;
        for (int j = 1; j < s->nkids; ++j) l = map_maxlen(l, s->kids[j]->len);
        if (s->nkids == 1 && l.k == L_EXT) { /* reduce over one element returns it */ }
        s->len = l;
        /* padded_signals = Extend.(signals, padding)  src/mapsignal.jl:26 */
        for (int j = 0; j < s->nkids; ++j)
            s->kids[j] = extend_sig(s->kids[j], nd->i2, nd->d0);
        break;
    }
    case SO_NODE_FILT_SOS: s->len = c0->len; break; /* src/filters.jl:162-163 */
    case SO_NODE_RESAMPLE: { /* src/filters.jl:165 */
        if (len_isinf(c0->len)) s->len = c0->len;
        else s->len = len_fin((int64_t)ceil((double)c0->len.n * nd->fs / c0->fs));
        break;
    }
    case SO_NODE_NORMPOWER: s->len = c0->len; break;
    default: fail(SO_ERR_INVALID, "oracle: unknown node kind %d", nd->kind);
    }
    return s;
}

/* ------------------------------------------------------------------------- */
/* DSP.jl streaming FIR kernel.  One per channel.       */
typedef struct {
    int arbitrary;
    int64_t L, M; /* rational: Nphi=L, decimation M */
    int nphi, taps; /* tapsPerphi */
    int hlen;
    const double* h;
    double* pfb;  /* [nphi][taps], pfb[p*taps+k] = h[p + nphi*k] (k = tap age) */
    double* dpfb;
    double rate, delta;
    int exact;     /* arbitrary kernel with integer frame rates: exact rational positions */
    int64_t eL, eM;
    double c0;     /* (hlen-1)/2 : fine-grid position of output 0 */
    int64_t m;     /* next output index (closed form) */
    /* accumulator mode (DSP.jl FIRArbitrary.update) */
    double phi_acc;
    int64_t x_idx_global; /* global 1-based newest-input index of next output */
    double* hist;  /* last taps-1 inputs, hist[taps-2] newest */
    int64_t consumed; /* inputs consumed so far */
} Fir;

static int64_t gcd64(int64_t a, int64_t b) {
    while (b) {
        int64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

static void fir_init(Fir* f, const so_node_t* nd, double fs_in) {
    memset(f, 0, sizeof *f);
    f->arbitrary = (nd->i0 == SO_RS_ARBITRARY);
    f->hlen = nd->i2;
    f->h = (const double*)nd->p0;
    if (f->arbitrary) {
        f->nphi = nd->i1;
        f->rate = nd->d0;
        f->delta = (double)f->nphi / f->rate; /* FIRArbitrary: Δ = Nϕ/rate */
        double fo = nd->fs, fi = fs_in;
        if (fo == floor(fo) && fi == floor(fi) && fo >= 1 && fi >= 1 && fo < 2147483648.0 &&
            fi < 2147483648.0 && fo / fi == f->rate && !g_phase_accumulate) {
            int64_t g = gcd64((int64_t)fo, (int64_t)fi);
            int64_t L = (int64_t)fo / g, M = (int64_t)fi / g;
            if (L <= 8192 && M <= 1048576) {
                f->exact = 1;
                f->eL = L;
                f->eM = M;
            }
        }
    } else {
        f->L = nd->l0;
        f->M = nd->l1;
        if (nd->i0 == SO_RS_FIR) f->L = f->M = 1; /* Filt(x,h): DF2TFilter(PolynomialRatio(h,[1])) */
        f->nphi = (int)f->L;
    }
    f->taps = (f->hlen + f->nphi - 1) / f->nphi; /* taps2pfb: ceil(hLen/Nϕ) */
    size_t sz = (size_t)f->nphi * (size_t)f->taps;
    f->pfb = (double*)xalloc(sizeof(double) * sz);
    f->dpfb = (double*)xalloc(sizeof(double) * sz);
    for (int p = 0; p < f->nphi; ++p)
        for (int k = 0; k < f->taps; ++k) {
            int64_t hi = p + (int64_t)f->nphi * k;
            double hv = hi < f->hlen ? f->h[hi] : 0.0;
            /* dh = [diff(h); 0] */
            double dv = (hi + 1 < f->hlen) ? f->h[hi + 1] - f->h[hi] : 0.0;
            f->pfb[(size_t)p * f->taps + k] = hv;
            f->dpfb[(size_t)p * f->taps + k] = dv;
        }
    /* timedelay + setphase!: output 0 sits at fine position c0=(hLen-1)/2 */
    f->c0 = nd->i0 == SO_RS_FIR ? 0.0 : (double)(f->hlen - 1) / 2.0; /* a plain FIR is causal */
    f->hist = (double*)xalloc(sizeof(double) * (size_t)(f->taps + 1));
    if (f->arbitrary) {
        double tau = (double)(f->hlen - 1) / 2.0 / (double)f->nphi;
        double w = floor(tau), fr = tau - w;
        f->x_idx_global = 1 + (int64_t)llround(w);
        f->phi_acc = fr * f->nphi + 1.0;
    }
}

/* closed-form position of output m: j = newest input (0-based), p = phase, alpha */
static void fir_pos(const Fir* f, int64_t m, int64_t* j, int* p, double* alpha) {
    if (f->arbitrary && f->exact) {
        int64_t N = m * ((int64_t)f->nphi * f->eM);
        int64_t qi = (int64_t)f->c0 + N / f->eL;
        *alpha = (double)(N % f->eL) / (double)f->eL;
        *j = qi / f->nphi;
        *p = (int)(qi % f->nphi);
    } else if (f->arbitrary) {
        double t = (double)m * f->delta;
        double q = f->c0 + t;
        double fl = floor(q);
        int64_t qi = (int64_t)fl;
        *alpha = q - fl;
        *j = qi / f->nphi;
        *p = (int)(qi % f->nphi);
    } else {
        int64_t qi = (int64_t)f->c0 + m * f->M;
        *alpha = 0.0;
        *j = qi / f->L;
        *p = (int)(qi % f->L);
    }
}

/* filter one block x[0..n) (global input offset f->consumed); append outputs */
static int64_t fir_block(Fir* f, const double* x, int64_t n, double* out, int64_t outcap,
                         int count_only) {
    int64_t produced = 0;
    int64_t base = f->consumed;
    int T = f->taps;
    if (f->arbitrary && g_phase_accumulate) {
        /* DSP.jl FIRArbitrary filt!/update, phase accumulated sample by sample */
        int64_t xidx = f->x_idx_global; /* 1-based global */
        double acc = f->phi_acc;
        int64_t m = f->m;
        while (xidx <= base + n) {
            int pidx = (int)floor(acc);
            double a = acc - pidx;
            if (!count_only) {
                if (produced >= outcap) fail(SO_ERR_RUNTIME, "oracle: resampler output overflow");
                double lo = 0, hi = 0;
                const double* pf = f->pfb + (size_t)(pidx - 1) * T;
                const double* df = f->dpfb + (size_t)(pidx - 1) * T;
                for (int k = T - 1; k >= 0; --k) {
                    int64_t gi = xidx - 1 - k; /* 0-based global input */
                    double xv;
                    int64_t li = gi - base;
                    if (li >= 0) xv = x[li];
                    else {
                        int64_t hidx = (T - 1) + li; /* hist has T-1 entries */
                        xv = hidx >= 0 ? f->hist[hidx] : 0.0;
                    }
                    lo += pf[k] * xv;
                    hi += df[k] * xv;
                }
                out[produced] = lo + hi * a;
            }
            produced++;
            m++;
            acc += f->delta;
            if (acc > f->nphi) {
                xidx += (int64_t)floor((acc - 1) / f->nphi);
                acc = fmod(acc - 1, (double)f->nphi) + 1;
            }
        }
        if (!count_only) {
            f->x_idx_global = xidx;
            f->phi_acc = acc;
            f->m = m;
        }
    } else {
        int64_t m = f->m;
        for (;;) {
            int64_t j;
            int p;
            double a;
            fir_pos(f, m, &j, &p, &a);
            if (j >= base + n) break;
            if (!count_only) {
                if (produced >= outcap) fail(SO_ERR_RUNTIME, "oracle: resampler output overflow");
                double lo = 0, hi = 0;
                const double* pf = f->pfb + (size_t)p * T;
                const double* df = f->dpfb + (size_t)p * T;
                for (int k = T - 1; k >= 0; --k) {
                    int64_t li = j - k - base;
                    double xv;
                    if (li >= 0) xv = x[li];
                    else {
                        int64_t hidx = (T - 1) + li;
                        xv = hidx >= 0 ? f->hist[hidx] : 0.0;
                    }
                    lo += pf[k] * xv;
                    if (f->arbitrary) hi += df[k] * xv;
                }
                out[produced] = f->arbitrary ? lo + hi * a : lo;
            }
            produced++;
            m++;
        }
        if (!count_only) f->m = m;
    }
    if (!count_only) {
        /* shiftin!(history, x) */
        int H = T - 1;
        if (n >= H) {
            for (int k = 0; k < H; ++k) f->hist[k] = x[n - H + k];
        } else {
            for (int k = 0; k + n < H; ++k) f->hist[k] = f->hist[k + n];
            for (int64_t k = 0; k < n; ++k) f->hist[H - n + k] = x[k];
        }
        f->consumed += n;
    }
    return produced;
}

/* ------------------------------------------------------------------------- */
typedef struct OState OState;
struct OState {
    OSig* s;
    int started;
    int64_t len;    /* nframes(block) */
    int64_t offset; /* meaning per kind */
    OState** kids;
    int64_t gout; /* FILT/RESAMPLE: global index of the current block's first frame */
    /* UNTIL */
    int64_t cut_n;
    /* PAD */
    int pad_mode; /* 0 child, 1 padding */
    double* padvals;
    int pad_dtype;
    /* APPEND */
    int k;
    /* RAMP */
    int ramp_active; /* block.Ramp !== nothing */
    int64_t marker, stop;
    /* MAP */
    int64_t* offsets;
    double* scratch; /* nkids * maxch */
    int* kdt;
    int child_nothing; /* AFTER/CUT: child returned nothing */
    /* FILT */
    int64_t last_output_index, available_output;
    double* input;  /* [rows_in * nch] column-major */
    double* output; /* [blocksize * nch] */
    int64_t rows_in, blocksize;
    double* sos_state; /* [nch][nsec][2] */
    Fir* firs;
    int filt_child_started;
    /* NORMPOWER */
    double* vals;
    int64_t nvals;
};

static OState* mkstate(OSig* s) {
    OState* st = (OState*)xalloc(sizeof(OState));
    st->s = s;
    st->kids = (OState**)xalloc(sizeof(OState*) * (size_t)(s->nkids + 1));
    for (int j = 0; j < s->nkids; ++j) st->kids[j] = mkstate(s->kids[j]);
    return st;
}
static int nextblock(OState* st, int64_t maxlen, int skip);
static int frame(OState* st, int64_t i, double* out); /* returns value dtype */

/* sink!(result,x,::IsSignal,block) src/sink.jl:227-241 into a column-major matrix.
 * `have_block`: a current block already exists in st. Returns frames written. */
static int64_t pull_into(OState* st, double* dst, int64_t rows, int nch, int dtype,
                         int have_block) {
    int64_t written = 0;
    int ok = have_block ? 1 : nextblock(st, rows, 0);
    double tmp[4096];
    double* fr = nch <= 4096 ? tmp : (double*)xalloc(sizeof(double) * (size_t)nch);
    while (ok && written < rows) {
        if (st->len <= 0) fail(SO_ERR_RUNTIME, "oracle: @assert nframes(block) > 0");
        for (int64_t i = 1; i <= st->len; ++i) { /* sink_helper! src/sink.jl:256-260 */
            frame(st, i, fr);
            for (int ch = 0; ch < nch; ++ch) /* writesink! :262-267 (convert) */
                dst[(written + i - 1) + (int64_t)ch * rows] = roundto(dtype, fr[ch]);
        }
        written += st->len;
        int64_t ml = rows - written;
        if (ml > 0) ok = nextblock(st, ml, 0);
    }
    return written;
}

static double ramp_fn(int code, double x) {
    return code == SO_RAMP_SINRAMP ? sinpi_(0.5 * x) : x;
}

static int nextblock(OState* st, int64_t maxlen, int skip) {
    OSig* s = st->s;
    const so_node_t* nd = s->nd;
    switch (s->kind) {
    case SO_NODE_ARRAY: { /* src/arrays.jl:126-132 */
        int64_t offset = st->started ? st->offset + st->len : 0;
        st->started = 1;
        if (offset < nd->l0) {
            st->len = imin(maxlen, nd->l0 - offset);
            st->offset = offset;
            return 1;
        }
        return 0;
    }
    case SO_NODE_CONST: /* src/numbers.jl:61-62 */
        st->len = maxlen;
        return 1;
    case SO_NODE_FUNC: /* src/functions.jl:48-51 */
        st->offset = st->started ? st->offset + st->len : 0;
        st->started = 1;
        st->len = maxlen;
        return 1;
    case SO_NODE_UNTIL: { /* src/cutting.jl:198-210 */
        if (!st->started) {
            st->started = 1;
            st->cut_n = nd->l0;
            st->len = 0;
        }
        int64_t nextlen = st->cut_n - st->len;
        if (nextlen > 0) {
            int ok = nextblock(st->kids[0], imin(nextlen, maxlen), skip);
            if (ok) {
                st->cut_n = nextlen;
                st->len = st->kids[0]->len;
                return 1;
            }
        }
        return 0;
    }
    case SO_NODE_AFTER: { /* src/cutting.jl:160-192 */
        OState* c = st->kids[0];
        if (!st->started) {
            st->started = 1;
            int64_t len = imax(0, nd->l0);
            if (len > 0) {
                int ok = nextblock(c, len, 1);
                int64_t skipped = ok ? c->len : 0;
                while (ok && skipped < len) {
                    ok = nextblock(c, imin(maxlen, len - skipped), 1);
                    if (!ok) break;
                    skipped += c->len;
                }
                if (skipped < len) fail(SO_ERR_LENGTH, "Signal is too short to skip %lld frames", (long long)len);
            }
        }
        int ok = nextblock(c, maxlen, skip);
        if (!ok) return 0;
        st->len = c->len;
        return 1;
    }
    case SO_NODE_PAD: { /* src/padding.jl:212-235 */
        OState* c = st->kids[0];
        if (st->pad_mode == 1) {
            st->offset = st->len + st->offset;
            st->len = maxlen;
            return 1;
        }
        int64_t newoff = st->started ? st->len + st->offset : 0;
        int had_block = st->started;
        int ok = nextblock(c, maxlen, skip);
        if (ok) {
            st->started = 1;
            st->offset = newoff;
            st->len = c->len;
            return 1;
        }
        /* usepad(x,block) src/padding.jl:150-192 */
        int T = s->kids[0]->dtype;
        st->padvals = (double*)xalloc(sizeof(double) * (size_t)s->nch);
        st->pad_dtype = T;
        switch (s->pad_kind) {
        case SO_PAD_VALUE:
            for (int ch = 0; ch < s->nch; ++ch) st->padvals[ch] = roundto(T, s->pad_value);
            break;
        case SO_PAD_VECTOR:
            for (int ch = 0; ch < s->nch; ++ch) st->padvals[ch] = roundto(T, s->pad_vec[ch]);
            break;
        case SO_PAD_ZERO:
            for (int ch = 0; ch < s->nch; ++ch) st->padvals[ch] = 0.0;
            break;
        case SO_PAD_ONE:
            for (int ch = 0; ch < s->nch; ++ch) st->padvals[ch] = 1.0;
            break;
        case SO_PAD_LASTFRAME:
            if (!had_block) fail(SO_ERR_LENGTH, "Signal is length zero; there is no last frame to pad with.");
            /* frame(x,block,nframes(block)) on the previous (still current) block */
            st->pad_dtype = frame(c, c->len, st->padvals);
            break;
        case SO_PAD_CYCLE:
        case SO_PAD_MIRROR:
            if (s->kids[0]->kind != SO_NODE_ARRAY)
                fail(SO_ERR_INVALID, "Attemped to specify an indexing pad function for a signal which is not known to support `getindex`.");
            if (s->kids[0]->nd->l0 == 0) fail(SO_ERR_LENGTH, "cannot index an empty array");
            break;
        default: fail(SO_ERR_INVALID, "oracle: bad pad kind");
        }
        st->pad_mode = 1;
        st->started = 1;
        st->offset = newoff;
        st->len = maxlen;
        return 1;
    }
    case SO_NODE_APPEND: { /* src/appending.jl:92-110 */
        int K = s->nkids;
        if (!st->started) {
            st->started = 1;
            st->k = 0;
        }
        int k0 = st->k;
        int ok = nextblock(st->kids[st->k], maxlen, skip);
        while (st->k < K - 1 && !ok) {
            st->k++;
            ok = nextblock(st->kids[st->k], maxlen, skip);
        }
        if (!ok) {
            /* advancechild returns nothing and the caller keeps its last AppendBlock, which still names
               the child that produced it (blocks are immutable, src/appending.jl:82-110): a `lastframe`
               pad after Append(x, <empty>) takes the last frame of x */
            st->k = k0;
            return 0;
        }
        st->len = st->kids[st->k]->len;
        return 1;
    }
    case SO_NODE_RAMP: { /* src/ramps.jl:74-119 */
        int64_t N = len_clean(s->len);
        int64_t R = nd->l0;
        if (nd->i0 == 0) { /* :on */
            if (!st->started) {
                st->started = 1;
                st->ramp_active = 1;
                st->marker = R;
                st->stop = N;
                st->offset = 0;
                st->len = imin(R, maxlen);
                return 1;
            }
            if (st->ramp_active) {
                int64_t offset = st->offset + st->len;
                int64_t len = imin(imin(N - offset, maxlen), st->marker - offset);
                st->offset = offset;
                if (len == 0) {
                    st->len = imin(N - offset, maxlen);
                    st->ramp_active = 0;
                } else st->len = len;
                return 1;
            } else {
                int64_t offset = st->offset + st->len;
                int64_t len = imin(imin(N - offset, maxlen), st->stop - offset);
                if (len > 0) {
                    st->offset = offset;
                    st->len = len;
                    return 1;
                }
                return 0;
            }
        } else { /* :off */
            if (!st->started) {
                int64_t rampstart = N - R;
                if (rampstart < 0)
                    fail(SO_ERR_UNSUPPORTED, "RampOff longer than the signal is undefined in the reference (src/ramps.jl:79-80)");
                st->started = 1;
                st->ramp_active = 0;
                st->marker = rampstart;
                st->stop = N;
                st->offset = 0;
                st->len = imin(rampstart, maxlen);
                if (st->len == 0) { /* zero-length flat part: go straight to the ramp */
                    st->len = imin(N, maxlen);
                    st->ramp_active = 1;
                }
                return 1;
            }
            if (!st->ramp_active) {
                int64_t offset = st->offset + st->len;
                int64_t len = imin(imin(N - offset, maxlen), st->marker - offset);
                st->offset = offset;
                if (len == 0) {
                    st->len = imin(N - offset, maxlen);
                    st->ramp_active = 1;
                } else st->len = len;
                return 1;
            } else {
                int64_t offset = st->offset + st->len;
                int64_t len = imin(imin(N - offset, maxlen), st->stop - offset);
                if (len > 0) {
                    st->offset = offset;
                    st->len = len;
                    return 1;
                }
                return 0;
            }
        }
    }
    case SO_NODE_MAP: { /* src/mapsignal.jl:216-244 */
        int N = s->nkids;
        if (!st->started) {
            st->started = 1;
            st->len = 0;
            st->offset = 0;
            st->offsets = (int64_t*)xalloc(sizeof(int64_t) * (size_t)N);
            int maxch = s->nch;
            for (int j = 0; j < N; ++j)
                if (s->kids[j]->nch > maxch) maxch = s->kids[j]->nch;
            st->scratch = (double*)xalloc(sizeof(double) * (size_t)N * (size_t)maxch);
            st->kdt = (int*)xalloc(sizeof(int) * (size_t)N);
            st->cut_n = maxch;
            for (int j = 0; j < N; ++j) st->kids[j]->len = 0; /* emptychild */
        }
        int64_t total = len_clean(s->len);
        maxlen = imin(maxlen, total - (st->offset + st->len));
        if (maxlen == 0) return 0;
        for (int j = 0; j < N; ++j) {
            int64_t off = st->offsets[j] + st->len;
            if (off == st->kids[j]->len) off = 0;
            st->offsets[j] = off;
        }
        for (int j = 0; j < N; ++j)
            if (st->offsets[j] == 0) {
                if (!nextblock(st->kids[j], maxlen, skip))
                    fail(SO_ERR_RUNTIME, "oracle: MapSignal child ended (should be extended)");
            }
        int64_t len = maxlen;
        for (int j = 0; j < N; ++j) len = imin(len, st->kids[j]->len - st->offsets[j]);
        st->offset = st->offset + st->len;
        st->len = len;
        return 1;
    }
    case SO_NODE_FILT_SOS:
    case SO_NODE_RESAMPLE: { /* src/filters.jl:204-262 */
        int nch = s->nch;
        int is_rs = (s->kind == SO_NODE_RESAMPLE);
        int64_t total = len_clean(s->len);
        if (!st->started) { /* FilterBlock(x) :204-211 */
            st->started = 1;
            st->blocksize = g_blocksize_override ? g_blocksize_override : (is_rs ? nd->i3 : nd->i1);
            if (st->blocksize <= 0) st->blocksize = 4096;
            st->len = 0;
            st->last_output_index = 0;
            st->available_output = 0;
            if (is_rs) {
                st->firs = (Fir*)xalloc(sizeof(Fir) * (size_t)nch);
                for (int ch = 0; ch < nch; ++ch) fir_init(&st->firs[ch], nd, s->kids[0]->fs); /* per channel: :205 */
                double ratio = st->firs[0].arbitrary ? st->firs[0].rate : (double)nd->l0 / (double)nd->l1;
                /* init_length :185-199 */
                int64_t n = (int64_t)trunc(fmax(1.0, (double)imin(total, st->blocksize) / ratio));
                int64_t out = fir_block(&st->firs[0], NULL, n, NULL, 0, 1);
                if (out <= 0) {
                    n = (int64_t)trunc(fmax(1.0, (double)st->blocksize / ratio));
                    out = fir_block(&st->firs[0], NULL, n, NULL, 0, 1);
                    if (out <= 0) fail(SO_ERR_INVALID, "Blocksize is too small for this resampling filter.");
                }
                st->rows_in = n;
                /* output buffer: the reference uses blocksize rows; the closed-form
                   count can exceed it by rounding, so keep slack */
                int64_t cap = (int64_t)ceil((double)n * ratio) + 4;
                if (cap < st->blocksize) cap = st->blocksize;
                st->output = (double*)xalloc(sizeof(double) * (size_t)cap * (size_t)nch);
                st->cut_n = cap; /* output rows capacity */
            } else {
                st->rows_in = imin(total, st->blocksize);
                int nsec = nd->i0;
                st->sos_state = (double*)xalloc(sizeof(double) * (size_t)nch * (size_t)nsec * 2);
                st->output = (double*)xalloc(sizeof(double) * (size_t)st->blocksize * (size_t)nch);
                st->cut_n = st->blocksize;
            }
            st->input = (double*)xalloc(sizeof(double) * (size_t)imax(st->rows_in, 1) * (size_t)nch);
        }
        int64_t last_output_index = st->last_output_index + st->len;
        if (total == last_output_index) return 0; /* :225-227 (quirk C-7 kept) */
        st->gout += st->len; /* frames handed out so far */
        if (g_intended) {
            if (st->gout >= total) return 0;
            maxlen = imin(maxlen, total - st->gout);
        }
        if (last_output_index < st->available_output) { /* leftover :230-235 */
            st->len = imin(maxlen, st->available_output - last_output_index);
            st->last_output_index = last_output_index;
            return 1;
        }
        /* refill :237-261; psig = Pad(x.signal,zero) is kids[0] (synthesized at mkfilt) */
        OState* c = st->kids[0];
        int64_t rows = st->rows_in;
        int have = 0;
        if (st->filt_child_started) {
            have = nextblock(c, rows, 0);
            if (!have) fail(SO_ERR_RUNTIME, "oracle: padded child ended");
        }
        st->filt_child_started = 1;
        pull_into(c, st->input, rows, nch, s->kids[0]->dtype, have);
        int64_t out_len;
        if (is_rs) {
            out_len = 0;
            for (int ch = 0; ch < nch; ++ch)
                out_len = fir_block(&st->firs[ch], st->input + (int64_t)ch * rows, rows,
                                    st->output + (int64_t)ch * st->cut_n, st->cut_n, 0);
            if (out_len <= 0) fail(SO_ERR_RUNTIME, "Unexpected non-positive output length!");
            for (int ch = 0; ch < nch; ++ch)
                for (int64_t i = 0; i < out_len; ++i) {
                    double* o = st->output + (int64_t)ch * st->cut_n + i;
                    *o = roundto(s->dtype, *o);
                }
        } else {
            out_len = rows;
            int nsec = nd->i0;
            const double* sos = (const double*)nd->p0;
            double g = nd->d0;
            for (int ch = 0; ch < nch; ++ch) { /* DSP.jl filt!(out, DF2TFilter{SOS}, x) */
                double* si = st->sos_state + (size_t)ch * nsec * 2;
                const double* x = st->input + (int64_t)ch * rows;
                double* o = st->output + (int64_t)ch * st->cut_n;
                for (int64_t i = 0; i < rows; ++i) {
                    double yi = x[i];
                    for (int f = 0; f < nsec; ++f) {
                        const double* b = sos + 6 * f;
                        double xi = yi;
                        yi = si[2 * f] + b[0] * xi;
                        si[2 * f] = si[2 * f + 1] + b[1] * xi - b[4] * yi;
                        si[2 * f + 1] = b[2] * xi - b[5] * yi;
                    }
                    o[i] = roundto(s->dtype, yi * g);
                }
            }
        }
        st->len = imin(maxlen, out_len);
        st->last_output_index = 0;
        st->available_output = out_len;
        return 1;
    }
    case SO_NODE_NORMPOWER: { /* src/filters.jl:296-314 */
        int64_t N = len_clean(s->len);
        if (!st->started) {
            if (N >= BIG) fail(SO_ERR_LENGTH, "Cannot normalize an infinite-length signal. Please use `Until` to take a prefix of the signal");
            st->started = 1;
            int nch = s->nch;
            st->vals = (double*)xalloc(sizeof(double) * (size_t)imax(N, 1) * (size_t)nch);
            st->nvals = N;
            if (N > 0) pull_into(st->kids[0], st->vals, N, nch, s->dtype, 0);
            /* rms = sqrt(mean(x -> float(x)^2, vals)); for Float32 Julia reduces in
               Float32 with pairwise summation (blocks of 1024) */
            int64_t cnt = N * nch;
            double rms;
            if (s->dtype == SO_F32) {
                /* iterative pairwise: sum blocks of 1024 sequentially in f32, then fold pairs */
                int64_t nb = (cnt + 1023) / 1024;
                float* part = (float*)xalloc(sizeof(float) * (size_t)imax(nb, 1));
                for (int64_t b = 0; b < nb; ++b) {
                    float acc = 0.f;
                    int64_t e = imin(cnt, (b + 1) * 1024);
                    for (int64_t i = b * 1024; i < e; ++i) {
                        float v = (float)st->vals[i];
                        acc += v * v;
                    }
                    part[b] = acc;
                }
                int64_t m = nb;
                while (m > 1) {
                    int64_t h = (m + 1) / 2;
                    for (int64_t i = 0; i < m / 2; ++i) part[i] = part[2 * i] + part[2 * i + 1];
                    if (m & 1) part[m / 2] = part[m - 1];
                    m = h;
                }
                float mean = (nb ? part[0] : 0.f) / (float)cnt;
                rms = (double)sqrtf(mean);
            } else {
                double acc = 0;
                /* pairwise in blocks of 1024 */
                int64_t nb = (cnt + 1023) / 1024;
                double* part = (double*)xalloc(sizeof(double) * (size_t)imax(nb, 1));
                for (int64_t b = 0; b < nb; ++b) {
                    double a2 = 0;
                    int64_t e = imin(cnt, (b + 1) * 1024);
                    for (int64_t i = b * 1024; i < e; ++i) a2 += st->vals[i] * st->vals[i];
                    part[b] = a2;
                }
                int64_t m = nb;
                while (m > 1) {
                    int64_t h = (m + 1) / 2;
                    for (int64_t i = 0; i < m / 2; ++i) part[i] = part[2 * i] + part[2 * i + 1];
                    if (m & 1) part[m / 2] = part[m - 1];
                    m = h;
                }
                acc = nb ? part[0] : 0;
                rms = sqrt(acc / (double)cnt);
            }
            for (int64_t i = 0; i < cnt; ++i) st->vals[i] = roundto(s->dtype, st->vals[i] / rms);
            st->offset = 0;
            st->len = 0;
        }
        int64_t off = st->offset + st->len;
        int64_t len = imin(maxlen, N - off);
        if (len <= 0) return 0;
        st->offset = off;
        st->len = len;
        return 1;
    }
    }
    fail(SO_ERR_INVALID, "oracle: nextblock on unknown kind");
    return 0;
}

static int frame(OState* st, int64_t i, double* out) {
    OSig* s = st->s;
    const so_node_t* nd = s->nd;
    switch (s->kind) {
    case SO_NODE_ARRAY: { /* view(block.data,i,:) src/arrays.jl:124 */
        int64_t row = st->offset + i - 1;
        if (nd->dtype == SO_F32) {
            const float* d = (const float*)nd->p0;
            for (int ch = 0; ch < s->nch; ++ch) out[ch] = (double)d[row * nd->s0 + ch * nd->s1];
        } else {
            const double* d = (const double*)nd->p0;
            for (int ch = 0; ch < s->nch; ++ch) out[ch] = d[row * nd->s0 + ch * nd->s1];
        }
        return nd->dtype;
    }
    case SO_NODE_CONST: /* src/numbers.jl:64 */
        out[0] = nd->d0;
        return nd->i0;
    case SO_NODE_FUNC: { /* src/functions.jl:53-60 */
        double n = (double)(i + st->offset);
        double v;
        if (nd->i1) { /* ω given */
            double ph = n / nd->fs * nd->d0 + nd->d1;
            if (nd->i0 == SO_FN_SIN) v = sinpi_(2 * ph);
            else {
                double a = 2 * M_PI * fmod(ph, 1.0);
                v = nd->i0 == SO_FN_COS ? cos(a) : a;
            }
        } else {
            double t = n / nd->fs + nd->d1;
            if (nd->i0 == SO_FN_SIN) v = sinpi_(2 * t);
            else v = nd->i0 == SO_FN_COS ? cos(t) : t;
        }
        out[0] = v;
        return SO_F64;
    }
    case SO_NODE_UNTIL:
    case SO_NODE_AFTER: /* src/cutting.jl:218-219 */
        return frame(st->kids[0], i, out);
    case SO_NODE_PAD: { /* src/padding.jl:208-213 */
        if (st->pad_mode == 0) return frame(st->kids[0], i, out);
        if (s->pad_kind == SO_PAD_CYCLE || s->pad_kind == SO_PAD_MIRROR) {
            const so_node_t* a = s->kids[0]->nd;
            int64_t N = a->l0;
            int64_t g = i + st->offset; /* 1-based global */
            int64_t row;
            if (s->pad_kind == SO_PAD_CYCLE) row = (g - 1) % N; /* src/padding.jl:132 */
            else { /* src/padding.jl:142-148 */
                int64_t cnt = (g - 1) / N, rem = (g - 1) % N;
                row = (cnt % 2 == 0) ? rem : N - rem - 1;
            }
            for (int ch = 0; ch < s->nch; ++ch)
                out[ch] = a->dtype == SO_F32 ? (double)((const float*)a->p0)[row * a->s0 + ch * a->s1]
                                             : ((const double*)a->p0)[row * a->s0 + ch * a->s1];
            return a->dtype;
        }
        for (int ch = 0; ch < s->nch; ++ch) out[ch] = st->padvals[ch];
        return st->pad_dtype;
    }
    case SO_NODE_APPEND: return frame(st->kids[st->k], i, out); /* src/appending.jl:89-90 */
    case SO_NODE_RAMP: { /* src/ramps.jl:56-72 */
        double v;
        int dt;
        if (!st->ramp_active) {
            v = 1.0;
            dt = float_of(s->kids[0]->dtype);
        } else if (nd->i0 == 0) {
            v = ramp_fn(nd->i1, (double)(i + st->offset - 1) / (double)st->marker);
            dt = SO_F64;
        } else {
            int64_t startramp = st->marker - st->offset;
            int64_t stop = st->stop - st->offset;
            v = ramp_fn(nd->i1, 1.0 - (double)(i - startramp) / (double)(stop - startramp));
            dt = SO_F64;
        }
        for (int ch = 0; ch < s->nch; ++ch) out[ch] = v;
        return dt;
    }
    case SO_NODE_MAP: { /* src/mapsignal.jl:249-272 */
        int N = s->nkids;
        int maxch = (int)st->cut_n;
        for (int j = 0; j < N; ++j)
            st->kdt[j] = frame(st->kids[j], i + st->offsets[j], st->scratch + (size_t)j * maxch);
        int fn = nd->i0;
        switch (fn) {
        case SO_MAP_ADD:
        case SO_MAP_MUL:
        case SO_MAP_SUB:
        case SO_MAP_DIV: {
            int dt = st->kdt[0];
            for (int ch = 0; ch < s->nch; ++ch) {
                double acc = st->scratch[ch];
                int t = st->kdt[0];
                if (N == 1) {
                    if (fn == SO_MAP_SUB) acc = -acc;
                }
                for (int j = 1; j < N; ++j) {
                    double b = st->scratch[(size_t)j * maxch + ch];
                    int tb = st->kdt[j];
                    int tr = promote(t, tb);
                    switch (fn) {
                    case SO_MAP_ADD: acc = acc + b; break;
                    case SO_MAP_MUL: acc = acc * b; break;
                    case SO_MAP_SUB: acc = acc - b; break;
                    default:
                        if (tr == SO_I64) tr = SO_F64;
                        acc = acc / b;
                    }
                    acc = roundto(tr, acc);
                    t = tr;
                }
                out[ch] = acc;
                dt = t;
            }
            return dt;
        }
        case SO_MAP_TUPLECAT: { /* src/mapsignal.jl:361-362 */
            int o = 0, dt = st->kdt[0];
            for (int j = 0; j < N; ++j) {
                for (int ch = 0; ch < s->kids[j]->nch; ++ch) out[o++] = st->scratch[(size_t)j * maxch + ch];
                dt = promote(dt, st->kdt[j]);
            }
            return dt;
        }
        case SO_MAP_GETCHAN: out[0] = st->scratch[nd->i3 - 1]; return st->kdt[0];
        case SO_MAP_AS1CHANNEL: { /* sum(x) src/reformatting.jl:156 */
            double acc = st->scratch[0];
            for (int ch = 1; ch < s->kids[0]->nch; ++ch) acc = roundto(st->kdt[0], acc + st->scratch[ch]);
            out[0] = acc;
            return st->kdt[0];
        }
        case SO_MAP_ASNCHANNELS:
            for (int ch = 0; ch < s->nch; ++ch) out[ch] = st->scratch[0];
            return st->kdt[0];
        case SO_MAP_TOELTYPE:
            for (int ch = 0; ch < s->nch; ++ch) out[ch] = roundto(nd->i3, st->scratch[ch]);
            return nd->i3;
        case SO_MAP_REVERSECH:
            for (int ch = 0; ch < s->nch; ++ch) out[ch] = st->scratch[s->nch - 1 - ch];
            return st->kdt[0];
        }
        fail(SO_ERR_INVALID, "oracle: unknown map fn %d", fn);
        return 0;
    }
    case SO_NODE_FILT_SOS:
    case SO_NODE_RESAMPLE: { /* view(x.output,i+x.last_output_index,:) src/filters.jl:213-214 */
        int64_t row = i + st->last_output_index - 1;
        for (int ch = 0; ch < s->nch; ++ch) out[ch] = st->output[(int64_t)ch * st->cut_n + row];
        return s->dtype;
    }
    case SO_NODE_NORMPOWER: {
        int64_t row = st->offset + i - 1;
        for (int ch = 0; ch < s->nch; ++ch) out[ch] = st->vals[row + (int64_t)ch * st->nvals];
        return s->dtype;
    }
    }
    fail(SO_ERR_INVALID, "oracle: frame on unknown kind");
    return 0;
}

/* FilteredSignal pulls from Pad(x.signal,zero) (src/filters.jl:240): splice the PAD in */
static void splice_filter_pads(OSig* s) {
    for (int j = 0; j < s->nkids; ++j) splice_filter_pads(s->kids[j]);
    if (s->kind == SO_NODE_FILT_SOS || s->kind == SO_NODE_RESAMPLE)
        s->kids[0] = make_pad(s->kids[0], SO_PAD_ZERO, 0, 0.0, NULL);
}

/* ------------------------------------------------------------------------- */
/* public entry points                                                        */

/* nframes(root) by the reference's length algebra: >=0, SO_LEN_INF; <-9 = error */
int64_t so_oracle_nframes(const so_node_t* nodes, int32_t n_nodes, int32_t root) {
    g_arena = NULL;
    if (setjmp(g_jmp)) {
        arena_free();
        return g_status - 100;
    }
    OSig* s = build(nodes, n_nodes, root);
    int64_t n = len_isinf(s->len) ? SO_LEN_INF : s->len.n;
    arena_free();
    return n;
}

/*
 * sink!(result,x) src/sink.jl:158-168,225-241.  `out` is host memory described by
 * `desc` (dtype f32/f64, strides in elements).  blocksize_override>0 replaces every
 * Filt/ToFramerate blocksize (for blocksize-invariance tests, runtests.jl:353-356).
 */
int32_t so_oracle_sink(const so_node_t* nodes, int32_t n_nodes, int32_t root,
                       const so_out_desc_t* desc, void* out, int32_t blocksize_override) {
    g_arena = NULL;
    g_err[0] = 0;
    if (setjmp(g_jmp)) {
        arena_free();
        return g_status;
    }
    if (g_position_mode >= 0) g_phase_accumulate = g_position_mode == 0;
    else {
        const char* pe = getenv("SO_ORACLE_EXACT_POSITIONS");
        g_phase_accumulate = !(pe && pe[0] == '1');
    }
    g_blocksize_override = blocksize_override;
    OSig* s = build(nodes, n_nodes, root);
    /* process_sink_params src/sink.jl:94-99 is the caller's check for sink();
       sink! itself only checks the buffer length :161-163 */
    int64_t N = len_clean(s->len);
    if (N < desc->nframes)
        fail(SO_ERR_LENGTH, "Signal is too short to fill buffer of length %lld.", (long long)desc->nframes);
    if (s->nch != desc->nch)
        fail(SO_ERR_CHANNELS, "oracle: host must apply ToChannels (src/sink.jl:164): signal has %d channels, buffer %d", s->nch, desc->nch);
    if (desc->dtype != SO_F32 && desc->dtype != SO_F64) fail(SO_ERR_UNSUPPORTED, "oracle: result eltype must be Float32/Float64");
    splice_filter_pads(s);
    OState* st = mkstate(s);
    int64_t rows = desc->nframes;
    int nch = desc->nch;
    int64_t written = 0;
    double* fr = (double*)xalloc(sizeof(double) * (size_t)(nch + 1));
    int ok = nextblock(st, rows, 0); /* src/sink.jl:225-226 */
    while (ok && written < rows) {
        if (st->len <= 0) fail(SO_ERR_RUNTIME, "oracle: @assert nframes(block) > 0");
        for (int64_t i = 1; i <= st->len; ++i) {
            frame(st, i, fr);
            int64_t r = written + i - 1;
            if (desc->dtype == SO_F32) {
                float* o = (float*)out;
                for (int ch = 0; ch < nch; ++ch) o[r * desc->frame_stride + ch * desc->chan_stride] = (float)fr[ch];
            } else {
                double* o = (double*)out;
                for (int ch = 0; ch < nch; ++ch) o[r * desc->frame_stride + ch * desc->chan_stride] = fr[ch];
            }
        }
        written += st->len;
        int64_t ml = rows - written;
        if (ml > 0) ok = nextblock(st, ml, 0);
    }
    if (written != rows) fail(SO_ERR_RUNTIME, "oracle: @assert written == nframes(result) (%lld != %lld)", (long long)written, (long long)rows);
    arena_free();
    return SO_OK;
}
