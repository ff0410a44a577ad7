// Stage setup: geometry and tables of the stateful stages (resampler variants, SOS IIR, Normpower),
// carrier fusion of pointwise sources into the periodic resampler, and the optional fusion of the
// IIR's state pass into the resampler in front of it.
#include "plan_impl.h"

namespace so {

static int env_int(const char* name, int dflt) {  // tuning knobs
    const char* ev = std::getenv(name);
    return ev ? std::atoi(ev) : dflt;
}

// ---------------------------------------------------------------------------
// small dense matrices for the SOS state propagation
Mat matmul(const Mat& a, const Mat& b, int D) {
    Mat c((size_t)D * D, 0.0);
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) {
            double v = a[(size_t)i * D + k];
            if (v == 0.0) continue;
            for (int j = 0; j < D; ++j) c[(size_t)i * D + j] += v * b[(size_t)k * D + j];
        }
    return c;
}
double maxabs(const Mat& a) {
    double m = 0;
    for (double v : a) {
        if (!std::isfinite(v)) return INFINITY;  // (an unstable cascade: powers overflow, then turn NaN)
        m = std::max(m, std::fabs(v));
    }
    return m;
}
Mat ident(int D) {
    Mat m((size_t)D * D, 0.0);
    for (int i = 0; i < D; ++i) m[(size_t)i * D + i] = 1.0;
    return m;
}
// one zero-input DF2T step applied to each unit state: columns of the state matrix A
Mat sos_state_matrix(const SosCoefs& cf) {
    int ns = cf.nsec, D = 2 * ns;
    Mat A((size_t)D * D, 0.0);
    for (int col = 0; col < D; ++col) {
        std::vector<double> s(D, 0.0);
        s[col] = 1.0;
        double y = 0.0;
        for (int f = 0; f < ns; ++f) {
            double xi = y;
            y = s[2 * f] + cf.b0[f] * xi;
            s[2 * f] = s[2 * f + 1] + cf.b1[f] * xi - cf.a1[f] * y;
            s[2 * f + 1] = cf.b2[f] * xi - cf.a2[f] * y;
        }
        for (int r = 0; r < D; ++r) A[(size_t)r * D + col] = s[r];
    }
    return A;
}
Mat matpow(Mat A, int64_t e, int D) {
    Mat R = ident(D);
    while (e > 0) {
        if (e & 1) R = matmul(R, A, D);
        e >>= 1;
        if (e) A = matmul(A, A, D);
    }
    return R;
}


// How far apart are two evaluation orders of this cascade?  The chunked scan and the fused
// multiply-adds of the device round differently from DSP.jl's sequential `filt!` (reference
// src/filters.jl:252-255); a well-conditioned cascade amplifies that to ~1e-12, but band-stops of
// order 7-12 whose sections have gains of 70 ... 1e-8 amplify rounding by 1e7 and more (round-2 soak:
// 1e-7 ... 1e-3 between engine and oracle on four of 520 designs).  Measured here directly: the same
// DF2T recurrence over a pseudo-random probe in Float64 and in the host's 80-bit long double, norm-wise
// relative difference.  Above kSosExactTol the stage runs the reference's own order of operations
// (one sequence per channel, no fused multiply-adds: k_sos_tiled<..., EXACT>), which is slow and exact.
constexpr double kSosExactTol = 1e-9;
static double sos_rounding_sensitivity(const double* sos, int nsec, double gain) {
    static std::mutex mu;
    static std::map<std::vector<double>, double> cache;
    std::vector<double> key(sos, sos + 6 * (size_t)nsec);
    key.push_back(gain);
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
        double dv;
        if (disk_value_get("sens", key, dv)) return cache[key] = dv;
    }
    std::vector<double> s(2 * (size_t)nsec, 0.0);
    std::vector<long double> sl(2 * (size_t)nsec, 0.0L);
    uint64_t st = 12345;
    long double num = 0, den = 0;
    for (int i = 0; i < 4096; ++i) {
        st = st * 6364136223846793005ULL + 1442695040888963407ULL;
        const double x = (double)((st >> 11) & 0xFFFFFFFFFFFFFULL) / 4503599627370496.0 - 0.5;
        volatile double y = x;  // (volatile: every operation rounded to Float64, no contraction)
        long double yl = x;
        for (int f = 0; f < nsec; ++f) {
            const double* b = sos + 6 * (size_t)f;
            const double xi = y;
            volatile double t0 = b[0] * xi;
            y = s[2 * f] + t0;
            volatile double t1 = b[1] * xi, t2 = b[4] * y, t3 = s[2 * f + 1] + t1;
            s[2 * f] = t3 - t2;
            volatile double t4 = b[2] * xi, t5 = b[5] * y;
            s[2 * f + 1] = t4 - t5;
            const long double xl = yl;
            yl = sl[2 * f] + (long double)b[0] * xl;
            sl[2 * f] = sl[2 * f + 1] + (long double)b[1] * xl - (long double)b[4] * yl;
            sl[2 * f + 1] = (long double)b[2] * xl - (long double)b[5] * yl;
        }
        const long double d = (long double)(y * gain) - yl * gain;
        num += d * d;
        den += yl * gain * yl * gain;
    }
    double r = den > 0 ? (double)std::sqrt((double)(num / den)) : 0.0;
    if (!std::isfinite(r)) r = 0.0;  // (an unstable design: nothing to protect)
    disk_value_put("sens", key, r);
    std::lock_guard<std::mutex> lk(mu);
    cache[key] = r;
    return r;
}

// ... and how far apart are the CHUNKED evaluation and the sequential one?  The probe above sees what a different
// rounding of the same recurrence does; it does not see the state hand-over of the time-parallel form, s0[k+1] =
// M s0[k] + v[k] with M = A^L by repeated squaring: for cascades with clustered poles (a band-stop whose upper edge is
// at Nyquist: ten poles next to -1) the powers of the nearly defective A lose digits that the recurrence itself does
// not -- 4e-5 between engine and oracle with a rounding sensitivity of 1e-8 (tools/soak_degenerate_filters.py).
// Measured the same way: the cascade over a pseudo-random probe sequentially and in chunks of the planned length
// (zero-state end states, Horner scan with the same M, outputs from the scanned states), both in Float64 with every
// operation rounded on its own; norm-wise relative difference.  Cached per coefficient set and length.
static double sos_chunk_sensitivity(const std::vector<SosCoefs>& groups, int64_t L, int64_t need) {
    static std::mutex mu;
    static std::map<std::vector<double>, double> cache;
    std::vector<double> key;
    for (auto& cf : groups) {
        for (int f = 0; f < cf.nsec; ++f) {
            key.push_back(cf.b0[f]); key.push_back(cf.b1[f]); key.push_back(cf.b2[f]);
            key.push_back(cf.a1[f]); key.push_back(cf.a2[f]);
        }
        key.push_back(cf.gain);
    }
    key.push_back((double)L);
    const int64_t nchunks = std::min<int64_t>(std::max<int64_t>(8, std::min<int64_t>(48, 65536 / L)), (need + L - 1) / L);
    const int64_t n = std::min<int64_t>(need, nchunks * L);
    key.push_back((double)n);  // (the probe's length: a short signal is probed over its own length)
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
        double dv;
        if (disk_value_get("chunk", key, dv)) return cache[key] = dv;
    }
    double worst = 0.0;
    if (nchunks >= 2) {
        std::vector<double> x((size_t)n), yseq((size_t)n), ychk((size_t)n);
        uint64_t st = 2463534242ULL;
        for (auto& v : x) {
            st = st * 6364136223846793005ULL + 1442695040888963407ULL;
            v = (double)((st >> 11) & 0xFFFFFFFFFFFFFULL) / 4503599627370496.0 - 0.5;
        }
        for (auto& cf : groups) {  // (groups of a long cascade filter one after the other, each chunked on its own)
            const int ns = cf.nsec, D = 2 * ns;
            // (host code is built for baseline x86-64: no fused multiply-add to contract into, every operation rounds)
            auto step = [&](double xin, std::vector<double>& sv) {
                double y = xin;
                double* sp = sv.data();
                for (int f = 0; f < ns; ++f) {
                    const double xi = y;
                    y = sp[2 * f] + cf.b0[f] * xi;
                    sp[2 * f] = (sp[2 * f + 1] + cf.b1[f] * xi) - cf.a1[f] * y;
                    sp[2 * f + 1] = cf.b2[f] * xi - cf.a2[f] * y;
                }
                return y;
            };
            std::vector<double> sv((size_t)D, 0.0);
            for (int64_t i = 0; i < n; ++i) yseq[(size_t)i] = step(x[(size_t)i], sv) * cf.gain;
            const Mat M = matpow(sos_state_matrix(cf), L, D);
            std::vector<double> s0((size_t)D, 0.0), v((size_t)D), t((size_t)D);
            for (int64_t k = 0; k * L < n; ++k) {
                const int64_t a = k * L, b = std::min<int64_t>(n, a + L);
                std::vector<double> run = s0;  // outputs from the scanned state
                for (int64_t i = a; i < b; ++i) ychk[(size_t)i] = step(x[(size_t)i], run) * cf.gain;
                std::fill(v.begin(), v.end(), 0.0);  // end state from rest
                for (int64_t i = a; i < b; ++i) (void)step(x[(size_t)i], v);
                for (int r = 0; r < D; ++r) {  // s0 <- M s0 + v (the device's Horner step)
                    double acc = v[(size_t)r];
                    for (int d = 0; d < D; ++d) acc = std::fma(M[(size_t)r * D + d], s0[(size_t)d], acc);
                    t[(size_t)r] = acc;
                }
                s0 = t;
            }
            long double num = 0, den = 0;
            for (int64_t i = 0; i < n; ++i) {
                const long double d = (long double)ychk[(size_t)i] - yseq[(size_t)i];
                num += d * d;
                den += (long double)yseq[(size_t)i] * yseq[(size_t)i];
            }
            double r = den > 0 ? (double)std::sqrt((double)(num / den)) : 0.0;
            if (!std::isfinite(r)) r = 0.0;  // (an unstable design: nothing to protect)
            worst = std::max(worst, r);
            x = yseq;  // the next group filters this group's output
        }
    }
    disk_value_put("chunk", key, worst);
    std::lock_guard<std::mutex> lk(mu);
    cache[key] = worst;
    return worst;
}

// Can this periodic resampler stage run the GA instantiation (Float32 tiles, Float64 gain at the A
// operand)?  Geometry the instantiations cover, and the LDS budget with three gain arrays.
static bool ga_fits(const Stage& S, int stage_dtype) {
    if (!S.periodic || S.rp.rows != 32 || stage_dtype != SO_F64 || std::getenv("SIGOPS_RS_NOGA")) return false;
    const RsPeriodic& rp = S.rp;
    const int gper = (rp.ngroups + rp.ncompute - 1) / std::max(1, rp.ncompute);
    if (rp.kw != 56 || gper != 1 || !(rp.ct == 8 || rp.ct == 4)) return false;
    const size_t avail = 160 * 1024 - sizeof(RsCtl) - 64;
    const size_t pitch4 = (size_t)((rp.tile_len + 31 + 8 + 3) / 4 * 4);
    const size_t tile_bytes = (size_t)rp.ct * pitch4 * 4;
    const size_t fpitch = (size_t)((rp.tile_len + 32 + 1) & ~1);  // (Float32 tiles start up to 31 frames below their first input)
    const bool ok = 3 * fpitch * 8 + kRsTwoDoubles * 8 + 2 * tile_bytes <= avail;
    if (std::getenv("SIGOPS_DEBUG_PLAN"))
        std::fprintf(stderr, "[sigops] GA geometry: kw=%d gper=%d ct=%d tile_len=%d -> %s\n", rp.kw, gper, rp.ct, rp.tile_len, ok ? "fits" : "no");
    return ok;
}

// Sequences (chunks x channels) a three-pass cascade is cut into when the signal is long enough: 512 workgroups of 256
// rows, two per CU, all resident at once (three fit) and finishing together.  The former 262 144 were 1 004 workgroups,
// 3.9 per CU: a CU runs three at a time, so a quarter of them waited for a slot and ran on a draining machine.  Sweeps
// with SIGOPS_SOS_CHUNK: the headline's K2 1.10 -> 1.01 ms (L 896 -> 1760), config 5's 1.94 -> 1.67, config 4's batch
// 1.76 -> 1.70; a length that gives 8.1 waves per CU instead of 8.0 (L 1728) costs 20 %, 768 workgroups (three per CU,
// L 1184) do as well as 512, fewer than 2 per CU lose again (L 2688: 1.32 ms).
constexpr int64_t kSosSequences = 131072;

// Chunk geometry of a three-pass cascade: L frames per chunk, W warm-up frames of the state pass, K terms of
// the scan, and the buffers they need.  `target`: chunks per channel that fill the machine -- kSosSequences / channels
// for a filter launched on its own, fewer and longer for the members of a batch (batch_sos_stages), which fill
// it together.  `groups` may be the stage's own vector (re-chunking).
void Plan::sos_chunking(int sid, int64_t need, int nch, int dtype, const std::vector<SosCoefs>& groups_in, bool exact, int64_t target) {
    const std::vector<SosCoefs> groups = groups_in;
    const int64_t BIG = (int64_t)1 << 60;
    auto sized_buf = [&](int have, size_t bytes) {
        if (have < 0) return raw_buf(bytes);
        bufs[have].bytes = bytes;
        return have;
    };
    // chunking: enough independent sequences to fill 256 CUs x 4 SIMDs x 4 waves
    SosGeom g{};
    g.n = need;
    g.nch = nch;
    const double tol = std::ldexp(1.0, -70);
    // chunk length: as many sequences (chunks x channels) as the machine can hold; every
    // pass is latency-bound per sequence, so shorter chunks win down to L = 64 (sweep on
    // config 2: L=64 0.205 ms, 128 0.208, 256 0.293, 512 0.531)
    int64_t nchunks = std::max<int64_t>(1, std::min<int64_t>(target, need / 64));
    int64_t L = (need + nchunks - 1) / nchunks;
    if (const char* ev = std::getenv("SIGOPS_SOS_CHUNK")) {  // tuning knob
        L = std::max(32, std::atoi(ev));
    }
    L = (L + 31) / 32 * 32;
    if (exact) L = std::max<int64_t>(need, 32);  // one sequence per channel, start to end
    std::vector<std::vector<double>> mp;
    int K = 1;
    int64_t W = BIG;
    for (;;) {
        nchunks = (need + L - 1) / L;
        mp.clear();
        K = 1;
        W = 0;
        if (nchunks <= 1) break;
        bool ok = true;
        for (auto& cf : groups) {
            int D = 2 * cf.nsec;
            Mat A = sos_state_matrix(cf);
            // W: first power of two with ||A^W|| < tol (pass-1 warm-up length)
            Mat P = A;
            int64_t w = 1;
            while (maxabs(P) >= tol && w < ((int64_t)1 << 40)) {
                P = matmul(P, P, D);
                w <<= 1;
            }
            W = std::max(W, w);
            Mat M = matpow(A, L, D);
            std::vector<double> pw_((size_t)D * D, 0.0);
            Mat cur = ident(D);
            std::vector<double> all;
            int k = 0;
            for (;;) {
                all.insert(all.end(), cur.begin(), cur.end());
                ++k;
                cur = matmul(cur, M, D);
                if (maxabs(cur) < tol) break;
                if (k >= 64) {
                    ok = false;
                    break;
                }
            }
            if (!ok) break;
            if (k < 2) {  // (the scan kernel reads M itself, entry 1, whatever K is)
                all.insert(all.end(), cur.begin(), cur.end());
                k = 2;
            }
            K = std::max(K, k);
            mp.push_back(all);
        }
        if (ok) break;
        L *= 2;  // slower-decaying filter: fewer, longer chunks
    }
    // Short signals are filtered without either cut (every earlier chunk enters the scan, the state
    // pass runs over whole chunks): what has decayed below 2^-70 of an earlier peak -- the tail of a
    // filter long after its input went silent -- then keeps the relative accuracy of the sequential
    // recurrence, which a `Normpower` of such a tail makes visible (tools/tree_soak.py 1396/7).
    if (nchunks > 1 && nchunks <= 64) {
        K = std::max<int>(K, (int)nchunks);
        W = std::max(W, L);
    }
    // ... and a filter that feeds a Normpower keeps it at every length: exact block scan
    // (kernels2.hip launch_sos_xscan) and a state pass over whole chunks
    const bool xscan = stages[sid].under_norm && nchunks > 64 && !stages[sid].onepass && !std::getenv("SIGOPS_SOS_NOXSCAN");
    if (xscan) W = std::max(W, L);
    // every group is scanned with the same K (pad shorter tables with zeros)
    for (size_t gi = 0; gi < mp.size(); ++gi) {
        int D = 2 * groups[gi].nsec;
        mp[gi].resize((size_t)K * D * D, 0.0);
    }
    g.chunk = L;
    g.nchunks = (int)nchunks;
    g.warm = W;
    g.kterms = K;
    g.in_dtype = g.out_dtype = dtype;
    g.exact = exact ? 1 : 0;
    stages[sid].groups = groups;
    stages[sid].mpow_host = mp;
    if (nchunks > 1 && !stages[sid].onepass) {
        size_t msz = 0;
        for (auto& v : mp) msz = std::max(msz, v.size());
        stages[sid].mpow_buf = sized_buf(stages[sid].mpow_buf, msz * 8 * groups.size());
        stages[sid].v_buf = sized_buf(stages[sid].v_buf, (size_t)nchunks * nch * 2 * kMaxSec * 8);
        stages[sid].s0_buf = sized_buf(stages[sid].s0_buf, (size_t)nchunks * nch * 2 * kMaxSec * 8);
        if (xscan) {
            stages[sid].xscan = true;
            stages[sid].xs_mats_host.clear();
            for (auto& cf : groups) {
                const int D = 2 * cf.nsec;
                Mat M = matpow(sos_state_matrix(cf), L, D), MB = matpow(M, kXsBlock, D);
                std::vector<double> both(M);
                both.insert(both.end(), MB.begin(), MB.end());
                stages[sid].xs_mats_host.push_back(both);
            }
            stages[sid].xs_mats_buf = sized_buf(stages[sid].xs_mats_buf, (size_t)groups.size() * 2 * 16 * 16 * 8);
            const int64_t nblk = (nchunks + kXsBlock - 1) / kXsBlock;
            stages[sid].sblk_buf = sized_buf(stages[sid].sblk_buf, (size_t)nblk * nch * 16 * 8);
        }
    }
    if (nchunks > 1 && !stages[sid].onepass && !exact) stages[sid].bad_buf = sized_buf(stages[sid].bad_buf, (size_t)nch * 4);
    stages[sid].sg = g;
}

void Plan::process_stage(int sid) {
    // NOTE: `stages` may grow while lowering the child; re-take references after.
    int ni = stages[sid].node;
    Node& N = nodes[ni];
    const so_node_t& nd = N.nd;
    int child = N.kids[0];
    // `Filt(Filt(x))` in Float64 (a band-pass written as low-pass |> high-pass): ONE cascade of all the sections -- the inner
    // filter's first, the gains multiplied -- as long as it fits one launch group: one run of the filter kernels instead of
    // two, no intermediate signal in HBM (reference src/filters.jl:240-255 filters the inner filter's blocks in place: the
    // same sections in the same order, rounded as one cascade instead of two).  Frames are one to one, so every bound below
    // holds for the innermost child; each skipped filter still pulls whole blocks of ITS child (error parity).
    std::vector<double> merged_sos;
    double merged_gain = nd.d0;
    std::vector<int64_t> merged_bs;
    if (stages[sid].kind == ST_SOS && N.dtype == SO_F64 && !std::getenv("SIGOPS_SOS_NOMERGE")) {
        int tot = nd.i0;
        while (nodes[child].nd.kind == SO_NODE_FILT_SOS && nodes[child].dtype == SO_F64 && nodes[child].nch == N.nch &&
               nodes[child].fs == N.fs && tot + nodes[child].nd.i0 <= kMaxSec && nodes[child].nd.i0 >= 1) {
            const so_node_t& cn = nodes[child].nd;
            if (merged_sos.empty()) merged_sos.assign((const double*)nd.p0, (const double*)nd.p0 + 6 * (size_t)nd.i0);
            merged_sos.insert(merged_sos.begin(), (const double*)cn.p0, (const double*)cn.p0 + 6 * (size_t)cn.i0);
            merged_gain *= cn.d0;
            merged_bs.push_back(std::max(1, cn.i1));
            tot += cn.i0;
            child = nodes[child].kids[0];
        }
    }
    Node& C = nodes[child];
    int64_t need = stages[sid].need;
    stages[sid].processed = true;
    if (need <= 0) return;

    int64_t in_frames = need;
    if (stages[sid].kind == ST_RESAMPLE) {
        RsGeom g{};
        g.arbitrary = nd.i0 == SO_RS_ARBITRARY;
        int hlen = nd.i2;
        g.nphi = g.arbitrary ? nd.i1 : (int)nd.l0;
        if (g.nphi < 1) fail(SO_ERR_INVALID, "resampler: bad phase count");
        g.L = nd.l0;
        g.M = nd.l1;
        const bool plain_fir = nd.i0 == SO_RS_FIR;  // Filt(x,h): ratio 1, causal, no delay compensation
        if (plain_fir) {
            g.nphi = 1;
            g.L = g.M = 1;
        }
        if (!g.arbitrary && (g.L < 1 || g.M < 1)) fail(SO_ERR_INVALID, "resampler: bad ratio");
        g.delta = g.arbitrary ? (double)g.nphi / nd.d0 : 0.0;
        g.c0 = plain_fir ? 0.0 : (double)(hlen - 1) / 2.0;
        g.c0i = plain_fir ? 0 : (hlen - 1) / 2;
        g.taps = (hlen + g.nphi - 1) / g.nphi;
        g.nch = N.nch;
        g.m0 = 0;
        g.n_out = need;
        if (g.arbitrary) rs_detect_exact(g, nd.fs, C.fs, nd.d0);
        // ---- warm start (see the IIR's below): the resampler is an FIR filter, so outputs from a
        //      whole number of periods before the first frame anybody reads on are the same whether
        //      the stage starts there or at frame 0, except the first few (their taps reach before
        //      the first input staged), which nobody reads either ----
        int64_t rbase = 0;
        if ((!g.arbitrary || g.exact) && stages[sid].lo >= 8192 && stages[sid].lo < need &&
            !std::getenv("SIGOPS_NO_WARM_START")) {
            const int64_t margin = (g.taps + 2 + g.M - 1) / g.M + 1;  // periods
            int64_t k = stages[sid].lo / g.L - margin;
            k = k / 16 * 16;  // (16 M inputs: the first staged input stays 128-byte aligned)
            if (k > 0 && k * g.L >= 4096) {
                rbase = k * g.L;
                stages[sid].base = rbase;
                stages[sid].in_base = k * g.M;
                need -= rbase;
                g.n_out = need;
            }
        }
        // ... and a rate without a period (no integer frame rates) has nothing to be aligned with: output m sits at
        // q_m = c0 + m delta whatever came before it (the accumulator's deviations from that are listed from the
        // nearest checkpoint of an earlier replay, accumulator.cpp), so the stage starts at the first frame anybody
        // reads -- g.m0 -- and stages its input from taps + 2 frames before that output's newest input -- g.j0
        if (g.arbitrary && !g.exact && stages[sid].lo >= 8192 && stages[sid].lo < need && !std::getenv("SIGOPS_NO_WARM_START")) {
            rbase = stages[sid].lo / 64 * 64;  // (whole lines for the kernels' stores)
            const double q = g.c0 + (double)rbase * g.delta;
            const int64_t jf = (int64_t)std::floor(q) / g.nphi;
            const int64_t ib = std::max<int64_t>(0, jf - g.taps - 2) / 16 * 16;
            stages[sid].base = rbase;
            stages[sid].in_base = ib;
            g.m0 = rbase;
            g.j0 = ib;
            need -= rbase;
            g.n_out = need;
        }
        // newest input of the last needed output
        int64_t jl;
        if (g.arbitrary && g.exact) {
            int64_t Nn = (need - 1) * ((int64_t)g.nphi * g.M);
            jl = (g.c0i + Nn / g.L) / g.nphi;
        } else if (g.arbitrary) {
            double q = g.c0 + (double)(g.m0 + need - 1) * g.delta;
            jl = (int64_t)std::floor(q) / g.nphi - g.j0;
        } else jl = (g.c0i + (need - 1) * g.M) / g.L;
        int64_t nin = jl + 2;  // +1 slack: host rounding of q may differ from the device's at ties
        if (!isinf_(C.len)) nin = std::min(nin, C.len.n - stages[sid].in_base);
        g.n_in = nin;
        in_frames = nin;
        // polyphase tables: pfb[p][k] = h[p + nphi*k]; dpfb from dh = [diff(h);0]
        const double* h = (const double*)nd.p0;
        stages[sid].pfb_host.assign((size_t)g.nphi * g.taps, 0.0);
        stages[sid].dpfb_host.assign((size_t)g.nphi * g.taps, 0.0);
        for (int p = 0; p < g.nphi; ++p)
            for (int k = 0; k < g.taps; ++k) {
                int64_t hi = p + (int64_t)g.nphi * k;
                stages[sid].pfb_host[(size_t)p * g.taps + k] = hi < hlen ? h[hi] : 0.0;
                stages[sid].dpfb_host[(size_t)p * g.taps + k] = hi + 1 < hlen ? h[hi + 1] - h[hi] : 0.0;
            }
        stages[sid].pfb_buf = raw_buf(stages[sid].pfb_host.size() * 8);
        stages[sid].dpfb_buf = raw_buf(stages[sid].dpfb_host.size() * 8);
        g.in_dtype = g.out_dtype = N.dtype;
        stages[sid].rg = g;
        // reference positions: DSP.jl's phase accumulator (SIGOPS_RS_EXACT=1 keeps the closed form)
        std::vector<uint8_t> wrap;
        if (g.arbitrary && !std::getenv("SIGOPS_RS_EXACT")) {
            // (period positions can only be baked into the tap tables of the periodic / row-tiled
            //  kernels: short outputs go to the thread-per-output kernel and list every deviation)
            replay_phase_accumulator(g, (const double*)nd.p0, hlen, rbase + need, g.exact && need >= 2048, wrap, stages[sid].fix_host, rbase);
        }
        // position of period output r as the tap tables see it: the closed form, or the
        // accumulator's wrap-around tie (previous input, last phase, alpha = 1) where it is the rule
        auto wrap_at = [&](int64_t r) { return !wrap.empty() && wrap[r % g.L]; };
        // ---- periodic (SGPR-tap) variant for rational rates ---------------------------
        // (32-row tiles first; 16-row tiles -- k_resample_periodic's Q = 1 -- where two 32-row slots do not fit LDS or the
        //  window needs more k-steps than a 32-row instantiation has: long periods, e.g. 44.1 -> 16 kHz)
        for (int rows_try : {32, 16}) {
        if (stages[sid].periodic || std::getenv(rows_try == 16 ? "SIGOPS_RS_NOQ1" : "SIGOPS_RS_NOPERIODIC_")) continue;
        if ((!g.arbitrary || g.exact) && need >= 2048) {
            constexpr int RM = 16;  // outputs per group = N of the 16x16x4 MFMA tile
            const int64_t Lb = g.L, Mb = g.M;
            // (8 channels per tile when possible: per-frame gains of a fused source are evaluated
            //  once per tile row-group, and 8 rows give every loader wave exactly one chunk)
            int ct = 1;
            for (int c : {8, 4, 2})
                if (N.nch % c == 0) {
                    ct = c;
                    break;
                }
            if (const char* ev = std::getenv("SIGOPS_RS_CT")) {  // tuning knob
                int c = std::atoi(ev);
                if ((c == 1 || c == 2 || c == 4 || c == 8) && N.nch % c == 0) ct = c;
            }
            if (rows_try == 16 && ((N.dtype != SO_F64 && N.dtype != SO_F32) || (ct != 8 && ct != 4))) continue;  // (the Q = 1 instantiations)
            const int pt = rows_try / ct;  // tile = 32 or 16 rows (k_resample_periodic's Q)
            // super-period: t periods so that (a) L*t is a multiple of 16 where possible and
            // (b) a tile (pt super-periods) covers ~1100 input frames per channel
            int64_t tmin = 16 / std::__gcd<int64_t>(Lb, 16);
            int64_t t = std::max<int64_t>(1, 1100 / (pt * Mb));
            t = std::max<int64_t>(tmin, t / tmin * tmin);
            if (Lb * t > 4096) t = std::max<int64_t>(1, 4096 / Lb);
            // ... and (c) at most twelve blocks of 16 outputs where tmin allows it: a compute wave keeps ONE block's taps
            // in registers.  Upsampling by small ratios (x 2, x 3, x 4: few inputs per period) used to get super-periods of
            // 30 - 70 blocks from (b), for which no instantiation exists -- the stage fell to the row-tiled kernel at a
            // tenth of the speed (x 2 of 8 channels x 300 s: 11 ms; tools/ratio_probe.py)
            if (!std::getenv("SIGOPS_RS_NOBLOCKCAP")) {
                // (twenty-four -- a compute wave keeps two blocks' taps -- for x 2 and x 3, whose twelve-block super-period is
                //  under a hundred input frames: tiles of 400 frames instead of 800, 0.64 ms instead of 0.39)
                const int64_t nblk = env_int("SIGOPS_RS_BLOCKCAP", (Lb <= 3 && Mb * (192 / Lb) < 100) ? 24 : 12);
                const int64_t tcap = std::max<int64_t>(tmin, (nblk * 16 / Lb) / tmin * tmin);
                t = std::min(t, tcap);
            }
            const int64_t Ls = Lb * t, Ms = Mb * t;
            auto pos = [&](int64_t r, int64_t& j, int& p, double& alpha) {
                int64_t qi;
                if (g.arbitrary) {
                    int64_t Nn = r * ((int64_t)g.nphi * Mb);
                    qi = g.c0i + Nn / Lb;
                    alpha = (double)(Nn % Lb) / (double)Lb;
                } else {
                    qi = g.c0i + r * Mb;
                    alpha = 0.0;
                }
                if (wrap_at(r)) {
                    qi -= 1;
                    alpha = 1.0;
                }
                j = qi / g.nphi;
                p = (int)(qi % g.nphi);
            };
            std::vector<int64_t> jr(Ls);
            std::vector<int> pr(Ls);
            std::vector<double> ar(Ls);
            for (int64_t r = 0; r < Ls; ++r) pos(r, jr[r], pr[r], ar[r]);
            const int ngroups = (int)((Ls + RM - 1) / RM);
            int64_t maxspan = 0;
            std::vector<int> jend(ngroups);
            for (int gi = 0; gi < ngroups; ++gi) {
                int64_t r0 = (int64_t)gi * RM, r1 = std::min<int64_t>(Ls, r0 + RM);
                jend[gi] = (int)jr[r1 - 1];
                maxspan = std::max(maxspan, jr[r1 - 1] - jr[r0]);
            }
            // k-steps: smallest instantiated KS covering taps + span (tab is zero padded);
            // compute waves: one (or two) groups each, taps stay in registers
            int kw = 0, ncomp = 0, gper = 0;
            {
                const int ksneed = (g.taps + (int)maxspan + 3) / 4;
                gper = ngroups <= 12 ? 1 : (ngroups <= 24 ? 2 : 0);
                if (const char* ev = std::getenv("SIGOPS_RS_GPER")) gper = std::atoi(ev);  // tuning knob
                const int ks1[] = {12, 14, 16, 20, 28}, ks2[] = {14}, ks16[] = {28, 36};
                if (rows_try == 16) {
                    if (gper == 1)
                        for (int k : ks16)
                            if (!kw && k >= ksneed) kw = 4 * k;
                } else if (gper == 1) {
                    for (int k : ks1)
                        if (!kw && k >= ksneed) kw = 4 * k;
                } else if (gper == 2 || gper == 3) {
                    for (int k : ks2)
                        if (!kw && k >= ksneed) kw = 4 * k;
                }
                if (kw) ncomp = (ngroups + gper - 1) / gper;
            }
            // first staged input, rounded down to a multiple of 4 frames so that tiles start
            // on a 16-byte boundary (vector loads) whenever pt*M is a multiple of 4
            int jlo = jend[0] - (kw - 1);
            jlo -= ((jlo % 4) + 4) % 4;
            // (+0..3 frames: a row's staged span is a whole number of MFMA k-steps, which the fused IIR
            //  state pass walks from jlo to the end)
            const int64_t tile_len = (pt - 1) * Ms + (jend[ngroups - 1] - jlo + 1 + 3) / 4 * 4;
            // tiles are kept in LDS in the sample type and staged from the 128-byte aligned frame
            // below their first input (+15 / +31 frames); rows are 16-byte multiples for LDS-DMA
            const int64_t esz_t = (int64_t)dsize(N.dtype), vfr = 16 / esz_t;
            int64_t pitch = (tile_len + (128 / esz_t - 1) + 2 * vfr + vfr - 1) / vfr * vfr;
            size_t lds_bytes = ((size_t)ct * pitch * esz_t + 7) / 8 * 8;
            size_t tab_bytes = (size_t)ngroups * kw * RM * 8;
            // LDS ring: as many tile slots as fit in 160 KiB, at most 4 (2 tiles in flight
            // beyond the one being retired), at least 2 (plain double buffering)
            int nslots = (int)std::min<size_t>(4, (160 * 1024 - sizeof(RsCtl) - 64) / std::max<size_t>(1, lds_bytes));
            if (const char* ev = std::getenv("SIGOPS_RS_SLOTS")) nslots = std::min(nslots, std::max(2, std::atoi(ev)));
            if (kw && nslots >= 2 && tab_bytes <= (16u << 20) && tile_len < (1 << 30)) {
                const double* h = (const double*)nd.p0;
                std::vector<double> tab((size_t)ngroups * kw * RM, 0.0);
                for (int gi = 0; gi < ngroups; ++gi) {
                    int64_t r0 = (int64_t)gi * RM, r1 = std::min<int64_t>(Ls, r0 + RM);
                    for (int64_t r = r0; r < r1; ++r)
                        for (int kk = 0; kk < kw; ++kk) {
                            int64_t rel = jend[gi] - (kw - 1) + kk;  // input index of slot kk
                            int64_t age = jr[r] - rel;               // tap age for output r
                            if (age < 0 || age >= g.taps) continue;
                            int64_t hi = pr[r] + (int64_t)g.nphi * age;
                            double hv = hi < hlen ? h[hi] : 0.0;
                            double dv = hi + 1 < hlen ? h[hi + 1] - h[hi] : 0.0;
                            tab[((size_t)gi * kw + kk) * RM + (r - r0)] = hv + ar[r] * dv;
                        }
                }
                RsPeriodic rp{};
                rp.n_in = g.n_in;
                rp.n_out = need;
                rp.L = Ls;
                rp.M = Ms;
                rp.nperiods = (need + Ls - 1) / Ls;
                rp.pt = pt;
                rp.ct = ct;
                rp.rows = rows_try;
                rp.ngroups = ngroups;
                rp.kw = kw;
                rp.tile_len = (int)tile_len;
                rp.lds_pitch = (int)pitch;
                rp.jlo = jlo;
                rp.nch = N.nch;
                rp.ptshift = 0;
                while ((1 << rp.ptshift) < pt) ++rp.ptshift;  // pt is a power of two
                rp.nslots = nslots;
                // persistent kernel: 16 waves per workgroup (8 when a wave owns three groups and
                // needs the registers), one workgroup per CU; the waves that do not compute load
                rp.nwaves = gper >= 3 ? 8 : 16;
                rp.ncompute = ncomp;
                rp.grid = 256;
                if (const char* ev = std::getenv("SIGOPS_RS_NWAVES")) rp.nwaves = std::max(2, std::min(16, std::atoi(ev)));
                if (const char* ev = std::getenv("SIGOPS_RS_GRID")) rp.grid = std::max(1, std::atoi(ev));
                if (const char* ev = std::getenv("SIGOPS_RS_DEBUG")) rp.pad = std::atoi(ev);  // ablation knob
                if (const char* ev = std::getenv("SIGOPS_RS_NLOAD")) rp.nload = std::max(1, std::atoi(ev));  // tuning knob
                // A Float32 signal all the way (the stage's own type: its source tile and its output are Float32): the products
                // on the Float32 MFMA where an instantiation exists (32-row tiles, one group per compute wave, windows of up to
                // 20 k-steps, 4 or 8 channels per tile); SIGOPS_RS_NO_F32MFMA keeps the Float64 products rounded once.
                rp.f32m = (N.dtype == SO_F32 && rows_try == 32 && gper == 1 && kw <= 80 && ct >= 4 && !std::getenv("SIGOPS_RS_NO_F32MFMA")) ? 1 : 0;
                stages[sid].periodic = true;
                stages[sid].per_j = jr;
                stages[sid].per_p = pr;
                stages[sid].per_a = ar;
                stages[sid].jend_last = jend[ngroups - 1];
                stages[sid].rp = rp;
                stages[sid].tab_host = tab;
                stages[sid].jend_host = jend;
                stages[sid].tab_buf = raw_buf(tab.size() * 8);
                stages[sid].jend_buf = raw_buf(jend.size() * 4);
                // what k_rs_fixup needs to recompute the groups a non-finite sample reached output by output
                if (!std::getenv("SIGOPS_RS_NO_FIXUP")) {
                    stages[sid].rs_jrel_host.resize((size_t)Ls);
                    for (int64_t r = 0; r < Ls; ++r) stages[sid].rs_jrel_host[(size_t)r] = (int)(jr[r] - jend[r / RM]);
                    stages[sid].rs_jrel_buf = raw_buf((size_t)Ls * 4);
                    stages[sid].rs_nf_buf = raw_buf((size_t)(4 + 4 * (kRsNfCap + 1)) * 4);
                }
            }
        }
        }
        // ---- row-tiled variant: rational rates the MFMA kernel's geometry does not cover -------
        if (!stages[sid].periodic && (!g.arbitrary || g.exact) && need >= 2048 && g.m0 == 0 &&
            !std::getenv("SIGOPS_RS_NOROWS")) {
            // (periods of fewer than 16 outputs -- decimation by 3, 4, 6 ...: L = 1 -- are walked as super-periods of
            //  whole 16-output blocks: a tile row is a (super-)period, and a row of ONE output left fifteen sixteenths of
            //  every MFMA tile and most of the staged window unused: 192 -> 48 kHz of 8 channels x 120 s took 12.3 ms,
            //  0.15 TB/s; tools/rate_matrix.py)
            const int64_t L1 = g.L, M1 = g.M;
            const int64_t tsup = (L1 < 16 && !std::getenv("SIGOPS_RR_NOSUPER")) ? 16 / std::__gcd<int64_t>(L1, 16) * env_int("SIGOPS_RR_SUPERK", 2) : 1;  // (32 outputs per row: 20-38 % over 16, 64 no better)
            const int64_t Lb = L1 * tsup, Mb = M1 * tsup;
            const double* h = (const double*)nd.p0;
            std::vector<int> jr(Lb);
            std::vector<double> ctab((size_t)Lb * g.taps, 0.0);
            int64_t jmin = INT64_MAX, jmax = INT64_MIN;
            for (int64_t r = 0; r < Lb; ++r) {
                int64_t qi;
                double alpha = 0.0;
                if (g.arbitrary) {
                    const int64_t Nn = r * ((int64_t)g.nphi * M1);
                    qi = g.c0i + Nn / L1;
                    alpha = (double)(Nn % L1) / (double)L1;
                } else qi = g.c0i + r * M1;
                if (wrap_at(r)) {
                    qi -= 1;
                    alpha = 1.0;
                }
                const int64_t j = qi / g.nphi;
                const int p = (int)(qi % g.nphi);
                jr[r] = (int)j;
                jmin = std::min(jmin, j - (g.taps - 1));
                jmax = std::max(jmax, j);
                for (int k = 0; k < g.taps; ++k) {
                    const int64_t hi = p + (int64_t)g.nphi * k;
                    const double hv = hi < hlen ? h[hi] : 0.0;
                    const double dv = hi + 1 < hlen ? h[hi + 1] - h[hi] : 0.0;
                    ctab[(size_t)r * g.taps + k] = hv + alpha * dv;
                }
            }
            const int64_t esz_t = (int64_t)dsize(N.dtype);
            int best_ct = 0, best_pb = 0;
            int64_t best_pitch = 0, best_len = 0;
            size_t rr_max_lds = 150 * 1024;
            if (const char* ev = std::getenv("SIGOPS_RR_MAXLDS")) rr_max_lds = (size_t)std::atoi(ev) * 1024;  // tuning knob
            // tile choice: the largest row count whose tile fits; two workgroups per CU (tiles of at
            // most 75 KB) overlap one's staging with the other's MFMAs (config 5: 0.77 -> 0.66 ms),
            // so that budget is tried first as long as it still gives an MFMA-able tile (>= 16 rows)
            const bool lds_forced = std::getenv("SIGOPS_RR_MAXLDS") != nullptr;
            for (int pass = 0; pass < 2 && !best_ct; ++pass) {
                const size_t budget = lds_forced ? rr_max_lds : (pass == 0 ? (size_t)75 * 1024 : rr_max_lds);
                for (int rows : {64, 32, 16, 8, 4, 2, 1}) {
                    if (pass == 0 && !lds_forced && rows < 16) break;
                    for (int ct : {8, 4, 2, 1}) {
                        if (best_ct || N.nch % ct || rows % ct) continue;
                        const int pb = rows / ct;
                        const int64_t tile_len = (pb - 1) * Mb + (jmax - jmin + 1);
                        const int64_t pitch = (tile_len + 3) | 1;  // odd: rows fall on different LDS banks
                        if ((size_t)ct * pitch * esz_t <= budget && tile_len < (1 << 30)) {
                            best_ct = ct;
                            best_pb = pb;
                            best_pitch = pitch;
                            best_len = tile_len;
                            rr_max_lds = budget;
                        }
                    }
                }
            }
            // MFMA path: groups of 16 phases against a [kw x 16] tap block (rows = 16, 32 or 64)
            std::vector<double> mtab;
            std::vector<int> mjend;
            int kw_m = 0, ngroups_m = 0;
            if (best_ct && (best_ct * best_pb) % 16 == 0 && !std::getenv("SIGOPS_RS_NOROWS_MFMA")) {
                const int64_t jmin_scalar = jmin;
                ngroups_m = (int)((Lb + 15) / 16);
                mjend.resize(ngroups_m);
                int64_t maxspan = 0;
                for (int gi = 0; gi < ngroups_m; ++gi) {
                    const int64_t r0 = 16 * (int64_t)gi, r1 = std::min<int64_t>(Lb, r0 + 16);
                    mjend[gi] = jr[r1 - 1];
                    maxspan = std::max<int64_t>(maxspan, jr[r1 - 1] - jr[r0]);
                }
                kw_m = (int)((g.taps + maxspan + 3) / 4 * 4);
                mtab.assign((size_t)ngroups_m * kw_m * 16, 0.0);
                for (int gi = 0; gi < ngroups_m; ++gi) {
                    const int64_t r0 = 16 * (int64_t)gi, r1 = std::min<int64_t>(Lb, r0 + 16);
                    jmin = std::min<int64_t>(jmin, mjend[gi] - (kw_m - 1));
                    for (int64_t r = r0; r < r1; ++r)
                        for (int kk = 0; kk < kw_m; ++kk) {
                            const int64_t age = jr[r] - (mjend[gi] - (kw_m - 1) + kk);
                            if (age >= 0 && age < g.taps)
                                mtab[((size_t)gi * kw_m + kk) * 16 + (r - r0)] = ctab[(size_t)r * g.taps + age];
                        }
                }
                // the window of a group may start a few frames before the oldest tap: re-size the tile
                const int64_t tile_len = (best_pb - 1) * Mb + (jmax - jmin + 1);
                const int64_t pitch = (tile_len + 3) | 1;
                if ((size_t)best_ct * pitch * esz_t <= rr_max_lds + 2048) {
                    best_len = tile_len;
                    best_pitch = pitch;
                } else {
                    kw_m = 0;
                    mtab.clear();
                    jmin = jmin_scalar;
                }
            }
            if (best_ct && jmin > INT32_MIN && jmax < INT32_MAX && (size_t)Lb * g.taps * 8 <= (64u << 20)) {
                RsRows rr{};
                rr.kw = kw_m;
                rr.ngroups = ngroups_m;
                rr.pbshift = 0;
                while ((1 << rr.pbshift) < best_pb) ++rr.pbshift;  // pb is a power of two
                stages[sid].mtab_host = mtab;
                stages[sid].mjend_host = mjend;
                stages[sid].mtab_buf = raw_buf(std::max<size_t>(mtab.size(), 1) * 8);
                stages[sid].mjend_buf = raw_buf(std::max<size_t>(mjend.size(), 1) * 4);
                rr.n_in = g.n_in;
                rr.n_out = need;
                rr.L = Lb;
                rr.M = Mb;
                rr.nperiods = (need + Lb - 1) / Lb;
                rr.taps = g.taps;
                rr.ct = best_ct;
                rr.pb = best_pb;
                rr.jlo = (int)jmin;
                rr.tile_len = (int)best_len;
                rr.pitch = (int)best_pitch;
                rr.nch = N.nch;
                if (const char* ev = std::getenv("SIGOPS_RS_DEBUG")) rr.debug = std::atoi(ev);  // ablation knob
                rr.threads = 1024;
                if (const char* ev = std::getenv("SIGOPS_RR_THREADS")) rr.threads = std::max(64, std::min(1024, std::atoi(ev) / 64 * 64));  // tuning knob
                stages[sid].rows = true;
                stages[sid].rr = rr;
                stages[sid].tab_host = ctab;
                stages[sid].jend_host = jr;
                stages[sid].tab_buf = raw_buf(ctab.size() * 8);
                stages[sid].jend_buf = raw_buf(jr.size() * 4);
            }
        }
        // ---- tiled variant for everything else that is long enough (no period to exploit) ----
        if (!stages[sid].periodic && !stages[sid].rows && need >= 2048 && !std::getenv("SIGOPS_RS_NOTILED")) {
            const int64_t esz_t = (int64_t)dsize(N.dtype);
            const size_t tabs = (size_t)2 * g.taps * g.nphi * 8;
            int ct = 1;
            for (int c : {8, 4, 2})
                if (N.nch % c == 0) {
                    ct = c;
                    break;
                }
            // inputs per output (fine-grid step / Nphi), for sizing the tile
            const double step = g.arbitrary ? g.delta / g.nphi : (double)g.M / (double)g.L;
            for (; ct >= 1; ct >>= 1) {
                if (N.nch % ct) continue;
                // two workgroups per CU when the tables allow it (their staging and arithmetic overlap)
                size_t budget = tabs <= (size_t)24 * 1024 ? (size_t)77 * 1024 - tabs - (size_t)2 * (g.taps + 2 * g.nphi + 2) * 8 : (size_t)150 * 1024 - std::min<size_t>(tabs, 150 * 1024);
                // (the two-outputs-per-lane form keeps the tile as doubles, frame by frame, in rows of ct + 2)
                const bool pair = step <= 6.0 && !std::getenv("SIGOPS_RS_NOPAIR");  // (the windows of a pair within 8 frames)
                const size_t frame_bytes = pair ? (size_t)(ct >= 2 ? ct + 2 : 1) * 8 : (size_t)ct * esz_t;
                int64_t tile_in = (int64_t)(budget / frame_bytes);
                int64_t tile_out = (int64_t)std::floor((double)(tile_in - g.taps - 24) / step);  // (k_resample_tiled2 stages 8 frames before the first window)
                tile_out = std::min<int64_t>(tile_out, 4096);
                if (tabs <= (size_t)100 * 1024 && tile_out >= 128) {
                    RsTiled rt{};
                    rt.g = g;
                    rt.ct = ct;
                    rt.tile_out = (int32_t)tile_out;
                    rt.tile_in = (int32_t)tile_in;
                    rt.pitch = (int32_t)(tile_in | 1);
                    rt.ntiles = (need + tile_out - 1) / tile_out;
                    rt.pair = pair;
                    stages[sid].tiled = true;
                    stages[sid].rt = rt;
                    stages[sid].pfbt_host.assign((size_t)g.taps * g.nphi, 0.0);
                    stages[sid].dpfbt_host.assign((size_t)g.taps * g.nphi, 0.0);
                    for (int p = 0; p < g.nphi; ++p)
                        for (int k = 0; k < g.taps; ++k) {
                            stages[sid].pfbt_host[(size_t)k * g.nphi + p] = stages[sid].pfb_host[(size_t)p * g.taps + k];
                            stages[sid].dpfbt_host[(size_t)k * g.nphi + p] = stages[sid].dpfb_host[(size_t)p * g.taps + k];
                        }
                    stages[sid].pfbt_buf = raw_buf(stages[sid].pfbt_host.size() * 8);
                    stages[sid].dpfbt_buf = raw_buf(stages[sid].dpfbt_host.size() * 8);
                    // ---- the persistent form (k_resample_arb): Float64, DSP.jl's 32 phases, windows of a pair close ----
                    // (a persistent workgroup zeroes its ring and builds its tables first: ~22 us before the first output, 15
                    //  for the tiled kernel; from 1.5 M output samples on the pipelined walk is ahead -- tools/arb_len_sweep.py)
                    const int64_t arb_min = env_int("SIGOPS_ARB_MIN", 1500000);
                    if (pair && g.nphi == 32 && (N.dtype == SO_F64 || N.dtype == SO_F32) && need >= 16384 && need * N.nch >= arb_min &&
                        !std::getenv("SIGOPS_RS_NOARB")) {
                        const int esz_a = (int)dsize(N.dtype), chf = 1024 / esz_a;  // (ring element bytes; frames per loader chunk)
                        RsArb ra{};
                        ra.g = g;
                        int cta = 1;
                        for (int c : {8, 4, 2})
                            if (N.nch % c == 0) {
                                cta = c;
                                break;
                            }
                        ra.ct = cta;
                        const int dmax = (int)std::ceil(step) + 1;  // newest inputs of two consecutive outputs, at most this far apart
                        // four outputs per lane where their windows stay close and all eight channels are in the
                        // workgroup: the LDS reads of a frame serve four outputs (the kernel is bound by them)
                        // (four outputs per lane -- the LDS reads of a frame serve four outputs -- is built and measured:
                        //  compute waves alone 0.204 ms instead of 0.185, its 32-byte-strided stores cost what the
                        //  reads save; opt-in)
                        int no = env_int("SIGOPS_ARB_NO", 2);
                        if (no != 4 || cta != 8 || step > 1.5) no = 2;
                        ra.no = no;
                        ra.zrows = (no - 1) * dmax + 3;
                        int ringf = 4096;
                        while (ringf >= 512 && resample_arb_lds_bytes(g.taps, ra.zrows, cta, ringf, esz_a) + 1024 > (size_t)160 * 1024) ringf >>= 1;
                        ringf = env_int("SIGOPS_ARB_RING", ringf);
                        ra.depth = env_int("SIGOPS_ARB_DEPTH", no == 4 ? 2 : 4);
                        ra.debug = env_int("SIGOPS_ARB_DEBUG", 0);
                        const int per_batch = (int)std::ceil(64.0 * no * step) + 2;
                        // a batch's own span + what the loader has in flight must fit next to NC batches in progress
                        int nc = (ringf - (ra.depth * chf + chf - 1 + g.taps + no * dmax + 32)) / per_batch;
                        nc = std::min(nc, no == 4 ? 7 : 8);  // (pairs: 4 waves 0.268 ms, 6 0.242, 8 0.231, 11 0.235 on the x pi / 3 bench)
                        nc = std::min(nc, env_int("SIGOPS_ARB_NC", nc));
                        ra.nc = nc;
                        ra.ringf = ringf;
                        ra.nbatches = (need + 64 * no - 1) / (64 * no);
                        const int64_t ncg = N.nch / cta;
                        int64_t nr = std::max<int64_t>(1, env_int("SIGOPS_ARB_GRID", 256) / ncg);
                        nr = std::min(nr, std::max<int64_t>(1, ra.nbatches / (2 * std::max(nc, 1))));
                        ra.bpr = (ra.nbatches + nr - 1) / nr;
                        ra.nranges = (int32_t)((ra.nbatches + ra.bpr - 1) / ra.bpr);
                        if (nc >= 4 && ringf >= 1024 && ringf >= 512 && ncg * ra.nranges < (1ll << 31)) {
                            stages[sid].arbk = true;
                            stages[sid].ra = ra;
                        }
                    }
                    break;
                }
            }
        }
        if (!wrap.empty() && !stages[sid].periodic && !stages[sid].rows)  // no tap table took the baked positions
            replay_phase_accumulator(stages[sid].rg, (const double*)nd.p0, nd.i2, stages[sid].base + need, false, wrap, stages[sid].fix_host,
                                     stages[sid].base);
        if (!stages[sid].fix_host.empty()) stages[sid].fix_buf = raw_buf(stages[sid].fix_host.size() * sizeof(RsFix));
    } else if (stages[sid].kind == ST_SOS) {
        if (!isinf_(C.len)) in_frames = std::min(need, C.len.n);
        int nsec = merged_sos.empty() ? nd.i0 : (int)(merged_sos.size() / 6);
        const double* sos = merged_sos.empty() ? (const double*)nd.p0 : merged_sos.data();
        const double gain_all = merged_gain;
        std::vector<SosCoefs> groups;
        for (int s0 = 0; s0 < nsec; s0 += kMaxSec) {
            SosCoefs cf{};
            cf.nsec = std::min(kMaxSec, nsec - s0);
            for (int f = 0; f < cf.nsec; ++f) {
                const double* b = sos + 6 * (s0 + f);
                if (b[3] != 1.0) fail(SO_ERR_INVALID, "SOS rows must be normalised (a0 == 1)");
                cf.b0[f] = b[0];
                cf.b1[f] = b[1];
                cf.b2[f] = b[2];
                cf.a1[f] = b[4];
                cf.a2[f] = b[5];
            }
            cf.gain = (s0 + kMaxSec >= nsec) ? gain_all : 1.0;
            groups.push_back(cf);
        }
        // ---- ill-conditioned cascade: the reference's own order of operations (see above) ----
        bool exact = false;
        {
            const char* ev = std::getenv("SIGOPS_SOS_EXACT");  // 1: always, 0: never (measurement aid)
            if (ev) exact = std::atoi(ev) != 0;
            else {
                exact = nsec >= 3 && sos_rounding_sensitivity(sos, nsec, gain_all) > kSosExactTol;
                if (!exact && nsec >= 2) {  // ... or the chunked form itself, at the chunk length this signal would get
                    const int64_t tgt = kSosSequences / std::max(1, N.nch);
                    const int64_t nck = std::max<int64_t>(1, std::min<int64_t>(tgt, need / 64));
                    const int64_t Lp = ((need + nck - 1) / nck + 31) / 32 * 32;
                    const double cs = nck > 1 ? sos_chunk_sensitivity(groups, Lp, need) : 0.0;
                    exact = cs > kSosExactTol;
                    if (std::getenv("SIGOPS_DEBUG_PLAN"))
                        std::fprintf(stderr, "[sigops] IIR of %d sections: rounding sensitivity %.3g, chunked (L = %lld) %.3g -> %s\n", nsec,
                                     nsec >= 3 ? sos_rounding_sensitivity(sos, nsec, gain_all) : 0.0, (long long)Lp, cs, exact ? "sequential" : "time-parallel");
                }
            }
        }
        // ---- warm start: frames before the first one anybody reads (After, a later window of a
        //      stream) matter only through the filter state, and what a state contributes has decayed
        //      below 2^-70 after W frames: start from zero state W frames early instead of at frame 0.
        //      (The reference filters the skipped frames, src/cutting.jl:160-173; same values.) ----
        // (not below a Normpower: the state the skipped frames leave is tiny, not negligible, once the
        //  result is divided by the rms of a decayed tail)
        if (!exact && !stages[sid].under_norm && stages[sid].lo < need && stages[sid].lo >= 8192 &&
            !std::getenv("SIGOPS_NO_WARM_START")) {
            int64_t Wd = 0;
            for (auto& cf : groups) {
                const int D = 2 * cf.nsec;
                Mat P = sos_state_matrix(cf);
                int64_t w = 1;
                while (maxabs(P) >= std::ldexp(1.0, -70) && w < ((int64_t)1 << 40)) {
                    P = matmul(P, P, D);
                    w <<= 1;
                }
                Wd += w;  // (groups are cascaded: decay times add up at worst)
            }
            if (stages[sid].lo - Wd >= 4096) {
                const int64_t base = (stages[sid].lo - Wd) / 64 * 64;
                stages[sid].base = stages[sid].in_base = base;
                need -= base;  // local frames from here on
                in_frames = need;
            }
        }
        // ---- single pass (one read, one write): wave tiles in time order with a look-back over the
        //      zero-state end states of the kt previous tiles (see k_sos_onepass) ----
        // Opt-in (SIGOPS_SOS_ONEPASS=1): its HBM traffic is the algorithmic minimum, but on MI355X it
        // is bound by fp64 vector work and dependent chains at two waves per SIMD (28.8 M x 8, order
        // 10: 1.7 ms against 1.13 ms for the three-pass form, which streams at ~5 TB/s) -- DESIGN.md.
        if (!exact && need >= 4096 && std::getenv("SIGOPS_SOS_ONEPASS") && !std::getenv("SIGOPS_SOS_3PASS")) {
            SosOne o{};
            const int tf = 64 * kSosLc;
            o.n = need;
            o.nch = N.nch;
            o.ntiles = (int)((need + tf - 1) / tf);
            o.nlev = 6;
            o.bt = std::min(4, N.nch);
            if (const char* ev = std::getenv("SIGOPS_SOS_BT")) o.bt = std::max(1, std::min(4, std::atoi(ev)));  // tuning knob
            if (const char* ev = std::getenv("SIGOPS_SOS_DEBUG")) o.debug = std::atoi(ev);  // ablation knob
            const double tol1 = std::ldexp(1.0, -70);
            bool ok = (int64_t)o.ntiles * o.nch < (1 << 30);
            // look-back depth: first kt with ||(A^tf)^kt|| < 2^-70, the same for every group
            int kt = 1;
            std::vector<Mat> As, Ts;
            for (auto& cf : groups) {
                const int D = 2 * cf.nsec;
                Mat A = sos_state_matrix(cf);
                Mat T = matpow(A, tf, D);
                As.push_back(A);
                Ts.push_back(T);
                Mat cur = T;
                int k = 1;
                while (ok && !(maxabs(cur) < tol1)) {
                    cur = matmul(cur, T, D);
                    if (++k > 64) ok = false;  // a pole this close to the unit circle: three-pass form
                }
                kt = std::max(kt, k);
            }
            if (ok) {
                o.kt = kt;
                std::vector<double> tabs;
                std::vector<size_t> offs;
                for (size_t gi = 0; gi < groups.size(); ++gi) {
                    const int D = 2 * groups[gi].nsec;
                    offs.push_back(tabs.size());
                    Mat P = matpow(As[gi], kSosLc, D);  // M = A^lc, then M^2, M^4, ...
                    for (int lev = 0; lev < o.nlev; ++lev) {
                        tabs.insert(tabs.end(), P.begin(), P.end());
                        P = matmul(P, P, D);
                    }
                    Mat cur = ident(D);
                    for (int j = 0; j < kt; ++j) {  // (A^tf)^j
                        tabs.insert(tabs.end(), cur.begin(), cur.end());
                        cur = matmul(cur, Ts[gi], D);
                    }
                }
                stages[sid].onepass = true;
                stages[sid].so1 = o;
                stages[sid].one_tabs_host = tabs;
                stages[sid].one_tabs_off = offs;
                stages[sid].one_tabs_buf = raw_buf(tabs.size() * 8);
                stages[sid].one_sync_buf = raw_buf(64);
                stages[sid].one_vpub_buf = raw_buf((size_t)o.ntiles * o.nch * 2 * kMaxSec * 8);
            }
        }
        sos_chunking(sid, need, N.nch, N.dtype, groups, exact, kSosSequences / std::max(1, N.nch));
    } else {  // ST_NORM
        in_frames = need;
        int64_t total = need * N.nch;
        int nparts = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (total + kBlock * 8 - 1) / (kBlock * 8)));
        stages[sid].nparts = nparts;
        // (Float32: one Float32 partial per block of 1024 values, two copies for the pairwise fold)
        const size_t nb32 = N.dtype == SO_F32 ? (size_t)((total + 1023) / 1024) : 0;
        stages[sid].partial_buf = raw_buf(std::max((size_t)nparts * 8, 2 * nb32 * 4));
    }

    // lower the child over the frames this stage consumes
    std::vector<Piece> ps;
    const int64_t in_base = stages[sid].in_base;
    if (in_base > 0) check_frames(child, in_base);
    {
        // whatever this stage reads feeds a Normpower too if the stage itself does (or is one): a
        // filter there must keep the RELATIVE accuracy of a decayed tail (see launch_sos_xscan)
        const bool norm_below = stages[sid].kind == ST_NORM || stages[sid].under_norm;
        in_norm += norm_below;
        if (in_frames > 0) ps = lower(child, Rect{0, in_frames, 0, N.nch}, Map{1, in_base, 1, 0});
        in_norm -= norm_below;
    }
    if (stages[sid].kind == ST_SOS && in_frames > 0) {  // the reference filters whole blocks of its input
        const int64_t bs = std::max(1, N.nd.i1);
        int64_t fr = (in_base + in_frames + bs - 1) / bs * bs;
        for (int64_t b2 : merged_bs) fr = (fr + b2 - 1) / b2 * b2;  // (... and so does every filter merged into this one)
        check_frames(child, fr);
    }
    if (stages[sid].kind == ST_RESAMPLE && in_frames > 0) {
        // ... and so does the resampler: it refills its input `rows` frames at a time, rows = trunc(max(1, min(N_out,
        // blocksize) / ratio)) (reference src/filters.jl:185-199 init_length, :237-244 refill), so an error that sits in
        // child frames the outputs asked for never depend on -- an indexing pad on a computed signal beyond them -- is
        // raised all the same (tools/tree_soak_multirate.py seed 16046)
        const RsGeom& rg = stages[sid].rg;
        const int64_t bs = nodes[ni].nd.i3 > 0 ? nodes[ni].nd.i3 : 4096;
        const double ratio = rg.arbitrary ? nodes[ni].nd.d0 : (double)rg.L / (double)rg.M;
        const int64_t total = isinf_(nodes[ni].len) ? bs : nodes[ni].len.n;
        const int64_t rows = (int64_t)std::trunc(std::max(1.0, (double)std::min<int64_t>(total, bs) / ratio));
        if (rows > 0 && ratio > 0) check_frames(child, (in_base + in_frames + rows - 1) / rows * rows);
    }
    if (stages[sid].out_buf >= 0) bufs[stages[sid].out_buf].frame0 = stages[sid].base;
    Stage& S = stages[sid];  // (re-taken: lower() may have appended stages)
    S.in_frames = in_frames;
    int in_dtype = S.kind == ST_NORM ? N.dtype : C.dtype;
    if (S.kind != ST_NORM && float_of(C.dtype) != N.dtype) fail(SO_ERR_INVALID, "filter dtype mismatch");
    if (S.kind != ST_NORM && C.dtype == SO_I64) in_dtype = SO_F64;
    // direct source: a single plain contiguous load of the right type
    bool direct = false;
    if ((S.kind != ST_NORM || S.norm_direct) && ps.size() == 1) {
        const Expr& e = exprs[ps[0].e];
        if (e.op == E_LOAD && e.leaf.mode == LM_PLAIN && e.leaf.sf == 1 && e.leaf.sc == 1 &&
            e.leaf.fstride == 1 && e.leaf.dtype == in_dtype && e.leaf.df >= 0 && e.leaf.dc >= 0 &&
            (e.leaf.cstride > 0 || e.leaf.cstride == -1 || N.nch == 1)) {
            direct = true;
            S.in_array_node = e.array_node;
            S.in_buf = e.leaf.buf;  // stage buffer or -1 (array)
            S.in_offset = e.leaf.df;
            S.in_pitch = e.leaf.cstride;  // -1: pitch of in_buf
            if (e.array_node >= 0) {
                // element offset = df*fstride + dc*cstride
                S.in_offset = e.leaf.df + e.leaf.dc * std::max<int64_t>(e.leaf.cstride, 0);
            } else {
                if (e.leaf.dc != 0) direct = false;
            }
        }
        // ... or that load plus / times ONE sine generator (`Mix(Signal(sin), x)`, `Amplify(x, Signal(sin))`
        // in Float64): the IIR forms the sum / product in its own loads (k_sos_tiled, SosGeom::src_op) --
        // no K1 pass, no intermediate in HBM (config 4: every scene is Mix |> Filt)
        if (!direct && S.kind == ST_SOS && (e.op == E_ADD || e.op == E_MUL) && e.dtype == SO_F64 && N.dtype == SO_F64 &&
            !S.sg.exact && !S.onepass && in_frames == need && !std::getenv("SIGOPS_SOS_NOSRC")) {
            auto strip = [&](int x) {
                while (exprs[x].op == E_RETYPE) x = exprs[x].a;
                return x;
            };
            const int ea = strip(e.a), eb = strip(e.b);
            const int li = exprs[ea].op == E_LOAD ? ea : (exprs[eb].op == E_LOAD ? eb : -1);
            const int fi = li == ea ? eb : ea;
            if (li >= 0) {
                const Expr& l = exprs[li];
                const Expr& f = exprs[fi];
                const bool load_ok = l.leaf.mode == LM_PLAIN && l.leaf.sf == 1 && l.leaf.sc == 1 && l.leaf.fstride == 1 &&
                                     l.leaf.dtype == SO_F64 && l.leaf.df >= 0 && l.leaf.dc >= 0 &&
                                     (l.leaf.cstride > 0 || l.leaf.cstride == -1 || N.nch == 1) &&
                                     (l.array_node >= 0 || l.leaf.dc == 0);
                const bool fn_ok = f.op == E_FUNC && f.leaf.mode == SO_FN_SIN && f.leaf.sf == 1 && f.dtype == SO_F64;
                if (load_ok && fn_ok) {
                    direct = true;
                    S.in_array_node = l.array_node;
                    S.in_buf = l.leaf.buf;
                    S.in_offset = l.leaf.df;
                    S.in_pitch = l.leaf.cstride;
                    if (l.array_node >= 0) S.in_offset = l.leaf.df + l.leaf.dc * std::max<int64_t>(l.leaf.cstride, 0);
                    S.src_op = e.op == E_ADD ? 1 : 2;
                    S.src_fn = f.leaf;
                    if (l.array_node >= 0) count_array(l.array_node);
                }
            }
        }
    }
    // A filter over a plain source (an array, a stage buffer, either one plus / times a sine) may run as ONE pass of the
    // block-state-space kernel (Plan::fuse_plain_sos: k_rsos with an identity resampler) instead of the three-pass chunked
    // scan: its source as a carrier, like a periodic resampler's.  The direct view above stays -- what the three passes
    // read if the fused form is not taken.
    // (below a Normpower: where every Normpower's region starts at this filter's first frame -- Stage::norm_df == 0.  What the
    //  one-pass form's warm starts cut, 2^-70 of what lay a decay time earlier, is then 2^-70 of something INSIDE the region
    //  the rms is taken over; a Normpower of a decayed tail alone -- `Filt |> After |> Normpower` -- keeps the exact scan.)
    if (direct && S.kind == ST_SOS && S.groups.size() == 1 && S.groups[0].nsec <= 6 && !S.sg.exact && !S.onepass &&
        (!S.under_norm || (S.norm_df == 0 && !std::getenv("SIGOPS_NORM_EXACT_FILT"))) &&
        S.base == 0 && in_base == 0 && in_frames == need && (N.dtype == SO_F64 || (N.dtype == SO_F32 && !std::getenv("SIGOPS_RSOS_NO32"))) &&
        need * N.nch >= (std::getenv("SIGOPS_RSOS_MINGROUPS") || (std::getenv("SIGOPS_RSOS_BATCH") && std::atoi(std::getenv("SIGOPS_RSOS_BATCH")) == 1) ? (int64_t)4096 : ((int64_t)1 << 22)) && !std::getenv("SIGOPS_NO_RSOS") && !std::getenv("SIGOPS_NO_PLAIN_RSOS")) {
        std::vector<DCarrier> cs;
        // (a Float32 signal: a plain Float32 array or buffer only -- the kernel's ring then keeps the Float32 samples)
        if (build_carriers(ps, N.nch, cs, false) && cs.size() == 1 && (N.dtype == SO_F64 || (cs[0].nsteps == 0 && cs[0].dtype == SO_F32))) S.carriers = cs;
    }
    if (S.kind == ST_NORM && S.norm_alias) {
        // `vals` is the child stage's buffer (planner.cpp): lowering the child above registered the frames it needs
        if (ps.size() != 1 || exprs[ps[0].e].op != E_LOAD || exprs[ps[0].e].leaf.buf != S.out_buf)
            fail(SO_ERR_RUNTIME, "internal: Normpower over a stage buffer expected that buffer");
        S.in_buf = S.out_buf;
        S.in_pitch = -1;
    } else if (S.kind == ST_NORM && S.norm_direct && direct && S.in_array_node >= 0) {
        // the array itself is `vals` (planner.cpp): K4 reads it in place
    } else if (S.kind == ST_NORM && S.norm_direct) {
        // planner.cpp took the in-place form from the node kinds (an array under Until / After), the lowered child is not
        // ONE plain load after all (pieces split at a boundary, a view the leaf test does not take): the sum of squares
        // is taken over a materialised copy as it used to be -- the readers, lowered as "the child's own pieces ./ rms",
        // are unaffected -- instead of refusing the plan
        S.norm_direct = false;
        S.in_array_node = -1;
        S.in_offset = 0;
        S.pw_step = emit_pointwise(ps, S.out_buf, N.dtype);
        S.in_buf = S.out_buf;
        S.in_pitch = -1;
    } else if (S.kind == ST_NORM) {
        // materialise the child straight into `vals` (the stage's own output buffer)
        S.pw_step = emit_pointwise(ps, S.out_buf, N.dtype);
        S.in_buf = S.out_buf;
        S.in_pitch = -1;
    } else if (!direct && S.kind == ST_RESAMPLE && S.periodic &&
               build_carriers(ps, N.nch, S.carriers, ga_fits(S, N.dtype),
                              // (the A2 instantiations: Float64 -- or Float32 all the way --, 32-row tiles of 4 or 8 channels, 14 k-steps, one group per compute wave)
                              // ... and two tile slots + the loader waves' staging rows of the second array fit LDS)
                              // (Float64 groups of FOUR channels: measured 1.81 ms for two 25 M x 4 arrays where K1's sum + the one-array
                              //  kernel take 0.87 -- round 6, tools/operator_matrix.py NCH=4; left to K1 unless SIGOPS_RS_ARR2_CT4.  With a
                              //  filter behind, the fused kernel takes both arrays itself: Stage::alt_carriers below)
                              (N.dtype == SO_F64 || (N.dtype == SO_F32 && S.rp.f32m)) && S.rp.rows == 32 &&
                                  (S.rp.ct == 8 || (S.rp.ct == 4 && (N.dtype == SO_F32 || std::getenv("SIGOPS_RS_ARR2_CT4")))) && S.rp.kw == 56 &&
                                  (S.rp.ngroups + S.rp.ncompute - 1) / S.rp.ncompute == 1 &&
                                  (size_t)2 * S.rp.ct * S.rp.lds_pitch * dsize(N.dtype) + (size_t)(S.rp.nwaves - S.rp.ncompute) * S.rp.ct * 1024 <=
                                      160 * 1024 - sizeof(RsCtl) - 64)) {
        // every piece is `array (op) per-frame values`: evaluated inside the kernel's LDS
        // staging, no intermediate in HBM
        S.in_buf = -1;
        S.in_array_node = -1;
        for (auto& c : S.carriers)
            if (car_has_arr2(c)) S.rp.arr2 = 1;
        if (S.rp.arr2) S.rp.nslots = 2;  // (the A2 instantiation: one tile of look-ahead, the rest of LDS for the second array's staging rows)
        if (S.carriers[0].pad_) {
            // GA instantiation: Float32 tiles (pitch in floats, 16-byte rows, 128-byte aligned start),
            // three gain arrays, no in-place work for the loaders
            RsPeriodic& rp = S.rp;
            const size_t avail = 160 * 1024 - sizeof(RsCtl) - 64;
            rp.ga = S.carriers[0].pad_ >= 3 ? 2 : 1;
            rp.lds_pitch = (int)((rp.tile_len + 31 + 8 + 3) / 4 * 4);
            const size_t tile_bytes = (size_t)rp.ct * rp.lds_pitch * 4;
            rp.fslots = 1;
            // gains are indexed from the 128-byte aligned frame the tile is staged from: up to 31 frames
            // below its first input for Float32 tiles (16 were reserved until a long-signal soak found the
            // gains of one tile running into the array the compute waves were reading)
            rp.fpitch = (rp.tile_len + 32 + 1) & ~1;
            const DLeaf& L0 = leaves[S.carriers[0].slot_leaf[0]];
            const bool two = !std::getenv("SIGOPS_RS_NOTWO") && rp.tile_len <= 64 * kRsTwoBases &&
                             (S.carriers[0].slot_kind[0] & 0xff) == OP_FUNC && L0.mode == SO_FN_SIN && L0.sf == 1;
            rp.ftwo = two ? 1 : 0;
            const size_t fbytes = (size_t)3 * rp.fpitch * 8 + (two ? kRsTwoDoubles * 8 : 0);
            rp.nslots = (int)std::min<size_t>(4, (avail - fbytes) / tile_bytes);
            if (const char* ev = std::getenv("SIGOPS_RS_SLOTS")) rp.nslots = std::min(rp.nslots, std::max(2, std::atoi(ev)));
            rp.nload = 0;
        } else {
        // gain ring: two LDS arrays [slots][tile frames] next to the tile ring, if at least two
        // tile slots still fit (see k_resample_periodic)
            RsPeriodic& rp = S.rp;
            const int ns0 = S.carriers[0].nslots;
            const size_t tile_bytes = (size_t)rp.ct * rp.lds_pitch * dsize(N.dtype);
            const size_t avail = 160 * 1024 - sizeof(RsCtl) - 64;
            const int fpitch = (rp.tile_len + 16 + 1) & ~1;
            // slot 0 (the only one) a sine generator: the kernel's two-level evaluation (TWO)
            bool two = !std::getenv("SIGOPS_RS_NOTWO") && rp.tile_len <= 64 * kRsTwoBases && ns0 == 1 && N.dtype == SO_F64 &&
                       (rp.ngroups + rp.ncompute - 1) / rp.ncompute == 1;
            if (two) {
                const DLeaf& L0 = leaves[S.carriers[0].slot_leaf[0]];
                two = (S.carriers[0].slot_kind[0] & 0xff) == OP_FUNC && L0.mode == SO_FN_SIN && L0.sf == 1;
            }
            const size_t fbytes = (size_t)2 * ns0 * fpitch * 8 + (two ? kRsTwoDoubles * 8 : 0);
            if (ns0 > 0 && S.carriers[0].nsteps > 0 && !std::getenv("SIGOPS_RS_NOFRING") &&
                fbytes + 2 * tile_bytes <= avail) {
                rp.fslots = ns0;
                rp.fpitch = fpitch;
                rp.ftwo = two ? 1 : 0;
                // the in-place multiply is vector-ALU work next to the MFMAs: keep it off the
                // SIMDs that carry the most compute waves (10 compute waves: loaders 10,11,14,15
                // on SIMD 2/3 copy and modify, 12,13 only keep the barrier count; measured
                // 0.790 -> 0.780 ms on config 3, three alternating runs each)
                if (!std::getenv("SIGOPS_RS_NLOAD")) {
                    auto ncomp_on = [&](int w) { return (rp.ncompute - (w & 3) + 3) >> 2; };
                    int minc = 1 << 30, cnt = 0;
                    for (int w = rp.ncompute; w < rp.nwaves; ++w) minc = std::min(minc, ncomp_on(w));
                    for (int w = rp.ncompute; w < rp.nwaves; ++w) cnt += ncomp_on(w) == minc;
                    if (cnt >= 2) rp.nload = cnt;
                }
                rp.nslots = (int)std::min<size_t>(rp.nslots, (avail - fbytes) / tile_bytes);
            }
        }
    } else if (!direct) {
        // (a periodic resampler over `x (op) y` of two Float64 arrays in a shape its own two-array form does not take -- two
        //  channels --: K1 materialises the map, but the fused resampler + IIR kernel could read both arrays itself; keep the
        //  map as a two-array carrier for fuse_resample_sos to swap in IF it takes the stage)
        if (S.kind == ST_RESAMPLE && S.periodic && N.dtype == SO_F64 && in_dtype == SO_F64 && N.nch % 2 == 0 && !std::getenv("SIGOPS_RSOS_NO_ARR2") &&
            !std::getenv("SIGOPS_NO_ARR2")) {
            std::vector<DCarrier> alt;
            if (build_carriers(ps, N.nch, alt, false, true) && alt.size() == 1 && alt[0].nsteps == 1 && car_has_arr2(alt[0])) S.alt_carriers = alt;
        }
        S.in_buf = new_buf(in_frames, N.nch, in_dtype);
        S.in_pitch = -1;
        S.in_array_node = -1;
        S.in_offset = 0;
        S.pw_step = emit_pointwise(ps, S.in_buf, in_dtype);
    }
    if (S.kind == ST_RESAMPLE && S.periodic && S.carriers.empty()) {
        // plain source (direct array / stage buffer / materialised input): one 0-step carrier
        DCarrier c{};
        c.a = 0;
        c.b = in_frames;
        c.dtype = in_dtype;
        c.array_node = S.in_array_node;
        c.buf = S.in_array_node >= 0 ? -1 : S.in_buf;
        c.df = S.in_offset;
        c.cstride = S.in_array_node >= 0 ? (N.nch == 1 ? 0 : S.in_pitch) : -1;  // -1: buffer pitch
        S.carriers.push_back(c);
    }
    if (!S.carriers.empty()) {
        S.car_buf = raw_buf(S.carriers.size() * sizeof(DCarrier));
        S.ctl_buf = raw_buf(sizeof(RsCtl));
    }
}

// e == carrier load combined with channel-independent operands by a short chain of ops?
bool Plan::match_carrier(int ei, DCarrier& C, std::vector<int>& monos) {
    const Expr e = exprs[ei];
    auto add_step = [&](int op, int mono_expr, bool flip, bool round32) {
        if (C.nsteps >= 4) return false;
        int slot = 0;
        if (mono_expr >= 0) {
            auto it = std::find(monos.begin(), monos.end(), mono_expr);
            if (it == monos.end()) {
                if ((int)monos.size() >= kMaxFrameSlots) return false;
                monos.push_back(mono_expr);
                slot = (int)monos.size() - 1;
            } else slot = (int)(it - monos.begin());
        }
        C.op[C.nsteps] = op;
        C.arg[C.nsteps] = slot | (flip ? 0x100 : 0) | (round32 ? 0x200 : 0);
        C.nsteps++;
        return true;
    };
    if (e.mono && C.nsteps == 0 && C.base == nullptr && C.buf == -1 && C.array_node == -1) {
        // channel-independent piece (generator, constant, padding value): no array at all,
        // the value is a per-frame slot
        C.dtype = e.dtype == SO_F32 ? SO_F32 : SO_F64;
        return add_step(OP_LOADF, ei, false, false);
    }
    switch (e.op) {
    case E_LOAD: {
        const DLeaf& L = e.leaf;
        if (L.mode != LM_PLAIN || L.sf != 1 || L.sc != 1 || L.fstride != 1) return false;
        if (e.array_node < 0 && L.dc != 0) return false;
        if (L.dc < 0) return false;
        C.dtype = L.dtype;
        C.array_node = e.array_node;
        C.buf = e.array_node >= 0 ? -1 : L.buf;
        C.cstride = L.cstride;  // -1: buffer pitch (patched in finalize)
        C.df = L.df + (e.array_node >= 0 ? L.dc * L.cstride : 0);
        if (e.array_node >= 0) count_array(e.array_node);
        return true;
    }
    case E_RETYPE: return match_carrier(e.a, C, monos);
    case E_ROUND32: return match_carrier(e.a, C, monos) && add_step(OP_ROUND32, -1, false, true);
    case E_NEG: return match_carrier(e.a, C, monos) && add_step(OP_NEG, -1, false, false);
    case E_ADD:
    case E_SUB:
    case E_MUL:
    case E_DIV: {
        int oc = e.op == E_ADD ? OP_ADD : e.op == E_SUB ? OP_SUB : e.op == E_MUL ? OP_MUL : OP_DIV;
        bool r32 = e.dtype == SO_F32;
        if (exprs[e.b].mono) {
            DCarrier c2 = C;
            std::vector<int> m2 = monos;
            if (match_carrier(e.a, c2, m2)) {
                C = c2;
                monos = m2;
                return add_step(oc, e.b, false, r32);
            }
        }
        if (exprs[e.a].mono) {
            DCarrier c2 = C;
            std::vector<int> m2 = monos;
            if (match_carrier(e.b, c2, m2)) {
                C = c2;
                monos = m2;
                return add_step(oc, e.a, true, r32);
            }
        }
        // `x (op) y` of two Float64 arrays (`Mix(x, y)`, `Amplify(x, y)`): x the carrier, y its one step's operand (the
        // resampler's A2 instantiation; the caller says whether its geometry has one)
        // (... or of two Float32 arrays of a Float32 signal: the step rounds to Float32, the Float32 tile's own arithmetic)
        const int adt = e.dtype == SO_F32 ? SO_F32 : SO_F64;
        if (carrier_arr2_ok && oc != OP_DIV && (e.dtype == SO_F64 ? !r32 : (e.dtype == SO_F32 && r32)) && C.nsteps == 0 && monos.empty()) {
            auto plain_array = [&](int ei2, DCarrier& c) {
                const Expr* x = &exprs[ei2];
                if (x->op != E_LOAD || x->array_node < 0) return false;  // (a caller's array; stage buffers stay with K1)
                const DLeaf& L = x->leaf;
                if (L.mode != LM_PLAIN || L.sf != 1 || L.sc != 1 || L.fstride != 1 || L.dc < 0 || L.dtype != adt) return false;
                c.dtype2 = L.dtype;
                c.array_node2 = x->array_node;
                c.buf2 = -1;
                c.cstride2 = L.cstride;
                c.df2 = L.df + L.dc * L.cstride;
                return true;
            };
            DCarrier c2 = C;
            std::vector<int> m2;
            if (exprs[e.a].op == E_LOAD && match_carrier(e.a, c2, m2) && c2.nsteps == 0 && c2.dtype == adt && m2.empty() && plain_array(e.b, c2)) {
                C = c2;
                C.op[0] = oc;
                C.arg[0] = kCarArr2 | (r32 ? 0x200 : 0);
                C.nsteps = 1;
                count_array(C.array_node2);
                return true;
            }
        }
        return false;
    }
    default: return false;
    }
}

bool Plan::build_carriers(const std::vector<Piece>& ps_in, int nch, std::vector<DCarrier>& out, bool allow_ga, bool allow_arr2) {
    struct Arr2Scope {  // (match_carrier's permission to form a step on a second array: for this call only)
        bool& flag;
        bool old;
        ~Arr2Scope() { flag = old; }
    } arr2_scope{carrier_arr2_ok, carrier_arr2_ok};
    carrier_arr2_ok = allow_arr2 && !std::getenv("SIGOPS_NO_ARR2");
    std::vector<Piece> ps = ps_in;
    for (auto& p : ps)
        if (p.r.c0 != 0 || p.r.c1 != nch) { if (std::getenv("SIGOPS_DEBUG_PLAN")) std::fprintf(stderr, "[sigops] carrier fusion rejected (#%d)\n", 1); return false; }
    std::sort(ps.begin(), ps.end(), [](const Piece& a, const Piece& b) { return a.r.a < b.r.a; });
    std::vector<DCarrier> cs;
    std::vector<std::vector<int>> monos_all;
    for (auto& p : ps) {
        DCarrier c{};
        c.buf = -1;
        c.array_node = -1;
        std::vector<int> monos;
        if (!match_carrier(p.e, c, monos)) { if (std::getenv("SIGOPS_DEBUG_PLAN")) std::fprintf(stderr, "[sigops] carrier fusion rejected (#%d)\n", 2); return false; }
        c.a = p.r.a;
        c.b = p.r.b;
        cs.push_back(c);
        monos_all.push_back(monos);
    }
    // a step on a second array: carrier 0's (the kernel's fast path is carrier 0's), and nothing else in that carrier
    for (size_t i = 0; i < cs.size(); ++i)
        if (car_has_arr2(cs[i]) && (cs[i].nsteps != 1 || !car_has_arr2(cs[0]))) { if (std::getenv("SIGOPS_DEBUG_PLAN")) std::fprintf(stderr, "[sigops] carrier fusion rejected (#%d)\n", 8); return false; }
    // compile the per-frame programs; everything must fit the kernel-argument control block
    if (cs.size() > (size_t)kCtlCar) { if (std::getenv("SIGOPS_DEBUG_PLAN")) std::fprintf(stderr, "[sigops] carrier fusion rejected (#%d)\n", 3); return false; }
    // fp32 stages: the kernel's in-place steps are fp64-only (fp32 tiles go through the general
    // staging path, ~6x slower than a K1 pass + the LDS-DMA fast path), so steps on fp32 data
    // are materialised by K1 instead of fused
    // ... except the commonest case, ONE Float32 array times ONE Float64 per-frame gain (`Amplify(x32,
    // Signal(sin))`, a Float64 product): the kernel's GA instantiation keeps the raw Float32 tile
    // and multiplies at the A operand (allow_ga: the caller has checked geometry and LDS budget)
    // Further carriers may only be generated pieces whose value is that same gain (the tail of an
    // infinite `Amplify`: the array's padding `one` times the gain): staged as 1.0f.
    bool ga = false;
    bool ga_add = false;  // ... or PLUS one Float64 per-frame operand (`Mix(x32, Signal(sin))`): added at the A operand
    if (allow_ga && !cs.empty() && cs[0].dtype == SO_F32 && cs[0].nsteps == 1 && (cs[0].op[0] == OP_MUL || cs[0].op[0] == OP_ADD) &&
        !(cs[0].arg[0] & 0x200) && (cs[0].array_node >= 0 || cs[0].buf >= 0) && monos_all[0].size() == 1) {
        ga = true;
        ga_add = cs[0].op[0] == OP_ADD;
        for (size_t i = 1; i < cs.size(); ++i)
            if (cs[i].base != nullptr || cs[i].array_node >= 0 || cs[i].buf >= 0 || cs[i].nsteps != 1 ||
                cs[i].op[0] != OP_LOADF || (cs[i].arg[0] & 0x300) || monos_all[i].size() != 1 || cs[i].dtype != SO_F64)
                ga = false;
    }
    if (std::getenv("SIGOPS_DEBUG_PLAN") && !cs.empty())
        std::fprintf(stderr, "[sigops] carriers=%zu allow_ga=%d dtype=%d nsteps=%d op=%d arg=%#x monos=%zu -> ga=%d\n", cs.size(), (int)allow_ga,
                     cs[0].dtype, cs[0].nsteps, cs[0].op[0], cs[0].arg[0], monos_all[0].size(), (int)ga);
    for (auto& c : cs)
        if (!ga && c.dtype == SO_F32 && c.nsteps > 0 && (c.array_node >= 0 || c.buf >= 0) && !car_has_arr2(c)) { if (std::getenv("SIGOPS_DEBUG_PLAN")) std::fprintf(stderr, "[sigops] carrier fusion rejected (#%d)\n", 4); return false; }
    std::vector<std::vector<DOp>> fcodes(cs.size());
    size_t nops_total = 0;
    std::set<int> leafset;
    const size_t leaves_before = leaves.size();
    for (size_t i = 0; i < cs.size(); ++i) {
        DCarrier& c = cs[i];
        std::vector<DOp>& fcode = fcodes[i];
        int dmax = 2;
        c.nslots = (int)monos_all[i].size();
        for (size_t k = 0; k < monos_all[i].size(); ++k) {
            // closed form for the kernel's hot loop: a single leaf, optionally rounded to
            // Float32; compound per-frame expressions are not fused
            int ei = monos_all[i][k], r32 = 0;
            for (;;) {
                const Expr& ex = exprs[ei];
                if (ex.op == E_RETYPE) ei = ex.a;
                else if (ex.op == E_ROUND32) { r32 = 0x100; ei = ex.a; }
                // `0 + g` / `g + 0`: the zero-padded tail of a `Mix` operand under a generator (the
                // sum differs from g only in the sign of a zero)
                else if (ex.op == E_ADD && is_const(ex.a, 0.0)) ei = ex.b;
                else if ((ex.op == E_ADD || ex.op == E_SUB) && is_const(ex.b, 0.0)) ei = ex.a;  // (g - 0 == g exactly)
                else break;
            }
            const int eop = exprs[ei].op;
            const int kind = eop == E_CONST ? OP_CONST : eop == E_SCALAR ? OP_SCALAR : eop == E_FUNC ? OP_FUNC : eop == E_RAMP ? OP_RAMP : -1;
            if (kind < 0) {
                leaves.resize(leaves_before);
                leaf_array_node.resize(leaves_before);
                { if (std::getenv("SIGOPS_DEBUG_PLAN")) std::fprintf(stderr, "[sigops] carrier fusion rejected (#%d)\n", 5); return false; }
            }
            c.slot_leaf[k] = add_leaf(exprs[ei]);
            c.slot_kind[k] = kind | r32;
            leafset.insert(c.slot_leaf[k]);
            std::map<int, int> none;
            gen(monos_all[i][k], fcode, none, fcode, false);
            fcode.push_back(DOp{OP_STOREF, (int)k});
            dmax = std::max(dmax, depth(monos_all[i][k]));
        }
        c.depth = dmax;
        nops_total += fcode.size();
        for (auto& o : fcode)
            if (o.code <= OP_RAMP) leafset.insert(o.arg);
        // the in-kernel frame interpreter is the 2-deep one
        if (dmax > 2 || nops_total > (size_t)kCtlOps || leafset.size() > (size_t)kCtlLeaves) {
            leaves.resize(leaves_before);  // drop what gen() appended
            leaf_array_node.resize(leaves_before);
            { if (std::getenv("SIGOPS_DEBUG_PLAN")) std::fprintf(stderr, "[sigops] carrier fusion rejected (#%d)\n", 6); return false; }
        }
    }
    // commit
    for (size_t i = 0; i < cs.size(); ++i) {
        cs[i].frame_pc = (int)ops.size();
        cs[i].frame_len = (int)fcodes[i].size();
        ops.insert(ops.end(), fcodes[i].begin(), fcodes[i].end());
    }
    if (ga) {
        // the generated pieces must be exactly carrier 0's gain
        auto same_leaf = [&](int a, int b) {
            const DLeaf &x = leaves[a], &y = leaves[b];
            return x.base == y.base && x.fstride == y.fstride && x.cstride == y.cstride && x.df == y.df && x.dc == y.dc &&
                   x.modn == y.modn && x.v0 == y.v0 && x.v1 == y.v1 && x.v2 == y.v2 && x.sf == y.sf && x.sc == y.sc &&
                   x.dtype == y.dtype && x.mode == y.mode && x.flag == y.flag && x.buf == y.buf;
        };
        for (size_t i = 1; i < cs.size(); ++i)
            if (cs[i].slot_kind[0] != cs[0].slot_kind[0] || !same_leaf(cs[i].slot_leaf[0], cs[0].slot_leaf[0])) ga = false;
        if (!ga) {
            leaves.resize(leaves_before);
            leaf_array_node.resize(leaves_before);
            ops.resize(ops.size() - nops_total);
            if (std::getenv("SIGOPS_DEBUG_PLAN")) std::fprintf(stderr, "[sigops] carrier fusion rejected (#%d)\n", 7);
            return false;
        }
        // the staging code copies the raw samples (1.0f for the generated pieces); the multiply
        // happens at the A operand
        for (size_t i = 0; i < cs.size(); ++i) {
            cs[i].pad_ = (i == 0 ? 1 : 2) + (ga_add ? 2 : 0);  // 1/2: multiply (generated pieces staged as 1.0f), 3/4: add (as 0.0f)
            cs[i].nsteps = 0;
            cs[i].dtype = SO_F32;
        }
    }
    out = cs;
    return true;
}

// Kernel-argument control block of a periodic resampler stage: carriers with their frame
// programs and leaves re-indexed into the block (built per execute from the patched tables).
RsCtl Plan::make_ctl(const Stage& S) const {
    RsCtl ctl{};
    std::map<int, int> leafmap;
    for (const DCarrier& c0 : S.carriers) {
        if (ctl.ncar >= kCtlCar) throw PlanError{SO_ERR_RUNTIME, "internal: carrier control block overflow"};
        DCarrier c = c0;
        c.frame_pc = ctl.nops;
        for (int k = 0; k < c0.frame_len; ++k) {
            DOp o = ops[c0.frame_pc + k];
            if (o.code <= OP_RAMP) {
                auto it = leafmap.find(o.arg);
                if (it == leafmap.end()) {
                    if (ctl.nleaves >= kCtlLeaves) throw PlanError{SO_ERR_RUNTIME, "internal: leaf control block overflow"};
                    ctl.leaves[ctl.nleaves] = leaves[o.arg];
                    if (o.code == OP_SCALAR) ctl.leaves[ctl.nleaves].flag = 0;  // (a COPY of the leaf: nobody patches its v0, RmsPatch)
                    it = leafmap.emplace(o.arg, ctl.nleaves++).first;
                }
                o.arg = it->second;
            }
            if (ctl.nops >= kCtlOps) throw PlanError{SO_ERR_RUNTIME, "internal: op control block overflow"};
            ctl.ops[ctl.nops++] = o;
        }
        for (int k = 0; k < c0.nslots; ++k) {
            auto it = leafmap.find(c0.slot_leaf[k]);
            if (it == leafmap.end()) {
                if (ctl.nleaves >= kCtlLeaves) throw PlanError{SO_ERR_RUNTIME, "internal: leaf control block overflow"};
                ctl.leaves[ctl.nleaves] = leaves[c0.slot_leaf[k]];
                if ((c0.slot_kind[k] & 0xff) == OP_SCALAR) ctl.leaves[ctl.nleaves].flag = 0;
                it = leafmap.emplace(c0.slot_leaf[k], ctl.nleaves++).first;
            }
            c.slot_leaf[k] = it->second;
        }
        ctl.car[ctl.ncar++] = c;
    }
    return ctl;
}


// ---------------------------------------------------------------------------
// Resampler -> IIR: fold the IIR's state pass into the resampler.
// The three-pass K2 reads its input twice; when that input is the output of a periodic resampler
// stage and nothing else reads it, the first read can go: a chunk's zero-state end state is linear
// in the resampler's INPUT,  v = sum_r G[r] y[r],  y[r] = sum_k Tap_r[k] x[j_r - k]
//                              = sum_i W[i] x[i],   W = G . Tap   (D x window of one period),
// and the resampler has that window staged in LDS anyway.  Two of its loader waves become state
// waves (k_resample_periodic): one MFMA pass over the window per period row, written as
// vper[ch][period][16]; K2 then combines pt periods into a chunk (k_sos_combine), scans and runs
// its output pass.  (Reference: the same filt! at src/filters.jl:252-255; values differ from the
// sequential recurrence by rounding of the start states, ~1e-16 relative.)
// Opt-in (SIGOPS_FUSE_STATE=1).  Measured on the north-star pipeline (28.8 M x 8, order 10): K2 1.13 ->
// 0.90 ms and its traffic 5.57 -> ~4.1 GB, but the two state waves' 48 MFMAs per tile are the
// resampler's critical path (tile period 9 800 -> 12 200 cycles): K3 0.72 -> 0.90 ms.  Break-even
// (1.79 vs 1.76-1.83 ms), so the three-pass form stays the default.
// Independent IIR stages of one shape share their launches.  The scenes under an `Append` (reference
// src/appending.jl:59-76: every child is evaluated on its own, with its own filter state) are filters of
// a few hundred workgroups each; three launches per scene each ramp up and drain on their own, and eight
// stream lanes hide only part of that (config 4, 64 scenes: 2.5 ms).  Members: three-pass cascades of one
// group that read a device array directly -- they wait for nothing, so the batch can run first.
void Plan::batch_sos_stages() {
    if (std::getenv("SIGOPS_SOS_NOBATCH")) return;
    std::map<std::pair<int, int>, std::vector<int>> kinds;
    std::vector<int> order;
    for (size_t i = 0; i < stages.size(); ++i)
        if (stages[i].need > 0) order.push_back((int)i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return stages[a].node < stages[b].node; });
    for (int sid : order) {
        const Stage& S = stages[sid];
        // (not below a Normpower: such a filter keeps every chunk in its scan, or the exact block scan, by its OWN
        //  chunk count -- sos_chunking -- which the batch would change)
        if (S.kind != ST_SOS || S.onepass || S.sg.exact || S.xscan || S.under_norm || S.pre_stage >= 0 || S.groups.size() != 1 ||
            S.in_array_node < 0 || S.pw_step >= 0 || S.sg.nchunks < 1 || S.rsos_src >= 0)
            continue;
        kinds[{S.groups[0].nsec, nodes[S.node].dtype}].push_back(sid);
    }
    for (auto& kv : kinds) {
        if (kv.second.size() < 2) continue;
        SosBatch B;
        B.members = kv.second;
        B.nsec = kv.first.first;
        B.dtype = kv.first.second;
        // the members fill the machine TOGETHER: chunks for kSosSequences sequences over all of them, not per member --
        // 64 two-channel scenes cut for themselves are 5.3 M sequences of 64 frames, and at that length the scan
        // (K = 64 terms of 64 multiply-adds per sequence) costs more than the filter it serves
        int64_t chans = 0;
        for (int sid : B.members) chans += stages[sid].sg.nch;
        if (!std::getenv("SIGOPS_SOS_BATCH_KEEPCHUNKS"))
            for (int sid : B.members) {
                Stage& S = stages[sid];
                sos_chunking(sid, S.sg.n, S.sg.nch, nodes[S.node].dtype, S.groups, false, std::max<int64_t>(1, kSosSequences / chans));
            }
        B.desc_buf = raw_buf((B.members.size() + 1) * sizeof(SosDesc));
        {
            int off = 0;
            for (int sid : B.members) {
                B.bad_off.push_back(off);
                off += stages[sid].sg.nch;
            }
            B.bad_buf = raw_buf((size_t)std::max(off, 1) * 4);
        }
        for (int sid : B.members) stages[sid].batch = (int)batches.size();
        if (std::getenv("SIGOPS_DEBUG_PLAN"))
            std::fprintf(stderr, "[sigops] %zu IIR stages of %d sections share their launches\n", B.members.size(), B.nsec);
        batches.push_back(std::move(B));
    }
}

// ---------------------------------------------------------------------------
// Resampler -> IIR in one kernel (k_rsos.hip).  The reference never materialises the resampled signal: after
// the rewrite src/filters.jl:143-148 the IIR pulls its resampling child block by block (src/filters.jl:240-255).
// When an SOS stage is the only reader of a periodic resampler stage, both run as ONE launch: the signal is cut
// into time ranges, 16 rows (ranges x channels) per workgroup, every range warm-started wp periods early from
// zero state, the cascade in block state-space form on the matrix cores (see the kernel's header).
// Values: the resampled samples are K3's (same products, same order); the IIR's differ from the sequential
// recurrence by the rounding of a different association, ~1e-14 norm-wise for cascades that pass the planner's
// conditioning probes (the others never get here: they run k_sos_exact).
static void rsos_block_matrices(const SosCoefs& cf, std::vector<double>& mats) {
    const int ns = cf.nsec, D = 2 * ns, B = 16;
    // columns of [T C; D A^16]: the DF2T recurrence over one block from a unit input / a unit state
    std::vector<double> Tm(B * B, 0.0), Cm(B * 12, 0.0), Dm(12 * B, 0.0), Am(12 * 12, 0.0);
    for (int col = 0; col < B + D; ++col) {
        std::vector<double> st(D, 0.0);
        if (col >= B) st[col - B] = 1.0;
        for (int t = 0; t < B; ++t) {
            double v = col == t ? 1.0 : 0.0;
            for (int f = 0; f < ns; ++f) {
                const double xi = v;
                v = st[2 * f] + cf.b0[f] * xi;
                st[2 * f] = st[2 * f + 1] + cf.b1[f] * xi - cf.a1[f] * v;
                st[2 * f + 1] = cf.b2[f] * xi - cf.a2[f] * v;
            }
            v *= cf.gain;
            if (col < B) Tm[t * B + col] = v;
            else Cm[t * 12 + (col - B)] = v;
        }
        for (int d = 0; d < D; ++d) {
            if (col < B) Dm[d * B + col] = st[d];
            else Am[d * 12 + (col - B)] = st[d];
        }
    }
    // MFMA operand form: every matrix as Mat[lane & 15][4 v + (lane >> 4)] per k-step v
    mats.assign((size_t)14 * 64, 0.0);
    for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kq = lane >> 4;
        for (int v = 0; v < 4; ++v) {
            const int k = 4 * v + kq;
            mats[(size_t)(0 + v) * 64 + lane] = i < 12 ? Dm[i * B + k] : 0.0;  // D[state i][time k]
            mats[(size_t)(7 + v) * 64 + lane] = Tm[i * B + k];                 // T[time i][time k]
            if (v < 3) {
                mats[(size_t)(4 + v) * 64 + lane] = i < 12 ? Am[i * 12 + k] : 0.0;  // A^16[state i][state k]
                mats[(size_t)(11 + v) * 64 + lane] = Cm[i * 12 + k];                // C[time i][state k]
            }
        }
    }
}

void Plan::fuse_resample_sos() {
    if (std::getenv("SIGOPS_NO_RSOS")) return;
    auto env_int = [](const char* name, int dflt) {
        const char* ev = std::getenv(name);
        return ev ? std::atoi(ev) : dflt;
    };
    for (size_t i2 = 0; i2 < stages.size(); ++i2) {
        Stage& S2 = stages[i2];
        if (S2.kind != ST_SOS || S2.onepass || S2.need <= 0 || S2.groups.size() != 1 || S2.groups[0].nsec > 6 ||
            S2.in_buf < 0 || S2.in_array_node >= 0 || S2.in_offset != S2.base || S2.pw_step >= 0 || S2.sg.exact || (S2.xscan && !S2.under_norm) ||
            (S2.under_norm && (S2.norm_df != 0 || S2.base != 0 || std::getenv("SIGOPS_NORM_EXACT_FILT"))) || S2.src_op || S2.batch >= 0 || S2.pre_stage >= 0 ||
            (nodes[S2.node].dtype != SO_F64 && nodes[S2.node].dtype != SO_F32) || S2.in_frames != S2.need - S2.base)
            continue;  // (below a Normpower: only where its region starts at the filter's first frame -- fuse_plain_sos, Stage::norm_df)
        // A Float32 signal all the way (Float32 source, resampler and filter stages): the kernel rounds the resampled values to
        // Float32 where the reference's resampler stores them (RsSos::x32) and stores a Float32 result; its sources are plain
        // Float32 arrays or buffers (a Float32 x Float32 step rounds too: K1 materialises it)
        const bool pure32 = nodes[S2.node].dtype == SO_F32;
        if (pure32 && std::getenv("SIGOPS_RSOS_NO32")) continue;
        if (S2.base > 0 && std::getenv("SIGOPS_RSOS_NOWINDOWS")) continue;
        int i3 = -1;
        for (size_t j = 0; j < stages.size(); ++j)
            if (stages[j].kind == ST_RESAMPLE && stages[j].out_buf == S2.in_buf) i3 = (int)j;
        if (i3 < 0 || i3 == alias_stage || stages[i3].win_off >= 0) continue;
        Stage& S3 = stages[i3];
        const RsPeriodic& rp = S3.rp;
        // A window (After, a block of so.stream, a rank's time range): both stages are warm-started -- the cascade a decay
        // time before the window (S2.base), the resampler a few periods before that (S3.base, whole periods).  The fused
        // kernel's coordinates are the resampler stage's: its output 0 is frame S3.base, the cascade starts from rest
        // there (earlier than the cascade stage alone would: a longer warm-up), and nothing below S2.base is stored.
        if (S3.base > S2.base) continue;
        if (rp.nstate || nodes[S3.node].dtype != nodes[S2.node].dtype || !S3.fix_host.empty() || S3.need < S2.need || (int)S3.carriers.size() > kCtlCar)
            continue;
        // (arr2: a source of two arrays -- K3's A2 instantiation's alone; this kernel's loader takes the second one for groups of
        //  two, four or eight channels, Float64, ONE carrier whose one step is `v (op) y`: k_rsos.hip, rsos_loader's A2)
        // ... also where K1 materialises the map for K3 (Stage::alt_carriers): tried with the two-array carrier in place of the
        // materialised buffer's; whatever makes this stage `continue` below puts K1's step and the buffer's carrier back
        struct AltGuard {
            Stage& S;
            std::vector<DCarrier> car;
            int pw;
            bool armed;
            ~AltGuard() {
                if (armed) {
                    S.carriers = car;
                    S.pw_step = pw;
                }
            }
        } alt{S3, S3.carriers, S3.pw_step, false};
        if (!rp.arr2 && !S3.alt_carriers.empty() && S3.pw_step >= 0 && S3.in_buf >= 0 && !std::getenv("SIGOPS_RSOS_NO_ARR2")) {
            S3.carriers = S3.alt_carriers;
            S3.pw_step = -1;
            alt.armed = true;
        }
        bool two_arrays = false;
        if (rp.arr2 || alt.armed) {
            const bool ok = !pure32 && nodes[S2.node].dtype == SO_F64 && S3.carriers.size() == 1 && S3.carriers[0].nsteps == 1 &&
                            (S3.carriers[0].arg[0] & kCarArr2) && !(S3.carriers[0].arg[0] & 0x200) && S3.carriers[0].dtype == SO_F64 &&
                            S3.carriers[0].dtype2 == SO_F64 && nodes[S2.node].nch % 2 == 0 &&
                            (S3.carriers[0].op[0] == OP_ADD || S3.carriers[0].op[0] == OP_SUB || S3.carriers[0].op[0] == OP_MUL) &&
                            !std::getenv("SIGOPS_RSOS_NO_ARR2");
            if (!ok) continue;
            two_arrays = true;
        }
        if (pure32) {
            bool plain = nodes[nodes[S3.node].kids[0]].dtype == SO_F32 && rp.ga == 0;
            for (auto& c : S3.carriers)
                if (c.nsteps != 0 || c.pad_ != 0 || c.dtype != SO_F32) plain = false;
            if (!plain) continue;
        }
        // (a stage that does not run the periodic kernel -- long super-periods: x 2 of eight channels -- has no carriers:
        //  its plain source, an array or a stage buffer, becomes one below)
        const bool plain_src = S3.carriers.empty() && (S3.in_buf >= 0 || S3.in_array_node >= 0);
        if (S3.carriers.empty() && !plain_src) continue;
        // The period the kernel walks: blocks of 16 outputs, each with its own [k-steps x 16] tap operand.  The resampler
        // stage's own (44.1 -> 48 kHz: 160 outputs, ten blocks) where it splits into whole blocks -- or, for the small
        // rational ratios (x 2, x 3, x 3/2 ...: DSP.jl's FIRInterpolator / FIRRational, reference src/reformatting.jl:103-111)
        // whose stage runs on super-periods sized for K3's tiles, the shortest super-period of whole blocks, with a tap
        // table of this stage's own.
        int64_t Lp = rp.L, Mp = rp.M;
        int ngp = rp.ngroups, kwp = rp.kw, jlop = rp.jlo;
        bool own_tab = false;
        std::vector<double> own_taps;
        std::vector<int> own_jend;
        std::vector<int64_t> own_jr;  // (newest input of every output of the own table's super-period)
        const bool blocks_ok = S3.periodic && Lp > 0 && Lp % 16 == 0 && (int64_t)ngp * 16 == Lp && ngp <= 256;
        if (!S3.rg.arbitrary && (!blocks_ok || ngp > 10) && !std::getenv("SIGOPS_RSOS_NOOWNTAB")) {
            const RsGeom& rg = S3.rg;
            const so_node_t& nd3 = nodes[S3.node].nd;
            const double* h = (const double*)nd3.p0;
            const int hlen = nd3.i2;
            const int64_t tmin = 16 / std::__gcd<int64_t>(rg.L, 16);
            bool found = false;
            for (int64_t m : {1, 2, 3, 5}) {
                const int64_t Ls = rg.L * tmin * m, Ms = rg.M * tmin * m;
                const int ng = (int)(Ls / 16);
                if (Ls > 4096 || (ng != 1 && ng != 2 && ng != 3 && ng != 5 && ng != 6 && ng != 10)) continue;
                std::vector<int64_t> jr((size_t)Ls);
                std::vector<int> prr((size_t)Ls);
                for (int64_t r = 0; r < Ls; ++r) {
                    const int64_t qi = rg.c0i + r * rg.M;  // (rational: nphi == L, no interpolation between phases)
                    jr[(size_t)r] = qi / rg.nphi;
                    prr[(size_t)r] = (int)(qi % rg.nphi);
                }
                std::vector<int> jend((size_t)ng);
                int64_t maxspan = 0;
                for (int gi = 0; gi < ng; ++gi) {
                    jend[(size_t)gi] = (int)jr[(size_t)gi * 16 + 15];
                    maxspan = std::max(maxspan, jr[(size_t)gi * 16 + 15] - jr[(size_t)gi * 16]);
                }
                const int ksneed = (int)((rg.taps + maxspan + 3) / 4);
                int ksc = 0;
                for (int k : {12, 14, 16, 20})
                    if (!ksc && k >= ksneed) ksc = k;
                if (std::getenv("SIGOPS_DEBUG_PLAN"))
                    std::fprintf(stderr, "[sigops] k_rsos own taps: L %lld M %lld x %lld -> %d blocks, taps %d span %lld: %d k-steps (%d)\n",
                                 (long long)rg.L, (long long)rg.M, (long long)(tmin * m), ng, rg.taps, (long long)maxspan, ksneed, ksc);
                if (!ksc) break;  // (a longer super-period does not shorten the windows)
                const int kw = 4 * ksc;
                int jlo = jend[0] - (kw - 1);
                jlo -= ((jlo % 4) + 4) % 4;
                std::vector<double> tab((size_t)ng * kw * 16, 0.0);
                for (int gi = 0; gi < ng; ++gi)
                    for (int rr = 0; rr < 16; ++rr)
                        for (int kk = 0; kk < kw; ++kk) {
                            const int64_t r = (int64_t)gi * 16 + rr;
                            const int64_t age = jr[(size_t)r] - ((int64_t)jend[(size_t)gi] - (kw - 1) + kk);
                            if (age < 0 || age >= rg.taps) continue;
                            const int64_t hi = prr[(size_t)r] + (int64_t)rg.nphi * age;
                            tab[((size_t)gi * kw + kk) * 16 + rr] = hi < hlen ? h[hi] : 0.0;
                        }
                Lp = Ls;
                Mp = Ms;
                ngp = ng;
                kwp = kw;
                jlop = jlo;
                own_taps.swap(tab);
                own_jend.swap(jend);
                own_jr = jr;
                own_tab = found = true;
                break;
            }
            (void)found;
        }
        if (!own_tab && (!blocks_ok || !S3.periodic)) continue;
        if (S3.base % Lp != 0 || Mp >= (1 << 20)) continue;
        // The resampler stage's table has the k-steps of K3's instantiations (12, 14, 16 ...); its windows end at the
        // group's newest input, so what a group does not need are its OLDEST slots.  Where every group's first slots are
        // zero the fused kernel takes a shorter window from the same table -- 44.1 -> 48 kHz: 38 taps + a span of 14
        // inputs are 52 = 13 k-steps, not 14: 27 MFMAs per block instead of 28 (same products, same order: the dropped
        // ones were multiplications by zero).  The staged range (ulo) stays the stage's.
        if (!own_tab && !std::getenv("SIGOPS_RSOS_NOTRIM") && !S3.tab_host.empty() && (size_t)ngp * kwp * 16 == S3.tab_host.size()) {
            int lead = kwp;  // leading all-zero slots common to all groups
            for (int gi = 0; gi < ngp && lead > 0; ++gi)
                for (int kk = 0; kk < lead; ++kk) {
                    bool z = true;
                    for (int rr = 0; rr < 16 && z; ++rr) z = S3.tab_host[((size_t)gi * kwp + kk) * 16 + rr] == 0.0;
                    if (!z) {
                        lead = kk;
                        break;
                    }
                }
            int ksc = 0;
            for (int k : {12, 13, 14, 16, 20})
                if (!ksc && 4 * k >= kwp - lead) ksc = k;
            if (ksc && 4 * ksc < kwp) {
                const int drop = kwp - 4 * ksc, kw = 4 * ksc;
                own_taps.assign((size_t)ngp * kw * 16, 0.0);
                for (int gi = 0; gi < ngp; ++gi)
                    for (int kk = 0; kk < kw; ++kk)
                        for (int rr = 0; rr < 16; ++rr)
                            own_taps[((size_t)gi * kw + kk) * 16 + rr] = S3.tab_host[((size_t)gi * kwp + kk + drop) * 16 + rr];
                own_jend = S3.jend_host;
                kwp = kw;
                own_tab = true;
                if (std::getenv("SIGOPS_DEBUG_PLAN"))
                    std::fprintf(stderr, "[sigops] k_rsos: %d leading zero slots in every group's window: %d k-steps instead of %d\n", lead, ksc,
                                 (kw + drop) / 4);
            }
        }
        const int ks = kwp / 4;
        if (!(ks == 12 || ks == 13 || ks == 14 || ks == 16 || ks == 20)) continue;  // (the window lengths k_rsos is instantiated for)
        bool ok = true;
        // GA carriers (a Float32 array whose Float64 gain or summand K3's compute waves apply at the MFMA operand): this
        // kernel's loader widens the landed Float32 chunk in place and applies the step on the way, so the carriers go
        // back to their plain form below -- array + one step, generated pieces that load the operand
        const bool was_ga = rp.ga != 0;
        for (auto& c : S3.carriers)
            if ((c.pad_ != 0) != was_ga) ok = false;
        if (was_ga && std::getenv("SIGOPS_RSOS_NO32")) ok = false;
        // nothing else may read the intermediate
        for (auto& L : leaves)
            if (L.buf == S2.in_buf) ok = false;
        for (size_t j = 0; j < stages.size(); ++j) {
            if (j != i2 && stages[j].in_buf == S2.in_buf) ok = false;
            if (j != i2)  // (the filter's own carrier over that buffer, a candidate of fuse_plain_sos, goes with it below)
                for (auto& c : stages[j].carriers)
                    if (c.buf == S2.in_buf) ok = false;
        }
        if (!ok) continue;
        const SosCoefs& cf = S2.groups[0];
        const int D = 2 * cf.nsec;
        const int nch = nodes[S2.node].nch;
        const int64_t need = S2.need - S3.base, L = Lp;
        const int64_t store_lo = S2.base - S3.base;
        const int64_t nperiods = (need + L - 1) / L;
        // warm-up: the first wp with ||A^(wp L)|| < 2^-56: what the frames in front of the warm-up leave in the state is an
        // eighth of the state's own rounding unit (2^-53) by the time the range's first output is stored -- below anything
        // the 50-term MFMA sums of a block can resolve.  (Round 4 cut at 2^-70 like the warm starts of windows, whose
        // results are compared with cold starts at 1e-13: 27 periods instead of 22 for the headline's band-stop, 1.3 % of
        // all blocks.  SIGOPS_RSOS_WTOL: another exponent.)
        int64_t wp = 1;
        {
            const double tol = std::ldexp(1.0, -std::abs(env_int("SIGOPS_RSOS_WTOL", S2.under_norm ? 70 : 56)));  // (below a Normpower: the plain form's cut)
            const Mat P = matpow(sos_state_matrix(cf), L, D);
            Mat cur = P;
            while (!(maxabs(cur) < tol) && wp < 1000000 && std::isfinite(maxabs(cur))) {
                cur = matmul(cur, P, D);
                ++wp;
            }
            if (!(maxabs(cur) < tol)) continue;
        }
        // rows of a sequence group: ct channels x rgs ranges
        int ct = 1;
        for (int c : {16, 8, 4, 2})
            if (nch % c == 0) {
                ct = c;
                break;
            }
        const int rgs = 16 / ct;
        const int64_t ncg = nch / ct;
        // ranges: one sequence group per CU (the last wave of groups of a longer grid would run on a draining machine),
        // fewer where a range would otherwise be shorter than its own warm-up
        const int cus = env_int("SIGOPS_RSOS_GRID", 256);
        int64_t rgroups = std::max<int64_t>(1, cus / ncg);  // range groups per channel group
        int64_t nranges = rgroups * rgs;
        const int64_t min_pr = std::max<int64_t>(1, wp);
        if (nperiods / nranges < min_pr) nranges = std::max<int64_t>(rgs, nperiods / min_pr / rgs * rgs);
        if (const char* ev = std::getenv("SIGOPS_RSOS_RANGES")) nranges = std::max<int64_t>(1, std::atoll(ev));
        int64_t pr = (nperiods + nranges - 1) / nranges;
        // groups of fewer than eight channels walk several ranges side by side: where a range is a multiple of 16 input frames long
        // all of them stage their chunks at the same alignment and the loader addresses its units by scalar additions
        // (k_rsos.hip, the loader's `klo_all`): a few periods more per range, fewer ranges
        if (ct < 8 && !std::getenv("SIGOPS_RSOS_NOALIGNPR")) {
            const int64_t mlt = 16 / std::__gcd<int64_t>(Mp % 16 == 0 ? 16 : Mp % 16, 16);
            if (pr % mlt != 0 && pr >= 8 * mlt) pr = (pr + mlt - 1) / mlt * mlt;
        }
        nranges = (nperiods + pr - 1) / pr;
        const int64_t ngrp = ncg * ((nranges + rgs - 1) / rgs);
        if ((pr + wp) * ngp >= (1 << 30) || (pr + wp) * Mp >= ((int64_t)1 << 30)) continue;
        // Worth it?  A workgroup walks its (pr + wp) periods block by block -- 0.30 us per block of 16 outputs x 16 rows
        // with 14 k-step windows -- however few workgroups there are, while the two kernels use the whole chip for any
        // length: 8 ps per output sample + 95 us of launches, scans and tails (tools/rsos_len_sweep.sh, 8 channels at
        // 44.1 -> 48 kHz: fused 0.176 / 0.189 / 0.241 / 0.466 ms for 20 / 46 / 75 / 200 s, two kernels 0.159 / 0.226 /
        // 0.306 / 0.710).  Short signals and short windows keep the two kernels (8 channels: below ~30 s).
        // SIGOPS_RSOS_MINGROUPS=n replaces the estimate by "at least n sequence groups" (tests, measurements).
        if (const char* ev = std::getenv("SIGOPS_RSOS_MINGROUPS")) {
            if (ngrp < std::atoll(ev)) continue;
        } else {
            // (groups of fewer than eight channels have 4, 8 or 16 loader units per chunk instead of 2, each with its own
            //  lane-table reads next to the chain wave's MFMAs: the loader sets the pace there -- 1e8 samples: 0.65 ms
            //  for 8 channels per group, 0.78 / 1.01 / 2.07 for 4 / 2 (two loader waves) / 1 (tools/channel_matrix.py, tools/rsos_ct_abl.sh;
            //  an affine, table-free addressing of unaligned rows was built and is slower still) -- two kernels: 1.0)
            // (round 6, measured with the step waves and without the spilled store offsets: 4 ch 1.04 / 0.98 with / without a fused
            //  step against 0.95 for 8 ch; 2 ch 1.06 / 1.06)
            const double unit_cost = ct >= 8 ? 1.0 : ct == 4 ? 1.1 : ct == 2 ? 1.15 : 3.2;
            const double t_fused = (double)((ngrp + cus - 1) / cus) * (double)((pr + wp) * ngp) * 0.30 * unit_cost * (ks + 14) / 28.0 + 15.0;
            const double t_two = 8.0e-6 * (double)need * nch + 95.0;
            if (std::getenv("SIGOPS_DEBUG_PLAN"))
                std::fprintf(stderr, "[sigops] k_rsos estimate: fused %.0f us (%lld groups, %lld + %lld periods), two kernels %.0f us\n", t_fused,
                             (long long)ngrp, (long long)pr, (long long)wp, t_two);
            if (t_fused > t_two) continue;
        }
        RsSos g{};
        g.n_in = S3.rg.n_in;
        g.n_out = need;
        g.store_lo = store_lo;
        g.L = L;
        g.M = Mp;
        g.pr = pr;
        g.wp = (int32_t)wp;
        g.nranges = (int32_t)nranges;
        g.ngroups = ngp;
        g.ks = ks;
        g.ulo = jlop;
        g.ct = ct;
        g.rgs = rgs;
        g.nch = nch;
        g.chunk = env_int("SIGOPS_RSOS_CHUNK", 128) == 64 ? 64 : 128;
        g.depth = std::max(1, std::min(4, env_int("SIGOPS_RSOS_DEPTH", 4)));
        g.nsec = cf.nsec;
        g.debug = env_int("SIGOPS_RSOS_DEBUG", 0);
        // waves: one chain, one loader, nwaves - 2 y waves that take the blocks round robin.  A y wave keeps the taps of
        // the phase groups its blocks cycle through in registers where they fit (10 y waves and the 10 groups of
        // 44.1 -> 48 kHz: one group each); otherwise the tap table goes to LDS and the input ring shrinks
        {
            auto cyc_of = [&](int nw) { return g.ngroups / std::__gcd(nw >= 16 ? 10 : nw - 2, g.ngroups); };
            auto fits = [&](int nw) {
                const int c = cyc_of(nw);
                return nw == 12 ? ((c == 1 || c == 2) && c * ks <= 32) : ((c == 1 || c == 2 || c == 3 || c == 5) && c * ks <= 80);
            };
            int nw = fits(12) ? 12 : fits(8) ? 8 : 12;
            // (groups of two channels: eight loader units per chunk -- a second loader wave pays: 1.22 -> 1.01 ms for 1e8
            //  samples, tools/rsos_nw16.sh; for four and more channels it costs what it saves)
            // (round 6: that was with the 128-register instantiation's store offsets spilled -- every result store behind a scratch
            //  reload and a wait for the previous store: 0.3 ms of a stereo signal's 1.34, k_rsos.hip `ystep`.  Without the spills
            //  12 waves win where the loader has no step to apply (2 ch x 2400 s: 1.07 against 1.10 ms); with the fused step
            //  the second loader wave still pays, 1.46 against 1.72)
            //  ... and since the step has waves of its own (k_rsos.hip, MODE: the two loaders issue and wait, waves 13 / 14 apply the
            //  step and publish -- two stages of a pipeline instead of one wave's serial time per chunk) sixteen waves carry groups of
            //  two AND four channels with a fused step at the plain pipeline's pace: 2 ch 1.42 -> 1.06 ms, 4 ch 1.24 -> 1.04
            const bool stepped = was_ga || (!plain_src && !S3.carriers.empty() && S3.carriers[0].nsteps > 0);
            if ((ct == 2 || (ct == 4 && !was_ga && nodes[S2.node].dtype == SO_F64 && !std::getenv("SIGOPS_RSOS_NOGSPLIT"))) && stepped && nw == 12 &&
                cyc_of(16) == 1 && ks <= 16)
                nw = 16;
            // (the helper geometry: sixteen waves of 128 registers, ONE loader, and a wave on the chain's SIMD that forms D . X for
            //  three of the y waves -- k_rsos.hip, NW = 17: 68 / 68 / 68 / 66 MFMAs per round on the four SIMDs instead of 72 / 72 / 72 / 54)
            // MEASURED (round 6) AND NOT THE DEFAULT: 0.978 ms against 0.978 for the plain pipeline, 1.08 against 1.00 with the fused
            // Mix (the loader at 128 registers; four waves on the chain's SIMD) -- SIGOPS_RSOS_NWAVES=17 selects it
            if (const char* ev = std::getenv("SIGOPS_RSOS_NWAVES")) nw = std::atoi(ev) == 8 ? 8 : (std::atoi(ev) == 16 || std::atoi(ev) == 17) && cyc_of(16) == 1 && ks <= 16 ? std::atoi(ev) : 12;
            // (a source of two arrays: the second one goes through the registers of the sixteen-wave geometry's two step waves)
            if (two_arrays) {
                if (!(cyc_of(16) == 1 && ks <= 16)) continue;
                nw = 16;
            }
            g.nwaves = nw;
            g.cyc = (nw >= 16 || fits(nw)) && !std::getenv("SIGOPS_RSOS_LDSTAPS") ? cyc_of(nw) : 0;
            // (waves, groups per wave, window) as launch_rsos_t instantiates them -- a combination it has not (the 16-wave
            //  geometry with its taps in LDS, SIGOPS_RSOS_LDSTAPS) keeps the two kernels HERE instead of failing the execute
            const bool inst = nw == 12 ? (g.cyc == 0 || g.cyc == 1 || (g.cyc == 2 && ks <= 16))
                            : nw >= 16 ? g.cyc == 1
                                       : (g.cyc == 0 || g.cyc == 1 || g.cyc == 2 || (g.cyc == 3 && ks <= 20) || (g.cyc == 5 && ks <= 16));
            if (!inst) continue;
        }
        // input ring: as large as fits next to the tap table (if any) and the exchange slots
        {
            const int ny = g.nwaves >= 16 ? 10 : g.nwaves - 2;
            // (a y wave's front part runs up to one of its own blocks ahead: the windows in use span 2 ny - 1 blocks)
            const int64_t span = (int64_t)(2 * ny - 1) * ((16 * Mp + L - 1) / L + 1) + kwp + 16 + 2 * g.chunk;
            int ring = 4096;  // (a multiple of 128: whole chunks, and rows of ring + 2 doubles fall on different banks)
            while (ring >= 128 && rsos_lds_bytes(g.ngroups, ks, ring + 2, g.nwaves, g.cyc) > rsos_lds_budget()) ring -= 128;
            if (const char* ev = std::getenv("SIGOPS_RSOS_RING")) ring = std::min(ring, std::max(128, std::atoi(ev) / 128 * 128));
            if (std::getenv("SIGOPS_DEBUG_PLAN"))
                std::fprintf(stderr, "[sigops] k_rsos: ring %d frames (needs %lld), LDS %zu of %zu\n", ring, (long long)span,
                             rsos_lds_bytes(g.ngroups, ks, ring + 2, g.nwaves, g.cyc), rsos_lds_budget());
            if (ring < 128 || ring < span) continue;
            g.ring = ring;
            g.rpitch = ring + 2;
        }
        if (plain_src) {
            DCarrier c{};
            c.a = 0;
            c.b = S3.in_frames;
            c.dtype = nodes[nodes[S3.node].kids[0]].dtype == SO_F32 ? SO_F32 : SO_F64;
            c.array_node = S3.in_array_node;
            c.buf = S3.in_array_node >= 0 ? -1 : S3.in_buf;
            c.df = S3.in_offset;
            c.cstride = S3.in_array_node >= 0 ? (nch == 1 ? 0 : S3.in_pitch) : -1;  // -1: buffer pitch
            S3.carriers.push_back(c);
            S3.car_buf = raw_buf(sizeof(DCarrier));
            S3.ctl_buf = raw_buf(sizeof(RsCtl));
        }
        if (was_ga) {
            for (size_t i = 0; i < S3.carriers.size(); ++i) {
                DCarrier& c = S3.carriers[i];
                c.pad_ = 0;
                c.nsteps = 1;  // (op[] / arg[] were left as build_carriers made them: MUL / ADD with slot 0, LOADF for the pieces)
                if (i > 0) c.dtype = SO_F64;
            }
            S3.rp.ga = 0;
            g.src32 = 1;
        } else
            g.src32 = S3.carriers[0].dtype == SO_F32 && S3.carriers[0].nsteps == 0 ? 1 : 0;
        g.x32 = pure32 ? 1 : 0;
        // carrier 0's step on the fast path
        {
            const DCarrier& c0 = S3.carriers[0];
            g.fuse = -1;
            if (two_arrays) {
                g.fuse = c0.op[0] == OP_MUL ? 0 : c0.op[0] == OP_ADD ? 1 : ((c0.arg[0] & 0x100) ? 3 : 2);
                g.fuse_sine = 0;
                g.arr2 = 1;
            } else if (c0.nsteps == 1 && (c0.arg[0] & 0x2ff) == 0 && c0.nslots >= 1 &&
                (c0.op[0] == OP_MUL || c0.op[0] == OP_ADD || c0.op[0] == OP_SUB)) {
                g.fuse = c0.op[0] == OP_MUL ? 0 : c0.op[0] == OP_ADD ? 1 : ((c0.arg[0] & 0x100) ? 3 : 2);
                const DLeaf& L0 = leaves[c0.slot_leaf[0]];
                const int kind = c0.slot_kind[0];
                if (kind == OP_FUNC && L0.mode == SO_FN_SIN && L0.sf == 1) g.fuse_sine = 1;
                else if (kind == OP_CONST || kind == OP_SCALAR) g.fuse_sine = 0;
                else g.fuse = -2;
            } else if (c0.nsteps > 0)
                g.fuse = -2;  // (every chunk takes the general staging path)
        }
        g.gsplit = g.nwaves == 16 && !g.src32 && g.fuse >= 0 && (g.arr2 || ((ct == 2 || ct == 4) && !std::getenv("SIGOPS_RSOS_NOGSPLIT"))) ? 1 : 0;
        g.ring32 = g.src32 && g.fuse == -1 && pure32 && !std::getenv("SIGOPS_RSOS_NO_RING32") ? 1 : 0;  // (no step: the ring keeps the Float32 samples)
        // A Float32 array -- with or without the fast path's one step -- into a Float32 RESULT (known at execute time: the stage
        // may turn out to write the sink's Float32 buffer itself, Plan::alias_narrow): the ring keeps Float32 samples, the step is
        // done on them, and the resampling product runs on the Float32 MFMA (k_rsos.hip, F32M; taps in registers only).  What
        // reaches the cascade is the Float32-rounded resampler output of a Float32 signal (reference src/filters.jl:105) -- for
        // a Float64 signal made of a Float32 array and a Float64 generator it is that signal rounded to Float32 on its way INTO
        // the resampler instead of on its way into the Float32 result: inside the 1e-6 contract (tools/soak_rsos_f32m.py,
        // profiles/r06/relerr_maxima_rsos_f32m.json: worst 1.5e-7; gate 3e-7), SIGOPS_RSOS_NO_F32MFMA=1 keeps the Float64 products.
        g.help = g.nwaves == 17 ? 1 : 0;
        g.f32m = g.src32 && g.fuse >= -1 && g.cyc > 0 && !std::getenv("SIGOPS_RSOS_NO_F32MFMA") && !std::getenv("SIGOPS_RSOS_NO_RING32") ? 1 : 0;
        S2.rs = g;
        S2.rsos_src = i3;
        S2.carriers.clear();  // (what process_stage prepared for the single-pass form of a plain filter: the resampler's serve now)
        {  // what k_rsos_fixup needs to recompute single outputs the reference's way
            const std::vector<int64_t>& jr = own_jr.empty() ? S3.per_j : own_jr;
            const std::vector<int>& je = own_tab ? own_jend : S3.jend_host;
            S2.rsos_jrel_host.clear();
            if ((int64_t)jr.size() >= Lp && (int64_t)je.size() >= ngp) {
                S2.rsos_jrel_host.resize((size_t)Lp);
                for (int64_t r = 0; r < Lp; ++r) S2.rsos_jrel_host[(size_t)r] = (int)(jr[(size_t)r] - je[(size_t)(r / 16)]);
                S2.rsos_jrel_buf = raw_buf(S2.rsos_jrel_host.size() * 4);
                S2.rsos_taps = S3.rg.taps;
            }
        }
        if (own_tab) {
            S2.rsos_tab_host.swap(own_taps);
            S2.rsos_jend_host.swap(own_jend);
            S2.rsos_tab_buf = raw_buf(S2.rsos_tab_host.size() * 8);
            S2.rsos_jend_buf = raw_buf(S2.rsos_jend_host.size() * 4);
        }
        S2.rsos_grid = (int)std::min<int64_t>(ngrp, cus);
        rsos_block_matrices(cf, S2.rsos_mats_host);
        S2.rsos_mats_buf = raw_buf(S2.rsos_mats_host.size() * 8);
        if (S2.bad_buf < 0) S2.bad_buf = raw_buf((size_t)nch * 4);  // first range per channel that ended in a non-finite state
        S3.fused_away = true;
        if (alt.armed) {  // (K1's copy of `x (op) y` is never made: its step is not in the plan, its buffer holds nothing)
            alt.armed = false;
            bufs[S3.in_buf].bytes = 0;
        }
        if (std::getenv("SIGOPS_DEBUG_PLAN"))
            std::fprintf(stderr,
                         "[sigops] resampler + IIR fused (k_rsos): %lld ranges of %lld periods (+%lld warm-up), %lld groups of %d ch x %d ranges, "
                         "ks=%d ring=%d chunk=%d waves=%d cyc=%d fuse=%d/%d\n",
                         (long long)nranges, (long long)pr, (long long)wp, (long long)ngrp, ct, rgs, ks, g.ring, g.chunk, g.nwaves, g.cyc,
                         g.fuse, g.fuse_sine);
    }
}

// ---------------------------------------------------------------------------
// A plain `Filt` in one pass over HBM: the fused kernel's roles with an identity resampler (reference: the IIR's
// nextblock filters every block of its child once, src/filters.jl:240-255; K2's chunked scan reads its input twice).
// Period = ten blocks of 16 frames, every block's [16 x 16] tap operand the identity over the block's own 16 inputs
// (4 k-steps: X = I . Win, exact), so a block costs 4 + 4 + 4 + 3 MFMAs on its y wave and 3 on the chain wave against 27
// for the headline -- and the input is read once.  Conditions: process_stage built a carrier for the source (whole-signal
// filters over an array / a buffer, optionally plus or times one sine; one group of at most 6 sections; not the
// sequential / exact-scan / under-Normpower variants), and the estimate below.
void Plan::fuse_plain_sos() {
    if (std::getenv("SIGOPS_NO_RSOS") || std::getenv("SIGOPS_NO_PLAIN_RSOS")) return;
    auto env_int = [](const char* name, int dflt) {
        const char* ev = std::getenv(name);
        return ev ? std::atoi(ev) : dflt;
    };
    const int cus = env_int("SIGOPS_RSOS_GRID", 256);
    // One filter's geometry for this kernel.  gpm == 0: a launch of its own (ranges to fill the chip, the estimate against the
    // three passes decides); gpm > 0: as a member of a batched launch with gpm workgroups of its own (the caller decides for
    // the batch as a whole; cost: the microseconds one of its workgroups works).
    auto fit = [&](size_t i2, int gpm, RsSos& g, double& cost) -> bool {
        Stage& S2 = stages[i2];
        const int dt = nodes[S2.node].dtype;
        if (dt != SO_F64 && dt != SO_F32) return false;
        const DCarrier& c0 = S2.carriers[0];
        const bool pure32 = dt == SO_F32;  // (a plain Float32 source: the ring keeps its samples, RsSos::ring32)
        if (pure32 && (c0.nsteps != 0 || c0.dtype != SO_F32)) return false;
        if (!pure32 && c0.dtype != SO_F64) return false;
        const SosCoefs& cf = S2.groups[0];
        const int D = 2 * cf.nsec;
        const int nch = nodes[S2.node].nch;
        const int64_t need = S2.need, L = 160;
        const int ngp = 10, ks = 4, kw = 16;
        const int64_t nperiods = (need + L - 1) / L;
        // warm-up cut of the PLAIN filter: 2^-70, as the warm starts of windows (its warm-up blocks are cheap -- 18 MFMAs, no
        // resampling --, and this is the default path of any long Filt: a range that starts in near silence behind a loud passage
        // must not see what the cut dropped; the fused resampler + IIR form above cuts at 2^-56)
        int64_t wp = 1;
        {
            const double tol = std::ldexp(1.0, -std::abs(env_int("SIGOPS_PLAIN_WTOL", 70)));
            const Mat P = matpow(sos_state_matrix(cf), L, D);
            Mat cur = P;
            while (!(maxabs(cur) < tol) && wp < 1000000 && std::isfinite(maxabs(cur))) {
                cur = matmul(cur, P, D);
                ++wp;
            }
            if (!(maxabs(cur) < tol)) return false;
        }
        int ct = 1;
        for (int c : {16, 8, 4, 2})
            if (nch % c == 0) {
                ct = c;
                break;
            }
        const int rgs = 16 / ct;
        const int64_t ncg = nch / ct;
        if (gpm > 0 && ncg > gpm) return false;
        int64_t rgroups = gpm > 0 ? gpm / ncg : std::max<int64_t>(1, cus / ncg);
        int64_t nranges = rgroups * rgs;
        const int64_t min_pr = std::max<int64_t>(1, 4 * wp);  // (a range at least four warm-ups long: <= 25 % of the blocks)
        if (nperiods / nranges < min_pr) nranges = std::max<int64_t>(rgs, nperiods / min_pr / rgs * rgs);
        if (const char* ev = std::getenv("SIGOPS_RSOS_RANGES"))
            if (gpm == 0) nranges = std::max<int64_t>(1, std::atoll(ev));
        const int64_t pr = (nperiods + nranges - 1) / nranges;
        nranges = (nperiods + pr - 1) / pr;
        const int64_t ngrp = ncg * ((nranges + rgs - 1) / rgs);
        if ((pr + wp) * ngp >= (1 << 30) || (pr + wp) * L >= ((int64_t)1 << 30)) return false;
        const bool stepped_few = (ct == 2 || ct == 4) && !pure32 && c0.nsteps == 1 && !std::getenv("SIGOPS_RSOS_NOGSPLIT");
        const int plain_nw = env_int("SIGOPS_PLAIN_NWAVES", stepped_few ? 16 : 12) == 16 ? 16 : 12;
        // a block of this form: 0.197 us on its workgroup (15 + 3 MFMAs; the chain wave's step sets the pace; groups of two
        // channels, eight loader units per chunk: 1.3 x); the three passes: 4.4 ps per sample + 25 us up to 1e8 samples,
        // 3.4 ps + 110 us beyond (tools/iir_one_pass_probe.py, Float64 Lowpass: 12.5 M x 8 0.344 against 0.442 ms, 28.8 M x 8
        // 0.735 / 0.763, 50 M x 2 0.574 / 0.464, 2.6 M x 2 0.055 / 0.048).  Float32 signals: widened chunk by chunk by the one
        // loader wave this form took 0.47 against 0.39 ms (12.5 M x 8); their samples stay Float32 in the ring now (ring32).
        // (unit costs re-measured with the loader's vectorised unit scan: 25 M x 4 0.417 against 0.444, 50 M x 2 0.452 / 0.465)
        // (50 M x 2: the three passes read 0.406 - 0.465 by box, this form 0.452 - 0.458: 1.4 keeps the three passes there)
        // (with the step waves -- a fused step on groups of two / four channels, sixteen waves --: 50 M x 2 0.440, 25 M x 4 0.419)
        const double unit_cost = stepped_few && plain_nw == 16 ? (ct == 4 ? 1.16 : 1.22) : ct >= 8 ? 1.0 : ct == 4 ? 1.19 : ct == 2 ? 1.4 : 3.4;
        cost = (double)((pr + wp) * ngp) * 0.197 * unit_cost;
        if (gpm > 0) {
            cost *= (double)((ngrp + gpm - 1) / gpm);
        } else if (const char* ev = std::getenv("SIGOPS_RSOS_MINGROUPS")) {
            if (ngrp < std::atoll(ev)) return false;
        } else {
            const double t_fused = (double)((ngrp + cus - 1) / cus) * cost + 15.0;
            const double nsamp = (double)need * nch;
            // (re-measured at the round's end: 12.5 M x 8 0.452 ms, 28.8 M x 8 0.891; Float32 signals: the three passes move half
            //  the bytes and take 0.9 of the time -- 0.40 / 0.82 --, this kernel's pace is the chain's: 0.335 / 0.72)
            // (a sine formed in K2's own loads costs it ~10 %: Mix(sine, x) |> Filt(Bandstop) of 50 M x 2 0.509 ms against 0.440 for
            //  this form with its step waves, 25 M x 4 0.512 against 0.419 -- tools/iir_mix_probe.py)
            // (below a Normpower the three passes scan exactly -- launch_sos_xscan: 0.76 ms for 12.5 M x 8 where the cut scan takes 0.47)
            const double t_three = (nsamp < 1e8 ? 4.4e-6 * nsamp + 25.0 : 3.4e-6 * nsamp + 110.0) * (pure32 ? 0.9 : 1.0) * (c0.nsteps > 0 ? 1.1 : 1.0) *
                                   (S2.under_norm ? 1.6 : 1.0);
            if (std::getenv("SIGOPS_DEBUG_PLAN"))
                std::fprintf(stderr, "[sigops] single-pass IIR estimate: %.0f us (%lld groups, %lld + %lld periods), three passes %.0f us\n", t_fused,
                             (long long)ngrp, (long long)pr, (long long)wp, t_three);
            if (t_fused > t_three) return false;
        }
        g = RsSos{};
        g.n_in = S2.in_frames;
        g.n_out = need;
        g.store_lo = 0;
        g.L = L;
        g.M = L;
        g.pr = pr;
        g.wp = (int32_t)wp;
        g.nranges = (int32_t)nranges;
        g.ngroups = ngp;
        g.ks = ks;
        g.ulo = 0;
        g.ct = ct;
        g.rgs = rgs;
        g.nch = nch;
        g.chunk = 128;
        g.depth = std::max(1, std::min(4, env_int("SIGOPS_RSOS_DEPTH", 4)));
        g.nsec = cf.nsec;
        g.debug = env_int("SIGOPS_RSOS_DEBUG", 0);
        // (groups of two / four channels with a fused step: sixteen waves -- two loaders that issue, two step waves -- as in the
        //  fused resampler + IIR form; SIGOPS_PLAIN_NWAVES: measurements)
        g.nwaves = plain_nw;
        g.cyc = 1;
        {
            const int ny = 10;
            const int64_t span = (int64_t)(2 * ny - 1) * 17 + kw + 16 + 2 * g.chunk;
            int ring = 4096;
            while (ring >= 128 && rsos_lds_bytes(g.ngroups, ks, ring + 2, g.nwaves, g.cyc) > rsos_lds_budget()) ring -= 128;
            if (const char* ev = std::getenv("SIGOPS_RSOS_RING")) ring = std::min(ring, std::max(128, std::atoi(ev) / 128 * 128));
            if (ring < 128 || ring < span) return false;
            g.ring = ring;
            g.rpitch = ring + 2;
        }
        g.help = 0;  // (this form's pace is the chain wave's step: nothing more onto its SIMD)
        g.src32 = pure32 ? 1 : 0;
        g.x32 = pure32 ? 1 : 0;
        g.ring32 = pure32 && !std::getenv("SIGOPS_RSOS_NO_RING32") ? 1 : 0;
        g.gsplit = 0;
        g.fuse = -1;
        if (c0.nsteps == 1 && (c0.arg[0] & 0x2ff) == 0 && c0.nslots >= 1 && (c0.op[0] == OP_MUL || c0.op[0] == OP_ADD || c0.op[0] == OP_SUB)) {
            g.fuse = c0.op[0] == OP_MUL ? 0 : c0.op[0] == OP_ADD ? 1 : ((c0.arg[0] & 0x100) ? 3 : 2);
            const DLeaf& L0 = leaves[c0.slot_leaf[0]];
            const int kind = c0.slot_kind[0];
            if (kind == OP_FUNC && L0.mode == SO_FN_SIN && L0.sf == 1) g.fuse_sine = 1;
            else if (kind == OP_CONST || kind == OP_SCALAR) g.fuse_sine = 0;
            else g.fuse = -2;
        } else if (c0.nsteps > 0)
            g.fuse = -2;
        g.gsplit = g.nwaves == 16 && !g.src32 && g.fuse >= 0 && (ct == 2 || ct == 4) && !std::getenv("SIGOPS_RSOS_NOGSPLIT") ? 1 : 0;
        return true;
    };
    // identity taps: block gi of the period reads inputs [16 gi, 16 gi + 16)
    auto commit = [&](size_t i2, const RsSos& g) {
        Stage& S2 = stages[i2];
        const int ngp = g.ngroups, kw = 16;
        const int64_t L = g.L;
        S2.rsos_tab_host.assign((size_t)ngp * kw * 16, 0.0);
        S2.rsos_jend_host.assign((size_t)ngp, 0);
        for (int gi = 0; gi < ngp; ++gi) {
            S2.rsos_jend_host[(size_t)gi] = 16 * gi + 15;
            for (int kk = 0; kk < kw; ++kk) S2.rsos_tab_host[((size_t)gi * kw + kk) * 16 + kk] = 1.0;
        }
        S2.rsos_tab_buf = raw_buf(S2.rsos_tab_host.size() * 8);
        S2.rsos_jend_buf = raw_buf(S2.rsos_jend_host.size() * 4);
        S2.rsos_jrel_host.resize((size_t)L);  // (identity: output r's one tap is its own input)
        for (int64_t r = 0; r < L; ++r) S2.rsos_jrel_host[(size_t)r] = (int)(r % 16) - 15;
        S2.rsos_jrel_buf = raw_buf(S2.rsos_jrel_host.size() * 4);
        S2.rsos_taps = 1;
        S2.rs = g;
        S2.rsos_src = (int)i2;  // (its own carriers, control block and tables)
        const int64_t ngrp = (int64_t)(g.nch / g.ct) * ((g.nranges + g.rgs - 1) / g.rgs);
        S2.rsos_grid = (int)std::min<int64_t>(ngrp, cus);
        rsos_block_matrices(S2.groups[0], S2.rsos_mats_host);
        S2.rsos_mats_buf = raw_buf(S2.rsos_mats_host.size() * 8);
        if (std::getenv("SIGOPS_DEBUG_PLAN"))
            std::fprintf(stderr, "[sigops] IIR in one pass (k_rsos, identity resampler%s): %lld ranges of %lld periods (+%d warm-up), %lld groups of %d ch x %d ranges, ring=%d fuse=%d/%d\n",
                         S2.rsb >= 0 ? ", batched" : "", (long long)g.nranges, (long long)g.pr, g.wp, (long long)ngrp, g.ct, g.rgs, g.ring, g.fuse, g.fuse_sine);
    };
    // Several filters of this kind in one plan (config 4's 64 scenes) are ONE batched launch per pass of the three-pass form
    // (batch_sos_stages, 26 us per scene); as launches of their own of this kernel they would be 42 us each.  Small ones go
    // through this kernel TOGETHER (k_rsos_batch) where that wins, and stay for the three-pass batch otherwise.
    int candidates = 0;
    for (auto& S : stages)
        if (S.kind == ST_SOS && S.rsos_src < 0 && S.carriers.size() == 1 && S.need > 0) ++candidates;
    std::vector<size_t> small;
    for (size_t i2 = 0; i2 < stages.size(); ++i2) {
        Stage& S2 = stages[i2];
        // (xscan: the exact scan the three-pass form takes below a Normpower -- a property of that form)
        if (S2.kind != ST_SOS || S2.rsos_src >= 0 || S2.carriers.size() != 1 || S2.need <= 0 || S2.onepass || S2.sg.exact || (S2.xscan && !S2.under_norm) ||
            (S2.under_norm && S2.norm_df != 0) || S2.batch >= 0 || S2.pre_stage >= 0 || S2.base != 0 || S2.groups.size() != 1 || S2.groups[0].nsec > 6 || S2.pw_step >= 0)
            continue;
        if (candidates > 1 && (int64_t)S2.need * nodes[S2.node].nch < ((int64_t)1 << 26) && !std::getenv("SIGOPS_RSOS_MINGROUPS")) {
            small.push_back(i2);
            continue;
        }
        RsSos g{};
        double cost = 0.0;
        if (!fit(i2, 0, g, cost)) continue;
        commit(i2, g);
        if (S2.bad_buf < 0) S2.bad_buf = raw_buf((size_t)g.nch * 4);
    }
    // ---- the small ones, together: members that share the kernel's instantiation (waves; the result's type) ----
    const char* const want_s = std::getenv("SIGOPS_RSOS_BATCH");  // 0: never, 1: whenever it fits (tests, measurements), default: by the estimate
    const int want = want_s && *want_s ? std::atoi(want_s) : -1;
    if (small.size() < 2 || want == 0) return;
    std::map<std::pair<int, int>, std::vector<size_t>> kinds;
    for (size_t i2 : small) {
        if ((int)i2 == alias_stage) continue;
        // (members wait for nothing: the batch's step stands where its first member stood -- a member that reads a stage's
        //  buffer, or a scalar some launch writes, could run before its producer; such filters keep launches of their own)
        {
            const DCarrier& c0 = stages[i2].carriers[0];
            bool plain = c0.array_node >= 0 && c0.buf < 0;
            for (int k = 0; k < c0.nslots && plain; ++k)
                if (leaves[c0.slot_leaf[k]].buf >= 0) plain = false;
            if (!plain) continue;
        }
        RsSos g{};
        double cost = 0.0;
        if (!fit(i2, 1 << 20, g, cost)) continue;  // (which instantiation it would take)
        kinds[{nodes[stages[i2].node].dtype == SO_F32 ? 1 : 0, g.nwaves}].push_back(i2);
    }
    for (auto& kv : kinds) {
        const std::vector<size_t>& cand = kv.second;
        if (cand.size() < 2) continue;
        const int gpm = (int)std::max<int64_t>(1, cus / (int64_t)cand.size());
        RsBatch B;
        B.gpm = gpm;
        std::vector<RsSos> gs;
        double worst = 0.0, nsamp = 0.0;
        bool stepped = false;
        for (size_t i2 : cand) {
            RsSos g{};
            double cost = 0.0;
            if (!fit(i2, gpm, g, cost) || g.nwaves != kv.first.second) continue;
            if (!gs.empty() && (g.ring != gs[0].ring || g.ngroups != gs[0].ngroups)) continue;
            B.members.push_back((int)i2);
            gs.push_back(g);
            worst = std::max(worst, cost);
            nsamp += (double)stages[i2].need * g.nch;
            stepped = stepped || stages[i2].carriers[0].nsteps > 0;
        }
        if (B.members.size() < 2) continue;
        // the batch as a whole: its workgroups in rounds of the chip's 256, the slowest member's pace; the three-pass batch: as
        // one filter over all the members' samples (bench.py --workload config4: 64 scenes of 2.6 M x 2 with a sine mixed in,
        // 1.55 ms in three passes)
        const double t_batch = (double)(((int64_t)B.members.size() * gpm + cus - 1) / cus) * worst + 15.0;
        const double t_three = (nsamp < 1e8 ? 4.4e-6 * nsamp + 25.0 : 3.4e-6 * nsamp + 110.0) * (kv.first.first ? 0.9 : 1.0) * (stepped ? 1.2 : 1.0);
        if (std::getenv("SIGOPS_DEBUG_PLAN"))
            std::fprintf(stderr, "[sigops] batched single-pass IIR estimate: %zu members x %d groups, %.0f us; three passes %.0f us\n", B.members.size(), gpm,
                         t_batch, t_three);
        if (want != 1 && t_batch > t_three) continue;
        const int bi = (int)rsbatches.size();
        int off = 0;
        for (size_t m = 0; m < B.members.size(); ++m) {
            stages[B.members[m]].rsb = bi;
            commit((size_t)B.members[m], gs[m]);
            B.bad_off.push_back(off);
            off += gs[m].nch;
        }
        B.bad_buf = raw_buf((size_t)std::max(off, 1) * 4);
        B.items_buf = raw_buf(B.members.size() * sizeof(RsosItem));
        B.fix_buf = raw_buf(B.members.size() * sizeof(RsFixup));
        rsbatches.push_back(std::move(B));
    }
}

void Plan::fuse_state_passes() {
    if (!std::getenv("SIGOPS_FUSE_STATE")) return;
    for (size_t i2 = 0; i2 < stages.size(); ++i2) {
        Stage& S2 = stages[i2];
        if (S2.kind != ST_SOS || S2.onepass || S2.need <= 0 || S2.base > 0 || S2.groups.size() != 1 || S2.in_buf < 0 ||
            S2.in_array_node >= 0 || S2.in_offset != 0 || S2.pw_step >= 0 || S2.sg.nchunks <= 1)
            continue;
        int i3 = -1;
        for (size_t j = 0; j < stages.size(); ++j)
            if (stages[j].kind == ST_RESAMPLE && stages[j].out_buf == S2.in_buf) i3 = (int)j;
        if (i3 < 0 || i3 == alias_stage || stages[i3].win_off >= 0) continue;
        Stage& S3 = stages[i3];
        const RsPeriodic& rp0 = S3.rp;
        if (!S3.periodic || S2.sg.exact || S2.xscan || S2.src_op || rp0.ga || rp0.arr2 || rp0.nstate || nodes[S3.node].dtype != SO_F64 || nodes[S2.node].dtype != SO_F64 ||
            rp0.rows != 32 || rp0.nwaves - rp0.ncompute < 4 || S3.need < S2.in_frames || S3.per_j.empty() || rp0.kw != 56 ||
            (rp0.ngroups + rp0.ncompute - 1) / rp0.ncompute != 1 || !(rp0.ct == 8 || rp0.ct == 4))
            continue;  // (the instantiations with state waves: k_resample.hip launch_rp_st)
        // nothing else may read the intermediate
        bool other = false;
        for (auto& L : leaves)
            if (L.buf == S2.in_buf) other = true;
        for (size_t j = 0; j < stages.size(); ++j) {
            if (j != i2 && stages[j].in_buf == S2.in_buf) other = true;
            for (auto& c : stages[j].carriers)
                if (c.buf == S2.in_buf) other = true;
        }
        if (other) continue;
        const SosCoefs& cf = S2.groups[0];
        const int D = 2 * cf.nsec;
        const int64_t Ls = rp0.L, L = (int64_t)rp0.pt * Ls;
        if (L < 32 || L > 16384) continue;
        // chunk geometry with L = pt periods
        const double tol = std::ldexp(1.0, -70);
        Mat A = sos_state_matrix(cf);
        Mat M = matpow(A, L, D), cur = ident(D);
        std::vector<double> mp;
        int K = 0;
        bool ok = true;
        for (;;) {
            mp.insert(mp.end(), cur.begin(), cur.end());
            ++K;
            cur = matmul(cur, M, D);
            if (maxabs(cur) < tol) break;
            if (K >= 64) {
                ok = false;
                break;
            }
        }
        const int64_t nchunks = (S2.need + L - 1) / L;
        if (!ok || nchunks <= 1) continue;
        // W = G . Tap over the staged span [jlo, jlo + 4*ksw) of a period row
        const so_node_t& nd3 = nodes[S3.node].nd;
        const double* h = (const double*)nd3.p0;
        const int hlen = nd3.i2, nphi = S3.rg.nphi, taps = S3.rg.taps;
        const int jlo = rp0.jlo;
        const int ksw = 2 * 24;  // two state waves x kSwK k-steps (k_resample.hip)
        if ((S3.jend_last - jlo + 1 + 3) / 4 > ksw) continue;  // the staged span of a row must fit
        // (window slots beyond the span have zero taps; there a row's window runs into the next
        //  row's staged frames or the slot's slack -- finite values: the kernel zeroes its LDS ring
        //  once at start when it has state waves, and 0 x finite is 0)
        std::vector<double> G((size_t)Ls * D, 0.0);  // G[r] = A^(Ls-1-r) B1
        {
            std::vector<double> st_(D, 0.0);
            double y = 1.0;
            for (int f = 0; f < cf.nsec; ++f) {  // one DF2T step with x = 1 from zero state
                const double xi = y;
                y = st_[2 * f] + cf.b0[f] * xi;
                st_[2 * f] = st_[2 * f + 1] + cf.b1[f] * xi - cf.a1[f] * y;
                st_[2 * f + 1] = cf.b2[f] * xi - cf.a2[f] * y;
            }
            for (int64_t r = Ls - 1; r >= 0; --r) {
                for (int d = 0; d < D; ++d) G[(size_t)r * D + d] = st_[d];
                std::vector<double> nx(D, 0.0);
                for (int a = 0; a < D; ++a)
                    for (int b = 0; b < D; ++b) nx[a] += A[(size_t)a * D + b] * st_[b];
                st_ = nx;
            }
        }
        std::vector<double> wt((size_t)4 * ksw * 16, 0.0);
        for (int64_t r = 0; r < Ls; ++r)
            for (int age = 0; age < taps; ++age) {
                const int64_t rel = S3.per_j[r] - age - jlo;  // input slot of this tap
                if (rel < 0 || rel >= 4 * ksw) {
                    ok = false;  // (cannot happen: the span covers every tap of the period)
                    continue;
                }
                const int64_t hi = S3.per_p[r] + (int64_t)nphi * age;
                const double hv = hi < hlen ? h[hi] : 0.0;
                const double dv = hi + 1 < hlen ? h[hi + 1] - h[hi] : 0.0;
                const double tv = hv + S3.per_a[r] * dv;
                for (int d = 0; d < D; ++d) wt[(size_t)rel * 16 + d] += G[(size_t)r * D + d] * tv;
            }
        if (!ok) continue;
        // LDS: the taps ([4*ksw][10] doubles) go behind the gain ring; keep at least three tile slots
        {
            const size_t avail = 160 * 1024 - sizeof(RsCtl) - 64;
            const size_t tile_bytes = (size_t)rp0.ct * rp0.lds_pitch * 8;
            const size_t fbytes = (size_t)2 * rp0.fslots * rp0.fpitch * 8 + (rp0.ftwo ? kRsTwoDoubles * 8 : 0);
            const size_t wbytes = (size_t)4 * ksw * 10 * 8;
            if (fbytes + wbytes + 3 * tile_bytes > avail) continue;
            S3.rp.nslots = (int)std::min<size_t>(S3.rp.nslots, (avail - fbytes - wbytes) / tile_bytes);
        }
        // commit: resampler side
        S3.rp.nstate = 2;
        S3.rp.ksw = ksw;
        S3.wtab_host = wt;
        S3.wtab_buf = raw_buf(wt.size() * 8);
        S3.vper_buf = raw_buf((size_t)2 * nodes[S3.node].nch * rp0.nperiods * 16 * 8);
        // ... and the IIR side
        S2.pre_stage = i3;
        S2.qmat_host = matpow(A, Ls, D);
        S2.qmat_buf = raw_buf(S2.qmat_host.size() * 8);
        S2.sg.chunk = L;
        S2.sg.nchunks = (int)nchunks;
        S2.sg.kterms = K;
        S2.mpow_host.assign(1, mp);
        if (S2.mpow_buf >= 0) bufs[S2.mpow_buf].bytes = std::max<size_t>(8, mp.size() * 8);
        else S2.mpow_buf = raw_buf(mp.size() * 8);
        const size_t vb = (size_t)nchunks * S2.sg.nch * 2 * kMaxSec * 8;
        if (S2.v_buf >= 0) bufs[S2.v_buf].bytes = vb;
        else S2.v_buf = raw_buf(vb);
        if (S2.s0_buf >= 0) bufs[S2.s0_buf].bytes = vb;
        else S2.s0_buf = raw_buf(vb);
        if (std::getenv("SIGOPS_DEBUG_PLAN"))
            std::fprintf(stderr, "[sigops] IIR state pass fused into the resampler: chunk %lld frames, K=%d, window %d inputs\n",
                         (long long)L, K, 4 * ksw);
    }
}


}  // namespace so
