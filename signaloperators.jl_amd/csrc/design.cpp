// Host-side (fp64) filter design: the DSP.jl calls the reference makes when a
// FilterBlock is built at sink time —
//   digitalfilter(Type(f...;fs=fs), Butterworth(n)|Chebyshev1(n,r)) |> DF2TFilter
//       (reference src/filters.jl:10-11,94)
//   resample_filter(ratio)                (reference src/reformatting.jl:93)
// DSP.jl 0.6.10 is not vendored under /root/reference; the algorithms are the
// published ones recorded in SURVEY.md Appendix B (analog prototype -> frequency
// transform of the prewarped band edges -> bilinear -> second-order sections;
// Kaiser-windowed sinc with kaiserord sizing).  tests/test_design.py checks both
// against scipy.signal (butter/cheby1 zpk, firwin).
#include <algorithm>
#include <cmath>
#include <complex>
#include <string>
#include <vector>

#include "../../include/sigops.h"
#include "plan.h"

namespace so {

using cd = std::complex<double>;

static double sinpi_h(double x) {
    double r = std::fmod(x, 2.0);
    double n = std::nearbyint(2.0 * r);
    double t = r - 0.5 * n;
    int q = (((int)n % 4) + 4) % 4;
    switch (q) {
    case 0: return std::sin(M_PI * t);
    case 1: return std::cos(M_PI * t);
    case 2: return -std::sin(M_PI * t);
    default: return -std::cos(M_PI * t);
    }
}
static double cospi_h(double x) { return sinpi_h(x + 0.5); }

struct ZPK {
    std::vector<cd> z, p;
    double k;
};

// Butterworth(n): DSP.jl poles (-sinpi(w) + i cospi(w)), w = (2i-1)/2n, conj pairs, -1 if odd
static ZPK butter_proto(int n) {
    ZPK f;
    f.k = 1.0;
    for (int i = 1; i <= n / 2; ++i) {
        double w = (2.0 * i - 1.0) / (2.0 * n);
        cd pole(-sinpi_h(w), cospi_h(w));
        f.p.push_back(pole);
        f.p.push_back(std::conj(pole));
    }
    if (n & 1) f.p.push_back(cd(-1.0, 0.0));
    return f;
}

// Chebyshev1(n, ripple dB)
static ZPK cheby1_proto(int n, double ripple) {
    ZPK f;
    double eps = std::sqrt(std::pow(10.0, ripple / 10.0) - 1.0);
    double mu = std::asinh(1.0 / eps) / n;
    cd prod(1.0, 0.0);
    for (int i = 1; i <= n / 2; ++i) {
        double w = (2.0 * i - 1.0) / (2.0 * n);
        cd pole(-std::sinh(mu) * sinpi_h(w), std::cosh(mu) * cospi_h(w));
        f.p.push_back(pole);
        f.p.push_back(std::conj(pole));
        prod *= (-pole) * (-std::conj(pole));
    }
    if (n & 1) {
        cd pole(-std::sinh(mu), 0.0);
        f.p.push_back(pole);
        prod *= -pole;
    }
    f.k = prod.real();
    if (!(n & 1)) f.k /= std::sqrt(1.0 + eps * eps);
    return f;
}

static double prewarp(double w) { return 4.0 * std::tan(M_PI * w / 2.0); }

static cd prod_neg(const std::vector<cd>& v) {
    cd r(1.0, 0.0);
    for (auto& x : v) r *= -x;
    return r;
}

static ZPK lp2lp(const ZPK& f, double w0) {
    ZPK g;
    for (auto& z : f.z) g.z.push_back(z * w0);
    for (auto& p : f.p) g.p.push_back(p * w0);
    g.k = f.k * std::pow(w0, (double)f.p.size() - (double)f.z.size());
    return g;
}
static ZPK lp2hp(const ZPK& f, double w0) {
    ZPK g;
    for (auto& z : f.z) g.z.push_back(w0 / z);
    for (auto& p : f.p) g.p.push_back(w0 / p);
    for (size_t i = f.z.size(); i < f.p.size(); ++i) g.z.push_back(cd(0, 0));
    g.k = f.k * (prod_neg(f.z) / prod_neg(f.p)).real();
    return g;
}
static ZPK lp2bp(const ZPK& f, double w1, double w2) {
    ZPK g;
    double bw = w2 - w1, w0 = std::sqrt(w1 * w2);
    for (auto& z : f.z) {
        cd a = z * (bw / 2.0), d = std::sqrt(a * a - w0 * w0);
        g.z.push_back(a + d);
        g.z.push_back(a - d);
    }
    for (auto& p : f.p) {
        cd a = p * (bw / 2.0), d = std::sqrt(a * a - w0 * w0);
        g.p.push_back(a + d);
        g.p.push_back(a - d);
    }
    for (size_t i = f.z.size(); i < f.p.size(); ++i) g.z.push_back(cd(0, 0));
    g.k = f.k * std::pow(bw, (double)f.p.size() - (double)f.z.size());
    return g;
}
static ZPK lp2bs(const ZPK& f, double w1, double w2) {
    ZPK g;
    double bw = w2 - w1, w0 = std::sqrt(w1 * w2);
    for (auto& z : f.z) {
        cd a = (bw / 2.0) / z, d = std::sqrt(a * a - w0 * w0);
        g.z.push_back(a + d);
        g.z.push_back(a - d);
    }
    for (auto& p : f.p) {
        cd a = (bw / 2.0) / p, d = std::sqrt(a * a - w0 * w0);
        g.p.push_back(a + d);
        g.p.push_back(a - d);
    }
    for (size_t i = f.z.size(); i < f.p.size(); ++i) {
        g.z.push_back(cd(0, w0));
        g.z.push_back(cd(0, -w0));
    }
    g.k = f.k * (prod_neg(f.z) / prod_neg(f.p)).real();
    return g;
}
// bilinear with fs = 2 (DSP.jl digitalfilter)
static ZPK bilinear(const ZPK& f) {
    const double fs2 = 4.0;
    ZPK g;
    cd num(1, 0), den(1, 0);
    for (auto& z : f.z) {
        g.z.push_back((fs2 + z) / (fs2 - z));
        num *= (fs2 - z);
    }
    for (auto& p : f.p) {
        g.p.push_back((fs2 + p) / (fs2 - p));
        den *= (fs2 - p);
    }
    for (size_t i = f.z.size(); i < f.p.size(); ++i) g.z.push_back(cd(-1, 0));
    g.k = f.k * (num / den).real();
    return g;
}

struct Group {
    cd a, b;  // the two roots (b == a for a single root with one==true)
    bool one;
};

static std::vector<Group> group_roots(std::vector<cd> r) {
    std::vector<Group> out;
    std::vector<cd> reals;
    std::vector<bool> used(r.size(), false);
    for (size_t i = 0; i < r.size(); ++i) {
        if (used[i]) continue;
        double tol = 1e-10 * std::max(1.0, std::abs(r[i]));
        if (std::abs(r[i].imag()) <= tol) {
            used[i] = true;
            reals.push_back(cd(r[i].real(), 0));
            continue;
        }
        // find the conjugate partner
        size_t best = r.size();
        double bd = 1e300;
        for (size_t j = i + 1; j < r.size(); ++j) {
            if (used[j]) continue;
            double d = std::abs(r[j] - std::conj(r[i]));
            if (d < bd) {
                bd = d;
                best = j;
            }
        }
        used[i] = true;
        if (best < r.size()) used[best] = true;
        cd a = r[i].imag() > 0 ? r[i] : std::conj(r[i]);
        out.push_back({a, std::conj(a), false});
    }
    std::sort(reals.begin(), reals.end(), [](cd x, cd y) { return x.real() < y.real(); });
    for (size_t i = 0; i + 1 < reals.size(); i += 2) out.push_back({reals[i], reals[i + 1], false});
    if (reals.size() & 1) out.push_back({reals.back(), reals.back(), true});
    return out;
}

// ZeroPoleGain -> SecondOrderSections: sections ordered so that the poles nearest
// the unit circle come last (a lone real pole first); each pole group takes the
// nearest remaining zero group.
static void zpk2sos(const ZPK& f, std::vector<double>& sos, double& gain) {
    std::vector<Group> pg = group_roots(f.p), zg = group_roots(f.z);
    auto dist = [](const Group& g) { return std::abs(std::abs(g.a) - 1.0); };
    std::sort(pg.begin(), pg.end(), [&](const Group& x, const Group& y) {
        if (x.one != y.one) return y.one;  // pairs first here; reversed below
        return dist(x) < dist(y);
    });
    std::vector<bool> zused(zg.size(), false);
    struct Sec { Group p; Group z; bool hasz; };
    std::vector<Sec> secs;
    for (auto& p : pg) {
        int best = -1;
        double bd = 1e300;
        for (size_t i = 0; i < zg.size(); ++i) {
            if (zused[i]) continue;
            if (p.one && !zg[i].one) continue;  // a first-order section takes a single zero
            double d = std::abs(zg[i].a - p.a);
            if (d < bd) {
                bd = d;
                best = (int)i;
            }
        }
        Sec s{p, Group{}, false};
        if (best >= 0) {
            zused[best] = true;
            s.z = zg[best];
            s.hasz = true;
        }
        secs.push_back(s);
    }
    std::reverse(secs.begin(), secs.end());  // nearest-to-unit-circle last, lone real first
    sos.clear();
    for (auto& s : secs) {
        double b0 = 1, b1 = 0, b2 = 0, a1, a2;
        if (s.p.one) {
            a1 = -s.p.a.real();
            a2 = 0;
        } else {
            a1 = -(s.p.a + s.p.b).real();
            a2 = (s.p.a * s.p.b).real();
        }
        if (s.hasz) {
            if (s.z.one) {
                b1 = -s.z.a.real();
            } else {
                b1 = -(s.z.a + s.z.b).real();
                b2 = (s.z.a * s.z.b).real();
            }
        }
        sos.insert(sos.end(), {b0, b1, b2, 1.0, a1, a2});
    }
    gain = f.k;
}

static int design_zpk(int type, double f1, double f2, double fs, int method, int order, double ripple, ZPK& d,
                      std::string& err) {
    if (order < 1 || order > 32) {
        err = "filter order must be in 1..32";
        return SO_ERR_INVALID;
    }
    if (!(fs > 0)) {
        err = "Filt needs a known frame rate";
        return SO_ERR_INVALID;
    }
    ZPK proto = method == SO_METHOD_CHEBYSHEV1 ? cheby1_proto(order, ripple) : butter_proto(order);
    double w1 = 2.0 * f1 / fs, w2 = 2.0 * f2 / fs;
    auto bad = [](double w) { return !(w > 0.0 && w < 1.0); };
    ZPK a;
    switch (type) {
    case SO_FILT_LOWPASS:
        if (bad(w1)) { err = "frequencies must be in (0, fs/2)"; return SO_ERR_INVALID; }
        a = lp2lp(proto, prewarp(w1));
        break;
    case SO_FILT_HIGHPASS:
        if (bad(w1)) { err = "frequencies must be in (0, fs/2)"; return SO_ERR_INVALID; }
        a = lp2hp(proto, prewarp(w1));
        break;
    case SO_FILT_BANDPASS:
        if (bad(w1) || bad(w2) || !(w1 < w2)) { err = "band edges must satisfy 0 < f1 < f2 < fs/2"; return SO_ERR_INVALID; }
        a = lp2bp(proto, prewarp(w1), prewarp(w2));
        break;
    case SO_FILT_BANDSTOP:
        if (bad(w1) || bad(w2) || !(w1 < w2)) { err = "band edges must satisfy 0 < f1 < f2 < fs/2"; return SO_ERR_INVALID; }
        a = lp2bs(proto, prewarp(w1), prewarp(w2));
        break;
    default: err = "unknown filter type"; return SO_ERR_INVALID;
    }
    d = bilinear(a);
    return SO_OK;
}

int design_iir(int type, double f1, double f2, double fs, int method, int order, double ripple,
               std::vector<double>& sos, double& gain, std::string& err) {
    ZPK d;
    int st = design_zpk(type, f1, f2, fs, method, order, ripple, d, err);
    if (st != SO_OK) return st;
    zpk2sos(d, sos, gain);
    return SO_OK;
}

// digitalfilter(...) as the ZeroPoleGain object itself (interleaved re,im pairs)
int design_iir_zpk(int type, double f1, double f2, double fs, int method, int order, double ripple,
                   std::vector<double>& z, std::vector<double>& p, double& k, std::string& err) {
    ZPK d;
    int st = design_zpk(type, f1, f2, fs, method, order, ripple, d, err);
    if (st != SO_OK) return st;
    z.clear();
    p.clear();
    for (auto& c : d.z) z.insert(z.end(), {c.real(), c.imag()});
    for (auto& c : d.p) p.insert(p.end(), {c.real(), c.imag()});
    k = d.k;
    return SO_OK;
}

// DF2TFilter(::ZeroPoleGain) -> SecondOrderSections (reference src/filters.jl:94 resolve_filter)
int zpk_to_sos(const double* z, int nz, const double* p, int np, double k, std::vector<double>& sos, double& gain,
               std::string& err) {
    if (nz < 0 || np < 0 || nz > np || np > 64) {
        err = "ZeroPoleGain: need 0 <= zeros <= poles <= 64";
        return SO_ERR_INVALID;
    }
    ZPK d;
    for (int i = 0; i < nz; ++i) d.z.push_back(cd(z[2 * i], z[2 * i + 1]));
    for (int i = 0; i < np; ++i) d.p.push_back(cd(p[2 * i], p[2 * i + 1]));
    d.k = k;
    zpk2sos(d, sos, gain);
    return SO_OK;
}

// ---- PolynomialRatio -> SecondOrderSections --------------------------------
// `filt(b, a, x)` and `Filt(x, PolynomialRatio(b, a))` (reference src/filters.jl:68-95) run DSP.jl's direct-form DF2T
// recurrence of order max(|b|, |a|) - 1.  The engine's IIR kernels run cascades of second-order sections, so the two
// polynomials are factored here: all roots at once by the Aberth-Ehrlich iteration (fp64 complex; cubic for simple roots), conjugates made exact, then zpk2sos.  H(w) = B(w) / A(w) in w = 1/z:
//   B(w) = b_d w^d prod_i (1 - z_i w)   (d leading zero coefficients = a pure delay, its own sections with b0 = 0)
// The factored form is another rounding of the same transfer function -- and where the polynomial is ill-conditioned
// not the same function at all; `resid` reports the relative l2 distance of the two impulse responses (direct form
// against cascade, DSP.jl's recurrences both) for the caller to gate on.
static cd poly_eval(const std::vector<double>& c, cd x, cd* deriv) {  // c: descending powers
    cd p = c[0], d = 0;
    for (size_t i = 1; i < c.size(); ++i) {
        d = d * x + p;
        p = p * x + c[i];
    }
    if (deriv) *deriv = d;
    return p;
}

static bool poly_roots(const std::vector<double>& c, std::vector<cd>& r) {  // c[0] != 0, c.back() != 0
    const int n = (int)c.size() - 1;
    r.assign(n, cd(0));
    if (n == 0) return true;
    if (n == 1) {
        r[0] = -c[1] / c[0];
        return true;
    }
    // starting circle: between the Cauchy bounds of the roots' moduli
    double hi = 0, lo = 0;
    for (int i = 1; i <= n; ++i) hi = std::max(hi, std::pow(std::abs(c[i] / c[0]), 1.0 / i));
    for (int i = 0; i < n; ++i) lo = std::max(lo, std::pow(std::abs(c[i] / c[n]), 1.0 / (n - i)));
    hi *= 2.0;
    lo = lo > 0 ? 0.5 / lo : 0.0;
    double rad = std::sqrt(std::max(hi, 1e-300) * std::max(lo, 1e-300));
    if (!(rad > 0) || !std::isfinite(rad)) rad = 1.0;
    for (int i = 0; i < n; ++i) r[i] = std::polar(rad, 2.0 * M_PI * i / n + 0.4);
    bool ok = false;
    for (int it = 0; it < 400 && !ok; ++it) {
        double step = 0, scale = 0;
        for (int i = 0; i < n; ++i) {
            cd d, p = poly_eval(c, r[i], &d);
            if (p == cd(0)) continue;
            cd nw = (d == cd(0)) ? cd(1e-3 * (std::abs(r[i]) + 1.0)) : p / d;
            cd rep = 0;
            for (int j = 0; j < n; ++j)
                if (j != i) {
                    cd df = r[i] - r[j];
                    if (df == cd(0)) df = cd(1e-12 * (std::abs(r[i]) + 1.0));
                    rep += 1.0 / df;
                }
            cd den = 1.0 - nw * rep;
            cd w = (std::abs(den) < 1e-30) ? nw : nw / den;
            r[i] -= w;
            step = std::max(step, std::abs(w));
            scale = std::max(scale, std::abs(r[i]));
        }
        ok = step <= 4e-16 * std::max(scale, 1e-300);
    }
    for (auto& x : r)
        if (!std::isfinite(x.real()) || !std::isfinite(x.imag())) return false;
    // Multiple roots (the n zeros at -1 of a Butterworth low-pass, at +-1 of a band-pass) come back as a ring of radius
    // ~eps^(1/m) around the true root, whose centroid is accurate to ~eps: cluster at a few radii, put every cluster at
    // its centroid, and keep the coarsest clustering whose product still reproduces the coefficients to rounding.
    using cl = std::complex<long double>;
    auto rebuilt_error = [&](const std::vector<cd>& q) {
        std::vector<cl> pc(1, cl(1));
        for (const cd& x : q) {
            pc.push_back(cl(0));
            for (size_t k = pc.size() - 1; k >= 1; --k) pc[k] -= cl(x) * pc[k - 1];
        }
        long double e = 0, m = 0;
        for (int k = 0; k <= n; ++k) {
            e = std::max(e, std::abs(pc[k] * (long double)c[0] - cl(c[k])));
            m = std::max(m, (long double)std::abs(c[k]));
        }
        return (double)(e / m);
    };
    std::vector<cd> best = r;
    const double e0 = rebuilt_error(r), accept = std::max(4.0 * e0, 4e-15 * n);
    for (double tau : {1e-7, 1e-5, 1e-4, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1}) {
        std::vector<int> id(n, -1);
        int ncl = 0;
        for (int i = 0; i < n; ++i) {  // single linkage
            if (id[i] >= 0) continue;
            std::vector<int> stack(1, i);
            id[i] = ncl;
            while (!stack.empty()) {
                int u = stack.back();
                stack.pop_back();
                for (int v = 0; v < n; ++v)
                    if (id[v] < 0 && std::abs(r[u] - r[v]) <= tau * std::max(1.0, std::abs(r[u]))) {
                        id[v] = ncl;
                        stack.push_back(v);
                    }
            }
            ++ncl;
        }
        if (ncl == n) continue;
        std::vector<cd> cen(ncl, cd(0));
        std::vector<int> cnt(ncl, 0);
        for (int i = 0; i < n; ++i) {
            cen[id[i]] += r[i];
            ++cnt[id[i]];
        }
        for (int k = 0; k < ncl; ++k) {
            if (cnt[k] < 2) continue;
            cd m = cen[k] / (double)cnt[k];
            // an m-fold root of p is a simple root of its (m-1)th derivative: Newton there from the centroid (p itself
            // is rounding noise within eps^(1/m) of the root, which is why the ring is no better than that)
            std::vector<double> dq(c);
            for (int t = 1; t < cnt[k]; ++t) {
                const int deg = (int)dq.size() - 1;
                for (int u = 0; u < deg; ++u) dq[u] *= (double)(deg - u);
                dq.pop_back();
            }
            for (int t = 0; t < 8; ++t) {
                cd d, v = poly_eval(dq, m, &d);
                if (d == cd(0)) break;
                cd w = v / d;
                if (!(std::abs(w) <= tau * std::max(1.0, std::abs(m)))) break;  // (left the cluster: not its root)
                m -= w;
            }
            if (std::abs(m.imag()) <= tau * std::max(1.0, std::abs(m))) m = cd(m.real(), 0.0);
            cen[k] = m;
        }
        std::vector<cd> q(n);
        for (int i = 0; i < n; ++i) q[i] = cnt[id[i]] > 1 ? cen[id[i]] : r[i];
        // (the ring itself reproduces the coefficients to rounding -- it is the exact root set of a neighbouring
        //  polynomial -- but is no set of conjugate pairs; the coarsest clustering that reproduces them as well wins)
        if (rebuilt_error(q) <= accept) best = q;
    }
    r = best;
    return true;
}

// direct-form DF2T (DSP.jl `_filt_iir!`): y = si[0] + b0 x; si[j] = si[j+1] + b[j+1] x - a[j+1] y
static void df2t_direct(const std::vector<double>& b, const std::vector<double>& a, std::vector<double>& si,
                        const double* x, double* y, int64_t n) {
    const int ord = (int)si.size();
    for (int64_t i = 0; i < n; ++i) {
        const double xi = x ? x[i] : 0.0;
        const double yi = (ord ? si[0] : 0.0) + b[0] * xi;
        for (int j = 0; j + 1 < ord; ++j) si[j] = si[j + 1] + b[j + 1] * xi - a[j + 1] * yi;
        if (ord) si[ord - 1] = b[ord] * xi - a[ord] * yi;
        y[i] = yi;
    }
}

static bool normalise_tf(const double* b, int nb, const double* a, int na, std::vector<double>& bn,
                         std::vector<double>& an, std::string& err) {
    if (nb < 1 || na < 1 || !b || !a) {
        err = "filt: empty coefficient vector";
        return false;
    }
    if (a[0] == 0.0 || !std::isfinite(a[0])) {
        err = "filt: a[1] must be nonzero";  // (DSP.jl: "filter must have non-zero leading denominator coefficient")
        return false;
    }
    const int sz = std::max(nb, na);
    bn.assign(sz, 0.0);
    an.assign(sz, 0.0);
    for (int i = 0; i < nb; ++i) bn[i] = b[i] / a[0];
    for (int i = 0; i < na; ++i) an[i] = a[i] / a[0];
    for (int i = 0; i < sz; ++i)
        if (!std::isfinite(bn[i]) || !std::isfinite(an[i])) {
            err = "filt: non-finite coefficient";
            return false;
        }
    return true;
}

int tf_to_sos(const double* b, int nb, const double* a, int na, std::vector<double>& sos, double& gain, double& resid,
              std::string& err) {
    std::vector<double> bn, an;
    if (!normalise_tf(b, nb, a, na, bn, an, err)) return SO_ERR_INVALID;
    if ((int)bn.size() > 129) {
        err = "filt: at most 128 coefficients besides the leading one";
        return SO_ERR_INVALID;
    }
    auto strip = [](std::vector<double> c, int& lead) {  // w-polynomial, ascending: lead = leading zeros; drop trailing zeros
        lead = 0;
        while (lead < (int)c.size() && c[lead] == 0.0) ++lead;
        c.erase(c.begin(), c.begin() + lead);
        while (!c.empty() && c.back() == 0.0) c.pop_back();
        return c;
    };
    int d = 0, da = 0;
    std::vector<double> bw = strip(bn, d), aw = strip(an, da);  // (da == 0: an[0] == 1)
    sos.clear();
    resid = 0.0;
    if (bw.empty()) {  // b == 0: the zero filter
        sos = {0.0, 0.0, 0.0, 1.0, 0.0, 0.0};
        gain = 1.0;
        return SO_OK;
    }
    // ascending in w = descending in z once multiplied by z^deg: the same coefficient order
    ZPK f;
    if (!poly_roots(bw, f.z) || !poly_roots(aw, f.p)) {
        err = "filt: the coefficient polynomials could not be factored";
        return SO_ERR_UNSUPPORTED;
    }
    f.k = bw[0];
    while (f.p.size() < f.z.size()) f.p.push_back(cd(0));  // more zeros than poles: FIR sections (poles at the origin)
    zpk2sos(f, sos, gain);
    for (int i = 0; i < d; i += 2) {
        const bool two = i + 1 < d;
        sos.insert(sos.begin(), {0.0, two ? 0.0 : 1.0, two ? 1.0 : 0.0, 1.0, 0.0, 0.0});
    }
    // the probe: impulse responses of the two forms over the filter's memory (until both states have decayed, <= 2^16)
    const int ord = (int)bn.size() - 1, nsec = (int)sos.size() / 6;
    std::vector<double> si(ord, 0.0), s1(nsec, 0.0), s2(nsec, 0.0);
    double num = 0, den = 0, tail = 0;
    const int64_t cap = 1 << 16;
    for (int64_t i = 0; i < cap; ++i) {
        const double xi = i == 0 ? 1.0 : 0.0;
        double yd;
        df2t_direct(bn, an, si, &xi, &yd, 1);
        double v = xi;
        for (int s0 = 0; s0 < nsec; ++s0) {  // DSP.jl's SOS recurrence (oracle/sigops_oracle.c, sos_filt)
            const double* c = &sos[6 * s0];
            const double y = s1[s0] + c[0] * v;
            s1[s0] = s2[s0] + c[1] * v - c[4] * y;
            s2[s0] = c[2] * v - c[5] * y;
            v = y;
        }
        v *= gain;
        num += (v - yd) * (v - yd);
        den += yd * yd;
        tail = (i & 255) ? std::max(tail, std::abs(yd)) : std::abs(yd);
        if (!std::isfinite(yd) || !std::isfinite(v)) break;
        if (i > 4 * ord + 64 && (i & 255) == 255 && tail * tail <= 1e-40 * den) break;  // decayed
    }
    resid = den > 0 ? std::sqrt(num / den) : (num > 0 ? 1.0 : 0.0);
    if (!std::isfinite(resid)) resid = 1.0;
    return SO_OK;
}

// zero-input response of the direct form from the initial state `si` (length max(|b|,|a|) - 1): what `filt(b, a, x, si)`
// adds to the zero-state response (linearity).  Writes at most `cap` frames and returns how many are needed until the
// response is below 2^-80 of its largest value (== cap where it has not decayed by then).
int tf_zero_input(const double* b, int nb, const double* a, int na, const double* si0, int nsi, double* out, int64_t cap,
                  int64_t& used, std::string& err) {
    std::vector<double> bn, an;
    if (!normalise_tf(b, nb, a, na, bn, an, err)) return SO_ERR_INVALID;
    const int ord = (int)bn.size() - 1;
    if (nsi != ord || (ord > 0 && !si0)) {
        err = "filt: the initial state must have max(length(a), length(b)) - 1 entries";  // (DSP.jl's ArgumentError)
        return SO_ERR_INVALID;
    }
    std::vector<double> si(si0, si0 + ord);
    double peak = 0;
    used = 0;
    for (int64_t i = 0; i < cap; ++i) {
        double y;
        df2t_direct(bn, an, si, nullptr, &y, 1);
        out[i] = y;
        peak = std::max(peak, std::abs(y));
        double st = 0;
        for (double v : si) st = std::max(st, std::abs(v));
        if (!(st > 0x1p-80 * peak) && std::isfinite(st)) {
            used = i + 1;
            return SO_OK;
        }
    }
    used = cap;
    return SO_OK;
}

// ---- resample_filter -------------------------------------------------------
static double bessel_i0(double x) {
    double s = 1.0, t = 1.0, h = x / 2.0;
    for (int k = 1; k < 500; ++k) {
        t *= (h / k) * (h / k);
        s += t;
        if (t < 1e-18 * s) break;
    }
    return s;
}

static void kaiser_lowpass(int hlen, double cutoff, double beta, double scale,
                           std::vector<double>& h) {
    h.resize(hlen);
    double i0b = bessel_i0(beta), sum = 0.0;
    for (int k = 0; k < hlen; ++k) {
        double u = hlen > 1 ? 2.0 * k / (hlen - 1) - 1.0 : 0.0;
        double w = bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - u * u))) / i0b;
        double x = cutoff * (k - (hlen - 1) / 2.0);
        double sinc = x == 0.0 ? 1.0 : sinpi_h(x) / (M_PI * x);
        h[k] = cutoff * sinc * w;
        sum += h[k];
    }
    for (auto& v : h) v = v / sum * scale;  // unity DC gain, then rmul!(h, Nphi)
}

static void resample_taps(double cutoff, int nphi, std::vector<double>& h) {
    const double att = 60.0;
    double tw = cutoff * 0.2;
    int n = (int)std::ceil((att - 7.95) / (M_PI * 2.285 * tw)) + 1;  // kaiserord
    double beta = 0.1102 * (att - 8.7);
    int hlen = nphi * (int)std::ceil((double)n / nphi);
    if (hlen % 2 == 0) hlen += 1;
    kaiser_lowpass(hlen, cutoff, beta, (double)nphi, h);
}

int design_resample_rational(int64_t num, int64_t den, std::vector<double>& h, std::string& err) {
    if (num < 1 || den < 1 || num > 4096 || den > 4096) {
        err = "rational resampling ratio out of range";
        return SO_ERR_INVALID;
    }
    double f_nyq = std::min(1.0 / (double)num, 1.0 / (double)den);
    resample_taps(f_nyq, (int)num, h);
    return SO_OK;
}

int design_resample_arbitrary(double rate, int nphi, std::vector<double>& h, std::string& err) {
    if (!(rate > 0.0) || nphi < 1) {
        err = "rate must be greater than 0";
        return SO_ERR_INVALID;
    }
    double f_nyq = rate >= 1.0 ? 1.0 / nphi : rate / nphi;
    resample_taps(f_nyq, nphi, h);
    return SO_OK;
}

}  // namespace so
