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
