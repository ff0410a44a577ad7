// extern "C" surface of libsigops (include/sigops.h).
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/sigops.h"
#include "plan.h"

namespace {
thread_local std::string g_err;
int set_err(int status, const std::string& msg) {
    g_err = msg;
    return status;
}
}  // namespace

struct so_plan {
    so::Plan* p;
};

extern "C" {

int32_t so_abi_version(void) { return SO_ABI_VERSION; }

const char* so_last_error(void) { return g_err.c_str(); }

int32_t so_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t so_plan_create(const so_node_t* nodes, int32_t n_nodes, int32_t root,
                       const so_out_desc_t* out, int32_t device, so_plan_t** plan) {
    if (!plan) return set_err(SO_ERR_INVALID, "so_plan_create: null plan pointer");
    *plan = nullptr;
    int status = SO_OK;
    std::string err;
    so::Plan* p = so::plan_create(nodes, n_nodes, root, out, device, status, err);
    if (!p) return set_err(status, err);
    *plan = new so_plan{p};
    return SO_OK;
}

int64_t so_plan_nframes(const so_plan_t* plan) {
    return plan ? so::plan_nframes(plan->p) : SO_LEN_MISSING;
}

int32_t so_plan_execute(so_plan_t* plan, void* out, void* hip_stream) {
    if (!plan) return set_err(SO_ERR_INVALID, "so_plan_execute: null plan");
    std::string err;
    int st = so::plan_execute(plan->p, out, hip_stream, err);
    if (st != SO_OK) return set_err(st, err);
    return SO_OK;
}

int32_t so_plan_check(so_plan_t* plan, void* hip_stream) {
    if (!plan) return set_err(SO_ERR_INVALID, "so_plan_check: null plan");
    std::string err;
    int st = so::plan_check(plan->p, hip_stream, err);
    if (st != SO_OK) return set_err(st, err);
    return SO_OK;
}

int32_t so_plan_set_array(so_plan_t* plan, int32_t node_index, const void* data) {
    if (!plan) return set_err(SO_ERR_INVALID, "so_plan_set_array: null plan");
    std::string err;
    int st = so::plan_set_array(plan->p, node_index, data, err);
    if (st != SO_OK) return set_err(st, err);
    return SO_OK;
}

int32_t so_plan_stats(const so_plan_t* plan, so_stats_t* stats) {
    if (!plan || !stats) return set_err(SO_ERR_INVALID, "so_plan_stats: null argument");
    so::plan_stats(plan->p, stats);
    return SO_OK;
}

int32_t so_plan_set_profiling(so_plan_t* plan, int32_t enable) {
    if (!plan) return set_err(SO_ERR_INVALID, "so_plan_set_profiling: null plan");
    so::plan_set_profiling(plan->p, enable);
    return SO_OK;
}

int32_t so_rtc_compile_check(const char* body, char* log, int32_t log_capacity) {
    if (!body) return set_err(SO_ERR_INVALID, "so_rtc_compile_check: null source");
    std::string err;
    const int st = so::rtc_compile_check(body, err);
    if (log && log_capacity > 0) std::snprintf(log, (size_t)log_capacity, "%s", err.c_str());
    return st == 0 ? SO_OK : set_err(SO_ERR_UNSUPPORTED, err);
}

int32_t so_rtc_wait_idle(void) {
    so::rtc_wait_idle();
    return SO_OK;
}

int32_t so_rtc_shutdown(void) {
    so::rtc_shutdown();
    return SO_OK;
}

int64_t so_plan_counter(const so_plan_t* plan, int32_t which) {
    if (!plan) return -1;
    return so::plan_counter(plan->p, which);
}

int32_t so_plan_step_info(const so_plan_t* plan, int32_t index, so_step_info_t* info) {
    if (!plan) return set_err(SO_ERR_INVALID, "so_plan_step_info: null plan");
    return so::plan_step_info(plan->p, index, info);
}

void so_plan_destroy(so_plan_t* plan) {
    if (!plan) return;
    so::plan_destroy(plan->p);
    delete plan;
}

int32_t so_design_iir(int32_t type, double f1, double f2, double fs, int32_t method,
                      int32_t order, double ripple_db, double* sos, int32_t sos_capacity,
                      int32_t* nsections, double* gain) {
    std::vector<double> s;
    double g = 1.0;
    std::string err;
    int st = so::design_iir(type, f1, f2, fs, method, order, ripple_db, s, g, err);
    if (st != SO_OK) return set_err(st, err);
    if (!sos || !nsections || !gain || (int)s.size() > sos_capacity)
        return set_err(SO_ERR_INVALID, "so_design_iir: output buffer too small");
    std::memcpy(sos, s.data(), s.size() * sizeof(double));
    *nsections = (int32_t)(s.size() / 6);
    *gain = g;
    return SO_OK;
}

int32_t so_design_iir_zpk(int32_t type, double f1, double f2, double fs, int32_t method, int32_t order,
                          double ripple_db, double* z, int32_t* nz, double* p, int32_t* np, int32_t capacity,
                          double* k) {
    std::vector<double> zz, pp;
    double kk = 1.0;
    std::string err;
    int st = so::design_iir_zpk(type, f1, f2, fs, method, order, ripple_db, zz, pp, kk, err);
    if (st != SO_OK) return set_err(st, err);
    if (!z || !p || !nz || !np || !k || (int)zz.size() > 2 * capacity || (int)pp.size() > 2 * capacity)
        return set_err(SO_ERR_INVALID, "so_design_iir_zpk: output buffer too small");
    std::memcpy(z, zz.data(), zz.size() * sizeof(double));
    std::memcpy(p, pp.data(), pp.size() * sizeof(double));
    *nz = (int32_t)(zz.size() / 2);
    *np = (int32_t)(pp.size() / 2);
    *k = kk;
    return SO_OK;
}

int32_t so_zpk_to_sos(const double* z, int32_t nz, const double* p, int32_t np, double k, double* sos,
                      int32_t sos_capacity, int32_t* nsections, double* gain) {
    std::vector<double> s;
    double g = 1.0;
    std::string err;
    if ((nz > 0 && !z) || (np > 0 && !p)) return set_err(SO_ERR_INVALID, "so_zpk_to_sos: null roots");
    int st = so::zpk_to_sos(z, nz, p, np, k, s, g, err);
    if (st != SO_OK) return set_err(st, err);
    if (!sos || !nsections || !gain || (int)s.size() > sos_capacity)
        return set_err(SO_ERR_INVALID, "so_zpk_to_sos: output buffer too small");
    std::memcpy(sos, s.data(), s.size() * sizeof(double));
    *nsections = (int32_t)(s.size() / 6);
    *gain = g;
    return SO_OK;
}

int32_t so_tf_to_sos(const double* b, int32_t nb, const double* a, int32_t na, double* sos, int32_t sos_capacity,
                     int32_t* nsections, double* gain, double* residual) {
    std::vector<double> s;
    double g = 1.0, r = 0.0;
    std::string err;
    int st = so::tf_to_sos(b, nb, a, na, s, g, r, err);
    if (st != SO_OK) return set_err(st, err);
    if (!sos || !nsections || !gain || (int)s.size() > sos_capacity)
        return set_err(SO_ERR_INVALID, "so_tf_to_sos: output buffer too small");
    std::memcpy(sos, s.data(), s.size() * sizeof(double));
    *nsections = (int32_t)(s.size() / 6);
    *gain = g;
    if (residual) *residual = r;
    return SO_OK;
}

int32_t so_tf_zero_input(const double* b, int32_t nb, const double* a, int32_t na, const double* si, int32_t nsi,
                         double* out, int64_t capacity, int64_t* nframes) {
    std::string err;
    int64_t used = 0;
    if (!out || !nframes || capacity < 0) return set_err(SO_ERR_INVALID, "so_tf_zero_input: null output");
    int st = so::tf_zero_input(b, nb, a, na, si, nsi, out, capacity, used, err);
    if (st != SO_OK) return set_err(st, err);
    *nframes = used;
    return SO_OK;
}

static int32_t copy_taps(const std::vector<double>& h, double* out, int32_t capacity, int32_t* hlen) {
    if (!hlen) return set_err(SO_ERR_INVALID, "null hlen");
    *hlen = (int32_t)h.size();
    if (!out) return SO_OK;
    if ((int)h.size() > capacity) return set_err(SO_ERR_INVALID, "tap buffer too small");
    std::memcpy(out, h.data(), h.size() * sizeof(double));
    return SO_OK;
}

int32_t so_design_resample_rational(int64_t num, int64_t den, double* h, int32_t capacity,
                                    int32_t* hlen) {
    std::vector<double> t;
    std::string err;
    int st = so::design_resample_rational(num, den, t, err);
    if (st != SO_OK) return set_err(st, err);
    return copy_taps(t, h, capacity, hlen);
}

int32_t so_design_resample_arbitrary(double rate, int32_t nphi, double* h, int32_t capacity,
                                     int32_t* hlen) {
    std::vector<double> t;
    std::string err;
    int st = so::design_resample_arbitrary(rate, nphi, t, err);
    if (st != SO_OK) return set_err(st, err);
    return copy_taps(t, h, capacity, hlen);
}

int32_t so_resample_positions(double fs_in, double fs_out, double rate, int32_t nphi, const double* h,
                              int32_t hlen, int64_t n_out, int64_t* j, int32_t* p, double* alpha,
                              int64_t* nfix, int64_t* nbaked) {
    if (!h || hlen < 1 || nphi < 1 || !(rate > 0) || n_out < 0 || !j || !p || !alpha)
        return set_err(SO_ERR_INVALID, "so_resample_positions: bad arguments");
    return so::resample_positions(fs_in, fs_out, rate, nphi, h, hlen, n_out, j, p, alpha, nfix, nbaked);
}

}  // extern "C"
