// Host-side planner interface (internal).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sigops.h"

namespace so {

int design_iir(int type, double f1, double f2, double fs, int method, int order, double ripple,
               std::vector<double>& sos, double& gain, std::string& err);
int design_iir_zpk(int type, double f1, double f2, double fs, int method, int order, double ripple,
                   std::vector<double>& z, std::vector<double>& p, double& k, std::string& err);
int zpk_to_sos(const double* z, int nz, const double* p, int np, double k, std::vector<double>& sos, double& gain,
               std::string& err);
int tf_to_sos(const double* b, int nb, const double* a, int na, std::vector<double>& sos, double& gain, double& resid,
              std::string& err);
int tf_zero_input(const double* b, int nb, const double* a, int na, const double* si0, int nsi, double* out, int64_t cap,
                  int64_t& used, std::string& err);
int design_resample_rational(int64_t num, int64_t den, std::vector<double>& h, std::string& err);
int design_resample_arbitrary(double rate, int nphi, std::vector<double>& h, std::string& err);

int resample_positions(double fs_in, double fs_out, double rate, int nphi, const double* h, int hlen,
                       int64_t n_out, int64_t* j, int32_t* p, double* alpha, int64_t* nfix, int64_t* nbaked);

int rtc_compile_check(const std::string& body, std::string& err);  // rtc.cpp
void rtc_wait_idle();                                                // rtc.cpp
void rtc_shutdown();                                                 // rtc.cpp

struct Plan;
Plan* plan_create(const so_node_t* nodes, int32_t n_nodes, int32_t root, const so_out_desc_t* out,
                  int32_t device, int& status, std::string& err);
int plan_execute(Plan* p, void* out, void* stream, std::string& err);
int plan_check(Plan* p, void* stream, std::string& err);
int plan_set_array(Plan* p, int32_t node_index, const void* data, std::string& err);
int64_t plan_nframes(const Plan* p);
void plan_stats(const Plan* p, so_stats_t* st);
void plan_set_profiling(Plan* p, int mode);
int64_t plan_counter(const Plan* p, int which);
int plan_step_info(const Plan* p, int index, so_step_info_t* info);
void plan_destroy(Plan* p);

}  // namespace so
