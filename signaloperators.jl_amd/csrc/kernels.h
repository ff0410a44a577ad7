// Host-callable launchers of the HIP kernels in k_pointwise.hip, k_sos.hip, k_resample.hip and kernels2.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include "sigops_internal.h"

namespace so {
void launch_pointwise(const DPiece* d_pieces, int npieces, int64_t nblocks, const DOp* d_ops,
                      const DLeaf* d_leaves, OutView out, bool deep, hipStream_t st, bool chain = false, bool il = false);
// returns number of kernel launches
int launch_sos_poison(void* y, const SosGeom& g, hipStream_t st);
int launch_fill_u32(void* p, size_t n, uint32_t v, hipStream_t st);  // (returns the launch's hipError_t)  // (instead of hipMemsetAsync: see k_sos.hip)
int launch_sos_poison_batch(const SosDesc* desc, int n, int dtype, hipStream_t st);
int launch_sos_batch(const SosDesc* desc, int n, int nsec, int dtype, const int64_t* total, hipStream_t st);
int launch_sos(const void* x, void* y, double* v, double* s0, const double* mpow,
               const SosGeom& g, const SosCoefs& cf, hipStream_t st);
// one pass of the three-pass IIR on its own: phase 1 (chunk end states from zero state -> v) or
// phase 3 (outputs from the chunk start states s0)
int launch_sos_phase(const void* x, void* y, double* v, const double* s0, const SosGeom& g, const SosCoefs& cf, int phase,
                     hipStream_t st);
// exact scan between them (kernels2.hip): s0[k+1] = M s0[k] + v[k] over ALL earlier chunks, no 2^-70 cut;
// mats = [M = A^L][MB = M^kXsBlock] (D x D each, row-major), sblk = [nch][nblocks][16] scratch.  3 launches.
constexpr int kXsBlock = 64;
int launch_sos_xscan(const double* v, double* s0, const double* mats, double* sblk, const SosGeom& g, int nsec, hipStream_t st);
// SOS IIR in the reference's order of operations, one sequence per channel (kernels2.hip); a: sections
// 1..8, b: sections 9..16 (b.nsec == 0: none).  0 when launched, -1: no instantiation
int launch_sos_exact(const void* x, void* y, const SosGeom& g, const SosCoefs& a, const SosCoefs& b, hipStream_t st);
// SOS IIR whose state pass was done by the resampler in front (vper: [nch][nper][16])
int launch_sos_prestate(const void* x, void* y, const double* vper, int64_t nper, const double* qmat, int pt,
                        double* v, double* s0, const double* mpow, const SosGeom& g, const SosCoefs& cf,
                        hipStream_t st);
// single-pass SOS IIR (the caller zeroes `sync` on the stream before every launch)
void launch_sos_onepass(const void* x, void* y, const SosOne& g, const SosCoefs& cf, const double* tabs,
                        int* sync, double* vpub, int dtype, hipStream_t st);
void launch_resample(const void* x, void* y, const double* pfb, const double* dpfb,
                     const RsGeom& g, hipStream_t st);
// pfbt / dpfbt: polyphase tables transposed to [taps][nphi]
size_t resample_arb_lds_bytes(int taps, int zrows, int ct, int ringf, int esz);
int launch_resample_arb(const void* x, void* y, const double* pfbt, const double* dpfbt, const RsArb& a, hipStream_t st);
void launch_resample_tiled2(const void* x, void* y, const double* pfbt, const double* dpfbt, const RsTiled& g, hipStream_t st);
void launch_resample_tiled(const void* x, void* y, const double* pfbt, const double* dpfbt, const RsTiled& g,
                           hipStream_t st);
// returns 0 when launched, -1 when no instantiation fits the geometry
int launch_resample_rows(const void* x, void* y, const double* ctab, const int* jr, const double* mtab,
                         const int* jend, const RsRows& g, int dtype, hipStream_t st);
int launch_resample_periodic(void* y, const double* tab, const int* jend, const RsPeriodic& g,
                             int dtype, const RsGlobalTables& gsrc, hipStream_t st);
void launch_resample_fix(const RsFixArgs& a, hipStream_t st);
// fused periodic resampler -> SOS IIR (k_rsos.hip): 0 when launched, -1 when no instantiation fits
int launch_rsos_fixup(const RsFixup& fx, hipStream_t st);  // k_exact.hip
int launch_rs_fixup(const RsPerFixup& fx, hipStream_t st);   // k_exact.hip
int launch_rsos_batch(const RsosItem* items, int nitems, int gpm, const RsSos& g0, hipStream_t st);
int launch_rsos_fixup_batch(const RsFixup* items, int nitems, int nch, int out_f32, hipStream_t st);  // k_exact.hip
int launch_rsos(const double* tab, const int* jend, const RsSos& g, void* y, const RsGlobalTables& gsrc, int grid, hipStream_t st);
size_t rsos_lds_bytes(int ngroups, int ks, int rpitch, int nwaves, int cyc);
size_t rsos_lds_budget();
void launch_rms(const void* x, int dtype, int64_t n, int nch, int64_t pitch, double* partial,
                int nparts, double* rms, hipStream_t st, const RmsPatch& patch = RmsPatch{});
}  // namespace so
