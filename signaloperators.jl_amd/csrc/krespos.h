// Output positions of the polyphase resamplers (shared by k_resample.hip and kernels2.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "sigops_internal.h"

namespace so {

// ---------------------------------------------------------------------------
// K3: polyphase resampler.  Output m sits at fine-grid position q_m (SURVEY.md
// Appendix A): j = newest input, p = phase, alpha = fractional phase;
//   y[m] = sum_k pfb[p][k] x[j-k]  +  alpha * sum_k dpfb[p][k] x[j-k]
// (DSP.jl FIRArbitrary: yLower + yUpper*alpha; rational kernels have alpha == 0).
// Position arithmetic is bit-exact with the oracle: two separately rounded fp64
// operations (no FMA contraction) or pure int64.
__device__ __forceinline__ void rs_pos(const RsGeom& g, int64_t m, int64_t& j, int& p,
                                       double& alpha) {
    if (g.arbitrary && g.exact) {
        const int64_t N = m * ((int64_t)g.nphi * g.M);
        const int64_t qi = g.c0i + N / g.L;
        alpha = __ddiv_rn((double)(N % g.L), (double)g.L);
        j = qi / g.nphi;
        p = (int)(qi % g.nphi);
    } else if (g.arbitrary) {
        const double t = __dmul_rn((double)m, g.delta);
        const double q = __dadd_rn(g.c0, t);
        const double fl = floor(q);
        const int64_t qi = (int64_t)fl;
        alpha = q - fl;
        if (g.nphi == 32) {  // (DSP.jl's N_phi; positions are never negative: a shift, not a 64-bit division)
            j = (qi >> 5) - g.j0;
            p = (int)(qi & 31);
        } else {
            j = qi / g.nphi - g.j0;
            p = (int)(qi % g.nphi);
        }
    } else {
        const int64_t qi = g.c0i + m * g.M;
        alpha = 0.0;
        j = qi / g.L;
        p = (int)(qi % g.L);
    }
}

}  // namespace so
