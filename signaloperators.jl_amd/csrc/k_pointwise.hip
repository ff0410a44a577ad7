// Hand-written HIP kernels for gfx950 (MI355X, CDNA4; wave64).  No CUDA shims, no dual paths.
// K1 k_pointwise: fused generator / map / ramp / index kernel (replaces frame() recursion + sink_helper!,
// reference src/sink.jl:256-260, src/mapsignal.jl:249-272); K4 k_sumsq_*: Normpower reduction (src/filters.jl:296-309)
#include "kcommon.h"

namespace so {

// K1: one workgroup = kBlock*E consecutive frames x a channel chunk of one piece.
// Lane l handles frames base + l + e*kBlock, so every load/store instruction is a
// fully coalesced run of 64 consecutive elements per wave.
// DEEP == false: every piece of the launch needs a stack depth <= 2 (left-fold chains: almost
// every tree), so the 4-deep interpreters are not even compiled in -- half the registers, twice
// the waves per SIMD, and this kernel is bound by bytes in flight.
// CHAIN: pieces whose per-sample program is `array (op) F_s (op) F_t ...` (Amplify / Mix / Ramp chains
// over one array: the commonest maps) skip the interpreter in the channel loop: the program is
// decoded once into scalar registers and eight channels' 16-byte loads are issued back to back,
// so a lane has 128 bytes in flight instead of 16 (the interpreter issues one load per channel
// pass and then waits for it: K1 was bound by bytes in flight).
// IL (with CHAIN): interleaved frames -- a result or a leaf with frame_stride = nch, chan_stride = 1
// (WAV buffers, `PermutedDimsArray` inputs; reference src/WAV.jl:3-6, src/AxisArrays.jl:38-39) -- go
// through an LDS tile of 512 frames x 8 channels: global accesses are runs of consecutive
// elements across the workgroup (whole frames when the piece has <= 8 channels), the lanes pick
// their (frame pair, channel) values out of LDS.  Without it a lane's accesses are nch elements
// apart and every 16-byte access moves a 64-byte sector.
template <int E, bool DEEP, bool CHAIN = false, bool IL = false>
__global__ __launch_bounds__(kBlock) void k_pointwise(const DPiece* __restrict__ pieces,
                                                      int npieces, const DOp* __restrict__ ops,
                                                      const DLeaf* __restrict__ leaves,
                                                      OutView out) {
    const int64_t bid = blockIdx.x;
    int lo = 0, hi = npieces - 1;
    while (lo < hi) {  // wave-uniform binary search: piece owning this workgroup
        int mid = (lo + hi + 1) >> 1;
        if (pieces[mid].block0 <= bid) lo = mid;
        else hi = mid - 1;
    }
    const DPiece P = pieces[lo];
    const int64_t rel = bid - P.block0;
    const int64_t bf = rel % P.nblk_f;
    const int bc = (int)(rel / P.nblk_f);
    const int cbeg = P.c0 + bc * P.chc;
    const int cend = min(P.c1, cbeg + P.chc);
    // A workgroup walks P.sub consecutive blocks of kBlock*E frames: the piece lookup above and the
    // program fetches are chains of dependent scalar loads (~a microsecond while the chip streams),
    // paid once per workgroup instead of once per 64 KB.
    for (int sb = 0; sb < P.sub; ++sb) {
    int64_t n[E], ns[E];
    bool valid[E];
    // light variant, block entirely inside the piece: lane l owns the PAIR of frames
    // (base + 2l, base + 2l + 1) and reads / writes it as one 16-byte access where alignment allows
    const int64_t blk0 = P.a + (bf * P.sub + sb) * (int64_t)(kBlock * E);
    if (blk0 >= P.b) break;
    const bool pair = !DEEP && E == 2 && blk0 + kBlock * E <= P.b;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        ns[e] = pair ? blk0 + (int64_t)E * threadIdx.x + e : blk0 + threadIdx.x + (int64_t)e * kBlock;
        valid[e] = ns[e] < P.b;
        n[e] = valid[e] ? ns[e] : P.b - 1;  // clamp: loads stay in range, store is skipped
    }
    double F[kMaxFrameSlots][E];
#pragma unroll
    for (int k = 0; k < kMaxFrameSlots; ++k)
#pragma unroll
        for (int e = 0; e < E; ++e) F[k][e] = 0.0;
    double v[E];
    const bool deep = DEEP && P.depth > 2;  // wave-uniform
    if (P.frame_len > 0) {
        if constexpr (DEEP) {
            if (deep) run_program<E, false, kStackDepth, true>(ops, P.frame_pc, P.frame_len, leaves, n, cbeg, F, v);
            else run_program<E, false, 2, true>(ops, P.frame_pc, P.frame_len, leaves, n, cbeg, F, v);
        } else run_program<E, false, 2, true>(ops, P.frame_pc, P.frame_len, leaves, n, cbeg, F, v);
    }
    if constexpr (CHAIN && !DEEP && E == 2) {
        constexpr int kIlPitch = 9;  // doubles per frame row of the LDS tile (8 channels + 1: bank spread)
        __shared__ double il_tile[IL ? kBlock * E * kIlPitch : 1];
        const bool out_il = IL && out.fstride > 1 && out.cstride == 1;
        if (P.chain && pair && (out.fstride == 1 || out_il)) {  // (wave-uniform)
            const DLeaf& L = leaves[ops[P.samp_pc].arg];
            const bool in_il = IL && L.fstride > 1 && L.cstride == 1;
            const int nst = (P.samp_len - 1) >> 1;
            int sop[4], sslot[4];  // operand: frame slot 0..3, or 4 = the constant cval[i]
            double cval[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const DOp o = ops[P.samp_pc + 1 + 2 * (i < nst ? i : 0)];
                sslot[i] = i < nst ? (o.code == OP_LOADF ? o.arg : 4) : 0;
                // (a device scalar -- a Normpower's rms -- is a constant of the launch: read once here, from the leaf its
                //  producer patched (RmsPatch) or through its pointer)
                cval[i] = i < nst && o.code == OP_CONST ? leaves[o.arg].v0
                          : i < nst && o.code == OP_SCALAR ? (leaves[o.arg].flag ? leaves[o.arg].v0 : scalar_leaf(leaves[o.arg].base)) : 0.0;
                sop[i] = i < nst ? ops[P.samp_pc + 2 + 2 * i].code : -1;
            }
            const bool in64 = L.dtype == SO_F64, out64 = out.dtype == SO_F64;
            const int isz = in64 ? 8 : 4, osz = out64 ? 8 : 4;
            constexpr int CB = 8;
            for (int cb = cbeg; cb < cend; cb += CB) {
                double val[CB][2];
                // ---- loads of up to eight channels, all in flight together ----
                const int nb = cend - cb < CB ? cend - cb : CB;  // channels of this batch
                if (in_il) {
                    // the batch's 512 x nb block of the interleaved leaf, element runs of nb per frame
                    // (every wave moves and reads only ITS 128 frames of the tile: wave barriers suffice,
                    //  the four waves of the workgroup stay independent)
                    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, f0w = wv * 64 * E;
                    const int64_t off0 = (blk0 + f0w + L.df) * L.fstride + ((int64_t)L.sc * cb + L.dc);
                    if (in64 && nb == L.fstride && !(nb & 1) && ((((uintptr_t)L.base) + off0 * 8) & 15) == 0) {
                        // whole frames: the block is one contiguous run -> 16-byte loads, all in flight
                        const double2* src = reinterpret_cast<const double2*>((const double*)L.base + off0);
                        const int nv = 64 * E * nb / 2;
                        for (int v0 = ln; v0 < nv; v0 += 8 * 64) {
                            double2 w[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (v0 + j * 64 < nv) w[j] = src[v0 + j * 64];
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (v0 + j * 64 < nv) {
                                    const int e0 = 2 * (v0 + j * 64), f = f0w + e0 / nb, cc = e0 % nb;
                                    il_tile[f * kIlPitch + cc] = w[j].x;
                                    il_tile[f * kIlPitch + cc + 1] = w[j].y;
                                }
                        }
                    } else {
                        for (int idx = ln; idx < 64 * E * nb; idx += 64) {
                            const int f = idx / nb, cc = idx - f * nb;
                            const int64_t off = off0 + (int64_t)f * L.fstride + (int64_t)L.sc * cc;
                            il_tile[(f0w + f) * kIlPitch + cc] = in64 ? ((const double*)L.base)[off] : (double)((const float*)L.base)[off];
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int u = 0; u < CB; ++u) {
                        val[u][0] = u < nb ? il_tile[(2 * threadIdx.x) * kIlPitch + u] : 0.0;
                        val[u][1] = u < nb ? il_tile[(2 * threadIdx.x + 1) * kIlPitch + u] : 0.0;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int u = 0; u < CB; ++u) {
                    if (in_il) break;
                    val[u][0] = val[u][1] = 0.0;
                    if (cb + u < cend) {
                        const int64_t off = ((int64_t)L.sc * (cb + u) + L.dc) * L.cstride + ns[0] + L.df;
                        const char* pa = (const char*)L.base + off * isz;
                        const uintptr_t a0 = (uintptr_t)rfl64((int64_t)(uintptr_t)pa);  // lane 0's address
                        if (in64) {
                            if ((a0 & 15) == 0) {
                                const double2 w = *reinterpret_cast<const double2*>(pa);
                                val[u][0] = w.x;
                                val[u][1] = w.y;
                            } else {
                                val[u][0] = ((const double*)pa)[0];
                                val[u][1] = ((const double*)pa)[1];
                            }
                        } else if ((a0 & 7) == 0) {
                            const float2 w = *reinterpret_cast<const float2*>(pa);
                            val[u][0] = (double)w.x;
                            val[u][1] = (double)w.y;
                        } else {
                            val[u][0] = (double)((const float*)pa)[0];
                            val[u][1] = (double)((const float*)pa)[1];
                        }
                    }
                }
                // ---- the chain: the opcode switch outside the element loops ----
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i >= nst) break;
                    double m[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        m[e] = sslot[i] == 0 ? F[0][e] : sslot[i] == 1 ? F[1][e] : sslot[i] == 2 ? F[2][e] : sslot[i] == 3 ? F[3][e] : cval[i];
#define SO_CH(EXPR)                                          \
    _Pragma("unroll") for (int u = 0; u < CB; ++u) _Pragma("unroll") for (int e = 0; e < 2; ++e) { \
        const double x = val[u][e];                          \
        val[u][e] = (EXPR);                                  \
    }
                    switch (sop[i]) {
                    case OP_ADD: SO_CH(x + m[e]) break;
                    case OP_SUB: SO_CH(x - m[e]) break;
                    case OP_MUL: SO_CH(x * m[e]) break;
                    default: SO_CH(x / m[e]) break;
                    }
#undef SO_CH
                }
                // ---- stores ----
                if (out_il) {
#pragma unroll
                    for (int u = 0; u < CB; ++u)
                        if (u < nb) {
                            il_tile[(2 * threadIdx.x) * kIlPitch + u] = val[u][0];
                            il_tile[(2 * threadIdx.x + 1) * kIlPitch + u] = val[u][1];
                        }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, f0w = wv * 64 * E;
                    const int64_t ooff0 = (blk0 + f0w) * out.fstride + cb;
                    if (!(out.pad & 1) && out64 && nb == out.fstride && !(nb & 1) && ((((uintptr_t)out.base) + ooff0 * 8) & 15) == 0) {
                        double2* dst = reinterpret_cast<double2*>((double*)out.base + ooff0);
                        const int nv = 64 * E * nb / 2;
                        for (int v0 = ln; v0 < nv; v0 += 64) {
                            const int e0 = 2 * v0, f = f0w + e0 / nb, cc = e0 % nb;
                            double2 w;
                            w.x = il_tile[f * kIlPitch + cc];
                            w.y = il_tile[f * kIlPitch + cc + 1];
                            dst[v0] = w;
                        }
                    } else {
                        for (int idx = ln; idx < 64 * E * nb; idx += 64) {
                            const int f = idx / nb, cc = idx - f * nb;
                            const int64_t off = ooff0 + (int64_t)f * out.fstride + cc;
                            const double w = il_tile[(f0w + f) * kIlPitch + cc];
                            if (out64) ((double*)out.base)[off] = w;
                            else ((float*)out.base)[off] = (float)w;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
#pragma unroll
                for (int u = 0; u < CB; ++u) {
                    if (cb + u < cend) {
                        const int64_t off = (int64_t)(cb + u) * out.cstride + ns[0];
                        char* pa = (char*)out.base + off * osz;
                        const uintptr_t a0 = (uintptr_t)rfl64((int64_t)(uintptr_t)pa);
                        if (out64) {
                            if ((a0 & 15) == 0) {
                                double2 w;
                                w.x = val[u][0];
                                w.y = val[u][1];
                                *reinterpret_cast<double2*>(pa) = w;
                            } else {
                                ((double*)pa)[0] = val[u][0];
                                ((double*)pa)[1] = val[u][1];
                            }
                        } else if ((a0 & 7) == 0) {
                            float2 w;
                            w.x = (float)val[u][0];
                            w.y = (float)val[u][1];
                            *reinterpret_cast<float2*>(pa) = w;
                        } else {
                            ((float*)pa)[0] = (float)val[u][0];
                            ((float*)pa)[1] = (float)val[u][1];
                        }
                    }
                }
            }
            continue;
        }
    }
    for (int c = cbeg; c < cend; ++c) {
        if constexpr (DEEP) {
            if (deep) run_program<E, false, kStackDepth, false>(ops, P.samp_pc, P.samp_len, leaves, n, c, F, v);
            else run_program<E, false, 2, false>(ops, P.samp_pc, P.samp_len, leaves, n, c, F, v);
        } else run_program<E, false, 2, false, true>(ops, P.samp_pc, P.samp_len, leaves, n, c, F, v, pair);
        if constexpr (!DEEP && E == 2) {
            if (pair && out.fstride == 1) {  // aligned pair store (wave-uniform alignment)
                const int64_t off = (int64_t)c * out.cstride + ns[0];
                const int par = __builtin_amdgcn_readfirstlane((int)off) & 1;
                if (out.dtype == SO_F64 && ((uintptr_t)out.base & 7) == 0 && ((((uintptr_t)out.base) >> 3) & 1) == (uintptr_t)par) {
                    double2 w;
                    w.x = v[0];
                    w.y = v[1];
                    *reinterpret_cast<double2*>((double*)out.base + off) = w;
                    continue;
                }
                if (out.dtype == SO_F32 && ((uintptr_t)out.base & 3) == 0 && ((((uintptr_t)out.base) >> 2) & 1) == (uintptr_t)par) {
                    float2 w;
                    w.x = (float)v[0];
                    w.y = (float)v[1];
                    *reinterpret_cast<float2*>((float*)out.base + off) = w;
                    continue;
                }
            }
        }
        if (out.dtype == SO_F32) {
            float* o = (float*)out.base + (int64_t)c * out.cstride;
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (valid[e]) o[ns[e] * out.fstride] = (float)v[e];
        } else {
            double* o = (double*)out.base + (int64_t)c * out.cstride;
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (valid[e]) o[ns[e] * out.fstride] = v[e];
        }
    }
    }  // sub-blocks
}

void launch_pointwise(const DPiece* d_pieces, int npieces, int64_t nblocks, const DOp* d_ops,
                      const DLeaf* d_leaves, OutView out, bool deep, hipStream_t st, bool chain, bool il) {
    if (nblocks <= 0) return;
    if (chain && !deep && il)
        hipLaunchKernelGGL((k_pointwise<kPointwiseE, false, true, true>), dim3((unsigned)nblocks), dim3(kBlock), 0, st, d_pieces,
                           npieces, d_ops, d_leaves, out);
    else if (chain && !deep)
        hipLaunchKernelGGL((k_pointwise<kPointwiseE, false, true>), dim3((unsigned)nblocks), dim3(kBlock), 0, st, d_pieces,
                           npieces, d_ops, d_leaves, out);
    else if (deep)
        hipLaunchKernelGGL((k_pointwise<kPointwiseE, true>), dim3((unsigned)nblocks), dim3(kBlock), 0, st, d_pieces,
                           npieces, d_ops, d_leaves, out);
    else
        hipLaunchKernelGGL((k_pointwise<kPointwiseE, false>), dim3((unsigned)nblocks), dim3(kBlock), 0, st, d_pieces,
                           npieces, d_ops, d_leaves, out);
}

// ---------------------------------------------------------------------------
// K4: sum of squares over a planar [nch][pitch] buffer with n valid frames per
// channel; deterministic two-stage tree (no atomics), fp64 accumulation.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sumsq_partial(const T* __restrict__ x, int64_t n,
                                                          int nch, int64_t pitch,
                                                          double* __restrict__ partial) {
    __shared__ double red[kBlock / 64];
    const int64_t total = n * nch;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    double acc = 0.0;
    // (channel, frame) of this thread's elements kept by increments: a 64-bit division per element cost more than its load
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int64_t ch = i / n, f = i - ch * n;
    constexpr int U = 4;  // loads in flight; the sums stay in element order
    for (; i + (U - 1) * stride < total; i += U * stride) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = (double)x[ch * pitch + f];
            f += stride;
            if (f >= n) {
                const int64_t k = f / n;
                f -= k * n;
                ch += k;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u] * v[u];
    }
    for (; i < total; i += stride) {
        const double v = (double)x[ch * pitch + f];
        acc += v * v;
        f += stride;
        if (f >= n) {
            const int64_t k = f / n;
            f -= k * n;
            ch += k;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) s += red[w];
        partial[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(kBlock) void k_sumsq_final(const double* __restrict__ partial,
                                                        int nparts, double count,
                                                        double* __restrict__ rms, RmsPatch patch) {
    __shared__ double red[kBlock];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += kBlock) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double r = sqrt(red[0] / count);
        rms[0] = r;
        for (int i = 0; i < patch.n; ++i) *patch.dst[i] = r;  // (the scalar leaves that read it: RmsPatch)
    }
}

// Float32 signals: Julia reduces `mean(x -> float(x)^2, vals)` in Float32 -- pairwise over blocks of
// 1024 values (Base.mapreduce_impl).  The same order as the oracle's restatement
// (oracle/sigops_oracle.c, NORMPOWER): every block summed front to back in Float32 (separate
// multiply and add), then neighbours folded level by level; rms = sqrt(sum / count) in Float32.
// One lane per block of 1024 (its sum is a chain of 1024 dependent adds), one wave per workgroup: the wave's 64 blocks are
// 256 KB of the signal in a row, read 64 values of every block at a time with whole-line requests (eight lanes per 128-byte
// line; the next chunk's requests are in flight while this one is summed) into an LDS tile the lanes then walk row by row.
// The wave then folds its 64 sums itself -- the first six levels of the neighbour tree: pairs (2i, 2i+1), an odd last one
// carried up unchanged, which for an aligned group of 64 is the same within the wave as over the whole list -- and leaves
// ONE value for k_sumsq32_fold (which, one workgroup walking every level through memory, took longer than the sums).
// (One scalar load per value and lane, 4 KB apart from its neighbours', and the whole tree in the fold kernel: 0.8 ms for
// 12.5 M x 8 values; `tools/operator_matrix.py`, Float32 `Normpower`.)
constexpr int kSqBlocks = 64, kSqChunk = 64;
__global__ __launch_bounds__(kSqBlocks) void k_sumsq32_blocks(const float* __restrict__ x, int64_t n, int nch,
                                                              int64_t pitch, float* __restrict__ part, int64_t nb) {
    __shared__ float tile[kSqBlocks][kSqChunk + 1];
    const int t = threadIdx.x;
    const int64_t total = n * nch;
    const int64_t b0 = (int64_t)blockIdx.x * kSqBlocks;
    constexpr int LPB = kSqChunk / 4;  // lanes (float4 requests) per block and chunk
    constexpr int NQ = LPB;            // requests per lane and chunk: 64 blocks x LPB requests / 64 lanes
    // request q = j * 64 + t of a chunk: block q / LPB of the wave, values (q % LPB) * 4 ... + 3 of its chunk
    int64_t qi[NQ], qf[NQ];
    int qc[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int q = j * kSqBlocks + t;
        qi[j] = (b0 + q / LPB) * 1024 + (q % LPB) * 4;
        qc[j] = (int)(qi[j] < total ? qi[j] / n : nch);
        qf[j] = qi[j] - (int64_t)qc[j] * n;
    }
    float v[NQ][4];
    // a wave whose 64 blocks lie inside one channel (all but a handful) issues a chunk's requests back to back; behind a
    // per-request edge test every load waited for the one before it
    const int64_t wlast = (b0 + kSqBlocks) * 1024 - 1;
    const bool interior = wlast < total && (b0 * 1024) / n == wlast / n;
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const f4u* src[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) src[j] = (const f4u*)(x + (interior ? (int64_t)qc[j] * pitch + qf[j] : 0));
    auto request_interior = [&]() {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const f4u w = *src[j];
            v[j][0] = w.x, v[j][1] = w.y, v[j][2] = w.z, v[j][3] = w.w;
            src[j] += kSqChunk / 4;
        }
    };
    // a wave that holds the signal's end or a channel boundary (rows of n >= 64 frames: at most one boundary per step):
    // every value its own predicated load, no branches, so that a chunk's loads are still in flight together -- the
    // handful of such waves otherwise ends long after all the others
    auto request_edges = [&]() {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int64_t f = qf[j] + e;
                int ch = qc[j];
                if (f >= n) f -= n, ++ch;
                const bool ok = qi[j] + e < total;
                const float w = x[ok ? (int64_t)ch * pitch + f : 0];
                v[j][e] = ok ? w : 0.f;
            }
            qi[j] += kSqChunk;
            qf[j] += kSqChunk;
            if (qf[j] >= n) qf[j] -= n, ++qc[j];
        }
    };
    auto request_short_rows = [&]() {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (qi[j] + 3 < total && qf[j] + 3 < n) {
                // (a 16-byte request from any 4-byte address: rows of arrays start where the caller put them)
                const f4u w = *(const f4u*)(x + (int64_t)qc[j] * pitch + qf[j]);
                v[j][0] = w.x, v[j][1] = w.y, v[j][2] = w.z, v[j][3] = w.w;
            } else {  // the signal's end (zeros add nothing to a sum of squares) or a channel boundary inside the four
                int64_t f = qf[j];
                int ch = qc[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    while (f >= n && ch < nch) {
                        f -= n;
                        ++ch;
                    }
                    v[j][e] = (qi[j] + e < total) ? x[(int64_t)ch * pitch + f] : 0.f;
                    ++f;
                }
            }
            qi[j] += kSqChunk;
            qf[j] += kSqChunk;
            while (qf[j] >= n && qc[j] < nch) {
                qf[j] -= n;
                ++qc[j];
            }
        }
    };
    float acc = 0.f;
    auto run = [&](auto request) {  // (the whole loop once per kind of wave: with the choice inside it the loads went back to waiting)
    request();
    for (int c = 0; c < 1024; c += kSqChunk) {
        __syncthreads();  // (the previous chunk has been read)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int q = j * kSqBlocks + t;
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[q / LPB][(q % LPB) * 4 + e] = v[j][e];
        }
        __syncthreads();
        if (c + kSqChunk < 1024) request();
#pragma unroll
        for (int k0 = 0; k0 < kSqChunk; k0 += 32) {
            float w[32];  // (read as a batch: behind a `volatile` asm every LDS read was waited for on its own)
#pragma unroll
            for (int k = 0; k < 32; ++k) w[k] = tile[t][k0 + k];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                // the square is rounded on its own (Julia's x^2, then +): __fmul_rn / __fadd_rn are plain * and + to
                // the compiler, which fuses them into v_fmac_f32 under its default contraction -- 1 ulp of the rms off
                // on two of twelve long signals
                float sq;
                asm("v_mul_f32 %0, %1, %1" : "=v"(sq) : "v"(w[k]));
                acc = acc + sq;
            }
        }
    }
    };
    if (interior) run(request_interior);
    else if (n >= kSqChunk) run(request_edges);
    else run(request_short_rows);
    // six levels of the neighbour tree inside the wave
    int valid = b0 + t < nb;
#pragma unroll
    for (int d = 1; d < kSqBlocks; d <<= 1) {
        const float other = __shfl_down(acc, d, 64);
        const int ov = __shfl_down(valid, d, 64);
        if ((t & (2 * d - 1)) == 0 && ov) acc = __fadd_rn(acc, other);
    }
    if (t == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(kBlock) void k_sumsq32_fold(float* __restrict__ a, float* __restrict__ b, int64_t nb,
                                                         float count, double* __restrict__ rms, RmsPatch patch) {
    float* in = a;
    float* out = b;
    int64_t m = nb;
    while (m > 1) {
        const int64_t h = (m + 1) / 2;
        for (int64_t i = threadIdx.x; i < m / 2; i += kBlock) out[i] = __fadd_rn(in[2 * i], in[2 * i + 1]);
        if ((m & 1) && threadIdx.x == 0) out[m / 2] = in[m - 1];
        __syncthreads();
        float* t = in;
        in = out;
        out = t;
        m = h;
    }
    // (Float32 division and square root through Float64: correctly rounded whatever the device's own
    //  single-precision sequences do -- v_sqrt_f32 alone is 1 ulp)
    if (threadIdx.x == 0) {
        const float mean = (float)((double)(nb ? in[0] : 0.f) / (double)count);
        const double r = (double)(float)sqrt((double)mean);
        rms[0] = r;
        for (int i = 0; i < patch.n; ++i) *patch.dst[i] = r;  // (the scalar leaves that read it: RmsPatch)
    }
}

void launch_rms(const void* x, int dtype, int64_t n, int nch, int64_t pitch, double* partial,
                int nparts, double* rms, hipStream_t st, const RmsPatch& patch) {
    if (dtype == SO_F32) {
        const int64_t nb = (n * nch + 1023) / 1024;
        const int64_t nw = (nb + kSqBlocks - 1) / kSqBlocks;  // one value per wave of 64 blocks goes on to the fold kernel
        float* pa = (float*)partial;
        hipLaunchKernelGGL(k_sumsq32_blocks, dim3((unsigned)nw), dim3(kSqBlocks), 0, st, (const float*)x, n, nch, pitch, pa, nb);
        hipLaunchKernelGGL(k_sumsq32_fold, dim3(1), dim3(kBlock), 0, st, pa, pa + nw, nw, (float)((double)n * (double)nch), rms, patch);
        return;
    }
    if (dtype == SO_F32)
        hipLaunchKernelGGL((k_sumsq_partial<float>), dim3(nparts), dim3(kBlock), 0, st,
                           (const float*)x, n, nch, pitch, partial);
    else
        hipLaunchKernelGGL((k_sumsq_partial<double>), dim3(nparts), dim3(kBlock), 0, st,
                           (const double*)x, n, nch, pitch, partial);
    hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(kBlock), 0, st, partial, nparts,
                       (double)n * (double)nch, rms, patch);
}

}  // namespace so
