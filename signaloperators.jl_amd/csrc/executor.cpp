// Executor: buffer allocation and table upload (finalize), multi-stream lanes, launches, HIP-graph
// replay, so_plan_set_array, statistics.
#include "plan_impl.h"

namespace so {

// May the kernels read this carrier's array 16 bytes at a time (LDS-DMA, vector loads)?  Rows only need their element's
// natural alignment: the memory pipeline takes a 16-byte access at any 8-byte address -- LDS-DMA included (measured:
// a [channels x frames] device tensor with an ODD number of frames, every second row 8 bytes off, used to send every chunk
// of K3 and of the fused kernel down the general staging path: 8 channels x 12 500 001 frames 5.06 ms, now 0.70; same
// values); Float32 rows likewise from any 4-byte address (8 ch x 12 500 001: 5.4 -> 0.77 ms).
static inline int carrier_vec_ok(const DCarrier& c, int64_t V) {
    if (!std::getenv("SIGOPS_STRICT_ALIGN")) return (uintptr_t)c.base % (c.dtype == SO_F64 ? 8 : 4) == 0;
    return ((uintptr_t)c.base % 16 == 0) && (c.cstride % V == 0);
}

// ---------------------------------------------------------------------------
// Plan tables go up in pieces of 16 KB: the runtime sets its staging path for pageable copies of 32 KB and more up on the
// first such copy of the PROCESS -- 6 ms, measured, which the 66 KB tap table of a first one-shot sink paid (a host that has
// already copied its signal to the device has paid it; one whose data was produced on the device has not).  Small copies go
// another way.  Tables of more than a megabyte are not chopped up.
static hipError_t h2d_small(void* dst, const void* src, size_t bytes) {
    constexpr size_t kPiece = 16 * 1024;
    if (bytes < 2 * kPiece || bytes > (1u << 20)) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    for (size_t off = 0; off < bytes; off += kPiece) {
        const hipError_t e = hipMemcpy((char*)dst + off, (const char*)src + off, std::min(kPiece, bytes - off), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void Plan::finalize() {
    // size stage output buffers now that every need is known
    for (size_t si = 0; si < stages.size(); ++si) {
        Stage& S = stages[si];
        if (S.out_buf >= 0 && ((int)si == alias_stage || S.win_off >= 0)) {
            bufs[S.out_buf].external = true;  // the kernel writes the final output directly
            bufs[S.out_buf].bytes = 0;
            continue;
        }
        if (S.out_buf >= 0 && (S.fused_away || S.norm_direct)) {  // (its consumer computes it on the fly: k_rsos; Normpower reading an array in place)
            bufs[S.out_buf].frames = 0;
            bufs[S.out_buf].bytes = 64;
            continue;
        }
        if (S.out_buf >= 0) {
            Buf& b = bufs[S.out_buf];
            b.frames = S.need - S.base;
            b.pitch = std::max<int64_t>(64, (S.need - S.base + 63) / 64 * 64);
            b.bytes = (size_t)b.pitch * (size_t)std::max(b.nch, 1) * dsize(b.dtype);
        }
    }
    // host array leaves get a device copy (the leaves of pointwise programs, the carriers of fused
    // resampler sources and the direct sources of stages go through the same validation)
    auto stage_host_array = [&](int an) {
        if (an < 0) return;
        const so_node_t& nd = nodes[an].nd;
        if (nd.i0 || array_buf.count(an)) return;  // device-resident, or already staged
        if (nd.s0 < 0 || nd.s1 < 0) fail(SO_ERR_UNSUPPORTED, "negative strides on host arrays are not supported");
        const size_t extent = nd.l0 > 0 ? (size_t)((nd.l0 - 1) * nd.s0 + (int64_t)(nd.nch - 1) * nd.s1 + 1) : 0;
        const int b = raw_buf(extent * dsize(nd.dtype));
        array_buf[an] = b;
        host_leaves.push_back(HostLeaf{an, nd.p0, extent * dsize(nd.dtype), b});
    };
    for (size_t i = 0; i < leaves.size(); ++i) stage_host_array(leaf_array_node[i]);
    for (auto& S : stages) {
        for (auto& c : S.carriers) {
            stage_host_array(c.array_node);
            if (car_has_arr2(c)) stage_host_array(c.array_node2);
        }
        stage_host_array(S.in_array_node);
    }
    if (!out.is_device && out.nframes > 0) {
        Buf b;
        b.frames = out.nframes;
        b.pitch = out.nframes;
        b.nch = out.nch;
        b.dtype = out.dtype;
        b.bytes = (size_t)out.nframes * out.nch * dsize(out.dtype);
        bufs.push_back(b);
        out_stage_buf = (int)bufs.size() - 1;
    }
    // allocate
    const bool dbg_t = std::getenv("SIGOPS_DEBUG_PLAN") != nullptr;
    const auto tf0 = std::chrono::steady_clock::now();
    int64_t scratch = 0;
    int nalloc = 0;
    for (auto& b : bufs) {
        if (b.external) continue;
        HIPCHECK(hipMalloc(&b.d, std::max<size_t>(b.bytes, 64)));
        scratch += (int64_t)b.bytes;
        ++nalloc;
    }
    stats.scratch_bytes = scratch;
    if (dbg_t)
        std::fprintf(stderr, "[sigops] finalize: %d hipMalloc calls, %lld bytes: %.3f ms\n", nalloc, (long long)scratch,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count());
    if (out_alias_buf >= 0 && !out.is_device) {
        bufs[out_alias_buf].d = bufs[out_stage_buf].d;
        bufs[out_alias_buf].pitch = bufs[out_stage_buf].pitch;
    }
    // patch leaves
    for (size_t i = 0; i < leaves.size(); ++i) {
        DLeaf& L = leaves[i];
        int an = leaf_array_node[i];
        if (an >= 0) {
            L.base = nodes[an].nd.i0 ? array_ptr[an] : bufs[array_buf[an]].d;
        } else if (L.buf >= 0) {
            L.base = bufs[L.buf].d;
            if (L.cstride == -1) L.cstride = bufs[L.buf].pitch;
            L.df -= bufs[L.buf].frame0;  // (stage buffers that start at a later frame: once, here)
        }
    }
    // scalar leaves of a Normpower's rms: patched on the device by the launch that computes it (RmsPatch)
    for (auto& S : stages) {
        if (S.kind != ST_NORM || S.rms_buf < 0 || std::getenv("SIGOPS_NO_RMSPATCH")) continue;
        S.rms_leaves.clear();
        for (size_t i = 0; i < leaves.size(); ++i)
            if (leaves[i].buf == S.rms_buf && leaf_array_node[i] < 0) S.rms_leaves.push_back((int)i);
        if (S.rms_leaves.size() > 8) S.rms_leaves.clear();
        for (int i : S.rms_leaves) leaves[(size_t)i].flag = 1;
    }
    for (auto& S : stages) {
        if (S.carriers.empty()) continue;
        for (auto& c : S.carriers) {
            if (c.array_node >= 0) {
                const so_node_t& nd = nodes[c.array_node].nd;
                if (!nd.i0 && !array_buf.count(c.array_node)) fail(SO_ERR_RUNTIME, "internal: carrier array without device copy");
                c.base = nd.i0 ? array_ptr[c.array_node] : bufs[array_buf[c.array_node]].d;
            } else if (c.buf >= 0) {
                c.base = bufs[c.buf].d;
                if (c.cstride == -1) c.cstride = bufs[c.buf].pitch;
                c.df -= bufs[c.buf].frame0;
            } else {
                c.base = nullptr;  // generated piece
                c.cstride = 0;
            }
            const int64_t V = 16 / (int64_t)dsize(c.dtype);
            c.vec_ok = carrier_vec_ok(c, V);
            if (car_has_arr2(c)) {  // the step's second array (a caller's: match_carrier)
                const so_node_t& nd = nodes[c.array_node2].nd;
                if (!nd.i0 && !array_buf.count(c.array_node2)) fail(SO_ERR_RUNTIME, "internal: carrier array without device copy");
                c.base2 = nd.i0 ? array_ptr[c.array_node2] : bufs[array_buf[c.array_node2]].d;
                c.vec_ok2 = (uintptr_t)c.base2 % (c.dtype2 == SO_F64 ? 8 : 4) == 0;
            } else {
                c.base2 = nullptr;
                c.array_node2 = c.buf2 = -1;
            }
        }
        if (std::getenv("SIGOPS_DEBUG_PLAN"))
            for (auto& c : S.carriers)
                std::fprintf(stderr, "[sigops] carrier [%lld,%lld) base=%p cstride=%lld df=%lld dtype=%d vec_ok=%d nsteps=%d frame_len=%d depth=%d\n",
                             (long long)c.a, (long long)c.b, c.base, (long long)c.cstride, (long long)c.df, c.dtype, c.vec_ok, c.nsteps, c.frame_len, c.depth);
        HIPCHECK(h2d_small(bufs[S.car_buf].d, S.carriers.data(), S.carriers.size() * sizeof(DCarrier)));
        {
            const RsCtl ctl = make_ctl(S);
            HIPCHECK(h2d_small(bufs[S.ctl_buf].d, &ctl, sizeof(RsCtl)));
        }
    }
    if (dbg_t)
        std::fprintf(stderr, "[sigops] finalize: leaves and carriers patched, carrier blocks uploaded: %.3f ms\n",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count());
    // upload tables
    if (!pieces.empty()) {
        HIPCHECK(hipMalloc(&d_pieces, pieces.size() * sizeof(DPiece)));
        HIPCHECK(h2d_small(d_pieces, pieces.data(), pieces.size() * sizeof(DPiece)));
    }
    if (!ops.empty()) {
        HIPCHECK(hipMalloc(&d_ops, ops.size() * sizeof(DOp)));
        HIPCHECK(h2d_small(d_ops, ops.data(), ops.size() * sizeof(DOp)));
    }
    if (!leaves.empty()) {
        HIPCHECK(hipMalloc(&d_leaves, leaves.size() * sizeof(DLeaf)));
        HIPCHECK(h2d_small(d_leaves, leaves.data(), leaves.size() * sizeof(DLeaf)));
    }
    for (auto& S : stages) {
        if (S.need <= 0) continue;
        if (S.kind == ST_SOS && S.qmat_buf >= 0)
            HIPCHECK(h2d_small(bufs[S.qmat_buf].d, S.qmat_host.data(), S.qmat_host.size() * 8));
        if (S.kind == ST_SOS && S.rsos_src >= 0) {
            HIPCHECK(h2d_small(bufs[S.rsos_mats_buf].d, S.rsos_mats_host.data(), S.rsos_mats_host.size() * 8));
            if (S.rsos_jrel_buf >= 0) HIPCHECK(h2d_small(bufs[S.rsos_jrel_buf].d, S.rsos_jrel_host.data(), S.rsos_jrel_host.size() * 4));
            if (S.rsos_tab_buf >= 0) {
                HIPCHECK(h2d_small(bufs[S.rsos_tab_buf].d, S.rsos_tab_host.data(), S.rsos_tab_host.size() * 8));
                HIPCHECK(h2d_small(bufs[S.rsos_jend_buf].d, S.rsos_jend_host.data(), S.rsos_jend_host.size() * 4));
            }
        }
        if (S.kind == ST_SOS && S.onepass)
            HIPCHECK(h2d_small(bufs[S.one_tabs_buf].d, S.one_tabs_host.data(), S.one_tabs_host.size() * 8));
        if (S.kind == ST_RESAMPLE && S.rs_jrel_buf >= 0 && S.rs_nf_buf >= 0) {
            HIPCHECK(h2d_small(bufs[S.rs_jrel_buf].d, S.rs_jrel_host.data(), S.rs_jrel_host.size() * 4));
            HIPCHECK(hipMemset(bufs[S.rs_nf_buf].d, 0, 64));  // (the list's count: k_rs_fixup leaves it at zero again)
        }
        if (S.kind == ST_RESAMPLE) {
            HIPCHECK(h2d_small(bufs[S.pfb_buf].d, S.pfb_host.data(), S.pfb_host.size() * 8));
            HIPCHECK(h2d_small(bufs[S.dpfb_buf].d, S.dpfb_host.data(), S.dpfb_host.size() * 8));
            if (S.wtab_buf >= 0)
                HIPCHECK(h2d_small(bufs[S.wtab_buf].d, S.wtab_host.data(), S.wtab_host.size() * 8));
            if (S.tiled) {
                HIPCHECK(h2d_small(bufs[S.pfbt_buf].d, S.pfbt_host.data(), S.pfbt_host.size() * 8));
                HIPCHECK(h2d_small(bufs[S.dpfbt_buf].d, S.dpfbt_host.data(), S.dpfbt_host.size() * 8));
            }
            if (S.fix_buf >= 0)
                HIPCHECK(h2d_small(bufs[S.fix_buf].d, S.fix_host.data(), S.fix_host.size() * sizeof(RsFix)));
            if (S.periodic || S.rows) {
                HIPCHECK(h2d_small(bufs[S.tab_buf].d, S.tab_host.data(), S.tab_host.size() * 8));
                HIPCHECK(h2d_small(bufs[S.jend_buf].d, S.jend_host.data(), S.jend_host.size() * 4));
            }
            if (S.rows && !S.mtab_host.empty()) {
                HIPCHECK(h2d_small(bufs[S.mtab_buf].d, S.mtab_host.data(), S.mtab_host.size() * 8));
                HIPCHECK(h2d_small(bufs[S.mjend_buf].d, S.mjend_host.data(), S.mjend_host.size() * 4));
            }
        } else if (S.kind == ST_SOS && S.mpow_buf >= 0) {
            for (size_t gi = 0; gi < S.xs_mats_host.size() && S.xs_mats_buf >= 0; ++gi)
                HIPCHECK(h2d_small((char*)bufs[S.xs_mats_buf].d + gi * 2 * 16 * 16 * 8, S.xs_mats_host[gi].data(),
                                   S.xs_mats_host[gi].size() * 8));
            size_t msz = 0;
            for (auto& v : S.mpow_host) msz = std::max(msz, v.size());
            for (size_t gi = 0; gi < S.mpow_host.size(); ++gi)
                HIPCHECK(h2d_small((char*)bufs[S.mpow_buf].d + gi * msz * 8, S.mpow_host[gi].data(),
                                   S.mpow_host[gi].size() * 8));
        }
    }
    if (dbg_t)
        std::fprintf(stderr, "[sigops] finalize: tables uploaded: %.3f ms\n",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count());
    // step list: stages in increasing node order (children first), then the root program
    std::vector<int> order;
    for (size_t i = 0; i < stages.size(); ++i)
        if (stages[i].need > 0) order.push_back((int)i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return stages[a].node < stages[b].node; });
    std::vector<char> batch_pushed(batches.size(), 0);
    std::vector<char> rsb_pushed(rsbatches.size(), 0);
    for (int sid : order) {
        Stage& S = stages[sid];
        if (S.batch >= 0) {  // one step for the whole batch, where its first member stood (members wait for nothing)
            if (batch_pushed[S.batch]) continue;
            batch_pushed[S.batch] = 1;
            Step st{2, S.batch, "k_sos_batch", 0};
            for (int m : batches[S.batch].members)
                st.bytes += 2 * (stages[m].need - stages[m].base) * stages[m].sg.nch * (int64_t)dsize(nodes[stages[m].node].dtype);
            steps.push_back(st);
            continue;
        }
        if (S.rsb >= 0) {  // ... and one for the plain filters that go through k_rsos together
            if (rsb_pushed[S.rsb]) continue;
            rsb_pushed[S.rsb] = 1;
            Step st{3, S.rsb, "k_rsos_batch", 0};
            for (int m : rsbatches[S.rsb].members) {
                const Stage& M = stages[m];
                const int64_t esz = (int64_t)dsize(nodes[M.node].dtype);
                st.bytes += (M.rs.n_in * (int64_t)dsize(M.carriers[0].dtype) + M.rs.n_out * esz) * M.rs.nch;
            }
            steps.push_back(st);
            continue;
        }
        if (S.pw_step >= 0) push_pw_step(S.pw_step);
        if (S.fused_away) continue;  // (runs inside its consumer's launch)
        const char* nm = S.kind == ST_SOS ? (S.rsos_src >= 0 ? "k_rsos" : S.sg.exact ? "k_sos_exact" : "k_sos") : S.kind == ST_RESAMPLE ? (S.periodic ? "k_resample_periodic" : S.rows ? "k_resample_rows" : S.tiled ? (S.arbk ? "k_resample_arb" : S.rt.pair ? "k_resample_tiled2" : "k_resample_tiled") : "k_resample") : "k_sumsq";
        Step st{1, sid, nm, 0};
        // algorithmic bytes of a stage: the samples it reads in the type they have WHERE THEY LIE (a Float32 array under a
        // Float64 map is 4 bytes a sample, whatever the node's promoted type) plus the samples it writes in the type of
        // the buffer they go to (a Float64 stage that rounds into the Float32 result itself writes 4)
        const int64_t esz = (int64_t)dsize(nodes[S.node].dtype);
        const int64_t osz = (sid == alias_stage && alias_narrow) ? (int64_t)dsize(out.dtype) : esz;
        auto src_esz = [&](const Stage& R) -> int64_t {  // a resampler's source: carrier 0's array / buffer, else its child
            if (!R.carriers.empty()) return (int64_t)dsize(R.carriers[0].dtype);
            const std::vector<int>& kids = nodes[R.node].kids;
            return !kids.empty() && kids[0] >= 0 ? (int64_t)dsize(nodes[kids[0]].dtype) : esz;
        };
        if (S.kind == ST_SOS && S.rsos_src >= 0) st.bytes = (S.rs.n_in * src_esz(stages[S.rsos_src]) * (S.rs.arr2 ? 2 : 1) + S.rs.n_out * osz) * S.rs.nch;  // (arr2: a second array read)
        else if (S.kind == ST_SOS) st.bytes = (S.need - S.base) * S.sg.nch * (esz + osz);
        else if (S.kind == ST_RESAMPLE) st.bytes = (S.rg.n_in * (src_esz(S) + (S.rp.arr2 ? src_esz(S) : 0)) + S.rg.n_out * osz) * S.rg.nch;  // (arr2: a second array of the source's type read)
        else st.bytes = (S.need - S.base) * nodes[S.node].nch * esz;
        steps.push_back(st);
    }
    stats.n_stages = (int)order.size() + 1;
    stats.algorithmic_bytes = algo_bytes + out.nframes * (int64_t)out.nch * (int64_t)dsize(out.dtype);
    int64_t h2d = 0;
    for (auto& h : host_leaves) h2d += (int64_t)h.bytes;
    stats.h2d_bytes = h2d;
    stats.d2h_bytes = out.is_device ? 0 : out.nframes * (int64_t)out.nch * (int64_t)dsize(out.dtype);
}

// Dependencies between steps from the plan buffers they read and write, then a lane (stream)
// per step: a step continues on the lane of its latest dependency, a step without
// dependencies opens the next lane (round robin over at most 8).
void Plan::plan_lanes() {
    const int n = (int)steps.size();
    step_deps.assign(n, {});
    step_lane.assign(n, 0);
    step_signals.assign(n, 0);
    nlanes = 1;
    if (n < 3 || std::getenv("SIGOPS_SINGLE_STREAM")) return;
    const int kFinal = -2;
    std::vector<std::set<int>> rd(n), wr(n);
    auto piece_reads = [&](const PwStep& w, std::set<int>& out) {
        for (int li : w.rtc_leaves)  // (a hipRTC step has no programs: its source names these leaves)
            if (li >= 0 && li < (int)leaves.size() && leaves[li].buf >= 0) out.insert(leaves[li].buf);
        for (int pi = w.piece0; pi < w.piece0 + w.npieces; ++pi) {
            const DPiece& P = pieces[pi];
            for (int k = 0; k < P.frame_len + P.samp_len; ++k) {
                const DOp& o = k < P.frame_len ? ops[P.frame_pc + k] : ops[P.samp_pc + (k - P.frame_len)];
                if ((o.code == OP_LOAD || o.code == OP_SCALAR) && o.arg >= 0 && o.arg < (int)leaves.size() &&
                    leaves[o.arg].buf >= 0)
                    out.insert(leaves[o.arg].buf);
            }
        }
    };
    auto carrier_reads = [&](const std::vector<DCarrier>& cs, std::set<int>& out) {
        for (auto& c : cs) {
            if (c.buf >= 0) out.insert(c.buf);
            for (int k = 0; k < c.frame_len; ++k) {
                const DOp& o = ops[c.frame_pc + k];
                if ((o.code == OP_LOAD || o.code == OP_SCALAR) && leaves[o.arg].buf >= 0) out.insert(leaves[o.arg].buf);
            }
            for (int k = 0; k < c.nslots; ++k)
                if (leaves[c.slot_leaf[k]].buf >= 0) out.insert(leaves[c.slot_leaf[k]].buf);
        }
    };
    for (int i = 0; i < n; ++i) {
        const Step& st = steps[i];
        if (st.kind == 0) {
            const PwStep& w = pw[st.idx];
            piece_reads(w, rd[i]);
            wr[i].insert(w.out_buf >= 0 ? w.out_buf : kFinal);
            // the root launch runs in place on the windows stages have written, and so does any
            // sub-expression of it that was materialised into a temporary first
            if (w.out_buf < 0 || (out_alias_buf >= 0 && rd[i].count(out_alias_buf)))
                for (size_t k = 0; k < stages.size(); ++k)
                    if (stages[k].win_off >= 0) rd[i].insert(-100 - (int)k);
        } else if (st.kind == 2) {  // members read device arrays only; each writes its window or buffer
            for (int m : batches[st.idx].members) {
                const Stage& S = stages[m];
                if (S.win_off >= 0) wr[i].insert(-100 - m);
                else wr[i].insert(m == alias_stage ? kFinal : S.out_buf);
            }
        } else if (st.kind == 3) {  // every member reads through its carriers and writes its window or buffer
            for (int m : rsbatches[st.idx].members) {
                const Stage& S = stages[m];
                carrier_reads(S.carriers, rd[i]);
                if (S.win_off >= 0) wr[i].insert(-100 - m);
                else wr[i].insert(m == alias_stage ? kFinal : S.out_buf);
            }
        } else {
            const Stage& S = stages[st.idx];
            if (S.in_buf >= 0 && S.rsos_src < 0) rd[i].insert(S.in_buf);
            carrier_reads(S.rsos_src >= 0 ? stages[S.rsos_src].carriers : S.carriers, rd[i]);
            if (S.win_off >= 0) wr[i].insert(-100 - st.idx);  // its own window of the result
            else wr[i].insert(st.idx == alias_stage ? kFinal : S.out_buf);
            if (S.kind == ST_NORM) {  // reads its own output buffer, writes the rms scalar
                rd[i].insert(S.out_buf);
                if (S.rms_buf >= 0) wr[i].insert(S.rms_buf);
            }
        }
    }
    auto meets = [](const std::set<int>& a, const std::set<int>& b) {
        for (int x : a)
            if (b.count(x)) return true;
        return false;
    };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (meets(rd[i], wr[j]) || meets(wr[i], wr[j]) || meets(wr[i], rd[j])) step_deps[i].push_back(j);
    int next = 0;
    const int kMaxLanes = 8;
    for (int i = 0; i < n; ++i) {
        if (step_deps[i].empty()) {
            step_lane[i] = next % kMaxLanes;
            ++next;
        } else step_lane[i] = step_lane[step_deps[i].back()];
    }
    step_lane[n - 1] = 0;  // the last step (it produces the result) runs on the caller's stream
    for (int i = 0; i < n; ++i) nlanes = std::max(nlanes, step_lane[i] + 1);
    for (int i = 0; i < n; ++i)
        for (int j : step_deps[i])
            if (step_lane[j] != step_lane[i]) step_signals[j] = 1;
    if (nlanes == 1) return;
    lane_streams.assign(nlanes, nullptr);
    for (int l = 1; l < nlanes; ++l) HIPCHECK(hipStreamCreateWithFlags(&lane_streams[l], hipStreamNonBlocking));
    step_done.assign(n, nullptr);
    for (int i = 0; i < n; ++i) HIPCHECK(hipEventCreateWithFlags(&step_done[i], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
}

void Plan::release() {
    for (auto st_ : lane_streams)
        if (st_) (void)hipStreamDestroy(st_);
    lane_streams.clear();
    for (auto e : step_done)
        if (e) (void)hipEventDestroy(e);
    step_done.clear();
    if (ev_start) (void)hipEventDestroy(ev_start);
    ev_start = nullptr;
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    graph_exec = nullptr;
    if (capture_stream) (void)hipStreamDestroy(capture_stream);
    capture_stream = nullptr;
    for (auto& b : bufs)
        if (b.d && !b.external) (void)hipFree(b.d);
    bufs.clear();
    if (d_pieces) (void)hipFree(d_pieces);
    if (d_ops) (void)hipFree(d_ops);
    if (d_leaves) (void)hipFree(d_leaves);
    if (kerr) {
        // (the hipFree calls above synchronised the device: whatever the plan's last launch had to say is in the word now.  An
        //  execute into a device result returns before its kernels have run, and the caller may never make another call on
        //  this plan -- a one-shot sink: so_plan_check is the call that reports it; where the host destroys the plan without
        //  it, the result it holds is invalid and nobody has been told: say so as loudly as a library can)
        (void)hipDeviceSynchronize();
        const uint32_t who = *(volatile uint32_t*)kerr;
        if (who)
            std::fprintf(stderr, "libsigops: %s: a wait between its waves did not end -- THE LAST RESULT OF THIS PLAN IS INVALID (so_plan_destroy "
                                 "without so_plan_check)\n", who == 2 ? "k_resample_arb" : "k_rsos");
        (void)hipHostFree(kerr);
    }
    kerr = nullptr;
    for (auto e : events) (void)hipEventDestroy(e);
    events.clear();
}


// executes a deferred-profiling plan keeps events for (plans of hundreds of steps keep fewer)
static size_t prof_cap(const Plan* P) {
    const size_t evset = 2 * std::max<size_t>(1, P->steps.size());
    return std::max<size_t>(1, std::min<size_t>(kProfExecs, 8192 / evset));
}

// the plan's host-mapped word a kernel reports a wait between its waves in that did not end (device address, or null: the
// kernel then traps -- also the choice of SIGOPS_RSOS_TRAP=1)
static uint32_t* kernel_error_word(Plan* P) {
    if (std::getenv("SIGOPS_RSOS_TRAP")) return nullptr;
    if (!P->kerr) {
        if (hipHostMalloc((void**)&P->kerr, 64, hipHostMallocMapped) == hipSuccess) *P->kerr = 0;
        else {
            P->kerr = nullptr;
            (void)hipGetLastError();
            return nullptr;
        }
    }
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, P->kerr, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return (uint32_t*)dp;
}

// a kernel gave up on a wait between its waves (k_rsos: a protocol that did not hold; its result and
// everything computed from it is wrong): reported once: by the execute itself where it synchronises (a host result), else by the next call on the plan
static bool kernel_gave_up(Plan* P, std::string& err) {
    if (!P->kerr || !*(volatile uint32_t*)P->kerr) return false;
    const uint32_t who = *(volatile uint32_t*)P->kerr;  // 1: k_rsos, 2: k_resample_arb
    *(volatile uint32_t*)P->kerr = 0;
    err = std::string(who == 2 ? "k_resample_arb" : "k_rsos") + ": a wait between its waves did not end (this plan's last result is invalid)";
    return true;
}

static int plan_execute_direct(Plan* P, void* outp, void* stream, std::string& err) {
    try {
        HIPCHECK(hipSetDevice(P->device));
        hipStream_t st = (hipStream_t)stream;
        if (P->out.nframes > 0 && !outp) fail(SO_ERR_INVALID, "so_plan_execute: null output");
        for (auto& h : P->host_leaves)
            if (h.bytes) HIPCHECK(hipMemcpyAsync(P->bufs[h.buf].d, P->array_ptr[h.node], h.bytes, hipMemcpyHostToDevice, st));
        // profiling 1: one set of events, read (after a synchronize) at the end of this execute;
        // profiling 2 (deferred): a set per execute, up to kProfExecs of them, nothing synchronised here --
        // so_plan_step_info averages them once the caller has synchronised (bench.py's timed region)
        const size_t evset = 2 * P->steps.size();
        const size_t pcap = prof_cap(P);
        const size_t evneed = P->profiling == 2 ? evset * pcap : evset;
        if (P->profiling && P->events.size() < evneed) {
            while (P->events.size() < evneed) {
                hipEvent_t e;
                HIPCHECK(hipEventCreate(&e));
                P->events.push_back(e);
            }
        }
        const size_t ev0 = P->profiling == 2 ? evset * ((size_t)P->prof_execs % pcap) : 0;
        if (P->out_alias_buf >= 0 && P->out.is_device && P->bufs[P->out_alias_buf].d != outp) {
            // in-place root pieces read the result: point their leaves at this execute's buffer
            P->bufs[P->out_alias_buf].d = outp;
            for (auto& L : P->leaves)
                if (L.buf == P->out_alias_buf) L.base = outp;
            if (P->d_leaves)
                HIPCHECK(hipMemcpyAsync(P->d_leaves, P->leaves.data(), P->leaves.size() * sizeof(DLeaf), hipMemcpyHostToDevice, st));
        }
        int launches = 0;
        // (profiling times the steps one after the other on the caller's stream)
        const bool lanes = P->nlanes > 1 && !P->profiling && !std::getenv("SIGOPS_RS_TRACE");
        hipStream_t const main_st = st;
        std::vector<char> lane_started(P->nlanes, 0);
        if (lanes) HIPCHECK(hipEventRecord(P->ev_start, main_st));  // after the H2D copies / earlier work
        for (size_t si = 0; si < P->steps.size(); ++si) {
            Step& s = P->steps[si];
            const int ln = lanes ? P->step_lane[si] : 0;
            hipStream_t st = ln == 0 ? main_st : P->lane_streams[ln];  // shadows the caller's stream
            if (lanes) {
                if (ln != 0 && !lane_started[ln]) {
                    HIPCHECK(hipStreamWaitEvent(st, P->ev_start, 0));
                    lane_started[ln] = 1;
                }
                for (int d : P->step_deps[si])
                    if (P->step_lane[d] != ln) HIPCHECK(hipStreamWaitEvent(st, P->step_done[d], 0));
            }
            if (P->profiling) HIPCHECK(hipEventRecord(P->events[ev0 + 2 * si], st));
            if (s.kind == 0) {
                PwStep& w = P->pw[s.idx];
                OutView ov{};
                if (w.nblocks <= 0) {
                    // empty rectangle (zero-frame sink): nothing to launch
                } else if (w.out_buf >= 0) {
                    Buf& b = P->bufs[w.out_buf];
                    ov.base = b.d;
                    ov.fstride = 1;
                    ov.cstride = b.pitch;
                    ov.dtype = b.dtype;
                } else if (P->out.is_device) {
                    ov.base = outp;
                    ov.fstride = P->out.frame_stride;
                    ov.cstride = P->out.chan_stride;
                    ov.dtype = P->out.dtype;
                } else {
                    Buf& b = P->bufs[P->out_stage_buf];
                    ov.base = b.d;
                    ov.fstride = P->interleaved_host ? P->out.nch : 1;
                    ov.cstride = P->interleaved_host ? 1 : b.pitch;
                    ov.dtype = b.dtype;
                }
                if (w.nblocks > 0) {
                    static const int il_scalar = std::getenv("SIGOPS_K1_ILSCALAR") ? 1 : 0;  // ablation knob
                    ov.pad = il_scalar;
                    if (w.rtc) {
                        if (rtc_launch(w.rtc, w.nblocks, P->d_pieces + w.piece0, w.npieces, P->d_leaves, ov, st) != 0)
                            fail(SO_ERR_RUNTIME, "hipRTC kernel launch failed");
                    } else
                        launch_pointwise(P->d_pieces + w.piece0, w.npieces, w.nblocks, P->d_ops, P->d_leaves, ov, w.deep, st, w.chain,
                                         w.il || (ov.fstride > 1 && ov.cstride == 1));
                    s.launches = 1;
                    launches++;
                }
            } else if (s.kind == 2) {
                SosBatch& B = P->batches[s.idx];
                const size_t nm = B.members.size();
                std::vector<SosDesc> d(nm + 1);
                int64_t first[3] = {0, 0, 0};
                for (size_t m = 0; m <= nm; ++m) {
                    SosDesc& D = d[m];
                    std::memset(&D, 0, sizeof D);
                    for (int q = 0; q < 3; ++q) D.first[q] = first[q];
                    if (m == nm) break;
                    const int sid = B.members[m];
                    const Stage& S = P->stages[sid];
                    const Node& N = P->nodes[S.node];
                    const size_t esz = dsize(N.dtype);
                    const so_node_t& nd = P->nodes[S.in_array_node].nd;
                    const char* base = nd.i0 ? (const char*)P->array_ptr[S.in_array_node] : (const char*)P->bufs[P->array_buf[S.in_array_node]].d;
                    Buf ob = P->bufs[S.out_buf];
                    if (sid == P->alias_stage) {
                        if (P->out.is_device) {
                            ob.d = outp;
                            ob.pitch = N.nch == 1 ? std::max<int64_t>(P->out.chan_stride, S.need) : P->out.chan_stride;
                        } else {
                            ob.d = P->bufs[P->out_stage_buf].d;
                            ob.pitch = P->bufs[P->out_stage_buf].pitch;
                        }
                        ob.d = (char*)ob.d - (size_t)P->alias_skip * (P->alias_narrow ? dsize(P->out.dtype) : esz);
                    } else if (S.win_off >= 0) {
                        const Buf& ab = P->bufs[P->out_alias_buf];
                        ob.d = (char*)(P->out.is_device ? outp : ab.d) + (size_t)S.win_off * esz;
                        ob.pitch = ab.pitch;
                    }
                    D.x = base + (size_t)S.in_offset * esz;
                    D.y = ob.d;
                    D.v = S.v_buf >= 0 ? (double*)P->bufs[S.v_buf].d : nullptr;
                    D.s0 = S.s0_buf >= 0 ? (double*)P->bufs[S.s0_buf].d : nullptr;
                    D.mpow = S.mpow_buf >= 0 ? (const double*)P->bufs[S.mpow_buf].d : nullptr;
                    D.g = S.sg;
                    D.g.in_pitch = N.nch == 1 ? 0 : S.in_pitch;
                    D.g.out_pitch = ob.pitch;
                    D.g.store_lo = sid == P->alias_stage ? P->alias_skip : 0;
                    if (sid == P->alias_stage && P->alias_narrow) D.g.out_dtype = SO_F32;
                    D.g.align_rows = !std::getenv("SIGOPS_SOS_NOALIGN");
                    D.g.bad = B.bad_buf >= 0 && S.sg.nchunks > 1 && !std::getenv("SIGOPS_SOS_NOPOISON")
                                  ? (int32_t*)P->bufs[B.bad_buf].d + B.bad_off[m] : nullptr;
                    if (S.src_op) {  // fused sine source of the cascade's input (as below)
                        const DLeaf& F = S.src_fn;
                        D.g.src_op = S.src_op;
                        D.g.src_has_omega = F.flag;
                        D.g.src_df = F.df;
                        D.g.src_omega = F.v0;
                        D.g.src_phi = F.v1;
                        D.g.src_fs = F.v2;
                        const double step = 6.283185307179586476925 * (F.flag ? F.v0 / F.v2 : 1.0 / F.v2);
                        D.g.src_cd = std::cos(step);
                        D.g.src_sd = std::sin(step);
                    }
                    D.cf = S.groups[0];
                    const int64_t nseq = (int64_t)D.g.nchunks * D.g.nch;
                    if (D.g.nchunks > 1) {
                        first[0] += ((int64_t)(D.g.nchunks - 1) * D.g.nch + kBlock - 1) / kBlock;
                        first[1] += (nseq + kBlock - 1) / kBlock;
                    } else
                        D.s0 = nullptr;  // one chunk: starts from rest
                    first[2] += (nseq + kBlock - 1) / kBlock;
                }
                if (first[2] >= ((int64_t)1 << 31)) fail(SO_ERR_RUNTIME, "internal: batched IIR grid too large");
                if (B.host.size() != d.size() || std::memcmp(B.host.data(), d.data(), d.size() * sizeof(SosDesc)) != 0) {
                    // the descriptors change only when the result or an array moves -- never between the direct
                    // execute for a result pointer and the capture that follows it (plan_execute), so a captured
                    // graph holds the three launches only and reads the table this copy left on the device
                    B.host = d;
                    HIPCHECK(hipMemcpyAsync(P->bufs[B.desc_buf].d, B.host.data(), B.host.size() * sizeof(SosDesc), hipMemcpyHostToDevice, st));
                }
                for (int q = 0; q < 3; ++q) B.total[q] = first[q];
                const bool poison = B.bad_buf >= 0 && !std::getenv("SIGOPS_SOS_NOPOISON");
                if (poison) HIPCHECK((hipError_t)launch_fill_u32(P->bufs[B.bad_buf].d, P->bufs[B.bad_buf].bytes / 4, 0x7f7f7f7fu, st));  // "no non-finite chunk yet"
                int nl = launch_sos_batch((const SosDesc*)P->bufs[B.desc_buf].d, (int)nm, B.nsec, B.dtype, B.total, st);
                if (poison) nl += launch_sos_poison_batch((const SosDesc*)P->bufs[B.desc_buf].d, (int)nm, B.dtype, st);
                s.launches = nl;
                launches += nl;
            } else if (s.kind == 3) {
                // plain filters, one pass each, ONE launch (k_rsos_batch): per member what the single launch below sets up
                RsBatch& B = P->rsbatches[s.idx];
                const size_t nm = B.members.size();
                std::vector<RsosItem> items(nm);
                std::vector<RsFixup> fix(nm);
                const bool poison = B.bad_buf >= 0 && !std::getenv("SIGOPS_SOS_NOPOISON");
                int maxch = 1, out_f32 = 0;
                for (size_t m = 0; m < nm; ++m) {
                    const int sid = B.members[m];
                    const Stage& S = P->stages[sid];
                    const Node& N = P->nodes[S.node];
                    const size_t esz = dsize(N.dtype);
                    Buf ob = P->bufs[S.out_buf];
                    if (S.win_off >= 0) {  // its window of the result
                        const Buf& ab = P->bufs[P->out_alias_buf];
                        ob.d = (char*)(P->out.is_device ? outp : ab.d) + (size_t)S.win_off * esz;
                        ob.pitch = ab.pitch;
                    }
                    RsosItem& I = items[m];
                    std::memset(&I, 0, sizeof I);
                    RsSos rs = S.rs;
                    rs.out_pitch = ob.pitch;
                    rs.out_f32 = N.dtype == SO_F32;
                    rs.f32m = 0;
                    rs.sring = 0;
                    rs.mats = (const double*)P->bufs[S.rsos_mats_buf].d;
                    rs.bad = poison ? (int32_t*)P->bufs[B.bad_buf].d + B.bad_off[m] : nullptr;
                    rs.err = kernel_error_word(P);
                    I.g = rs;
                    I.tab = (const double*)P->bufs[S.rsos_tab_buf].d;
                    I.jend = (const int*)P->bufs[S.rsos_jend_buf].d;
                    I.y = (char*)ob.d - (size_t)S.rs.store_lo * (rs.out_f32 ? 4 : 8);
                    I.gsrc = RsGlobalTables{(const RsCtl*)P->bufs[S.ctl_buf].d, (const DCarrier*)P->bufs[S.car_buf].d, P->d_ops, P->d_leaves};
                    RsFixup& F = fix[m];
                    std::memset(&F, 0, sizeof F);
                    F.g = rs;
                    F.tab = I.tab;
                    F.jend = I.jend;
                    F.jrel = (const int*)P->bufs[S.rsos_jrel_buf].d;
                    F.taps = S.rsos_taps;
                    F.cf = S.groups[0];
                    F.gsrc = I.gsrc;
                    F.y = I.y;
                    maxch = std::max(maxch, (int)rs.nch);
                    out_f32 = rs.out_f32;
                }
                if (B.host.size() != nm || std::memcmp(B.host.data(), items.data(), nm * sizeof(RsosItem)) != 0 ||
                    std::memcmp(B.fhost.data(), fix.data(), nm * sizeof(RsFixup)) != 0) {
                    // (as the three-pass batch's descriptors: the tables change only when the result or an array moves -- never
                    //  between the direct execute for a result pointer and the capture that follows it)
                    B.host = items;
                    B.fhost = fix;
                    HIPCHECK(hipMemcpyAsync(P->bufs[B.items_buf].d, B.host.data(), nm * sizeof(RsosItem), hipMemcpyHostToDevice, st));
                    HIPCHECK(hipMemcpyAsync(P->bufs[B.fix_buf].d, B.fhost.data(), nm * sizeof(RsFixup), hipMemcpyHostToDevice, st));
                }
                if (poison) HIPCHECK((hipError_t)launch_fill_u32(P->bufs[B.bad_buf].d, P->bufs[B.bad_buf].bytes / 4, 0x7f7f7f7fu, st));  // "no non-finite range yet"
                if (launch_rsos_batch((const RsosItem*)P->bufs[B.items_buf].d, (int)nm, B.gpm, items[0].g, st) != 0)
                    fail(SO_ERR_RUNTIME, "internal: no batched one-pass IIR instantiation for this geometry");
                int nl = 1;
                if (poison) nl += launch_rsos_fixup_batch((const RsFixup*)P->bufs[B.fix_buf].d, (int)nm, maxch, out_f32, st);
                s.launches = nl;
                launches += nl;
            } else {
                Stage& S = P->stages[s.idx];
                Node& N = P->nodes[S.node];
                size_t esz = dsize(N.dtype);
                const char* inp;
                int64_t in_pitch;
                if (S.kind == ST_RESAMPLE && S.periodic) {
                    inp = nullptr;  // the periodic kernel reads through its carriers
                    in_pitch = 0;
                } else if (S.in_array_node >= 0) {
                    const so_node_t& nd = P->nodes[S.in_array_node].nd;
                    const char* base = nd.i0 ? (const char*)P->array_ptr[S.in_array_node]
                                             : (const char*)P->bufs[P->array_buf[S.in_array_node]].d;
                    inp = base + (size_t)S.in_offset * esz;
                    in_pitch = N.nch == 1 ? 0 : S.in_pitch;
                } else {
                    Buf& b = P->bufs[S.in_buf];
                    inp = (const char*)b.d + (size_t)(S.in_offset - b.frame0) * esz;
                    in_pitch = b.pitch;
                }
                Buf ob = P->bufs[S.out_buf];
                if (s.idx == P->alias_stage) {  // write the sink buffer directly
                    if (P->out.is_device) {
                        ob.d = outp;
                        ob.pitch = N.nch == 1 ? std::max<int64_t>(P->out.chan_stride, S.need) : P->out.chan_stride;
                    } else {
                        ob.d = P->bufs[P->out_stage_buf].d;
                        ob.pitch = P->bufs[P->out_stage_buf].pitch;
                    }
                    // (local frame alias_skip is the result's frame 0; earlier frames are not stored)
                    ob.d = (char*)ob.d - (size_t)P->alias_skip * (P->alias_narrow ? dsize(P->out.dtype) : esz);
                } else if (S.win_off >= 0) {  // ... or its window of it
                    const Buf& ab = P->bufs[P->out_alias_buf];
                    ob.d = (char*)(P->out.is_device ? outp : ab.d) + (size_t)S.win_off * esz;
                    ob.pitch = ab.pitch;
                }
                if (S.kind == ST_SOS) {
                    SosGeom g = S.sg;
                    g.in_pitch = in_pitch;
                    g.out_pitch = ob.pitch;
                    g.store_lo = s.idx == P->alias_stage ? P->alias_skip : 0;
                    if (s.idx == P->alias_stage && P->alias_narrow) g.out_dtype = SO_F32;
                    g.align_rows = S.pre_stage < 0 && !std::getenv("SIGOPS_SOS_NOALIGN");
                    g.bad = nullptr;
                    if (S.bad_buf >= 0 && S.pre_stage < 0 && S.rsos_src < 0 && !S.onepass && !S.xscan && !g.exact &&
                        !std::getenv("SIGOPS_SOS_NOPOISON")) {
                        g.bad = (int32_t*)P->bufs[S.bad_buf].d;
                        HIPCHECK((hipError_t)launch_fill_u32(g.bad, (size_t)N.nch, 0x7f7f7f7fu, st));  // "no non-finite chunk yet"
                    }
                    size_t msz = 0;
                    for (auto& v : S.mpow_host) msz = std::max(msz, v.size());
                    int nl = 0;
                    if (S.rsos_src >= 0) {  // the resampler in front and this cascade in one launch (k_rsos)
                        const Stage& S3 = P->stages[S.rsos_src];
                        RsSos rs = S.rs;
                        rs.out_pitch = ob.pitch;
                        rs.out_f32 = g.out_dtype == SO_F32 || N.dtype == SO_F32;
                        if (rs.f32m && rs.out_f32) rs.ring32 = 1;  // (a Float32 result: Float32 samples in the ring, the Float32 MFMA)
                        else rs.f32m = 0;
                        rs.sring = rs.f32m && (rs.fuse == 1 || rs.fuse == 2) && !std::getenv("SIGOPS_RSOS_NO_SRING") ? 1 : 0;
                        // the kernel's output m is frame m - store_lo of this stage's buffer (a window: the resampler's warm
                        // start lies store_lo frames before the cascade's); the sink's own skipped frames come on top
                        char* const yk = (char*)ob.d - (size_t)S.rs.store_lo * (rs.out_f32 ? 4 : 8);
                        rs.store_lo = S.rs.store_lo + g.store_lo;
                        rs.mats = (const double*)P->bufs[S.rsos_mats_buf].d;
                        rs.bad = nullptr;
                        if (S.bad_buf >= 0 && rs.nranges >= 1 && !std::getenv("SIGOPS_SOS_NOPOISON")) {
                            rs.bad = (int32_t*)P->bufs[S.bad_buf].d;
                            HIPCHECK((hipError_t)launch_fill_u32(rs.bad, (size_t)N.nch, 0x7f7f7f7fu, st));  // "no non-finite range yet"
                        }
                        rs.err = kernel_error_word(P);
                        static long long* d_rtrace = nullptr;  // SIGOPS_RSOS_TRACE tuning aid
                        // (SIGOPS_RSOS_TRACE_SKIP=n: not the first n launches of the process -- a traced launch synchronises, and a run
                        //  of traced launches never leaves the chip's power-management transient)
                        static int rtrace_seen = 0;
                        const char* const rtrace_skip = std::getenv("SIGOPS_RSOS_TRACE_SKIP");
                        const bool rtracing = std::getenv("SIGOPS_RSOS_TRACE") != nullptr && rtrace_seen++ >= (rtrace_skip ? std::atoi(rtrace_skip) : 0);
                        const size_t rtrace_n = (size_t)16 * kRsosTraceIters * 8;
                        if (rtracing) {
                            if (!d_rtrace) HIPCHECK(hipMalloc(&d_rtrace, rtrace_n * 8));
                            HIPCHECK(hipMemsetAsync(d_rtrace, 0, rtrace_n * 8, st));
                            rs.trace = d_rtrace;
                        }
                        const double* rtab = (const double*)P->bufs[S.rsos_tab_buf >= 0 ? S.rsos_tab_buf : S3.tab_buf].d;
                        const int* rjend = (const int*)P->bufs[S.rsos_jend_buf >= 0 ? S.rsos_jend_buf : S3.jend_buf].d;
                        if (launch_rsos(rtab, rjend, rs, yk,
                                        RsGlobalTables{(const RsCtl*)P->bufs[S3.ctl_buf].d, (const DCarrier*)P->bufs[S3.car_buf].d, P->d_ops, P->d_leaves},
                                        S.rsos_grid, st) != 0)
                            fail(SO_ERR_RUNTIME, "internal: no fused resampler + IIR instantiation for this geometry");
                        if (rtracing) {
                            std::vector<long long> tr(rtrace_n);
                            HIPCHECK(hipStreamSynchronize(st));
                            HIPCHECK(hipMemcpy(tr.data(), d_rtrace, rtrace_n * 8, hipMemcpyDeviceToHost));
                            long long t0 = 0;
                            for (size_t i = 0; i < rtrace_n; ++i)
                                if (tr[i] && (!t0 || tr[i] < t0)) t0 = tr[i];
                            if (std::atoi(std::getenv("SIGOPS_RSOS_TRACE")) == 2) {  // a -DSO_RSOS_COUNT=1 build: poll counters, raw
                                for (int w = 0; w < rs.nwaves; ++w)
                                    for (int it = 0; it < 5; ++it) {
                                        const long long* q = &tr[((size_t)w * kRsosTraceIters + it) * 8];
                                        bool any = false;
                                        for (int k = 0; k < 8; ++k) any = any || q[k];
                                        if (!any) continue;
                                        std::fprintf(stderr, "[rsos-count] w%02d row%d", w, it);
                                        for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %lld", q[k] - 1);
                                        std::fprintf(stderr, "\n");
                                    }
                            } else
                            for (int w = 0; w < rs.nwaves; ++w)
                                for (int it = 0; it < kRsosTraceIters; ++it) {
                                    const long long* q = &tr[((size_t)w * kRsosTraceIters + it) * 8];
                                    if (!q[0] && !q[2]) continue;
                                    std::fprintf(stderr, "[rsos-trace] %s w%02d it%02d", w == 0 ? "C" : (w & 3) == 0 ? "L" : "Y", w, it);
                                    for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %lld", q[k] ? q[k] - t0 : -1);
                                    std::fprintf(stderr, "\n");
                                }
                        }
                        nl = 1;
                        if (rs.bad && S.rsos_jrel_buf >= 0 && !std::getenv("SIGOPS_RSOS_NO_FIXUP")) {
                            // behind a non-finite sample the reference stays non-finite: NaN over everything after the first bad
                            // range -- and INSIDE that range the block the sample fell into (the block form makes all 16 outputs
                            // non-finite, and a group's window is wider than an output's) output by output the reference's way
                            RsFixup fx{};
                            fx.g = rs;
                            fx.tab = rtab;
                            fx.jend = rjend;
                            fx.jrel = (const int*)P->bufs[S.rsos_jrel_buf].d;
                            fx.taps = S.rsos_taps;
                            fx.cf = S.groups[0];
                            fx.gsrc = RsGlobalTables{(const RsCtl*)P->bufs[S3.ctl_buf].d, (const DCarrier*)P->bufs[S3.car_buf].d, P->d_ops, P->d_leaves};
                            fx.y = yk;
                            nl += launch_rsos_fixup(fx, st);
                        } else if (rs.bad) {
                            SosGeom pg = g;
                            pg.n = rs.n_out;
                            pg.chunk = rs.pr * rs.L;
                            pg.nchunks = rs.nranges;
                            pg.store_lo = rs.store_lo;
                            pg.bad = rs.bad;
                            nl += launch_sos_poison(yk, pg, st);
                        }
                    }
                    for (size_t gi = 0; gi < S.groups.size() && S.onepass; ++gi) {
                        const void* x = gi == 0 ? (const void*)inp : (const void*)ob.d;
                        SosOne o = S.so1;
                        o.in_pitch = gi == 0 ? in_pitch : ob.pitch;
                        o.out_pitch = ob.pitch;
                        const int64_t al = 16 / (int64_t)esz;
                        o.vec_in = ((uintptr_t)x % 16 == 0) && (o.in_pitch % al == 0);
                        o.vec_out = ((uintptr_t)ob.d % 16 == 0) && (o.out_pitch % al == 0);
                        Buf& sb = P->bufs[S.one_sync_buf];
                        Buf& vb = P->bufs[S.one_vpub_buf];
                        // (kernel nodes, not memset nodes, as for the poison words: this sequence is graph-capturable too, and a
                        //  stale ticket counter would hang the kernel -- k_small.hip)
                        if (sb.bytes % 4 == 0 && vb.bytes % 4 == 0) {
                            HIPCHECK((hipError_t)launch_fill_u32(sb.d, sb.bytes / 4, 0u, st));           // ticket counter
                            HIPCHECK((hipError_t)launch_fill_u32(vb.d, vb.bytes / 4, 0xffffffffu, st));  // "not published yet"
                        } else {
                            HIPCHECK(hipMemsetAsync(sb.d, 0, sb.bytes, st));
                            HIPCHECK(hipMemsetAsync(vb.d, 0xff, vb.bytes, st));
                        }
                        launch_sos_onepass(x, ob.d, o, S.groups[gi], (const double*)P->bufs[S.one_tabs_buf].d + S.one_tabs_off[gi],
                                           (int*)sb.d, (double*)P->bufs[S.one_vpub_buf].d, N.dtype, st);
                        nl += 1;
                    }
                    if (S.pre_stage >= 0) {
                        const Stage& S3 = P->stages[S.pre_stage];
                        nl += launch_sos_prestate(inp, ob.d, (const double*)P->bufs[S3.vper_buf].d, S3.rp.nperiods,
                                                  (const double*)P->bufs[S.qmat_buf].d, S3.rp.pt, (double*)P->bufs[S.v_buf].d,
                                                  (double*)P->bufs[S.s0_buf].d, (const double*)P->bufs[S.mpow_buf].d, g,
                                                  S.groups[0], st);
                    }
                    bool exact_done = false;
                    if (g.exact && !S.onepass && S.pre_stage < 0 && S.groups.size() <= 2) {
                        // ill-conditioned cascade: DSP.jl's order of operations, one sequence per channel
                        SosGeom gg = g;
                        exact_done = launch_sos_exact(inp, ob.d, gg, S.groups[0], S.groups.size() > 1 ? S.groups[1] : SosCoefs{}, st) == 0;
                        if (exact_done) nl += 1;
                    }
                    for (size_t gi = 0; gi < S.groups.size() && !S.onepass && S.pre_stage < 0 && !exact_done && S.rsos_src < 0; ++gi) {
                        const void* x = gi == 0 ? (const void*)inp : (const void*)ob.d;
                        SosGeom gg = g;
                        if (gi > 0) gg.in_pitch = ob.pitch;
                        if (gi == 0 && S.src_op) {  // fused sine source of the cascade's input
                            const DLeaf& F = S.src_fn;
                            gg.src_op = S.src_op;
                            gg.src_has_omega = F.flag;
                            gg.src_df = F.df;
                            gg.src_omega = F.v0;
                            gg.src_phi = F.v1;
                            gg.src_fs = F.v2;
                            const double step = 6.283185307179586476925 * (F.flag ? F.v0 / F.v2 : 1.0 / F.v2);
                            gg.src_cd = std::cos(step);
                            gg.src_sd = std::sin(step);
                        }
                        // frames beyond the child's end are zero (Pad(x.signal,zero), reference
                        // src/filters.jl:240): the materialised input covers them; a direct
                        // source always has in_frames == need
                        if (S.xscan) {  // exact scan between the state pass and the output pass
                            double* vb = (double*)P->bufs[S.v_buf].d;
                            double* sb = (double*)P->bufs[S.s0_buf].d;
                            nl += launch_sos_phase(x, ob.d, vb, nullptr, gg, S.groups[gi], 1, st);
                            nl += launch_sos_xscan(vb, sb, (const double*)((char*)P->bufs[S.xs_mats_buf].d + gi * 2 * 16 * 16 * 8),
                                                   (double*)P->bufs[S.sblk_buf].d, gg, S.groups[gi].nsec, st);
                            nl += launch_sos_phase(x, ob.d, nullptr, sb, gg, S.groups[gi], 3, st);
                            continue;
                        }
                        nl += launch_sos(x, ob.d, S.v_buf >= 0 ? (double*)P->bufs[S.v_buf].d : nullptr,
                                         S.s0_buf >= 0 ? (double*)P->bufs[S.s0_buf].d : nullptr,
                                         S.mpow_buf >= 0 ? (const double*)((char*)P->bufs[S.mpow_buf].d + gi * msz * 8) : nullptr,
                                         gg, S.groups[gi], st);
                        nl += launch_sos_poison(ob.d, gg, st);  // (behind a NaN the reference stays NaN: SosGeom::bad)
                        if (gg.bad && gi + 1 < S.groups.size()) HIPCHECK((hipError_t)launch_fill_u32(gg.bad, (size_t)N.nch, 0x7f7f7f7fu, st));
                    }
                    s.launches = nl;
                    launches += nl;
                } else if (S.kind == ST_RESAMPLE) {
                    RsGeom g = S.rg;
                    g.in_pitch = in_pitch;
                    g.out_pitch = ob.pitch;
                    if (S.periodic) {
                        RsPeriodic rp = S.rp;
                        rp.in_pitch = in_pitch;
                        rp.out_pitch = ob.pitch;
                        rp.out_f32 = s.idx == P->alias_stage && P->alias_narrow;
                        if (rp.nstate > 0) {
                            rp.wtab = (const double*)P->bufs[S.wtab_buf].d;
                            rp.vper = (double*)P->bufs[S.vper_buf].d;
                        }
                        const int64_t al = 16 / (int64_t)esz;
                        rp.vec_ok = ((uintptr_t)ob.d % 16 == 0) && (ob.pitch % al == 0) && (rp.L % al == 0);
                        static long long* d_trace = nullptr;  // SIGOPS_RS_TRACE tuning aid
                        const bool tracing = std::getenv("SIGOPS_RS_TRACE") != nullptr;
                        const size_t trace_n = (size_t)16 * kRsTraceIters * kRsTraceStamps;
                        if (tracing) {
                            if (!d_trace) HIPCHECK(hipMalloc(&d_trace, trace_n * 8));
                            HIPCHECK(hipMemsetAsync(d_trace, 0, trace_n * 8, st));
                            rp.trace = d_trace;
                        }
                        // (the list of (tile, group)s a non-finite sample reached, and the launch that recomputes them the reference's
                        //  way: plain sources only -- a gain applied at the MFMA operand, a second array, a state pass keep the kernel's set)
                        const bool nf_fix = S.rs_nf_buf >= 0 && S.rs_jrel_buf >= 0 && rp.ga == 0 && rp.arr2 == 0 && rp.nstate == 0 && S.ctl_buf >= 0 &&
                                            S.car_buf >= 0 && !S.carriers.empty() && rp.rows <= 32 && rp.kw <= 160;
                        rp.nf = nf_fix ? (uint32_t*)P->bufs[S.rs_nf_buf].d : nullptr;
                        if (launch_resample_periodic(ob.d, (const double*)P->bufs[S.tab_buf].d,
                                                     (const int*)P->bufs[S.jend_buf].d, rp, N.dtype,
                                                     RsGlobalTables{(const RsCtl*)P->bufs[S.ctl_buf].d, (const DCarrier*)P->bufs[S.car_buf].d, P->d_ops, P->d_leaves},
                                                     st) != 0)
                            fail(SO_ERR_RUNTIME, "internal: no periodic resampler instantiation for this geometry");
                        if (nf_fix) {
                            RsPerFixup fx{};
                            fx.g = rp;
                            fx.tab = (const double*)P->bufs[S.tab_buf].d;
                            fx.jend = (const int*)P->bufs[S.jend_buf].d;
                            fx.jrel = (const int*)P->bufs[S.rs_jrel_buf].d;
                            fx.taps = S.rg.taps;
                            fx.out_f32 = (N.dtype == SO_F32 || rp.out_f32) ? 1 : 0;
                            fx.gsrc = RsGlobalTables{(const RsCtl*)P->bufs[S.ctl_buf].d, (const DCarrier*)P->bufs[S.car_buf].d, P->d_ops, P->d_leaves};
                            fx.y = ob.d;
                            s.launches += launch_rs_fixup(fx, st);
                        }
                        if (tracing) {
                            std::vector<long long> tr(trace_n);
                            HIPCHECK(hipStreamSynchronize(st));
                            HIPCHECK(hipMemcpy(tr.data(), d_trace, trace_n * 8, hipMemcpyDeviceToHost));
                            long long t0 = 0;
                            for (size_t i = 0; i < trace_n; ++i)
                                if (tr[i] && (!t0 || tr[i] < t0)) t0 = tr[i];
                            for (int w = 0; w < rp.nwaves; ++w)
                                for (int it = 0; it < kRsTraceIters; ++it) {
                                    const long long* q = &tr[((size_t)w * kRsTraceIters + it) * kRsTraceStamps];
                                    if (!q[0]) continue;
                                    std::fprintf(stderr, "[rs-trace] %s w%02d it%02d", w < rp.ncompute ? "C" : "L", w, it);
                                    for (int k = 0; k < kRsTraceStamps; ++k)
                                        std::fprintf(stderr, " %lld", q[k] ? q[k] - t0 : -1);
                                    std::fprintf(stderr, "\n");
                                }
                        }
                    } else if (S.rows) {
                        RsRows rr = S.rr;
                        rr.in_pitch = in_pitch;
                        rr.out_pitch = ob.pitch;
                        if (std::getenv("SIGOPS_DEBUG_PLAN"))
                            std::fprintf(stderr, "[sigops] k_resample_rows: ct=%d pb=%d tile_len=%d pitch=%d kw=%d ngroups=%d L=%lld M=%lld taps=%d\n", rr.ct, rr.pb,
                                         rr.tile_len, rr.pitch, rr.kw, rr.ngroups, (long long)rr.L, (long long)rr.M, rr.taps);
                        launch_resample_rows(inp, ob.d, (const double*)P->bufs[S.tab_buf].d,
                                             (const int*)P->bufs[S.jend_buf].d, (const double*)P->bufs[S.mtab_buf].d,
                                             (const int*)P->bufs[S.mjend_buf].d, rr, N.dtype, st);
                    } else if (S.tiled) {
                        RsTiled rt = S.rt;
                        rt.g.in_pitch = in_pitch;
                        rt.g.out_pitch = ob.pitch;
                        bool done = false;
                        if (S.arbk) {
                            RsArb ra = S.ra;
                            ra.g.in_pitch = in_pitch;
                            ra.g.out_pitch = ob.pitch;
                            ra.err = kernel_error_word(P);
                            done = launch_resample_arb(inp, ob.d, (const double*)P->bufs[S.pfbt_buf].d,
                                                       (const double*)P->bufs[S.dpfbt_buf].d, ra, st) == 0;
                        }
                        if (done) {
                        } else if (rt.pair)
                            launch_resample_tiled2(inp, ob.d, (const double*)P->bufs[S.pfbt_buf].d,
                                                  (const double*)P->bufs[S.dpfbt_buf].d, rt, st);
                        else
                            launch_resample_tiled(inp, ob.d, (const double*)P->bufs[S.pfbt_buf].d,
                                                  (const double*)P->bufs[S.dpfbt_buf].d, rt, st);
                    } else
                        launch_resample(inp, ob.d, (const double*)P->bufs[S.pfb_buf].d,
                                        (const double*)P->bufs[S.dpfb_buf].d, g, st);
                    s.launches = 1;
                    launches++;
                    if (S.fix_buf >= 0) {  // the outputs DSP.jl's phase accumulator places differently
                        RsFixArgs fa{};
                        fa.fix = (const RsFix*)P->bufs[S.fix_buf].d;
                        fa.nfix = (int64_t)S.fix_host.size();
                        fa.pfb = (const double*)P->bufs[S.pfb_buf].d;
                        fa.dpfb = (const double*)P->bufs[S.dpfb_buf].d;
                        fa.n_in = g.n_in;
                        fa.taps = g.taps;
                        fa.nch = N.nch;
                        fa.stage_dtype = N.dtype;
                        fa.out_dtype = (s.idx == P->alias_stage && P->alias_narrow) ? SO_F32 : N.dtype;
                        if (S.periodic) {
                            fa.car = (const DCarrier*)P->bufs[S.car_buf].d;
                            fa.ncar = (int)S.carriers.size();
                            fa.ops = P->d_ops;
                            fa.leaves = P->d_leaves;
                        } else {
                            fa.x = inp;
                            fa.in_pitch = in_pitch;
                            fa.in_dtype = N.dtype;
                        }
                        fa.y = ob.d;
                        fa.out_pitch = ob.pitch;
                        launch_resample_fix(fa, st);
                        s.launches++;
                        launches++;
                    }
                } else {
                    RmsPatch patch{};
                    for (int li : S.rms_leaves) patch.dst[patch.n++] = &P->d_leaves[li].v0;
                    launch_rms(S.norm_direct ? (const void*)inp : ob.d, N.dtype, S.need, N.nch, S.norm_direct ? in_pitch : ob.pitch,
                               (double*)P->bufs[S.partial_buf].d, S.nparts, (double*)P->bufs[S.rms_buf].d, st, patch);
                    s.launches = 2;
                    launches += 2;
                }
            }
            if (P->profiling) HIPCHECK(hipEventRecord(P->events[ev0 + 2 * si + 1], st));
            if (lanes && (P->step_signals[si] || (ln != 0 && si + 1 == P->steps.size())))
                HIPCHECK(hipEventRecord(P->step_done[si], st));
        }
        if (lanes) {
            // join: everything the side lanes did is ordered before what follows on the caller's
            // stream (the last step of every side lane signals; wait for the last step per lane)
            std::vector<int> last(P->nlanes, -1);
            for (size_t si = 0; si < P->steps.size(); ++si) last[P->step_lane[si]] = (int)si;
            for (int l = 1; l < P->nlanes; ++l)
                if (last[l] >= 0) {
                    if (!P->step_signals[last[l]]) HIPCHECK(hipEventRecord(P->step_done[last[l]], P->lane_streams[l]));
                    HIPCHECK(hipStreamWaitEvent(main_st, P->step_done[last[l]], 0));
                }
        }
        HIPCHECK(hipGetLastError());
        P->stats.n_launches = launches;
        if (!P->out.is_device && P->out.nframes > 0) {
            Buf& b = P->bufs[P->out_stage_buf];
            size_t esz = dsize(P->out.dtype);
            bool planar = P->out.frame_stride == 1 && (P->out.nch == 1 || P->out.chan_stride == P->out.nframes);
            if (planar || P->interleaved_host) {
                HIPCHECK(hipMemcpyAsync(outp, b.d, b.bytes, hipMemcpyDeviceToHost, st));
                HIPCHECK(hipStreamSynchronize(st));
            } else {
                P->host_tmp.resize(b.bytes);
                HIPCHECK(hipMemcpyAsync(P->host_tmp.data(), b.d, b.bytes, hipMemcpyDeviceToHost, st));
                HIPCHECK(hipStreamSynchronize(st));
                for (int c = 0; c < P->out.nch; ++c)
                    for (int64_t f = 0; f < P->out.nframes; ++f)
                        std::memcpy((char*)outp + (size_t)(f * P->out.frame_stride + c * P->out.chan_stride) * esz,
                                    P->host_tmp.data() + (size_t)(c * P->out.nframes + f) * esz, esz);
            }
        } else if (!P->host_leaves.empty() || P->profiling == 1) {
            HIPCHECK(hipStreamSynchronize(st));
        }
        if (!P->out.is_device && P->out.nframes > 0) {  // (synchronised above)
            std::string why;
            if (kernel_gave_up(P, why)) fail(SO_ERR_RUNTIME, why);
        }
        if (P->profiling == 2) P->prof_execs++;
        if (P->profiling == 1) {
            HIPCHECK(hipStreamSynchronize(st));
            double total = 0, best = -1;
            for (size_t si = 0; si < P->steps.size(); ++si) {
                float ms = 0;
                HIPCHECK(hipEventElapsedTime(&ms, P->events[2 * si], P->events[2 * si + 1]));
                P->steps[si].ms = ms;
                total += ms;
                if (ms > best) {
                    best = ms;
                    P->stats.dominant_kernel_ms = ms;
                    P->stats.dominant_kernel_bytes = P->steps[si].bytes;
                    std::snprintf(P->stats.dominant_kernel, sizeof P->stats.dominant_kernel, "%s", P->steps[si].name.c_str());
                }
            }
            P->stats.last_exec_ms = total;
        }
    } catch (const PlanError& e) {
        err = e.msg;
        return e.status;
    }
    return SO_OK;
}

// so_plan_execute.  Plans with many small launches (config 4: 33 launches and ~40 event
// operations per execute) are host-bound, so from the second execute with the same result
// pointer on, the whole multi-stream launch sequence is replayed from a captured HIP graph.
int plan_execute(Plan* P, void* outp, void* stream, std::string& err) {
    DeviceGuard guard(P->device);
    if (kernel_gave_up(P, err)) return SO_ERR_RUNTIME;
    const bool eligible = P->steps.size() >= 4 && P->out.is_device && P->host_leaves.empty() && !P->profiling &&
                          !std::getenv("SIGOPS_NO_GRAPH") && !std::getenv("SIGOPS_RS_TRACE");
    if (!eligible) {
        P->n_direct++;
        return plan_execute_direct(P, outp, stream, err);
    }
    hipStream_t st = (hipStream_t)stream;
    // The captured launches read the leaf table on the device as it stood at capture time.  In-place root
    // pieces read the RESULT through leaves that plan_execute_direct re-points at every new result
    // pointer: after a direct execute into another buffer the table no longer matches the graph (A, A,
    // B, A would apply the in-place gains to B's values a second time), so the graph is only replayed
    // while those leaves still point at its buffer; otherwise this execute goes the direct way, which
    // points them back.
    const bool leaves_match = P->out_alias_buf < 0 || P->bufs[P->out_alias_buf].d == outp;
    if (P->graph_exec && P->graph_out == outp && P->graph_epoch == P->array_epoch && leaves_match) {
        if (hipSetDevice(P->device) == hipSuccess && hipGraphLaunch(P->graph_exec, st) == hipSuccess) {
            P->n_replays++;
            return SO_OK;
        }
        (void)hipGetLastError();
        (void)hipGraphExecDestroy(P->graph_exec);  // fall back to direct launches for good
        P->graph_exec = nullptr;
        P->graph_failed = true;
    }
    if (P->graph_failed || P->last_out != outp || P->last_epoch != P->array_epoch) {
        // first execute for this result / these arrays: plain launches (also performs the
        // one-time function attribute calls, which must not happen inside a capture)
        P->last_out = outp;
        P->last_epoch = P->array_epoch;
        P->n_direct++;
        return plan_execute_direct(P, outp, stream, err);
    }
    if (P->graph_exec && P->graph_out == outp && P->graph_epoch == P->array_epoch) {
        // (the graph is still the right one; only the leaf table had moved)
        P->n_direct++;
        return plan_execute_direct(P, outp, stream, err);
    }
    if (P->graph_exec) {
        (void)hipGraphExecDestroy(P->graph_exec);
        P->graph_exec = nullptr;
    }
    hipGraph_t graph = nullptr;
    // capture on a stream of our own (the caller's may be the legacy default stream, which cannot
    // be captured); the graph is then launched on the caller's stream
    if (!P->capture_stream && hipStreamCreateWithFlags(&P->capture_stream, hipStreamNonBlocking) != hipSuccess)
        P->capture_stream = nullptr;
    if (!P->capture_stream || hipSetDevice(P->device) != hipSuccess ||
        hipStreamBeginCapture(P->capture_stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        P->graph_failed = true;
        return plan_execute_direct(P, outp, stream, err);
    }
    const int rc = plan_execute_direct(P, outp, (void*)P->capture_stream, err);
    const hipError_t ec = hipStreamEndCapture(P->capture_stream, &graph);
    if (std::getenv("SIGOPS_DEBUG_PLAN"))
        std::fprintf(stderr, "[sigops] graph capture: rc=%d end=%d (%s) graph=%p err=%s\n", rc, (int)ec, hipGetErrorString(ec), (void*)graph, err.c_str());
    if (rc != SO_OK || ec != hipSuccess || !graph ||
        hipGraphInstantiate(&P->graph_exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        P->graph_exec = nullptr;
        P->graph_failed = true;
        return rc != SO_OK ? rc : plan_execute_direct(P, outp, stream, err);
    }
    (void)hipGraphDestroy(graph);
    P->graph_out = outp;
    P->graph_epoch = P->array_epoch;
    P->n_captures++;
    if (hipGraphLaunch(P->graph_exec, st) != hipSuccess) {
        (void)hipGetLastError();
        P->graph_failed = true;
        return plan_execute_direct(P, outp, stream, err);
    }
    return SO_OK;
}

int plan_set_array(Plan* P, int32_t node_index, const void* data, std::string& err) {
    if (node_index < 0 || node_index >= (int)P->nodes.size() || P->nodes[node_index].nd.kind != SO_NODE_ARRAY) {
        err = "so_plan_set_array: not an ARRAY node";
        return SO_ERR_INVALID;
    }
    DeviceGuard guard(P->device);
    P->array_ptr[node_index] = data;
    P->array_epoch++;  // invalidates a captured launch graph
    if (P->nodes[node_index].nd.i0) {  // device leaf: patch the leaf table
        bool changed = false;
        for (size_t i = 0; i < P->leaves.size(); ++i)
            if (P->leaf_array_node[i] == node_index) {
                P->leaves[i].base = data;
                changed = true;
            }
        if (changed && P->d_leaves)
            if (hipMemcpy(P->d_leaves, P->leaves.data(), P->leaves.size() * sizeof(DLeaf), hipMemcpyHostToDevice) != hipSuccess) {
                err = "so_plan_set_array: leaf upload failed";
                return SO_ERR_RUNTIME;
            }
        // carriers of fused resampler stages (by value at launch + a device copy for the slow path)
        for (auto& S : P->stages) {
            bool touched = false;
            for (auto& c : S.carriers) {
                if (c.array_node == node_index) {
                    c.base = data;
                    const int64_t V = 16 / (int64_t)dsize(c.dtype);
                    c.vec_ok = carrier_vec_ok(c, V);
                    touched = true;
                }
                if (car_has_arr2(c) && c.array_node2 == node_index) {
                    c.base2 = data;
                    c.vec_ok2 = (uintptr_t)c.base2 % (c.dtype2 == SO_F64 ? 8 : 4) == 0;
                    touched = true;
                }
            }
            if (touched) {
                const RsCtl ctl = P->make_ctl(S);
                if (hipMemcpy(P->bufs[S.car_buf].d, S.carriers.data(), S.carriers.size() * sizeof(DCarrier), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(P->bufs[S.ctl_buf].d, &ctl, sizeof(RsCtl), hipMemcpyHostToDevice) != hipSuccess) {
                    err = "so_plan_set_array: carrier upload failed";
                    return SO_ERR_RUNTIME;
                }
            }
        }
    }
    return SO_OK;
}

int64_t plan_nframes(const Plan* P) {
    const Node& R = P->nodes[P->root];
    return isinf_(R.len) ? SO_LEN_INF : R.len.n;
}
void plan_stats(const Plan* P, so_stats_t* st) { *st = P->stats; }
void plan_set_profiling(Plan* P, int mode) {
    P->profiling = mode < 0 || mode > 2 ? 1 : mode;
    P->prof_execs = 0;
}
int64_t plan_counter(const Plan* P, int which) {
    switch (which) {
    case SO_COUNTER_GRAPH_REPLAYS: return P->n_replays;
    case SO_COUNTER_GRAPH_CAPTURES: return P->n_captures;
    case SO_COUNTER_DIRECT_EXECUTES: return P->n_direct;
    case SO_COUNTER_FUSED_MFMAS_PER_BLOCK: {  // k_rsos: window k-steps + 14 (D . X, X^T T^T, the chain's A^16 . S, S^T C^T)
        for (const Stage& S : P->stages)
            if (S.kind == ST_SOS && S.rsos_src >= 0 && S.need > 0) return S.rs.ks + 14;
        return 0;
    }
    default: return -1;
    }
}
int plan_step_info(const Plan* P, int index, so_step_info_t* info) {
    if (info && index >= 0 && index < (int)P->steps.size()) {
        const Step& s = P->steps[index];
        std::memset(info, 0, sizeof *info);
        std::snprintf(info->name, sizeof info->name, "%s", s.name.c_str());
        info->algorithmic_bytes = s.bytes;
        info->ms = s.ms;
        if (P->profiling == 2 && P->prof_execs > 0) {
            // deferred mode: mean over the (up to kProfExecs most recent) executes recorded since
            // so_plan_set_profiling(plan, 2); the caller has synchronised the stream
            const size_t evset = 2 * P->steps.size();
            const int64_t n = std::min<int64_t>(P->prof_execs, (int64_t)prof_cap(P));
            double sum = 0;
            int64_t got = 0;
            for (int64_t e = 0; e < n; ++e) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, P->events[evset * e + 2 * index], P->events[evset * e + 2 * index + 1]) == hipSuccess) {
                    sum += ms;
                    ++got;
                } else (void)hipGetLastError();
            }
            info->ms = got ? sum / (double)got : 0.0;
        }
        info->launches = s.launches;
    }
    return (int)P->steps.size();
}
// so_plan_check: waits for everything the plan has launched so far (the stream its last execute ran on -- or the whole
// device where the plan ran on lanes of its own) and reports what only shows once the kernels have run: a kernel that
// gave up on a wait between its waves.  The call a host makes behind an execute into a DEVICE result before it trusts it
// (a host result's execute synchronises and reports by itself).
int plan_check(Plan* P, void* stream, std::string& err) {
    DeviceGuard guard(P->device);
    try {
        if (P->nlanes > 1 || P->graph_exec) HIPCHECK(hipDeviceSynchronize());
        else HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
    } catch (const PlanError& e) {
        err = e.msg;
        return e.status;
    }
    if (kernel_gave_up(P, err)) return SO_ERR_RUNTIME;
    return SO_OK;
}

void plan_destroy(Plan* P) {
    if (!P) return;
    {
        DeviceGuard guard(P->device);
        P->release();
    }
    delete P;
}


}  // namespace so
