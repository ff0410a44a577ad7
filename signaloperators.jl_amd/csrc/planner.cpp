// Planner: turns the flattened operator tree (include/sigops.h) into a short list
// of device steps.  This replaces the per-node `nextblock` state machines of the
// reference (src/cutting.jl:154-210, src/padding.jl:200-235, src/appending.jl:82-110,
// src/ramps.jl:45-119, src/mapsignal.jl:194-244, src/filters.jl:169-262) by a single
// host-side pass using the closed-form semantics of every node:
//
//   lower(node, rectangle, index-map) -> pieces {rectangle, expression}
//
// Index-only nodes (Until/After/Pad/Extend/Append/Ramp regions, channel maps) never
// reach the device: they become piece boundaries and leaf index offsets.  Stateful
// nodes (Filt IIR, resampler, Normpower) become *stages* with device buffers.
#include "plan_impl.h"

namespace so {

// ===========================================================================
void Plan::build_nodes(const so_node_t* in, int n) {
    nodes.resize(n);
    for (int i = 0; i < n; ++i) {
        Node& N = nodes[i];
        N.nd = in[i];
        const so_node_t& nd = in[i];
        if (nd.n_children < 0 || (nd.n_children > 0 && !nd.children))
            fail(SO_ERR_INVALID, "node " + std::to_string(i) + ": bad children");
        for (int j = 0; j < nd.n_children; ++j) {
            int c = nd.children[j];
            if (c < 0 || c >= i) fail(SO_ERR_INVALID, "node table must be in post-order (children before parents)");
            N.kids.push_back(c);
        }
        N.dtype = nd.dtype;
        N.nch = nd.nch;
        N.fs = nd.fs;
        auto kid = [&](int j) -> Node& {
            if ((int)N.kids.size() <= j) fail(SO_ERR_INVALID, "node " + std::to_string(i) + ": missing child");
            return nodes[N.kids[j]];
        };
        switch (nd.kind) {
        case SO_NODE_ARRAY:
            if (nd.dtype != SO_F32 && nd.dtype != SO_F64)
                fail(SO_ERR_UNSUPPORTED, "array leaves must be Float32 or Float64");
            if (nd.l0 < 0 || nd.nch < 1) fail(SO_ERR_INVALID, "bad array shape");
            if (nd.l0 > 0 && !nd.p0) fail(SO_ERR_INVALID, "array leaf without data");
            N.len = Len{LK_FIN, nd.l0};
            array_ptr[i] = nd.p0;
            break;
        case SO_NODE_CONST:
            N.len = Len{LK_NUMEXT, 0};
            N.nch = 1;
            N.dtype = nd.i0;
            break;
        case SO_NODE_FUNC:
            if (!(nd.fs > 0)) fail(SO_ERR_LENGTH, "Unknown frame rate: function signals need a frame rate");
            N.len = Len{LK_INF, 0};
            N.nch = 1;
            N.dtype = SO_F64;
            break;
        case SO_NODE_UNTIL: {  // reference src/cutting.jl:130
            Node& c = kid(0);
            int64_t L = std::max<int64_t>(0, nd.l0);
            N.len = isinf_(c.len) ? Len{LK_FIN, L} : Len{LK_FIN, std::min(c.len.n, L)};
            N.nch = c.nch;
            N.dtype = c.dtype;
            break;
        }
        case SO_NODE_AFTER: {  // reference src/cutting.jl:134,174-181
            Node& c = kid(0);
            if (isinf_(c.len)) N.len = c.len;
            else {
                // The reference raises this from After's first nextblock (src/cutting.jl:160-181), i.e.
                // only when the node is evaluated: an After inside a Mix / Amplify whose result has
                // no frames is never asked for a block.  Recorded here, raised by lower() / plan_create.
                if (nd.l0 > c.len.n) N.short_skip = nd.l0;
                N.len = Len{LK_FIN, std::min(std::max<int64_t>(c.len.n - nd.l0, 0), c.len.n)};
            }
            N.nch = c.nch;
            N.dtype = c.dtype;
            break;
        }
        case SO_NODE_PAD: {  // reference src/padding.jl:13-14
            Node& c = kid(0);
            N.len = nd.i1 ? Len{LK_EXT, clean(c.len)} : Len{LK_INF, 0};
            N.nch = c.nch;
            N.dtype = c.dtype;
            if (nd.i0 == SO_PAD_VECTOR && !nd.p0) fail(SO_ERR_INVALID, "vector padding without values");
            // (an indexing pad -- cycle, mirror -- over something that is not an array is only an
            //  error once padding actually starts, reference src/padding.jl:163-177: see pad_pieces)
            break;
        }
        case SO_NODE_APPEND: {  // reference src/appending.jl:59-76
            if (N.kids.empty()) fail(SO_ERR_INVALID, "Append without signals");
            int64_t tot = 0;
            bool inf = false;
            for (size_t j = 0; j < N.kids.size(); ++j) {
                Node& c = nodes[N.kids[j]];
                if (c.nch != kid(0).nch) fail(SO_ERR_CHANNELS, "Append: children must be Uniform in channels");
                if (isinf_(c.len)) {
                    if (j + 1 < N.kids.size()) fail(SO_ERR_LENGTH, "Cannot Append to the end of an infinite signal");
                    inf = true;
                } else tot += c.len.n;
            }
            N.len = inf ? Len{LK_INF, 0} : Len{LK_FIN, tot};
            N.nch = kid(0).nch;
            break;
        }
        case SO_NODE_RAMP: {
            Node& c = kid(0);
            N.len = c.len;
            N.nch = c.nch;
            N.dtype = float_of(c.dtype);
            if (nd.l0 < 1) fail(SO_ERR_INVALID, "ramp length must be >= 1 frame");
            break;
        }
        case SO_NODE_MAP: {
            if (N.kids.empty()) fail(SO_ERR_INVALID, "MapSignal without signals");
            Len l = kid(0).len;
            for (size_t j = 1; j < N.kids.size(); ++j) l = map_maxlen(l, nodes[N.kids[j]].len);
            N.len = l;
            int fn = nd.i0;
            int t = kid(0).dtype;
            for (size_t j = 1; j < N.kids.size(); ++j) t = promote(t, nodes[N.kids[j]].dtype);
            if (fn == SO_MAP_DIV && t == SO_I64) t = SO_F64;
            if (fn == SO_MAP_TOELTYPE) t = nd.i3;
            N.dtype = t;
            if (nd.i1) {  // bychannel: Uniform(channels=true) already applied by the host
                for (int k : N.kids)
                    if (nodes[k].nch != kid(0).nch)
                        fail(SO_ERR_CHANNELS, "OperateOn: children must be Uniform in channels (host applies ToChannels)");
                N.nch = kid(0).nch;
            } else {
                switch (fn) {
                case SO_MAP_TUPLECAT: {
                    int s = 0;
                    for (int k : N.kids) s += nodes[k].nch;
                    N.nch = s;
                    break;
                }
                case SO_MAP_GETCHAN:
                    if (nd.i3 < 1 || nd.i3 > kid(0).nch) fail(SO_ERR_CHANNELS, "SelectChannel: channel out of range");
                    N.nch = 1;
                    break;
                case SO_MAP_AS1CHANNEL: N.nch = 1; break;
                case SO_MAP_ASNCHANNELS:
                    if (kid(0).nch != 1) fail(SO_ERR_CHANNELS, "No rule to convert signal with " + std::to_string(kid(0).nch) + " channels to a signal with " + std::to_string(nd.i3) + " channels.");
                    N.nch = nd.i3;
                    break;
                case SO_MAP_REVERSECH: N.nch = kid(0).nch; break;
                default: fail(SO_ERR_UNSUPPORTED, "cross-channel map function is not lowerable");
                }
            }
            break;
        }
        case SO_NODE_FILT_SOS: {
            Node& c = kid(0);
            N.len = c.len;
            N.nch = c.nch;
            N.dtype = float_of(c.dtype);
            if (nd.i0 < 1 || !nd.p0) fail(SO_ERR_INVALID, "Filt without second-order sections");
            break;
        }
        case SO_NODE_RESAMPLE: {  // reference src/filters.jl:165
            Node& c = kid(0);
            if (!(c.fs > 0) || !(nd.fs > 0)) fail(SO_ERR_LENGTH, "resampling needs known frame rates");
            if (isinf_(c.len)) N.len = c.len.k == LK_EXT ? Len{LK_INF, 0} : c.len;
            else N.len = Len{LK_FIN, (int64_t)std::ceil((double)c.len.n * nd.fs / c.fs)};
            N.nch = c.nch;
            N.dtype = float_of(c.dtype);
            if (!nd.p0 || nd.i2 < 1) fail(SO_ERR_INVALID, "resampler without taps");
            if (nd.i0 != SO_RS_FIR && (nd.i2 & 1) == 0) fail(SO_ERR_UNSUPPORTED, "resample_filter taps must have odd length");
            break;
        }
        case SO_NODE_NORMPOWER: {
            Node& c = kid(0);
            N.len = c.len;
            N.nch = c.nch;
            N.dtype = float_of(c.dtype);
            break;
        }
        default: fail(SO_ERR_INVALID, "unknown node kind " + std::to_string(nd.kind));
        }
        // nodes that ask their (first) child for a block whenever they are asked for one themselves
        if (!N.short_skip && !N.kids.empty() && nd.kind != SO_NODE_MAP) N.short_skip = nodes[N.kids[0]].short_skip;
        if (N.dtype == SO_I64 && nd.kind != SO_NODE_CONST && nd.kind != SO_NODE_UNTIL &&
            nd.kind != SO_NODE_AFTER && nd.kind != SO_NODE_PAD && nd.kind != SO_NODE_APPEND &&
            nd.kind != SO_NODE_MAP)
            fail(SO_ERR_UNSUPPORTED, "integer sample types are not lowered (SURVEY.md §8(b))");
        // cross-check against what the host computed (catches glue bugs early)
        if (nd.nframes != SO_LEN_UNCHECKED) {
            int64_t mine = isinf_(N.len) ? SO_LEN_INF : N.len.n;
            if (nd.nframes != SO_LEN_MISSING && nd.nframes != mine)
                fail(SO_ERR_INVALID, "node " + std::to_string(i) + " (kind " + std::to_string(nd.kind) + "): host nframes " + std::to_string(nd.nframes) + " != planner " + std::to_string(mine));
        }
        if (nd.nch != N.nch && nd.kind != SO_NODE_CONST && nd.kind != SO_NODE_FUNC)
            fail(SO_ERR_CHANNELS, "node " + std::to_string(i) + ": host nchannels " + std::to_string(nd.nch) + " != planner " + std::to_string(N.nch));
        if (nd.dtype != N.dtype && nd.kind != SO_NODE_CONST)
            fail(SO_ERR_INVALID, "node " + std::to_string(i) + ": host sampletype " + std::to_string(nd.dtype) + " != planner " + std::to_string(N.dtype));
    }
}

// ---------------------------------------------------------------------------
void Plan::count_array(int ni) {
    if (array_counted[ni]) return;
    array_counted[ni] = true;
    const so_node_t& nd = nodes[ni].nd;
    algo_bytes += nd.l0 * (int64_t)nd.nch * (int64_t)dsize(nd.dtype);
}

int Plan::stage_for(int ni, int kind) {
    auto it = stage_of_node.find(ni);
    if (it != stage_of_node.end()) return it->second;
    Stage S;
    S.kind = kind;
    S.node = ni;
    stages.push_back(S);
    stage_of_node[ni] = (int)stages.size() - 1;
    return (int)stages.size() - 1;
}

void Plan::use_stage(Stage& S, const Rect& r, const Map& m) {
    int64_t hi = m.sf ? r.b + m.df : m.df + 1;
    if (S.processed && hi > S.need) fail(SO_ERR_RUNTIME, "internal: stage need raised after processing");
    S.need = std::max(S.need, hi);
    const int64_t lo = m.sf ? r.a + m.df : m.df;
    if (S.processed && lo < S.base) fail(SO_ERR_RUNTIME, "internal: stage read before its first frame after processing");
    S.lo = std::min(S.lo, std::max<int64_t>(lo, 0));
}

// Frames the reference evaluates although nobody uses their values -- what `After` skips
// (src/cutting.jl:160-173 pulls the skipped blocks through the whole tree below it) and what a
// filter reads ahead of the frames asked for (src/filters.jl:221-262: whole blocks of `blocksize`
// input frames) -- raise the same errors there as used frames do.  Lower them for the errors only.
void Plan::check_frames(int ni, int64_t upto) {
    Node& C = nodes[ni];
    const int64_t hi = isinf_(C.len) ? upto : std::min(upto, C.len.n);
    if (hi <= C.checked_upto || hi >= BIG) return;
    const int64_t lo = C.checked_upto;
    C.checked_upto = hi;
    ++dry;
    try {
        (void)lower(ni, Rect{lo, hi, 0, C.nch}, Map{1, 0, 1, 0});
    } catch (const PlanError& e) {
        --dry;
        // (frames the engine never computes cannot run into its own limits)
        if (e.status == SO_ERR_UNSUPPORTED || e.status == SO_ERR_RUNTIME) return;
        throw;
    }
    --dry;
}

static DLeaf mk_leafmap(const Map& m) {
    DLeaf L{};
    L.sf = m.sf;
    L.df = m.df;
    L.sc = m.sc;
    L.dc = m.dc;
    L.buf = -1;
    L.mode = LM_PLAIN;
    return L;
}

std::vector<Piece> Plan::pad_pieces(int child, int padkind, double padvalue, const double* padvec,
                                    Rect r, Map m) {
    Node& C = nodes[child];
    int T = C.dtype;
    std::vector<Piece> out;
    if (r.a >= r.b) return out;
    switch (padkind) {
    case SO_PAD_VALUE: out.push_back({r, mk_const(roundto(T, padvalue), T)}); break;
    case SO_PAD_ZERO: out.push_back({r, mk_const(0.0, T)}); break;
    case SO_PAD_ONE: out.push_back({r, mk_const(1.0, T)}); break;
    case SO_PAD_VECTOR: {
        // per-channel constants: one piece per channel (channel counts of padded
        // vectors are tiny in practice)
        for (int c = r.c0; c < r.c1; ++c) {
            int64_t cn = (int64_t)m.sc * c + m.dc;
            out.push_back({Rect{r.a, r.b, c, c + 1}, mk_const(roundto(T, padvec[cn]), T)});
            if (m.sc == 0) {
                out.back().r.c1 = r.c1;
                break;
            }
        }
        break;
    }
    case SO_PAD_LASTFRAME: {
        if (isinf_(C.len) || C.len.n == 0)
            fail(SO_ERR_LENGTH, "Signal is length zero; there is no last frame to pad with.");
        out = lower(child, r, Map{0, C.len.n - 1, m.sc, m.dc});
        break;
    }
    case SO_PAD_CYCLE:
    case SO_PAD_MIRROR: {
        if (C.nd.kind != SO_NODE_ARRAY)
            fail(SO_ERR_INVALID, "Attemped to specify an indexing pad function for a signal which is not known to support `getindex`.");
        if (C.len.n == 0) fail(SO_ERR_LENGTH, "cannot index an empty array");
        // one piece per wrap of the array: cycle -> x[f - k*N]; mirror -> odd wraps run
        // backwards x[(k+1)*N - 1 - f]  (reference src/padding.jl:132-148)
        const int64_t Nc = C.len.n;
        auto wrap_load = [&](Rect rr, int sf, int64_t df) {
            Expr e;
            e.op = E_LOAD;
            e.dtype = C.dtype;
            e.leaf = mk_leafmap(Map{sf, df, m.sc, m.dc});
            e.leaf.fstride = C.nd.s0;
            e.leaf.cstride = C.nd.s1;
            e.leaf.dtype = C.dtype;
            e.array_node = child;
            e.mono = (m.sc == 0);
            out.push_back({rr, add_expr(e)});
        };
        count_array(child);
        if (m.sf == 0) {
            int64_t k = m.df / Nc, rem = m.df % Nc;
            bool rev = padkind == SO_PAD_MIRROR && (k & 1);
            wrap_load(r, 0, rev ? Nc - 1 - rem : rem);
            break;
        }
        if ((r.b - r.a) / Nc > 65536)
            fail(SO_ERR_UNSUPPORTED, "cycle/mirror padding over more than 65536 repetitions is not lowered");
        for (int64_t f0 = r.a + m.df; f0 < r.b + m.df;) {
            int64_t k = f0 / Nc;
            int64_t f1 = std::min(r.b + m.df, (k + 1) * Nc);
            Rect rr{f0 - m.df, f1 - m.df, r.c0, r.c1};
            bool rev = padkind == SO_PAD_MIRROR && (k & 1);
            if (rev) wrap_load(rr, -1, (k + 1) * Nc - 1 - m.df);
            else wrap_load(rr, 1, m.df - k * Nc);
            f0 = f1;
        }
        break;
    }
    default: fail(SO_ERR_INVALID, "unknown padding kind");
    }
    return out;
}

// child extended with padding past its end (Pad / Extend / MapSignal's
// Extend.(signals,padding), reference src/mapsignal.jl:26, src/padding.jl:97-101)
std::vector<Piece> Plan::lower_padded(int ni, int padkind, double padvalue, const double* padvec,
                                      Rect r, Map m, bool always_pad) {
    Node& C = nodes[ni];
    if (isinf_(C.len) && !always_pad) return lower(ni, r, m);
    if (isinf_(C.len)) return lower(ni, r, m);
    int64_t Nc = C.len.n;
    std::vector<Piece> out;
    if (m.sf == 0) {
        if (m.df < Nc) return lower(ni, r, m);
        return pad_pieces(ni, padkind, padvalue, padvec, r, m);
    }
    int64_t s = std::min(std::max(Nc - m.df, r.a), r.b);
    if (s > r.a) out = lower(ni, Rect{r.a, s, r.c0, r.c1}, m);
    if (s < r.b) {
        auto p = pad_pieces(ni, padkind, padvalue, padvec, Rect{s, r.b, r.c0, r.c1}, m);
        out.insert(out.end(), p.begin(), p.end());
    }
    return out;
}

// intersect the children's piece partitions of `r` and fold `op` left to right
std::vector<Piece> Plan::combine(const std::vector<std::vector<Piece>>& kids, Rect r, int op,
                                 int force_dtype) {
    std::vector<int64_t> fb{r.a, r.b};
    std::vector<int> cb{r.c0, r.c1};
    for (auto& ps : kids)
        for (auto& p : ps) {
            fb.push_back(p.r.a);
            fb.push_back(p.r.b);
            cb.push_back(p.r.c0);
            cb.push_back(p.r.c1);
        }
    std::sort(fb.begin(), fb.end());
    fb.erase(std::unique(fb.begin(), fb.end()), fb.end());
    std::sort(cb.begin(), cb.end());
    cb.erase(std::unique(cb.begin(), cb.end()), cb.end());
    std::vector<Piece> out;
    for (size_t ci = 0; ci + 1 < cb.size(); ++ci)
        for (size_t fi = 0; fi + 1 < fb.size(); ++fi) {
            Rect cell{fb[fi], fb[fi + 1], cb[ci], cb[ci + 1]};
            if (cell.a < r.a || cell.b > r.b || cell.c0 < r.c0 || cell.c1 > r.c1) continue;
            int acc = -1;
            for (auto& ps : kids) {
                int e = -1;
                for (auto& p : ps)
                    if (p.r.a <= cell.a && cell.b <= p.r.b && p.r.c0 <= cell.c0 && cell.c1 <= p.r.c1) {
                        e = p.e;
                        break;
                    }
                if (e < 0) fail(SO_ERR_RUNTIME, "internal: child pieces do not cover the cell");
                acc = acc < 0 ? e : mk_bin(op, acc, e);
            }
            (void)force_dtype;
            // merge with the previous piece along frames when the expression is identical
            out.push_back({cell, acc});
        }
    return out;
}

std::vector<Piece> Plan::lower(int ni, Rect r, Map m) {
    std::vector<Piece> out;
    if (r.a >= r.b || r.c0 >= r.c1) return out;
    Node& N = nodes[ni];
    const so_node_t& nd = N.nd;
    if (nd.kind == SO_NODE_MAP || nd.kind == SO_NODE_APPEND)  // frames wanted: every child is evaluated
        for (int k : N.kids)
            if (nodes[k].short_skip)
                fail(SO_ERR_LENGTH, "Signal is too short to skip " + std::to_string(nodes[k].short_skip) + " frames");
    switch (nd.kind) {
    case SO_NODE_ARRAY: {
        Expr e;
        e.op = E_LOAD;
        e.dtype = N.dtype;
        e.leaf = mk_leafmap(m);
        e.leaf.fstride = nd.s0;
        e.leaf.cstride = nd.s1;
        e.leaf.dtype = N.dtype;
        e.array_node = ni;
        e.mono = (m.sc == 0);
        if (!dry && nd.l1 > 0) {
            // a leaf of which only the frames >= l1 are resident (streams fed block by block keep a
            // bounded tail of their input): nothing earlier may be read
            const int64_t first = (m.sf > 0 ? r.a : m.sf < 0 ? r.b - 1 : 0) * m.sf + m.df;
            if (first < nd.l1)
                fail(SO_ERR_LENGTH, "frame " + std::to_string(first) + " of a streamed leaf is needed, but frames before " +
                                        std::to_string(nd.l1) + " are no longer resident (raise the stream's history)");
        }
        if (!dry) count_array(ni);
        out.push_back({r, add_expr(e)});
        return out;
    }
    case SO_NODE_CONST: out.push_back({r, mk_const(nd.d0, nd.i0)}); return out;
    case SO_NODE_FUNC: {
        Expr e;
        e.op = E_FUNC;
        e.dtype = SO_F64;
        e.leaf = mk_leafmap(m);
        e.leaf.mode = nd.i0;
        e.leaf.flag = nd.i1;
        e.leaf.v0 = nd.d0;
        e.leaf.v1 = nd.d1;
        e.leaf.v2 = nd.fs;
        e.heavy = true;
        out.push_back({r, add_expr(e)});
        return out;
    }
    case SO_NODE_UNTIL: return lower(N.kids[0], r, m);
    case SO_NODE_AFTER:
        if (nd.l0 > 0) check_frames(N.kids[0], nd.l0);
        return lower(N.kids[0], r, Map{m.sf, m.df + std::max<int64_t>(0, nd.l0), m.sc, m.dc});
    case SO_NODE_PAD:
        return lower_padded(N.kids[0], nd.i0, nd.d0, (const double*)nd.p0, r, m, true);
    case SO_NODE_APPEND: {  // reference src/appending.jl:92-110
        int64_t off = 0;
        for (size_t k = 0; k < N.kids.size(); ++k) {
            Node& C = nodes[N.kids[k]];
            int64_t end = isinf_(C.len) ? BIG : off + C.len.n;
            if (m.sf == 0) {
                if (m.df >= off && m.df < end) return lower(N.kids[k], r, Map{0, m.df - off, m.sc, m.dc});
            } else {
                int64_t a = std::max(r.a, off - m.df), b = std::min(r.b, end == BIG ? r.b : end - m.df);
                if (a < b) {
                    auto p = lower(N.kids[k], Rect{a, b, r.c0, r.c1}, Map{1, m.df - off, m.sc, m.dc});
                    out.insert(out.end(), p.begin(), p.end());
                }
            }
            off = end;
        }
        return out;
    }
    case SO_NODE_RAMP: {  // reference src/ramps.jl:56-119
        int64_t Ntot = clean(N.len), R = nd.l0;
        int onedt = float_of(nodes[N.kids[0]].dtype);
        auto ramp_expr = [&]() {
            Expr e;
            e.op = E_RAMP;
            e.dtype = SO_F64;
            e.leaf = mk_leafmap(m);
            e.leaf.mode = nd.i1;
            e.leaf.flag = nd.i0;
            e.leaf.v0 = (double)R;
            e.leaf.modn = nd.i0 ? Ntot - R : 0;
            e.heavy = true;
            return add_expr(e);
        };
        int64_t B;  // boundary in node frames: [0,B) first region, [B,inf) second
        if (nd.i0 == 0) B = R;
        else {
            if (Ntot >= BIG) B = BIG;
            else {
                B = Ntot - R;
                if (B < 0) fail(SO_ERR_INVALID, "RampOff longer than the signal is undefined in the reference (src/ramps.jl:79-80)");
            }
        }
        auto region = [&](Rect rr, bool first) {
            bool is_ramp = (nd.i0 == 0) ? first : !first;
            out.push_back({rr, is_ramp ? ramp_expr() : mk_const(1.0, onedt)});
        };
        if (m.sf == 0) {
            region(r, m.df < B);
            return out;
        }
        int64_t s = B >= BIG ? r.b : std::min(std::max(B - m.df, r.a), r.b);
        if (s > r.a) region(Rect{r.a, s, r.c0, r.c1}, true);
        if (s < r.b) region(Rect{s, r.b, r.c0, r.c1}, false);
        return out;
    }
    case SO_NODE_MAP: {
        int fn = nd.i0;
        const double* pv = nullptr;
        switch (fn) {
        case SO_MAP_ADD:
        case SO_MAP_MUL:
        case SO_MAP_SUB:
        case SO_MAP_DIV: {
            std::vector<std::vector<Piece>> ks;
            for (int k : N.kids) ks.push_back(lower_padded(k, nd.i2, nd.d0, pv, r, m, false));
            if (N.kids.size() == 1) {
                if (fn != SO_MAP_SUB) return ks[0];
                for (auto& p : ks[0]) out.push_back({p.r, mk_un(E_NEG, p.e, exprs[p.e].dtype)});
                return out;
            }
            int op = fn == SO_MAP_ADD ? E_ADD : fn == SO_MAP_MUL ? E_MUL : fn == SO_MAP_SUB ? E_SUB : E_DIV;
            return combine(ks, r, op, -1);
        }
        case SO_MAP_TOELTYPE: {
            auto ps = lower_padded(N.kids[0], nd.i2, nd.d0, pv, r, m, false);
            for (auto& p : ps) {
                int t = exprs[p.e].dtype;
                int e = p.e;
                if (nd.i3 == SO_F32 && t != SO_F32) e = mk_un(E_ROUND32, e, SO_F32);
                else if (nd.i3 != t) e = mk_un(E_RETYPE, e, nd.i3);
                out.push_back({p.r, e});
            }
            return out;
        }
        case SO_MAP_TUPLECAT: {  // reference src/mapsignal.jl:361-362
            int off = 0;
            for (int k : N.kids) {
                int nc = nodes[k].nch;
                if (m.sc == 0) {
                    if (m.dc >= off && m.dc < off + nc)
                        return lower_padded(k, nd.i2, nd.d0, pv, r, Map{m.sf, m.df, 0, m.dc - off}, false);
                } else {
                    // node channel cn = sc*c + dc must lie in [off, off+nc)
                    int c_lo, c_hi;
                    if (m.sc > 0) {
                        c_lo = (int)std::max<int64_t>(r.c0, off - m.dc);
                        c_hi = (int)std::min<int64_t>(r.c1, off + nc - m.dc);
                    } else {
                        // cn = dc - c  in [off, off+nc)  =>  c in (dc-off-nc, dc-off]
                        c_lo = (int)std::max<int64_t>(r.c0, m.dc - off - nc + 1);
                        c_hi = (int)std::min<int64_t>(r.c1, m.dc - off + 1);
                    }
                    if (c_lo < c_hi) {
                        auto p = lower_padded(k, nd.i2, nd.d0, pv, Rect{r.a, r.b, c_lo, c_hi},
                                              Map{m.sf, m.df, m.sc, m.dc - off}, false);
                        out.insert(out.end(), p.begin(), p.end());
                    }
                }
                off += nc;
            }
            return out;
        }
        case SO_MAP_GETCHAN:
            return lower_padded(N.kids[0], nd.i2, nd.d0, pv, r, Map{m.sf, m.df, 0, nd.i3 - 1}, false);
        case SO_MAP_ASNCHANNELS:
            return lower_padded(N.kids[0], nd.i2, nd.d0, pv, r, Map{m.sf, m.df, 0, 0}, false);
        case SO_MAP_REVERSECH: {
            int nc = nodes[N.kids[0]].nch;
            return lower_padded(N.kids[0], nd.i2, nd.d0, pv, r,
                                Map{m.sf, m.df, -m.sc, (int64_t)nc - 1 - m.dc}, false);
        }
        case SO_MAP_AS1CHANNEL: {  // sum(x) over channels, reference src/reformatting.jl:156
            int nc = nodes[N.kids[0]].nch;
            if (nc > 64) fail(SO_ERR_UNSUPPORTED, "ToChannels(x,1) over more than 64 channels is not lowered yet");
            std::vector<std::vector<Piece>> ks;
            for (int c = 0; c < nc; ++c)
                ks.push_back(lower_padded(N.kids[0], nd.i2, nd.d0, pv, r, Map{m.sf, m.df, 0, c}, false));
            if (nc == 1) return ks[0];
            return combine(ks, r, E_ADD, -1);
        }
        }
        fail(SO_ERR_UNSUPPORTED, "map function not lowerable");
    }
    case SO_NODE_FILT_SOS:
    case SO_NODE_RESAMPLE:
    case SO_NODE_NORMPOWER: {
        int kind = nd.kind == SO_NODE_FILT_SOS ? ST_SOS : nd.kind == SO_NODE_RESAMPLE ? ST_RESAMPLE : ST_NORM;
        if (dry) {  // the frames of the child this node's frames [0, F) are made of
            const Node& C = nodes[N.kids[0]];
            const int64_t F = m.sf ? r.b + m.df : m.df + 1;
            int64_t Fc = BIG;
            if (kind == ST_SOS) Fc = (F + std::max(1, nd.i1) - 1) / std::max(1, nd.i1) * std::max(1, nd.i1);
            else if (kind == ST_RESAMPLE && N.fs > 0 && C.fs > 0) Fc = (int64_t)std::ceil((double)F * C.fs / N.fs) + 1;
            if (kind == ST_NORM && isinf_(N.len))
                fail(SO_ERR_LENGTH, "Cannot normalize an infinite-length signal. Please use `Until` to take a prefix of the signal");
            check_frames(N.kids[0], Fc);
            out.push_back({r, mk_const(0.0, N.dtype)});
            return out;
        }
        int sid = stage_for(ni, kind);
        if (in_norm > 0) {
            stages[sid].under_norm = true;
            stages[sid].norm_df = std::max(stages[sid].norm_df, m.sf == 1 ? m.df : (int64_t)1 << 60);
        }
        if (kind == ST_NORM) {
            if (isinf_(N.len))
                fail(SO_ERR_LENGTH, "Cannot normalize an infinite-length signal. Please use `Until` to take a prefix of the signal");
            stages[sid].need = N.len.n;
        } else {
            use_stage(stages[sid], r, m);
        }
        if (stages[sid].out_buf < 0) {
            // Normpower of a filter's or resampler's output: `vals` IS that stage's buffer (nothing scales it in place --
            // the division by the rms is part of whoever reads it), not a copy of it: one pass over the signal less
            // (Filt |> Normpower: 1.70 -> 1.4 ms for 12.5 M x 8; reference src/filters.jl:296-305 fills vals from its child)
            const Node& Cn = nodes[N.kids[0]];
            const bool stage_child = (Cn.nd.kind == SO_NODE_FILT_SOS || Cn.nd.kind == SO_NODE_RESAMPLE) && Cn.dtype == N.dtype &&
                                     Cn.nch == N.nch && !isinf_(Cn.len) && Cn.len.n == N.len.n;
            if (kind == ST_NORM && stage_child && !std::getenv("SIGOPS_NORM_COPY")) {
                const int cs = stage_for(N.kids[0], Cn.nd.kind == SO_NODE_FILT_SOS ? ST_SOS : ST_RESAMPLE);
                if (stages[cs].out_buf < 0) stages[cs].out_buf = new_buf(0, Cn.nch, Cn.dtype);
                stages[sid].out_buf = stages[cs].out_buf;
                stages[sid].norm_alias = true;
            } else {
                stages[sid].out_buf = new_buf(0, N.nch, N.dtype);  // sized in finalize()
                // ... and of a plain array (through `Until` / `After`): the sum of squares is taken over the array where it
                // lies and whoever reads `vals` divides the array's own frames (three passes over the signal -> two)
                int k = N.kids[0];
                while (nodes[k].nd.kind == SO_NODE_UNTIL || (nodes[k].nd.kind == SO_NODE_AFTER && !nodes[k].short_skip)) k = nodes[k].kids[0];
                const so_node_t& a = nodes[k].nd;
                if (kind == ST_NORM && a.kind == SO_NODE_ARRAY && a.s0 == 1 && a.l1 == 0 && (a.s1 > 0 || a.nch == 1) && nodes[k].dtype == N.dtype &&
                    nodes[k].nch == N.nch && !std::getenv("SIGOPS_NORM_COPY"))
                    stages[sid].norm_direct = true;
            }
            if (kind == ST_NORM) stages[sid].rms_buf = raw_buf(8);
        }
        if (kind == ST_NORM && stages[sid].norm_direct) {  // the child's own pieces ./ rms
            Expr s;
            s.op = E_SCALAR;
            s.dtype = SO_F64;
            s.leaf.buf = stages[sid].rms_buf;
            const int se = add_expr(s);
            auto ps = lower(N.kids[0], r, m);
            for (auto& p : ps) {
                Expr d;
                d.op = E_DIV;
                d.dtype = N.dtype;
                d.a = p.e;
                d.b = se;
                d.mono = exprs[p.e].mono;
                p.e = add_expr(d);
            }
            return ps;
        }
        Expr e;
        e.op = E_LOAD;
        e.dtype = N.dtype;
        e.leaf = mk_leafmap(m);
        e.leaf.fstride = 1;
        e.leaf.cstride = -1;  // = pitch of the buffer, patched in finalize()
        e.leaf.dtype = N.dtype;
        e.leaf.buf = stages[sid].out_buf;
        e.mono = (m.sc == 0);
        int le = add_expr(e);
        if (kind == ST_NORM) {  // vals ./= rms   (reference src/filters.jl:304-305)
            Expr s;
            s.op = E_SCALAR;
            s.dtype = SO_F64;
            s.leaf.buf = stages[sid].rms_buf;
            int se = add_expr(s);
            Expr d;
            d.op = E_DIV;
            d.dtype = N.dtype;  // stored back into the Float32/Float64 `vals`
            d.a = le;
            d.b = se;
            d.mono = exprs[le].mono;
            le = add_expr(d);
        }
        out.push_back({r, le});
        return out;
    }
    }
    fail(SO_ERR_INVALID, "unknown node kind");
}

// ---------------------------------------------------------------------------
int Plan::depth(int ei) const {
    const Expr& e = exprs[ei];
    switch (e.op) {
    case E_CONST:
    case E_LOAD:
    case E_SCALAR:
    case E_FUNC:
    case E_RAMP: return 1;
    case E_NEG:
    case E_ROUND32:
    case E_RETYPE: return depth(e.a);
    default: return std::max(depth(e.a), depth(e.b) + 1);
    }
}

int Plan::add_leaf(const Expr& e) {
    leaves.push_back(e.leaf);
    leaf_array_node.push_back(e.op == E_LOAD ? e.array_node : -1);
    return (int)leaves.size() - 1;
}

void Plan::gen(int ei, std::vector<DOp>& code, std::map<int, int>& hoisted,
               std::vector<DOp>& fcode, bool allow_hoist) {
    const Expr e = exprs[ei];
    if (allow_hoist && e.mono && e.heavy) {
        auto it = hoisted.find(ei);
        if (it == hoisted.end() && (int)hoisted.size() < kMaxFrameSlots) {
            std::map<int, int> none;
            gen(ei, fcode, none, fcode, false);
            int slot = (int)hoisted.size();
            fcode.push_back(DOp{OP_STOREF, slot});
            hoisted[ei] = slot;
            it = hoisted.find(ei);
        }
        if (it != hoisted.end()) {
            code.push_back(DOp{OP_LOADF, it->second});
            return;
        }
    }
    switch (e.op) {
    case E_CONST: code.push_back(DOp{OP_CONST, add_leaf(e)}); return;
    case E_LOAD: code.push_back(DOp{OP_LOAD, add_leaf(e)}); return;
    case E_SCALAR: code.push_back(DOp{OP_SCALAR, add_leaf(e)}); return;
    case E_FUNC: code.push_back(DOp{OP_FUNC, add_leaf(e)}); return;
    case E_RAMP: code.push_back(DOp{OP_RAMP, add_leaf(e)}); return;
    case E_RETYPE: gen(e.a, code, hoisted, fcode, allow_hoist); return;
    case E_NEG:
        gen(e.a, code, hoisted, fcode, allow_hoist);
        code.push_back(DOp{OP_NEG, 0});
        return;
    case E_ROUND32:
        gen(e.a, code, hoisted, fcode, allow_hoist);
        code.push_back(DOp{OP_ROUND32, 0});
        return;
    default: {
        // `per-frame value (+|*) samples`: samples first (the same sum / product bit for bit), so that
        // Mix(sin, x) and Amplify(gain, x) compile to the kernel's chain form like Mix(x, sin)
        // (only when the per-frame operand is ONE opcode -- a hoisted slot or a constant -- so that the
        //  stack depth depth() predicts for the original order still holds)
        const Expr& ea = exprs[e.a];
        const bool one_op = ea.op == E_CONST || (allow_hoist && ea.heavy && (hoisted.count(e.a) || (int)hoisted.size() < kMaxFrameSlots));
        const bool swap = (e.op == E_ADD || e.op == E_MUL) && ea.mono && one_op && !exprs[e.b].mono;
        gen(swap ? e.b : e.a, code, hoisted, fcode, allow_hoist);
        gen(swap ? e.a : e.b, code, hoisted, fcode, allow_hoist);
        int oc = e.op == E_ADD ? OP_ADD : e.op == E_SUB ? OP_SUB : e.op == E_MUL ? OP_MUL : OP_DIV;
        code.push_back(DOp{oc, 0});
        if (e.dtype == SO_F32) code.push_back(DOp{OP_ROUND32, 0});  // Julia Float32 arithmetic
        return;
    }
    }
}

// Per-frame slots the expression needs: its maximal channel-independent generator / ramp
// sub-expressions (what gen() hoists; identical sub-expressions are counted twice here).
int Plan::frame_slots(int ei) const {
    const Expr& e = exprs[ei];
    if (e.mono && e.heavy) return 1;
    switch (e.op) {
    case E_CONST:
    case E_LOAD:
    case E_SCALAR:
    case E_FUNC:
    case E_RAMP: return 0;
    case E_NEG:
    case E_ROUND32:
    case E_RETYPE: return frame_slots(e.a);
    default: return frame_slots(e.a) + frame_slots(e.b);
    }
}

// copy of expression `ei` that, evaluated at (n, c), gives the original at (n + a, c + c0)
int Plan::shift_expr(int ei, int64_t a, int c0) {
    Expr e = exprs[ei];
    switch (e.op) {
    case E_CONST:
    case E_SCALAR: return ei;
    case E_LOAD:
        e.leaf.df += (int64_t)e.leaf.sf * a;
        e.leaf.dc += (int64_t)e.leaf.sc * c0;
        return add_expr(e);
    case E_FUNC:
    case E_RAMP:
        if (e.leaf.sf) e.leaf.df += a;  // (func_eval / ramp_eval: (sf ? n : 0) + df)
        return add_expr(e);
    case E_NEG:
    case E_ROUND32:
    case E_RETYPE: e.a = shift_expr(e.a, a, c0); return add_expr(e);
    default:
        e.a = shift_expr(e.a, a, c0);
        e.b = shift_expr(e.b, a, c0);
        return add_expr(e);
    }
}

// Evaluate `ei` over rectangle r into a scratch buffer with one more pointwise step (appended
// to `pre`) and return a plain load of that buffer.  Values are stored in the expression's own
// sample type, i.e. exactly as the interpreter would have passed them on.
int Plan::materialise(int ei, const Rect& r, std::vector<int>& pre) {
    const int dt = exprs[ei].dtype == SO_F32 ? SO_F32 : SO_F64;
    const int nchp = r.c1 - r.c0;
    const int64_t nf = r.b - r.a;
    const int buf = new_buf(nf, nchp, dt);
    std::vector<Piece> one{Piece{Rect{0, nf, 0, nchp}, shift_expr(ei, r.a, r.c0)}};
    pre.push_back(emit_pointwise(one, buf, dt));
    Expr l;
    l.op = E_LOAD;
    l.dtype = exprs[ei].dtype;
    l.leaf = mk_leafmap(Map{1, -r.a, 1, -(int64_t)r.c0});
    l.leaf.fstride = 1;
    l.leaf.cstride = -1;  // = pitch of the buffer, patched in finalize()
    l.leaf.dtype = dt;
    l.leaf.buf = buf;
    l.mono = false;
    return add_expr(l);
}

// Rewrite an expression that exceeds the interpreter's limits (stack depth kStackDepth,
// kMaxFrameSlots per-frame slots) into one that fits, by materialising sub-expressions: the
// reference has no such limits (it recurses through `frame`), so neither may the lowering.
int Plan::legalise(int ei, const Rect& r, std::vector<int>& pre) {
    Expr e = exprs[ei];
    switch (e.op) {
    case E_CONST:
    case E_LOAD:
    case E_SCALAR:
    case E_FUNC:
    case E_RAMP: return ei;
    case E_NEG:
    case E_ROUND32:
    case E_RETYPE: {
        const int a = legalise(e.a, r, pre);
        if (a == e.a) return ei;
        e.a = a;
        e.mono = exprs[a].mono;
        e.heavy = exprs[a].heavy;
        return add_expr(e);
    }
    default: break;
    }
    int a = legalise(e.a, r, pre), b = legalise(e.b, r, pre);
    if (std::max(depth(a), depth(b) + 1) > kStackDepth) b = materialise(b, r, pre);
    auto slots_of = [&](int x, int y) {
        return exprs[x].mono && exprs[y].mono && (exprs[x].heavy || exprs[y].heavy) ? 1 : frame_slots(x) + frame_slots(y);
    };
    while (slots_of(a, b) > kMaxFrameSlots) {
        if (frame_slots(a) >= frame_slots(b)) a = materialise(a, r, pre);
        else b = materialise(b, r, pre);
    }
    if (a == e.a && b == e.b) return ei;
    e.a = a;
    e.b = b;
    e.mono = exprs[a].mono && exprs[b].mono;
    e.heavy = exprs[a].heavy || exprs[b].heavy;
    return add_expr(e);
}

void Plan::push_pw_step(int idx) {
    for (int q : pw[idx].pre) push_pw_step(q);
    steps.push_back(Step{0, idx, pw[idx].rtc ? "k_pointwise_rtc" : "k_pointwise", pw[idx].bytes});
}

// compile pieces into one pointwise launch writing `out_buf` (or the final output)
int Plan::emit_pointwise(const std::vector<Piece>& ps_in, int out_buf, int out_dtype) {
    std::vector<Piece> ps = ps_in;
    // ---- hipRTC (rtc.cpp): the step as straight-line source instead of interpreter programs.
    //      SIGOPS_RTC=1: every pointwise step of up to 32 pieces; 0: never; default: steps the interpreter
    //      could only run after materialising sub-expressions (too deep / too many per-frame values), when
    //      they are big enough to pay for a compile ----
    const void* rtc_fn = nullptr;
    std::vector<int> rtc_leaves;
    {
        const char* ev = std::getenv("SIGOPS_RTC");
        const int mode = ev ? std::atoi(ev) : 2;
        bool over = false;
        int64_t elems = 0;
        int np = 0;
        for (auto& p : ps) {
            if (p.r.a >= p.r.b || p.r.c0 >= p.r.c1) continue;
            ++np;
            elems += (p.r.b - p.r.a) * (int64_t)(p.r.c1 - p.r.c0);
            over = over || depth(p.e) > kStackDepth || frame_slots(p.e) > kMaxFrameSlots;
        }
        const bool want = np > 0 && np <= 32 && (mode == 1 || (mode == 2 && over && elems >= (1 << 20)));
        // ... and big steps the interpreter can run as they are: the specialised kernel if it is already there (this
        // process, or the code objects on disk), else the interpreter now and a background compile for later plans
        // (the straight-line form runs 15-20 % faster on Float64 maps, 1.6-2.1x on Float32 ones: tools/k1_probe.py)
        const bool later = !want && mode == 2 && np > 0 && np <= 32 && elems >= (1 << 22) && !std::getenv("SIGOPS_RTC_NOASYNC");
        if ((want || later) && !dry) {
            const size_t leaves_before = leaves.size();
            std::string err;
            try {
                const std::string src = rtc_source(ps);
                rtc_fn = want ? rtc_kernel(src, device, err) : rtc_kernel_if_ready(src, device);
                if (!rtc_fn && later) err = "not compiled yet (queued)";
            } catch (const PlanError& e) {
                err = e.msg;
            }
            if (rtc_fn)
                for (size_t i = leaves_before; i < leaves.size(); ++i) rtc_leaves.push_back((int)i);
            else if (std::getenv("SIGOPS_DEBUG_PLAN"))
                std::fprintf(stderr, "[sigops] hipRTC not used: %s\n", err.c_str());
        }
    }
    std::vector<int> pre;
    for (auto& p : ps) {
        if (rtc_fn) break;
        if (p.r.a >= p.r.b || p.r.c0 >= p.r.c1) continue;
        if (depth(p.e) > kStackDepth || frame_slots(p.e) > kMaxFrameSlots) p.e = legalise(p.e, p.r, pre);
    }
    PwStep st;
    st.pre = pre;
    st.rtc = rtc_fn;
    st.rtc_leaves = rtc_leaves;
    st.piece0 = (int)pieces.size();
    st.out_buf = out_buf;
    int64_t blk = 0;
    constexpr int E = kPointwiseE;
    for (auto& p : ps) {
        if (p.r.a >= p.r.b || p.r.c0 >= p.r.c1) continue;
        if (!rtc_fn && depth(p.e) > kStackDepth)
            fail(SO_ERR_UNSUPPORTED, "expression too deep for the fused pointwise kernel (stack depth > 4)");
        std::vector<DOp> code, fcode;
        std::map<int, int> hoisted;
        int nchp = p.r.c1 - p.r.c0;
        if (rtc_fn) {  // geometry only: the programs are in the compiled kernel
            DPiece d{};
            d.depth = 2;
            d.a = p.r.a;
            d.b = p.r.b;
            d.c0 = p.r.c0;
            d.c1 = p.r.c1;
            d.nblk_f = (d.b - d.a + kBlock * E - 1) / (kBlock * E);
            d.sub = (int)std::max<int64_t>(1, std::min<int64_t>(2, d.nblk_f / 16384));
            d.nblk_f = (d.nblk_f + d.sub - 1) / d.sub;
            int chc = nchp;
            if (d.nblk_f < 2048 && nchp > 1) {
                int64_t want = (2048 + d.nblk_f - 1) / d.nblk_f;
                chc = (int)std::max<int64_t>(1, nchp / std::min<int64_t>(want, nchp));
            }
            d.chc = chc;
            d.block0 = blk;
            blk += d.nblk_f * ((nchp + chc - 1) / chc);
            pieces.push_back(d);
            st.bytes += (d.b - d.a) * (int64_t)nchp * (int64_t)dsize(out_dtype);
            continue;
        }
        // generators / ramps always go to the per-frame program: the per-sample interpreter
        // has no transcendental opcodes
        gen(p.e, code, hoisted, fcode, true);
        for (auto& o : code)
            if (o.code == OP_FUNC || o.code == OP_RAMP)
                fail(SO_ERR_UNSUPPORTED, "more than 4 distinct generator/ramp sub-expressions in one fused piece");
        // a Float32 operation at the root of a piece that is stored as Float32: its rounding IS the store's (the same value
        // rounded twice) -- without the op the program of `x32 ./ rms` is `array (op) scalar` and takes the chain path below
        if (out_dtype == SO_F32 && code.size() >= 2 && code.back().code == OP_ROUND32 && !std::getenv("SIGOPS_K1_KEEPROUND")) code.pop_back();
        DPiece d{};
        d.depth = std::max(2, depth(p.e));
        if (d.depth > 2) st.deep = true;
        // `array (op) F_s (op) F_t ...` over a planar unit-stride leaf: the kernel's chain path
        if (d.depth <= 2 && !code.empty() && code[0].code == OP_LOAD && (code.size() & 1) && code.size() <= 9 &&
            !std::getenv("SIGOPS_K1_NOCHAIN")) {
            const DLeaf& L = leaves[code[0].arg];
            // (a unit frame stride, or an interleaved leaf: channels adjacent, frames nch apart)
            const bool leaf_il = L.cstride == 1 && L.fstride > 1 && L.sc == 1;
            bool ok = L.mode == LM_PLAIN && L.sf == 1 && (L.fstride == 1 || leaf_il);
            for (size_t i = 1; ok && i + 1 < code.size(); i += 2)
                ok = (code[i].code == OP_LOADF || code[i].code == OP_CONST || code[i].code == OP_SCALAR) && code[i + 1].code >= OP_ADD && code[i + 1].code <= OP_DIV;
            if (ok) {
                d.chain = 1;
                st.chain = true;
                if (leaf_il) st.il = true;
            }
        }
        d.a = p.r.a;
        d.b = p.r.b;
        d.c0 = p.r.c0;
        d.c1 = p.r.c1;
        d.frame_pc = (int)ops.size();
        d.frame_len = (int)fcode.size();
        ops.insert(ops.end(), fcode.begin(), fcode.end());
        d.samp_pc = (int)ops.size();
        d.samp_len = (int)code.size();
        ops.insert(ops.end(), code.begin(), code.end());
        d.nblk_f = (d.b - d.a + kBlock * E - 1) / (kBlock * E);
        // long pieces: several blocks per workgroup (amortises the per-workgroup lookup latency),
        // keeping at least ~16 workgroups per CU
        d.sub = (int)std::max<int64_t>(1, std::min<int64_t>(2, d.nblk_f / 16384));  // (sweep on 26 M x 8: 1-2 best, 8 -10 %)
        if (const char* ev = std::getenv("SIGOPS_K1_SUB")) d.sub = std::max(1, std::min(64, std::atoi(ev)));  // tuning knob
        d.nblk_f = (d.nblk_f + d.sub - 1) / d.sub;
        // channel chunking: keep all channels in one workgroup unless the piece is
        // too short to fill the machine along frames
        int chc = nchp;
        if (d.nblk_f < 2048 && nchp > 1) {
            int64_t want = (2048 + d.nblk_f - 1) / d.nblk_f;
            chc = (int)std::max<int64_t>(1, nchp / std::min<int64_t>(want, nchp));
        }
        if (const char* ev = std::getenv("SIGOPS_K1_CHC")) chc = std::max(1, std::min(nchp, std::atoi(ev)));  // tuning knob
        d.chc = chc;
        int nbc = (nchp + chc - 1) / chc;
        d.block0 = blk;
        blk += d.nblk_f * nbc;
        pieces.push_back(d);
        st.bytes += (d.b - d.a) * (int64_t)nchp * (int64_t)dsize(out_dtype);
    }
    st.npieces = (int)pieces.size() - st.piece0;
    st.nblocks = blk;
    pw.push_back(st);
    return (int)pw.size() - 1;
}

// The C-ABI entry points run on the plan's device and leave the caller's current device as they
// found it (a single process may drive several GPUs).

// ---------------------------------------------------------------------------
// Window aliasing.  When the root of the tree is `Append` / `Ramp` / `Amplify(number)` ... over
// whole stage outputs (config 4: Append of Mix |> Filt |> Ramp scenes; any Append of filtered or
// resampled children), the root pointwise launch used to copy every stage buffer into the result:
// a read and a write of the whole output for nothing.  Here a stage whose buffer is read by root
// pieces only, 1:1 (same channel, frame + constant offset, every frame exactly once), writes its
// window of the result itself; identity pieces disappear and the remaining ones (ramp edges,
// gains) run IN PLACE on the result (element-wise with the same index on both sides).
void Plan::try_window_alias(std::vector<Piece>& rootp) {
    if (std::getenv("SIGOPS_NO_WINDOW_ALIAS") || out.frame_stride != 1 || interleaved_host || out.nframes <= 0) return;
    if (out.is_device && out.nch > 1 && out.chan_stride < out.nframes) return;
    // stage buffers loaded by each root piece
    auto loads_of = [&](int e, std::vector<int>& ls) {
        std::vector<int> stk{e};
        while (!stk.empty()) {
            const int x = stk.back();
            stk.pop_back();
            if (x < 0) continue;
            const Expr& ex = exprs[x];
            if (ex.op == E_LOAD && ex.leaf.buf >= 0) ls.push_back(x);
            if (ex.op >= E_ADD) {
                stk.push_back(ex.a);
                if (ex.op <= E_DIV) stk.push_back(ex.b);
            }
        }
    };
    struct Use { std::vector<size_t> pieces; bool bad = false; int64_t off = 0; };
    std::map<int, Use> uses;  // buffer -> root pieces
    for (size_t pi = 0; pi < rootp.size(); ++pi) {
        const Piece& p = rootp[pi];
        if (p.r.a >= p.r.b || p.r.c0 >= p.r.c1) continue;
        std::vector<int> ls;
        loads_of(p.e, ls);
        std::set<int> bufs_here;
        for (int x : ls) bufs_here.insert(exprs[x].leaf.buf);
        for (int x : ls) {
            const DLeaf& L = exprs[x].leaf;
            Use& u = uses[L.buf];
            const bool plain = L.mode == LM_PLAIN && L.sf == 1 && L.sc == 1 && L.dc == 0 && L.fstride == 1 && L.cstride == -1 &&
                               L.dtype == out.dtype && p.r.c0 == 0 && p.r.c1 == out.nch && bufs_here.size() == 1 && ls.size() == 1;
            if (!plain) u.bad = true;
            if (u.pieces.empty()) u.off = -L.df;
            else if (u.off != -L.df) u.bad = true;
            u.pieces.push_back(pi);
        }
    }
    std::vector<char> drop(rootp.size(), 0);
    bool any = false;
    for (auto& kv : uses) {
        const int b = kv.first;
        Use& u = kv.second;
        if (u.bad || u.off < 0) continue;
        int sid = -1;
        for (size_t i = 0; i < stages.size(); ++i)
            if (stages[i].out_buf == b) sid = (int)i;
        if (sid < 0) continue;
        Stage& S = stages[sid];
        const Node& N = nodes[S.node];
        if (S.kind == ST_NORM || S.need <= 0 || N.dtype != out.dtype || N.nch != out.nch || bufs[b].dtype != out.dtype) continue;
        // every frame of the stage exactly once, in a window that starts at frame 0 of the stage
        std::vector<std::pair<int64_t, int64_t>> iv;
        for (size_t pi : u.pieces) iv.emplace_back(rootp[pi].r.a - u.off, rootp[pi].r.b - u.off);
        std::sort(iv.begin(), iv.end());
        int64_t at = 0;
        bool cover = true;
        for (auto& x : iv) {
            if (x.first != at) cover = false;
            at = x.second;
        }
        if (!cover || at != S.need || u.off + S.need > out.nframes) continue;
        // no other reader of the buffer
        bool other = false;
        for (auto& L : leaves)
            if (L.buf == b) other = true;
        for (auto& S2 : stages) {
            if (S2.in_buf == b) other = true;
            for (auto& c : S2.carriers)
                if (c.buf == b) other = true;
        }
        if (other) continue;
        // commit
        if (out_alias_buf < 0) {
            Buf ob;
            ob.frames = out.nframes;
            ob.nch = out.nch;
            ob.dtype = out.dtype;
            ob.pitch = out.is_device ? (out.nch == 1 ? std::max<int64_t>(out.chan_stride, out.nframes) : out.chan_stride) : out.nframes;
            ob.bytes = 0;
            ob.external = true;
            bufs.push_back(ob);
            out_alias_buf = (int)bufs.size() - 1;
        }
        S.win_off = u.off;
        any = true;
        for (size_t pi : u.pieces) {
            Expr& top = exprs[rootp[pi].e];
            if (top.op == E_LOAD) {
                drop[pi] = 1;  // the stage has written these frames
                continue;
            }
            std::vector<int> ls;
            loads_of(rootp[pi].e, ls);
            for (int x : ls) {  // in place on the result
                exprs[x].leaf.buf = out_alias_buf;
                exprs[x].leaf.df = 0;
            }
        }
    }
    if (!any) return;
    std::vector<Piece> keep;
    for (size_t pi = 0; pi < rootp.size(); ++pi)
        if (!drop[pi]) keep.push_back(rootp[pi]);
    rootp.swap(keep);
    if (std::getenv("SIGOPS_DEBUG_PLAN")) {
        int n = 0;
        for (auto& S : stages) n += S.win_off >= 0;
        std::fprintf(stderr, "[sigops] window aliasing: %d stage(s) write the result themselves, %zu root piece(s) left\n", n, rootp.size());
    }
}


// ===========================================================================
Plan* plan_create(const so_node_t* nodes, int32_t n_nodes, int32_t root, const so_out_desc_t* out,
                  int32_t device, int& status, std::string& err) {
    std::unique_ptr<Plan> P(new Plan());
    const auto t_create0 = std::chrono::steady_clock::now();
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    struct Restore {
        int d;
        ~Restore() {
            if (d >= 0) (void)hipSetDevice(d);
        }
    } restore{prev_device};
    try {
        if (!nodes || n_nodes < 1 || root < 0 || root >= n_nodes || !out)
            fail(SO_ERR_INVALID, "so_plan_create: bad arguments");
        if (out->dtype != SO_F32 && out->dtype != SO_F64)
            fail(SO_ERR_UNSUPPORTED, "result eltype must be Float32 or Float64");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
            fail(SO_ERR_NODEVICE, "no HIP device visible: the sink engine has no CPU path");
        if (device < 0 || device >= ndev) fail(SO_ERR_INVALID, "bad device ordinal");
        P->device = device;
        HIPCHECK(hipSetDevice(device));
        const auto t_dev = std::chrono::steady_clock::now();
        P->out = *out;
        P->root = root;
        P->build_nodes(nodes, n_nodes);
        Node& R = P->nodes[root];
        // sink! length check, reference src/sink.jl:161-163
        if (clean(R.len) < out->nframes)
            fail(SO_ERR_LENGTH, "Signal is too short to fill buffer of length " + std::to_string(out->nframes) + ".");
        if (R.nch != out->nch)
            fail(SO_ERR_CHANNELS, "signal has " + std::to_string(R.nch) + " channels, buffer has " + std::to_string(out->nch) + " (the host applies ToChannels, reference src/sink.jl:164)");
        if (R.short_skip)  // the sink asks the root for a block even when the result is empty (src/sink.jl:225-226)
            fail(SO_ERR_LENGTH, "Signal is too short to skip " + std::to_string(R.short_skip) + " frames");
        std::vector<Piece> rootp;
        if (out->nframes > 0) rootp = P->lower(root, Rect{0, out->nframes, 0, out->nch}, Map{1, 0, 1, 0});
        const auto t_low = std::chrono::steady_clock::now();
        // stages: largest node index first (all users of a stage have larger indices)
        for (;;) {
            int best = -1;
            for (size_t i = 0; i < P->stages.size(); ++i)
                if (!P->stages[i].processed && (best < 0 || P->stages[i].node > P->stages[best].node))
                    best = (int)i;
            if (best < 0) break;
            P->process_stage(best);
        }
        // If the root is nothing but a full plain read of one stage's output, let that
        // stage's kernel write the sink buffer itself (saves a read+write pass).
        // (a host result in interleaved layout -- WAV frames -- is produced by the root K1 step
        //  writing the staging buffer with the result's own strides: one D2H copy, no host loop)
        P->interleaved_host = !out->is_device && out->nch > 1 && out->chan_stride == 1 && out->frame_stride == out->nch;
        if (rootp.size() == 1 && (out->frame_stride == 1 || !out->is_device) && !P->interleaved_host) {
            int re = rootp[0].e;
            // a Float64 signal stored into a Float32 result (`convert(Float32, ·)` on write, reference
            // src/sink.jl:262-266): the periodic resampler can round in its own store
            if (out->dtype == SO_F32)
                while (P->exprs[re].op == E_RETYPE || P->exprs[re].op == E_ROUND32) re = P->exprs[re].a;
            const Expr& e = P->exprs[re];
            const bool narrowing = e.leaf.dtype == SO_F64 && out->dtype == SO_F32 && !std::getenv("SIGOPS_NO_NARROW_STORE");
            if (e.op == E_LOAD && e.leaf.buf >= 0 && e.leaf.mode == LM_PLAIN && e.leaf.sf == 1 &&
                e.leaf.df >= 0 && e.leaf.sc == 1 && e.leaf.dc == 0 && (e.leaf.dtype == out->dtype || narrowing)) {
                for (size_t i = 0; i < P->stages.size(); ++i)
                    if (P->stages[i].out_buf == e.leaf.buf && P->stages[i].kind != ST_NORM &&
                        P->stages[i].need == out->nframes + e.leaf.df &&
                        // (a window of the stage: the three-pass IIR can leave out the frames before it)
                        (e.leaf.df == 0 || (P->stages[i].kind == ST_SOS && P->stages[i].groups.size() == 1 &&
                                            !P->stages[i].onepass && out->frame_stride == 1 &&
                                            e.leaf.df >= P->stages[i].base && !std::getenv("SIGOPS_NO_WINDOW_ALIAS"))) &&
                        (e.leaf.dtype == out->dtype || (P->stages[i].kind == ST_RESAMPLE && P->stages[i].periodic &&
                                                        P->stages[i].rp.ct >= 4 && P->stages[i].rp.rows == 32 && !P->stages[i].rp.arr2 &&
                                                        (P->stages[i].rp.ngroups + P->stages[i].rp.ncompute - 1) / P->stages[i].rp.ncompute == 1) ||
                         // (one group: later groups of a cascade filter the result buffer in place)
                         (P->stages[i].kind == ST_SOS && P->stages[i].groups.size() == 1 && !P->stages[i].onepass &&
                          out->frame_stride == 1)))
                        P->alias_stage = (int)i;
                P->alias_narrow = P->alias_stage >= 0 && e.leaf.dtype != out->dtype;
                if (P->alias_stage >= 0) P->alias_skip = e.leaf.df - P->stages[P->alias_stage].base;
                if (P->alias_stage >= 0)
                    for (auto& L : P->leaves)  // any other consumer of that buffer forbids aliasing
                        if (L.buf == e.leaf.buf) P->alias_stage = -1;
                if (P->alias_stage >= 0)
                    for (auto& S : P->stages)
                        if (S.in_buf == e.leaf.buf) P->alias_stage = -1;
            }
        }
        const auto t_stg = std::chrono::steady_clock::now();
        int rootstep = -1;
        if (P->alias_stage < 0) P->try_window_alias(rootp);
        if (P->alias_stage < 0) rootstep = P->emit_pointwise(rootp, -1, out->dtype);
        P->fuse_state_passes();
        P->fuse_resample_sos();
        P->fuse_plain_sos();
        P->batch_sos_stages();
        const auto t_lowered = std::chrono::steady_clock::now();
        P->finalize();
        const auto t_final = std::chrono::steady_clock::now();
        if (rootstep >= 0) P->push_pw_step(rootstep);
        P->plan_lanes();
        if (std::getenv("SIGOPS_DEBUG_PLAN")) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "[sigops] plan_create: lowering and stage setup %.3f ms (device %.3f, tree %.3f, stages %.3f, fusion %.3f), allocation and uploads %.3f ms, lanes %.3f ms\n",
                         ms(t_create0, t_lowered), ms(t_create0, t_dev), ms(t_dev, t_low), ms(t_low, t_stg), ms(t_stg, t_lowered), ms(t_lowered, t_final),
                         ms(t_final, std::chrono::steady_clock::now()));
        }
    } catch (const PlanError& e) {
        status = e.status;
        err = e.msg;
        P->release();
        return nullptr;
    }
    status = SO_OK;
    return P.release();
}


}  // namespace so
