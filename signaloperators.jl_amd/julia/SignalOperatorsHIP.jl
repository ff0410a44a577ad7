# SignalOperatorsHIP.jl — reference-side glue for libsigops (include/sigops.h).
#
# NOTE: Julia is not installed in the build image or on the GPU box (SURVEY.md probe
# table), so this file has never been executed.  It is the binding a maintainer of
# haberdashPI/SignalOperators.jl would add (see INTEGRATION.md): a sink type `HIPSink`
# whose `sink!` method lowers the lazy operator tree to the C-ABI node table instead of
# pulling blocks (reference src/sink.jl:225-241).  Tree construction (Signal / Until /
# Ramp / Filt / Mix / ToFramerate ...), length algebra and the ToFramerate rewrite rules
# stay exactly as they are in the reference; only the block loop is replaced.
#
# The tested host-side mirror of this glue is the Python package next to this file
# (lowering.py builds the identical node table).
module SignalOperatorsHIP

using SignalOperators
using SignalOperators: MapSignal, FilteredSignal, CutApply, PaddedSignal, AppendSignals,
    RampSignal, NormedSignal, SignalFunction, NumberSignal, ResamplerFn, RawFilterFn,
    FnBr, GetChanFn, As1Channel, AsNChannels, ToEltypeFn, tuplecat, sinramp,
    resolvelen, nframes_helper, IsSignal, SignalTrait, child
using DSP

const libsigops = get(ENV, "LIBSIGOPS", "libsigops.so")

# ---- mirror of so_node_t / so_out_desc_t (include/sigops.h) -----------------------
struct SoNode
    kind::Int32; dtype::Int32; nch::Int32; n_children::Int32
    children::Ptr{Int32}
    nframes::Int64
    fs::Float64
    i0::Int32; i1::Int32; i2::Int32; i3::Int32
    l0::Int64; l1::Int64
    d0::Float64; d1::Float64; d2::Float64; d3::Float64
    p0::Ptr{Cvoid}; p1::Ptr{Cvoid}
    s0::Int64; s1::Int64
end
struct SoOutDesc
    dtype::Int32; nch::Int32; nframes::Int64
    frame_stride::Int64; chan_stride::Int64
    is_device::Int32; reserved::Int32
end

const SO_F32, SO_F64, SO_I64 = Int32(0), Int32(1), Int32(2)
const LEN_INF, LEN_MISSING = Int64(-1), Int64(-2)
@enum Kind::Int32 ARRAY=0 CONST=1 FUNC=2 UNTIL=3 AFTER=4 PAD=5 APPEND=6 RAMP=7 MAP=8 FILT_SOS=9 RESAMPLE=10 NORMPOWER=11

sodtype(::Type{Float32}) = SO_F32
sodtype(::Type{Float64}) = SO_F64
sodtype(::Type{<:Integer}) = SO_I64
sodtype(T) = error("HIPSink lowers Float32/Float64 signals only; use the stock sink for $T")
solen(n::Number) = Int64(n)
solen(::Missing) = LEN_MISSING
solen(n) = isinf(n) ? LEN_INF : error("unexpected length $n")
sofs(fs) = ismissing(fs) ? NaN : Float64(fs)

"""
    HIPSink{T}

Sink type: `sink(x, HIPSink)` evaluates `x` on the MI355X engine and returns a
`(Array, framerate)` tuple like the stock `Tuple` sink.
"""
struct HIPSink{T} <: AbstractMatrix{T}
    data::Matrix{T}
end
Base.size(x::HIPSink) = size(x.data)
Base.getindex(x::HIPSink, i...) = getindex(x.data, i...)
SignalOperators.initsink(x, ::Type{<:HIPSink}) =
    HIPSink(Array{sampletype(x)}(undef, nframes(x), nchannels(x)))

mutable struct Lowering
    nodes::Vector{SoNode}
    keep::Vector{Any}          # GC roots for every pointer handed to C
    memo::IdDict{Any,Int32}
    need::IdDict{Any,Tuple{Int,Int}}   # randn leaves: (frames reached, leading frames skipped)
end
Lowering() = Lowering(SoNode[], Any[], IdDict{Any,Int32}(), IdDict{Any,Tuple{Int,Int}}())

# How many frames of every `randn` leaf a sink of n frames reaches, and how many of the leading
# ones an `After` SKIPS rather than evaluates (src/cutting.jl:160-181: skipped blocks never reach
# `frame`, so they draw nothing; `Filt` ignores the flag, src/filters.jl:241-244).  Mirrors
# lowering.py `_demand` of the tested Python host.
finite_min(x, m) = (n = nframes(x); (ismissing(n) || isinf(n)) ? m : min(m, Int(n)))
demand!(need, x, n, skip=0) = nothing
function demand!(need, x::SignalFunction{<:SignalOperators.RandFn}, n, skip=0)
    n0, s0 = get(need, x, (0, skip))
    need[x] = (max(n0, n), min(s0, skip))
end
function demand!(need, x::CutApply{<:Any,<:Any,K}, n, skip=0) where K
    L = max(0, resolvelen(x))
    K <: Val{:Until} ? demand!(need, child(x), finite_min(child(x), min(n, L)), skip) :
                       demand!(need, child(x), finite_min(child(x), n + L), skip + L)
end
# sub-trees the engine cannot lower are sunk by the stock CPU sink (stock_leaf! below): record how many
# of their frames the GPU sink reaches and stop there
record!(need, x, n, skip) = (need[x] = (max(n, get(need, x, (0, skip))[1]), min(skip, get(need, x, (0, skip))[2])); nothing)
demand!(need, x::SignalFunction, n, skip=0) = lowerable_fn(x.fn) ? nothing : record!(need, x, n, skip)
function demand!(need, x::Union{PaddedSignal,RampSignal}, n, skip=0)
    opaque = x isa RampSignal ? !(x.fn === sinramp || x.fn === identity) :
        !(x.Pad === zero || x.Pad === one || x.Pad === lastframe || x.Pad === cycle || x.Pad === mirror ||
          x.Pad isa Number || x.Pad isa Union{Tuple,AbstractVector})
    opaque ? record!(need, x, n, skip) : demand!(need, child(x), finite_min(child(x), n), skip)
end
function demand!(need, x::MapSignal, n, skip=0)
    fn = x.fn isa FnBr ? x.fn.fn : x.fn
    (lowerable_map(fn) && (x.padding === one || x.padding === zero)) || return record!(need, x, n, skip)
    foreach(s -> demand!(need, s, finite_min(s, n), skip), x.signals)
end
function demand!(need, x::AppendSignals, n, skip=0)
    rem, sk = n, skip
    for s in x.signals
        m = finite_min(s, rem)
        demand!(need, s, m, min(sk, m))
        rem -= m; sk = max(0, sk - m)
        rem <= 0 && break
    end
end
function demand!(need, x::FilteredSignal, n, skip=0)
    m = n
    if x.fn isa ResamplerFn   # newest input of the last output + the filter's group delay
        h = DSP.resample_filter(x.fn.ratio)
        nphi = x.fn.ratio isa Rational ? numerator(x.fn.ratio) : 32
        m = ceil(Int, max(n - 1, 0) / x.fn.ratio) + ceil(Int, (length(h) - 1) / (2nphi)) + 2
    end
    demand!(need, child(x), finite_min(child(x), m), 0)
end
demand!(need, x::NormedSignal, n, skip=0) = demand!(need, child(x), Int(nframes(child(x))), 0)

function push_node!(lw, x, kind; kids=Int32[], i0=0, i1=0, i2=0, i3=0, l0=0, l1=0,
                    d0=0.0, d1=0.0, p0=C_NULL, p1=C_NULL, s0=0, s1=0,
                    dtype=sodtype(sampletype(x)), nch=nchannels(x))
    push!(lw.keep, kids)
    node = SoNode(Int32(kind), dtype, Int32(nch), Int32(length(kids)),
                  isempty(kids) ? C_NULL : pointer(kids), solen(nframes(x)), sofs(framerate(x)),
                  i0, i1, i2, i3, l0, l1, d0, d1, 0.0, 0.0, p0, p1, s0, s1)
    push!(lw.nodes, node)
    Int32(length(lw.nodes) - 1)
end

lower!(lw, x) = get!(() -> lower_node!(lw, x), lw.memo, x)

function lower_node!(lw, x::Union{AbstractArray,Tuple{<:AbstractArray,<:Number}})
    data = x isa Tuple ? x[1] : x
    push!(lw.keep, data)
    push_node!(lw, x, ARRAY; p0=pointer(data), l0=size(data, 1), i0=0,
               s0=stride(data, 1), s1=ndims(data) > 1 ? stride(data, 2) : 0)
end
lower_node!(lw, x::NumberSignal) =
    push_node!(lw, x, CONST; d0=Float64(x.val), i0=sodtype(typeof(x.val)), nch=1)
# `Signal(randn)` (src/functions.jl:98-114): host-materialised -- one draw per EVALUATED frame in
# increasing frame order from the leaf's own generator, zeros for the frames an `After` skips --
# then an ARRAY node under an (unreachable) zero Pad, so that the leaf stays infinite for the planner.
function lower_node!(lw, x::SignalFunction{<:SignalOperators.RandFn})
    n, skipped = get(lw.need, x, (0, 0))
    data = zeros(Float64, max(n, 0), 1)
    for i in skipped+1:n
        data[i, 1] = randn(x.fn.rng)
    end
    push!(lw.keep, data)
    push!(lw.nodes, SoNode(Int32(ARRAY), SO_F64, Int32(1), Int32(0), C_NULL, Int64(-3) #= SO_LEN_UNCHECKED =#,
                           sofs(framerate(x)), 0, 0, 0, 0, size(data, 1), 0, 0.0, 0.0, 0.0, 0.0, pointer(data), C_NULL,
                           1, size(data, 1)))
    inner = Int32(length(lw.nodes) - 1)
    kids = Int32[inner]
    push!(lw.keep, kids)
    push!(lw.nodes, SoNode(Int32(PAD), SO_F64, Int32(1), Int32(1), pointer(kids), LEN_INF, sofs(framerate(x)),
                           2 #= zero =#, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, C_NULL, C_NULL, 0, 0))
    Int32(length(lw.nodes) - 1)
end
# SURVEY.md section 8(b): a sub-tree the engine cannot lower (opaque closures) is materialised by the
# STOCK CPU sink -- `sink(sub, Array)`, reference src/sink.jl:64-92 -- for the frames the GPU sink
# reaches, and enters the node table as an ARRAY leaf under an (unreachable) zero Pad and, for a
# finite sub-tree, an Until of its own length, so that the length algebra above it is unchanged.
function stock_leaf!(lw, x)
    n, _ = get(lw.need, x, (isinf(nframes(x)) ? 0 : Int(nframes(x)), 0))
    data = n > 0 ? SignalOperators.sink(x |> Until(n * frames), Array) : zeros(sampletype(x), 0, nchannels(x))
    data = convert(Matrix{float(sampletype(x))}, reshape(data, size(data, 1), :))
    push!(lw.keep, data)
    dt = sodtype(eltype(data))
    push!(lw.nodes, SoNode(Int32(ARRAY), dt, Int32(size(data, 2)), Int32(0), C_NULL, Int64(-3) #= SO_LEN_UNCHECKED =#,
                           sofs(framerate(x)), 0, 0, 0, 0, size(data, 1), 0, 0.0, 0.0, 0.0, 0.0, pointer(data), C_NULL,
                           1, max(size(data, 1), 1)))
    kids = Int32[Int32(length(lw.nodes) - 1)]
    push!(lw.keep, kids)
    push!(lw.nodes, SoNode(Int32(PAD), dt, Int32(size(data, 2)), Int32(1), pointer(kids), LEN_INF, sofs(framerate(x)),
                           2 #= zero =#, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, C_NULL, C_NULL, 0, 0))
    idx = Int32(length(lw.nodes) - 1)
    if !isinf(nframes(x))
        kids2 = Int32[idx]
        push!(lw.keep, kids2)
        push!(lw.nodes, SoNode(Int32(UNTIL), dt, Int32(size(data, 2)), Int32(1), pointer(kids2), Int64(nframes(x)),
                               sofs(framerate(x)), 0, 0, 0, 0, Int64(nframes(x)), 0, 0.0, 0.0, 0.0, 0.0, C_NULL, C_NULL, 0, 0))
        idx = Int32(length(lw.nodes) - 1)
    end
    idx
end
# (demand analysis stops at such a sub-tree: it is sunk on its own)
lowerable_fn(fn) = fn === sin || fn === cos || fn === identity
lowerable_map(fn) = fn === (+) || fn === (*) || fn === (-) || fn === (/) || fn === tuplecat || fn isa GetChanFn ||
    fn isa As1Channel || fn isa AsNChannels || fn isa ToEltypeFn || fn === reverse

function lower_node!(lw, x::SignalFunction)
    lowerable_fn(x.fn) || return stock_leaf!(lw, x)
    code = x.fn === sin ? 0 : x.fn === cos ? 1 : 2
    push_node!(lw, x, FUNC; i0=code, i1=ismissing(x.ω) ? 0 : 1,
               d0=ismissing(x.ω) ? 0.0 : Float64(x.ω), d1=x.ϕ, nch=1, dtype=SO_F64)
end
function lower_node!(lw, x::CutApply{<:Any,<:Any,K}) where K
    c = lower!(lw, child(x))
    push_node!(lw, x, K <: Val{:Until} ? UNTIL : AFTER; kids=Int32[c], l0=resolvelen(x))
end
function lower_node!(lw, x::PaddedSignal{<:Any,<:Any,E}) where E
    c = lower!(lw, child(x))
    p = x.Pad
    kind, val, vec = p === zero ? (2, 0.0, C_NULL) : p === one ? (3, 0.0, C_NULL) :
        p === lastframe ? (4, 0.0, C_NULL) : p === cycle ? (5, 0.0, C_NULL) :
        p === mirror ? (6, 0.0, C_NULL) : p isa Number ? (0, Float64(p), C_NULL) :
        p isa Union{Tuple,AbstractVector} ? (1, 0.0, pointer(push!(lw.keep, Float64.(collect(p)))[end])) :
        return stock_leaf!(lw, x)   # an opaque padding closure (src/padding.jl:150-192): the padded signal as a whole
    push_node!(lw, x, PAD; kids=Int32[c], i0=kind, i1=E ? 1 : 0, d0=val, p0=vec)
end
lower_node!(lw, x::AppendSignals) =
    push_node!(lw, x, APPEND; kids=Int32[lower!(lw, s) for s in x.signals])
function lower_node!(lw, x::RampSignal{D}) where D
    (x.fn === sinramp || x.fn === identity) || return stock_leaf!(lw, x)   # a custom ramp function (src/ramps.jl:26)
    fn = x.fn === sinramp ? 0 : 1
    push_node!(lw, x, RAMP; kids=Int32[lower!(lw, child(x))], i0=D === :on ? 0 : 1, i1=fn,
               l0=resolvelen(x))
end
function lower_node!(lw, x::MapSignal)
    fn = x.fn isa FnBr ? x.fn.fn : x.fn
    (lowerable_map(fn) && (x.padding === one || x.padding === zero)) || return stock_leaf!(lw, x)   # src/mapsignal.jl:131-145
    code, extra = fn === (+) ? (0, 0) : fn === (*) ? (1, 0) : fn === (-) ? (2, 0) : fn === (/) ? (3, 0) :
        fn === tuplecat ? (4, 0) : fn isa GetChanFn ? (5, fn.n) : fn isa As1Channel ? (6, 0) :
        fn isa AsNChannels ? (7, fn.ch) : fn isa ToEltypeFn ? (8, sodtype(typeof(fn).parameters[1])) :
        (9, 0)
    pad = x.padding === one ? 3 : 2
    push_node!(lw, x, MAP; kids=Int32[lower!(lw, s) for s in x.signals], i0=code,
               i1=x.bychannel ? 1 : 0, i2=pad, i3=extra)
end
function lower_node!(lw, x::FilteredSignal)
    c = lower!(lw, child(x))
    if x.fn isa ResamplerFn                         # reference src/reformatting.jl:92-98
        h = Float64.(DSP.resample_filter(x.fn.ratio))
        push!(lw.keep, h)
        if x.fn.ratio isa Rational
            push_node!(lw, x, RESAMPLE; kids=Int32[c], i0=0, i1=numerator(x.fn.ratio),
                       l0=numerator(x.fn.ratio), l1=denominator(x.fn.ratio), p0=pointer(h),
                       i2=length(h), i3=x.blocksize)
        else
            push_node!(lw, x, RESAMPLE; kids=Int32[c], i0=1, i1=32, d0=Float64(x.fn.ratio),
                       p0=pointer(h), i2=length(h), i3=x.blocksize)
        end
    elseif (hobj = x.fn(framerate(x))) isa PolynomialRatio && length(coefa(hobj)) == 1
        # Filt(x,h) with FIR coefficients (RawFilterFn, src/filters.jl:89-97): RESAMPLE kind SO_RS_FIR
        h = Float64.(coefb(hobj) ./ coefa(hobj)[1])
        push!(lw.keep, h)
        push_node!(lw, x, RESAMPLE; kids=Int32[c], i0=2, i1=1, l0=1, l1=1, p0=pointer(h), i2=length(h), i3=x.blocksize)
    elseif hobj isa PolynomialRatio
        # DF2TFilter(::PolynomialRatio) is DSP.jl's direct form; the engine runs second-order sections.  The library
        # factors the polynomials and reports how far the two impulse responses are apart (ill-conditioned ones: refuse)
        b, a = Float64.(coefb(hobj)), Float64.(coefa(hobj))
        sos = zeros(Float64, 6 * (max(length(a), length(b)) ÷ 2 + 2))
        nsec, gain, resid = Ref{Int32}(0), Ref{Float64}(0), Ref{Float64}(0)
        check(ccall((:so_tf_to_sos, libsigops), Int32,
                    (Ptr{Float64}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Ref{Int32}, Ref{Float64}, Ref{Float64}),
                    b, length(b), a, length(a), sos, length(sos), nsec, gain, resid))
        resid[] <= 1e-9 || error("PolynomialRatio: too ill-conditioned to factor into second-order sections " *
                                 "(impulse responses differ by $(resid[])); pass SecondOrderSections or ZeroPoleGain")
        push!(lw.keep, sos)
        push_node!(lw, x, FILT_SOS; kids=Int32[c], i0=nsec[], p0=pointer(sos), d0=gain[], i1=x.blocksize)
    else                                            # reference src/filters.jl:10-11,94
        f = convert(SecondOrderSections, x.fn(framerate(x)))
        sos = Float64[c for b in f.biquads for c in (b.b0, b.b1, b.b2, 1.0, b.a1, b.a2)]
        push!(lw.keep, sos)
        push_node!(lw, x, FILT_SOS; kids=Int32[c], i0=length(f.biquads), p0=pointer(sos),
                   d0=Float64(f.g), i1=x.blocksize)
    end
end
lower_node!(lw, x::NormedSignal) = push_node!(lw, x, NORMPOWER; kids=Int32[lower!(lw, child(x))])

check(st) = st == 0 || error(unsafe_string(ccall((:so_last_error, libsigops), Cstring, ())))

# Large map nests are specialised by hipRTC off the caller's path (first sink of a shape: interpreter kernel; later
# sinks, and later sessions where ENV["SIGOPS_CACHE_DIR"] names a directory: the compiled one, same values).  A long-running service
# can wait for the queue once it has seen its workload:
warmup_done() = check(ccall((:so_rtc_wait_idle, libsigops), Int32, ()))
# (the compile thread ends before the session tears the GPU runtime down)
atexit(() -> ccall((:so_rtc_shutdown, libsigops), Int32, ()))

# The method the engine plugs into: reference src/sink.jl:225-226 dispatch point.
function SignalOperators.sink!(result::HIPSink{T}, x, ::IsSignal) where T
    lw = Lowering()
    demand!(lw.need, x, size(result, 1))
    root = lower!(lw, x)
    desc = Ref(SoOutDesc(sodtype(T), size(result, 2), size(result, 1), 1, size(result, 1), 0, 0))
    plan = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve lw result begin
        check(ccall((:so_plan_create, libsigops), Int32,
                    (Ptr{SoNode}, Int32, Int32, Ref{SoOutDesc}, Int32, Ref{Ptr{Cvoid}}),
                    lw.nodes, length(lw.nodes), root, desc, 0, plan))
        try
            check(ccall((:so_plan_execute, libsigops), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                        plan[], result.data, C_NULL))
        finally
            ccall((:so_plan_destroy, libsigops), Cvoid, (Ptr{Cvoid},), plan[])
        end
    end
    nothing
end

# Block-by-block evaluation (the Python mirror's `so.stream`): block k is the sink of
# `x |> After(k*blocksize frames) |> Until(blocksize frames)`.  Stateful stages of a later block start a
# decay time (IIR) / a few periods (resampler) before it instead of at frame 0 -- the planner's warm
# start, DESIGN.md section 2 -- so the work per block does not grow with its position.
function sink_blocks(f, x, blocksize::Integer; T=Float64)
    n, pos = nframes(x), 0
    while isinf(n) || pos < n
        m = isinf(n) ? blocksize : min(blocksize, n - pos)
        blk = Until(pos == 0 ? x : After(x, pos * frames), m * frames)
        result = HIPSink{T}(Array{T}(undef, m, nchannels(x)))
        SignalOperators.sink!(result, blk, SignalOperators.SignalTrait(blk))
        f(result.data)
        pos += m
    end
end

# ---- the exchange steps of a sharded sink (include/sigops.h so_comm_*; one process per GPU, RCCL bound at run time) ----
struct SoSlab
    rows::Int64; row_elems::Int64
    dst_offset::Int64; dst_row_stride::Int64
end

"rank 0 makes the 128-byte id; the host's launcher (MPI.jl, a file, a socket) hands it to every rank"
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:so_comm_unique_id, libsigops), Int32, (Ptr{UInt8},), id))
    id
end

function comm_create(id::Vector{UInt8}, world::Integer, rank::Integer, device::Integer=0)
    comm = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:so_comm_create, libsigops), Int32, (Ptr{UInt8}, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
                id, world, rank, device, comm))
    comm[]
end

"every rank's share (`mine`: device pointer, rows `src_row_stride` elements apart) into every rank's `full` buffer"
function comm_allgather!(comm::Ptr{Cvoid}, mine::Ptr{Cvoid}, src_row_stride::Integer, full::Ptr{Cvoid},
                         slabs::Vector{SoSlab}, dtype::Integer=SO_F64, stream::Ptr{Cvoid}=C_NULL)
    check(ccall((:so_comm_allgather, libsigops), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{SoSlab}, Int32, Ptr{Cvoid}),
                comm, mine, src_row_stride, full, slabs, dtype, stream))
end

"the operands of a root `Mix` evaluated on different ranks: the ranks' partial sums added up in place (root < 0: on every rank)"
function comm_reduce_sum!(comm::Ptr{Cvoid}, buf::Ptr{Cvoid}, rows::Integer, row_elems::Integer, row_stride::Integer,
                          dtype::Integer=SO_F64, root::Integer=-1, stream::Ptr{Cvoid}=C_NULL)
    check(ccall((:so_comm_reduce_sum, libsigops), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Int32, Int32, Ptr{Cvoid}),
                comm, buf, rows, row_elems, row_stride, dtype, root, stream))
end

comm_destroy(comm::Ptr{Cvoid}) = ccall((:so_comm_destroy, libsigops), Cvoid, (Ptr{Cvoid},), comm)

end # module
