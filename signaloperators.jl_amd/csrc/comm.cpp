// The one exchange step of the sharded sink: every rank has evaluated its share
// of the result -- the frames of its Append children or of its time range, or its slab of channels --
// and the shares are gathered into the full planar buffer on every rank.  Shares are uneven, so this is
// a grouped ncclSend / ncclRecv exchange over RCCL (xGMI inside a node), one message per contiguous run.
// Behind the C-ABI so that the Julia host named by BASELINE.json's north_star has the collective without
// PyTorch; the Python harness can use it too (sharding.py, native=True) or torch.distributed.
//
// RCCL is bound at run time (dlopen): libsigops itself stays loadable where RCCL is not installed, and
// inside a PyTorch process the copy of RCCL that process has already loaded is used (two HIP runtimes
// in one process do not work).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/sigops.h"

namespace {

struct UniqueId {  // ncclUniqueId
    char internal[128];
};

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;  // (the id is passed BY VALUE: 128 bytes)
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // a copy this process already has
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) {
            r.err = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed");
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(r.h, s);
            if (!p && r.err.empty()) r.err = std::string("RCCL symbol missing: ") + s;
            return p;
        };
        r.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
        r.CommInitRank = (int (*)(void**, int, UniqueId, int))sym("ncclCommInitRank");
        r.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        r.GroupStart = (int (*)())sym("ncclGroupStart");
        r.GroupEnd = (int (*)())sym("ncclGroupEnd");
        r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
        r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
        r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclAllReduce");
        r.Reduce = (int (*)(const void*, void*, size_t, int, int, int, void*, hipStream_t))sym("ncclReduce");
        r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    });
    return r;
}

thread_local std::string g_comm_err;
int fail(int st, const std::string& m) {
    g_comm_err = m;
    return st;
}

}  // namespace

struct so_comm {
    void* comm = nullptr;
    int world = 1, rank = 0, device = 0;
};

extern "C" {

const char* so_comm_last_error(void) { return g_comm_err.c_str(); }

int32_t so_comm_unique_id(void* id128) {
    if (!id128) return fail(SO_ERR_INVALID, "so_comm_unique_id: null id");
    Rccl& r = rccl();
    if (!r.err.empty()) return fail(SO_ERR_UNSUPPORTED, r.err);
    const int st = r.GetUniqueId(id128);
    if (st != 0) return fail(SO_ERR_RUNTIME, std::string("ncclGetUniqueId: ") + r.GetErrorString(st));
    return SO_OK;
}

int32_t so_comm_create(const void* id128, int32_t world, int32_t rank, int32_t device, so_comm_t** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return fail(SO_ERR_INVALID, "so_comm_create: bad arguments");
    Rccl& r = rccl();
    if (!r.err.empty()) return fail(SO_ERR_UNSUPPORTED, r.err);
    if (hipSetDevice(device) != hipSuccess) return fail(SO_ERR_NODEVICE, "so_comm_create: no such HIP device");
    UniqueId id;
    std::memcpy(id.internal, id128, 128);
    so_comm* c = new so_comm;
    c->world = world;
    c->rank = rank;
    c->device = device;
    const int st = r.CommInitRank(&c->comm, world, id, rank);
    if (st != 0) {
        delete c;
        return fail(SO_ERR_RUNTIME, std::string("ncclCommInitRank: ") + r.GetErrorString(st));
    }
    *out = c;
    return SO_OK;
}

void so_comm_destroy(so_comm_t* c) {
    if (!c) return;
    Rccl& r = rccl();
    if (c->comm && r.CommDestroy) (void)r.CommDestroy(c->comm);
    delete c;
}

int32_t so_comm_allgather(so_comm_t* c, const void* mine, int64_t src_row_stride, void* full, const so_slab_t* slabs,
                          int32_t dtype, void* stream) {
    if (!c || !slabs || !full) return fail(SO_ERR_INVALID, "so_comm_allgather: null argument");
    if (dtype != SO_F32 && dtype != SO_F64) return fail(SO_ERR_INVALID, "so_comm_allgather: Float32 / Float64 only");
    Rccl& r = rccl();
    if (!r.err.empty()) return fail(SO_ERR_UNSUPPORTED, r.err);
    if (hipSetDevice(c->device) != hipSuccess) return fail(SO_ERR_NODEVICE, "so_comm_allgather: device");
    const size_t esz = dtype == SO_F32 ? 4 : 8;
    const int nt = dtype == SO_F32 ? 7 : 8;  // ncclFloat32 / ncclFloat64
    hipStream_t st = (hipStream_t)stream;
    const so_slab_t& me = slabs[c->rank];
    if (me.rows > 0 && me.row_elems > 0 && !mine) return fail(SO_ERR_INVALID, "so_comm_allgather: null slab");
    // this rank's own share (unless the engine already wrote it in place)
    char* own = (char*)full + (size_t)me.dst_offset * esz;
    // SIGOPS_COMM_SELF_EXCHANGE=1 (test aid: a box has one GPU): the own share travels through RCCL too, as
    // a send to and a receive from this very rank, so that the grouped exchange runs on a single device
    const bool self_x = std::getenv("SIGOPS_COMM_SELF_EXCHANGE") != nullptr && (const void*)own != mine;
    if (self_x && me.rows > 0 && me.row_elems > 0) {
        int rc = r.GroupStart();
        for (int64_t row = 0; row < me.rows && rc == 0; ++row) {
            rc = r.Send((const char*)mine + (size_t)(row * src_row_stride) * esz, (size_t)me.row_elems, nt, c->rank, c->comm, st);
            if (rc == 0) rc = r.Recv(own + (size_t)(row * me.dst_row_stride) * esz, (size_t)me.row_elems, nt, c->rank, c->comm, st);
        }
        const int rc2 = r.GroupEnd();
        if (rc != 0 || rc2 != 0) return fail(SO_ERR_RUNTIME, std::string("RCCL self exchange: ") + r.GetErrorString(rc ? rc : rc2));
    } else if (me.rows > 0 && me.row_elems > 0 && (const void*)own != mine) {
        if (hipMemcpy2DAsync(own, (size_t)me.dst_row_stride * esz, mine, (size_t)src_row_stride * esz,
                             (size_t)me.row_elems * esz, (size_t)me.rows, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(SO_ERR_RUNTIME, "so_comm_allgather: local copy failed");
    }
    if (c->world == 1) return SO_OK;
    int rc = r.GroupStart();
    for (int p = 0; p < c->world && rc == 0; ++p) {
        if (p == c->rank) continue;
        for (int64_t row = 0; row < me.rows && me.row_elems > 0 && rc == 0; ++row)
            rc = r.Send((const char*)mine + (size_t)(row * src_row_stride) * esz, (size_t)me.row_elems, nt, p, c->comm, st);
        const so_slab_t& sp = slabs[p];
        for (int64_t row = 0; row < sp.rows && sp.row_elems > 0 && rc == 0; ++row)
            rc = r.Recv((char*)full + (size_t)(sp.dst_offset + row * sp.dst_row_stride) * esz, (size_t)sp.row_elems, nt, p, c->comm, st);
    }
    const int rc2 = r.GroupEnd();
    if (rc != 0 || rc2 != 0) return fail(SO_ERR_RUNTIME, std::string("RCCL exchange: ") + r.GetErrorString(rc ? rc : rc2));
    return SO_OK;
}

// The other exchange the path has (BASELINE.json north_star: "RCCL ... only for the final concatenate/sum"): the
// operands of a root `Mix(xs...) = OperateOn(+, xs...)` (reference src/mapsignal.jl:307-308) evaluated on different
// ranks, every rank's partial sum in a buffer of the result's shape, and ONE grouped reduction adds them up in place --
// into every rank's buffer (root < 0: ncclAllReduce) or into `root`'s (ncclReduce; the others' buffers are then
// unspecified).  `rows` runs of `row_elems` elements, `row_stride` elements apart (a planar result whose channel
// rows are padded; rows = 1 for a contiguous one).  Floating-point note: the reference folds its operands left to
// right; a reduction over ranks associates them as the collective's algorithm does -- the same values for two ranks,
// a different rounding of the same sum (<= 1 ulp per addition) beyond.
int32_t so_comm_reduce_sum(so_comm_t* c, void* buf, int64_t rows, int64_t row_elems, int64_t row_stride, int32_t dtype,
                           int32_t root, void* stream) {
    if (!c || (!buf && rows > 0 && row_elems > 0)) return fail(SO_ERR_INVALID, "so_comm_reduce_sum: null argument");
    if (dtype != SO_F32 && dtype != SO_F64) return fail(SO_ERR_INVALID, "so_comm_reduce_sum: Float32 / Float64 only");
    if (rows < 0 || row_elems < 0 || root >= c->world) return fail(SO_ERR_INVALID, "so_comm_reduce_sum: bad shape or root");
    Rccl& r = rccl();
    if (!r.err.empty()) return fail(SO_ERR_UNSUPPORTED, r.err);
    if (rows == 0 || row_elems == 0) return SO_OK;
    if (hipSetDevice(c->device) != hipSuccess) return fail(SO_ERR_NODEVICE, "so_comm_reduce_sum: device");
    const size_t esz = dtype == SO_F32 ? 4 : 8;
    const int nt = dtype == SO_F32 ? 7 : 8;  // ncclFloat32 / ncclFloat64
    const int sum = 0;                        // ncclSum
    hipStream_t st = (hipStream_t)stream;
    if (row_stride == row_elems) {  // contiguous: one collective
        row_elems *= rows;
        rows = 1;
    }
    int rc = rows > 1 ? r.GroupStart() : 0;
    for (int64_t row = 0; row < rows && rc == 0; ++row) {
        char* p = (char*)buf + (size_t)(row * row_stride) * esz;
        rc = root < 0 ? r.AllReduce(p, p, (size_t)row_elems, nt, sum, c->comm, st) : r.Reduce(p, p, (size_t)row_elems, nt, sum, root, c->comm, st);
    }
    const int rc2 = rows > 1 ? r.GroupEnd() : 0;
    if (rc != 0 || rc2 != 0) return fail(SO_ERR_RUNTIME, std::string("RCCL reduction: ") + r.GetErrorString(rc ? rc : rc2));
    return SO_OK;
}

}  // extern "C"
