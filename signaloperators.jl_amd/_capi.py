"""ctypes binding of libsigops.so (include/sigops.h).  The library is built in-tree
by `csrc/build.py` (hipcc --offload-arch=gfx950); if it is missing every entry point
raises — there is no CPU fallback on the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIGOPS_LIB") or os.path.join(_HERE, "csrc", "libsigops.so")  # SIGOPS_LIB: A/B tuning builds

SO_F32, SO_F64, SO_I64 = 0, 1, 2
SO_LEN_INF, SO_LEN_MISSING, SO_LEN_UNCHECKED = -1, -2, -3
(NODE_ARRAY, NODE_CONST, NODE_FUNC, NODE_UNTIL, NODE_AFTER, NODE_PAD, NODE_APPEND, NODE_RAMP,
 NODE_MAP, NODE_FILT_SOS, NODE_RESAMPLE, NODE_NORMPOWER) = range(12)
FN = {"sin": 0, "cos": 1, "identity": 2}
RAMPFN = {"sinramp": 0, "identity": 1}
MAPFN = {"add": 0, "mul": 1, "sub": 2, "div": 3, "tuplecat": 4, "getchan": 5, "as1channel": 6,
         "asnchannels": 7, "toeltype": 8, "reversech": 9}
PAD = {"value": 0, "vector": 1, "zero": 2, "one": 3, "lastframe": 4, "cycle": 5, "mirror": 6}
RS_RATIONAL, RS_ARBITRARY, RS_FIR = 0, 1, 2
FILT = {"lowpass": 0, "highpass": 1, "bandpass": 2, "bandstop": 3}
METHOD = {"butterworth": 0, "chebyshev1": 1}


class so_node_t(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dtype", C.c_int32), ("nch", C.c_int32),
                ("n_children", C.c_int32), ("children", C.POINTER(C.c_int32)),
                ("nframes", C.c_int64), ("fs", C.c_double),
                ("i0", C.c_int32), ("i1", C.c_int32), ("i2", C.c_int32), ("i3", C.c_int32),
                ("l0", C.c_int64), ("l1", C.c_int64),
                ("d0", C.c_double), ("d1", C.c_double), ("d2", C.c_double), ("d3", C.c_double),
                ("p0", C.c_void_p), ("p1", C.c_void_p), ("s0", C.c_int64), ("s1", C.c_int64)]


class so_out_desc_t(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("nch", C.c_int32), ("nframes", C.c_int64),
                ("frame_stride", C.c_int64), ("chan_stride", C.c_int64),
                ("is_device", C.c_int32), ("reserved", C.c_int32)]


class so_stats_t(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("n_launches", C.c_int32),
                ("algorithmic_bytes", C.c_int64), ("scratch_bytes", C.c_int64),
                ("h2d_bytes", C.c_int64), ("d2h_bytes", C.c_int64),
                ("last_exec_ms", C.c_double), ("dominant_kernel_ms", C.c_double),
                ("dominant_kernel_bytes", C.c_int64), ("dominant_kernel", C.c_char * 64)]


class so_slab_t(C.Structure):
    _fields_ = [("rows", C.c_int64), ("row_elems", C.c_int64), ("dst_offset", C.c_int64), ("dst_row_stride", C.c_int64)]


class so_step_info_t(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("algorithmic_bytes", C.c_int64), ("ms", C.c_double),
                ("launches", C.c_int32), ("pad", C.c_int32)]


EXPORTS = ["so_abi_version", "so_last_error", "so_device_count", "so_plan_create",
           "so_plan_nframes", "so_plan_execute", "so_plan_check", "so_plan_set_array", "so_plan_stats",
           "so_plan_set_profiling", "so_plan_destroy", "so_design_iir",
           "so_design_resample_rational", "so_design_resample_arbitrary",
           "so_resample_positions", "so_plan_step_info", "so_design_iir_zpk", "so_zpk_to_sos", "so_tf_to_sos", "so_tf_zero_input", "so_plan_counter",
           "so_rtc_compile_check", "so_rtc_wait_idle", "so_rtc_shutdown", "so_comm_unique_id", "so_comm_create", "so_comm_allgather", "so_comm_reduce_sum", "so_comm_last_error", "so_comm_destroy"]

_lib = None


class EngineMissing(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineMissing(
            f"{LIB_PATH} not built: run `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). The HIP engine has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 with the same
    # SONAME as /opt/rocm's.  If ours were loaded first, a later `import torch` would bind
    # to it and fail ("No HIP GPUs are available"); so when torch is installed, load its
    # copy first and let libsigops resolve against it.
    # Importing torch AFTER libsigops has initialised HIP was observed to hang inside
    # torch._C's static initialisation on the GPU box, so torch (when installed) goes first.
    try:
        import importlib.util as _ilu

        if _ilu.find_spec("torch") is not None:
            import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    L.so_abi_version.restype = C.c_int32
    L.so_last_error.restype = C.c_char_p
    L.so_device_count.restype = C.c_int32
    L.so_plan_create.restype = C.c_int32
    L.so_plan_create.argtypes = [C.POINTER(so_node_t), C.c_int32, C.c_int32,
                                 C.POINTER(so_out_desc_t), C.c_int32, C.POINTER(C.c_void_p)]
    L.so_plan_nframes.restype = C.c_int64
    L.so_plan_nframes.argtypes = [C.c_void_p]
    L.so_plan_execute.restype = C.c_int32
    L.so_plan_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.so_plan_check.restype = C.c_int32
    L.so_plan_check.argtypes = [C.c_void_p, C.c_void_p]
    L.so_plan_set_array.restype = C.c_int32
    L.so_plan_set_array.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.so_plan_stats.restype = C.c_int32
    L.so_plan_stats.argtypes = [C.c_void_p, C.POINTER(so_stats_t)]
    L.so_plan_set_profiling.restype = C.c_int32
    L.so_plan_set_profiling.argtypes = [C.c_void_p, C.c_int32]
    L.so_plan_destroy.restype = None
    L.so_plan_destroy.argtypes = [C.c_void_p]
    L.so_design_iir.restype = C.c_int32
    L.so_design_iir.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32,
                                C.c_int32, C.c_double, C.POINTER(C.c_double), C.c_int32,
                                C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.so_design_resample_rational.restype = C.c_int32
    L.so_design_resample_rational.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_double),
                                              C.c_int32, C.POINTER(C.c_int32)]
    L.so_design_resample_arbitrary.restype = C.c_int32
    L.so_design_resample_arbitrary.argtypes = [C.c_double, C.c_int32, C.POINTER(C.c_double),
                                               C.c_int32, C.POINTER(C.c_int32)]
    L.so_design_iir_zpk.restype = C.c_int32
    L.so_design_iir_zpk.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_double,
                                    C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_double)]
    L.so_zpk_to_sos.restype = C.c_int32
    L.so_zpk_to_sos.argtypes = [C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.c_int32, C.c_double,
                                C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.so_tf_to_sos.restype = C.c_int32
    L.so_tf_to_sos.argtypes = [C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double),
                               C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.so_tf_zero_input.restype = C.c_int32
    L.so_tf_zero_input.argtypes = [C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.c_int32,
                                   C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.c_int64,
                                   C.POINTER(C.c_int64)]
    L.so_rtc_compile_check.restype = C.c_int32
    L.so_rtc_compile_check.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
    L.so_rtc_wait_idle.restype = C.c_int32
    L.so_rtc_wait_idle.argtypes = []
    L.so_rtc_shutdown.restype = C.c_int32
    L.so_rtc_shutdown.argtypes = []
    import atexit

    atexit.register(L.so_rtc_shutdown)  # (before the interpreter tears torch's GPU state down: include/sigops.h)
    L.so_comm_unique_id.restype = C.c_int32
    L.so_comm_unique_id.argtypes = [C.c_void_p]
    L.so_comm_create.restype = C.c_int32
    L.so_comm_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.so_comm_allgather.restype = C.c_int32
    L.so_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(so_slab_t), C.c_int32, C.c_void_p]
    L.so_comm_reduce_sum.restype = C.c_int32
    L.so_comm_reduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    L.so_comm_last_error.restype = C.c_char_p
    L.so_comm_destroy.restype = None
    L.so_comm_destroy.argtypes = [C.c_void_p]
    L.so_plan_counter.restype = C.c_int64
    L.so_plan_counter.argtypes = [C.c_void_p, C.c_int32]
    L.so_plan_step_info.restype = C.c_int32
    L.so_plan_step_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(so_step_info_t)]
    L.so_resample_positions.restype = C.c_int32
    L.so_resample_positions.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_double),
                                        C.c_int32, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    if L.so_abi_version() != 1:
        raise EngineMissing("libsigops ABI version mismatch")
    _lib = L
    return L


def last_error():
    return lib().so_last_error().decode("utf-8", "replace")
