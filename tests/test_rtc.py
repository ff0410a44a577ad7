"""hipRTC-specialised pointwise kernels: the compile path needs no device --
the generated source of a pointwise step compiles for gfx950 against the library's embedded
definitions (CPU test); on the GPU the specialised kernels are compared with the interpreter kernel
and the oracle (tests/test_gpu_rtc.py)."""
import ctypes as C

from sigops_amd import _capi as K

BODY = r'''
__device__ __forceinline__ void p0_frame(const DLeaf* __restrict__ L, long long N, double* M) {
    const int C = 0; (void)C; (void)N; (void)L; (void)M;
    M[0] = (func_eval(L[1], N) * ramp_eval(L[2], N));
}
__device__ __forceinline__ double p0_samp(const DLeaf* __restrict__ L, long long N, int C, const double* M) {
    (void)N; (void)C; (void)L; (void)M;
    return ((double)(float)((leaf_load(L[0], N, C) * M[0]) + (leaf_load(L[3], N, C) / L[4].v0)));
}
extern "C" __global__ __launch_bounds__(256) void k_rtc(const DPiece* __restrict__ pieces, int npieces, const DLeaf* __restrict__ L, OutView out) {
    const long long bid = blockIdx.x;
    const DPiece P = pieces[0];
    const long long n0 = P.a + bid * 512ll + 2ll * threadIdx.x;
    if (n0 >= P.b || npieces < 1) return;
    const bool ok1 = n0 + 1 < P.b;
    const long long n1 = ok1 ? n0 + 1 : n0;
    double M0[1], M1[1];
    p0_frame(L, n0, M0);
    p0_frame(L, n1, M1);
    for (int c = P.c0; c < P.c1; ++c) so_store2(out, n0, c, p0_samp(L, n0, c, M0), p0_samp(L, n1, c, M1), ok1);
}
'''


def test_generated_source_compiles_for_gfx950_without_a_device():
    log = C.create_string_buffer(8000)
    st = K.lib().so_rtc_compile_check(BODY.encode(), log, 8000)
    assert st == 0, log.value.decode()


def test_compile_errors_come_back_with_the_log():
    log = C.create_string_buffer(8000)
    st = K.lib().so_rtc_compile_check(BODY.replace("func_eval", "no_such_function").encode(), log, 8000)
    assert st != 0 and "no_such_function" in log.value.decode()
