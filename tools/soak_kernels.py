"""Soak: the seeded kernel-geometry fuzz tests of tests/test_gpu_fuzz.py (resampler geometries, IIR
geometries, fused sine gains) with many more seeds.  python tools/soak_kernels.py SEED0 SEED1"""
import sys, traceback
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_gpu_fuzz as t
bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    for fn in (t.test_random_resampler_geometries, t.test_random_iir_geometries, t.test_random_fused_sine_gains):
        n += 1
        try:
            fn(seed)
        except Exception as e:
            bad += 1
            print('FAIL', fn.__name__, seed, str(e)[:300].replace('\n', ' '), flush=True)
print('runs', n, 'bad', bad)
