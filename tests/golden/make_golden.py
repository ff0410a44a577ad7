"""Generates tests/golden/*.npz (run in the build container:  python tests/golden/make_golden.py).

The reference is Julia and cannot run here or on the GPU box, and its own tests hold no filter /
resampler vectors, so the fixtures come from two independent sources:
  * design.npz  -- SciPy (NOT this repo's code): Butterworth / Chebyshev-I ZPK for the filters the
                   configs use, and `firwin` Kaiser taps for the resampling ratios; pins the
                   C-ABI design entry points.
  * cases.npz   -- the CPU oracle's result for every seeded tree of tests/cases.py (full arrays up
                   to 20 000 elements, else head / tail / norm / sum); pins the oracle against
                   silent drift and gives the GPU box expected values that do not depend on the
                   oracle binary built there.
Inputs are regenerated from the seeds in tests/cases.py; only expected outputs are stored."""
import os
import sys

import numpy as np
from scipy import signal as sps

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def kaiser_taps(rate, nphi):
    """DSP.jl resample_filter restated with SciPy"""
    f_nyq = 1.0 / nphi if rate >= 1 else rate / nphi
    tw = 0.2 * f_nyq
    n = int(np.ceil((60 - 7.95) / (np.pi * 2.285 * tw))) + 1
    hlen = nphi * int(np.ceil(n / nphi))
    if hlen % 2 == 0:
        hlen += 1
    beta = 0.1102 * (60 - 8.7)
    return sps.firwin(hlen, f_nyq, window=("kaiser", beta)) * nphi


def main():
    design = {}
    for name, (N, wn, bt, fs) in {"bandstop_500_2000_44100": (5, [500, 2000], "bandstop", 44100),
                                  "bandstop_500_2000_48000": (5, [500, 2000], "bandstop", 48000),
                                  "lowpass_4000_16000": (5, 4000, "lowpass", 16000)}.items():
        z, p, k = sps.butter(N, wn, bt, fs=fs, output="zpk")
        design[name + "_z"], design[name + "_p"], design[name + "_k"] = z, p, np.array(k)
    z, p, k = sps.cheby1(5, 1, 8, "highpass", fs=100, output="zpk")
    design["cheby1_hp_8_100_z"], design["cheby1_hp_8_100_p"], design["cheby1_hp_8_100_k"] = z, p, np.array(k)
    design["taps_48000_44100"] = kaiser_taps(48000 / 44100, 32)
    design["taps_16000_44100"] = kaiser_taps(16000 / 44100, 32)
    design["taps_2_1"] = kaiser_taps(2.0, 2)
    np.savez_compressed(os.path.join(HERE, "design.npz"), **design)

    from cases import CASES
    from oracle_bridge import oracle_sink

    out = {}
    for name in sorted(CASES):
        y = np.asarray(oracle_sink(CASES[name]()))
        out[name + "__shape"] = np.array(y.shape)
        out[name + "__dtype"] = np.array(str(y.dtype))
        if y.size <= 20000:
            out[name + "__full"] = y
        else:
            out[name + "__head"] = y[:64].copy()
            out[name + "__tail"] = y[-64:].copy()
            out[name + "__norm"] = np.array(np.linalg.norm(y.astype(np.float64)))
            out[name + "__sum"] = np.array(y.astype(np.float64).sum())
    np.savez_compressed(os.path.join(HERE, "cases.npz"), **out)
    print("design.npz", os.path.getsize(os.path.join(HERE, "design.npz")), "bytes; cases.npz",
          os.path.getsize(os.path.join(HERE, "cases.npz")), "bytes;", len(CASES), "cases")


if __name__ == "__main__":
    main()
