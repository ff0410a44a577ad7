"""The reference's only value-level pins of `Filt` (test/runtests.jl:314-350, "Filtering"): two sines at
10 Hz and 5 Hz, every filter type, inequalities between the means of what comes out.  One statement of
them, run against the CPU oracle (tests/test_oracle_golden.py) and against the HIP engine
(tests/test_gpu_parity.py) -- `evaluate` is whichever sink is under test."""
import numpy as np

import sigops_amd as so
from sigops_amd import (Signal, Until, Mix, Amplify, Filt, Normpower, ToChannels, Lowpass, Highpass, Bandpass, Bandstop,
                        Butterworth, Chebyshev1, ErrorException, s, Hz, sin)


def filtering_inequalities(evaluate, nch):
    a = Signal(sin, 100 * Hz, ω=10 * Hz) | ToChannels(nch) | Until(5 * s)
    b = Signal(sin, 100 * Hz, ω=5 * Hz) | ToChannels(nch) | Until(5 * s)
    cmplx = Mix(a, b)
    cheb = Chebyshev1(5, 1)
    high = evaluate(cmplx | Filt(Highpass, 8 * Hz, method=cheb))
    low_tree = cmplx | Filt(Lowpass, 6 * Hz, method=Butterworth(5))
    low = evaluate(low_tree)
    # (the reference filters the materialised `low` again: a data signal at the same rate)
    highlow = evaluate(Signal(np.asfortranarray(low), 100 * Hz) | Filt(Highpass, 8 * Hz, method=cheb))
    bandp1 = evaluate(cmplx | Filt(Bandpass, 20 * Hz, 30 * Hz, method=cheb))
    bandp2 = evaluate(cmplx | Filt(Bandpass, 2 * Hz, 12 * Hz, method=cheb))
    bands1 = evaluate(cmplx | Filt(Bandstop, 20 * Hz, 30 * Hz, method=cheb))
    bands2 = evaluate(cmplx | Filt(Bandstop, 2 * Hz, 12 * Hz, method=cheb))
    for ctor in (lambda: Filt(a, Highpass, 75 * Hz), lambda: Filt(a, Lowpass, 75 * Hz),
                 lambda: Filt(a, Bandpass, 75 * Hz, 80 * Hz), lambda: Filt(a, Bandstop, 75 * Hz, 80 * Hz)):
        try:  # runtests.jl:334-337: a band beyond the Nyquist rate is an error (at construction or at the sink)
            evaluate(ctor())
        except ErrorException:
            pass
        else:
            raise AssertionError("a filter band beyond the Nyquist rate must raise")
    mabs = lambda v: float(np.mean(np.abs(v)))
    assert high.shape == low.shape == highlow.shape == (500, nch)           # runtests.jl:339-341
    assert np.mean(high) < 0.01 and np.mean(low) < 0.02                      # :342-343
    assert 10 * mabs(highlow) < mabs(low) and 10 * mabs(highlow) < mabs(high)  # :344-345
    assert 10 * mabs(bandp1) < mabs(bandp2)                                   # :346
    assert 10 * mabs(bands2) < mabs(bands1)                                   # :347
    assert mabs(evaluate(cmplx | Amplify(10) | Normpower)) < mabs(evaluate(cmplx | Amplify(10)))  # :349-350
    return high
