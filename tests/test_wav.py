"""WAV glue (reference src/WAV.jl): container logic on the CPU, save/load round trip through the
engine on the GPU (checked against the oracle and against SciPy's independent WAV reader)."""
import struct

import numpy as np
import pytest

import sigops_amd as so
from sigops_amd import Hz, kHz, frames, s  # noqa: F401
from sigops_amd.wav import _read_chunks, load_signal


def _write_pcm16(path, data, fs):
    pcm = np.clip(np.round(data * 32768.0), -32768, 32767).astype("<i2")
    payload = pcm.tobytes()
    fmt = struct.pack("<HHIIHH", 1, data.shape[1], fs, fs * data.shape[1] * 2, data.shape[1] * 2, 16)
    body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return pcm


def test_load_pcm16_and_framerate_check(tmp_path):
    rng = np.random.default_rng(3)
    data = rng.uniform(-0.9, 0.9, size=(50, 2))
    pcm = _write_pcm16(tmp_path / "a.wav", data, 8000)
    x = load_signal(str(tmp_path / "a.wav"))
    assert so.framerate(x) == 8000 and so.nchannels(x) == 2 and so.nframes(x) == 50
    np.testing.assert_array_equal(x.data, pcm.astype(np.float64) / 32768.0)
    assert x.data.flags.f_contiguous  # planar, like Julia's Array
    load_signal(str(tmp_path / "a.wav"), 8 * kHz)  # consistent rate: fine
    with pytest.raises(so.ErrorException, match="ToFramerate"):  # src/WAV.jl:10-13
        load_signal(str(tmp_path / "a.wav"), 44.1 * kHz)


def test_rejects_non_wave(tmp_path):
    p = tmp_path / "b.wav"
    p.write_bytes(b"RIFX" + b"\0" * 40)
    with pytest.raises(so.ErrorException):
        load_signal(str(p))
    assert _read_chunks(b"RIFF" + struct.pack("<I", 4) + b"WAVE") == {}


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_save_signal_round_trip(tmp_path, dtype):
    from oracle_bridge import oracle_sink
    from scipy.io import wavfile

    rng = np.random.default_rng(11)
    noise = rng.standard_normal((4000, 2)).astype(dtype)
    x = so.Signal(noise, 8 * kHz) | so.Until(3500 * frames) | so.Ramp(100 * frames) | so.Amplify(dtype(0.25))
    want = oracle_sink(x)
    path = str(tmp_path / "c.wav")
    so.save_signal(path, x)
    rate, got = wavfile.read(path)  # independent reader: interleaved IEEE float
    # (Float32 data * a Float64 literal promotes to Float64, as in Julia; a Float32 gain keeps Float32)
    assert rate == 8000 and got.dtype == want.dtype and got.shape == want.shape
    assert np.linalg.norm(got.astype(np.float64) - want) <= 1e-6 * np.linalg.norm(want)
    y = so.load_signal(path, 8 * kHz)  # and back through our own reader
    np.testing.assert_array_equal(y.data, got)
