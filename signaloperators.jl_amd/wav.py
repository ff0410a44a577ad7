"""WAV glue: `save_signal` / `load_signal` (reference src/WAV.jl:3-15).

The reference calls WAV.jl (`wavwrite(data, file, Fs=round(Int, fs))` on the `sink(x, Tuple)`
result, `wavread(file)` → `Signal(data, fs)`).  WAV.jl is a third-party dependency that is not
vendored under the reference tree; what is restated here is the container it produces for
floating-point data: RIFF/WAVE, `fmt ` chunk with format tag 3 (IEEE float), 32 or 64 bits
per sample, interleaved frames, a `fact` chunk as the format requires for non-PCM data.

MI355X-side: the engine writes the sink result directly in the file's interleaved layout
(`frame_stride = nch, chan_stride = 1` in `so_out_desc_t`), so there is no host transpose
between the D2H copy and the file.  Integer PCM files are read and scaled to [-1, 1) by
2^(bits-1) (parity with WAV.jl's integer scaling is unpinned: no fixture in the reference).
"""
import struct

import numpy as np

from . import signals as S
from .engine import process_sink_params, sink_into

_FMT_PCM, _FMT_FLOAT, _FMT_EXT = 1, 3, 0xFFFE


def save_signal(filename, x, *, device=0):
    """`x |> sink("file.wav")` (reference src/sink.jl:139-142 → src/WAV.jl:3-6)"""
    x = process_sink_params(x)
    if x.dtype == S.I64:
        S.error("the HIP engine sinks Float32/Float64 signals only; integer signals take the stock CPU sink")
    n, nch = S.nframes(x), x.nch
    data = np.empty((n, nch), dtype=x.dtype, order="C")  # interleaved, as in the file
    if n:
        sink_into(data, x, device=device)
    fs = int(round(float(x.fs)))
    bits = data.dtype.itemsize * 8
    payload = data.tobytes()
    fmt = struct.pack("<HHIIHH", _FMT_FLOAT, nch, fs, fs * nch * bits // 8, nch * bits // 8, bits)
    fact = struct.pack("<I", n)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact
    body += b"data" + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")
    with open(filename, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return filename


def _read_chunks(buf):
    if buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        S.error("not a RIFF/WAVE file")
    pos, out = 12, {}
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        out.setdefault(cid, buf[pos + 8:pos + 8 + size])
        pos += 8 + size + (size & 1)
    return out


def load_signal(filename, fs=None):
    """`Signal("file.wav"[, fs])` (reference src/WAV.jl:8-15): frame-rate mismatch is an error,
    conversion is the caller's `ToFramerate`."""
    with open(filename, "rb") as f:
        ch = _read_chunks(f.read())
    if b"fmt " not in ch or b"data" not in ch:
        S.error(f"{filename}: missing fmt/data chunk")
    tag, nch, rate, _, _, bits = struct.unpack("<HHIIHH", ch[b"fmt "][:16])
    if tag == _FMT_EXT and len(ch[b"fmt "]) >= 26:
        tag = struct.unpack("<H", ch[b"fmt "][24:26])[0]  # sub-format GUID's first field
    raw = ch[b"data"]
    if tag == _FMT_FLOAT and bits in (32, 64):
        data = np.frombuffer(raw, dtype="<f4" if bits == 32 else "<f8")
    elif tag == _FMT_PCM and bits in (8, 16, 24, 32):
        if bits == 8:
            data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
        elif bits == 24:
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            data = (v - ((v & 0x800000) << 1)).astype(np.float64) / float(1 << 23)
        else:
            data = np.frombuffer(raw, dtype="<i2" if bits == 16 else "<i4").astype(np.float64) / float(1 << (bits - 1))
    else:
        S.error(f"{filename}: unsupported WAV format tag {tag} / {bits} bits")
    data = np.asfortranarray(data.reshape(-1, nch))  # planar, like Julia's Array
    if fs is not None:
        from . import units as U

        want = U.inHz(fs)
        if want is not None and float(want) != float(rate):
            S.error(f"Expected file {filename} to have framerate {fs}. If you wish to convert the frame rate, "
                    "you can use `ToFramerate`.")
    return S.Signal(data, rate)
