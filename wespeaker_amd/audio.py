"""RIFF/WAVE PCM reader (stdlib only; torchaudio is not available here).

Stands in for `torchaudio.load(path, normalize=...)` at cli/speaker.py:126-127 and the native
`WavReader::Open` (runtime/core/frontend/wav.h:71-117): returns a (C, N) tensor -- int16 samples
when normalize=False (the CLI default, `wavform_norm=False`, cli/speaker.py:49), float32 in
[-1, 1) when normalize=True."""
import wave

import numpy as np
import torch


def load_wav(path: str, normalize: bool = False):
    with wave.open(path, "rb") as w:
        nch, width, sr, nframes = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(nframes)
    if width == 2:
        data = np.frombuffer(raw, dtype="<i2").reshape(-1, nch).T
        if normalize:
            out = torch.from_numpy(data.astype(np.float32) / 32768.0)
        else:
            out = torch.from_numpy(data.copy())
    elif width == 1:          # unsigned 8-bit
        data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0).reshape(-1, nch).T
        out = torch.from_numpy(np.ascontiguousarray(data / 128.0 if normalize else data * 256.0))
    elif width == 4:
        data = np.frombuffer(raw, dtype="<i4").reshape(-1, nch).T
        out = torch.from_numpy(np.ascontiguousarray(
            data.astype(np.float32) / 2147483648.0 if normalize else data.astype(np.float32) / 65536.0))
    else:
        raise ValueError("unsupported sample width %d in %s" % (width, path))
    return out, sr


def load_pcm16_fast(source):
    """Channel 0 of a RIFF/WAVE file (or bytes) as a 1-D numpy array + sample rate, for the batch driver's
    decode threads: one read(), a chunk walk over the header, np.frombuffer on the data chunk -- 16-bit PCM
    keeps its int16 samples (what load_wav(normalize=False) returns), other widths fall back to load_wav."""
    import struct
    if isinstance(source, (bytes, bytearray, memoryview)):
        raw = bytes(source)
    else:
        with open(source, "rb") as f:
            raw = f.read()
    if len(raw) >= 12 and raw[:4] == b"RIFF" and raw[8:12] == b"WAVE":
        pos, fmt = 12, None
        while pos + 8 <= len(raw):
            cid, size = raw[pos:pos + 4], struct.unpack_from("<I", raw, pos + 4)[0]
            if cid == b"fmt ":
                fmt = struct.unpack_from("<HHIIHH", raw, pos + 8)
            elif cid == b"data":
                if fmt is not None and fmt[0] == 1 and fmt[5] == 16 and fmt[1] >= 1:
                    n = min(size, len(raw) - pos - 8) // (2 * fmt[1])
                    data = np.frombuffer(raw, dtype="<i2", count=n * fmt[1], offset=pos + 8)
                    return (data if fmt[1] == 1 else data[::fmt[1]].copy()), fmt[2]
                break
            pos += 8 + size + (size & 1)
    import io
    pcm, sr = load_wav(io.BytesIO(raw), normalize=False)
    return pcm[0].numpy(), sr


# ------------------------------------------------------------------------------------ resampling
def resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Filter bank of torchaudio.transforms.Resample's default method (sinc_interp_hann), the
    transform the reference applies at cli/speaker.py:157-160.  torchaudio is a third-party dependency
    that is not vendored in the reference nor installed here: this restates its published algorithm
    (functional._get_sinc_resample_kernel) -- float64 construction, float32 result.
    Returns (kernel float32 [new][2*width+orig], orig, new, width) with the rates divided by their gcd."""
    import math
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base_freq))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = t * base_freq
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        kernels = np.where(t == 0, 1.0, np.sin(t) / t)
    kernels = kernels * window * scale
    return kernels.astype(np.float32), orig, new, width


def resample(pcm: torch.Tensor, orig_freq: int, new_freq: int, device=None) -> torch.Tensor:
    """(C, N) or (N,) -> same rank at new_freq, computed by ws_resample on the GPU (float32 like
    torchaudio); output length ceil(new * N / orig)."""
    from . import _lib
    if orig_freq == new_freq:
        return pcm
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    kern, orig, new, width = resample_kernel(orig_freq, new_freq)
    kt = torch.from_numpy(kern).to(dev)
    x = pcm.to(device=dev, dtype=torch.float32)
    squeeze = x.dim() == 1
    if squeeze:
        x = x.unsqueeze(0)
    x = x.contiguous()
    n_in = int(x.shape[1])
    n_out = -(-new * n_in // orig)
    y = torch.empty((x.shape[0], n_out), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        for c in range(x.shape[0]):
            _lib.check(_lib.lib().ws_resample(_lib.ptr(x[c]), n_in, _lib.ptr(kt), orig, new, width,
                                              _lib.ptr(y[c]), n_out, _lib.current_stream_ptr(dev)),
                       "ws_resample")
        torch.cuda.current_stream(dev).synchronize()
    return y[0] if squeeze else y
