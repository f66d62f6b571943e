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
