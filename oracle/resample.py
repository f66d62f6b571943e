"""ORACLE (test infrastructure, never shipped): numpy restatement of torchaudio's default
`Resample` (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99), the transform the reference
applies when the file's rate differs from `resample_rate` (wespeaker/cli/speaker.py:157-160).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY UNPINNED: torchaudio is a third-party dependency (setup.py: torchaudio>=2.0.0, unpinned) that
is neither vendored in /root/reference nor installed here; this restates its published algorithm
(torchaudio.functional.resample: _get_sinc_resample_kernel + _apply_sinc_resample_kernel) with an
independent formulation (per-output-sample loop in float64 products) of the same filter."""
import math

import numpy as np


def resample(x, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    x = np.asarray(x, dtype=np.float32)
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = orig_freq // g, new_freq // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    n_out = -(-new * x.shape[-1] // orig)
    xp = np.concatenate([np.zeros(width, np.float32), x, np.zeros(width + orig, np.float32)])
    out = np.zeros(n_out, dtype=np.float32)
    k = np.arange(-width, width + orig, dtype=np.float64)
    for ph in range(new):
        t = (-ph / new + k / orig) * base
        t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
        win = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        tt = t * math.pi
        with np.errstate(invalid="ignore", divide="ignore"):
            kern = (np.where(tt == 0, 1.0, np.sin(tt) / tt) * win * (base / orig)).astype(np.float32)
        idx = np.arange(ph, n_out, new)
        q = idx // new
        starts = q * orig                       # into the padded signal (already shifted by width)
        frames = np.stack([xp[s:s + kern.shape[0]] for s in starts]) if len(starts) else np.zeros((0, kern.shape[0]))
        out[idx] = (frames.astype(np.float32) * kern[None, :]).sum(1, dtype=np.float32)
    return out
