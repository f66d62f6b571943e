"""ORACLE (test infrastructure, never shipped): CPU restatement of the Kaldi fbank
frontend the reference calls.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

What it restates
----------------
`torchaudio.compliance.kaldi.fbank` -- a THIRD-PARTY dependency of the reference
(setup.py:36 `torchaudio>=2.0.0`, unpinned; not vendored in /root/reference, not
installed in this image).  The published algorithm is restated below for exactly
the arguments the reference passes:

  * wespeaker/cli/speaker.py:92-97      num_mel_bins=80, frame_length=25, frame_shift=10,
                                        sample_frequency=sr, window_type=self.window_type
  * wespeaker/dataset/processor.py:516-525   same + dither (0.0 at extraction), 'hamming',
                                        use_energy=False, waveform*(1<<15)
  * CMN: wespeaker/cli/speaker.py:98-99  feat - feat.mean(0)

torchaudio defaults that therefore apply: snip_edges=True, remove_dc_offset=True,
preemphasis 0.97, round_to_power_of_two=True, use_power=True, use_log_fbank=True,
low_freq=20, high_freq=0 (-> Nyquist), no VTLN, no energy column, channel 0, float32.

PARITY PINNING: the reference has no tests and no golden vectors for this path
(SURVEY.md section 4), and torchaudio cannot be run here.  This restatement is instead
pinned against the reference's OWN native statement of the same algorithm,
runtime/core/frontend/fbank.h:33-97,138-198 + fft.cc, compiled from /root/reference
into oracle/_ref/ (see oracle/Makefile, tests/golden/fbank_ref_native.npz).  At the
torchaudio boundary itself: parity unpinned.
"""
import math

import numpy as np

EPS = np.float32(1.1920928955078125e-07)  # torch.finfo(torch.float32).eps


def next_pow2(n: int) -> int:
    return 1 if n <= 1 else 2 ** (n - 1).bit_length()


def window_function(window_type: str, size: int) -> np.ndarray:
    j = np.arange(size, dtype=np.float64)
    a = 2.0 * math.pi / (size - 1)
    if window_type == "hamming":
        w = 0.54 - 0.46 * np.cos(a * j)
    elif window_type == "hanning":
        w = 0.5 - 0.5 * np.cos(a * j)
    elif window_type == "povey":
        w = (0.5 - 0.5 * np.cos(a * j)) ** 0.85
    elif window_type == "rectangular":
        w = np.ones(size)
    else:
        raise ValueError("Invalid window type " + window_type)
    return w.astype(np.float32)


def mel_scale(freq):
    return np.float32(1127.0) * np.log(np.float32(1.0) + freq / np.float32(700.0))


def mel_banks(num_bins: int, padded_window: int, sample_freq: float,
              low_freq: float = 20.0, high_freq: float = 0.0) -> np.ndarray:
    """(num_bins, padded_window//2) triangular filters, triangular in the mel domain."""
    num_fft_bins = padded_window // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded_window
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float32)[:, None]
    left = np.float32(mel_low) + b * np.float32(delta)
    center = np.float32(mel_low) + (b + np.float32(1.0)) * np.float32(delta)
    right = np.float32(mel_low) + (b + np.float32(2.0)) * np.float32(delta)
    mel = mel_scale(np.float32(fft_bin_width) * np.arange(num_fft_bins, dtype=np.float32))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return np.maximum(np.float32(0.0), np.minimum(up, down)).astype(np.float32)


def num_frames(num_samples: int, frame_len: int, frame_shift: int) -> int:
    if num_samples < frame_len:
        return 0
    return 1 + (num_samples - frame_len) // frame_shift


def kaldi_fbank(waveform, num_mel_bins=80, frame_length=25.0, frame_shift=10.0,
                sample_frequency=16000.0, window_type="povey", dither=0.0,
                preemphasis=0.97, remove_dc_offset=True, low_freq=20.0, high_freq=0.0,
                cmn=False) -> np.ndarray:
    """waveform: (C, N) or (N,) in int16-range floats (or [-1,1] floats) -> (T, num_mel_bins) f32.

    NB torchaudio's own default window is 'povey'; the reference passes 'hamming' unless
    `set_window_type('povey')` was called (cli/speaker.py:50,66-67).
    """
    x = np.asarray(waveform)
    if x.ndim == 2:
        x = x[0]                                      # channel=-1 -> channel 0
    x = x.astype(np.float32)
    wshift = int(sample_frequency * frame_shift * 0.001)
    wsize = int(sample_frequency * frame_length * 0.001)
    padded = next_pow2(wsize)
    m = num_frames(x.shape[0], wsize, wshift)
    if m == 0:
        return np.zeros((0, num_mel_bins), np.float32)
    idx = np.arange(m)[:, None] * wshift + np.arange(wsize)[None, :]
    frames = x[idx]                                   # (m, wsize) snip_edges=True
    if dither != 0.0:
        raise NotImplementedError("dither is forced to 0.0 on the extraction path "
                                  "(reference bin/extract.py:84-85)")
    if remove_dc_offset:
        frames = frames - frames.mean(axis=1, dtype=np.float32, keepdims=True)
    if preemphasis != 0.0:
        prev = np.concatenate([frames[:, :1], frames[:, :-1]], axis=1)   # replicate pad
        frames = frames - np.float32(preemphasis) * prev
    frames = frames * window_function(window_type, wsize)[None, :]
    if padded != wsize:
        frames = np.pad(frames, ((0, 0), (0, padded - wsize)))
    spec = np.fft.rfft(frames.astype(np.float32), axis=1)
    power = (np.abs(spec).astype(np.float32)) ** np.float32(2.0)        # (m, padded/2+1)
    banks = mel_banks(num_mel_bins, padded, sample_frequency, low_freq, high_freq)
    banks = np.pad(banks, ((0, 0), (0, 1)))                              # zero Nyquist column
    mel = power.astype(np.float32) @ banks.T.astype(np.float32)
    feat = np.log(np.maximum(mel.astype(np.float32), EPS)).astype(np.float32)
    if cmn:
        feat = feat - feat.mean(axis=0, dtype=np.float32, keepdims=True)
    return feat.astype(np.float32)


def speaker_features(pcm_int16, window_type="hamming", sample_rate=16000, cmn=True):
    """The exact `Speaker.compute_features` recipe (cli/speaker.py:90-106) for PCM16 input
    loaded with normalize=False (cli/speaker.py:126-127,156): int16-scale floats."""
    x = np.asarray(pcm_int16).astype(np.float32)
    return kaldi_fbank(x, num_mel_bins=80, frame_length=25, frame_shift=10,
                       sample_frequency=sample_rate, window_type=window_type, cmn=cmn)


def apply_cmvn(feats, norm_mean=True, norm_var=False):
    """The reference's test-time CMVN (wespeaker/dataset/dataset_utils.py:19-26, called at bin/extract.py:124-127):
    feats (B, T, F) float32; mean over T subtracted, then divided by sqrt(unbiased var over T + 1e-7) -- the variance
    is taken of the already mean-subtracted tensor.  numpy float32 restatement (pinned to the reference function on
    tests/golden/cmvn_ref.npz)."""
    x = np.asarray(feats, dtype=np.float32)
    if norm_mean:
        x = x - x.mean(axis=1, keepdims=True, dtype=np.float32)
    if norm_var:
        with np.errstate(invalid="ignore", divide="ignore"):
            x = x / np.sqrt(x.var(axis=1, keepdims=True, ddof=1, dtype=np.float32) + np.float32(1e-7))
    return x.astype(np.float32)
