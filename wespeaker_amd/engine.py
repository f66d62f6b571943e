"""Host-side wrappers around the native frontend and model engine.

`NativeSpeakerModel` stands where the reference has a torch.nn.Module
(wespeaker/models/speaker_model.py:31-62 `get_speaker_model(name)(**args)` +
wespeaker/utils/checkpoint.py:20-85 `load_checkpoint`): it is callable on a (B, T, F) float32
feature tensor and returns a tuple whose last element is the (B, E) embedding, which is all the
reference's callers use (`outputs[-1] if isinstance(outputs, tuple) else outputs`,
cli/speaker.py:119-120,165; bin/extract.py:134).  PyTorch is used only to own device memory and
streams; every FLOP of the forward runs in the HIP library.
"""
import ctypes
from ctypes import c_int64, c_void_p

import numpy as np
import torch

from . import _lib

WINDOW_TYPES = {"hamming": 0, "povey": 1}

SUPPORTED_MODELS = ("ECAPA_TDNN_c512", "ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_c1024",
                    "ECAPA_TDNN_GLOB_c1024", "ResNet18", "ResNet34", "ResNet50", "ResNet101",
                    "ResNet152", "ResNet221", "ResNet293", "CAMPPlus")

DEFAULT_EMBED_DIM = {"ECAPA": 192, "ResNe": 256, "CAMPP": 512}


def default_device() -> torch.device:
    import os
    _lib.require_gpu()
    idx = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    return torch.device("cuda", idx)


class Frontend:
    """Kaldi fbank (+CMN) on the GPU: replaces torchaudio.compliance.kaldi.fbank as called at
    cli/speaker.py:92-99 and dataset/processor.py:516-525."""

    def __init__(self, sample_rate=16000, num_mel_bins=80, device=None):
        self.device = torch.device(device) if device is not None else default_device()
        self.sample_rate = sample_rate
        self.num_mel_bins = num_mel_bins
        h = c_void_p()
        _lib.check(_lib.lib().ws_frontend_create(sample_rate, num_mel_bins, self.device.index or 0,
                                                 ctypes.byref(h)), "ws_frontend_create")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:
            _lib._lib.ws_frontend_destroy(h)
            self._h = None

    def num_frames(self, num_samples: int) -> int:
        return _lib.lib().ws_num_frames(int(num_samples), self.sample_rate)

    def set_cmvn(self, norm_mean=True, norm_var=False):
        """Which apply_cmvn (dataset/dataset_utils.py:19-26) this frontend's normalisation step is -- the `cmn=True`
        of fbank / fbank_ragged and the fused extract calls; (False, False) = the config's `cmvn: False`."""
        _lib.check(_lib.lib().ws_frontend_set_cmvn(self._h, int(bool(norm_mean)), int(bool(norm_var))),
                   "ws_frontend_set_cmvn")
        self.norm_mean, self.norm_var = bool(norm_mean), bool(norm_var)
        return self

    def fbank_ragged(self, wav: torch.Tensor, num_samples, window_type="hamming", cmn=True, scale=1.0):
        """Padded (B, Nmax) waveforms + per-utterance sample counts -> (B, Tmax, bins) features whose rows
        beyond an utterance's own frames are zero; CMN over the utterance's own frames (ws_fbank_ragged)."""
        if wav.dtype == torch.int16:
            dt = 0
        else:
            wav = wav.to(torch.float32)
            dt = 1
        wav = wav.to(self.device).contiguous()
        B, N = wav.shape
        ns = np.ascontiguousarray(np.asarray(num_samples, dtype=np.int32).reshape(-1))
        n_max = int(ns.max()) if B else 0
        T = self.num_frames(n_max)
        feats = torch.empty((B, T, self.num_mel_bins), dtype=torch.float32, device=self.device)
        if B and T:
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().ws_fbank_ragged(self._h, _lib.ptr(wav), dt, B, _lib.ptr(ns), n_max,
                                                      wav.stride(0), float(scale), WINDOW_TYPES[window_type],
                                                      int(bool(cmn)), _lib.ptr(feats),
                                                      _lib.current_stream_ptr(self.device)), "ws_fbank_ragged")
        return feats

    def fbank(self, wav: torch.Tensor, window_type="hamming", cmn=True, scale=1.0) -> torch.Tensor:
        """wav: (B, N) int16 or float32 (int16-range unless `scale` says otherwise), any device
        -> (B, T, num_mel_bins) float32 on the GPU."""
        if window_type not in WINDOW_TYPES:
            raise ValueError("Invalid window type " + str(window_type))
        if wav.dim() == 1:
            wav = wav.unsqueeze(0)
        if wav.dtype == torch.int16:
            dt = 0
        else:
            wav = wav.to(torch.float32)
            dt = 1
        wav = wav.to(self.device).contiguous()
        B, N = wav.shape
        T = self.num_frames(N)
        feats = torch.empty((B, T, self.num_mel_bins), dtype=torch.float32, device=self.device)
        if B == 0 or T == 0:
            return feats
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ws_fbank(self._h, _lib.ptr(wav), dt, B, N, wav.stride(0),
                                           float(scale), WINDOW_TYPES[window_type], int(bool(cmn)),
                                           _lib.ptr(feats), _lib.current_stream_ptr(self.device)),
                       "ws_fbank")
        return feats


def save_native_model(path, model_name, state_dict, feat_dim=80, embed_dim=None):
    """Write the flat weight file ws_engine_load reads (the native runtime's counterpart of an
    exported .onnx: callers without Python -- wespeaker_amd/csrc/bin/extract_emb_main.cc -- load it).
    Only floating-point tensors are written, under the reference's state_dict names."""
    import struct
    embed_dim = int(embed_dim or DEFAULT_EMBED_DIM.get(model_name[:5], 256))
    items = []
    for key, val in state_dict.items():
        arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
        if arr.dtype.kind == "f" and arr.ndim <= 4:
            items.append((key, np.ascontiguousarray(arr, dtype="<f4")))
    with open(path, "wb") as f:
        name = model_name.encode()
        f.write(b"WSAMDW01" + struct.pack("<i", len(name)) + name)
        f.write(struct.pack("<iii", int(feat_dim), embed_dim, len(items)))
        for key, arr in items:
            k = key.encode()
            f.write(struct.pack("<i", len(k)) + k + struct.pack("<i", arr.ndim))
            f.write(struct.pack("<%dq" % arr.ndim, *arr.shape))
            f.write(arr.tobytes())
    return path


def dispatch_log(on=True, clear=False):
    """Diagnostic: start / stop noting which kernel every distinct conv / linear problem is given
    (ws_debug_dispatch_log; process-wide).  `clear` forgets what was noted so far."""
    _lib.check(_lib.lib().ws_debug_dispatch_log(2 if (on and clear) else (1 if on else 0)), "ws_debug_dispatch_log")
    if clear and not on:
        _lib.check(_lib.lib().ws_debug_dispatch_log(2), "ws_debug_dispatch_log")
        _lib.check(_lib.lib().ws_debug_dispatch_log(0), "ws_debug_dispatch_log")


def dispatch_report():
    """The noted (problem -> kernel) table as sorted text lines (ws_debug_dispatch_report)."""
    import ctypes
    need = int(_lib.lib().ws_debug_dispatch_report(None, 0))
    buf = ctypes.create_string_buffer(max(need, 1))
    _lib.lib().ws_debug_dispatch_report(buf, need)
    return [l for l in buf.value.decode().split("\n") if l]


class NativeSpeakerModel:
    """The engine behind `Speaker.model`."""

    @classmethod
    def from_file(cls, path, device=None, max_batch=64, max_frames=400):
        """Engine from a flat weight file (save_native_model) through ws_engine_load."""
        self = cls.__new__(cls)
        self.device = torch.device(device) if device is not None else default_device()
        h = c_void_p()
        _lib.check(_lib.lib().ws_engine_load(str(path).encode(), self.device.index or 0, int(max_batch),
                                             int(max_frames), ctypes.byref(h)), "ws_engine_load")
        self._h = h
        with open(path, "rb") as f:
            f.seek(8)
            n = int.from_bytes(f.read(4), "little")
            self.model_name = f.read(n).decode()
        self.feat_dim = _lib.lib().ws_engine_feat_dim(h)
        self.embed_dim = _lib.lib().ws_engine_embed_dim(h)
        self.frontend_type = "fbank"
        self.max_batch, self.max_frames = int(max_batch), int(max_frames)
        self._rows_budget = self.max_batch * self.max_frames
        self._max_batch0 = self.max_batch
        self.ignored_keys = []
        self.precision = "fp32"
        return self

    def __init__(self, model_name: str, state_dict, feat_dim=80, embed_dim=None, device=None,
                 max_batch=64, max_frames=400, **unused_model_args):
        self.device = torch.device(device) if device is not None else default_device()
        self.model_name = model_name
        self.feat_dim = int(feat_dim)
        self.embed_dim = int(embed_dim or DEFAULT_EMBED_DIM.get(model_name[:5], 256))
        self.frontend_type = "fbank"
        self.max_batch = int(max_batch)
        self.max_frames = int(max_frames)
        L = _lib.lib()
        h = c_void_p()
        _lib.check(L.ws_engine_create(model_name.encode(), self.feat_dim, self.embed_dim,
                                      self.device.index or 0, ctypes.byref(h)), "ws_engine_create")
        self._h = h
        self.ignored_keys = []
        for key, val in state_dict.items():
            arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
            if arr.dtype.kind != "f":
                continue                       # num_batches_tracked etc.
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (c_int64 * max(1, arr.ndim))(*arr.shape)
            used = _lib.check(L.ws_engine_set_tensor(h, key.encode(), _lib.ptr(arr), arr.ndim, shape),
                              "ws_engine_set_tensor(%s)" % key)
            if not used:
                self.ignored_keys.append(key)
        _lib.check(L.ws_engine_finalize(h, self.max_batch, self.max_frames), "ws_engine_finalize")
        self._rows_budget = self.max_batch * self.max_frames
        self._max_batch0 = self.max_batch
        self.precision = "fp32"

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:
            _lib._lib.ws_engine_destroy(h)
            self._h = None

    # nn.Module look-alikes so reference-style call sites keep working
    def eval(self):
        return self

    def to(self, device):
        dev = torch.device(device)
        if dev.type == "cuda" and (dev.index is None or dev.index == self.device.index):
            return self
        if dev.type == "cpu":
            return self          # outputs are returned wherever the caller asks; compute stays on the GPU
        raise _lib.NativeError("engine was created on %s; create a new one for %s" % (self.device, dev))

    PRECISIONS = {"fp32": 0, "f16x3": 1, "f16": 2}

    def set_precision(self, mode):
        """'fp32' (exact fp32 MFMA, default) or 'f16x3' (3-pass split-binary16 MFMA with fp32
        accumulation; fp32-grade accuracy, several times faster)."""
        code = self.PRECISIONS[mode] if isinstance(mode, str) else int(mode)
        _lib.check(_lib.lib().ws_engine_set_precision(self._h, code), "ws_engine_set_precision")
        self.precision = {v: k for k, v in self.PRECISIONS.items()}[code]
        return self

    def check_range(self):
        """Binary16 back-ends only: synchronise the current stream and raise NativeError if an activation
        left the binary16 range since the last check (ws_engine_check_range).  No-op in fp32."""
        if self.precision != "fp32":
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().ws_engine_check_range(self._h, _lib.current_stream_ptr(self.device)),
                           "ws_engine_check_range")

    def flops(self, batch, frames) -> float:
        return float(_lib.lib().ws_engine_flops(self._h, int(batch), int(frames)))

    # kernel classes of ws_engine_profile_read: every conv/linear GEMM launch with N > 64 (whatever tile
    # and back-end the dispatcher picks: the dominant class), the narrow ones (N <= 64, fused Res2 chain),
    # reductions / element-wise / frontend, split-K GEMMs of the M = B layers
    PROFILE_CLASSES = ("gemm_main", "gemm_narrow", "reduce_elementwise", "gemm_splitk")

    def profile(self, on):
        """on: False/0 = off, True = all kernel classes, int = bit mask (1 = dominant GEMM only)."""
        mask = 0xF if on is True else int(on)
        _lib.check(_lib.lib().ws_engine_profile_enable(self._h, mask))

    def profile_read(self):
        """-> {class: dict(ms, flops, bytes, launches)}; synchronises the recorded events."""
        ms = (ctypes.c_double * 4)()
        fl = (ctypes.c_double * 4)()
        by = (ctypes.c_double * 4)()
        ln = (ctypes.c_int * 4)()
        _lib.check(_lib.lib().ws_engine_profile_read(self._h, ms, fl, by, ln))
        return {n: dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=ln[i])
                for i, n in enumerate(self.PROFILE_CLASSES)}

    def reserve(self, max_batch, max_frames):
        """Re-size the engine workspace (ws_engine_reserve; synchronises the device)."""
        _lib.check(_lib.lib().ws_engine_reserve(self._h, int(max_batch), int(max_frames)),
                   "ws_engine_reserve")
        self.max_batch, self.max_frames = int(max_batch), int(max_frames)

    def _ensure_capacity(self, frames, batch=0):
        """The reference takes utterances of any length (cli/speaker.py:125-167): when one exceeds the
        finalized capacity the workspace is re-laid-out for it, keeping rows (= max_batch x max_frames,
        i.e. the memory footprint) about constant -- longer utterances run in smaller chunks.  When much
        shorter batches come back (a length-bucketed list), the layout is turned back towards many short
        rows so that they do not run in needlessly small chunks.  Each change synchronises the device."""
        rows = self._rows_budget
        if frames > self.max_frames:
            new_frames = -(-int(frames * 1.25) // 100) * 100
            self.reserve(max(1, min(self._max_batch0, rows // new_frames)), new_frames)
        elif batch > self.max_batch and self.max_batch < self._max_batch0:
            new_frames = max(-(-int(frames * 1.25) // 100) * 100, 100)
            new_batch = max(1, min(self._max_batch0, rows // new_frames))
            if new_frames < self.max_frames and new_batch >= 2 * self.max_batch:
                self.reserve(new_batch, new_frames)

    def embed(self, feats: torch.Tensor) -> torch.Tensor:
        """(B, T, F) float32 -> (B, E) float32 on the GPU."""
        if feats.dim() != 3 or feats.shape[2] != self.feat_dim:
            raise ValueError("expected (B, T, %d) features, got %s" % (self.feat_dim, tuple(feats.shape)))
        feats = feats.to(device=self.device, dtype=torch.float32).contiguous()
        B, T, _ = feats.shape
        self._ensure_capacity(T)
        emb = torch.empty((B, self.embed_dim), dtype=torch.float32, device=self.device)
        if B:
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().ws_forward(self._h, _lib.ptr(feats), B, T, _lib.ptr(emb),
                                                 _lib.current_stream_ptr(self.device)), "ws_forward")
        return emb

    def __call__(self, feats: torch.Tensor):
        """Return convention of the reference modules (callers use `outputs[-1] if tuple`):
        ECAPA -> (out4, embed) (ecapa_tdnn.py:227-234), ResNet -> (tensor(0.0), embed_a) or
        (embed_a, embed_b) (resnet.py:192-204), CAM++ -> tensor (campplus.py:409-413).
        The leading element is not materialised on this path (None for ECAPA / two-layer ResNet)."""
        emb = self.embed(feats)
        if self.model_name.startswith("CAMPPlus"):
            return emb
        if self.model_name.startswith("ResNet"):
            return torch.tensor(0.0), emb
        return None, emb

    def extract(self, frontend: Frontend, wav: torch.Tensor, window_type="hamming", scale=1.0):
        """Fused wav -> fbank -> CMN -> forward (ws_extract).  wav (B, N) int16/float32."""
        if wav.dim() == 1:
            wav = wav.unsqueeze(0)
        if wav.dtype == torch.int16:
            dt = 0
        else:
            wav = wav.to(torch.float32)
            dt = 1
        wav = wav.to(self.device).contiguous()
        B, N = wav.shape
        self._ensure_capacity(frontend.num_frames(N), B)
        emb = torch.empty((B, self.embed_dim), dtype=torch.float32, device=self.device)
        if B:
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().ws_extract(self._h, frontend._h, _lib.ptr(wav), dt, B, N,
                                                 wav.stride(0), float(scale),
                                                 WINDOW_TYPES[window_type], _lib.ptr(emb),
                                                 _lib.current_stream_ptr(self.device)), "ws_extract")
        return emb

    def embed_ragged(self, feats: torch.Tensor, num_frames, cmvn=None) -> torch.Tensor:
        """(B, Tmax, F) float32 features of utterances of different lengths + their frame counts ->
        (B, E): row b equals embed(feats[b:b+1, :num_frames[b]]) (ws_forward_ragged); the content of the
        padding rows is ignored.  cmvn=(norm_mean, norm_var): the features are raw -- apply_cmvn over every
        utterance's own frames first (ws_forward_ragged_cmvn; `data_type: feat` lists)."""
        if feats.dim() != 3 or feats.shape[2] != self.feat_dim:
            raise ValueError("expected (B, T, %d) features, got %s" % (self.feat_dim, tuple(feats.shape)))
        feats = feats.to(device=self.device, dtype=torch.float32).contiguous()
        B, T, _ = feats.shape
        lens = np.ascontiguousarray(np.asarray(num_frames, dtype=np.int32).reshape(-1))
        if lens.shape[0] != B:
            raise ValueError("num_frames has %d entries for a batch of %d" % (lens.shape[0], B))
        self._ensure_capacity(T)
        emb = torch.empty((B, self.embed_dim), dtype=torch.float32, device=self.device)
        if B:
            with torch.cuda.device(self.device):
                if cmvn is not None and (cmvn[0] or cmvn[1]):
                    _lib.check(_lib.lib().ws_forward_ragged_cmvn(self._h, _lib.ptr(feats), B, T, _lib.ptr(lens),
                                                                 int(bool(cmvn[0])), int(bool(cmvn[1])), _lib.ptr(emb),
                                                                 _lib.current_stream_ptr(self.device)),
                               "ws_forward_ragged_cmvn")
                else:
                    _lib.check(_lib.lib().ws_forward_ragged(self._h, _lib.ptr(feats), B, T, _lib.ptr(lens),
                                                            _lib.ptr(emb), _lib.current_stream_ptr(self.device)),
                               "ws_forward_ragged")
        return emb

    def extract_ragged(self, frontend: Frontend, wav: torch.Tensor, num_samples, window_type="hamming",
                       scale=1.0):
        """Fused wav -> fbank -> CMN -> forward for utterances of different lengths (ws_extract_ragged):
        wav (B, Nmax) int16/float32 padded rows (padding content irrelevant), num_samples[b] valid samples.
        Row b equals extract(wav[b:b+1, :num_samples[b]])."""
        if wav.dtype == torch.int16:
            dt = 0
        else:
            wav = wav.to(torch.float32)
            dt = 1
        wav = wav.to(self.device)
        if wav.stride(-1) != 1:
            wav = wav.contiguous()
        B, N = wav.shape
        ns = np.ascontiguousarray(np.asarray(num_samples, dtype=np.int32).reshape(-1))
        if ns.shape[0] != B:
            raise ValueError("num_samples has %d entries for a batch of %d" % (ns.shape[0], B))
        n_max = int(ns.max()) if B else 0
        self._ensure_capacity(frontend.num_frames(n_max), B)
        emb = torch.empty((B, self.embed_dim), dtype=torch.float32, device=self.device)
        if B:
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().ws_extract_ragged(self._h, frontend._h, _lib.ptr(wav), dt, B, _lib.ptr(ns),
                                                        n_max, wav.stride(0), float(scale),
                                                        WINDOW_TYPES[window_type], _lib.ptr(emb),
                                                        _lib.current_stream_ptr(self.device)),
                           "ws_extract_ragged")
        return emb

    def extract_chunked(self, frontend: Frontend, wav: torch.Tensor, samples_per_chunk: int,
                        window_type="hamming", scale=1.0):
        """Chunk-and-average extraction of ONE utterance: the native runtime's
        SpeakerEngine::ExtractEmbedding (runtime/core/speaker/speaker_engine.cc:83-159) via
        ws_extract_chunked.  samples_per_chunk <= 0 = full mode.  Returns ((E,) tensor, n_chunks)."""
        wav = wav.reshape(-1)
        if wav.dtype == torch.int16:
            dt = 0
        else:
            wav = wav.to(torch.float32)
            dt = 1
        wav = wav.to(self.device).contiguous()
        n = int(wav.shape[0])
        total = frontend.num_frames(n)
        if samples_per_chunk > 0:
            ms = frontend.sample_rate // 1000
            self._ensure_capacity(max(1, 1 + (samples_per_chunk - ms * 25) // (ms * 10)))
        else:
            self._ensure_capacity(total)
        emb = torch.empty((self.embed_dim,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            n_chunks = _lib.check(_lib.lib().ws_extract_chunked(
                self._h, frontend._h, _lib.ptr(wav), dt, n, int(samples_per_chunk), float(scale),
                WINDOW_TYPES[window_type], _lib.ptr(emb), _lib.current_stream_ptr(self.device)),
                "ws_extract_chunked")
        return emb, n_chunks


    def extract_windows(self, frontend: Frontend, wav: torch.Tensor, seg_length=None, window_frames=150,
                        period_frames=75, subseg_cmn=True, window_type="hamming", scale=1.0):
        """Diarization sub-segment embeddings of ONE speech segment in one call (ws_extract_windows): fbank of the
        whole segment -> the windows of diar/extract_emb.py:55-83 (`window_frames` every `period_frames`, laid out
        from `seg_length`, the last one completed by tiling its own frames) -> optional per-window CMN -> forward
        (cli/speaker.py:232-251, 108-123).  seg_length None = num_frames + 2, what the reference's time stamps give
        for its own VAD segments.  Returns an (n_windows, E) tensor on the GPU."""
        wav = wav.reshape(-1)
        if wav.dtype == torch.int16:
            dt = 0
        else:
            wav = wav.to(torch.float32)
            dt = 1
        wav = wav.to(self.device).contiguous()
        n = int(wav.shape[0])
        total = frontend.num_frames(n)
        if seg_length is None:
            seg_length = total + 2
        n_win = _lib.lib().ws_num_windows(int(seg_length), int(window_frames), int(period_frames))
        if total <= 0 or n_win <= 0:
            raise ValueError("segment of %d samples is shorter than one frame" % n)
        self._ensure_capacity(int(window_frames))
        emb = torch.empty((n_win, self.embed_dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            got = _lib.check(_lib.lib().ws_extract_windows(
                self._h, frontend._h, _lib.ptr(wav), dt, n, int(seg_length), int(window_frames), int(period_frames),
                float(scale), WINDOW_TYPES[window_type], 1 if subseg_cmn else 0, _lib.ptr(emb), n_win,
                _lib.current_stream_ptr(self.device)), "ws_extract_windows")
        assert got == n_win
        return emb


class LaneResult:
    """One batch in flight on a lane: the embeddings tensor becomes valid when `event` has fired."""

    __slots__ = ("tensor", "event", "lane")

    def __init__(self, tensor, event, lane):
        self.tensor, self.event, self.lane = tensor, event, lane

    def wait(self):
        """Make the CURRENT stream wait for this batch (no host synchronisation) and return its embeddings."""
        cur = torch.cuda.current_stream(self.tensor.device)
        cur.wait_event(self.event)
        self.tensor.record_stream(cur)          # (allocated under the lane's stream, consumed on this one)
        return self.tensor

    def synchronize(self):
        self.event.synchronize()
        return self.tensor


class SpeakerModelLanes:
    """`lanes` engines of one model (weights replicated, one workspace each) on `lanes` HIP streams; batch i runs on
    lane i % lanes, so two (or three) batches are in flight on the GPU at any time.

    Why: one 256 x 2 s ECAPA forward is ~35 launches.  Its seven persistent GEMMs fill the chip, everything between
    them does not -- the SE / context FCs and the split-K layers are chains of L2 round trips on a few waves per
    CU, the launches for the rows behind the last whole round of GEMM tiles occupy 192 of 256 CUs at 0.3 MFMA
    utilisation, every persistent kernel ramps up and drains -- and inside one stream those gaps cannot be filled,
    because every launch depends on the one before it.  A second batch has no such dependence: with two batches in
    flight the hardware dispatcher places the other lane's workgroups on whatever a lane leaves idle (measured on
    ECAPA-GLOB-512, fp32: 61.0 k -> 63.8 k utt/s with two lanes, 64.4 k with three).  Per-batch latency doubles;
    results are the same bits (same kernels, same launch parameters per batch).

    extract() / embed() return a LaneResult immediately; call .wait() (stream-side) or .synchronize() (host-side)
    before the embeddings are consumed.  Inputs must stay unmodified until then (they may be dropped: the lane's stream
    is recorded on them)."""

    def __init__(self, model_name, state_dict, lanes=2, **kwargs):
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        self.engines = [NativeSpeakerModel(model_name, state_dict, **kwargs) for _ in range(int(lanes))]
        self.device = self.engines[0].device
        self.streams = [torch.cuda.Stream(self.device) for _ in self.engines]
        self._next = 0
        self.model_name, self.embed_dim, self.feat_dim = model_name, self.engines[0].embed_dim, self.engines[0].feat_dim

    @property
    def lanes(self):
        return len(self.engines)

    def _run(self, fn, *inputs):
        lane = self._next
        self._next = (lane + 1) % len(self.engines)
        stream = self.streams[lane]
        stream.wait_stream(torch.cuda.current_stream(self.device))      # the inputs are ready
        for t in inputs:                      # the caller may drop / rebind them as soon as we return: keep the
            if isinstance(t, torch.Tensor) and t.is_cuda:       # caching allocator from reusing the block before the
                t.record_stream(stream)                          # lane's kernels have read it
        with torch.cuda.stream(stream):
            out = fn(self.engines[lane])
            ev = torch.cuda.Event()
            ev.record(stream)
        return LaneResult(out, ev, lane)

    def extract(self, frontend, wav, window_type="hamming", scale=1.0):
        return self._run(lambda e: e.extract(frontend, wav, window_type=window_type, scale=scale), wav)

    def extract_ragged(self, frontend, wav, num_samples, **kw):
        return self._run(lambda e: e.extract_ragged(frontend, wav, num_samples, **kw), wav)

    def embed(self, feats):
        return self._run(lambda e: e.embed(feats), feats)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def set_precision(self, mode):
        """Every lane gets the back-end.  (Round 3 refused the binary16 back-ends here: next to their kernels the
        fbank kernel of any engine returned a few mel bins wrong.  Root cause and fix: DESIGN.md 6.0 -- a packed-fp32
        instruction form of fbank's power-spectrum loop; the shipped kernel has none, and the library is checked for
        that form at build time.)"""
        for e in self.engines:
            e.set_precision(mode)
        return self

    def check_range(self):
        self.synchronize()
        for e in self.engines:
            e.check_range()

    def flops(self, batch, frames):
        return self.engines[0].flops(batch, frames)
