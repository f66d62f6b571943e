"""Drop-in for the reference's Python API surface on the extraction path:

    wespeaker.load_model(dir) -> Speaker            (wespeaker/cli/speaker.py:300-301)
    wespeaker.load_model_pt(dir) -> model           (:306-335)
    Speaker.extract_embedding / _from_pcm / _list / _from_feats, setters, cosine helpers (:60-211)

Same names, argument meaning, return types and error behaviour; the arithmetic (fbank, CMN, model
forward) runs in the HIP library on an MI355X.  Out-of-scope methods (VAD, diarization, model-hub
download -- SURVEY.md section 2 rows 1/17) raise NotImplementedError instead of silently doing
something else.
"""
import os

import numpy as np
import torch
import yaml

from . import _lib
from .audio import load_wav
from .engine import Frontend, NativeSpeakerModel, default_device


def _load_state_dict(path: str):
    """utils/checkpoint.py:20-33: torch.load, unwrap {'state_dict': ...}."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(ckpt, dict) and "state_dict" in ckpt:
        ckpt = ckpt["state_dict"]
    return ckpt


def load_model_pt(model_name_or_path: str, device=None, max_batch=64, max_frames=400):
    """`config.yaml` + `avg_model.pt` -> native model (cli/speaker.py:306-335).  Raises
    FileNotFoundError exactly like the reference when either file is missing (:312-315)."""
    model_dir = model_name_or_path
    for file in ("config.yaml", "avg_model.pt"):
        if not os.path.exists(os.path.join(model_dir, file)):
            raise FileNotFoundError(f"{file} not found in {model_dir}")
    with open(os.path.join(model_dir, "config.yaml"), "r") as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    frontend_type = "fbank"
    if "dataset_args" in config and "frontend" in config["dataset_args"]:
        frontend_type = config["dataset_args"]["frontend"]
    if frontend_type != "fbank":
        raise NotImplementedError("only the 'fbank' frontend is on the MI355X hot path "
                                  "(got %r; SSL frontends are out of scope)" % frontend_type)
    model_args = dict(config.get("model_args") or {})
    pooling = model_args.pop("pooling_func", None)
    if config["model"].startswith("ECAPA") and pooling not in (None, "ASTP"):
        raise NotImplementedError("ECAPA-TDNN with pooling_func=%r (only ASTP)" % pooling)
    if config["model"].startswith(("ResNet", "CAMPPlus")) and pooling not in (None, "TSTP"):
        raise NotImplementedError("%s with pooling_func=%r (only TSTP)" % (config["model"], pooling))
    sd = _load_state_dict(os.path.join(model_dir, "avg_model.pt"))
    if model_args.pop("emb_bn", False) and "bn2.running_mean" not in sd:
        raise KeyError("emb_bn=True but bn2.* is missing from the checkpoint")
    if model_args.pop("two_emb_layer", False) and "seg_2.weight" not in sd:
        raise KeyError("two_emb_layer=True but seg_2.* is missing from the checkpoint")
    model = NativeSpeakerModel(config["model"], sd, device=device, max_batch=max_batch,
                               max_frames=max_frames, **model_args)
    model.frontend_type = frontend_type
    model.config = config
    return model


def subsegment_ids(seg_id, seg_length, window_fs, period_fs):
    """The sub-segment names of diar/extract_emb.py:55-83 ("<seg_id>-<first frame:08d>-<last frame:08d>")."""
    if seg_length <= window_fs:
        return [seg_id + "-{:08d}-{:08d}".format(0, seg_length)]
    out = []
    for begin in range(0, seg_length - window_fs + period_fs, period_fs):
        out.append(seg_id + "-{:08d}-{:08d}".format(begin, min(begin + window_fs, seg_length)))
    return out


def subsegment(fbank, seg_id, window_fs, period_fs, frame_shift):
    """Host mirror of diar/extract_emb.py:55-83 (same name, arguments and return value): (subseg ids, list of
    (window_fs, F) arrays).  `extract_subsegment_embeddings` does this on the device; this form is for callers that
    hold features on the host, like the reference's `extract_emb.py` reading an fbank scp."""
    seg_begin, seg_end = seg_id.split('-')[-2:]
    seg_length = (int(seg_end) - int(seg_begin)) // frame_shift
    fbank = np.asarray(fbank)
    feat_dim = fbank.shape[1]
    subsegs = subsegment_ids(seg_id, seg_length, window_fs, period_fs)
    if seg_length <= window_fs:
        return subsegs, [np.resize(fbank, (window_fs, feat_dim))]
    feats = []
    for name in subsegs:
        b, e = (int(t) for t in name.split('-')[-2:])
        feats.append(np.resize(fbank[b:e], (window_fs, feat_dim)))
    return subsegs, feats


class Speaker:

    def __init__(self, model_dir: str, device=None, max_batch=64, max_frames=400):
        _lib.require_gpu()
        self.device = torch.device(device) if device is not None else default_device()
        self.model = load_model_pt(model_dir, device=self.device, max_batch=max_batch,
                                   max_frames=max_frames)
        self.vad = None
        self.table = {}
        self.resample_rate = 16000
        self.apply_vad = False
        self.output_device = torch.device("cpu")
        self.wavform_norm = False
        self.window_type = "hamming"
        self._frontends = {}
        # diarization params kept for API compatibility
        self.diar_min_duration = 0.255
        self.diar_window_secs = 1.5
        self.diar_period_secs = 0.75
        self.diar_frame_shift = 10
        self.diar_batch_size = 32
        self.diar_subseg_cmn = True

    # ------------------------------------------------------------------ setters (speaker.py:60-88)
    def set_wavform_norm(self, wavform_norm: bool):
        self.wavform_norm = wavform_norm

    def set_window_type(self, window_type: str):
        self.window_type = window_type

    def set_resample_rate(self, resample_rate: int):
        self.resample_rate = resample_rate

    def set_vad(self, apply_vad: bool):
        if apply_vad:
            raise NotImplementedError("silero VAD is outside the MI355X hot path (SURVEY.md s.2 row 1)")
        self.apply_vad = False

    def set_device(self, device: str):
        """The engine always computes on its MI355X; 'cpu' is accepted and only means "hand results
        back on the CPU" (what extract_embedding does anyway, speaker.py:166)."""
        dev = torch.device(device)
        if dev.type == "cuda":
            idx = self.device.index if dev.index is None else dev.index
            if idx != self.device.index:
                raise _lib.NativeError("Speaker was built on %s; build a new one for cuda:%d"
                                       % (self.device, idx))
        self.model = self.model.to(dev)

    def set_precision(self, mode: str):
        """Extension (not in the reference API): GEMM back-end of the forward -- 'fp32' (default,
        exact fp32 MFMA), 'f16x3' (fp32-grade split binary16) or 'f16' (binary16 operands and
        inter-layer tensors with fp32 accumulation: what the reference's TensorRT-fp16 runtime
        computes).  See include/wespeaker_amd.h, ws_engine_set_precision."""
        self.model.set_precision(mode)

    def set_diarization_params(self, min_duration=0.255, window_secs=1.5, period_secs=0.75,
                               frame_shift=10, batch_size=32, subseg_cmn=True):
        self.diar_min_duration = min_duration
        self.diar_window_secs = window_secs
        self.diar_period_secs = period_secs
        self.diar_frame_shift = frame_shift
        self.diar_batch_size = batch_size
        self.diar_subseg_cmn = subseg_cmn

    # ------------------------------------------------------------------------------- features
    def _frontend(self, sample_rate):
        fe = self._frontends.get(sample_rate)
        if fe is None:
            fe = Frontend(sample_rate, self.model.feat_dim, device=self.device)
            self._frontends[sample_rate] = fe
        return fe

    def compute_features(self, wavform, sample_rate=16000, cmn=True):
        """(C, N) waveform -> (1, T, F) features on the GPU (speaker.py:90-106; channel 0)."""
        wav = wavform[0:1] if wavform.dim() == 2 else wavform.unsqueeze(0)
        return self._frontend(sample_rate).fbank(wav, window_type=self.window_type, cmn=cmn)

    def extract_embedding_from_feats(self, fbanks, batch_size, subseg_cmn):
        """list of (T, F) arrays (or one (N, T, F) array / tensor) -> (N, E) numpy (speaker.py:108-123).  One upload of
        the stacked windows; the per-window CMN (`fbanks_array - np.mean(fbanks_array, axis=1)`, :110-112) runs on the
        device (ws_cmn), then `batch_size` windows per forward like the reference's loop."""
        if isinstance(fbanks, torch.Tensor):
            arr_t = fbanks.to(device=self.device, dtype=torch.float32)
            arr_t = arr_t.clone() if subseg_cmn else arr_t            # (the CMN below is in place)
        else:
            arr = np.ascontiguousarray(np.stack(fbanks), dtype=np.float32)
            arr_t = torch.from_numpy(arr).to(self.device)
        arr_t = arr_t.contiguous()
        if arr_t.dim() != 3:
            raise ValueError("expected N windows of (T, F) features, got shape %s" % (tuple(arr_t.shape),))
        n, t, f = arr_t.shape
        if subseg_cmn and n:
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().ws_cmn(_lib.ptr(arr_t), n, t, f, _lib.current_stream_ptr(self.device)), "ws_cmn")
        out = []
        for i in range(0, n, batch_size):
            emb = self.model(arr_t[i:i + batch_size])
            emb = emb[-1] if isinstance(emb, tuple) else emb
            out.append(emb)
        res = torch.cat(out).cpu().numpy() if out else np.zeros((0, self.model.embed_dim), np.float32)
        self.model.check_range()
        return res

    def extract_subsegment_embeddings(self, pcm, sample_rate, begin_ms=0, end_ms=None):
        """One speech segment -> (subseg ids, (n, E) numpy embeddings): the body of the per-segment loop of
        `Speaker.diarize` (cli/speaker.py:232-251) -- compute_features(cmn=False), `subsegment()`
        (diar/extract_emb.py:55-83) with this Speaker's window / period / frame shift, per-window CMN if
        `diar_subseg_cmn`, forward -- as ONE device call (ws_extract_windows); no feature tensor or window stack is
        built on the host.  `pcm` is the segment's samples ((N,) or (1, N), int16-scale like the reference's
        `pcm[0, begin_idx:end_idx]`); [begin_ms, end_ms) its position in the recording (end_ms None = begin_ms + its
        duration): the reference lays the windows out from (end_ms - begin_ms) // frame_shift, not from the frame
        count, and names them "{begin_ms:08d}-{end_ms:08d}-{first:08d}-{last:08d}".  (The VAD that produces the segments
        and the clustering behind them stay out of scope: SURVEY.md s.2.)

        One deviation (stated in INTEGRATION.md too): a segment whose `sample_rate` differs from `resample_rate` is
        resampled first, like `extract_embedding_from_pcm` does (cli/speaker.py:157-160); the reference's `diarize`
        computes its features at the file's own rate (cli/speaker.py:232-237).  For 16 kHz input -- every recipe
        of the reference -- the two are the same; the window NAMES are the same at any rate (they come from the
        millisecond time stamps)."""
        pcm = pcm.reshape(-1)
        if sample_rate != self.resample_rate:
            from .audio import resample
            pcm = resample(pcm.to(torch.float), sample_rate, self.resample_rate, self.device)
        n = int(pcm.shape[0])
        if end_ms is None:
            end_ms = int(begin_ms) + n * 1000 // self.resample_rate
        window_fs = int(self.diar_window_secs * 1000) // self.diar_frame_shift
        period_fs = int(self.diar_period_secs * 1000) // self.diar_frame_shift
        seg_length = (int(end_ms) - int(begin_ms)) // self.diar_frame_shift
        seg_id = "{:08d}-{:08d}".format(int(begin_ms), int(end_ms))
        subsegs = subsegment_ids(seg_id, seg_length, window_fs, period_fs)
        emb = self.model.extract_windows(self._frontend(self.resample_rate), pcm, seg_length=seg_length,
                                         window_frames=window_fs, period_frames=period_fs,
                                         subseg_cmn=self.diar_subseg_cmn, window_type=self.window_type)
        res = emb.cpu().numpy()
        self.model.check_range()
        assert len(subsegs) == res.shape[0]
        return subsegs, res

    # ------------------------------------------------------------------------------ extraction
    def extract_embedding(self, audio_path: str):
        pcm, sample_rate = load_wav(audio_path, normalize=self.wavform_norm)
        return self.extract_embedding_from_pcm(pcm, sample_rate)

    def extract_embedding_from_pcm(self, pcm: torch.Tensor, sample_rate: int):
        if self.apply_vad:
            raise NotImplementedError("VAD is outside the MI355X hot path")
        if sample_rate != self.resample_rate:       # cli/speaker.py:157-160 (torchaudio Resample)
            from .audio import resample
            pcm = resample(pcm.to(torch.float), sample_rate, self.resample_rate, self.device)
        wav = pcm[0:1] if pcm.dim() == 2 else pcm.unsqueeze(0)
        fe = self._frontend(self.resample_rate)
        emb = self.model.extract(fe, wav, window_type=self.window_type)
        out = emb[0].to(torch.device("cpu"))
        self.model.check_range()          # binary16 back-ends: loud if the checkpoint left their range
        return out

    def extract_embedding_batch(self, wavs: torch.Tensor, sample_rate: int = 16000):
        """(B, N) equal-length utterances -> (B, E) on the GPU (the batched form of
        extract_embedding_from_pcm; results are identical to B single calls)."""
        if sample_rate != self.resample_rate:
            from .audio import resample
            wavs = resample(wavs.to(torch.float), sample_rate, self.resample_rate, self.device)
        return self.model.extract(self._frontend(self.resample_rate), wavs,
                                  window_type=self.window_type)

    def extract_embedding_list(self, scp_path: str):
        """scp of `name wav_path` lines -> (names, [np.ndarray(E)]) (speaker.py:169-178).
        Utterances are independent, so equal-length ones share a device batch; file decode, H2D and
        the forward overlap (wespeaker_amd.extract.GpuExtractor); the order of the output follows
        the scp like the reference's batch-1 loop."""
        from . import extract as wx

        def entries():
            with open(scp_path, "r") as read_scp:
                for line in read_scp:
                    if not line.strip():
                        continue
                    name, wav_path = line.strip().split()
                    yield name, (lambda w=wav_path: load_wav(w, normalize=self.wavform_norm))

        def resample_fn(pcm, sr, target):
            from .audio import resample
            return resample(pcm.to(torch.float), sr, target, self.device).cpu()

        ex = wx.GpuExtractor(self.model, self._frontend(self.resample_rate), self.window_type)
        names, emb = wx.extract_entries(entries(), ex, batch_size=1, max_batch=self.model.max_batch,
                                        resample_rate=self.resample_rate, resample_fn=resample_fn)
        return names, [e for e in emb]

    # ------------------------------------------------------------------- similarity (:180-211)
    def compute_similarity(self, audio_path1: str, audio_path2: str) -> float:
        e1 = self.extract_embedding(audio_path1)
        e2 = self.extract_embedding(audio_path2)
        if e1 is None or e2 is None:
            return 0.0
        return self.cosine_similarity(e1, e2)

    def cosine_similarity(self, e1, e2):
        cosine_score = torch.dot(e1, e2) / (torch.norm(e1) * torch.norm(e2))
        return (cosine_score.item() + 1.0) / 2      # [-1, 1] => [0, 1]  (speaker.py:188-191)

    def register(self, name: str, audio_path: str):
        if name in self.table:
            print("Speaker {} already registered, ignore".format(name))
        else:
            self.table[name] = self.extract_embedding(audio_path)

    def recognize(self, audio_path: str):
        q = self.extract_embedding(audio_path)
        best_score, best_name = 0.0, ""
        for name, e in self.table.items():
            score = self.cosine_similarity(q, e)
            if best_score < score:
                best_score, best_name = score, name
        return {"name": best_name, "confidence": best_score}

    def diarize(self, audio_path: str, utt: str = "unk"):
        raise NotImplementedError("diarization is outside the MI355X hot path (SURVEY.md s.2 row 17)")

    def diarize_list(self, scp_path: str):
        raise NotImplementedError("diarization is outside the MI355X hot path")


def load_model(model_name_or_path: str, **kw) -> Speaker:
    """wespeaker.load_model: local model directory only (no network here for the model hub)."""
    if not os.path.isdir(model_name_or_path):
        raise FileNotFoundError("model directory %r not found (hub download is not available)"
                                % model_name_or_path)
    return Speaker(model_name_or_path, **kw)
