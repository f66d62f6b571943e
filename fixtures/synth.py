"""Seeded synthetic inputs: wavs, model weights, PLDA parameters, trial lists.

There is no network (no VoxCeleb, no pretrained checkpoints), so every test,
fixture and benchmark in this repo runs on the generators below.  They follow
SURVEY.md section 8(d).  Everything is driven by numpy's PCG64 so the same seed
gives the same bytes on every machine (the GPU box cannot read fixtures that
are too large to commit, e.g. 6 M-parameter weight sets).

State-dict key names/shapes are the reference's parameter names
(wespeaker/models/ecapa_tdnn.py:160-206, pooling_layers.py:97-117,
resnet.py:110-169, campplus.py:333-407) because the weight ingestion path
(`load_model` -> `avg_model.pt`) is keyed by them.
"""
from collections import OrderedDict

import numpy as np

SAMPLE_RATE = 16000


# --------------------------------------------------------------------------- wavs
def synth_wav(utt_idx: int, num_samples: int = 32000) -> np.ndarray:
    """One PCM16 utterance: gaussian noise (sigma 3000) + an 8000-amplitude tone."""
    rng = np.random.Generator(np.random.PCG64(1234 + int(utt_idx)))
    f0 = rng.uniform(80.0, 400.0)
    t = np.arange(num_samples, dtype=np.float64) / SAMPLE_RATE
    x = 3000.0 * rng.standard_normal(num_samples) + 8000.0 * np.sin(2 * np.pi * f0 * t)
    # a slow amplitude envelope so that frames differ in energy
    x *= 0.6 + 0.4 * np.sin(2 * np.pi * (0.7 + 0.1 * (utt_idx % 5)) * t)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def synth_wav_batch(first_idx: int, batch: int, num_samples: int = 32000) -> np.ndarray:
    return np.stack([synth_wav(first_idx + i, num_samples) for i in range(batch)])


def write_wav(path: str, pcm: np.ndarray, sample_rate: int = SAMPLE_RATE) -> None:
    """Minimal RIFF/WAVE PCM16 mono writer (stdlib only)."""
    import wave
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(np.ascontiguousarray(pcm, dtype="<i2").tobytes())


# ------------------------------------------------------------------------ weights
class _Init:
    def __init__(self, seed):
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.sd = OrderedDict()

    def conv(self, name, cout, cin, *k, bias=True, gain=2.0):
        fan_in = cin * int(np.prod(k)) if k else cin
        shape = (cout, cin) + tuple(k)
        std = np.sqrt(gain / fan_in)
        self.sd[name + ".weight"] = (std * self.rng.standard_normal(shape)).astype(np.float32)
        if bias:
            self.sd[name + ".bias"] = (0.1 * self.rng.standard_normal(cout)).astype(np.float32)

    def linear(self, name, cout, cin, bias=True, gain=1.0):
        self.conv(name, cout, cin, bias=bias, gain=gain)

    def bn(self, name, c, affine=True):
        """Randomised eval-mode BN (default init would make BN ~identity and hide bugs)."""
        if affine:
            self.sd[name + ".weight"] = self.rng.uniform(0.8, 1.2, c).astype(np.float32)
            self.sd[name + ".bias"] = (0.1 * self.rng.standard_normal(c)).astype(np.float32)
        self.sd[name + ".running_mean"] = (0.1 * self.rng.standard_normal(c)).astype(np.float32)
        self.sd[name + ".running_var"] = self.rng.uniform(0.5, 1.5, c).astype(np.float32)
        self.sd[name + ".num_batches_tracked"] = np.array(1000, dtype=np.int64)


def ecapa_config(model_name: str):
    """(channels, global_context_att) for the reference constructor names."""
    table = {
        "ECAPA_TDNN_c512": (512, False),
        "ECAPA_TDNN_GLOB_c512": (512, True),
        "ECAPA_TDNN_c1024": (1024, False),
        "ECAPA_TDNN_GLOB_c1024": (1024, True),
    }
    if model_name not in table:
        raise KeyError(model_name)
    return table[model_name]


def synth_ecapa_state_dict(model_name="ECAPA_TDNN_GLOB_c512", feat_dim=80, embed_dim=192,
                           emb_bn=False, seed=42):
    C, glob = ecapa_config(model_name)
    w = C // 8
    g = _Init(seed)
    g.conv("layer1.conv", C, feat_dim, 5)
    g.bn("layer1.bn", C)
    for L in (2, 3, 4):
        p = "layer%d.se_res2block" % L
        g.conv(p + ".0.conv", C, C, 1)
        g.bn(p + ".0.bn", C)
        for i in range(7):
            g.conv(p + ".1.convs.%d" % i, w, w, 3)
            g.bn(p + ".1.bns.%d" % i, w)
        g.conv(p + ".2.conv", C, C, 1, gain=0.7)
        g.bn(p + ".2.bn", C)
        g.linear(p + ".3.linear1", 128, C)
        g.linear(p + ".3.linear2", C, 128, gain=4.0)
    g.conv("conv", 1536, 3 * C, 1, gain=1.0)
    g.conv("pool.linear1", 128, 1536 * (3 if glob else 1), 1, gain=1.0)
    g.conv("pool.linear2", 1536, 128, 1, gain=8.0)   # wide logits -> non-trivial softmax
    g.bn("bn", 3072)
    g.linear("linear", embed_dim, 3072)
    if emb_bn:
        g.bn("bn2", embed_dim)
    return g.sd


RESNET_LAYOUTS = {      # reference resnet.py:207-260: (bottleneck, blocks per stage)
    "ResNet18": (False, [2, 2, 2, 2]), "ResNet34": (False, [3, 4, 6, 3]),
    "ResNet50": (True, [3, 4, 6, 3]), "ResNet101": (True, [3, 4, 23, 3]),
    "ResNet152": (True, [3, 8, 36, 3]), "ResNet221": (True, [6, 16, 48, 3]),
    "ResNet293": (True, [10, 20, 64, 3]),
}


def synth_resnet_state_dict(model_name="ResNet34", feat_dim=80, embed_dim=256,
                            two_emb_layer=False, seed=42):
    bottleneck, blocks = RESNET_LAYOUTS[model_name]
    exp = 4 if bottleneck else 1
    m = 32
    g = _Init(seed)
    g.conv("conv1", m, 1, 3, 3, bias=False)
    g.bn("bn1", m)
    in_planes = m
    for s, nb in enumerate(blocks):
        planes = m * (2 ** s)
        for b in range(nb):
            stride = (1 if s == 0 else 2) if b == 0 else 1
            p = "layer%d.%d" % (s + 1, b)
            if bottleneck:
                g.conv(p + ".conv1", planes, in_planes, 1, 1, bias=False)
                g.bn(p + ".bn1", planes)
                g.conv(p + ".conv2", planes, planes, 3, 3, bias=False)
                g.bn(p + ".bn2", planes)
                g.conv(p + ".conv3", planes * 4, planes, 1, 1, bias=False, gain=0.02)
                g.bn(p + ".bn3", planes * 4)
            else:
                g.conv(p + ".conv1", planes, in_planes, 3, 3, bias=False)
                g.bn(p + ".bn1", planes)
                g.conv(p + ".conv2", planes, planes, 3, 3, bias=False, gain=0.25)
                g.bn(p + ".bn2", planes)
            if stride != 1 or in_planes != planes * exp:
                g.conv(p + ".shortcut.0", planes * exp, in_planes, 1, 1, bias=False, gain=1.0)
                g.bn(p + ".shortcut.1", planes * exp)
            in_planes = planes * exp
    stats_dim = (feat_dim // 8) * m * 8 * exp * 2
    g.linear("seg_1", embed_dim, stats_dim)
    if two_emb_layer:
        g.bn("seg_bn_1", embed_dim, affine=False)
        g.linear("seg_2", embed_dim, embed_dim)
    return g.sd


def synth_campplus_state_dict(feat_dim=80, embed_dim=512, seed=42):
    g = _Init(seed)
    g.conv("head.conv1", 32, 1, 3, 3, bias=False)
    g.bn("head.bn1", 32)
    for layer in ("head.layer1", "head.layer2"):
        for b in (0, 1):
            p = "%s.%d" % (layer, b)
            g.conv(p + ".conv1", 32, 32, 3, 3, bias=False)
            g.bn(p + ".bn1", 32)
            g.conv(p + ".conv2", 32, 32, 3, 3, bias=False, gain=0.5)
            g.bn(p + ".bn2", 32)
            if b == 0:
                g.conv(p + ".shortcut.0", 32, 32, 1, 1, bias=False, gain=1.0)
                g.bn(p + ".shortcut.1", 32)
    g.conv("head.conv2", 32, 32, 3, 3, bias=False)
    g.bn("head.bn2", 32)
    ch = 32 * (feat_dim // 8)
    g.conv("xvector.tdnn.linear", 128, ch, 5, bias=False)
    g.bn("xvector.tdnn.nonlinear.batchnorm", 128)
    ch = 128
    for k, layers in enumerate((12, 24, 16)):
        for j in range(layers):
            p = "xvector.block%d.tdnnd%d" % (k + 1, j + 1)
            cin = ch + 32 * j
            g.bn(p + ".nonlinear1.batchnorm", cin)
            g.conv(p + ".linear1", 128, cin, 1, bias=False)
            g.bn(p + ".nonlinear2.batchnorm", 128)
            g.conv(p + ".cam_layer.linear_local", 32, 128, 3, bias=False)
            g.conv(p + ".cam_layer.linear1", 64, 128, 1)
            g.conv(p + ".cam_layer.linear2", 32, 64, 1, gain=6.0)
        ch = ch + 32 * layers
        p = "xvector.transit%d" % (k + 1)
        g.bn(p + ".nonlinear.batchnorm", ch)
        g.conv(p + ".linear", ch // 2, ch, 1, bias=False)
        ch //= 2
    g.bn("xvector.out_nonlinear.batchnorm", ch)
    g.conv("xvector.dense.linear", embed_dim, ch * 2, 1, bias=False, gain=1.0)
    g.bn("xvector.dense.nonlinear.batchnorm", embed_dim, affine=False)
    return g.sd


def synth_state_dict(model_name, feat_dim=80, embed_dim=None, seed=42, **kw):
    if model_name.startswith("ECAPA_TDNN"):
        kw.pop("pooling_func", None)
        return synth_ecapa_state_dict(model_name, feat_dim, embed_dim or 192, seed=seed, **kw)
    if model_name.startswith("ResNet"):
        kw.pop("pooling_func", None)
        return synth_resnet_state_dict(model_name, feat_dim, embed_dim or 256, seed=seed, **kw)
    if model_name.startswith("CAMPPlus"):
        kw.pop("pooling_func", None)
        return synth_campplus_state_dict(feat_dim, embed_dim or 512, seed=seed)
    raise KeyError("no synthetic weights for model %r" % model_name)


def write_model_dir(model_dir, model_name, feat_dim=80, embed_dim=192, seed=42, **model_kw):
    """Write `config.yaml` + `avg_model.pt` exactly as wespeaker.load_model expects
    (reference cli/speaker.py:306-335)."""
    import os
    import torch
    import yaml
    os.makedirs(model_dir, exist_ok=True)
    sd = synth_state_dict(model_name, feat_dim, embed_dim, seed=seed, **model_kw)
    torch.save(OrderedDict((k, torch.from_numpy(np.asarray(v))) for k, v in sd.items()),
               os.path.join(model_dir, "avg_model.pt"))
    model_args = dict(feat_dim=feat_dim, embed_dim=embed_dim)
    model_args["pooling_func"] = "ASTP" if model_name.startswith("ECAPA") else "TSTP"
    model_args.update(model_kw)
    cfg = {
        "model": model_name,
        "model_args": model_args,
        "dataset_args": {
            "resample_rate": 16000,
            "frontend": "fbank",
            "fbank_args": {"num_mel_bins": feat_dim, "frame_shift": 10,
                           "frame_length": 25, "dither": 0.0},
        },
    }
    with open(os.path.join(model_dir, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    return sd


# --------------------------------------------------------------------------- PLDA
def synth_plda(dim=192, seed=7, normalize_length=False):
    """transform = Q diag(s) (Q Haar), psi sorted-descending Gamma(2,1)+0.01,
    mu ~ N(0, 0.1), offset = -T mu.  All float64 like the reference's HDF5 models."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.standard_normal((dim, dim))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    s = rng.uniform(0.5, 2.0, dim)
    transform = q * s[None, :]
    psi = np.sort(rng.gamma(2.0, 1.0, dim) + 0.01)[::-1].copy()
    mu = 0.1 * rng.standard_normal(dim)
    offset = -transform @ mu
    return {"mu": mu, "transform": transform, "psi": psi, "offset": offset,
            "normalize_length": bool(normalize_length), "subtract_train_set_mean": False}


def synth_embeddings(n, dim=192, seed=11, num_speakers=None):
    """float32 embeddings with speaker structure (so LLRs span target/non-target ranges)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    num_speakers = num_speakers or max(1, n // 8)
    centers = rng.standard_normal((num_speakers, dim))
    spk = rng.integers(0, num_speakers, n)
    x = centers[spk] + 0.6 * rng.standard_normal((n, dim))
    return x.astype(np.float32), spk


def synth_trial_pairs(num_trials, n_enroll, n_test, seed=99):
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.integers(0, n_enroll, num_trials).astype(np.int32),
            rng.integers(0, n_test, num_trials).astype(np.int32))


def synth_scoring_set(n_eval=60, n_cohort=150, dim=192, num_trials=300, seed=21):
    """Cosine-scoring / score-norm fixture: eval + cohort embeddings (a shared offset makes the
    mean subtraction matter), utterance names and a labelled trial list."""
    eval_emb, spk = synth_embeddings(n_eval, dim, seed=seed, num_speakers=max(2, n_eval // 6))
    cohort_emb, _ = synth_embeddings(n_cohort, dim, seed=seed + 1, num_speakers=max(2, n_cohort // 4))
    rng = np.random.Generator(np.random.PCG64(seed + 2))
    offset = (0.5 * rng.standard_normal(dim)).astype(np.float32)
    eval_emb, cohort_emb = eval_emb + offset, cohort_emb + offset
    eval_names = ["utt%04d" % i for i in range(n_eval)]
    cohort_names = ["coh%05d" % i for i in range(n_cohort)]
    ia = rng.integers(0, n_eval, num_trials)
    ib = rng.integers(0, n_eval, num_trials)
    trials = [(eval_names[a], eval_names[b], "target" if spk[a] == spk[b] else "nontarget")
              for a, b in zip(ia, ib)]
    return {"eval_emb": eval_emb, "cohort_emb": cohort_emb, "eval_names": eval_names,
            "cohort_names": cohort_names, "trials": trials,
            "idx_a": ia.astype(np.int32), "idx_b": ib.astype(np.int32)}


def write_scoring_files(fix, out_dir):
    """Write the fixture as the files the reference's score.py / score_norm.py consume."""
    import os
    from wespeaker_amd.kaldi_io import VectorWriter
    os.makedirs(out_dir, exist_ok=True)
    paths = {"eval_scp": os.path.join(out_dir, "xvector.scp"),
             "cohort_scp": os.path.join(out_dir, "cohort.scp"),
             "trials": os.path.join(out_dir, "trials.kaldi"),
             "mean_vec": os.path.join(out_dir, "mean_vec.npy")}
    with VectorWriter(os.path.join(out_dir, "xvector.ark"), paths["eval_scp"]) as w:
        for k, v in zip(fix["eval_names"], fix["eval_emb"]):
            w(k, v)
    with VectorWriter(os.path.join(out_dir, "cohort.ark"), paths["cohort_scp"]) as w:
        for k, v in zip(fix["cohort_names"], fix["cohort_emb"]):
            w(k, v)
    with open(paths["trials"], "w") as f:
        for t in fix["trials"]:
            f.write("%s %s %s\n" % t)
    np.save(paths["mean_vec"], fix["cohort_emb"].mean(0))
    return paths


def synth_plda_training_set(n=400, dim=64, num_speakers=40, n_adapt=300, seed=41):
    """PLDA training / adaptation fixture: labelled training embeddings (variable utterances per
    speaker) and an unlabelled, shifted and rescaled adaptation set."""
    emb, spk = synth_embeddings(n, dim, seed=seed, num_speakers=num_speakers)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    emb = emb + (0.3 * rng.standard_normal(dim)).astype(np.float32)
    adp, _ = synth_embeddings(n_adapt, dim, seed=seed + 2, num_speakers=max(2, n_adapt // 10))
    adp = (1.3 * adp + (0.5 * rng.standard_normal(dim))).astype(np.float32)
    names = ["trn%05d" % i for i in range(n)]
    return {"emb": emb, "spk": spk, "names": names, "adapt": adp,
            "adapt_names": ["adp%05d" % i for i in range(n_adapt)]}


def write_plda_training_files(fix, out_dir):
    import os
    from wespeaker_amd.kaldi_io import VectorWriter
    os.makedirs(out_dir, exist_ok=True)
    paths = {"scp": os.path.join(out_dir, "train.scp"), "utt2spk": os.path.join(out_dir, "utt2spk"),
             "adapt_scp": os.path.join(out_dir, "adapt.scp")}
    with VectorWriter(os.path.join(out_dir, "train.ark"), paths["scp"]) as w:
        for k, v in zip(fix["names"], fix["emb"]):
            w(k, v)
    with open(paths["utt2spk"], "w") as f:
        for k, s in zip(fix["names"], fix["spk"]):
            f.write("%s spk%03d\n" % (k, s))
    with VectorWriter(os.path.join(out_dir, "adapt.ark"), paths["adapt_scp"]) as w:
        for k, v in zip(fix["adapt_names"], fix["adapt"]):
            w(k, v)
    return paths


# ------------------------------------------------------------------ Kaldi <Plda> model files
def write_kaldi_plda(path, mu, transform, psi, binary=True, double=True):
    """A Kaldi `<Plda>` object the way Kaldi's Plda::Write lays it out (the format
    wespeaker/utils/plda/kaldi_utils.py:24-55 reads): token, mean vector, transform matrix, psi
    vector, closing token.  The reference has no writer; this exists to build test fixtures."""
    import struct
    mu, transform, psi = (np.asarray(a, dtype=np.float64) for a in (mu, transform, psi))
    if binary:
        dt, vt, mt = ("<f8", b"DV ", b"DM ") if double else ("<f4", b"FV ", b"FM ")

        def vec(v):
            return vt + b"\x04" + struct.pack("<i", v.shape[0]) + v.astype(dt).tobytes()

        with open(path, "wb") as f:
            f.write(b"\0B<Plda> " + vec(mu))
            f.write(mt + b"\x04" + struct.pack("<i", transform.shape[0]) + b"\x04" +
                    struct.pack("<i", transform.shape[1]) + transform.astype(dt).tobytes())
            f.write(vec(psi) + b"</Plda> ")
        return
    fmt = lambda row: " ".join(repr(float(x)) for x in row)      # noqa: E731
    with open(path, "w") as f:
        f.write("<Plda>  [ " + fmt(mu) + " ]\n [\n")
        for i, row in enumerate(transform):
            f.write("  " + fmt(row) + (" ]\n" if i == transform.shape[0] - 1 else "\n"))
        f.write(" [ " + fmt(psi) + " ]\n</Plda> ")
