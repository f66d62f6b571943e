"""GPU (-m gpu): the HIP path, called through the C-ABI, against the oracle on the same seeded
inputs and against the committed golden vectors produced by the reference itself.

Bars (BASELINE.json north_star): embeddings within 1e-4 cosine of the reference CPU path, PLDA
LLRs within 1e-3.  The tests additionally hold tighter bounds (relative L2 error) so that a
wrong-but-correlated embedding cannot pass."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import ecapa as oecapa
from oracle import fbank as ofbank
from oracle import plda as oplda
from fixtures import synth

pytestmark = pytest.mark.gpu

COS_TOL = 1e-4          # north_star: 1 - cosine <= 1e-4
REL_TOL = 2e-4          # our own: ||e - e_ref|| / ||e_ref||   (fp32 contraction-order noise is ~1e-6)
LLR_TOL = 1e-3          # north_star; float64 path is far inside (checked at 1e-8 below)


def _cos_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return 1.0 - np.sum(a * b, -1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def _rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)


@pytest.fixture(scope="module")
def frontend():
    from wespeaker_amd.engine import Frontend
    return Frontend(16000, 80)


# ------------------------------------------------------------------------------------- fbank
@pytest.mark.parametrize("window", ["hamming", "povey"])
def test_fbank_matches_oracle(frontend, window):
    wav = synth.synth_wav_batch(0, 5)
    got = frontend.fbank(torch.from_numpy(wav), window_type=window, cmn=False).cpu().numpy()
    ref = np.stack([ofbank.kaldi_fbank(w.astype(np.float32), window_type=window) for w in wav])
    assert got.shape == ref.shape == (5, 198, 80)
    assert np.abs(got - ref).max() < 2e-3            # log-mel values span ~8..26
    assert np.abs(got - ref).mean() < 2e-5
    got_c = frontend.fbank(torch.from_numpy(wav), window_type=window, cmn=True).cpu().numpy()
    ref_c = ref - ref.mean(1, keepdims=True)
    assert np.abs(got_c - ref_c).max() < 2e-3


@pytest.mark.parametrize("bins", [23, 24, 40, 64, 128])
def test_fbank_other_mel_bin_counts(bins):
    """Kaldi's own default (23) and the common 40 / 64: triangular filters longer than the mel phase's 48 register taps
    take its tail loop (round 3 rejected such frontends: ADVICE r3)."""
    from wespeaker_amd.engine import Frontend
    fe = Frontend(16000, bins)
    wav = synth.synth_wav_batch(3, 3)
    got = fe.fbank(torch.from_numpy(wav), cmn=False).cpu().numpy()
    ref = np.stack([ofbank.kaldi_fbank(w.astype(np.float32), num_mel_bins=bins, window_type="hamming") for w in wav])
    assert got.shape == ref.shape == (3, 198, bins)
    assert np.abs(got - ref).max() < 2e-3 and np.abs(got - ref).mean() < 2e-5


def test_fbank_matches_reference_native_golden(frontend, golden_dir):
    g = np.load(os.path.join(golden_dir, "fbank_ref_native.npz"))
    wav = np.stack([synth.synth_wav(int(i)) for i in g["utt_idx"]])
    got = frontend.fbank(torch.from_numpy(wav), cmn=False).cpu().numpy()
    assert np.abs(got - g["logmel"]).max() < 5e-4     # reference-owned native arithmetic


def test_fbank_edge_cases(frontend):
    # ragged / minimal lengths, float input, strided rows, silence
    for n in (400, 559, 560, 4000, 16000 * 5):
        wav = synth.synth_wav(11, n)
        got = frontend.fbank(torch.from_numpy(wav)[None], cmn=False).cpu().numpy()[0]
        ref = ofbank.kaldi_fbank(wav.astype(np.float32), window_type="hamming")
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 2e-3
    assert frontend.fbank(torch.zeros(2, 399, dtype=torch.int16)).shape == (2, 0, 80)
    z = frontend.fbank(torch.zeros(1, 1000, dtype=torch.int16), cmn=False).cpu().numpy()
    assert np.allclose(z, np.log(np.float32(1.1920929e-07)))
    wav = synth.synth_wav_batch(20, 3)
    f32 = frontend.fbank(torch.from_numpy(wav.astype(np.float32)), cmn=False).cpu().numpy()
    i16 = frontend.fbank(torch.from_numpy(wav), cmn=False).cpu().numpy()
    assert np.array_equal(f32, i16)
    # [-1, 1] floats with scale 32768 (processor.py:516)
    sc = frontend.fbank(torch.from_numpy(wav.astype(np.float32) / 32768.0), cmn=False,
                        scale=32768.0).cpu().numpy()
    assert np.abs(sc - i16).max() < 1e-3
    # max values: full-scale square wave must not overflow / NaN
    sq = np.where(np.arange(32000) % 64 < 32, 32767, -32768).astype(np.int16)
    out = frontend.fbank(torch.from_numpy(sq)[None], cmn=False).cpu().numpy()
    assert np.isfinite(out).all()
    assert np.abs(out[0] - ofbank.kaldi_fbank(sq.astype(np.float32), window_type="hamming")).max() < 2e-3


# ------------------------------------------------------------------ fbank at other sample rates
def test_fbank_other_sample_rates_match_reference_native_and_oracle(golden_dir):
    """VERDICT r4 missing #1: the reference passes `sample_frequency=sample_rate` straight through
    (wespeaker/cli/speaker.py:90-97), its native twin sizes the FFT from the frame (runtime/core/frontend/fbank.h:33-52)
    and its PLDA recipe runs at 8 kHz (examples/sre/v2/conf/resnet.yaml:31).  8 kHz (256 points; 80 / 23 / 40 bins),
    32 kHz (1024) and 48 kHz (2048) against the reference's own native fbank (golden) and the oracle."""
    from wespeaker_amd.engine import Frontend
    g = np.load(os.path.join(golden_dir, "fbank_ref_native_rates.npz"))
    assert len(g.files) == 5
    for key in g.files:
        rate, bins, n = (int(t[1:]) for t in key.split("_"))
        fe = Frontend(rate, bins)
        wav = synth.synth_wav(21 + rate // 8000, n)
        got = fe.fbank(torch.from_numpy(wav)[None], cmn=False).cpu().numpy()[0]
        assert got.shape == g[key].shape
        assert np.abs(got - g[key]).max() < 5e-4, key                     # reference-owned native arithmetic
        for window in ("hamming", "povey"):
            got_w = fe.fbank(torch.from_numpy(wav)[None], window_type=window, cmn=True).cpu().numpy()[0]
            ref = ofbank.kaldi_fbank(wav.astype(np.float32), num_mel_bins=bins, sample_frequency=rate,
                                     window_type=window, cmn=True)
            assert np.abs(got_w - ref).max() < 2e-3 and np.abs(got_w - ref).mean() < 2e-5, (key, window)


@pytest.mark.parametrize("rate", [4000, 8000, 11025, 22050, 24000, 44100])
def test_fbank_any_rate_edge_lengths_and_batches(rate):
    """Rates on both sides of the 512-point class (11025 Hz: 275 samples -> 512 points, the specialised kernel with
    another frame length; 4000: 128; 22050 / 24000: 1024; 44100: 2048), shortest inputs, batches, float input."""
    from wespeaker_amd.engine import Frontend
    fe = Frontend(rate, 80)
    flen, fshift = int(rate * 0.025), int(rate * 0.010)
    assert fe.fbank(torch.zeros(2, flen - 1, dtype=torch.int16)).shape == (2, 0, 80)
    for n in (flen, flen + fshift - 1, flen + fshift, 2 * rate + 17):
        wav = synth.synth_wav_batch(31, 3, n)
        got = fe.fbank(torch.from_numpy(wav), cmn=False).cpu().numpy()
        ref = np.stack([ofbank.kaldi_fbank(w.astype(np.float32), sample_frequency=rate, window_type="hamming")
                        for w in wav])
        assert got.shape == ref.shape and got.shape[1] == fe.num_frames(n)
        assert np.abs(got - ref).max() < 2e-3 and np.abs(got - ref).mean() < 3e-5, (rate, n)
    wav = synth.synth_wav_batch(32, 2, rate)
    f32 = fe.fbank(torch.from_numpy(wav.astype(np.float32)), cmn=False).cpu().numpy()
    assert np.array_equal(f32, fe.fbank(torch.from_numpy(wav), cmn=False).cpu().numpy())
    z = fe.fbank(torch.zeros(1, rate, dtype=torch.int16), cmn=False).cpu().numpy()
    assert np.allclose(z, np.log(np.float32(1.1920929e-07)))


def test_fbank_any_length_kernel_has_the_specialised_kernels_bits_at_16k(frontend):
    """The any-length kernel (every rate outside the 512-point class) run on 16 kHz input through ws_debug_fbank_mode(2):
    the same operations in the same order as the specialised kernel -> the same bits, for both windows, with and
    without CMN, and for other mel-bin counts (whose filters have other tap counts per lane)."""
    from wespeaker_amd import _lib
    from wespeaker_amd.engine import Frontend
    wav = torch.from_numpy(synth.synth_wav_batch(40, 6))
    for window in ("hamming", "povey"):
        fast = frontend.fbank(wav, window_type=window, cmn=False).clone()
        fast_c = frontend.fbank(wav, window_type=window, cmn=True).clone()
        try:
            assert _lib.lib().ws_debug_fbank_mode(2) == 0
            anyk = frontend.fbank(wav, window_type=window, cmn=False).clone()
            anyk_c = frontend.fbank(wav, window_type=window, cmn=True).clone()
        finally:
            assert _lib.lib().ws_debug_fbank_mode(0) == 0
        assert torch.equal(fast, anyk) and torch.equal(fast_c, anyk_c)
    for bins in (23, 40, 128):
        fe = Frontend(16000, bins)
        a = fe.fbank(wav, cmn=False).clone()
        try:
            _lib.lib().ws_debug_fbank_mode(2)
            b = fe.fbank(wav, cmn=False).clone()
        finally:
            _lib.lib().ws_debug_fbank_mode(0)
        assert torch.equal(a, b), bins
    assert _lib.lib().ws_debug_fbank_mode(3) != 0


def test_speaker_api_at_8khz_resample_rate(tmp_path):
    """`Speaker.set_resample_rate(8000)` (cli/speaker.py:66-67; the SRE recipe's rate): 16 kHz files are resampled to
    8 kHz, 8 kHz files pass through, features come from the 256-point frontend; checked against the oracle chain
    resample -> fbank(8 kHz) -> CMN -> ECAPA."""
    import wespeaker_amd as wespeaker
    from oracle import resample as oresample
    mdir = str(tmp_path / "model")
    sd = synth.write_model_dir(mdir, "ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    spk = wespeaker.load_model(mdir)
    spk.set_resample_rate(8000)
    p16, p8 = str(tmp_path / "a16.wav"), str(tmp_path / "a8.wav")
    w16, w8 = synth.synth_wav(300, 32000), synth.synth_wav(301, 16000)
    synth.write_wav(p16, w16, 16000)
    synth.write_wav(p8, w8, 8000)
    e8 = spk.extract_embedding(p8).numpy()
    r8 = oecapa.ecapa_forward(sd, ofbank.speaker_features(w8, sample_rate=8000)[None]).numpy()[0]
    assert _cos_err(e8, r8) < COS_TOL and _rel_err(e8, r8) < 5e-4
    e16 = spk.extract_embedding(p16).numpy()
    x8 = oresample.resample(w16.astype(np.float32), 16000, 8000)
    r16 = oecapa.ecapa_forward(sd, ofbank.speaker_features(x8, sample_rate=8000)[None]).numpy()[0]
    assert _cos_err(e16, r16) < COS_TOL and _rel_err(e16, r16) < 2e-3
    scp = tmp_path / "wav.scp"
    scp.write_text("u8 %s\nu16 %s\n" % (p8, p16))
    names, embs = spk.extract_embedding_list(str(scp))
    assert names == ["u8", "u16"]
    assert _rel_err(embs[0], e8) < 1e-5 and _rel_err(embs[1], e16) < 1e-5


# ------------------------------------------------------------------------------------- ECAPA
def _engine(name, embed_dim=192, seed=42, **kw):
    from wespeaker_amd.engine import NativeSpeakerModel
    sd = synth.synth_ecapa_state_dict(name, 80, embed_dim, seed=seed, **kw)
    return sd, NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=embed_dim, max_batch=8,
                                  max_frames=400)


@pytest.mark.parametrize("name", ["ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_c512",
                                  "ECAPA_TDNN_GLOB_c1024", "ECAPA_TDNN_c1024"])
def test_ecapa_forward_matches_oracle_and_golden(name, golden_dir):
    sd, model = _engine(name)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(3)])
    got = model(torch.from_numpy(feats))[-1].cpu().numpy()
    ref = oecapa.ecapa_forward(sd, feats).numpy()
    assert got.shape == ref.shape == (3, 192)
    assert _cos_err(got, ref).max() < COS_TOL
    assert _rel_err(got, ref).max() < REL_TOL
    g = np.load(os.path.join(golden_dir, "ecapa_ref.npz"))       # reference nn.Module outputs
    assert _cos_err(got[:2], g[name + "/emb"]).max() < COS_TOL
    assert _rel_err(got[:2], g[name + "/emb"]).max() < REL_TOL
    got_s = model(torch.from_numpy(feats[:2, :57].copy()))[-1].cpu().numpy()
    assert _rel_err(got_s, g[name + "/emb_T57"]).max() < REL_TOL


def _cancelling_channel_fixture(seed=7, c=5):
    """An ECAPA-GLOB weight set in which channel `c` of h = ReLU(conv(cat)) sits at ~200 with a spread of a few 1e-2
    (|mean| / std ~ 1e3..1e4), its context STD drives the attention strongly, and its pooled output std (which the
    reference itself computes by the cancelling form, pooling_layers.py:141-143) is kept out of the embedding."""
    sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=seed)
    sd = {k: np.array(v, copy=True) for k, v in sd.items()}
    sd["conv.weight"][c] *= 1e-3
    sd["conv.bias"][c] = 200.0
    sd["pool.linear1.weight"][:, 3072 + c, 0] = 3.0 * np.sign(sd["pool.linear1.weight"][:, 3072 + c, 0] + 1e-9)
    sd["linear.weight"][:, 1536 + c] = 0.0
    return sd


def test_ecapa_glob_context_std_two_pass_fallback():
    """VERDICT r4 missing #4 / ADVICE r3: astp_std_from_sums takes mean and std of the global context from the GEMM
    epilogue's column sums and sums of squares; when S2 - T mean^2 has cancelled (std below ~3 % of the mean) that
    (utterance, channel) is recomputed in two passes like torch.var (csrc/ecapa_ops.hip, `cancelled`).  The standard
    fixtures never get there (BN statistics N(0, 0.1) / U(0.5, 1.5)); this one drives a channel with |mean| / std ~ 1e3
    through it, with the context std weighted heavily in the attention, and checks against the oracle.  The test first
    shows that the single-pass fp32 form WOULD be wrong on this channel (so a regression of the fallback cannot hide)."""
    from wespeaker_amd.engine import NativeSpeakerModel
    c = 5
    sd = _cancelling_channel_fixture(c=c)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(50 + i)) for i in range(4)])
    ref, mid = oecapa.ecapa_forward(sd, feats, return_intermediates=True)
    ref = ref.numpy()
    h = mid["h"].numpy()[:, c, :]                                   # (B, T)
    T = h.shape[1]
    two_pass = h.astype(np.float64).var(axis=1, ddof=1)
    assert (np.sqrt(two_pass) / h.mean(1) < 0.03).all()             # the kernel's `cancelled` condition holds
    s1 = np.float32(0); s2 = np.float32(0)
    single = []
    for b in range(h.shape[0]):                                     # fp32 sums like the epilogue leaves them
        s1 = np.add.reduce(h[b], dtype=np.float32); s2 = np.add.reduce(h[b] * h[b], dtype=np.float32)
        m = np.float32(s1 / np.float32(T))
        single.append(max(float(np.float32(s2) - np.float32(T) * m * m), 0.0) / (T - 1))
    assert (np.abs(np.array(single) - two_pass) / two_pass > 0.1).any()   # the single-pass form has lost the variance
    model = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, feat_dim=80, embed_dim=192, max_batch=8, max_frames=400)
    got = model(torch.from_numpy(feats))[-1].cpu().numpy()
    assert np.isfinite(got).all()
    assert _cos_err(got, ref).max() < COS_TOL and _rel_err(got, ref).max() < REL_TOL
    # and it matters: with the channel's context std off by what the single-pass form loses, the oracle moves far
    # beyond the tolerance (the attention column weighs it with |w| = 3)
    sd_bad = dict(sd)
    sd_bad["pool.linear1.bias"] = sd["pool.linear1.bias"] + 3.0 * 0.05 * np.sign(sd["pool.linear1.weight"][:, 3072 + c, 0])
    moved = oecapa.ecapa_forward(sd_bad, feats).numpy()
    assert _rel_err(moved, ref).max() > 10 * REL_TOL


@pytest.mark.parametrize("name", ["ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_c1024"])
def test_ecapa_f16x3_split_precision_matches_oracle(name, golden_dir):
    """The 3-pass split-binary16 MFMA back-end (hi*hi + hi*lo + lo*hi, fp32 accumulate) must meet
    the same bars as the exact fp32 path."""
    sd, model = _engine(name)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(3)])
    exact = model(torch.from_numpy(feats))[-1].cpu().numpy()
    model.set_precision("f16x3")
    got = model(torch.from_numpy(feats))[-1].cpu().numpy()
    ref = oecapa.ecapa_forward(sd, feats).numpy()
    assert _cos_err(got, ref).max() < COS_TOL
    assert _rel_err(got, ref).max() < REL_TOL
    assert _rel_err(got, exact).max() < REL_TOL
    g = np.load(os.path.join(golden_dir, "ecapa_ref.npz"))
    assert _rel_err(got[:2], g[name + "/emb"]).max() < REL_TOL
    for T in (5, 57, 201):
        f = np.random.RandomState(T).randn(2, T, 80).astype(np.float32)
        e = model(torch.from_numpy(f))[-1].cpu().numpy()
        assert _rel_err(e, oecapa.ecapa_forward(sd, f).numpy()).max() < REL_TOL
    # large-magnitude and tiny-magnitude inputs (binary16 range / subnormal lo parts)
    for scale in (1e-3, 30.0):
        f = (np.random.RandomState(7).randn(2, 100, 80) * scale).astype(np.float32)
        e = model(torch.from_numpy(f))[-1].cpu().numpy()
        assert _rel_err(e, oecapa.ecapa_forward(sd, f).numpy()).max() < REL_TOL
    model.set_precision("fp32")
    back = model(torch.from_numpy(feats))[-1].cpu().numpy()
    assert np.array_equal(back, exact)


F16_REL_TOL = 1.5e-3    # binary16 operands (11-bit significand): ~5e-4 measured on the embeddings
F16_REL_TOL_STRESS = 5e-3   # 30x-scaled random features (|x| up to ~120, far outside CMN'd log-mels): 2.0e-3 (c512) ..
                            # 3.3e-3 (c1024) measured
F16_REL_TOL_DEEP = 2.5e-3   # ResNet221 (Bottleneck [6,16,48,3] = 219 binary16-rounded conv outputs in series): 1.6e-3
                            # measured at T = 9


@pytest.mark.parametrize("name", ["ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_c512", "ECAPA_TDNN_GLOB_c1024",
                                  "ECAPA_TDNN_c1024"])
def test_ecapa_f16_backend_meets_the_cosine_bar(name, golden_dir):
    """WS_PREC_F16 (binary16 MFMA operands, fp32 accumulation, binary16 activation copies between
    the 1x1 layers): the north-star bar 1 - cos <= 1e-4 against the oracle AND the reference golden,
    at the golden shapes, ragged lengths (K-tile-64 kernel tails, T < 64 fallbacks) and scales."""
    sd, model = _engine(name)
    model.set_precision("f16")
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(3)])
    got = model(torch.from_numpy(feats))[-1].cpu().numpy()
    ref = oecapa.ecapa_forward(sd, feats).numpy()
    assert _cos_err(got, ref).max() < COS_TOL
    assert _rel_err(got, ref).max() < F16_REL_TOL
    g = np.load(os.path.join(golden_dir, "ecapa_ref.npz"))
    assert _cos_err(got[:2], g[name + "/emb"]).max() < COS_TOL
    assert _rel_err(got[:2], g[name + "/emb"]).max() < F16_REL_TOL
    for T in (5, 57, 64, 201, 333):
        f = np.random.RandomState(T).randn(3, T, 80).astype(np.float32)
        e = model(torch.from_numpy(f))[-1].cpu().numpy()
        r = oecapa.ecapa_forward(sd, f).numpy()
        assert _cos_err(e, r).max() < COS_TOL and _rel_err(e, r).max() < F16_REL_TOL, T
    for scale in (1e-2, 30.0):
        f = (np.random.RandomState(7).randn(2, 100, 80) * scale).astype(np.float32)
        e = model(torch.from_numpy(f))[-1].cpu().numpy()
        r = oecapa.ecapa_forward(sd, f).numpy()
        assert _cos_err(e, r).max() < COS_TOL and _rel_err(e, r).max() < F16_REL_TOL_STRESS, scale
    # batch invariance: row tiles never mix utterances, but the position of an utterance inside the
    # 64-row tiles moves the fp32 summation order of the SE / context statistics by ~1e-7, which a
    # binary16 rounding downstream can turn into one half-ulp (5e-4) on single activations
    one = model(torch.from_numpy(feats[1:2]))[-1].cpu().numpy()
    assert _rel_err(one, got[1:2]).max() < F16_REL_TOL


def test_resnet_and_campplus_f16_backend_meet_the_cosine_bar(golden_dir):
    from oracle import campplus as ocam
    from oracle import resnet as oresnet
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
    sd = synth.synth_resnet_state_dict("ResNet34", 80, 256, seed=42)
    model = _native("ResNet34", sd, 256)
    model.set_precision("f16")
    got = model(torch.from_numpy(feats))[-1].cpu().numpy()
    ref = oresnet.resnet_forward(sd, feats, "ResNet34").numpy()
    assert _cos_err(got, ref).max() < COS_TOL and _rel_err(got, ref).max() < F16_REL_TOL
    g = np.load(os.path.join(golden_dir, "resnet_ref.npz"))
    assert _cos_err(got, g["ResNet34/emb"]).max() < COS_TOL
    sd = synth.synth_campplus_state_dict(80, 512, seed=42)
    model = _native("CAMPPlus", sd, 512, max_batch=4, max_frames=400)
    model.set_precision("f16")
    got = model(torch.from_numpy(feats)).cpu().numpy()
    ref = ocam.campplus_forward(sd, feats).numpy()
    assert _cos_err(got, ref).max() < COS_TOL and _rel_err(got, ref).max() < F16_REL_TOL
    g = np.load(os.path.join(golden_dir, "campplus_ref.npz"))
    assert _cos_err(got, g["emb"]).max() < COS_TOL


@pytest.mark.parametrize("name", ["ResNet18", "ResNet50", "ResNet221"])
def test_resnet_other_depths_f16_backend(name, golden_dir):
    """The f16 back-end on the BasicBlock-18 and the Bottleneck families (ResNet221 = BASELINE config 3's
    "r=221": 48-layer stage 3, 1x1-3x3-1x1 blocks with x4 expansion) against the oracle, the reference
    nn.Module golden, and at sizes that leave partial tiles in every stride-2 stage."""
    from oracle import resnet as oresnet
    sd = synth.synth_resnet_state_dict(name, 80, 256, seed=42)
    model = _native(name, sd, 256)
    model.set_precision("f16")
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
    got = model(torch.from_numpy(feats))[-1].cpu().numpy()
    ref = oresnet.resnet_forward(sd, feats, name).numpy()
    assert _cos_err(got, ref).max() < COS_TOL and _rel_err(got, ref).max() < F16_REL_TOL
    g = np.load(os.path.join(golden_dir, "resnet_ref.npz"))
    assert _cos_err(got, g[name + "/emb"]).max() < COS_TOL
    assert _rel_err(got, g[name + "/emb"]).max() < F16_REL_TOL
    got_s = model(torch.from_numpy(feats[:, :57].copy()))[-1].cpu().numpy()
    assert _cos_err(got_s, g[name + "/emb_T57"]).max() < COS_TOL
    for T in (9, 131):
        f = np.random.RandomState(T).randn(2, T, 80).astype(np.float32)
        e = model(torch.from_numpy(f))[-1].cpu().numpy()
        r = oresnet.resnet_forward(sd, f, name).numpy()
        tol = F16_REL_TOL_DEEP if name == "ResNet221" else F16_REL_TOL
        assert _cos_err(e, r).max() < COS_TOL and _rel_err(e, r).max() < tol, T
    model.check_range()


def test_campplus_f16_backend_multi_segment_context(golden_dir):
    """CAM++ in f16 with more than one 100-frame context segment (T = 328 -> 2 segments, golden from the
    reference module; T = 603 -> 4 segments with a 2-frame ceil_mode tail; T = 7 single short one)."""
    from oracle import campplus as ocam
    sd = synth.synth_campplus_state_dict(80, 512, seed=42)
    model = _native("CAMPPlus", sd, 512, max_batch=4, max_frames=700)
    model.set_precision("f16")
    g = np.load(os.path.join(golden_dir, "campplus_ref.npz"))
    long_feats = np.stack([ofbank.speaker_features(synth.synth_wav(i, 52800)) for i in range(2)])
    got_l = model(torch.from_numpy(long_feats)).cpu().numpy()
    assert _cos_err(got_l, g["emb_T328"]).max() < COS_TOL
    assert _rel_err(got_l, g["emb_T328"]).max() < F16_REL_TOL
    for T in (7, 201, 399, 603):
        f = np.random.RandomState(T).randn(2, T, 80).astype(np.float32)
        e = model(torch.from_numpy(f)).cpu().numpy()
        r = ocam.campplus_forward(sd, f).numpy()
        assert _cos_err(e, r).max() < COS_TOL and _rel_err(e, r).max() < F16_REL_TOL, T
    model.check_range()


def test_binary16_range_guard_never_silently_wrong():
    """Adversarial ranges for the binary16 back-ends (include/wespeaker_amd.h: activations must stay
    below 65504).  (1) features of magnitude 1e5: fp32 still matches the oracle, f16 and f16x3 produce
    inf and ws_engine_check_range reports it (WS_ERR_RANGE) instead of returning NaN quietly.
    (2) checkpoints whose BN running_var spans 1e-2..1e2 (x0.1..x10 per layer): every back-end either
    meets the bar or raises -- never a finite-looking wrong embedding."""
    from wespeaker_amd._lib import NativeError
    sd, model = _engine("ECAPA_TDNN_GLOB_c512")
    huge = (np.random.RandomState(1).randn(2, 120, 80) * 1e5).astype(np.float32)
    ref = oecapa.ecapa_forward(sd, huge).numpy()
    assert np.isfinite(ref).all()
    assert _rel_err(model(torch.from_numpy(huge))[-1].cpu().numpy(), ref).max() < REL_TOL
    model.check_range()                                   # fp32: nothing to report
    for mode in ("f16", "f16x3"):
        model.set_precision(mode)
        out = model(torch.from_numpy(huge))[-1]
        with pytest.raises(NativeError, match="binary16 range"):
            model.check_range()
        assert not bool(torch.isfinite(out).all())
        model.check_range()                               # the counter was cleared by the report
        ok = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
        model(torch.from_numpy(ok))
        model.check_range()                               # in-range input: clean
    rng = np.random.Generator(np.random.PCG64(77))
    raised = 0
    for trial in range(3):
        sd2 = dict(synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=60 + trial))
        for k in list(sd2):
            if k.endswith("running_var"):
                sd2[k] = (10.0 ** rng.uniform(-2, 2, sd2[k].shape)).astype(np.float32)
        from wespeaker_amd.engine import NativeSpeakerModel
        m2 = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd2, feat_dim=80, embed_dim=192, max_batch=4,
                                max_frames=200)
        feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
        ref2 = oecapa.ecapa_forward(sd2, feats).numpy()
        assert np.isfinite(ref2).all()
        assert _rel_err(m2(torch.from_numpy(feats))[-1].cpu().numpy(), ref2).max() < REL_TOL
        for mode in ("f16x3", "f16"):
            m2.set_precision(mode)
            e = m2(torch.from_numpy(feats))[-1].cpu().numpy()
            try:
                m2.check_range()
            except NativeError:
                raised += 1
                continue
            assert _cos_err(e, ref2).max() < COS_TOL, (trial, mode)
    print("range guard: %d of 6 wide-range runs reported binary16 overflow" % raised)


def test_ecapa_emb_bn_and_shapes(golden_dir):
    sd, model = _engine("ECAPA_TDNN_c512", embed_dim=256, seed=5, emb_bn=True)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
    got = model(torch.from_numpy(feats))[-1].cpu().numpy()
    g = np.load(os.path.join(golden_dir, "ecapa_ref.npz"))
    assert _rel_err(got, g["ECAPA_TDNN_c512_embbn/emb"]).max() < REL_TOL


def test_ecapa_batch_invariance_chunking_and_ragged_lengths():
    """Utterances are independent: any batch split / chunking gives the same rows; odd lengths
    (T not a multiple of any tile) work; B larger than the engine's chunk is processed in chunks."""
    sd, model = _engine("ECAPA_TDNN_GLOB_c512")
    wav = synth.synth_wav_batch(30, 19)
    feats = np.stack([ofbank.speaker_features(w) for w in wav])
    full = model(torch.from_numpy(feats))[-1].cpu().numpy()            # 19 > max_batch 8 -> 3 chunks
    one = np.concatenate([model(torch.from_numpy(feats[i:i + 1]))[-1].cpu().numpy()
                          for i in range(19)])
    assert _rel_err(full, one).max() < 1e-5
    ref = oecapa.ecapa_forward(sd, feats).numpy()
    assert _rel_err(full, ref).max() < REL_TOL
    for T in (5, 33, 129, 201, 399):
        f = np.random.RandomState(T).randn(2, T, 80).astype(np.float32)
        got = model(torch.from_numpy(f))[-1].cpu().numpy()
        assert _rel_err(got, oecapa.ecapa_forward(sd, f).numpy()).max() < REL_TOL
    # raw C-ABI: over the finalized capacity is a loud WS_ERR_CAPACITY ...
    from wespeaker_amd import _lib
    big = torch.zeros(1, model.max_frames + 1, 80, device="cuda")
    out = torch.empty(1, 192, device="cuda")
    assert _lib.lib().ws_forward(model._h, _lib.ptr(big), 1, model.max_frames + 1, _lib.ptr(out), None) == -7
    # ... and the Python wrapper re-sizes the workspace instead (the reference has no length cap,
    # cli/speaker.py:125-167): a 8 s utterance through an engine finalized for 4 s
    f = np.random.RandomState(7).randn(2, 801, 80).astype(np.float32)
    got = model(torch.from_numpy(f))[-1].cpu().numpy()
    assert model.max_frames >= 801
    assert _rel_err(got, oecapa.ecapa_forward(sd, f).numpy()).max() < REL_TOL
    got = model(torch.from_numpy(feats[:3]))[-1].cpu().numpy()           # still fine after the re-size
    assert _rel_err(got, ref[:3]).max() < REL_TOL
    assert model(torch.zeros(0, 100, 80))[-1].shape == (0, 192)           # empty batch


# ---------------------------------------------------------------------- Speaker API end-to-end
def test_speaker_api_end_to_end(tmp_path):
    import wespeaker_amd as wespeaker
    mdir = str(tmp_path / "model")
    sd = synth.write_model_dir(mdir, "ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    with pytest.raises(FileNotFoundError):
        wespeaker.load_model_pt(str(tmp_path))
    spk = wespeaker.load_model(mdir)
    assert spk.model.frontend_type == "fbank"
    scp = tmp_path / "wav.scp"
    lines = []
    lengths = [32000, 32000, 24000, 32000, 16000]
    for i, n in enumerate(lengths):
        p = str(tmp_path / ("u%d.wav" % i))
        synth.write_wav(p, synth.synth_wav(100 + i, n))
        lines.append("utt%d %s" % (i, p))
    scp.write_text("\n".join(lines) + "\n")
    names, embs = spk.extract_embedding_list(str(scp))
    assert names == ["utt%d" % i for i in range(5)]
    ref = [oecapa.ecapa_forward(sd, ofbank.speaker_features(synth.synth_wav(100 + i, n))[None]).numpy()[0]
           for i, n in enumerate(lengths)]
    for e, r in zip(embs, ref):
        assert isinstance(e, np.ndarray) and e.shape == (192,)
        assert _cos_err(e, r) < COS_TOL and _rel_err(e, r) < 5e-4
    e0 = spk.extract_embedding(str(tmp_path / "u0.wav"))
    assert isinstance(e0, torch.Tensor) and e0.device.type == "cpu"
    assert _rel_err(e0.numpy(), embs[0]) < 1e-5
    s = spk.compute_similarity(str(tmp_path / "u0.wav"), str(tmp_path / "u0.wav"))
    assert abs(s - 1.0) < 1e-5
    spk.set_window_type("povey")
    ep = spk.extract_embedding(str(tmp_path / "u0.wav")).numpy()
    rp = oecapa.ecapa_forward(sd, ofbank.speaker_features(synth.synth_wav(100), window_type="povey")[None]).numpy()[0]
    assert _rel_err(ep, rp) < 5e-4
    spk.set_window_type("hamming")
    fb = [ofbank.speaker_features(synth.synth_wav(100 + i, 24000 + 400), cmn=False)[:150] for i in range(3)]
    ef = spk.extract_embedding_from_feats(fb, batch_size=2, subseg_cmn=True)
    rf = oecapa.ecapa_forward(sd, np.stack(fb) - np.stack(fb).mean(1, keepdims=True)).numpy()
    assert _rel_err(ef, rf).max() < REL_TOL


def test_extract_windows_matches_reference_subsegment_rule(golden_dir, tmp_path):
    """VERDICT r4 missing #3: one speech segment -> diarization sub-segment embeddings in ONE device call
    (ws_extract_windows).  The window layout is the reference's own subsegment() (golden row maps made by running
    diar/extract_emb.py:55-83 on row-index features); features and forward from the oracle."""
    import wespeaker_amd as wespeaker
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    g = np.load(os.path.join(golden_dir, "subsegment_ref.npz"))
    name = "ECAPA_TDNN_GLOB_c512"
    sd = synth.synth_ecapa_state_dict(name, 80, 192, seed=42)
    model = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=192, max_batch=8, max_frames=200)
    fe = Frontend(16000, 80)
    for k in range(7):                                              # (the 15 s case below, through the Speaker API)
        nf, seg_len, win, per = (int(v) for v in g["case%d_params" % k])
        rows = g["case%d_rows" % k]
        wav = synth.synth_wav(400 + k, 400 + 160 * (nf - 1))
        fb = ofbank.speaker_features(wav, cmn=False)
        assert fb.shape[0] == nf
        for cmn in (True, False):
            wins = fb[rows]                                         # (n_windows, window, 80)
            if cmn:
                wins = wins - wins.mean(1, keepdims=True)
            ref = oecapa.ecapa_forward(sd, wins).numpy()
            got = model.extract_windows(fe, torch.from_numpy(wav), seg_length=seg_len, window_frames=win,
                                        period_frames=per, subseg_cmn=cmn).cpu().numpy()
            assert got.shape == ref.shape
            assert _cos_err(got, ref).max() < COS_TOL and _rel_err(got, ref).max() < REL_TOL, (k, cmn)
    # default seg_length = num_frames + 2 (the reference's own VAD segments)
    wav = synth.synth_wav(411, 400 + 160 * 497)
    a = model.extract_windows(fe, torch.from_numpy(wav))
    b = model.extract_windows(fe, torch.from_numpy(wav), seg_length=500)
    assert torch.equal(a, b) and a.shape == (6, 192)
    with pytest.raises(ValueError):
        model.extract_windows(fe, torch.zeros(100, dtype=torch.int16))
    # Speaker API: names + embeddings of a 15 s segment; extract_embedding_from_feats on the same windows (device CMN)
    mdir = str(tmp_path / "model")
    sd2 = synth.write_model_dir(mdir, name, 80, 192, seed=42)
    spk = wespeaker.load_model(mdir)
    nf, seg_len, win, per = (int(v) for v in g["case7_params"])
    wav = synth.synth_wav(420, 400 + 160 * (nf - 1))
    names, embs = spk.extract_subsegment_embeddings(torch.from_numpy(wav), 16000, begin_ms=1230, end_ms=1230 + seg_len * 10)
    assert names == [str(x) for x in g["case7_names"]] and embs.shape == (19, 192)
    fb = ofbank.speaker_features(wav, cmn=False)
    wins = fb[g["case7_rows"]]
    ref = oecapa.ecapa_forward(sd2, wins - wins.mean(1, keepdims=True)).numpy()
    assert _cos_err(embs, ref).max() < COS_TOL and _rel_err(embs, ref).max() < REL_TOL
    from wespeaker_amd.speaker import subsegment
    feats_dev = spk.compute_features(torch.from_numpy(wav)[None].to(torch.float), cmn=False)[0]
    names2, host_wins = subsegment(feats_dev.cpu().numpy(), "{:08d}-{:08d}".format(1230, 1230 + seg_len * 10), win, per, 10)
    assert names2 == names
    via_feats = spk.extract_embedding_from_feats(host_wins, batch_size=8, subseg_cmn=True)
    assert _rel_err(via_feats, embs).max() < 1e-5
    via_tensor = spk.extract_embedding_from_feats(torch.from_numpy(np.stack(host_wins)), batch_size=32, subseg_cmn=True)
    assert _rel_err(via_tensor, embs).max() < 1e-5
    no_cmn = spk.extract_embedding_from_feats(host_wins, batch_size=5, subseg_cmn=False)
    assert _rel_err(no_cmn, oecapa.ecapa_forward(sd2, wins).numpy()).max() < REL_TOL


# -------------------------------------------------------------------------------------- PLDA
@pytest.mark.parametrize("normalize_length", [False, True])
def test_plda_matches_oracle_and_reference_golden(normalize_length, golden_dir):
    from wespeaker_amd import TwoCovPLDA
    p = synth.synth_plda(192, seed=7, normalize_length=normalize_length)
    plda = TwoCovPLDA.from_params(p["mu"], p["transform"], p["psi"], p["offset"], normalize_length)
    emb, _ = synth.synth_embeddings(40, 192, seed=11)
    g = np.load(os.path.join(golden_dir, "plda_ref.npz"))
    tag = "nl%d" % int(normalize_length)
    tr = plda.transform_rows(emb.astype(np.float64)).cpu().numpy()
    assert np.abs(tr - g[tag + "/transformed"]).max() < 1e-10
    one = plda.transform_embedding(emb[3].astype(np.float64))
    assert np.abs(one - g[tag + "/transformed"][3]).max() < 1e-10
    for n in (1, 3):
        ref = g["%s/llr_n%d" % (tag, n)]
        mat = plda.llr_matrix(tr[:20], np.full(20, n, np.int32), tr[20:]).cpu().numpy()
        assert np.abs(mat - ref).max() < 1e-8 < LLR_TOL
        s = plda.log_likelihood_ratio(tr[2], tr[25], n)
        assert abs(s - ref[2, 5]) < 1e-8
    # mixed n per enrollment row, odd sizes, explicit pairs
    E, T = tr[:17], tr[17:40]
    nn = (np.arange(17) % 4 + 1).astype(np.int32)
    mat = plda.llr_matrix(E, nn, T).cpu().numpy()
    ref = np.array([[oplda.log_likelihood_ratio(p, E[i], T[j], nn[i]) for j in range(23)] for i in range(17)])
    assert np.abs(mat - ref).max() < 1e-8
    ie, it = synth.synth_trial_pairs(1000, 17, 23)
    pr = plda.llr_pairs(E, nn, T, ie, it).cpu().numpy()
    assert np.abs(pr - ref[ie, it]).max() < 1e-8
    assert plda.llr_pairs(E, nn, T, ie[:0], it[:0]).shape == (0,)


@pytest.mark.parametrize("normalize_length", [False, True])
def test_plda_large_tables_take_the_gemm_paths(normalize_length):
    """>= 128 vectors: pre-processing runs as rows -> f64 MFMA GEMM -> row norm; uniform and per-model
    session counts take different contraction lengths (D vs 2D).  All must agree with the oracle."""
    from wespeaker_amd import TwoCovPLDA
    p = synth.synth_plda(192, seed=7, normalize_length=normalize_length)
    plda = TwoCovPLDA.from_params(p["mu"], p["transform"], p["psi"], p["offset"], normalize_length)
    emb, _ = synth.synth_embeddings(700, 192, seed=3)
    mean = emb.mean(0).astype(np.float64)
    t_t = plda.prepare_test(emb[:300], mean).cpu().numpy()
    ref_t = np.stack([oplda.prepare_test(p, e, mean) for e in emb[:300]])
    assert np.abs(t_t - ref_t).max() < 1e-9
    offs = np.arange(0, 401, 2)                       # 200 enrollment models of 2 utterances
    e_t = plda.prepare_enroll(emb[300:700], offs, mean).cpu().numpy()
    ref_e = np.stack([oplda.prepare_enroll(p, emb[300 + 2 * i:302 + 2 * i], mean)[0] for i in range(200)])
    assert np.abs(e_t - ref_e).max() < 1e-9
    for n in (1, 2):
        mat = plda.llr_matrix(e_t, n, t_t).cpu().numpy()                    # uniform-n path
        gen = plda.llr_matrix(e_t, torch.full((200,), n, dtype=torch.int32), t_t).cpu().numpy()
        ref = oplda.llr_matrix_vectorised(p, ref_e, np.full(200, n), ref_t)
        assert np.abs(mat - ref).max() < 1e-8 and np.abs(gen - ref).max() < 1e-8
        ie, it = synth.synth_trial_pairs(5000, 200, 300, seed=n)
        pr = plda.llr_pairs(e_t, n, t_t, ie, it).cpu().numpy()
        assert np.abs(pr - ref[ie, it]).max() < 1e-8


def test_score_plda_orders_long_lists_by_enrollment_model():
    """score_plda scores a long list ordered by enrollment model (the pair kernel then reads the enrollment row of
    consecutive trials from L1) and returns the scores in the CALLER's order: a shuffled 20 k-trial list equals the
    same list scored trial by trial through the oracle, position by position."""
    from wespeaker_amd import TwoCovPLDA, score_plda
    p = synth.synth_plda(64, seed=3)
    plda = TwoCovPLDA.from_params(p["mu"], p["transform"], p["psi"], p["offset"], False)
    emb, _ = synth.synth_embeddings(260, 64, seed=9)
    enroll = {"m%03d" % i: [emb[i]] for i in range(60)}
    test = {"t%03d" % i: emb[60 + i] for i in range(200)}
    rs = np.random.RandomState(4)
    trials = [("m%03d" % rs.randint(60), "t%03d" % rs.randint(200)) for _ in range(20000)]
    got = score_plda(plda, enroll, test, trials)
    e_t = {k: oplda.prepare_enroll(p, v, None, True)[0] for k, v in enroll.items()}
    t_t = {k: oplda.prepare_test(p, v, None) for k, v in test.items()}
    for i in range(0, 20000, 97):
        m, t = trials[i]
        assert abs(got[i] - oplda.log_likelihood_ratio(p, e_t[m], t_t[t], 1)) < 1e-8, i
    again = score_plda(plda, enroll, test, sorted(trials))
    lookup = {}
    for (m, t), v in zip(sorted(trials), again):
        lookup[(m, t)] = v
    assert all(got[i] == lookup[trials[i]] for i in range(0, 20000, 13))       # same kernel, same bits per trial


@pytest.mark.parametrize("normalize_length,multisession_avg", [(False, True), (True, False)])
def test_score_plda_and_eval_sv_files(tmp_path, normalize_length, multisession_avg):
    from wespeaker_amd import TwoCovPLDA, kaldi_io, score_plda
    p = synth.synth_plda(64, seed=3, normalize_length=normalize_length)
    plda = TwoCovPLDA.from_params(p["mu"], p["transform"], p["psi"], p["offset"], normalize_length)
    emb, spk = synth.synth_embeddings(60, 64, seed=5, num_speakers=6)
    enroll = {}
    for i in range(30):
        enroll.setdefault("spk%d" % spk[i], []).append(emb[i])
    test = {"t%d" % i: emb[30 + i] for i in range(30)}
    trials = [(m, t, "target") for m in enroll for t in list(test)[::3]]
    indomain = emb.mean(0).astype(np.float64)
    got = score_plda(plda, enroll, test, trials, multisession_avg=multisession_avg,
                     indomain_mean=indomain)
    e_t, n_e = {}, {}
    for k, v in enroll.items():
        e_t[k], n_e[k] = oplda.prepare_enroll(p, v, indomain, multisession_avg)
    t_t = {k: oplda.prepare_test(p, v, indomain) for k, v in test.items()}
    ref = np.array([oplda.log_likelihood_ratio(p, e_t[m], t_t[t], n_e[m]) for m, t, _ in trials])
    assert np.abs(got - ref).max() < 1e-8
    # file-level eval_sv with Kaldi ark/scp I/O
    with kaldi_io.VectorWriter(str(tmp_path / "e.ark"), str(tmp_path / "e.scp")) as w:
        utt2spk = []
        for k, vs in enroll.items():
            for j, v in enumerate(vs):
                w("%s_u%d" % (k, j), v)
                utt2spk.append("%s_u%d %s" % (k, j, k))
    (tmp_path / "utt2spk").write_text("\n".join(utt2spk) + "\n")
    with kaldi_io.VectorWriter(str(tmp_path / "t.ark"), str(tmp_path / "t.scp")) as w:
        for k, v in test.items():
            w(k, v)
    (tmp_path / "trials").write_text("".join("%s %s %s\n" % t for t in trials))
    plda.save_model(str(tmp_path / "plda.npz"))
    plda2 = TwoCovPLDA.load_model(str(tmp_path / "plda.npz"))
    assert plda2.normalize_length == normalize_length
    with kaldi_io.VectorWriter(str(tmp_path / "i.ark"), str(tmp_path / "i.scp")) as w:
        for i in range(60):
            w("i%d" % i, emb[i])
    plda2.eval_sv(str(tmp_path / "e.scp"), str(tmp_path / "utt2spk"), str(tmp_path / "t.scp"),
                  str(tmp_path / "trials"), str(tmp_path / "scores"),
                  multisession_avg=multisession_avg, indomain_scp=str(tmp_path / "i.scp"))
    lines = (tmp_path / "scores").read_text().strip().split("\n")
    assert len(lines) == len(trials)
    for ln, (m, t, lab), r in zip(lines, trials, ref):
        a = ln.split()
        assert a[0] == m and a[1] == t and a[3] == lab
        assert abs(float(a[2]) - r) < 1e-3          # printed with %.5f; indomain mean in f32 ark


# -------------------------------------------------------------------------- ResNet / CAM++
def _native(name, sd, embed_dim, max_batch=4, max_frames=400):
    from wespeaker_amd.engine import NativeSpeakerModel
    return NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=embed_dim, max_batch=max_batch,
                              max_frames=max_frames)


@pytest.mark.parametrize("name,kw", [("ResNet18", {}), ("ResNet34", {}),
                                     ("ResNet34", {"two_emb_layer": True}), ("ResNet50", {}),
                                     ("ResNet221", {})])
def test_resnet_forward_matches_oracle_and_golden(name, kw, golden_dir):
    from oracle import resnet as oresnet
    sd = synth.synth_resnet_state_dict(name, 80, 256, seed=42, **kw)
    model = _native(name, sd, 256)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(3)])
    out = model(torch.from_numpy(feats))
    assert isinstance(out, tuple)
    got = out[-1].cpu().numpy()
    ref = oresnet.resnet_forward(sd, feats, name).numpy()
    assert got.shape == ref.shape == (3, 256)
    assert _cos_err(got, ref).max() < COS_TOL
    assert _rel_err(got, ref).max() < REL_TOL
    g = np.load(os.path.join(golden_dir, "resnet_ref.npz"))
    tag = name + ("_2emb" if kw else "")
    assert _rel_err(got[:2], g[tag + "/emb"]).max() < REL_TOL
    got_s = model(torch.from_numpy(feats[:2, :57].copy()))[-1].cpu().numpy()
    assert _rel_err(got_s, g[tag + "/emb_T57"]).max() < REL_TOL
    for T in (9, 64, 131, 200):              # odd sizes through the three stride-2 stages
        f = np.random.RandomState(T).randn(2, T, 80).astype(np.float32)
        e = model(torch.from_numpy(f))[-1].cpu().numpy()
        assert _rel_err(e, oresnet.resnet_forward(sd, f, name).numpy()).max() < REL_TOL


def test_campplus_forward_matches_oracle_and_golden(golden_dir):
    from oracle import campplus as ocam
    sd = synth.synth_campplus_state_dict(80, 512, seed=42)
    model = _native("CAMPPlus", sd, 512, max_batch=4, max_frames=700)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(3)])
    got = model(torch.from_numpy(feats))
    assert isinstance(got, torch.Tensor)                 # CAM++ returns a bare tensor
    got = got.cpu().numpy()
    ref = ocam.campplus_forward(sd, feats).numpy()
    assert got.shape == ref.shape == (3, 512)
    assert _cos_err(got, ref).max() < COS_TOL
    assert _rel_err(got, ref).max() < REL_TOL
    g = np.load(os.path.join(golden_dir, "campplus_ref.npz"))
    assert _rel_err(got[:2], g["emb"]).max() < REL_TOL
    long_feats = np.stack([ofbank.speaker_features(synth.synth_wav(i, 52800)) for i in range(2)])
    got_l = model(torch.from_numpy(long_feats)).cpu().numpy()
    assert _rel_err(got_l, g["emb_T328"]).max() < REL_TOL        # two context segments
    got_s = model(torch.from_numpy(feats[:2, :57].copy())).cpu().numpy()
    assert _rel_err(got_s, g["emb_T57"]).max() < REL_TOL
    for T in (7, 201, 399, 603):             # 603 -> T' = 302 -> 4 segments, last one 2 frames
        f = np.random.RandomState(T).randn(2, T, 80).astype(np.float32)
        e = model(torch.from_numpy(f)).cpu().numpy()
        assert _rel_err(e, ocam.campplus_forward(sd, f).numpy()).max() < REL_TOL
    # the row-block variants of the one-kernel dense layer (cam_dense.hip): T' = 32 q + r trunk frames run q blocks of
    # 32 rows + a 4-row tail on 4x4x1 MFMAs when r <= 4: T' = 32, 33, 35, 64, 65, 68, 69, 75, 96, 99 (the 2-s
    # utterance), 100 (last frame of segment 0 in the tail), 101 (a one-frame second segment), 128
    for T in (64, 66, 70, 128, 130, 136, 138, 150, 192, 198, 200, 202, 256):
        f = np.random.RandomState(T).randn(3, T, 80).astype(np.float32)
        e = model(torch.from_numpy(f)).cpu().numpy()
        assert _rel_err(e, ocam.campplus_forward(sd, f).numpy()).max() < REL_TOL, T


def test_speaker_api_precision_switch(tmp_path):
    """Speaker.set_precision (extension): all three back-ends through the file-level API agree with
    each other inside the north-star cosine bar."""
    import wespeaker_amd
    d = str(tmp_path / "ecapa")
    synth.write_model_dir(d, "ECAPA_TDNN_GLOB_c512", embed_dim=192, seed=42)
    wav = str(tmp_path / "u.wav")
    synth.write_wav(wav, synth.synth_wav(5))
    spk = wespeaker_amd.load_model(d)
    embs = {}
    for mode in ("fp32", "f16x3", "f16"):
        spk.set_precision(mode)
        embs[mode] = spk.extract_embedding(wav).numpy()
    assert _cos_err(embs["f16x3"][None], embs["fp32"][None]).max() < 1e-6
    assert _cos_err(embs["f16"][None], embs["fp32"][None]).max() < COS_TOL
    with pytest.raises(KeyError):
        spk.set_precision("int8")


def test_speaker_api_resnet_and_campplus_model_dirs(tmp_path):
    import wespeaker_amd as wespeaker
    from oracle import campplus as ocam
    from oracle import resnet as oresnet
    wav = synth.synth_wav(200)
    p = str(tmp_path / "a.wav")
    synth.write_wav(p, wav)
    feats = ofbank.speaker_features(wav)[None]
    for name, ed, fwd in (("ResNet34", 256, lambda sd, f: oresnet.resnet_forward(sd, f, "ResNet34")),
                          ("CAMPPlus", 512, ocam.campplus_forward)):
        mdir = str(tmp_path / name)
        sd = synth.write_model_dir(mdir, name, 80, ed, seed=9)
        spk = wespeaker.load_model(mdir)
        e = spk.extract_embedding(p).numpy()
        r = fwd(sd, feats).numpy()[0]
        assert e.shape == (ed,)
        assert _cos_err(e, r) < COS_TOL and _rel_err(e, r) < 5e-4


# ------------------------------------------------ BASELINE-size checks through domain properties
def test_full_size_batch_properties_ecapa():
    """configs[1]/[4] sizes (256 x 2 s): the oracle is too slow for every row, so check
    size-independent properties: utterances are independent (any permutation of the batch permutes
    the embeddings; a row equals its single-utterance result) plus an oracle spot check."""
    from bench import device_wavs
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    model = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, max_batch=256, max_frames=198)
    fe = Frontend(16000, 80)
    wav = device_wavs(256, 32000, model.device, 5)
    perm = torch.randperm(256, device=model.device)
    for prec in ("fp32", "f16x3"):
        model.set_precision(prec)
        full = model.extract(fe, wav)
        shuffled = model.extract(fe, wav[perm])
        # (not bit-equal: the position of an utterance inside the 64-row tiles moves the fp32 summation order
        # of the SE / context column sums by ~1e-7)
        assert _rel_err(shuffled.cpu().numpy(), full[perm].cpu().numpy()).max() < 1e-5
        single = torch.cat([model.extract(fe, wav[i:i + 1]) for i in (0, 131, 255)])
        assert _rel_err(single.cpu().numpy(), full[[0, 131, 255]].cpu().numpy()).max() < 1e-5
        rows = list(range(3, 256, 17))                      # 15 rows spread over every 64-row tile phase
        ref = oecapa.ecapa_forward(sd, np.stack([ofbank.speaker_features(wav[i].cpu().numpy()) for i in rows])).numpy()
        assert _rel_err(full[rows].cpu().numpy(), ref).max() < 5e-4
        assert _cos_err(full[rows].cpu().numpy(), ref).max() < COS_TOL
        assert bool(torch.isfinite(full).all())
    # the headline back-end at the headline size (this is the only size at which the wide layer runs on
    # the phase-staggered 256x256 kernel).  A row's binary16 roundings depend on fp32 summation-order
    # noise of the column-sum tiles it shares, so batch position moves it by ~1e-4, not 0.
    model.set_precision("f16")
    full = model.extract(fe, wav)
    shuffled = model.extract(fe, wav[perm])
    assert _rel_err(shuffled.cpu().numpy(), full[perm].cpu().numpy()).max() < F16_REL_TOL
    assert _cos_err(shuffled.cpu().numpy(), full[perm].cpu().numpy()).max() < 1e-5
    single = torch.cat([model.extract(fe, wav[i:i + 1]) for i in (0, 131, 255)])
    assert _rel_err(single.cpu().numpy(), full[[0, 131, 255]].cpu().numpy()).max() < F16_REL_TOL
    assert _cos_err(full[rows].cpu().numpy(), ref).max() < COS_TOL
    assert _rel_err(full[rows].cpu().numpy(), ref).max() < F16_REL_TOL
    assert torch.equal(model.extract(fe, wav), full)          # same launch sequence -> same bits
    assert bool(torch.isfinite(full).all())


def test_full_size_ecapa1024_f16_spot_check():
    """configs[1] (ECAPA-TDNN-1024, 256 x 2 s) on the headline back-end: its 1024-channel layers all run
    on the phase-staggered 256x256 kernel at this size.  Oracle spot rows + run-to-run bit equality."""
    from bench import device_wavs
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c1024", 80, 192, seed=11)
    model = NativeSpeakerModel("ECAPA_TDNN_GLOB_c1024", sd, max_batch=256, max_frames=198)
    fe = Frontend(16000, 80)
    wav = device_wavs(256, 32000, model.device, 9)
    model.set_precision("f16")
    full = model.extract(fe, wav)
    rows = [0, 31, 77, 128, 129, 200, 254, 255]
    ref = oecapa.ecapa_forward(sd, np.stack([ofbank.speaker_features(wav[i].cpu().numpy()) for i in rows])).numpy()
    assert _cos_err(full[rows].cpu().numpy(), ref).max() < COS_TOL
    assert _rel_err(full[rows].cpu().numpy(), ref).max() < F16_REL_TOL
    assert torch.equal(model.extract(fe, wav), full)
    model.set_precision("f16x3")
    assert _rel_err(model.extract(fe, wav)[rows].cpu().numpy(), ref).max() < 5e-4


def test_ragged_batch_partial_tile_f16():
    """250 x 2 s utterances: 49 500 rows = 193.4 tiles of 256 -- the last tile of every 256x256-kernel launch
    is partial (row clamps in the LDS-DMA, masked stores and column sums in the binary16 epilogue)."""
    from bench import device_wavs
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    model = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, max_batch=250, max_frames=198)
    fe = Frontend(16000, 80)
    wav = device_wavs(250, 32000, model.device, 3)
    model.set_precision("f16")
    full = model.extract(fe, wav)
    rows = [0, 100, 248, 249]
    small = torch.cat([model.extract(fe, wav[i:i + 1]) for i in rows])
    assert _rel_err(full[rows].cpu().numpy(), small.cpu().numpy()).max() < F16_REL_TOL
    assert _cos_err(full[rows].cpu().numpy(), small.cpu().numpy()).max() < 1e-5
    assert bool(torch.isfinite(full).all())


def test_full_size_resnet34_f16_spot_check():
    """configs[2] (ResNet34) at 512 x 2 s on the f16 back-end: the only size at which its 256-wide 3x3
    layers run on the convolution form of the phase-staggered 256x256 kernel (>= 65 536 output pixels);
    also the 64-channel direct kernel and the grouped residual epilogue at full size.  Oracle spot rows,
    equality with a small-batch run of the same utterances (within binary16 rounding), run-to-run bits."""
    from oracle import resnet as oresnet
    from bench import device_wavs
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    sd = synth.synth_resnet_state_dict("ResNet34", 80, 256, seed=3)
    model = NativeSpeakerModel("ResNet34", sd, feat_dim=80, embed_dim=256, max_batch=512, max_frames=198)
    fe = Frontend(16000, 80)
    wav = device_wavs(512, 32000, model.device, 21)
    model.set_precision("f16")
    full = model.extract(fe, wav)
    rows = [1, 64, 255, 300, 448, 511]
    feats = np.stack([ofbank.speaker_features(wav[i].cpu().numpy()) for i in rows])
    ref = oresnet.resnet_forward(sd, feats, "ResNet34").numpy()
    assert _cos_err(full[rows].cpu().numpy(), ref).max() < COS_TOL
    assert _rel_err(full[rows].cpu().numpy(), ref).max() < F16_REL_TOL
    small = model.extract(fe, wav[rows])
    assert _rel_err(small.cpu().numpy(), full[rows].cpu().numpy()).max() < F16_REL_TOL
    assert torch.equal(model.extract(fe, wav), full)
    assert bool(torch.isfinite(full).all())


@pytest.mark.parametrize("name,E,batch,prec", [
    ("ECAPA_TDNN_GLOB_c1024", 192, 256, "fp32"),
    ("ResNet34", 256, 512, "fp32"),
    ("ResNet221", 256, 256, "fp32"), ("ResNet221", 256, 256, "f16"),
    ("CAMPPlus", 512, 512, "fp32"), ("CAMPPlus", 512, 512, "f16"),
])
def test_full_size_every_bench_workload(name, E, batch, prec):
    """Every (workload, back-end) pair bench.py reports that had no full-size check: the BASELINE batch of 2 s
    utterances at the engine chunk bench.py uses -- the shapes at which the dispatcher picks its big-tile / persistent
    kernels (tests/golden/dispatch_* pins WHICH kernel, this pins that it is RIGHT) and at which the activation maps
    come closest to the 2^31-element guards.  Oracle rows spread over the tile phases, equality with a small-batch
    run of the same utterances, run-to-run bits, finiteness of all rows."""
    from oracle import campplus as ocampplus
    from oracle import resnet as oresnet
    from bench import device_wavs
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    sd = synth.synth_state_dict(name, 80, E, seed=13)
    model = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=E, max_batch=batch, max_frames=198)
    fe = Frontend(16000, 80)
    wav = device_wavs(batch, 32000, model.device, 31)
    model.set_precision(prec)
    full = model.extract(fe, wav)
    rows = [0, 1, 63, batch // 2 - 1, batch // 2, batch - 65, batch - 2, batch - 1]
    if name == "ResNet221":
        rows = rows[::2] + [batch - 1]                       # the oracle takes ~1 s per utterance here
    feats = np.stack([ofbank.speaker_features(wav[i].cpu().numpy()) for i in rows])
    sdt = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    if name.startswith("ECAPA"):
        ref = oecapa.ecapa_forward(sdt, feats).numpy()
    elif name.startswith("ResNet"):
        ref = oresnet.resnet_forward(sdt, feats, name).numpy()
    else:
        ref = ocampplus.campplus_forward(sdt, feats).numpy()
    tol = 5e-4 if prec == "fp32" else (F16_REL_TOL_DEEP if name == "ResNet221" else F16_REL_TOL)
    got = full[rows].cpu().numpy()
    assert _cos_err(got, ref).max() < COS_TOL, _cos_err(got, ref).max()
    assert _rel_err(got, ref).max() < tol, _rel_err(got, ref).max()
    small = model.extract(fe, wav[rows])
    assert _rel_err(small.cpu().numpy(), got).max() < (1e-5 if prec == "fp32" else tol)
    assert torch.equal(model.extract(fe, wav), full)          # same launch sequence -> same bits
    assert bool(torch.isfinite(full).all())
    if name != "ResNet221":
        # utterances are independent (VERDICT r5 weak #1b: this property was pinned for ECAPA only): any permutation
        # of the batch permutes the embeddings, and a row equals its single-utterance result -- every one of the
        # `batch` rows is covered by this, not just the oracle's spot rows
        perm = torch.randperm(batch, device=model.device, generator=torch.Generator(model.device).manual_seed(3))
        shuffled = model.extract(fe, wav[perm])
        ptol = 1e-5 if prec == "fp32" else tol
        assert _rel_err(shuffled.cpu().numpy(), full[perm].cpu().numpy()).max() < ptol
        ones = [0, batch // 3, batch - 1]
        single = torch.cat([model.extract(fe, wav[i:i + 1]) for i in ones])
        assert _rel_err(single.cpu().numpy(), full[ones].cpu().numpy()).max() < ptol


@pytest.mark.parametrize("name", ["ResNet101", "ResNet152", "ResNet293"])
@pytest.mark.parametrize("prec", ["fp32", "f16"])
def test_resnet_deep_constructors(name, prec, golden_dir):
    """The remaining constructors of wespeaker/models/resnet.py:231-260 -- Bottleneck [3,4,23,3], [3,8,36,3] and
    [10,20,64,3] (ResNet293's 64-block stage 3 is the longest launch sequence the engine can be asked for) -- on both
    back-ends against the oracle and the reference nn.Module golden: one 2-s case, one T = 57 case, and two odd sizes
    that leave partial tiles in every stride-2 stage."""
    from oracle import resnet as oresnet
    sd = synth.synth_resnet_state_dict(name, 80, 256, seed=42)
    model = _native(name, sd, 256)
    model.set_precision(prec)
    tol = REL_TOL if prec == "fp32" else F16_REL_TOL_DEEP
    g = np.load(os.path.join(golden_dir, "resnet_deep_ref.npz"))
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
    got = model(torch.from_numpy(feats))[-1].cpu().numpy()
    assert got.shape == (2, 256)
    assert _cos_err(got, g[name + "/emb"]).max() < COS_TOL and _rel_err(got, g[name + "/emb"]).max() < tol
    ref = oresnet.resnet_forward(sd, feats, name).numpy()
    assert _cos_err(got, ref).max() < COS_TOL and _rel_err(got, ref).max() < tol
    got_s = model(torch.from_numpy(feats[:, :57].copy()))[-1].cpu().numpy()
    assert _cos_err(got_s, g[name + "/emb_T57"]).max() < COS_TOL and _rel_err(got_s, g[name + "/emb_T57"]).max() < tol
    for T in (9, 131):
        f = np.random.RandomState(T).randn(2, T, 80).astype(np.float32)
        e = model(torch.from_numpy(f))[-1].cpu().numpy()
        r = oresnet.resnet_forward(sd, f, name).numpy()
        assert _cos_err(e, r).max() < COS_TOL and _rel_err(e, r).max() < tol, T
    model.check_range()


def test_cmvn_modes_against_the_reference_apply_cmvn(golden_dir):
    """apply_cmvn(norm_mean, norm_var) (dataset/dataset_utils.py:19-26; bin/extract.py:124-127 with
    test_conf['cmvn'] / ['cmvn_args']): the four settings through ws_cmvn, ws_fbank's `cmn` step with
    ws_frontend_set_cmvn, the ragged form (statistics over an utterance's own frames), and the fused ws_extract --
    against the reference function's own outputs (tests/golden/cmvn_ref.npz) and the reference ECAPA module on them."""
    from wespeaker_amd import _lib
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    g = np.load(os.path.join(golden_dir, "cmvn_ref.npz"))
    fe = Frontend(16000, 80)
    dev = fe.device
    wav = torch.from_numpy(np.stack([synth.synth_wav(i) for i in range(3)]))
    raw = fe.fbank(wav, cmn=False)
    assert np.abs(raw.cpu().numpy() - g["m0v0"]).max() < 2e-3            # (the fbank's own tolerance, un-normalised)
    raw_g = torch.from_numpy(g["m0v0"]).to(dev)
    for nm in (0, 1):
        for nv in (0, 1):
            tag = "m%dv%d" % (nm, nv)
            for suffix, x in (("", raw_g), ("_T57", raw_g[:, :57].contiguous())):
                y = x.clone()
                _lib.check(_lib.lib().ws_cmvn(_lib.ptr(y), y.shape[0], y.shape[1], 80, nm, nv,
                                              _lib.current_stream_ptr(dev)), "ws_cmvn")
                want = g[tag + suffix]
                assert np.abs(y.cpu().numpy() - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (tag, suffix)
                assert np.array_equal(ofbank.apply_cmvn(x.cpu().numpy(), bool(nm), bool(nv)).shape, want.shape)
            fe.set_cmvn(nm, nv)
            got = fe.fbank(wav, cmn=True).cpu().numpy()
            want = ofbank.apply_cmvn(raw.cpu().numpy(), bool(nm), bool(nv))
            assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), tag
            # ragged: utterance 1 keeps 57 frames' worth of samples; its statistics use those frames only
            ns = np.array([32000, 160 * 56 + 400, 32000], np.int32)
            rg = fe.fbank_ragged(wav, ns, cmn=True).cpu().numpy()
            assert np.abs(rg[1, :57] - ofbank.apply_cmvn(raw[1:2, :57].cpu().numpy(), bool(nm), bool(nv))[0]).max() \
                <= 2e-5 * max(1.0, np.abs(want).max()), tag
            assert not rg[1, 57:].any() and np.abs(rg[0] - got[0]).max() == 0
    # mean-only keeps its bits: ws_cmn == ws_cmvn(1, 0)
    a, b = raw_g.clone(), raw_g.clone()
    _lib.check(_lib.lib().ws_cmn(_lib.ptr(a), 3, 198, 80, _lib.current_stream_ptr(dev)), "ws_cmn")
    _lib.check(_lib.lib().ws_cmvn(_lib.ptr(b), 3, 198, 80, 1, 0, _lib.current_stream_ptr(dev)), "ws_cmvn")
    assert torch.equal(a, b)
    # T = 1: torch.var of one sample is NaN (0 / 0) -- the same here, not an exception
    one = raw_g[:, :1].contiguous().clone()
    _lib.check(_lib.lib().ws_cmvn(_lib.ptr(one), 3, 1, 80, 1, 1, _lib.current_stream_ptr(dev)), "ws_cmvn")
    assert bool(torch.isnan(one).all())
    # the fused extract with `cmvn_args: {norm_var: True}` and with `cmvn: False` vs the reference module
    sd = synth.synth_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    model = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, feat_dim=80, embed_dim=192, max_batch=4, max_frames=198)
    for (nm, nv), key in (((1, 1), "ecapa512_m1v1/emb"), ((0, 0), "ecapa512_m0v0/emb")):
        fe.set_cmvn(nm, nv)
        e = model.extract(fe, wav.to(dev)).cpu().numpy()
        assert _cos_err(e, g[key]).max() < COS_TOL and _rel_err(e, g[key]).max() < 5e-4, key
        er = model.extract_ragged(fe, wav.to(dev), np.array([32000, 32000, 32000], np.int32)).cpu().numpy()
        assert _rel_err(er, g[key]).max() < 5e-4, key
    fe.set_cmvn(True, False)


def test_persistent_fp32_gemm_against_the_tile_kernels(tmp_path):
    """gemm_f32_stream.hip (the persistent kernel of the plain 1x1 layers) keeps the tile kernels' k order per
    accumulator, so its GEMM outputs are the tile kernels' bits (tools/stream_probe compares whole layers); only its
    per-tile column sums are folded in another order.  Whole models, with WS_STREAM=0 (tile kernels only), each form
    forced (2: 128x128 tile, 3: 256x128 tile) and the dispatcher's choice: ResNet221 (no column sums) must give the
    same BITS, the ECAPA models (SE means / context statistics from column sums) the same embeddings to 1e-6.
    ECAPA-512 (column sums, dual store), ECAPA-1024 (K = 3072 layer), and a batch whose row count is not a multiple of
    the tile rows."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ab.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from fixtures import synth\n"
        "from bench import device_wavs\n"
        "from wespeaker_amd import Frontend, NativeSpeakerModel\n"
        "dev = torch.device('cuda:0'); fe = Frontend(16000, 80, device=dev); out = {}\n"
        "for name, E, B in (('ECAPA_TDNN_GLOB_c512', 192, 256), ('ECAPA_TDNN_GLOB_c1024', 192, 256),\n"
        "                   ('ResNet221', 256, 64), ('ECAPA_TDNN_c512', 192, 171)):\n"
        "    sd = synth.synth_state_dict(name, 80, E, seed=5)\n"
        "    m = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=E, device=dev, max_batch=B, max_frames=198)\n"
        "    out[name] = m.extract(fe, device_wavs(B, 32000, dev, 17)).cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n" % root)
    res = {}
    for mode in ("0", "2", "3", ""):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("WS_STREAM", None)
        if mode:
            env["WS_STREAM"] = mode
        path = str(tmp_path / ("emb_%s.npz" % (mode or "auto")))
        r = subprocess.run([sys.executable, str(script), path], env=env, cwd=root, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:]
        res[mode] = np.load(path)
    for mode in ("2", "3", ""):
        for k in res["0"].files:
            assert np.isfinite(res[mode][k]).all()
            if k.startswith("ResNet"):
                assert np.array_equal(res["0"][k], res[mode][k]), (mode, k, np.abs(res["0"][k] - res[mode][k]).max())
            else:
                assert _rel_err(res[mode][k], res["0"][k]).max() < 1e-6, (mode, k)


@pytest.mark.parametrize("switch", ["WS_ASTP_FUSED=0", "WS_CHAIN_SMALL=1"])
def test_ab_switches_agree_with_the_shipped_path(tmp_path, switch):
    """The test hooks of INTEGRATION.md section 6 that select another kernel for the same arithmetic (round 6 removed the
    fifteen closed A/B switches; their fallback kernels are reached by shape in the other tests): full-size batches
    (so that the shipped side does take the kernel in question), switched against shipped.  Different summation
    orders only: embeddings agree to 1e-5 relative; the Res2 chain's two forms keep the same k order: the same bits."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    models = {"WS_ASTP_FUSED=0": "(('ECAPA_TDNN_GLOB_c512', 192, 256, 'fp32'), ('ECAPA_TDNN_c512', 192, 256, 'fp32'))",
              "WS_CHAIN_SMALL=1": "(('ECAPA_TDNN_GLOB_c512', 192, 256, 'fp32'), ('ECAPA_TDNN_c1024', 192, 128, 'fp32'))"}[switch]
    script = tmp_path / "ab.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from fixtures import synth\n"
        "from bench import device_wavs\n"
        "from wespeaker_amd import Frontend, NativeSpeakerModel\n"
        "dev = torch.device('cuda:0'); fe = Frontend(16000, 80, device=dev); out = {}\n"
        "for name, E, B, prec in %s:\n"
        "    sd = synth.synth_state_dict(name, 80, E, seed=5)\n"
        "    m = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=E, device=dev, max_batch=B, max_frames=198)\n"
        "    m.set_precision(prec)\n"
        "    out[name] = m.extract(fe, device_wavs(B, 32000, dev, 17)).cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n" % (root, models))
    res = {}
    for tag in ("shipped", "switched"):
        env = dict(os.environ, PYTHONPATH=root)
        for k in ("WS_ASTP_FUSED", "WS_CHAIN_SMALL", "WS_STREAM", "WS_BIG_TILES", "WS_PLDA_BIG_TILES"):
            env.pop(k, None)
        if tag == "switched":
            k, v = switch.split("=")
            env[k] = v
        path = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, str(script), path], env=env, cwd=root, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:]
        res[tag] = np.load(path)
    for k in res["shipped"].files:
        assert np.isfinite(res["switched"][k]).all()
        assert _rel_err(res["switched"][k], res["shipped"][k]).max() < 1e-5, (switch, k)
        if switch == "WS_CHAIN_SMALL=1":          # (the four- and eight-wavefront chain kernels: same k order)
            assert np.array_equal(res["switched"][k], res["shipped"][k]), (switch, k)


def test_lanes_two_batches_in_flight_same_bits():
    """SpeakerModelLanes: batch i on lane i % lanes (own engine, own HIP stream).  Every batch runs the same kernels
    with the same launch parameters as on a single engine, so the embeddings are the single engine's BITS however the
    lanes' kernels interleave on the GPU; results are joined stream-side (wait) or host-side (synchronize)."""
    from bench import device_wavs
    from wespeaker_amd import SpeakerModelLanes
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=11)
    one = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, max_batch=64, max_frames=198)
    lanes = SpeakerModelLanes("ECAPA_TDNN_GLOB_c512", sd, lanes=3, max_batch=64, max_frames=198)
    fe = Frontend(16000, 80)
    wavs = [device_wavs(b, 32000, one.device, 40 + i) for i, b in enumerate((64, 64, 33, 64, 1, 64, 64))]
    for prec in ("fp32", "f16x3", "f16"):     # the binary16 back-ends too (round 3 had to refuse them: DESIGN.md 6.0)
        one.set_precision(prec)
        lanes.set_precision(prec)
        ref = [one.extract(fe, w) for w in wavs]
        pending = [lanes.extract(fe, w) for w in wavs]               # all seven enqueued before anything is joined
        assert [(p.lane - pending[0].lane) % 3 for p in pending] == [0, 1, 2, 0, 1, 2, 0]
        got = [p.wait() for p in pending[:4]] + [p.synchronize() for p in pending[4:]]
        torch.cuda.synchronize()
        for r, g in zip(ref, got):
            assert torch.equal(r, g)
    lanes.check_range()
    # the bench configuration: full batches, two lanes, many steps in flight
    one256 = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, max_batch=256, max_frames=198)
    two = SpeakerModelLanes("ECAPA_TDNN_GLOB_c512", sd, lanes=2, max_batch=256, max_frames=198)
    w256 = [device_wavs(256, 32000, one.device, 70 + i) for i in range(3)]
    ref = [one256.extract(fe, w) for w in w256]
    for rep in range(4):
        pending = [two.extract(fe, w256[i % 3]) for i in range(12)]
        for i, p in enumerate(pending):
            assert torch.equal(p.synchronize(), ref[i % 3]), (rep, i)


@pytest.mark.parametrize("name,E,batch", [("ResNet34", 256, 64), ("ResNet221", 256, 32), ("CAMPPlus", 512, 64),
                                          ("ECAPA_TDNN_GLOB_c1024", 192, 64)])
def test_lanes_same_bits_other_families(name, E, batch):
    """The configurations bench.py reports with two batches in flight (BASELINE configs 1 - 3): on two lanes every
    batch carries the single engine's bits, fp32 and binary16 back-ends."""
    from bench import device_wavs
    from wespeaker_amd import SpeakerModelLanes
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    sd = synth.synth_state_dict(name, 80, E, seed=11)
    one = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=E, max_batch=batch, max_frames=198)
    two = SpeakerModelLanes(name, sd, lanes=2, feat_dim=80, embed_dim=E, max_batch=batch, max_frames=198)
    fe = Frontend(16000, 80)
    wavs = [device_wavs(b, 32000, one.device, 50 + i) for i, b in enumerate((batch, batch, batch // 2 + 1, batch))]
    for prec in ("fp32", "f16"):
        one.set_precision(prec)
        two.set_precision(prec)
        ref = [one.extract(fe, w) for w in wavs]
        for rep in range(3):
            pending = [two.extract(fe, wavs[i % 4]) for i in range(8)]
            for i, p in enumerate(pending):
                assert torch.equal(p.synchronize(), ref[i % 4]), (name, prec, rep, i)
    two.check_range()


def test_fbank_next_to_binary16_engines_keeps_its_bits():
    """DESIGN.md 6.0, the defect of rounds 2 - 3: next to a binary16 engine on another stream the fbank kernel returned
    wrong power-spectrum values in lanes 48..63 (mel bins 35-42 / 56-60 / 68-72 / 78-79; ~70 % of the launches of this
    very loop with the round-3 kernel, tools/fbank_race_probe.py).  Cause: one packed-fp32 instruction form
    (op_sel:[0,1]) in its power-spectrum loop.  The shipped kernel has no packed-fp32 instructions: >= 1000 launches
    next to f16x3 / f16 ECAPA and f16 ResNet34 forwards, plain and ragged, every output the serial bits."""
    from bench import device_wavs
    from wespeaker_amd import _lib
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    dev = torch.device("cuda:0")
    fe, fe2 = Frontend(16000, 80), Frontend(16000, 80)
    w = device_wavs(64, 32000, dev, 40)
    ns = np.full(64, 32000, dtype=np.int32)
    ns[1::3] = 24000
    ref = fe.fbank(w, cmn=False).clone()
    ref_cmn = fe.fbank(w, cmn=True).clone()
    ref_rag = fe.fbank_ragged(w, ns, cmn=True).clone()
    torch.cuda.synchronize()
    feats_p = fe2.fbank(w, cmn=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    launches = 0
    for pname, pE, pprec, reps in (("ECAPA_TDNN_GLOB_c512", 192, "f16x3", 14), ("ECAPA_TDNN_GLOB_c512", 192, "f16", 8),
                                   ("ResNet34", 256, "f16", 5)):
        P = NativeSpeakerModel(pname, synth.synth_state_dict(pname, 80, pE, seed=12), feat_dim=80, embed_dim=pE,
                               max_batch=64, max_frames=198)
        P.set_precision(pprec)
        for rep in range(reps):
            outs = []
            for _ in range(5):
                with torch.cuda.stream(s2):
                    P.embed(feats_p)
                with torch.cuda.stream(s1):
                    for _ in range(6):
                        outs.append((fe.fbank(w, cmn=False), ref))
                    outs.append((fe.fbank(w, cmn=True), ref_cmn))
                    outs.append((fe.fbank_ragged(w, ns, cmn=True), ref_rag))
            torch.cuda.synchronize()
            for got, want in outs:
                launches += 1
                assert torch.equal(got, want), (pname, pprec, rep, float((got - want).abs().max()))
    assert launches >= 1000
    assert _lib.lib().ws_debug_fbank_mode(7) == -1 and b"mode" in _lib.lib().ws_last_error()


def test_fbank_packed_reproducer_still_fails_next_to_the_f16x3_engine():
    """Sentinel of the mitigation above: the `no-packed-fp32-ops` build of the fbank kernel and the
    build-time ISA gate exist because `v_pk_{mul,add,fma}_f32 ... op_sel:[0,1]` returns wrong values in lanes 48..63
    next to binary16 GEMMs of another stream.  The round-3 packed build is still in the library
    (ws_debug_fbank_mode(1)): run it in the loop in which it failed ~70 % of the launches.  It must STILL differ from
    the serial bits; if a driver / firmware update ever makes it clean, this test reports XFAIL-style (`xfail` with the
    reason) instead of silently keeping a mitigation nobody needs any more."""
    from bench import device_wavs
    from wespeaker_amd import _lib
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    dev = torch.device("cuda:0")
    fe, fe2 = Frontend(16000, 80), Frontend(16000, 80)
    w = device_wavs(64, 32000, dev, 40)
    ref = fe.fbank(w, cmn=False).clone()                       # shipped kernel, nothing else running
    torch.cuda.synchronize()
    feats_p = fe2.fbank(w, cmn=True)
    P = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", synth.synth_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=12),
                           feat_dim=80, embed_dim=192, max_batch=64, max_frames=198)
    P.set_precision("f16x3")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    dirty = launches = 0
    bad_bins = set()
    try:
        assert _lib.lib().ws_debug_fbank_mode(1) == 0
        alone = fe.fbank(w, cmn=False).clone()
        torch.cuda.synchronize()
        # the packed build is right when it runs alone (its own bits: packed FMAs round differently from the shipped
        # kernel's separate multiplies and adds), and repeatably so
        assert float((alone - ref).abs().max()) < 1e-3 and torch.equal(fe.fbank(w, cmn=False), alone)
        ref = alone
        for rep in range(12):
            outs = []
            for _ in range(5):
                with torch.cuda.stream(s2):
                    P.embed(feats_p)
                with torch.cuda.stream(s1):
                    for _ in range(6):
                        outs.append(fe.fbank(w, cmn=False))
            torch.cuda.synchronize()
            for got in outs:
                launches += 1
                if not torch.equal(got, ref):
                    dirty += 1
                    bad_bins.update(torch.nonzero((got != ref).any(0).any(0)).flatten().tolist())
    finally:
        assert _lib.lib().ws_debug_fbank_mode(0) == 0
    assert launches == 360
    if dirty == 0:
        pytest.xfail("the packed-fp32 reproducer ran %d launches next to the f16x3 engine without one wrong value: the "
                     "erratum of DESIGN.md 6.0 no longer reproduces on this driver / firmware -- revisit the "
                     "no-packed-fp32-ops build of fbank_kernel.inc and build.check_isa" % launches)
    # the damage is where round 3 saw it: mel bins fed by lanes 48..63 of the power-spectrum loop
    assert bad_bins and bad_bins <= set(range(35, 43)) | set(range(56, 61)) | set(range(68, 73)) | {78, 79}, sorted(bad_bins)


def test_fbank_ragged_on_a_shared_frontend_from_two_streams():
    """ws_fbank_ragged uploads a per-call frame-count table; round 3 kept ONE table per frontend and overwrote it
    while kernels of an earlier call on another stream could still read it.  The tables now live in a ring of four
    slots whose reuse waits for the consumers: ten ragged calls with different length sets, alternating between two
    streams on ONE frontend, none synchronised in between -- every result equals its serial run."""
    from bench import device_wavs
    from wespeaker_amd.engine import Frontend
    dev = torch.device("cuda:0")
    fe = Frontend(16000, 80)
    w = device_wavs(48, 40000, dev, 77)
    rs = np.random.RandomState(3)
    lens = [rs.randint(400, 40001, size=48).astype(np.int32) for _ in range(10)]
    for ns in lens:
        ns[rs.randint(48)] = 40000                      # (same padded length for every call)
    ref = [fe.fbank_ragged(w, ns, cmn=True).clone() for ns in lens]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for i, ns in enumerate(lens):
        with torch.cuda.stream(streams[i & 1]):
            outs.append(fe.fbank_ragged(w, ns, cmn=True))
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, ref)):
        assert torch.equal(o, r), i


def test_engines_of_different_families_on_two_streams_keep_their_bits():
    """Two ws_engine handles on two streams is a documented use of the C-ABI (include/wespeaker_amd.h): an fp32 engine
    of every family next to a binary16 ECAPA engine, and a binary16 CAM++ / ResNet engine next to it, each keep the
    bits they give alone (the screen for any other kernel with the defect of DESIGN.md 6.0)."""
    from bench import device_wavs
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    dev = torch.device("cuda:0")
    fe, fe2 = Frontend(16000, 80), Frontend(16000, 80)
    w = device_wavs(48, 32000, dev, 61)
    partner = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", synth.synth_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=12),
                                 feat_dim=80, embed_dim=192, max_batch=48, max_frames=198)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, E in (("ECAPA_TDNN_GLOB_c512", 192), ("ResNet34", 256), ("CAMPPlus", 512)):
        A = NativeSpeakerModel(name, synth.synth_state_dict(name, 80, E, seed=11), feat_dim=80, embed_dim=E,
                               max_batch=48, max_frames=198)
        for prec in ("fp32", "f16"):
            A.set_precision(prec)
            ref = A.extract(fe, w).clone()
            torch.cuda.synchronize()
            for pprec in ("f16x3", "f16"):
                partner.set_precision(pprec)
                outs = []
                for _ in range(6):
                    with torch.cuda.stream(s2):
                        partner.extract(fe2, w)
                    with torch.cuda.stream(s1):
                        outs.append(A.extract(fe, w))
                        outs.append(A.extract(fe, w))
                torch.cuda.synchronize()
                for o in outs:
                    assert torch.equal(o, ref), (name, prec, pprec, float((o - ref).abs().max()))


def test_attentive_pooling_kernel_forced_on_small_batches(tmp_path):
    """astp_fused.hip is chosen by a cost model (a full round of one-workgroup-per-utterance work must beat the three
    tile launches), so small test batches never reach it.  WS_ASTP_FUSED=2 forces it: uniform batches at the edges
    of its three row-block variants (T = 64 .. 208) and ragged batches, GLOB and plain ECAPA, against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "forced.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from fixtures import synth\n"
        "from wespeaker_amd import NativeSpeakerModel\n"
        "out = {}\n"
        "for name in ('ECAPA_TDNN_GLOB_c512', 'ECAPA_TDNN_c512'):\n"
        "    sd = synth.synth_state_dict(name, 80, 192, seed=42)\n"
        "    m = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=192, max_batch=8, max_frames=208)\n"
        "    for T in (64, 100, 112, 113, 160, 161, 198, 208):\n"
        "        f = np.random.RandomState(T).randn(3, T, 80).astype(np.float32)\n"
        "        out['%%s/u%%d' %% (name, T)] = m(torch.from_numpy(f))[-1].cpu().numpy()\n"
        "    for k, lens in enumerate(([198, 150, 57, 208, 5, 203], [160, 113, 129, 75], [112, 64, 100, 97])):\n"
        "        pad = np.full((len(lens), max(lens), 80), np.nan, dtype=np.float32)\n"
        "        for i, L in enumerate(lens):\n"
        "            pad[i, :L] = np.random.RandomState(100 + i).randn(L, 80).astype(np.float32)\n"
        "        out['%%s/r%%d' %% (name, k)] = m.embed_ragged(torch.from_numpy(pad), lens).cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n" % root)
    env = dict(os.environ, PYTHONPATH=root, WS_ASTP_FUSED="2", WS_DISPATCH_LOG="0")
    path = str(tmp_path / "forced.npz")
    r = subprocess.run([sys.executable, str(script), path], env=env, cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    got = np.load(path)
    for name in ("ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_c512"):
        sd = synth.synth_state_dict(name, 80, 192, seed=42)
        for T in (64, 100, 112, 113, 160, 161, 198, 208):
            f = np.random.RandomState(T).randn(3, T, 80).astype(np.float32)
            ref = oecapa.ecapa_forward(sd, f).numpy()
            assert _rel_err(got["%s/u%d" % (name, T)], ref).max() < REL_TOL, (name, T)
        for k, lens in enumerate(([198, 150, 57, 208, 5, 203], [160, 113, 129, 75], [112, 64, 100, 97])):
            feats = [np.random.RandomState(100 + i).randn(L, 80).astype(np.float32) for i, L in enumerate(lens)]
            ref = _oracle_rows(lambda f: oecapa.ecapa_forward(sd, f).numpy(), feats)
            g = got["%s/r%d" % (name, k)]
            assert np.isfinite(g).all()
            assert _cos_err(g, ref).max() < COS_TOL and _rel_err(g, ref).max() < REL_TOL, (name, lens)


def test_bench_two_ranks_with_lanes_on_one_gpu():
    """bench.py --gpus 2 (self-launching) with its default number of batches in flight per rank (three since round 6),
    both ranks on this one GPU (WS_SHARE_GPU=1, gloo): the per-step gathers are issued under the lane's stream and
    joined `lanes` steps later."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WS_SHARE_GPU="1", WS_DIST_BACKEND="gloo", PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--windows", "1", "--batch", "64", "--headline-only"], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 128
    assert line["config"]["batches_in_flight_per_gpu"] == 3 and line["value"] > 0
    assert line["roofline"]["frac"] > 0


def _rccl_env(root, port, log_path):
    """A lone rank on a REAL RCCL process group: WS_DIST_FORCE_GROUP=1 makes init_distributed build the group and
    every collective run at world size 1 (parallel.collectives_active); NCCL_DEBUG=INFO (into a file: RCCL writes it
    through its own buffer, in the middle of other stdout lines) proves it was RCCL."""
    return dict(os.environ, PYTHONPATH=root, WS_DIST_FORCE_GROUP="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
                MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NCCL_DEBUG="INFO", NCCL_DEBUG_FILE=str(log_path),
                HSA_ENABLE_IPC_MODE_LEGACY="0")


def _assert_rccl_ran(log_path):
    log = open(log_path).read()
    assert "NCCL INFO" in log, log[-1500:]
    assert re.search(r"nranks 1\b", log) or re.search(r"nRanks 1\b", log), log[-1500:]
    assert "Init COMPLETE" in log or "init complete" in log.lower(), log[-1500:]


def test_rccl_group_of_one_runs_bench_default_and_set_modes(tmp_path):
    """The nccl (= RCCL) branches of bench.py and parallel.py executed for real on the one GPU there is:
    init_process_group("nccl", device_id), the device barrier of fence(), the device-side max_over_ranks all_reduce,
    gather_rows_async's all_gather_into_tensor(async_op=True) under a lane's stream (default mode) and gather_rows'
    blocking all_gather_into_tensor (set mode); the gathered embeddings equal the un-gathered run's checksum."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"]
    for extra in (["--windows", "1", "--batch", "64", "--headline-only"],
                  ["--workload", "stream10k", "--total-utts", "700", "--batch", "256"]):
        lines = {}
        for forced in (True, False):
            log = tmp_path / ("rccl_%d.log" % len(extra))
            env = _rccl_env(root, 29873, log) if forced else dict(os.environ, PYTHONPATH=root)
            r = subprocess.run(base + extra, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-3000:]
            lines[forced] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            if forced:
                _assert_rccl_ran(log)
                assert lines[forced].get("collective_backend", "nccl") == "nccl"
        a, b = lines[True], lines[False]
        assert a["n_gpus"] == 1 and a["value"] > 0
        ka = a.get("set", a)
        kb = b.get("set", b)
        assert ka["embedding_checksum"] == kb["embedding_checksum"], (extra, ka["embedding_checksum"], kb["embedding_checksum"])


def test_rccl_group_of_one_runs_the_extract_driver(tmp_path):
    """python -m wespeaker_amd.extract --gather_npz on an RCCL group of one: the barrier, the all_reduce of the
    embedding width / the per-job counts on device tensors, gather_rows and all_gather_object (extract.py run_jobs)
    execute under nccl; arks and gathered rows equal the plain single-process run bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mdir = str(tmp_path / "exp")
    synth.write_model_dir(mdir, "ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    n = len(_driver_corpus(tmp_path))
    base = [sys.executable, "-m", "wespeaker_amd.extract", "--exp_dir", mdir, "--model_path",
            os.path.join(mdir, "avg_model.pt"), "--data_type", "raw", "--data_list", str(tmp_path / "raw.list"),
            "--wavs_num", str(n), "--nj", "3", "--batch_size", "1"]
    outs = {}
    for tag, env in (("plain", dict(os.environ, PYTHONPATH=root)),
                     ("rccl", _rccl_env(root, 29874, tmp_path / "rccl.log"))):
        r = subprocess.run(base + ["--store_dir", tag, "--gather_npz", str(tmp_path / (tag + ".npz"))], env=env, cwd=root,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        outs[tag] = r.stdout
    _assert_rccl_ran(tmp_path / "rccl.log")
    g0, g1 = np.load(str(tmp_path / "plain.npz")), np.load(str(tmp_path / "rccl.npz"))
    assert list(g0["keys"]) == list(g1["keys"]) and np.array_equal(g0["emb"], g1["emb"])
    for j in range(3):
        assert open(os.path.join(mdir, "embeddings", "plain", "xvector_%03d.ark" % j), "rb").read() == \
               open(os.path.join(mdir, "embeddings", "rccl", "xvector_%03d.ark" % j), "rb").read()


def test_rccl_group_of_one_sharded_plda_matrix(tmp_path):
    """SURVEY 8(e), last sentence: the LLR matrix with the enrollment rows sharded over the ranks and the blocks
    collected by one all_gather -- here on an RCCL group of one (the real ws_plda_llr_matrix per block, a real
    all_gather_into_tensor of float64 rows); equal to the plain call bit for bit.  World size 2 runs on gloo with the
    oracle as the block scorer (tests/test_host_logic.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "sharded.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from fixtures import synth\n"
        "from wespeaker_amd import TwoCovPLDA, parallel\n"
        "rank, world, local = parallel.init_distributed()\n"
        "assert parallel.collectives_active() and torch.distributed.get_backend() == 'nccl'\n"
        "p = synth.synth_plda(192, seed=7)\n"
        "plda = TwoCovPLDA.from_params(p['mu'], p['transform'], p['psi'], p['offset'], False)\n"
        "emb, _ = synth.synth_embeddings(900, 192, seed=23)\n"
        "e_t, t_t = plda.prepare_test(emb[:300]), plda.prepare_test(emb[300:])\n"
        "full = plda.llr_matrix(e_t, 1, t_t)\n"
        "got = parallel.llr_matrix_sharded(lambda lo, hi: plda.llr_matrix(e_t[lo:hi], 1, t_t), 300)\n"
        "assert got.shape == (300, 600) and torch.equal(got, full)\n"
        "torch.distributed.barrier(device_ids=[0]); torch.distributed.destroy_process_group()\n"
        "print('sharded ok')\n" % root)
    r = subprocess.run([sys.executable, str(script)], env=_rccl_env(root, 29875, tmp_path / "rccl.log"), cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "sharded ok" in r.stdout, r.stdout[-3000:]
    _assert_rccl_ran(tmp_path / "rccl.log")


def test_full_size_plda_one_million_trials():
    """configs[4]: 1 M trial pairs.  pairs == gather of the dense matrix; the uniform-n and per-model-n
    code paths agree; LLR(e, t, n) is invariant to the order in which the tables are given."""
    from wespeaker_amd import TwoCovPLDA
    p = synth.synth_plda(192, seed=7)
    plda = TwoCovPLDA.from_params(p["mu"], p["transform"], p["psi"], p["offset"], False)
    emb, _ = synth.synth_embeddings(4000, 192, seed=21)
    e_t = plda.prepare_test(emb[:2000])
    t_t = plda.prepare_test(emb[2000:])
    ie, it = synth.synth_trial_pairs(1000000, 2000, 2000, seed=99)
    pairs_u = plda.llr_pairs(e_t, 1, t_t, ie, it)
    pairs_g = plda.llr_pairs(e_t, torch.ones(2000, dtype=torch.int32), t_t, ie, it)
    mat = plda.llr_matrix(e_t, 1, t_t)
    gathered = mat[torch.from_numpy(ie).long().to(mat.device), torch.from_numpy(it).long().to(mat.device)]
    assert float((pairs_u - gathered).abs().max()) < 1e-9
    assert float((pairs_u - pairs_g).abs().max()) < 1e-9
    # spot check 200 trials against the reference's per-trial formula
    e_np, t_np = e_t.cpu().numpy(), t_t.cpu().numpy()
    ref = np.array([oplda.log_likelihood_ratio(p, e_np[ie[k]], t_np[it[k]], 1) for k in range(0, 1000000, 5000)])
    assert np.abs(pairs_u.cpu().numpy()[::5000] - ref).max() < 1e-8


def test_plda_dense_matrix_big_tiles_same_bits_and_oracle():
    """The 128x128-tile f64 GEMM that large LLR matrices take (>= 1024 such tiles: 4096 x 4096 and up, i.e. the >= 1e8
    -trial blocks of llr_matrix_sharded) accumulates every output's k-steps in the 64x64 kernel's order: a 4100 x 4130
    matrix (ragged edges) equals, bit for bit, the blocks the small kernel computes for row / column sub-ranges; both
    n-session forms (K = D with a per-test constant, K = 2 D per-model n); spot trials against the reference's formula."""
    from wespeaker_amd import TwoCovPLDA
    p = synth.synth_plda(192, seed=7)
    plda = TwoCovPLDA.from_params(p["mu"], p["transform"], p["psi"], p["offset"], False)
    emb, _ = synth.synth_embeddings(4100 + 4130, 192, seed=23)
    e_t = plda.prepare_test(emb[:4100])
    t_t = plda.prepare_test(emb[4100:])
    n_per = torch.from_numpy((1 + np.arange(4100) % 3).astype(np.int32))
    for n_sessions in (1, n_per):
        big = plda.llr_matrix(e_t, n_sessions, t_t)                       # 33 x 33 big tiles
        assert big.shape == (4100, 4130) and bool(torch.isfinite(big).all())
        for r0, r1, c0, c1 in ((0, 900, 0, 1000), (3300, 4100, 3200, 4130), (2000, 2064, 100, 164)):
            ns = n_sessions if isinstance(n_sessions, int) else n_sessions[r0:r1]
            small = plda.llr_matrix(e_t[r0:r1], ns, t_t[c0:c1])           # < 1024 big tiles: the 64x64 kernel
            assert torch.equal(small, big[r0:r1, c0:c1]), (r0, c0)
    e_np, t_np = e_t.cpu().numpy(), t_t.cpu().numpy()
    got = plda.llr_matrix(e_t, 1, t_t).cpu().numpy()
    for (i, j) in ((0, 0), (4099, 4129), (2047, 2048), (128, 4000), (3333, 77)):
        assert abs(got[i, j] - oplda.log_likelihood_ratio(p, e_np[i], t_np[j], 1)) < 1e-8


def test_full_size_fbank_scale_property(frontend):
    """log-mel(alpha x) = log-mel(x) + 2 log alpha, so CMN'd features are scale invariant
    (floor effects aside); checked on a 256-utterance batch."""
    from bench import device_wavs
    wav = device_wavs(256, 32000, frontend.device, 9).float()
    a = frontend.fbank(wav, cmn=False)
    b = frontend.fbank(wav * 0.25, cmn=False)
    assert float((b - a - 2.0 * np.log(0.25)).abs().max()) < 2e-3
    ac = frontend.fbank(wav, cmn=True)
    bc = frontend.fbank(wav * 0.25, cmn=True)
    assert float((ac - bc).abs().max()) < 2e-3
    assert float(ac.mean(dim=1).abs().max()) < 1e-3


# ============================================================ cosine scoring + AS-norm
# Tolerances: cosines are float32 dot products of unit vectors (reference: float32 numpy / sklearn)
# -> 2e-6 absolute; cohort mean / std 2e-6 / 5e-6 absolute (the kernel accumulates in float64, numpy
# in pairwise float32); normalised scores divide by std ~ 0.05: 1e-3 absolute like the PLDA LLR bar.
COS_ATOL, STAT_ATOL, NORM_ATOL = 2e-6, 5e-6, 1e-3


@pytest.mark.parametrize("n,n_cohort,dim,top_n", [(60, 150, 192, 20), (37, 301, 100, 300),
                                                   (130, 1000, 256, 1000), (5, 7, 512, 3),
                                                   (300, 999, 192, 2000)])
def test_cohort_stats_match_oracle(n, n_cohort, dim, top_n):
    from oracle import score as oscore
    from wespeaker_amd import score as wscore
    emb, _ = synth.synth_embeddings(n, dim, seed=5)
    cohort, _ = synth.synth_embeddings(n_cohort, dim, seed=6)
    cohort[3] = cohort[1]                      # exact ties inside the cohort scores
    if n_cohort > 10:
        cohort[10] = cohort[1]
    m_ref, s_ref = oscore.get_mean_std(emb, cohort, top_n)
    t, c = wscore.UnitTable(emb), wscore.UnitTable(cohort)
    m, s = wscore.cohort_stats(t, c, top_n)
    assert np.abs(m.cpu().numpy() - m_ref).max() <= STAT_ATOL
    assert np.abs(s.cpu().numpy() - s_ref).max() <= STAT_ATOL
    # chunked path (scratch for 128 rows only) gives the same numbers bit for bit
    m2, s2 = wscore.cohort_stats(t, c, top_n, scratch_bytes=4 * 128 * ((n_cohort + 3) // 4 * 4))
    assert torch.equal(m, m2) and torch.equal(s, s2)
    # dense matrix against numpy
    unit = lambda x: x / np.sqrt(np.sum(x ** 2, axis=1, keepdims=True))
    S = wscore.cosine_matrix(t, c).cpu().numpy()
    assert np.abs(S - unit(emb) @ unit(cohort).T).max() <= COS_ATOL


def test_cosine_and_score_norm_files_match_reference_golden(tmp_path, golden_dir):
    """The file-level tools (bin/score.py main, bin/score_norm.py main) on the same ark / scp /
    trial files the reference was run on to make tests/golden/score_ref.npz."""
    from wespeaker_amd import score as wscore
    g = np.load(os.path.join(golden_dir, "score_ref.npz"))
    fix = synth.synth_scoring_set()
    paths = synth.write_scoring_files(fix, str(tmp_path / "emb"))
    for tag, mean_path in (("nomean", None), ("mean", paths["mean_vec"])):
        store = str(tmp_path / ("scores_" + tag))
        os.makedirs(store)
        wscore.trials_cosine_score(paths["eval_scp"], store, mean_path, [paths["trials"]])
        score_file = os.path.join(store, "trials.kaldi.score")
        rows = [l.split() for l in open(score_file)]
        assert [tuple(r[:2]) + (r[3],) for r in rows] == [tuple(t) for t in fix["trials"]]
        cos = np.array([float(r[2]) for r in rows])
        assert np.abs(cos - g[tag + "/cosine"]).max() <= 1.01e-5       # both sides are '%.5f' text
        for method in ("asnorm", "snorm"):
            nf = os.path.join(store, method + ".score")
            wscore.score_norm(method, 20, score_file, nf, paths["cohort_scp"], paths["eval_scp"],
                              mean_path)
            lines = [l.split() for l in open(nf)]
            assert all(len(l) == 8 for l in lines)
            cols = np.array([[float(x) for x in (l[2:3] + l[4:8])] for l in lines])
            ref = g["%s/%s" % (tag, method)]
            assert np.abs(cols[:, 0] - ref[:, 0]).max() <= NORM_ATOL
            assert np.abs(cols[:, 1:] - ref[:, 1:]).max() <= 1.01e-4   # '%.4f' text on both sides


def test_cosine_pairs_match_oracle_and_edge_cases():
    from oracle import score as oscore
    from wespeaker_amd import score as wscore
    emb, _ = synth.synth_embeddings(200, 192, seed=8)
    mv = emb.mean(0)
    ia, ib = synth.synth_trial_pairs(5000, 200, 200, seed=3)
    t = wscore.UnitTable(emb, mv)
    got = wscore.cosine_pairs(t, t, ia, ib).cpu().numpy()
    assert np.abs(got - oscore.cosine_pairs(emb, mv, ia, ib)).max() <= COS_ATOL
    assert np.abs(t.mag.cpu().numpy() - np.linalg.norm(emb - mv, axis=1)).max() <= 1e-4
    assert wscore.cosine_pairs(t, t, [], []).numel() == 0
    with pytest.raises(IndexError):
        wscore.cosine_pairs(t, t, [200], [0])
    with pytest.raises(ValueError):
        wscore.score_norm("znorm", 10, "x", "y", "z", "w")


def test_full_size_score_norm_properties():
    """VoxCeleb1-O-sized tables (4874 x 256) against a 20k cohort: size-independent properties."""
    from wespeaker_amd import score as wscore
    emb, _ = synth.synth_embeddings(4874, 256, seed=31)
    cohort, _ = synth.synth_embeddings(20000, 256, seed=32)
    t, c = wscore.UnitTable(emb), wscore.UnitTable(cohort)
    m, s = wscore.cohort_stats(t, c, 300)
    # (1) cohort order does not matter (exact selection -> only float64 summation order changes)
    perm = np.random.Generator(np.random.PCG64(1)).permutation(20000)
    m_p, s_p = wscore.cohort_stats(t, wscore.UnitTable(cohort[perm]), 300)
    assert (m - m_p).abs().max().item() <= 1e-6 and (s - s_p).abs().max().item() <= 1e-6
    # (2) top_n >= cohort size == plain row statistics of the dense matrix
    S = wscore.cosine_matrix(t, c).double()
    m_all, s_all = wscore.cohort_stats(t, c, 10 ** 9)
    assert (m_all.double() - S.mean(1)).abs().max().item() <= 1e-6
    assert (s_all.double() - S.std(1, unbiased=False)).abs().max().item() <= 1e-6
    # (3) top-N against torch.topk on the same matrix
    top = S.topk(300, dim=1).values
    assert (m.double() - top.mean(1)).abs().max().item() <= 1e-6
    assert (s.double() - top.std(1, unbiased=False)).abs().max().item() <= 1e-6
    # (4) monotone: a larger N can only lower the mean of the top-N
    m_600, _ = wscore.cohort_stats(t, c, 600)
    assert bool((m_600 <= m + 1e-7).all())


# ============================================== chunk-and-average mode (native runtime)
@pytest.mark.parametrize("seconds,samples_per_chunk", [(5.0, 32000), (1.0, 32000), (3.975, 32000),
                                                       (2.5, 0), (6.3, 24000)])
def test_extract_chunked_matches_oracle(frontend, seconds, samples_per_chunk):
    from oracle import chunked
    sd, model = _engine("ECAPA_TDNN_GLOB_c512")
    wav = synth.synth_wav(3, int(round(seconds * 16000)))
    emb, n_chunks = model.extract_chunked(frontend, torch.from_numpy(wav), samples_per_chunk)
    feats = ofbank.speaker_features(wav, cmn=False)           # no CMN: the rule applies it per chunk
    ref, n_ref = chunked.extract_chunked(
        feats, 16000, samples_per_chunk,
        lambda b: oecapa.ecapa_forward(sd, torch.from_numpy(b)).numpy())
    assert n_chunks == n_ref
    assert _cos_err(emb.cpu().numpy()[None], ref[None]) <= 1e-4
    assert _rel_err(emb.cpu().numpy(), ref) <= 2e-3


@pytest.mark.parametrize("prec", ["fp32", "f16x3", "f16"])
def test_extract_chunked_matches_reference_engine_golden(frontend, golden_dir, prec):
    """ws_extract_chunked against the reference's OWN SpeakerEngine::ExtractEmbedding
    (speaker_engine.cc:83-159 compiled into oracle/_ref, model = the pinned ECAPA oracle;
    tests/golden/chunked_ref.npz): chunk counts equal, embeddings inside the north_star bar on every
    back-end (the reference engine sits on its native float-FFT fbank: 1.7e-4 log-mel noise)."""
    g = np.load(os.path.join(golden_dir, "chunked_ref.npz"))
    sd, model = _engine("ECAPA_TDNN_GLOB_c512")
    model.set_precision(prec)
    for i, (seed, n, spc) in enumerate(g["cases"]):
        wav = synth.synth_wav(int(seed), int(n))
        emb, n_chunks = model.extract_chunked(frontend, torch.from_numpy(wav), int(spc))
        assert n_chunks == int(g["%d/n_chunks" % i])
        e, ref = emb.cpu().numpy(), g["%d/emb" % i]
        assert _cos_err(e[None], ref[None]) <= COS_TOL, (i, prec)
        assert _rel_err(e, ref) <= (2e-3 if prec != "f16" else 3e-3), (i, prec, _rel_err(e, ref))


def test_cpp_caller_of_the_c_abi(frontend, golden_dir, tmp_path):
    """wespeaker_amd/lib/extract_emb_main (plain g++ C++, no Python, no torch; twin of the reference's
    runtime/core/bin/extract_emb_main.cc:43-117) loads a flat weight file through ws_engine_load and
    prints `key e0 e1 ...` per wav.scp line: same numbers as the ctypes path, and inside the bar of the
    reference engine's golden."""
    import subprocess
    from wespeaker_amd import build as wbuild
    from wespeaker_amd.engine import NativeSpeakerModel, save_native_model
    assert os.path.exists(wbuild.MAIN_BIN), "build with python -m wespeaker_amd.build"
    g = np.load(os.path.join(golden_dir, "chunked_ref.npz"))
    sd, model = _engine("ECAPA_TDNN_GLOB_c512")
    mpath = str(tmp_path / "model.wsamd")
    save_native_model(mpath, "ECAPA_TDNN_GLOB_c512", sd, 80, 192)
    loaded = NativeSpeakerModel.from_file(mpath, max_batch=8, max_frames=400)     # ws_engine_load via ctypes
    assert loaded.model_name == "ECAPA_TDNN_GLOB_c512" and loaded.embed_dim == 192
    cases = [(i, int(s), int(n)) for i, (s, n, spc) in enumerate(g["cases"]) if int(spc) == 32000]
    lines = []
    for i, seed, n in cases:
        p = str(tmp_path / ("c%d.wav" % i))
        synth.write_wav(p, synth.synth_wav(seed, n))
        lines.append("case%d %s" % (i, p))
    (tmp_path / "wav.scp").write_text("\n".join(lines) + "\n")
    res = subprocess.run([wbuild.MAIN_BIN, "--wav_scp", str(tmp_path / "wav.scp"), "--speaker_model_path", mpath,
                          "--samples_per_chunk", "32000", "--result", str(tmp_path / "emb.txt")],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    assert "RTF" in res.stderr
    rows = [l.split() for l in open(tmp_path / "emb.txt")]
    assert [r[0] for r in rows] == ["case%d" % i for i, _, _ in cases]
    for r, (i, seed, n) in zip(rows, cases):
        e = np.array([float(x) for x in r[1:]], dtype=np.float32)
        assert e.shape == (192,)
        via_ctypes, _ = loaded.extract_chunked(frontend, torch.from_numpy(synth.synth_wav(seed, n)), 32000)
        assert np.array_equal(e, via_ctypes.cpu().numpy())          # %.9g round-trips float32
        assert _cos_err(e[None], g["%d/emb" % i][None]) <= COS_TOL
    bad = subprocess.run([wbuild.MAIN_BIN, "--wav_path", str(tmp_path / "c0.wav"), "--speaker_model_path",
                          str(tmp_path / "wav.scp")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert bad.returncode == 1 and "bad magic" in bad.stderr


def test_extract_chunked_errors(frontend):
    from wespeaker_amd._lib import NativeError
    _, model = _engine("ECAPA_TDNN_GLOB_c512")
    with pytest.raises(NativeError):
        model.extract_chunked(frontend, torch.zeros(100, dtype=torch.int16), 32000)   # < 1 frame
    with pytest.raises(NativeError):
        model.extract_chunked(frontend, torch.zeros(32000, dtype=torch.int16), 100)   # chunk < frame


# ================================================= PLDA training / adaptation
def test_plda_stats_kernel_matches_numpy_float64():
    from wespeaker_amd import plda_train
    rng = np.random.Generator(np.random.PCG64(3))
    for n, dim, n_cls, norm, use_mean in [(1000, 192, 37, False, False), (777, 100, 50, True, True),
                                          (5000, 256, 300, True, False), (64, 512, 1, False, True)]:
        x = rng.standard_normal((n, dim)).astype(np.float32) + 0.3
        cuts = np.sort(rng.choice(np.arange(1, n), n_cls - 1, replace=False)) if n_cls > 1 else []
        offs = np.concatenate([[0], cuts, [n]]).astype(np.int32)
        mv = x.astype(np.float64).mean(0) if use_mean else None
        cm, sc = plda_train.gpu_stats(x, offs, mv, norm)
        y = x.astype(np.float64) - (mv if use_mean else 0.0)
        if norm:
            y = np.sqrt(dim) * y / np.linalg.norm(y, axis=1, keepdims=True)
        ref_cm = np.stack([y[a:b].mean(0) for a, b in zip(offs[:-1], offs[1:])])
        ref_sc = sum((y[a:b] - m).T @ (y[a:b] - m) for a, b, m in zip(offs[:-1], offs[1:], ref_cm))
        assert np.abs(cm - ref_cm).max() <= 1e-12 * max(1.0, np.abs(ref_cm).max())
        assert np.abs(sc - ref_sc).max() <= 1e-11 * np.abs(ref_sc).max()
        assert np.abs(sc - sc.T).max() <= 1e-11 * np.abs(ref_sc).max()


@pytest.mark.parametrize("tag,sub,nl", [("plain", False, False), ("sub_nl", True, True)])
def test_plda_train_and_adapt_match_oracle_and_reference_golden(tmp_path, golden_dir, tag, sub, nl):
    """TwoCovPLDA(scp_file, utt2spk_file, ...).train(3) / .adapt(scp) on the same ark/scp files the
    reference was trained on.  The reference keeps per-speaker statistics in float32; ours are
    float64, hence 2e-5 relative on B / W / psi; LLRs at the north-star 1e-3."""
    from wespeaker_amd import TwoCovPLDA
    g = np.load(os.path.join(golden_dir, "plda_train_ref.npz"))
    fix = synth.synth_plda_training_set()
    paths = synth.write_plda_training_files(fix, str(tmp_path))
    plda = TwoCovPLDA(scp_file=paths["scp"], utt2spk_file=paths["utt2spk"], embed_dim=64,
                      subtract_train_set_mean=sub, normalize_length=nl)
    np.testing.assert_allclose(plda.stats.offset_scatter, g[tag + "/offset_scatter"], rtol=0,
                               atol=2e-5 * np.abs(g[tag + "/offset_scatter"]).max())
    plda.train(3)
    for k in ("B", "W"):
        ref = g["%s/%s" % (tag, k)]
        assert np.abs(getattr(plda, k) - ref).max() <= 2e-5 * np.abs(ref).max(), k
    np.testing.assert_allclose(plda.psi, g[tag + "/psi"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(plda.mu, g[tag + "/mu"], rtol=0, atol=1e-6)
    probe, _ = synth.synth_embeddings(24, 64, seed=47)

    def llr(model, n):
        tr = model.transform_rows(probe.astype(np.float64))
        return model.llr_matrix(tr[:12], np.full(12, n, dtype=np.int32), tr[12:]).cpu().numpy()
    assert np.abs(llr(plda, 2) - g[tag + "/llr"]).max() <= LLR_TOL
    adapted = plda.adapt(paths["adapt_scp"], 0.5, 0.5)
    assert adapted.normalize_length is False                  # reference quirk: not inherited
    np.testing.assert_allclose(np.sort(adapted.psi), np.sort(g[tag + "/adapt_psi"]), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(adapted.mu, g[tag + "/adapt_mu"], rtol=0, atol=1e-5)
    assert np.abs(llr(adapted, 1) - g[tag + "/adapt_llr"]).max() <= LLR_TOL


def test_full_size_plda_stats_properties():
    """200k x 256 training set, 5000 speakers: identities that do not need a CPU reference."""
    from wespeaker_amd import plda_train
    n, dim, n_cls = 200000, 256, 5000
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((n, dim), generator=g, dtype=torch.float32)
    offs = np.linspace(0, n, n_cls + 1).astype(np.int32)
    cm, sc = plda_train.gpu_stats(x.numpy(), offs, None, False)
    xd = x.double().cuda()
    cls = torch.from_numpy(np.repeat(np.arange(n_cls), np.diff(offs))).cuda()
    cm_t = torch.zeros((n_cls, dim), dtype=torch.float64, device="cuda").index_add_(0, cls, xd)
    cm_t /= torch.from_numpy(np.diff(offs)).double().cuda()[:, None]
    assert np.abs(cm - cm_t.cpu().numpy()).max() <= 1e-12
    yc = xd - cm_t[cls]
    # trace identity and a full check through torch's float64 matmul
    assert abs(np.trace(sc) - float((yc * yc).sum())) <= 1e-9 * np.trace(sc)
    assert np.abs(sc - (yc.T @ yc).cpu().numpy()).max() <= 1e-10 * np.abs(sc).max()


# ================================================= embedding-processing chain (LDA)
def test_embedding_processing_chain_matches_reference_golden(tmp_path, golden_dir):
    """EmbeddingProcessingChain built from ark/scp files (statistics by ws_plda_stats, links applied
    by ws_rows_affine) against the reference's own chain; eigenvector signs aligned per column."""
    from oracle import embedding_processing as oproc
    from wespeaker_amd import embedding_processing as wproc
    from wespeaker_amd.kaldi_io import read_vec_scp
    g = np.load(os.path.join(golden_dir, "embd_proc_ref.npz"))
    fix = synth.synth_plda_training_set()
    probe, _ = synth.synth_embeddings(24, 64, seed=47)
    paths = synth.write_plda_training_files(fix, str(tmp_path))
    chain_str = ("mean-subtract --scp %s | length-norm | lda --scp %s --utt2spk %s --dim 20 | length-norm"
                 % (paths["scp"], paths["scp"], paths["utt2spk"]))
    assert wproc.chain_string_to_dict("a --x 1 | b --y=2")[1] == ['b', {'y': '2'}]
    chain = wproc.prep_embd_proc(chain_str, str(tmp_path / "chain.pkl"))
    out = chain(probe)
    sgn = np.sign(np.sum(out * g["out"], axis=0))
    assert np.abs(out * sgn - g["out"]).max() <= 1e-5
    ref, _ = oproc.chain_fit_apply(fix["emb"], fix["spk"], 20, probe)
    assert np.abs(out * np.sign(np.sum(out * ref, axis=0)) - ref).max() <= 1e-5
    np.testing.assert_allclose(chain.chain_of_classes[0].mean, g["mean1"], rtol=0, atol=1e-6)
    # save / load / apply through files (bin/apply_embd_proc.py)
    res = wproc.apply_embd_proc(str(tmp_path / "chain.pkl"), paths["adapt_scp"], str(tmp_path / "o.ark,scp"))
    back = np.vstack(list(read_vec_scp(str(tmp_path / "o.scp")).values()))
    assert back.shape == (300, 20) and np.abs(back - res).max() <= 1e-12
    assert np.abs(np.linalg.norm(back, axis=1) - 1.0).max() <= 1e-12
    # update_link: replace the LDA by a 10-dimensional one
    chain.update_link(2, "lda --scp %s --utt2spk %s --dim 10" % (paths["scp"], paths["utt2spk"]))
    assert chain(probe).shape == (24, 10)


def test_resample_matches_oracle_and_speaker_api(tmp_path):
    """ws_resample (torchaudio.transforms.Resample arithmetic, cli/speaker.py:157-160) vs the numpy
    restatement, for down- and up-sampling, and through Speaker.extract_embedding on an 8 kHz file."""
    import wespeaker_amd
    from oracle import resample as oresample
    from wespeaker_amd import audio
    rng = np.random.RandomState(0)
    for orig, new, n in [(8000, 16000, 8000), (44100, 16000, 22050), (16000, 8000, 4001), (48000, 16000, 4800)]:
        x = (rng.randn(n) * 3000).astype(np.float32)
        got = audio.resample(torch.from_numpy(x), orig, new).cpu().numpy()
        ref = oresample.resample(x, orig, new)
        assert got.shape == ref.shape == (-(-new * n // orig),)
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max()          # float32 summation order
    # a pure tone keeps its frequency and amplitude
    t = np.arange(8000) / 8000.0
    y = audio.resample(torch.from_numpy(np.sin(2 * np.pi * 440 * t).astype(np.float32)), 8000, 16000).cpu().numpy()
    t2 = np.arange(y.shape[0]) / 16000.0
    assert np.abs(y[200:-200] - np.sin(2 * np.pi * 440 * t2)[200:-200]).max() < 2e-2
    d = str(tmp_path / "ecapa")
    synth.write_model_dir(d, "ECAPA_TDNN_GLOB_c512", embed_dim=192, seed=42)
    wav8 = str(tmp_path / "u8k.wav")
    synth.write_wav(wav8, synth.synth_wav(3, 16000), sample_rate=8000)      # 2 s at 8 kHz
    emb = wespeaker_amd.load_model(d).extract_embedding(wav8)
    assert emb.shape == (192,) and bool(torch.isfinite(emb).all())


# ================================== batch extraction driver (extract.py / extract_embedding.sh)
def _driver_corpus(tmp_path, n=14):
    import json
    lengths = [32000, 24000, 40000, 32000, 16160][:5] * 3
    lengths = lengths[:n]
    lines = []
    for i, L in enumerate(lengths):
        p = str(tmp_path / ("d%02d.wav" % i))
        synth.write_wav(p, synth.synth_wav(500 + i, L))
        lines.append(json.dumps({"key": "utt%02d" % i, "wav": p, "spk": "s%d" % (i % 4)}))
    (tmp_path / "raw.list").write_text("\n".join(lines) + "\n")
    return lengths


def test_extract_job_matches_oracle_and_per_file_api(tmp_path):
    """wespeaker_amd.extract.extract (= bin/extract.py, one job): config.yaml + avg_model.pt + raw.list ->
    ark/scp whose rows equal the batch-1 oracle (the reference's whole-utterance mode) in list order."""
    import wespeaker_amd
    from wespeaker_amd import extract as wx, kaldi_io
    mdir = str(tmp_path / "exp")
    sd = synth.write_model_dir(mdir, "ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    lengths = _driver_corpus(tmp_path)
    keys, emb = wx.extract(config=os.path.join(mdir, "config.yaml"), model_path=os.path.join(mdir, "avg_model.pt"),
                           data_type="raw", data_list=str(tmp_path / "raw.list"),
                           embed_ark=str(tmp_path / "out" / "xvector_000.ark"), batch_size=1, num_workers=2)
    assert keys == ["utt%02d" % i for i in range(len(lengths))]
    ref = np.stack([oecapa.ecapa_forward(sd, ofbank.speaker_features(synth.synth_wav(500 + i, L))[None]).numpy()[0]
                    for i, L in enumerate(lengths)])
    assert _cos_err(emb, ref).max() < COS_TOL and _rel_err(emb, ref).max() < 5e-4
    back = kaldi_io.read_vec_scp(str(tmp_path / "out" / "xvector_000.scp"))
    assert list(back) == keys and all(np.array_equal(back[k], emb[i]) for i, k in enumerate(keys))
    spk = wespeaker_amd.load_model(mdir)
    one = spk.extract_embedding(str(tmp_path / "d02.wav")).numpy()
    assert _rel_err(one, emb[2]) < 1e-5
    # cohort mode: batch 16 of random 200-frame crops (extract_vox.sh:31) -> each row = oracle on its crop
    keys_c, emb_c = wx.extract(config=os.path.join(mdir, "config.yaml"), model_path=os.path.join(mdir, "avg_model.pt"),
                               data_type="raw", data_list=str(tmp_path / "raw.list"),
                               embed_ark=str(tmp_path / "coh" / "xvector_000.ark"), batch_size=16, seed=3)
    for i in (0, 2, 4):
        w = torch.from_numpy(synth.synth_wav(500 + i, lengths[i]))
        crop = wx.random_chunk(w, "utt%02d" % i, 32240, 3).numpy()
        r = oecapa.ecapa_forward(sd, ofbank.speaker_features(crop)[None]).numpy()[0]
        assert _cos_err(emb_c[i][None], r[None]).max() < COS_TOL and _rel_err(emb_c[i], r) < 5e-4


def test_extract_driver_two_ranks_on_one_gpu(tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 -m wespeaker_amd.extract ... (both ranks on GPU 0 over
    gloo: WS_SHARE_GPU / WS_DIST_BACKEND) writes the same arks as the single-process run, bit for bit, plus
    the merged xvector.scp and extract.result of tools/extract_embedding.sh."""
    import subprocess
    import sys
    from wespeaker_amd import kaldi_io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mdir = str(tmp_path / "exp")
    synth.write_model_dir(mdir, "ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    n = len(_driver_corpus(tmp_path))
    base = [sys.executable, "-m", "wespeaker_amd.extract", "--exp_dir", mdir, "--model_path",
            os.path.join(mdir, "avg_model.pt"), "--data_type", "raw", "--data_list", str(tmp_path / "raw.list"),
            "--wavs_num", str(n), "--nj", "4", "--batch_size", "1"]
    # WS_COLLECTIVES_ON_GPU: the gather=True collectives carry GPU tensors like under nccl / RCCL (whose process
    # group has no CPU path -- the placement bug of round 2 could only show on a multi-GPU node)
    env = dict(os.environ, PYTHONPATH=root, WS_SHARE_GPU="1", WS_DIST_BACKEND="gloo", WS_COLLECTIVES_ON_GPU="1")
    r1 = subprocess.run(base + ["--store_dir", "one"], env=env, cwd=root, stdout=subprocess.PIPE,
                        stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r1.returncode == 0, r1.stdout[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29871"] + base[1:] +
                        ["--store_dir", "two", "--gather_npz", str(tmp_path / "gathered.npz")],
                        env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:]
    d1, d2 = os.path.join(mdir, "embeddings", "one"), os.path.join(mdir, "embeddings", "two")
    for j in range(4):
        assert open(os.path.join(d1, "xvector_%03d.ark" % j), "rb").read() == \
               open(os.path.join(d2, "xvector_%03d.ark" % j), "rb").read()
    m1, m2 = kaldi_io.read_vec_scp(os.path.join(d1, "xvector.scp")), kaldi_io.read_vec_scp(os.path.join(d2, "xvector.scp"))
    assert list(m1) == list(m2) == ["utt%02d" % i for i in range(n)]
    assert all(np.array_equal(m1[k], m2[k]) for k in m1)
    assert open(os.path.join(d2, "extract.result")).read().startswith("Successfully extract embedding")
    # gather=True: every row of the list, in list order, on rank 0 -- equal to what the arks hold
    g = np.load(str(tmp_path / "gathered.npz"))
    assert list(g["keys"]) == list(m2) and np.array_equal(g["emb"], np.stack([m2[k] for k in m2]))


# =============================== ragged batches: utterances of different lengths in one device batch
def _oracle_rows(forward, feats_list):
    return np.concatenate([np.asarray(forward(f[None])) for f in feats_list])


def test_fbank_ragged_rows_equal_single_utterance_rows(frontend):
    lens = [32000, 20000, 400, 27311, 559]
    wavs = [synth.synth_wav(700 + i, n) for i, n in enumerate(lens)]
    pad = np.full((len(lens), max(lens)), 12345, dtype=np.int16)         # junk in the padding
    for i, w in enumerate(wavs):
        pad[i, :lens[i]] = w
    for cmn in (False, True):
        got = frontend.fbank_ragged(torch.from_numpy(pad), lens, cmn=cmn).cpu().numpy()
        assert got.shape == (5, 198, 80)
        for i, w in enumerate(wavs):
            one = frontend.fbank(torch.from_numpy(w), cmn=cmn).cpu().numpy()[0]
            T = one.shape[0]
            assert np.array_equal(got[i, :T], one), (i, cmn)              # same kernel, same arithmetic
            assert not got[i, T:].any()                                   # padding rows are zero


RAGGED_CASES = [("ECAPA_TDNN_GLOB_c512", 192, [198, 150, 57, 399, 5, 201]),
                ("ECAPA_TDNN_c1024", 192, [198, 64, 230, 131]),
                ("ResNet34", 256, [198, 131, 9, 200, 64]),
                ("ResNet221", 256, [150, 57, 98]),
                ("CAMPPlus", 512, [603, 328, 201, 99, 57, 7]),
                # longer than the fused Res2 chain's 416-frame window (the 21-launch form) / many CAM segments:
                # what a real VoxCeleb list (4 - 20 s utterances) looks like
                ("ECAPA_TDNN_GLOB_c512", 192, [1203, 417, 798, 416, 500]),
                ("ECAPA_TDNN_c1024", 192, [600, 450]),
                ("ResNet34", 256, [1001, 640, 431]),
                ("CAMPPlus", 512, [1500, 1001, 777]),
                # <= 256 frames = <= 128 trunk frames: the one-kernel dense layer (cam_dense.hip) with per-utterance
                # lengths, one and two context segments (trunk lengths 125, 99, 66, 29, 4, 102), and a full 128
                ("CAMPPlus", 512, [250, 198, 131, 57, 7, 203]),
                ("CAMPPlus", 512, [256, 255, 201, 200]),
                # ... with the 4-row tail on 4x4x1 MFMAs: T' max = 99 (3 blocks + 3 rows; utterances ending inside a
                # block, inside the tail and before it), 100 (a 4-row tail), 35 (one block + 3)
                ("CAMPPlus", 512, [198, 197, 195, 193, 192, 150, 9]),
                ("CAMPPlus", 512, [200, 199, 131]),
                ("CAMPPlus", 512, [70, 69, 64, 33]),
                # 160 < T <= 208: the one-kernel attentive pooling (astp_fused.hip) with per-utterance lengths,
                # a full 208-frame window, and its shortest window
                ("ECAPA_TDNN_GLOB_c512", 192, [198, 150, 57, 208, 5, 203]),
                ("ECAPA_TDNN_c512", 192, [161, 100, 161]),
                # ... and its 10- and 7-row-block variants (T <= 160 / <= 112), incl. the shortest window it takes
                ("ECAPA_TDNN_GLOB_c512", 192, [160, 113, 129, 75]),
                ("ECAPA_TDNN_c512", 192, [112, 64, 100, 97]),
                ("ECAPA_TDNN_GLOB_c512", 192, [64, 64, 5])]


@pytest.mark.parametrize("name,E,lens", RAGGED_CASES)
@pytest.mark.parametrize("prec", ["fp32", "f16x3", "f16"])
def test_ragged_forward_rows_equal_the_batch1_oracle(name, E, lens, prec):
    """ws_forward_ragged: utterance b uses num_frames[b] rows of its padded slot; every row of the result
    equals the batch-1 oracle on that utterance alone (the reference's whole-utterance mode), on every
    back-end.  The padding rows are filled with NaN to prove that nothing reads them."""
    from oracle import campplus as ocam, resnet as oresnet
    sd = synth.synth_state_dict(name, 80, E, seed=42)
    model = _native(name, sd, E, max_batch=4, max_frames=max(lens))       # batch 6 > chunk 4: two chunks
    model.set_precision(prec)
    fwd = {"ECAPA": lambda f: oecapa.ecapa_forward(sd, f).numpy(),
           "ResNe": lambda f: oresnet.resnet_forward(sd, f, name).numpy(),
           "CAMPP": lambda f: ocam.campplus_forward(sd, f).numpy()}[name[:5]]
    feats = [np.random.RandomState(100 + i).randn(T, 80).astype(np.float32) for i, T in enumerate(lens)]
    pad = np.full((len(lens), max(lens), 80), np.nan, dtype=np.float32)
    for i, f in enumerate(feats):
        pad[i, :lens[i]] = f
    got = model.embed_ragged(torch.from_numpy(pad), lens).cpu().numpy()
    ref = _oracle_rows(fwd, feats)
    assert np.isfinite(got).all()
    tol = REL_TOL if prec != "f16" else (F16_REL_TOL_DEEP if name == "ResNet221" else F16_REL_TOL)
    assert _cos_err(got, ref).max() < COS_TOL, (_cos_err(got, ref), prec)
    assert _rel_err(got, ref).max() < tol, (_rel_err(got, ref), prec)
    # and against the engine's own uniform path on each utterance alone
    for i, f in enumerate(feats[:3]):
        out = model(torch.from_numpy(f[None]))
        one = (out[-1] if isinstance(out, tuple) else out).cpu().numpy()
        assert _rel_err(got[i:i + 1], one).max() < (1e-5 if prec == "fp32" else 2 * tol), (i, prec)
    model.check_range()
    with pytest.raises(Exception):
        model.embed_ragged(torch.from_numpy(pad), [max(lens) + 1] + lens[1:])       # longer than the slot
    with pytest.raises(Exception):
        model.embed_ragged(torch.from_numpy(pad), [0] + lens[1:])                   # shorter than the model minimum


@pytest.mark.parametrize("name,E", [("ECAPA_TDNN_GLOB_c512", 192), ("ECAPA_TDNN_c512", 192), ("ResNet34", 256),
                                    ("ResNet221", 256)])
def test_full_size_ragged_batch_on_the_persistent_kernel(name, E):
    """A ragged batch big enough for the persistent fp32 GEMM (gemm_f32_stream.hip, MASK: output pixels at or beyond
    their utterance's own width are stored as zeros in its epilogue -- tiles and 64x64 row units, the plain forms at
    run time, the residual / 3x3 forms as masked twins; until round 6 every ragged batch ran on the tile kernels).
    250 utterances of 70 .. 230 frames in 230-frame slots, NaN in the padding: spot rows against the batch-1 oracle
    (the reference's whole-utterance mode, bin/extract.py:95) and against the engine's own uniform batch-1 forward on
    that utterance alone (tile kernels: a batch of one never reaches the persistent kernel)."""
    from oracle import resnet as oresnet
    sd = synth.synth_state_dict(name, 80, E, seed=42)
    B, TMAX = 250, 230                  # (57 500 rows: the row units' last strip has 28 rows)
    model = _native(name, sd, E, max_batch=B, max_frames=TMAX)
    fwd = (lambda f: oecapa.ecapa_forward(sd, f).numpy()) if name.startswith("ECAPA") else \
          (lambda f: oresnet.resnet_forward(sd, f, name).numpy())
    rng = np.random.RandomState(3)
    lens = rng.randint(70, TMAX + 1, size=B)
    lens[0], lens[1], lens[B - 1] = TMAX, 70, 199           # a full slot, the shortest, a strip-boundary case
    pad = np.full((B, TMAX, 80), np.nan, dtype=np.float32)
    feats = {}
    for i in range(B):
        f = np.random.RandomState(1000 + i).randn(lens[i], 80).astype(np.float32)
        pad[i, :lens[i]] = f
        feats[i] = f
    got = model.embed_ragged(torch.from_numpy(pad), [int(x) for x in lens]).cpu().numpy()
    assert np.isfinite(got).all()
    rows = [0, 1, 2, 77, 128, B - 2, B - 1]
    for i in rows:
        out = model(torch.from_numpy(feats[i][None]))
        one = (out[-1] if isinstance(out, tuple) else out).cpu().numpy()
        assert _rel_err(got[i:i + 1], one).max() < 1e-5, (i, lens[i])
    ref = _oracle_rows(fwd, [feats[i] for i in rows[:2]])
    assert _cos_err(got[rows[:2]], ref).max() < COS_TOL and _rel_err(got[rows[:2]], ref).max() < REL_TOL
    # the dispatcher really sent masked layers to the persistent kernel
    from wespeaker_amd.engine import dispatch_log, dispatch_report
    dispatch_log(True, clear=True)
    try:
        model.embed_ragged(torch.from_numpy(pad), [int(x) for x in lens])
        lines = dispatch_report()
    finally:
        dispatch_log(False, clear=True)
    hits = [l for l in lines if "+mask" in l and "gemm_f32_stream_kernel" in l]
    assert hits, lines[:8]
    if name == "ResNet221":                                 # 3x3, residual and plain 1x1 forms
        assert any("conv" in l for l in hits) and any("+res" in l for l in hits), hits


@pytest.mark.parametrize("name", ["ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_c512", "ECAPA_TDNN_GLOB_c1024"])
@pytest.mark.parametrize("B,T", [(256, 209), (128, 330), (128, 398), (64, 798)])
def test_attentive_pooling_kernel_on_segments_of_long_utterances(name, B, T):
    """astp_fused_kernel beyond 208 frames (round 6; pooling_layers.py:119-144): workgroup (u, seg) takes <= 208
    frames of utterance u and leaves its online-softmax tuple per channel, astp_combine_kernel merges the tuples
    (2 x 105, 2 x 165, 2 x 199, 4 x 200 frames here: all three row-block forms).  Uniform and ragged batches big
    enough for the cost model to pick the kernel; spot rows against the batch-1 oracle, and the dispatch log must
    name the kernel.  The same shapes put the time-tiled Res2 chain on the windows chain_pick_mtw64 / chain_pick_mtw128
    choose (160 .. 256 rows; ECAPA-1024: 160 or 208)."""
    sd = synth.synth_state_dict(name, 80, 192, seed=42)
    model = _native(name, sd, 192, max_batch=B, max_frames=T)
    f = np.random.RandomState(T).randn(B, T, 80).astype(np.float32)
    from wespeaker_amd.engine import dispatch_log, dispatch_report
    dispatch_log(True, clear=True)
    try:
        out = model(torch.from_numpy(f))
        got = (out[-1] if isinstance(out, tuple) else out).cpu().numpy()
        lines = dispatch_report()
    finally:
        dispatch_log(False, clear=True)
    assert any("astp_fused_kernel" in l for l in lines), lines[:6]
    rows = [0, B // 2, B - 1]
    ref = oecapa.ecapa_forward(sd, f[rows]).numpy()
    assert _cos_err(got[rows], ref).max() < COS_TOL and _rel_err(got[rows], ref).max() < REL_TOL, (B, T)
    lens = np.random.RandomState(B).randint(T - 150, T + 1, size=B)
    lens[0], lens[B - 1] = T, T - 150
    pad = np.full((B, T, 80), np.nan, dtype=np.float32)
    for i in range(B):
        pad[i, :lens[i]] = f[i, :lens[i]]
    rag = model.embed_ragged(torch.from_numpy(pad), [int(x) for x in lens]).cpu().numpy()
    assert np.isfinite(rag).all()
    ref = _oracle_rows(lambda x: oecapa.ecapa_forward(sd, x).numpy(), [f[i, :lens[i]] for i in rows])
    assert _cos_err(rag[rows], ref).max() < COS_TOL and _rel_err(rag[rows], ref).max() < REL_TOL, (B, T)


@pytest.mark.parametrize("name,E", [("ECAPA_TDNN_GLOB_c512", 100), ("ResNet18", 37), ("CAMPPlus", 200)])
def test_small_m_gemm_edges(name, E):
    """small_m_gemm_f32_kernel (csrc/small_m_gemm.hip): the M = batch linear layers of the fp32 back-end in one launch
    -- ECAPA's final BN + Linear and global-context bias (ecapa_tdnn.py:214-218, pooling_layers.py:128-133), the ResNets'
    seg_1 (resnet.py:196-202), CAM++'s dense layer (campplus.py:322-330).  Embedding sizes that are NOT multiples of its
    16-column tile and batches that are not multiples of its 16-row tile (1, 3, 17, 33), against the oracle."""
    from oracle import campplus as ocam, resnet as oresnet
    sd = synth.synth_state_dict(name, 80, E, seed=42)
    model = _native(name, sd, E, max_batch=33, max_frames=120)
    fwd = {"ECAPA": lambda f: oecapa.ecapa_forward(sd, f).numpy(),
           "ResNe": lambda f: oresnet.resnet_forward(sd, f, name).numpy(),
           "CAMPP": lambda f: ocam.campplus_forward(sd, f).numpy()}[name[:5]]
    for B in (1, 3, 17, 33):
        f = np.random.RandomState(B).randn(B, 100, 80).astype(np.float32)
        out = model(torch.from_numpy(f))
        got = (out[-1] if isinstance(out, tuple) else out).cpu().numpy()
        ref = fwd(f)
        assert got.shape == ref.shape == (B, E)
        assert _cos_err(got, ref).max() < COS_TOL and _rel_err(got, ref).max() < REL_TOL, (name, B)


@pytest.mark.parametrize("name", ["ResNet50", "ResNet101"])
def test_bottleneck_conv3_conv1_fusion_partial_pixel_blocks(name):
    """bneck_c3c1_kernel (csrc/bneck_fuse.hip; resnet.py:72-107): conv3 + residual + ReLU of a Bottleneck block and conv1 +
    ReLU of the next block in one launch (fp32, stages 1 - 2, incl. the stage 1 -> 2 transition).  A wavefront owns 32
    pixels; ONE utterance of an odd number of frames makes the pixel count 16 (mod 32) in both stages, so the last
    block of pixels is half empty (dropped by the buffer bounds).  Against the oracle; and the ragged entry point with
    the same lengths -- the kernel's masked twin -- gives the same rows."""
    from oracle import resnet as oresnet
    sd = synth.synth_resnet_state_dict(name, 80, 256, seed=42)
    model = _native(name, sd, 256, max_batch=4, max_frames=200)
    for B, T in ((1, 131), (1, 57), (3, 99), (4, 198)):
        assert (B * 80 * T) % 32 in (0, 16)
        f = np.random.RandomState(T).randn(B, T, 80).astype(np.float32)
        got = model(torch.from_numpy(f))[-1].cpu().numpy()
        ref = oresnet.resnet_forward(sd, f, name).numpy()
        assert _cos_err(got, ref).max() < COS_TOL and _rel_err(got, ref).max() < REL_TOL, (B, T)
        rag = model.embed_ragged(torch.from_numpy(f), [T] * B).cpu().numpy()      # per-utterance lengths: the masked twin
        assert _rel_err(rag, got).max() < 1e-5, (B, T)
    model.check_range()


def test_campplus_direct_conv_and_tile_kernels_same_rows():
    """conv3x3_direct_f32_kernel against the implicit-GEMM tile kernels on CAM++'s FCM head (campplus.py:245-330; maps
    of 40 / 20 / 10 mel rows, strides (1,1) and (2,1)).  The direct kernel takes a layer from 131 072 pixels on: a
    batch of 96 utterances runs every FCM layer on it (8 x 16-pixel patches: the 20- and 10-row maps leave the last
    patch row partly empty, 198 / 161 frames the last patch column), batches of 4 run the same layers on the tile
    kernels.  The same k order per pixel: the embeddings agree bit for bit; spot rows against the oracle.
    (Round 6 measured a second patch shape, 4 x 32, for the 20- / 10-row maps: 10 - 19 % fewer patches, the same
    launch times -- 418 vs 410 us -- because the halo DMA, not the MFMA count, sets the patch time; not kept.)"""
    from oracle import campplus as ocam
    sd = synth.synth_state_dict("CAMPPlus", 80, 192, seed=11)
    model = _native("CAMPPlus", sd, 192, max_batch=96, max_frames=200)
    for T in (198, 161):
        f = np.random.RandomState(T).randn(96, T, 80).astype(np.float32)
        big = model(torch.from_numpy(f)).cpu().numpy()
        small = np.concatenate([model(torch.from_numpy(f[i:i + 4])).cpu().numpy() for i in (0, 44, 92)])
        assert np.array_equal(big[[0, 1, 2, 3, 44, 45, 46, 47, 92, 93, 94, 95]], small), T
        ref = ocam.campplus_forward(sd, f[[0, 95]]).numpy()
        assert _cos_err(big[[0, 95]], ref).max() < COS_TOL and _rel_err(big[[0, 95]], ref).max() < REL_TOL, T
    model.check_range()


def test_res2_chain_four_wavefront_kernel_sizes():
    """res2_chain4_kernel (csrc/res2_chain4.hip; ecapa_tdnn.py:58-78): the fp32 chain of ECAPA-512 takes it when the
    batch fills the chip (> 64 utterances) and 129 <= T <= 208 -- one instantiation per number of 16-row tiles (9 .. 13).
    Each of them, uniform and as a ragged batch whose shorter utterances end inside other tiles (down to T = 70: most
    tiles are past the end, dropped by the buffer bounds), against the batch-1 oracle on spot rows and against the
    engine's own small-batch path (the eight-wavefront kernel: same k order, so the rows must agree to fp32 noise)."""
    sd = synth.synth_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    B = 72
    model = _native("ECAPA_TDNN_GLOB_c512", sd, 192, max_batch=B, max_frames=208)
    small = _native("ECAPA_TDNN_GLOB_c512", sd, 192, max_batch=4, max_frames=208)
    for T in (129, 145, 170, 190, 198, 208):
        f = np.random.RandomState(T).randn(B, T, 80).astype(np.float32)
        got = model(torch.from_numpy(f))[-1].cpu().numpy()
        rows = [0, 17, B - 1]
        ref = oecapa.ecapa_forward(sd, f[rows]).numpy()
        assert _cos_err(got[rows], ref).max() < COS_TOL and _rel_err(got[rows], ref).max() < REL_TOL, T
        other = small(torch.from_numpy(f[:4]))[-1].cpu().numpy()
        assert _rel_err(got[:4], other).max() < 1e-5, T
        assert np.isfinite(got).all()
    T = 200
    lens = [200 - (i * 37) % 131 for i in range(B)]                 # 70 .. 200, every tile count from 5 to 13
    assert min(lens) == 70 and max(lens) == 200
    feats = [np.random.RandomState(500 + i).randn(n, 80).astype(np.float32) for i, n in enumerate(lens)]
    pad = np.full((B, T, 80), np.nan, dtype=np.float32)
    for i, x in enumerate(feats):
        pad[i, :lens[i]] = x
    got = model.embed_ragged(torch.from_numpy(pad), lens).cpu().numpy()
    assert np.isfinite(got).all()
    rows = [0, 1, 2, 3, 40, B - 1]
    ref = _oracle_rows(lambda x: oecapa.ecapa_forward(sd, x).numpy(), [feats[i] for i in rows])
    assert _cos_err(got[rows], ref).max() < COS_TOL and _rel_err(got[rows], ref).max() < REL_TOL
    for i in rows[:4]:
        one = small(torch.from_numpy(feats[i][None]))[-1].cpu().numpy()
        assert _rel_err(got[i:i + 1], one).max() < 1e-5, i


def test_feat_lists_of_precomputed_kaldi_features(tmp_path, golden_dir):
    """`data_type: feat` (dataset/dataset.py:136-273, processor.parse_feat :171-196, bin/extract.py:112-139): json lines
    {key, feat: ark:offset, spk} of RAW Kaldi fbank matrices; CMVN (test_conf cmvn / cmvn_args) and the forward on the
    GPU.  Whole-utterance mode on utterances of different lengths == the batch-1 oracle on apply_cmvn'd features, with
    the default CMVN and with norm_var; the random-chunk mode (batch_size > 1) crops / tiles to num_frms frames; the
    (True, True) embeddings of the first three utterances equal the reference module's golden."""
    import json
    from wespeaker_amd import kaldi_io
    from wespeaker_amd import extract as wx
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    sd = synth.synth_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    model = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, feat_dim=80, embed_dim=192, max_batch=8, max_frames=260)
    ex = wx.GpuExtractor(model, Frontend(16000, 80, device=model.device))
    ns = [32000, 32000, 32000, 24000, 41000, 9000, 32000, 30000, 16160, 28000]
    raws = [ofbank.speaker_features(synth.synth_wav(i, n), cmn=False) for i, n in enumerate(ns)]
    ark = str(tmp_path / "feats.ark")
    lines = []
    with open(ark, "wb") as f:
        for i, m in enumerate(raws):
            lines.append(json.dumps({"key": "utt%d" % i, "feat": "%s:%d" % (ark, kaldi_io.write_mat(f, "utt%d" % i, m)),
                                     "spk": "s%d" % (i % 3)}))
    fwd = lambda x: oecapa.ecapa_forward(sd, x).numpy()          # noqa: E731
    for cmvn in ((True, False), (True, True), (False, False)):
        keys, emb = wx.extract_list("feat", lines, ex, batch_size=1, max_batch=8, cmvn=cmvn)
        assert keys == ["utt%d" % i for i in range(len(ns))]
        ref = _oracle_rows(fwd, [ofbank.apply_cmvn(m[None], *cmvn)[0] for m in raws])
        assert _cos_err(emb, ref).max() < COS_TOL and _rel_err(emb, ref).max() < 5e-4, cmvn
    g = np.load(os.path.join(golden_dir, "cmvn_ref.npz"))
    keys, emb = wx.extract_list("feat", lines[:3], ex, batch_size=1, cmvn=(True, True))
    assert _rel_err(emb, g["ecapa512_m1v1/emb"]).max() < 5e-4
    # random-chunk mode: 100 frames per utterance, cropped at the seeded start (long) or tiled (the 55-frame one)
    keys, emb = wx.extract_list("feat", lines, ex, batch_size=4, num_frms=100, seed=5, cmvn=(True, False))
    chunks = []
    for i, m in enumerate(raws):
        c = wx.random_chunk(m, "utt%d" % i, 100, 5) if m.shape[0] >= 100 else np.tile(m, (100 // m.shape[0] + 1, 1))[:100]
        chunks.append(ofbank.apply_cmvn(c[None], True, False)[0])
    assert raws[5].shape[0] < 100 and all(c.shape == (100, 80) for c in chunks)
    ref = _oracle_rows(fwd, chunks)
    assert _cos_err(emb, ref).max() < COS_TOL and _rel_err(emb, ref).max() < 5e-4


@pytest.mark.parametrize("name,E", [("ECAPA_TDNN_GLOB_c512", 192), ("ResNet34", 256), ("CAMPPlus", 512)])
def test_ragged_extract_from_waveforms(frontend, name, E):
    """ws_extract_ragged (wav -> fbank -> CMN -> forward on a padded batch) against the oracle run on every
    waveform alone."""
    from oracle import campplus as ocam, resnet as oresnet
    sd = synth.synth_state_dict(name, 80, E, seed=42)
    model = _native(name, sd, E, max_batch=3, max_frames=250)
    fwd = {"ECAPA": lambda f: oecapa.ecapa_forward(sd, f).numpy(),
           "ResNe": lambda f: oresnet.resnet_forward(sd, f, name).numpy(),
           "CAMPP": lambda f: ocam.campplus_forward(sd, f).numpy()}[name[:5]]
    ns = [32000, 24000, 40000, 31840, 16160, 32000, 9000]
    wavs = [synth.synth_wav(800 + i, n) for i, n in enumerate(ns)]
    pad = np.zeros((len(ns), max(ns)), dtype=np.int16)
    for i, w in enumerate(wavs):
        pad[i, :ns[i]] = w
    got = model.extract_ragged(frontend, torch.from_numpy(pad), ns).cpu().numpy()
    ref = _oracle_rows(fwd, [ofbank.speaker_features(w) for w in wavs])
    assert _cos_err(got, ref).max() < COS_TOL and _rel_err(got, ref).max() < 5e-4, _rel_err(got, ref)
    model.set_precision("f16")
    got16 = model.extract_ragged(frontend, torch.from_numpy(pad), ns).cpu().numpy()
    assert _cos_err(got16, ref).max() < COS_TOL and _rel_err(got16, ref).max() < F16_REL_TOL_DEEP


def test_driver_batches_similar_lengths_through_the_ragged_path(tmp_path):
    """Whole-utterance lists with all-different lengths (a real test set) still run as device batches: the
    driver pads utterances within 12 % of each other into ws_extract_ragged calls; rows = batch-1 oracle."""
    import json
    import wespeaker_amd
    from wespeaker_amd import extract as wx
    mdir = str(tmp_path / "exp")
    sd = synth.write_model_dir(mdir, "ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    lengths = [32000 + 137 * i for i in range(9)] + [48000 - 211 * i for i in range(6)]
    lines = []
    for i, L in enumerate(lengths):
        p = str(tmp_path / ("r%02d.wav" % i))
        synth.write_wav(p, synth.synth_wav(900 + i, L))
        lines.append(json.dumps({"key": "utt%02d" % i, "wav": p, "spk": "s"}))
    spk = wespeaker_amd.load_model(mdir)
    calls = []
    ex = wx.GpuExtractor(spk.model, spk._frontend(16000))
    orig = ex.submit
    ex.submit = lambda utts: (calls.append([int(u.shape[0]) for u in utts]), orig(utts))[1]
    keys, emb = wx.extract_entries(wx.iter_entries("raw", lines), ex, batch_size=1, max_batch=8)
    assert keys == ["utt%02d" % i for i in range(15)]
    assert len(calls) < 15 and max(len(c) for c in calls) >= 6              # real batches of different lengths
    assert all(max(c) <= min(c) * 1.13 for c in calls)
    ref = np.stack([oecapa.ecapa_forward(sd, ofbank.speaker_features(synth.synth_wav(900 + i, L))[None]).numpy()[0]
                    for i, L in enumerate(lengths)])
    assert _cos_err(emb, ref).max() < COS_TOL and _rel_err(emb, ref).max() < 5e-4


def test_driver_with_two_lanes_equals_one_engine(tmp_path):
    """GpuExtractor over a SpeakerModelLanes: staging slot i = lane i (own engine + stream), so upload, forward and
    download of consecutive batches overlap.  Same batches, same kernels: the rows are the one-engine driver's bits,
    in list order -- uniform files (ws_extract) and all-different lengths (ragged batches)."""
    from wespeaker_amd import SpeakerModelLanes, extract as wx
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    lines = {"uniform": [], "varied": []}
    for i in range(45):
        for tag, n in (("uniform", 32000), ("varied", 24000 + 353 * i)):
            p = str(tmp_path / ("%s%02d.wav" % (tag[0], i)))
            synth.write_wav(p, synth.synth_wav(300 + i, n))
            lines[tag].append("utt%02d %s" % (i, p))
    fe = Frontend(16000, 80)
    one = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, max_batch=8, max_frames=250)
    two = SpeakerModelLanes("ECAPA_TDNN_GLOB_c512", sd, lanes=2, max_batch=8, max_frames=250)
    for prec in ("fp32", "f16"):                # (the driver's default is two lanes for every back-end since round 4)
        one.set_precision(prec)
        two.set_precision(prec)
        for tag in ("uniform", "varied"):
            k1, e1 = wx.extract_list("scp", lines[tag], wx.GpuExtractor(one, fe), batch_size=1, max_batch=8, num_workers=2)
            k2, e2 = wx.extract_list("scp", lines[tag], wx.GpuExtractor(two, fe), batch_size=1, max_batch=8, num_workers=2)
            assert k1 == k2 == ["utt%02d" % i for i in range(45)]
            assert np.array_equal(e1, e2), (prec, tag)


def test_driver_head_first_pipeline_and_its_fallback(tmp_path):
    """extract_files on a list long enough for its head-first start-up (>= 6 x max_batch files: the first 2 x max_batch
    are probed, planned and sent to the GPU alone, the decode thread prepares the rest behind them): uniform and
    all-different lengths equal the general path (extract_entries) row for row, in list order.  A file BEHIND the head
    that the native loader cannot take (8-bit PCM) makes the call start over on the general path -- same rows again."""
    import struct
    from wespeaker_amd import extract as wx
    from wespeaker_amd.engine import Frontend, NativeSpeakerModel
    sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    fe = Frontend(16000, 80)
    model = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, max_batch=8, max_frames=260)
    n = 70                                                   # head = 16 files, rest = 54
    for tag in ("uniform", "varied"):
        lines = []
        for i in range(n):
            p = str(tmp_path / ("%s%02d.wav" % (tag[0], i)))
            synth.write_wav(p, synth.synth_wav(700 + i, 32000 if tag == "uniform" else 24000 + 229 * ((i * 37) % n)))
            lines.append("utt%02d %s" % (i, p))
        kp = wx.split_path_list("scp", lines)
        k1, e1 = wx.extract_files(kp[0], kp[1], wx.GpuExtractor(model, fe), batch_size=1, max_batch=8, threads=4)
        k2, e2 = wx.extract_entries(wx.iter_entries("scp", lines), wx.GpuExtractor(model, fe), batch_size=1,
                                    max_batch=8, num_workers=2)
        assert k1 == k2 == ["utt%02d" % i for i in range(n)]
        assert _rel_err(e1, e2).max() < 1e-5, tag              # (other batch compositions: fp32 summation order only)
        rows = [0, 15, 16, 40, n - 1]                          # head, the head / rest seam, rest
        ref = _oracle_rows(lambda f: oecapa.ecapa_forward(sd, f).numpy(),
                           [ofbank.speaker_features(wx.load_pcm16_fast(kp[1][i])[0]) for i in rows])
        assert _cos_err(e1[rows], ref).max() < COS_TOL and _rel_err(e1[rows], ref).max() < 5e-4, tag
    # an 8-bit PCM file at position 50: the head's batches are already on the GPU when the rest is probed
    bad = str(tmp_path / "u50.wav")
    pcm8 = ((synth.synth_wav(750, 32000).astype(np.int32) >> 8) + 128).astype(np.uint8)
    with open(bad, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + pcm8.size) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 16000, 1, 8)
                + b"data" + struct.pack("<I", pcm8.size) + pcm8.tobytes())
    assert wx.extract_files(kp[0], kp[1][:50] + [bad] + kp[1][51:], wx.GpuExtractor(model, fe), batch_size=1, max_batch=8,
                            threads=4) is None
    lines[50] = "utt50 " + bad
    k3, e3 = wx.extract_list("scp", lines, wx.GpuExtractor(model, fe), batch_size=1, max_batch=8, num_workers=2)
    assert k3 == k1 and np.isfinite(e3).all() and _rel_err(np.delete(e3, 50, 0), np.delete(e1, 50, 0)).max() < 1e-5


# ------------------------------------------------------------------------------------------ dispatch tables
def _dispatch_cases():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import dispatch_tables
    return dispatch_tables


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(5))
@pytest.mark.parametrize("prec", ["fp32", "f16"])
def test_dispatch_tables_are_pinned(case, prec):
    """Which kernel every conv / linear problem of the five BASELINE workloads is given
    (ws_debug_dispatch_report) equals tests/golden/dispatch_<model>_<prec>.txt.  A layer that falls off the fast
    kernels costs speed, not correctness: this is the test that makes it a reviewed change
    (`python tools/dispatch_tables.py --write` regenerates the tables)."""
    dt = _dispatch_cases()
    model, E, batch, chunk = dt.CASES[case]
    got = dt.table(model, E, batch, chunk, prec)
    want = [l.rstrip("\n") for l in open(dt.golden_path(model, prec)) if l.strip()]
    assert got == want, "dispatch of %s/%s changed:\n%s" % (
        model, prec, "\n".join(sorted(set(got) ^ set(want))))
    # and the log is really off again: nothing is noted by a later forward
    from wespeaker_amd.engine import dispatch_report
    assert dispatch_report() == []


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "f16"])
def test_one_minute_utterance(prec):
    """A single 60 s utterance (T = 5998: 47 row tiles of one image, every per-utterance kernel on its long-loop
    form, engine workspace grown from a 2 s finalisation) against the oracle."""
    sd = synth.synth_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    model = _native("ECAPA_TDNN_GLOB_c512", sd, 192, max_batch=2, max_frames=198)
    model.set_precision(prec)
    f = np.random.RandomState(60).randn(1, 5998, 80).astype(np.float32)
    out = model(torch.from_numpy(f))
    got = (out[-1] if isinstance(out, tuple) else out).cpu().numpy()
    ref = oecapa.ecapa_forward(sd, f).numpy()
    assert _cos_err(got, ref).max() < COS_TOL
    assert _rel_err(got, ref).max() < (REL_TOL if prec == "fp32" else F16_REL_TOL)
    model.check_range()


@pytest.mark.gpu
def test_engine_chunk_beyond_32bit_offsets_is_refused():
    """The binary16 convolution kernels address a tensor with 32-bit element offsets: an engine chunk whose
    stage-1 activation map would pass 2^31 elements is refused at finalisation (WS_ERR_CAPACITY), not computed
    wrongly; the same model with a sane chunk works."""
    from wespeaker_amd._lib import NativeError
    from wespeaker_amd.engine import NativeSpeakerModel
    sd = synth.synth_state_dict("ResNet221", 80, 256, seed=42)
    with pytest.raises(NativeError, match="2\\^31"):
        NativeSpeakerModel("ResNet221", sd, feat_dim=80, embed_dim=256, max_batch=1100, max_frames=198)
    m = NativeSpeakerModel("ResNet221", sd, feat_dim=80, embed_dim=256, max_batch=2, max_frames=198)
    assert m.embed(torch.zeros(1, 198, 80, device="cuda")).shape == (1, 256)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "f16"])
def test_res2_time_tiles_are_bit_identical_to_whole_utterance_windows(tmp_path, prec):
    """The fused Res2 chain cuts an utterance into time tiles with a 7 * dilation halo when it is longer than a
    workgroup's rows or when the batch is small (res2_fused.hip).  A tile computes its owned rows from exactly the
    same operands in the same order as the whole-utterance window does, so the embeddings must be IDENTICAL bit for
    bit whichever form runs: small windows (WS_CHAIN_SMALL=1: three tiles per 198-frame utterance, five at 301)
    against big ones (=0), in two processes because the switch is read once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "from wespeaker_amd import NativeSpeakerModel\n"
        "from fixtures import synth\n"
        "out = {}\n"
        "for name, E in (('ECAPA_TDNN_GLOB_c512', 192), ('ECAPA_TDNN_c1024', 192)):\n"
        "    sd = synth.synth_state_dict(name, 80, E, seed=42)\n"
        "    m = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=E, max_batch=4, max_frames=301)\n"
        "    m.set_precision(sys.argv[2])\n"
        "    for T in (198, 301, 57):\n"
        "        f = torch.from_numpy(np.random.RandomState(T).randn(3, T, 80).astype(np.float32)).cuda()\n"
        "        out['%s_%d' % (name, T)] = m.embed(f).cpu().numpy()\n"
        "    lens = [301, 120, 250]\n"
        "    f = torch.from_numpy(np.random.RandomState(9).randn(3, 301, 80).astype(np.float32)).cuda()\n"
        "    out[name + '_ragged'] = m.embed_ragged(f, lens).cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n")
    res = []
    for small in ("0", "1"):
        path = str(tmp_path / ("small%s.npz" % small))
        env = dict(os.environ, PYTHONPATH=root, WS_CHAIN_SMALL=small)
        r = subprocess.run([sys.executable, "-c", code, path, prec], env=env, cwd=root, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        res.append(np.load(path))
    assert sorted(res[0].files) == sorted(res[1].files) and len(res[0].files) == 8
    for k in res[0].files:
        assert np.isfinite(res[0][k]).all()
        assert np.array_equal(res[0][k], res[1][k]), (k, np.abs(res[0][k] - res[1][k]).max())
