"""CPU: pin the oracle (oracle/*.py) against golden vectors produced by the REFERENCE ITSELF
(oracle/make_golden.py ran the reference's nn.Modules / TwoCovPLDA / native fbank in the build
container).  If /root/reference is present, also re-check against the live reference."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import ecapa as oecapa
from oracle import fbank as ofbank
from oracle import plda as oplda
from oracle import ref_shim
from fixtures import synth


def test_fbank_restatement_vs_reference_native(golden_dir):
    g = np.load(os.path.join(golden_dir, "fbank_ref_native.npz"))
    for k, idx in enumerate(g["utt_idx"]):
        mine = ofbank.speaker_features(synth.synth_wav(int(idx)), cmn=False)
        ref = g["logmel"][k]
        assert mine.shape == ref.shape == (198, 80)        # 2 s @ 16 kHz -> 198 frames
        # the reference's native code uses a hand-rolled float FFT / logf: agreement to ~2e-4
        assert np.abs(mine - ref).max() < 5e-4
        assert np.abs(mine - ref).mean() < 2e-5
    short = ofbank.speaker_features(synth.synth_wav(int(g["short_idx"]), int(g["short_len"])), cmn=False)
    assert short.shape == g["short_logmel"].shape == (23, 80)
    assert np.abs(short - g["short_logmel"]).max() < 5e-4


def test_fbank_restatement_vs_reference_native_other_rates(golden_dir):
    """8 kHz (the SRE recipe's rate, examples/sre/v2/conf/resnet.yaml:31), 32 kHz, 48 kHz: the restatement against the
    reference's own native fbank (frontend/fbank.h sizes its FFT from the frame: 256 / 1024 / 2048 points)."""
    g = np.load(os.path.join(golden_dir, "fbank_ref_native_rates.npz"))
    assert len(g.files) == 5
    for key in g.files:
        rate, bins, n = (int(t[1:]) for t in key.split("_"))
        wav = synth.synth_wav(21 + rate // 8000, n)
        mine = ofbank.kaldi_fbank(wav.astype(np.float32), num_mel_bins=bins, sample_frequency=rate,
                                  window_type="hamming")
        assert mine.shape == g[key].shape == (1 + (n - rate // 40) // (rate // 100), bins)
        assert np.abs(mine - g[key]).max() < 5e-4, key


def test_fbank_edge_cases():
    assert ofbank.kaldi_fbank(np.zeros(399, np.float32), window_type="hamming").shape == (0, 80)
    assert ofbank.kaldi_fbank(np.zeros(400, np.float32), window_type="hamming").shape == (1, 80)
    z = ofbank.kaldi_fbank(np.zeros(1000, np.float32), window_type="hamming")
    assert np.allclose(z, np.log(np.float32(1.1920929e-07)))          # floor at eps
    x = synth.synth_wav(1).astype(np.float32)
    a = ofbank.kaldi_fbank(x, window_type="hamming")
    b = ofbank.kaldi_fbank(np.stack([x, -x]), window_type="hamming")   # channel 0 only
    assert np.array_equal(a, b)
    c = ofbank.kaldi_fbank(x, window_type="hamming", cmn=True)
    assert np.abs(c.mean(0)).max() < 1e-4
    p = ofbank.kaldi_fbank(x, window_type="povey")
    assert np.abs(p - a).max() > 1e-3                                 # the window matters
    # [-1,1] floats scaled by 1<<15 (processor.py:516) == int16-range input
    d = ofbank.kaldi_fbank((x / 32768.0) * np.float32(32768.0), window_type="hamming")
    assert np.abs(d - a).max() < 1e-3


@pytest.mark.parametrize("name", ["ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_c512",
                                  "ECAPA_TDNN_GLOB_c1024", "ECAPA_TDNN_c1024"])
def test_ecapa_restatement_vs_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "ecapa_ref.npz"))
    sd = synth.synth_ecapa_state_dict(name, 80, 192, seed=42)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
    emb = oecapa.ecapa_forward(sd, feats).numpy()
    ref = g[name + "/emb"]
    assert emb.shape == ref.shape == (2, 192)
    assert np.abs(emb - ref).max() <= 2e-4 * np.abs(ref).max()
    emb_s = oecapa.ecapa_forward(sd, feats[:, :57]).numpy()
    assert np.abs(emb_s - g[name + "/emb_T57"]).max() <= 2e-4 * np.abs(ref).max()


def test_ecapa_emb_bn_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "ecapa_ref.npz"))
    sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_c512", 80, 256, emb_bn=True, seed=5)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
    emb = oecapa.ecapa_forward(sd, feats).numpy()
    ref = g["ECAPA_TDNN_c512_embbn/emb"]
    assert np.abs(emb - ref).max() <= 2e-4 * np.abs(ref).max()


def test_plda_restatement_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "plda_ref.npz"))
    emb, _ = synth.synth_embeddings(40, 192, seed=11)
    for nl in (False, True):
        p = synth.synth_plda(192, seed=7, normalize_length=nl)
        tag = "nl%d" % int(nl)
        tr = np.stack([oplda.transform_embedding(p, e) for e in emb])
        assert np.abs(tr - g[tag + "/transformed"]).max() < 1e-12
        for n in (1, 3):
            ref = g["%s/llr_n%d" % (tag, n)]
            loop = np.array([[oplda.log_likelihood_ratio(p, tr[i], tr[20 + j], n) for j in range(20)]
                             for i in range(20)])
            assert np.abs(loop - ref).max() < 1e-10
            mat = oplda.llr_matrix_vectorised(p, tr[:20], np.full(20, n), tr[20:])
            assert np.abs(mat - ref).max() < 1e-9


@pytest.mark.skipif(not ref_shim.available(), reason="reference checkout not present (GPU box)")
def test_live_reference_agrees_with_oracle():
    """In the build container the reference is importable: bit-compare the restatement."""
    name = "ECAPA_TDNN_GLOB_c512"
    sd = synth.synth_ecapa_state_dict(name, 80, 192, seed=3)
    m = ref_shim.ref_model(name, feat_dim=80, embed_dim=192, pooling_func="ASTP")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m.eval()
    feats = torch.from_numpy(np.stack([ofbank.speaker_features(synth.synth_wav(5, 16000))]))
    with torch.no_grad():
        ref = m(feats)[-1]
    mine = oecapa.ecapa_forward(sd, feats)
    assert torch.equal(ref, mine)
    p = synth.synth_plda(64, seed=1, normalize_length=True)
    rp = ref_shim.ref_plda(p)
    x = np.random.RandomState(0).randn(64)
    assert np.allclose(rp.transform_embedding(x), oplda.transform_embedding(p, x), atol=1e-13)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref",
                                                    "libref_fbank.so")),
                    reason="oracle/_ref not built")
def test_reference_native_fbank_library_runs():
    lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_fbank.so"))
    lib.ref_fbank.restype = ctypes.c_int
    x = synth.synth_wav(7).astype(np.float32)
    out = np.zeros((300, 80), np.float32)
    n = lib.ref_fbank(x.ctypes.data_as(ctypes.c_void_p), x.shape[0], 80, 16000,
                      out.ctypes.data_as(ctypes.c_void_p), 300)
    assert n == 198
    mine = ofbank.speaker_features(synth.synth_wav(7), cmn=False)
    assert np.abs(mine - out[:n]).max() < 5e-4


@pytest.mark.parametrize("name,kw", [("ResNet18", {}), ("ResNet34", {}),
                                     ("ResNet34", {"two_emb_layer": True}), ("ResNet50", {}),
                                     ("ResNet221", {})])
def test_resnet_restatement_vs_reference_golden(golden_dir, name, kw):
    from oracle import resnet as oresnet
    g = np.load(os.path.join(golden_dir, "resnet_ref.npz"))
    tag = name + ("_2emb" if kw else "")
    sd = synth.synth_resnet_state_dict(name, 80, 256, seed=42, **kw)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
    emb = oresnet.resnet_forward(sd, feats, name).numpy()
    ref = g[tag + "/emb"]
    assert emb.shape == ref.shape == (2, 256)
    assert np.abs(emb - ref).max() <= 2e-4 * np.abs(ref).max()
    emb_s = oresnet.resnet_forward(sd, feats[:, :57], name).numpy()
    assert np.abs(emb_s - g[tag + "/emb_T57"]).max() <= 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize("name", ["ResNet101", "ResNet152", "ResNet293"])
def test_resnet_deep_restatement_vs_reference_golden(golden_dir, name):
    """wespeaker/models/resnet.py:231-260: the constructors the first golden file left out."""
    from oracle import resnet as oresnet
    g = np.load(os.path.join(golden_dir, "resnet_deep_ref.npz"))
    sd = synth.synth_resnet_state_dict(name, 80, 256, seed=42)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
    ref = g[name + "/emb"]
    emb = oresnet.resnet_forward(sd, feats, name).numpy()
    assert emb.shape == ref.shape == (2, 256)
    assert np.abs(emb - ref).max() <= 2e-4 * np.abs(ref).max()
    emb_s = oresnet.resnet_forward(sd, feats[:, :57], name).numpy()
    assert np.abs(emb_s - g[name + "/emb_T57"]).max() <= 2e-4 * np.abs(ref).max()


def test_apply_cmvn_restatement_vs_reference_golden(golden_dir):
    """oracle.fbank.apply_cmvn vs the reference's own function (dataset/dataset_utils.py:19-26) for the four
    (norm_mean, norm_var) settings at 198 and 57 frames; live against the reference when it is here."""
    g = np.load(os.path.join(golden_dir, "cmvn_ref.npz"))
    raw = g["m0v0"]
    for nm in (False, True):
        for nv in (False, True):
            for suffix, x in (("", raw), ("_T57", raw[:, :57])):
                want = g["m%dv%d%s" % (nm, nv, suffix)]
                got = ofbank.apply_cmvn(x, nm, nv)
                assert got.dtype == np.float32 and got.shape == want.shape
                assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), (nm, nv, suffix)
    assert np.isnan(ofbank.apply_cmvn(raw[:, :1], True, True)).all()      # torch.var of one frame is NaN
    if ref_shim.available():
        du = ref_shim.ref_module("wespeaker.dataset.dataset_utils")
        live = du.apply_cmvn(torch.from_numpy(raw), norm_mean=True, norm_var=True).numpy()
        assert np.array_equal(live, g["m1v1"])
        assert bool(torch.isnan(du.apply_cmvn(torch.from_numpy(raw[:, :1].copy()), True, True)).all())


def test_campplus_restatement_vs_reference_golden(golden_dir):
    from oracle import campplus as ocam
    g = np.load(os.path.join(golden_dir, "campplus_ref.npz"))
    sd = synth.synth_campplus_state_dict(80, 512, seed=42)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(2)])
    emb = ocam.campplus_forward(sd, feats).numpy()
    assert emb.shape == (2, 512)
    assert np.abs(emb - g["emb"]).max() <= 2e-4 * np.abs(g["emb"]).max()
    # 328 frames -> T' = 164 -> two context segments (100 + 64, ceil_mode)
    long_feats = np.stack([ofbank.speaker_features(synth.synth_wav(i, 52800)) for i in range(2)])
    emb_l = ocam.campplus_forward(sd, long_feats).numpy()
    assert np.abs(emb_l - g["emb_T328"]).max() <= 2e-4 * np.abs(g["emb_T328"]).max()
    emb_s = ocam.campplus_forward(sd, feats[:, :57]).numpy()
    assert np.abs(emb_s - g["emb_T57"]).max() <= 2e-4 * np.abs(g["emb_T57"]).max()


def test_score_restatement_vs_reference_golden(golden_dir):
    """oracle/score.py vs the outputs of the reference's own bin/score.py + bin/score_norm.py
    (tests/golden/score_ref.npz, oracle/make_golden.py:make_score)."""
    from oracle import score as oscore
    g = np.load(os.path.join(golden_dir, "score_ref.npz"))
    fix = synth.synth_scoring_set()
    mean_vec = fix["cohort_emb"].mean(0)
    names = fix["eval_names"]
    enroll_list = sorted(set(t[0] for t in fix["trials"]))
    test_list = sorted(set(t[1] for t in fix["trials"]))
    idx = {k: i for i, k in enumerate(names)}
    e_emb = fix["eval_emb"][[idx[u] for u in enroll_list]]
    t_emb = fix["eval_emb"][[idx[u] for u in test_list]]
    ie = np.array([enroll_list.index(t[0]) for t in fix["trials"]])
    it = np.array([test_list.index(t[1]) for t in fix["trials"]])
    for tag, mv in (("nomean", None), ("mean", mean_vec)):
        cos = oscore.cosine_pairs(fix["eval_emb"], mv, fix["idx_a"], fix["idx_b"])
        # the golden scores went through the '%.5f' text of the score file
        assert np.abs(cos - g[tag + "/cosine"]).max() <= 5.1e-6 + 1e-6
        for method in ("asnorm", "snorm"):
            cols = oscore.score_norm(method, 20, g[tag + "/cosine"], ie, it, e_emb, t_emb,
                                     fix["cohort_emb"], mv)
            ref = g["%s/%s" % (tag, method)]
            assert np.abs(cols["normed"] - ref[:, 0]).max() <= 5.1e-6 + 1e-4
            for j, key in enumerate(("enroll_mag", "test_mag", "enroll_mean", "test_mean")):
                assert np.abs(cols[key] - ref[:, 1 + j]).max() <= 5.1e-5 + 1e-5, key
    m, s = oscore.get_mean_std(fix["eval_emb"] - mean_vec, fix["cohort_emb"] - mean_vec, 20)
    np.testing.assert_allclose(m, g["get_mean_std/mean"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(s, g["get_mean_std/std"], rtol=0, atol=1e-6)


def test_chunk_rule_known_answers():
    """oracle/chunked.py against hand-worked cases of speaker_engine.cc:96-131 (2 s chunks =
    198 frames, runtime/server/x86_gpu/README.md:35-53)."""
    from oracle import chunked
    cf, ch = chunked.chunk_frames(498, 16000, 32000)          # 5 s
    assert cf == 198 and len(ch) == 3
    assert ch[0][0] == 0 and ch[1][0] == 198 and ch[1][-1] == 395
    assert list(ch[2][:102]) == list(range(396, 498)) and list(ch[2][102:]) == list(range(0, 96))
    cf, ch = chunked.chunk_frames(98, 16000, 32000)           # 1 s: cyclic tiling
    assert len(ch) == 1 and list(ch[0]) == [i % 98 for i in range(198)]
    cf, ch = chunked.chunk_frames(396, 16000, 32000)          # exact multiple: no partial chunk
    assert len(ch) == 2 and ch[1][-1] == 395
    cf, ch = chunked.chunk_frames(498, 16000, 0)              # full mode
    assert cf == 498 and len(ch) == 1 and len(ch[0]) == 498
    cf, ch = chunked.chunk_frames(199, 16000, 32000)          # one full chunk + 1 frame
    assert len(ch) == 2 and list(ch[1]) == [198] + list(range(0, 197))


@pytest.mark.parametrize("tag,sub,nl", [("plain", False, False), ("sub_nl", True, True)])
def test_plda_training_restatement_vs_reference_golden(golden_dir, tag, sub, nl):
    """oracle/plda_train.py vs the reference's own TwoCovPLDA.train(3) / .adapt() outputs
    (tests/golden/plda_train_ref.npz).  Eigenvector signs are arbitrary, so the transforms are
    compared through psi and through LLRs of fixed pairs."""
    from oracle import plda_train as otrain
    g = np.load(os.path.join(golden_dir, "plda_train_ref.npz"))
    fix = synth.synth_plda_training_set()
    mats = {}
    for e, s in zip(fix["emb"], fix["spk"]):
        mats.setdefault(int(s), []).append(e)
    # class order of the reference = first appearance in the scp
    class_mats = [np.vstack(v) for v in mats.values()]
    p = otrain.train(class_mats, 3, subtract_train_set_mean=sub, normalize_length=nl,
                     samples=fix["emb"])
    for k in ("B", "W", "mu", "psi"):
        np.testing.assert_allclose(p[k], g["%s/%s" % (tag, k)], rtol=1e-9, atol=1e-11, err_msg=k)
    probe, _ = synth.synth_embeddings(24, 64, seed=47)

    def llr(params, n):
        tr = np.stack([oplda.transform_embedding(params, e) for e in probe])
        return np.array([[oplda.log_likelihood_ratio(params, tr[i], tr[12 + j], n)
                          for j in range(12)] for i in range(12)])
    np.testing.assert_allclose(llr(p, 2), g[tag + "/llr"], rtol=0, atol=1e-8)
    a = otrain.adapt(p, fix["adapt"])
    np.testing.assert_allclose(np.sort(a["psi"]), np.sort(g[tag + "/adapt_psi"]), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(a["mu"], g[tag + "/adapt_mu"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(llr(a, 1), g[tag + "/adapt_llr"], rtol=0, atol=1e-6)


def _align_signs(a, ref):
    """eigh leaves each LDA eigenvector's sign arbitrary: flip columns of `a` to match `ref`."""
    s = np.sign(np.sum(a * ref, axis=0))
    s[s == 0] = 1.0
    return a * s


def test_embedding_processing_restatement_vs_reference_golden(golden_dir):
    from oracle import embedding_processing as oproc
    g = np.load(os.path.join(golden_dir, "embd_proc_ref.npz"))
    fix = synth.synth_plda_training_set()
    probe, _ = synth.synth_embeddings(24, 64, seed=47)
    out, (mean1, lda_m, lda) = oproc.chain_fit_apply(fix["emb"], fix["spk"], 20, probe)
    np.testing.assert_allclose(mean1, g["mean1"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(lda_m, g["lda_m"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(np.abs(lda).sum(0), g["lda_abs_colsum"], rtol=1e-5)
    assert np.abs(_align_signs(out, g["out"]) - g["out"]).max() <= 1e-5


def test_chunk_rule_vs_reference_engine_golden(golden_dir):
    """oracle/chunked.py against the reference's OWN SpeakerEngine (speaker_engine.cc:63-159 compiled
    into oracle/_ref/libref_engine.so, run by oracle/make_golden.py chunked): chunk counts equal, the
    padded chunk tensor within the native-fbank noise (the engine sits on the reference's float FFT),
    averaged embeddings inside the north_star bar."""
    from oracle import chunked
    g = np.load(os.path.join(golden_dir, "chunked_ref.npz"))
    sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    for i, (seed, n, spc) in enumerate(g["cases"]):
        feats = ofbank.speaker_features(synth.synth_wav(int(seed), int(n)), cmn=False)
        seen = []

        def forward(batch):
            seen.append(batch)
            return oecapa.ecapa_forward(sd, batch).numpy()

        emb, n_chunks = chunked.extract_chunked(feats, 16000, int(spc), forward)
        assert n_chunks == int(g["%d/n_chunks" % i])
        last = seen[0][-1]
        assert last.shape == g["%d/last_chunk" % i].shape
        assert np.abs(last - g["%d/last_chunk" % i]).max() < 1e-3
        ref = g["%d/emb" % i]
        cos = 1.0 - float(np.dot(emb, ref) / (np.linalg.norm(emb) * np.linalg.norm(ref)))
        assert cos < 1e-5 and np.linalg.norm(emb - ref) / np.linalg.norm(ref) < 2e-3, (i, cos)


def test_reference_engine_library_runs_live(golden_dir):
    """When oracle/_ref/libref_engine.so is present (it travels to the GPU box), the reference engine
    itself re-produces the committed golden bit for bit with a cheap stand-in model."""
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                      "libref_engine.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libref_engine.so not built")
    from oracle.make_golden import ref_engine_extract
    g = np.load(os.path.join(golden_dir, "chunked_ref.npz"))
    seed, n, spc = (int(v) for v in g["cases"][0])
    cap = []
    emb, n_chunks = ref_engine_extract(synth.synth_wav(seed, n), spc,
                                       lambda f: f.mean(1)[:, :8], 8, cap)      # stand-in: 8 bin means
    assert n_chunks == int(g["0/n_chunks"]) and np.array_equal(cap[-1], g["0/last_chunk"])
    assert np.allclose(emb, np.mean([c.mean(0)[:8] for c in cap], axis=0), atol=1e-6)
