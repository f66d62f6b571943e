#!/usr/bin/env python
"""ORACLE tooling (test infrastructure): generate tests/golden/*.npz by running the
REFERENCE ITSELF in this container.

  python oracle/make_golden.py          # needs /root/reference; writes tests/golden/

What is recorded (all seeds live in fixtures/synth.py, inputs are regenerated from them):
  * ecapa_ref.npz   -- embeddings of the reference's own nn.Modules
                       (wespeaker/models/ecapa_tdnn.py, imported from /root/reference) for the four
                       ECAPA constructors on synthetic utterances 0..1, weights = synth seed 42.
  * resnet_ref.npz / campplus_ref.npz -- the same for the reference's ResNet18/34/50/221 and
                       CAMPPlus modules (wespeaker/models/resnet.py, campplus.py).
  * resnet_deep_ref.npz -- ResNet101 / ResNet152 / ResNet293 (resnet.py:231-260), same inputs, 198 and 57 frames.
  * cmvn_ref.npz    -- the reference's own apply_cmvn (dataset/dataset_utils.py:19-26) for the four
                       (norm_mean, norm_var) settings + ECAPA-512 embeddings of the (1,1) and (0,0) features.
  * plda_ref.npz    -- outputs of the reference's own TwoCovPLDA.transform_embedding /
                       log_likelihood_ratio (wespeaker/utils/plda/two_cov_plda.py:156-184).
  * score_ref.npz   -- outputs of the reference's own bin/score.py (trials_cosine_score) and
                       bin/score_norm.py (get_mean_std, main with asnorm and snorm) run on the
                       synth_scoring_set fixture through real ark/scp/trial files.
  * plda_train_ref.npz -- the reference's own TwoCovPLDA(scp_file, utt2spk_file, ...).train(3) and
                       .adapt(adapt_scp) run on ark/scp files of the synth_plda_training_set
                       fixture: B, W, mu, psi and LLRs of fixed pairs under the trained / adapted model.
  * embd_proc_ref.npz -- the reference's own EmbeddingProcessingChain ("mean-subtract | length-norm |
                       lda --dim 20 | length-norm") built from ark/scp files of the PLDA training
                       fixture and applied to 24 probe embeddings.
  * fbank_ref_native.npz -- log-mel output of the reference's own native fbank
                       (runtime/core/frontend/fbank.h, built into oracle/_ref by oracle/Makefile).
  * fbank_ref_native_rates.npz -- the same native fbank at 8 kHz (80 / 23 / 40 bins), 32 kHz and 48 kHz
                       (256-, 1024- and 2048-point transforms: fbank.h:33-52 sizes the FFT from the frame).
  * subsegment_ref.npz -- the reference's own diar/extract_emb.py subsegment() on row-index features: which source
                       frame every (window, frame) of the diarization sub-segments holds, and the sub-segment names.
  * kaldi_plda_*.bin / .txt + kaldi_plda_ref.npz -- Kaldi <Plda> files and what the reference's own
                       read_plda returns for them.
  * chunked_ref.npz -- the reference's own native SpeakerEngine (speaker_engine.cc, built into
                       oracle/_ref/libref_engine.so) on synthetic utterances: chunk counts, averaged
                       embeddings, the padded chunk tensors.
The GPU box has no /root/reference; tests there compare against these committed files.
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from oracle.fbank import kaldi_fbank, speaker_features  # noqa: E402
from fixtures import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def ref_native_fbank(pcm_int16, sample_rate=16000, bins=80):
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_fbank.so"))
    lib.ref_fbank.restype = ctypes.c_int
    x = np.ascontiguousarray(pcm_int16, dtype=np.float32)
    out = np.zeros((2000, bins), np.float32)
    n = lib.ref_fbank(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(x.shape[0]), bins, sample_rate,
                      out.ctypes.data_as(ctypes.c_void_p), 2000)
    return out[:n].copy()


# (sample rate, mel bins, samples): 8 kHz = the reference's SRE recipe (examples/sre/v2/conf/resnet.yaml:31), 200-sample
# frames -> 256-point transform; 32 kHz -> 1024; 48 kHz -> 2048.  The native reference sizes its frame as
# sample_rate / 1000 * 25 (runtime/core/frontend/feature_pipeline.h), which equals torchaudio's int(rate * 0.025) at
# these three rates (not at 44.1 kHz: 1100 vs 1102 samples -- that rate is checked against the restatement only).
FBANK_RATES = ((8000, 80, 16000), (8000, 23, 16000), (8000, 40, 2000), (32000, 80, 64000), (48000, 80, 96000))


def make_fbank_rates():
    out = {}
    for rate, bins, n in FBANK_RATES:
        wav = synth.synth_wav(21 + rate // 8000, n)
        ref = ref_native_fbank(wav, rate, bins)
        mine = kaldi_fbank(wav.astype(np.float32), num_mel_bins=bins, sample_frequency=rate, window_type="hamming")
        assert ref.shape == mine.shape, (rate, ref.shape, mine.shape)
        out["r%d_b%d_n%d" % (rate, bins, n)] = ref
        print("fbank %5d Hz %3d bins %6d samples: %d frames, native-ref vs restatement max|d| = %.3e mean|d| = %.3e"
              % (rate, bins, n, ref.shape[0], np.abs(mine - ref).max(), np.abs(mine - ref).mean()))
    np.savez_compressed(os.path.join(GOLD, "fbank_ref_native_rates.npz"), **out)


def make_fbank():
    idx = [0, 7]
    feats = np.stack([ref_native_fbank(synth.synth_wav(i)) for i in idx])
    short = ref_native_fbank(synth.synth_wav(3, 4000))     # 23 frames
    np.savez_compressed(os.path.join(GOLD, "fbank_ref_native.npz"), utt_idx=np.array(idx),
                        logmel=feats, short_idx=np.array(3), short_len=np.array(4000),
                        short_logmel=short)
    mine = np.stack([speaker_features(synth.synth_wav(i), cmn=False) for i in idx])
    print("fbank: native-ref vs restatement max|d| = %.3e mean|d| = %.3e"
          % (np.abs(mine - feats).max(), np.abs(mine - feats).mean()))


def make_ecapa():
    out = {}
    feats = np.stack([speaker_features(synth.synth_wav(i)) for i in range(2)])
    feats_short = feats[:, :57, :].copy()
    for name in ("ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_c512", "ECAPA_TDNN_GLOB_c1024",
                 "ECAPA_TDNN_c1024"):
        sd = synth.synth_ecapa_state_dict(name, 80, 192, seed=42)
        m = ref_shim.ref_model(name, feat_dim=80, embed_dim=192, pooling_func="ASTP")
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m.eval()
        with torch.no_grad():
            out4, emb = m(torch.from_numpy(feats))
            _, emb_s = m(torch.from_numpy(feats_short))
        out[name + "/emb"] = emb.numpy()
        out[name + "/emb_T57"] = emb_s.numpy()
        out[name + "/out4_absmean"] = np.array(out4.abs().mean().item())
        print(name, "emb", emb.shape, float(emb.abs().mean()))
    # emb_bn=True variant (bn2 path, ecapa_tdnn.py:202-206,232-233)
    sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_c512", 80, 256, emb_bn=True, seed=5)
    m = ref_shim.ref_model("ECAPA_TDNN_c512", feat_dim=80, embed_dim=256, pooling_func="ASTP",
                           emb_bn=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m.eval()
    with torch.no_grad():
        out["ECAPA_TDNN_c512_embbn/emb"] = m(torch.from_numpy(feats))[-1].numpy()
    np.savez_compressed(os.path.join(GOLD, "ecapa_ref.npz"), **out)


def _ref_forward(name, sd, feats, **model_args):
    m = ref_shim.ref_model(name, **model_args)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m.eval()
    with torch.no_grad():
        out = m(torch.from_numpy(feats))
    return (out[-1] if isinstance(out, tuple) else out).numpy()


def make_resnet_campplus():
    feats = np.stack([speaker_features(synth.synth_wav(i)) for i in range(2)])
    long_feats = np.stack([speaker_features(synth.synth_wav(i, 52800)) for i in range(2)])   # 328 frames
    out = {}
    for name, kw in (("ResNet18", {}), ("ResNet34", {}), ("ResNet34", {"two_emb_layer": True}),
                     ("ResNet50", {}), ("ResNet221", {})):
        sd = synth.synth_resnet_state_dict(name, 80, 256, seed=42, **kw)
        tag = name + ("_2emb" if kw else "")
        out[tag + "/emb"] = _ref_forward(name, sd, feats, feat_dim=80, embed_dim=256,
                                         pooling_func="TSTP", **kw)
        out[tag + "/emb_T57"] = _ref_forward(name, sd, feats[:, :57].copy(), feat_dim=80,
                                             embed_dim=256, pooling_func="TSTP", **kw)
        print(tag, float(np.abs(out[tag + "/emb"]).mean()))
    np.savez_compressed(os.path.join(GOLD, "resnet_ref.npz"), **out)
    sd = synth.synth_campplus_state_dict(80, 512, seed=42)
    cam = {"emb": _ref_forward("CAMPPlus", sd, feats, feat_dim=80, embed_dim=512, pooling_func="TSTP"),
           "emb_T328": _ref_forward("CAMPPlus", sd, long_feats, feat_dim=80, embed_dim=512,
                                    pooling_func="TSTP"),
           "emb_T57": _ref_forward("CAMPPlus", sd, feats[:, :57].copy(), feat_dim=80, embed_dim=512,
                                   pooling_func="TSTP")}
    np.savez_compressed(os.path.join(GOLD, "campplus_ref.npz"), **cam)
    print("CAMPPlus", float(np.abs(cam["emb"]).mean()))


def make_resnet_deep():
    """The three remaining constructors of wespeaker/models/resnet.py:231-260 (ResNet101 / 152 / 293: Bottleneck
    [3,4,23,3] / [3,8,36,3] / [10,20,64,3]) through the reference's own nn.Module: one 2-s and one T = 57 case each."""
    feats = np.stack([speaker_features(synth.synth_wav(i)) for i in range(2)])
    out = {}
    for name in ("ResNet101", "ResNet152", "ResNet293"):
        sd = synth.synth_resnet_state_dict(name, 80, 256, seed=42)
        out[name + "/emb"] = _ref_forward(name, sd, feats, feat_dim=80, embed_dim=256, pooling_func="TSTP")
        out[name + "/emb_T57"] = _ref_forward(name, sd, feats[:, :57].copy(), feat_dim=80, embed_dim=256,
                                              pooling_func="TSTP")
        print(name, float(np.abs(out[name + "/emb"]).mean()))
    np.savez_compressed(os.path.join(GOLD, "resnet_deep_ref.npz"), **out)


def make_cmvn():
    """The reference's own apply_cmvn (wespeaker/dataset/dataset_utils.py:19-26) on un-normalised fbank features of
    synthetic utterances 0..2 (198 frames) and a 57-frame cut, for the four (norm_mean, norm_var) settings, plus the
    ECAPA-512 embedding of the (True, True) features through the reference module (the path bin/extract.py:124-135
    takes with `cmvn_args: {norm_var: True}`)."""
    du = ref_shim.ref_module("wespeaker.dataset.dataset_utils")
    raw = np.stack([speaker_features(synth.synth_wav(i), cmn=False) for i in range(3)])
    out = {}
    for nm in (False, True):
        for nv in (False, True):
            tag = "m%dv%d" % (nm, nv)
            out[tag] = du.apply_cmvn(torch.from_numpy(raw), norm_mean=nm, norm_var=nv).numpy()
            out[tag + "_T57"] = du.apply_cmvn(torch.from_numpy(raw[:, :57].copy()), norm_mean=nm, norm_var=nv).numpy()
    sd = synth.synth_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
    out["ecapa512_m1v1/emb"] = _ref_forward("ECAPA_TDNN_GLOB_c512", sd, out["m1v1"], feat_dim=80, embed_dim=192,
                                            pooling_func="ASTP")
    out["ecapa512_m0v0/emb"] = _ref_forward("ECAPA_TDNN_GLOB_c512", sd, out["m0v0"], feat_dim=80, embed_dim=192,
                                            pooling_func="ASTP")
    np.savez_compressed(os.path.join(GOLD, "cmvn_ref.npz"), **out)
    print("cmvn", {k: float(np.abs(v).mean()) for k, v in out.items() if not k.endswith("T57")})


def make_plda():
    out = {}
    emb, _ = synth.synth_embeddings(40, 192, seed=11)
    for nl in (False, True):
        p = synth.synth_plda(192, seed=7, normalize_length=nl)
        ref = ref_shim.ref_plda(p)
        tr = np.stack([ref.transform_embedding(e.astype(np.float64)) for e in emb])
        tag = "nl%d" % int(nl)
        out[tag + "/transformed"] = tr
        for n in (1, 3):
            llr = np.array([[ref.log_likelihood_ratio(tr[i], tr[20 + j], n) for j in range(20)]
                            for i in range(20)])
            out["%s/llr_n%d" % (tag, n)] = llr
    np.savez_compressed(os.path.join(GOLD, "plda_ref.npz"), **out)
    print("plda llr range", float(llr.min()), float(llr.max()))


def make_plda_train():
    import tempfile
    mod = ref_shim.ref_module("wespeaker.utils.plda.two_cov_plda")
    utils = ref_shim.ref_module("wespeaker.utils.plda.plda_utils")
    fix = synth.synth_plda_training_set()
    probe, _ = synth.synth_embeddings(24, 64, seed=47)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        paths = synth.write_plda_training_files(fix, d)
        for tag, sub, nl in (("plain", False, False), ("sub_nl", True, True)):
            plda = mod.TwoCovPLDA(scp_file=paths["scp"], utt2spk_file=paths["utt2spk"], embed_dim=64,
                                  subtract_train_set_mean=sub, normalize_length=nl)
            plda.train(3)
            for k in ("B", "W", "mu", "psi", "transform", "offset"):
                out["%s/%s" % (tag, k)] = np.array(getattr(plda, k))
            out[tag + "/offset_scatter"] = plda.stats.offset_scatter
            tr = np.stack([plda.transform_embedding(e.astype(np.float64)) for e in probe])
            out[tag + "/llr"] = np.array([[plda.log_likelihood_ratio(tr[i], tr[12 + j], 2)
                                           for j in range(12)] for i in range(12)])
            adp = plda.adapt(paths["adapt_scp"], 0.5, 0.5)
            adp.dim = adp.mu.shape[0]
            out[tag + "/adapt_mu"], out[tag + "/adapt_psi"] = adp.mu, adp.psi
            tr = np.stack([adp.transform_embedding(e.astype(np.float64)) for e in probe])
            out[tag + "/adapt_llr"] = np.array([[adp.log_likelihood_ratio(tr[i], tr[12 + j], 1)
                                                 for j in range(12)] for i in range(12)])
    np.savez_compressed(os.path.join(GOLD, "plda_train_ref.npz"), **out)
    print("plda train: psi[:4]", out["plain/psi"][:4], "adapt psi[:4]", out["plain/adapt_psi"][:4])


def make_embd_proc():
    import tempfile
    mod = ref_shim.ref_module("wespeaker.utils.embedding_processing")
    fix = synth.synth_plda_training_set()
    probe, _ = synth.synth_embeddings(24, 64, seed=47)
    with tempfile.TemporaryDirectory() as d:
        paths = synth.write_plda_training_files(fix, d)
        chain = ("mean-subtract --scp %s | length-norm | lda --scp %s --utt2spk %s --dim 20 | length-norm"
                 % (paths["scp"], paths["scp"], paths["utt2spk"]))
        c = mod.EmbeddingProcessingChain(chain=chain)
        out = c(probe)
        np.savez_compressed(os.path.join(GOLD, "embd_proc_ref.npz"), out=out,
                            mean1=c.chain_of_classes[0].mean, lda_m=c.chain_of_classes[2].m,
                            lda_abs_colsum=np.abs(c.chain_of_classes[2].lda).sum(0))
    print("embd proc: out", out.shape, out.dtype)


def make_score():
    import tempfile
    score_mod = ref_shim.ref_module("wespeaker.bin.score")
    norm_mod = ref_shim.ref_module("wespeaker.bin.score_norm")
    fix = synth.synth_scoring_set()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        paths = synth.write_scoring_files(fix, d)
        for tag, mean_path in (("nomean", None), ("mean", paths["mean_vec"])):
            store = os.path.join(d, "scores_" + tag)
            os.makedirs(store)
            score_mod.trials_cosine_score(paths["eval_scp"], store, mean_path, [paths["trials"]])
            score_file = os.path.join(store, "trials.kaldi.score")
            rows = [l.split() for l in open(score_file)]
            out[tag + "/cosine"] = np.array([float(r[2]) for r in rows])
            for method, top_n in (("asnorm", 20), ("snorm", 20)):
                nf = os.path.join(store, method + ".score")
                norm_mod.main(method, top_n, score_file, nf, paths["cohort_scp"], paths["eval_scp"],
                              mean_path)
                cols = np.array([[float(x) for x in (l.split()[2:3] + l.split()[4:8])]
                                 for l in open(nf)])
                out["%s/%s" % (tag, method)] = cols      # normed, e_mag, t_mag, e_mean, t_mean
        mv = np.load(paths["mean_vec"])
        m, s = norm_mod.get_mean_std(fix["eval_emb"] - mv, fix["cohort_emb"] - mv, 20)
        out["get_mean_std/mean"], out["get_mean_std/std"] = m, s
    np.savez_compressed(os.path.join(GOLD, "score_ref.npz"), **out)
    print("score: cosine range", out["mean/cosine"].min(), out["mean/cosine"].max(),
          "asnorm range", out["mean/asnorm"][:, 0].min(), out["mean/asnorm"][:, 0].max())


def make_kaldi_plda():
    """Kaldi <Plda> files (float and double binary, text) written by fixtures.synth.write_kaldi_plda and
    read back by the REFERENCE's own read_plda (utils/plda/kaldi_utils.py:24-109; only kaldi_io's
    trivial open_or_fd is stubbed -- the binary parsing is the reference's code)."""
    ku = ref_shim.ref_module("wespeaker.utils.plda.kaldi_utils")
    ku.open_or_fd = lambda f: open(f, "rb")
    p = synth.synth_plda(6, seed=13)
    out = {}
    for tag, kw in (("f32", dict(double=False)), ("f64", dict(double=True))):
        path = os.path.join(GOLD, "kaldi_plda_%s.bin" % tag)
        synth.write_kaldi_plda(path, p["mu"], p["transform"], p["psi"], binary=True, **kw)
        mu, tr, psi = ku.read_plda(path)
        out[tag + "/mu"], out[tag + "/transform"], out[tag + "/psi"] = mu, tr, psi
    synth.write_kaldi_plda(os.path.join(GOLD, "kaldi_plda.txt"), p["mu"], p["transform"], p["psi"],
                           binary=False)
    np.savez_compressed(os.path.join(GOLD, "kaldi_plda_ref.npz"), **out)
    print("kaldi plda: dim", out["f64/mu"].shape[0], "f32 vs f64 max diff",
          np.abs(out["f32/transform"] - out["f64/transform"]).max())


CHUNK_CASES = [  # (utt seed, num_samples, samples_per_chunk)
    (200, 48000, 32000),     # 298 frames: one full 198-frame chunk + 100 frames completed with head frames
    (201, 20000, 32000),     # 123 frames < one chunk: cyclic tiling
    (202, 63600, 32000),     # 396 frames: exactly two chunks, no partial one
    (203, 40000, 0),         # full mode
    (204, 70000, 16000),     # 98-frame chunks: 4 full + a partial one
    (205, 32160, 32000),     # 199 frames: one full chunk + ONE frame
]


def ref_engine_extract(pcm, samples_per_chunk, forward, emb_dim, capture=None):
    """The reference's SpeakerEngine::ExtractEmbedding (oracle/_ref/libref_engine.so) with `forward`
    ((1, T, 80) float32 -> (1, E)) standing in for the ONNX model.  capture: list that receives the
    (T, 80) chunk tensors exactly as the engine hands them to the model (after its ApplyMean)."""
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_engine.so"))
    CB = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int,
                          ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_void_p)
    calls = []

    def cb(feats, T, D, emb, E, _user):
        f = np.ctypeslib.as_array(feats, shape=(T, D)).copy()
        calls.append(f)
        out = np.asarray(forward(f[None]), dtype=np.float32).reshape(-1)
        ctypes.memmove(emb, out.ctypes.data, 4 * E)

    lib.ref_engine_extract.restype = ctypes.c_int
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    avg = np.zeros(emb_dim, np.float32)
    n = lib.ref_engine_extract(pcm.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(pcm.shape[0]), 80, 16000,
                               emb_dim, ctypes.c_int(samples_per_chunk), CB(cb), None,
                               avg.ctypes.data_as(ctypes.c_void_p))
    # ExtractFeature ran once for the count and once inside ExtractEmbedding: the model saw n chunks
    assert len(calls) == n, (len(calls), n)
    if capture is not None:
        capture.extend(calls)
    return avg, n


def make_chunked():
    """chunked_ref.npz -- the reference's own native SpeakerEngine (runtime/core/speaker/
    speaker_engine.cc:63-159 + frontend/feature_pipeline.cc + fbank.h, compiled into
    oracle/_ref/libref_engine.so) run on synthetic utterances, with the pinned ECAPA oracle as the
    model behind its SpeakerModel interface: chunk count, averaged embedding, and for the two
    padding cases the chunk tensors the engine hands to the model."""
    from oracle import ecapa as oecapa
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in
          synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42).items()}
    forward = lambda f: oecapa.ecapa_forward(sd, f).numpy()      # noqa: E731
    out = {"cases": np.array(CHUNK_CASES)}
    for i, (seed, n, spc) in enumerate(CHUNK_CASES):
        cap = []
        emb, n_chunks = ref_engine_extract(synth.synth_wav(seed, n), spc, forward, 192, cap)
        out["%d/emb" % i] = emb
        out["%d/n_chunks" % i] = np.array(n_chunks)
        out["%d/last_chunk" % i] = cap[-1]            # the padded / completed chunk (or the only one)
        print("chunked case", (seed, n, spc), "->", n_chunks, "chunks of", cap[0].shape[0], "frames")
    np.savez_compressed(os.path.join(GOLD, "chunked_ref.npz"), **out)


# (num_frames, seg_length, window, period): seg_length = num_frames + 2 is what the reference's own VAD segments give
# (diar/extract_emb.py:62-65); the others cover the single-window branch, a last window that runs past the frames, a
# layout from exactly the frame count, and a non-default window / period pair
SUBSEG_CASES = ((498, 500, 150, 75), (148, 150, 150, 75), (98, 100, 150, 75), (149, 151, 150, 75), (223, 225, 150, 75),
                (300, 300, 150, 75), (331, 333, 100, 30), (1498, 1500, 150, 75))


def make_subsegment():
    """The reference's own subsegment() (wespeaker/diar/extract_emb.py:55-83) on feature matrices whose row r holds the
    value r: the windows it returns ARE the (window, frame) -> source-row maps.  onnxruntime (imported at module scope,
    never used by subsegment) is stubbed."""
    import types
    for stub in ("onnxruntime",):
        if stub not in sys.modules:
            sys.modules[stub] = types.ModuleType(stub)
    ref_shim.ref_module("wespeaker.models.ecapa_tdnn")          # seeds the package shims
    diar = types.ModuleType("wespeaker.diar")
    diar.__path__ = [os.path.join(ref_shim.REF_ROOT, "wespeaker", "diar")]
    sys.modules["wespeaker.diar"] = diar
    mod = ref_shim.ref_module("wespeaker.diar.extract_emb")
    out = {}
    for k, (nf, seg_len, win, per) in enumerate(SUBSEG_CASES):
        fb = np.repeat(np.arange(nf, dtype=np.float32)[:, None], 4, axis=1)
        seg_id = "{:08d}-{:08d}".format(1230, 1230 + seg_len * 10)
        names, wins = mod.subsegment(fbank=fb, seg_id=seg_id, window_fs=win, period_fs=per, frame_shift=10)
        idx = np.stack(wins)[:, :, 0].astype(np.int32)
        assert idx.shape == (len(names), win) and (np.stack(wins) == idx[:, :, None]).all()
        out["case%d_params" % k] = np.array([nf, seg_len, win, per], np.int32)
        out["case%d_rows" % k] = idx
        out["case%d_names" % k] = np.array(names)
        print("subsegment", (nf, seg_len, win, per), "->", len(names), "windows, last", names[-1])
    np.savez_compressed(os.path.join(GOLD, "subsegment_ref.npz"), **out)


SECTIONS = {"fbank": make_fbank, "fbank_rates": make_fbank_rates, "subsegment": make_subsegment, "ecapa": make_ecapa, "resnet_campplus": make_resnet_campplus,
            "resnet_deep": make_resnet_deep, "cmvn": make_cmvn,
            "plda": make_plda, "score": make_score, "plda_train": make_plda_train,
            "embd_proc": make_embd_proc, "kaldi_plda": make_kaldi_plda, "chunked": make_chunked}

if __name__ == "__main__":
    assert ref_shim.available(), "needs /root/reference"
    os.makedirs(GOLD, exist_ok=True)
    for name in (sys.argv[1:] or list(SECTIONS)):       # python oracle/make_golden.py [section ...]
        SECTIONS[name]()
    print("golden fixtures written to", GOLD)
