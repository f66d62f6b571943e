"""ORACLE helper (this container only): import the reference's own Python modules from
/root/reference WITHOUT executing wespeaker/__init__.py (which needs torchaudio, kaldiio,
silero_vad -- all absent).  Used by oracle/make_golden.py to generate tests/golden/*.npz and by
tests that are skipped when /root/reference does not exist (i.e. on the GPU box)."""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("WESPEAKER_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "wespeaker", "models"))


def _seed_packages():
    if "wespeaker" in sys.modules and getattr(sys.modules["wespeaker"], "_oracle_shim", False):
        return
    for name, sub in (("wespeaker", ""), ("wespeaker.models", "models"),
                      ("wespeaker.utils", "utils"), ("wespeaker.utils.plda", "utils/plda"),
                      ("wespeaker.bin", "bin"), ("wespeaker.dataset", "dataset")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF_ROOT, "wespeaker", sub)]
        m._oracle_shim = True
        sys.modules[name] = m
    # third-party modules the PLDA files import at module scope but never use on the numeric path
    for stub in ("h5py", "kaldiio", "kaldi_io", "kaldi_io.kaldi_io", "fire"):
        if stub not in sys.modules:
            try:
                importlib.import_module(stub)
            except Exception:
                s = types.ModuleType(stub)
                if stub == "kaldi_io":
                    s.open_or_fd = s.BadSampleSize = s.UnknownMatrixHeader = None
                if stub == "kaldi_io.kaldi_io":
                    s._read_compressed_mat = s._read_mat_ascii = None
                if stub == "kaldiio":
                    # bin/score.py and bin/score_norm.py read embeddings through
                    # kaldiio.load_scp_sequential: serve the same ark/scp files with our reader
                    def load_scp_sequential(scp_path):
                        from wespeaker_amd.kaldi_io import read_vec_scp
                        return iter(read_vec_scp(scp_path).items())
                    s.load_scp_sequential = load_scp_sequential
                if stub == "fire":
                    s.Fire = lambda *a, **k: None
                sys.modules[stub] = s


def ref_module(dotted: str):
    """e.g. ref_module('wespeaker.models.ecapa_tdnn')"""
    if not available():
        raise RuntimeError("reference not available at " + REF_ROOT)
    _seed_packages()
    return importlib.import_module(dotted)


def ref_model(model_name: str, **model_args):
    fam = {"ECAPA": "ecapa_tdnn", "ResNe": "resnet", "CAMPP": "campplus"}[model_name[:5]]
    mod = ref_module("wespeaker.models." + fam)
    return getattr(mod, model_name)(**model_args)


def ref_plda(params: dict):
    mod = ref_module("wespeaker.utils.plda.two_cov_plda")
    p = mod.TwoCovPLDA()
    p.mu, p.transform, p.psi, p.offset = (params["mu"], params["transform"], params["psi"],
                                          params["offset"])
    p.dim = p.mu.shape[0]
    p.normalize_length = params["normalize_length"]
    return p
