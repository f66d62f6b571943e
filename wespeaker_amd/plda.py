"""Two-covariance PLDA scoring on the MI355X behind the reference's interface.

Mirrors `wespeaker/utils/plda/two_cov_plda.py`:
    TwoCovPLDA.load_model(path, from_kaldi=False)          :341-363
    .transform_embedding(x) / .log_likelihood_ratio(e, t, n)   :156-184
    .eval_sv(enroll_scp, enroll_utt2spk, test_scp, trials, score_file,
             multisession_avg=True, indomain_scp=None)     :186-256
plus the matrix / pair APIs and the new in-memory entry point `score_plda(...)` (semantics of
`eval_sv`; the reference only has it as bin/eval_plda.py + local/score_plda.sh).
All arithmetic runs in float64 in the HIP library (the reference computes in numpy float64);
numpy/torch are used here only for plumbing (name -> row index maps, device buffers).
EM training / adaptation (two_cov_plda.py:38-154,258-309): `TwoCovPLDA(scp_file=..., utt2spk_file=...)`,
`.train(iters)`, `.adapt(scp)` -- the utterance-level statistics run on the GPU (ws_plda_stats), the
D x D algebra in numpy float64 (wespeaker_amd/plda_train.py).
"""
import ctypes
import os
import struct
from collections import OrderedDict
from ctypes import c_void_p

import numpy as np
import torch

from . import _lib
from .engine import default_device
from .kaldi_io import read_vec_scp


def _f64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


class TwoCovPLDA:

    def __init__(self, scp_file=None, utt2spk_file=None, embed_dim=256, subtract_train_set_mean=False,
                 normalize_length=False, *, mu=None, transform=None, psi=None, offset=None, device=None):
        """Positional parameters are the reference's (two_cov_plda.py:68-73); the in-memory parameters
        of a trained model are keyword-only (see `from_params`)."""
        self.normalize_length = bool(normalize_length)
        self.subtract_train_set_mean = bool(subtract_train_set_mean)
        self.dim = int(embed_dim if mu is None else np.asarray(mu).shape[0])
        self.mu = _f64(np.zeros(self.dim) if mu is None else mu)
        self.transform = _f64(np.zeros((self.dim, self.dim)) if transform is None else transform)
        self.psi = _f64(np.zeros(self.dim) if psi is None else psi)
        self.offset = _f64(-self.transform @ self.mu if offset is None else offset)
        self._device = device
        self._h = None
        self.stats = None
        self.B, self.W = np.eye(self.dim), np.eye(self.dim)
        if scp_file is not None:                      # training constructor, two_cov_plda.py:95-107
            from . import plda_train
            samples, embeddings_dict = plda_train.get_data_for_plda(scp_file, utt2spk_file)
            train_mean = samples.mean(0) if self.subtract_train_set_mean else None
            self.stats = plda_train.collect_stats(embeddings_dict, train_mean, self.normalize_length,
                                                  device)
            self.dim = self.stats.dim
            self.B, self.W = np.eye(self.dim), np.eye(self.dim)
            self.mu = self.stats.sum_ / self.stats.class_weight

    @classmethod
    def from_params(cls, mu, transform, psi, offset=None, normalize_length=False,
                    subtract_train_set_mean=False, device=None):
        """A trained model from its parameters (what load_model builds attribute by attribute in the
        reference, two_cov_plda.py:341-363)."""
        return cls(normalize_length=normalize_length, subtract_train_set_mean=subtract_train_set_mean,
                   mu=mu, transform=transform, psi=psi, offset=offset, device=device)

    # ---------------------------------------------------------------- training (host f64 on GPU stats)
    def em_one_iter(self):
        from . import plda_train
        self.B, self.W = plda_train.em_one_iter(self.stats, self.B, self.W)

    def get_output(self):
        from . import plda_train
        self.mu, self.transform, self.psi, self.offset = plda_train.get_output(self.stats, self.B, self.W)
        self._invalidate()

    def train(self, num_em_iters):
        for i in range(num_em_iters):
            print("Plda estimation %d of %d" % (i, num_em_iters))
            self.em_one_iter()
        self.get_output()

    def adapt(self, adapt_scp, ac_scale=0.5, wc_scale=0.5):
        """two_cov_plda.py:258-309.  Like the reference, the returned model does NOT inherit
        normalize_length / subtract_train_set_mean (it is built by a bare TwoCovPLDA())."""
        from . import plda_train
        rows = np.array(list(read_vec_scp(adapt_scp).values()))
        mu, tr, psi, off = plda_train.adapt_parameters(self.mu, self.transform, self.psi, rows,
                                                       self.normalize_length, ac_scale, wc_scale,
                                                       self._device)
        return TwoCovPLDA.from_params(mu, tr, psi, off, device=self._device)

    # ------------------------------------------------------------------------------ native handle
    @property
    def device(self):
        if self._device is None:
            self._device = default_device()
        return torch.device(self._device)

    def _handle(self):
        if self._h is None:
            _lib.require_gpu()
            h = c_void_p()
            _lib.check(_lib.lib().ws_plda_create(
                self.dim, _lib.ptr(self.mu), _lib.ptr(self.transform), _lib.ptr(self.psi),
                _lib.ptr(self.offset), int(self.normalize_length), self.device.index or 0,
                ctypes.byref(h)), "ws_plda_create")
            self._h = h
        return self._h

    def _invalidate(self):
        if self._h and _lib is not None and getattr(_lib, "_lib", None) is not None:
            _lib._lib.ws_plda_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self._invalidate()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------- I/O
    @staticmethod
    def load_model(model_name, from_kaldi=False, device=None):
        if from_kaldi:
            mu, tr, psi = read_kaldi_plda(model_name)
            return TwoCovPLDA.from_params(mu, tr, psi, -1.0 * np.matmul(tr, mu), device=device)
        # the format is read off the file's magic, not its name: save_model keeps the caller's exact
        # path (reference recipes pass suffix-less names such as ${exp_dir}/plda)
        path = str(model_name)
        if not os.path.exists(path) and os.path.exists(path + ".npz"):
            path = path + ".npz"
        with open(path, "rb") as fh:
            magic = fh.read(8)
        if magic[:2] == b"PK":                       # zip container = numpy .npz
            with np.load(path) as f:
                return TwoCovPLDA.from_params(f["mu"], f["transform"], f["psi"], f["offset"],
                                              bool(f["normalize_length"]),
                                              bool(f["subtract_train_set_mean"]), device=device)
        if magic != b"\x89HDF\r\n\x1a\n":
            raise ValueError("%s is neither a .npz nor an HDF5 PLDA model" % path)
        # the reference's own format (two_cov_plda.py:348-355 reads it with h5py): here through the HDF5 C library
        # (wespeaker_amd/hdf5_io.py, ctypes), which this image has while h5py is absent
        from . import hdf5_io
        try:
            f = hdf5_io.read_datasets(path, ("mu", "transform", "psi", "offset", "normalize_length",
                                             "subtract_train_set_mean"))
        except hdf5_io.Hdf5Error as e:
            raise ImportError("reading the reference's HDF5 PLDA model %s needs libhdf5 (WS_HDF5_LIB) : %s; "
                              "use .npz or from_kaldi=True" % (path, e)) from e
        return TwoCovPLDA.from_params(f["mu"], f["transform"], f["psi"], f["offset"], bool(f["normalize_length"]),
                                      bool(f["subtract_train_set_mean"]), device=device)

    def save_model(self, output_file_name, fmt=None):
        """Writes exactly `output_file_name` (np.savez on a path would append '.npz').  fmt "hdf5" = the
        reference's format (two_cov_plda.py:311-339: six datasets, arrays chunked + gzip + fletcher32 with
        unlimited maxshape, the two flags as integer scalars), written through libhdf5 (hdf5_io.py); "npz" =
        numpy's container.  Default: hdf5 where the HDF5 library is found, npz otherwise; load_model tells them
        apart by the file magic."""
        from . import hdf5_io
        if fmt is None:
            fmt = "hdf5" if hdf5_io.available() else "npz"
        items = dict(mu=self.mu, transform=self.transform, psi=self.psi, offset=self.offset,
                     normalize_length=int(self.normalize_length),
                     subtract_train_set_mean=int(self.subtract_train_set_mean))
        if fmt == "hdf5":
            hdf5_io.write_datasets(output_file_name, items)
        elif fmt == "npz":
            with open(output_file_name, "wb") as fh:
                np.savez(fh, **items)
        else:
            raise ValueError("save_model: fmt must be 'hdf5' or 'npz'")

    # ------------------------------------------------------------------------- device primitives
    def _dev(self, x, dtype):
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def prepare_test(self, emb, mean_vec=None) -> torch.Tensor:
        """(N, D) float32 embeddings -> (N, D) float64 transformed (eval_sv :237-244)."""
        emb = self._dev(emb, torch.float32)
        n = emb.shape[0]
        out = torch.empty((n, self.dim), dtype=torch.float64, device=self.device)
        mv = _f64(mean_vec) if mean_vec is not None else None
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ws_plda_prepare_test(
                self._handle(), _lib.ptr(emb), n, _lib.ptr(mv) if mv is not None else None,
                _lib.ptr(out), _lib.current_stream_ptr(self.device)), "ws_plda_prepare_test")
        return out

    def prepare_enroll(self, emb, group_offsets, mean_vec=None) -> torch.Tensor:
        """Rows of emb grouped contiguously per enrollment model (eval_sv :216-235)."""
        emb = self._dev(emb, torch.float32)
        offs = self._dev(np.asarray(group_offsets, dtype=np.int32), torch.int32)
        g = offs.shape[0] - 1
        out = torch.empty((g, self.dim), dtype=torch.float64, device=self.device)
        mv = _f64(mean_vec) if mean_vec is not None else None
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ws_plda_prepare_enroll(
                self._handle(), _lib.ptr(emb), _lib.ptr(offs), g,
                _lib.ptr(mv) if mv is not None else None, _lib.ptr(out),
                _lib.current_stream_ptr(self.device)), "ws_plda_prepare_enroll")
        return out

    def _sessions(self, n_sessions, n_rows):
        """-> (device int32 array or None, n_uniform).  A python int / a host array whose entries
        are all equal selects the uniform-n fast path (the common multisession_avg=True case)."""
        if isinstance(n_sessions, torch.Tensor):
            return self._dev(n_sessions, torch.int32), 0
        arr = np.asarray(n_sessions, dtype=np.int32).reshape(-1)
        if arr.size == 1 or (arr.size and np.all(arr == arr[0])):
            return None, int(arr[0])
        return self._dev(np.broadcast_to(arr, (n_rows,)).copy(), torch.int32), 0

    def llr_matrix(self, enroll_t, n_sessions, test_t) -> torch.Tensor:
        """(Ne, D), (Ne,) or int, (Nt, D) transformed float64 -> (Ne, Nt) float64 LLRs on the GPU."""
        e = self._dev(enroll_t, torch.float64)
        t = self._dev(test_t, torch.float64)
        n, nu = self._sessions(n_sessions, e.shape[0])
        out = torch.empty((e.shape[0], t.shape[0]), dtype=torch.float64, device=self.device)
        if out.numel() == 0:
            return out
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ws_plda_llr_matrix(
                self._handle(), _lib.ptr(e), _lib.ptr(n) if n is not None else None, nu, e.shape[0],
                _lib.ptr(t), t.shape[0], _lib.ptr(out), _lib.current_stream_ptr(self.device)),
                "ws_plda_llr_matrix")
        return out

    def llr_pairs(self, enroll_t, n_sessions, test_t, idx_e, idx_t) -> torch.Tensor:
        e = self._dev(enroll_t, torch.float64)
        t = self._dev(test_t, torch.float64)
        n, nu = self._sessions(n_sessions, e.shape[0])
        ie = self._dev(idx_e, torch.int32)
        it = self._dev(idx_t, torch.int32)
        if ie.shape != it.shape:
            raise ValueError("idx_e and idx_t differ in length")
        out = torch.empty((ie.shape[0],), dtype=torch.float64, device=self.device)
        if ie.shape[0] == 0:
            return out
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ws_plda_llr_pairs(
                self._handle(), _lib.ptr(e), _lib.ptr(n) if n is not None else None, nu, e.shape[0],
                _lib.ptr(t), t.shape[0], _lib.ptr(ie), _lib.ptr(it), ie.shape[0], _lib.ptr(out),
                _lib.current_stream_ptr(self.device)), "ws_plda_llr_pairs")
        return out

    # ---------------------------------------------------------- reference per-vector methods
    def transform_rows(self, x) -> torch.Tensor:
        """(N, D) float64 -> (N, D) float64: `transform_embedding` applied row-wise on the GPU."""
        x = self._dev(x, torch.float64)
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ws_plda_transform(self._handle(), _lib.ptr(x), x.shape[0],
                                                    _lib.ptr(out),
                                                    _lib.current_stream_ptr(self.device)),
                       "ws_plda_transform")
        return out

    def transform_embedding(self, embedding):
        """(D,) -> (D,) float64 numpy (two_cov_plda.py:156-163)."""
        x = np.asarray(embedding, dtype=np.float64)[None, :]
        return self.transform_rows(x).cpu().numpy()[0]

    def log_likelihood_ratio(self, transformed_train_embedding, transformed_test_embedding, n):
        out = self.llr_matrix(np.asarray(transformed_train_embedding, dtype=np.float64)[None, :],
                              [int(n)],
                              np.asarray(transformed_test_embedding, dtype=np.float64)[None, :])
        return float(out.cpu().numpy()[0, 0])

    # --------------------------------------------------------------------------------- eval_sv
    def eval_sv(self, enroll_scp, enroll_utt2spk, test_scp, trials, score_file,
                multisession_avg=True, indomain_scp=None):
        enroll_vecs = read_vec_scp(enroll_scp)
        labels = {}
        with open(enroll_utt2spk, "r") as fin:
            for line in fin:
                tokens = line.strip().split()
                labels[tokens[0]] = tokens[1]
        enroll_dict = OrderedDict()
        for key, vec in enroll_vecs.items():
            if key in labels:
                enroll_dict.setdefault(labels[key], []).append(vec)
            else:
                print("WARNING: {} not in utt2spk ({}), skipping it.".format(key, enroll_utt2spk))
        test_dict = read_vec_scp(test_scp)
        mean_vec = None
        if indomain_scp is not None:
            mean_vec = np.vstack(list(read_vec_scp(indomain_scp).values())).mean(0)
        trial_list = []
        with open(trials, "r") as read_trials:
            for line in read_trials:
                tokens = line.strip().split()
                if tokens:
                    trial_list.append(tokens)
        scores = score_plda(self, enroll_dict, test_dict, [(t[0], t[1]) for t in trial_list],
                            multisession_avg=multisession_avg, indomain_mean=mean_vec)
        with open(score_file, "w") as write_score:
            for segs, score in zip(trial_list, scores):
                write_score.write("{} {} {:.5f} {}\n".format(segs[0], segs[1], score,
                                                             segs[2] if len(segs) > 2 else ""))


def score_plda(plda: TwoCovPLDA, enroll_embeddings, test_embeddings, trials,
               multisession_avg=True, indomain_mean=None, return_tensor=False):
    """In-memory `eval_sv` (two_cov_plda.py:186-256).

    enroll_embeddings: dict model_id -> list/array of (D,) utterance embeddings
    test_embeddings:   dict utt_id -> (D,) embedding
    trials:            sequence of (model_id, utt_id[, ...])
    Returns the LLR of every trial, in order (numpy float64, or a GPU tensor)."""
    e_names = list(enroll_embeddings.keys())
    rows, offs, counts = [], [0], []
    for k in e_names:
        v = np.vstack(enroll_embeddings[k]).astype(np.float32)
        rows.append(v)
        offs.append(offs[-1] + v.shape[0])
        counts.append(1 if multisession_avg else v.shape[0])
    t_names = list(test_embeddings.keys())
    e_index = {k: i for i, k in enumerate(e_names)}
    t_index = {k: i for i, k in enumerate(t_names)}
    idx_e = np.fromiter((e_index[t[0]] for t in trials), dtype=np.int32, count=len(trials))
    idx_t = np.fromiter((t_index[t[1]] for t in trials), dtype=np.int32, count=len(trials))
    if not e_names or not t_names:
        if len(trials):
            raise KeyError("trials reference embeddings but a table is empty")
        return np.zeros(0, dtype=np.float64)
    enroll_t = plda.prepare_enroll(np.vstack(rows), offs, indomain_mean)
    test_t = plda.prepare_test(np.vstack([np.asarray(test_embeddings[k], dtype=np.float32)
                                          for k in t_names]), indomain_mean)
    # the pair kernel reads the enrollment row of consecutive trials from L1 when they share it (1 M trials: 203 us
    # grouped, 388 us in random order): score the list ordered by enrollment model -- trial files mostly are already,
    # then the sort is skipped -- and put the scores back in the caller's order
    order = None
    if len(idx_e) > 4096 and np.any(idx_e[1:] < idx_e[:-1]):
        order = np.argsort(idx_e, kind="stable")
        idx_e, idx_t = idx_e[order], idx_t[order]
    out = plda.llr_pairs(enroll_t, np.asarray(counts, dtype=np.int32), test_t, idx_e, idx_t)
    if order is not None:
        inv = torch.from_numpy(order).to(out.device)
        back = torch.empty_like(out)
        back[inv] = out
        out = back
    return out if return_tensor else out.cpu().numpy()


# ------------------------------------------------------------------------------- Kaldi <Plda>
def read_kaldi_plda(path):
    """Kaldi `<Plda>` object (mean vector, transform matrix, psi vector), binary (float or double
    payloads) or text, with the layout utils/plda/kaldi_utils.py:24-55 reads:
        binary:  '\\0B' '<Plda> ' vec mat vec '</Plda> '   (vec = 'FV '|'DV ' \\4 <i32 n> data;
                                                           mat = 'FM '|'DM ' \\4 <i32 r> \\4 <i32 c> data)
        text:    '<Plda> ' ' [ m ... ]\\n' ' [\\n  row\\n  row ]\\n' ' [ psi ... ]\\n' '</Plda> '
    Returns float64 (mu, transform, psi)."""
    with open(path, "rb") as fd:
        head = fd.read(2)
        if head == b"\0B":
            if fd.read(7) != b"<Plda> ":
                raise ValueError("not a Kaldi <Plda> file: " + str(path))

            def payload(tags, n_dims):
                tag = fd.read(3)
                if tag not in tags:
                    raise ValueError("unexpected Kaldi data tag %r in %s" % (tag, path))
                size = 4 if tag[:1] == b"F" else 8
                dims = []
                for _ in range(n_dims):
                    if fd.read(1) != b"\x04":
                        raise ValueError("bad int-size byte in " + str(path))
                    dims.append(struct.unpack("<i", fd.read(4))[0])
                count = int(np.prod(dims))
                buf = fd.read(count * size)
                if len(buf) != count * size:
                    raise ValueError("truncated Kaldi <Plda> file: " + str(path))
                return np.frombuffer(buf, dtype="<f4" if size == 4 else "<f8").astype(np.float64).reshape(dims)

            mu = payload((b"FV ", b"DV "), 1)
            tr = payload((b"FM ", b"DM "), 2)
            psi = payload((b"FV ", b"DV "), 1)
            tail = fd.read(8)
        else:
            if head + fd.read(5) != b"<Plda> ":
                raise ValueError("not a Kaldi <Plda> file: " + str(path))

            def numbers(line):
                return [float(t) for t in line.decode().replace("[", " ").replace("]", " ").split()]

            mu = np.array(numbers(fd.readline()), dtype=np.float64)
            rows = []
            opened = False
            while True:                         # " [" then one row per line, the last one closed by "]"
                line = fd.readline()
                if not line:
                    raise ValueError("truncated Kaldi <Plda> text file: " + str(path))
                opened = opened or b"[" in line
                vals = numbers(line)
                if vals:
                    rows.append(vals)
                if opened and b"]" in line:
                    break
            tr = np.array(rows, dtype=np.float64)
            psi = np.array(numbers(fd.readline()), dtype=np.float64)
            tail = fd.read(8)
        if tail != b"</Plda> ":
            raise ValueError("missing </Plda> terminator in " + str(path))
    if tr.shape != (mu.shape[0], mu.shape[0]) or psi.shape != mu.shape:
        raise ValueError("inconsistent <Plda> dimensions in %s: mean %s transform %s psi %s"
                         % (path, mu.shape, tr.shape, psi.shape))
    return mu, tr, psi
