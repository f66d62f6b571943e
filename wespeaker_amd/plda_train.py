"""Two-covariance PLDA training (EM) and unsupervised adaptation behind the reference's interface.

Mirrors `wespeaker/utils/plda/two_cov_plda.py`:
    PldaStats / TwoCovPLDA.__init__(scp_file, utt2spk_file, ...)   :38-107   -> collect_stats (GPU)
    train / em_one_iter / get_output                               :109-154  -> em_train (host f64)
    adapt(adapt_scp, ac_scale, wc_scale)                           :258-309  -> adapt (GPU stats + host f64)
and `wespeaker/bin/train_plda.py`, `bin/adapt_plda.py` (see `train_plda` / `adapt_plda` below).

Division of labour: everything that touches the N utterances -- mean
subtraction, length normalisation, class means, the D x D offset scatter / data covariance -- is one
`ws_plda_stats` call on the GPU (float64 MFMA).  What is left is D x D linear algebra on class-level
statistics (inv, cholesky, eigh: a few GFLOP in total), done here in numpy float64 exactly like the
reference, except that the per-speaker loop of em_one_iter is regrouped by the number of
utterances n (mix_var depends on the speaker only through n), which removes the O(#speakers) matrix
inversions without changing the mathematics.
"""
import collections

import numpy as np
import torch

from . import _lib
from .kaldi_io import read_vec_scp

PldaStats = collections.namedtuple(
    "PldaStats", ["dim", "num_example", "num_classes", "class_weight", "example_weight", "sum_",
                  "offset_scatter", "class_mean", "class_count"])


def _device(device=None):
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def gpu_stats(rows, group_offsets, mean_vec=None, normalize_length=False, device=None):
    """rows (N, D) float32 or float64 (kept as is: the reference's chain stays in float64) grouped by
    class, group_offsets int32[C+1] -> (class_mean (C, D), offset_scatter (D, D)) numpy float64,
    computed by ws_plda_stats."""
    dev = _device(device)
    L = _lib.lib()
    rows = np.asarray(rows)
    is64 = rows.dtype == np.float64
    x = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float64 if is64 else np.float32)).to(dev)
    n, dim = int(x.shape[0]), int(x.shape[1])
    offs = torch.from_numpy(np.ascontiguousarray(group_offsets, dtype=np.int32)).to(dev)
    n_groups = int(offs.numel()) - 1
    mv = None
    if mean_vec is not None:
        mv = torch.from_numpy(np.ascontiguousarray(mean_vec, dtype=np.float64)).to(dev)
    cm = torch.empty((n_groups, dim), dtype=torch.float64, device=dev)
    sc = torch.empty((dim, dim), dtype=torch.float64, device=dev)
    need = int(L.ws_plda_stats_scratch(n, dim))
    scratch = torch.empty((need,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.ws_plda_stats(_lib.ptr(x), int(is64), n, dim, _lib.ptr(offs), n_groups,
                                   _lib.ptr(mv) if mv is not None else None, int(bool(normalize_length)),
                                   _lib.ptr(cm), _lib.ptr(sc), _lib.ptr(scratch), need,
                                   _lib.current_stream_ptr(dev)), "ws_plda_stats")
        out = cm.cpu().numpy(), sc.cpu().numpy()
    return out


def collect_stats(embeddings_dict, train_mean_vec=None, normalize_length=False, device=None):
    """TwoCovPLDA.__init__ :95-107 -- one PldaStats.add_samples(1.0, mat) per speaker, on the GPU.
    embeddings_dict: speaker -> list of (D,) vectors (insertion order = class order)."""
    mats, offs = [], [0]
    for mat in embeddings_dict.values():
        mat = np.vstack(mat)
        mats.append(mat)
        offs.append(offs[-1] + mat.shape[0])
    rows = np.vstack(mats)
    class_mean, scatter = gpu_stats(rows, offs, train_mean_vec, normalize_length, device)
    counts = np.diff(np.asarray(offs)).astype(np.int64)
    c = len(mats)
    return PldaStats(dim=rows.shape[1], num_example=int(counts.sum()), num_classes=c,
                     class_weight=float(c), example_weight=float(counts.sum()),
                     sum_=class_mean.sum(0), offset_scatter=scatter, class_mean=class_mean,
                     class_count=counts)


def get_data_for_plda(scp_file, utt2spk_file):
    """plda_utils.py:64-81 with our ark/scp reader."""
    samples_dict = read_vec_scp(scp_file)
    labels = {}
    with open(utt2spk_file, "r") as fin:
        for line in fin:
            tokens = line.strip().split()
            labels[tokens[0]] = tokens[1]
    samples, model_dict = [], collections.OrderedDict()
    for key, vec in samples_dict.items():
        samples.append(vec)
        if key in labels:
            model_dict.setdefault(labels[key], []).append(vec)
        else:
            print("WARNING: {} not in utt2spk ({}), skipping it.".format(key, utt2spk_file))
    return np.vstack(samples), model_dict


def em_one_iter(stats: PldaStats, B, W):
    """two_cov_plda.py:116-142 regrouped by utterance count (class weight is 1.0 as in :106)."""
    inv = np.linalg.inv
    W_stats = stats.offset_scatter.copy()
    W_count = stats.example_weight - stats.class_weight
    B_stats, B_count = np.zeros((stats.dim, stats.dim)), 0.0
    B_inv, W_inv = inv(B), inv(W)
    m_all = stats.class_mean - stats.sum_ / stats.class_weight
    for n in np.unique(stats.class_count):
        sel = stats.class_count == n
        k = float(sel.sum())
        mix_var = inv(B_inv + n * W_inv)
        m = m_all[sel]
        w = (n * (m @ W_inv.T)) @ mix_var.T           # rows: mix_var @ (n W_inv m)
        m_w = m - w
        B_stats += k * mix_var + w.T @ w
        B_count += k
        W_stats += n * (k * mix_var + m_w.T @ m_w)
        W_count += k
    W = W_stats / W_count
    B = B_stats / B_count
    return 0.5 * (B + B.T), 0.5 * (W + W.T)


def compute_normalizing_transform(covar):
    """plda_utils.py:84-90."""
    try:
        c = np.linalg.cholesky(covar)
    except np.linalg.LinAlgError:
        c = np.linalg.cholesky(covar + np.eye(covar.shape[0]) * 1e-6)
    return np.linalg.inv(c)


def get_output(stats: PldaStats, B, W):
    """two_cov_plda.py:144-157 -> (mu, transform, psi, offset)."""
    mu = stats.sum_ / stats.class_weight
    transform1 = compute_normalizing_transform(W)
    B_proj = transform1 @ B @ transform1.T
    s, U = np.linalg.eigh(B_proj)
    s = np.where(s > 0.0, s, 0.0)
    idx = np.argsort(-s)                              # sort_svd, plda_utils.py:93-103
    s, U = s[idx], U[:, idx]
    transform = U.T @ transform1
    return mu, transform, s, -1.0 * (transform @ mu)


def em_train(stats: PldaStats, num_em_iters, verbose=True):
    B, W = np.eye(stats.dim), np.eye(stats.dim)
    for i in range(num_em_iters):
        if verbose:
            print("Plda estimation %d of %d" % (i, num_em_iters))
        B, W = em_one_iter(stats, B, W)
        if verbose:
            print("Trace of W:", np.trace(W), "Trace of B:", np.trace(B))
    return B, W


def adapt_parameters(mu, transform, psi, adp_rows, normalize_length, ac_scale=0.5, wc_scale=0.5,
                     device=None):
    """two_cov_plda.py:258-300 (BUT's unsupervised adaptation).  The data mean and np.cov of the
    adaptation set come from ws_plda_stats with a single class; the rest is D x D float64."""
    import scipy.linalg as spl
    adp_rows = np.ascontiguousarray(adp_rows, dtype=np.float32)
    n = adp_rows.shape[0]
    mean_vec = adp_rows.astype(np.float64).mean(0) if n else None
    cm, scatter = gpu_stats(adp_rows, [0, n], mean_vec, normalize_length, device)
    data_cov = scatter / (n - 1)                      # np.cov(adp_data.T)
    mu_adp = cm[0]                                    # np.mean(adp_data, axis=0)
    W = np.linalg.inv(transform.T.dot(transform))
    W = (W + W.T) / 2
    B = np.linalg.inv((transform.T / psi).dot(transform))
    B = (B + B.T) / 2
    T = B + W
    v, e = spl.eigh(data_cov, (T + T.T) / 2)
    iet = np.linalg.inv(e.T)
    excess = iet[:, v > 1].dot(np.diag(np.sqrt(v[v > 1] - 1)))
    V_adp = excess * np.sqrt(ac_scale)
    B_adp = B + V_adp.dot(V_adp.T)
    U_adp = excess * np.sqrt(wc_scale)
    W_adp = W + U_adp.dot(U_adp.T)
    A, Bm = (B_adp + B_adp.T) / 2.0, (W_adp + W_adp.T) / 2.0
    eps = 1e-9
    D, V = np.linalg.eigh(Bm)
    T1 = np.dot(np.diag(1.0 / np.sqrt(D + eps)), V.T)
    A1 = np.dot(np.dot(T1, A), T1.T)
    _, T2 = np.linalg.eigh(A1)
    Tj = np.dot(T2.T, T1)
    A2 = np.dot(np.dot(Tj, A), Tj.T)
    return mu_adp, Tj, np.diag(A2).copy(), -1.0 * np.matmul(Tj, mu_adp)


# ------------------------------------------------------- bin/train_plda.py, bin/adapt_plda.py
def train_plda(scp_path, utt2spk, indim, exp_dir, iter=5, type="2cov"):   # noqa: A002 (reference flag names)
    """bin/train_plda.py: --type 2cov --scp_path --utt2spk --indim --exp_dir --iter.
    Saves `exp_dir/plda` like the reference: HDF5 (written through libhdf5, wespeaker_amd/hdf5_io.py; numpy's
    .npz container under the same name where no HDF5 library exists -- TwoCovPLDA.load_model sniffs the format)."""
    import os
    from .plda import TwoCovPLDA
    if type != "2cov":
        raise ValueError("only the kaldi 2cov version is supported (as in the reference)")
    plda = TwoCovPLDA(scp_file=scp_path, utt2spk_file=utt2spk, embed_dim=indim)
    plda.train(iter)
    path = os.path.join(exp_dir, "plda")
    plda.save_model(path)
    return path


def adapt_plda(adp_scp, mdl_org, mdl_adp, across_class_scale=0.5, within_class_scale=0.5,
               mdl_format="wespeaker"):
    """bin/adapt_plda.py: -ad -as -ws -mo -ma -mf."""
    from .plda import TwoCovPLDA
    plda = TwoCovPLDA.load_model(mdl_org, mdl_format == "kaldi")
    adapted = plda.adapt(adp_scp, across_class_scale, within_class_scale)
    adapted.save_model(mdl_adp)
    return adapted
