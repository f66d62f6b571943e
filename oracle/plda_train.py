"""ORACLE (test infrastructure, never shipped): numpy restatement of the reference's
two-covariance PLDA training (EM) and unsupervised adaptation, per-speaker loops included.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows wespeaker/utils/plda/two_cov_plda.py in /root/reference:
  :38-66    PldaStats.add_samples  (np.mean / matmul in the embeddings' dtype -- float32 from the
                                    ark files -- accumulated into float64)
  :95-107   constructor pre-processing (train-set mean, norm_embeddings, one add_samples per speaker)
  :116-142  em_one_iter            (per-speaker inv(B_inv + n W_inv))
  :144-157  get_output             (Cholesky whitening of W, eigh of the projected B, sort_svd)
  :258-300  adapt                  (np.cov of the adaptation data, generalised eigh, re-diagonalisation)

PINNED: tests/golden/plda_train_ref.npz holds B, W, mu, psi and LLRs of the reference's own
TwoCovPLDA(scp_file, utt2spk_file, ...).train(3) / .adapt(...) run in this container on ark/scp
files (oracle/make_golden.py:make_plda_train); tests/test_oracle_golden.py checks this file
against it.
"""
import math

import numpy as np
from numpy.linalg import inv


def norm_embeddings(x):
    scale = math.sqrt(x.shape[-1])
    return (scale * x.transpose() / np.linalg.norm(x, axis=1)).transpose()


def collect_stats(class_mats, train_mean_vec, normalize_length):
    """class_mats: list of (n_c, D) arrays in the embeddings' own dtype."""
    dim = class_mats[0].shape[1]
    st = {"dim": dim, "class_weight": 0.0, "example_weight": 0.0, "sum_": np.zeros(dim),
          "offset_scatter": np.zeros((dim, dim)), "classinfo": []}
    for mat in class_mats:
        mat = mat - train_mean_vec
        if normalize_length:
            mat = norm_embeddings(mat)
        n = mat.shape[0]
        mean = np.mean(mat, axis=0)
        tmp = mat - mean
        st["offset_scatter"] += np.matmul(tmp.T, tmp)
        st["classinfo"].append((1.0, n, mean))
        st["class_weight"] += 1.0
        st["example_weight"] += n
        st["sum_"] += mean
    return st


def em_one_iter(st, B, W):
    dim = st["dim"]
    B_stats, B_count = np.zeros((dim, dim)), 0.0
    W_stats = st["offset_scatter"].copy()
    W_count = st["example_weight"] - st["class_weight"]
    B_inv, W_inv = inv(B), inv(W)
    for weight, n, mu in st["classinfo"]:
        m = mu - st["sum_"] / st["class_weight"]
        mix_var = inv(B_inv + n * W_inv)
        w = np.matmul(mix_var, n * np.matmul(W_inv, m))
        m_w = m - w
        B_stats += weight * (mix_var + np.outer(w, w))
        B_count += weight
        W_stats += weight * n * (mix_var + np.outer(m_w, m_w))
        W_count += weight
    W = W_stats / W_count
    B = B_stats / B_count
    return 0.5 * (B + B.T), 0.5 * (W + W.T)


def get_output(st, B, W):
    mu = st["sum_"] / st["class_weight"]
    try:
        c = np.linalg.cholesky(W)
    except np.linalg.LinAlgError:
        c = np.linalg.cholesky(W + np.eye(W.shape[0]) * 1e-6)
    transform1 = inv(c)
    B_proj = np.matmul(np.matmul(transform1, B), transform1.T)
    s, U = np.linalg.eigh(B_proj)
    s = np.where(s > 0.0, s, 0.0)
    idx = np.argsort(-s)
    s, U = s[idx], U.T[idx].T
    transform = np.matmul(U.T, transform1)
    return {"mu": mu, "transform": transform, "psi": s, "offset": -1.0 * np.matmul(transform, mu)}


def train(class_mats, num_iters, subtract_train_set_mean=False, normalize_length=False,
          samples=None):
    """samples: all training rows in scp order (the reference takes the train-set mean of THAT
    float32 array, :96-98); defaults to the class-grouped order."""
    all_rows = np.vstack(class_mats) if samples is None else samples
    tm = all_rows.mean(0) if subtract_train_set_mean else np.zeros(all_rows.shape[1])
    st = collect_stats(class_mats, tm, normalize_length)
    B, W = np.eye(st["dim"]), np.eye(st["dim"])
    for _ in range(num_iters):
        B, W = em_one_iter(st, B, W)
    out = get_output(st, B, W)
    out.update(B=B, W=W, normalize_length=normalize_length)
    return out


def adapt(p, adp_data, ac_scale=0.5, wc_scale=0.5):
    import scipy.linalg as spl
    mean_vec = adp_data.mean(0)
    adp_data = adp_data - mean_vec
    if p["normalize_length"]:
        adp_data = norm_embeddings(adp_data)
    tr, psi = p["transform"], p["psi"]
    W = inv(tr.T.dot(tr))
    W = (W + W.T) / 2
    B = inv((tr.T / psi).dot(tr))
    B = (B + B.T) / 2
    T = B + W
    data_cov = np.cov(adp_data.T)
    v, e = spl.eigh(data_cov, (T + T.T) / 2)
    iet = inv(e.T)
    excess = iet[:, v > 1].dot(np.diag(np.sqrt(v[v > 1] - 1)))
    V_adp = excess * np.sqrt(ac_scale)
    B_adp = B + V_adp.dot(V_adp.T)
    U_adp = excess * np.sqrt(wc_scale)
    W_adp = W + U_adp.dot(U_adp.T)
    mu = np.mean(adp_data, axis=0)
    A, Bm = (B_adp + B_adp.T) / 2.0, (W_adp + W_adp.T) / 2.0
    D, V = np.linalg.eigh(Bm)
    T1 = np.dot(np.diag(1.0 / np.sqrt(D + 1e-9)), V.T)
    A1 = np.dot(np.dot(T1, A), T1.T)
    _, T2 = np.linalg.eigh(A1)
    Tj = np.dot(T2.T, T1)
    A2 = np.dot(np.dot(Tj, A), Tj.T)
    return {"mu": mu, "transform": Tj, "psi": np.diag(A2).copy(),
            "offset": -1.0 * np.matmul(Tj, mu), "normalize_length": False}
