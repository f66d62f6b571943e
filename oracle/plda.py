"""ORACLE (test infrastructure, never shipped): numpy float64 restatement of the
reference's two-covariance PLDA scoring.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows (file:line in /root/reference):
  wespeaker/utils/plda/two_cov_plda.py:156-163  transform_embedding
  wespeaker/utils/plda/two_cov_plda.py:165-184  log_likelihood_ratio (per-trial form)
  wespeaker/utils/plda/two_cov_plda.py:204-256  eval_sv pre-processing + trial loop
  wespeaker/utils/plda/plda_utils.py:46-58      norm_embeddings (sqrt(D) x / |x|)

PINNED: tests/golden/plda_ref.npz was generated in this container by running the reference's
own TwoCovPLDA methods (imported from /root/reference with stub h5py/kaldiio/kaldi_io modules,
oracle/make_golden.py) on seeded parameters; tests/test_oracle_golden.py checks this file
against it to 1e-12.
"""
import math

import numpy as np

M_LOG_2PI = 1.8378770664093454835606594728112


def norm_embeddings(x):
    x = np.asarray(x, dtype=np.float64)
    scale = math.sqrt(x.shape[-1])
    if x.ndim == 2:
        return scale * x / np.linalg.norm(x, axis=1, keepdims=True)
    return scale * x / np.linalg.norm(x)


def transform_embedding(p, x):
    """p: dict(mu, transform, psi, offset, normalize_length).  x: (D,) -> (D,) float64."""
    y = np.matmul(p["transform"], np.asarray(x, dtype=np.float64)) + p["offset"]
    if p["normalize_length"]:
        y = (math.sqrt(y.shape[0]) / np.linalg.norm(y)) * y
    return y


def log_likelihood_ratio(p, enroll_t, test_t, n):
    """Per-trial LLR exactly in the reference's operation order."""
    psi = p["psi"]
    dim = psi.shape[0]
    mean = n * psi / (n * psi + 1.0) * enroll_t
    variance = 1.0 + psi / (n * psi + 1.0)
    logdet = np.sum(np.log(variance))
    sqdiff = np.power(test_t - mean, 2.0)
    given = -0.5 * (logdet + M_LOG_2PI * dim + np.dot(sqdiff, 1.0 / variance))
    variance = psi + 1.0
    logdet = np.sum(np.log(variance))
    without = -0.5 * (logdet + M_LOG_2PI * dim + np.dot(np.power(test_t, 2.0), 1.0 / variance))
    return given - without


def prepare_enroll(p, utt_embeddings, mean_vec=None, multisession_avg=True):
    """eval_sv :216-235 for ONE enrollment model: list/array of utterance embeddings ->
    (transformed vector, n)."""
    v = np.vstack(utt_embeddings).astype(np.float64)
    n = 1 if multisession_avg else v.shape[0]
    if mean_vec is not None:
        v = v - mean_vec
    m = np.mean(v, 0)
    if p["normalize_length"]:
        m = norm_embeddings(m)
    return transform_embedding(p, m), n


def prepare_test(p, emb, mean_vec=None):
    """eval_sv :237-244 for ONE test utterance."""
    v = np.asarray(emb, dtype=np.float64)
    if mean_vec is not None:
        v = v - mean_vec
    if p["normalize_length"]:
        v = norm_embeddings(v)
    return transform_embedding(p, v)


def llr_pairs(p, enroll_t, n_enroll, test_t, idx_e, idx_t):
    """Loop form over an explicit trial list (what eval_sv's trial loop does)."""
    out = np.empty(len(idx_e), dtype=np.float64)
    for k, (i, j) in enumerate(zip(idx_e, idx_t)):
        out[k] = log_likelihood_ratio(p, enroll_t[i], test_t[j], n_enroll[i])
    return out


def llr_matrix_vectorised(p, enroll_t, n_enroll, test_t):
    """Vectorised numpy float64 closed form (the 'fair' CPU baseline):
    LLR[i,j] = K(n_i) - 1/2 (sum_d a t^2 + sum_d b e^2) + sum_d g e t."""
    psi = p["psi"][None, :]
    n = np.asarray(n_enroll, dtype=np.float64)[:, None]
    c = n * psi / (n * psi + 1.0)
    v = 1.0 + psi / (n * psi + 1.0)
    K = -0.5 * (np.sum(np.log(v), 1) - np.sum(np.log(psi + 1.0)))
    a = 1.0 / v - 1.0 / (psi + 1.0)
    b = c * c / v
    g = c / v
    E = np.asarray(enroll_t, dtype=np.float64)
    T = np.asarray(test_t, dtype=np.float64)
    return (K[:, None] - 0.5 * ((T * T) @ a.T).T - 0.5 * np.sum(b * E * E, 1)[:, None]
            + (g * E) @ T.T)
