"""ORACLE (test infrastructure, never shipped): numpy float32 restatement of the reference's
cosine scoring and AS-norm / S-norm.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows (file:line in /root/reference):
  wespeaker/bin/score.py:38-72        trials_cosine_score: emb - mean_vec, cosine per trial
                                      (sklearn.metrics.pairwise.cosine_similarity on (1, D) rows)
  wespeaker/bin/score_norm.py:26-36   get_mean_std: rows / |row|, emb x cohort^T, sort each row
                                      descending, mean and np.std (ddof 0) of the first top_n
  wespeaker/bin/score_norm.py:84-110  asnorm: top_n as given; snorm: top_n = #cohort;
                                      normed = 0.5 ((s - mu_e)/sd_e + (s - mu_t)/sd_t); the score
                                      `s` is re-read from the 5-decimal text of the score file;
                                      extra columns |enroll|, |test|, mu_e, mu_t

PINNED: tests/golden/score_ref.npz was generated in this container by running the reference's
own score.py / score_norm.py functions (imported from /root/reference with a kaldiio stub that
reads the same ark files, oracle/make_golden.py); tests/test_oracle_golden.py checks this file
against it.
"""
import numpy as np


def cosine_pairs(emb, mean_vec, idx_a, idx_b):
    """score.py:44-63 for index lists.  emb (n, D) float32; returns float32 cosines."""
    x = np.asarray(emb, dtype=np.float32)
    if mean_vec is not None:
        x = x - np.asarray(mean_vec, dtype=np.float32)
    out = np.empty(len(idx_a), dtype=np.float32)
    for p, (a, b) in enumerate(zip(idx_a, idx_b)):
        u, v = x[a], x[b]
        out[p] = np.dot(u, v) / (np.sqrt(np.dot(u, u)) * np.sqrt(np.dot(v, v)))
    return out


def get_mean_std(emb, cohort, top_n):
    """score_norm.py:26-36, dtype-preserving like the reference (float32 in the recipes)."""
    emb = emb / np.sqrt(np.sum(emb ** 2, axis=1, keepdims=True))
    cohort = cohort / np.sqrt(np.sum(cohort ** 2, axis=1, keepdims=True))
    s = np.matmul(emb, cohort.T)
    s = np.sort(s, axis=1)[:, ::-1]
    top = s[:, :top_n]
    return np.mean(top, axis=1), np.std(top, axis=1)


def score_norm(method, top_n, scores, idx_e, idx_t, enroll_emb, test_emb, cohort_emb, mean_vec=None):
    """score_norm.py:84-110 on arrays.  scores: the raw trial scores AS READ FROM THE SCORE FILE
    (i.e. already rounded to 5 decimals by score.py).  Returns dict of per-trial columns."""
    mv = 0.0 if mean_vec is None else np.asarray(mean_vec)
    e, t, c = enroll_emb - mv, test_emb - mv, cohort_emb - mv
    if method == "asnorm":
        n_top = top_n
    elif method == "snorm":
        n_top = c.shape[0]
    else:
        raise ValueError(method)
    e_mean, e_std = get_mean_std(e, c, n_top)
    t_mean, t_std = get_mean_std(t, c, n_top)
    s = np.asarray(scores, dtype=np.float64)
    normed = 0.5 * ((s - e_mean[idx_e]) / e_std[idx_e] + (s - t_mean[idx_t]) / t_std[idx_t])
    return {"normed": normed, "enroll_mag": np.linalg.norm(e, axis=1)[idx_e],
            "test_mag": np.linalg.norm(t, axis=1)[idx_t], "enroll_mean": e_mean[idx_e],
            "test_mean": t_mean[idx_t], "e_mean": e_mean, "e_std": e_std, "t_mean": t_mean,
            "t_std": t_std}
