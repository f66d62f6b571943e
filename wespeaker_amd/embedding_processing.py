"""Embedding-processing chain (mean-subtract | length-norm | lda | whitening) behind the reference's
interface, `wespeaker/utils/embedding_processing.py`:

    chain_string_to_dict            :23-67
    Lda                             :70-178   (scatter statistics + eigen-decomposition, `__call__`)
    Length_norm / Whitening / MeanSubtraction   :181-216
    EmbeddingProcessingChain        :219-271  (`__call__`, save / load (pickle), update_link)
and the tools bin/prep_embd_proc.py, bin/apply_embd_proc.py (`prep_embd_proc`, `apply_embd_proc`).

Everything that touches the N embeddings runs on the GPU: a link is applied to all rows by
`ws_rows_affine` (y = (x - sub) M, optional length normalisation, float64), and the LDA / mean
statistics (per-speaker means, within-class scatter) come from `ws_plda_stats`.  The D x D algebra
(eigh, whitening) is numpy / scipy float64 exactly like the reference.
"""
import pickle
import re

import numpy as np
import torch

from . import _lib, plda_train
from .kaldi_io import VectorWriter, read_vec_scp


_OPTION = re.compile(r"--\s*([^\s=]+)\s*[=\s]\s*(\S+)\s*")


def chain_string_to_dict(chain_string=None):
    """'mean-subtract --scp a.scp | lda --scp b.scp --utt2spk u --dim 100 | length-norm' ->
    [['mean-subtract', {'scp': 'a.scp'}], ['lda', {...}], ['length-norm', {}]]: links are separated
    by '|', every option is `--name value` or `--name=value`, values stay strings
    (same grammar and return structure as embedding_processing.py:23-67)."""
    if chain_string is None:
        return []
    parsed = []
    for piece in chain_string.split("|"):
        head, dashes, tail = piece.partition("--")
        options = {}
        pos, tail = 0, dashes + tail
        while pos < len(tail):
            hit = _OPTION.match(tail, pos)
            if hit is None:
                raise AssertionError("malformed option in chain link %r" % piece)
            options[hit.group(1)] = hit.group(2)
            pos = hit.end()
        parsed.append([head.strip(" "), options])
    return parsed


def _apply_link(embd, sub=None, M=None, normalize=False):
    """rows (n, d_in) numpy float32/float64 -> (n, d_out) numpy float64 through ws_rows_affine."""
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    x = np.ascontiguousarray(embd)
    is64 = x.dtype == np.float64
    if not is64:
        x = x.astype(np.float32, copy=False)
    xt = torch.from_numpy(x).to(dev)
    n, d_in = int(xt.shape[0]), int(xt.shape[1])
    st = torch.from_numpy(np.ascontiguousarray(sub, dtype=np.float64)).to(dev) if sub is not None else None
    mt = torch.from_numpy(np.ascontiguousarray(M, dtype=np.float64)).to(dev) if M is not None else None
    d_out = int(mt.shape[1]) if mt is not None else d_in
    out = torch.empty((n, d_out), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().ws_rows_affine(
            _lib.ptr(xt), int(is64), n, d_in, _lib.ptr(st) if st is not None else None,
            _lib.ptr(mt) if mt is not None else None, d_out, int(bool(normalize)), _lib.ptr(out),
            _lib.current_stream_ptr(dev)), "ws_rows_affine")
        res = out.cpu().numpy()
    return res


class Lda:

    def compute_mean_and_lda_scatter_matrices(self, scp_file, utt2spk_file, equal_speaker_weight=False,
                                              current_chain=None):
        """embedding_processing.py:72-131.  Speakers with a single utterance are skipped like there;
        per-speaker means and the within-class scatter come from one ws_plda_stats call."""
        if current_chain is None:
            current_chain = lambda e: e  # noqa: E731
        _, embeddings_dict = plda_train.get_data_for_plda(scp_file, utt2spk_file)
        mats = [np.vstack(v) for v in embeddings_dict.values() if len(v) > 1]
        n_skipped = len(embeddings_dict) - len(mats)
        counts = np.array([m.shape[0] for m in mats])
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        rows = current_chain(np.vstack(mats))
        means, scatter = plda_train.gpu_stats(rows, offs)          # float64 rows stay float64
        print("  #speakers: {}, #used {}, #skipped {} (only having one utterances)".format(
            len(embeddings_dict), len(mats), n_skipped))
        if equal_speaker_weight:
            mean = means.mean(0)
            between = np.cov(means, rowvar=False, bias=True)
            # sum of the per-speaker (biased) covariances / #speakers: needs the per-speaker split
            covs = [np.cov(rows[a:b], rowvar=False, bias=True) for a, b in zip(offs[:-1], offs[1:])]
            within = np.sum(covs, axis=0) / len(embeddings_dict)
        else:
            mean = np.sum(counts[:, None] * means, axis=0) / counts.sum()
            between = np.cov(means, rowvar=False, bias=True, fweights=counts)
            within = scatter / counts.sum()          # sum_s n_s cov_s(bias) / sum_s n_s
        return mean, between, within

    def __init__(self, args, current_chain=None):
        """Attributes `m` (global mean) and `lda` (D x dim projection) as in the reference (:132-178):
        the within-class covariance is whitened through its own eigen-decomposition (eigenvalues
        floored at eps x the largest, as Kaldi does), the between-class covariance is diagonalised in
        that whitened space, and the `dim` leading directions are mapped back."""
        import scipy.linalg as spl
        out_dim = int(args['dim'])
        floor_ratio = float(args.get('eps', 1e-6))
        self.m, between, within = self.compute_mean_and_lda_scatter_matrices(
            args['scp'], args['utt2spk'], current_chain=current_chain)
        w_val, w_vec = spl.eigh(within)
        w_val = np.maximum(w_val, w_val.max() * floor_ratio)
        whiten = w_vec / np.sqrt(w_val)                   # columns scaled: whiten.T @ within @ whiten = I
        b_val, b_vec = spl.eigh(whiten.T @ between @ whiten)      # ascending eigenvalues
        self.lda = whiten @ b_vec[:, -out_dim:]
        print("LDA: %d -> %d dimensions; between-class eigenvalue mass kept %.2f of %.2f"
              % (b_val.shape[0], out_dim, b_val[-out_dim:].sum(), b_val.sum()))

    def __call__(self, embd):
        return _apply_link(embd, sub=self.m, M=self.lda)


class Length_norm:

    def __init__(self, args=None, current_chain=None):
        pass

    def __call__(self, embd):
        return _apply_link(embd, normalize=True)


class Whitening:
    """Declared but empty in the reference (:198-201): constructing it works, calling it does not."""

    def __init__(self, args, current_chain):
        pass


class MeanSubtraction:

    def __init__(self, args, current_chain=None):
        if current_chain is None:
            current_chain = lambda e: e  # noqa: E731
        e = np.vstack(list(read_vec_scp(args['scp']).values()))
        rows = np.asarray(current_chain(e))
        means, _ = plda_train.gpu_stats(rows, [0, rows.shape[0]])
        self.mean = means[0]

    def __call__(self, embd):
        return _apply_link(embd, sub=self.mean)


class EmbeddingProcessingChain:
    string2class = {'lda': Lda, 'length-norm': Length_norm, 'whitening': Whitening,
                    'mean-subtract': MeanSubtraction}

    def __init__(self, chain=None):
        # `chain_of_classes` is the pickled attribute of the reference (:232-244) -- name kept so that
        # chains saved by either implementation load in the other
        self.chain_of_classes = []
        for method, options in chain_string_to_dict(chain):
            self.chain_of_classes.append(self._build(method, options))

    def _build(self, method, options):
        if method not in self.string2class:
            raise KeyError("unknown embedding-processing link %r (known: %s)"
                           % (method, ", ".join(sorted(self.string2class))))
        print("embedding processing: building link %r with %s" % (method, options))
        return self.string2class[method](options, self)

    def __call__(self, embd):
        for link in self.chain_of_classes:
            embd = link(embd)
        return embd

    def save(self, path, data_format='pickle'):
        with open(path, 'wb') as f:
            pickle.dump(self.chain_of_classes, f)
        print("embedding processing: %d links written to %s" % (len(self.chain_of_classes), path))

    def load(self, path, data_format='pickle'):
        with open(path, 'rb') as f:
            self.chain_of_classes = pickle.load(f)
        print("embedding processing: %d links read from %s" % (len(self.chain_of_classes), path))

    def update_link(self, link_no_to_replace, new_link):
        """Rebuild ONE link in place (:246-271); the replacement is estimated on the output of the
        links in front of it, so it is constructed against the truncated chain."""
        spec = chain_string_to_dict(new_link)
        assert len(spec) == 1, "Length of new chain must be one."
        if not 0 <= link_no_to_replace < len(self.chain_of_classes):
            return
        method, options = spec[0]
        tail = self.chain_of_classes[link_no_to_replace + 1:]
        self.chain_of_classes = self.chain_of_classes[:link_no_to_replace]
        self.chain_of_classes.append(self._build(method, options))
        self.chain_of_classes.extend(tail)


# --------------------------------------------------- bin/prep_embd_proc.py, bin/apply_embd_proc.py
def prep_embd_proc(chain='whitening | length-norm ', path=None):
    c = EmbeddingProcessingChain(chain=chain)
    if path:
        c.save(path)
    return c


def apply_embd_proc(path, input, output):  # noqa: A002 (reference flag names)
    """bin/apply_embd_proc.py: --path --input (scp) --output ('x.ark,scp' | 'x.ark')."""
    chain = EmbeddingProcessingChain()
    chain.load(path)
    d = read_vec_scp(input)
    utt = list(d.keys())
    embd = chain(np.array(list(d.values())))
    print("Read {} embeddings of dimension {}.".format(len(utt), embd.shape[1] if len(utt) else 0))
    if output.endswith('ark,scp') or output.endswith('scp,ark'):
        stem = output[:-len('ark,scp')]
        with VectorWriter(stem + "ark", stem + "scp") as w:
            for u, e in zip(utt, embd):
                w(u, e)
    elif output.endswith('ark'):
        with VectorWriter(output) as w:
            for u, e in zip(utt, embd):
                w(u, e)
    else:
        raise ValueError("output must end in 'ark,scp', 'scp,ark' or 'ark'")
    return embd
