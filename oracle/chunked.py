"""ORACLE (test infrastructure, never shipped): the native runtime's chunk-and-average rule.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows runtime/core/speaker/speaker_engine.cc in /root/reference:
  :63-75    ApplyMean                (per-chunk CMN)
  :83-138   ExtractFeature           (chunk cutting; head-frame completion of the last chunk;
                                      cyclic tiling of an utterance shorter than one chunk)
  :140-159  ExtractEmbedding         (sum of chunk embeddings / chunk count)

Pinned: the reference's own speaker_engine.cc + feature_pipeline.cc + fbank.h are compiled in place
into oracle/_ref/libref_engine.so (oracle/Makefile, oracle/ref_engine_wrap.cc; only the ONNX/MNN
model back-end is replaced, through the reference's own SpeakerModel interface, by the pinned
oracle/ecapa.py forward) and run by oracle/make_golden.py -> tests/golden/chunked_ref.npz: chunk
counts, averaged embeddings and the padded chunk tensors of six cases (head-frame completion,
cyclic tiling, exact multiple, full mode, short chunks, one-frame tail).
tests/test_oracle_golden.py holds this restatement to them.
"""
import numpy as np


def chunk_frames(total, sample_rate, samples_per_chunk):
    """-> (frames per chunk, list of per-chunk source-frame index arrays)."""
    if samples_per_chunk <= 0:                       # full mode (:90-95)
        return total, [np.arange(total)]
    ms = sample_rate // 1000
    cf = 1 + (samples_per_chunk - ms * 25) // (ms * 10)      # :101-103
    chunks = []
    pos = 0
    while total - pos >= cf:                          # :108-112
        chunks.append(np.arange(pos, pos + cf))
        pos += cf
    last = total - pos
    if last > 0:                                      # :114-133
        idx = list(range(pos, pos + last))
        if not chunks:                                # wav_len < chunk_len: tile, then a prefix
            num_pad = cf // last
            for _ in range(1, num_pad):
                idx += idx[:last]
            idx += idx[:cf - len(idx)]
        else:                                         # head frames of the first chunk
            idx += list(chunks[0][:cf - len(idx)])
        assert len(idx) == cf
        chunks.append(np.array(idx))
    return cf, chunks


def extract_chunked(feats, sample_rate, samples_per_chunk, forward):
    """feats (total, F) WITHOUT CMN; forward maps (B, T, F) float32 -> (B, E).  -> ((E,), n_chunks)."""
    feats = np.asarray(feats, dtype=np.float32)
    _, chunks = chunk_frames(feats.shape[0], sample_rate, samples_per_chunk)
    batch = np.stack([feats[i] - feats[i].mean(0, keepdims=True) for i in chunks])
    emb = np.asarray(forward(batch), dtype=np.float32)
    acc = np.zeros(emb.shape[1], dtype=np.float32)
    for e in emb:                                     # :147-154 accumulate in chunk order
        acc += e
    return acc / np.float32(len(chunks)), len(chunks)
