"""Batch extraction driver: data list -> per-job `xvector_XXX.ark/scp` -> merged `xvector.scp`.

The MI355X twin of the reference's batch path
    tools/extract_embedding.sh:39-67      split the list into nj contiguous sub-lists, job k -> GPU k % G,
                                          `cat xvector_*.scp > xvector.scp`, count check -> extract.result
    wespeaker/bin/extract.py:33-139       one job: config + checkpoint -> Dataset -> model -> ark,scp writer
    wespeaker/dataset/dataset.py:136-268  the test-time pipeline (no shuffle / filter / augmentation):
                                          parse -> resample -> [random_chunk] -> fbank; CMN applied in extract.py
    examples/voxceleb/v2/local/extract_vox.sh:31   vox2_dev cohort: batch 16 of random 200-frame crops;
                                                   test sets: whole utterances (batch_size 1)
with the same flags, file names and on-disk formats, run as ONE process per GPU:

    python -m torch.distributed.run --nproc-per-node G -m wespeaker_amd.extract \
        --exp_dir exp/X --model_path exp/X/models/avg_model.pt --data_type raw \
        --data_list data/vox1/raw.list --store_dir vox1 --wavs_num 4874 --batch_size 1 --nj 8

What is different by design (MI355X-first, not a translation):
  * whole-utterance mode is still "batch_size 1" in its RESULT (every utterance is embedded on its own
    frames), but utterances of similar length (within 12 %) share one device batch: equal lengths go
    through ws_extract, different ones through ws_extract_ragged, whose rows equal the batch-1 result
    (padding never enters a convolution tap or a statistic) -- the engine sees real batches instead of
    launch-bound single rows; the output order is the list order;
  * file decode runs on host threads ahead of the GPU, PCM goes through rotating pinned buffers and a copy
    stream (H2D of batch i+1 overlaps the forward of batch i), embeddings come back through pinned memory;
  * fbank + CMN + forward are one C-ABI call (ws_extract) on int16 PCM resident in HBM;
  * rank r runs jobs r, r+G, ... of the nj sub-lists; the only collective is the optional all_gather of
    the embeddings (`gather=True`: every rank gets all rows in list order, e.g. for PLDA scoring on rank 0).
The random crops of the cohort mode use a seeded per-utterance generator (the reference draws from the
unseeded global `random` in DataLoader workers, i.e. is not reproducible; the crop length rule is the same).
"""
import json
import os
import time
import subprocess
import tarfile
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import yaml

from . import _lib, parallel
from .audio import load_pcm16_fast
from .kaldi_io import VectorWriter, write_vectors

AUDIO_SUFFIXES = ("flac", "mp3", "m4a", "ogg", "opus", "wav", "wma")     # dataset/processor.py:33


# ----------------------------------------------------------------------------------- list handling
def read_lists(list_file):
    """dataset/lmdb_data or file_utils.read_lists: one entry per non-empty line."""
    with open(list_file, "r", encoding="utf8") as f:
        return [line.rstrip("\n") for line in f if line.strip()]


def split_rule(n_lines, nj):
    """tools/extract_embedding.sh:40-42: `split -l $((data_num / nj + 1))` -> [lo, hi) per job
    (contiguous; the last jobs may be short or empty, exactly like the files split writes)."""
    per = n_lines // nj + 1
    return [(min(n_lines, j * per), min(n_lines, (j + 1) * per)) for j in range(nj)]


def _read_audio_entry(wav):
    """processor.parse_raw.read_audio (:129-136): a path, or a shell command ending in '|'.
    -> (1-D numpy samples of channel 0: int16 for PCM16 files, int16-range float32 otherwise; sample rate)"""
    if wav.endswith("|"):
        return load_pcm16_fast(subprocess.run(wav[:-1], shell=True, stdout=subprocess.PIPE, check=True).stdout)
    return load_pcm16_fast(wav)


def iter_entries(data_type, lines):
    """-> (key, loader) pairs in list order; loader() -> (1-D numpy samples of channel 0, sample_rate).
    raw: json lines {key, wav, spk} (processor.parse_raw); scp: `key path`; shard: tar files of
    key.wav/key.spk members (processor.tar_file_and_group) -- a shard's members are read when reached."""
    if data_type == "raw":
        for line in lines:
            obj = json.loads(line)
            yield obj["key"], (lambda w=obj["wav"]: _read_audio_entry(w))
    elif data_type == "scp":
        for line in lines:
            key, path = line.strip().split(None, 1)
            yield key, (lambda w=path: _read_audio_entry(w))
    elif data_type == "shard":
        for line in lines:
            with tarfile.open(line.strip(), mode="r:*") as tar:
                for info in tar:
                    pos = info.name.rfind(".")
                    if pos <= 0 or info.name[pos + 1:] not in AUDIO_SUFFIXES:
                        continue
                    data = tar.extractfile(info).read()
                    yield info.name[:pos], (lambda d=data: load_pcm16_fast(d))
    else:
        raise NotImplementedError("data_type %r (raw / scp / shard entries are waveforms; 'feat' lists of precomputed "
                                  "Kaldi features go through extract_feats)" % data_type)


def crop_start(key, data_len, chunk_len, seed):
    """Start of the random crop of `key` (processor.get_random_chunk:315-347 draws
    random.randint(0, data_len - chunk_len)); seeded per utterance so that runs repeat."""
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
    return int(rng.integers(0, data_len - chunk_len + 1))


def random_chunk(pcm, key, chunk_len, seed):
    """get_random_chunk: crop if long enough, else tile (`repeat` / np.tile) and cut to chunk_len.
    pcm: 1-D numpy array or torch tensor."""
    n = pcm.shape[0]
    if n >= chunk_len:
        s = crop_start(key, n, chunk_len, seed)
        return pcm[s:s + chunk_len]
    reps = chunk_len // n + 1
    tiled = pcm.repeat(reps) if isinstance(pcm, torch.Tensor) else np.tile(pcm, reps)
    return tiled[:chunk_len]


# --------------------------------------------------------------------------- native wave-file loader
class PathTable:
    """`const char* const*` views of a list of file names for the C-ABI loaders: ONE encode pass, the NUL-terminated
    names in one buffer, the pointers as a numpy array -- `take(idx)` is the table of any sub-list without touching a
    Python string again (round 5 built a ctypes array of c_char_p per batch and one for the probe: 1.4 us per file
    each time, 5 ms of an 80-ms pass over 4 096 files)."""

    def __init__(self, paths):
        import ctypes
        enc = [os.fsencode(p) for p in paths]
        self.n = len(enc)
        blob = b"\0".join(enc) + b"\0"
        self._buf = ctypes.create_string_buffer(blob, len(blob))
        lens = np.fromiter((len(e) + 1 for e in enc), dtype=np.int64, count=self.n)
        offs = np.concatenate(([0], np.cumsum(lens)[:-1])) if self.n else np.zeros(0, np.int64)
        self.ptrs = (offs + ctypes.addressof(self._buf)).astype(np.uint64)

    def take(self, idx):
        """(pointer table of paths[idx] as a contiguous uint64 array, its length); keep the array alive over the call."""
        t = np.ascontiguousarray(self.ptrs[idx])
        return t, int(t.shape[0])


def _path_table(paths):
    return paths if isinstance(paths, PathTable) else PathTable(paths)


def probe_wavs(paths, threads=16):
    """ws_wav_probe: (num_samples int32[n] (-1 = not a PCM16 RIFF file), sample_rate int32[n]) read by C++ threads.
    paths: a list of names or a PathTable."""
    tab = _path_table(paths)
    n = tab.n
    ns, sr = np.empty(n, np.int32), np.empty(n, np.int32)
    if n:
        rc = _lib.lib().ws_wav_probe(_lib.ptr(tab.ptrs), n, int(threads), _lib.ptr(ns), _lib.ptr(sr))
        if rc < 0:
            _lib.check(rc, "ws_wav_probe")
    return ns, sr


def load_wav_rows(paths, dst, counts, starts=None, threads=16, idx=None):
    """ws_wav_load_rows: samples [starts[i], starts[i] + counts[i]) of file i into row i of the int16 host array /
    tensor `dst` (n, stride), decoded by C++ threads (the call releases the GIL).  paths: a list of names, or a
    PathTable with `idx` = the rows of the table this batch holds."""
    table = _path_table(paths)                 # (kept alive over the call: the pointers point into its buffer)
    ptrs, n = table.take(slice(None) if idx is None else idx)
    if not n:
        return
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    st = None if starts is None else np.ascontiguousarray(starts, dtype=np.int32)
    stride = int(dst.stride(0)) if isinstance(dst, torch.Tensor) else int(dst.strides[0] // 2)
    _lib.check(_lib.lib().ws_wav_load_rows(_lib.ptr(ptrs), n, int(threads), _lib.ptr(dst), stride,
                                           _lib.ptr(st) if st is not None else None, _lib.ptr(counts)),
               "ws_wav_load_rows")


# ------------------------------------------------------------------------------ the overlapped engine
class GpuExtractor:
    """(B, N) int16 host batches -> (B, E) float32 host rows through ws_extract, with rotating pinned
    staging buffers, a copy stream for H2D and non-blocking D2H: batch i+1 uploads while batch i computes."""

    def __init__(self, model, frontend, window_type="hamming", depth=2):
        """model: a NativeSpeakerModel, or a SpeakerModelLanes -- then staging slot i belongs to lane i (its own
        engine and HIP stream: upload, forward and download of batch i + 1 are enqueued while batch i computes, and
        the two forwards share the GPU, engine.SpeakerModelLanes)."""
        self.model, self.frontend, self.window_type = model, frontend, window_type
        self.device = model.device
        self.copy_stream = torch.cuda.Stream(device=self.device)
        engines = getattr(model, "engines", None)
        if engines is not None and len(engines) > 1:
            depth = len(engines)
            self._engines, self._streams = list(engines), list(model.streams)
        else:
            eng = engines[0] if engines else model
            self._engines, self._streams = [eng] * depth, [None] * depth
        self.depth = depth
        self._slots = [dict(pin=None, dev=None, out=None, handle=None, free=None) for _ in range(depth)]
        self._turn = 0
        self.embed_dim = model.embed_dim
        # where the submitting thread's time goes (seconds, cumulative): tools/driver_phases.py reads it
        self.timing = {"wait_device_slot_s": 0.0, "submit_s": 0.0, "batches": 0, "wait_decode_s": 0.0,
                       "wait_result_s": 0.0, "probe_plan_s": 0.0}

    supports_ragged = True

    def submit_files(self, paths, counts, starts=None, threads=16):
        """Enqueue one batch of PCM16 files: C++ threads (ws_wav_load_rows) decode samples
        [starts[i], starts[i] + counts[i]) of file i straight into a pinned staging buffer."""
        counts = np.asarray(counts, dtype=np.int32)
        stage = self.stage(len(paths), int(counts.max()) if len(paths) else 0, torch.int16)
        load_wav_rows(paths, stage.view, counts, starts, threads)
        return self.submit_staged(stage, [int(c) for c in counts])

    # -- pinned staging: a ring of STAGES buffers that is independent of the device slots.  A buffer is free again as
    # soon as its upload has left it (an event on the copy stream), not when the forward that consumed the batch has
    # finished -- so a decode thread can fill buffer i + 2 while batch i computes and batch i + 1 uploads.
    STAGES = 4

    class _Stage:
        __slots__ = ("pin", "view", "uploaded")

        def __init__(self):
            self.pin, self.view, self.uploaded = None, None, None

    def stage(self, B, N, dtype=torch.int16):
        """The next pinned staging buffer as a (B, N) view (blocks until its previous upload has completed).
        Safe to call from ONE decode thread while another thread calls submit_staged."""
        if not hasattr(self, "_stages"):
            self._stages, self._stage_turn = [self._Stage() for _ in range(self.STAGES)], 0
        st = self._stages[self._stage_turn % self.STAGES]
        self._stage_turn += 1
        if st.uploaded is not None:
            st.uploaded.synchronize()
            st.uploaded = None
        nbytes = B * N * (2 if dtype == torch.int16 else 4)
        if st.pin is None or st.pin.numel() < nbytes:
            st.pin = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        st.view = st.pin[:nbytes].view(dtype).view(B, N)
        return st

    def submit(self, utts):
        """Enqueue one batch (a list of 1-D waveforms, or a stacked (B, N) tensor); returns a handle whose
        .result() is the (B, E) numpy array.  Different lengths -> one padded ragged batch."""
        if isinstance(utts, torch.Tensor):
            utts = list(utts.numpy())
        utts = [u.numpy() if isinstance(u, torch.Tensor) else u for u in utts]
        lens = [int(u.shape[0]) for u in utts]
        npdt = np.int16 if all(u.dtype == np.int16 for u in utts) else np.float32
        # (8- / 32-bit files arrive as int16-range floats)
        stage = self.stage(len(lens), max(lens), torch.int16 if npdt is np.int16 else torch.float32)
        pin_np = stage.view.numpy()                          # numpy row copies: ~10 us each (torch's: ~60)
        for b, u in enumerate(utts):                         # (padding bytes are never read by the kernels)
            pin_np[b, :lens[b]] = u
        return self.submit_staged(stage, lens)

    def submit_staged(self, stage, lens):
        """Upload a filled staging buffer (copy stream), run the fused extract on the next lane, download the rows."""
        lane = self._turn % self.depth
        slot = self._slots[lane]
        engine = self._engines[lane]
        self._turn += 1
        pin = stage.view
        B, N = pin.shape
        ragged = min(lens) != N
        dtype = pin.dtype
        nbytes = B * N * pin.element_size()
        t_w = time.perf_counter()
        if slot["free"] is not None:
            slot["free"].synchronize()                       # the forward that read this slot's DEVICE buffer has finished
        self.timing["wait_device_slot_s"] += time.perf_counter() - t_w
        main = self._streams[lane] or torch.cuda.current_stream(self.device)
        if slot["dev"] is None or slot["dev"].numel() < nbytes:
            # The device staging buffer is written by the COPY stream: it must come from that stream's
            # allocator pool.  A block taken from the main stream's pool may be memory a still-pending
            # main-stream kernel (e.g. the previous batch's embedding store) is about to write -- legal for
            # same-stream reuse, a race for the copy stream (seen as a corrupted first utterance).
            with torch.cuda.stream(self.copy_stream):
                slot["dev"] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=self.device)
        dev = slot["dev"][:nbytes].view(dtype).view(B, N)
        with torch.cuda.stream(self.copy_stream):
            dev.copy_(pin, non_blocking=True)
            uploaded = torch.cuda.Event()
            uploaded.record(self.copy_stream)
        stage.uploaded = uploaded
        main.wait_event(uploaded)
        dev.record_stream(main)
        # pinned result buffer of the slot (pin_memory() costs ~10 ms per call: never per batch).  The previous
        # batch of this slot may not have been collected yet: materialise it before its buffer is reused.
        if slot["handle"] is not None:
            slot["handle"].materialize()
        if slot["out"] is None or slot["out"].shape[0] < B:
            slot["out"] = torch.empty((max(B, 256), self.embed_dim), dtype=torch.float32).pin_memory()
        out = slot["out"][:B]
        with torch.cuda.stream(main):
            if ragged:
                emb = engine.extract_ragged(self.frontend, dev, lens, window_type=self.window_type)
            else:
                emb = engine.extract(self.frontend, dev, window_type=self.window_type)
            slot["free"] = torch.cuda.Event()
            slot["free"].record(main)
            out.copy_(emb, non_blocking=True)
            done = torch.cuda.Event()
            done.record(main)
        slot["handle"] = _Pending(out, done)
        self.timing["submit_s"] += time.perf_counter() - t_w
        self.timing["batches"] += 1
        return slot["handle"]

    def finish(self):
        for st in self._streams:
            (st or torch.cuda.current_stream(self.device)).synchronize()
        self.model.check_range()


class _Pending:
    def __init__(self, out, event):
        self._out, self._event, self._value = out, event, None

    def materialize(self):
        if self._value is None:
            self._event.synchronize()
            self._value = self._out.numpy().copy()       # (the pinned buffer belongs to the slot)
            self._out = None
        return self._value

    def result(self):
        return self.materialize()


class HostExtractor:
    """Same submit()/finish() protocol around any callable (B, N) tensor -> (B, E) array: the CPU tests use it
    to exercise list handling, bucketing, sharding and file output without a GPU."""

    def __init__(self, fn, embed_dim, ragged_fn=None):
        self.fn, self.embed_dim, self.ragged_fn = fn, embed_dim, ragged_fn
        self.supports_ragged = ragged_fn is not None

    def submit_files(self, paths, counts, starts=None, threads=4):
        counts = np.asarray(counts, dtype=np.int32)
        buf = np.zeros((len(paths), int(counts.max())), dtype=np.int16)
        load_wav_rows(paths, buf, counts, starts, threads)
        return self.submit([buf[b, :int(c)] for b, c in enumerate(counts)])

    def submit(self, utts):
        utts = [torch.from_numpy(np.array(u)) if isinstance(u, np.ndarray) else u for u in utts]
        lens = [int(u.shape[0]) for u in utts]
        if min(lens) == max(lens):
            return _Done(np.asarray(self.fn(torch.stack(list(utts))), dtype=np.float32))
        padded = torch.zeros((len(lens), max(lens)), dtype=utts[0].dtype)
        for b, u in enumerate(utts):
            padded[b, :lens[b]] = u
        return _Done(np.asarray(self.ragged_fn(padded, lens), dtype=np.float32))

    def finish(self):
        pass


class _Done:
    def __init__(self, out):
        self._out = out

    def result(self):
        return self._out


# ------------------------------------------------------------------------------------- one job
def _prefetch(pool, fn, items, depth, chunk=32):
    """fn(item) results in item order, computed on the pool in tasks of `chunk` items with at most `depth`
    tasks ahead of the consumer (Executor.map would submit the whole list -- and hold every decoded waveform
    -- at once; one future per file costs more Python time than decoding a PCM16 file does)."""
    from collections import deque
    import itertools
    q = deque()
    it = iter(items)
    while True:
        group = list(itertools.islice(it, chunk))
        if group:
            q.append(pool.submit(lambda g=group: [fn(x) for x in g]))
        if not group or len(q) >= depth:
            if not q:
                return
            yield from q.popleft().result()


def extract_entries(entries, extractor, batch_size=1, whole_utt=None, chunk_len=32240, max_batch=256,
                    resample_rate=16000, num_workers=4, seed=0, resample_fn=None,
                    max_buffered_samples=64 << 20, length_tolerance=0.12):
    """Embeddings of `entries` ((key, loader) pairs) in list order -> (keys, (n, E) float32).

    whole_utt (default: batch_size == 1, the rule of bin/extract.py:95): every utterance on all of its
    samples; utterances whose lengths lie within length_tolerance of each other (extractors with
    supports_ragged; exact equality otherwise) share a device batch of up to max_batch rows.  Otherwise every
    utterance is cut/tiled to chunk_len samples (random_chunk) and batches hold max(batch_size, ...) rows
    -- the batch size does not change any row's value, so the cohort mode also fills up to max_batch.
    Utterances waiting for a full bucket hold at most max_buffered_samples samples in host memory."""
    if whole_utt is None:
        whole_utt = batch_size == 1
    keys, done = [], []                # done: (indices, (b, E) array) per finished batch
    pending = []                       # (indices, handle)
    buckets = {}                       # (length, dtype) -> (indices, tensors)
    buffered = 0

    def load(item):
        idx, (key, loader) = item
        pcm, sr = loader()
        if isinstance(pcm, torch.Tensor):                  # (C, N) tensors from custom loaders: channel 0
            pcm = (pcm[0] if pcm.dim() == 2 else pcm).numpy()
        if sr != resample_rate:
            if resample_fn is None:
                raise RuntimeError("%s is sampled at %d Hz, expected %d (no resampler given)" % (key, sr, resample_rate))
            pcm = resample_fn(torch.from_numpy(np.ascontiguousarray(pcm)), sr, resample_rate).numpy()
        if not whole_utt:
            pcm = random_chunk(pcm, key, chunk_len, seed)
        return idx, key, pcm

    import math
    ragged_ok = whole_utt and getattr(extractor, "supports_ragged", False) and length_tolerance > 0
    log_step = math.log1p(length_tolerance) if ragged_ok else 1.0

    def bucket_of(n, dtype):
        # geometric length classes: everything in a class is within length_tolerance of the class maximum
        return (int(math.log(max(n, 1)) / log_step) if ragged_ok else n, dtype)

    def flush(key):
        idxs, tensors = buckets.pop(key)
        pending.append((idxs, extractor.submit(tensors)))

    def drain(keep):
        while len(pending) > keep:
            idxs, handle = pending.pop(0)
            done.append((idxs, np.asarray(handle.result(), dtype=np.float32)))

    workers = max(1, num_workers)
    with ThreadPoolExecutor(max_workers=workers) as pool:
        # file decode runs ahead of the GPU on the pool; results arrive in list order
        for idx, key, pcm in _prefetch(pool, load, enumerate(entries), depth=2 * workers + 2):
            keys.append(key)
            n = int(pcm.shape[0])
            key = bucket_of(n, pcm.dtype.str)
            b = buckets.setdefault(key, ([], []))
            b[0].append(idx)
            b[1].append(pcm)
            buffered += n
            if len(b[0]) >= max_batch:
                buffered -= sum(int(t.shape[0]) for t in b[1])
                flush(key)
                drain(keep=2)
            elif buffered > max_buffered_samples:
                for length in sorted(buckets, key=lambda t: t[0]):
                    flush(length)
                    drain(keep=2)
                buffered = 0
        for length in sorted(buckets, key=lambda t: t[0], reverse=True):
            flush(length)
            drain(keep=2)
        drain(keep=0)
    extractor.finish()
    emb = np.zeros((len(keys), extractor.embed_dim), np.float32)
    for idxs, block in done:
        emb[np.asarray(idxs, dtype=np.int64)] = block
    return keys, emb


def plan_batches(counts, max_batch, tolerance):
    """Index batches over `counts` (samples per file): longest first, consecutive files of the SORTED list share a
    batch of at most max_batch rows, cut where the shortest would fall below longest / (1 + tolerance)
    (tolerance 0: equal lengths only).  One numpy search per batch, nothing per file."""
    n = len(counts)
    c64 = counts.astype(np.int64)
    order = np.argsort(-c64, kind="stable")
    neg = -c64[order]                                    # ascending
    batches, b0 = [], 0
    while b0 < n:
        top = -int(neg[b0])
        lim = top / (1.0 + tolerance) if tolerance > 0 else top
        b1 = int(np.searchsorted(neg, -lim, side="right"))          # first index whose count is < lim
        b1 = max(b0 + 1, min(b1, b0 + max_batch))
        batches.append(order[b0:b1])
        b0 = b1
    return batches


def extract_files(keys, paths, extractor, batch_size=1, whole_utt=None, chunk_len=32240, max_batch=256,
                  resample_rate=16000, threads=16, seed=0, length_tolerance=0.12):
    """The same result as extract_entries for a list of wave FILES, with no per-file Python work: C++ threads
    probe every header (ws_wav_probe), the batches are planned on the length array (numpy), and every batch is
    decoded by C++ threads straight into the pinned staging ring (ws_wav_load_rows) while earlier batches upload and
    compute.  Returns None when the list needs the general path (files that are not 16-bit PCM, another sample
    rate, or -- in the random-crop mode -- files shorter than the crop, which are tiled there).

    Batches.  The rows come back by index, so the list may be walked in ANY order: longest first (the pinned / device
    staging buffers and the engine workspace are then sized once), and -- with a ragged extractor -- consecutive files
    of the SORTED list share a batch: 4 096 files of 1.5 .. 2.5 s are 16 batches whose lengths differ by 3 % each (1.6 %
    of padding rows), where round 5's geometric 12 % length classes gave five classes with 6 % padding and a
    part-filled last batch apiece.  A batch is still cut where the spread would pass length_tolerance.

    Start-up.  Encoding, probing and planning 4 096 names costs ~7 ms during which the GPU would idle (a tenth of the
    pass): the first 2 * max_batch files (the HEAD: ~8 ms of GPU work) are probed and planned alone and go to the GPU
    at once; the rest of the list is encoded, probed and planned by the decode thread behind the head's batches."""
    if whole_utt is None:
        whole_utt = batch_size == 1
    n = len(paths)
    if n == 0:
        return [], np.zeros((0, extractor.embed_dim), np.float32)
    t_begin = time.perf_counter()
    ragged_ok = whole_utt and getattr(extractor, "supports_ragged", False) and length_tolerance > 0
    tol = length_tolerance if ragged_ok else 0.0
    staged = hasattr(extractor, "stage") and hasattr(extractor, "submit_staged")
    n_head = 2 * max_batch if (staged and n >= 6 * max_batch) else n

    class _Unfit(Exception):
        pass

    def prepare(lo, hi):
        """(path table, counts, starts, batches of LIST indices) of paths[lo:hi]; _Unfit when the general path is needed."""
        table = PathTable(paths[lo:hi])
        ns, sr = probe_wavs(table, threads)
        if (ns <= 0).any() or (sr != resample_rate).any():
            raise _Unfit()
        starts = None
        if whole_utt:
            counts = ns
        else:
            if (ns < chunk_len).any():
                raise _Unfit()
            counts = np.full(hi - lo, chunk_len, np.int32)
            starts = np.array([crop_start(k, int(m), chunk_len, seed) for k, m in zip(keys[lo:hi], ns)], dtype=np.int32)
        return table, counts, starts, plan_batches(counts, max_batch, tol)

    try:
        head = prepare(0, n_head)
    except _Unfit:
        return None
    emb = np.zeros((n, extractor.embed_dim), np.float32)
    pending = []
    timing = getattr(extractor, "timing", None) or {}

    def drain(keep):
        t_d = time.perf_counter()
        while len(pending) > keep:
            idx, handle = pending.pop(0)
            emb[idx] = handle.result()
        timing["wait_result_s"] = timing.get("wait_result_s", 0.0) + time.perf_counter() - t_d

    timing["probe_plan_s"] = timing.get("probe_plan_s", 0.0) + time.perf_counter() - t_begin
    if not staged:                                       # (HostExtractor and other stand-ins: decode inside submit_files)
        table, counts, starts, batches = head
        for idx in batches:
            pending.append((idx, extractor.submit_files([paths[i] for i in idx], counts[idx],
                                                        None if starts is None else starts[idx], threads)))
            drain(keep=2)
        drain(keep=0)
        extractor.finish()
        return list(keys), emb
    # decode-ahead: ONE Python thread walks the plan and has the C++ pool (ws_wav_load_rows, GIL released) fill the
    # staging ring, up to two batches ahead of the submitting thread
    import queue
    import threading
    ready = queue.Queue(maxsize=max(1, extractor.STAGES - 2))
    failure = []

    def decode_segment(lo, seg):
        table, counts, starts, batches = seg
        for idx in batches:
            c = counts[idx]
            st = extractor.stage(len(idx), int(c.max()), torch.int16)
            load_wav_rows(table, st.view, c, None if starts is None else starts[idx], threads, idx=idx)
            ready.put((idx + lo, st, [int(x) for x in c]))

    def decode_all():
        try:
            decode_segment(0, head)
            if n_head < n:
                decode_segment(n_head, prepare(n_head, n))
        except BaseException as err:  # noqa: BLE001  (re-raised / acted on by the submitting thread)
            failure.append(err)
        finally:
            ready.put(None)

    th = threading.Thread(target=decode_all, name="ws-decode-ahead", daemon=True)
    th.start()
    try:
        while True:
            t_q = time.perf_counter()
            item = ready.get()
            timing["wait_decode_s"] = timing.get("wait_decode_s", 0.0) + time.perf_counter() - t_q
            if item is None:
                break
            idx, st, lens = item
            pending.append((idx, extractor.submit_staged(st, lens)))
            drain(keep=2)
    finally:
        while th.is_alive():                              # (an exception on this side: let the decode thread finish)
            try:
                ready.get(timeout=0.05)
            except queue.Empty:
                pass
        th.join()
    drain(keep=0)
    extractor.finish()
    if failure:
        if isinstance(failure[0], _Unfit):               # a file behind the head needs the general path: start over there
            return None
        raise failure[0]
    return list(keys), emb


def decode_threads(num_workers=0):
    """C++ decode threads of the file path: 2 per requested worker (bin/extract.py's `num_workers`), or -- 0 / None,
    the default -- scaled with the host: an eighth of its cores, between 8 and 32.  tools/probe_threads.py on the
    256-core GPU box, 4 096 two-second files in /dev/shm: 35 / 8.4 / 6.6 / 6.6 / 24 / 89 ms with 1 / 8 / 16 / 32 /
    64 / 128 threads -- past 32 the open() / pread() calls of one directory contend."""
    if num_workers:
        return max(4, 2 * int(num_workers))
    return int(min(32, max(8, (os.cpu_count() or 8) // 8)))


def split_path_list(data_type, lines):
    """(keys, paths) of a raw / scp list whose entries are plain files, else None (pipes, shards)."""
    keys, paths = [], []
    try:
        for line in lines:
            if data_type == "raw":
                obj = json.loads(line)
                key, path = obj["key"], obj["wav"]
            elif data_type == "scp":
                key, path = line.strip().split(None, 1)
            else:
                return None
            if path.endswith("|"):
                return None
            keys.append(key)
            paths.append(path)
    except (ValueError, KeyError):
        return None
    return keys, paths


def extract_feats(lines, extractor, batch_size=1, whole_utt=None, num_frms=200, max_batch=256, num_workers=4, seed=0,
                  cmvn=(True, False), length_tolerance=0.12, load_block=2048):
    """`data_type: feat` lists (dataset/dataset.py:136-273 with processor.parse_feat :171-196): json lines
    {key, feat, spk} whose `feat` is a Kaldi matrix (`file.ark:offset`, what kaldiio.load_mat reads) -> (keys, (n, E)).

    The features come from disk; CMVN (test_conf['cmvn'] / ['cmvn_args'], bin/extract.py:124-127) and the forward run
    on the GPU in one call per batch (ws_forward_ragged_cmvn: statistics over every utterance's own frames).
    whole_utt (default: batch_size == 1, bin/extract.py:95): every utterance on all of its frames, utterances within
    length_tolerance share a padded batch; otherwise each is cut / tiled to num_frms frames
    (processor.random_chunk(chunk_len=num_frms, 'feat')).  Matrices are read `load_block` utterances at a time by a
    thread pool, so host memory holds one block."""
    from .kaldi_io import read_mat
    if whole_utt is None:
        whole_utt = batch_size == 1
    model = extractor.model
    engine = getattr(model, "engines", [model])[0]
    entries = []
    for line in lines:
        obj = json.loads(line)
        entries.append((obj["key"], obj["feat"]))
    keys = [k for k, _ in entries]
    emb = np.zeros((len(entries), extractor.embed_dim), np.float32)

    def load(item):
        key, spec = item
        m = read_mat(spec)
        if not whole_utt:
            m = random_chunk(m, key, num_frms, seed) if m.shape[0] >= num_frms else \
                np.tile(m, (num_frms // m.shape[0] + 1, 1))[:num_frms]
        return np.ascontiguousarray(m, dtype=np.float32)

    with ThreadPoolExecutor(max_workers=max(1, num_workers)) as pool:
        for lo in range(0, len(entries), load_block):
            mats = list(pool.map(load, entries[lo:lo + load_block]))
            frames = np.array([m.shape[0] for m in mats], dtype=np.int32)
            for idx in plan_batches(frames, max_batch, length_tolerance):
                T = int(frames[idx].max())
                pad = torch.zeros((len(idx), T, mats[idx[0]].shape[1]), dtype=torch.float32)
                for r, i in enumerate(idx):
                    pad[r, :frames[i]] = torch.from_numpy(mats[i])
                out = engine.embed_ragged(pad, frames[idx], cmvn=cmvn)
                emb[lo + idx] = out.cpu().numpy()
    extractor.finish()
    return keys, emb


def extract_list(data_type, lines, extractor, **kw):
    """extract_files when the list allows it, extract_entries otherwise (same result either way); `feat` lists of
    precomputed Kaldi features go through extract_feats."""
    if data_type == "feat":
        fk = {k: v for k, v in kw.items() if k in ("batch_size", "whole_utt", "num_frms", "max_batch", "num_workers",
                                                     "seed", "cmvn", "length_tolerance")}
        return extract_feats(lines, extractor, **fk)
    kw = {k: v for k, v in kw.items() if k not in ("num_frms", "cmvn")}         # (the feat path's own arguments)
    kp = split_path_list(data_type, lines)
    if kp is not None and hasattr(extractor, "submit_files"):
        fast = {k: v for k, v in kw.items() if k in ("batch_size", "whole_utt", "chunk_len", "max_batch",
                                                       "resample_rate", "seed", "length_tolerance")}
        out = extract_files(kp[0], kp[1], extractor, threads=decode_threads(kw.get("num_workers", 0)), **fast)
        if out is not None:
            return out
    return extract_entries(iter_entries(data_type, lines), extractor, **kw)


def write_ark_scp(keys, emb, embed_ark):
    """bin/extract.py:105-111,137-139: `ark,scp:<embed_ark>,<embed_ark[:-3]>scp`, absolute ark path."""
    d = os.path.dirname(embed_ark)
    if d:
        os.makedirs(d, exist_ok=True)
    embed_ark = os.path.abspath(embed_ark)
    embed_scp = embed_ark[:-3] + "scp"
    if isinstance(emb, np.ndarray) and emb.ndim == 2:
        write_vectors(list(keys), emb, embed_ark, embed_scp)     # one pass over the whole table
    else:
        with VectorWriter(embed_ark, embed_scp) as w:
            for k, e in zip(keys, emb):
                w(k, e)
    return embed_scp


# ----------------------------------------------------------------------------- config + model
def load_config(config, **overrides):
    """utils.parse_config_or_kwargs (utils/utils.py:37-51): yaml keys, overridden by keyword arguments."""
    with open(config) as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader)
    return dict(cfg, **overrides)


def check_frontend_config(configs):
    """The test-time dataset settings this driver implements; anything else is refused loudly."""
    ds = dict(configs.get("dataset_args") or {})
    if ds.get("frontend", "fbank") != "fbank":
        raise NotImplementedError("frontend %r: only 'fbank' is on the MI355X hot path" % ds.get("frontend"))
    fb = dict(ds.get("fbank_args") or {})
    if fb.get("frame_length", 25) != 25 or fb.get("frame_shift", 10) != 10:
        raise NotImplementedError("fbank frame_length/frame_shift other than 25/10 ms")
    # test_conf.get('cmvn', True) / apply_cmvn(features, **test_conf.get('cmvn_args', {})): bin/extract.py:124-127,
    # dataset_utils.py:19-26 (defaults norm_mean=True, norm_var=False); `cmvn: False` = neither
    cm = dict(ds.get("cmvn_args") or {})
    unknown = set(cm) - {"norm_mean", "norm_var"}
    if unknown:
        raise TypeError("apply_cmvn() got an unexpected keyword argument %r" % sorted(unknown)[0])
    on = bool(ds.get("cmvn", True))
    return {"resample_rate": int(ds.get("resample_rate", 16000)),
            "num_mel_bins": int(fb.get("num_mel_bins", 80)),
            "num_frms": int(ds.get("num_frms", 200)),
            "norm_mean": on and bool(cm.get("norm_mean", True)),
            "norm_var": on and bool(cm.get("norm_var", False))}


def chunk_samples(num_frms, resample_rate, frame_shift=10, frame_length=25):
    """dataset.py:237-241: ((num_frms - 1) * frame_shift + frame_length) * resample_rate // 1000."""
    return ((num_frms - 1) * frame_shift + frame_length) * resample_rate // 1000


def build_gpu_extractor(configs, model_path, device=None, max_batch=256, max_frames=400, precision="fp32", lanes=0):
    """get_speaker_model(...)(**model_args) + load_checkpoint (bin/extract.py:66-77) on the native engine.
    lanes: batches in flight on the GPU (engine.SpeakerModelLanes); 0 = two (every back-end: the fbank defect that kept
    the binary16 ones on one stream in round 3 is closed)."""
    from .engine import Frontend, NativeSpeakerModel, SpeakerModelLanes
    from .speaker import _load_state_dict
    fc = check_frontend_config(configs)
    margs = dict(configs.get("model_args") or {})
    margs.pop("pooling_func", None)
    feat_dim = int(margs.pop("feat_dim", fc["num_mel_bins"]))
    embed_dim = margs.pop("embed_dim", None)
    sd = _load_state_dict(model_path)
    if lanes <= 0:
        lanes = 2
    if lanes > 1:
        model = SpeakerModelLanes(configs["model"], sd, lanes=lanes, feat_dim=feat_dim, embed_dim=embed_dim,
                                  device=device, max_batch=max_batch, max_frames=max_frames)
    else:
        model = NativeSpeakerModel(configs["model"], sd, feat_dim=feat_dim, embed_dim=embed_dim, device=device,
                                   max_batch=max_batch, max_frames=max_frames)
    model.set_precision(precision)
    fe = Frontend(fc["resample_rate"], feat_dim, device=model.device)
    fe.set_cmvn(fc["norm_mean"], fc["norm_var"])
    return GpuExtractor(model, fe), fc


def _gpu_resampler(device):
    from .audio import resample

    def fn(pcm, sr, target):
        return resample(pcm.to(torch.float), sr, target, device).cpu()
    return fn


# ------------------------------------------------------------------------------------ entry points
def extract(config="conf/config.yaml", **kwargs):
    """ONE job = wespeaker/bin/extract.py: flags model_path, data_type, data_list, embed_ark, batch_size,
    num_workers (+ precision, max_batch extensions).  Writes embed_ark and its .scp; returns (keys, emb)."""
    configs = load_config(config, **kwargs)
    batch_size = int(configs.get("batch_size", 1))
    extractor, fc = build_gpu_extractor(configs, configs["model_path"], device=configs.get("device"),
                                        max_batch=int(configs.get("max_batch", 256)),
                                        precision=configs.get("precision", "fp32"))
    lines = read_lists(configs["data_list"])
    keys, emb = extract_list(
        configs["data_type"], lines, extractor, batch_size=batch_size,
        chunk_len=chunk_samples(fc["num_frms"], fc["resample_rate"]),
        max_batch=int(configs.get("max_batch", 256)), resample_rate=fc["resample_rate"],
        num_workers=int(configs.get("num_workers", 4)), seed=int(configs.get("seed", 0)),
        resample_fn=_gpu_resampler(extractor.device), num_frms=fc["num_frms"],
        cmvn=(fc["norm_mean"], fc["norm_var"]))
    write_ark_scp(keys, emb, configs["embed_ark"])
    return keys, emb


def run_jobs(lines, data_type, embed_dir, make_extractor, nj, rank=0, world=1, batch_size=1, chunk_len=32240,
             max_batch=256, resample_rate=16000, num_workers=4, seed=0, resample_fn=None, gather=False,
             wavs_num=None, store_dir="", num_frms=200, cmvn=(True, False)):
    """tools/extract_embedding.sh on `world` ranks: job j of the nj contiguous sub-lists runs on rank
    j % world and writes embed_dir/xvector_{j:03d}.ark/.scp (+ log/split_{j:03d}); after a barrier rank 0
    concatenates the scp files in job order into xvector.scp, compares the count with wavs_num and writes
    extract.result with the reference's two messages.  gather=True additionally returns, on every rank,
    (keys, (N, E) embeddings) of the whole list in list order through ONE all_gather of padded blocks."""
    import torch.distributed as dist
    log_dir = os.path.join(embed_dir, "log")
    os.makedirs(log_dir, exist_ok=True)
    spans = split_rule(len(lines), nj)
    extractor = None
    my_keys, my_emb = [], []
    for j, (lo, hi) in enumerate(spans):
        if j % world != rank:
            continue
        with open(os.path.join(log_dir, "split_%03d" % j), "w") as f:
            f.write("".join(l + "\n" for l in lines[lo:hi]))
        if hi <= lo:
            continue                                # `split` writes no file for an empty tail job
        if extractor is None:
            extractor = make_extractor()
        keys, emb = extract_list(data_type, lines[lo:hi], extractor, batch_size=batch_size,
                                 chunk_len=chunk_len, max_batch=max_batch, resample_rate=resample_rate,
                                 num_workers=num_workers, seed=seed, resample_fn=resample_fn, num_frms=num_frms,
                                 cmvn=cmvn)
        write_ark_scp(keys, emb, os.path.join(embed_dir, "xvector_%03d.ark" % j))
        my_keys.append((j, keys))
        my_emb.append((j, emb))
    multi = parallel.collectives_active()          # several ranks, or a forced one-rank group (WS_DIST_FORCE_GROUP)
    if multi:
        if dist.get_backend() == "nccl":           # RCCL's barrier is a device collective: name the device
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
    result = None
    if rank == 0:
        merged = os.path.join(embed_dir, "xvector.scp")
        count = 0
        with open(merged, "w") as out:
            for j in range(nj):
                p = os.path.join(embed_dir, "xvector_%03d.scp" % j)
                if os.path.exists(p):
                    with open(p) as f:
                        for line in f:
                            out.write(line)
                            count += 1
        ok = wavs_num is None or count == int(wavs_num)
        result = ("Successfully extract embedding for %s" if ok else "Failed to extract embedding for %s") % store_dir
        with open(os.path.join(embed_dir, "extract.result"), "w") as f:
            f.write(result + "\n")
        print(result)
    if not gather:
        return result
    # ---- all ranks: the whole list in list order (job j sits at rows [lo_j, hi_j) for raw/scp lists)
    E = extractor.embed_dim if extractor is not None else 0
    # (an nccl = RCCL process group has no CPU path: every tensor a collective touches lives on this rank's GPU)
    # (WS_COLLECTIVES_ON_GPU=1: the same placement under gloo -- how the single-GPU test box runs this branch)
    use_cuda = multi and (dist.get_backend() == "nccl" or os.environ.get("WS_COLLECTIVES_ON_GPU") == "1")
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    if multi:
        e_t = torch.tensor([E], dtype=torch.int64, device=dev)
        dist.all_reduce(e_t, op=dist.ReduceOp.MAX)
        E = int(e_t.item())
    local = np.concatenate([e for _, e in my_emb]) if my_emb else np.zeros((0, E), np.float32)
    local_keys = [k for _, ks in my_keys for k in ks]
    if not multi:
        return local_keys, local
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[rank] = local.shape[0]
    dist.all_reduce(counts)
    per = int(counts.max().item())
    block = torch.zeros((per, E), dtype=torch.float32, device=dev)
    block[:local.shape[0]] = torch.from_numpy(local).to(dev)
    full = parallel.gather_rows(block, per * world).cpu().numpy()
    key_lists = [None] * world
    dist.all_gather_object(key_lists, [(j, ks) for j, ks in my_keys])
    # rank r's block holds its jobs in increasing j; re-assemble in job order
    pieces = {}
    for r in range(world):
        off = 0
        for j, ks in key_lists[r]:
            pieces[j] = (ks, full[r * per + off: r * per + off + len(ks)])
            off += len(ks)
    keys = [k for j in sorted(pieces) for k in pieces[j][0]]
    emb = np.concatenate([pieces[j][1] for j in sorted(pieces)]) if pieces else np.zeros((0, E), np.float32)
    return keys, emb


def main(argv=None):
    """Flags of tools/extract_embedding.sh (+ --config, --precision, --max_batch)."""
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--exp_dir", default="exp/XVEC")
    ap.add_argument("--config", default=None, help="default: <exp_dir>/config.yaml")
    ap.add_argument("--model_path", default="avg_model.pt")
    ap.add_argument("--data_type", default="shard", choices=["shard", "raw", "scp", "feat"])
    ap.add_argument("--data_list", default="shard.list")
    ap.add_argument("--wavs_num", type=int, default=None)
    ap.add_argument("--store_dir", default="")
    ap.add_argument("--batch_size", "--batch-size", type=int, default=1)
    ap.add_argument("--num_workers", "--num-workers", type=int, default=4)
    ap.add_argument("--nj", type=int, default=0, help="number of sub-lists (default: the number of ranks)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "f16x3", "f16"])
    ap.add_argument("--max_batch", type=int, default=256)
    ap.add_argument("--lanes", type=int, default=0,
                    help="batches in flight on the GPU (one engine + stream each); 0 = 2")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpus", default=None, help="accepted for compatibility; ranks map to LOCAL_RANK")
    ap.add_argument("--gather_npz", default=None,
                    help="also all_gather the embeddings (every rank gets the whole list in list order); rank 0 "
                         "writes them here as keys / emb arrays")
    args = ap.parse_args(argv)
    rank, world, local_rank = parallel.init_distributed()
    # WS_SHARE_GPU=1 (debug / single-GPU boxes, with WS_DIST_BACKEND=gloo): every rank uses GPU 0
    device = torch.device("cuda", 0 if os.environ.get("WS_SHARE_GPU") else local_rank)
    torch.cuda.set_device(device)
    configs = load_config(args.config or os.path.join(args.exp_dir, "config.yaml"))
    fc = check_frontend_config(configs)
    embed_dir = os.path.join(args.exp_dir, "embeddings", args.store_dir)

    def make():
        ex, _ = build_gpu_extractor(configs, args.model_path, device=device, max_batch=args.max_batch,
                                    precision=args.precision, lanes=args.lanes)
        return ex

    out = run_jobs(read_lists(args.data_list), args.data_type, embed_dir, make, args.nj or world, rank, world,
                   batch_size=args.batch_size, chunk_len=chunk_samples(fc["num_frms"], fc["resample_rate"]),
                   max_batch=args.max_batch, resample_rate=fc["resample_rate"], num_workers=args.num_workers,
                   seed=args.seed, resample_fn=_gpu_resampler(device), wavs_num=args.wavs_num,
                   store_dir=args.store_dir, gather=bool(args.gather_npz), num_frms=fc["num_frms"],
                   cmvn=(fc["norm_mean"], fc["norm_var"]))
    if args.gather_npz and rank == 0:
        keys, emb = out
        np.savez(args.gather_npz, keys=np.asarray(keys), emb=emb)
    if parallel.collectives_active():
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
