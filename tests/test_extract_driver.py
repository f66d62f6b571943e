"""CPU: the batch-extraction driver (wespeaker_amd/extract.py = tools/extract_embedding.sh +
wespeaker/bin/extract.py) with a host stand-in for the GPU call: list formats, the `split -l` rule,
length bucketing with list-order output, the random-crop cohort mode, per-job ark/scp + merged scp +
extract.result, and the 2-rank gloo run against the single-rank run, bit for bit."""
import io
import json
import os
import tarfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fixtures import synth
from wespeaker_amd import extract as wx
from wespeaker_amd import kaldi_io, parallel


def _fake_rows(batch):
    """Row-wise deterministic 'embedding' (independent of the batch a row travels in)."""
    x = batch.to(torch.float64)
    return torch.stack([x.mean(1), x.std(1), x[:, 0], x[:, -1],
                        torch.full((x.shape[0],), float(x.shape[1]), dtype=torch.float64),
                        x.abs().sum(1) % 977.0], 1).to(torch.float32).numpy()


def _make_corpus(d, n=23):
    lengths = [16000 + 1600 * (i % 4) for i in range(n)]          # 4 distinct lengths, interleaved
    lines_raw, lines_scp = [], []
    for i, L in enumerate(lengths):
        p = os.path.join(d, "u%02d.wav" % i)
        synth.write_wav(p, synth.synth_wav(300 + i, L))
        lines_raw.append(json.dumps({"key": "utt%02d" % i, "wav": p, "spk": "s%d" % (i % 3)}))
        lines_scp.append("utt%02d %s" % (i, p))
    return lengths, lines_raw, lines_scp


def test_split_rule_is_the_scripts_split_minus_l():
    # tools/extract_embedding.sh:40-42: subfile_num = data_num / nj + 1 (integer division), split -l
    assert wx.split_rule(100, 4) == [(0, 26), (26, 52), (52, 78), (78, 100)]
    assert wx.split_rule(8, 4) == [(0, 3), (3, 6), (6, 8), (8, 8)]       # last job gets no file
    for n, nj in [(4874, 8), (7, 8), (1, 1), (0, 2), (1000, 16)]:
        spans = wx.split_rule(n, nj)
        assert [i for lo, hi in spans for i in range(lo, hi)] == list(range(n))
    assert wx.chunk_samples(200, 16000) == 32240                          # dataset.py:237-241


def test_whole_utterance_mode_buckets_by_length_and_keeps_list_order(tmp_path):
    lengths, lines_raw, lines_scp = _make_corpus(str(tmp_path))
    seen = []

    def fn(batch):
        seen.append(tuple(batch.shape))
        return _fake_rows(batch)

    ex = wx.HostExtractor(fn, 6)
    keys, emb = wx.extract_entries(wx.iter_entries("raw", lines_raw), ex, batch_size=1, max_batch=4, num_workers=3)
    assert keys == ["utt%02d" % i for i in range(23)]
    ref = np.concatenate([_fake_rows(torch.from_numpy(synth.synth_wav(300 + i, L))[None]) for i, L in enumerate(lengths)])
    assert np.array_equal(emb, ref)
    assert all(b <= 4 for b, _ in seen) and max(b for b, _ in seen) == 4           # real batches
    assert {n for _, n in seen} == set(lengths)                                     # never mixed lengths
    keys2, emb2 = wx.extract_entries(wx.iter_entries("scp", lines_scp), ex, batch_size=1, max_batch=64)
    assert keys2 == keys and np.array_equal(emb2, emb)
    # bounded host buffering: a tiny budget forces early flushes, same result
    keys3, emb3 = wx.extract_entries(wx.iter_entries("raw", lines_raw), ex, max_batch=64, max_buffered_samples=40000)
    assert keys3 == keys and np.array_equal(emb3, emb)
    assert wx.extract_entries(iter(()), ex)[1].shape == (0, 6)


def test_shard_lists_and_pipe_commands(tmp_path):
    lengths, lines_raw, _ = _make_corpus(str(tmp_path), n=6)
    shard = str(tmp_path / "shards_000.tar")
    with tarfile.open(shard, "w") as tar:
        for i in range(6):
            for suffix, data in (("wav", open(str(tmp_path / ("u%02d.wav" % i)), "rb").read()),
                                 ("spk", ("s%d" % i).encode())):
                info = tarfile.TarInfo("utt%02d.%s" % (i, suffix))
                info.size = len(data)
                tar.addfile(info, io.BytesIO(data))
    ex = wx.HostExtractor(_fake_rows, 6)
    keys_s, emb_s = wx.extract_entries(wx.iter_entries("shard", [shard]), ex, max_batch=8)
    keys_r, emb_r = wx.extract_entries(wx.iter_entries("raw", lines_raw), ex, max_batch=8)
    assert keys_s == keys_r and np.array_equal(emb_s, emb_r)
    piped = [json.dumps({"key": "p0", "wav": "cat %s |" % (tmp_path / "u00.wav"), "spk": "x"})]
    _, emb_p = wx.extract_entries(wx.iter_entries("raw", piped), ex)
    assert np.array_equal(emb_p[0], emb_r[0])
    with pytest.raises(NotImplementedError):
        list(wx.iter_entries("feat", ["x"]))


def test_random_chunk_cohort_mode(tmp_path):
    """bin/extract.py:95 + dataset.py:235-242: batch_size > 1 -> every utterance becomes ONE random crop of
    ((num_frms-1)*10+25) ms, tiled first when shorter (processor.get_random_chunk)."""
    paths = []
    for i, L in enumerate([48000, 20000, 32240, 9000]):
        p = str(tmp_path / ("c%d.wav" % i))
        synth.write_wav(p, synth.synth_wav(400 + i, L))
        paths.append("c%d %s" % (i, p))
    got = []
    ex = wx.HostExtractor(lambda b: (got.append(b.clone()), _fake_rows(b))[1], 6)
    keys, emb = wx.extract_entries(wx.iter_entries("scp", paths), ex, batch_size=16, chunk_len=32240, seed=5)
    assert len(got) == 1 and got[0].shape == (4, 32240)                   # one equal-length batch
    w0 = torch.from_numpy(synth.synth_wav(400, 48000))
    s0 = wx.crop_start("c0", 48000, 32240, 5)
    assert 0 <= s0 <= 48000 - 32240 and torch.equal(got[0][0], w0[s0:s0 + 32240])
    w1 = torch.from_numpy(synth.synth_wav(401, 20000))
    assert torch.equal(got[0][1], torch.cat([w1, w1])[:32240])           # tiled, then cut
    assert torch.equal(got[0][2], torch.from_numpy(synth.synth_wav(402, 32240)))   # exact length: start 0
    w3 = torch.from_numpy(synth.synth_wav(403, 9000))
    assert torch.equal(got[0][3], w3.repeat(4)[:32240])
    keys_b, emb_b = wx.extract_entries(wx.iter_entries("scp", paths), ex, batch_size=16, chunk_len=32240, seed=5)
    assert np.array_equal(emb, emb_b)                                     # seeded: runs repeat
    _, emb_c = wx.extract_entries(wx.iter_entries("scp", paths), ex, batch_size=16, chunk_len=32240, seed=6)
    assert not np.array_equal(emb[0], emb_c[0])


def test_frontend_config_is_checked_loudly():
    ok = {"model": "ECAPA_TDNN_GLOB_c512", "dataset_args": {"resample_rate": 16000, "fbank_args":
          {"num_mel_bins": 80, "frame_shift": 10, "frame_length": 25, "dither": 1.0}}}
    assert wx.check_frontend_config(ok) == {"resample_rate": 16000, "num_mel_bins": 80, "num_frms": 200,
                                            "norm_mean": True, "norm_var": False}
    for bad in ({"frontend": "s3prl"}, {"fbank_args": {"frame_shift": 20}}):
        with pytest.raises(NotImplementedError):
            wx.check_frontend_config({"dataset_args": bad})
    # test_conf['cmvn'] / ['cmvn_args'] (bin/extract.py:124-127 -> apply_cmvn, dataset_utils.py:19-26)
    for ds, want in (({"cmvn": False}, (False, False)), ({"cmvn_args": {"norm_var": True}}, (True, True)),
                     ({"cmvn_args": {"norm_mean": False, "norm_var": True}}, (False, True)),
                     ({"cmvn": False, "cmvn_args": {"norm_var": True}}, (False, False)),
                     ({"cmvn_args": {"norm_mean": False}}, (False, False))):
        fc = wx.check_frontend_config({"dataset_args": ds})
        assert (fc["norm_mean"], fc["norm_var"]) == want, ds
    with pytest.raises(TypeError):                       # apply_cmvn(**{'norm_std': ...}) is a TypeError there too
        wx.check_frontend_config({"dataset_args": {"cmvn_args": {"norm_std": True}}})


def _job_worker(rank, world, port, embed_dir, lines, q):
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                          WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        parallel.init_distributed(backend="gloo")
    out = wx.run_jobs(lines, "raw", embed_dir, lambda: wx.HostExtractor(_fake_rows, 6), nj=4, rank=rank,
                      world=world, max_batch=4, gather=True, wavs_num=len(lines), store_dir="vox1")
    q.put((rank, out[0], out[1]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_run_writes_the_same_files_as_one_rank(tmp_path):
    """extract_embedding.sh semantics on 2 ranks (gloo): per-job ark/scp, merged xvector.scp in job order,
    extract.result -- identical bytes to the single-rank run; gather=True hands every rank all rows."""
    _, lines_raw, _ = _make_corpus(str(tmp_path), n=23)
    ctx = mp.get_context("spawn")
    results = {}
    for world, tag in ((1, "one"), (2, "two")):
        d = str(tmp_path / tag)
        q = ctx.Queue()
        port = 29950 + os.getpid() % 40
        procs = [ctx.Process(target=_job_worker, args=(r, world, port, d, lines_raw, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=180) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        results[tag] = res
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    for j in range(4):
        a = open(os.path.join(one, "xvector_%03d.ark" % j), "rb").read()
        b = open(os.path.join(two, "xvector_%03d.ark" % j), "rb").read()
        assert a == b and len(a) > 0
        assert open(os.path.join(one, "log", "split_%03d" % j)).read() == open(os.path.join(two, "log", "split_%03d" % j)).read()
    all_ark = b"".join(open(os.path.join(two, "xvector_%03d.ark" % j), "rb").read() for j in range(4))
    single = kaldi_io.read_vec_scp(os.path.join(one, "xvector.scp"))
    merged = kaldi_io.read_vec_scp(os.path.join(two, "xvector.scp"))
    assert list(merged) == list(single) == ["utt%02d" % i for i in range(23)]
    assert all(np.array_equal(merged[k], single[k]) for k in single)
    assert len(all_ark) == sum(os.path.getsize(os.path.join(one, "xvector_%03d.ark" % j)) for j in range(4))
    assert open(os.path.join(two, "extract.result")).read() == "Successfully extract embedding for vox1\n"
    keys1, emb1 = results["one"][0][1], results["one"][0][2]
    for rank, keys, emb in results["two"]:
        assert keys == keys1 and np.array_equal(emb, emb1), rank
    assert np.array_equal(emb1, np.stack(list(single.values())))


def test_count_mismatch_is_reported_like_the_script(tmp_path):
    _, lines_raw, _ = _make_corpus(str(tmp_path), n=5)
    msg = wx.run_jobs(lines_raw, "raw", str(tmp_path / "o"), lambda: wx.HostExtractor(_fake_rows, 6), nj=2,
                      wavs_num=6, store_dir="vox1")
    assert msg == "Failed to extract embedding for vox1"


def test_ragged_capable_extractors_get_length_classes(tmp_path):
    """Extractors that take padded batches (GpuExtractor: ws_extract_ragged) receive utterances whose
    lengths lie within the tolerance of each other; exact-length extractors never see mixed lengths."""
    lines = []
    lengths = [16000 + 331 * i for i in range(20)]                  # 16000 .. 22289: all different
    for i, L in enumerate(lengths):
        p = str(tmp_path / ("g%02d.wav" % i))
        synth.write_wav(p, synth.synth_wav(600 + i, L))
        lines.append("utt%02d %s" % (i, p))
    seen = []

    def ragged_fn(padded, lens):
        seen.append(list(lens))
        return np.concatenate([_fake_rows(padded[b:b + 1, :n]) for b, n in enumerate(lens)])

    ex = wx.HostExtractor(_fake_rows, 6, ragged_fn=ragged_fn)
    keys, emb = wx.extract_entries(wx.iter_entries("scp", lines), ex, batch_size=1, max_batch=8)
    ref = np.concatenate([_fake_rows(torch.from_numpy(synth.synth_wav(600 + i, L))[None]) for i, L in enumerate(lengths)])
    assert keys == ["utt%02d" % i for i in range(20)] and np.array_equal(emb, ref)
    assert seen and all(max(c) <= min(c) * 1.12 + 1 for c in seen) and sum(len(c) for c in seen) >= 15
    ex2 = wx.HostExtractor(_fake_rows, 6)                             # no ragged support: exact lengths only
    keys2, emb2 = wx.extract_entries(wx.iter_entries("scp", lines), ex2, batch_size=1, max_batch=8)
    assert np.array_equal(emb2, ref)


def _write_riff(path, pcm, rate=16000, channels=1, bits=16, extra_chunk=False):
    import struct
    data = np.ascontiguousarray(pcm).tobytes()
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * channels * bits // 8,
                                           channels * bits // 8, bits)
    if extra_chunk:
        body += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"          # odd size + pad byte
    body += b"data" + struct.pack("<I", len(data)) + data
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_native_wav_loader_rows_and_probe(tmp_path):
    """ws_wav_probe / ws_wav_load_rows (C++ decode threads, host only): canonical files, a LIST chunk in front of
    the data, stereo (channel 0 kept), sub-ranges; unreadable / non-PCM16 files are reported, not guessed."""
    a = synth.synth_wav(1, 5000)
    b = synth.synth_wav(2, 7000)
    st = np.stack([synth.synth_wav(3, 3000), synth.synth_wav(4, 3000)], 1)       # interleaved stereo
    pa, pb, pc, pd, pe = (str(tmp_path / n) for n in ("a.wav", "b.wav", "c.wav", "d.wav", "e.wav"))
    _write_riff(pa, a)
    _write_riff(pb, b, extra_chunk=True)
    _write_riff(pc, st, channels=2)
    _write_riff(pd, (a[:100] >> 8).astype(np.int8), bits=8)                     # 8-bit: not on this path
    open(pe, "wb").write(b"not a wave file")
    ns, sr = wx.probe_wavs([pa, pb, pc, pd, pe, str(tmp_path / "absent.wav")], threads=3)
    assert list(ns) == [5000, 7000, 3000, -1, -1, -1] and list(sr[:3]) == [16000] * 3
    buf = np.full((3, 7000), 77, np.int16)
    wx.load_wav_rows([pa, pb, pc], buf, [5000, 7000, 3000], threads=2)
    assert np.array_equal(buf[0, :5000], a) and (buf[0, 5000:] == 77).all()
    assert np.array_equal(buf[1], b) and np.array_equal(buf[2, :3000], st[:, 0])
    wx.load_wav_rows([pb, pa], buf[:2], [100, 200], starts=[6900, 4800], threads=1)
    assert np.array_equal(buf[0, :100], b[6900:]) and np.array_equal(buf[1, :200], a[4800:])
    from wespeaker_amd._lib import NativeError
    with pytest.raises(NativeError, match="a.wav"):
        wx.load_wav_rows([pb, pa], buf[:2], [100, 5001], threads=2)             # more than the file holds
    with pytest.raises(NativeError):
        wx.load_wav_rows([pd], buf[:1], [10])
    # the Python fast reader agrees (incl. the chunk walk)
    from wespeaker_amd.audio import load_pcm16_fast
    assert np.array_equal(load_pcm16_fast(pb)[0], b) and np.array_equal(load_pcm16_fast(pc)[0], st[:, 0])
    x8, r8 = load_pcm16_fast(pd)
    assert r8 == 16000 and x8.dtype == np.float32 and x8.shape == (100,)          # falls back to load_wav


def test_native_wav_loader_survives_damaged_files(tmp_path):
    """Truncations at every header byte, random byte flips in the header and absurd chunk sizes: the C++ chunk walk
    either reports the file as unusable (-1) or returns a sample count the file really holds -- it never reads past
    what it was given and never crashes the process."""
    good = str(tmp_path / "good.wav")
    _write_riff(good, synth.synth_wav(9, 2000), extra_chunk=True)
    raw = open(good, "rb").read()
    hdr = len(raw) - 4000                       # bytes in front of the samples
    paths, limits = [], []
    for cut in list(range(0, hdr + 3)) + [hdr + 101, len(raw) - 1]:           # cut anywhere in / just behind the header
        q = str(tmp_path / ("cut%d.wav" % cut))
        open(q, "wb").write(raw[:cut])
        paths.append(q); limits.append(max(0, cut - hdr) // 2)
    rng = np.random.RandomState(3)
    for k in range(200):                                                      # flipped header bytes
        b = bytearray(raw)
        for _ in range(rng.randint(1, 4)):
            b[rng.randint(0, hdr)] = rng.randint(0, 256)
        q = str(tmp_path / ("flip%d.wav" % k))
        open(q, "wb").write(bytes(b))
        paths.append(q); limits.append(len(raw) // 2)
    for size in (0xffffffff, 0x7fffffff, 0x80000001, 3999, 4001):            # lying data-chunk sizes
        b = bytearray(raw)
        b[hdr - 4:hdr] = int(size).to_bytes(4, "little")
        q = str(tmp_path / ("size%x.wav" % size))
        open(q, "wb").write(bytes(b))
        paths.append(q); limits.append(2000)
    ns, _ = wx.probe_wavs(paths, threads=4)
    for n, lim, q in zip(ns, limits, paths):
        assert n == -1 or 0 <= n <= lim, (q, n, lim)
    usable = [(q, int(n)) for q, n in zip(paths, ns) if n > 0]
    assert usable                                                               # e.g. the lying sizes clamp to the file
    buf = np.zeros((len(usable), 2000), np.int16)
    wx.load_wav_rows([q for q, _ in usable], buf, [n for _, n in usable], threads=4)


def test_file_fast_path_equals_the_general_path(tmp_path):
    """extract_files (native loader, batches planned on the probed lengths) == extract_entries, row for row, in
    both modes; lists it cannot take (pipes, shards, other formats / rates, crops of short files) fall back."""
    lengths, lines_raw, lines_scp = _make_corpus(str(tmp_path), n=21)
    ragged = lambda padded, lens: np.concatenate([_fake_rows(padded[b:b + 1, :n]) for b, n in enumerate(lens)])  # noqa: E731
    ex = wx.HostExtractor(_fake_rows, 6, ragged_fn=ragged)
    for kw in (dict(batch_size=1), dict(batch_size=16, chunk_len=12000, seed=4)):
        k0, e0 = wx.extract_entries(wx.iter_entries("raw", lines_raw), ex, max_batch=5, **kw)
        kp = wx.split_path_list("raw", lines_raw)
        k1, e1 = wx.extract_files(kp[0], kp[1], ex, max_batch=5, threads=3, **kw)
        assert k0 == k1 and np.array_equal(e0, e1)
        k2, e2 = wx.extract_list("scp", lines_scp, ex, max_batch=5, num_workers=2, **kw)
        assert k2 == k0 and np.array_equal(e2, e0)
    assert wx.split_path_list("raw", [json.dumps({"key": "p", "wav": "cat x.wav |", "spk": "s"})]) is None
    assert wx.split_path_list("shard", ["a.tar"]) is None
    kp = wx.split_path_list("scp", lines_scp)
    assert wx.extract_files(kp[0], kp[1], ex, batch_size=16, chunk_len=30000) is None      # shorter than the crop
    assert wx.extract_files(kp[0], kp[1], ex, resample_rate=8000) is None                  # other sample rate
    k3, e3 = wx.extract_list("scp", lines_scp, ex, batch_size=16, chunk_len=30000, seed=1)  # ... general path: tiled
    assert len(k3) == 21 and e3.shape == (21, 6) and (e3[:, 4] == 30000).all()


def test_plan_batches_and_path_table(tmp_path):
    """The file path's batch plan (longest first, consecutive files of the sorted list, cut at max_batch rows and at
    the length tolerance) and the one-pass path table the C++ loaders read names from."""
    rng = np.random.RandomState(0)
    for tol, draw in ((0.0, lambda n: rng.choice([32000, 16000, 24000], size=n)),
                      (0.12, lambda n: rng.randint(24000, 40001, size=n))):
        for n in (1, 5, 300, 4096):
            c = draw(n).astype(np.int32)
            bs = wx.plan_batches(c, 256, tol)
            assert sorted(np.concatenate(bs).tolist()) == list(range(n))            # every file once
            tops = [int(c[b].max()) for b in bs]
            assert tops == sorted(tops, reverse=True)                               # longest first
            for b in bs:
                assert 1 <= len(b) <= 256
                assert c[b].min() * (1 + tol) >= c[b].max() if tol else c[b].min() == c[b].max()
    assert len(wx.plan_batches(np.full(4096, 32000, np.int32), 256, 0.12)) == 16
    # 1.5 .. 2.5 s lengths: 16 full batches (the geometric classes of round 5 made 20, five of them part-filled)
    assert [len(b) for b in wx.plan_batches(rng.randint(24000, 40001, size=4096).astype(np.int32), 256, 0.12)] == [256] * 16
    paths = []
    for i, nsamp in enumerate((32000, 1234, 16000)):
        p = str(tmp_path / ("f%d é.wav" % i))                                  # (a non-ASCII name too)
        synth.write_wav(p, synth.synth_wav(i, nsamp))
        paths.append(p)
    tab = wx.PathTable(paths)
    ns, sr = wx.probe_wavs(tab, 2)
    assert ns.tolist() == [32000, 1234, 16000] and sr.tolist() == [16000] * 3
    dst = np.zeros((2, 16000), np.int16)
    wx.load_wav_rows(tab, dst, np.array([16000, 1234], np.int32), None, 2, idx=np.array([2, 1]))
    assert np.array_equal(dst[0], synth.synth_wav(2, 16000)) and np.array_equal(dst[1, :1234], synth.synth_wav(1, 1234))
    assert wx.decode_threads(3) == 6 and 8 <= wx.decode_threads(0) <= 32


def test_bulk_ark_writer_equals_the_record_writer(tmp_path):
    """kaldi_io.write_vectors (one pass over the table) writes VectorWriter's bytes: float32 / float64, keys of equal and
    of different lengths, the scp offsets, an empty table."""
    from wespeaker_amd import kaldi_io
    for keys in (["utt%05d" % i for i in range(300)], ["u%d" % i for i in range(300)]):
        for dt in (np.float32, np.float64):
            m = np.random.RandomState(1).randn(300, 19).astype(dt)
            a, b = str(tmp_path / "a.ark"), str(tmp_path / "b.ark")
            with kaldi_io.VectorWriter(a, a[:-3] + "scp") as w:
                for k, e in zip(keys, m):
                    w(k, e)
            kaldi_io.write_vectors(keys, m, b, b[:-3] + "scp")
            assert open(a, "rb").read() == open(b, "rb").read()
            assert open(a[:-3] + "scp").read().replace("a.ark", "X") == open(b[:-3] + "scp").read().replace("b.ark", "X")
            back = kaldi_io.read_vec_scp(b[:-3] + "scp")
            assert np.array_equal(back[keys[7]], m[7].astype(np.float32) if dt is np.float32 else m[7])
    kaldi_io.write_vectors([], np.zeros((0, 4), np.float32), str(tmp_path / "e.ark"), str(tmp_path / "e.scp"))
    assert os.path.getsize(str(tmp_path / "e.ark")) == 0
