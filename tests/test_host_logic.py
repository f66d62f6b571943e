"""CPU: host-side logic that needs no GPU -- I/O formats, synthetic generators, sharding rule,
and the world_size-2 gather path on the gloo backend."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wespeaker_amd import audio, kaldi_io, parallel
from fixtures import synth


def test_synth_is_deterministic():
    a, b = synth.synth_wav(3), synth.synth_wav(3)
    assert a.dtype == np.int16 and a.shape == (32000,) and np.array_equal(a, b)
    assert not np.array_equal(a, synth.synth_wav(4))
    s1 = synth.synth_ecapa_state_dict("ECAPA_TDNN_c512", seed=1)
    s2 = synth.synth_ecapa_state_dict("ECAPA_TDNN_c512", seed=1)
    assert all(np.array_equal(s1[k], s2[k]) for k in s1)
    n_params = sum(v.size for k, v in s1.items() if v.dtype == np.float32 and "running" not in k)
    assert abs(n_params - 5.80e6) < 0.05e6        # 5.80 M (runtime/onnxruntime/README.md:77-88 scale)
    g = synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512")
    n_glob = sum(v.size for k, v in g.items() if v.dtype == np.float32 and "running" not in k)
    assert abs(n_glob - 6.19e6) < 0.05e6


def test_wav_roundtrip(tmp_path):
    pcm = synth.synth_wav(2, 8000)
    p = str(tmp_path / "a.wav")
    synth.write_wav(p, pcm)
    x, sr = audio.load_wav(p, normalize=False)
    assert sr == 16000 and x.dtype == torch.int16 and x.shape == (1, 8000)
    assert np.array_equal(x[0].numpy(), pcm)
    y, _ = audio.load_wav(p, normalize=True)
    assert y.dtype == torch.float32 and torch.allclose(y[0], torch.from_numpy(pcm / 32768.0).float())


def test_kaldi_vector_ark_scp_roundtrip(tmp_path):
    ark, scp = str(tmp_path / "x.ark"), str(tmp_path / "x.scp")
    vecs = {"utt%d" % i: np.random.RandomState(i).randn(192).astype(np.float32) for i in range(5)}
    with kaldi_io.VectorWriter(ark, scp) as w:
        for k, v in vecs.items():
            w(k, v)
    back = kaldi_io.read_vec_scp(scp)
    assert list(back) == list(vecs)
    assert all(np.array_equal(back[k], vecs[k]) for k in vecs)
    seq = kaldi_io.read_vec_ark(ark)
    assert all(np.array_equal(seq[k], vecs[k]) for k in vecs)
    # byte layout: key ' ' \0 B F V ' ' \4 int32 dim payload   (utils/plda/kaldi_utils.py:58-79)
    raw = open(ark, "rb").read()
    assert raw.startswith(b"utt0 \x00BFV \x04\xc0\x00\x00\x00")


def test_shard_rule_matches_reference_split():
    # tools/extract_embedding.sh:40-42: split -l ceil(N/nj) -> contiguous chunks, last one short
    for n, g in [(100, 8), (4874, 8), (7, 8), (0, 4), (10000, 4), (5, 1)]:
        ranges = [parallel.shard_range(n, r, g) for r in range(g)]
        covered = [i for lo, hi in ranges for i in range(lo, hi)]
        assert covered == list(range(n))
        per = parallel.shard_size(n, g)
        assert all(hi - lo <= per for lo, hi in ranges)


def _gloo_worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    wavs = torch.arange(n_total * 4, dtype=torch.float32).reshape(n_total, 4)

    def fake_extract(block):            # "embedding" = row sums, width 3
        s = block.sum(1, keepdim=True)
        return torch.cat([s, s * 2, s * 3], 1)

    out = parallel.extract_sharded(fake_extract, wavs, batch_size=3)
    expect = fake_extract(wavs)
    ok = bool(torch.equal(out, expect))
    # the pipelined form bench.py --gpus N uses: several gathers in flight, joined in order
    lo, hi = parallel.shard_range(n_total, rank, world)
    pend = [parallel.gather_rows_async(fake_extract(wavs[lo:hi] + k), n_total) for k in range(4)]
    for k, h in enumerate(pend):
        ok = ok and bool(torch.equal(h.wait(), fake_extract(wavs + k)))
        ok = ok and bool(torch.equal(h.wait(), fake_extract(wavs + k)))      # wait() twice is harmless
    q.put((rank, ok, tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _gloo_llr_worker(rank, world, port, n_enroll, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from oracle import plda as oplda
    from wespeaker_amd import parallel
    parallel.init_distributed("gloo")
    p = synth.synth_plda(16, seed=3)
    emb, _ = synth.synth_embeddings(n_enroll + 9, 16, seed=5)
    e_t = np.stack([oplda.prepare_test(p, v) for v in emb[:n_enroll]])
    t_t = np.stack([oplda.prepare_test(p, v) for v in emb[n_enroll:]])
    full_ref = oplda.llr_matrix_vectorised(p, e_t, np.ones(n_enroll, dtype=np.int64), t_t)
    calls = []

    def block(lo, hi):                     # the reference's arithmetic on this rank's enrollment rows only
        calls.append((lo, hi))
        return torch.from_numpy(oplda.llr_matrix_vectorised(p, e_t[lo:hi], np.ones(hi - lo, dtype=np.int64), t_t)
                                .reshape(hi - lo, 9))
    got = parallel.llr_matrix_sharded(block, n_enroll)
    only0 = parallel.llr_matrix_sharded(block, n_enroll, to_rank0_only=True)
    # (numpy's BLAS rounds a 1-row product differently from the same row inside a larger one: compare to 1e-12)
    ok = bool(np.abs(got.numpy() - full_ref).max() < 1e-12) and calls[0] == parallel.shard_range(n_enroll, rank, world)
    ok = ok and ((only0 is None) == (rank != 0))
    q.put((rank, ok, tuple(got.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_enroll", [7, 2, 1])
def test_world_size_2_sharded_plda_matrix_on_gloo(n_enroll):
    """the LLR matrix with the enrollment rows sharded over two ranks equals the unsharded matrix, entry
    for entry (7 rows: 4 + 3; 2: 1 + 1; 1: 1 + an empty shard)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() + n_enroll) % 90
    procs = [ctx.Process(target=_gloo_llr_worker, args=(r, 2, port, n_enroll, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape == (n_enroll, 9), (rank, ok, shape)


@pytest.mark.parametrize("n_total", [11, 2, 1])
def test_world_size_2_gather_on_gloo(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() + n_total) % 300
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape == (n_total, 3), (rank, ok, shape)


# ------------------------------------------------------------------------- PLDA model files
def test_kaldi_plda_reader_matches_reference_reader(golden_dir):
    """read_kaldi_plda against what the REFERENCE's read_plda (utils/plda/kaldi_utils.py:24-109)
    returned for the same committed files (oracle/make_golden.py kaldi_plda); the text form has no
    reference-side run (kaldi_io._read_mat_ascii is third-party) and is held to the same arrays."""
    from wespeaker_amd.plda import TwoCovPLDA, read_kaldi_plda
    g = np.load(os.path.join(golden_dir, "kaldi_plda_ref.npz"))
    for tag in ("f32", "f64"):
        mu, tr, psi = read_kaldi_plda(os.path.join(golden_dir, "kaldi_plda_%s.bin" % tag))
        assert mu.dtype == tr.dtype == psi.dtype == np.float64
        assert np.array_equal(mu, g[tag + "/mu"]) and np.array_equal(tr, g[tag + "/transform"])
        assert np.array_equal(psi, g[tag + "/psi"])
    mu, tr, psi = read_kaldi_plda(os.path.join(golden_dir, "kaldi_plda.txt"))
    assert np.array_equal(mu, g["f64/mu"]) and np.array_equal(tr, g["f64/transform"])
    assert np.array_equal(psi, g["f64/psi"])
    # load_model(from_kaldi=True): offset = -transform @ mu (two_cov_plda.py:344-347)
    plda = TwoCovPLDA.load_model(os.path.join(golden_dir, "kaldi_plda_f64.bin"), from_kaldi=True)
    assert plda.dim == 6 and np.allclose(plda.offset, -g["f64/transform"] @ g["f64/mu"], atol=0)
    assert plda.normalize_length is False


def test_kaldi_plda_reader_rejects_damaged_files(tmp_path, golden_dir):
    from wespeaker_amd.plda import read_kaldi_plda
    raw = open(os.path.join(golden_dir, "kaldi_plda_f64.bin"), "rb").read()
    for name, data in (("trunc", raw[:-20]), ("noend", raw[:-8] + b"garbage "), ("tag", raw.replace(b"DM ", b"XM "))):
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(ValueError):
            read_kaldi_plda(str(p))
    p = tmp_path / "other"
    p.write_bytes(b"\0B<Nnet> ")
    with pytest.raises(ValueError):
        read_kaldi_plda(str(p))


def test_plda_model_roundtrip_keeps_the_callers_file_name(tmp_path):
    """Reference recipes pass suffix-less names (${exp_dir}/plda, exp/plda_adapt): save_model must
    write exactly that path and load_model must read it back."""
    from wespeaker_amd.plda import TwoCovPLDA
    p = synth.synth_plda(8, seed=3, normalize_length=True)
    plda = TwoCovPLDA.from_params(p["mu"], p["transform"], p["psi"], p["offset"], True, True)
    for name in ("plda", "plda_adapt.npz", "model.h5"):
        path = str(tmp_path / name)
        plda.save_model(path)
        assert os.path.exists(path) and not os.path.exists(path + ".npz")
        back = TwoCovPLDA.load_model(path)
        for k in ("mu", "transform", "psi", "offset"):
            assert np.array_equal(getattr(back, k), getattr(plda, k))
        assert back.normalize_length is True and back.subtract_train_set_mean is True
    (tmp_path / "junk").write_bytes(b"not a model at all")
    with pytest.raises(ValueError):
        TwoCovPLDA.load_model(str(tmp_path / "junk"))


def test_plda_hdf5_files_have_the_reference_layout_and_survive_a_rewrite(tmp_path):
    """save_model writes the reference's HDF5 layout (two_cov_plda.py:311-339) through libhdf5: checked with the
    HDF5 project's own tools, not with our reader -- h5dump must show six datasets, the arrays float64, chunked,
    gzip + fletcher32, unlimited maxshape, the flags int64 scalars -- and load_model must read the file after
    h5repack has rewritten it contiguous and unfiltered (another writer's layout).  h5py is not in this image, so
    an h5py-written file cannot be pinned here; both sides sit on the same C library h5py wraps."""
    import shutil
    import subprocess
    from wespeaker_amd import hdf5_io
    from wespeaker_amd.plda import TwoCovPLDA
    if not hdf5_io.available():
        pytest.skip("no HDF5 C library in this environment")
    p = synth.synth_plda(10, seed=5, normalize_length=True)
    plda = TwoCovPLDA.from_params(p["mu"], p["transform"], p["psi"], p["offset"], True, False)
    path = str(tmp_path / "plda")
    plda.save_model(path)                                        # default format = hdf5 here
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    back = TwoCovPLDA.load_model(path)
    for k in ("mu", "transform", "psi", "offset"):
        assert np.array_equal(getattr(back, k), getattr(plda, k)) and getattr(back, k).dtype == np.float64
    assert back.normalize_length is True and back.subtract_train_set_mean is False
    plda.save_model(str(tmp_path / "plda_npz"), fmt="npz")       # the other container on request
    assert open(str(tmp_path / "plda_npz"), "rb").read(2) == b"PK"
    with pytest.raises(KeyError):
        hdf5_io.read_datasets(path, ("mu", "no_such_dataset"))
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    h5repack = shutil.which("h5repack") or "/opt/conda/bin/h5repack"
    if not (os.path.exists(h5dump) and os.path.exists(h5repack)):
        pytest.skip("HDF5 command-line tools not installed")
    hdr = subprocess.run([h5dump, "-p", "-H", path], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert hdr.count("DATASET ") == 6
    for name, shape in (("mu", "( 10 ) / ( 10 )"), ("transform", "( 10, 10 ) / ( H5S_UNLIMITED, H5S_UNLIMITED )")):
        block = hdr[hdr.index('DATASET "%s"' % name):]
        block = block[:block.index("ALLOCATION_TIME")]
        assert "H5T_IEEE_F64LE" in block and shape in block and "CHUNKED" in block
        assert "CHECKSUM FLETCHER32" in block and "COMPRESSION DEFLATE { LEVEL 4 }" in block
    flag = hdr[hdr.index('DATASET "normalize_length"'):]
    assert "H5T_STD_I64LE" in flag[:300] and "SCALAR" in flag[:300]
    val = subprocess.run([h5dump, "-d", "/psi", "-w", "0", path], stdout=subprocess.PIPE, text=True, check=True).stdout
    got = np.array([float(x) for x in val[val.index("(0):") + 4:val.index("}", val.index("(0):"))].split(",")])
    assert np.allclose(got, plda.psi, rtol=1e-5)                 # (h5dump prints 6 significant digits)
    plain = str(tmp_path / "plda_contiguous.h5")
    subprocess.run([h5repack, "-l", "CONTI", "-f", "NONE", path, plain], check=True, stdout=subprocess.PIPE)
    again = TwoCovPLDA.load_model(plain)
    for k in ("mu", "transform", "psi", "offset"):
        assert np.array_equal(getattr(again, k), getattr(plda, k))
    assert again.normalize_length is True and again.subtract_train_set_mean is False


def _layout_lines(path):
    """h5dump's header of a file, minus what legitimately differs between writers (byte offsets, sizes, fill / allocation
    timing): data types, data spaces, chunking and the filter pipeline remain."""
    import shutil
    import subprocess
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if not os.path.exists(h5dump):
        pytest.skip("h5dump not installed")
    out = subprocess.run([h5dump, "-p", "-H", path], stdout=subprocess.PIPE, text=True, check=True).stdout
    keep = []
    for line in out.splitlines()[1:]:
        t = line.strip()
        if t.startswith(("SIZE", "OFFSET", "FILL_TIME", "H5D_ALLOC_TIME", "VALUE")):
            continue
        keep.append(t)
    return keep


def test_plda_hdf5_against_a_file_written_by_h5py(golden_dir, tmp_path):
    """tests/golden/plda_h5py.h5 was written by the real h5py (3.3.0, the Anaconda interpreter of this image) with
    the reference's six create_dataset calls (oracle/make_golden_h5py.py): load_model reads it to the last bit, and
    the file save_model writes has the SAME data types, data spaces, chunking and filter pipeline, dataset by
    dataset.  Where that interpreter exists, h5py also reads back what we wrote."""
    import json
    import subprocess
    from wespeaker_amd import hdf5_io
    from wespeaker_amd.plda import TwoCovPLDA
    if not hdf5_io.available():
        pytest.skip("no HDF5 C library in this environment")
    gold = os.path.join(golden_dir, "plda_h5py.h5")
    exp = np.load(os.path.join(golden_dir, "plda_h5py_expected.npz"))
    plda = TwoCovPLDA.load_model(gold)
    for k in ("mu", "transform", "psi", "offset"):
        assert np.array_equal(getattr(plda, k), exp[k]) and getattr(plda, k).dtype == np.float64
    assert plda.normalize_length is True and plda.subtract_train_set_mean is False and plda.dim == 12
    ours = str(tmp_path / "plda")
    plda.save_model(ours)
    assert _layout_lines(ours) == _layout_lines(gold)
    py39 = "/opt/conda/bin/python3.9"
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "make_golden_h5py.py")
    if os.path.exists(py39) and subprocess.run([py39, "-c", "import h5py"], stderr=subprocess.DEVNULL).returncode == 0:
        r = subprocess.run([py39, "-W", "ignore", script, "read", ours], stdout=subprocess.PIPE, text=True, check=True)
        back = json.loads(r.stdout.strip().splitlines()[-1])
        for k in ("mu", "transform", "psi", "offset"):
            assert np.array_equal(np.array(back[k]), exp[k]) and back["_dtypes"][k] == "float64"
        assert back["normalize_length"] == 1 and back["subtract_train_set_mean"] == 0
        assert back["_shapes"]["normalize_length"] == [] and back["_dtypes"]["normalize_length"] == "int64"


def test_twocovplda_constructor_keeps_the_reference_positional_order():
    """two_cov_plda.py:68-73: (scp_file, utt2spk_file, embed_dim, subtract_train_set_mean,
    normalize_length); trained parameters are keyword-only here."""
    import inspect
    from wespeaker_amd.plda import TwoCovPLDA
    sig = list(inspect.signature(TwoCovPLDA.__init__).parameters.values())[1:]
    assert [p.name for p in sig[:5]] == ["scp_file", "utt2spk_file", "embed_dim",
                                         "subtract_train_set_mean", "normalize_length"]
    assert all(p.kind == inspect.Parameter.KEYWORD_ONLY for p in sig[5:])
    bare = TwoCovPLDA(None, None, 32, True, False)
    assert bare.dim == 32 and bare.subtract_train_set_mean and not bare.normalize_length
    assert bare.mu.shape == (32,) and bare.transform.shape == (32, 32)


def test_chain_string_parser_known_answers():
    """The grammar of wespeaker/utils/embedding_processing.py:23-67 (its own docstring example first);
    checked live against the reference function when /root/reference is present."""
    from wespeaker_amd.embedding_processing import chain_string_to_dict
    example = ("mean-subtract --scp mean1_xvector.scp | length-norm | lda  --scp lda_xvector.scp "
               "--utt2spk utt2spk --dim 100 | length-norm")
    assert chain_string_to_dict(example) == [
        ["mean-subtract", {"scp": "mean1_xvector.scp"}], ["length-norm", {}],
        ["lda", {"scp": "lda_xvector.scp", "utt2spk": "utt2spk", "dim": "100"}], ["length-norm", {}]]
    assert chain_string_to_dict(None) == []
    assert chain_string_to_dict("lda --dim=20 --eps 1e-5") == [["lda", {"dim": "20", "eps": "1e-5"}]]
    with pytest.raises(AssertionError):
        chain_string_to_dict("lda --dim")
    from oracle import ref_shim
    if ref_shim.available():
        ref = ref_shim.ref_module("wespeaker.utils.embedding_processing").chain_string_to_dict
        for c in (example, "whitening | length-norm ", " length-norm", "mean-subtract --scp a|length-norm",
                  "lda --dim=20 --scp  x --utt2spk=u --eps 1e-5"):
            assert chain_string_to_dict(c) == ref(c)


def test_subsegment_mirror_matches_reference_golden():
    """wespeaker_amd.speaker.subsegment / subsegment_ids / ws_num_windows against what the reference's own
    diar/extract_emb.py:55-83 returned (tests/golden/subsegment_ref.npz, row-index features)."""
    import os
    from wespeaker_amd import _lib
    from wespeaker_amd.speaker import subsegment, subsegment_ids
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "subsegment_ref.npz"))
    n_cases = len([k for k in g.files if k.endswith("_params")])
    assert n_cases == 8
    for k in range(n_cases):
        nf, seg_len, win, per = (int(v) for v in g["case%d_params" % k])
        fb = np.repeat(np.arange(nf, dtype=np.float32)[:, None], 4, axis=1)
        seg_id = "{:08d}-{:08d}".format(1230, 1230 + seg_len * 10)
        names, wins = subsegment(fb, seg_id, win, per, 10)
        assert names == [str(x) for x in g["case%d_names" % k]]
        assert np.array_equal(np.stack(wins)[:, :, 0].astype(np.int32), g["case%d_rows" % k])
        assert subsegment_ids(seg_id, seg_len, win, per) == names
        assert _lib.lib().ws_num_windows(seg_len, win, per) == len(names)
    assert _lib.lib().ws_num_windows(0, 150, 75) == 0 and _lib.lib().ws_num_windows(10, 0, 75) == 0


def test_kaldi_matrix_reader_formats(tmp_path):
    """kaldi_io.read_mat / read_mat_scp (what kaldiio.load_mat is to processor.parse_feat, dataset/processor.py:171-196):
    binary float / double records at scp offsets, the text form, and the three compressed layouts of Kaldi's
    compressed-matrix.h (round trip through this package's own writer: kaldiio is not in the image)."""
    import io
    from wespeaker_amd import kaldi_io
    rs = np.random.RandomState(3)
    mats = {"a": rs.randn(57, 80).astype(np.float32) * 4 + 2, "b": rs.randn(3, 23).astype(np.float32),
            "c": rs.randn(200, 80).astype(np.float64)}
    ark, scp = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")
    with open(ark, "wb") as f, open(scp, "w") as g:
        for k, m in mats.items():
            g.write("%s %s:%d\n" % (k, ark, kaldi_io.write_mat(f, k, m)))
    back = kaldi_io.read_mat_scp(scp)
    for k, m in mats.items():
        assert back[k].dtype == np.float32 and np.array_equal(back[k], m.astype(np.float32))
    txt = str(tmp_path / "t.txt")
    with open(txt, "w") as f:
        f.write(" [\n  1 2 3\n  4 5.5 -6 ]\n")
    assert np.array_equal(kaldi_io.read_mat(txt), np.array([[1, 2, 3], [4, 5.5, -6]], np.float32))
    m = mats["a"]
    span = float(m.max() - m.min())
    for fmt, tol in ((2, span / 65535.0), (3, span / 255.0), (1, span / 60.0)):
        f = io.BytesIO()
        off = kaldi_io.write_mat(f, "x", m, compress=fmt)
        f.seek(off)
        r = kaldi_io._read_matrix_at(f)
        assert r.shape == m.shape and np.abs(r - m).max() <= tol, (fmt, np.abs(r - m).max())
    with pytest.raises(ValueError):
        kaldi_io._read_matrix_at(io.BytesIO(b"\0BXX 1234"))
