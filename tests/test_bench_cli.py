"""bench.py's own launch / sharding / gather / JSON logic on CPU (world size 2, gloo) with the host stand-in for
the extractor (WS_BENCH_STUB=1): `python bench.py --gpus 2` must launch itself -- the driver's N = 1 command form
used at N > 1 -- like the reference's launcher spawns its per-GPU jobs (tools/extract_embedding.sh:46-65)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None):
    env = dict(os.environ, WS_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                      # a clean shell: the self-launch path
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_fixed_size_set_two_ranks_equals_one_rank():
    """--total-utts: contiguous shards (37 utterances -> 19 + 18), one gather, strong scaling; the gathered table of the
    self-launched two-rank run equals the one-rank run's."""
    one = run_bench("--gpus", "1", "--total-utts", "37", "--batch", "8", "--steps", "2", "--warmup", "1")
    two = run_bench("--gpus", "2", "--total-utts", "37", "--batch", "8", "--steps", "2", "--warmup", "1")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["scaling"] == two["scaling"] == "strong"
    assert two["set"]["per_rank_utts"] == 19 and one["set"]["per_rank_utts"] == 37
    assert two["config"]["total_utts"] == 37 and two["steps"] == 2 and two["warmup"] == 1
    assert two["value"] > 0 and two["unit"] == "embeddings/s"
    a, b = one["set"]["embedding_checksum"], two["set"]["embedding_checksum"]
    assert abs(a - b) <= 1e-6 * abs(a)


def test_default_mode_self_launch_is_weak_scaling():
    d = run_bench("--gpus", "2", "--batch", "8", "--steps", "2", "--warmup", "1", "--windows", "2")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 16 and d["config"]["per_gpu_batch"] == 8
    assert d["value"] > 0 and d["higher_is_better"] is True
    assert d["metric"].startswith("embeddings/sec")


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WS_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus" in r.stderr


@pytest.mark.parametrize("workload,world,total,trials", [("vox1o", 8, 4874, 37611), ("vox1o", 4, 4874, 37611),
                                                         ("stream10k", 4, 10000, 0), ("stream10k", 8, 10000, 0)])
def test_baseline_sets_on_four_and_eight_ranks(workload, world, total, trials):
    """BASELINE configs 2 / 3 as the driver's 8-GPU node will run them (`bench.py --gpus N --workload ...`), here with
    the host stand-in on gloo: contiguous ceil(U / G) shards in list order with a short last shard (VoxCeleb1-O over 8
    ranks: 7 x 610 + 604, tools/extract_embedding.sh:39-67), equal batches inside a shard, one gather that keeps the
    list order (the order-sensitive checksum equals the one-rank run's), rank 0 scoring the full trial list, and the
    collective the line says it ran on."""
    args = ("--workload", workload, "--batch", "512", "--steps", "1", "--warmup", "1", "--seconds", "0.1")
    one = run_bench("--gpus", "1", *args)
    many = run_bench("--gpus", str(world), *args)
    assert many["n_gpus"] == world and many["scaling"] == "strong" and many["config"]["total_utts"] == total
    per = -(-total // world)
    shards = many["set"]["shards"]
    assert len(shards) == world and shards[0][0] == 0 and shards[-1][1] == total
    assert all(shards[r][1] == shards[r + 1][0] for r in range(world - 1))          # contiguous, in list order
    assert [hi - lo for lo, hi in shards[:-1]] == [per] * (world - 1)
    assert shards[-1][1] - shards[-1][0] == total - per * (world - 1) <= per         # the short last shard
    if workload == "vox1o" and world == 8:
        assert [hi - lo for lo, hi in shards] == [610] * 7 + [604]
    b = many["set"]["batches_per_rank"]                                              # rank 0's plan: equal batches
    assert sum(b) == per and max(b) - min(b) <= 1 and len(b) == -(-per // 512)
    assert many["set"]["trials_scored_per_step"] == trials
    if trials:
        assert many["set"]["trials_scored_on_rank0"] == trials and many["set"]["scores_finite"]
    for key in ("embedding_checksum", "embedding_order_checksum"):
        x, y = one["set"][key], many["set"][key]
        assert abs(x - y) <= 1e-9 * abs(x), key
    assert many["collective_backend"] == "gloo" and many["collective"]["ranks"] == world
    assert one["collective"]["ranks"] == 1 and one["collective_backend"] is None


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


def test_compact_line_of_a_full_record_stays_under_8k():
    """round 5's 24-KB line was not parsed by the driver.  The committed full record of that run goes
    through bench.compact_line: strict JSON, < 8 KB, the contract's keys + `roofline` + `cpu_baseline` + one figure per
    BASELINE config; the reference prints ONE figure (runtime/core/bin/extract_emb_main.cc:100-117)."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")) as fp:
        full = json.load(fp)
    assert len(json.dumps(full)) > 20000                     # the record that broke the parser
    text = bench.compact_line(full, "bench_detail.json")
    assert len(text) < 8192 and "\n" not in text
    line = json.loads(text, parse_constant=lambda c: pytest.fail("non-strict JSON constant " + c))
    for k in REQUIRED:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and line["config"]["workload"]
    roof = line["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
              "algorithmic_flops_per_launch", "whole_step_frac_of_peak", "traffic_source"):
        assert k in roof, k
    assert len(roof["kernel"]) <= 200 and roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-4)
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert line["plda"]["pairs_trials_per_s"] > 0 and line["plda"]["roofline"]["frac"] > 0
    assert line["plda"]["cpu_baseline"]["value"] > 0
    assert set(line["configs"]) == {"ECAPA_TDNN_GLOB_c1024", "ResNet34", "ResNet221", "CAMPPlus"}
    assert all(line["configs"][m]["fp32"]["value"] > 0 for m in line["configs"])
    assert len(line["fixed_size_sets"]) == 3
    assert line["collective"]["ranks"] == 1 and line["detail"] == "bench_detail.json"


def test_stub_lines_are_short_and_the_detail_file_is_written(tmp_path):
    """Every mode's stdout line is one strict-JSON line < 8 KB; the full record lands in --detail-file."""
    det = str(tmp_path / "detail.json")
    for argv in (("--gpus", "2", "--batch", "8", "--steps", "2", "--warmup", "1", "--windows", "2"),
                 ("--gpus", "8", "--workload", "vox1o", "--batch", "512", "--steps", "1", "--warmup", "1",
                  "--seconds", "0.1")):
        env = dict(os.environ, WS_BENCH_STUB="1")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--detail-file", det] + list(argv),
                           env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        allout = [l for l in r.stdout.splitlines() if l.strip()]
        out = [l for l in allout if l.startswith("{")]           # (gloo prints its own connection notes on stdout)
        assert len(out) == 1 and len(out[0]) < 8192 and allout[-1] == out[0], r.stdout[:500]
        line = json.loads(out[0])
        for k in REQUIRED:
            assert k in line, k
        with open(det) as fp:
            full = json.load(fp)
        assert full["value"] == pytest.approx(line["value"], rel=1e-5)


def test_eight_ranks_value_is_priced_on_the_slowest_rank_and_scoring_is_inside_the_step():
    """What the driver's 8-GPU node will rely on, with the host stand-in on gloo at world size 8:
    (a) `value` = all ranks' utterances / the MAX-over-ranks time -- rank 5 is made 1 s slower per extract call, rank 0
    (which prints the line) is not, and ms_per_step must carry those seconds; (b) rank 0's scoring of the VoxCeleb1-O
    trial list lies INSIDE the timed region of `--workload vox1o` -- a 2-s sleep inside the scoring step shows in
    ms_per_step; (c) the compact line of an N > 1 run keeps `collective{backend, ranks, library_version}` and the
    shard table."""
    args = ("--gpus", "8", "--workload", "vox1o", "--batch", "512", "--steps", "1", "--warmup", "1", "--seconds", "0.1")
    base = run_bench(*args)
    slow = run_bench(*args, env_extra={"WS_BENCH_STUB_SLOW": "5:1.0"})
    # one pass of a rank = 2 batches of 305 utterances -> two extract calls -> >= 2 s on rank 5 only
    assert slow["ms_per_step"] >= 2000.0 and slow["ms_per_step"] > base["ms_per_step"] + 1000.0, \
        (slow["ms_per_step"], base["ms_per_step"])
    assert slow["value"] == pytest.approx(4874 / (slow["ms_per_step"] * 1e-3), rel=1e-3)
    scored = run_bench(*args, env_extra={"WS_BENCH_STUB_SCORE_SLEEP": "2.0"})
    assert scored["ms_per_step"] >= 2000.0 and scored["set"]["trials_scored_on_rank0"] == 37611
    for d in (base, slow, scored):
        assert d["n_gpus"] == 8 and d["scaling"] == "strong"
        assert d["collective"] == {"backend": "gloo", "ranks": 8, "library_version": None}
        assert len(d["set"]["shards"]) == 8 and d["set"]["shards"][-1] == [4270, 4874]
    # weak-scaling default mode: the same pricing
    w = run_bench("--gpus", "8", "--batch", "8", "--steps", "2", "--warmup", "1", "--windows", "1",
                  env_extra={"WS_BENCH_STUB_SLOW": "3:0.5"})
    assert w["n_gpus"] == 8 and w["scaling"] == "weak" and w["config"]["global_batch"] == 64
    assert w["ms_per_step"] >= 500.0 and w["value"] == pytest.approx(64 / (w["ms_per_step"] * 1e-3), rel=1e-3)
    assert w["collective"]["ranks"] == 8
