"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, host logic of the matcher
API (tiler, pyramid, geometric verification, timers), the sequence sharding + match-table all-gather on gloo
(world_size 2), and the synthetic data generators."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from icepy4d_amd import synthetic


def test_library_exports_every_declared_symbol():
    from icepy4d_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()  # binds every name of _lib.SIGNATURES; no compute, no GPU needed
    header = open(os.path.join(ROOT, "include", "icematch.h")).read()
    declared = set(re.findall(r"\b(im_[a-z0-9_]+)\s*\(", header))
    declared -= {"im_ctx"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.im_version() >= 100


def test_no_cpu_fallback_without_gpu():
    """The product path fails loudly when there is no HIP device."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from icepy4d_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(0)
    from icepy4d_amd.matching import LightGlueMatcher
    with pytest.raises(RuntimeError):
        LightGlueMatcher({"state_dicts": {"superpoint": {}, "lightglue": {}}})


def test_product_never_imports_oracle():
    """No product source imports, includes, loads or executes anything under oracle/ (docstrings may NAME the checker files)."""
    import ast
    import re
    for dirpath, _, files in os.walk(os.path.join(ROOT, "icepy4d_amd")):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                tree = ast.parse(open(path).read())
                for node in ast.walk(tree):
                    if isinstance(node, ast.Import):
                        assert not any(a.name.split(".")[0] == "oracle" for a in node.names), path
                    elif isinstance(node, ast.ImportFrom):
                        assert (node.module or "").split(".")[0] != "oracle", path
                    elif isinstance(node, ast.Call):     # importlib.import_module("oracle...") / __import__ / open / CDLL on oracle paths
                        for arg in node.args:
                            if isinstance(arg, ast.Constant) and isinstance(arg.value, str):
                                assert not re.match(r"^(oracle($|[./]))", arg.value), (path, arg.value)
            elif f.endswith((".hip", ".h", ".cpp")):
                for line in open(path):
                    if line.lstrip().startswith("#include"):
                        assert "oracle" not in line, (path, line)


def test_tiler_matches_reference():
    from icepy4d_amd.matching import Tiler
    g = load_golden("g4_wrappers")
    t = Tiler(grid=[2, 3], overlap=15, origin=[0, 0])
    lims, origin = t.compute_limits_by_grid(g["image0"])
    assert np.array_equal(np.array([lims[i] for i in sorted(lims)]), g["tiler_limits"])
    assert np.array_equal(t.extract_patch(g["image0"], lims[4]), g["tiler_patch"])
    # q4: a 1x1 grid drops the last row and column
    lims, _ = Tiler(grid=[1, 1]).compute_limits_by_grid(np.zeros((800, 1200), np.uint8))
    assert Tiler.extract_patch(np.zeros((800, 1200), np.uint8), lims[0]).shape == (799, 1199)


def test_enums_match_reference_values():
    from icepy4d_amd.matching import GeometricVerification, Quality, TileSelection
    assert [e.value for e in TileSelection] == [0, 1, 2, 3] and TileSelection.PRESELECTION.value == 3
    assert GeometricVerification.NONE.value == 1 and GeometricVerification.MAGSAC.value == 3
    assert Quality.LOW.value == 1 and Quality.HIGHEST.value == 4


def test_pyramid():
    """The numpy restatement of cv2.pyrDown / pyrUp that the device kernels are held to (tests/test_gpu_models.py), against
    hand-computed samples and an independent scipy formulation."""
    from oracle.pyramid_cpu import pyr_down, pyr_up
    a = np.full((37, 50), 100, np.uint8)
    assert pyr_down(a).shape == (19, 25) and (pyr_down(a) == 100).all()
    assert pyr_up(a).shape == (74, 100) and (pyr_up(a) == 100).all()
    rng = np.random.default_rng(0)
    b = synthetic.band_limited_noise(rng, 64, 96)
    d = pyr_down(b)
    # hand-computed interior sample of the 5x5 binomial filter
    k = np.array([1, 4, 6, 4, 1], np.int64)
    y, x = 10, 20
    ref = (np.outer(k, k) * b[2 * y - 2:2 * y + 3, 2 * x - 2:2 * x + 3].astype(np.int64)).sum()
    assert d[y, x] == (ref + 128) >> 8
    assert abs(float(pyr_up(d).mean()) - float(b.mean())) < 2.0
    # whole-image cross-check against an independent formulation of the documented algorithm (scipy correlate, mode
    # 'mirror' = BORDER_REFLECT_101): separable [1 4 6 4 1] on the full-resolution image, every second sample, (s + 128) >> 8;
    # odd sizes and 3 channels included. (cv2 itself is absent: parity with an OpenCV build stays unpinned.)
    from scipy import ndimage
    k2 = np.outer(k, k)
    for img in (b, b[:63, :95], np.stack([b, b[::-1], b[:, ::-1]], -1)):
        planes = img[..., None] if img.ndim == 2 else img
        ref = np.stack([(ndimage.correlate(planes[..., c].astype(np.int64), k2, mode="mirror")[::2, ::2] + 128) >> 8
                        for c in range(planes.shape[-1])], -1).astype(np.uint8)
        got = pyr_down(img)
        assert np.array_equal(got if img.ndim == 3 else got[..., None], ref)
    # pyrUp: zero insertion, the same kernel, (s + 32) >> 6; interior samples (the border rule acts on source indices)
    z = np.zeros((2 * d.shape[0], 2 * d.shape[1]), np.int64)
    z[::2, ::2] = d
    ref_up = (ndimage.correlate(z, k2, mode="constant") + 32) >> 6
    assert np.array_equal(pyr_up(d)[2:-2, 2:-2], np.clip(ref_up, 0, 255)[2:-2, 2:-2].astype(np.uint8))
    # border rule of OpenCV's `pyrUp_` (recalled from its source; cv2 is absent, so this pins the CHOSEN rule, not a cv2 build):
    # before the first sample reflect-101 (6 cur + 2 next), after the last one the last one itself (prev + 7 cur, 8 cur)
    u, dd = pyr_up(d).astype(np.int64), d.astype(np.int64)
    assert u[-1, -1] == dd[-1, -1]
    assert u[-2, -1] == ((dd[-2, -1] + 7 * dd[-1, -1]) * 8 + 32) >> 6
    assert u[-1, -2] == ((dd[-1, -2] + 7 * dd[-1, -1]) * 8 + 32) >> 6
    assert u[0, 0] == ((6 * dd[0, 0] + 2 * dd[1, 0]) * 6 + 2 * (6 * dd[0, 1] + 2 * dd[1, 1]) + 32) >> 6


def test_geometric_verification():
    from icepy4d_amd.matching import GeometricVerification, geometric_verification
    rng = np.random.default_rng(1)
    # points on two depth planes seen by two translated cameras -> a valid epipolar geometry
    X = np.c_[rng.uniform(-1, 1, 200), rng.uniform(-1, 1, 200), rng.uniform(4, 8, 200)]
    K = np.array([[800, 0, 320], [0, 800, 240], [0, 0, 1.0]])
    p0 = (K @ X.T).T
    p0 = p0[:, :2] / p0[:, 2:]
    X1 = X + np.array([0.5, 0.05, 0.1])
    p1 = (K @ X1.T).T
    p1 = p1[:, :2] / p1[:, 2:]
    p1[:40] += rng.uniform(20, 60, size=(40, 2))  # outliers
    from oracle import gv_cpu
    with pytest.raises(RuntimeError, match="no host fallback"):       # the product scores hypotheses on the device only
        geometric_verification(p0.astype(np.float32), p1.astype(np.float32), GeometricVerification.PYDEGENSAC, threshold=1.0)
    F, mask = geometric_verification(p0.astype(np.float32), p1.astype(np.float32), GeometricVerification.PYDEGENSAC, threshold=1.0,
                                     hypothesis_fn=gv_cpu.hypothesis_fn(p0, p1, 1.0))   # the oracle's numpy RANSAC in the device stage's place
    assert F is not None and mask[40:].mean() > 0.95 and mask[:40].mean() < 0.2
    F, mask = geometric_verification(p0[:3], p1[:3], GeometricVerification.MAGSAC)
    assert F is None and mask.all()
    F, mask = geometric_verification(p0, p1, GeometricVerification.NONE)
    assert F is None and mask.all()


def test_average_timer_and_timeit(capsys):
    from icepy4d_amd.utils import AverageTimer, timeit
    t = AverageTimer()
    t.update("matching")
    t.update("matching")
    assert "matching" in t.times and t.will_print["matching"]
    t.print("Matching")
    assert not t.will_print["matching"]

    @timeit
    def f(x):
        return x + 1

    assert f(1) == 2
    assert "Function f took" in capsys.readouterr().out


def test_matcher_option_errors():
    from icepy4d_amd.matching import ImageMatcherBase, check_dict_keys
    with pytest.raises(TypeError):
        ImageMatcherBase("nope")
    with pytest.raises(KeyError):
        check_dict_keys({"a": 1}, ["a", "b"])
    m = ImageMatcherBase({})
    with pytest.raises(NotImplementedError):
        m._match_images(np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint8))
    assert m.mkpts0 is None and m.mconf is None


def test_store_and_filter_features():
    from icepy4d_amd.matching import FeaturesBase, ImageMatcherBase
    m = ImageMatcherBase({})
    f0 = FeaturesBase(np.arange(10, dtype=np.float32).reshape(5, 2), np.arange(256 * 5, dtype=np.float32).reshape(256, 5), np.arange(5, dtype=np.float32))
    f1 = FeaturesBase(f0.keypoints + 100, f0.descriptors + 1, f0.scores + 10)
    matches0 = np.array([2, -1, 0, -1, 4])
    m._store_features(f0, f1, matches0)
    m._mconf = f0.scores[matches0 > -1]
    assert np.array_equal(m.mkpts0, f0.keypoints[[0, 2, 4]]) and np.array_equal(m.mkpts1, f1.keypoints[[2, 0, 4]])
    assert m.descriptors1.shape == (256, 3) and np.array_equal(m.scores1, f1.scores[[2, 0, 4]])
    m._filter_matches_by_mask(np.array([True, False, True]))
    assert len(m.mkpts0) == 2 and m.descriptors0.shape == (256, 2) and len(m.mconf) == 2


def test_synthetic_generators_are_deterministic():
    a0, a1 = synthetic.stereo_pair(3, 64, 96)
    b0, b1 = synthetic.stereo_pair(3, 64, 96)
    assert np.array_equal(a0, b0) and np.array_equal(a1, b1) and a0.dtype == np.uint8
    sd1, sd2 = synthetic.superpoint_state_dict(0), synthetic.superpoint_state_dict(0)
    assert all(torch.equal(sd1[k], sd2[k]) for k in sd1)
    assert set(synthetic.lightglue_state_dict(0)) == set(synthetic.lightglue_state_dict(0, "prune"))
    i0, i1 = synthetic.translated_pair(0, 64, 96, 16, 8)
    assert i0.shape == i1.shape == (64, 96)


def test_shard_and_records():
    from icepy4d_amd import sequence as sq
    assert sq.shard_epochs(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((sq.shard_epochs(2048, r, 8) for r in range(8)), [])) == list(range(2048))
    K = 16
    t = sq.new_table(2, K, "cpu")
    m0 = torch.full((K,), -1, dtype=torch.int32)
    m0[3] = 7
    ms = torch.zeros(K)
    ms[3] = 0.5
    sq.write_record(t, 1, 42, torch.tensor([10, 12], dtype=torch.int32), m0, ms, torch.tensor([9, 0, 0, 0], dtype=torch.int32))
    r = sq.decode_record(t[1].numpy(), K)
    assert r["epoch"] == 42 and r["n0"] == 10 and r["n1"] == 12 and r["n_matches"] == 1 and r["stop"] == 9
    assert r["matches0"][3] == 7 and r["matching_scores0"][3] == 0.5 and len(r["matches0"]) == 10


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from icepy4d_amd import sequence as sq
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
K, n_epochs = 8, 7
mine = sq.shard_epochs(n_epochs, rank, world)
t = sq.new_table(len(mine), K, "cpu")
for row, ep in enumerate(mine):
    m0 = torch.full((K,), -1, dtype=torch.int32); m0[ep % K] = ep
    sq.write_record(t, row, ep, torch.tensor([K, K], dtype=torch.int32), m0, torch.full((K,), float(ep)), torch.tensor([9, 0, 0, 0], dtype=torch.int32))
full = sq.all_gather_tables(t)
assert full.shape[0] == n_epochs, full.shape
assert full[:, 0].tolist() == list(range(n_epochs))
for ep in range(n_epochs):
    r = sq.decode_record(full[ep].numpy(), K)
    assert r["matches0"][ep % K] == ep and r["matching_scores0"][0] == float(ep) and r["n_matches"] == 1
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_all_gather_match_tables_gloo_world2(tmp_path):
    """The N > 1 path: round-robin epoch shards with unequal counts, one all-gather, every rank gets the full table."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


def test_match_table_checkpoint_and_resume(tmp_path):
    from icepy4d_amd import sequence as sq
    K = 8
    t = sq.new_table(4, K, "cpu")
    for row, ep in enumerate([5, 1, 3]):
        m0 = torch.full((K,), -1, dtype=torch.int32)
        m0[0] = ep
        sq.write_record(t, row, ep, torch.tensor([K, K], dtype=torch.int32), m0, torch.ones(K), torch.tensor([9, 0, 0, 0], dtype=torch.int32))
    path = str(tmp_path / "table.npz")
    sq.save_table(path, t, K)                      # the unused 4th row (epoch -1) is dropped
    rec, k = sq.load_table(path)
    assert k == K and rec[:, 0].tolist() == [1, 3, 5]
    assert sq.pending_epochs(6, rec) == [0, 2, 4]
    kp0 = np.arange(2 * K, dtype=np.float32).reshape(K, 2)
    kp1 = kp0 + 100
    a, b, conf = sq.records_to_features(rec[1], K, kp0, kp1)
    assert a.shape == (1, 2) and np.array_equal(b[0], kp1[3]) and conf[0] == 1.0


def test_committed_bench_line_has_the_contract_fields():
    """The bench line committed under profiles/ (produced by `python bench.py` on the GPU box) carries every field the
    measurement contract names; guards bench.py against silently dropping one."""
    import glob
    import json
    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")))[-1]
    line = open(latest).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0
    assert abs(d["value"] - 1e3 / d["ms_per_step"] * d["n_gpus"]) < 1e-6 * d["value"]
    if latest >= os.path.join(ROOT, "profiles", "r04_bench.json"):
        # since round 4 the default line witnesses the other configurations too: configs[2] over distinct epochs, configs[4] (attention
        # fraction, Sinkhorn time and rate) and one match() call through the plugin API
        sm = d["side_measurements"]
        assert sm["config3_distinct_epochs"]["pairs"] >= 60 and sm["config3_distinct_epochs"]["epochs_distinct_and_in_order"] is True
        assert sm["config5"]["mean_keypoints"] == 16384 and sm["config5"]["sinkhorn"]["solve_ms"] > 0 and sm["config5"]["attention"].get("frac_of_peak", sm["config5"]["attention"].get("frac_of_fp32_mfma_peak")) < 1
        assert sm["match_call_ms"]["keypoints"] == 4096 and sm["match_call_ms"]["median"] > 0
    if latest >= os.path.join(ROOT, "profiles", "r05_bench.json"):
        # since round 5: `value` is the median of five back-to-back timed regions, the line lists failed epochs, the production call of
        # main_dev.py:115-132 and the adaptive variants (with the check of the timed launch mode against direct launches); no roofline
        # fraction above 1 anywhere
        vr = d["value_repeats"]["pairs_per_s"]
        assert len(vr) == 5 and abs(sorted(vr)[2] - d["value"]) < 1e-9 * d["value"] and d["failed_epochs"] == []
        assert sm["production_call_ms"]["median"] > 0 and set(sm["production_call_ms"]["split_ms_of_the_median_call"]) >= {"preselection", "matching", "geometric_verification"}
        ad = sm["adaptive_depth_and_width"]
        for variant in ("prune_gradual", "earlystop_late"):
            assert ad[variant]["matches_equal_to_direct_launches"] is True and ad[variant]["pairs_per_s"] > d["value"]
        live = ad["prune_gradual"]["live_points_per_layer_of_one_pair"]
        assert live[0] == [4096, 4096] and live[-1][0] < 1024 and any(1024 < w[0] <= 2048 for w in live) and any(2048 < w[0] < 4096 for w in live)
        assert d["roofline"]["frac"] < 1 and sm["config5"]["sinkhorn"]["frac_of_hbm_peak_moved"] < 1
        c5 = os.path.join(ROOT, "profiles", "r05_bench_config5.json")
        if os.path.exists(c5):
            assert json.loads(open(c5).read().strip().splitlines()[-1])["roofline_sinkhorn"]["frac"] < 1


def test_sfm_relative_orientation_and_triangulation():
    """Row f-4: `estimate_pose` recovers a synthetic relative orientation (R exactly, t up to scale and sign convention
    x1 = R x0 + t) with gross outliers present; the vectorised linear triangulation equals a restatement of the
    reference's per-point formulation (`sfm/triangulation.py:166-186`: null vector of [P_i | -x_i]) and reproduces the 3-D
    points."""
    from icepy4d_amd import sfm
    rng = np.random.default_rng(3)
    n = 300
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(5, 9, n)]
    K = np.array([[1200.0, 0, 640], [0, 1200.0, 480], [0, 0, 1]])
    ang = 0.12
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([-1.0, 0.05, 0.1])
    P0, P1 = K @ np.eye(3, 4), K @ np.c_[R, t]
    h = np.c_[X, np.ones(n)]
    x0 = (P0 @ h.T).T; x0 = x0[:, :2] / x0[:, 2:]
    x1 = (P1 @ h.T).T; x1 = x1[:, :2] / x1[:, 2:]
    x1n = x1 + rng.normal(0, 0.1, x1.shape)
    x1n[:40] += rng.uniform(30, 80, size=(40, 2))
    from oracle import gv_cpu
    xn0, xn1 = (x0 - K[[0, 1], [2, 2]]) / K[[0, 1], [0, 1]], (x1n - K[[0, 1], [2, 2]]) / K[[0, 1], [0, 1]]
    Re, te, inl = sfm.estimate_pose(x0, x1n, K, K, thresh=1.0, hypothesis_fn=gv_cpu.hypothesis_fn(xn0, xn1, 1.0 / 1200.0))
    assert inl[40:].mean() > 0.95 and inl[:40].mean() < 0.1
    assert np.abs(Re - R).max() < 5e-3
    assert np.abs(te / np.linalg.norm(te) - t / np.linalg.norm(t)).max() < 2e-2
    assert sfm.estimate_pose(x0[:4], x1[:4], K, K, 1.0) is None

    # triangulation against the REFERENCE's outputs (G10, `tools/gen_golden.py triangulation`): tests/test_oracle_golden.py; here the known scene
    h0, h1 = np.c_[x0, np.ones(n)], np.c_[x1, np.ones(n)]
    Xt = sfm.triangulate_points_linear(P0, P1, h0, h1)
    assert np.abs(Xt[:, :3] - X).max() < 1e-8 and np.abs(Xt[:, 3] - 1).max() == 0


def test_bench_gpus2_as_typed_spawns_its_ranks_dry_run():
    """`python bench.py --gpus 2 ...` with no launcher around it: the parent starts one process per rank through
    torch.distributed.run before anything touches a GPU and relays rank 0's JSON line. --dry-run fabricates the records, so
    spawn, rendezvous (gloo), epoch sharding, the table all-gather and the JSON contract are exercised on a CPU-only machine."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["dry_run"] is True and d["scaling"] == "weak"
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    # the N > 1 line is self-verifying: who took part, with how many pairs each, and what the gathered table holds
    rk = d["ranks"]
    assert rk["world"] == 2 and rk["backend"] == "gloo" and rk["pairs_per_rank"] == [5, 5] and len(rk["per_rank_pairs_per_s"]) == 2
    assert rk["epochs_in_gathered_table"] == 10 and rk["epochs_complete_and_sorted"] is True and len(rk["all_gather_ms"]) == 2
    assert rk["record_bytes"] == 4 * (8 + 2 * 4096) and rk["gathered_table_bytes"] == 10 * rk["record_bytes"]
    # config 4: the 98 KB records with the keypoints of both images, sharded over the ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "4", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d4 = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d4["ranks"]["record_bytes"] == 4 * (8 + 6 * 4096) == 98336 and d4["ranks"]["epochs_complete_and_sorted"] is True
    # a mismatch between --gpus and the launcher's world size is an error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


def test_bench_gpus8_config4_dry_run_with_counts_that_do_not_divide():
    """The shape of the first 8-GPU lease (BASELINE configs[3]: epochs sharded e = rank (mod 8), 98 KB records, one all-gather),
    rehearsed on CPU: 8 gloo ranks, 37 steps + 3 warm-up pairs per rank (neither divides the ten pairs of a launch group), the
    gathered table holds exactly the timed epochs in order. With IM_BENCH_ISOLATE_DEVICES=1 - the documented fallback if the plain
    run faults on a device index other than 0 - every rank narrows HIP_VISIBLE_DEVICES to its own GPU before anything touches the
    HIP runtime (an environment variable read at the top of the rank process, never a re-exec) and then runs as device 0."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HIP_VISIBLE_DEVICES")}
    env["OMP_NUM_THREADS"] = "1"
    for isolate in ("0", "1"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "4", "--steps", "37", "--warmup", "3",
                            "--dry-run"], env=dict(env, IM_BENCH_ISOLATE_DEVICES=isolate), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        rk = d["ranks"]
        assert d["n_gpus"] == 8 and d["steps"] == 37 and rk["world"] == 8 and rk["pairs_per_rank"] == [37] * 8
        assert rk["epochs_in_gathered_table"] == 8 * 37 and rk["epochs_complete_and_sorted"] is True
        assert rk["record_bytes"] == 98336 and rk["gathered_table_bytes"] == 8 * 37 * 98336
        assert rk["devices_isolated"] is (isolate == "1")
        assert rk["hip_visible_device_per_rank"] == (list(range(8)) if isolate == "1" else [-1] * 8)
    # a launcher that already narrowed the list: rank r takes the r-th entry of it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=dict(env, IM_BENCH_ISOLATE_DEVICES="1", HIP_VISIBLE_DEVICES="5,3"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["ranks"]["hip_visible_device_per_rank"] == [5, 3]


def test_ransac_iteration_count_survives_tiny_inlier_ratios():
    """`_needed` (shared by the fundamental- and the essential-matrix RANSAC): w^8 underflows below ~1 % inliers; the count saturates
    instead of dividing by log(1) = 0 (OverflowError once the first batch of hypotheses found nothing)."""
    from icepy4d_amd.matching.geometric_verification import NEEDED_CAP, _needed
    assert _needed(0.9999, 0.0) == NEEDED_CAP and _needed(0.9999, 1e-3) == NEEDED_CAP and _needed(0.9999, 0.005) == NEEDED_CAP
    assert _needed(0.9999, 0.5) == int(np.ceil(np.log(1e-4) / np.log(1 - 0.5 ** 8))) == 2354
    assert _needed(0.9999, 1.0) == 1 and _needed(0.99, 0.2) < NEEDED_CAP
    # the caller's loop `while done < min(needed, max_iters)` then simply runs to max_iters


def test_out_of_scope_names_of_the_reference_exist_and_fail_loudly():
    """`from icepy4d.matching import *` exports LOFTRMatcher (`matchers.py:1005`) and the matcher classes carry `viz_*` methods
    (`:702, 739, 942`): scripts written against the reference must import and run - LoFTR fails at construction with a clear
    message, the drawing methods are accepted and skipped with a warning."""
    import logging
    from icepy4d_amd import matching
    from icepy4d_amd.matching import matchers
    with pytest.raises(NotImplementedError, match="LOFTRMatcher"):
        matching.LOFTRMatcher({})
    for cls, names in ((matchers.ImageMatcherBase, ("viz_matches_mpl", "viz_matches_cv2")), (matchers.SuperGlueMatcher, ("viz_matches",))):
        for n in names:
            assert callable(getattr(cls, n))
    obj = matchers.ImageMatcherBase.__new__(matchers.ImageMatcherBase)
    assert obj.viz_matches_mpl(None, None, None, None, "x.png", hide_fig=True) is None and obj.viz_matches_cv2(None, None) is None


def test_record_width_mismatch_is_an_error_with_a_message():
    """A match table built without the keypoint payload handed to a matcher that writes 98 KB records (or the reverse) raises a
    ValueError that says so (it used to surface as a size mismatch inside `Tensor.copy_`)."""
    from icepy4d_amd import sequence as sq

    class FakeEngine:
        max_kpts = 16
    with pytest.raises(ValueError, match="words wide"):
        sq.write_records(torch.zeros(2, 8 + 3 * 16, dtype=torch.int32), 0, 0, 1, FakeEngine())
    sm = sq.SequenceMatcher.__new__(sq.SequenceMatcher)
    sm.e, sm.P, sm.use_graph = FakeEngine(), 1, False
    sm._rec = sq.new_table(1, 16, "cpu", True)
    sm._pending = [(0, sq.new_table(1, 16, "cpu", False), 0)]
    sm._enqueue = lambda *a, **k: None
    sm._record = lambda *a, **k: None
    sm._inp = torch.zeros(2, 8, 8, dtype=torch.uint8)
    with pytest.raises(ValueError, match="with_keypoints"):
        sm._run_group()


def test_margin_extractor_explains_perturbed_decisions():
    """tests/margins.py on the CPU: perturb the oracle's score map by +-d, redo the oracle's integer stages on the perturbed
    map, and every keypoint that changed must be traced to an oracle decision (NMS equality / threshold / top-k cut) with margin
    <= 2d; an unrelated keypoint swap is NOT explained. Same for match indices against arg-max gaps."""
    import margins
    from icepy4d_amd import synthetic
    from oracle import ref_cpu as o
    sd = synthetic.superpoint_state_dict(0)
    img = synthetic.band_limited_noise(np.random.default_rng(3), 240, 320)
    with torch.inference_mode():
        tr = {}
        ref = o.superpoint_lg(o.frame_to_tensor(img), sd, 300, trace=tr)
    S, N = tr["score_map"][0], tr["nms"][0]
    ref_kp = ref["keypoints"].numpy()
    n_diff = 0
    for seed, d in enumerate((1e-5, 1e-4, 1e-4)):
        g = torch.Generator().manual_seed(seed)
        Sp = S + (torch.rand(S.shape, generator=g) * 2 - 1) * d
        kp, _ = o.select_keypoints_lg(o.simple_nms(Sp[None], 4)[0], 4, 0.0005, 300)
        ex = margins.explain_keypoint_diffs(S, N, kp.numpy(), ref_kp, 4, 4, 0.0005, 300, 2 * d)
        assert ex["unexplained"] == [], ex
        n_diff += ex["n_diff"]
    assert n_diff > 0                                              # the perturbation did flip decisions
    # a keypoint replaced by a far-away non-candidate pixel has no explaining margin at the float tolerance
    fake = ref_kp.copy()
    fake[0] = [7.0, 9.0] if (7.0, 9.0) not in {tuple(p) for p in ref_kp} else [9.0, 7.0]
    ex = margins.explain_keypoint_diffs(S, N, fake, ref_kp, 4, 4, 0.0005, 300, 1e-7)
    assert ex["n_diff"] == 2 and len(ex["unexplained"]) >= 1
    # order: swapping two keypoints whose scores differ by more than eps is unexplained, equal scores are explained
    sc = ref["keypoint_scores"].numpy()
    sw = ref_kp.copy()
    sw[[0, 200]] = sw[[200, 0]]
    assert margins.explain_order_diffs(sw, ref_kp, sc, 1e-7)["unexplained"] != []
    assert margins.explain_order_diffs(ref_kp, ref_kp, sc, 1e-7)["n_moved"] == 0
    # matches: a changed arg-max is explained only when the row / column gap is within eps
    la = torch.full((4, 4), -5.0)
    la[0, 0], la[0, 1] = -0.5, -0.5 - 5e-5
    la[1, 2], la[2, 1] = -0.3, -0.2
    m_ref = np.array([0, 2, 1])
    assert margins.explain_match_diffs(la, np.array([1, 2, 1]), m_ref, 0.1, 1e-4)["unexplained"] == []
    assert margins.explain_match_diffs(la, np.array([0, 0, 1]), m_ref, 0.1, 1e-4)["unexplained"] != []


def test_features_pickle_is_the_references_format(tmp_path):
    """Row f-3: `icepy4d_amd.features_io` writes the pickle `icepy4d.core.Features.save_as_pickle` writes (fixtures
    g7_features_ref*.pkl were produced by the reference's own class from the arrays in g7_features_in.npz): same bytes, and a
    reference-written file reads back to the arrays that went in."""
    from conftest import load_golden, GOLDEN
    from icepy4d_amd import features_io as fio
    g = load_golden("g7_features_in")
    ref = fio.load_features_pickle(os.path.join(GOLDEN, "g7_features_ref.pkl"))
    assert np.array_equal(ref["kpts"], g["kpts"]) and np.array_equal(ref["descr"], g["descr"]) and np.array_equal(ref["scores"], g["scores"])
    assert ref["track_ids"].tolist() == list(range(12)) and ref["epoch"] is None
    out = tmp_path / "features_0.pkl"
    fio.save_features_pickle(out, g["kpts"], g["descr"], g["scores"])
    assert out.read_bytes() == open(os.path.join(GOLDEN, "g7_features_ref.pkl"), "rb").read()
    out2 = tmp_path / "features_ep.pkl"
    fio.save_features_pickle(out2, g["kpts"], g["descr"], g["scores"].reshape(-1, 1), epoch=3)
    assert out2.read_bytes() == open(os.path.join(GOLDEN, "g7_features_ref_epoch.pkl"), "rb").read()
    back = fio.load_features_pickle(out2)
    assert int(back["epoch"]) == 3 and np.array_equal(back["descr"], g["descr"])
    assert "icepy4d.core.features" not in sys.modules            # the stand-in module is only registered while (un)pickling
    # COLMAP-export arithmetic (`io/export2colmap.py:27-88`): rounded unique keypoints, re-indexed pairs
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 50, size=(40, 2)).astype(np.float32)
    b = a + np.float32(3.2)
    a[5] = a[4]
    kp, mt = fio.matches_to_h5_arrays(a, b, "im0.jpg", "im1.jpg")
    m = mt["im0.jpg"]["im1.jpg"]
    assert np.array_equal(kp["im0.jpg"][m[:, 0]], np.round(a)) and np.array_equal(kp["im1.jpg"][m[:, 1]], np.round(b))
    assert len(kp["im0.jpg"]) <= 39 and fio.matches_to_h5_arrays(a[:10], b[:10], "x", "y") == ({}, {})
    txt = tmp_path / "k.txt"
    fio.save_features_txt(txt, g["kpts"])
    assert np.array_equal(np.loadtxt(txt, delimiter=",", skiprows=1), g["kpts"])


def test_geometric_verification_lo_degeneracy_and_magsac():
    """Row f-2 beyond plain RANSAC: (a) MAGSAC ignores the caller's threshold like the reference's fallback call and keeps the
    tight sigma-consensus inliers; (b) a scene dominated by one plane plus a few off-plane points: the degeneracy repair
    (plane-and-parallax) recovers the off-plane inliers that a plane-degenerate F misses; (c) `confidence` changes how many
    hypotheses are drawn."""
    from icepy4d_amd.matching import GeometricVerification, geometric_verification
    gv = sys.modules["icepy4d_amd.matching.geometric_verification"]     # the module (the package exports the function by that name)
    rng = np.random.default_rng(5)
    K = np.array([[900, 0, 400], [0, 900, 300], [0, 0, 1.0]])
    t = np.array([0.6, 0.02, 0.05])

    def project(X, noise):
        p0 = (K @ X.T).T
        p1 = (K @ (X + t).T).T
        p0, p1 = p0[:, :2] / p0[:, 2:], p1[:, :2] / p1[:, 2:]
        return p0, p1 + rng.normal(0, noise, p1.shape)

    # (a) general scene, 0.2 px noise, 25 % gross outliers
    X = np.c_[rng.uniform(-1, 1, 400), rng.uniform(-1, 1, 400), rng.uniform(4, 9, 400)]
    p0, p1 = project(X, 0.2)
    p1[:100] += rng.uniform(15, 40, size=(100, 2))
    from oracle import gv_cpu      # the oracle's numpy RANSAC stands in for the device hypothesis stage (CPU-only test)
    F, m_mag = geometric_verification(p0, p1, GeometricVerification.MAGSAC, threshold=50.0,      # threshold is ignored
                                      hypothesis_fn=gv_cpu.hypothesis_fn(p0, p1, gv.MAGSAC_K * gv.MAGSAC_SIGMA_MAX))
    F2, m_deg = geometric_verification(p0, p1, GeometricVerification.PYDEGENSAC, threshold=1.0, hypothesis_fn=gv_cpu.hypothesis_fn(p0, p1, 1.0))
    assert m_mag[:100].mean() < 0.05 and m_mag[100:].mean() > 0.9 and m_deg[100:].mean() > 0.95
    x0, x1 = np.c_[p0[100:], np.ones(300)], np.c_[p1[100:], np.ones(300)]
    assert np.abs(np.einsum("ni,ij,nj->n", x1, F, x0)).mean() < 2e-3 * np.abs(F).max() * 900
    # (b) 90 % of the points on one plane
    n_pl, n_off = 360, 40
    Xp = np.c_[rng.uniform(-1, 1, n_pl), rng.uniform(-1, 1, n_pl), np.full(n_pl, 6.0)]
    Xo = np.c_[rng.uniform(-1, 1, n_off), rng.uniform(-1, 1, n_off), rng.uniform(3, 9, n_off)]
    p0, p1 = project(np.r_[Xp, Xo], 0.1)
    _, m_on = geometric_verification(p0, p1, GeometricVerification.PYDEGENSAC, threshold=0.7, seed=3, hypothesis_fn=gv_cpu.hypothesis_fn(p0, p1, 0.7))
    assert m_on[:n_pl].mean() > 0.95 and m_on[n_pl:].mean() > 0.8, (m_on[:n_pl].mean(), m_on[n_pl:].mean())
    # (c) confidence drives the hypothesis count
    assert gv._needed(0.9999, 0.5) > gv._needed(0.9, 0.5) > 0 and gv._needed(0.99, 0.9) < 10
    w = gv._magsac_weights(np.array([0.0, 0.5, 1.0, 1.81, 1.83, 5.0]))
    assert (np.diff(w[:4]) < 0).all() and w[4] == 0 and w[5] == 0


def test_state_dict_fingerprint_memo_and_weight_file_cache(tmp_path):
    """A matcher per epoch (`main_dev.py:115-132`) must not re-hash or re-load unchanged weights: the fingerprint of an unchanged
    dict object is memoised, an in-place edit changes it, an equal copy hashes to the same value, and a weights file is loaded
    once per (path, mtime, size)."""
    import time
    import torch
    from icepy4d_amd import engine
    from icepy4d_amd.matching import matchers
    sd = synthetic.lightglue_state_dict(0, "passthrough")
    f1 = engine.state_dict_fingerprint(sd)
    t = time.perf_counter()
    assert engine.state_dict_fingerprint(sd) == f1
    assert time.perf_counter() - t < 0.02                      # memo hit: no bytes hashed
    assert engine.state_dict_fingerprint({k: v.clone() for k, v in sd.items()}) == f1
    k = next(iter(sd))
    sd[k].add_(1.0)
    assert engine.state_dict_fingerprint(sd) != f1
    # a weight set that dies and another one of the same shapes: storage addresses may be reused by the allocator, the memo must
    # not answer with the dead set's hash (entries pin their tensors, so the addresses cannot be handed out again)
    seen = set()
    for seed in range(4):
        tmp = {kk: torch.full_like(v, float(seed)) for kk, v in sd.items()}
        seen.add(engine.state_dict_fingerprint(tmp))
        del tmp
    assert len(seen) == 4
    p = tmp_path / "superpoint_v1.pth"
    torch.save(synthetic.superpoint_state_dict(0), str(p))
    a = matchers._load_state_dict({"weights_dir": str(tmp_path)}, "superpoint", ["superpoint_v1.pth"])
    b = matchers._load_state_dict({"weights_dir": str(tmp_path)}, "superpoint", ["superpoint_v1.pth"])
    assert a is b


def test_estimate_pose_like_the_reference_tests():
    """The reference's own tests of `estimate_pose` (`tests/test_sfm_geometry.py:8-32`, same inputs): None below five matches, a
    pose for five."""
    from icepy4d_amd import sfm
    assert sfm.estimate_pose(np.array([[0, 0], [0, 1]]), np.array([[0, 0], [0, 1]]), np.eye(3), np.eye(3), 0.5, 0.9999) is None
    kpts0 = np.array([[1853, 2632], [2122, 2744], [416, 2867], [1880, 2582], [2100, 2770]]).astype(np.float32)
    kpts1 = np.array([[0, 0], [0, 1], [1, 0], [1, 1], [0.5, 0.5]])
    res = sfm.estimate_pose(kpts0, kpts1, np.eye(3), np.eye(3), 0.5, 0.9999)
    assert res is not None
    R, t, inl = res
    assert R.shape == (3, 3) and t.shape == (3,) and inl.shape == (5,) and inl.dtype == bool
    assert abs(np.linalg.det(R) - 1) < 1e-9 and np.abs(R @ R.T - np.eye(3)).max() < 1e-9


def test_five_point_solver_known_answers():
    """`essential_five_point` (what cv2.findEssentialMat evaluates on a minimal sample): on exact synthetic correspondences the
    true essential matrix is among the solutions to 1e-9, every solution satisfies x1^T E x0 = 0, det E = 0 and
    2 E E^T E = trace(E E^T) E; with 5-7 noisy matches and one gross outlier `estimate_pose` recovers the motion and flags it."""
    from icepy4d_amd import sfm
    rng = np.random.default_rng(11)
    for _ in range(10):
        X = np.c_[rng.uniform(-2, 2, 5), rng.uniform(-1.5, 1.5, 5), rng.uniform(4, 9, 5)]
        ang = rng.uniform(-0.3, 0.3)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        t = rng.normal(size=3)
        x0 = X[:, :2] / X[:, 2:]
        Xc = X @ R.T + t
        x1 = Xc[:, :2] / Xc[:, 2:]
        Es = sfm.essential_five_point(x0, x1)
        assert 1 <= len(Es) <= 10
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Et = tx @ R
        Et /= np.linalg.norm(Et)
        assert min(min(np.abs(E - Et).max(), np.abs(E + Et).max()) for E in Es) < 1e-9
        h0, h1 = np.c_[x0, np.ones(5)], np.c_[x1, np.ones(5)]
        for E in Es:
            assert np.abs(np.einsum("ni,ij,nj->n", h1, E, h0)).max() < 1e-10
            assert abs(np.linalg.det(E)) < 1e-10 and np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max() < 1e-9
    K = np.array([[1000.0, 0, 640], [0, 1000.0, 480], [0, 0, 1]])
    ang = 0.15
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([-1.0, 0.1, 0.2])
    for n, outlier in ((6, False), (7, True)):   # with six matches a single outlier cannot be told apart: any five define a model
        X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(4, 9, n)]
        x0 = (K @ X.T).T
        x0 = x0[:, :2] / x0[:, 2:]
        x1 = (K @ (X @ R.T + t).T).T
        x1 = x1[:, :2] / x1[:, 2:] + rng.normal(0, 0.05, (n, 2))
        if outlier:
            x1[2] += 60.0
        Re, te, inl = sfm.estimate_pose(x0, x1, K, K, 1.0)
        assert inl.sum() == n - int(outlier) and (not outlier or not inl[2])
        assert np.abs(Re - R).max() < 2e-2 and np.abs(te / np.linalg.norm(te) - t / np.linalg.norm(t)).max() < 5e-2


def test_engine_reserve_failure_forgets_the_stale_workspace():
    """A failed workspace growth (out of device memory inside `im_ctx_reserve`, which has released the old workspace by then)
    must not leave the Python side with the old sizes: the next reserve() has to reach the library again, and graphs captured
    against the released buffers must be dropped (engines are shared by every matcher object with equal weights)."""
    from icepy4d_amd.engine import Engine

    class FakeCtx:
        def __init__(self):
            self.calls, self.fail_next = [], False

        def call(self, name, *a):
            self.calls.append((name, a))
            if self.fail_next:
                self.fail_next = False
                raise RuntimeError("im_ctx_reserve: out of device memory (-31)")

    e = object.__new__(Engine)
    e.device = torch.device("cpu")
    e.ctx = FakeCtx()
    e.max_h = e.max_w = e.max_images = e.max_kpts = 0
    e.generation, e._loaded, e.graphs = 0, {}, {}
    e.reserve(64, 64, 2, 128)
    assert (e.max_h, e.max_kpts) == (64, 128) and len(e.ctx.calls) == 1
    e.graphs["k"] = object()
    gen = e.generation
    e.ctx.fail_next = True
    with pytest.raises(RuntimeError, match="out of device memory"):
        e.reserve(64, 64, 2, 1 << 14)
    assert (e.max_h, e.max_w, e.max_images, e.max_kpts) == (0, 0, 0, 0) and e.graphs == {} and e.generation > gen
    e.reserve(32, 32, 2, 64)                        # smaller than the stale sizes: must still call the library
    assert len(e.ctx.calls) == 3 and e.ctx.calls[-1][1] == (32, 32, 2, 64) and e.max_kpts == 64


def test_kernels_keep_their_register_budget(tmp_path):
    """Occupancy is a property of the compiled code, not of the source: the assignment sweeps once needed 284-320 registers (every row
    load a basic block of its own, all of a strip's loads collected at the top of the loop) and ran at one wave per SIMD whatever the
    grid. Compile the HBM / latency-bound kernels' file for gfx950 (no GPU needed) and hold every kernel to the budget its launch
    bounds promise: no scratch spills, `lse_stats` and `best_sweep` <= 168 VGPRs (three waves per SIMD; `lse_stats` was held at 128 = four
    waves until round 5 found WHY it fitted: the compiler reused one 4-register temporary for all 16 row loads of a group and waited for each
    load on its own - one kilobyte in flight per wave). So the listing is also checked for what the budget is for: in each sweep at least
    eight 16-byte row loads are issued back to back before the first wait."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "lg_misc.s"
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "icepy4d_amd", "csrc", "lg_misc.hip"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = {}
    for blk in re.split(r"\n  - \.agpr_count:", out.read_text())[1:]:
        def field(k):
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            return m.group(1) if m else "0"
        seen[field("name")] = (int(field("vgpr_count")), int(field("vgpr_spill_count")))
    assert len(seen) >= 20
    assert all(spill == 0 for _, spill in seen.values()), {k: v for k, v in seen.items() if v[1]}
    lse = [v for k, v in seen.items() if "lse_stats_kernelILb1" in k]
    best = [v for k, v in seen.items() if "best_sweep_kernel" in k]
    assert lse and best
    assert all(v <= 168 for v, _ in lse), lse
    assert all(v <= 168 for v, _ in best), best
    text = out.read_text()
    for sym in ("_ZN2im16lse_stats_kernelILb1EEEvNS_10AssignArgsE", "_ZN2im17best_sweep_kernelILi0ELb1EEEvNS_10AssignArgsE",
                "_ZN2im22col_lse_combine_kernelENS_10AssignArgsE", "_ZN2im23col_best_combine_kernelENS_10AssignArgsE"):
        body = text[text.index(sym + ":"):]
        body = body[:body.index(".Lfunc_end")]
        mem = [l.strip() for l in body.split("\n") if l.strip().startswith(("global_load_dwordx4", "global_load_dwordx2", "s_waitcnt vmcnt"))]
        run = best_run = 0
        for l in mem:
            run = run + 1 if l.startswith("global_load") else 0
            best_run = max(best_run, run)
        assert best_run >= 8, (sym, best_run)      # loads in flight together, not one at a time
    # the Winograd convolution: every instantiation at two waves per SIMD (<= 256 registers) without scratch
    out2 = tmp_path / "conv_wino.s"
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "icepy4d_amd", "csrc", "conv_wino.hip"), "-o", str(out2)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    conv = {}
    for blk in re.split(r"\n  - \.agpr_count:", out2.read_text())[1:]:
        def field2(k):
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            return m.group(1) if m else "0"
        conv[field2("name")] = (int(field2("vgpr_count")), int(field2("vgpr_spill_count")), int(field2("private_segment_fixed_size")))
    # f32-input form: pool / plain x fused / plain x U through registers / LDS, minus the unpooled fused register form (7); BX (round 6, the
    # products on the bf16 matrix cores): plain, pooled, fused + pooled (3)
    assert len(conv) == 10, sorted(conv)
    assert all(v <= 256 and sp == 0 and scratch == 0 for v, sp, scratch in conv.values()), conv
    text = out2.read_text()

    def kernel_body(txt, sym):
        body = txt[txt.index(sym + ":"):]
        return body[:body.index(".Lfunc_end")]

    def hot_block(body, mfma):
        return max(re.split(r"\n\.LBB\d+_\d+:", body), key=lambda b: b.count(mfma))

    # The counted waits: `s_waitcnt vmcnt(N)` in front of a barrier is right only if the compiler emitted exactly the loads the
    # source counts on, in that order. f32 form with U through registers: the two patch transfers of a slab are followed by EIGHT register loads
    # (the next slab's U fragments) before `vmcnt(8)`; nothing else touches vector memory in the loop (no scratch either).
    f32_plain = kernel_body(text, "_ZN2im19conv3x3_wino_kernelILb0ELb0ELb1ELb0EEEvNS_8ConvArgsE")
    assert "scratch_" not in f32_plain and "flat_" not in f32_plain
    lines = [l.strip() for l in f32_plain.split("\n")]
    i8 = [i for i, l in enumerate(lines) if l.startswith("s_waitcnt vmcnt(8)")]
    assert i8, "the counted wait of the register-U form is gone"
    for i in i8:        # walking back from the wait: exactly eight register loads, then the slab's transfers (listing order = program order in the loop)
        back = [l for l in lines[:i] if l.startswith(("buffer_load", "buffer_store", "global_"))][::-1]
        n_reg = next(k for k, l in enumerate(back) if l.endswith(" lds"))
        assert n_reg == 8 and all(l.startswith("buffer_load_dwordx4") for l in back[:8]), back[:12]
    # BX: per 16-channel chunk 48 bf16 MFMAs, 24 fragment loads (8 steps x 3 planes) and - plain layers - 5 patch transfers; the transfers are
    # issued BEFORE the chunk's fragment loads, `vmcnt(3 * X_AHEAD = 9)` in front of the chunk's barrier therefore covers them: at least nine
    # register loads must follow the last transfer
    bx_plain = kernel_body(text, "_ZN2im19conv3x3_wino_kernelILb0ELb0ELb1ELb1EEEvNS_8ConvArgsE")
    hot = hot_block(bx_plain, "v_mfma_f32_32x32x16_bf16")
    assert hot.count("v_mfma_f32_32x32x16_bf16") == 48 and "v_mfma_f32_32x32x2_f32" not in hot and "scratch_" not in bx_plain
    assert sum(1 for l in hot.split("\n") if l.strip().startswith("buffer_load_dwordx4") and not l.rstrip().endswith(" lds")) == 24
    lines = [l.strip() for l in bx_plain.split("\n")]
    code = [(i, l) for i, l in enumerate(lines) if l and not l.startswith((";", ".")) ]       # instructions only (labels and comments dropped)
    i9 = [i for k, (i, l) in enumerate(code) if l.startswith("s_waitcnt vmcnt(9)") and any(x[1].startswith("s_barrier") for x in code[k + 1:k + 10]) and
          not any(x[1].startswith(("v_", "buffer_", "ds_")) for x in code[k + 1:k + 10] if code.index(x) < next(j for j in range(k + 1, k + 10) if code[j][1].startswith("s_barrier")))]
    assert len(i9) == 2, "the counted wait in front of the chunk barrier (peeled first chunk + loop) is gone"
    for i in i9:
        back = [l for l in lines[:i] if l.startswith(("buffer_load", "buffer_store", "global_"))][::-1]
        n_reg = next(k for k, l in enumerate(back) if l.endswith(" lds"))
        assert n_reg >= 9 and all(l.endswith(" lds") for l in back[n_reg:n_reg + 5]), (n_reg, back[:30])   # five patch pieces per wave and chunk
    # the fragment ring really runs ahead: the fragment loads are not each waited for at once (the machine scheduler had sunk every load to its use)
    waits = [int(m) for m in re.findall(r"s_waitcnt vmcnt\((\d+)\)", hot)]
    assert waits and min(waits) >= 6, waits
    # BX2 (opt-in, conv_wino_bx2.hip): every transfer of the persistent loop is an LDS-DMA piece with a counted wait in front of each quarter's
    # barrier - 3 U pieces per wave and quarter (vmcnt(3): the quarter's own pieces may still fly), the wave's patch pieces of the next chunk (each
    # under its own predicate: blocks of their own) between quarters 3 and 4 (vmcnt(8) = 3 + 5 in quarters 4 and 1); nothing else in vector memory
    # inside the chunk loop, no scratch
    out3 = tmp_path / "conv_wino_bx2.s"
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "icepy4d_amd", "csrc", "conv_wino_bx2.hip"), "-o", str(out3)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    t3 = out3.read_text()
    bx2 = {}
    for blk in re.split(r"\n  - \.agpr_count:", t3)[1:]:
        def field4(k):
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            return m.group(1) if m else "0"
        bx2[field4("name")] = (int(field4("vgpr_count")), int(field4("vgpr_spill_count")), int(field4("private_segment_fixed_size")))
    assert len(bx2) == 2 and all(v <= 256 and sp == 0 and scratch == 0 for v, sp, scratch in bx2.values()), bx2
    for sym in ("_ZN2im23conv3x3_wino_bx2_kernelILb0EEEvNS_8ConvArgsEi", "_ZN2im23conv3x3_wino_bx2_kernelILb1EEEvNS_8ConvArgsEi"):
        body = kernel_body(t3, sym)
        assert "scratch_" not in body and "flat_" not in body
        blocks = [b for b in re.split(r"\n\.LBB\d+_\d+:", body) if "v_mfma_f32_32x32x16_bf16" in b]
        assert blocks and sum(b.count("v_mfma_f32_32x32x16_bf16") for b in blocks) % 48 == 0, [b.count("v_mfma_f32_32x32x16_bf16") for b in blocks]
        for b in blocks:
            n = b.count("v_mfma_f32_32x32x16_bf16")
            vm = [l.strip() for l in b.split("\n") if l.strip().startswith(("buffer_", "global_", "scratch_", "flat_"))]
            assert all(l.startswith("buffer_load_dwordx4") and l.endswith(" lds") for l in vm), vm
            w = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", b)]
            assert (n, w) in ((36, [8, 3, 3]), (12, [8])), (n, w)
            assert b.count("s_barrier") == len(w)
    # the kernels on the bf16 matrix cores (round 5): two waves per SIMD (<= 256 registers), no scratch; the attention's main loop must hold its 48
    # bf16 MFMAs apart (the vector work of a tile is dealt over the MFMA slots by hand: at most two MFMAs back to back outside the last PV group),
    # and the product form stages by LDS-DMA (no ds_write in its loop)
    for name, n_kernels in (("attention_bx.hip", 4), ("ffn_fused.hip", 4), ("gemm.hip", 26)):
        o = tmp_path / (name + ".s")
        r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "icepy4d_amd", "csrc", name), "-o", str(o)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        ks = {}
        for blk in re.split(r"\n  - \.agpr_count:", o.read_text())[1:]:
            def field3(k):
                m = re.search(r"\." + k + r":\s+(\S+)", blk)
                return m.group(1) if m else "0"
            ks[field3("name")] = (int(field3("vgpr_count")), int(field3("vgpr_spill_count")), int(field3("private_segment_fixed_size")))
        assert len(ks) == n_kernels, (name, sorted(ks))
        assert all(v <= 256 and sp == 0 and scratch == 0 for v, sp, scratch in ks.values()), (name, ks)
        if name == "ffn_fused.hip":      # the split form is there to have TWO blocks of 8 waves per CU: four waves per SIMD = 128 registers
            assert all(v <= 128 for k, (v, _, _) in ks.items() if "ffn_fused_split_kernel" in k) and sum("ffn_fused_split_kernel" in k for k in ks) == 2, ks
        if name == "attention_bx.hip":
            text = o.read_text()
            sym = "_ZN2im20flash_attn_bx_kernelILb1ELb1EEEvNS_8AttnArgsE"
            body = text[text.index(sym + ":"):]
            body = body[:body.index(".Lfunc_end")]
            assert body.count("v_mfma_f32_32x32x16_bf16") >= 48 and "v_mfma_f32_32x32x2_f32" not in body
            assert body.count("buffer_load_dwordx4") >= 6 and " lds" in body
            # the hand-dealt loop: the basic block with the most bf16 MFMAs holds the 48 of a step; count its longest run of adjacent MFMAs
            blocks = re.split(r"\n\.LBB\d+_\d+:", body)
            hot = max(blocks, key=lambda b: b.count("v_mfma_f32_32x32x16_bf16"))
            assert hot.count("v_mfma_f32_32x32x16_bf16") == 48 and "ds_write" not in hot, hot.count("v_mfma_f32_32x32x16_bf16")
            ops = [l.split()[0] for l in hot.split("\n") if l.strip() and not l.strip().startswith((";", ".", "s_nop", "s_waitcnt"))]
            run = best_run = 0
            for op in ops:
                run = run + 1 if op.startswith("v_mfma") else 0
                best_run = max(best_run, run)
            assert best_run <= 6, best_run         # only the last PV group (nothing left to deal) runs its six back to back
            # the counted wait of the step: `vmcnt(6)` = "this step's six transfers may still fly, everything older has landed" is right only
            # if a step issues EXACTLY six vector-memory operations, all of them LDS-DMA transfers: no register load, no store, no scratch in the loop
            vm = [l.strip() for l in hot.split("\n") if l.strip().startswith(("buffer_", "global_", "scratch_", "flat_"))]
            assert len(vm) == 6 and all(l.startswith("buffer_load_dwordx4") and l.endswith(" lds") for l in vm), vm
            assert "s_waitcnt vmcnt(6)" in body and "scratch_" not in body



def test_bf16_triple_arithmetic_behind_the_matchers_products():
    """The arithmetic of csrc/attention_bx.hip / ffn_fused.hip / gemm.hip (BX), emulated in numpy (bf16 = the upper half of an fp32, round to
    nearest even, as v_cvt_pk_bf16_f32): an fp32 value IS the sum of its three bf16 cuts (8 + 8 + 8 significant bits: exact, not approximate),
    with |x1| <= 2^-8 |x| and |x2| <= 2^-17 |x|; the two subtractions of the cut are exact in fp32; every bf16 x bf16 product is exact in fp32;
    and the six products with i + j <= 2 miss the exact product by at most 2^-24 |a b| - the rounding of ONE fp32 operation - and by 2^-27 in the
    root mean square, where a rounded fp32 product itself sits at 2^-25.2; three products (i + j <= 1) miss by up to 2^-16. How the matrix core
    accumulates them is measured on the device (profiles/r05_bf16x_probe.txt, tests/test_gpu_kernels.py); this pins the part that is arithmetic."""
    rng = np.random.default_rng(5)

    def bf16(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)

    def cut(x):
        x = x.astype(np.float32)
        h = bf16(x)
        r1 = (x - h).astype(np.float32)
        assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))        # exact
        m = bf16(r1)
        r2 = (r1 - m).astype(np.float32)
        assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))       # exact
        return h, m, bf16(r2)

    x = np.concatenate([rng.standard_normal(400000), rng.uniform(0, 256, 200000), rng.standard_normal(100000) * 1e-6]).astype(np.float32)
    y = rng.permutation(x)
    xh, xm, xl = cut(x)
    yh, ym, yl = cut(y)
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    assert np.array_equal(xh.astype(np.float64) + xm.astype(np.float64) + xl.astype(np.float64), x64)   # the three cuts ARE the value
    ax = np.abs(x64)
    assert (np.abs(xm.astype(np.float64)) <= ax * 2.0 ** -8).all() and (np.abs(xl.astype(np.float64)) <= ax * 2.0 ** -17).all()
    for a in (xh, xm, xl):
        for b in (yh, ym, yl):      # 8 x 8 significant bits: exact in fp32
            assert np.array_equal((a * b).astype(np.float64), a.astype(np.float64) * b.astype(np.float64))
    exact = x64 * y64
    scale = np.abs(exact)
    six = sum(a.astype(np.float64) * b.astype(np.float64) for a, b in ((xh, yl), (xl, yh), (xm, ym), (xh, ym), (xm, yh), (xh, yh)))
    three = sum(a.astype(np.float64) * b.astype(np.float64) for a, b in ((xh, ym), (xm, yh), (xh, yh)))
    e6, e3 = np.abs(six - exact) / scale, np.abs(three - exact) / scale
    e32 = np.abs((x * y).astype(np.float32).astype(np.float64) - exact) / scale            # one rounded fp32 product, for scale
    assert e6.max() <= 2.0 ** -24 and np.sqrt((e6 ** 2).mean()) <= 2.0 ** -27, (np.log2(e6.max()), np.log2(np.sqrt((e6 ** 2).mean())))
    assert np.sqrt((e6 ** 2).mean()) < 0.25 * np.sqrt((e32 ** 2).mean())
    assert e3.max() >= 2.0 ** -17       # what the other three products are for

# ------------------------------------------------------------------------------------------- failure isolation
FAKE_SEQUENCE = r'''
import torch
from icepy4d_amd import sequence as sq

class FakeEngine:
    """What SequenceMatcher needs of an engine, on the CPU: the "forward" reads the parked inputs and fails for poisoned pairs
    (first pixel 255); the "record" step writes n_matches = first pixel of image 0."""
    max_kpts = 8
    device = torch.device("cpu")

def fake_matcher(P, bad_group_once=False):
    sm = sq.SequenceMatcher.__new__(sq.SequenceMatcher)
    sm.e, sm.P, sm.use_graph, sm._graph = FakeEngine(), P, False, None
    sm._inp = torch.zeros(2 * P, 4, 4, dtype=torch.uint8)
    sm._rec = sq.new_table(P, 8, "cpu")
    sm._pending, sm.failed, sm._pinned, sm._pin_i = [], [], None, 0
    state = {"seen": None}
    def enqueue(pairs):
        if (pairs[0::2, 0, 0] == 255).any():
            raise RuntimeError("im_superpoint_forward: injected failure (-31)")
        state["seen"] = pairs.clone()
    def record(table, row, epoch, n_pairs):
        for j in range(n_pairs):
            r = table[row + j]
            r[0] = epoch + j; r[1] = 8; r[2] = 8; r[3] = int(state["seen"][2 * j, 0, 0]); r[4] = 9
    sm._enqueue, sm._record = enqueue, record
    return sm
'''


def test_sequence_matcher_isolates_a_failing_pair():
    """A pair whose enqueue raises (library error code, wrong input shape) gets the record {epoch, n0 = n1 = 0, n_matches = -1}, the
    other pairs of its launch group are re-run one by one and keep their results, the sequence goes on (`matchers.py:199-207`,
    `main_dev.py:270-274`: the reference logs and continues), `failed` names the epoch and `pending_epochs` hands it to a resumed run."""
    ns = {}
    exec(FAKE_SEQUENCE, ns)
    sq = ns["sq"]
    for P in (1, 3):
        sm = ns["fake_matcher"](P)
        table = sq.new_table(7, 8, "cpu")
        for ep in range(7):
            pair = torch.full((2, 4, 4), ep + 1, dtype=torch.uint8)
            if ep == 4:
                pair[0, 0, 0] = 255                                   # the forward of this pair fails
            if ep == 5 and P > 1:
                pair = torch.zeros(2, 5, 5, dtype=torch.uint8)        # cannot even be parked in the launch group's input buffer: wrong shape
            elif ep == 5:
                pair[0, 0, 0] = 255                                   # (one pair per direct launch takes any shape the workspace holds)
            sm.match_pair(pair, ep, table, ep)
        sm.flush()
        assert table[:, 0].tolist() == list(range(7))
        assert table[:, 3].tolist() == [1, 2, 3, 4, -1, -1, 7], (P, table[:, 3].tolist())
        assert table[4, 1:3].tolist() == [0, 0] and [e for e, _ in sm.failed] == ([4, 5] if P == 1 else [5, 4])
        assert "injected failure" in dict(sm.failed)[4]
        rec = table.numpy()
        assert sq.pending_epochs(7, rec) == [4, 5]
        assert sq.decode_record(rec[4], 8)["n_matches"] == -1 and len(sq.decode_record(rec[4], 8)["matches0"]) == 0


def test_sequence_matcher_logs_a_group_failure_and_lets_device_errors_through(caplog):
    """(a) the error of a launch group is logged even when the one-by-one retries of its pairs succeed (it used to vanish);
    (b) an error that is not a pair's - out of device memory, a sticky HIP error - is NOT retried pair by pair: it reaches the caller, as
    `matchers.py` does for `IcematchError`."""
    import logging
    ns = {}
    exec(FAKE_SEQUENCE, ns)
    sq = ns["sq"]
    sm = ns["fake_matcher"](2)
    real = sm._enqueue
    state = {"n": 0}

    def flaky(pairs):                       # the first enqueue of a whole group fails, the retries work
        state["n"] += 1
        if state["n"] == 1:
            raise ValueError("transient launch-group failure")
        return real(pairs)
    sm._enqueue = flaky
    table = sq.new_table(2, 8, "cpu")
    with caplog.at_level(logging.ERROR, logger="icepy4d_amd"):
        for ep in range(2):
            sm.match_pair(torch.full((2, 4, 4), ep + 1, dtype=torch.uint8), ep, table, ep)
        sm.flush()
    assert table[:, 3].tolist() == [1, 2] and sm.failed == []
    assert any("failed as a whole" in r.getMessage() and "transient launch-group failure" in r.getMessage() for r in caplog.records)

    sm = ns["fake_matcher"](2)

    def oom(pairs):
        raise RuntimeError("im_superpoint_forward failed (-11): out of device memory")
    sm._enqueue = oom
    table = sq.new_table(2, 8, "cpu")
    sm.match_pair(torch.ones(2, 4, 4, dtype=torch.uint8), 0, table, 0)
    with pytest.raises(RuntimeError, match="out of device memory"):
        sm.match_pair(torch.ones(2, 4, 4, dtype=torch.uint8), 1, table, 1)


ISOLATION_WORKER = FAKE_SEQUENCE + r'''
import os, sys, torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_epochs = 9
mine = sq.shard_epochs(n_epochs, rank, world)
sm = fake_matcher(2)
t = sq.new_table(len(mine), 8, "cpu")
for row, ep in enumerate(mine):
    pair = torch.full((2, 4, 4), ep + 1, dtype=torch.uint8)
    if ep == 3:
        pair[0, 0, 0] = 255             # rank 1's second pair fails
    sm.match_pair(pair, ep, t, row)
sm.flush()
full = sq.all_gather_tables(t)          # every rank arrives here, also the one with the failed pair
assert full[:, 0].tolist() == list(range(n_epochs)), full[:, 0].tolist()
assert full[:, 3].tolist() == [1, 2, 3, -1, 5, 6, 7, 8, 9], full[:, 3].tolist()
assert sq.pending_epochs(n_epochs, full.numpy()) == [3]
assert [e for e, _ in sm.failed] == ([3] if rank == 1 else [])
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_failed_pair_on_one_rank_does_not_stall_the_all_gather_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(ISOLATION_WORKER)
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29537", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


def test_bench_dry_run_lists_failed_epochs_and_still_gathers():
    """`bench.py --gpus 2 --dry-run --fail-epochs`: the line carries `failed_epochs`, the gathered table is complete and in order."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "5", "--warmup", "1",
                        "--fail-epochs", "3,6"], capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["failed_epochs"] == [3, 6] and line["ranks"]["epochs_complete_and_sorted"] and line["ranks"]["epochs_in_gathered_table"] == 10
