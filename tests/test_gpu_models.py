"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors.

Stage-isolated tests feed each integer/compare stage the oracle's own input, so their outputs must be
bit-exact (NMS map, keypoint coordinates and order, match indices). Floating-point stages are compared
within 1e-4 (north_star tolerance; observed ~1e-6)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from icepy4d_amd import synthetic

pytestmark = pytest.mark.gpu

SP_SD = synthetic.superpoint_state_dict(0)


@pytest.fixture(scope="module")
def eng():
    from icepy4d_amd.engine import Engine
    e = Engine(0)
    e.load_state_dict("superpoint", SP_SD)
    e.reserve(256, 320, 2, 512)
    yield e
    e.close()


def oracle():
    from oracle import ref_cpu
    return ref_cpu


def ptrs(*ts):
    from icepy4d_amd._lib import ptr
    return [ptr(t) for t in ts]


# ------------------------------------------------------------------------------------------- SuperPoint stages
@pytest.mark.parametrize("radius", [3, 4])
def test_nms_exact(eng, radius):
    from icepy4d_amd._lib import stream_ptr
    o = oracle()
    g = load_golden("g1_superpoint_b")
    smap = torch.from_numpy(g["score_map"])
    rng = np.random.default_rng(5)
    plateau = torch.from_numpy(np.round(rng.uniform(0, 1, size=smap.shape) * 8).astype(np.float32) / 8)  # many exact ties
    maps = torch.stack([smap, plateau]).contiguous()
    ref = o.simple_nms(maps, radius)
    d_in = maps.cuda()
    d_out = torch.full_like(d_in, float("nan"))
    eng.ctx.call("im_nms", *ptrs(d_in, d_out), 2, maps.shape[1], maps.shape[2], radius, stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(d_out.cpu(), ref)
    gold = g["nms4"] if radius == 4 else g["nms3"]
    assert np.array_equal(d_out[0].cpu().numpy(), gold)


def check_keypoints(kp, sc, ref_kp, ref_sc):
    """Same keypoints in the same order; inside a group of exactly equal scores any order is accepted
    (torch.topk's tie order is unspecified)."""
    assert kp.shape == ref_kp.shape
    assert np.array_equal(sc, ref_sc)
    if np.array_equal(kp, ref_kp):
        return
    for v in np.unique(sc):
        idx = np.where(sc == v)[0]
        a = {tuple(p) for p in kp[idx]}
        b = {tuple(p) for p in ref_kp[idx]}
        assert a == b, f"tie group {v}"


@pytest.mark.parametrize("tag,k", [("a", 64), ("b", 2000), ("b", 100), ("b", 512)])
def test_select_topk_exact(eng, tag, k):
    from icepy4d_amd._lib import stream_ptr
    o = oracle()
    g = load_golden(f"g1_superpoint_{tag}")
    nms = torch.from_numpy(g["nms4"])
    h, w = nms.shape
    d_nms = torch.stack([nms, nms.flip(0)]).contiguous().cuda()
    kk = min(k, 512)
    eng.ctx.call("im_select_topk", *ptrs(d_nms), 2, h, w, 4, 0.0005, kk, *ptrs(eng.kpts, eng.scores, eng.n), stream_ptr())
    torch.cuda.synchronize()
    for b, m in enumerate((nms, nms.flip(0))):
        ref_kp, ref_sc = o.select_keypoints_lg(m, 4, 0.0005, kk)
        n = int(eng.n[b].item())
        assert n == len(ref_sc)
        check_keypoints(eng.kpts[b, :n].cpu().numpy(), eng.scores[b, :n].cpu().numpy(), ref_kp.numpy(), ref_sc.numpy())
    if k == int(g["max_k"]) and k <= 512:
        assert np.array_equal(eng.kpts[0, :int(eng.n[0])].cpu().numpy(), g["keypoints"])


def test_sample_descriptors(eng):
    from icepy4d_amd._lib import stream_ptr
    o = oracle()
    g = load_golden("g1_superpoint_a")
    x = o.frame_to_tensor(g["image"])[None]
    with torch.inference_mode():
        feat = o.sp_encoder(x, SP_SD)
        raw = o._conv(o._conv(feat, SP_SD, "convDa"), SP_SD, "convDb", relu=False)  # before F.normalize
        kp = torch.from_numpy(g["keypoints"])
        ref = o.sample_descriptors(kp, F.normalize(raw, p=2, dim=1)[0]).t()
    n = kp.shape[0]
    eng.kpts.zero_()
    eng.kpts[0, :n] = kp.cuda()
    eng.n[:] = torch.tensor([n, 0], dtype=torch.int32)
    d_raw = raw.permute(0, 2, 3, 1).contiguous().cuda()
    hc, wc = raw.shape[-2:]
    eng.ctx.call("im_sample_descriptors", *ptrs(d_raw), 1, hc, wc, *ptrs(eng.kpts, eng.n, eng.desc), stream_ptr())
    torch.cuda.synchronize()
    err = (eng.desc[0, :n].cpu() - ref).abs().max().item()
    assert err < 1e-5, err
    assert (eng.desc[0, :n].cpu() - torch.from_numpy(g["descriptors"])).abs().max().item() < 1e-5


# ------------------------------------------------------------------------------------------- SuperPoint end to end
@pytest.mark.parametrize("tag", ["a", "b"])
def test_superpoint_end_to_end(eng, tag):
    """Convolutions accumulate in another order than torch-CPU, so score bits differ (~1e-7) and a handful of
    near-tie decisions may flip; require >= 98 % identical keypoints and 1e-4 descriptors on the common ones."""
    g = load_golden(f"g1_superpoint_{tag}")
    img = torch.from_numpy(g["image"])
    k = min(int(g["max_k"]), 512)
    o = oracle()
    with torch.inference_mode():
        ref = o.superpoint_lg(o.frame_to_tensor(g["image"]), SP_SD, k)
    d_img = torch.stack([img, img]).contiguous().cuda()
    eng.superpoint(d_img, 4, 0.0005, 4, k)
    torch.cuda.synchronize()
    kp, desc, sc = eng.features_to_host(0)
    kp1, desc1, sc1 = eng.features_to_host(1)
    assert np.array_equal(kp, kp1) and np.array_equal(desc, desc1)  # batch invariance
    ref_kp = ref["keypoints"].numpy()
    assert kp.shape == ref_kp.shape
    ours = {tuple(p): i for i, p in enumerate(kp)}
    common = [(ours[tuple(p)], j) for j, p in enumerate(ref_kp) if tuple(p) in ours]
    assert len(common) >= 0.98 * len(ref_kp), (len(common), len(ref_kp))
    ii, jj = np.array(common).T
    assert np.abs(sc[ii] - ref["keypoint_scores"].numpy()[jj]).max() < 1e-5
    assert np.abs(desc[ii] - ref["descriptors"].numpy()[jj]).max() < 1e-4


# ------------------------------------------------------------------------------------------- LightGlue
@pytest.fixture(scope="module")
def lg_eng():
    from icepy4d_amd.engine import Engine
    e = Engine(0)
    e.reserve(64, 64, 2, 320)
    yield e
    e.close()


def run_lightglue(e, f, **conf):
    K = e.max_kpts
    m, n = f["kpts0"].shape[0], f["kpts1"].shape[0]
    e.kpts.zero_(); e.desc.zero_()
    e.kpts[0, :m] = torch.from_numpy(f["kpts0"]).cuda(); e.kpts[1, :n] = torch.from_numpy(f["kpts1"]).cuda()
    e.desc[0, :m] = torch.from_numpy(f["desc0"]).cuda(); e.desc[1, :n] = torch.from_numpy(f["desc1"]).cuda()
    e.n[:] = torch.tensor([m, n], dtype=torch.int32)
    e.lightglue(tuple(f["size0"]), tuple(f["size1"]), **conf)
    torch.cuda.synchronize()
    return e.matches_to_host(m, n)


@pytest.mark.parametrize("ci", range(6))
def test_lightglue_golden(lg_eng, ci):
    g = load_golden(f"g2_lightglue_{ci}")
    variant = str(g["variant"])
    lg_eng.load_state_dict("lightglue", synthetic.lightglue_state_dict(0, variant))
    f = synthetic.synthetic_features(int(g["seed"]), int(g["m"]), int(g["n"]))
    wc, dc = float(g["width_confidence"]), float(g["depth_confidence"])
    out = run_lightglue(lg_eng, f, depth_confidence=dc, width_confidence=wc)
    assert out["stop"] == int(g["stop"])
    assert np.array_equal(out["matches0"], g["matches0"])
    assert np.array_equal(out["matches1"], g["matches1"])
    assert np.abs(out["matching_scores0"] - g["matching_scores0"]).max() < 1e-4
    assert np.abs(out["matching_scores1"] - g["matching_scores1"]).max() < 1e-4
    if wc > 0:
        assert np.array_equal(out["prune0"], g["prune0"])
        assert np.array_equal(out["prune1"], g["prune1"])


def test_assign_from_sim_exact(lg_eng):
    """Stage-isolated: the oracle's own similarity matrix in, match indices bit-exact out."""
    from icepy4d_amd._lib import stream_ptr
    o = oracle()
    g = load_golden("g2_lightglue_2")
    sim = torch.from_numpy(g["sim_l0"])
    m, n = sim.shape
    rng = np.random.default_rng(3)
    z0 = torch.from_numpy(rng.normal(2, 2, size=(1, m, 1)).astype(np.float32))
    z1 = torch.from_numpy(rng.normal(2, 2, size=(1, n, 1)).astype(np.float32))
    scores = o.double_softmax_scores(sim[None], z0, z1)
    r0, r1, s0, s1 = o.mutual_nn_filter(scores, 0.1)
    d_sim = sim.cuda()
    d_z0, d_z1 = z0.flatten().cuda(), z1.flatten().cuda()
    dm0 = torch.zeros(m, dtype=torch.int32, device="cuda"); dm1 = torch.zeros(n, dtype=torch.int32, device="cuda")
    ds0 = torch.zeros(m, device="cuda"); ds1 = torch.zeros(n, device="cuda")
    lg_eng.ctx.call("im_assign_from_sim", *ptrs(d_sim), m, n, n, *ptrs(d_z0, d_z1), 0.1, *ptrs(dm0, dm1, ds0, ds1), stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(dm0.cpu().numpy(), r0[0].numpy())
    assert np.array_equal(dm1.cpu().numpy(), r1[0].numpy())
    assert (ds0.cpu() - s0[0]).abs().max().item() < 1e-5
    assert (ds1.cpu() - s1[0]).abs().max().item() < 1e-5
    assert int((r0[0] > -1).sum()) > 20
