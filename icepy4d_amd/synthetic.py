"""Seeded synthetic weights and stereo images.

No pretrained SuperPoint / LightGlue / SuperGlue weights are obtainable offline
(reference: `.MISSING_LARGE_BLOBS:1-3`, `lightglue/superpoint.py:139-140`,
`lightglue/lightglue.py:376-380`), so parity and benchmarks run on seeded
weights that use the *official state-dict key names and shapes*; a user who has
the real ``.pth`` files loads them through the same `weights.pack_*` functions.

Everything here is generated with ``numpy.random.default_rng`` so that the
values do not depend on torch's module-construction order and are identical in
the build container and on the GPU box.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

SP_CONVS = [
    # name, cout, cin, k
    ("conv1a", 64, 1, 3), ("conv1b", 64, 64, 3),
    ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 128, 64, 3), ("conv3b", 128, 128, 3),
    ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 256, 128, 3), ("convPb", 65, 256, 1),
    ("convDa", 256, 128, 3), ("convDb", 256, 256, 1),
]


def _uniform(rng: np.random.Generator, shape, bound: float) -> torch.Tensor:
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


def superpoint_state_dict(seed: int = 0, gain: float = math.sqrt(6.0), calibrate: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded SuperPoint weights with the official key names
    (`lightglue/superpoint.py:118-137`, same names in `SuperGlue/models/superpoint.py:122-140`).

    He-uniform weights (bound = gain / sqrt(fan_in)) keep the activation variance
    through the ReLU stack so that the 65-way detector logits are spread out and
    the score map has well separated local maxima.
    """
    rng = np.random.default_rng(1000 + seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, cout, cin, k in SP_CONVS:
        fan_in = cin * k * k
        sd[f"{name}.weight"] = _uniform(rng, (cout, cin, k, k), gain / math.sqrt(fan_in))
        sd[f"{name}.bias"] = _uniform(rng, (cout,), 0.1)
    if calibrate:
        _center_descriptor_head(sd)
    return sd


def _center_descriptor_head(sd: Dict[str, torch.Tensor]) -> None:
    """Random ReLU features have a large common-mode component that makes all L2-normalised
    descriptors nearly collinear. Remove it by folding the mean convDb response on a fixed
    calibration image into convDb.bias, so that seeded weights give discriminative descriptors
    (end-to-end tests and the benchmark then produce real matches). Weight generation only."""
    import torch.nn.functional as F
    img = band_limited_noise(np.random.default_rng(999), 128, 160)
    x = torch.from_numpy(img.astype(np.float32) / 255.0)[None, None]
    with torch.inference_mode():
        for i, name in enumerate(("conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convDa")):
            x = F.relu(F.conv2d(x, sd[f"{name}.weight"], sd[f"{name}.bias"], padding=1))
            if name in ("conv1b", "conv2b", "conv3b"):
                x = F.max_pool2d(x, 2, 2)
        d = F.conv2d(x, sd["convDb.weight"], sd["convDb.bias"])
        sd["convDb.bias"] = (sd["convDb.bias"] - d.mean(dim=(0, 2, 3))).contiguous()


LG_LAYERS = 9
# `prune_gradual`: a descriptor channel of a unit vector in 256 dimensions is ~N(0, 1/256); sigmoid(400 x + b) > 0.01 (the keep test of
# `get_pruning_mask`, `lightglue.py:563-569`) fails for x < (-4.595 - b) / 400, the token is confident for 400 x + b > logit(~0.9)
PRUNE_GRADUAL_MATCH_BIAS = -1.5       # unmatchable below x = -0.0077: ~45 %
PRUNE_GRADUAL_TOKEN_BIAS = 12.0       # confident above x = -0.0245: ~65 %


def lightglue_state_dict(seed: int = 0, variant: str = "default", channel_stats=None) -> Dict[str, torch.Tensor]:
    """Seeded LightGlue(features='superpoint') weights, official key names
    (`lightglue/lightglue.py:350-373`).

    variants:
      default      nn.Linear-like U(+-1/sqrt(fan_in)); no early stop, no pruning, ~0 matches
      passthrough  last FFN linear x0.01, final_proj = 20*I, matchability bias +4:
                   descriptors survive the transformer, hundreds of mutual matches
      earlystop    passthrough + token-confidence bias +6 from layer 3 on (stop=4)
      prune        passthrough + matchability weight 0 / bias -10 in layers 0..7 and
                   confident tokens from layer 1 on: points get pruned, then
                   depth stop is disabled by the caller (depth_confidence=-1)
      prune_gradual   passthrough + matchability and token confidence that each read ONE descriptor channel (a different
                   one per layer, weight 400): every layer finds ~65 % of the live points confident and ~45 % of them
                   unmatchable, so with the reference's default options (`depth_confidence` 0.95, `width_confidence` 0.99,
                   pruning evaluated after every layer on the CPU path, `lightglue.py:326-331, 495-510`) ~30 % of the live
                   points go per layer - 4096 points walk through (2048, 4096], (1024, 2048] and <= 1024 - until the
                   pruned points (which count as confident, `:571-579`) lift the ratio over 0.95 and the pair stops at
                   layer 6-8 on its own; with depth_confidence=-1 (no tokens: `keep` is the matchability test alone)
                   ~45 % go per layer through all nine layers. `channel_stats` = (mean [256], std [256]) of the descriptors
                   the weights will see (e.g. of the keypoint descriptors of one image; default: random unit vectors,
                   0 and 1 / 16) keeps those rates on descriptors extracted from pixels
      earlystop_late  passthrough + token-confidence bias +6 from layer 6 on: full width, stop = 7
    """
    rng = np.random.default_rng(2000 + seed)
    sd: Dict[str, torch.Tensor] = {}

    def linear(prefix, out_f, in_f):
        b = 1.0 / math.sqrt(in_f)
        sd[f"{prefix}.weight"] = _uniform(rng, (out_f, in_f), b)
        sd[f"{prefix}.bias"] = _uniform(rng, (out_f,), b)

    sd["posenc.Wr.weight"] = torch.from_numpy(rng.normal(0.0, 1.0, size=(32, 2)).astype(np.float32))
    for i in range(LG_LAYERS):
        for blk, projs in (("self_attn", (("Wqkv", 768, 256), ("out_proj", 256, 256))),
                           ("cross_attn", (("to_qk", 256, 256), ("to_v", 256, 256), ("to_out", 256, 256)))):
            p = f"transformers.{i}.{blk}"
            for nm, o, k in projs:
                linear(f"{p}.{nm}", o, k)
            linear(f"{p}.ffn.0", 512, 512)
            sd[f"{p}.ffn.1.weight"] = 1.0 + _uniform(rng, (512,), 0.1)
            sd[f"{p}.ffn.1.bias"] = _uniform(rng, (512,), 0.1)
            linear(f"{p}.ffn.3", 256, 512)
        linear(f"log_assignment.{i}.matchability", 1, 256)
        linear(f"log_assignment.{i}.final_proj", 256, 256)
    for i in range(LG_LAYERS - 1):
        linear(f"token_confidence.{i}.token.0", 1, 256)
    # `confidence_threshold` (`lightglue/lightglue.py:558-561`): float64 then stored in an f32 buffer
    thr = [float(np.clip(0.8 + 0.1 * np.exp(-4.0 * i / LG_LAYERS), 0, 1)) for i in range(LG_LAYERS)]
    sd["confidence_thresholds"] = torch.tensor(thr, dtype=torch.float32)

    if variant != "default":
        for i in range(LG_LAYERS):
            for blk in ("self_attn", "cross_attn"):
                sd[f"transformers.{i}.{blk}.ffn.3.weight"] *= 0.01
                sd[f"transformers.{i}.{blk}.ffn.3.bias"] *= 0.01
            sd[f"log_assignment.{i}.final_proj.weight"] = 20.0 * torch.eye(256)
            sd[f"log_assignment.{i}.final_proj.bias"] = torch.zeros(256)
            sd[f"log_assignment.{i}.matchability.weight"] *= 0.1
            sd[f"log_assignment.{i}.matchability.bias"] = torch.full((1,), 4.0)
    if variant == "earlystop":
        for i in range(3, LG_LAYERS - 1):
            sd[f"token_confidence.{i}.token.0.weight"] *= 0.01
            sd[f"token_confidence.{i}.token.0.bias"] = torch.full((1,), 6.0)
    if variant == "prune":
        for i in range(1, LG_LAYERS - 1):
            sd[f"token_confidence.{i}.token.0.weight"] *= 0.01
            sd[f"token_confidence.{i}.token.0.bias"] = torch.full((1,), 6.0)
        for i in range(LG_LAYERS - 1):
            # matchability depends on the first descriptor channel only -> about half
            # of the points fall below the 0.01 keep threshold
            w = torch.zeros(1, 256)
            w[0, 0] = 400.0
            sd[f"log_assignment.{i}.matchability.weight"] = w
            sd[f"log_assignment.{i}.matchability.bias"] = torch.full((1,), -4.0)
    if variant == "earlystop_late":
        for i in range(6, LG_LAYERS - 1):
            sd[f"token_confidence.{i}.token.0.weight"] *= 0.01
            sd[f"token_confidence.{i}.token.0.bias"] = torch.full((1,), 6.0)
    if variant == "prune_gradual":
        mean, std = channel_stats if channel_stats is not None else (torch.zeros(256), torch.full((256,), 1.0 / 16.0))
        for i in range(LG_LAYERS - 1):
            for key, ch, bias in ((f"log_assignment.{i}.matchability", 2 * i, PRUNE_GRADUAL_MATCH_BIAS),
                                  (f"token_confidence.{i}.token.0", 2 * i + 1, PRUNE_GRADUAL_TOKEN_BIAS)):
                w = torch.zeros(1, 256)
                w[0, ch] = 25.0 / float(std[ch])        # 400 for random unit vectors
                sd[f"{key}.weight"] = w
                sd[f"{key}.bias"] = torch.full((1,), bias - float(w[0, ch]) * float(mean[ch]))
    if variant not in ("default", "passthrough", "earlystop", "prune", "prune_gradual", "earlystop_late"):
        raise ValueError(f"unknown LightGlue weight variant {variant!r}")
    return sd


SG_KENC = [3, 32, 64, 128, 256, 256]
SG_LAYERS = 18


def superglue_state_dict(seed: int = 0, variant: str = "default") -> Dict[str, torch.Tensor]:
    """Seeded SuperGlue weights, official key names (`SuperGlue/models/superglue.py:51-61,
    74-84, 96-149, 221-247`). BatchNorm running statistics are randomised so that the
    eval-mode folding is exercised.

    variant 'passthrough': GNN MLP output x0.01 and final_proj = 20*I so that the input
    descriptors dominate the score matrix and Sinkhorn yields hundreds of matches.
    """
    rng = np.random.default_rng(3000 + seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv1d(prefix, out_c, in_c, zero_bias=False):
        b = 1.0 / math.sqrt(in_c)
        sd[f"{prefix}.weight"] = _uniform(rng, (out_c, in_c, 1), b)
        sd[f"{prefix}.bias"] = torch.zeros(out_c) if zero_bias else _uniform(rng, (out_c,), b)

    def bn(prefix, c):
        sd[f"{prefix}.weight"] = 1.0 + _uniform(rng, (c,), 0.2)
        sd[f"{prefix}.bias"] = _uniform(rng, (c,), 0.1)
        sd[f"{prefix}.running_mean"] = _uniform(rng, (c,), 0.1)
        sd[f"{prefix}.running_var"] = 1.0 + _uniform(rng, (c,), 0.3)
        sd[f"{prefix}.num_batches_tracked"] = torch.tensor(100, dtype=torch.long)

    n = len(SG_KENC)
    idx = 0
    for i in range(1, n):
        conv1d(f"kenc.encoder.{idx}", SG_KENC[i], SG_KENC[i - 1], zero_bias=(i == n - 1))
        idx += 1
        if i < n - 1:
            bn(f"kenc.encoder.{idx}", SG_KENC[i])
            idx += 2  # BatchNorm1d, ReLU
    for l in range(SG_LAYERS):
        p = f"gnn.layers.{l}"
        conv1d(f"{p}.attn.merge", 256, 256)
        for j in range(3):
            conv1d(f"{p}.attn.proj.{j}", 256, 256)
        conv1d(f"{p}.mlp.0", 512, 512)
        bn(f"{p}.mlp.1", 512)
        conv1d(f"{p}.mlp.3", 256, 512, zero_bias=True)
    conv1d("final_proj", 256, 256)
    sd["bin_score"] = torch.tensor(1.0)
    if variant == "passthrough":
        for l in range(SG_LAYERS):
            sd[f"gnn.layers.{l}.mlp.3.weight"] *= 0.01
        sd["final_proj.weight"] = (20.0 * torch.eye(256)).unsqueeze(-1).contiguous()
        sd["final_proj.bias"] = torch.zeros(256)
        sd["kenc.encoder.12.weight"] *= 0.05
    elif variant != "default":
        raise ValueError(f"unknown SuperGlue weight variant {variant!r}")
    return sd


# --------------------------------------------------------------------------------------
# images
# --------------------------------------------------------------------------------------

def _box3(a: np.ndarray) -> np.ndarray:
    p = np.pad(a, 1, mode="edge")
    h, w = a.shape
    s = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            s += p[dy:dy + h, dx:dx + w]
    return s / 9.0


def band_limited_noise(rng: np.random.Generator, h: int, w: int) -> np.ndarray:
    """Uniform u8 noise, 3x3 box-blurred twice, contrast-stretched to 0..255:
    textured everywhere, so the SuperPoint score map has no flat regions."""
    a = rng.integers(0, 256, size=(h, w)).astype(np.float64)
    a = _box3(_box3(a))
    lo, hi = a.min(), a.max()
    return np.clip(np.rint((a - lo) / (hi - lo) * 255.0), 0, 255).astype(np.uint8)


def warp_homography(img: np.ndarray, Hm: np.ndarray) -> np.ndarray:
    """Bilinear inverse warp: out(x, y) = img(Hm @ [x, y, 1]); edge-clamped."""
    h, w = img.shape
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    d = Hm[2, 0] * xs + Hm[2, 1] * ys + Hm[2, 2]
    sx = (Hm[0, 0] * xs + Hm[0, 1] * ys + Hm[0, 2]) / d
    sy = (Hm[1, 0] * xs + Hm[1, 1] * ys + Hm[1, 2]) / d
    sx = np.clip(sx, 0, w - 1.001)
    sy = np.clip(sy, 0, h - 1.001)
    x0 = np.floor(sx).astype(np.int64)
    y0 = np.floor(sy).astype(np.int64)
    fx, fy = sx - x0, sy - y0
    f = img.astype(np.float64)
    out = (f[y0, x0] * (1 - fx) * (1 - fy) + f[y0, x0 + 1] * fx * (1 - fy)
           + f[y0 + 1, x0] * (1 - fx) * fy + f[y0 + 1, x0 + 1] * fx * fy)
    return out


def stereo_pair(epoch: int = 0, h: int = 1080, w: int = 1920) -> Tuple[np.ndarray, np.ndarray]:
    """Synthetic stereo pair of epoch `epoch`: seeds 1234+2e / 1235+2e,
    right image = left warped by a slowly varying homography (about 40 px disparity at 1080p)
    plus N(0, 2) noise. Returns two gray uint8 [h, w] arrays."""
    left = band_limited_noise(np.random.default_rng(1234 + 2 * epoch), h, w)
    t = 0.02 * epoch
    disp = 40.0 * w / 1920.0
    Hm = np.array([[1.0 + 0.01 * math.cos(t), 0.004 * math.sin(t), disp],
                   [-0.003, 1.0 - 0.005 * math.sin(t), 3.0 * math.cos(t)],
                   [2e-6, -1e-6, 1.0]])
    right = warp_homography(left, Hm)
    right = right + np.random.default_rng(1235 + 2 * epoch).normal(0.0, 2.0, size=right.shape)
    return left, np.clip(np.rint(right), 0, 255).astype(np.uint8)


def translated_pair(seed: int = 0, h: int = 240, w: int = 320, dx: int = 40, dy: int = 8,
                    noise: float = 2.0) -> Tuple[np.ndarray, np.ndarray]:
    """Pair related by a pure translation of (dx, dy) pixels, both multiples of 8 so that the
    CNN (three 2x2 pools) is shift-equivariant away from the borders: a keypoint at (x, y) in
    image 0 appears at (x + dx, y + dy) in image 1. Used for end-to-end match-index tests."""
    assert dx % 8 == 0 and dy % 8 == 0 and dx >= 0 and dy >= 0
    base = band_limited_noise(np.random.default_rng(7000 + seed), h + dy, w + dx)
    img0 = base[dy:dy + h, dx:dx + w]
    img1 = base[0:h, 0:w].astype(np.float64)
    img1 = img1 + np.random.default_rng(7500 + seed).normal(0.0, noise, size=img1.shape)
    return np.ascontiguousarray(img0), np.clip(np.rint(img1), 0, 255).astype(np.uint8)


def synthetic_features(seed: int, m: int, n: int, width: int = 640, height: int = 480,
                       overlap: float = 0.7, sigma: float = 0.05, dim: int = 256):
    """Matcher-only fixture input: keypoints uniform in the image (integer
    pixel positions like SuperPoint's), desc0 random unit vectors, desc1 = permuted desc0
    + sigma noise for the first overlap*min(m, n) points, fresh random vectors otherwise."""
    rng = np.random.default_rng(4000 + seed)

    def unit(a):
        return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)

    k0 = np.stack([rng.integers(4, width - 4, size=m), rng.integers(4, height - 4, size=m)], 1).astype(np.float32)
    d0 = unit(rng.normal(size=(m, dim)))
    n_common = int(overlap * min(m, n))
    perm = rng.permutation(n)
    k1 = np.stack([rng.integers(4, width - 4, size=n), rng.integers(4, height - 4, size=n)], 1).astype(np.float32)
    d1 = unit(rng.normal(size=(n, dim)))
    src = rng.permutation(m)[:n_common]
    dst = perm[:n_common]
    d1[dst] = unit(d0[src] + sigma * rng.normal(size=(n_common, dim)))
    k1[dst] = np.clip(k0[src] + rng.integers(-20, 21, size=(n_common, 2)), 4, [width - 5, height - 5]).astype(np.float32)
    s0 = rng.uniform(0.01, 0.5, size=m).astype(np.float32)
    s1 = rng.uniform(0.01, 0.5, size=n).astype(np.float32)
    gt = np.full(m, -1, dtype=np.int64)
    gt[src] = dst
    return dict(kpts0=k0, kpts1=k1, desc0=d0, desc1=d1, scores0=s0, scores1=s1,
                size0=np.array([width, height], np.float32), size1=np.array([width, height], np.float32), gt=gt)
