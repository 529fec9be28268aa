"""CPU oracle for the learned extraction + matching hot path.   *** TEST INFRASTRUCTURE ***

This file is a from-scratch restatement (torch-CPU fp32 + numpy) of what the reference computes on
its CPU path between `_frame2tensor` and the numpy results of `_match_images`
(reference: `src/icepy4d/matching/matchers.py:1226-1304` and `:892-940`, which call the vendored
`thirdparty/LightGlue/lightglue/{superpoint,lightglue}.py` and `thirdparty/SuperGlue/models/*.py`).
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the
product (`icepy4d_amd/`) never does.

Pinning status: the reference ships **no** golden vectors for this path (`tests/test_matching.py:1-19`
is commented out) and no weights (`.MISSING_LARGE_BLOBS`). The oracle is pinned against outputs of the
reference modules themselves, imported in the build container by `tools/gen_golden.py` with seeded
weights; those outputs are committed under `tests/golden/` and `tests/test_oracle_golden.py` checks
every function here against them (bit-exact: both sides run the same torch-CPU kernels).
Parity against *pretrained* weights is unpinned (none obtainable offline).

Every function cites the reference lines it follows. State dicts use the official key names.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# image -> tensor
# ----------------------------------------------------------------------------------------------

def frame_to_tensor(image: np.ndarray) -> Tensor:
    """`LightGlueMatcher._frame2tensor` (`matchers.py:1212-1220`): HxWxC -> CxHxW (or HxW -> 1xHxW),
    divide by 255 in float64 and round once to float32. Returns [C, H, W]."""
    if image.ndim == 3:
        image = image.transpose(2, 0, 1)
    elif image.ndim == 2:
        image = image[None]
    else:
        raise ValueError(f"Not an image: {image.shape}")
    return torch.tensor(image / 255.0, dtype=torch.float)


def rgb_to_gray(img: Tensor) -> Tensor:
    """kornia.color.rgb_to_grayscale as called by `ImagePreprocessor` (`lightglue/utils.py:35-36`);
    kornia is un-vendored => parity unpinned; evaluation order fixed as (0.299 r + 0.587 g) + 0.114 b."""
    r, g, b = img[..., 0:1, :, :], img[..., 1:2, :, :], img[..., 2:3, :, :]
    return (0.299 * r + 0.587 * g) + 0.114 * b


def rgb_to_gray_u8_cv2(image: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(image, cv2.COLOR_RGB2GRAY) on uint8 [H,W,3] as `SuperGlueMatcher._match_images` calls it
    (`matchers.py:911-914`); OpenCV is un-vendored => parity unpinned; its published fixed-point form with 14
    fractional bits: (4899 R + 9617 G + 1868 B + 8192) >> 14."""
    a = image.astype(np.uint32)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


# ----------------------------------------------------------------------------------------------
# SuperPoint
# ----------------------------------------------------------------------------------------------

def _conv(x: Tensor, sd: SD, name: str, relu: bool = True) -> Tensor:
    w = sd[f"{name}.weight"]
    y = F.conv2d(x, w, sd[f"{name}.bias"], stride=1, padding=w.shape[-1] // 2)
    return F.relu(y) if relu else y


def sp_encoder(image: Tensor, sd: SD) -> Tensor:
    """Shared VGG encoder (`lightglue/superpoint.py:155-165`): [B,1,H,W] -> [B,128,H/8,W/8]."""
    x = _conv(_conv(image, sd, "conv1a"), sd, "conv1b")
    x = F.max_pool2d(x, 2, 2)
    x = _conv(_conv(x, sd, "conv2a"), sd, "conv2b")
    x = F.max_pool2d(x, 2, 2)
    x = _conv(_conv(x, sd, "conv3a"), sd, "conv3b")
    x = F.max_pool2d(x, 2, 2)
    return _conv(_conv(x, sd, "conv4a"), sd, "conv4b")


def sp_score_map(feat: Tensor, sd: SD) -> Tensor:
    """Detector head, 65-way softmax, drop dustbin, depth-to-space (`lightglue/superpoint.py:168-173`):
    channel c of cell (i, j) lands at pixel (8i + c//8, 8j + c%8). -> [B, Hc*8, Wc*8]."""
    logits = _conv(_conv(feat, sd, "convPa"), sd, "convPb", relu=False)
    prob = F.softmax(logits, 1)[:, :64]
    b, _, hc, wc = prob.shape
    prob = prob.permute(0, 2, 3, 1).reshape(b, hc, wc, 8, 8)
    return prob.permute(0, 1, 3, 2, 4).reshape(b, hc * 8, wc * 8)


def simple_nms(scores: Tensor, radius: int) -> Tensor:
    """`simple_nms` (`lightglue/superpoint.py:50-65`, twin `SuperGlue/models/superpoint.py:48-64`):
    exact-equality local maxima with two suppression/recovery rounds; max-pool padding is -inf."""
    def pool(t):
        return F.max_pool2d(t, kernel_size=2 * radius + 1, stride=1, padding=radius)

    zero = torch.zeros_like(scores)
    keep = scores == pool(scores)
    for _ in range(2):
        near_max = pool(keep.float()) > 0
        rest = torch.where(near_max, zero, scores)
        keep = keep | ((rest == pool(rest)) & ~near_max)
    return torch.where(keep, scores, zero)


def select_keypoints_lg(nms: Tensor, border: int, threshold: float, max_k: Optional[int]
                        ) -> Tuple[Tensor, Tensor]:
    """LightGlue-flavour selection on one [H8, W8] NMS map (`lightglue/superpoint.py:177-200`):
    border frame := -1, `> threshold`, row-major candidates, top-k (descending) only when there are
    more than k candidates, (y, x) -> (x, y) float."""
    s = nms.clone()
    if border:
        s[:border] = -1
        s[:, :border] = -1
        s[-border:] = -1
        s[:, -border:] = -1
    ys, xs = torch.where(s > threshold)
    val = s[ys, xs]
    if max_k is not None and max_k < len(val):
        val, idx = torch.topk(val, max_k, dim=0, sorted=True)
        ys, xs = ys[idx], xs[idx]
    return torch.stack([xs, ys], -1).float(), val


def select_keypoints_sg(nms: Tensor, border: int, threshold: float, max_k: int) -> Tuple[Tensor, Tensor]:
    """SuperGlue-flavour selection (`SuperGlue/models/superpoint.py:176-203`): threshold first, then
    the border mask on coordinates, top-k when `max_k >= 0` and fewer than all are wanted."""
    h, w = nms.shape
    yx = torch.nonzero(nms > threshold)
    val = nms[yx[:, 0], yx[:, 1]]
    ok = (yx[:, 0] >= border) & (yx[:, 0] < h - border) & (yx[:, 1] >= border) & (yx[:, 1] < w - border)
    yx, val = yx[ok], val[ok]
    if max_k >= 0 and max_k < len(val):
        val, idx = torch.topk(val, max_k, dim=0)
        yx = yx[idx]
    return torch.flip(yx, [1]).float(), val


def sp_dense_descriptors(feat: Tensor, sd: SD) -> Tensor:
    """Descriptor head + per-cell L2 normalisation (`lightglue/superpoint.py:203-205`)."""
    d = _conv(_conv(feat, sd, "convDa"), sd, "convDb", relu=False)
    return F.normalize(d, p=2, dim=1)


def sample_descriptors(kpts: Tensor, dense: Tensor, s: int = 8) -> Tensor:
    """`sample_descriptors` (`lightglue/superpoint.py:75-87`): kpts [K,2] (x,y), dense [C,Hc,Wc]
    -> [C, K]; bilinear, align_corners=True, zero padding, then L2 normalise."""
    c, hc, wc = dense.shape
    k = kpts - s / 2 + 0.5
    k = k / torch.tensor([wc * s - s / 2 - 0.5, hc * s - s / 2 - 0.5]).to(k)[None]
    k = k * 2 - 1
    out = F.grid_sample(dense[None], k.view(1, 1, -1, 2), mode="bilinear", align_corners=True)
    return F.normalize(out.reshape(1, c, -1), p=2, dim=1)[0]


def superpoint_lg(image: Tensor, sd: SD, max_k: Optional[int], nms_radius: int = 4,
                  threshold: float = 0.0005, border: int = 4, trace: Optional[dict] = None) -> dict:
    """LightGlue-flavour `SuperPoint.extract(img, resize=None)` for one image [C,H,W] or [1,C,H,W]
    (`lightglue/superpoint.py:217-231` + `:146-215`). Returns batch-free tensors:
    keypoints [K,2], keypoint_scores [K], descriptors [K,256], image_size [2] = (W, H)."""
    if image.dim() == 3:
        image = image[None]
    size = torch.tensor([image.shape[-1], image.shape[-2]], dtype=torch.float)
    if image.shape[1] == 3:
        image = rgb_to_gray(image)
    feat = sp_encoder(image, sd)
    smap = sp_score_map(feat, sd)
    nms = simple_nms(smap, nms_radius)
    kpts, kscores = select_keypoints_lg(nms[0], border, threshold, max_k)
    dense = sp_dense_descriptors(feat, sd)
    desc = sample_descriptors(kpts, dense[0]).t().contiguous()
    if trace is not None:
        trace.update(feat=feat, score_map=smap, nms=nms, dense=dense)
    # extract(): kpts = (kpts + .5) / scale - .5 with scale == 1 (exact identity)
    kpts = (kpts + 0.5) / torch.ones(2) - 0.5
    return dict(keypoints=kpts, keypoint_scores=kscores, descriptors=desc, image_size=size)


def superpoint_sg(image: Tensor, sd: SD, nms_radius: int = 3, threshold: float = 0.001,
                  max_k: int = -1, border: int = 4) -> dict:
    """MagicLeap-flavour `SuperPoint.forward` for one image [1,1,H,W]
    (`SuperGlue/models/superpoint.py:151-220`): keypoints [K,2], scores [K], descriptors [256,K]."""
    feat = sp_encoder(image, sd)
    nms = simple_nms(sp_score_map(feat, sd), nms_radius)
    kpts, kscores = select_keypoints_sg(nms[0], border, threshold, max_k)
    dense = sp_dense_descriptors(feat, sd)
    return dict(keypoints=kpts, scores=kscores, descriptors=sample_descriptors(kpts, dense[0]))


# ----------------------------------------------------------------------------------------------
# LightGlue
# ----------------------------------------------------------------------------------------------

LG_HEADS = 4


def lg_normalize_keypoints(kpts: Tensor, size: Tensor) -> Tensor:
    """`normalize_keypoints` (`lightglue/lightglue.py:23-35`): (k - size/2) / (max(size)/2)."""
    size = size.to(kpts)
    return (kpts - (size / 2)[None, :]) / (size.max() / 2)


def lg_posenc(kpts_n: Tensor, sd: SD) -> Tensor:
    """`LearnableFourierPositionalEncoding.forward` (`lightglue/lightglue.py:68-74`): [K,2] -> [2,1,K,64]
    (cos | sin, each frequency duplicated pairwise)."""
    proj = F.linear(kpts_n, sd["posenc.Wr.weight"])
    emb = torch.stack([torch.cos(proj), torch.sin(proj)], 0).unsqueeze(-3)
    return emb.repeat_interleave(2, dim=-1)


def _rot_half(x: Tensor) -> Tensor:
    x = x.unflatten(-1, (-1, 2))
    a, b = x.unbind(dim=-1)
    return torch.stack((-b, a), dim=-1).flatten(start_dim=-2)


def _rope(freqs: Tensor, t: Tensor) -> Tensor:
    """`apply_cached_rotary_emb` (`lightglue/lightglue.py:55-57`)."""
    return (t * freqs[0]) + (_rot_half(t) * freqs[1])


def _ffn(sd: SD, p: str, x: Tensor, msg: Tensor) -> Tensor:
    """`x + ffn(cat[x, msg])` with Linear(512,512) -> LayerNorm(512) -> GELU -> Linear(512,256)
    (`lightglue/lightglue.py:144-149, 162`)."""
    h = F.linear(torch.cat([x, msg], -1), sd[f"{p}.ffn.0.weight"], sd[f"{p}.ffn.0.bias"])
    h = F.layer_norm(h, (h.shape[-1],), sd[f"{p}.ffn.1.weight"], sd[f"{p}.ffn.1.bias"], 1e-5)
    h = F.gelu(h)
    return x + F.linear(h, sd[f"{p}.ffn.3.weight"], sd[f"{p}.ffn.3.bias"])


def lg_self_block(sd: SD, i: int, x: Tensor, enc: Tensor) -> Tensor:
    """`SelfBlock.forward` CPU branch (`lightglue/lightglue.py:151-162` with `Attention.forward`
    `:120-123`): x [1,K,256], enc [2,1,1,K,64]."""
    p = f"transformers.{i}.self_attn"
    qkv = F.linear(x, sd[f"{p}.Wqkv.weight"], sd[f"{p}.Wqkv.bias"])
    qkv = qkv.unflatten(-1, (LG_HEADS, -1, 3)).transpose(1, 2)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q, k = _rope(enc, q), _rope(enc, k)
    ctx = F.scaled_dot_product_attention(q.contiguous(), k.contiguous(), v.contiguous())
    msg = F.linear(ctx.transpose(1, 2).flatten(start_dim=-2), sd[f"{p}.out_proj.weight"], sd[f"{p}.out_proj.bias"])
    return _ffn(sd, p, x, msg)


def lg_cross_block(sd: SD, i: int, x0: Tensor, x1: Tensor) -> Tuple[Tensor, Tensor]:
    """`CrossBlock.forward` CPU branch (`lightglue/lightglue.py:190-216`): one similarity matrix,
    softmax over its rows for image 0 and over its columns for image 1; weights shared."""
    p = f"transformers.{i}.cross_attn"

    def heads(t):
        return t.unflatten(-1, (LG_HEADS, -1)).transpose(1, 2)

    qk0 = heads(F.linear(x0, sd[f"{p}.to_qk.weight"], sd[f"{p}.to_qk.bias"]))
    qk1 = heads(F.linear(x1, sd[f"{p}.to_qk.weight"], sd[f"{p}.to_qk.bias"]))
    v0 = heads(F.linear(x0, sd[f"{p}.to_v.weight"], sd[f"{p}.to_v.bias"]))
    v1 = heads(F.linear(x1, sd[f"{p}.to_v.weight"], sd[f"{p}.to_v.bias"]))
    scale = qk0.shape[-1] ** -0.5
    qk0, qk1 = qk0 * scale ** 0.5, qk1 * scale ** 0.5
    sim = torch.einsum("bhid,bhjd->bhij", qk0, qk1)
    a01 = F.softmax(sim, dim=-1)
    a10 = F.softmax(sim.transpose(-2, -1).contiguous(), dim=-1)
    m0 = torch.einsum("bhij,bhjd->bhid", a01, v1)
    m1 = torch.einsum("bhji,bhjd->bhid", a10.transpose(-2, -1), v0)
    m0 = m0.transpose(1, 2).flatten(start_dim=-2)
    m1 = m1.transpose(1, 2).flatten(start_dim=-2)
    m0 = F.linear(m0, sd[f"{p}.to_out.weight"], sd[f"{p}.to_out.bias"])
    m1 = F.linear(m1, sd[f"{p}.to_out.weight"], sd[f"{p}.to_out.bias"])
    return _ffn(sd, p, x0, m0), _ffn(sd, p, x1, m1)


def lg_log_assignment(sd: SD, i: int, d0: Tensor, d1: Tensor) -> Tuple[Tensor, Tensor]:
    """`MatchAssignment.forward` + `sigmoid_log_double_softmax` (`lightglue/lightglue.py:253-285`):
    returns (scores [1,M+1,N+1], sim [1,M,N])."""
    p = f"log_assignment.{i}"
    md0 = F.linear(d0, sd[f"{p}.final_proj.weight"], sd[f"{p}.final_proj.bias"])
    md1 = F.linear(d1, sd[f"{p}.final_proj.weight"], sd[f"{p}.final_proj.bias"])
    dim = md0.shape[-1]
    md0, md1 = md0 / dim ** 0.25, md1 / dim ** 0.25
    sim = torch.einsum("bmd,bnd->bmn", md0, md1)
    z0 = F.linear(d0, sd[f"{p}.matchability.weight"], sd[f"{p}.matchability.bias"])
    z1 = F.linear(d1, sd[f"{p}.matchability.weight"], sd[f"{p}.matchability.bias"])
    return double_softmax_scores(sim, z0, z1), sim


def double_softmax_scores(sim: Tensor, z0: Tensor, z1: Tensor) -> Tensor:
    """`sigmoid_log_double_softmax` (`lightglue/lightglue.py:253-265`)."""
    b, m, n = sim.shape
    cert = F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    s_row = F.log_softmax(sim, 2)
    s_col = F.log_softmax(sim.transpose(-1, -2).contiguous(), 2).transpose(-1, -2)
    out = sim.new_zeros((b, m + 1, n + 1))
    out[:, :m, :n] = s_row + s_col + cert
    out[:, :-1, -1] = F.logsigmoid(-z0.squeeze(-1))
    out[:, -1, :-1] = F.logsigmoid(-z1.squeeze(-1))
    return out


def mutual_nn_filter(scores: Tensor, th: float):
    """`filter_matches` (`lightglue/lightglue.py:290-306`; same logic `SuperGlue/models/superglue.py:288-298`)
    on the inner [M,N] block of a [1,M+1,N+1] log-assignment. torch.max returns the first index among ties."""
    inner = scores[:, :-1, :-1]
    max0, max1 = inner.max(2), inner.max(1)
    m0, m1 = max0.indices, max1.indices
    i0 = torch.arange(m0.shape[1])[None]
    i1 = torch.arange(m1.shape[1])[None]
    mutual0 = i0 == m1.gather(1, m0)
    mutual1 = i1 == m0.gather(1, m1)
    e0 = max0.values.exp()
    zero = e0.new_tensor(0)
    ms0 = torch.where(mutual0, e0, zero)
    ms1 = torch.where(mutual1, ms0.gather(1, m1), zero)
    valid0 = mutual0 & (ms0 > th)
    valid1 = mutual1 & valid0.gather(1, m1)
    return torch.where(valid0, m0, -1), torch.where(valid1, m1, -1), ms0, ms1


def lightglue(feats0: dict, feats1: dict, sd: SD, depth_confidence: float = 0.95,
              width_confidence: float = 0.99, filter_threshold: float = 0.1, n_layers: int = 9,
              trace: Optional[dict] = None, pruning_min_kpts: int = -1) -> dict:
    """`LightGlue._forward` on the CPU path (`lightglue/lightglue.py:436-556`): pruning threshold is -1
    on CPU (`:326-331`) so pruning is evaluated after every layer but the last (quirk q10);
    `pruning_min_kpts` = 1024 / 1536 restates the CUDA path's `desc.shape[-2] > pruning_th` test (`:495, 503`).
    feats: keypoints [K,2], descriptors [K,256], image_size [2] (batch-free). Returns batch-free tensors."""
    k0, k1 = feats0["keypoints"][None], feats1["keypoints"][None]
    m, n = k0.shape[1], k1.shape[1]
    kn0 = lg_normalize_keypoints(k0[0], feats0["image_size"])[None]
    kn1 = lg_normalize_keypoints(k1[0], feats1["image_size"])[None]
    d0 = feats0["descriptors"][None].contiguous()
    d1 = feats1["descriptors"][None].contiguous()
    e0, e1 = lg_posenc(kn0, sd), lg_posenc(kn1, sd)
    thr = sd.get("confidence_thresholds")
    if thr is None:   # registered buffer computed in `LightGlue.__init__` (`lightglue/lightglue.py:371-373, 558-561`)
        thr = torch.Tensor([float(np.clip(0.8 + 0.1 * np.exp(-4.0 * i / n_layers), 0, 1)) for i in range(n_layers)])
    do_stop = depth_confidence > 0
    do_prune = width_confidence > 0
    ind0, ind1 = torch.arange(m)[None], torch.arange(n)[None]
    prune0, prune1 = torch.ones_like(ind0), torch.ones_like(ind1)
    t0 = t1 = None
    layers: List[dict] = []
    i = 0
    for i in range(n_layers):
        d0 = lg_self_block(sd, i, d0, e0)
        d1 = lg_self_block(sd, i, d1, e1)
        d0, d1 = lg_cross_block(sd, i, d0, d1)
        if trace is not None:
            layers.append(dict(desc0=d0[0].clone(), desc1=d1[0].clone(), ind0=ind0[0].clone(), ind1=ind1[0].clone()))
        if i == n_layers - 1:
            continue
        if do_stop:
            # TokenConfidence (`:77-89`) + check_if_stop (`:571-579`): ratio over the ORIGINAL m+n
            tw, tb = sd[f"token_confidence.{i}.token.0.weight"], sd[f"token_confidence.{i}.token.0.bias"]
            t0 = torch.sigmoid(F.linear(d0, tw, tb)).squeeze(-1)
            t1 = torch.sigmoid(F.linear(d1, tw, tb)).squeeze(-1)
            conf = torch.cat([t0, t1], -1)
            ratio = 1.0 - (conf < thr[i]).float().sum() / (m + n)
            if trace is not None:      # decision margins of this layer (tests/parity_report.py reports them): confidence tests, stop ratio
                layers[-1].update(token_margin=float((conf - thr[i]).abs().min()), stop_margin=float(ratio - depth_confidence))
            if ratio > depth_confidence:
                break
        if do_prune:
            # get_pruning_mask (`:563-569`) + order-preserving compaction (`:495-510`)
            mw, mb = sd[f"log_assignment.{i}.matchability.weight"], sd[f"log_assignment.{i}.matchability.bias"]
            for side in (0, 1):
                d, t = (d0, t0) if side == 0 else (d1, t1)
                if not d.shape[-2] > pruning_min_kpts:
                    continue
                keep = torch.sigmoid(F.linear(d, mw, mb)).squeeze(-1) > (1 - width_confidence)
                if trace is not None:
                    layers[-1][f"match_margin{side}"] = float((torch.sigmoid(F.linear(d, mw, mb)) - (1 - width_confidence)).abs().min())
                if t is not None:
                    keep |= t <= thr[i]
                idx = torch.where(keep)[1]
                if side == 0:
                    ind0, d0, e0 = ind0.index_select(1, idx), d0.index_select(1, idx), e0.index_select(-2, idx)
                    prune0[:, ind0] += 1
                else:
                    ind1, d1, e1 = ind1.index_select(1, idx), d1.index_select(1, idx), e1.index_select(-2, idx)
                    prune1[:, ind1] += 1
    scores, sim = lg_log_assignment(sd, i, d0, d1)
    a0, a1, ms0, ms1 = mutual_nn_filter(scores, filter_threshold)
    valid = a0[0] > -1
    mi0 = torch.where(valid)[0]
    mi1 = a0[0][valid]
    if do_prune:
        pairs = torch.stack([ind0[0, mi0], ind1[0, mi1]], -1)
        f0 = torch.full((1, m), -1, dtype=a0.dtype)
        f1 = torch.full((1, n), -1, dtype=a1.dtype)
        f0[:, ind0] = torch.where(a0 == -1, -1, ind1.gather(1, a0.clamp(min=0)))
        f1[:, ind1] = torch.where(a1 == -1, -1, ind0.gather(1, a1.clamp(min=0)))
        g0, g1 = torch.zeros((1, m)), torch.zeros((1, n))
        g0[:, ind0] = ms0
        g1[:, ind1] = ms1
        mscore = ms0[0][valid]
        a0, a1, ms0, ms1 = f0, f1, g0, g1
    else:
        pairs = torch.stack([mi0, mi1], -1)
        mscore = ms0[0][valid]
        prune0 = torch.ones_like(ms0) * n_layers
        prune1 = torch.ones_like(ms1) * n_layers
    if trace is not None:
        trace.update(layers=layers, log_assignment=scores[0], sim=sim[0], kept0=ind0[0], kept1=ind1[0])
    return dict(matches0=a0[0], matches1=a1[0], matching_scores0=ms0[0], matching_scores1=ms1[0],
                stop=i + 1, matches=pairs, scores=mscore, prune0=prune0[0], prune1=prune1[0])


def match_images_lightglue(image0: np.ndarray, image1: np.ndarray, sp_sd: SD, lg_sd: SD,
                           max_keypoints: int = 10240, **lg_conf):
    """`LightGlueMatcher._match_images` (`matchers.py:1226-1304`) result tuple:
    (kpts0 [K,2], desc0 [256,K], scores0 [K]), (…1), matches0 [K] int64, mconf [S]."""
    with torch.inference_mode():
        f0 = superpoint_lg(frame_to_tensor(image0), sp_sd, max_keypoints)
        f1 = superpoint_lg(frame_to_tensor(image1), sp_sd, max_keypoints)
        out = lightglue(f0, f1, lg_sd, **lg_conf)

    def pack(f):
        return (f["keypoints"].numpy(), f["descriptors"].numpy().T, f["keypoint_scores"].numpy())

    return pack(f0), pack(f1), out["matches0"].numpy(), out["scores"].numpy(), out


# ----------------------------------------------------------------------------------------------
# SuperGlue
# ----------------------------------------------------------------------------------------------

def _conv1d(sd: SD, name: str, x: Tensor) -> Tensor:
    return F.conv1d(x, sd[f"{name}.weight"], sd[f"{name}.bias"])


def _bn(sd: SD, name: str, x: Tensor) -> Tensor:
    return F.batch_norm(x, sd[f"{name}.running_mean"], sd[f"{name}.running_var"],
                        sd[f"{name}.weight"], sd[f"{name}.bias"], training=False, eps=1e-5)


def sg_normalize_keypoints(kpts: Tensor, height: int, width: int) -> Tensor:
    """`normalize_keypoints` (`SuperGlue/models/superglue.py:64-71`): (k - [W,H]/2) / (0.7 max(W,H))."""
    one = kpts.new_tensor(1)
    size = torch.stack([one * width, one * height])[None]
    center = size / 2
    scaling = size.max(1, keepdim=True).values * 0.7
    return (kpts - center[:, None, :]) / scaling[:, None, :]


def sg_keypoint_encoder(sd: SD, kpts_n: Tensor, scores: Tensor) -> Tensor:
    """`KeypointEncoder.forward` (`SuperGlue/models/superglue.py:74-84`): MLP 3->32->64->128->256->256
    (Conv1d k=1 + BatchNorm1d(eval) + ReLU, last layer plain). kpts_n [1,K,2], scores [1,K] -> [1,256,K]."""
    x = torch.cat([kpts_n.transpose(1, 2), scores.unsqueeze(1)], dim=1)
    for j in range(4):
        x = F.relu(_bn(sd, f"kenc.encoder.{3 * j + 1}", _conv1d(sd, f"kenc.encoder.{3 * j}", x)))
    return _conv1d(sd, "kenc.encoder.12", x)


def sg_propagate(sd: SD, l: int, x: Tensor, src: Tensor) -> Tensor:
    """`AttentionalPropagation.forward` (`SuperGlue/models/superglue.py:96-128`): 4-head attention with
    channel c = d*4 + head (`view(b, 64, 4, N)` `:111-114`), softmax over source points, merge conv,
    MLP(512 -> 512 BN ReLU -> 256)."""
    p = f"gnn.layers.{l}"
    b = x.shape[0]
    q = _conv1d(sd, f"{p}.attn.proj.0", x).view(b, 64, 4, -1)
    k = _conv1d(sd, f"{p}.attn.proj.1", src).view(b, 64, 4, -1)
    v = _conv1d(sd, f"{p}.attn.proj.2", src).view(b, 64, 4, -1)
    att = torch.einsum("bdhn,bdhm->bhnm", q, k) / 64 ** 0.5
    prob = F.softmax(att, dim=-1)
    msg = torch.einsum("bhnm,bdhm->bdhn", prob, v)
    msg = _conv1d(sd, f"{p}.attn.merge", msg.contiguous().view(b, 256, -1))
    h = _conv1d(sd, f"{p}.mlp.0", torch.cat([x, msg], dim=1))
    h = F.relu(_bn(sd, f"{p}.mlp.1", h))
    return _conv1d(sd, f"{p}.mlp.3", h)


def log_optimal_transport(scores: Tensor, alpha: Tensor, iters: int) -> Tensor:
    """`log_optimal_transport` + `log_sinkhorn_iterations` (`SuperGlue/models/superglue.py:152-186`)."""
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    ms, ns = (m * one).to(scores), (n * one).to(scores)
    a = alpha.expand(b, 1, 1)
    z = torch.cat([torch.cat([scores, alpha.expand(b, m, 1)], -1),
                   torch.cat([alpha.expand(b, 1, n), a], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])[None].expand(b, -1)
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])[None].expand(b, -1)
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(z + u.unsqueeze(2), dim=1)
    return z + u.unsqueeze(2) + v.unsqueeze(1) - norm


def superglue(data: dict, sd: SD, sinkhorn_iterations: int = 20, match_threshold: float = 0.3,
              n_layers: int = 18, trace: Optional[dict] = None) -> dict:
    """`SuperGlue.forward` (`SuperGlue/models/superglue.py:250-305`). data: keypoints{0,1} [K,2],
    scores{0,1} [K], descriptors{0,1} [256,K], shape{0,1} = (H, W) of the image tensors fed.
    Layer names alternate self/cross; in a cross layer both sides use the pre-update other side."""
    k0, k1 = data["keypoints0"][None], data["keypoints1"][None]
    if k0.shape[1] == 0 or k1.shape[1] == 0:
        return dict(matches0=torch.full(k0.shape[1:2], -1, dtype=torch.int),
                    matches1=torch.full(k1.shape[1:2], -1, dtype=torch.int),
                    matching_scores0=k0.new_zeros(k0.shape[1:2]), matching_scores1=k1.new_zeros(k1.shape[1:2]))
    kn0 = sg_normalize_keypoints(k0, *data["shape0"])
    kn1 = sg_normalize_keypoints(k1, *data["shape1"])
    d0 = data["descriptors0"][None] + sg_keypoint_encoder(sd, kn0, data["scores0"][None])
    d1 = data["descriptors1"][None] + sg_keypoint_encoder(sd, kn1, data["scores1"][None])
    if trace is not None:
        trace["kenc0"], trace["kenc1"] = d0[0].clone(), d1[0].clone()
    for l in range(n_layers):
        s0, s1 = (d1, d0) if l % 2 == 1 else (d0, d1)
        delta0, delta1 = sg_propagate(sd, l, d0, s0), sg_propagate(sd, l, d1, s1)
        d0, d1 = d0 + delta0, d1 + delta1
    md0, md1 = _conv1d(sd, "final_proj", d0), _conv1d(sd, "final_proj", d1)
    sc = torch.einsum("bdn,bdm->bnm", md0, md1) / 256 ** 0.5
    z = log_optimal_transport(sc, sd["bin_score"], sinkhorn_iterations)
    a0, a1, ms0, ms1 = mutual_nn_filter(z, match_threshold)
    if trace is not None:
        trace.update(gnn0=d0[0], gnn1=d1[0], scores=sc[0], ot=z[0])
    return dict(matches0=a0[0], matches1=a1[0], matching_scores0=ms0[0], matching_scores1=ms1[0])


def match_images_superglue(image0: np.ndarray, image1: np.ndarray, sp_sd: SD, sg_sd: SD,
                           nms_radius: int = 3, keypoint_threshold: float = 0.001, max_keypoints: int = -1,
                           sinkhorn_iterations: int = 20, match_threshold: float = 0.3):
    """`SuperGlueMatcher._match_images` (`matchers.py:892-940`) for gray uint8 inputs; note quirk q5:
    mconf = keypoint scores of the valid matches, not match confidences (`matchers.py:936-938`)."""
    if image0.ndim > 2:
        image0 = rgb_to_gray_u8_cv2(image0)
    if image1.ndim > 2:
        image1 = rgb_to_gray_u8_cv2(image1)
    with torch.inference_mode():
        t0 = torch.tensor(image0 / 255.0, dtype=torch.float)[None, None]
        t1 = torch.tensor(image1 / 255.0, dtype=torch.float)[None, None]
        p0 = superpoint_sg(t0, sp_sd, nms_radius, keypoint_threshold, max_keypoints)
        p1 = superpoint_sg(t1, sp_sd, nms_radius, keypoint_threshold, max_keypoints)
        out = superglue(dict(keypoints0=p0["keypoints"], keypoints1=p1["keypoints"], scores0=p0["scores"],
                             scores1=p1["scores"], descriptors0=p0["descriptors"], descriptors1=p1["descriptors"],
                             shape0=t0.shape[-2:], shape1=t1.shape[-2:]), sg_sd, sinkhorn_iterations, match_threshold)
    m0 = out["matches0"].numpy()
    f0 = (p0["keypoints"].numpy(), p0["descriptors"].numpy(), p0["scores"].numpy())
    f1 = (p1["keypoints"].numpy(), p1["descriptors"].numpy(), p1["scores"].numpy())
    return f0, f1, m0, f0[2][m0 > -1], out
