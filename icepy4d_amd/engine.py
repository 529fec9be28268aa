"""Thin host wrapper around one `im_ctx` of libicematch: weights in, device buffers, forward calls.

torch is plumbing here (device memory, the current HIP stream); every computation happens in the library.
"""
from __future__ import annotations

import ctypes as C
import hashlib
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import LightGlueConf, SuperGlueConf, ptr


_FP_MEMO: Dict[tuple, tuple] = {}   # cheap key -> (sha1, the tensors themselves)


def state_dict_fingerprint(state_dict: Dict[str, torch.Tensor]) -> str:
    """Identity of a weight set: SHA-1 over names, shapes and fp32 bytes (50 ms for LightGlue's 12 M parameters). Two
    matcher objects built from equal state dicts share device weights; different ones never share a context.

    The reference builds a fresh matcher per epoch (`main_dev.py:115-132`), so the hash of an UNCHANGED dict object is memoised
    under a cheap key - (name, storage address, shape, dtype, in-place version counter) of every tensor - and a per-epoch
    construction costs microseconds; any in-place edit bumps a version counter and the bytes are hashed again. A memo entry keeps
    its tensors alive: a freed storage address could otherwise be handed to a different weight set of the same shapes and hit."""
    keys = [k for k in sorted(state_dict) if not k.endswith("num_batches_tracked")]
    cheap = None
    if not any(state_dict[k].is_inference() for k in keys):     # inference tensors carry no version counter: always hashed
        cheap = tuple((k, state_dict[k].data_ptr(), tuple(state_dict[k].shape), str(state_dict[k].dtype), state_dict[k]._version) for k in keys)
        hit = _FP_MEMO.get(cheap)
        if hit is not None:
            return hit[0]
    h = hashlib.sha1()
    for key in keys:
        t = state_dict[key].detach().cpu().to(torch.float32).contiguous()
        h.update(key.encode())
        h.update(str(tuple(t.shape)).encode())
        h.update(t.numpy().tobytes())
    if cheap is None:
        return h.hexdigest()
    if len(_FP_MEMO) >= 8:              # a handful of weight sets at most (each entry pins its tensors in host memory)
        _FP_MEMO.pop(next(iter(_FP_MEMO)))
    _FP_MEMO[cheap] = (h.hexdigest(), [state_dict[k] for k in keys])
    return _FP_MEMO[cheap][0]


class Engine:
    """One context on one device. Not re-entrant (results live in the engine's buffers until the next call)."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("icepy4d_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.ctx = _lib.Context(device)
        self.max_h = self.max_w = self.max_images = self.max_kpts = 0
        self.generation = 0          # bumped whenever buffers are reallocated OR weights replaced (captured graphs become stale)
        self._loaded: Dict[str, str] = {}     # model -> fingerprint of the weights on the device
        self.graphs: Dict[tuple, object] = {}  # captured whole-pair HIP graphs, keyed by (shape, K, settings, generation)

    def stream_ptr(self) -> int:
        """The current HIP stream of THIS engine's device (not of whatever device is current in the process)."""
        return torch.cuda.current_stream(self.device).cuda_stream

    def synchronize(self) -> None:
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, model: str, state_dict: Dict[str, torch.Tensor]) -> None:
        """Pass a state dict with the official key names (`lightglue/superpoint.py:118-137`,
        `lightglue/lightglue.py:350-373`, `SuperGlue/models/superglue.py:221-242`); the old LightGlue names
        `self_attn.{i}` / `cross_attn.{i}` are remapped like the reference does (`lightglue.py:388-392`).
        Loading the weights that are already on the device is a no-op; replacing them frees the old device copy, so
        `generation` is bumped and every HIP graph captured against the old pointers is dropped by its owner."""
        fp = state_dict_fingerprint(state_dict)
        if self._loaded.get(model) == fp:
            return
        m = model.encode()
        for key, t in state_dict.items():
            if key.endswith("num_batches_tracked"):
                continue
            if model == "lightglue":
                for i in range(9):
                    if key.startswith(f"self_attn.{i}."):
                        key = key.replace(f"self_attn.{i}", f"transformers.{i}.self_attn", 1)
                    elif key.startswith(f"cross_attn.{i}."):
                        key = key.replace(f"cross_attn.{i}", f"transformers.{i}.cross_attn", 1)
            a = np.ascontiguousarray(t.detach().cpu().to(torch.float32).numpy())
            self.ctx.call("im_set_tensor", m, key.encode(), a.ctypes.data, a.size)
        self.ctx.call("im_finalize_weights", m)
        self._loaded[model] = fp
        self.generation += 1

    def holds(self, model: str, fingerprint: str) -> bool:
        """True if `model` is not loaded yet or loaded with exactly these weights (i.e. a matcher with them may share this engine)."""
        return self._loaded.get(model, fingerprint) == fingerprint

    # ------------------------------------------------------------------ workspace
    def reserve(self, max_h: int, max_w: int, max_images: int = 2, max_kpts: int = 4096) -> None:
        """Grow-only: buffers keep the largest sizes seen (row strides are the reserved max_kpts)."""
        if (max_h <= self.max_h and max_w <= self.max_w and max_images <= self.max_images and max_kpts <= self.max_kpts):
            return
        max_h, max_w, max_images = max(max_h, self.max_h), max(max_w, self.max_w), max(max_images, self.max_images)
        max_kpts = max(max_kpts, self.max_kpts)
        try:
            self.ctx.call("im_ctx_reserve", max_h, max_w, max_images, max_kpts)
        except Exception:
            # the library has already released the old workspace (a failed growth, e.g. out of device memory, leaves the context
            # without one): forget the old sizes and buffers, invalidate every captured graph, and let the caller see the error.
            # A later reserve() then really calls the library instead of returning early on stale sizes; the engine stays usable
            # (also for the other matcher objects that share it through matchers.get_engine).
            self.max_h = self.max_w = self.max_images = self.max_kpts = 0
            self.generation += 1
            self.graphs.clear()
            raise
        self.max_h, self.max_w, self.max_images, self.max_kpts = max_h, max_w, max_images, max_kpts
        self.generation += 1
        K, B = max_kpts, max_images
        d = self.device
        self.kpts = torch.zeros(B, K, 2, device=d)
        self.scores = torch.zeros(B, K, device=d)
        self.desc = torch.zeros(B, K, 256, device=d)
        self.n = torch.zeros(B, dtype=torch.int32, device=d)
        NI = 2 * ((B + 1) // 2)          # matcher outputs: rows 2p, 2p + 1 = image 0 / 1 of pair p
        self.matches = torch.zeros(NI, K, dtype=torch.int32, device=d)
        self.mscores = torch.zeros(NI, K, device=d)
        self.prune = torch.zeros(NI, K, dtype=torch.int32, device=d)
        self.info = torch.zeros(NI // 2, 4, dtype=torch.int32, device=d)

    # ------------------------------------------------------------------ forwards (enqueue only)
    def superpoint(self, gray_u8: torch.Tensor, nms_radius: int = 4, threshold: float = 0.0005, border: int = 4,
                   max_kpts: Optional[int] = None, flavour: int = 0, slot: int = 0) -> None:
        """gray_u8: device uint8 [B, H, W] (gray) or [B, H, W, 3] (RGB: converted per pixel inside the first convolution,
        `flavour` picks the reference's conversion). Fills self.kpts / scores / desc / n for images slot .. slot + B - 1
        (two images of different size are two calls with slot 0 and 1)."""
        assert gray_u8.is_cuda and gray_u8.is_contiguous()
        if gray_u8.dtype == torch.float32:           # float gray [B, H, W], already in [0, 1] (the resize path)
            assert gray_u8.dim() == 3
            C_ = 4
        else:
            assert gray_u8.dtype == torch.uint8
            assert gray_u8.dim() == 3 or (gray_u8.dim() == 4 and gray_u8.shape[3] == 3), gray_u8.shape
            C_ = 1 if gray_u8.dim() == 3 else 3
        B, H, W = gray_u8.shape[:3]
        assert 0 <= slot and slot + B <= self.max_images
        k = -1 if max_kpts is None else int(max_kpts)
        self._last_sp = (slot, B)
        self.ctx.call("im_superpoint_forward", ptr(gray_u8), B, H, W, C_, int(nms_radius), float(threshold), int(border), k,
                      int(flavour), ptr(self.kpts[slot:]), ptr(self.scores[slot:]), ptr(self.desc[slot:]), ptr(self.n[slot:]),
                      self.stream_ptr())

    def candidates(self) -> list:
        """Candidate counts (before the top-k / capacity cut) of the images of the last `superpoint` call. Synchronises."""
        slot, B = self._last_sp
        h = np.zeros(B, dtype=np.int32)
        self.ctx.call("im_superpoint_candidates", B, h.ctypes.data, self.stream_ptr())
        return h.tolist()

    def lightglue(self, size0: Tuple[float, float], size1: Tuple[float, float], depth_confidence: float = 0.95,
                  width_confidence: float = 0.99, filter_threshold: float = 0.1, n_layers: int = 9,
                  kpts: Optional[torch.Tensor] = None, desc: Optional[torch.Tensor] = None,
                  n: Optional[torch.Tensor] = None, pruning_min_kpts: int = -1, n_pairs: int = 1) -> None:
        """Matches image 0 and 1 of (kpts, desc, n) (default: the SuperPoint outputs held by the engine).
        size = (W, H). Fills self.matches / mscores / prune / info. pruning_min_kpts: -1 = the reference's CPU-path
        semantics (pruning evaluated after every layer; what the golden vectors pin), 1024 / 1536 = its CUDA path without /
        with FlashAttention (`lightglue.py:326-331`). n_pairs > 1: images 2p, 2p + 1 of the inputs are pair p, all pairs go
        through one sequence of launches (reserve max_images >= 2 n_pairs; every pair has the sizes size0 / size1)."""
        kpts = self.kpts if kpts is None else kpts
        desc = self.desc if desc is None else desc
        n = self.n if n is None else n
        conf = LightGlueConf(float(depth_confidence), float(width_confidence), float(filter_threshold), int(n_layers),
                             int(pruning_min_kpts))
        size = np.array([size0[0], size0[1], size1[0], size1[1]], dtype=np.float32)
        assert 1 <= n_pairs and 2 * n_pairs <= self.matches.shape[0], (n_pairs, self.matches.shape)
        self.ctx.call("im_lightglue_forward_pairs", int(n_pairs), ptr(kpts), ptr(desc), ptr(n), size.ctypes.data, C.byref(conf),
                      ptr(self.matches), ptr(self.mscores), ptr(self.prune), ptr(self.info), self.stream_ptr())

    def superglue(self, shape0: Tuple[int, int], shape1: Tuple[int, int], sinkhorn_iterations: int = 20,
                  match_threshold: float = 0.3, n_layers: int = 18, kpts=None, scores=None, desc=None, n=None) -> None:
        """shape = (H, W) of the image tensors. Fills self.matches / mscores / info."""
        kpts = self.kpts if kpts is None else kpts
        scores = self.scores if scores is None else scores
        desc = self.desc if desc is None else desc
        n = self.n if n is None else n
        conf = SuperGlueConf(int(sinkhorn_iterations), float(match_threshold), int(n_layers))
        shp = np.array([shape0[0], shape0[1], shape1[0], shape1[1]], dtype=np.float32)
        self.ctx.call("im_superglue_forward", ptr(kpts), ptr(scores), ptr(desc), ptr(n), shp.ctypes.data, C.byref(conf),
                      ptr(self.matches), ptr(self.mscores), ptr(self.info), self.stream_ptr())

    # ------------------------------------------------------------------ results to host (synchronises)
    def features_to_host(self, image: int, channels_first: bool = False):
        """(keypoints [n, 2], descriptors [n, 256] - or [256, n], the reference's layout, transposed on the device when
        channels_first - and scores [n])."""
        n = int(self.n[image].item())
        d = self.desc[image, :n]
        if channels_first:
            d = d.T.contiguous()
        return (self.kpts[image, :n].cpu().numpy(), d.cpu().numpy(), self.scores[image, :n].cpu().numpy())

    def matches_to_host(self, n0: int, n1: int, pair: int = 0):
        a = 2 * pair
        m = self.matches[a:a + 2].cpu().numpy().astype(np.int64)
        s = self.mscores[a:a + 2].cpu().numpy()
        p = self.prune[a:a + 2].cpu().numpy()
        info = self.info[pair].cpu().numpy()
        return dict(matches0=m[0, :n0], matches1=m[1, :n1], matching_scores0=s[0, :n0], matching_scores1=s[1, :n1],
                    prune0=p[0, :n0], prune1=p[1, :n1], stop=int(info[0]))

    def pair_to_host(self, pair: int = 0, channels_first: bool = True, pageable: bool = False):
        """Everything a matcher call returns for one pair: the two live counts first (8 bytes, one synchronisation), then ONE round of
        asynchronous device-to-host copies of the LIVE rows only into page-locked buffers and a second synchronisation (the separate
        `features_to_host` / `matches_to_host` calls make about fourteen blocking copies into pageable memory: ~1 ms of a 10.6 ms
        `match()` at 1080p / 4096 keypoints; copying all K reserved rows instead of the live ones moved 21 MB per call at K = 10240, 33 MB
        once a shared engine had grown to 16384, whatever n was). The page-locked buffers come from torch's caching host allocator (sizes in
        power-of-two buckets: no allocation after the first call of a bucket) and are owned by the returned arrays - no aliasing with
        later calls. Returns ((kpts0, desc0, scores0), (kpts1, desc1, scores1), matches dict); descriptors as [256, n] when
        channels_first: a transposed VIEW of the [n, 256] rows, which is also what the reference hands out (`feats['descriptors'].T`,
        `matchers.py:1281-1288`). pageable=True copies the results out of the page-locked buffers (for callers that archive thousands
        of result sets: an array that stays alive keeps its page-locked block)."""
        a = 2 * pair
        host = {}
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device)
            hn = torch.empty(2, dtype=torch.int32, pin_memory=True)
            hn.copy_(self.n[a:a + 2], non_blocking=True)
            stream.synchronize()
            n0, n1 = (int(v) for v in hn.tolist())
            rows = max(n0, n1, 1)
            host["info"] = torch.empty(self.info[pair].shape, dtype=self.info.dtype, pin_memory=True)
            host["info"].copy_(self.info[pair], non_blocking=True)
            for name, t in (("kpts", self.kpts), ("scores", self.scores), ("desc", self.desc), ("matches", self.matches),
                            ("mscores", self.mscores), ("prune", self.prune)):
                live = t[a:a + 2, :rows]
                h = torch.empty(live.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(live, non_blocking=True)
                host[name] = h
            stream.synchronize()
        if pageable:
            host = {k: torch.from_numpy(v.numpy().copy()) for k, v in host.items()}
        feats = []
        for i, n in ((0, n0), (1, n1)):
            d = host["desc"][i, :n].numpy()
            feats.append((host["kpts"][i, :n].numpy(), d.T if channels_first else d, host["scores"][i, :n].numpy()))
        m = host["matches"].numpy().astype(np.int64)
        s, p = host["mscores"].numpy(), host["prune"].numpy()
        out = dict(matches0=m[0, :n0], matches1=m[1, :n1], matching_scores0=s[0, :n0], matching_scores1=s[1, :n1],
                   prune0=p[0, :n0], prune1=p[1, :n1], stop=int(host["info"][0]))
        return feats[0], feats[1], out

    def close(self):
        self.ctx.close()
