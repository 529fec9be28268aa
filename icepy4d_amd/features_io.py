"""Result formats of the reference for the matched features of one epoch (row f-3 of the scope table).

The reference's driver turns the matcher's result properties into one `icepy4d.core.Features` object per camera
(`main_dev.py:160-173` -> `Features.append_features_from_numpy`, `src/icepy4d/core/features.py:362-453`) and persists it with
`Features.save_as_pickle` (`:596-600`) / `save_as_txt` (`:585-594`); `Epoch.save_pickle` (`core/epoch.py:455-473`) pickles the
same objects inside the epoch. This module writes and reads exactly that pickle WITHOUT the reference package being installed:
stand-in classes with the reference's module path, class names, slots and attribute values are registered under
`icepy4d.core.features` for the duration of a dump / load, so a file written here unpickles with the real classes in an
icepy4d installation, and a file written by icepy4d loads here (tests/golden/g7_features_ref.pkl is one).

The COLMAP-style `features_to_h5` of the reference (`io/export2colmap.py:27-88`) refers to undefined names (`images`, `cams`,
`epoch`, `epochdir`) and cannot run there; its h5py dependency is absent from this image. `matches_to_h5_arrays` restates its
arithmetic (rounded, de-duplicated keypoints per image and re-indexed match pairs) and `write_h5` stores them when h5py exists.
"""
from __future__ import annotations

import contextlib
import pickle
import sys
import types
from pathlib import Path
from typing import Dict, Optional, Tuple, Union

import numpy as np

_REF_MODULE = "icepy4d.core.features"


class Feature:
    """Stand-in for `icepy4d.core.features.Feature` (`features.py:72-209`): same slots, same value types."""
    __slots__ = ("_x", "_y", "_track", "_descr", "_score", "epoch")

    def __init__(self, x, y, track_id=None, descr=None, score=None, epoch=None):
        self._x = np.float32(x)
        self._y = np.float32(y)
        self._track = None if track_id is None else np.int32(track_id)
        self._descr = None if descr is None else descr.reshape(-1, 1)
        self._score = None if score is None else np.float32(score)
        self.epoch = None if epoch is None else np.int32(epoch)     # a fresh scalar per feature, as `features.py:151-158` makes


class Features:
    """Stand-in for `icepy4d.core.features.Features` (`features.py:208-230`): `{track_id: Feature}` plus counters."""

    def __init__(self):
        self._values = {}
        self._last_id = -1
        self._iter = 0
        self._descriptor_size = 256

    def __len__(self):
        return len(self._values)


Feature.__module__ = Features.__module__ = _REF_MODULE


@contextlib.contextmanager
def _reference_names():
    """pickle stores classes by (module, name) and checks that the name resolves to the very object: when the reference package
    is not importable, a module object with the stand-ins answers under the reference's module path."""
    created = []
    try:                                                # only the import is guarded: an error raised by the with-body must
        import importlib                                # propagate, not be swallowed here and answered by a second yield
        real = importlib.import_module(_REF_MODULE)     # an icepy4d installation: use its own classes
    except Exception:
        real = None
    if real is not None:
        yield real.Feature, real.Features
        return
    try:
        for name in ("icepy4d", "icepy4d.core", _REF_MODULE):
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
                created.append(name)
        mod = sys.modules[_REF_MODULE]
        mod.Feature, mod.Features = Feature, Features
        yield Feature, Features
    finally:
        for name in created:
            sys.modules.pop(name, None)


def build_features(kpts: np.ndarray, descr: Optional[np.ndarray] = None, scores: Optional[np.ndarray] = None,
                   epoch: Optional[int] = None, classes=(Feature, Features)):
    """`Features().append_features_from_numpy(x, y, descr, scores, epoch=epoch)` (`features.py:362-453`): track ids 0..n-1,
    descr [256 | 128, n], scores [n] or [n, 1]."""
    F, FS = classes
    fs = FS()
    x = np.asarray(kpts[:, 0]).astype(np.float32).flatten()
    y = np.asarray(kpts[:, 1]).astype(np.float32).flatten()
    if not np.any(x):
        return fs                                    # "Empty input feature arrays. Nothing done." (`:385-387`)
    dT = None
    if descr is not None:
        assert descr.shape[0] in (128, 256), "descriptor array must be [128 | 256, n]"
        fs._descriptor_size = descr.shape[0]
        dT = np.asarray(descr.T, dtype=np.float32)
    sc = None if scores is None else np.asarray(scores, dtype=np.float32)
    ep = None
    if epoch is not None:
        ep = np.int32(epoch)
        fs.epoch = ep
    for t_id in range(len(x)):
        s = None
        if sc is not None:
            s = sc[t_id]
            s = s[0] if getattr(s, "shape", ()) == (1,) else s
        fs._values[t_id] = F(x[t_id], y[t_id], track_id=t_id, descr=None if dT is None else dT[t_id], score=s, epoch=ep)
        fs._last_id = t_id
    return fs


def save_features_pickle(path: Union[str, Path], kpts: np.ndarray, descr: Optional[np.ndarray] = None,
                         scores: Optional[np.ndarray] = None, epoch: Optional[int] = None) -> None:
    """`Features.save_as_pickle` (`features.py:596-600`) of the features built from the matcher's arrays."""
    with _reference_names() as classes:
        fs = build_features(kpts, descr, scores, epoch, classes)
        with open(Path(path), "wb") as f:
            pickle.dump(fs, f, protocol=pickle.HIGHEST_PROTOCOL)


def load_features_pickle(path: Union[str, Path]) -> Dict[str, np.ndarray]:
    """A Features pickle (written here or by icepy4d) -> dict(kpts [n, 2], descr [d, n] or None, scores [n] or None,
    track_ids [n], epoch), the array forms of `Features.to_numpy` (`features.py:455-500`)."""
    with _reference_names():
        with open(Path(path), "rb") as f:
            fs = pickle.load(f)
    vals = fs._values
    ids = list(vals.keys())
    kpts = np.array([[vals[i]._x, vals[i]._y] for i in ids], dtype=np.float32).reshape(-1, 2)
    has_d = len(ids) > 0 and vals[ids[0]]._descr is not None
    has_s = len(ids) > 0 and vals[ids[0]]._score is not None
    descr = np.concatenate([vals[i]._descr for i in ids], axis=1).astype(np.float32) if has_d else None
    scores = np.array([vals[i]._score for i in ids], dtype=np.float32) if has_s else None
    return dict(kpts=kpts, descr=descr, scores=scores, track_ids=np.array(ids, dtype=np.int32), epoch=getattr(fs, "epoch", None))


def save_matcher_features(matcher, path0: Union[str, Path], path1: Union[str, Path], epoch: Optional[int] = None) -> None:
    """The `main_dev.py:160-173` step for a matcher object after `match()`: one Features pickle per camera."""
    save_features_pickle(path0, matcher.mkpts0, matcher.descriptors0, matcher.scores0, epoch)
    save_features_pickle(path1, matcher.mkpts1, matcher.descriptors1, matcher.scores1, epoch)


def save_features_txt(path: Union[str, Path], kpts: np.ndarray, fmt: str = "%i", delimiter: str = ",", header: str = "x,y") -> None:
    """`Features.save_as_txt` (`features.py:585-594`)."""
    np.savetxt(path, np.asarray(kpts, dtype=np.float32), fmt=fmt, delimiter=delimiter, newline="\n", header=header)


def matches_to_h5_arrays(mkpts0: np.ndarray, mkpts1: np.ndarray, key0: str, key1: str, min_matches: int = 20
                         ) -> Tuple[Dict[str, np.ndarray], Dict[str, Dict[str, np.ndarray]]]:
    """The arithmetic of the reference's COLMAP export (`io/export2colmap.py:27-88`) for one image pair: keypoints rounded to
    integers and de-duplicated per image (`torch.unique(dim=0, return_inverse=True)` = lexicographic rows), match pairs
    re-indexed into the unique lists. Returns ({image: keypoints [u, 2]}, {image0: {image1: matches [S, 2]}}); empty when there
    are fewer than `min_matches` matches (`MIN_MATCHES = 20`, `:24, 43`)."""
    if len(mkpts0) < min_matches:
        return {}, {}
    kp, inv = {}, {}
    for k, pts in ((key0, mkpts0), (key1, mkpts1)):
        u, r = np.unique(np.round(np.asarray(pts, dtype=np.float32)), axis=0, return_inverse=True)
        kp[k], inv[k] = u, np.asarray(r).reshape(-1)
    matches = np.stack([inv[key0], inv[key1]], 1).astype(np.int64)
    return kp, {key0: {key1: matches}}


def write_h5(output_dir: Union[str, Path], keypoints: Dict[str, np.ndarray], matches: Dict[str, Dict[str, np.ndarray]]) -> None:
    """`keypoints.h5` / `matches.h5` as `export2colmap.py:78-88` lays them out. Needs h5py (not in this image)."""
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError("h5py is not installed: keypoints.h5 / matches.h5 cannot be written (the arrays are available from "
                           "matches_to_h5_arrays)") from e
    output_dir = Path(output_dir)
    with h5py.File(output_dir / "keypoints.h5", mode="w") as f_kp:
        for k, v in keypoints.items():
            f_kp[k] = v
    with h5py.File(output_dir / "matches.h5", mode="w") as f_match:
        for k1, gr in matches.items():
            group = f_match.require_group(k1)
            for k2, m in gr.items():
                group[k2] = m
