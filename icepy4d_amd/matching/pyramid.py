"""Gaussian pyramid steps used by `Quality` resizing and tile preselection (`matchers.py:529-530, 599-609`: `cv2.pyrDown` /
`cv2.pyrUp` in the reference), run on the device (csrc/pyramid.hip). Host arrays in and out, like the reference's calls; several
levels stay on the device in between. There is no host fallback."""
import numpy as np
import torch

from .. import _lib


def _levels(engine, image: np.ndarray, levels: int, device_image=None) -> np.ndarray:
    """levels > 0: that many pyrDown steps; levels < 0: pyrUp steps. uint8 [H, W] or [H, W, C] (C <= 4). `device_image`: the same image
    already on the engine's device (the tile matcher uploads each image once and shares it with its tile crops)."""
    assert image.dtype == np.uint8 and image.ndim in (2, 3), "pyramid: uint8 [H, W] or [H, W, C] images"
    if levels == 0:
        return image
    c = 1 if image.ndim == 2 else image.shape[2]
    h, w = image.shape[:2]
    with torch.cuda.device(engine.device):
        cur = device_image if device_image is not None else torch.from_numpy(np.ascontiguousarray(image)).to(engine.device)
        assert cur.dtype == torch.uint8 and cur.is_contiguous() and tuple(cur.shape[:2]) == (h, w), "pyramid: device image of another shape"
        for _ in range(abs(levels)):
            if levels > 0:
                oh, ow = (h + 1) // 2, (w + 1) // 2
            else:
                oh, ow = 2 * h, 2 * w
            out = torch.empty((oh, ow, c), dtype=torch.uint8, device=engine.device)
            engine.ctx.call("im_pyr_down" if levels > 0 else "im_pyr_up", _lib.ptr(cur), _lib.ptr(out), 1, h, w, c, engine.stream_ptr())
            cur, h, w = out, oh, ow
        res = cur.cpu().numpy()
    return res[:, :, 0] if image.ndim == 2 else res


def pyr_down(image: np.ndarray, engine, levels: int = 1, device_image=None) -> np.ndarray:
    return _levels(engine, image, levels, device_image)


def pyr_up(image: np.ndarray, engine, levels: int = 1) -> np.ndarray:
    return _levels(engine, image, -levels)
