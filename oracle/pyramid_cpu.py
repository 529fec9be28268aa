"""TEST INFRASTRUCTURE (oracle): numpy restatement of the Gaussian pyramid steps the product runs on the device
(icepy4d_amd/csrc/pyramid.hip); only tests/ may import it.

Gaussian pyramid steps used by `Quality` resizing and tile preselection. The reference calls OpenCV
(`cv2.pyrDown` / `cv2.pyrUp`, `matchers.py:529-530, 599-609`), an un-vendored dependency that is absent
here, so this is a restatement of OpenCV's documented algorithm for 8-bit images: separable 5-tap kernel
[1 4 6 4 1]/16, BORDER_REFLECT_101 (pyrUp: reflected before the first sample, clamped after the last one, as in
OpenCV's `pyrUp_`), fixed-point rounding (sum + 128) >> 8 for pyrDown and (sum + 32) >> 6 for
pyrUp. Parity with a specific OpenCV build is unpinned (no cv2 in the image)."""
import numpy as np

_K = np.array([1, 4, 6, 4, 1], dtype=np.int64)


def _reflect101(idx: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(idx)
    p = 2 * (n - 1)
    idx = np.mod(idx, p)
    return np.where(idx >= n, p - idx, idx)


def pyr_down(img: np.ndarray) -> np.ndarray:
    """uint8 [H, W] or [H, W, C] -> [(H+1)//2, (W+1)//2(, C)]."""
    a = img.astype(np.int64)
    h, w = a.shape[:2]
    oh, ow = (h + 1) // 2, (w + 1) // 2
    ys = _reflect101(2 * np.arange(oh)[:, None] + np.arange(-2, 3)[None, :], h)   # [oh, 5]
    xs = _reflect101(2 * np.arange(ow)[:, None] + np.arange(-2, 3)[None, :], w)   # [ow, 5]
    kshape = (1, 5) + (1,) * (a.ndim - 1)
    rows = (a[ys] * _K.reshape(kshape)).sum(1)                                    # [oh, W(, C)]
    rows = np.moveaxis(rows, 1, 0)                                                # [W, oh(, C)]
    out = (rows[xs] * _K.reshape(kshape)).sum(1)                                  # [ow, oh(, C)]
    out = np.moveaxis(out, 0, 1)
    return ((out + 128) >> 8).astype(np.uint8)


def pyr_up(img: np.ndarray) -> np.ndarray:
    """uint8 [H, W(, C)] -> [2H, 2W(, C)]: zero-insertion upsampling filtered with 4 x the Gaussian kernel."""
    a = img.astype(np.int64)
    h, w = a.shape[:2]

    def up1(x, n):  # along axis 0
        out = np.zeros((2 * n,) + x.shape[1:], dtype=np.int64)
        i = np.arange(n)
        prev = x[_reflect101(i - 1, n)]
        nxt = x[np.minimum(i + 1, n - 1)]     # OpenCV's pyrUp: the sample after the last one is the last one itself
        out[0::2] = prev + 6 * x + nxt        # even samples: [1 6 1] / 8
        out[1::2] = 4 * (x + nxt)             # odd samples:  [4 4] / 8
        return out

    t = up1(a, h)
    t = np.moveaxis(up1(np.moveaxis(t, 1, 0), w), 0, 1)
    return np.clip((t + 32) >> 6, 0, 255).astype(np.uint8)
