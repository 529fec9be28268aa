"""Grid tiler with the reference's semantics (`src/icepy4d/matching/tiling.py:93-135`), including quirk q4:
limits are (xmin, ymin, xmax, ymax) with xmax = xmin + DX + overlap - 1 and `extract_patch` slices
[ymin:ymax, xmin:xmax] (Python-exclusive), so every tile loses its last row and column."""
from typing import Dict, List, Tuple

import numpy as np


class Tiler:
    def __init__(self, grid: List[int] = [1, 1], overlap: int = 0, origin: List[int] = [0, 0], max_length: int = 2000):
        self._origin = origin
        self._overlap = overlap
        self._nrow, self._ncol = grid[0], grid[1]
        self._limits = None

    @property
    def grid(self) -> List[int]:
        return [self._nrow, self._ncol]

    @property
    def origin(self) -> List[int]:
        return self._origin

    @property
    def overlap(self) -> int:
        return self._overlap

    @property
    def limits(self) -> Dict[int, tuple]:
        return self._limits

    def compute_limits_by_grid(self, image: np.ndarray) -> Tuple[Dict[int, tuple], List[int]]:
        h, w = image.shape[0], image.shape[1]
        # tile pitch rounded to a multiple of 10 px (`tiling.py:105-106`; Python round = half to even)
        dx = round((w - self._origin[0]) / self._ncol / 10) * 10
        dy = round((h - self._origin[1]) / self._nrow / 10) * 10
        lim = {}
        for col in range(self._ncol):
            for row in range(self._nrow):
                idx = row * self._ncol + col
                xmin = max(self._origin[0], col * dx - self._overlap)
                ymin = max(self._origin[1], row * dy - self._overlap)
                lim[idx] = (xmin, ymin, xmin + dx + self._overlap - 1, ymin + dy + self._overlap - 1)
        self._limits = lim
        return lim, self._origin

    @staticmethod
    def extract_patch(image: np.ndarray, limits) -> np.ndarray:
        return image[limits[1]:limits[3], limits[0]:limits[2]]
