"""Matching option enumerations; names and values follow the reference (`src/icepy4d/matching/enums.py:4-27`)."""
from enum import Enum


class TileSelection(Enum):
    NONE = 0
    EXHAUSTIVE = 1
    GRID = 2
    PRESELECTION = 3


class GeometricVerification(Enum):
    NONE = 1
    PYDEGENSAC = 2
    MAGSAC = 3


class Quality(Enum):
    LOW = 1
    MEDIUM = 2
    HIGH = 3
    HIGHEST = 4
