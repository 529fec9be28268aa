"""Option enumerations of the matcher API.

The member names and integer values are part of the drop-in contract (reference `src/icepy4d/matching/enums.py:4-27`:
user code writes `TileSelection.PRESELECTION`, `Quality.HIGH`, ...), so they are reproduced exactly; the tables below
are the whole specification."""
from enum import Enum

_SPEC = {
    "TileSelection": ("how tile pairs are chosen", ("NONE", 0), ("EXHAUSTIVE", 1), ("GRID", 2), ("PRESELECTION", 3)),
    "GeometricVerification": ("outlier rejection after matching", ("NONE", 1), ("PYDEGENSAC", 2), ("MAGSAC", 3)),
    "Quality": ("image pyramid level the matcher runs on", ("LOW", 1), ("MEDIUM", 2), ("HIGH", 3), ("HIGHEST", 4)),
}


def _make(name: str) -> type:
    doc, *members = _SPEC[name]
    cls = Enum(name, dict(members), module=__name__)
    cls.__doc__ = f"{name}: {doc}."
    return cls


TileSelection = _make("TileSelection")
GeometricVerification = _make("GeometricVerification")
Quality = _make("Quality")
