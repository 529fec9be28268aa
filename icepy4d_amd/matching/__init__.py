"""Drop-in for `icepy4d.matching` (reference `src/icepy4d/matching/__init__.py:1-3`): same public names."""
from .enums import GeometricVerification, Quality, TileSelection  # noqa: F401
from .geometric_verification import geometric_verification  # noqa: F401
from .matchers import (FeaturesBase, ImageMatcherBase, LightGlueMatcher, SuperGlueMatcher,  # noqa: F401
                       check_dict_keys, get_engine)
from .tiling import Tiler  # noqa: F401
