"""Drop-in for `icepy4d.matching` (reference `src/icepy4d/matching/__init__.py:1-3`): same public names."""
from .enums import GeometricVerification, Quality, TileSelection  # noqa: F401
from .geometric_verification import geometric_verification  # noqa: F401
from .matchers import (FeaturesBase, ImageMatcherABC, ImageMatcherBase, LightGlueMatcher,  # noqa: F401
                       SuperGlueMatcher, check_dict_keys, get_engine)
from .tiling import Tiler  # noqa: F401
