"""Drop-in for `icepy4d.matching` (reference `src/icepy4d/matching/__init__.py:1-3`): same public names. `LOFTRMatcher` is a name
only (constructing it raises NotImplementedError: out of scope); the `viz_*` methods are accepted and skipped with a warning."""
from .enums import GeometricVerification, Quality, TileSelection  # noqa: F401
from .geometric_verification import geometric_verification  # noqa: F401
from .matchers import (FeaturesBase, ImageMatcherABC, ImageMatcherBase, LightGlueMatcher, LOFTRMatcher,  # noqa: F401
                       SuperGlueMatcher, check_dict_keys, get_engine)
from .tiling import Tiler  # noqa: F401
