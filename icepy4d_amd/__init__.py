"""icepy4d_amd: MI355X-native implementation of icepy4d's learned feature extraction + matching hot path
(SuperPoint -> LightGlue / SuperGlue) behind the reference's matcher plugin API. See DESIGN.md."""
__version__ = "0.1.0"
