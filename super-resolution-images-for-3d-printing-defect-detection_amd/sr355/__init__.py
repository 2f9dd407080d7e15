"""sr355 -- host side of the MI355X-native super-resolution + defect-classifier path.

`sr355.runtime` binds libsr355.so (hand-written HIP for gfx950) through ctypes; the reference-shaped
classes live in the sibling `SRModels` package.
"""
from .runtime import Context, Model, Sr355Error  # noqa: F401
