"""stainx_amd -- MI355X-native (gfx950) stain normalisation behind stainx's fit/transform API.

Drop-in surface: ``Macenko``, ``Reinhard``, ``HistogramMatching``, ``StainNormalizerTransform``,
``StainNormalizerBase`` (reference src/stainx/__init__.py).  Importing the package never touches
the GPU; the native library is loaded when a backend is first instantiated and its absence raises.
"""
from stainx_amd.base import StainNormalizerBase
from stainx_amd.normalizers import HistogramMatching, Macenko, Reinhard
from stainx_amd.transforms import StainNormalizerTransform

__version__ = "0.1.0"
__all__ = ["HistogramMatching", "Macenko", "Reinhard", "StainNormalizerBase", "StainNormalizerTransform", "__version__"]
