from stainx_amd.normalizers.histogram_matching import HistogramMatching
from stainx_amd.normalizers.macenko import Macenko
from stainx_amd.normalizers.reinhard import Reinhard

__all__ = ["HistogramMatching", "Macenko", "Reinhard"]
