"""Abstract normaliser interface (mirrors reference src/stainx/base.py:12-61)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any

from stainx_amd.utils import get_device


class StainNormalizerBase(ABC):
    """scikit-learn style ``fit`` / ``transform`` / ``fit_transform`` contract."""

    def __init__(self, device: str | Any | None = None):
        self.device = get_device(device)
        self._is_fitted = False

    @abstractmethod
    def fit(self, images: Any) -> "StainNormalizerBase":
        """Estimate the reference parameters; returns ``self``."""

    @abstractmethod
    def transform(self, images: Any) -> Any:
        """Normalise ``images`` with the fitted parameters."""

    def fit_transform(self, images: Any) -> Any:
        """``fit(images)`` then ``transform(images)`` (base.py:51-61)."""
        return self.fit(images).transform(images)
