"""Device resolution and channel-layout helpers (mirrors reference src/stainx/utils.py)."""
from __future__ import annotations

from typing import Any

import numpy as np
import torch


def get_device(device: str | Any | None) -> Any:
    """``None`` -> the visible GPU if any, else CPU (utils.py:12-34); strings become ``torch.device``."""
    if device is None:
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if isinstance(device, str):
        return torch.device(device)
    return device


def device_type_of(device: Any) -> str | None:
    if hasattr(device, "type"):
        return device.type
    if isinstance(device, str):
        return device.split(":")[0]
    return None


class ChannelFormatConverter:
    """NHWC <-> NCHW helper with the reference's ``channel_axis`` vocabulary (utils.py:37-100)."""

    _CHANNELS_FIRST = (1, -3)
    _CHANNELS_LAST = (-1, 3)

    def __init__(self, channel_axis: int = 1):
        if channel_axis not in self._CHANNELS_FIRST + self._CHANNELS_LAST:
            raise ValueError(f"Unsupported channel_axis={channel_axis}. Valid values: {sorted(self._CHANNELS_FIRST + self._CHANNELS_LAST)}")
        self.channel_axis = channel_axis
        self.is_channels_first = channel_axis in self._CHANNELS_FIRST
        self.permute_to_hwc = (1, 2, 0) if self.is_channels_first else None

    def to_hwc(self, images: Any, squeeze_batch: bool = False) -> np.ndarray:
        arr = images.detach().cpu().numpy() if isinstance(images, torch.Tensor) else np.asarray(images)
        if squeeze_batch:
            arr = np.squeeze(arr, axis=0)
        return np.transpose(arr, self.permute_to_hwc) if self.permute_to_hwc is not None else arr

    def prepare_for_normalizer(self, images: Any) -> Any:
        """Channels-last input -> NCHW (a lone HWC image gains a batch axis); NCHW passes through."""
        if self.is_channels_first:
            return images
        is_tensor = isinstance(images, torch.Tensor)
        if images.ndim == 4:
            return images.permute(0, 3, 1, 2) if is_tensor else np.transpose(images, (0, 3, 1, 2))
        if images.ndim == 3:
            return images.permute(2, 0, 1).unsqueeze(0) if is_tensor else np.expand_dims(np.transpose(images, (2, 0, 1)), 0)
        raise ValueError(f"prepare_for_normalizer expects 3D or 4D images, got ndim={images.ndim}")
