"""``StainNormalizerTransform``: the ``nn.Module`` wrapper for DataLoader / torchvision pipelines.

Mirrors reference src/stainx/transforms.py:26-230 (constructor validation matrix :93-140, device
following :173-198, layout checks :200-216, ``reference`` / ``batch`` modes :218-230).  Fitted stain
parameters live on the inner normaliser, not in buffers, so ``state_dict()`` never holds them
(transforms.py:63-67).
"""
from __future__ import annotations

from typing import Any

import torch
import torch.nn as nn

from stainx_amd.normalizers import HistogramMatching, Macenko, Reinhard

_METHODS = {"macenko": Macenko, "reinhard": Reinhard, "histogram_matching": HistogramMatching}
_CHANNELS_FIRST = frozenset({1, -3})
_CHANNELS_LAST = frozenset({-1, 3})
_GPU_BACKENDS = frozenset({"torch_hip", "torch_cuda"})
_FIT_TENSOR_ATTRS = ("_stain_matrix", "_target_max_conc", "_concentration_matrix", "_reference_histogram", "_ref_vals", "_ref_cdf",
                     "_ref_histograms_256", "_reference_mean", "_reference_std")


def _same_layout(a: int, b: int) -> bool:
    return (a in _CHANNELS_FIRST) == (b in _CHANNELS_FIRST)


def _resolved(device) -> torch.device:
    device = torch.device(device)
    if device.type == "cuda" and device.index is None and torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return device


class StainNormalizerTransform(nn.Module):
    """Apply a stain normaliser to CHW / NCHW (histogram matching: also HWC / NHWC) tensors.

    ``mode="reference"`` fits once on ``reference``; ``mode="batch"`` re-fits on
    ``batch[batch_ref_index]`` at every call.  ``device=None`` follows the input tensor's device.
    For Macenko ``normalize_to_0_1`` defaults to True (float [0,1] pipelines).
    """

    def __init__(self, method: str = "macenko", *, mode: str = "reference", reference: torch.Tensor | None = None,
                 device: str | torch.device | None = None, backend: str | None = None, channel_axis: int = 1, batch_ref_index: int = 0,
                 normalize_to_0_1: bool | None = None, normalizer: Any | None = None):
        super().__init__()
        if mode not in ("reference", "batch"):
            raise ValueError(f"Unsupported mode '{mode}'. Use 'reference' or 'batch'.")
        self.mode = mode
        self.channel_axis = channel_axis
        self.batch_ref_index = batch_ref_index
        self.device = None if device is None else torch.device(device)
        self._requested_backend = backend
        if backend in _GPU_BACKENDS and self.device is not None and self.device.type != "cuda":
            raise ValueError(f"backend='{backend}' requires a CUDA device, got {self.device}.")

        explicit = normalize_to_0_1
        if normalizer is not None:
            self.normalizer = normalizer
            if isinstance(normalizer, Macenko):
                if explicit is not None:
                    normalizer.normalize_to_0_1 = bool(explicit)
            elif explicit:
                raise ValueError("normalize_to_0_1 only applies to Macenko normalizers.")
            if isinstance(normalizer, HistogramMatching):
                inner_axis = int(normalizer.channel_axis)
                if channel_axis != 1 and not _same_layout(channel_axis, inner_axis):
                    raise ValueError(f"channel_axis={channel_axis} conflicts with prebuilt HistogramMatching(channel_axis={inner_axis}).")
                self.channel_axis = inner_axis
            elif channel_axis not in _CHANNELS_FIRST:
                raise ValueError(f"channel_axis={channel_axis} is only supported for histogram_matching; Macenko/Reinhard require NCHW (channel_axis=1).")
        else:
            if method not in _METHODS:
                raise ValueError(f"Unknown method '{method}'. Choose from {sorted(_METHODS)}")
            if method != "histogram_matching" and channel_axis not in _CHANNELS_FIRST:
                raise ValueError(f"channel_axis={channel_axis} is only supported for histogram_matching; {method} requires NCHW (channel_axis=1).")
            if explicit and method != "macenko":
                raise ValueError("normalize_to_0_1 only applies to Macenko (method='macenko').")
            start_device = self._initial_device(backend)
            if method == "macenko":
                unit = True if explicit is None else bool(explicit)
                self.normalizer = Macenko(device=start_device, backend=backend, normalize_to_0_1=unit)
            elif method == "histogram_matching":
                self.normalizer = HistogramMatching(device=start_device, backend=backend, channel_axis=channel_axis)
            else:
                self.normalizer = Reinhard(device=start_device, backend=backend)

        if mode == "reference":
            if reference is None and not getattr(self.normalizer, "_is_fitted", False):
                raise ValueError("mode='reference' requires a reference tensor (or a pre-fitted normalizer).")
            if reference is not None:
                self.fit_reference(reference)

    def _initial_device(self, backend: str | None):
        if self.device is not None:
            return self.device
        if backend in _GPU_BACKENDS:
            if not torch.cuda.is_available():
                raise ValueError(f"backend='{backend}' requires a CUDA device; pass device='cuda' or use CUDA input tensors with device=None.")
            return torch.device("cuda")
        return "cpu"      # placeholder until the first tensor shows where the data lives

    def _layout_axis(self) -> int:
        if isinstance(self.normalizer, HistogramMatching):
            return int(self.normalizer.channel_axis)
        return self.channel_axis

    def fit_reference(self, reference: torch.Tensor) -> "StainNormalizerTransform":
        self.normalizer.fit(self._prepare(reference))
        return self

    def _sync_normalizer_device(self, device: torch.device) -> None:
        device = torch.device(device)
        if self._requested_backend in _GPU_BACKENDS and device.type != "cuda":
            raise ValueError(f"backend='{self._requested_backend}' requires CUDA tensors when device=None; got {device}.")
        # compare RESOLVED devices: an index-less "cuda" names the current device (that is where the engine pinned itself), and a
        # tensor on another GPU must move the normaliser there instead of being copied to the engine's GPU behind the caller's back
        engine = self.normalizer._engine
        current = _resolved(engine.device if engine is not None else self.normalizer.device)
        device = _resolved(device)
        if current == device:
            return
        self.normalizer.device = device
        self.normalizer._engine = None
        for name in _FIT_TENSOR_ATTRS:
            value = getattr(self.normalizer, name, None)
            if isinstance(value, torch.Tensor):
                setattr(self.normalizer, name, value.to(device))
            elif isinstance(value, (list, tuple)) and value and all(isinstance(v, torch.Tensor) for v in value):
                setattr(self.normalizer, name, type(value)(v.to(device) for v in value))

    def _prepare(self, images: torch.Tensor) -> torch.Tensor:
        if images.dim() == 3:
            images = images.unsqueeze(0)
        if images.dim() != 4:
            raise ValueError(f"Expected CHW/NCHW or HWC/NHWC image tensor, got shape {tuple(images.shape)}")
        if isinstance(self.normalizer, HistogramMatching) and self._layout_axis() in _CHANNELS_LAST:
            if images.shape[-1] != 3:
                raise ValueError(f"channels-last histogram matching expects shape (N, H, W, 3), got {tuple(images.shape)}")
        elif images.shape[1] != 3:
            raise ValueError(f"Expected NCHW with C=3 (got shape {tuple(images.shape)}). Macenko/Reinhard do not accept NHWC; "
                             "use channel_axis=-1 only with histogram_matching, or permute to NCHW first.")
        target = self.device if self.device is not None else images.device
        self._sync_normalizer_device(target)
        return images.to(target)

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        single = img.dim() == 3
        batch = self._prepare(img)
        if self.mode == "batch":
            idx = self.batch_ref_index
            if idx < 0 or idx >= batch.shape[0]:
                raise IndexError(f"batch_ref_index={idx} out of range for batch size {batch.shape[0]}")
            self.normalizer.fit(batch[idx : idx + 1])
        result = self.normalizer.transform(batch)
        return result.squeeze(0) if single else result
