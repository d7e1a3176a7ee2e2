"""Backend selection + ``fit`` / ``transform`` plumbing (mirrors reference normalizers/_template.py).

Backend ids of this package:

``"torch_hip"``   libstainx_hip.so on an MI355X.  ``"torch_cuda"`` is accepted as an alias so code
                  written for the reference's compiled backend drops in unchanged (on ROCm a GPU
                  tensor's ``device.type`` is ``"cuda"``, so the reference's device checks carry over).

There is deliberately no ``"torch"`` id here: the pure-PyTorch path belongs to upstream stainx, and
this package never computes on the CPU.  Requesting it raises ``ValueError``; a missing native
library raises ``ImportError`` (reference _template.py:31-38).
"""
from __future__ import annotations

from typing import Any

from stainx_amd.base import StainNormalizerBase
from stainx_amd.utils import device_type_of

_VALID_BACKENDS = frozenset({"torch_hip", "torch_cuda"})
_CANONICAL = {"torch_hip": "torch_hip", "torch_cuda": "torch_hip"}


class NormalizerTemplate(StainNormalizerBase):
    def __init__(self, device: str | Any | None = None, backend: str | None = None):
        super().__init__(device)
        if backend is not None and backend not in _VALID_BACKENDS:
            hint = " (the pure-PyTorch backend lives in upstream stainx; this package is the HIP backend)" if backend == "torch" else ""
            raise ValueError(f"Unsupported backend '{backend}'. Valid backends: {sorted(_VALID_BACKENDS)}{hint}")
        if backend is not None:
            from stainx_amd.backends.torch_hip_backend import HIP_AVAILABLE

            if not HIP_AVAILABLE:
                raise ImportError(f"Backend '{backend}' requires libstainx_hip.so. Build it with hipcc (see __graft_entry__.build); there is no CPU fallback.")
        self.backend = _CANONICAL[backend] if backend is not None else self._select_backend()
        self._backend_impl = None
        self._init_algorithm_attributes()

    def _init_algorithm_attributes(self):
        """Algorithm-specific fitted attributes; overridden by subclasses."""

    def _select_backend(self) -> str:
        """Only one backend exists; whether it can run is checked when it is instantiated."""
        return "torch_hip"

    def _device_type(self) -> str | None:
        return device_type_of(self.device)

    def _get_backend_impl(self):
        if self._backend_impl is None:
            self._backend_impl = self._get_torch_hip_class()(self.device, **self._get_backend_kwargs())
        return self._backend_impl

    def _get_torch_hip_class(self):
        raise NotImplementedError("Subclasses must implement _get_torch_hip_class")

    # the reference's hook name, kept for code that subclasses its normalisers
    def _get_torch_cuda_class(self):
        return self._get_torch_hip_class()

    def _get_backend_kwargs(self) -> dict:
        return {}

    def fit(self, images: Any) -> "NormalizerTemplate":
        self._compute_reference_params(images)
        self._is_fitted = True
        return self

    def transform(self, images: Any) -> Any:
        if not self._is_fitted:
            raise ValueError("Must call fit() before transform()")
        return self._run_transform(self._get_backend_impl(), images, self._get_reference_params())

    def _run_transform(self, impl, images, params):
        return impl.transform(images, *params)

    def _compute_reference_params(self, images: Any) -> None:
        raise NotImplementedError("Subclasses must implement _compute_reference_params")

    def _get_reference_params(self) -> tuple:
        raise NotImplementedError("Subclasses must implement _get_reference_params")
