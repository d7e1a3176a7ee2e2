"""Backend selection and the ``fit`` / ``transform`` plumbing shared by the three normalisers.

The public surface is stainx's (``Normalizer(device=..., backend=...)``, ``fit``, ``transform``, ``fit_transform`` from the
base class; reference: src/stainx/normalizers/_template.py).  The inside is declarative: a normaliser class states

* ``engine``          -- name of its backend class in ``stainx_amd.backends.torch_hip_backend``,
* ``fitted_slots``    -- the attributes a fit fills (they start as ``None``; their names are the reference's, user code
                         reads them),
* ``learn(engine, images)``   -- returns the values for those slots,
* ``arguments()``     -- what the backend's ``transform`` receives after the images,

and this class does the rest.  Backend ids: ``"torch_hip"`` (libstainx_hip.so on an MI355X) and ``"torch_cuda"`` as an
alias, so code written for the reference's compiled backend drops in unchanged (on ROCm a GPU tensor's
``device.type`` is ``"cuda"``).  There is deliberately no ``"torch"`` id: the pure-PyTorch path belongs to upstream
stainx and this package never computes on the CPU -- requesting it raises ``ValueError``, a missing native library
``ImportError``.
"""
from __future__ import annotations

import importlib
from typing import Any

from stainx_amd.base import StainNormalizerBase
from stainx_amd.utils import device_type_of

BACKEND_IDS = {"torch_hip": "torch_hip", "torch_cuda": "torch_hip"}      # accepted id -> canonical id
_VALID_BACKENDS = frozenset(BACKEND_IDS)


class NormalizerTemplate(StainNormalizerBase):
    engine: str = ""                      # backend class name
    fitted_slots: tuple[str, ...] = ()    # attributes filled by fit()

    def __init__(self, device: str | Any | None = None, backend: str | None = None):
        super().__init__(device)
        if backend is None:
            self.backend = "torch_hip"      # the only one; whether it can run shows when the engine is built
        else:
            if backend not in BACKEND_IDS:
                where = " (the pure-PyTorch backend lives in upstream stainx; this package is the HIP backend)" if backend == "torch" else ""
                raise ValueError(f"Unsupported backend '{backend}'. Valid backends: {sorted(BACKEND_IDS)}{where}")
            from stainx_amd.backends.torch_hip_backend import HIP_AVAILABLE

            if not HIP_AVAILABLE:
                raise ImportError(f"Backend '{backend}' requires libstainx_hip.so. Build it with hipcc (see __graft_entry__.build); there is no CPU fallback.")
            self.backend = BACKEND_IDS[backend]
        self._engine = None
        for slot in self.fitted_slots:
            setattr(self, slot, None)

    # ---- what a normaliser declares ---------------------------------------------------------------------------
    def engine_options(self) -> dict:
        """Keyword arguments of the backend class besides the device."""
        return {}

    def learn(self, engine, images: Any) -> tuple:
        raise NotImplementedError

    def arguments(self) -> tuple:
        return tuple(getattr(self, slot) for slot in self.fitted_slots)

    def call_options(self) -> dict:
        """Keyword arguments of the backend's ``transform``."""
        return {}

    # ---- plumbing ---------------------------------------------------------------------------------------------
    def _device_type(self) -> str | None:
        return device_type_of(self.device)

    def _get_backend_impl(self):
        if self._engine is None:
            cls = getattr(importlib.import_module("stainx_amd.backends.torch_hip_backend"), self.engine)
            self._engine = cls(self.device, **self.engine_options())
        return self._engine

    def fit(self, images: Any) -> "NormalizerTemplate":
        values = self.learn(self._get_backend_impl(), images)
        for slot, value in zip(self.fitted_slots, values):
            setattr(self, slot, value)
        self._is_fitted = True
        return self

    def transform(self, images: Any) -> Any:
        if not self._is_fitted:
            raise ValueError("Must call fit() before transform()")
        return self._get_backend_impl().transform(images, *self.arguments(), **self.call_options())
