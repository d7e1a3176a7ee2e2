"""Macenko stain normalisation (API of stainx.Macenko, incl. ``normalize_to_0_1`` and ``precision``)."""
from __future__ import annotations

from typing import Any

from stainx_amd.normalizers._template import NormalizerTemplate


class Macenko(NormalizerTemplate):
    """``normalize_to_0_1`` defaults to False here (output ~[0,255]); ``StainNormalizerTransform`` defaults it to True.
    ``precision`` takes the reference's two values, ``"stable"`` and ``"fast"``: both run the same exact kernels (fp64 covariance, exact
    nearest-rank percentiles) -- the reference's "fast" is a reduced-precision variant of its CUDA path (fp16 tensors, MAE ~0.05 grey
    levels), and the exact result lies inside its tolerance.  ``"sampled"`` is an extension (opt-in approximation: percentiles of a
    4096-pixel sample per tile, mean error ~0.5 / worst ~5 grey levels, about twice the throughput)."""

    engine = "MacenkoHIP"
    fitted_slots = ("_stain_matrix", "_target_max_conc", "_concentration_matrix")      # (3,2), (2,), unused

    def __init__(self, device: Any | None = None, backend: str | None = None, normalize_to_0_1: bool = False, precision: str = "stable", *,
                 output_dtype: Any | None = None):
        if precision not in ("stable", "fast", "sampled"):
            raise ValueError(f"precision must be 'stable' or 'fast' (or the extension 'sampled'), got {precision!r}")
        self._precision = precision
        self.normalize_to_0_1 = normalize_to_0_1
        # extension (not in the reference): uint8 tiles come out as torch.bfloat16 / torch.float16, the `.to(dtype)` of the
        # result fused into the call; None keeps the reference's output type
        self.output_dtype = output_dtype
        super().__init__(device=device, backend=backend)

    def engine_options(self) -> dict:
        return {} if self._precision == "stable" else {"precision": self._precision}

    def learn(self, engine, images):
        stain_matrix, max_conc = engine.compute_reference_stain_matrix(images)
        return stain_matrix, max_conc, None

    def arguments(self) -> tuple:
        return (self._stain_matrix, self._target_max_conc)

    def call_options(self) -> dict:
        # the `/255` after the cast to the input dtype is fused into the last kernel
        options = {"normalize_to_0_1": bool(self.normalize_to_0_1)}
        if self.output_dtype is not None:
            options["out_dtype"] = self.output_dtype
        return options
