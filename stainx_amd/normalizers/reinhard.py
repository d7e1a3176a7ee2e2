"""Reinhard colour normalisation: LAB mean / standard deviation matching (API of stainx.Reinhard)."""
from __future__ import annotations

from stainx_amd.normalizers._template import NormalizerTemplate


class Reinhard(NormalizerTemplate):
    engine = "ReinhardHIP"
    fitted_slots = ("_reference_mean", "_reference_std")      # LAB (3,) each, float32 on the device

    def learn(self, engine, images):
        return engine.compute_reference_mean_std(images)
