"""pytest configuration: the ``gpu`` marker and shared fixture helpers."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name: str) -> dict:
    with np.load(GOLDEN / name, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def bf16_from_bits(bits: np.ndarray) -> torch.Tensor:
    """Golden files store bfloat16 tensors as their uint16 bit patterns."""
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)


def golden_tensor(arr: np.ndarray, name: str) -> torch.Tensor:
    if name == "bf16":
        return bf16_from_bits(arr)
    return torch.from_numpy(arr.copy())


TORCH_DTYPES = {"f32": torch.float32, "u8": torch.uint8, "bf16": torch.bfloat16, "f16": torch.float16, "f64": torch.float64}


@pytest.fixture(scope="session")
def golden():
    cache: dict[str, dict] = {}

    def get(name: str) -> dict:
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get
