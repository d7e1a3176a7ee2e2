"""The two benchmark command lines carry the reference's flags (benchmarks/benchmark_stainx_backend.py:83-97,
benchmarks/run_stainx.py:20-45) and produce what they document: run in-process on the GPU, flags and refusals on the CPU."""
from __future__ import annotations

import importlib.util
import json
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def _load(name: str):
    spec = importlib.util.spec_from_file_location(name, ROOT / "benchmarks" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(mod, argv, capsys):
    old = sys.argv
    sys.argv = [mod.__name__] + argv
    try:
        rc = mod.main()
    finally:
        sys.argv = old
    io = capsys.readouterr()
    return rc, io.out, io.err


def _help_text(mod) -> str:
    import contextlib
    import io

    buf = io.StringIO()
    old = sys.argv
    sys.argv = [mod.__name__, "--help"]
    try:
        with contextlib.redirect_stdout(buf), pytest.raises(SystemExit) as exc:
            mod.main()
        assert exc.value.code == 0
    finally:
        sys.argv = old
    return buf.getvalue()


def test_cli_flags_are_the_references(capsys):
    grid, single = _load("benchmark_stainx_amd_grid"), _load("run_stainx_amd")
    for mod, flags in ((grid, ("--method", "--image-size", "--channels", "--warmup", "--runs", "--seed", "--batch-size", "--backend1", "--backend2")),
                       (single, ("--batch-size", "--height", "--width", "--channels", "--device", "--runs", "--seed"))):
        text = _help_text(mod)
        for f in flags:
            assert f in text, (mod.__name__, f)
    with pytest.raises(SystemExit):
        _run(grid, [], capsys)                                   # --method is required, as in the reference
    rc, _, err = _run(single, ["macenko", "--device", "cpu"], capsys)
    assert rc == 2 and "not supported" in err                    # no CPU path in this package


@pytest.mark.gpu
def test_grid_cli_rows_and_relative_error(capsys):
    grid = _load("benchmark_stainx_amd_grid")
    rc, out, err = _run(grid, ["--method", "macenko", "--image-size", "64", "--batch-size", "2", "--runs", "3", "--warmup", "1", "--backend2", "oracle"], capsys)
    assert rc == 0, err
    rows = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(rows) == 1 and rows[0]["batch"] == 2 and rows[0]["size"] == 64 and rows[0]["method"] == "macenko"
    assert rows[0]["ms_per_call"] > 0 and rows[0]["images_per_s"] > 0 and rows[0]["speedup"] > 0
    assert rows[0]["relative_error"] < 5e-3                      # uint8 tiles: a grey level here and there at truncation boundaries
    assert "rel. error" in err                                   # the table the reference prints at the end
    for method, data in (("reinhard", "noise"), ("histogram_matching", "noise")):
        rc, out, err = _run(grid, ["--method", method, "--image-size", "32", "48", "--batch-size", "2", "3", "--runs", "2", "--warmup", "1", "--backend2", "oracle", "--dtype", "float32"], capsys)
        assert rc == 0, err
        rows = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
        assert [(r["batch"], r["size"]) for r in rows] == [(2, 32), (2, 48), (3, 32), (3, 48)]
        assert all(r["relative_error"] < (1e-5 if method == "histogram_matching" else 1e-3) for r in rows), rows


@pytest.mark.gpu
def test_single_method_cli(capsys):
    single = _load("run_stainx_amd")
    for method in ("macenko", "reinhard", "histogram_matching"):
        rc, out, err = _run(single, [method, "--batch-size", "4", "--height", "96", "--width", "64", "--runs", "2", "--data", "he" if method == "macenko" else "noise"], capsys)
        assert rc == 0, err
        assert "Result shape: (4, 3, 96, 64)" in out and "Images per second" in out and f"Method: {method}" in out
