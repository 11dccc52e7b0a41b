"""qgtc_ppopp22_amd — MI355X-native QGTC bit-GEMM hot path.

The product is two in-tree native libraries (built by ``__graft_entry__.build()``):

* ``libqgtc_hip.so``  — hand-written HIP kernels for gfx950 behind the C-ABI of ``include/qgtc.h``;
* ``QGTC*.so``        — the PyTorch-ROCm extension that keeps the reference's ``QGTC`` operator
                        surface (``val2bit, bit2val, bitMM2Bit, bitMM2Bit_profile, bitMM2Bit_base_cnt,
                        bitMM2Bit_zerojump_cnt, bitMM2Bit_col, bitMM2Int``).

There is no CPU or PyTorch fallback: importing :func:`load_ext` raises if the extension is missing.
"""
from __future__ import annotations

import importlib
import os

from .shapes import P8, P128, S8, S128, cols_shape, rows_shape  # noqa: F401

__all__ = ["load_ext", "lib_path", "S8", "S128", "P8", "P128", "rows_shape", "cols_shape"]

_PKG = os.path.dirname(os.path.abspath(__file__))


def lib_path() -> str:
    """Path of the C-ABI shared library (for ctypes / cgo-style consumers)."""
    return os.path.join(_PKG, "libqgtc_hip.so")


def load_ext():
    """Import and return the compiled ``QGTC`` extension module. Fails loudly if it is not built."""
    import torch  # noqa: F401  (libtorch must be loaded before the extension)

    try:
        return importlib.import_module("qgtc_ppopp22_amd.QGTC")
    except ImportError as e:  # pragma: no cover - exercised only on a broken install
        raise ImportError(
            "the QGTC HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` from the repo root); there is no fallback path"
        ) from e
