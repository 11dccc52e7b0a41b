"""Drop-in module name of the reference's extension: ``import QGTC`` (main_qgtc.py:16, sampler.py:10,
2_7c_QGTC_GEMM_INT8.py:3, unitest.py:3) resolves to the MI355X build."""
import torch  # noqa: F401

from qgtc_ppopp22_amd import load_ext as _load_ext

_ext = _load_ext()
globals().update({k: getattr(_ext, k) for k in dir(_ext) if not k.startswith("_") and k != "torch"})
__doc__ = _ext.__doc__
