import sys, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import tests.test_abi_gpu as T
from oracle.qgtc_oracle import Oracle
import qgtc_ppopp22_amd
lib = ctypes.CDLL(qgtc_ppopp22_amd.lib_path())
for n in ("qgtc_rows_words", "qgtc_cols_words"):
    getattr(lib, n).restype = ctypes.c_size_t
lib.qgtc_strerror.restype = ctypes.c_char_p
orig = np.testing.assert_array_equal
def spy(a, b, err_msg="", **kw):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape == b.shape and a.dtype == np.float32 and not (a == b).all():
        bad = np.argwhere(a != b)
        print(err_msg, "shape", a.shape, "bad", len(bad), "rows", sorted(set(bad[:, 0]))[:40], "cols", sorted(set(bad[:, 1]))[:64])
        r, c = bad[0]
        print(" first", r, c, a[r, c], b[r, c], " row", r, "got", a[r, :16], "want", b[r, :16])
        return
    return orig(a, b, err_msg=err_msg, **kw)
np.testing.assert_array_equal = spy
T.test_chain_entries_with_raw_descriptors.__wrapped__ if hasattr(T.test_chain_entries_with_raw_descriptors, "__wrapped__") else None
T.test_chain_entries_with_raw_descriptors(lib, Oracle(), 40, 40, 7, 5, 3, False)
