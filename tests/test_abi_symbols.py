"""The C-ABI library loads without a GPU, exports every symbol include/qgtc.h declares, agrees with
the oracle on the shape algebra and rejects bad arguments before touching the device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import qgtc_ppopp22_amd

    path = qgtc_ppopp22_amd.lib_path()
    assert os.path.exists(path), "libqgtc_hip.so is not built (run __graft_entry__.build())"
    L = ctypes.CDLL(path)
    L.qgtc_rows_words.restype = ctypes.c_size_t
    L.qgtc_cols_words.restype = ctypes.c_size_t
    L.qgtc_strerror.restype = ctypes.c_char_p
    return L


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "qgtc.h")).read()
    return sorted(set(re.findall(r"\b(qgtc_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n


def test_abi_version_and_errors(lib):
    assert lib.qgtc_abi_version() == 11
    assert lib.qgtc_strerror(0) == b"ok"
    for code in range(1, 6):
        assert lib.qgtc_strerror(code) not in (b"ok", b"unknown error")


@pytest.mark.parametrize("H,W,b", [(1, 1, 1), (8, 128, 2), (9, 129, 3), (1213, 128, 2), (4096, 64, 8)])
def test_shape_algebra_matches_oracle(lib, oracle, H, W, b):
    assert lib.qgtc_rows_words(H, W, b) == oracle.rows_words(H, W, b)
    assert lib.qgtc_cols_words(H, W, b, 0) == oracle.cols_words(H, W, b, False)
    assert lib.qgtc_cols_words(H, W, b, 1) == oracle.cols_words(H, W, b, True)


def test_bad_arguments_are_rejected_without_a_device(lib):
    EINVAL, ESIZE, EALIGN = 1, 2, 3
    buf = (ctypes.c_uint32 * 64)()
    f = (ctypes.c_float * 64)()
    assert lib.qgtc_val2bit(None, 4, 4, 1, 0, 0, buf, ctypes.c_size_t(64), None) == EINVAL
    assert lib.qgtc_val2bit(f, 4, 4, 0, 0, 0, buf, ctypes.c_size_t(64), None) == EINVAL
    assert lib.qgtc_val2bit(f, 4, 4, 33, 0, 0, buf, ctypes.c_size_t(64), None) == EINVAL
    assert lib.qgtc_val2bit(f, 4, 4, 1, 0, 0, buf, ctypes.c_size_t(3), None) == ESIZE
    sz = ctypes.c_size_t
    assert lib.qgtc_bitmm2bit(buf, sz(64), buf, sz(64), 0, 8, 8, 1, 1, 1, buf, sz(64), 0, None) == EINVAL
    assert lib.qgtc_bitmm2bit(buf, sz(64), buf, sz(64), 8, 8, 8, 1, 40, 1, buf, sz(64), 0, None) == EINVAL
    assert lib.qgtc_bitmm2bit(buf, sz(64), buf, sz(64), 8, 8, 8, 1, 1, 1, buf, sz(1), 0, None) == ESIZE
    mis = ctypes.cast(ctypes.addressof(buf) + 4, ctypes.POINTER(ctypes.c_uint32))
    assert lib.qgtc_bitmm2bit(mis, sz(60), buf, sz(64), 8, 8, 8, 1, 1, 1, buf, sz(64), 0, None) == EALIGN


def test_extension_imports_and_keeps_the_reference_surface():
    import qgtc_ppopp22_amd

    ext = qgtc_ppopp22_amd.load_ext()
    for name in ("val2bit", "bit2val", "bitMM2Bit", "bitMM2Bit_profile", "bitMM2Bit_base_cnt",
                 "bitMM2Bit_zerojump_cnt", "bitMM2Bit_col", "bitMM2Int"):
        assert callable(getattr(ext, name)), name
    import QGTC  # the reference's module name

    assert QGTC.bitMM2Bit is ext.bitMM2Bit
    # the names BASELINE.json's north_star uses for the same operators (QGTC_conv.py's vocabulary): aliases, not copies
    assert QGTC.bit_qnt is QGTC.val2bit and QGTC.mm_v1 is QGTC.bitMM2Bit and QGTC.mm_v2 is QGTC.bitMM2Int
    assert ext.bit_qnt is ext.val2bit and ext.mm_v1 is ext.bitMM2Bit and ext.mm_v2 is ext.bitMM2Int


def test_engine_env_default():
    """QGTC_ENGINE sets the engine unmodified callers start with (the extension imports without a GPU)."""
    import subprocess
    import sys

    code = "import QGTC; print(QGTC.get_engine())"
    for env_val, want in ((None, "auto"), ("popcount", "popcount"), ("mfma", "mfma")):
        env = dict(os.environ)
        env.pop("QGTC_ENGINE", None)
        if env_val:
            env["QGTC_ENGINE"] = env_val
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-500:]
        assert out.stdout.strip().splitlines()[-1] == want


def test_epoch_pool_layout_is_host_side(lib, oracle):
    """Entry points that do no device work: the pool rule of a device-filled epoch plan (outputs in (stage, batch)
    order, each a multiple of four words, sizes = the reference's allocation rules QGTC_device.cu:223,456,507) and the size
    helpers of the chain / tile formats."""
    from helpers import QgtcStage as Stage

    lib.qgtc_epoch_pool_layout.restype = ctypes.c_size_t
    lib.qgtc_epoch_pool_layout.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.qgtc_chain_words.restype = lib.qgtc_weight_codes_words.restype = ctypes.c_size_t
    ns = [1213, 599, 37, 8, 1]
    stages = (Stage * 4)(Stage(2, 16, 128, 128, 2, 2, 2, 1, 0, 0, 0), Stage(0, 32, -1, 128, 1, 2, 3, 0, 0, 1, 0), Stage(0, 33, -1, 10, 1, 2, 1, 2, 1, 1, 0),
                         Stage(2, 16, 128, 100, 2, 2, 2, 1, 0, 0, 1))
    nodes = (ctypes.c_int32 * len(ns))(*ns)
    offs = (ctypes.c_uint64 * (4 * len(ns)))()
    total = lib.qgtc_epoch_pool_layout(ctypes.addressof(nodes), len(ns), ctypes.addressof(stages), 4, ctypes.addressof(offs))
    sizes = [oracle.cols_words(n, 128, 2, False) for n in ns] + [oracle.rows_words(n, 128, 3) for n in ns] + [(n * 10 + 3) // 4 * 4 for n in ns] \
        + [lib.qgtc_chain_words(n, 100, 2) for n in ns]                       # fmt 1: the chain format of qgtc_chain_*
    assert [lib.qgtc_chain_words(n, 100, 2) for n in ns] == [(n + 127) // 128 * 128 * 16 for n in ns]
    lib.qgtc_adj_tiles_words.restype = ctypes.c_size_t                     # 512-byte tiles: [32-row block][k-quad][32 rows][4 words]
    assert lib.qgtc_adj_tiles_words(1213, 1213) == 38 * 10 * 128 and lib.qgtc_adj_tiles_words(1, 129) == 2 * 128 and lib.qgtc_adj_tiles_words(0, 5) == 0
    assert lib.qgtc_weight_codes_words(128, 100, 2, 1) == 4 * 2 * 64 * 4 and lib.qgtc_weight_codes_words(64, 50, 4, 1) == 2 * 2 * 2 * 64 * 4
    assert lib.qgtc_weight_codes_words(128, 70, 3, 0) == 4 * 2 * 2 * 64 * 4      # three column blocks are kept as four (the fourth: zero codes)
    assert lib.qgtc_weight_codes_words(602, 128, 2, 0) == 5 * 4 * 2 * 64 * 4     # order 0: a table per k-quad of K (ABI 11)
    assert lib.qgtc_weight_codes_words(128, 128, 2, 1) == 4 * 2 * 64 * 4 and lib.qgtc_weight_codes_words(0, 128, 2, 0) == 0
    # ABI 11's wider chains: 129 .. 256 columns of K in the second product take four 64-column slices a column block, five to seven
    # column blocks are kept as eight; 5 .. 8-bit weights are four base-4 digits; T at 5 .. 8 bits is two arrays of the 4-bit form
    assert lib.qgtc_weight_codes_words(200, 200, 2, 1) == 8 * 4 * 64 * 4 and lib.qgtc_weight_codes_words(602, 128, 2, 1) == 0
    assert lib.qgtc_weight_codes_words(128, 128, 8, 1) == 4 * 2 * 4 * 64 * 4 and lib.qgtc_weight_codes_words(128, 128, 5, 0) == 4 * 2 * 4 * 64 * 4
    assert lib.qgtc_chain_words(1213, 128, 8) == 2 * lib.qgtc_chain_words(1213, 128, 4) and lib.qgtc_chain_words(1213, 200, 2) == 10 * 256 * 16
    assert total == sum(sizes)
    assert list(offs) == [sum(sizes[:i]) for i in range(len(sizes))]
    assert lib.qgtc_epoch_pool_layout(None, 3, ctypes.addressof(stages), 3, None) == 0


def test_route_functions_name_the_kernel_behind_a_call(lib):
    """qgtc_bitmm_route / qgtc_bitmm_batched_route: host-only, the SAME rule functions the launchers switch on (qgtc_hip.hip:
    single_route / batched_route), so the routing table of DESIGN.md (tools/routing_table.py) cannot drift from the code."""
    lib.qgtc_bitmm_route.restype = lib.qgtc_bitmm_batched_route.restype = ctypes.c_char_p
    POP, MFMA, AUTO, JUMP = 0x0, 0x8, 0x10, 0x4
    r = lambda *a: lib.qgtc_bitmm_route(*a).decode()             # noqa: E731
    g = lambda *a: lib.qgtc_bitmm_batched_route(*a).decode()     # noqa: E731
    # bench.py's step (BASELINE.json configs[1]): the FP4 narrow-operand kernel on the default engine, AND + popcount when asked
    assert r(4096, 4096, 64, 1, 1, 1, 0, AUTO) == "k_bitmm_fp4_one" and r(4096, 4096, 64, 1, 1, 1, 0, POP) == "k_bitmm"
    assert r(4096, 4096, 64, 1, 8, 8, 0, AUTO) == "k_bitmm_fp4_one"
    assert r(32768, 32768, 64, 1, 1, 1, 0, AUTO) == "k_bitmm_fp4_stream"           # 5_9's largest: the long-K kernel (round 6)
    assert r(16384, 16384, 256, 1, 1, 1, 1, AUTO) == "k_bitmm_fp4_stream" and r(32768, 32768, 1024, 1, 1, 1, 0, AUTO) == "k_bitmm_fp4_wide"
    assert r(4096, 8192, 64, 1, 1, 1, 0, AUTO) == "k_bitmm_fp4_skinny" and r(4096, 8192, 64, 1, 1, 1, 0, MFMA) == "k_bitmm_fp4_stream"   # few rows
    assert r(32768, 32768, 64, 1, 2, 2, 0, AUTO) == "k_bitmm_fp4_skinny"           # more than one plane: K > 4096 stays there
    assert r(4096, 4096, 1024, 1, 1, 1, 0, AUTO) == "k_bitmm_fp4_wide"
    assert r(4096, 4096, 64, 8, 8, 8, 0, AUTO) == "k_bitmm_mfma"                   # 4096 * 255 * 255 >= 2^24: int8 form, int32 sums
    assert r(512, 512, 64, 9, 2, 4, 0, MFMA) == "k_bitmm"                          # nine planes: AND + popcount only
    assert r(0, 4, 4, 1, 1, 1, 0, AUTO) == "invalid" and r(4, 4, 4, 1, 1, 1, 3, AUTO) == "invalid"
    assert g(1213, 128, 128, 2, 2, 2, 1, AUTO) == "k_bitmm_fp4_xw_rows"
    assert g(1213, 1213, 128, 1, 2, 2, 0, AUTO | JUMP) == "k_bitmm_fp4_rows"
    assert g(1213, 1213, 128, 1, 2, 2, 0, POP | JUMP) == "k_bitmm_batched"
    assert g(1213, 128, 128, 8, 8, 8, 1, AUTO) == "k_bitmm_fp4_rows" and g(1213, 300, 512, 2, 2, 2, 1, AUTO) == "k_bitmm_mfma_batched"
    # the table in DESIGN.md is this tool's output
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "routing_table.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "k_bitmm_fp4_one" in out.stdout
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for line in out.stdout.strip().splitlines():
        assert line in design, "DESIGN.md's routing table is stale: re-run tools/routing_table.py\n" + line


def test_the_library_alone_decides_the_loaders_route(lib, monkeypatch):
    """qgtc_load_work_words is the one decider (it reads QGTC_NO_LOAD_SORT on every call): 0 = the bitmap route, whatever was asked before."""
    lib.qgtc_load_work_words.restype = ctypes.c_size_t
    lib.qgtc_load_work_words.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
    assert lib.qgtc_load_work_words(3, 257, 1000) > 0
    monkeypatch.setenv("QGTC_NO_LOAD_SORT", "1")
    assert lib.qgtc_load_work_words(3, 257, 1000) == 0
    monkeypatch.setenv("QGTC_NO_LOAD_SORT", "0")
    assert lib.qgtc_load_work_words(3, 257, 1000) > 0
    monkeypatch.delenv("QGTC_NO_LOAD_SORT")
    assert lib.qgtc_load_work_words(3, 257, 1000) > 0
