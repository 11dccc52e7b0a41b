"""The grouped data loader (qgtc_load_batches / QGTC.EpochPlan.load): every batch of an iterator packed by one library call.
Every output must be word for word what the single-batch entries give - which tests/test_gpu_parity.py pins to the oracle -
and, directly, what the oracle packs (sampler.py:76-106: dense adjacency from the edges, val2bit(A, 1), val2bit(X, b, True))."""
import ctypes
import random

import numpy as np
import pytest

from helpers import to_np_u32

pytestmark = pytest.mark.gpu


def _batches(rng, sizes, F, dup=True, empty_first=False):
    rows, cols, feats, ecounts = [], [], [], []
    for i, n in enumerate(sizes):
        ne = 0 if (empty_first and i == 0) else int(rng.integers(n, 8 * n))
        r, c = rng.integers(0, n, ne), rng.integers(0, n, ne)
        if dup and ne > 8:      # multiplicities 2, 3 and 4 of some cells: the 1-bit quantiser maps them to 0, 1, 1 (kernel.h:39-44)
            r = np.concatenate([r, r[:4], r[:2], r[:2], r[1:2]])
            c = np.concatenate([c, c[:4], c[:2], c[:2], c[1:2]])
        rows.append(r.astype(np.int64))
        cols.append(c.astype(np.int64))
        x = rng.normal(size=(n, F)).astype(np.float32) * 2.0
        x.flat[:7] = [-1.0, -0.0, 0.5, 1.5, 2.5, np.nan, 1e9][: min(7, x.size)]
        feats.append(x)
        ecounts.append(int(r.size))
    return rows, cols, feats, ecounts


def _dense(n, r, c):
    A = np.zeros((n, n), dtype=np.float32)
    np.add.at(A, (r, c), 1.0)      # torch.sparse.FloatTensor(...).to_dense() sums duplicates (sampler.py:87-89)
    return A


@pytest.mark.parametrize("route", ["buckets", "bitmaps", "too-wide", "widest-buckets"])
@pytest.mark.parametrize("bits,F,chain", [(2, 128, 0), (4, 50, 4), (1, 33, 1), (3, 64, 3), (8, 16, 0), (8, 40, 8), (5, 128, 5)])
def test_grouped_loader_equals_the_oracle_and_the_single_batch_entries(qgtc, oracle, bits, F, chain, route, monkeypatch):
    """route: "buckets" = the default (edges bucketed by 32-row block, every word of rows + tiles + bitmaps written once from LDS);
    "bitmaps" = the three dense multiplicity bitmaps (QGTC_NO_LOAD_SORT, and every iterator with a batch above 5120 nodes: "too-wide")."""
    import torch
    if route == "bitmaps":
        monkeypatch.setenv("QGTC_NO_LOAD_SORT", "1")
    if route in ("too-wide", "widest-buckets") and bits != 2:
        pytest.skip("one width is enough for the 5200-node fallback / the 5100-node batch (40 k-quads: 60 KB of LDS a row block)")
    rng = np.random.default_rng(bits * 100 + F)
    sizes = [1213, 599, 37, 8, 1, 130, 2049, 300]          # ragged; 2049 nodes = 17 k-quads (the kernel's wide-row path)
    if route == "too-wide":
        sizes = [5200, 64, 1213]
    if route == "widest-buckets":
        sizes = [5100, 64, 1213]
    rows, cols, feats, ecounts = _batches(rng, sizes, F, empty_first=(bits == 3))
    src, dst = (torch.from_numpy(np.concatenate(v)).cuda() for v in (rows, cols))
    X = torch.from_numpy(np.concatenate(feats)).cuda()
    plan = qgtc.EpochPlan.load(src, dst, ecounts, X, sizes, bits, True, chain, True, True)
    torch.cuda.synchronize()
    tiles_occupied = tiles_all = 0
    pA, pX, pXr = plan.As, plan.Xs, plan.Xrs
    for i, n in enumerate(sizes):
        A = _dense(n, rows[i], cols[i])
        oA = oracle.val2bit(A, 1)
        np.testing.assert_array_equal(to_np_u32(pA[i]), oA, err_msg=f"A of batch {i}")
        np.testing.assert_array_equal(to_np_u32(pX[i]), oracle.val2bit(feats[i], bits, True), err_msg=f"X (cols) of batch {i}")
        np.testing.assert_array_equal(to_np_u32(pXr[i]), oracle.val2bit(feats[i], bits, False), err_msg=f"X (rows) of batch {i}")
        assert pA[i].shape == (oracle.rows_words(n, n, 1) // ((n + 127) // 128 * 4), (n + 127) // 128 * 4)
        # the formats of the grouped epoch against the single-batch entries on the SAME packed tensors
        dA = torch.from_numpy(oA.view(np.int32)).cuda()
        lib_tiles = _adj_tiles(qgtc, dA, n)
        assert torch.equal(plan.format_of(i, qgtc.SRC_AT), lib_tiles), f"tiles of batch {i}"
        occ = qgtc.tile_occupancy(dA, n, n, 1)
        assert torch.equal(plan.format_of(i, -1), occ), f"bitmap of batch {i}"
        tiles_occupied += int(sum(bin(int(w) & (2 ** 64 - 1)).count("1") for w in occ.cpu().numpy().view(np.uint64)))
        tiles_all += ((n + 31) // 32) * ((n + 127) // 128)
        if chain:
            assert torch.equal(plan.format_of(i, qgtc.SRC_XC), _chain_from_cols(qgtc, pX[i], n, F, bits)), f"X (chain) of batch {i}"
    assert abs(plan.occupied_fraction - tiles_occupied / tiles_all) < 1e-12
    # out-of-range indices: skipped, and reported when asked
    bad_src = src.clone()
    bad_src[3] = sizes[0]
    with pytest.raises(RuntimeError, match="out of range"):
        qgtc.EpochPlan.load(bad_src, dst, ecounts, X, sizes, bits, False, 0, False, True)
    with pytest.raises(RuntimeError, match="add up"):
        qgtc.EpochPlan.load(src, dst, ecounts[:-1] + [ecounts[-1] + 1], X, sizes, bits)



@pytest.mark.parametrize("seed", range(16))
def test_grouped_loader_random_shapes_equal_the_oracle(qgtc, oracle, seed):
    """Random batch counts, sizes, feature widths (odd, below and above a 64-column chunk) and bit widths: the adjacency and BOTH layouts of
    X - the rows layout comes out of a 32 x 32 bit transpose over the lanes of the cols-layout words (pack_kernels.hip.h::transpose32_lanes)."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    sizes = [int(v) for v in rng.integers(1, 900, int(rng.integers(1, 5)))]
    if seed % 4 == 0:
        sizes.append(1700)                       # (up to 13 600 edges: the passes of k_load_sort beyond its register-resident form)
    F, bits = int(rng.integers(1, 200)), int(rng.integers(1, 9))
    rows, cols, feats, ecounts = _batches(rng, sizes, F, empty_first=(seed % 5 == 0))
    src, dst = (torch.from_numpy(np.concatenate(v)).cuda() for v in (rows, cols))
    X = torch.from_numpy(np.concatenate(feats)).cuda()
    plan = qgtc.EpochPlan.load(src, dst, ecounts, X, sizes, bits, True, 0, True, True)
    torch.cuda.synchronize()
    occupied = 0
    for i, n in enumerate(sizes):
        oA = oracle.val2bit(_dense(n, rows[i], cols[i]), 1)
        np.testing.assert_array_equal(to_np_u32(plan.As[i]), oA, err_msg=f"A of batch {i} (sizes {sizes}, F {F}, {bits} bits)")
        np.testing.assert_array_equal(to_np_u32(plan.Xs[i]), oracle.val2bit(feats[i], bits, True), err_msg=f"X (cols) of batch {i} (F {F}, {bits} bits)")
        np.testing.assert_array_equal(to_np_u32(plan.Xrs[i]), oracle.val2bit(feats[i], bits, False), err_msg=f"X (rows) of batch {i} (F {F}, {bits} bits)")
        occ = qgtc.tile_occupancy(torch.from_numpy(oA.view(np.int32)).cuda(), n, n, 1)
        assert torch.equal(plan.format_of(i, -1), occ), f"bitmap of batch {i}"
        occupied += int(sum(bin(int(w) & (2 ** 64 - 1)).count("1") for w in occ.cpu().numpy().view(np.uint64)))
    tiles_all = sum(((n + 31) // 32) * ((n + 127) // 128) for n in sizes)
    assert abs(plan.occupied_fraction - occupied / tiles_all) < 1e-12       # (k_load_stats: the sum of the per-row-block counts)


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        import qgtc_ppopp22_amd
        _LIB = ctypes.CDLL(qgtc_ppopp22_amd.lib_path())
        _LIB.qgtc_adj_tiles_words.restype = ctypes.c_size_t
        _LIB.qgtc_chain_words.restype = ctypes.c_size_t
    return _LIB


def _adj_tiles(qgtc, dA, n):
    import torch
    lib = _lib()
    words = lib.qgtc_adj_tiles_words(n, n)
    out = torch.empty(words, dtype=torch.int32, device="cuda")
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    lib.qgtc_adj_tiles_from_rows.argtypes = [vp, sz, ctypes.c_int, ctypes.c_int, vp, sz, vp]
    assert lib.qgtc_adj_tiles_from_rows(dA.data_ptr(), dA.numel(), n, n, out.data_ptr(), words, torch.cuda.current_stream().cuda_stream) == 0
    return out


def _chain_from_cols(qgtc, dX, n, F, bits):
    import torch
    lib = _lib()
    words = lib.qgtc_chain_words(n, F, bits)
    out = torch.empty(words, dtype=torch.int32, device="cuda")
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    lib.qgtc_chain_from_cols.argtypes = [vp, sz, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, sz, vp]
    assert lib.qgtc_chain_from_cols(dX.data_ptr(), dX.numel(), n, F, bits, out.data_ptr(), words, torch.cuda.current_stream().cuda_stream) == 0
    return out


@pytest.mark.parametrize("dataset,bits,gin", [("ogbn-arxiv", 2, False), ("ppi", 4, True), ("tiny", 3, False)])
def test_cluster_iter_grouped_equals_batch_by_batch(qgtc, dataset, bits, gin):
    """ClusterIter's default (one EpochPlan.load call for the iterator) against its batch-by-batch route (QGTC.pack_edges +
    QGTC.val2bit per batch, the structure of sampler.py:76-106): the same packed tensors, the same per-batch parameters, and
    the same grouped epoch outputs from either iterator's plan."""
    import torch
    from qgtc_ppopp22_amd import driver, graph as G
    from qgtc_ppopp22_amd.sampler import ClusterIter

    psize, bsz = (40, 4) if dataset == "tiny" else (1500, 20)
    g = G.make_graph(dataset, psize)
    ids = list(range(0, psize // bsz, 3))
    its = []
    for grouped in (True, False):
        random.seed(2)
        its.append(ClusterIter(dataset, g, psize, bsz, bit_width=bits, run_GIN=gin, device="cuda", qgtc=qgtc, batch_ids=ids, with_rows_X=True,
                               keep_raw=True, grouped=grouped))
    a, b = its
    assert a.cluster_param_li == b.cluster_param_li and a.n_edges == b.n_edges and len(a) == len(ids)
    for x, y, ra, rb in zip(a.cTensor_li, b.cTensor_li, a.raw_li, b.raw_li):
        assert torch.equal(x.bit_A, y.bit_A) and torch.equal(x.bit_X, y.bit_X) and torch.equal(x.bit_X_rows, y.bit_X_rows)
        assert x.bit_A.shape == y.bit_A.shape and x.bit_X.shape == y.bit_X.shape and x.bit_X_rows.shape == y.bit_X_rows.shape
        assert all(torch.equal(p, q) for p, q in zip(ra, rb))
    F = g.feat.shape[1]
    H = 64 if gin else 128
    W = driver.pack_weights(qgtc, F, H, 10, bits, torch.device("cuda"))
    outs = []
    for it in its:
        plan = driver.PlannedEpoch(qgtc, it.epoch_data(qgtc), it.cluster_param_li, W, bits, "correct", gin)
        plan.run()
        torch.cuda.synchronize()
        outs.append([o.clone() for o in plan.outs])
    assert all(torch.equal(p, q) for p, q in zip(*outs))
    # packing again from the resident raw arrays (a driver that packs inside its epoch loop) gives the same tensors
    again = a.pack_now(qgtc)
    assert all(torch.equal(p, q.bit_A) for p, q in zip(again.As, a.cTensor_li)) and all(torch.equal(p, q.bit_X) for p, q in zip(again.Xs, a.cTensor_li))


@pytest.mark.parametrize("with_work", [False, True])
def test_load_batches_through_the_raw_abi(oracle, with_work):
    """qgtc_load_batches through ctypes: the table written by the host with the struct layout of include/qgtc.h, raw device
    pointers, no optional format - against the oracle. with_work (ABI 11): the caller hands over qgtc_load_work_words words and
    neither clears A nor supplies scratch; without it: the three multiplicity bitmaps inside the cleared region."""
    import torch
    lib = _lib()

    from helpers import QgtcLoaderBatch as LoaderBatch

    assert ctypes.sizeof(LoaderBatch) == 88
    lib.qgtc_rows_words.restype = lib.qgtc_cols_words.restype = ctypes.c_size_t
    rng = np.random.default_rng(11)
    sizes, F, bits = [70, 257, 5], 20, 2
    rows, cols, feats, ecounts = _batches(rng, sizes, F)
    src, dst = (torch.from_numpy(np.concatenate(v)).cuda() for v in (rows, cols))
    X = torch.from_numpy(np.concatenate(feats)).cuda()
    a_words = [lib.qgtc_rows_words(n, n, 1) for n in sizes]
    x_words = [lib.qgtc_cols_words(n, F, bits, 0) for n in sizes]
    zero = torch.full((3 * sum(a_words) + 4,), -1, dtype=torch.int32, device="cuda")
    xp = torch.full((sum(x_words),), -1, dtype=torch.int32, device="cuda")
    table, e0, f0, a0, x0 = [], 0, 0, 0, 0
    for n, ne, aw, xw in zip(sizes, ecounts, a_words, x_words):
        table.append(LoaderBatch(e0, ne, f0, n, 0, zero.data_ptr() + 4 * a0, None if with_work else zero.data_ptr() + 4 * (sum(a_words) + 4 + 2 * a0), None, None,
                                 xp.data_ptr() + 4 * x0, None, None))
        e0, f0, a0, x0 = e0 + ne, f0 + n, a0 + aw, x0 + xw
    dev_table = torch.frombuffer(bytearray(bytes((LoaderBatch * len(sizes))(*table))), dtype=torch.uint8).cuda()
    vp = ctypes.c_void_p
    lib.qgtc_load_batches.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t,
                                      vp, vp, ctypes.c_uint, vp, ctypes.c_size_t, vp]
    lib.qgtc_load_work_words.restype = ctypes.c_size_t
    lib.qgtc_load_work_words.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
    assert lib.qgtc_load_work_words(3, 257, sum(ecounts)) == 3 * (9 + 1) + sum(ecounts) + (sum(ecounts) & 1) + 3 * 9 and lib.qgtc_load_work_words(3, 5121, 10) == 0
    st = torch.cuda.current_stream().cuda_stream
    stats_ptr = zero.data_ptr() + 4 * sum(a_words)
    if with_work:    # only `stats` is cleared; A stays poisoned until the call has written every word of it
        work = torch.full((lib.qgtc_load_work_words(len(sizes), max(sizes), sum(ecounts)),), -1, dtype=torch.int32, device="cuda")
        rc = lib.qgtc_load_batches(dev_table.data_ptr(), len(sizes), max(sizes), max(ecounts), src.data_ptr(), dst.data_ptr(), X.data_ptr(), F, bits,
                                   stats_ptr, 16, stats_ptr, None, 0, work.data_ptr(), work.numel(), st)
    else:
        rc = lib.qgtc_load_batches(dev_table.data_ptr(), len(sizes), max(sizes), max(ecounts), src.data_ptr(), dst.data_ptr(), X.data_ptr(), F, bits,
                                   zero.data_ptr(), zero.numel() * 4, stats_ptr, None, 0, None, 0, st)
    assert rc == 0
    torch.cuda.synchronize()
    got_a, got_x = zero.cpu().numpy().view(np.uint32), xp.cpu().numpy().view(np.uint32)
    occupied = int(zero[sum(a_words):sum(a_words) + 2].cpu().numpy().view(np.uint64)[0])      # stats[0]: occupied 32-row x 128-bit tiles
    want_occupied = 0
    for i, n in enumerate(sizes):
        w = oracle.val2bit(_dense(n, rows[i], cols[i]), 1).reshape(-1, (n + 127) // 128, 4)
        want_occupied += sum(int(np.any(w[32 * t:32 * t + 32, q] != 0)) for t in range((n + 31) // 32) for q in range((n + 127) // 128))
    assert occupied == want_occupied
    a0 = x0 = 0
    for i, n in enumerate(sizes):
        np.testing.assert_array_equal(got_a[a0:a0 + a_words[i]], oracle.val2bit(_dense(n, rows[i], cols[i]), 1))
        np.testing.assert_array_equal(got_x[x0:x0 + x_words[i]], oracle.val2bit(feats[i], bits, True))
        a0, x0 = a0 + a_words[i], x0 + x_words[i]
    # bad arguments are error codes
    assert lib.qgtc_load_batches(None, 3, 300, 10, src.data_ptr(), dst.data_ptr(), X.data_ptr(), F, bits, zero.data_ptr(), 16, None, None, 0, None, 0, st) == 1
    assert lib.qgtc_load_batches(dev_table.data_ptr(), 3, 300, 10, None, dst.data_ptr(), X.data_ptr(), F, bits, zero.data_ptr(), 16, None, None, 0, None, 0, st) == 1
    assert lib.qgtc_load_batches(dev_table.data_ptr(), 0, 300, 10, src.data_ptr(), dst.data_ptr(), X.data_ptr(), F, bits, zero.data_ptr(), 16, None, None, 0, None, 0, st) == 1
    # a work buffer that cannot serve is an error, never a silent switch to the route whose cleared region was not supplied
    if with_work:
        args = (dev_table.data_ptr(), len(sizes), max(sizes), max(ecounts), src.data_ptr(), dst.data_ptr(), X.data_ptr(), F, bits, stats_ptr, 16, stats_ptr, None, 0)
        assert lib.qgtc_load_batches(*args, work.data_ptr() + 4, work.numel() - 1, st) == 3          # QGTC_EALIGN: not 8-byte aligned
        assert lib.qgtc_load_batches(*args, work.data_ptr(), 10, st) == 2                            # QGTC_ESIZE: below the fixed part of the layout
        assert lib.qgtc_load_batches(dev_table.data_ptr(), len(sizes), 6000, max(ecounts), src.data_ptr(), dst.data_ptr(), X.data_ptr(), F, bits, stats_ptr, 16,
                                     stats_ptr, None, 0, work.data_ptr(), work.numel(), st) == 1     # QGTC_EINVAL: batches above 5120 nodes have no bucketed route
        torch.cuda.synchronize()
