"""Shared pieces of the measurement legs."""
from __future__ import annotations

import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REF_TFLOPS_4096_64 = {1: 46.768, 2: 26.818, 4: 14.196, 8: 7.324}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
# VALU issue rate of the v_and_b32 + v_bcnt_u32_b32 pair measured on MI355X with tools/valu_peak.hip (4.2e13 lane-instr/s
# at 8 waves/SIMD); one pair = 32 bit-MACs = 64 bit-ops. SURVEY.md 8(d) states the same roofline at full-rate issue
# (256 CUs x 4 SIMDs x 32 lane-instr/clk x 2.4 GHz = 7.864e13), which the VOP3 v_bcnt does not reach. Both are reported.
VALU_PEAK_BITOPS = 4.2e13 * 32
VALU_PEAK_BITOPS_SURVEY = 7.864e13 * 32
FP4_PEAK_TFLOPS = 10000.0        # MI355X_MICROARCH.md: ~10 PF dense FP4 / FP6 MFMA

PROFILE_DIR = "profiles/r06"
CLOCK_WARMUP_S = 0.3
EVENT_MIN_LAUNCHES = 200     # launches per HIP-event window (QGTC_device.cu:409 times 200 too)


@contextlib.contextmanager
def engine(Q, name):
    """Run a block on one engine of the library ("auto" is the shipped default) and put the previous one back."""
    prev = Q.get_engine()
    Q.set_engine(name)
    try:
        yield
    finally:
        Q.set_engine(prev)


@contextlib.contextmanager
def quiet_fd1():
    """Send file descriptor 1 itself to /dev/null for a block: the counter operators print their `counter_global:` /
    `counter:` lines with C printf (qgtc_torch.cpp, as the reference does), which sys.stdout redirection does not catch -
    and a pipe-buffered printf would surface at exit, BEHIND the JSON line. Flushed on both sides."""
    import ctypes

    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    null = os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    os.close(null)
    try:
        yield
    finally:
        sys.stdout.flush()
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def flush_c_stdio():
    import ctypes

    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)


def median_of_5(Q, ba, bx, M, K, N, w, reps=200):
    """The reference's measurement (QGTC_device.cu:403-422): `reps` launches between two events; the median of five such
    windows after an untimed one (200 reps per point, median of >= 5 runs). Milliseconds per window."""
    Q.profile(ba, bx, M, K, N, 1, w, w, max(reps // 10, 1))
    return sorted(Q.profile(ba, bx, M, K, N, 1, w, w, reps) for _ in range(5))[2]


def make_workload(Q, M, K, N, w, device, seed, ones=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if ones:      # 2_7c_QGTC_GEMM_INT8.py:6-12
        A = torch.ones((M, K))
        X = torch.ones((K, N))
    else:
        A = (torch.rand((M, K), generator=g) < 0.5).float()
        X = torch.randint(0, 2 ** w, (K, N), generator=g).float()
    bit_A = Q.val2bit(A.to(device), 1, False, False)
    bit_X = Q.val2bit(X.to(device), w, True, False)
    return A, X, bit_A, bit_X


def keep_clock_up(run, burst=20):
    """The chip drops into a low power state within milliseconds of idling (the same 20 launches take 85 us warm and
    200-450 us after 50-500 ms of idle): CLOCK_WARMUP_S of untimed, drained bursts ahead of a measured region."""
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < CLOCK_WARMUP_S:
        for _ in range(burst):
            run()
        torch.cuda.synchronize()


def profile_summary(name):
    """The committed rocprofv3 summary of a target (tools/collect_profiles.sh) - only when it was collected from THESE kernel
    sources: the summary records the hash of csrc/ + include/qgtc.h it ran (qgtc_ppopp22_amd/_build.py::kernel_source_hash).
    Returns (summary or None, a short string that says where the counter figures come from or why there are none)."""
    from qgtc_ppopp22_amd._build import kernel_source_hash

    rel = f"{PROFILE_DIR}/summary_{name}.json"
    here = kernel_source_hash()
    try:
        with open(os.path.join(ROOT, rel)) as f:
            summ = json.load(f)
    except (OSError, ValueError):
        return None, f"none: {rel} absent"
    got = summ.get("kernel_source_hash")
    if got != here:
        return None, f"none: {rel} is of sources {got}, tree is {here}"
    return summ, f"{rel} (pmc, sources {here})"


def hip_events_us(fn, reps=100, untimed=3):
    """Mean microseconds of `fn` between two HIP events on the current stream (torch's current stream is the one the
    library launches on)."""
    for _ in range(untimed):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
