"""RCCL has to have run on the hardware at least once (missing 2): BASELINE.json configs[4] gathers per-epoch
results over RCCL / xGMI, and a one-GPU box never reaches those calls through the world == 1 short-cuts. Here a FRESH child
process (benchmarks/rccl_check.py) sets up a world-size-1 process group BEFORE touching the GPU and drives DEVICE tensors
through every exchange of qgtc_ppopp22_amd/dist.py - all_gather_into_tensor of 75 ragged per-batch outputs (padded) and of
the per-batch summaries, all_reduce(MAX), barrier - with the short-cuts bypassed (a live group always gets the real
collective). The nccl (= RCCL) result must equal the gloo result of the same script on host tensors.
(File name: collected first, like test_aa_two_ranks_one_gpu.py, so the children start before this process holds the GPU.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(backend, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "rccl_check.py"), backend], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("RCCL_WORLD1 ")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0][len("RCCL_WORLD1 "):])


def test_forced_collectives_at_world_1_over_gloo():
    """CPU: a live world-1 group runs the real collectives (no short-cut) and they return what the short-cut would."""
    r = _run("gloo")
    assert r["ok"] is True and r["backend"] == "gloo" and r["ranks_seen"] == 1 and r["device"] == "cpu"
    assert r["gathered_bytes"] == 75 * 1239 * 10 * 4 and r["max_over_ranks"] == 1.25


@pytest.mark.gpu
def test_rccl_world_1_moves_device_tensors_and_equals_gloo():
    r = _run("nccl")
    g = _run("gloo")
    assert r["ok"] is True, r
    assert r["backend"] == "nccl" and r["ranks_seen"] == 1 and r["device"].startswith("cuda")
    for k in ("checksum", "summaries_checksum", "gathered_bytes", "max_over_ranks"):
        assert r[k] == g[k], (k, r[k], g[k])
