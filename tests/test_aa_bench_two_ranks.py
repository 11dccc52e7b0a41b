"""`bench.py --gpus 2` end to end on the one-GPU box: the parent starts two rank processes, both bind cuda:0 (test hook
QGTC_BENCH_SHARE_GPU) and talk over gloo (RCCL refuses two ranks on one device) - the whole N > 1 flow of main(): timed region with
barriers, max over ranks, checksum gather, the strong- and the weak-scaled epoch legs with their end-of-epoch gathers, ONE line from
rank 0. What the driver runs on an 8-GPU node as `--gpus 2 / 4 / 8` (BASELINE.json configs[4]) with backend nccl."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["summaries", "outputs"])
def test_bench_with_two_ranks_sharing_the_gpu(gather):
    env = dict(os.environ, QGTC_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    extras = os.path.join("/tmp", f"qgtc_bench2_{gather}.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "10", "--warmup", "3",
                          "--gather", gather, "--extras-file", extras], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert all(l.startswith("[Gloo]") for l in lines[:-1]), out.stdout[-2000:]   # (gloo announces its ranks on stdout; nothing else but the line)
    assert len(lines[-1]) < 4096                                                # rank 0 only, one short line, LAST
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 10 and line["scaling"] == "weak" and line["value"] > 0
    ex = json.load(open(extras))
    assert len(ex["rank_checksums"]) == 2 and all(c != 0 for c in ex["rank_checksums"])   # (int32 words summed: every output bit is set here)
    strong, weak = ex["cluster_gcn_epoch_ogbn_arxiv_shape"], ex["cluster_gcn_epoch_ogbn_arxiv_shape_weak_scaled"]
    assert strong["gathered_batches"] == 75 and weak["gathered_batches"] == 150             # 75 round-robin; 75 per rank
    assert strong["batched_correct_chain_ms"] > 0 and weak["batches_per_second"] > 0
    assert ex["batched_gin_epoch_ppi_shape_4bit"]["gathered_batches"] == 75
    if gather == "outputs":
        assert strong["gathered_output_bytes"] > 75 * 1100 * 10 * 4
