"""PlannedEpoch (device-filled plan) against BatchedEpoch (host-built plan): every variant on the 'tiny' graph, sharded and not,
one and several epochs."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
from qgtc_ppopp22_amd.sampler import ClusterIter

g = G.make_graph("tiny", 40)
dev = torch.device("cuda:0")
bad = 0
for gin in (False, True):
    for bits in (2, 4):
        for chain in ("correct", "reference"):
            for ids in (None, [0, 2, 4, 6, 8], [1, 3, 5, 7, 9]):
                random.seed(2)
                it = ClusterIter("tiny", g, 40, 4, bit_width=bits, run_GIN=gin, device=dev, qgtc=Q, batch_ids=ids, with_rows_X=True)
                W = driver.pack_weights(Q, 32, 64, 10, bits, dev)
                host = driver.BatchedEpoch(Q, it.cTensor_li, it.cluster_param_li, W, bits, chain, gin)
                ref = [o.clone() for o in host.run()]
                for kw in ({}, {"fuse": False}, {"chain_stages": False}, {"keep_aggregates": True}):
                    plan = driver.PlannedEpoch(Q, it.epoch_data(Q), it.cluster_param_li, W, bits, chain, gin, **kw)
                    for rep in range(3):
                        plan.run()
                        torch.cuda.synchronize()
                        ok = all(torch.equal(a, b) for a, b in zip(plan.outs, ref))
                        if not ok:
                            bad += 1
                            wrong = [i for i, (a, b) in enumerate(zip(plan.outs, ref)) if not torch.equal(a, b)]
                            print("MISMATCH", "gin" if gin else "gcn", bits, chain, ids, kw, "rep", rep, "batches", wrong)
                            break
print("bad", bad)
