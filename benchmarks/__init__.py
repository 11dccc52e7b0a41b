"""Measurement legs of bench.py, one module per concern:

  common.py    constants (peaks, the reference's published numbers), workload builder, clock warm-up, profile lookup
  headline.py  the timed region of the contract (K bitMM2Bit launches) and the dominant kernel's roofline block
  cpu.py       cpu_baseline: the C oracle on the host cores + the DGL-style fp32 CPU epoch (baselines, not targets)
  epochs.py    Cluster-GCN / Batched-GIN epoch legs, their roofline blocks, zero-tile rows, README's epoch table
  tables.py    the reference's benchmark tables (2_7c, 5_9, Fig. 8a int8 comparison, width sweep) - extras file only
  launcher.py  `--gpus N` without a launcher, and the GPU-free dry run of the collectives

bench.py prints ONE short JSON line; everything a table needs goes to the extras file the line names.
"""
