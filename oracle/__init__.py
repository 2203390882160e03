"""TEST INFRASTRUCTURE ONLY: CPU parity oracles for the MI355X pool-scoring path.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this package.
See enet_oracle.c / enet_oracle.py (C restatement, fixed accumulation order) and
torch_restatement.py (independent torch-CPU restatement).  Parity against real TensorFlow is
unpinned (no TF, no reference fixtures): SURVEY.md 8c, DESIGN.md.
"""
