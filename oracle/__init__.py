"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference algorithms (see oracle/nrmc_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
