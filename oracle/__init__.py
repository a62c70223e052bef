"""ORACLE package: CPU restatements of the reference's hot path. TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg - never by
nerfail_amd/ (the product fails loudly without its HIP library; there is no CPU fallback).
Pinned against tests/golden/*.npz, which were produced by running the reference itself
(tests/golden/make_golden.py).
"""
