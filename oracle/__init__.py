"""TEST INFRASTRUCTURE: CPU oracle (restatement) and reference-build bindings.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
