"""TEST INFRASTRUCTURE: CPU oracle (plain C + numpy) for the semantic_slam hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
