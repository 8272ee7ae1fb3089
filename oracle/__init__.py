"""oracle/ — CPU restatements of the reference's hot-path algorithms.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker — the product (mangatranslator_amd/) never does.
"""
