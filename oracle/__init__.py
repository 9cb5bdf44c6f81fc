"""TEST INFRASTRUCTURE: CPU restatement of the reference's hot-path algorithms (the parity oracle).
Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this package; the product
(aicovergen_amd/) never does."""
