"""oracle — CPU restatement of the reference's FCOS/HISFCOS hot path.  TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package."""
