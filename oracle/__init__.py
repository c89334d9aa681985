"""Parity oracle (CPU, numpy).  TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  See oracle/README.md."""
