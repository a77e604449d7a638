"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see shepseg_oracle.c header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
