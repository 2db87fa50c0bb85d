"""CPU oracle for the SCDA hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  Nothing under scda_amd/ imports it.
"""
