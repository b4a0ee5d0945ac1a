"""TEST INFRASTRUCTURE — CPU oracle for the render hot path (see oracle/oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
