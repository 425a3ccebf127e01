"""ORACLE package - test infrastructure only (see oracle/ode_numpy.py header).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never from the product package tfdiffeq_amd.
"""
