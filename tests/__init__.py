"""Parity tests of tfdiffeq_amd: `-m "not gpu"` = oracle vs golden fixtures + host logic, `-m gpu` = the HIP path vs the oracle."""
