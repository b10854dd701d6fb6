"""CPU oracle for the linear-association hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product (normalisr_amd/) never does.  See oracle/normalisr_oracle.c and oracle/reference_path.py.
"""
from .reference_path import (  # noqa: F401
	ensure_built, pvalues, beta_cdf_half, inv_rank, association_test_1, association_tests, de, coex,
	block_plain_c, pearson_r_t, bh, binnet, normvar, normvar1)
