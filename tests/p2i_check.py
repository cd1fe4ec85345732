"""The winner-id comparison of the p2i max splat lives with the oracle (oracle/p2i_check.py), so that
`__graft_entry__.smoke()` applies the same rule as the tests; re-exported here for the tests' imports."""
from oracle.p2i_check import _value, assert_ids_exact_up_to_ulp_ties  # noqa: F401
