"""The kernel solves the forward dynamics by composite inertias + Newton-Euler bias forces + leg-wise block elimination
(openroborl_amd/csrc/orr_physics.h, leg_dynamics / row_response), the oracle by the articulated-body algorithm.
tools/crba_proto.py is the numpy statement of the kernel's formulation; here it is checked against the oracle
(accelerations and the full inverse mass matrix) on random states of both robots.  CPU only."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_block_elimination_matches_articulated_body_algorithm():
    import crba_proto
    crba_proto.main()   # asserts 1e-9 on accelerations (relative) and on M^-1
