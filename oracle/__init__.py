"""CPU oracle for the folve convolution hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (folve_amd/) never does.
"""
from .oracle import *  # noqa: F401,F403
