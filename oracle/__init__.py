"""CPU oracle for the kangaroo jump path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (kangaroo_amd) never does.  See oracle/kng_oracle.h for the reference citations.
"""
from .binding import Oracle, load_oracle  # noqa: F401
