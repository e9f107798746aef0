"""CPU oracle for the STMask hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  Nothing under ``stmask_amd/`` imports it (tests/test_no_oracle_in_product.py
enforces that).  See ``stm_oracle.c`` for the pinning status of each function.
"""
from .oracle import *  # noqa: F401,F403
