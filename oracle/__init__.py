"""CPU oracle for the registration hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package (it is the checker, never the thing shipped or
measured).  The product package ``probabilistic_point_clouds_registration_amd``
never imports it.

``binding`` wraps ``libppcr_oracle.so`` (built by ``oracle/Makefile`` from
``ppcr_oracle.c``) with ctypes.
"""
