"""CPU oracle for the LiteralKG hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``literalkg_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and there only as the checker / the timed CPU baseline.
"""
