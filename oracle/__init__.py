"""Test infrastructure only: CPU restatements of the reference E2FGVI forward.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
"""
