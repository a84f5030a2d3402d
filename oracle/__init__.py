"""CPU oracle for the MMBERT hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is imported by the product package (``mm-vqa_amd/``).
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may use it, and only as the checker / reported baseline.
"""
