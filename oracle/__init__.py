"""ORACLE -- CPU restatements of the reference hot path (test infrastructure only).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``audiodeepfake-detection_amd/``) never does and fails
loudly when its HIP library is missing.
"""
