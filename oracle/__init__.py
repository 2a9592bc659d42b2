"""CPU oracle package — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package
(capsaicin_amd) never imports it.  See cap_oracle.h for the parity statement ("parity unpinned" at the
TraceRay boundary; pure functions pinned by tests/golden/kat.json).
"""
