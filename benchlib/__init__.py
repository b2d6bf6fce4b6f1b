"""The probes behind bench.py's secondary objects, one module per BASELINE config; bench.py keeps the headline and the line.

This package is PART OF bench.py (split out of it in round 6 for readability), not of the product: like bench.py it may use the
CPU oracle, and does so only where bench.py did -- as the timed CPU leg (`cpu_baseline`) and as the checker of a parity figure,
always outside a timed GPU region.  Nothing under `nerfmeshes_amd/` imports it or the oracle."""
