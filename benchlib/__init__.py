"""The probes behind bench.py's secondary objects, one module per BASELINE config; bench.py keeps the headline and the line."""
