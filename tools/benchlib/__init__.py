"""Helpers of bench.py (repo root): input generators, the pipelined workload, roofline / counter bookkeeping, CPU baselines.
bench.py holds the measurement contract (timed_loop, the JSON line); everything here is plumbing around it."""
