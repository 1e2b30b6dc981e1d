"""Pieces of bench.py: shapes and peaks (shapes), HIP-event timers and the committed-profile
look-ups (timers), the CPU baselines -- the only importers of oracle/ on the bench side (baselines)
-- and the formatter of the final line (format)."""
