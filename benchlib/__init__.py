"""The sections of bench.py (the driver-facing entry point at the repository root)."""
