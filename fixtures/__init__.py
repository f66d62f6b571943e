"""Seeded synthetic inputs / weights / PLDA parameters for tests, bench.py and the oracle goldens.
Not part of the product package (wespeaker_amd never imports it)."""
