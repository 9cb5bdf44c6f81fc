"""Seeded synthetic parameters (in the reference's checkpoint formats) and synthetic audio.  No trained weights or
audio files exist offline: tests, bench.py and smoke() all draw from here.  Data only -- no
algorithm of the hot path lives in this package."""
