"""Minimal stand-in for the un-vendored `radiotools` dependency (>=0.2.1).

Used ONLY by tests/golden/gen/*.py inside the build container to import the
reference's pure-Python path; it never ships to the GPU box as product code.
Only the trivial trigonometry the hot path uses is restated here.
"""
