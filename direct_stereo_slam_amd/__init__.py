"""MI355X-native direct photometric hot path of IRVLab/direct_stereo_slam.

Package layout: csrc/ (HIP kernels + C ABI), host/ (C++ adaptor with the reference's class
surface), tracker.py / ringdb.py (Python mirror of the same interface for tests and bench),
synth.py (seeded synthetic stereo scenes).  The compute path is the shared library
lib/libdsm_hotpath.so; importing the package does not load it, using it does (and fails loudly
if it is missing -- there is no CPU fallback).
"""
__version__ = "0.1.0"
