"""Name-only stand-in for NVIDIA Isaac Gym, used ONLY by tests/golden/make_fixtures.py.

Isaac Gym is a closed binary that is absent from this pipeline.
This package lets the reference's pure-PyTorch task functions be imported so their outputs
can be recorded as golden vectors.  It contains no physics.  `torch_utils` restates the
publicly documented helper semantics; every other module only
provides the names the reference files mention at import time.
"""
