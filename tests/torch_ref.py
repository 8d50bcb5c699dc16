"""The independent torch-CPU formulation lives in oracle/torch_oracle.py (bench.py's cpu_baseline leg times it
too); the tests keep importing it under this name."""
from oracle.torch_oracle import *                      # noqa: F401,F403
from oracle.torch_oracle import _same_pad, _t          # noqa: F401
