"""Re-export of the oracle loader (oracle/pyoracle.py) for the tests."""
from oracle.pyoracle import *  # noqa: F401,F403
from oracle.pyoracle import lib, call, grid_build, orb_params, orb_tables, orb_extract, orb_level, orb_candidates, frame_pipeline  # noqa: F401
