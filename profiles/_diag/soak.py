"""Full-length soak (tests/soak.py::run_soak; the suite runs it at reduced length, test_soak_reduced):
   python profiles/_diag/soak.py [steps] [arenas] [peds]
Defaults 1500 steps x 256 arenas: 2 passes (scan stack 1 and 2) = 2 x 384 k env-steps compared bit for bit."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in ("nav-gym_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from soak import run_soak

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
E = int(sys.argv[2]) if len(sys.argv) > 2 else 256
peds = len(sys.argv) > 3 and sys.argv[3] == "peds"
run_soak(steps, E, peds, log=lambda m: print(m, flush=True))
