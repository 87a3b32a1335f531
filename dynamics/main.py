#!/usr/bin/env python3
"""`python dynamics/main.py <flags of dynamics/train_dynamics_2d.sh>`: trains the 2-D dynamics model on the HIP path."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dgdm_amd.dynamics.main import train  # noqa: E402
from dgdm_amd.dynamics.parser import parse  # noqa: E402

if __name__ == "__main__":
    train(parse(sys.argv[1:]))
