#!/usr/bin/env python3
"""`python generator/train.py ...` as the reference's guided_sample_{2d,3d}.sh call it."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dgdm_amd.generator.train import main, train  # noqa: E402,F401

if __name__ == "__main__":
    main()
