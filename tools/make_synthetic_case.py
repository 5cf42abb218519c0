#!/usr/bin/env python
"""Write the synthetic sphere capture to data/<case>/ in the reference's on-disk layout, so the drop-in entry
point can be exercised end to end:

    python tools/make_synthetic_case.py [--root data] [--views 24 --height 480 --width 270]
    python PMVO.py --yaml=configs/reconstruct/synthetic_sphere
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monohair_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--root", default="data")
ap.add_argument("--case", default="synthetic_sphere")
ap.add_argument("--views", type=int, default=24)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=270)
a = ap.parse_args()
print("written:", synth.write_case(a.root, a.case, a.views, a.height, a.width))
