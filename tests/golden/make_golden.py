#!/usr/bin/env python3
"""Regenerate the golden fixtures in this directory from the CPU oracle.

The reference has no golden vectors for this path and its kernel cannot run
in the build container, so these fixtures are ORACLE outputs: they pin (a) the oracle against
silent drift across compilers / CPUs (tests/test_oracle_golden.py reproduces them bit for bit
on whatever host runs the suite) and (b) the HIP kernels on the GPU box, where the reference
tree does not exist.  The decoded scene table is derived from the data file
oclpathtracer_amd/data/cornellbox.bin (a copy of the reference's test/cornellbox.bin).

Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oclpathtracer_amd import scene  # noqa: E402
from oracle import ptoracle  # noqa: E402

CASES = [  # name, W, H, frames, max_bounces
    ("cornell_64x64_f1_d16", 64, 64, 1, 16),
    ("cornell_64x64_f2_d16", 64, 64, 2, 16),
    ("cornell_64x64_f8_d16", 64, 64, 8, 16),
    ("cornell_64x64_f8_d2", 64, 64, 8, 2),
    ("cornell_40x24_f5_d16", 40, 24, 5, 16),  # ragged: W*H % 64 != 0, W != H
]


def main():
    ptoracle.build()
    tris, mats = scene.load_model()
    table = {
        "triangles": [{"p1": t["p1"].tolist(), "p2": t["p2"].tolist(), "p3": t["p3"].tolist(), "id": int(t["id"])} for t in tris],
        "materials": [{"albedo": m["albedo"].tolist(), "emissive": m["emissive"].tolist(),
                       "roughness": float(m["roughness"]), "type": int(m["type"])} for m in mats],
    }
    with open(os.path.join(HERE, "cornell_scene_table.json"), "w") as f:
        json.dump(table, f, indent=1)
    stats = {}
    for name, W, H, frames, depth in CASES:
        fb, st = ptoracle.render(tris, mats, W, H, frames, max_bounces=depth, want_stats=True)
        np.save(os.path.join(HERE, name + ".npy"), fb)
        stats[name] = {"W": W, "H": H, "frames": frames, "max_bounces": depth, **st}
    with open(os.path.join(HERE, "work_counters.json"), "w") as f:
        json.dump(stats, f, indent=1)
    print("wrote", len(CASES), "framebuffers + scene table + work counters")


if __name__ == "__main__":
    main()
