#!/usr/bin/env python3
"""Extracts the numbers of the reference's robot description (examples/examples_files/sawyer_arm.urdf: link masses, COM
offsets, inertia tensors, joint frames and axes) into tests/golden/sawyer_arm_tables.json.  Data only; run where
/root/reference exists."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as g
pkg = g.load_package()
tab = pkg.parse_urdf("/root/reference/examples/examples_files/sawyer_arm.urdf")
json.dump(tab, open(os.path.join(HERE, "sawyer_arm_tables.json"), "w"), indent=1)
print(len(tab["links"]), "links", len(tab["joints"]), "joints")
