#!/usr/bin/env python3
"""Extracts the numbers of the reference's robot descriptions (examples/examples_files/sawyer_arm.urdf, and the whole robot sawyer.urdf with its
fixed joints: link masses, COM offsets, inertia tensors, joint frames and axes) into tests/golden/sawyer_arm_tables.json and
tests/golden/sawyer_full_tables.json.  Data only; run where /root/reference exists."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as g
pkg = g.load_package()
tab = pkg.parse_urdf("/root/reference/examples/examples_files/sawyer_arm.urdf")
json.dump(tab, open(os.path.join(HERE, "sawyer_arm_tables.json"), "w"), indent=1)
print(len(tab["links"]), "links", len(tab["joints"]), "joints")
full = pkg.parse_urdf("/root/reference/examples/examples_files/sawyer.urdf", keep_fixed=True)
json.dump(full, open(os.path.join(HERE, "sawyer_full_tables.json"), "w"), indent=1)
print(len(full["links"]), "links", len(full["joints"]), "joints,", sum(j["type"] == "fixed" for j in full["joints"]), "of them fixed")
