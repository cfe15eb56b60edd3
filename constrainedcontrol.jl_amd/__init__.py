"""cclqr — MI355X-native batched LQR rollout behind ConstrainedControl.jl's LQR / TrackingLQR / Mechanism surface.

The directory name carries a dot, so import it through `__graft_entry__.load_package()` (registers it as `cclqr`).
Arithmetic on the hot path happens only in csrc/ (HIP, through the C-ABI of include/cclqr.h); this package is
the host-side mirror of the reference's plugin interface and never falls back to a CPU implementation.
"""
from .mechanism import (Body, Box, EqualityConstraint, FixedOrientation, MechTables, Mechanism, Origin, Prismatic, Quaternion, Revolute, RotX, RotY, RotZ,
                        getid, mechanism_from_urdf_tables, minimal_to_maximal, one_quaternion, parse_urdf, qconj, qmul, setJointPosition, setPosition, setVelocity, vrotate, joint_position_states, urdf_lump_fixed)
from . import examples
from . import _capi
from . import dist
from .lqr import LQR, PID, BatchState, Controller, OpenLoop, Storage, TrackingLQR, control_lqr, on_device, setForce, simulate, state_error
