#!/bin/bash
# ADVICE r3 / VERDICT r3 weak 9: what can still be learnt, WITHOUT a GPU, about the round-1 wrong-result / hang / memory fault.
# The misbehaving builds were uncommitted variants of the round-1 kernel; what IS in history is the revision they were edits of (40d496b).
# This script takes that revision's phase functions and its own emulator out of git, poisons the emulated LDS image with signalling NaNs
# (the round-1 KERNEL initialised only Z and LAM of a valid instance: everything else started as whatever the CU's LDS held), builds it
# with -fsanitize=address,undefined at the image's exact size and runs multi-step feedback rollouts at the three shapes whose GPU
# instantiations misbehaved: rollout_kernel<16, true> (dual-pole cart), <32, false> (8-body chain), <64, false> (17-body chain).
# Result (round 4): ARCHAEOLOGY_OK -- no out-of-range LDS offset, no read of a slot the kernel had not written, no undefined behaviour,
# every state finite and every step converged.  (first row.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
mkdir -p $W/constrainedcontrol.jl_amd/csrc $W/tests/emu $W/include
for f in cclqr_dev.h cclqr_internal.h cclqr_lin_dev.h cclqr_newton.h cclqr_tables.h; do git -C $R show 40d496b:constrainedcontrol.jl_amd/csrc/$f > $W/constrainedcontrol.jl_amd/csrc/$f; done
git -C $R show 40d496b:tests/emu/emu_rollout.cpp > $W/tests/emu/emu_rollout.cpp
git -C $R show 40d496b:include/cclqr.h > $W/include/cclqr.h
python3 - $W/tests/emu/emu_rollout.cpp <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = "        for (int e = 0; e < Y.total; e++) L[e] = 0.0;"
assert old in s
s = s.replace(old, "        for (int e = 0; e < Y.total; e++) L[e] = std::numeric_limits<double>::signaling_NaN();\n        for (int e = 0; e < 5 * nb; e++) L[Y.LAM + e] = 0.0;")
s = s.replace("#include <vector>", "#include <vector>\n#include <limits>", 1)
open(p, "w").write(s)
PY
cp $R/tools/archaeology_r1_driver.cpp $W/driver.cpp
cd $W
/opt/rocm/lib/llvm/bin/clang++ -x hip --offload-host-only -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -ffp-contract=off -I/opt/rocm/include -o r1_asan tests/emu/emu_rollout.cpp driver.cpp 2>/dev/null
./r1_asan
