"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/cclqr.h declares (no compute
calls without a GPU), the mechanism mirror reproduces the reference scripts' placements, sharding helpers, gloo gather."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_library_exports_every_declared_symbol(cclqr):
    import __graft_entry__ as graft
    if not os.path.exists(cclqr._capi.LIB_PATH):
        graft.build()
    hdr = open(os.path.join(ROOT, "include", "cclqr.h")).read()
    declared = sorted(set(re.findall(r"\b(cclqr_[a-z_]+)\s*\(", hdr)))
    assert len(declared) >= 14
    lib = ctypes.CDLL(cclqr._capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libcclqr.so does not export %s" % name
    assert sorted(cclqr._capi.EXPORTS) == declared
    assert cclqr._capi.lib().cclqr_version() >= 100


def test_every_entry_point_cites_the_reference():
    hdr = open(os.path.join(ROOT, "include", "cclqr.h")).read()
    for name in ("cclqr_linearize", "cclqr_riccati", "cclqr_riccati_tracking", "cclqr_rollout", "cclqr_mech_create", "cclqr_ctrl_create"):
        i = hdr.index("int " + name)
        assert re.search(r"\.jl:\d+", hdr[max(0, i - 900):i]), name


def test_product_has_no_cpu_fallback(cclqr, monkeypatch):
    capi = cclqr._capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libcclqr.so")
    with pytest.raises(ImportError):
        capi.lib()
    src = "".join(open(os.path.join(ROOT, "constrainedcontrol.jl_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "constrainedcontrol.jl_amd"))
                  if f.endswith(".py"))
    assert "oracle" not in src.replace("the oracle", "").replace("# oracle", ""), "the product must not import or call the oracle"


def test_mechanism_mirror_reproduces_script_placements(cclqr):
    ex = cclqr.examples.cartpole_n(1)
    mech = ex["mech"]
    assert [cclqr.getid(b) for b in ex["bodies"]] == [1, 2] and [cclqr.getid(j) for j in ex["joints"]] == [3, 4]
    z = mech.state()
    assert np.allclose(z[0, 0:3], [0, 0.5, 0])                               # setPosition!(origin, cart, Δx=[0;0.5;0])
    assert np.allclose(z[1, 3:7], [np.cos(0.1), np.sin(0.1), 0, 0])          # Δq = RotX(0.2)
    assert np.allclose(z[1, 0:3], [0, 0.5 - 0.5 * np.sin(0.2), 0.5 * np.cos(0.2)])
    # desired states of the scripts are consistent with setPosition! (SURVEY 8a-bis cross-check)
    exp = cclqr.examples.pendulum(θ0=np.pi)
    assert np.allclose(exp["mech"].state()[0, 0:3], exp["xd"][0], atol=1e-15)
    n = 5
    exn = cclqr.examples.cartpole_n(n, y0=0.0, φ0=[0.0] * n)
    assert np.allclose(exn["mech"].state()[1:, 2], [i + 0.5 for i in range(n)])
    # batch initial-state generator == the script's setPosition! sequence
    φ = np.array([0.3, -0.2, 0.1, 0.05, -0.4])
    exr = cclqr.examples.cartpole_n(n, y0=0.17, φ0=φ)
    assert np.abs(cclqr.examples.cartpole_states(n, [0.17], φ[None])[0] - exr["mech"].state()).max() < 1e-14
    t = exr["mech"].tables()
    assert t.nb == 6 and t.mx == 72 and t.ml == 30 and list(t.parent) == [-1, 0, 1, 2, 3, 4]
    assert np.allclose(t.inertia[1].reshape(3, 3), np.diag([1.01, 1.01, 0.02]) / 12)


def test_mechanism_topologies(cclqr):
    o = cclqr.Origin()
    a, b = cclqr.Box(1, 1, 1, 1), cclqr.Box(1, 1, 1, 1)
    j1 = cclqr.EqualityConstraint(cclqr.Revolute(o, a, [1, 0, 0]))
    j2 = cclqr.EqualityConstraint(cclqr.Revolute(a, b, [1, 0, 0]))
    j3 = cclqr.EqualityConstraint(cclqr.Revolute(o, b, [1, 0, 0]))
    m = cclqr.Mechanism(o, [a, b], [j1, j2, j3])       # closed loop (lqr_deltabot.jl): accepted, flagged, rolled out by rollout_loop.hip
    t = m.tables()
    assert m.has_loops and (t.nb, t.ne, t.ml) == (2, 3, 15)
    assert not cclqr.Mechanism(o, [a, b], [j1, j2]).has_loops
    with pytest.raises(ValueError):
        cclqr.Mechanism(o, [a, b], [j1])               # a body without any joint
    ex = cclqr.examples.deltabot()                     # examples/lqr_deltabot.jl:25-41
    t = ex["mech"].tables()
    assert ex["mech"].has_loops and (t.nb, t.ne) == (5, 7) and list(t.type) == [0, 0, 0, 0, 2, 0, 0]
    assert sum(len(e) for e in ex["mech"].eqconstraints) == 33


def test_shard_bounds_cover_everything(cclqr):
    d = cclqr.dist
    for n in (0, 1, 7, 8, 65536, 65537):
        for w in (1, 2, 3, 8):
            b = [d.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


_GLOO_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
pkg = g.load_package()
rank, world, local = pkg.dist.init_from_env(backend="gloo")
n_total = 11
z0 = np.arange(n_total * 2 * 13, dtype=np.float64).reshape(n_total, 2, 13)
out = pkg.dist.sharded_rollout_numpy(lambda blk: blk * 2.0 + rank * 0.0, z0, rank, world)
t = pkg.dist.max_over_ranks(1.0 + rank)
# chunked trajectory collection (dist.TrajectoryGather): 4 chunks of 6 steps, slabs reused (2 buffers), final Storage layout on rank 0
n_local, T, nb, H = 5, 24, 3, 4
full = (np.arange(n_local * T * nb * 13, dtype=np.float64).reshape(n_local, T, nb, 13) + 1e6 * rank)
tg = pkg.dist.TrajectoryGather(rank, world, n_local, T, nb, H, "cpu")
for c in range(H):
    tg.wait_slab_free(c)
    tg.slab(c).copy_(torch.from_numpy(full[:, c * (T // H):(c + 1) * (T // H)]))     # stands for the chunk's rollout launch
    tg.submit(c)
traj = tg.finish()
if rank == 0:
    assert out.shape == z0.shape and np.array_equal(out, z0 * 2.0), "gather mismatch"
    assert t == float(world)
    want = np.concatenate([np.arange(n_local * T * nb * 13, dtype=np.float64).reshape(n_local, T, nb, 13) + 1e6 * r for r in range(world)])
    assert traj.shape == (n_local * world, T, nb, 13) and np.array_equal(traj.numpy(), want), "chunked trajectory gather mismatch"
    assert tg.bytes_gathered == n_local * T * nb * 13 * 8 * (world - 1)
    print("GLOO_OK")
else:
    assert out is None and traj is None
import torch.distributed as _d
_d.barrier(); _d.destroy_process_group()
'''


def test_workspace_cache_bookkeeping_across_devices(tmp_path):
    """the per-thread device workspace cache of capi.hip (csrc/cclqr_wscache.h) releases every cached block on ITS device
    when the thread has switched GPU and never hands a block of another device out; driven on the CPU with a fake allocator, under
    ASAN/UBSAN"""
    exe = str(tmp_path / "ws_cache_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-fsanitize=address,undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "emu", "ws_cache_test.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "WS_CACHE_OK" in r.stdout, r.stdout + r.stderr


def test_bench_self_launch_starts_the_ranks_before_touching_the_gpu(tmp_path):
    """`python bench.py --gpus 2` outside a torchrun environment spawns `python -m torch.distributed.run
    --nproc-per-node 2 bench.py ...` as a child and relays its exit code; the parent path imports neither torch nor the HIP library
    (here, without a GPU, both ranks end with bench.py's 'needs a GPU' and the parent reports the failure instead of printing a line)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[src.index("def main():"):src.index("    import torch\n", src.index("def main():"))]
    assert "self_launch(" in head and "load_package" not in head                  # the launch decision comes before any GPU-touching import
    body = src[src.index("def self_launch"):src.index("def build_workload")]
    assert "torch.distributed.run" in body and "--nproc-per-node" in body and "os.exec" not in body and "import torch" not in body
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: tests/test_gpu_rollout.py::test_bench_self_launches_its_ranks runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--instances", "8", "--sim-steps", "8",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "needs a GPU" in (r.stdout + r.stderr)
    assert not [x for x in r.stdout.splitlines() if x.strip().startswith("{")]


def test_no_silent_gloo_fallback():
    """dist.init_from_env never replaces RCCL by gloo on its own: the only way to gloo is to ask for it by name"""
    src = open(os.path.join(ROOT, "constrainedcontrol.jl_amd", "dist.py")).read()
    body = src[src.index("def init_from_env"):src.index("def gather_to_root")]
    assert "except" not in body and body.count("init_process_group") == 2
    bench = open(os.path.join(ROOT, "bench.py")).read()
    timed = bench[bench.index("    def one_rollout(timed):"):bench.index("    for _ in range(args.warmup):")]
    assert "torch.empty" not in timed and "torch.zeros" not in timed and "torch.cat" not in timed and "gather_to_root" not in timed
    assert 'dist.get_backend() == "nccl"' in bench and "--allow-gloo" in bench


def test_gloo_world2_shard_and_gather(tmp_path):
    """the N>1 path of bench.py with 2 CPU ranks: shard instances, collect the trajectory chunks (slab reuse, final Storage layout
    bit for bit) and the final states on rank 0"""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT})
    from conftest import free_port
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GLOO_OK" in r.stdout


_GLOO_WORKER8 = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
pkg = g.load_package()
rank, world, local = pkg.dist.init_from_env(backend="gloo")
assert world == 8
# uneven shards: 43 instances over 8 ranks = 6 6 6 5 5 5 5 5; 8 chunks of 3 steps; two slabs reused four times each
n_total, T, nb, H = 43, 24, 3, 8
lo, hi = pkg.dist.shard_bounds(n_total, rank, world)
n_local = hi - lo
whole = np.arange(n_total * T * nb * 13, dtype=np.float64).reshape(n_total, T, nb, 13) * 0.5 + 7.0
tg = pkg.dist.TrajectoryGather(rank, world, n_local, T, nb, H, "cpu", n_total=n_total)
assert tg.nmax == 6 and sum(tg.sizes) == n_total
for c in range(H):
    tg.wait_slab_free(c)
    tg.slab(c)[:n_local].copy_(torch.from_numpy(whole[lo:hi, c * (T // H):(c + 1) * (T // H)]))     # stands for the chunk's rollout launch
    tg.submit(c)
traj = tg.finish()
zT = pkg.dist.RootGather(n_total, (nb, 13), torch.float64, "cpu", rank, world)(torch.from_numpy(whole[lo:hi, -1].copy()))
worst = pkg.dist.max_over_ranks(10.0 + rank)
plan = pkg.dist.collection_plan(rank, world, n_local, T, nb, H, True, "trajectory", n_total=n_total)
held = sum(x.numel() * 8 for x in tg.slabs) + (tg.out.numel() * 8 + sum(b.numel() * 8 for pair in tg.recv for b in pair) if rank == 0 else 0)
assert abs(plan["total"] - held) <= 2 * (2 * n_local * nb * 13 * 8 + 4 * H * n_local + 5 * nb * 8 * n_local) + (8 * 6 + n_total + 6) * nb * 13 * 8, (plan, held)
if rank == 0:
    assert traj.shape == whole.shape and np.array_equal(traj.numpy(), whole), "chunked trajectory gather mismatch (world 8, uneven shards)"
    assert np.array_equal(zT.numpy(), whole[:, -1]) and worst == 17.0
    print("GLOO8_OK")
else:
    assert traj is None and zT is None
torch.distributed.barrier(); torch.distributed.destroy_process_group()      # every rank leaves together (a rank that exits while a peer's gloo threads
                                                                            # still talk to it aborts in its teardown under load)
'''


def test_gloo_world8_uneven_shards_eight_chunks(tmp_path):
    """the collection path of an 8-GPU run rehearsed with 8 CPU ranks over gloo -- uneven shards (43 instances: 6 6 6 5 5 5 5 5,
    slabs padded to the largest so that one fixed-size gather per chunk moves everything), eight chunks through two slabs, the assembled
    Storage layout and the final-state gather bit-identical to the unsharded arrays, max-over-ranks timing; and rank 0's allocation plan
    (dist.collection_plan, what bench.py checks against the free HBM) equal to what the collectors really hold"""
    script = tmp_path / "worker8.py"
    script.write_text(_GLOO_WORKER8 % {"root": ROOT})
    from conftest import free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GLOO8_OK" in r.stdout


def test_collection_plan_of_the_headline_run():
    """the bytes bench.py announces before it allocates (and refuses on): rank 0 of 8 x 8192 instances x 1000 steps x 17 bodies holds the 116 GB
    Storage layout + 29 GB of receive slabs + its own slabs when trajectories are collected, 14.8 GB when only final states travel"""
    import __graft_entry__ as graft
    d = graft.load_package().dist
    p = d.collection_plan(0, 8, 8192, 1000, 17, 8, True, "trajectory")
    assert abs(p["total"] / 1e9 - 148.7) < 0.5 and p["assembled Storage layout out[n_total][T][nb][13] (rank 0)"] == 65536 * 1000 * 17 * 13 * 8
    assert d.collection_plan(3, 8, 8192, 1000, 17, 8, True, "trajectory")["total"] < 4e9
    f = d.collection_plan(0, 8, 8192, 1000, 17, 1, True, "final")
    assert abs(f["total"] / 1e9 - 14.8) < 0.2 and f["own recorded trajectory [n_local][T][nb][13]"] == 8192 * 1000 * 17 * 13 * 8
    assert d.collection_plan(0, 1, 8192, 1000, 17, 1, True, "trajectory")["total"] == f["total"] - (8 * 8192 + 65536 + 8192) * 17 * 13 * 8


_URDF = """<?xml version="1.0"?>
<robot name="two_link">
  <link name="base"><inertial><origin xyz="0 0 0.1" rpy="0 0 0"/><mass value="3"/><inertia ixx="1" ixy="0" ixz="0" iyy="1" iyz="0" izz="1"/></inertial></link>
  <link name="l0"><inertial><origin xyz="0.1 0.02 0.3" rpy="0 0 0"/><mass value="2.5"/><inertia ixx="0.05" ixy="0.001" ixz="-0.002" iyy="0.06" iyz="0.003" izz="0.02"/></inertial></link>
  <link name="l1"><inertial><origin xyz="0 -0.05 0.2" rpy="0 0 0"/><mass value="1.5"/><inertia ixx="0.03" ixy="0" ixz="0.001" iyy="0.03" iyz="0" izz="0.01"/></inertial></link>
  <joint name="j0" type="revolute"><origin xyz="0 0 0.08" rpy="0 0 0"/><parent link="base"/><child link="l0"/><axis xyz="0 0 1"/></joint>
  <joint name="j1" type="revolute"><origin xyz="0.081 0.05 0.237" rpy="-1.5707963 1.5707963 0"/><parent link="l0"/><child link="l1"/><axis xyz="0 0 1"/></joint>
</robot>
"""


def test_urdf_subset_reader(cclqr, orc, tmp_path):
    """Mechanism(path, floating=false, g=0.0) (examples/lqr_sawyer.jl:9): COM-frame bodies, full inertia, rpy joint frames"""
    f = tmp_path / "two_link.urdf"
    f.write_text(_URDF)
    mech = cclqr.Mechanism(str(f), floating=False, g=0.0)
    assert len(mech.bodies) == 2 and [e.name for e in mech.eqconstraints] == ["j0", "j1"]
    t = mech.tables()
    assert list(t.parent) == [-1, 0] and np.allclose(t.mass, [2.5, 1.5])
    assert np.allclose(t.inertia[0].reshape(3, 3), [[0.05, 0.001, -0.002], [0.001, 0.06, 0.003], [-0.002, 0.003, 0.02]])
    assert np.allclose(t.p1[0], [0, 0, 0.08]) and np.allclose(t.p2[0], [-0.1, -0.02, -0.3])
    assert np.allclose(t.p1[1], np.array([0.081, 0.05, 0.237]) - [0.1, 0.02, 0.3])
    z = mech.state()
    assert np.abs(orc.constraints(t, z)).max() < 1e-14          # zero pose satisfies every joint
    assert np.allclose(z[0, 0:3], [0.1, 0.02, 0.38])
    # the joint axis really is the free direction: rotate joint 1 and the constraints still hold
    cclqr.setJointPosition(mech, mech.geteqconstraint("j1"), 0.7)
    assert np.abs(orc.constraints(t, mech.state())).max() < 1e-14
    with pytest.raises(NotImplementedError):
        cclqr.Mechanism(str(f), floating=True)


def test_sawyer_fixture_matches_reference_urdf_when_present(cclqr):
    import json
    tab = json.load(open(os.path.join(ROOT, "tests", "golden", "sawyer_arm_tables.json")))
    assert len(tab["joints"]) == 7 and len(tab["links"]) == 8
    ref = "/root/reference/examples/examples_files/sawyer_arm.urdf"
    if os.path.exists(ref):
        assert cclqr.parse_urdf(ref) == tab
    ex = cclqr.examples.sawyer(tab)
    assert len(ex["mech"].bodies) == 7 and ex["mech"].tables().ml == 35      # Nb = 7, mλ = 35


_URDF_FIXED = """<?xml version="1.0"?>
<robot name="lumped">
  <link name="base"/>
  <link name="arm"><inertial><origin xyz="0 0 0.5" rpy="0 0 0"/><mass value="2.0"/><inertia ixx="0.2" ixy="0" ixz="0" iyy="0.2" iyz="0" izz="0.01"/></inertial></link>
  <link name="tool"><inertial><origin xyz="0.1 0 0" rpy="0 0 0"/><mass value="1.0"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.03" iyz="0" izz="0.03"/></inertial></link>
  <link name="marker"/>
  <link name="finger"><inertial><origin xyz="0 0 0.05" rpy="0 0 0"/><mass value="0.2"/><inertia ixx="0.001" ixy="0" ixz="0" iyy="0.001" iyz="0" izz="0.0002"/></inertial></link>
  <joint name="shoulder" type="revolute"><origin xyz="0 0 0.1" rpy="0 0 0"/><parent link="base"/><child link="arm"/><axis xyz="1 0 0"/></joint>
  <joint name="tool_mount" type="fixed"><origin xyz="0 0 1.0" rpy="0 0 1.5707963267948966"/><parent link="arm"/><child link="tool"/></joint>
  <joint name="marker_mount" type="fixed"><origin xyz="0.2 0 0" rpy="0 0 0"/><parent link="tool"/><child link="marker"/></joint>
  <joint name="grip" type="prismatic"><origin xyz="0.2 0 0" rpy="0 0 0"/><parent link="tool"/><child link="finger"/><axis xyz="1 0 0"/></joint>
</robot>
"""


def test_urdf_fixed_joints_are_lumped(cclqr, orc, tmp_path):
    """a URDF `fixed` joint (examples_files/sawyer.urdf has 15) welds its child link to its parent: the reader lumps the two into one rigid body --
    summed mass, common COM, parallel-axis inertia, the child's own joints re-anchored through the composed transform"""
    f = tmp_path / "lumped.urdf"
    f.write_text(_URDF_FIXED)
    with pytest.raises(ValueError):
        cclqr.parse_urdf(str(f))                                   # the strict reader still refuses what is not a 1-DoF joint
    lump = cclqr.urdf_lump_fixed(cclqr.parse_urdf(str(f), keep_fixed=True))
    assert sorted(lump["links"]) == ["arm", "base", "finger"] and [j["name"] for j in lump["joints"]] == ["shoulder", "grip"]
    arm = lump["links"]["arm"]
    # tool frame = arm frame moved to (0, 0, 1) and turned 90 deg about z: its COM (0.1, 0, 0) sits at (0, 0.1, 1) in the arm frame
    assert np.isclose(arm["mass"], 3.0) and np.allclose(arm["com"], (2.0 * np.array([0, 0, 0.5]) + 1.0 * np.array([0, 0.1, 1.0])) / 3.0)
    com = np.array(arm["com"])
    S = lambda d: (d @ d) * np.eye(3) - np.outer(d, d)
    I_tool_in_arm = np.diag([0.03, 0.01, 0.03])                    # diag(ixx, iyy, izz) with x and y swapped by the quarter turn
    I = np.diag([0.2, 0.2, 0.01]) + 2.0 * S(np.array([0, 0, 0.5]) - com) + I_tool_in_arm + 1.0 * S(np.array([0, 0.1, 1.0]) - com)
    got = arm["inertia"]
    assert np.allclose([[got[0], got[1], got[2]], [got[1], got[3], got[4]], [got[2], got[4], got[5]]], I)
    grip = lump["joints"][1]
    assert grip["parent"] == "arm" and np.allclose(grip["xyz"], [0.0, 0.2, 1.0])          # (0.2, 0, 0) in the tool frame
    mech = cclqr.Mechanism(str(f), floating=False, g=-9.81)
    t = mech.tables()
    assert t.nb == 2 and list(t.parent) == [-1, 0] and np.allclose(t.mass, [3.0, 0.2])
    assert np.abs(orc.constraints(t, mech.state())).max() < 1e-14
    # the finger slides along the TOOL's x axis = the arm's y axis
    cclqr.setJointPosition(mech, mech.geteqconstraint("grip"), 0.05)
    z = mech.state()
    assert np.abs(orc.constraints(t, z)).max() < 1e-14 and np.allclose(z[1, 0:3], [0.0, 0.25, 1.1 + 0.05])


def test_sawyer_full_fixture_is_a_branching_robot(cclqr, orc):
    """examples_files/sawyer.urdf (24 links, 15 fixed joints; tests/golden/sawyer_full_tables.json holds its numbers): lumped it is the arm of
    sawyer_arm.urdf plus the head on the first link -- eight bodies, a BRANCHING tree"""
    import json
    tab = json.load(open(os.path.join(ROOT, "tests", "golden", "sawyer_full_tables.json")))
    assert len(tab["links"]) == 24 and sum(j["type"] == "fixed" for j in tab["joints"]) == 15
    ref = "/root/reference/examples/examples_files/sawyer.urdf"
    if os.path.exists(ref):
        assert cclqr.parse_urdf(ref, keep_fixed=True) == tab
    mech = cclqr.mechanism_from_urdf_tables(tab, g=0.0)
    t = mech.tables()
    assert t.nb == 8 and t.ne == 8 and sorted(e.name for e in mech.eqconstraints) == sorted(["head_pan"] + ["right_j%d" % i for i in range(7)])
    assert sum(1 for a in t.parent if a == 0) == 2                      # head and right_l1 both hang off right_l0
    assert np.abs(orc.constraints(t, mech.state())).max() < 1e-14
    total = sum(L["mass"] for L in tab["links"].values())
    lumped = cclqr.urdf_lump_fixed(tab)["links"]
    assert np.isclose(sum(L["mass"] for L in lumped.values()), total)


def test_minimal_to_maximal_kinematics(cclqr, orc):
    """6-argument linearsystem's setpoint conversion (lqr.jl:80): joint coordinates/rates -> consistent maximal state"""
    ex = cclqr.examples.cartpole_n(3)
    mech = ex["mech"]
    ids = [cclqr.getid(j) for j in ex["joints"]]
    θ = np.array([0.3, 0.2, -0.4, 0.1])
    θd = np.array([0.5, -1.0, 0.7, 0.2])
    xd, vd, qd, ωd = cclqr.minimal_to_maximal(mech, ids, θ, θd)
    t = mech.tables()
    z = np.zeros((4, 13))
    for i in range(4):
        z[i, 0:3], z[i, 3:7], z[i, 7:10], z[i, 10:13] = xd[i], qd[i], vd[i], ωd[i]
    assert np.abs(orc.constraints(t, z)).max() < 1e-14
    assert np.allclose(z, np.concatenate([cclqr.examples.cartpole_states(3, [0.3], [[0.2, -0.4, 0.1]])[0][:, :7], z[:, 7:]], axis=1))
    # velocities are the time derivative of the forward kinematics
    h = 1e-6
    xp = cclqr.minimal_to_maximal(mech, ids, θ + h * θd, θd)[0]
    xm = cclqr.minimal_to_maximal(mech, ids, θ - h * θd, θd)[0]
    for i in range(4):
        assert np.allclose((xp[i] - xm[i]) / (2 * h), vd[i], atol=1e-8)
    assert np.allclose(ωd[1], [θd[1], 0, 0]) and np.allclose(ωd[3], [θd[1] + θd[2] + θd[3], 0, 0])


def _build_c_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "c_abi_cartpole")
    libdir = os.path.join(ROOT, "constrainedcontrol.jl_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_cartpole.c"),
                           "-o", exe, "-L", libdir, "-lcclqr", "-lm", "-Wl,-rpath," + libdir])
    return exe


def test_c_example_compiles_and_links_against_the_abi(cclqr, tmp_path):
    """examples/c_abi_cartpole.c uses include/cclqr.h from plain C (no Python, no torch): it must compile warning-free and link"""
    import __graft_entry__ as graft
    if not os.path.exists(cclqr._capi.LIB_PATH):
        graft.build()
    exe = _build_c_example(tmp_path)
    assert os.path.exists(exe)
    # the part of it that needs no GPU: cclqr_version + cclqr_abi_layout against the C compiler's own sizeof / offsetof of include/cclqr.h
    import subprocess
    out = subprocess.run([exe, "--abi"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "abi %d layout ok" % cclqr._capi.ABI_VERSION, out.stdout + out.stderr


def test_abi_layout_is_what_the_bindings_mirror(cclqr):
    """cclqr_abi_layout (include/cclqr.h): the library's sizeof / offsetof of the four structs that cross the boundary equal the ctypes mirrors
    of _capi.py field by field (checked at every load too), a deliberately wrong mirror is caught, and julia/CCLQR.jl declares the same fields in
    the same order with the same widths (it cannot be executed here: compared as text against the ctypes mirror)"""
    import ctypes as C
    import re
    capi = cclqr._capi
    L = capi.lib()
    want = capi.mirrored_layout()
    got = (C.c_int32 * 64)()
    n = L.cclqr_abi_layout(got, C.c_int32(64))
    assert n == len(want) == 48 and list(got)[:n] == want
    assert L.cclqr_abi_layout(None, C.c_int32(0)) == 48                    # size query

    class Wrong(C.Structure):                                                # a mirror that forgot a field: size and the offsets behind it differ
        _fields_ = [f for f in capi.RolloutOpts._fields_ if f[0] != "pid_state_len"]
    assert C.sizeof(Wrong) != want[39] and Wrong.newton_eps_alone.offset != want[47]
    # the Julia mirror, as text: field names and types in order
    jl = open(os.path.join(ROOT, "julia", "CCLQR.jl")).read()
    width = {"Int32": 4, "Int64": 8, "UInt64": 8, "Float64": 8}
    for jname, S in (("MechDesc", capi.MechDesc), ("CtrlDesc", capi.CtrlDesc), ("RiccatiOpts", capi.RiccatiOpts), ("RolloutOpts", capi.RolloutOpts)):
        body = re.search(r"struct %s\b.*?\n(.*?)\nend" % jname, jl, re.S).group(1)
        body = re.sub(r"#.*", "", body)
        fields = re.findall(r"(\w+)::(Ptr\{\w+\}|\w+)", body)
        assert [f for f, _ in fields] == [f for f, _ in S._fields_], jname
        for (f, ty), (_, cty) in zip(fields, S._fields_):
            assert (8 if ty.startswith("Ptr") else width[ty]) == C.sizeof(cty), (jname, f)
    assert "ABI_VERSION = %d" % capi.ABI_VERSION in jl and "cclqr_abi_layout" in jl


def _kernel_resources(tmp_path, src_name, match):
    import subprocess
    asm = str(tmp_path / (src_name + ".s"))
    src = os.path.join(ROOT, "constrainedcontrol.jl_amd", "csrc", src_name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=fast", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                           "-o", asm, src], stderr=subprocess.DEVNULL)
    txt = open(asm).read()
    kernels = {}
    for blk in txt.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if match not in name:
            continue
        g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
        kernels[name] = dict(agpr=int(blk.split()[0]), vgpr=g("vgpr_count"), scratch=g("private_segment_fixed_size"),
                             lds=g("group_segment_fixed_size"), sgpr_spill=g("sgpr_spill_count"), vgpr_spill=g("vgpr_spill_count"))
    return kernels


def test_chain_rollout_kernel_resources(tmp_path):
    """the register-resident chain kernel (csrc/rollout_chain.hip), every instantiation: (lanes, layout links) in
    {(8, 4), (16, 8), (32, 16), (32, 17), (32, 32), (64, 64)} x control variant {plain LQR, + friction/noise, + PID, + friction/noise with the Philox
    sample generated in the kernel (the step-per-launch form of configs[4], round 5)}.  The plain-LQR instantiations
    -- every BASELINE config but the friction/noise law of config 5 -- must not spill a single scalar register and must not touch
    scratch memory (the kernel must not live in the regime where a spilled pointer or mask can go wrong);
    nor does the friction/noise variant; the PID variant may spill a few scalars (atan2 constants) but no vector register to scratch either."""
    kernels = _kernel_resources(tmp_path, "rollout_chain.hip", "rollout_chain_kernel")
    # 7 shapes (lanes, layout links, lanes per link, links per sub-lane group): since round 5 the 1- and 2-link mechanisms give a link three lanes
    # (<8, 4, 3, 2> next to <8, 4, 1, 8> for 3-4 links) x 4 control variants with the exact Newton rule + the 7 plain-law kernels of the measured-error
    # Newton mode (RELAX)
    assert len(kernels) == 35, sorted(kernels)
    assert sum("ELb1ELi" in name for name in kernels) == 7
    shapes = set()
    for name, k in kernels.items():
        assert k["lds"] == 0 and k["scratch"] == 0 and k["vgpr"] <= 512, (name, k)
        m = re.search(r"ILi(\d+)ELi(\d+)ELi(\d)ELb([01])ELi(\d)ELi(\d+)EEEv", name)
        G, nbp, variant, relax, kl, nl = (int(x) for x in m.groups())
        shapes.add((G, nbp, kl, nl))
        assert kl * nl <= G and (kl > 1 or nl == G), name
        reduction_level = G == 32 and nbp <= 17
        if variant == 0:
            assert k["sgpr_spill"] == 0, (name, k)
            # No scalar spill (pointers, lane masks) and no scratch anywhere.  "vgpr_spill" counts values the allocator moves to AGPRs after
            # its first pass -- the same v_accvgpr moves as the ~230 values it places there itself, never memory (scratch == 0 above): the
            # 32-lane kernels with the reduction level and the two-level / partner-assisted line search carry up to 8 of them, the rest none
            # (the 8- and 16-lane kernels carry up to 8 since their line search hands trial points to idle groups: TrialIn, round 4)
            # (the measured-error 8-lane kernel: 12 since the prologue stopped clearing the LDS image)
            assert k["vgpr_spill"] <= (16 if G == 8 else 8 if (reduction_level or G <= 16) else 0), (name, k)
            # the instantiations with the odd-even reduction level (32 lanes, <= 17 links) park more values in AGPRs around it
            assert k["vgpr"] <= (504 if reduction_level else 440), (name, k)
        elif variant == 1:
            assert k["sgpr_spill"] == 0, (name, k)
        else:       # PID (atan2 constants) / Philox in the kernel (a call): a few scalars saved around them
            assert k["sgpr_spill"] <= 16, (name, k)
    assert shapes == {(8, 4, 3, 2), (8, 4, 1, 8), (16, 8, 1, 16), (32, 16, 1, 32), (32, 17, 1, 32), (32, 32, 1, 32), (64, 64, 1, 64)}, shapes


def test_headline_kernel_instruction_budget(tmp_path):
    """ISA-level regressions of the headline kernel rollout_chain_kernel<32, 17, 0, false> that cost time without changing a result (round 3 found
    two by reading the assembly): every DPP shift must be a lone v_mov_b32_dpp with bound_ctrl -- without it the compiler zero-initialises the
    destination first (265 extra v_mov_b32) --, and an LDS operand of a lane-conditional select must not come back as per-element EXEC-masked
    loads.  Bounds = today's counts + a margin; a failure here means: read the block that grew."""
    import subprocess
    asm = str(tmp_path / "rollout_chain.s")
    src = os.path.join(ROOT, "constrainedcontrol.jl_amd", "csrc", "rollout_chain.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=fast", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", asm, src],
                          stderr=subprocess.DEVNULL)
    lines = open(asm).read().splitlines()
    start = [i for i, l in enumerate(lines) if l.startswith("_ZN5cclqr20rollout_chain_kernelILi32ELi17ELi0ELb0ELi1ELi32EEEvNS_11RolloutArgsE:")][0]
    end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
    ops = [l.split()[0] for l in (x.strip() for x in lines[start:end]) if l and not l.startswith((";", ".")) and not l.endswith(":")]
    count = lambda prefix: sum(o.startswith(prefix) for o in ops)
    assert len(ops) <= 10800, len(ops)                                   # 10 648 today (10 941 before the two fixes)
    assert count("v_mov_b32_e32") <= 200, count("v_mov_b32_e32")         # 152 today (417 with zero-initialised DPP destinations)
    dpp = [x.strip() for x in lines[start:end] if "v_mov_b32_dpp" in x]
    assert dpp and all("bound_ctrl" in x for x in dpp)
    assert count("s_and_saveexec_b64") + count("s_or_saveexec_b64") <= 135      # 126 today


def test_linearize_and_riccati_kernel_resources(tmp_path):
    """csrc/linearize.hip and csrc/riccati.hip: no scratch memory and no vector-register spill in any kernel; the scalar spills the
    compiler makes today (to VGPR lanes, never to memory) are recorded as upper bounds so that an edit that pushes a kernel further
    into the spilling regime fails here, on the CPU, before it reaches a GPU"""
    lin = _kernel_resources(tmp_path, "linearize.hip", "linearize_kernel")
    assert len(lin) == 2, sorted(lin)
    for name, k in lin.items():
        assert k["scratch"] == 0 and k["vgpr_spill"] == 0 and k["vgpr"] <= 512 and k["lds"] == 0, (name, k)
        assert k["sgpr_spill"] <= (25 if "ILb1E" in name else 8), (name, k)      # (25 / 8 since the chain masks are 64 bits wide: 64-link chains; 23 / 6 before)
    ric = _kernel_resources(tmp_path, "riccati.hip", "ric")
    assert len(ric) >= 24, sorted(ric)
    for name, k in ric.items():
        if "riccati_resident_kernel" in name and not re.search(r"ILi\dELi0ELi0EEEv", name):
            assert k["sgpr_spill"] <= 69, (name, k)          # the specialised shapes
    # riccati_resident_kernel<MU, NG, BF>: the register-fragment specialisations (NG > 0: the BASELINE shapes) spill 0 .. 69 scalars, the
    # generic forms (any mx: operands streamed through the double-buffered tiles, unrolled per-input loops) up to 136 -- to VGPR lanes, never
    # to memory
    bound = {"riccati_resident_kernel": 136, "ric_gain_update_kernel": 42}
    for name, k in ric.items():
        assert k["scratch"] == 0 and k["vgpr_spill"] == 0, (name, k)
        # every riccati kernel runs >= 2 wavefronts per SIMD (512-thread workgroups / tiles): <= 256 registers per lane
        assert k["vgpr"] <= 256, (name, k)
        lim = max([v for key, v in bound.items() if key in name] + [0])
        assert k["sgpr_spill"] <= lim, (name, k)


def test_loop_rollout_kernel_resources(tmp_path):
    """the closed-loop rollout kernel (csrc/rollout_loop.hip; one instantiation per width of the register-resident rows of the dense
    system, 8 .. 64 columns) cross-compiles for gfx950 without scratch memory: the one access into the row that the register file cannot
    index -- the lane's own entry in the pivot column -- is a compare-and-select inside a uniform branch, not an indexed array"""
    kernels = _kernel_resources(tmp_path, "rollout_loop.hip", "rollout_loop_kernel")
    assert len(kernels) == 8, sorted(kernels)
    for name, k in kernels.items():
        # two wavefronts per SIMD (eight instances per CU instead of four): 256 registers and a small, bounded spill of the evaluation and
        # assembly phases (measured: +19 % against the 307-register build without scratch at six per CU, +28 % more at eight per CU once the
        # dense system is assembled in registers and its 12.6 KB leave the LDS image)
        assert k["scratch"] <= 384 and k["vgpr"] <= 256 and k["lds"] == 0, (name, k)


def test_tree_rollout_kernel_resources(tmp_path):
    """every instantiation of the register-resident tree kernel (csrc/rollout_treereg.hip; (lanes, layout links) x control variant + the
    measured-error Newton mode) cross-compiles for gfx950 within one wavefront's register file (512 VGPR + AGPR per lane) WITHOUT scratch
    memory -- the child / sibling lists and the schedule records are indexed by compile-time constants only (an index the compiler cannot
    resolve puts them into scratch: 128 bytes in the first cut) -- and without vector-register spills to memory; the plain-LQR and
    friction/noise instantiations spill no scalar register, the PID ones a few (atan2 constants)"""
    kernels = _kernel_resources(tmp_path, "rollout_treereg.hip", "rollout_treereg_kernel")
    assert len(kernels) == 44, sorted(kernels)          # {(16,4), (16,8), (32,8), (32,10), (32,12), (32,14), (32,16), (32,24), (32,32), (64,48), (64,64)} x 4
    for name, k in kernels.items():
        assert k["vgpr"] <= 512 and k["lds"] == 0, (name, k)      # all LDS is dynamic (one instance image per lane group)
        assert k["scratch"] == 0 and k["vgpr_spill"] <= 8, (name, k)
        assert k["sgpr_spill"] <= (16 if "ELi2ELb0EEEv" in name else 0), (name, k)


def test_controlfunction_helpers_on_the_host(cclqr, orc):
    """the building blocks a custom controlfunction closure is given (lqr.py: BatchState, state_error, setForce): the batched error
    coordinates equal the per-instance statement of lqr.jl:92-103 in oracle/loops.py, setForce broadcasts and maps constraint ids to joints.
    (No device call: the closure path itself is covered by tests/test_gpu_setup.py::test_custom_controlfunction_closure.)"""
    from oracle import loops
    ex = cclqr.examples.cartpole_n(2)
    mech = ex["mech"]
    rng = np.random.default_rng(2)
    z = cclqr.examples.cartpole_states(2, rng.uniform(-0.5, 0.5, 5), rng.uniform(-0.4, 0.4, (5, 2)))
    z[:, :, 7:13] = rng.normal(size=(5, 3, 6)) * 0.1
    zd = cclqr.examples.cartpole_states(2, [0.1], [[0.3, -0.2]])

    class Ctl:
        pass
    c = Ctl(); c.zd = zd
    batch = cclqr.BatchState(mech, z, 1)
    dz = cclqr.state_error(batch, c, 1)
    for i in range(5):
        assert np.abs(dz[i] - loops.state_error(z[i], zd[0])).max() < 1e-15
    cclqr.setForce(batch, mech.eqconstraints[1], 2.0)
    cclqr.setForce(batch, cclqr.getid(mech.eqconstraints[0]), np.arange(5.0))
    assert sorted(batch.u) == [0, 1] and np.array_equal(batch.u[1], np.full(5, 2.0)) and np.array_equal(batch.u[0], np.arange(5.0))
    # the same helpers on a torch batch (what a closure registered with cclqr.on_device is handed; here the tensors simply live on the CPU):
    # state_error / control_lqr in torch equal the numpy forms, setForce takes tensors and scalars
    import torch
    c.K = rng.normal(size=(7, 1, 36)); c.Fd = rng.normal(size=(7, 1)); c.N = 8; c.eqcids = [cclqr.getid(mech.eqconstraints[0])]
    c.zd = np.repeat(zd, 7, 0) + rng.normal(size=(7, 3, 13)) * 0.01
    tb = cclqr.BatchState(mech, torch.from_numpy(z.copy()), 3)
    assert tb.on_device and not batch.on_device
    assert np.abs(cclqr.state_error(tb, c, 3).numpy() - cclqr.state_error(batch, c, 3)).max() < 1e-15
    un = cclqr.control_lqr(cclqr.BatchState(mech, z, 3), c, 3)
    ut = cclqr.control_lqr(tb, c, 3)
    assert np.abs(ut.numpy() - un).max() < 1e-13 and np.abs(tb.u[0].numpy() - un[:, 0]).max() < 1e-13
    cclqr.setForce(tb, mech.eqconstraints[1], 2.0)
    assert tb.u[1].shape == (5,) and float(tb.u[1][4]) == 2.0
    assert cclqr.control_lqr(cclqr.BatchState(mech, torch.from_numpy(z.copy()), 8), c, 8).abs().max() == 0.0      # k < N gate (lqr.jl:106)

