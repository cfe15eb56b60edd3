# export_reference_fixtures.jl -- run by a maintainer who HAS Julia + ConstrainedDynamics 0.9.x + ConstrainedControl 0.3.0, once:
#
#     julia --project=<an environment with both packages> tools/export_reference_fixtures.jl  [output directory, default tests/golden]
#
# STATUS: source only.  There is no Julia in this repository's build / test pipeline (SURVEY.md 8c), so this file has never been executed here;
# it only uses what the reference's own example scripts use (cited line by line below) plus `linearsystem`, `simulate!` and `Storage` exactly as
# src/control/lqr.jl:63 and src/control/lqr_tracking.jl:32-35 call them.  It imports nothing from this repository and writes nothing but data.
#
# What it writes: for each BASELINE config one directory  <out>/ref_<config>/  of .npy files (a 20-line writer below: no NPZ.jl / JSON.jl needed)
# -- the numbers SURVEY.md 8c lists "to capture if a Julia environment ever becomes available":
#     z0[nb][13]                       the initial state the run starts from (x, q = (s, v1, v2, v3), v, ω per body, as body.state holds it)
#     A, Bu, Bl, G                     linearsystem(...) at the setpoint (lqr.jl:63)                       (configs 1-4)
#     K_all[nK][mu][12 nb]             lqr.K[k][i] (lqr.jl:4);  K_distinct_from = first k with K[k] !== K[k+1] (the back-fill of lqr.jl:179-181)
#     zd[nb][13], Fd[mu], Q[12nb][12nb] (already times Δt), R[mu][mu], N, dt, g
#     k_list, storage_x[len(k_list)][nb][3], storage_q[..][4], storage_v[..][3], storage_w[..][3]
#                                      storage.{x,q,v,ω}[body][k] of simulate!(mech, tend, lqr, record = true) at k in {1, 2, 10, 100, last}
#     (config 5) U[1000], storage0_* of the open-loop swing-up, K_all of the TrackingLQR, storage_* of the run under the PACKAGE's law
#                                      control_trackinglqr! (lqr_tracking.jl:46-71): the script's own law adds friction and randn() noise,
#                                      which cannot be reproduced; the friction / noise option of this repository is checked against its oracle
#     newton_iters[20]                 Newton iterations the dependency's newton! takes in each of the first 20 controlled steps (round 5: the one
#                                      recollected rule that costs 45 % of the rollout kernel -- does a solve stop on ||f|| < ε AND ||Δs|| < ε, or on the
#                                      residual alone?).  Measured without touching the solver's internals: the mechanism is deep-copied inside the control
#                                      callback of every step (after the law has set its forces), and the copy is solved with newtonIter = 1, 2, ... until the
#                                      solution equals the one of the default call bit for bit -- a converged newton! returns early, so that is its count
#     newton_defaults[3]               ε, newtonIter, lineIter of newton! as its method declares them (NaN where the declaration could not be read)
#     versions[4][3]                   major, minor, patch of ConstrainedDynamics, ConstrainedControl, StaticArrays, Julia
#
# tests/test_reference_fixtures.py consumes these directories when they exist (on the CPU oracle and on the HIP path) and answers, the moment a
# file appears: do the gains agree, does the linear model agree, does the integrator agree, and WHICH KNOT does Storage record (DESIGN.md 2).
using ConstrainedDynamics
using ConstrainedControl
using LinearAlgebra
import Pkg

const OUT = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")

# ---------------------------------------------------------------- .npy writer (format 1.0, little-endian, Fortran order = Julia's own layout)
function write_npy(path::String, a::AbstractArray{T}) where {T<:Union{Float64,Int64}}
    descr = T == Float64 ? "<f8" : "<i8"
    shape = ndims(a) == 1 ? "($(length(a)),)" : "(" * join(size(a), ", ") * ")"
    hdr = "{'descr': '$descr', 'fortran_order': True, 'shape': $shape, }"
    pad = 64 - mod(10 + length(hdr) + 1, 64)
    hdr = hdr * " "^pad * "\n"
    open(path, "w") do io
        write(io, UInt8[0x93]); write(io, "NUMPY"); write(io, UInt8[1, 0]); write(io, UInt16(length(hdr)))
        write(io, hdr); write(io, Array{T}(a))
    end
end
save(dir, name, a::AbstractArray) = write_npy(joinpath(dir, name * ".npy"), a isa AbstractArray{<:Integer} ? Int64.(a) : Float64.(a))
save(dir, name, x::Real) = save(dir, name, [Float64(x)])

# ---------------------------------------------------------------- small accessors (the only ConstrainedDynamics internals touched)
# scalar-first components of a unit quaternion: `imag` is what the reference itself imports (src/ConstrainedControl.jl:4)
quat4(q) = [hasproperty(q, :w) ? q.w : q.s; ConstrainedDynamics.imag(q)...]
body_state(b) = [b.state.xc...; quat4(b.state.qc)...; b.state.vc...; b.state.ωc...]              # examples/lqr_sawyer.jl:16-17 read xc, qc like this
state_matrix(mech) = permutedims(hcat([body_state(b) for b in mech.bodies]...))                 # [nb][13]
function storage_samples(storage, nb, ks)
    x = zeros(length(ks), nb, 3); q = zeros(length(ks), nb, 4); v = zeros(length(ks), nb, 3); w = zeros(length(ks), nb, 3)
    for (j, k) in enumerate(ks), i = 1:nb                                                        # storage.x[i][k]: lqr_tracking.jl:32-35
        x[j, i, :] = storage.x[i][k]; q[j, i, :] = quat4(storage.q[i][k]); v[j, i, :] = storage.v[i][k]; w[j, i, :] = storage.ω[i][k]
    end
    return x, q, v, w
end
gains(K) = begin                                                                                 # K[k][i] is a 1 x 12nb row (lqr.jl:4)
    nK, mu, mx = length(K), length(K[1]), length(K[1][1])
    out = zeros(nK, mu, mx)
    for k = 1:nK, i = 1:mu
        out[k, i, :] = vec(K[k][i])
    end
    out
end
distinct_from(K) = begin k = 1; while k < length(K) && K[k] === K[k+1]; k += 1; end; k end          # the converged gain is aliased: lqr.jl:179-181
sample_ks(n) = unique(filter(k -> k <= n, [1, 2, 10, 100, n]))

# ---------------------------------------------------------------- the stopping rule of the dependency's newton!, by observation
const NEWTON_STEPS = 20
# the solution a (copied) mechanism holds after newton! with an iteration cap: v+, ω+ of every body (what lqr.jl:98-103 reads as vsol[2], ωsol[2]) and,
# where the constraint type has the field, the multipliers
function solution_after(snapshot, cap)
    m = deepcopy(snapshot)
    ConstrainedDynamics.newton!(m, newtonIter = cap)
    s = Float64[]
    for b in m.bodies
        append!(s, b.state.vsol[2]); append!(s, b.state.ωsol[2])
    end
    for e in m.eqconstraints
        hasproperty(e, :λsol) && append!(s, e.λsol[2])
    end
    return s
end
function newton_iterations(snapshot; maxit = 100)
    ref = solution_after(snapshot, maxit)
    for n = 1:maxit
        solution_after(snapshot, n) == ref && return n
    end
    return maxit
end
# simulate! under `controller` for NEWTON_STEPS steps with a wrapper callback that keeps a copy of the mechanism as newton! is about to see it
function newton_iteration_counts(mech, controller)
    snaps = Vector{Any}(undef, NEWTON_STEPS)
    storage = Storage{Float64}(Base.OneTo(NEWTON_STEPS), length(mech.bodies))
    wrapped!(m, k) = begin
        controller.control!(m, controller, k)                        # the plugin contract: lqr.jl:14, :89; lqr_tracking.jl:14, :46
        snaps[k] = deepcopy(m)
    end
    simulate!(mech, storage, wrapped!, record = true)                 # the closure form of simulate!: trackingLQR_triple_cartpole.jl:46-53
    # (if the dependency moves the inputs from the constraints onto the bodies between control! and newton!, a stand-alone newton! on the copy still
    # does the same, because that transfer is part of evaluating the residual; should a version do it elsewhere, the counts below describe a step without
    # its input -- tests/test_reference_fixtures.py then reports "neither rule" rather than a wrong answer)
    return Int64[newton_iterations(snaps[k]) for k = 1:NEWTON_STEPS]
end
# ε, newtonIter, lineIter as the method of newton! declares them: read as NUMBERS out of the declaration (no source text is kept)
function newton_defaults()
    out = [NaN, NaN, NaN]
    try
        m = first(methods(ConstrainedDynamics.newton!))
        file, line = String(m.file), m.line
        txt = join(readlines(file)[line:min(line + 3, end)], " ")
        for (i, key) in enumerate(["ε", "newtonIter", "lineIter"])
            mt = match(Regex(key * "\\s*=\\s*([0-9.eE+-]+)"), txt)
            mt === nothing || (out[i] = parse(Float64, mt.captures[1]))
        end
    catch err
        @warn "newton! declaration not readable" err
    end
    return out
end
function package_versions()
    v = zeros(Int64, 4, 3)
    names = ["ConstrainedDynamics", "ConstrainedControl", "StaticArrays"]
    try
        for (uuid, info) in Pkg.dependencies()                        # (Julia >= 1.4; the reference's compat is 1.6 - 1.8, Project.toml:14)
            i = findfirst(==(info.name), names)
            (i === nothing || info.version === nothing) && continue
            v[i, :] = [info.version.major, info.version.minor, info.version.patch]
        end
    catch err
        @warn "package versions not readable" err
    end
    v[4, :] = [VERSION.major, VERSION.minor, VERSION.patch]
    return v
end
function save_newton_facts(dir, mech, controller)
    save(dir, "newton_defaults", newton_defaults())
    save(dir, "versions", package_versions())
    try       # on a COPY of the placed mechanism (bodies, multipliers, everything): the run the other arrays come from starts from an untouched one
        save(dir, "newton_iters", newton_iteration_counts(deepcopy(mech), controller))
    catch err
        @warn "Newton iteration counts not recorded" err
    end
end

function export_lqr(name, mech, bodyids, eqcids, Q, R, horizon, tend; xd, qd, Fτd = [[0.0] for _ in eqcids])
    dir = joinpath(OUT, "ref_" * name); mkpath(dir)
    nb = length(mech.bodies)
    vd = [zeros(3) for _ = 1:nb]; ωd = [zeros(3) for _ = 1:nb]
    save(dir, "z0", state_matrix(mech))
    save(dir, "dt", mech.Δt); save(dir, "g", mech.g)
    A, Bu, Bλ, G = linearsystem(mech, xd, vd, qd, ωd, Fτd, bodyids, eqcids)                       # the call of lqr.jl:63
    save(dir, "A", A); save(dir, "Bu", Bu); save(dir, "Bl", Bλ); save(dir, "G", G)
    lqr = LQR(mech, bodyids, eqcids, Q, R, horizon, xd = xd, qd = qd, Fτd = Fτd)
    save(dir, "K_all", gains(lqr.K)); save(dir, "K_distinct_from", [distinct_from(lqr.K)])
    save(dir, "Q", cat(Q..., dims = (1, 2)) * mech.Δt); save(dir, "R", cat(R..., dims = (1, 2)) * mech.Δt)   # lqr.jl:18-19
    save(dir, "N", [horizon == Inf ? 0 : Int(ceil(horizon / mech.Δt))])                           # lqr.jl:21-27 (0 = the Inf-horizon controller)
    zd = zeros(nb, 13)
    for i = 1:nb
        zd[i, :] = [xd[i]...; quat4(qd[i])...; 0; 0; 0; 0; 0; 0]
    end
    save(dir, "zd", zd); save(dir, "Fd", [f[1] for f in Fτd])
    save(dir, "ctrl_joint_ids", Int64.(eqcids)); save(dir, "body_ids", Int64.(bodyids))
    save_newton_facts(dir, mech, lqr)                                                             # (the controller holds ids, not the mechanism: lqr.jl:3-15)
    storage = simulate!(mech, tend, lqr, record = true)                                           # examples/lqr_cartpole.jl:44
    ks = sample_ks(length(storage.x[1]))
    x, q, v, w = storage_samples(storage, nb, ks)
    save(dir, "k_list", ks); save(dir, "storage_x", x); save(dir, "storage_q", q); save(dir, "storage_v", v); save(dir, "storage_w", w)
    println("wrote ", dir)
end

# ================================================================ configs[0]  examples/lqr_pendulum.jl
let
    joint_axis = [1.0; 0.0; 0.0]; length1 = 1.0; width, depth = 0.1, 0.1
    p2 = [0.0; 0.0; length1 / 2]
    origin = Origin{Float64}()
    link1 = Box(width, depth, length1, length1)
    j = EqualityConstraint(Revolute(origin, link1, joint_axis; p2 = p2))
    mech = Mechanism(origin, [link1], [j])
    setPosition!(origin, link1, p2 = p2, Δq = Quaternion(RotX(pi - 0.4)))                          # :30
    xd = [[0; 0.0; 0.5]]; qd = [Quaternion(RotX(1.0 * pi))]                                       # :32-33
    Q = [diagm(ones(12)) * 0.0]; Q[1][7, 7] = 1000.0; Q[1][10, 10] = 100.0; R = [ones(1, 1)]        # :35-38
    export_lqr("pendulum_cfg1", mech, getid.([link1]), getid.([j]), Q, R, Inf, 10; xd = xd, qd = qd)
end

# ================================================================ configs[1]  examples/lqr_cartpole.jl  (the script's own initial state: y = 0.5, φ = 0.2)
function cart_chain(N; g = -9.81)
    ex = [1.0; 0.0; 0.0]; ey = [0.0; 1.0; 0.0]; length1 = 1.0; width, depth = 0.1, 0.1
    p2 = [0.0; 0.0; length1 / 2]
    origin = Origin{Float64}()
    cart = Box(0.1, 0.5, 0.1, length1 / 2)
    bodies = [cart; [Box(width, depth, length1, length1) for i = 1:N]]
    joint1 = EqualityConstraint(Prismatic(origin, cart, ey))
    joint2 = EqualityConstraint(Revolute(cart, bodies[2], ex; p2 = -p2))
    constraints = [joint1; joint2]
    if N > 1
        constraints = [constraints; [EqualityConstraint(Revolute(bodies[i], bodies[i+1], ex; p1 = p2, p2 = -p2)) for i = 2:N]]   # lqr_cartpole_n_pendulum.jl:34
    end
    mech = Mechanism(origin, bodies, constraints, g = g)
    place!(y, φ) = begin
        setPosition!(origin, cart, Δx = [0; y; 0])
        setPosition!(cart, bodies[2], p2 = -p2, Δq = Quaternion(RotX(φ[1])))
        for i = 2:N
            setPosition!(bodies[i], bodies[i+1], p1 = p2, p2 = -p2, Δq = Quaternion(RotX(φ[i])))
        end
    end
    return mech, origin, bodies, constraints, place!
end
let
    mech, origin, bodies, constraints, place! = cart_chain(1)
    place!(0.5, [0.2])                                                                            # lqr_cartpole.jl:33-34
    xd = [[0; 0; 0.0], [0; 0; 0.5]]; qd = [one(Quaternion{Float64}) for _ = 1:2]
    Q = [diagm(ones(12)) * 1.0 for i = 1:2]; R = [ones(1, 1)]
    export_lqr("cartpole_cfg2", mech, getid.(bodies), [getid(constraints[1])], Q, R, 10.0, 10; xd = xd, qd = qd)
end

# ================================================================ configs[2]  examples/lqr_cartpole_n_pendulum.jl
# (a) as scripted with N = 3 (the value the script ships), with a fixed draw instead of rand(): y = 0.25, φ = [0.03, 0.02, 0.01] (< 3^-3 = 0.037)
let
    N = 3
    mech, origin, bodies, constraints, place! = cart_chain(N)
    place!(0.25, [0.030, 0.020, 0.010])                      # a fixed stand-in for `rand(N)/(3^N)`, `rand()-0.5` (:21-22); inside the script's range
    xd = [[[0; 0; 0.0]]; [[0; 0; i - 1 + 0.5] for i = 1:N]]; qd = [one(Quaternion{Float64}) for _ = 1:N+1]
    Q = [diagm(ones(12)) * 1.0 for i = 1:N+1]; R = [ones(1, 1)]
    export_lqr("chain3_upright_cfg3_as_scripted", mech, getid.(bodies), [getid(constraints[1])], Q, R, 10.0, 10; xd = xd, qd = qd)
end
# (b) the bench workload of this repository: N = 16, regulated about the HANGING equilibrium (first link turned by π, the others straight), the
#     start a fixed small swing: y = 0.1, φ_1 = π + 0.15, φ_i = (-1)^i 0.1
let
    N = 16
    mech, origin, bodies, constraints, place! = cart_chain(N)
    place!(0.0, [1.0 * pi; zeros(N - 1)])
    xd = [Vector(b.state.xc) for b in mech.bodies]; qd = [b.state.qc for b in mech.bodies]          # the hanging configuration itself is the setpoint
    place!(0.1, [pi + 0.15; [(-1.0)^i * 0.1 for i = 2:N]])
    Q = [diagm(ones(12)) * 1.0 for i = 1:N+1]; R = [ones(1, 1)]
    export_lqr("chain16_hanging_cfg3_bench", mech, getid.(bodies), [getid(constraints[1])], Q, R, 10.0, 10; xd = xd, qd = qd)
end

# ================================================================ configs[3]  examples/lqr_sawyer.jl  (paths relative to the reference's root)
let
    path = joinpath(dirname(pathof(ConstrainedControl)), "..", "examples", "examples_files", "sawyer_arm.urdf")
    mech = Mechanism(path, floating = false, g = 0.0)                                              # :9
    names = ["right_j0", "right_j1", "right_j2", "right_j3", "right_j4", "right_j5", "right_j6"]
    for n in names; setPosition!(mech, mech.eqconstraints[n], [0.0]); end                          # :11-14 (all seven: the zero pose)
    xd = [Vector(b.state.xc) for b in mech.bodies]; qd = [b.state.qc for b in mech.bodies]          # :16-17
    for (i, n) in enumerate(names); setPosition!(mech, mech.eqconstraints[n], [0.002 * (-1.0)^i]); end   # a start inside the controller's region of attraction
    Q = [diagm(ones(12)) * 1000.0 for i = 1:7]; R = [ones(1, 1) for i = 1:7]                       # :25-26
    export_lqr("sawyer_cfg4", mech, getid.(mech.bodies), getid.(mech.eqconstraints), Q, R, 20.0, 20; xd = xd, qd = qd)
end

# ================================================================ configs[4]  examples/trackingLQR_triple_cartpole.jl
let
    src = joinpath(dirname(pathof(ConstrainedControl)), "..", "examples", "trackingLQR_triple_cartpole.jl")
    U = eval(Meta.parse(split(readline(src), "=", limit = 2)[2]))                                   # line 1 of the script: `U = [ ... ]`
    dir = joinpath(OUT, "ref_triple_cartpole_tracking_cfg5"); mkpath(dir)
    ex = [1.0; 0.0; 0.0]; ey = [0.0; 1.0; 0.0]; length1 = 1.0; width, depth = 0.1, 0.1
    p2 = [0.0; 0.0; length1 / 2]
    origin = Origin{Float64}()
    cart = Box(0.1, 0.5, 0.1, length1 / 2); pole1 = Box(width, depth, length1, length1); pole2 = deepcopy(pole1); pole3 = deepcopy(pole1)
    joint1 = EqualityConstraint(Prismatic(origin, cart, ey))
    joint2 = EqualityConstraint(Revolute(cart, pole1, ex; p2 = p2))
    joint3 = EqualityConstraint(Revolute(pole1, pole2, ex; p1 = -p2, p2 = p2))
    joint4 = EqualityConstraint(Revolute(pole2, pole3, ex; p1 = -p2, p2 = p2))
    links = [cart; pole1; pole2; pole3]; constraints = [joint1; joint2; joint3; joint4]
    mech = Mechanism(origin, links, constraints, g = -9.81, Δt = 0.01)                              # :40
    zero_pose!() = begin
        setPosition!(origin, cart, Δx = [0; 0.0; 0]); setPosition!(cart, pole1, p2 = p2, Δq = Quaternion(RotX(0.0)))
        setPosition!(pole1, pole2, p1 = -p2, p2 = p2, Δq = Quaternion(RotX(0.0))); setPosition!(pole2, pole3, p1 = -p2, p2 = p2, Δq = Quaternion(RotX(0.0)))
        for b in links; setVelocity!(b); end                                                      # :144-147
    end
    zero_pose!()
    save(dir, "z0", state_matrix(mech)); save(dir, "U", Float64.(U)); save(dir, "dt", mech.Δt); save(dir, "g", mech.g)
    steps = Base.OneTo(1000)
    storage0 = Storage{Float64}(steps, 4)
    simulate!(mech, storage0, (m, k) -> setForce!(m, joint1, [U[k]]), record = true)               # :46-53: the open-loop swing-up
    ks = sample_ks(1000)
    x, q, v, w = storage_samples(storage0, 4, ks)
    save(dir, "k_list", ks); save(dir, "storage0_x", x); save(dir, "storage0_q", q); save(dir, "storage0_v", v); save(dir, "storage0_w", w)
    xa, qa, va, wa = storage_samples(storage0, 4, collect(1:1000))                                 # every knot: the setpoints TrackingLQR is built from
    save(dir, "storage0_all_x", xa); save(dir, "storage0_all_q", qa); save(dir, "storage0_all_v", va); save(dir, "storage0_all_w", wa)
    Q = [diagm(ones(12)) * 0.0 for i = 1:4]
    Q[1][2, 2] = 10; Q[1][5, 5] = 1; Q[2][7, 7] = 40; Q[2][10, 10] = 1; Q[3][7, 7] = 40; Q[3][10, 10] = 1; Q[4][7, 7] = 40; Q[4][10, 10] = 1   # :62-70
    R = [ones(1, 1) * 0.1]
    zero_pose!()
    lqr = TrackingLQR(mech, storage0, [[[U[k]]] for k = 1:1000], [joint1.id], Q, R)      # :117 without the script-local controlfunction:
                                                                                          # the package's own control_trackinglqr! (lqr_tracking.jl:46-71), no friction, no noise
    save(dir, "K_all", gains(lqr.K)); save(dir, "K_distinct_from", [distinct_from(lqr.K)])
    save(dir, "Q", cat(Q..., dims = (1, 2)) * mech.Δt); save(dir, "R", cat(R..., dims = (1, 2)) * mech.Δt)
    zero_pose!()
    save_newton_facts(dir, mech, lqr)
    storage = Storage{Float64}(steps, 4)
    simulate!(mech, storage, lqr, record = true)
    x, q, v, w = storage_samples(storage, 4, ks)
    save(dir, "storage_x", x); save(dir, "storage_q", q); save(dir, "storage_v", v); save(dir, "storage_w", w)
    println("wrote ", dir)
end
