# CCLQR.jl -- `ccall` shim between ConstrainedControl.jl and libcclqr.so (include/cclqr.h).
#
# STATUS: source only.  No Julia toolchain exists in the build/test pipeline of this repository, so this file has never been
# executed; the ABI it binds IS exercised (by constrainedcontrol.jl_amd/_capi.py through ctypes).  Struct layouts below mirror
# include/cclqr.h field by field.  `mech_tables` / `ctrl_joints`, which read ConstrainedDynamics internals, are the only
# parts that depend on that package's field names (0.9.x); everything else is plain arrays.
#
# Drop-in points in the reference (janbruedigam/ConstrainedControl.jl v0.3.0):
#   src/control/lqr.jl:63   A, Bu, Bλ, G = linearsystem(...)          ->  CCLQR.linearsystem(h, zd, ctrl, Fd)
#   src/control/lqr.jl:39   Ku = dlqr(A, Bu, Bλ, G, Q, R, Ntemp)      ->  CCLQR.dlqr(A, Bu, Bλ, G, Q, R, Ntemp)
#   src/control/lqr_tracking.jl:40  Ku = dlqr(mechanism, xd, ...)      ->  CCLQR.dlqr_tracking(h, zd, Fd, ctrl, Q, R, N)
#   simulate!(mech, tend, lqr; record) on a batch                     ->  CCLQR.simulate_batch!(h, c, z0, steps; record, noise)
module CCLQR

using LinearAlgebra

const lib = get(ENV, "CCLQR_LIB", joinpath(@__DIR__, "..", "constrainedcontrol.jl_amd", "libcclqr.so"))

struct MechDesc            # cclqr_mech_desc
    nb::Int32; ne::Int32
    dt::Float64; g::Float64
    mass::Ptr{Float64}; inertia::Ptr{Float64}
    parent::Ptr{Int32}; child::Ptr{Int32}; type::Ptr{Int32}
    p1::Ptr{Float64}; p2::Ptr{Float64}; axis::Ptr{Float64}; qoff::Ptr{Float64}
end

struct CtrlDesc            # cclqr_ctrl_desc
    mu::Int32; ctrl_joint::Ptr{Int32}
    nK::Int32; N::Int32; K::Ptr{Float64}
    nsp::Int32; zd::Ptr{Float64}; Fd::Ptr{Float64}
    fric::Ptr{Float64}; noise_scale::Float64
    npid::Int32; pid_joint::Ptr{Int32}
    pid_P::Ptr{Float64}; pid_I::Ptr{Float64}; pid_D::Ptr{Float64}; pid_goal::Ptr{Float64}     # PID{T,N}, src/control/pid.jl:3-11
    noise_philox::Int32; noise_seed::UInt64                                                    # reproducible stand-in for randn()
    n_ctrl::Int32                                                                              # > 1: one (K, zd, Fd) table per instance
end

struct RolloutOpts         # cclqr_rollout_opts
    first_instance::Int64
    pid_state_dev::Ptr{Float64}
    pid_state_len::Int64
    noise_ws_dev::Ptr{Float64}
    noise_ws_len::Int64
    newton_mode::Int32
    flags::Int32               # ROLLOUT_NO_ALLOC: the call may neither allocate nor synchronise (a hipGraph capture is open on the device)
    newton_eps_alone::Float64
end
const ROLLOUT_NO_ALLOC = Int32(1)
const ROLLOUT_CARRY_STATUS = Int32(4)      # `status` carries an instance's status across launches: a lost instance stays frozen (step-per-launch loops)
const ROLLOUT_PACK_WAVEFRONTS = Int32(2)   # every wavefront of a chain launch full whatever the batch size (many launches sharing the device at once)

struct RiccatiOpts         # cclqr_riccati_opts
    path::Int32
    bf16_terms::Int32
    keep_last::Int32
    reserved::Int32
end

const ABI_VERSION = 201    # include/cclqr.h CCLQR_ABI_VERSION: the structs above mirror THAT header field by field
"sizeof / fieldoffset of the four mirrors above in the order cclqr_abi_layout reports the library's own (include/cclqr.h)"
mirrored_layout() = Int32[x for S in (MechDesc, CtrlDesc, RiccatiOpts, RolloutOpts) for x in (sizeof(S), (fieldoffset(S, i) for i in 1:fieldcount(S))...)]
"call once after loading: a library built from another header would read these structs past their end (ADVICE r2).  The version number says which
header the library was built from; cclqr_abi_layout says what the C compiler made of it, so the mirrors are verified field by field (VERDICT r4)."
function check_abi()
    v = ccall((:cclqr_version, lib), Cint, ())
    v == ABI_VERSION || error("libcclqr.so has ABI version $v, CCLQR.jl was written for $ABI_VERSION")
    want = mirrored_layout()
    got = zeros(Int32, length(want))
    n = ccall((:cclqr_abi_layout, lib), Cint, (Ptr{Int32}, Int32), got, Int32(length(got)))
    (n == length(want) && got == want) || error("struct layout mismatch: libcclqr.so reports $got, CCLQR.jl mirrors $want")
    nothing
end

const REVOLUTE = Int32(0)
const PRISMATIC = Int32(1)
const FIXED_ORIENTATION = Int32(2)     # FixedOrientation(a, b; qoffset): one Rotational3 constraint (examples/lqr_deltabot.jl:25)

# Closed-loop mechanisms (examples/lqr_deltabot.jl): `linearize_projected` returns A' = A - Bλ (G Bλ)^-1 G A and D = Bu - Bλ (G Bλ)^-1 G Bu
# (lqr.jl:151) from cclqr_linearize_projected; pass them to `riccati` with empty Bλ (mx x 0) and G (0 x mx).
function linearize_projected(mech::Ptr{Cvoid}, zd::Vector{Float64}, nb::Int, ctrl_joint::Vector{Int32}, Fd::Vector{Float64}; h = 0.0)
    mx, mu = 12nb, length(ctrl_joint)
    Ap, D = zeros(mx * mx), zeros(mx * mu)
    check(ccall((:cclqr_linearize_projected, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Ptr{Int32}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
                mech, Int32(1), zd, Int32(mu), ctrl_joint, Fd, h, Ap, D))
    return permutedims(reshape(Ap, mx, mx)), permutedims(reshape(D, mu, mx))      # row-major on the C side
end

# One LQR per setpoint, built and kept on the device (cclqr_ctrl_create_lqr_batch): zd is nb*13*n doubles, Q / R the Δt-scaled weights
# (lqr.jl:18-19) row-major; returns the controller handle for rollout!(...) and the break indices.
function lqr_batch(mech::Ptr{Cvoid}, n::Int, zd::Vector{Float64}, ctrl_joint::Vector{Int32}, Q::Vector{Float64}, R::Vector{Float64}, N::Int;
                   tol = 1e-5, infinite_horizon = false)      # infinite_horizon: LQR{T,Inf} -- N = ceil(10/Δt) (lqr.jl:26), only Ku[1] is kept
    h = Ref{Ptr{Cvoid}}(C_NULL); kb = zeros(Int32, n)
    check(ccall((:cclqr_ctrl_create_lqr_batch, lib), Cint,
                (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Float64, Ptr{Int32}, Ref{Ptr{Cvoid}}),
                mech, Int32(n), zd, Int32(length(ctrl_joint)), ctrl_joint, C_NULL, Q, R, Int32(N), Int32(infinite_horizon ? 1 : 0), tol, kb, h))
    return h[], kb
end

# Time-varying recursion (lqr_tracking.jl:73-122) on per-knot models the caller brings (row-major [N-1][..][..] flattened), e.g. the
# projected pairs of a closed-loop mechanism with ml = 0
function dlqr_tv(mx::Int, mu::Int, ml::Int, A::Vector{Float64}, Bu::Vector{Float64}, Bl::Vector{Float64}, G::Vector{Float64},
                 Q::Vector{Float64}, R::Vector{Float64}, N::Int; tol = 1e-5)
    K = zeros((N - 1) * mu * mx); kb = zeros(Int32, 1)
    check(ccall((:cclqr_riccati_tv, lib), Cint,
                (Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Ptr{Float64}, Ptr{Int32}),
                Int32(mx), Int32(mu), Int32(ml), A, Bu, ml > 0 ? pointer(Bl) : C_NULL, ml > 0 ? pointer(G) : C_NULL, Q, R, Int32(N), tol, K, kb))
    return K, Int(kb[1])
end

lasterror() = unsafe_string(ccall((:cclqr_last_error, lib), Cstring, ()))
function check(rc::Integer)
    rc == 0 && return nothing
    msg = lasterror()
    rc == -1 && throw(AssertionError(msg))          # CCLQR_EINVAL   <-> the reference's @assert (lqr.jl:59-60)
    rc == -2 && throw(SingularException(0))         # CCLQR_ESINGULAR <-> LAPACK exception from lqr.jl:151,160
    rc == -3 && (@info msg; return nothing)         # CCLQR_ENOCONV  <-> `@info "Riccati recursion did not converge."` (lqr.jl:41)
    error("cclqr ($rc): $msg")
end

# ---------------------------------------------------------------------------------------------------------------------
# Mechanism -> flat tables.  Assumes ConstrainedDynamics 0.9.x field names: mechanism.bodies / eqconstraints / origin / Δt / g,
# body.m, body.J, eqc.parentid, eqc.childids, eqc.constraints with .vertices, .V3 (axis row), .qoffset.
# Body ids 1..Nb, joint ids Nb+1.. (examples/trackingLQR_triple_cartpole.jl:109-111).
#
# An EqualityConstraint may bundle SEVERAL joints off one parent: examples/lqr_deltabot.jl:25 builds
#   EqualityConstraint(Revolute(origin, lowerlegl, ...), Revolute(origin, lowerlegr, ...), FixedOrientation(origin, platform))
# whose `constraints` is the flat tuple (Translational3, Rotational2, Translational3, Rotational2, Rotational3) with
# `childids` = (l, l, r, r, platform) -- one entry per component.  The C side knows 1-DoF joints and the FixedOrientation, one child
# each, so every EqualityConstraint is EXPANDED into one table joint per distinct child (components grouped by child id, in order):
#   Translational{3} + Rotational{2} -> REVOLUTE,  Translational{2} + Rotational{3} -> PRISMATIC,  a lone Rotational{3} -> FIXED_ORIENTATION;
# anything else is refused (a silent flattening would simulate a different mechanism).  `joint_of_eqc[id]` lists the 0-based table
# joints an EqualityConstraint id became: an LQR's `eqcids` (lqr.jl:49-57; one input per 1-DoF joint) map through it.
nrows(c) = length(c)                                  # Joint{T,N}: N constraint rows
istrans(c) = occursin("Translational", string(typeof(c)))
function mech_tables(mechanism)
    bodies = collect(mechanism.bodies); eqcs = collect(mechanism.eqconstraints)
    nb = length(bodies)
    bodyindex = Dict(b.id => Int32(i - 1) for (i, b) in enumerate(bodies))
    mass = Float64[b.m for b in bodies]
    inertia = reduce(vcat, [vec(permutedims(Matrix(b.J))) for b in bodies])          # row-major 3x3 per body
    parent = Int32[]; child = Int32[]; typ = Int32[]
    p1 = Float64[]; p2 = Float64[]; axis = Float64[]; qoff = Float64[]
    joint_of_eqc = Dict{Int,Vector{Int32}}()
    for e in eqcs
        cs = collect(e.constraints); cids = collect(e.childids)
        length(cs) == length(cids) || error("EqualityConstraint $(e.id): constraints and childids differ in length")
        joint_of_eqc[e.id] = Int32[]
        i = 1
        while i <= length(cs)
            j = i
            while j < length(cs) && cids[j+1] == cids[i]; j += 1; end
            comp = cs[i:j]
            tr = [c for c in comp if istrans(c)]; ro = [c for c in comp if !istrans(c)]
            kind = if length(tr) == 1 && length(ro) == 1 && nrows(tr[1]) == 3 && nrows(ro[1]) == 2
                REVOLUTE
            elseif length(tr) == 1 && length(ro) == 1 && nrows(tr[1]) == 2 && nrows(ro[1]) == 3
                PRISMATIC
            elseif isempty(tr) && length(ro) == 1 && nrows(ro[1]) == 3
                FIXED_ORIENTATION
            else
                error("EqualityConstraint $(e.id), child $(cids[i]): only Revolute, Prismatic and FixedOrientation components are supported")
            end
            push!(joint_of_eqc[e.id], Int32(length(typ)))
            push!(parent, get(bodyindex, e.parentid, Int32(-1)))                       # origin -> -1
            push!(child, bodyindex[cids[i]]); push!(typ, kind)
            if kind == FIXED_ORIENTATION
                append!(p1, zeros(3)); append!(p2, zeros(3)); append!(axis, [1.0, 0.0, 0.0])
            else
                append!(p1, Vector{Float64}(tr[1].vertices[1])); append!(p2, Vector{Float64}(tr[1].vertices[2]))
                append!(axis, Vector{Float64}(vec(kind == REVOLUTE ? ro[1].V3 : tr[1].V3)))
            end
            q = ro[1].qoffset
            append!(qoff, Float64[q.s, q.v1, q.v2, q.v3])
            i = j + 1
        end
    end
    ne = length(typ)
    return (; nb, ne, dt = Float64(mechanism.Δt), g = Float64(mechanism.g), mass, inertia, parent, child, typ, p1, p2, axis, qoff, joint_of_eqc)
end
"0-based table joints of the 1-DoF joints named by an LQR's eqcids (lqr.jl:49-57): every id must have become exactly one 1-DoF joint"
function ctrl_joints(t, eqcids)
    out = Int32[]
    for id in eqcids
        js = [j for j in t.joint_of_eqc[id] if t.typ[j+1] != FIXED_ORIENTATION]
        length(js) == 1 || throw(AssertionError("eqcid $id bundles $(length(js)) 1-DoF joints: name the joint (lqr.jl:1-2 'Only for 1dof joints')"))
        push!(out, js[1])
    end
    out
end

mutable struct MechHandle
    ptr::Ptr{Cvoid}
    nb::Int
    function MechHandle(t)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve t begin
            d = MechDesc(t.nb, t.ne, t.dt, t.g, pointer(t.mass), pointer(t.inertia), pointer(t.parent), pointer(t.child), pointer(t.typ),
                         pointer(t.p1), pointer(t.p2), pointer(t.axis), pointer(t.qoff))
            check_abi()
            check(ccall((:cclqr_mech_create, lib), Cint, (Ref{MechDesc}, Ref{Ptr{Cvoid}}), d, h))
        end
        obj = new(h[], t.nb)
        finalizer(o -> ccall((:cclqr_mech_destroy, lib), Cint, (Ptr{Cvoid},), o.ptr), obj)
        obj
    end
end
MechHandle(mechanism, ::Val{:mechanism}) = MechHandle(mech_tables(mechanism))

# body states as the 13 x Nb matrix the C side reads as [nb][13] = x(3) q(4) v(3) ω(3)
pack_state(xd, qd, vd, ωd) = reduce(hcat, [Float64[x...; q.s; q.v1; q.v2; q.v3; v...; ω...] for (x, q, v, ω) in zip(xd, qd, vd, ωd)])

# C is row-major, Julia column-major: a C matrix [r][c] is read here as a (c, r) array and transposed lazily.
cmat(buf, r, c) = permutedims(reshape(buf, c, r))

"linearsystem(mechanism, xd, vd, qd, ωd, Fτd, bodyids, eqcids) replacement (lqr.jl:63).  ctrl = 0-based joint indices."
function linearsystem(h::MechHandle, zd::Matrix{Float64}, ctrl::Vector{Int32}, Fd::Vector{Float64})
    nb = h.nb; mx, ml, mu = 12nb, 5nb, length(ctrl)
    A = zeros(mx * mx); Bu = zeros(mx * max(mu, 1)); Bl = zeros(mx * ml); G = zeros(ml * mx)
    check(ccall((:cclqr_linearize, lib), Cint,
                (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                h.ptr, 1, zd, mu, ctrl, Fd, A, Bu, Bl, G))
    cmat(A, mx, mx), cmat(Bu[1:mx*mu], mx, mu), cmat(Bl, mx, ml), cmat(G, ml, mx)
end

"dlqr(A,Bu,Bλ,G,Q,R,N) replacement (lqr.jl:141-184).  Q, R are the Δt-scaled block diagonals of lqr.jl:18-19.  Returns Ku[k][i] and the break index."
function dlqr(A, Bu, Bλ, G, Q, R, N::Integer; tol = 1e-5)
    mx, mu, ml = size(A, 1), size(Bu, 2), size(Bλ, 2)
    rowmajor(M) = vec(permutedims(Matrix{Float64}(M)))
    K = zeros(mx * mu * max(N - 1, 0)); kb = Ref{Int32}(0)
    check(ccall((:cclqr_riccati, lib), Cint,
                (Int32, Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Float64,
                 Ptr{Float64}, Ref{Int32}),
                1, mx, mu, ml, rowmajor(A), rowmajor(Bu), rowmajor(Bλ), rowmajor(G), rowmajor(Q), rowmajor(R), N, tol, K, kb))
    K3 = reshape(K, mx, mu, N - 1)                                   # C layout [N-1][mu][mx]
    Ku = [[reshape(K3[:, i, k], 1, mx) for i = 1:mu] for k = 1:N-1]   # the layout of lqr.jl:145
    Ku, Int(kb[])
end

"dlqr(mechanism, xd, vd, qd, ωd, Fτd, eqcids, Q, R, N) replacement (lqr_tracking.jl:73-122): zd is 13 x Nb x N, Fd is mu x N."
function dlqr_tracking(h::MechHandle, zd::Array{Float64,3}, Fd::Matrix{Float64}, ctrl::Vector{Int32}, Q, R, N::Integer; tol = 1e-5)
    mx, mu = 12h.nb, length(ctrl)
    rowmajor(M) = vec(permutedims(Matrix{Float64}(M)))
    K = zeros(mx * mu * (N - 1)); kb = Ref{Int32}(0)
    check(ccall((:cclqr_riccati_tracking, lib), Cint,
                (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Ptr{Float64}, Ref{Int32}),
                h.ptr, mu, ctrl, zd, Fd, rowmajor(Q), rowmajor(R), N, tol, K, kb))
    K3 = reshape(K, mx, mu, N - 1)
    [[reshape(K3[:, i, k], 1, mx) for i = 1:mu] for k = 1:N-1], Int(kb[])
end

mutable struct CtrlHandle
    ptr::Ptr{Cvoid}
end
"Device controller tables from the fields of an LQR / TrackingLQR (lqr.jl:3-15, lqr_tracking.jl:3-15).
 K: mx x mu x nK, zd: 13 x Nb x nsp, Fd: mu x nsp, N = horizon steps (0 for LQR{T,Inf}), ctrl = 0-based joint indices."
function CtrlHandle(h::MechHandle, ctrl::Vector{Int32}, K::Array{Float64,3}, N::Integer, zd::Array{Float64,3}, Fd::Matrix{Float64};
                    fric::Union{Nothing,Vector{Float64}} = nothing, noise_scale = 0.0, n_ctrl::Integer = 0)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    nc = max(n_ctrl, 1)          # n_ctrl > 1: K is mx x mu x (nK * n_ctrl), zd 13 x Nb x (nsp * n_ctrl), Fd mu x (nsp * n_ctrl): one table per instance
    GC.@preserve ctrl K zd Fd fric begin
        d = CtrlDesc(length(ctrl), pointer(ctrl), size(K, 3) ÷ nc, N, pointer(K), size(zd, 3) ÷ nc, pointer(zd), pointer(Fd),
                     fric === nothing ? Ptr{Float64}(C_NULL) : pointer(fric), noise_scale,
                     0, Ptr{Int32}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), 0, UInt64(0),
                     n_ctrl)
        check(ccall((:cclqr_ctrl_create, lib), Cint, (Ptr{Cvoid}, Ref{CtrlDesc}, Ref{Ptr{Cvoid}}), h.ptr, d, c))
    end
    obj = CtrlHandle(c[])
    finalizer(o -> ccall((:cclqr_ctrl_destroy, lib), Cint, (Ptr{Cvoid},), o.ptr), obj)
    obj
end

"Batched simulate!: z0 is 13 x Nb x n_inst; returns (traj 13 x Nb x steps x n_inst or nothing, zT, status)."
function simulate_batch!(h::MechHandle, c::CtrlHandle, z0::Array{Float64,3}, steps::Integer; record = true, noise = nothing, k0 = 1,
                         first_instance = 0, newton_mode = 0, newton_eps_alone = 0.0)
    n = size(z0, 3)
    traj = record ? zeros(13, h.nb, steps, n) : nothing
    zT = similar(z0); status = zeros(Int32, n)
    o = RolloutOpts(first_instance, C_NULL, 0, C_NULL, 0, newton_mode, 0, newton_eps_alone)      # first_instance: global index of instance 1 of this shard
    check(ccall((:cclqr_rollout_host_ex, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ref{RolloutOpts}),
                h.ptr, c.ptr, n, steps, k0, z0, noise === nothing ? C_NULL : noise, record ? traj : C_NULL, zT, status, o))
    traj, zT, status
end

"Device-pointer rollout with explicit options (cclqr_rollout_ex): all pointers are device addresses (e.g. from AMDGPU.jl), the launch is
 asynchronous on `stream`.  first_instance = global index of instance 1 of this shard (noise stream and per-instance controller table),
 pid_state = device buffer n_inst x Nb x 2 carrying the PID integrators between launches (C_NULL: none), noise_ws = device workspace of
 n_inst x steps doubles for the Philox samples (C_NULL: the controller handle's own; see reserve_noise!), newton_mode 0 = exact rule,
 flags = ROLLOUT_NO_ALLOC | ROLLOUT_PACK_WAVEFRONTS | ROLLOUT_CARRY_STATUS (include/cclqr.h)."
function rollout_dev!(h::MechHandle, c::CtrlHandle, n::Integer, steps::Integer, k0::Integer, z0::Ptr{Float64}, lam::Ptr{Float64},
                      traj::Ptr{Float64}, zT::Ptr{Float64}, status::Ptr{Int32}; noise::Ptr{Float64} = Ptr{Float64}(C_NULL), noise_stride = 0,
                      first_instance = 0, pid_state::Ptr{Float64} = Ptr{Float64}(C_NULL), noise_ws::Ptr{Float64} = Ptr{Float64}(C_NULL),
                      newton_mode = 0, newton_eps_alone = 0.0, flags = Int32(0), stream::Ptr{Cvoid} = C_NULL)
    o = RolloutOpts(first_instance, pid_state, pid_state == C_NULL ? 0 : n * h.nb * 2, noise_ws, noise_ws == C_NULL ? 0 : n * steps, newton_mode, Int32(flags), newton_eps_alone)
    check(ccall((:cclqr_rollout_ex, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Int32},
                 Ref{RolloutOpts}, Ptr{Cvoid}),
                h.ptr, c.ptr, n, steps, k0, z0, lam, noise, noise_stride, traj, zT, status, o, stream))
end

"Size the controller handle's Philox workspace before step-per-launch rollouts are captured into a hipGraph (cclqr_ctrl_reserve_noise)."
reserve_noise!(c::CtrlHandle, n::Integer, steps::Integer) = check(ccall((:cclqr_ctrl_reserve_noise, lib), Cint, (Ptr{Cvoid}, Int64, Int32), c.ptr, n, steps))

"The `controlfunction` hook (lqr.jl:14, :56) with the closure on the host: U (mu x nsp x n_ctrl, as given to CtrlHandle) are the joint inputs the
closure computed for the next single-step launch (cclqr_ctrl_set_feedforward; host array, copied synchronously).  Step the batch with
`rollout_dev!(...; flags = ROLLOUT_CARRY_STATUS)` and a zeroed status array: an instance lost in an earlier launch then stays frozen, as in one launch over the horizon."
set_feedforward!(c::CtrlHandle, U::Array{Float64}) =
    check(ccall((:cclqr_ctrl_set_feedforward, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int32, Ptr{Cvoid}), c.ptr, U, length(U), 0, C_NULL))

"Storage{T}(steps, Nb) view of instance n of a batched trajectory: storage.x[i][k] etc. (lqr_tracking.jl:32-35)."
function storage_fields(traj::Array{Float64,4}, n::Integer)
    nb, steps = size(traj, 2), size(traj, 3)
    x = [[traj[1:3, i, k, n] for k = 1:steps] for i = 1:nb]
    q = [[traj[4:7, i, k, n] for k = 1:steps] for i = 1:nb]
    v = [[traj[8:10, i, k, n] for k = 1:steps] for i = 1:nb]
    ω = [[traj[11:13, i, k, n] for k = 1:steps] for i = 1:nb]
    (; x, q, v, ω)
end

end # module
