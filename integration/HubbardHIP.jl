# HubbardHIP.jl -- reference-side binding of libhubbardtn_hip.so (INTEGRATION.md section 1).
#
# STATUS: UNTESTED.  The build image has no Julia and none of the reference's dependencies, so nothing in this file has
# ever been parsed or run.  It is written against the documented data model of the versions the reference pins
# (Manifest.toml: MPSKit 0.13.1, TensorKit 0.14.6, BlockTensorKit 0.1.6, TensorKitSectors 0.1.4) and is meant as the
# concrete starting point for the maintainer who wires the library in, not as a claim of a working package.  The table
# formats it produces are the ones `tests/test_cengine_cpu.py::test_mps_import_export_tables_shuffled_order_and_padded_
# leading_dimensions` and `::test_whole_sweep_through_the_c_abi_only` feed to the same entry points from Python.
#
# What it replaces: the method `find_groundstate(psi, H, alg)` that `compute_groundstate` reaches at
# src/HubbardFunctions.jl:1010.  `hamiltonian(simul)` (src:386-472, 811-910) and `initialize_mps` (src:917-959) stay as
# they are; their results are exported as tables, the sweeps run in the library, the state is imported back.
module HubbardHIP

using MPSKit, TensorKit, BlockTensorKit
using TensorKit: FusionTree, fusiontrees, sectors, dim, space, domain, codomain

const lib = "libhubbardtn_hip.so"

# ---- mirrors of the C structs of include/hubbardtn_hip.h (sizes asserted by tests/test_abi.py on the Python side) ----
struct HtnSymmetry
    kind::Int32
    n_site::Int32
    site_N::NTuple{4,Int32}
    site_j::NTuple{4,Int32}
end
struct HtnSiteOp            # 136 bytes: rank 2k, charge dN, reduced matrix elements red[out * 4 + in]
    k::Int32
    dN::Int32
    red::NTuple{16,Float64}
end
struct HtnMpoEntry
    wl::Int32
    wr::Int32
    op::Int32
    pad::Int32
    re::Float64
    im::Float64
end
struct HtnSubblock
    lN::Int32
    lj::Int32
    s::Int32
    rN::Int32
    rj::Int32
    ld::Int32
    off::Int64
end
struct HtnSector
    N::Int32
    j::Int32
    count::Int32
end
struct HtnSweepOpts
    chi_full::Int32
    weighting::Int32
    cutoff::Float64
    krylovdim::Int32
    maxrestart::Int32
    lanczos_tol::Float64
    jacobi_tol::Float64
    jacobi_max_sweeps::Int32
    svd_split_elems::Int32
    rank_cut::Float64
    profile::Int32
    pad::Int32
end

check(rc) = rc == 0 || error(unsafe_string(ccall((:htn_last_error, lib), Cstring, ())))

# ---- sector labels --------------------------------------------------------------------------------------------------
# reference: I = fZ2 ⊠ SU2Irrep ⊠ U1Irrep (src:250), charge shifted per site: k = n Q - P (src:251).  Library: (N, 2S)
# with N the electron number counted from the left end of the chain.  `sites_left` = number of sites left of the bond.
function label(c, P::Int, Q::Int, sites_left::Int)
    twoS = Int(2 * c.sectors[2].j)
    k = Int(c.sectors[3].charge)
    num = k + P * sites_left
    num % Q == 0 || error("charge $k on a bond with $sites_left sites to its left is not a multiple of Q = $Q")
    return Int32(num ÷ Q), Int32(twoS)
end
# site multiplets in the library's order: empty, singly occupied, doubly occupied
site_index(c, P, Q) = (n = (Int(c.sectors[3].charge) + P) ÷ Q; n + 1)       # 1-based: 1, 2, 3

# ---- sweep options ----------------------------------------------------------------------------------------------------
# truncdim(D) -> chi_full, truncbelow(eta) -> cutoff (src:1010 uses truncbelow(10^-svalue), src:1363-1365 truncdim).
# (TensorKit 0.14: TruncationDimension has field `dim`, TruncationCutoff fields `ϵ`, `add_back`; MultipleTruncation `truncations`.)
function sweep_opts(trscheme; krylovdim = 30, tol = 1e-10, maxrestart = 1)
    chi, cut = Int32(0), 0.0
    schemes = trscheme isa TensorKit.MultipleTruncation ? trscheme.truncations : (trscheme,)
    for t in schemes
        if t isa TensorKit.TruncationDimension
            chi = Int32(t.dim)
        elseif t isa TensorKit.TruncationCutoff
            cut = Float64(t.ϵ)
        elseif !(t isa TensorKit.NoTruncation)
            error("HubbardHIP: unsupported truncation scheme $(typeof(t))")
        end
    end
    return HtnSweepOpts(chi, 0, cut, Int32(krylovdim), Int32(maxrestart), tol, 1e-14, 40, 0, 0.0, 0, 0)
end

# ---- MPS export: one flat ComplexF64 vector per site + the fusion-tree sub-block table -------------------------------
# A site tensor A : V_l ⊗ P ← V_r of TensorKit stores, per coupled sector c, a (rows × cols) column-major block inside
# `A.data`; rows run over the fusion trees (a, s → c), cols over the single tree (c ← c).  `A[f1, f2]` is a StridedView
# of that window: its parent is `A.data`, `sv.offset` the element offset, `sv.strides[1] == 1` and the stride of the last
# (right-bond) index the leading dimension.  That is exactly an `htn_subblock` -- no copy, any order.
# NORMALISATION (DESIGN.md section 2, INTEGRATION.md section 2): the library's blocks are "tilde" normalised; for the
# centre tensor x̃ = sqrt(2S_c + 1) t_c with c the RIGHT sector for AC and for right-canonical tensors.  The factor is
# applied to a copy of the data vector below.  UNVERIFIED against TensorKit's Clebsch-Gordan phase convention: validate once
# on a small state (norm, energy and singular values do not depend on it).
function export_mps(ψ::FiniteMPS, P::Int, Q::Int)
    L = length(ψ)
    bond_ptr, secs = Int32[0], HtnSector[]
    for b in 0:L
        V = b == 0 ? left_virtualspace(ψ, 1) : right_virtualspace(ψ, b)
        labs = sort([(label(c, P, Q, b)..., Int32(dim(V, c))) for c in sectors(V)])
        append!(secs, (HtnSector(l...) for l in labs))
        push!(bond_ptr, Int32(length(secs)))
    end
    sub_ptr, subs, data_ptr, data = Int32[0], HtnSubblock[], Int64[0], ComplexF64[]
    for i in 1:L
        A = i == 1 ? ψ.AC[1] : ψ.AR[i]                     # centre on site 1, right-canonical to its right
        d = copy(A.data)
        for (f1, f2) in fusiontrees(A)
            a, s = f1.uncoupled
            c = f2.uncoupled[1]
            sv = A[f1, f2]                                   # StridedView{ComplexF64,3}: (n_a, 1, n_c)
            ld = sv.strides[3]
            scale = sqrt(Float64(dim(c)))                    # tilde normalisation, see above
            n_a, _, n_c = size(sv)
            for col in 0:n_c-1, row in 0:n_a-1
                d[sv.offset + 1 + row + col * ld] *= scale
            end
            lN, lj = label(a, P, Q, i - 1)
            rN, rj = label(c, P, Q, i)
            push!(subs, HtnSubblock(lN, lj, Int32(site_index(s, P, Q) - 1), rN, rj, Int32(ld), Int64(sv.offset)))
        end
        append!(data, d)
        push!(sub_ptr, Int32(length(subs)))
        push!(data_ptr, Int64(length(data)))
    end
    return bond_ptr, secs, sub_ptr, subs, data_ptr, data
end

# ---- MPS import: htn_mps_bond / htn_mps_site_size / htn_mps_get_site per site -----------------------------------------
# Builds new virtual spaces from the bond tables the library returns, allocates TensorMaps on them and copies every
# sub-block into its fusion-tree window (undoing the tilde factor).  The centre is on site 1 after htn_dmrg2_sweep.
function import_mps(mps::Ptr{Cvoid}, ψ::FiniteMPS, P::Int, Q::Int)
    L = length(ψ)
    I = sectortype(ψ.AC[1])
    bond(b) = begin
        n = ccall((:htn_mps_bond, lib), Int32, (Ptr{Cvoid}, Int32, Ptr{HtnSector}), mps, b, C_NULL)
        out = Vector{HtnSector}(undef, n)
        ccall((:htn_mps_bond, lib), Int32, (Ptr{Cvoid}, Int32, Ptr{HtnSector}), mps, b, out)
        out
    end
    sector(rec::HtnSector, b) = I(isodd(rec.N), rec.j // 2, rec.N * Q - P * b)      # (parity, S, shifted charge)
    spaces = [Vect[I]((sector(r, b) => Int(r.count) for r in bond(b))...) for b in 0:L]
    phys = physicalspace(ψ, 1)
    tensors = map(1:L) do i
        A = zeros(ComplexF64, spaces[i] ⊗ phys, spaces[i+1])
        nb = ccall((:htn_mps_get_site, lib), Int32, (Ptr{Cvoid}, Int32, Ptr{HtnSubblock}, Ptr{ComplexF64}), mps, i - 1, C_NULL, C_NULL)
        sz = ccall((:htn_mps_site_size, lib), Int64, (Ptr{Cvoid}, Int32, Ptr{Int32}), mps, i - 1, C_NULL)
        subs, flat = Vector{HtnSubblock}(undef, nb), Vector{ComplexF64}(undef, max(sz, 1))
        ccall((:htn_mps_get_site, lib), Int32, (Ptr{Cvoid}, Int32, Ptr{HtnSubblock}, Ptr{ComplexF64}), mps, i - 1, subs, flat) == nb ||
            error(unsafe_string(ccall((:htn_last_error, lib), Cstring, ())))
        table = Dict((sb.lN, sb.lj, sb.s, sb.rN, sb.rj) => sb for sb in subs)
        for (f1, f2) in fusiontrees(A)
            a, s = f1.uncoupled
            c = f2.uncoupled[1]
            key = (label(a, P, Q, i - 1)..., Int32(site_index(s, P, Q) - 1), label(c, P, Q, i)...)
            haskey(table, key) || continue
            sb = table[key]
            sv = A[f1, f2]
            n_a, _, n_c = size(sv)
            scale = 1 / sqrt(Float64(dim(c)))
            for col in 1:n_c, row in 1:n_a
                sv[row, 1, col] = scale * flat[sb.off+row+(col-1)*sb.ld]
            end
        end
        A
    end
    return FiniteMPS(tensors; normalize = false)       # site 1 holds the centre; MPSKit re-gauges on construction
end

# ---- MPO export --------------------------------------------------------------------------------------------------------
# MPSKit 0.13: a FiniteMPOHamiltonian is a vector of BlockTensorKit.SparseBlockTensorMap W[i] with legs
# (V_left ⊗ P ← P ⊗ V_right), V_left / V_right being SumSpaces whose summands are the MPO "levels"; level 1 is the
# start ("nothing applied yet"), the last level the end ("term complete") (Jordan form, SURVEY App. A.3).  Every stored
# block W[i][j, 1, 1, k] is either a BraidingTensor (identity on P, levels of trivial charge) or a TensorMap
# V_j ⊗ P ← P ⊗ V_k whose single-sector virtual legs carry the charge of the operator string that is open between the
# sites.  The library wants, per MPO bond, the (dN, 2k) label of every level (electron number and doubled spin carried by the
# open string) and, per site, entries (wl, wr, op, coef) with `op` an index into a table of reduced site operators
# <out||O||in> over the three site multiplets.
# The reduced matrix element of a block is read off its fusion-tree sub-blocks: for t : V_j ⊗ P ← P ⊗ V_k with one-dimensional
# virtual sectors every sub-block is a 1x1x1x1 array, and (for the isometric fusion-tensor convention of TensorKit, src:264-292)
# its value IS the reduced element up to the recoupling factor the planner applies itself.  UNVERIFIED: whether TensorKit's
# (V_j ⊗ P ← P ⊗ V_k) tree basis needs an F-move to reach the library's <out|| O ||in> with the level spin coupled from the
# left (DESIGN.md section 2); for rank-0 operators (number, double occupancy, identity) no recoupling arises.
function export_mpo(H, P::Int, Q::Int)
    L = length(H)
    sym = HtnSymmetry(0, 3, (0, 1, 2, 0), (0, 1, 0, 0))          # HTN_SYM_SU2_U1: site multiplets (N, 2S) = (0,0), (1,1), (2,0)
    ops = HtnSiteOp[]
    opindex = Dict{Tuple{Int32,Int32,NTuple{16,Float64}},Int32}()
    level_ptr, levels, entry_ptr, entries = Int32[0], Int32[], Int32[0], HtnMpoEntry[]
    level_label(V) = begin                                          # one-sector auxiliary space -> (dN, 2k)
        c = only(sectors(V))
        (Int32((Int(c.sectors[3].charge)) ÷ Q), Int32(2 * c.sectors[2].j))
    end
    for b in 0:L
        Vsum = b == 0 ? left_virtualspace(H, 1) : right_virtualspace(H, b)
        for V in Vsum.spaces
            dN, k2 = level_label(V)
            push!(levels, dN, k2)
        end
        push!(level_ptr, Int32(length(levels) ÷ 2))
    end
    for i in 1:L
        W = H[i]
        for (idx, t) in nonzero_pairs(W)                            # BlockTensorKit: stored blocks only
            j, _, _, k = Tuple(idx)
            red = zeros(16)
            if t isa BraidingTensor
                red[1] = red[6] = red[11] = 1.0                     # identity on the three site multiplets
                kk, dN, coef = Int32(0), Int32(0), 1.0 + 0.0im
            else
                coef = 1.0 + 0.0im
                for (f1, f2) in fusiontrees(t)
                    sout = site_index(f1.uncoupled[2], P, Q)
                    sin = site_index(f2.uncoupled[1], P, Q)
                    red[(sout-1)*4+sin] = real(only(t[f1, f2]))
                end
                dNl, kl = level_label(space(t, 1))
                dNr, kr = level_label(space(t, 4)')
                kk, dN = Int32(abs(kr - kl)), Int32(dNr - dNl)     # rank and charge of the site operator itself
            end
            key = (kk, dN, Tuple(red))
            op = get!(opindex, key) do
                push!(ops, HtnSiteOp(kk, dN, Tuple(red)))
                Int32(length(ops) - 1)
            end
            push!(entries, HtnMpoEntry(Int32(j - 1), Int32(k - 1), op, 0, real(coef), imag(coef)))
        end
        push!(entry_ptr, Int32(length(entries)))
    end
    return sym, ops, level_ptr, levels, entry_ptr, entries
end

# ---- the algorithm struct + method ----------------------------------------------------------------------------------------
struct HIPDMRG2{T} <: MPSKit.Algorithm
    trscheme::T
    tol::Float64
    maxiter::Int
    krylovdim::Int
    P::Int
    Q::Int
end
HIPDMRG2(; trscheme, tol = 1e-6, maxiter = 100, krylovdim = 30, P = 1, Q = 1) = HIPDMRG2(trscheme, tol, maxiter, krylovdim, P, Q)

function MPSKit.find_groundstate(ψ::FiniteMPS, H, alg::HIPDMRG2, envs = nothing)
    ctx, mpo, mps = Ref{Ptr{Cvoid}}(C_NULL), Ref{Ptr{Cvoid}}(C_NULL), Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:htn_ctx_create, lib), Cint, (Int32, Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), 1, 0, C_NULL, ctx))     # HTN_BACKEND_HIP, device 0
    try
        sym, ops, level_ptr, levels, entry_ptr, entries = export_mpo(H, alg.P, alg.Q)
        check(ccall((:htn_mpo_create, lib), Cint,
            (Ptr{Cvoid}, Ref{HtnSymmetry}, Int32, Ptr{HtnSiteOp}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{HtnMpoEntry}, Ref{Ptr{Cvoid}}),
            ctx[], sym, length(H), ops, length(ops), level_ptr, levels, entry_ptr, entries, mpo))
        bond_ptr, sectors_, sub_ptr, subs, data_ptr, data = export_mps(ψ, alg.P, alg.Q)
        check(ccall((:htn_mps_create, lib), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{HtnSector}, Ptr{Int32}, Ptr{HtnSubblock}, Ptr{Int64}, Ptr{ComplexF64},
             Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
            ctx[], mpo[], length(ψ), bond_ptr, sectors_, sub_ptr, subs, data_ptr, data, C_NULL, C_NULL, mps))
        opts = sweep_opts(alg.trscheme; krylovdim = alg.krylovdim)
        E, Eprev, δ = Ref{Float64}(0.0), Inf, Inf
        for it in 1:alg.maxiter
            check(ccall((:htn_dmrg2_sweep, lib), Cint, (Ptr{Cvoid}, Ref{HtnSweepOpts}, Ptr{Cvoid}, Ref{Float64}), mps[], opts, C_NULL, E))
            δ = abs(E[] - Eprev) / length(ψ)
            Eprev = E[]
            δ < alg.tol && break
        end
        ψ′ = import_mps(mps[], ψ, alg.P, alg.Q)
        return ψ′, environments(ψ′, H), δ
    finally            # handles are reference counted: any order is safe
        mps[] == C_NULL || ccall((:htn_mps_destroy, lib), Cvoid, (Ptr{Cvoid},), mps[])
        mpo[] == C_NULL || ccall((:htn_mpo_destroy, lib), Cvoid, (Ptr{Cvoid},), mpo[])
        ccall((:htn_ctx_destroy, lib), Cvoid, (Ptr{Cvoid},), ctx[])
    end
end

# In compute_groundstate (src/HubbardFunctions.jl:1010) a finite-chain run then reads
#     ψ, envs, δ = find_groundstate(ψ₀, H, HubbardHIP.HIPDMRG2(; trscheme = truncbelow(10.0^(-svalue)), tol = tol, P = P, Q = Q))
# The infinite-chain IDMRG2 the reference hard-codes needs the growing-window loop of hubbardtn_amd/idmrg.py on this side
# (htn_mps_create with boundary environments, htn_mps_get_env / htn_mps_env_bond); it is not written here.
end # module
