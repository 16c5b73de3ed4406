"""On-disk formats of the hot path's results: the reference's result cache and per-site state files.

  produce_or_load  -- `produce_groundstate` (src/HubbardFunctions.jl:1133-1166): DrWatson `produce_or_load(compute_groundstate,
                      simul, datadir("sims", name(simul)); prefix=...)`; the result is written once under
                      data/sims/<OB|MB>/<prefix>_<savename(simul)> and reloaded unless force=true.
  save_state / load_state -- src:1669-1691: one file per site holding `convert(Dict, AL[i])`, i.e. the site tensor as a
                      dictionary {fusion-tree sub-block -> matrix}.

File CONTAINER: the reference writes JLD2 (an HDF5 dialect); neither JLD2 nor HDF5 is readable or writable in this
image, so the same dictionaries are stored as NumPy .npz archives (names end in .npz instead of .jld2).  Directory
layout, prefixes and the DrWatson `savename` rule (fields of type Real / String sorted by name, 3 significant digits)
follow the reference; byte-level interchange with Julia needs an HDF5 writer and is NOT claimed.
Site tensors are stored in this library's Euclidean ("tilde") normalisation; INTEGRATION.md section 2 gives the
one-real-factor-per-block conversion to TensorKit's reduced data.
"""
from __future__ import annotations

import json
import os

import numpy as np

from .models import MB_Sim, MBC_Sim, OB_Sim, OBC_Sim2


def datadir(*parts):
    """DrWatson's datadir(): <project>/data/...; project = $HTN_PROJECT_DIR or the current directory"""
    return os.path.join(os.environ.get("HTN_PROJECT_DIR", os.getcwd()), "data", *parts)


def _fmt(v):
    if isinstance(v, (bool, np.bool_)):
        return "true" if v else "false"
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    if isinstance(v, (float, np.floating)):
        r = float(f"{float(v):.3g}")                    # DrWatson default: sigdigits = 3
        return repr(r) if r != int(r) or abs(r) >= 1e16 else f"{r:.1f}"
    return str(v)


def savename(simul) -> str:
    """DrWatson.savename of the model struct: scalar fields (Real, String) as key=value sorted by key, joined by '_'.
    Vector / matrix fields (t, u, J) and kwargs are skipped by DrWatson's default allowedtypes."""
    if isinstance(simul, OB_Sim):
        fields = {"P": simul.P, "Q": simul.Q, "bond_dim": simul.bond_dim, "period": simul.period, "svalue": simul.svalue,
                  "μ": simul.mu}
    else:
        fields = {"P": simul.P, "Q": simul.Q, "bond_dim": simul.bond_dim, "svalue": simul.svalue}
    if isinstance(simul, (OBC_Sim2, MBC_Sim)):             # the chemical-potential structs have no P, Q fields (src:154-238)
        fields.pop("P"), fields.pop("Q")
    return "_".join(f"{k}={_fmt(v)}" for k, v in sorted(fields.items()))


def _jl_vec(v):
    return "[" + "_".join(repr(float(x)) for x in v) + "]"


def cache_name(simul, what="groundstate"):
    """(sub-directory, file stem) of the reference's cache entry (src:1134-1166)"""
    spin = "spin_" if simul.kwargs.get("spin", False) else "nospin_"
    if isinstance(simul, MB_Sim):
        prefix = f"{what}_{spin}{simul.kwargs.get('code', '')}"
        sub = "MBC" if isinstance(simul, MBC_Sim) else "MB"
    else:
        U13 = simul.kwargs.get("U13", [0.0])
        JMs = simul.kwargs.get("JMs", (0.0, 0.0))
        prefix = (f"{what}_{spin}t{_jl_vec(simul.t)}_u{_jl_vec(simul.u)}_J{_jl_vec(simul.J)}_U13{_jl_vec(U13)}"
                  f"_JMs{float(JMs[0])!r}_{float(JMs[1])!r}")
        sub = "OBC" if isinstance(simul, OBC_Sim2) else "OB"
    L = simul.kwargs.get("L")
    tail = savename(simul) + (f"_L={int(L)}" if L else "")           # finite chains (not in the reference) get their length
    return sub, f"{prefix}_{tail}"


# ---- state <-> dictionaries ----------------------------------------------------------------------------------------
def state_dicts(eng):
    """per-site dictionaries of an engine's MPS: [{'kind': 'L'|'R', 'blocks': {(l, s, r): matrix}}], and the bond tables"""
    sites = [{"kind": eng.site_kind(i), "blocks": eng.download_site(i)} for i in range(eng.L)]
    bonds = [dict(b.dims) for b in eng.bonds]
    return sites, bonds


def _pack_site(d):
    out = {"kind": np.array(d["kind"])}
    keys = sorted(d["blocks"])
    out["labels"] = np.array([[l[0], l[1], s, r[0], r[1]] for (l, s, r) in keys], dtype=np.int32).reshape(-1, 5)
    for k, key in enumerate(keys):
        out[f"b{k}"] = np.asarray(d["blocks"][key], dtype=np.complex128)
    return out


def _unpack_site(z):
    labels = z["labels"]
    blocks = {((int(a), int(b)), int(s), (int(c), int(d))): z[f"b{k}"] for k, (a, b, s, c, d) in enumerate(labels)}
    return {"kind": str(z["kind"]), "blocks": blocks}


def save_state(psi, path: str, name: str):
    """src:1669-1677: one file per site, `state<i>.npz` (1-based like the reference), under path/name"""
    eng = psi.engine if hasattr(psi, "engine") else psi.result.engine
    path = os.path.join(path, name)
    os.makedirs(path, exist_ok=False)                     # the reference's mkdir fails on an existing directory, too
    sites, bonds = state_dicts(eng)
    for i, d in enumerate(sites, start=1):
        np.savez(os.path.join(path, f"state{i}.npz"), **_pack_site(d))
    with open(os.path.join(path, "bonds.json"), "w") as f:
        json.dump([[[N, j, n] for (N, j), n in sorted(b.items())] for b in bonds], f)
    return path


def load_state(path: str):
    """src:1679-1691: reads state1 .. stateN -> (bonds, [site dict]) ready for engine.DMRG2 / initialize from state"""
    n = len([e for e in os.listdir(path) if e.startswith("state") and e.endswith(".npz")])
    sites = []
    for i in range(1, n + 1):
        with np.load(os.path.join(path, f"state{i}.npz"), allow_pickle=False) as z:
            sites.append(_unpack_site(z))
    with open(os.path.join(path, "bonds.json")) as f:
        bonds = [{(N, j): n for N, j, n in b} for b in json.load(f)]
    return bonds, sites


# ---- result cache ---------------------------------------------------------------------------------------------------
def produce_or_load(compute, simul, force=False, directory=None, **kw):
    """DrWatson.produce_or_load for compute_groundstate: returns the result dictionary {"groundstate", "environments",
    "ham", "delta", "config"} -- loaded from the cache entry when it exists (the state is re-instantiated on the device
    from the stored site tensors, no sweep is run), computed and stored otherwise.  `cache=False` in kw, or models given
    an explicit init_state, bypass the cache like a direct compute_groundstate call."""
    from . import api
    use_cache = kw.pop("cache", True) and kw.get("init_state") is None
    if not use_cache:
        return compute(simul, **kw)
    if kw.get("L") and not simul.kwargs.get("L"):
        simul.kwargs["L"] = kw["L"]
    sub, stem = cache_name(simul)
    if kw.get("chi"):                                     # fixed-D runs (not in the reference) are separate entries
        stem += f"_chi={int(kw['chi'])}"
    directory = directory or datadir("sims", sub)
    entry = os.path.join(directory, stem)
    if os.path.isdir(entry) and not force:
        return api._load_result(simul, entry, **kw)
    res = compute(simul, **kw)
    try:
        os.makedirs(directory, exist_ok=True)
        if os.path.isdir(entry):
            import shutil
            shutil.rmtree(entry)
        api._save_result(res, entry)
    except OSError:
        pass                                              # a read-only project directory must not lose the result
    return res
