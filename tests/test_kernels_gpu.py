"""GPU parity of every C-ABI primitive against its numpy statement (tests/emul.py)."""
import numpy as np
import pytest

from emul import NumpyOps
from hubbardtn_amd import abi
from ref_planner import TaskList

pytestmark = pytest.mark.gpu


def _rand_z(rng, n):
    return rng.standard_normal(n) + 1j * rng.standard_normal(n)


def _random_tasks(rng, nblocks, maxdim, maxk, nbuf_elems):
    """random grouped-GEMM problem exercising all ops, ragged sizes, COPY segments, empty blocks"""
    tl = TaskList()
    out_off = 0
    for b in range(nblocks):
        m, n = int(rng.integers(1, maxdim)), int(rng.integers(1, maxdim))
        ld = m + int(rng.integers(0, 3))
        tl.block(b, 1, out_off, m, n, ld)
        out_off += ld * n
        for s in range(int(rng.integers(0, 5))):
            alpha = complex(rng.standard_normal(), rng.standard_normal())
            if rng.random() < 0.2:
                ldb = m + int(rng.integers(0, 4))
                off = int(rng.integers(0, nbuf_elems - ldb * n - 1))
                tl.copy(b, 0, off, ldb, alpha)
                continue
            k = int(rng.integers(1, maxk))
            op_a, op_b = int(rng.integers(0, 3)), int(rng.integers(0, 3))
            lda = (m if op_a == abi.OP_N else k) + int(rng.integers(0, 3))
            ldb = (k if op_b == abi.OP_N else n) + int(rng.integers(0, 3))
            sa = lda * (k if op_a == abi.OP_N else m)
            sb = ldb * (n if op_b == abi.OP_N else k)
            a_off = int(rng.integers(0, nbuf_elems - sa - 1))
            b_off = int(rng.integers(0, nbuf_elems - sb - 1))
            tl.gemm(b, 0, a_off, lda, op_a, 2, b_off, ldb, op_b, k, alpha)
    return tl.finalize(), out_off


@pytest.mark.parametrize("seed,nblocks,maxdim,maxk", [(0, 40, 20, 30), (1, 25, 75, 90), (2, 6, 140, 300)])
def test_grouped_gemm_matches_numpy(hip_ops, seed, nblocks, maxdim, maxk):
    rng = np.random.default_rng(seed)
    nel = 200_000
    tasks, out_size = _random_tasks(rng, nblocks, maxdim, maxk, nel)
    src0, src2 = _rand_z(rng, nel), _rand_z(rng, nel)
    ref = np.zeros(out_size, dtype=np.complex128)
    emu = NumpyOps()
    emu.grouped_gemm([src0, ref, src2] + [None] * 5, emu.upload_tasks(tasks))
    d0, d2 = hip_ops.to_device(src0), hip_ops.to_device(src2)
    out = hip_ops.zeros_z(out_size)
    hip_ops.grouped_gemm([d0, out, d2] + [None] * 5, hip_ops.upload_tasks(tasks))
    got = hip_ops.to_host(out)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-12 * scale      # f64 MFMA, tolerance = accumulation order only
    # the same list through the balancing pass: tiles cut into their quadrants, layered launch order
    dev = hip_ops.upload_tasks(tasks, balance=True)
    out2 = hip_ops.zeros_z(out_size)
    hip_ops.grouped_gemm([d0, out2, d2] + [None] * 5, dev)
    assert np.abs(hip_ops.to_host(out2) - ref).max() <= 1e-12 * scale
    if maxdim > 32 and maxk > 80:
        assert dev[1] > tasks.ntiles                     # some multi-quadrant tile was long enough to be cut


def test_grouped_gemm_segments_longer_than_one_slab_and_every_quadrant_shape(hip_ops):
    """the kernel-level ABI also takes segments whose k exceeds one 16-deep slab (htn_tile.pad[1] = 0: the waves walk
    (segment, k offset) cursors) -- the planner never emits them, so they are built by hand here; block extents are chosen so
    that the tiles cover every quadrant shape of the kernel (one, two side by side, two stacked, four; 1-, 15-, 16-, 17-wide
    edges), with all op combinations, k not a multiple of 4 or 16, a COPY segment, and both with and without the library's
    balancing pass"""
    import ref_planner as pl
    rng = np.random.default_rng(11)
    nel = 300_000
    shapes = [(1, 1), (15, 33), (16, 16), (17, 5), (32, 32), (40, 50), (31, 64), (64, 17), (3, 48)]
    seg_rows, blk = [], []
    pos = out_off = 0
    for bi, (m, n) in enumerate(shapes):
        ld = m + bi % 3
        cnt = 0
        for (k, op_a, op_b) in [(37, 0, 0), (100, 1, 2), (5, 2, 1), (16, 0, 1), (61, 2, 0)][:2 + bi % 4]:
            lda = (m if op_a == abi.OP_N else k) + 1
            ldb = (k if op_b == abi.OP_N else n) + 2
            sa, sb = lda * (k if op_a == abi.OP_N else m), ldb * (n if op_b == abi.OP_N else k)
            al = complex(rng.standard_normal(), rng.standard_normal() if (bi + cnt) % 2 else 0.0)
            seg_rows.append((int(rng.integers(0, nel - sa - 1)), int(rng.integers(0, nel - sb - 1)), 0, 2, lda, ldb, k, op_a, op_b,
                             abi.SEG_GEMM, al.real, al.imag))
            cnt += 1
        ncopy = 0
        if bi % 2:
            seg_rows.append((0, int(rng.integers(0, nel - (m + 3) * n - 1)), 0, 2, 0, m + 3, 0, 0, 0, abi.SEG_COPY, 0.7, -0.2))
            cnt, ncopy = cnt + 1, 1
        blk.append((out_off, ld, m, n, pos, cnt, ncopy))
        pos += cnt
        out_off += ld * n
    segs = np.array(seg_rows, dtype=abi.SEG_DT)
    tiles = []
    for (off, ld, m, n, sb, sc, nc) in blk:
        for r0 in range(0, m, 32):
            for c0 in range(0, n, 32):
                t = np.zeros(1, dtype=abi.TILE_DT)[0]
                t["c_off"], t["buf_c"], t["ldc"], t["m"], t["n"], t["row0"], t["col0"] = off, 1, ld, min(32, m - r0), min(32, n - c0), r0, c0
                t["seg_begin"], t["seg_count"], t["pad0"], t["pad1"] = sb, sc, nc, 0
                tiles.append(t)
    tarr = np.array(tiles, dtype=abi.TILE_DT)
    tasks = pl.Tasks(tarr, len(tarr), segs, len(segs), 0)
    src0, src2 = _rand_z(rng, nel), _rand_z(rng, nel)
    ref = np.zeros(out_off, dtype=np.complex128)
    NumpyOps().grouped_gemm([src0, ref, src2] + [None] * 5, (tarr, len(tarr), segs))
    d0, d2 = hip_ops.to_device(src0), hip_ops.to_device(src2)
    for balance in (False, True):
        out = hip_ops.zeros_z(out_off)
        dev = hip_ops.upload_tasks(tasks, balance=balance)
        hip_ops.grouped_gemm([d0, out, d2] + [None] * 5, dev)
        assert np.abs(hip_ops.to_host(out) - ref).max() <= 1e-12 * np.abs(ref).max(), balance


def test_mfma_layout_identity_asymmetric(hip_ops):
    """A = I with an asymmetric B catches a transposed C/D lane map (cdna guide section 3)"""
    n = 32
    tl = TaskList()
    tl.block(0, 1, 0, n, n, n)
    tl.gemm(0, 0, 0, n, abi.OP_N, 2, 0, n, abi.OP_N, n, 1.0)
    tasks = tl.finalize()
    I = np.eye(n, dtype=np.complex128).T.reshape(-1)
    B = (np.arange(n)[:, None] * 100 + np.arange(n)[None, :]) * (1 + 0.5j)
    out = hip_ops.zeros_z(n * n)
    hip_ops.grouped_gemm([hip_ops.to_device(I), out, hip_ops.to_device(B.T.reshape(-1).copy())] + [None] * 5,
                         hip_ops.upload_tasks(tasks))
    got = hip_ops.to_host(out).reshape(n, n).T
    assert np.array_equal(got, B)


def test_krylov_vector_algebra(hip_ops):
    rng = np.random.default_rng(3)
    n, nv = 54_321, 17
    V = _rand_z(rng, nv * n)
    w = _rand_z(rng, n)
    dV, dw = hip_ops.to_device(V), hip_ops.to_device(w)
    out = hip_ops.zeros_z(nv)
    hip_ops.dots(dV, n, nv, dw, n, out)
    ref = np.array([np.vdot(V[i * n:(i + 1) * n], w) for i in range(nv)])
    got = hip_ops.to_host(out)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    out2 = hip_ops.zeros_z(nv)
    hip_ops.dots(dV, n, nv, dw, n, out2)
    assert np.array_equal(hip_ops.to_host(out2), got)          # deterministic reduction order
    coef = _rand_z(rng, nv)
    hip_ops.axpys(dw, dV, n, nv, hip_ops.to_device(coef), -1.0, n)
    wref = w - (coef[:, None] * V.reshape(nv, n)).sum(0)
    assert np.abs(hip_ops.to_host(dw) - wref).max() <= 1e-12 * np.abs(wref).max()
    nr = hip_ops.zeros_z(1)
    hip_ops.dots(dw, n, 1, dw, n, nr)
    hip_ops.scale_inv_sqrt(dw, dw, nr, n)
    assert abs(np.linalg.norm(hip_ops.to_host(dw)) - 1.0) < 1e-13


@pytest.mark.parametrize("shapes,acc", [([(1, 1), (2, 1), (5, 5), (17, 9), (64, 64), (70, 33)], True),
                                        ([(130, 130), (200, 90), (3, 2)], True),
                                        ([(60, 60), (40, 52), (140, 150), (3, 7)], False)])
def test_jacobi_svd_matches_lapack(hip_ops, shapes, acc):
    """G' = G J has orthogonal columns whose norms are the singular values; wide blocks (m < n) end with
    n - m zero columns; J is only written when HTN_SVD_ACCUMULATE is set"""
    rng = np.random.default_rng(4)
    desc = np.zeros(len(shapes), dtype=abi.SVD_DT)
    go = vo = so = 0
    mats = []
    for i, (m, n) in enumerate(shapes):
        desc[i] = (go, vo, so, m, n, abi.SVD_ACCUMULATE if acc else 0, 0)
        r = min(m, n)
        # graded spectrum like a Schmidt spectrum: singular values over 12 decades
        U, _ = np.linalg.qr(_rand_z(rng, m * r).reshape(m, r))
        W, _ = np.linalg.qr(_rand_z(rng, n * r).reshape(n, r))
        s = 10.0 ** (-12 * np.arange(r) / max(r - 1, 1))
        mats.append((U * s) @ W.conj().T)
        go, vo, so = go + m * n, vo + n * n, so + n
    G = np.concatenate([M.T.reshape(-1) for M in mats])
    dG = hip_ops.to_device(G)
    sentinel = 7.0 + 3.0j
    dV = hip_ops.to_device(np.full(vo, sentinel))
    dS, info = hip_ops.empty_f64(so), hip_ops.empty_i32(len(shapes))
    hip_ops.jacobi_svd(dG, dV, dS, hip_ops.to_device(desc), len(shapes), max(m for m, _ in shapes), 40, 1e-14, info)
    Gp, J, S, inf = hip_ops.to_host(dG), hip_ops.to_host(dV), hip_ops.to_host(dS), hip_ops.to_host(info)
    assert inf.min() >= 0, inf          # negative = not converged; 0 = nothing to do (n < 2)
    if not acc:
        assert np.all(J == sentinel)
    for i, (m, n) in enumerate(shapes):
        d = desc[i]
        gp = Gp[d["g_off"]:d["g_off"] + m * n].reshape(n, m).T
        s = S[d["s_off"]:d["s_off"] + n]
        ref = np.zeros(n)
        ref[:min(m, n)] = np.linalg.svd(mats[i], compute_uv=False)
        got = np.sort(s)[::-1]
        # LAPACK (the reference's zgesvd) is accurate to eps * sigma_max ABSOLUTE only, so the relative
        # 1e-8 bar of north_star is checkable against it down to sigma ~ 1e-6 sigma_max
        assert np.abs(got - ref).max() <= 1e-13 * ref[0]
        big = ref > 1e-6 * ref[0]
        assert np.abs(got[big] / ref[big] - 1).max() < 1e-8
        gram = gp.conj().T @ gp
        dg = np.sqrt(np.abs(np.diag(gram)))
        live = dg > 1e-14 * ref[0]        # columns below 1e-15 |G|_F are treated as zero by the kernel
        off = np.abs(gram - np.diag(np.diag(gram)))[np.ix_(live, live)]
        assert (off <= 1e-13 * np.outer(dg[live], dg[live])).all()      # live columns mutually orthogonal
        assert live.sum() >= (ref > 1e-13 * ref[0]).sum()
        if acc:
            j = J[d["v_off"]:d["v_off"] + n * n].reshape(n, n).T
            assert np.abs(j.conj().T @ j - np.eye(n)).max() < 1e-13          # J unitary
            assert np.abs(mats[i] @ j - gp).max() < 1e-13                     # G' = M J


@pytest.mark.parametrize("multi", [False, True])
@pytest.mark.parametrize("shapes", [[(1, 1), (2, 3), (40, 40), (33, 57), (57, 33)], [(107, 107), (150, 93), (93, 150)],
                                    [(230, 230), (60, 60), (300, 170), (129, 140)],
                                    # <= 256 rows and >= 160 columns: the pivoted QR runs with helper workgroups AND with all 16
                                    # waves of a workgroup on its chunks of the trailing update (qr_trailing_coop)
                                    [(202, 202), (180, 200), (96, 210)]])
def test_jacobi_svd_qr_preconditioned(hip_ops, shapes, multi):
    """HTN_SVD_QRCP: G0 (m0 x n0) -> (right singular vectors of G0) x Sigma, n0 x min(m0, n0), rows in the
    ORIGINAL column order of G0; few sweeps on graded spectra"""
    rng = np.random.default_rng(8)
    desc = np.zeros(len(shapes), dtype=abi.SVD_DT)
    go = vo = so = 0
    mats = []
    for i, (m0, n0) in enumerate(shapes):
        r = min(m0, n0)
        desc[i] = (go, vo, so, n0, r, abi.SVD_QRCP, m0)
        U, _ = np.linalg.qr(_rand_z(rng, m0 * r).reshape(m0, r))
        W, _ = np.linalg.qr(_rand_z(rng, n0 * r).reshape(n0, r))
        s = 10.0 ** (-12 * np.arange(r) / max(r - 1, 1))
        mats.append((U * s) @ W.conj().T)
        go, vo, so = go + m0 * n0, vo + ((n0 + 63) // 64 * 64) * r, so + r      # v region: padded R^H workspace
    dG = hip_ops.to_device(np.concatenate([M.T.reshape(-1) for M in mats]))
    dV, dS, info = hip_ops.zeros_z(vo), hip_ops.empty_f64(so), hip_ops.empty_i32(len(shapes))
    # multi = True hands the host copy of the descriptors over: blocks larger than one CU's LDS then take the
    # multi-launch block-Jacobi path (panel pairs on different CUs)
    hip_ops.jacobi_svd(dG, dV, dS, hip_ops.to_device(desc), len(shapes), max(max(s_) for s_ in shapes), 40, 1e-14, info,
                       desc_host=desc if multi else None)
    Gp, S, inf = hip_ops.to_host(dG), hip_ops.to_host(dS), hip_ops.to_host(info)
    assert inf.min() >= 0, inf
    assert inf.max() <= 12, inf                   # preconditioning keeps the sweep count small
    for i, (m0, n0) in enumerate(shapes):
        d = desc[i]
        r = min(m0, n0)
        out = Gp[d["g_off"]:d["g_off"] + n0 * r].reshape(r, n0).T        # n0 x r, ld n0
        s = S[d["s_off"]:d["s_off"] + r]
        ref = np.linalg.svd(mats[i], compute_uv=False)
        order = np.argsort(-s)
        assert np.abs(s[order] - ref).max() <= 1e-13 * ref[0]
        big = ref > 1e-6 * ref[0]
        assert np.abs(s[order][big] / ref[big] - 1).max() < 1e-8
        live = s > 1e-13 * ref[0]
        Viso = out[:, live] / s[live]
        assert np.abs(Viso.conj().T @ Viso - np.eye(live.sum())).max() < 1e-12       # orthonormal columns
        # they are right singular vectors of G0:  |G0 v| = sigma and G0^H G0 v = sigma^2 v
        assert np.abs(np.linalg.norm(mats[i] @ Viso, axis=0) - s[live]).max() <= 1e-12 * ref[0]
        resid = mats[i].conj().T @ (mats[i] @ Viso) - Viso * s[live] ** 2
        assert np.abs(resid).max() <= 1e-12 * ref[0] ** 2


def test_large_block_svd_result_does_not_depend_on_the_sweep_hint(hip_ops):
    """htn_svd_opts.sweeps_hint only bounds the SPECULATIVE enqueue of outer sweeps (the sweep expected to be the last is
    not followed by an empty one): singular values, vectors and the reported sweep count are bit-identical for no hint,
    the right hint, a hint that is too small (extra sweeps are then enqueued one at a time) and one that is too large"""
    rng = np.random.default_rng(21)
    shapes = [(230, 230), (300, 170), (330, 310)]        # (310 columns: the pivoted QR runs with helper workgroups, above 288)
    desc = np.zeros(len(shapes), dtype=abi.SVD_DT)
    go = vo = so = 0
    mats = []
    for i, (m0, n0) in enumerate(shapes):
        r = min(m0, n0)
        desc[i] = (go, vo, so, n0, r, abi.SVD_QRCP, m0)
        U, _ = np.linalg.qr(_rand_z(rng, m0 * r).reshape(m0, r))
        W, _ = np.linalg.qr(_rand_z(rng, n0 * r).reshape(n0, r))
        mats.append((U * 10.0 ** (-10 * np.arange(r) / (r - 1))) @ W.conj().T)
        go, vo, so = go + m0 * n0, vo + ((n0 + 63) // 64 * 64) * r, so + r
    src = hip_ops.to_device(np.concatenate([M.T.reshape(-1) for M in mats]))
    d_desc = hip_ops.to_device(desc)
    runs = []
    used0 = None
    for hint in (0, None, 2, 30):
        dG = src.clone()
        dV, dS, info = hip_ops.zeros_z(vo), hip_ops.empty_f64(so), hip_ops.empty_i32(len(shapes))
        used = hip_ops.jacobi_svd(dG, dV, dS, d_desc, len(shapes), 330, 40, 1e-14, info, desc_host=desc,
                                  sweeps_hint=used0 if hint is None else hint)
        if used0 is None:
            used0 = used
        runs.append((used, hip_ops.to_host(dS), hip_ops.to_host(dG), hip_ops.to_host(info)))
    assert 3 <= used0 <= 12
    for used, S, G, inf in runs[1:]:
        assert used == used0 and np.array_equal(S, runs[0][1]) and np.array_equal(G, runs[0][2]) and np.array_equal(inf, runs[0][3])
    for i, (m0, n0) in enumerate(shapes):
        ref = np.linalg.svd(mats[i], compute_uv=False)
        got = np.sort(runs[0][1][desc[i]["s_off"]:desc[i]["s_off"] + min(m0, n0)])[::-1]
        assert np.abs(got - ref).max() <= 1e-13 * ref[0]


@pytest.mark.parametrize("multi", [False, True])
def test_jacobi_svd_rank_deficient_large_blocks(hip_ops, multi):
    """blocks whose numerical rank is below min(m0, n0) -- the two-site block of a bond whose neighbour is not
    saturated yet -- through the blocked QR (early panel exit) and the Gram-matrix panel visits (zero columns)"""
    rng = np.random.default_rng(11)
    shapes = [(200, 200, 120), (150, 180, 33), (260, 140, 1), (120, 120, 120)]
    desc = np.zeros(len(shapes), dtype=abi.SVD_DT)
    go = vo = so = 0
    mats = []
    for i, (m0, n0, rank) in enumerate(shapes):
        r = min(m0, n0)
        desc[i] = (go, vo, so, n0, r, abi.SVD_QRCP, m0)
        U, _ = np.linalg.qr(_rand_z(rng, m0 * rank).reshape(m0, rank))
        W, _ = np.linalg.qr(_rand_z(rng, n0 * rank).reshape(n0, rank))
        s = 10.0 ** (-8 * np.arange(rank) / max(rank - 1, 1))
        mats.append((U * s) @ W.conj().T)
        go, vo, so = go + m0 * n0, vo + ((n0 + 63) // 64 * 64) * r, so + r
    dG = hip_ops.to_device(np.concatenate([M.T.reshape(-1) for M in mats]))
    dV, dS, info = hip_ops.zeros_z(vo), hip_ops.empty_f64(so), hip_ops.empty_i32(len(shapes))
    hip_ops.jacobi_svd(dG, dV, dS, hip_ops.to_device(desc), len(shapes), 260, 40, 1e-14, info,
                       desc_host=desc if multi else None)
    Gp, S, inf = hip_ops.to_host(dG), hip_ops.to_host(dS), hip_ops.to_host(info)
    assert inf.min() >= 0, inf
    for i, (m0, n0, rank) in enumerate(shapes):
        d = desc[i]
        r = min(m0, n0)
        out = Gp[d["g_off"]:d["g_off"] + n0 * r].reshape(r, n0).T
        s = S[d["s_off"]:d["s_off"] + r]
        assert np.isfinite(out).all() and np.isfinite(s).all()
        ref = np.linalg.svd(mats[i], compute_uv=False)
        order = np.argsort(-s)
        assert np.abs(s[order] - ref).max() <= 1e-13 * ref[0]
        live = s > 1e-11 * ref[0]
        assert live.sum() == rank
        Viso = out[:, live] / s[live]
        assert np.abs(Viso.conj().T @ Viso - np.eye(rank)).max() < 1e-12
        assert np.abs(np.linalg.norm(mats[i] @ Viso, axis=0) - s[live]).max() <= 1e-12 * ref[0]


def test_rank_revealing_qr_cut(hip_ops):
    """htn_svd_opts.rank_cut (per call): directions below the cut are dropped before the Jacobi sweeps (their singular values
    come back as 0); the kept ones move by at most cut^2 / (2 sigma) (interlacing) and stay an isometry"""
    rng = np.random.default_rng(21)
    shapes = [(220, 220), (150, 190), (260, 130), (90, 90), (60, 100), (70, 40)]     # large-block path and one-CU kernel
    cut = 1e-6
    desc = np.zeros(len(shapes), dtype=abi.SVD_DT)
    go = vo = so = 0
    mats = []
    for i, (m0, n0) in enumerate(shapes):
        r = min(m0, n0)
        desc[i] = (go, vo, so, n0, r, abi.SVD_QRCP, m0)
        U, _ = np.linalg.qr(_rand_z(rng, m0 * r).reshape(m0, r))
        W, _ = np.linalg.qr(_rand_z(rng, n0 * r).reshape(n0, r))
        s = 10.0 ** (-12 * np.arange(r) / max(r - 1, 1))
        mats.append((U * s) @ W.conj().T)
        go, vo, so = go + m0 * n0, vo + ((n0 + 63) // 64 * 64) * r, so + r
    dG = hip_ops.to_device(np.concatenate([M.T.reshape(-1) for M in mats]))
    dV, dS, info = hip_ops.zeros_z(vo), hip_ops.empty_f64(so), hip_ops.empty_i32(len(shapes))
    hip_ops.jacobi_svd(dG, dV, dS, hip_ops.to_device(desc), len(shapes), 260, 40, 1e-14, info, desc_host=desc,
                       rank_cut=cut)
    Gp, S, inf = hip_ops.to_host(dG), hip_ops.to_host(dS), hip_ops.to_host(info)
    assert inf.min() >= 0, inf
    for i, (m0, n0) in enumerate(shapes):
        d = desc[i]
        r = min(m0, n0)
        out = Gp[d["g_off"]:d["g_off"] + n0 * r].reshape(r, n0).T
        s = np.sort(S[d["s_off"]:d["s_off"] + r])[::-1]
        ref = np.linalg.svd(mats[i], compute_uv=False)
        nlive = int((s > 0).sum())
        assert nlive < r                                              # something was dropped ...
        assert ref[nlive] <= cut * 1.0001 if nlive < r else True      # ... and only what lies below the cut
        assert ref[max(nlive - 17, 0)] > 0.0
        big = ref[:nlive] > 10 * cut
        assert np.abs(s[:nlive][big] - ref[:nlive][big]).max() <= cut ** 2 / (2 * 10 * cut) + 1e-13
        sraw = S[d["s_off"]:d["s_off"] + r]
        live = sraw > 0
        Viso = out[:, live] / sraw[live]
        assert np.abs(Viso.conj().T @ Viso - np.eye(live.sum())).max() < 1e-12
        keep = sraw > 100 * cut                                        # well above the cut: still singular pairs of G0
        Vk = out[:, keep] / sraw[keep]
        assert np.abs(np.linalg.norm(mats[i] @ Vk, axis=0) - sraw[keep]).max() <= 1e-8


def test_batched_copy_matches_numpy(hip_ops):
    rng = np.random.default_rng(5)
    src = _rand_z(rng, 5000)
    scl = rng.random(200) + 0.5
    idx = rng.integers(0, 9, size=64).astype(np.int32)
    items = np.zeros(6, dtype=abi.COPY_DT)
    cfg = [(abi.OP_N, 1, 1, 1), (abi.OP_N, 1, -1, 0), (abi.OP_C, 0, 0, 0), (abi.OP_C, 0, 0, 1), (abi.OP_N, 0, -1, 0),
           (abi.OP_C, 1, -1, 0)]
    off = 0
    for k, (op, gd, sd, inv) in enumerate(cfg):
        rows, cols = 7 + k, 5 + 2 * k
        it = items[k]
        it["dst_off"], it["src_off"], it["idx_off"], it["scl_off"] = off, 100 * k, (3 * k if k != 4 else -1), 10 * k
        it["rows"], it["cols"], it["ldd"], it["lds"] = rows, cols, rows + 1, 23
        it["op"], it["gather_dim"], it["scale_dim"], it["inv_norm"] = op, gd, sd, inv
        off += (rows + 1) * cols
    ref = np.zeros(off, dtype=np.complex128)
    NumpyOps().batched_copy(ref, src, idx, scl, items, len(items), 0.7)
    dst = hip_ops.zeros_z(off)
    hip_ops.batched_copy(dst, hip_ops.to_device(src), hip_ops.to_device(idx), hip_ops.to_device(scl),
                         hip_ops.to_device(items), len(items), 0.7)
    assert np.abs(hip_ops.to_host(dst) - ref).max() < 1e-14


def test_lanczos_driver_matches_dense_eigh(hip_ops):
    """htn_lanczos_z on a Hermitian map given as one grouped-GEMM stage (y = H x, H dense 200x200 here,
    x stored as a 200 x 3 block so the stage exercises tiles + segments) vs numpy eigh"""
    rng = np.random.default_rng(6)
    m, nc = 200, 3
    H = _rand_z(rng, m * m).reshape(m, m)
    H = H + H.conj().T
    tl = TaskList()
    tl.block(0, 1, 0, m, nc, m)
    tl.gemm(0, 2, 0, m, abi.OP_N, 0, 0, m, abi.OP_N, m, 1.0)
    tasks = tl.finalize()
    n = m * nc
    kd = 30
    V = hip_ops.zeros_z((kd + 2) * n)
    x0 = _rand_z(rng, n)
    V[0:n] = hip_ops.to_device(x0)
    stages = [([None, None, hip_ops.to_device(H.T.reshape(-1).copy())] + [None] * 5, hip_ops.upload_tasks(tasks))]
    eig, nmv, res = hip_ops.lanczos(stages, 0, 1, V, n, kd, 1e-10, 20)
    ref = np.linalg.eigvalsh(H)[0]
    assert abs(eig - ref) <= 1e-9 * abs(ref)
    x = hip_ops.to_host(V[0:n]).reshape(nc, m).T
    assert abs(np.linalg.norm(x) - 1) < 1e-12
    assert np.linalg.norm(H @ x - eig * x) < 1e-6


@pytest.mark.parametrize("seed,nblocks,maxdim,maxk", [(5, 12, 70, 900), (6, 3, 140, 2500)])
def test_grouped_gemm_split_k_across_workgroups_matches_numpy(hip_ops, seed, nblocks, maxdim, maxk):
    """tiles with long K loops are cut into parts by the library's balancing pass (htn_balance_tiles): the parts run on
    different workgroups / XCDs and meet through the ticket + slab workspace; the result must equal the numpy statement
    of the UNSPLIT list, be bit-identical from launch to launch (parts are added in part order, whatever the arrival
    order), and leave the tickets at zero"""
    rng = np.random.default_rng(seed)
    nel = 400_000
    tasks, out_size = _random_tasks(rng, nblocks, maxdim, maxk, nel)
    src0, src2 = _rand_z(rng, nel), _rand_z(rng, nel)
    ref = np.zeros(out_size, dtype=np.complex128)
    emu = NumpyOps()
    emu.grouped_gemm([src0, ref, src2] + [None] * 5, emu.upload_tasks(tasks))
    d0, d2 = hip_ops.to_device(src0), hip_ops.to_device(src2)
    dev = hip_ops.upload_tasks(tasks, balance=True)
    tiles_h = hip_ops.to_host(dev[0]).view(abi.TILE_DT)[:dev[1]]
    assert dev[1] > tasks.ntiles and int(tiles_h["nparts"].max()) >= 2          # something was split
    # the numpy emulator on the balanced list agrees with the unsplit statement (host-side check of the pass itself)
    chk = np.zeros(out_size, dtype=np.complex128)
    emu.grouped_gemm([src0, chk, src2] + [None] * 5, (tiles_h, dev[1], tasks.segs))
    assert np.abs(chk - ref).max() <= 1e-12 * np.abs(ref).max()
    outs = []
    for _ in range(3):
        out = hip_ops.zeros_z(out_size)
        hip_ops.grouped_gemm([d0, out, d2] + [None] * 5, dev)
        outs.append(hip_ops.to_host(out))
    assert np.abs(outs[0] - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    tickets = hip_ops.to_host(hip_ops._ws)[:abi.WS_TICKET_ELEMS].view(np.int32)
    assert not tickets.any()


def test_ring_jacobi_many_uneven_blocks_in_several_batches(hip_ops):
    """the one-launch ring block Jacobi under uneven load: 18 blocks between 100 and 430 columns in one call need more CU
    slots than the chip has (the launch is cut into batches of co-resident workgroups), blocks of 100 columns run beside
    blocks of 430, the 16-lane and the 64-lane forms and the helper-workgroup QR side by side with the one-CU kernel of
    the small blocks.  Every block must come back as a right-singular-vector matrix times Sigma of ITS matrix."""
    rng = np.random.default_rng(29)
    shapes = [(430, 420), (420, 430), (400, 410), (415, 400), (300, 310), (330, 300), (310, 320), (305, 300), (200, 210), (220, 200),
              (202, 202), (190, 230), (128, 120), (100, 140), (130, 110), (150, 100), (60, 70), (30, 30)]
    desc = np.zeros(len(shapes), dtype=abi.SVD_DT)
    go = vo = so = 0
    mats = []
    for i, (m0, n0) in enumerate(shapes):
        r = min(m0, n0)
        desc[i] = (go, vo, so, n0, r, abi.SVD_QRCP, m0)
        U, _ = np.linalg.qr(_rand_z(rng, m0 * r).reshape(m0, r))
        W, _ = np.linalg.qr(_rand_z(rng, n0 * r).reshape(n0, r))
        s = 10.0 ** (-9 * np.arange(r) / max(r - 1, 1)) * (1.0 + 0.3 * i)
        mats.append((U * s) @ W.conj().T)
        go, vo, so = go + m0 * n0, vo + ((n0 + 63) // 64 * 64) * r, so + r
    dG = hip_ops.to_device(np.concatenate([M.T.reshape(-1) for M in mats]))
    dV, dS, info = hip_ops.zeros_z(vo), hip_ops.empty_f64(so), hip_ops.empty_i32(len(shapes))
    used = hip_ops.jacobi_svd(dG, dV, dS, hip_ops.to_device(desc), len(shapes), 430, 40, 1e-14, info, desc_host=desc)
    Gp, S, inf = hip_ops.to_host(dG), hip_ops.to_host(dS), hip_ops.to_host(info)
    assert inf.min() >= 0 and inf.max() <= 13 and 3 <= used <= 13, (inf, used)
    for i, (m0, n0) in enumerate(shapes):
        d = desc[i]
        r = min(m0, n0)
        out = Gp[d["g_off"]:d["g_off"] + n0 * r].reshape(r, n0).T
        s = S[d["s_off"]:d["s_off"] + r]
        ref = np.linalg.svd(mats[i], compute_uv=False)
        order = np.argsort(-s)
        assert np.abs(s[order] - ref).max() <= 1e-13 * ref[0], i
        big = ref > 1e-6 * ref[0]
        assert np.abs(s[order][big] / ref[big] - 1).max() < 1e-8, i
        live = s > 1e-12 * ref[0]
        Viso = out[:, live] / s[live]
        assert np.abs(Viso.conj().T @ Viso - np.eye(live.sum())).max() < 1e-12, i
        assert np.abs(np.linalg.norm(mats[i] @ Viso, axis=0) - s[live]).max() <= 1e-12 * ref[0], i
