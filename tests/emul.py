"""TEST INFRASTRUCTURE: numpy interpreter of the C-ABI primitives (include/hubbardtn_hip.h).

Lets the `-m "not gpu"` suite execute the host logic (planner task lists, sweep driver) on CPU
and compare it with the oracle.  It is never imported by the product package; the product's only
device-ops provider is hubbardtn_amd.device.HipOps.
"""
import numpy as np

from hubbardtn_amd import abi


class NumpyOps:
    name = "numpy-emulator"

    def __init__(self, seed=0):
        self.rng = np.random.default_rng(seed)
        self.launches = 0

    def empty_z(self, n):
        return np.full(int(n), np.nan + 1j * np.nan, dtype=np.complex128)

    def zeros_z(self, n):
        return np.zeros(int(n), dtype=np.complex128)

    def empty_f64(self, n):
        return np.full(int(n), np.nan)

    def empty_i32(self, n):
        return np.zeros(int(n), dtype=np.int32)

    def to_device(self, arr):
        return np.array(arr, copy=True)

    def to_device_packed(self, arrays):
        return [np.array(a, copy=True) for a in arrays]

    def to_host(self, t):
        return np.array(t, copy=True)

    def zero(self, t):
        t[...] = 0

    def scale_inplace(self, t, f):
        t *= f

    def sync(self):
        pass

    def upload_tasks(self, tasks):
        return (tasks.tiles.copy(), tasks.ntiles, tasks.segs.copy())

    @staticmethod
    def _op(buf, off, ld, op, rows, cols, r0, c0):
        """rows x cols window of op(M) starting at (r0, c0)"""
        i = (r0 + np.arange(rows))[:, None]
        j = (c0 + np.arange(cols))[None, :]
        if op == abi.OP_N:
            return buf[off + i + j * ld]
        v = buf[off + j + i * ld]
        return v.conj() if op == abi.OP_C else v

    def grouped_gemm(self, bufs, dev_tasks, tag=None, flops=0):
        tiles, ntiles, segs = dev_tasks
        self.launches += 1
        outs = []
        for t in tiles[:ntiles]:
            m, n, r0, c0 = int(t["m"]), int(t["n"]), int(t["row0"]), int(t["col0"])
            acc = np.zeros((m, n), dtype=np.complex128)
            for s in segs[int(t["seg_begin"]):int(t["seg_begin"]) + int(t["seg_count"])]:
                alpha = complex(s["alpha_re"], s["alpha_im"])
                B = bufs[int(s["buf_b"])]
                if int(s["type"]) == abi.SEG_COPY:
                    acc += alpha * self._op(B, int(s["b_off"]), int(s["ldb"]), abi.OP_N, m, n, r0, c0)
                else:
                    A = bufs[int(s["buf_a"])]
                    k = int(s["k"])
                    a = self._op(A, int(s["a_off"]), int(s["lda"]), int(s["op_a"]), m, k, r0, 0)
                    b = self._op(B, int(s["b_off"]), int(s["ldb"]), int(s["op_b"]), k, n, 0, c0)
                    acc += alpha * (a @ b)
            outs.append((t, acc))
        outs.sort(key=lambda ta: int(ta[0]["part"]) if int(ta[0]["nparts"]) > 1 else 0)      # part 0 writes first, the others add
        for t, acc in outs:      # write after all reads: tiles never alias their inputs anyway
            C = bufs[int(t["buf_c"])]
            m, n, r0, c0 = int(t["m"]), int(t["n"]), int(t["row0"]), int(t["col0"])
            i = (r0 + np.arange(m))[:, None]
            j = (c0 + np.arange(n))[None, :]
            if int(t["nparts"]) > 1 and int(t["part"]) > 0:      # split-K part: the parts of one tile add up
                C[int(t["c_off"]) + i + j * int(t["ldc"])] += acc
            else:
                C[int(t["c_off"]) + i + j * int(t["ldc"])] = acc

    def dots(self, V, ldv, nvec, w, n, out):
        for i in range(nvec):
            out[i] = np.vdot(V[i * ldv:i * ldv + n], w[:n])

    def axpys(self, w, V, ldv, nvec, coef, sign, n):
        for i in range(nvec):
            w[:n] += sign * coef[i] * V[i * ldv:i * ldv + n]

    def scale_inv_sqrt(self, dst, src, nrm2, n):
        dst[:n] = src[:n] / np.sqrt(nrm2[0].real)

    def lanczos(self, stages, x_slot, y_slot, V, n, krylovdim, tol, max_restart, zero_y=False, exchange=None):
        """numpy statement of htn_lanczos_z (same algorithm: CGS2 full reorthogonalisation, eager stop)"""
        def matvec(x, y):
            if zero_y:
                y[...] = 0
            for bufs, tasks in stages:
                b = list(bufs)
                b[x_slot], b[y_slot] = x, y
                self.grouped_gemm(b, tasks)
            if exchange is not None:
                exchange(y)
        kd = krylovdim
        V[0:n] /= np.linalg.norm(V[0:n])
        nmv, theta, res, beta = 0, 0.0, 0.0, 0.0
        for restart in range(max_restart + 1):
            alphas, betas = [], []
            for j in range(kd):
                w = V[(j + 1) * n:(j + 2) * n]
                matvec(V[j * n:(j + 1) * n], w)
                nmv += 1
                Vm = V[:(j + 1) * n].reshape(j + 1, n)
                c1 = Vm.conj() @ w
                w -= Vm.T @ c1
                c2 = Vm.conj() @ w
                w -= Vm.T @ c2
                beta = float(np.linalg.norm(w))
                alphas.append(float((c1[j] + c2[j]).real))
                T = np.diag(alphas) + np.diag(betas, 1) + np.diag(betas, -1)
                ev, evec = np.linalg.eigh(T)
                theta, y = float(ev[0]), evec[:, 0]
                res = abs(beta * y[-1])
                if beta > 0:
                    w /= beta
                if res < tol or beta < 1e-14 or j == kd - 1:
                    break
                betas.append(beta)
            k = len(y)
            x = V[:k * n].reshape(k, n).T @ y.astype(np.complex128)
            V[0:n] = x / np.linalg.norm(x)
            if res < tol or beta < 1e-14:
                break
        return theta, nmv, res

    def jacobi_svd(self, G, Vj, S, desc, nblocks, max_m, max_sweeps, tol, info, desc_host=None):
        for b in range(nblocks):
            d = desc[b]
            m, n = int(d["m"]), int(d["n"])
            if int(d["flags"]) & abi.SVD_QRCP:
                # G0 (m0 x m) -> right singular vectors x Sigma (m x n), unsorted, arbitrary phases
                m0 = int(d["pad"])
                g0 = G[int(d["g_off"]):int(d["g_off"]) + m0 * m].reshape(m, m0).T
                U, s, Vh = np.linalg.svd(g0, full_matrices=False)
                perm = self.rng.permutation(n)
                ph = np.exp(2j * np.pi * self.rng.random(n))
                out = (Vh.conj().T[:, :n] * s[:n])[:, perm] * ph
                G[int(d["g_off"]):int(d["g_off"]) + m * n] = out.T.reshape(-1)
                S[int(d["s_off"]):int(d["s_off"]) + n] = s[:n][perm]
                info[b] = 1
                continue
            g = G[int(d["g_off"]):int(d["g_off"]) + m * n].reshape(n, m).T
            U, s, Vh = np.linalg.svd(g, full_matrices=True)
            r = min(m, n)
            sfull = np.zeros(n)
            sfull[:r] = s
            gp = np.zeros((m, n), dtype=np.complex128)
            gp[:, :r] = U[:, :r] * s
            perm = self.rng.permutation(n)             # Jacobi leaves the columns unsorted
            ph = np.exp(2j * np.pi * self.rng.random(n))  # and with arbitrary phases
            gp = gp[:, perm] * ph
            J = Vh.conj().T[:, perm] * ph
            G[int(d["g_off"]):int(d["g_off"]) + m * n] = gp.T.reshape(-1)
            if int(d["flags"]) & abi.SVD_ACCUMULATE:
                Vj[int(d["v_off"]):int(d["v_off"]) + n * n] = J.T.reshape(-1)
            S[int(d["s_off"]):int(d["s_off"]) + n] = sfull[perm]
            info[b] = 1

    def batched_copy(self, dst, src, idx, scl, items, nitems, gscale):
        for it in items[:nitems]:
            rows, cols = int(it["rows"]), int(it["cols"])
            i = np.arange(rows)[:, None] * np.ones((1, cols), dtype=np.int64)
            j = np.arange(cols)[None, :] * np.ones((rows, 1), dtype=np.int64)
            gi, gj = i, j
            if int(it["idx_off"]) >= 0:
                if int(it["gather_dim"]) == 0:
                    gi = idx[int(it["idx_off"]) + i]
                else:
                    gj = idx[int(it["idx_off"]) + j]
            if int(it["op"]) == abi.OP_N:
                v = src[int(it["src_off"]) + gi + gj * int(it["lds"])]
            else:
                v = src[int(it["src_off"]) + gj + gi * int(it["lds"])].conj()
            f = np.full((rows, cols), gscale)
            if int(it["scale_dim"]) >= 0 and int(it["scl_off"]) >= 0:
                sv = scl[int(it["scl_off"]) + (gi if int(it["scale_dim"]) == 0 else gj)]
                f = f / sv if int(it["inv_norm"]) else f * sv
            dst[int(it["dst_off"]) + i + j * int(it["ldd"])] = v * f
