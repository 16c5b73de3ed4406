// CPU BASELINE / CHECKER (test + benchmark infrastructure; the product never loads this).
//
// libhubbardtn_cpu.so = the SAME host core as the product (hubbardtn_amd/csrc/htn_plan.cpp, htn_engine.cpp: sector
// layouts, 9j recoupling, task lists, truncation, sweep driver) linked against THIS file instead of the HIP kernels:
// a plain C++/OpenMP execution of the task lists on host memory.  It is the CPU comparator SURVEY.md 8(d) and
// BASELINE.md section 4 specify ("the build's own CPU backend behind the same C ABI: task-parallel over
// environment-GEMM tasks, one BLAS thread per task", mirroring src/HubbardFunctions.jl:28-39), used by
//   * bench.py's cpu_baseline leg (whole sweeps, all host cores, `kind: "port"`),
//   * tests/ (the C++ planner + sweep driver run here without a GPU and are checked against the numpy oracle).
// It is NOT a fallback: libhubbardtn_hip.so refuses HTN_BACKEND_CPU, this library refuses HTN_BACKEND_HIP, and
// hubbardtn_amd/api.py only ever loads the HIP library.
//
// Per-sector SVD: LAPACK zgesvd when an OpenBLAS/LAPACKE shared object is given through $HTN_CPU_LAPACK (the reference
// path: TensorKit tsvd! -> LAPACK), else a built-in one-sided Jacobi.
#include <dlfcn.h>
#include <math.h>
#include <omp.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "htn_core.h"

using namespace htn;

namespace {

// ---- optional LAPACKE zgesvd from a shared object named by the environment ------------------------------------------
typedef int (*zgesvd32_fn)(int, char, char, int, int, void*, int, double*, void*, int, void*, int, double*);
typedef long (*zgesvd64_fn)(int, char, char, long, long, void*, long, double*, void*, long, void*, long, double*);
struct Lapack {
    zgesvd32_fn svd32 = nullptr;
    zgesvd64_fn svd64 = nullptr;
    void (*set_threads)(int) = nullptr;
};
Lapack& lapack() {
    static Lapack L;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* p = getenv("HTN_CPU_LAPACK");
        if (!p || !*p) return;
        void* h = dlopen(p, RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        L.svd32 = (zgesvd32_fn)dlsym(h, "scipy_LAPACKE_zgesvd");
        if (!L.svd32) L.svd32 = (zgesvd32_fn)dlsym(h, "LAPACKE_zgesvd");
        if (!L.svd32) L.svd64 = (zgesvd64_fn)dlsym(h, "scipy_LAPACKE_zgesvd64_");
        const char* names[] = {"scipy_openblas_set_num_threads", "scipy_openblas_set_num_threads64_", "openblas_set_num_threads"};
        for (const char* n : names)
            if (!L.set_threads) L.set_threads = (void (*)(int))dlsym(h, n);
        if (L.set_threads) L.set_threads(1);       // one BLAS thread per task (src:29); parallelism is over blocks
    });
    return L;
}

inline cplx ld_op(const cplx* p, int64_t off, int ld, int op, int64_t i, int64_t j) {      // element (i, j) of op(M)
    if (op == HTN_OP_N) return p[off + i + j * ld];
    const cplx v = p[off + j + i * ld];
    return op == HTN_OP_C ? std::conj(v) : v;
}

// one-sided (Hestenes) Jacobi on the columns of X (m x n, ld): X <- X J with mutually orthogonal columns; J (n x n,
// ldj) accumulated if given.  Returns the sweep count (negative: not converged).
int jacobi_cols(cplx* X, int m, int n, int ld, cplx* J, int ldj, double tol, int max_sweeps) {
    if (J)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) J[i + (int64_t)j * ldj] = i == j ? 1.0 : 0.0;
    const double thr = tol * tol;
    // a column below 1e-15 |X|_F is numerically zero (surplus columns of a rank-deficient block, e.g. a centre that is moved
    // without being optimised): its direction is rounding noise and must not keep the sweep loop alive (same rule as the
    // HIP kernels, htn_svd.hip)
    double frob2 = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) frob2 += std::norm(X[i + (int64_t)j * ld]);
    const double zero2 = 1e-30 * frob2;
    for (int sweep = 1; sweep <= max_sweeps; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                cplx* xp = X + (int64_t)p * ld;
                cplx* xq = X + (int64_t)q * ld;
                double a = 0.0, b = 0.0;
                cplx g = 0.0;
                for (int i = 0; i < m; ++i) {
                    a += std::norm(xp[i]);
                    b += std::norm(xq[i]);
                    g += std::conj(xp[i]) * xq[i];
                }
                const double g2 = std::norm(g);
                if (a <= zero2 || b <= zero2 || g2 <= thr * a * b || g2 == 0.0) continue;
                rotated = true;
                const double ag = sqrt(g2);
                const cplx ph = g / ag;                          // xq <- conj(ph) xq makes the inner product real
                const double zeta = (b - a) / (2.0 * ag);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < m; ++i) {
                    const cplx u = xp[i], v = std::conj(ph) * xq[i];
                    xp[i] = c * u - s * v;
                    xq[i] = s * u + c * v;
                }
                if (J) {
                    cplx* jp = J + (int64_t)p * ldj;
                    cplx* jq = J + (int64_t)q * ldj;
                    for (int i = 0; i < n; ++i) {
                        const cplx u = jp[i], v = std::conj(ph) * jq[i];
                        jp[i] = c * u - s * v;
                        jq[i] = s * u + c * v;
                    }
                }
            }
        if (!rotated) return sweep;
    }
    return -max_sweeps;
}

// lowest eigenpair of the real symmetric tridiagonal (alpha, beta): bisection on the Sturm count + inverse iteration
void tridiag_lowest(const std::vector<double>& alpha, const std::vector<double>& beta, double* val, std::vector<double>& vec) {
    const int k = (int)alpha.size();
    vec.assign(k, 0.0);
    if (k == 1) {
        *val = alpha[0];
        vec[0] = 1.0;
        return;
    }
    double lo = 1e300, hi = -1e300;
    for (int i = 0; i < k; ++i) {
        const double r = (i > 0 ? fabs(beta[i - 1]) : 0.0) + (i + 1 < k ? fabs(beta[i]) : 0.0);
        lo = std::min(lo, alpha[i] - r);
        hi = std::max(hi, alpha[i] + r);
    }
    auto count_below = [&](double x) {
        int cnt = 0;
        double d = 1.0;
        for (int i = 0; i < k; ++i) {
            d = (alpha[i] - x) - (i > 0 ? beta[i - 1] * beta[i - 1] / d : 0.0);
            if (d == 0.0) d = 1e-300;
            if (d < 0.0) ++cnt;
        }
        return cnt;
    };
    for (int it = 0; it < 200 && hi - lo > 4e-16 * std::max(fabs(lo), fabs(hi)); ++it) {
        const double mid = 0.5 * (lo + hi);
        if (count_below(mid) >= 1) hi = mid;
        else lo = mid;
    }
    const double mu = 0.5 * (lo + hi);
    *val = mu;
    std::vector<double> d(k), l(k), z(k);
    const double shift = mu - 1e-13 * std::max(1.0, fabs(mu));
    d[0] = alpha[0] - shift;
    if (d[0] == 0.0) d[0] = 1e-300;
    for (int i = 1; i < k; ++i) {
        l[i] = beta[i - 1] / d[i - 1];
        d[i] = (alpha[i] - shift) - l[i] * beta[i - 1];
        if (d[i] == 0.0) d[i] = 1e-300;
    }
    for (int i = 0; i < k; ++i) vec[i] = (i & 1) ? -0.7 : 1.0;
    for (int iter = 0; iter < 4; ++iter) {
        z[0] = vec[0];
        for (int i = 1; i < k; ++i) z[i] = vec[i] - l[i] * z[i - 1];
        z[k - 1] /= d[k - 1];
        for (int i = k - 2; i >= 0; --i) z[i] = z[i] / d[i] - l[i + 1] * z[i + 1];
        double nn = 0.0;
        for (int i = 0; i < k; ++i) nn += z[i] * z[i];
        nn = 1.0 / sqrt(nn);
        for (int i = 0; i < k; ++i) vec[i] = z[i] * nn;
    }
    if (vec[0] < 0.0)
        for (int i = 0; i < k; ++i) vec[i] = -vec[i];
    // Rayleigh quotient: the eigenvalue to full accuracy
    double num = 0.0;
    for (int i = 0; i < k; ++i) {
        double t = alpha[i] * vec[i];
        if (i > 0) t += beta[i - 1] * vec[i - 1];
        if (i + 1 < k) t += beta[i] * vec[i + 1];
        num += vec[i] * t;
    }
    *val = num;
}

struct CpuBackend : Backend {
    int kind() const override { return HTN_BACKEND_CPU; }
    void* alloc(size_t bytes) override {
        void* p = nullptr;
        if (posix_memalign(&p, 64, std::max<size_t>(bytes, 64))) return nullptr;
        // HTN_DEBUG_POISON (test switch, same meaning as in the HIP backend): blocks start as 0xFF bytes (NaN), so a planner
        // that leaves part of a buffer unwritten shows up as NaN instead of depending on what malloc returned
        static const bool poison = [] {
            const char* v = getenv("HTN_DEBUG_POISON");
            return v && v[0] && v[0] != '0';
        }();
        if (poison) memset(p, 0xFF, std::max<size_t>(bytes, 64));
        return p;
    }
    void release(void* p) override { free(p); }
    int upload(void* dst, const void* src, size_t bytes) override {
        memcpy(dst, src, bytes);
        return 0;
    }
    int download(void* dst, const void* src, size_t bytes) override {
        memcpy(dst, src, bytes);
        return 0;
    }
    int zero(void* p, size_t bytes) override {
        memset(p, 0, bytes);
        return 0;
    }
    int sync() override { return 0; }

    // task-parallel over output tiles (OpenMP, dynamic: tiles arrive longest first), serial arithmetic inside a task
    int grouped_gemm(const void* const* bufs, const htn_tile* tiles, int32_t n_tiles, const htn_seg* segs) override {
#pragma omp parallel for schedule(dynamic, 1)
        for (int t = 0; t < n_tiles; ++t) {
            const htn_tile& T = tiles[t];
            cplx acc[HTN_TILE * HTN_TILE];
            cplx a[HTN_TILE * 16], b[16 * HTN_TILE];
            for (int e = 0; e < T.m * T.n; ++e) acc[e] = 0.0;
            for (int s = T.seg_begin; s < T.seg_begin + T.seg_count; ++s) {
                const htn_seg& S = segs[s];
                const cplx alpha(S.alpha_re, S.alpha_im);
                const cplx* Bp = (const cplx*)bufs[S.buf_b];
                if (S.type == HTN_SEG_COPY) {
                    for (int j = 0; j < T.n; ++j)
                        for (int i = 0; i < T.m; ++i)
                            acc[i + j * T.m] += alpha * Bp[S.b_off + (T.row0 + i) + (int64_t)(T.col0 + j) * S.ldb];
                    continue;
                }
                const cplx* Ap = (const cplx*)bufs[S.buf_a];
                for (int k0 = 0; k0 < S.k; k0 += 16) {
                    const int kk = std::min(16, S.k - k0);
                    for (int k = 0; k < kk; ++k)
                        for (int i = 0; i < T.m; ++i) a[i + k * T.m] = alpha * ld_op(Ap, S.a_off, S.lda, S.op_a, T.row0 + i, k0 + k);
                    for (int j = 0; j < T.n; ++j)
                        for (int k = 0; k < kk; ++k) b[k + j * 16] = ld_op(Bp, S.b_off, S.ldb, S.op_b, k0 + k, T.col0 + j);
                    for (int j = 0; j < T.n; ++j)
                        for (int k = 0; k < kk; ++k) {
                            const cplx bv = b[k + j * 16];
                            const cplx* ac = a + k * T.m;
                            cplx* cc = acc + j * T.m;
                            for (int i = 0; i < T.m; ++i) cc[i] += ac[i] * bv;
                        }
                }
            }
            cplx* Cp = (cplx*)bufs[T.buf_c] + T.c_off;
            for (int j = 0; j < T.n; ++j)
                for (int i = 0; i < T.m; ++i) Cp[(T.row0 + i) + (int64_t)(T.col0 + j) * T.ldc] = acc[i + j * T.m];
        }
        return 0;
    }

    // fixed chunking + ordered final sum: bit-reproducible whatever the thread count (replicated ranks of the sharded
    // apply must take identical truncation decisions)
    static double nrm2(const cplx* x, int64_t n) {
        const int64_t chunk = 4096, nch = (n + chunk - 1) / chunk;
        std::vector<double> part((size_t)nch, 0.0);
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < nch; ++c) {
            double s = 0.0;
            const int64_t hi = std::min(n, (c + 1) * chunk);
            for (int64_t i = c * chunk; i < hi; ++i) s += std::norm(x[i]);
            part[(size_t)c] = s;
        }
        double s = 0.0;
        for (double v : part) s += v;
        return s;
    }

    // KrylovKit-style Lanczos: full basis, two-pass classical Gram-Schmidt, eager stop, restart
    int lanczos(const htn_gemm_launch* stages, int n_stages, int x_slot, int y_slot, void* Vv, int64_t n, int kd, double tol,
                int max_restart, int zero_y, htn_exchange2_fn exchange, void* user, double* eig, int* n_matvec,
                double* residual, double* matvec_ms) override {
        cplx* V = (cplx*)Vv;
        if (kd < 2 || kd > 63) return set_error("lanczos: krylovdim must be in 2..63");
        auto matvec = [&](cplx* x, cplx* y) -> int {
            if (zero_y) memset((void*)y, 0, sizeof(cplx) * n);
            for (int s = 0; s < n_stages; ++s) {
                const void* bufs[HTN_MAX_BUFS];
                for (int b = 0; b < HTN_MAX_BUFS; ++b) bufs[b] = stages[s].bufs[b];
                bufs[x_slot] = x;
                bufs[y_slot] = y;
                if (grouped_gemm(bufs, stages[s].tiles, stages[s].n_tiles, stages[s].segs)) return 1;
            }
            if (exchange && exchange(y, n, user)) return set_error("lanczos: the exchange hook reported a failure");
            return 0;
        };
        {
            const double s = 1.0 / sqrt(nrm2(V, n));
            for (int64_t i = 0; i < n; ++i) V[i] *= s;
        }
        int nmv = 0;
        double theta = 0.0, res = 0.0, beta = 0.0, amax = 0.0;
        std::vector<double> y;
        std::vector<cplx> c(kd + 1);
        for (int restart = 0; restart <= max_restart; ++restart) {
            std::vector<double> alphas, betas;
            for (int j = 0; j < kd; ++j) {
                cplx* vj = V + (int64_t)j * n;
                cplx* w = V + (int64_t)(j + 1) * n;
                if (matvec(vj, w)) return 1;
                ++nmv;
                double alpha = 0.0;
                for (int pass = 0; pass < 2; ++pass) {
#pragma omp parallel for
                    for (int i = 0; i <= j; ++i) {
                        const cplx* vi = V + (int64_t)i * n;
                        cplx s = 0.0;
                        for (int64_t e = 0; e < n; ++e) s += std::conj(vi[e]) * w[e];
                        c[i] = s;
                    }
                    alpha += c[j].real();
#pragma omp parallel for
                    for (int64_t e = 0; e < n; ++e) {
                        cplx s = 0.0;
                        for (int i = 0; i <= j; ++i) s += c[i] * V[(int64_t)i * n + e];
                        w[e] -= s;
                    }
                }
                beta = sqrt(nrm2(w, n));
                alphas.push_back(alpha);
                tridiag_lowest(alphas, betas, &theta, y);
                res = fabs(beta * y.back());
                amax = std::max(amax, std::max(fabs(alpha), beta));
                if (res < tol || beta < 1e-14 * std::max(amax, 1e-300) || j == kd - 1) break;
                betas.push_back(beta);
                const double inv = 1.0 / beta;
                for (int64_t e = 0; e < n; ++e) w[e] *= inv;
            }
            const int k = (int)y.size();
            cplx* xrow = V + (int64_t)(kd + 1) * n;
#pragma omp parallel for
            for (int64_t e = 0; e < n; ++e) {
                cplx s = 0.0;
                for (int i = 0; i < k; ++i) s += y[i] * V[(int64_t)i * n + e];
                xrow[e] = s;
            }
            const double s = 1.0 / sqrt(nrm2(xrow, n));
            for (int64_t e = 0; e < n; ++e) V[e] = xrow[e] * s;
            if (res < tol || beta < 1e-14 * std::max(amax, 1e-300)) break;
        }
        *eig = theta;
        *n_matvec = nmv;
        *residual = res;
        if (matvec_ms) *matvec_ms = 0.0;
        return 0;
    }

    // contract of htn_jacobi_svd_z (include/hubbardtn_hip.h) executed block by block on the host
    int jacobi_svd(void* Gv, void* Vjv, double* S, const htn_svd_block* desc, const htn_svd_block*, int n_blocks, int,
                   int max_sweeps, double tol, int32_t* info, const htn_svd_opts*) override {
        cplx* G = (cplx*)Gv;
        cplx* Vj = (cplx*)Vjv;
        Lapack& LP = lapack();
        int failed = 0;
#pragma omp parallel for schedule(dynamic, 1)
        for (int bi = 0; bi < n_blocks; ++bi) {
            const htn_svd_block& D = desc[bi];
            if (D.flags & HTN_SVD_QRCP) {
                // G0 (pad x m, ld pad) at g_off; wanted: A = G0^H = U Sigma W^H  ->  U Sigma (m x n, ld m), n = min(pad, m)
                const int pad = D.pad, m = D.m, n = D.n;
                std::vector<cplx> A((size_t)m * pad);
                for (int j = 0; j < pad; ++j)
                    for (int i = 0; i < m; ++i) A[i + (size_t)j * m] = std::conj(G[D.g_off + j + (int64_t)i * pad]);
                cplx* out = G + D.g_off;
                double* s = S + D.s_off;
                if (LP.svd32 || LP.svd64) {
                    std::vector<cplx> U((size_t)m * n);
                    std::vector<double> superb(std::max(1, std::min(m, pad)));
                    long rc;
                    if (LP.svd32) rc = LP.svd32(102, 'S', 'N', m, pad, A.data(), m, s, U.data(), m, nullptr, 1, superb.data());
                    else rc = LP.svd64(102, 'S', 'N', m, pad, A.data(), m, s, U.data(), m, nullptr, 1, superb.data());
                    if (rc != 0) {
#pragma omp atomic write
                        failed = 1;
                    }
                    for (int j = 0; j < n; ++j)
                        for (int i = 0; i < m; ++i) out[i + (int64_t)j * m] = U[i + (size_t)j * m] * s[j];
                    info[bi] = 1;
                    continue;
                }
                int sw;
                if (pad <= m) {                       // Jacobi on the pad = n columns of A
                    sw = jacobi_cols(A.data(), m, pad, m, nullptr, 0, tol, max_sweeps);
                    for (int j = 0; j < n; ++j) {
                        double nn = 0.0;
                        for (int i = 0; i < m; ++i) nn += std::norm(A[i + (size_t)j * m]);
                        s[j] = sqrt(nn);
                        for (int i = 0; i < m; ++i) out[i + (int64_t)j * m] = A[i + (size_t)j * m];
                    }
                } else {                              // wide A: orthogonalise the m columns of G0 = A^H, U = the rotation
                    std::vector<cplx> X((size_t)pad * m), J((size_t)m * m);
                    for (int j = 0; j < m; ++j)
                        for (int i = 0; i < pad; ++i) X[i + (size_t)j * pad] = G[D.g_off + i + (int64_t)j * pad];
                    sw = jacobi_cols(X.data(), pad, m, pad, J.data(), m, tol, max_sweeps);
                    for (int j = 0; j < m; ++j) {
                        double nn = 0.0;
                        for (int i = 0; i < pad; ++i) nn += std::norm(X[i + (size_t)j * pad]);
                        s[j] = sqrt(nn);
                        for (int i = 0; i < m; ++i) out[i + (int64_t)j * m] = J[i + (size_t)j * m] * s[j];
                    }
                }
                info[bi] = sw;
            } else {
                const int m = D.m, n = D.n;
                cplx* X = G + D.g_off;
                cplx* J = (D.flags & HTN_SVD_ACCUMULATE) ? Vj + D.v_off : nullptr;
                const int sw = jacobi_cols(X, m, n, m, J, n, tol, max_sweeps);
                for (int j = 0; j < n; ++j) {
                    double nn = 0.0;
                    for (int i = 0; i < m; ++i) nn += std::norm(X[i + (int64_t)j * m]);
                    S[D.s_off + j] = sqrt(nn);
                }
                info[bi] = sw;
            }
        }
        if (failed) return set_error("LAPACK zgesvd failed");
        return 0;
    }

    int batched_copy(void* dstv, const void* srcv, const int32_t* idx, const double* scl, const htn_copy_item* items, int n_items,
                     double gscale) override {
        cplx* dst = (cplx*)dstv;
        const cplx* src = (const cplx*)srcv;
#pragma omp parallel for schedule(dynamic, 1)
        for (int q = 0; q < n_items; ++q) {
            const htn_copy_item& I = items[q];
            for (int j = 0; j < I.cols; ++j)
                for (int i = 0; i < I.rows; ++i) {
                    int gi = i, gj = j;
                    if (I.idx_off >= 0) {
                        if (I.gather_dim == 0) gi = idx[I.idx_off + i];
                        else gj = idx[I.idx_off + j];
                    }
                    cplx v;
                    if (I.op == HTN_OP_N) v = src[I.src_off + gi + (int64_t)gj * I.lds];
                    else v = std::conj(src[I.src_off + gj + (int64_t)gi * I.lds]);
                    double f = gscale;
                    if (I.scale_dim >= 0 && I.scl_off >= 0) {
                        const double sv = scl[I.scl_off + (I.scale_dim == 0 ? gi : gj)];
                        f = I.inv_norm ? (sv > 0.0 ? f / sv : 0.0) : f * sv;
                    }
                    dst[I.dst_off + i + (int64_t)j * I.ldd] = v * f;
                }
        }
        return 0;
    }
    int scale(void* x, int64_t n, double f) override {
        cplx* p = (cplx*)x;
        for (int64_t i = 0; i < n; ++i) p[i] *= f;
        return 0;
    }
};

}  // namespace

namespace htn {
Backend* make_backend(int backend, int, void*) {
    if (backend != HTN_BACKEND_CPU) {
        set_error("libhubbardtn_cpu.so is the CPU baseline: backend %d is not available here", backend);
        return nullptr;
    }
    return new CpuBackend();
}
}  // namespace htn

// CPU-baseline extra (not part of the product header): OpenMP team size of the task-parallel kernels
extern "C" int htn_cpu_set_threads(int n) {
    const int prev = omp_get_max_threads();
    if (n > 0) omp_set_num_threads(n);
    return prev;
}
extern "C" int htn_comm_unique_id(void*) { return set_error("htn_comm_unique_id: the CPU baseline has no RCCL"); }
extern "C" int htn_device_init(int, char* name_host, int* cu_count_host) {
    if (name_host) strcpy(name_host, "cpu");
    if (cu_count_host) *cu_count_host = 0;
    return 0;
}
