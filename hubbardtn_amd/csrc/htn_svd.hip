// batched Jacobi SVD + strided copy kernels (libhubbardtn_hip.so)
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "htn_common.h"

// ----------------------------------------------------------------------------------------------
// batched one-sided (Hestenes) Jacobi SVD, one 16-wave workgroup per coupled-sector block
// ----------------------------------------------------------------------------------------------
// Round-robin (circle) ordering: n-1 rounds of n/2 disjoint column pairs per sweep, rounds separated by
// a workgroup barrier.  The critical path is (#sweeps) x (n-1) rounds x (latency of one round), so the
// kernel is built to make a round short:
//   * a pair is rotated by a GROUP of GS = 16 / 32 / 64 lanes chosen from the column length
//     (m <= 128 / 256 / 512): 1024 / GS pairs rotate concurrently, i.e. one pass per round up to n = 128,
//     and both columns stay in registers between the dot products and the rotation;
//   * the matrix lives in LDS when it fits the dynamic LDS window (round latency ~ LDS instead of L2),
//     otherwise in global memory (L2 resident: a few hundred KB);
//   * sweeps stop early through the quadratic convergence of cyclic Jacobi: a sweep whose largest squared
//     cosine, seen BEFORE the rotations, is below 0.1 tol leaves less than tol^2 behind, so neither the
//     textbook's final "checking" sweep nor the sweep before it is run.
#define JAC_THREADS 1024
#define JAC_MAXEL 8           // column elements per lane kept in registers: m <= GS * JAC_MAXEL

template <int GS>
__device__ __forceinline__ double group_sum(double v) {      // all-reduce inside aligned groups of GS lanes
    v = row_sum16(v);
    if (GS == 16) return v;
    // row sums as wave-uniform scalars (v_readlane) instead of ds_bpermute butterflies; same summation order
    const double a = lane_bcast<0>(v) + lane_bcast<16>(v);
    const double b = lane_bcast<32>(v) + lane_bcast<48>(v);
    if (GS == 32) return (threadIdx.x & 32) ? b : a;
    return a + b;
}

template <int GS>
__device__ __forceinline__ double jacobi_pair(double2* ga, double2* gb, double2* __restrict__ va,
                                              double2* __restrict__ vb, int m, int n, int sub, double tol,
                                              bool accumulate, double zero2) {
    double2 a[JAC_MAXEL], b[JAC_MAXEL];
    double aa = 0.0, bb = 0.0, gr = 0.0, gi = 0.0;
#pragma unroll
    for (int e = 0; e < JAC_MAXEL; ++e) {
        const int i = sub + GS * e;
        if (i < m) {
            a[e] = ga[i];
            b[e] = gb[i];
            aa += a[e].x * a[e].x + a[e].y * a[e].y;
            bb += b[e].x * b[e].x + b[e].y * b[e].y;
            gr += a[e].x * b[e].x + a[e].y * b[e].y;      // conj(a) * b
            gi += a[e].x * b[e].y - a[e].y * b[e].x;
        }
    }
    aa = group_sum<GS>(aa);
    bb = group_sum<GS>(bb);
    gr = group_sum<GS>(gr);
    gi = group_sum<GS>(gi);
    // a column below 1e-15 |G|_F is numerically zero (surplus columns of a wide or rank-deficient block):
    // its direction is rounding noise and must not keep the sweep loop alive
    if (aa <= zero2 || bb <= zero2) return 0.0;
    const double g2 = gr * gr + gi * gi;
    const double ab = aa * bb;
    const double ratio2 = g2 * fast_rcp(ab);          // (|a.b| / (|a||b|))^2
    if (g2 == 0.0 || g2 <= tol * tol * ab) return ratio2;
    // rotation diagonalising [[aa, gamma], [conj gamma, bb]]:  t = sign(h) 2g / (|h| + sqrt(h^2 + 4 g^2)), h = bb - aa
    const double ig = fast_rsq(g2);
    const double g = g2 * ig;
    const double h = bb - aa;
    const double w2 = fma(h, h, 4.0 * g2);
    const double w = w2 * fast_rsq(w2);
    double t = 2.0 * g * fast_rcp(fabs(h) + w);
    t = h >= 0.0 ? t : -t;
    const double c = fast_rsq(fma(t, t, 1.0));
    const double s = c * t;
    // b~ = exp(-i phi) b with exp(i phi) = gamma / |gamma|;  a' = c a - s b~ ; b' = s a + c b~
    const double pr = gr * ig, pi = -gi * ig;   // exp(-i phi)
#pragma unroll
    for (int e = 0; e < JAC_MAXEL; ++e) {
        const int i = sub + GS * e;
        if (i < m) {
            const double2 bt = make_double2(pr * b[e].x - pi * b[e].y, pr * b[e].y + pi * b[e].x);
            ga[i] = make_double2(c * a[e].x - s * bt.x, c * a[e].y - s * bt.y);
            gb[i] = make_double2(s * a[e].x + c * bt.x, s * a[e].y + c * bt.y);
        }
    }
    if (accumulate)
        for (int i = sub; i < n; i += GS) {
            const double2 x = va[i], y = vb[i];
            const double2 yt = make_double2(pr * y.x - pi * y.y, pr * y.y + pi * y.x);
            va[i] = make_double2(c * x.x - s * yt.x, c * x.y - s * yt.y);
            vb[i] = make_double2(s * x.x + c * yt.x, s * x.y + c * yt.y);
        }
    return ratio2;
}

template <int GS>
__device__ __forceinline__ int jacobi_sweeps(double2* g, double2* __restrict__ v, int m, int n, int max_sweeps,
                                             double tol, double* s_ratio, int tid, bool accumulate, double zero2) {
    const int grp = tid / GS, sub = tid % GS;
    const int ngroups = JAC_THREADS / GS;
    const int np = n + (n & 1);      // padded to even; index np-1 == n is a bye when n is odd
    int sweeps = 0;
    bool done = (n < 2);
    while (!done && sweeps < max_sweeps) {
        if (tid == 0) *s_ratio = 0.0;
        __syncthreads();
        double ratio = 0.0;
        for (int r = 0; r < np - 1; ++r) {
            for (int p = grp; p < np / 2; p += ngroups) {
                int i, j;
                if (p == 0) {
                    i = np - 1;
                    j = r;
                } else {
                    i = (r + p) % (np - 1);
                    j = (r + np - 1 - p) % (np - 1);
                }
                if (i < n && j < n) {
                    const int lo = i < j ? i : j, hi = i < j ? j : i;
                    const double rr = jacobi_pair<GS>(g + (int64_t)lo * m, g + (int64_t)hi * m, v + (int64_t)lo * n,
                                                      v + (int64_t)hi * n, m, n, sub, tol, accumulate, zero2);
                    ratio = rr > ratio ? rr : ratio;
                }
            }
            __syncthreads();
        }
        // workgroup max of the non-negative ratio: doubles order like their bit patterns
        if (sub == 0 && ratio > 0.0) atomicMax((unsigned long long*)s_ratio, (unsigned long long)__double_as_longlong(ratio));
        __syncthreads();
        const double mx = *s_ratio;  // max over the sweep of the SQUARED cosine between column pairs
        ++sweeps;
        // mx is what the sweep SAW before its rotations; by the quadratic convergence of cyclic Jacobi the sweep
        // leaves about C mx^2 behind (C = 0.005 .. 0.2 measured on Schmidt-type spectra).  A sweep that saw
        // no squared cosine above 0.1 tol has therefore left < tol^2: no checking sweep afterwards.
        done = mx <= fmax(tol * tol, 0.1 * tol);
        __syncthreads();
    }
    return done ? sweeps : -sweeps;
}

// column storage split between LDS (first nl columns) and global memory (the rest): blocks slightly larger
// than the LDS window keep most of their columns at LDS latency.  Generic (flat) pointers.
struct SplitCols {
    double2* lds;
    double2* glob;
    int nl, mp;
    __device__ __forceinline__ double2* col(int j) const {
        return j < nl ? lds + (int64_t)j * mp : glob + (int64_t)(j - nl) * mp;
    }
};

// ---- padded fast path (QR-preconditioned blocks) -----------------------------------------------------
// Columns are stored with leading dimension GS * E (zero padded), so every lane owns exactly E elements
// of each column: no per-element predication, all loads issued back to back, and the LDS / global address
// space is known at compile time (ds_* / global_* instead of flat_*).
template <int GS, int E, typename P>
__device__ __forceinline__ double jacobi_pair_pad(P ga, P gb, int sub, double tol2, double zero2) {
    double2 a[E], b[E];
    double aa = 0.0, bb = 0.0, gr = 0.0, gi = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        a[e] = ga[sub + GS * e];
        b[e] = gb[sub + GS * e];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        aa = fma(a[e].x, a[e].x, fma(a[e].y, a[e].y, aa));
        bb = fma(b[e].x, b[e].x, fma(b[e].y, b[e].y, bb));
        gr = fma(a[e].x, b[e].x, fma(a[e].y, b[e].y, gr));      // conj(a) * b
        gi = fma(a[e].x, b[e].y, fma(-a[e].y, b[e].x, gi));
    }
    aa = group_sum<GS>(aa);
    bb = group_sum<GS>(bb);
    gr = group_sum<GS>(gr);
    gi = group_sum<GS>(gi);
    if (aa <= zero2 || bb <= zero2) return 0.0;
    const double g2 = gr * gr + gi * gi;
    const double ab = aa * bb;
    const double ratio2 = g2 * fast_rcp(ab);
    if (g2 == 0.0 || g2 <= tol2 * ab) return ratio2;
    const double ig = fast_rsq(g2);
    const double g = g2 * ig;
    const double h = bb - aa;
    const double w2 = fma(h, h, 4.0 * g2);
    const double w = w2 * fast_rsq(w2);
    double t = 2.0 * g * fast_rcp(fabs(h) + w);
    t = h >= 0.0 ? t : -t;
    const double c = fast_rsq(fma(t, t, 1.0));
    const double s = c * t;
    // a' = c a - sig b ; b' = conj(sig)... written with sig = s exp(-i phi), tau = c exp(-i phi):
    //   a' = c a - sig b,  b' = s a + tau b
    const double sr = s * gr * ig, si = -s * gi * ig;
    const double tr = c * gr * ig, ti = -c * gi * ig;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double2 na, nb;
        na.x = fma(c, a[e].x, fma(-sr, b[e].x, si * b[e].y));
        na.y = fma(c, a[e].y, fma(-sr, b[e].y, -si * b[e].x));
        nb.x = fma(s, a[e].x, fma(tr, b[e].x, -ti * b[e].y));
        nb.y = fma(s, a[e].y, fma(tr, b[e].y, ti * b[e].x));
        ga[sub + GS * e] = na;
        gb[sub + GS * e] = nb;
    }
    return ratio2;
}

template <int GS, int E, typename C>
__device__ __forceinline__ int jacobi_sweeps_pad(C cols, int n, int max_sweeps, double tol, double* s_ratio, int tid,
                                                 double zero2) {
    const int grp = tid / GS, sub = tid % GS;
    constexpr int ngroups = JAC_THREADS / GS;
    const int np = n + (n & 1);      // padded to even; index np-1 == n is a bye when n is odd
    const int md = np - 1;
    const double tol2 = tol * tol;
    int sweeps = 0;
    bool done = (n < 2);
    while (!done && sweeps < max_sweeps) {
        if (tid == 0) *s_ratio = 0.0;
        __syncthreads();
        double ratio = 0.0;
        for (int r = 0; r < md; ++r) {
            for (int p = grp; p < np / 2; p += ngroups) {
                int i, j;
                if (p == 0) {
                    i = np - 1;
                    j = r;
                } else {
                    i = r + p;
                    i = i >= md ? i - md : i;
                    j = r + md - p;
                    j = j >= md ? j - md : j;
                }
                if (i < n && j < n) {
                    const int lo = i < j ? i : j, hi = i < j ? j : i;
                    const double rr = jacobi_pair_pad<GS, E>(cols.col(lo), cols.col(hi), sub, tol2, zero2);
                    ratio = rr > ratio ? rr : ratio;
                }
            }
            __syncthreads();
        }
        if (sub == 0 && ratio > 0.0) atomicMax((unsigned long long*)s_ratio, (unsigned long long)__double_as_longlong(ratio));
        __syncthreads();
        const double mx = *s_ratio;  // max over the sweep of the SQUARED cosine between column pairs
        ++sweeps;
        done = mx <= fmax(tol2, 0.1 * tol);                // quadratic convergence, see jacobi_sweeps
        __syncthreads();
    }
    return done ? sweeps : -sweeps;
}

// all columns in one array of known address space (LDS when P is an LDS pointer after inlining)
template <typename P>
struct DenseCols {
    P base;
    int mp;
    __device__ __forceinline__ P col(int j) const { return base + j * mp; }
};

template <int GS, typename C>
__device__ __forceinline__ int jacobi_dispatch_e(C g, int E, int n, int max_sweeps, double tol, double* s_ratio, int tid,
                                                 double zero2) {
    switch (E) {
        case 1: return jacobi_sweeps_pad<GS, 1>(g, n, max_sweeps, tol, s_ratio, tid, zero2);
        case 2: return jacobi_sweeps_pad<GS, 2>(g, n, max_sweeps, tol, s_ratio, tid, zero2);
        case 3: return jacobi_sweeps_pad<GS, 3>(g, n, max_sweeps, tol, s_ratio, tid, zero2);
        case 4: return jacobi_sweeps_pad<GS, 4>(g, n, max_sweeps, tol, s_ratio, tid, zero2);
        case 5: return jacobi_sweeps_pad<GS, 5>(g, n, max_sweeps, tol, s_ratio, tid, zero2);
        case 6: return jacobi_sweeps_pad<GS, 6>(g, n, max_sweeps, tol, s_ratio, tid, zero2);
        case 7: return jacobi_sweeps_pad<GS, 7>(g, n, max_sweeps, tol, s_ratio, tid, zero2);
        default: return jacobi_sweeps_pad<GS, 8>(g, n, max_sweeps, tol, s_ratio, tid, zero2);
    }
}

// ---- QR with column pivoting (modified Gram-Schmidt, R only) ---------------------------------------
// Preconditioner of Drmac-Veselic type: G0 P = Q R, then Jacobi runs on X = R^H whose columns are graded
// by the pivoting, which cuts the sweep count from ~18 to a few on Schmidt-type (graded) spectra.  Q is
// never needed: G0 = (Q J) Sigma (P W)^H, and P W -- the normalised Jacobi output with its rows permuted
// back -- is exactly the isometry the caller asked for (it staged G0 = M^H or M accordingly).
// Columns are not swapped physically: s_col[k] is the physical column of logical position k.
template <int GS>
__device__ __forceinline__ int qrcp_mgs(double2* __restrict__ g0, int m0, int n0, int r, const SplitCols X,
                                        int* s_col, double* s_cn2, double* s_piv, int tid, double cut2) {
    const int grp = tid / GS, sub = tid % GS, lane = tid & 63, wave = tid >> 6;
    const int ngroups = JAC_THREADS / GS;
    for (int idx = tid; idx < X.mp * r; idx += JAC_THREADS) X.col(idx / X.mp)[idx % X.mp] = make_double2(0.0, 0.0);
    for (int k = grp; k < n0; k += ngroups) {
        double c = 0.0;
        for (int i = sub; i < m0; i += GS) {
            const double2 x = g0[(int64_t)k * m0 + i];
            c += x.x * x.x + x.y * x.y;
        }
        c = group_sum<GS>(c);
        if (sub == 0) {
            s_cn2[k] = c;
            s_col[k] = k;
        }
    }
    __syncthreads();
    if (wave == 0) {
        double f = 0.0;
        for (int k = lane; k < n0; k += 64) f += s_cn2[k];
        f = wave_sum(f);
        if (lane == 0) s_piv[1] = 1e-30 * f;       // "numerically zero" threshold on squared norms
    }
    __syncthreads();
    const double zero2 = s_piv[1];
    int rank = 0;
    for (int j = 0; j < r; ++j) {
        if (wave == 0) {                            // pivot = remaining column of largest norm
            double best = -1.0, mass = 0.0;
            int bi = j;
            for (int k = j + lane; k < n0; k += 64) {
                mass += s_cn2[k];                   // |R22|_F^2: bounds every remaining singular value (rank cut)
                if (s_cn2[k] > best) {
                    best = s_cn2[k];
                    bi = k;
                }
            }
            mass = wave_sum(mass);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ob > best || (ob == best && oi < bi)) {
                    best = ob;
                    bi = oi;
                }
            }
            if (bi != j)                             // the finished part of R moves with its column
                for (int jj = lane; jj < j; jj += 64) {
                    double2* xc = X.col(jj);
                    const double2 t = xc[j];
                    xc[j] = xc[bi];
                    xc[bi] = t;
                }
            if (lane == 0) {
                const int t = s_col[j];
                s_col[j] = s_col[bi];
                s_col[bi] = t;
                s_cn2[bi] = s_cn2[j];
                s_cn2[j] = best;
                s_piv[0] = mass <= cut2 ? 0.0 : best;
            }
        }
        __syncthreads();
        if (s_piv[0] <= zero2) break;               // numerically rank deficient: remaining R rows are zero
        rank = j + 1;
        const double2* __restrict__ p = g0 + (int64_t)s_col[j] * m0;
        for (int k = j + grp; k < n0; k += ngroups) {
            // every group recomputes |p|^2 itself (bitwise identical everywhere): no normalisation pass
            double2 pv[JAC_MAXEL], av[JAC_MAXEL];
            double pp = 0.0, dr = 0.0, di = 0.0;
            double2* __restrict__ a = g0 + (int64_t)s_col[k] * m0;
#pragma unroll
            for (int e = 0; e < JAC_MAXEL; ++e) {
                const int i = sub + GS * e;
                if (i < m0) {
                    pv[e] = p[i];
                    pp += pv[e].x * pv[e].x + pv[e].y * pv[e].y;
                    if (k > j) {
                        av[e] = a[i];
                        dr += pv[e].x * av[e].x + pv[e].y * av[e].y;      // conj(p) * a
                        di += pv[e].x * av[e].y - pv[e].y * av[e].x;
                    }
                }
            }
            pp = group_sum<GS>(pp);
            const double pn = sqrt(pp);
            if (k == j) {
                if (sub == 0) X.col(j)[j] = make_double2(pn, 0.0);
                continue;
            }
            dr = group_sum<GS>(dr);
            di = group_sum<GS>(di);
            const double fr = dr / pp, fi = di / pp;      // (p^H a) / |p|^2
            double c = 0.0;
#pragma unroll
            for (int e = 0; e < JAC_MAXEL; ++e) {
                const int i = sub + GS * e;
                if (i < m0) {
                    const double2 x = make_double2(av[e].x - (fr * pv[e].x - fi * pv[e].y),
                                                   av[e].y - (fr * pv[e].y + fi * pv[e].x));
                    a[i] = x;
                    c += x.x * x.x + x.y * x.y;
                }
            }
            c = group_sum<GS>(c);
            if (sub == 0) {
                s_cn2[k] = c;
                X.col(j)[k] = make_double2(dr / pn, -di / pn);   // conj(r_jk), r_jk = q^H a
            }
        }
        __syncthreads();
    }
    __syncthreads();
    return rank;
}

// workgroup barrier that orders LDS traffic only (__syncthreads() also drains the vector-memory queue; the
// panel loop below issues global stores of R entries in every step that nobody reads before the kernel ends)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- panel-blocked pivoted QR for blocks that live in global memory ---------------------------------
// qrcp_mgs above streams the whole trailing matrix through one CU once per COLUMN (268 MB for a 256 x 256
// block: 3.8 ms at the ~70 GB/s one CU gets out of L2).  Here the trailing matrix is touched once per PANEL
// of 16 columns:
//   (a) window pivoting: the 16 remaining columns of largest (exactly recomputed) norm form the panel;
//   (b) panel factorisation: one wave owns one panel column IN REGISTERS; 16 steps of modified Gram-Schmidt
//       with greedy pivoting inside the panel, the pivot column broadcast through LDS; a pivot whose norm
//       fell by more than 100x inside the panel is projected a second time ("twice is enough"), so the
//       panel basis Q_p is orthonormal to rounding and (I - Q_p Q_p^H) below is a projector;
//   (c) trailing update, one wave per chunk of 16 columns, on v_mfma_f64_16x16x4_f64:
//       C = Q_p^H A (rows of R), A <- A - Q_p C, new column norms recomputed from the updated columns.
// Between panels this is block MODIFIED Gram-Schmidt (every panel sees the updated trailing matrix), whose
// R factor is backward stable like that of column MGS.  R entries are stored by PHYSICAL column index
// (X[j][phys k] = conj(r_jk)), so nothing is ever swapped and the row permutation handed on is the identity.
// ---- trailing update shared between workgroups ----------------------------------------------------------------------
// The trailing update of a panel (C = Q_p^H A, A <- A - Q_p C on the MFMA pipe) is 55 % of k_qr_large on a 202 x 202 block and
// runs at the MFMA rate of ONE CU (measured 471 of 860 us; window 81, panel steps 299), 77 % on a 400 x 400 block (3.0 of
// 3.9 ms).  With NW > 1 workgroups per block (see the launch site for when) the master (role 0) keeps the pivoting and the
// panel factorisation and publishes, per panel, the panel basis Q_p, the column map and the panel's geometry; every workgroup
// then updates the 16-column chunks ch with ch % NW == role -- all its waves together, qr_trailing_coop -- and the helpers hand
// back the new column norms.  Everything that crosses workgroups inside the launch follows cdna_hip_programming.md section 6,
// Guideline 16 (first table row of MI355X_MICROARCH.md "visibility"): every storing wave drains, workgroup barrier, ONE lane
// raises an epoch with an agent-scope atomic store; the other side polls that word from one lane, workgroup barrier, and every
// load of shared bytes is an sc1 load (served by the L2).  The STORES of the shared bytes -- Q_p, the updated columns of A, the
// rows of R in X -- come in two flavours: write-through (sc1), correct wherever the workgroups run; or plain, kept in the
// XCD's L2, used only when every workgroup of the block has read the same XCD id from the hardware (k_qr_large) -- the
// launch puts a block's workgroups at grid positions of one residue mod 8, which the dispatcher has been seen to deal to one
// XCD, but nothing is assumed: a block that finds itself spread out takes the first flavour.  Polls are bounded (qr_wait_ge).
// Measured, 202 x 202 (tools/ring_prof.py, -DHTN_QR_PROF): one workgroup 858 us (trailing 452); four workgroups through
// memory 964; four through the shared L2 755; the same with all 16 waves of a workgroup on its chunks 619 (trailing 212).
typedef unsigned int qr_u4 __attribute__((ext_vector_type(4)));
struct QrShare {
    int NW, role;
    bool local;              // every workgroup of the block reports the same XCD: bulk stores stay in that XCD's L2 (plain), see k_qr_large
    double2* qn;             // [16 * ldq] panel basis (global copy)
    int* scol;               // [n0] logical -> physical column
    double* cn2;             // [n0] squared norms the helpers computed (by logical position)
    int* meta;               // j0, nb_eff, k_first, last_panel, finished
    unsigned* qflag;         // master -> helpers: number of panels published so far
    unsigned* hflag;         // [NW]: helper -> master: panels completed
    unsigned* fail;
};
__device__ __forceinline__ bool qr_wait_ge(unsigned* flag, unsigned want, unsigned* fail) {
    const long long t0 = wall_clock64();
    for (unsigned spins = 1;; ++spins) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 0u) {
            if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (wall_clock64() - t0 > 300000000ll) {             // 3 s of the 100 MHz constant clock
                __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}
template <bool SH>
__device__ __forceinline__ double2 qr_ld(const double2* base, __amdgpu_buffer_rsrc_t rs, int idx) {
    if (!SH) return base[idx];
    const qr_u4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, idx * 16, 0, 16);
    double2 v;
    __builtin_memcpy(&v, &u, 16);
    return v;
}
template <bool SH>
__device__ __forceinline__ void qr_st(double2* base, __amdgpu_buffer_rsrc_t rs, int idx, double2 v, bool local) {
    if (!SH) {
        base[idx] = v;
        return;
    }
    qr_u4 u;
    __builtin_memcpy(&u, &v, 16);
    if (local) __builtin_amdgcn_raw_buffer_store_b128(u, rs, idx * 16, 0, 0);      // stays in the XCD's L2; the readers' sc1 loads hit it there
    else __builtin_amdgcn_raw_buffer_store_b128(u, rs, idx * 16, 0, 16);
}

// one 16-column chunk of the trailing update by one wave: rows j0 .. of R for its columns (-> X), then the columns themselves
// and their new norms (norm_out[k], LDS for the master, the shared array for a helper)
template <bool SH>
__device__ __forceinline__ void qr_trailing_chunk(double2* __restrict__ g0, __amdgpu_buffer_rsrc_t rg, double2* __restrict__ Xg,
                                                  __amdgpu_buffer_rsrc_t rx, const double2* Qn, int ldq, const int* s_col, int m0,
                                                  int m0p, int n0, int mp, int j0, int nb_eff, int k_first, int ch, bool last_panel,
                                                  double* norm_out, bool norm_shared, int lane, bool local) {
    const int l15 = lane & 15, l4 = lane >> 4;
    const int nks = m0p >> 2;
    const int k = k_first + 16 * ch + l15;
    const bool valid = k < n0;
    const int pk = valid ? s_col[k] : 0;
    const int abase = pk * m0;
    d4 cr = {0.0, 0.0, 0.0, 0.0}, ci = {0.0, 0.0, 0.0, 0.0};
    const double2* qrow = Qn + l15 * ldq + l4;
    double2 ring[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = 4 * u + l4;
        ring[u] = (valid && u < nks && i < m0) ? qr_ld<SH>(g0, rg, abase + i) : make_double2(0.0, 0.0);
    }
    for (int ks0 = 0; ks0 < nks; ks0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ks = ks0 + u;
            if (ks < nks) {
                const double2 b = ring[u];
                const int inx = 4 * (ks + 4) + l4;
                ring[u] = (valid && ks + 4 < nks && inx < m0) ? qr_ld<SH>(g0, rg, abase + inx) : make_double2(0.0, 0.0);
                const double2 qv = qrow[4 * ks];
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(qv.x, b.x, cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(qv.x, b.y, ci, 0, 0, 0);
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(qv.y, b.y, cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-qv.y, b.x, ci, 0, 0, 0);
            }
        }
    }
    // rows j0 + (l4 + 4 reg) of R, column k
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int t = l4 + 4 * reg;
        if (valid && t < nb_eff) qr_st<SH>(Xg, rx, (j0 + t) * mp + pk, make_double2(cr[reg], -ci[reg]), local);
    }
    if (last_panel) return;                       // no later panel reads the trailing columns
    double nrm = 0.0;
    for (int i0 = 0; i0 < m0p; i0 += 16) {
        double2 old[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = i0 + l4 + 4 * reg;
            old[reg] = (valid && i < m0) ? qr_ld<SH>(g0, rg, abase + i) : make_double2(0.0, 0.0);
        }
        d4 dr = {0.0, 0.0, 0.0, 0.0}, di = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double2 qa = Qn[(l4 + 4 * kk) * ldq + i0 + l15];
            dr = __builtin_amdgcn_mfma_f64_16x16x4f64(qa.x, cr[kk], dr, 0, 0, 0);
            di = __builtin_amdgcn_mfma_f64_16x16x4f64(qa.x, ci[kk], di, 0, 0, 0);
            dr = __builtin_amdgcn_mfma_f64_16x16x4f64(-qa.y, ci[kk], dr, 0, 0, 0);
            di = __builtin_amdgcn_mfma_f64_16x16x4f64(qa.y, cr[kk], di, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = i0 + l4 + 4 * reg;
            if (valid && i < m0) {
                const double2 x = make_double2(old[reg].x - dr[reg], old[reg].y - di[reg]);
                qr_st<SH>(g0, rg, abase + i, x, local);
                nrm += x.x * x.x + x.y * x.y;
            }
        }
    }
    nrm += __shfl_xor(nrm, 16, 64);      // over the 4 lanes (l4) that share a column
    nrm += __shfl_xor(nrm, 32, 64);
    if (valid && l4 == 0) {
        if (norm_shared) __hip_atomic_store(norm_out + k, nrm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else norm_out[k] = nrm;
    }
}

// a helper workgroup of a block: per published panel, fetch Q_p and the column map, update its chunks, report
// The trailing update of one workgroup's chunks (ch = role, role + NW, ... < nchunks) by ALL its 16 waves.  One wave per chunk
// (qr_trailing_chunk) leaves most of the SIMDs idle once helpers share the chunks: four workgroups x three chunks = three
// busy waves per CU for 12 us (404 f64 MFMAs at 64 clk).  Here S = 4 waves share a chunk when the block is small enough for
// the partial sums to fit the LDS beside the panel (m0p <= 256; `part` != nullptr): part p of a chunk takes every S-th group
// of four k-steps of C = Q_p^H A -- the S partial C tiles are added through LDS in part order, every part ends up with the
// same full C in the MFMA's D layout -- and every S-th row tile of A <- A - Q_p C; part 0 writes the rows of R, the column
// norms are added through LDS in part order.  S depends on the block's size only, never on NW or on which workgroup has the
// chunk: the result is the same bit for bit with or without helpers.  Called by every wave of the workgroup (barriers).
template <bool SH>
__device__ __forceinline__ void qr_trailing_coop(double2* __restrict__ g0, __amdgpu_buffer_rsrc_t rg, double2* __restrict__ Xg,
                                                 __amdgpu_buffer_rsrc_t rx, const double2* Qn, int ldq, const int* s_col, int m0,
                                                 int m0p, int n0, int mp, int j0, int nb_eff, int k_first, int nchunks, int role,
                                                 int NW, bool last_panel, double* norm_out, bool norm_shared, double* part,
                                                 int tid, bool local) {
    __shared__ double s_np[JAC_THREADS / 64][16];
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int cnt = nchunks > role ? (nchunks - role + NW - 1) / NW : 0;
    if (part == nullptr) {             // one wave per chunk
        for (int q = wave; q < cnt; q += JAC_THREADS / 64)
            qr_trailing_chunk<SH>(g0, rg, Xg, rx, Qn, ldq, s_col, m0, m0p, n0, mp, j0, nb_eff, k_first, role + NW * q, last_panel, norm_out,
                                  norm_shared, lane, local);
        return;
    }
    constexpr int S = 4, CPR = (JAC_THREADS / 64) / S;      // waves per chunk, chunks per round
    const int nks = m0p >> 2;
    for (int r0 = 0; r0 < cnt; r0 += CPR) {                 // (uniform over the workgroup)
        const int ci_ = r0 + wave / S, pp = wave % S;
        const bool act = ci_ < cnt;
        const int ch = role + NW * ci_;
        const int k = k_first + 16 * ch + l15;
        const bool valid = act && k < n0;
        const int pk = valid ? s_col[k] : 0;
        const int abase = pk * m0;
        d4 cr = {0.0, 0.0, 0.0, 0.0}, ci = {0.0, 0.0, 0.0, 0.0};
        if (act) {
            const double2* qrow = Qn + l15 * ldq + l4;
            double2 ring[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ks = 4 * pp + u, i = 4 * ks + l4;
                ring[u] = (valid && ks < nks && i < m0) ? qr_ld<SH>(g0, rg, abase + i) : make_double2(0.0, 0.0);
            }
            for (int ks0 = 4 * pp; ks0 < nks; ks0 += 4 * S) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ks = ks0 + u;
                    if (ks < nks) {
                        const double2 b = ring[u];
                        const int ksn = ks + 4 * S, inx = 4 * ksn + l4;
                        ring[u] = (valid && ksn < nks && inx < m0) ? qr_ld<SH>(g0, rg, abase + inx) : make_double2(0.0, 0.0);
                        const double2 qv = qrow[4 * ks];
                        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(qv.x, b.x, cr, 0, 0, 0);
                        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(qv.x, b.y, ci, 0, 0, 0);
                        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(qv.y, b.y, cr, 0, 0, 0);
                        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-qv.y, b.x, ci, 0, 0, 0);
                    }
                }
            }
        }
        // partial C tiles -> LDS ([wave][reg][lane]: conflict-free), added in part order by every part
        {
            double* mine = part + (size_t)wave * 512 + lane;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                mine[64 * reg] = cr[reg];
                mine[64 * (4 + reg)] = ci[reg];
            }
        }
        __syncthreads();
        {
            const double* base = part + (size_t)(wave - pp) * 512 + lane;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int q = 0; q < S; ++q) {
                    a += base[q * 512 + 64 * reg];
                    b += base[q * 512 + 64 * (4 + reg)];
                }
                cr[reg] = a;
                ci[reg] = b;
            }
        }
        if (pp == 0) {                                  // rows j0 + (l4 + 4 reg) of R, column k
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int t = l4 + 4 * reg;
                if (valid && t < nb_eff) qr_st<SH>(Xg, rx, (j0 + t) * mp + pk, make_double2(cr[reg], -ci[reg]), local);
            }
        }
        if (!last_panel) {                              // (uniform) no later panel reads the trailing columns otherwise
            double nrm = 0.0;
            if (act) {
                for (int i0 = 16 * pp; i0 < m0p; i0 += 16 * S) {
                    double2 old[4];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int i = i0 + l4 + 4 * reg;
                        old[reg] = (valid && i < m0) ? qr_ld<SH>(g0, rg, abase + i) : make_double2(0.0, 0.0);
                    }
                    d4 dr = {0.0, 0.0, 0.0, 0.0}, di = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double2 qa = Qn[(l4 + 4 * kk) * ldq + i0 + l15];
                        dr = __builtin_amdgcn_mfma_f64_16x16x4f64(qa.x, cr[kk], dr, 0, 0, 0);
                        di = __builtin_amdgcn_mfma_f64_16x16x4f64(qa.x, ci[kk], di, 0, 0, 0);
                        dr = __builtin_amdgcn_mfma_f64_16x16x4f64(-qa.y, ci[kk], dr, 0, 0, 0);
                        di = __builtin_amdgcn_mfma_f64_16x16x4f64(qa.y, cr[kk], di, 0, 0, 0);
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int i = i0 + l4 + 4 * reg;
                        if (valid && i < m0) {
                            const double2 x = make_double2(old[reg].x - dr[reg], old[reg].y - di[reg]);
                            qr_st<SH>(g0, rg, abase + i, x, local);
                            nrm += x.x * x.x + x.y * x.y;
                        }
                    }
                }
            }
            nrm += __shfl_xor(nrm, 16, 64);      // over the 4 lanes (l4) that share a column
            nrm += __shfl_xor(nrm, 32, 64);
            if (l4 == 0) s_np[wave][l15] = nrm;
            __syncthreads();
            if (valid && pp == 0 && l4 == 0) {
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < S; ++q) t += s_np[wave + q][l15];
                if (norm_shared) __hip_atomic_store(norm_out + k, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else norm_out[k] = t;
            }
        }
        __syncthreads();                                // the partial tiles and norms are free for the next round
    }
}

__device__ __forceinline__ void qr_helper(double2* __restrict__ g0, int m0, int n0, int r, double2* __restrict__ Xg, int mp, int* s_col,
                                          double2* Qn, int tid, const QrShare sh, double* part) {
    __shared__ int s_meta[8];
    const int lane = tid & 63, wave = tid >> 6;
    const int m0p = (m0 + 15) & ~15, ldq = m0p + 1;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)g0, 0, m0 * n0 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)Xg, 0, mp * r * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)sh.qn, 0, 16 * ldq * 16, 0x00020000);
    for (unsigned epoch = 1;; ++epoch) {
        if (tid == 0) {
            const bool ok = qr_wait_ge(sh.qflag, epoch, sh.fail);
            s_meta[5] = ok ? 0 : 1;
            for (int q = 0; q < 5; ++q) s_meta[q] = ok ? __hip_atomic_load(sh.meta + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
        __syncthreads();
        const int j0 = s_meta[0], nb_eff = s_meta[1], k_first = s_meta[2];
        const bool last_panel = s_meta[3] != 0, finished = s_meta[4] != 0 || s_meta[5] != 0;
        if (finished) break;
        for (int idx = tid; idx < 16 * ldq; idx += JAC_THREADS) Qn[idx] = qr_ld<true>(sh.qn, rq, idx);
        for (int k = k_first + tid; k < n0; k += JAC_THREADS) s_col[k] = __hip_atomic_load(sh.scol + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int nchunks = (n0 - k_first + 15) >> 4;
        qr_trailing_coop<true>(g0, rg, Xg, rx, Qn, ldq, s_col, m0, m0p, n0, mp, j0, nb_eff, k_first, nchunks, sh.role, sh.NW, last_panel, sh.cn2,
                               true, part, tid, sh.local);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(sh.hflag + sh.role, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#ifdef HTN_QR_PROF          // diagnostic build only (tools/ring_prof.py --qr): 100 MHz ticks per phase of k_qr_large, block 0
__device__ long long g_qr_prof[8];
extern "C" int htn_qr_prof_dump(long long* out) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qr_prof), sizeof(long long) * 8));
    return 0;
}
#define QR_T(var) const long long var = wall_clock64()
#define QR_ACC(slot, a, b) qprof[slot] += (b) - (a)
#else
#define QR_T(var)
#define QR_ACC(slot, a, b)
#endif
template <int EL, bool SH>
__device__ __forceinline__ int qrcp_blocked(double2* __restrict__ g0, int m0, int n0, int r,
                                            double2* __restrict__ Xg, int mp, int* s_col, double* s_cn2,
                                            double* s_piv, double2* Qn, int tid, double cut2, const QrShare sh, double* part) {
    __shared__ double s_pn[16];
    __shared__ int s_share_ok;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)g0, 0, m0 * n0 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)Xg, 0, mp * r * 16, 0x00020000);
    unsigned pub_epoch = 0;
    if (tid == 0) s_share_ok = 1;
    __shared__ double2 s_r[16][17];      // the panel's own R block, [pivot step][owner wave]; flushed once per panel
    __shared__ int s_pc[16];
    __shared__ int s_pos[16];
    __shared__ int s_nb;
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int m0p = (m0 + 15) & ~15, ldq = m0p + 1;
    for (int idx = tid; idx < mp * r; idx += JAC_THREADS) qr_st<SH>(Xg, rx, idx, make_double2(0.0, 0.0), sh.local);
    for (int k = wave; k < n0; k += JAC_THREADS / 64) {
        double c = 0.0;
        for (int i = lane; i < m0; i += 64) {
            const double2 x = g0[(int64_t)k * m0 + i];       // (written before this launch: plain loads)
            c += x.x * x.x + x.y * x.y;
        }
        c = wave_sum(c);
        if (lane == 0) {
            s_cn2[k] = c;
            s_col[k] = k;
        }
    }
    __syncthreads();
    if (wave == 0) {
        double f = 0.0;
        for (int k = lane; k < n0; k += 64) f += s_cn2[k];
        f = wave_sum(f);
        if (lane == 0) s_piv[1] = 1e-30 * f;       // "numerically zero" threshold on squared norms
    }
    __syncthreads();
    const double zero2 = s_piv[1];
    bool stop = false;
    int j0 = 0, rank = 0;
#ifdef HTN_QR_PROF
    long long qprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // window | panel load | panel steps | flush | trailing | total | panels
    const long long qt0 = wall_clock64();
#endif
    while (j0 < r && !stop) {
        QR_T(q0);
        const int nbmax = r - j0 < 16 ? r - j0 : 16;
        // ---- (a) window: the nbmax largest remaining columns move to logical positions j0 .. ----
        if (wave == 0) {
            int nb = 0;
            // rank-revealing stop: the squared Frobenius norm of everything not yet factorised (= |R22|_F^2, the
            // residual column norms are recomputed exactly) bounds every remaining singular value; below the
            // caller's cut (htn_jacobi_set_rank_cut) the remaining rows of R are dropped
            double mass = 0.0;
            for (int k = j0 + lane; k < n0; k += 64) mass += s_cn2[k];
            mass = wave_sum(mass);
            for (int t = 0; t < (mass <= cut2 ? 0 : nbmax); ++t) {
                const int j = j0 + t;
                // key = norm bits with the low 10 mantissa bits replaced by 1023 - k: one 64-bit max finds the
                // largest norm, ties (to 2^-42 relative) going to the smallest position
                unsigned long long key = 0ull;
                for (int k = j + lane; k < n0; k += 64) {
                    const unsigned long long kk =
                        ((unsigned long long)__double_as_longlong(s_cn2[k]) & ~0x3FFull) | (unsigned long long)(1023 - k);
                    key = kk > key ? kk : key;
                }
                key = wave_max_u64(key);
                const int bi = 1023 - (int)(key & 0x3FFull);
                const double best = s_cn2[bi];
                if (best <= zero2) break;
                if (lane == 0) {
                    const int tc = s_col[j];
                    s_col[j] = s_col[bi];
                    s_col[bi] = tc;
                    s_cn2[bi] = s_cn2[j];
                    s_cn2[j] = best;
                }
                ++nb;
            }
            if (lane == 0) s_nb = nb;
        }
        __syncthreads();
        QR_T(q1);
        QR_ACC(0, q0, q1);
        int nb = s_nb;
        if (nb == 0) break;                          // numerically rank deficient: remaining R rows are zero
        if (nb < nbmax) stop = true;
        // ---- (b) panel factorisation: wave u < nb owns logical column j0 + u in registers ----
        const bool owner = wave < nb;
        const int pc = owner ? s_col[j0 + wave] : 0;
        double2 a[EL];
        double cn = owner ? s_cn2[j0 + wave] : -1.0;
        const double ns = cn;
#pragma unroll
        for (int e = 0; e < EL; ++e) {
            const int i = lane + 64 * e;
            a[e] = (owner && i < m0) ? qr_ld<SH>(g0, rg, pc * m0 + i) : make_double2(0.0, 0.0);
        }
        if (!owner)                                   // unused panel positions project onto nothing
            for (int i = lane; i < m0p; i += 64) Qn[wave * ldq + i] = make_double2(0.0, 0.0);
        // no global store inside the step loop: a store keeps its source registers busy until it has left the
        // memory pipeline, and the next step would wait for that (s_waitcnt vmcnt(0)) before reusing them
        if (tid < 256) s_r[tid >> 4][tid & 15] = make_double2(0.0, 0.0);
        if (lane == 0) s_pc[wave] = pc;
        bool pending = owner;
        int nb_eff = nb;
        QR_T(q2);
        QR_ACC(1, q1, q2);
        for (int t = 0; t < nb; ++t) {
            if (lane == 0) s_pn[wave] = pending ? cn : -1.0;
            lds_barrier();
            unsigned long long key = 0ull;             // same packing, 4 tag bits: 15 - wave
            {
                const double v = s_pn[l15];
                if (v >= 0.0) key = ((unsigned long long)__double_as_longlong(v) & ~0xFull) | (unsigned long long)(15 - l15);
            }
            key = row_max16_u64(key);
            const int bu = 15 - (int)(key & 0xFull);
            const double best = key ? s_pn[bu] : -1.0;
            if (best <= zero2) {                      // the rest of the panel is numerically zero (uniform)
                nb_eff = t;
                stop = true;
                break;
            }
            if (wave == bu) {
                if (cn < 1e-4 * ns) {                 // second projection against the panel's earlier pivots
                    for (int s2 = 0; s2 < t; ++s2) {
                        double dr = 0.0, di = 0.0;
                        const double2* qs = Qn + s2 * ldq + lane;
#pragma unroll
                        for (int e = 0; e < EL; ++e) {
                            if (lane + 64 * e < m0p) {
                                const double2 q = qs[64 * e];
                                dr += q.x * a[e].x + q.y * a[e].y;
                                di += q.x * a[e].y - q.y * a[e].x;
                            }
                        }
                        dr = wave_sum(dr);
                        di = wave_sum(di);
#pragma unroll
                        for (int e = 0; e < EL; ++e) {
                            if (lane + 64 * e < m0p) {
                                const double2 q = qs[64 * e];
                                a[e].x -= dr * q.x - di * q.y;
                                a[e].y -= dr * q.y + di * q.x;
                            }
                        }
                        if (lane == 0) {
                            const double2 old = s_r[s2][wave];
                            s_r[s2][wave] = make_double2(old.x + dr, old.y - di);
                        }
                    }
                    double c = 0.0;
#pragma unroll
                    for (int e = 0; e < EL; ++e) c += a[e].x * a[e].x + a[e].y * a[e].y;
                    cn = wave_sum(c);
                }
                const double ipn = fast_rsq(cn);
                const double pn = cn * ipn;
#pragma unroll
                for (int e = 0; e < EL; ++e) {
                    const int i = lane + 64 * e;
                    if (i < m0p) Qn[t * ldq + i] = make_double2(a[e].x * ipn, a[e].y * ipn);
                }
                if (lane == 0) {
                    s_r[t][wave] = make_double2(pn, 0.0);
                    s_pos[t] = pc;
                }
                pending = false;
            }
            lds_barrier();
            if (pending) {
                double dr = 0.0, di = 0.0;
                const double2* qt = Qn + t * ldq + lane;       // zero padded up to m0p; a[e] is zero beyond m0
#pragma unroll
                for (int e = 0; e < EL; ++e) {
                    if (lane + 64 * e < m0p) {
                        const double2 q = qt[64 * e];
                        dr += q.x * a[e].x + q.y * a[e].y;          // conj(q) * a
                        di += q.x * a[e].y - q.y * a[e].x;
                    }
                }
                dr = wave_sum(dr);
                di = wave_sum(di);
                double c = 0.0;
#pragma unroll
                for (int e = 0; e < EL; ++e) {
                    if (lane + 64 * e < m0p) {
                        const double2 q = qt[64 * e];
                        a[e].x -= dr * q.x - di * q.y;
                        a[e].y -= dr * q.y + di * q.x;
                        c += a[e].x * a[e].x + a[e].y * a[e].y;
                    }
                }
                cn = wave_sum(c);
                if (lane == 0) s_r[t][wave] = make_double2(dr, -di);   // conj(r_tk)
            }
        }
        __syncthreads();
        QR_T(q3);
        QR_ACC(2, q2, q3);
        if (nb_eff < nb) {                            // zero the positions the early exit left unwritten
            for (int tt = nb_eff + wave; tt < nb; tt += JAC_THREADS / 64)
                for (int i = lane; i < m0p; i += 64) Qn[tt * ldq + i] = make_double2(0.0, 0.0);
        }
        if (tid < nb_eff) s_col[j0 + tid] = s_pos[tid];
        if (tid < 256 && (tid >> 4) < nb_eff && (tid & 15) < nb)      // rows of R before a column's own pivot step: 0
            qr_st<SH>(Xg, rx, (j0 + (tid >> 4)) * mp + s_pc[tid & 15], s_r[tid >> 4][tid & 15], sh.local);
        __syncthreads();
        QR_T(q4);
        QR_ACC(3, q3, q4);
        // ---- (c) trailing update: wave per chunk of 16 logical columns ----
        const int k_first = j0 + nb;
        const int nchunks = (n0 - k_first + 15) >> 4;
        const bool last_panel = stop || k_first >= r;
        const int NW = SH ? sh.NW : 1;
        if (SH && nchunks > 0) {
            // publish the panel: basis, column map of the trailing part, geometry; drain; ONE lane raises the epoch
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)sh.qn, 0, 16 * ldq * 16, 0x00020000);
            for (int idx = tid; idx < 16 * ldq; idx += JAC_THREADS) qr_st<true>(sh.qn, rq, idx, Qn[idx], sh.local);
            for (int k = k_first + tid; k < n0; k += JAC_THREADS) __hip_atomic_store(sh.scol + k, s_col[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) {
                __hip_atomic_store(sh.meta + 0, j0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sh.meta + 1, nb_eff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sh.meta + 2, k_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sh.meta + 3, last_panel ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sh.meta + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ++pub_epoch;
            if (tid == 0) __hip_atomic_store(sh.qflag, pub_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the master's share: chunks ch with ch % NW == 0
        qr_trailing_coop<SH>(g0, rg, Xg, rx, Qn, ldq, s_col, m0, m0p, n0, mp, j0, nb_eff, k_first, nchunks, 0, NW, last_panel, s_cn2, false, part,
                             tid, sh.local);
        if (SH && nchunks > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's own column updates are out before the next panel reads them
            __syncthreads();
            if (tid == 0) {
                bool ok = true;
                for (int h = 1; h < NW && ok; ++h) ok = qr_wait_ge(sh.hflag + h, pub_epoch, sh.fail);
                if (!ok) s_share_ok = 0;
            }
            __syncthreads();
            if (!s_share_ok) {
                stop = true;
            } else if (!last_panel) {
                for (int k = k_first + tid; k < n0; k += JAC_THREADS)
                    if (((k - k_first) >> 4) % NW != 0) s_cn2[k] = __hip_atomic_load(sh.cn2 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        QR_T(q5);
        QR_ACC(4, q4, q5);
#ifdef HTN_QR_PROF
        qprof[6] += 1;
#endif
        j0 += nb;
        rank += nb_eff;
    }
    __syncthreads();
#ifdef HTN_QR_PROF
    if (tid == 0 && blockIdx.x == 0) {
        qprof[5] = wall_clock64() - qt0;
        for (int q = 0; q < 8; ++q) g_qr_prof[q] = qprof[q];
    }
#endif
    if (SH) {                  // release the helpers (every path out of the panel loop ends here)
        if (tid == 0) {
            __hip_atomic_store(sh.meta + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(sh.qflag, pub_epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_share_ok) rank = -1;
    }
    for (int i = tid; i < n0; i += JAC_THREADS) s_col[i] = i;      // rows of X are physical columns already
    __syncthreads();
    return rank;
}

// one workgroup per LARGE block (X = R^H does not fit the LDS window): pivoted QR only; the sweeps follow as
// multi-launch block Jacobi.  Runs beside k_jacobi_svd (small blocks) on a forked stream.
#define QR_SYNC_WORDS 16         // per large block: panel epoch | helper epochs [1 .. 4] | ... | XCD ids [8 .. 11]
#define QR_BOX_BYTES 137728      // per large block: 16 x 513 complex128 (Q_p) | 512 int (column map) | 512 double (norms) | meta
__global__ __launch_bounds__(JAC_THREADS) void k_qr_large(double2* __restrict__ G, double2* __restrict__ Vj,
                                                          const htn_svd_block* __restrict__ desc,
                                                          const int* __restrict__ large_ids, int* __restrict__ perm,
                                                          double* __restrict__ zero2_out, double cut2,
                                                          int* __restrict__ rank_host, int NW, char* __restrict__ qr_box,
                                                          unsigned* __restrict__ qr_sync, int nl, int nx, int part_off) {
    extern __shared__ double2 g_lds[];
    __shared__ double s_piv[2];
    __shared__ int s_col[64 * JAC_MAXEL];
    __shared__ double s_cn2[64 * JAC_MAXEL];
    __shared__ int s_qlocal;
    // grid position -> (block, role): with helpers (NW > 1) the workgroups of one block sit at positions p = nx s + x with ONE x
    // (nx = 8: the dispatcher has been seen to deal position p to XCD p mod 8), blocks x, x + nx, ... stacked along s
    const int px = (int)(blockIdx.x % (unsigned)nx), ps = (int)(blockIdx.x / (unsigned)nx);
    const int li = (ps / NW) * nx + px, role = ps % NW;
    if (li >= nl) return;             // (a gap of the placement)
    const htn_svd_block D = desc[large_ids[li]];
    const int m = D.m, n = D.n, m0 = D.pad, tid = threadIdx.x;
    const int gsx = m <= 16 * JAC_MAXEL ? 16 : (m <= 32 * JAC_MAXEL ? 32 : 64);
    const int mp = gsx * ((m + gsx - 1) / gsx);
    double2* __restrict__ X = Vj + D.v_off;
    QrShare sh;
    sh.NW = NW, sh.role = role;
    char* box = qr_box + (size_t)li * QR_BOX_BYTES;
    sh.qn = (double2*)box;
    sh.scol = (int*)(box + 16 * 513 * 16);
    sh.cn2 = (double*)(box + 16 * 513 * 16 + 512 * 4);
    sh.meta = (int*)(box + 16 * 513 * 16 + 512 * 4 + 512 * 8);
    sh.qflag = qr_sync + li * QR_SYNC_WORDS;
    sh.hflag = qr_sync + li * QR_SYNC_WORDS + 1;
    sh.fail = qr_sync + nl * QR_SYNC_WORDS;
    sh.local = false;
    if (NW > 1) {
        // the same check as in the ring kernel (ring_run): take the hand-off through the XCD's L2 only if every workgroup of the
        // block READS the same XCD id from the hardware; any other outcome keeps the placement-independent sc1 form
        if (tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc = (xcc & 0xfu) + 1u;
            unsigned* xw = qr_sync + li * QR_SYNC_WORDS + 8;
            __hip_atomic_store(xw + role, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int local = 1;
            for (int q = 0; q < NW && local >= 0; ++q) {
                if (!qr_wait_ge(xw + q, 1u, sh.fail)) local = -1;
                else if (__hip_atomic_load(xw + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc) local = 0;
            }
            s_qlocal = local;
        }
        __syncthreads();
        sh.local = s_qlocal > 0;
        // (a timed-out wait has raised the failure word: the master's and the helpers' first hand-off see it and give up)
    }
    // LDS beside the panel for the partial sums of the shared chunks (qr_trailing_coop): blocks of up to 256 rows, when the launch
    // reserved it (part_off = the panel's extent in complex elements)
    double* part = (part_off > 0 && ((m0 + 15) & ~15) <= 256) ? (double*)(g_lds + part_off) : nullptr;
    if (role > 0) {
        qr_helper(G + D.g_off, m0, m, n, X, mp, s_col, (double2*)g_lds, tid, sh, part);
        return;
    }
    int rank;
    if (NW > 1) {
        if (m0 <= 256) rank = qrcp_blocked<4, true>(G + D.g_off, m0, m, n, X, mp, s_col, s_cn2, s_piv, (double2*)g_lds, tid, cut2, sh, part);
        else rank = qrcp_blocked<8, true>(G + D.g_off, m0, m, n, X, mp, s_col, s_cn2, s_piv, (double2*)g_lds, tid, cut2, sh, part);
    } else {
        if (m0 <= 256) rank = qrcp_blocked<4, false>(G + D.g_off, m0, m, n, X, mp, s_col, s_cn2, s_piv, (double2*)g_lds, tid, cut2, sh, part);
        else rank = qrcp_blocked<8, false>(G + D.g_off, m0, m, n, X, mp, s_col, s_cn2, s_piv, (double2*)g_lds, tid, cut2, sh, part);
    }
    if (tid == 0) {
        rank_host[li] = rank;                       // host-pinned: the host sizes the Jacobi tournament with it (< 0: a hand-off timed out)
        // threshold for "numerically zero" columns of X = R^H: |X|_F^2 = |G0|_F^2 (the QR preserves it), summed in fixed order above
        zero2_out[li] = s_piv[1];
        __threadfence_system();
    }
    for (int i = tid; i < m; i += JAC_THREADS) perm[li * 64 * JAC_MAXEL + i] = s_col[i];
}

__global__ __launch_bounds__(JAC_THREADS) void k_jacobi_svd(double2* __restrict__ G, double2* __restrict__ Vj,
                                                            double* __restrict__ S,
                                                            const htn_svd_block* __restrict__ desc,
                                                            int max_sweeps, double tol, int* __restrict__ info,
                                                            int lds_elems, const int* __restrict__ large_slot,
                                                            double cut2) {
    extern __shared__ double2 g_lds[];
    __shared__ double s_ratio;
    __shared__ double s_piv[2];
    __shared__ int s_col[64 * JAC_MAXEL];
    __shared__ double s_cn2[64 * JAC_MAXEL];
    if (large_slot && large_slot[blockIdx.x] >= 0) return;      // k_qr_large + block Jacobi own this block
    const htn_svd_block D = desc[blockIdx.x];
    const int m = D.m, n = D.n;
    double2* __restrict__ gglob = G + D.g_off;
    double2* __restrict__ v = Vj + D.v_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = JAC_THREADS / 64;
    const bool in_lds = (int64_t)m * n <= lds_elems;
    const bool accumulate = (D.flags & HTN_SVD_ACCUMULATE) != 0;
    const bool qrcp = (D.flags & HTN_SVD_QRCP) != 0;
    double2* g;
    if (qrcp) {
        // G0 (m0 x n0 = D.pad x D.m) at gglob; X = R^H (n0 x r = m x n), leading dimension mp = GS * E (zero padded),
        // in LDS when it fits, else in the Vj workspace (the planner reserves roundup(m, 64) * n elements there)
        const int m0 = D.pad;
        const int gsx = m <= 16 * JAC_MAXEL ? 16 : (m <= 32 * JAC_MAXEL ? 32 : 64);
        const int E = (m + gsx - 1) / gsx;
        const int mp = gsx * E;
        // (keeping only the leading columns in LDS and the rest in global memory was measured SLOWER than
        // all-global: the mixed case needs flat addressing, 4.4 ms vs 3.1 ms for a 107 x 107 block)
        const bool x_lds = (int64_t)mp * n <= lds_elems;
        const int nl = x_lds ? n : 0;
        const SplitCols X = {(double2*)g_lds, v, nl, mp};
        int rank;                 // columns of X the sweeps have to visit (the others are zero)
        if (m0 <= 16 * JAC_MAXEL) rank = qrcp_mgs<16>(gglob, m0, m, n, X, s_col, s_cn2, s_piv, tid, cut2);
        else if (m0 <= 32 * JAC_MAXEL) rank = qrcp_mgs<32>(gglob, m0, m, n, X, s_col, s_cn2, s_piv, tid, cut2);
        else rank = qrcp_mgs<64>(gglob, m0, m, n, X, s_col, s_cn2, s_piv, tid, cut2);
        {   // |X|_F^2 -> threshold for "numerically zero" columns
            double f = 0.0;
            for (int idx = tid; idx < mp * n; idx += JAC_THREADS) {
                const double2 x = X.col(idx / mp)[idx % mp];
                f += x.x * x.x + x.y * x.y;
            }
            f = wave_sum(f);
            if (lane == 0) s_cn2[wave] = f;          // fixed-order sum over the 16 waves: bit-reproducible
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int q = 0; q < JAC_THREADS / 64; ++q) t += s_cn2[q];
                s_ratio = t;
            }
            __syncthreads();
        }
        const double zero2q = 1e-30 * s_ratio;
        __syncthreads();
        int swq;
        if (x_lds) {        // everything in LDS: ds_* addressing
            const DenseCols<double2*> dc = {(double2*)g_lds, mp};
            if (gsx == 16) swq = jacobi_dispatch_e<16>(dc, E, rank, max_sweeps, tol, &s_ratio, tid, zero2q);
            else if (gsx == 32) swq = jacobi_dispatch_e<32>(dc, E, rank, max_sweeps, tol, &s_ratio, tid, zero2q);
            else swq = jacobi_dispatch_e<64>(dc, E, rank, max_sweeps, tol, &s_ratio, tid, zero2q);
        } else {            // everything in global memory (L2 resident): global_* addressing
            const DenseCols<double2*> dc = {v, mp};
            if (gsx == 16) swq = jacobi_dispatch_e<16>(dc, E, rank, max_sweeps, tol, &s_ratio, tid, zero2q);
            else if (gsx == 32) swq = jacobi_dispatch_e<32>(dc, E, rank, max_sweeps, tol, &s_ratio, tid, zero2q);
            else swq = jacobi_dispatch_e<64>(dc, E, rank, max_sweeps, tol, &s_ratio, tid, zero2q);
        }
        __syncthreads();
        // column norms + write back with the pivoting undone: row k of X is row s_col[k] of the result
        for (int j = wave; j < n; j += nwaves) {
            double sn = 0.0;
            for (int i = lane; i < m; i += 64) {
                const double2 x = X.col(j)[i];
                sn += x.x * x.x + x.y * x.y;
                gglob[(int64_t)j * m + s_col[i]] = x;
            }
            sn = wave_sum(sn);
            if (lane == 0) S[D.s_off + j] = sqrt(sn);
        }
        if (tid == 0) info[blockIdx.x] = swq;
        return;
    } else {
        // V = identity ; stage G into LDS when it fits
        if (accumulate)
            for (int idx = tid; idx < n * n; idx += JAC_THREADS) {
                const int i = idx % n, j = idx / n;
                v[idx] = make_double2(i == j ? 1.0 : 0.0, 0.0);
            }
        if (in_lds)
            for (int idx = tid; idx < m * n; idx += JAC_THREADS) g_lds[idx] = gglob[idx];
        __syncthreads();
        g = in_lds ? (double2*)g_lds : gglob;
    }
    // |G|_F^2 -> threshold for "numerically zero" columns
    {
        double f = 0.0;
        for (int idx = tid; idx < m * n; idx += JAC_THREADS) {
            const double2 x = g[idx];
            f += x.x * x.x + x.y * x.y;
        }
        f = wave_sum(f);
        if (lane == 0) s_cn2[wave] = f;              // fixed-order sum over the 16 waves: bit-reproducible
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int q = 0; q < JAC_THREADS / 64; ++q) t += s_cn2[q];
            s_ratio = t;
        }
        __syncthreads();
    }
    const double zero2 = 1e-30 * s_ratio;
    __syncthreads();
    int sw;
    if (m <= 16 * JAC_MAXEL) sw = jacobi_sweeps<16>(g, v, m, n, max_sweeps, tol, &s_ratio, tid, accumulate, zero2);
    else if (m <= 32 * JAC_MAXEL) sw = jacobi_sweeps<32>(g, v, m, n, max_sweeps, tol, &s_ratio, tid, accumulate, zero2);
    else sw = jacobi_sweeps<64>(g, v, m, n, max_sweeps, tol, &s_ratio, tid, accumulate, zero2);
    __syncthreads();
    // column norms (+ write back; the QR path undoes the pivoting: row k of X is row s_col[k] of the result)
    for (int j = wave; j < n; j += nwaves) {
        double s = 0.0;
        for (int i = lane; i < m; i += 64) {
            const double2 x = g[(int64_t)j * m + i];
            s += x.x * x.x + x.y * x.y;
            if (in_lds) gglob[(int64_t)j * m + i] = x;
        }
        s = wave_sum(s);
        if (lane == 0) S[D.s_off + j] = sqrt(s);
    }
    if (tid == 0) info[blockIdx.x] = sw;
}

// ---- large blocks: block one-sided Jacobi over several CUs ---------------------------------------------
// A block whose R^H does not fit one CU's LDS is VALU-bound on that CU (all n/2 pairs of a round on 16
// waves).  Here the columns are cut into panels of w = 8; one WORKGROUP handles one PAIR of panels (a "visit"),
// the panel pairs of one tournament round run on different CUs, and rounds are separated by kernel boundaries
// (the only cross-CU synchronisation used: no in-kernel grid barriers).  The host reads one number per block
// and outer sweep (max squared cosine) to stop.
#define JAC_PANEL 8
struct JacPairItem {
    int32_t blk, ci, ni, cj, nj, pad[3];
};

// ---- panel-pair visit on the Gram matrix ---------------------------------------------------------------
// Rotating the 2 w = 16 columns of mp rows directly costs, per round, a chain of LDS reads, dots over the
// columns, a lane reduction, the rotation and the write back -- about 1.7 us, almost all of it latency (the
// first version of this path did that, 37 us per visit).  Here only ONE pass touches the long columns:
//   (1) G = P^H P  (16 x 16, Hermitian) on v_mfma_f64_16x16x4_f64, the row range split over the 4 waves;
//   (2) one cyclic sweep of two-sided Jacobi on G, 8 disjoint rotations per round, accumulating U (16 x 16);
//       one thread per matrix element, no reductions, two workgroup barriers per round;
//   (3) P <- P U on the MFMA pipe, written straight back to global.
// In exact arithmetic this is one-sided Jacobi on the columns (a one-sided rotation of two columns IS the
// two-sided rotation of their Gram matrix).  In floating point G is re-formed from the columns at every visit, so rounding in
// (2) only perturbs the rotation angles, never the orthogonality test: the stopping criterion is evaluated on
// the exact Gram of the visit.  The pivot 2 x 2 block is updated with Rutishauser's formulas
// (a' = a - t|g|, b' = b + t|g|, 0 off-diagonal), which keep the small diagonal entries of a graded Gram
// matrix relatively accurate (Demmel & Veselic 1992); the remaining entries use the plain bilinear update.
// partner of LDS column slot x in round r.  mode 0 (cross visit, 8 rounds): slot p of panel A meets slot
// 8 + (p + r) % 8 of panel B -- every A-B pair exactly once.  mode 1 (intra visit, 7 rounds): round-robin
// tournament inside each group of 8 slots.
__device__ __forceinline__ int jg_partner(int x, int r, int mode) {
    if (mode == 0) return x < 8 ? 8 + ((x + r) & 7) : ((x - r) & 7);
    const int g = x & 8, k = x & 7;
    if (k == 7) return g + r;
    if (k == r) return g + 7;
    int y = 2 * r - k + 7;
    y = y >= 14 ? y - 14 : (y >= 7 ? y - 7 : y);
    return g + y;
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cmulc(double2 ac, double2 b) {      // conj(ac) * b
    return make_double2(fma(ac.x, b.x, ac.y * b.y), fma(ac.x, b.y, -ac.y * b.x));
}


// rotation that annihilates the (lo, hi) entry g of a 2 x 2 Hermitian pivot [[aa, g], [conj g, bb]]: c = cos, cq = cos *
// tan / |g|, dq = tan * |g|; identity (returns false) for dead slots, numerically zero columns and pairs already
// orthogonal to tol.  No branches: cos from cos(2 theta) = |h| / w runs in parallel with the reciprocal that gives q.
__device__ __forceinline__ bool jg_rotation(double aa, double bb, double2 g, bool live, double z2, double tol2,
                                            double& c, double& cq, double& dq) {
    const double g2 = fma(g.x, g.x, g.y * g.y);
    const bool on = live && aa > z2 && bb > z2 && g2 > 0.0 && g2 > tol2 * aa * bb;
    const double h = on ? bb - aa : 0.0, g2s = on ? g2 : 1.0, ah = fabs(h);
    const double w2 = fma(h, h, 4.0 * g2s);
    const double iw = fast_rsq(w2);
    double q = 2.0 * fast_rcp(fma(w2, iw, ah));
    q = h >= 0.0 ? q : -q;
    const double c2 = fma(0.5 * ah, iw, 0.5);
    const double cc = c2 * fast_rsq(c2);
    c = on ? cc : 1.0;
    cq = on ? cc * q : 0.0;
    dq = on ? q * g2s : 0.0;
    return on;
}

#define JG_LD 17
#define JG_PS 9            // doubles per lane of the Gram partials (8 used): odd stride, no 16-way bank conflict
#define JG_GU_ELEMS (4 * 64 * JG_PS / 2)        // complex128 elements of the G/U region (>= 4 * 16 * JG_LD; the partials alias it)
__global__ __launch_bounds__(256) void k_jacobi_pairs_gram(double2* __restrict__ Vj,
                                                           const htn_svd_block* __restrict__ desc,
                                                           const int* __restrict__ large_ids,
                                                           const JacPairItem* __restrict__ items,
                                                           const double* __restrict__ zero2,
                                                           unsigned long long* __restrict__ ratio_bits,
                                                           const int* __restrict__ done, double tol, int inner) {
    extern __shared__ double2 g_lds[];
    __shared__ unsigned long long s_rbits;
    const JacPairItem it = items[blockIdx.x];
    if (done[it.blk]) return;
    const htn_svd_block D = desc[large_ids[it.blk]];
    const int m = D.m, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int gsx = m <= 16 * JAC_MAXEL ? 16 : (m <= 32 * JAC_MAXEL ? 32 : 64);
    const int mp = gsx * ((m + gsx - 1) / gsx);
    const int ldp = mp + 1;
    double2* __restrict__ X = Vj + D.v_off;
    const int mode = it.pad[0], nrounds = mode ? 7 : 8;
    double2* P = g_lds;                              // [16][ldp]
    double2* GU = g_lds + 16 * ldp;                  // G[2][16][17], U[2][16][17]; the Gram partials alias it
    const double z2 = zero2[it.blk];
    const double tol2 = tol * tol;
    if (tid == 0) s_rbits = 0ull;
    // ---- (0) columns -> LDS: 16 lanes per column, 256 contiguous bytes per request ----
    {
        const int c = tid >> 4, r = tid & 15;
        const bool live = c < 8 ? c < it.ni : c - 8 < it.nj;      // slots 0..7: first panel, 8..15: second
        const int col = c < 8 ? it.ci + c : it.cj + (c - 8);
        const double2* __restrict__ src = X + (int64_t)(live ? col : 0) * mp;
        double2* dst = P + c * ldp;
        const int ne = mp >> 4;
        for (int e0 = 0; e0 < ne; e0 += 16) {          // 16 requests in flight per lane: one round trip up to 256 rows
            double2 v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r + 16 * (e0 + e);
                v[e] = (live && e0 + e < ne) ? src[row] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (e0 + e < ne) dst[r + 16 * (e0 + e)] = v[e];
        }
    }
    __syncthreads();
    // ---- (1) Gram: lane (l15, l4) feeds P[4 ks + l4][l15] as A and as B operand ----
    {
        d4 g1 = {0.0, 0.0, 0.0, 0.0}, g2 = {0.0, 0.0, 0.0, 0.0}, mm = {0.0, 0.0, 0.0, 0.0};
        const double2* pc = P + l15 * ldp + l4;
        const int nit = mp >> 4;                     // k-steps of this wave: ks = wave + 4 it
        // operands of four steps are fetched together, the next four are in flight during the 12 MFMAs
        double2 nx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) nx[u] = u < nit ? pc[4 * (wave + 4 * u)] : make_double2(0.0, 0.0);
        for (int it0 = 0; it0 < nit; it0 += 4) {
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = nx[u];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (it0 + 4 + u < nit) nx[u] = pc[4 * (wave + 4 * (it0 + 4 + u))];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (it0 + u < nit) {                 // wave-uniform
                    g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u].x, v[u].x, g1, 0, 0, 0);
                    g2 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u].y, v[u].y, g2, 0, 0, 0);
                    mm = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u].x, v[u].y, mm, 0, 0, 0);      // M[a][b] = sum re_a im_b
                }
            }
        }
        double* part = (double*)GU + (wave * 64 + lane) * JG_PS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            part[r] = g1[r] + g2[r];
            part[4 + r] = mm[r];
        }
    }
    __syncthreads();
    const int ei = tid >> 4, ej = tid & 15;
    {
        // accumulator element (row = l4 + 4 reg, col = l15): G[i][j] sits in lane j + 16 (i & 3), reg i >> 2
        const double* pa = (const double*)GU + (ej + 16 * (ei & 3)) * JG_PS + (ei >> 2);
        const double* pb = (const double*)GU + (ei + 16 * (ej & 3)) * JG_PS + 4 + (ej >> 2);
        double gr = 0.0, mij = 0.0, mji = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {               // fixed order: deterministic
            gr += pa[w * 64 * JG_PS];
            mij += pa[w * 64 * JG_PS + 4];
            mji += pb[w * 64 * JG_PS];
        }
        __syncthreads();                            // partials consumed; the region becomes G / U
        GU[ei * JG_LD + ej] = make_double2(gr, mij - mji);
        GU[2 * 16 * JG_LD + ei * JG_LD + ej] = make_double2(ei == ej ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    // ---- convergence measure of this visit: max squared cosine over its column pairs (exact Gram) ----
    {
        double ratio = 0.0;
        const bool vi = ei < 8 ? ei < it.ni : ei - 8 < it.nj, vj = ej < 8 ? ej < it.ni : ej - 8 < it.nj;
        if (vi && vj && ei < ej && (mode ? (ei >> 3) == (ej >> 3) : (ei < 8 && ej >= 8))) {
            const double dii = GU[ei * JG_LD + ei].x, djj = GU[ej * JG_LD + ej].x;
            const double2 g = GU[ei * JG_LD + ej];
            if (dii > z2 && djj > z2) ratio = fma(g.x, g.x, g.y * g.y) * fast_rcp(dii * djj);
        }
        const unsigned long long key = wave_max_u64((unsigned long long)__double_as_longlong(ratio));
        if (lane == 0 && key) atomicMax(&s_rbits, key);
    }
    __syncthreads();
    const unsigned long long first_bits = s_rbits;
    if (__longlong_as_double((long long)first_bits) <= tol2) {     // nothing to rotate in this visit (uniform)
        if (tid == 0 && first_bits) atomicMax(&ratio_bits[it.blk], first_bits);
        return;
    }
    // ---- (2) two-sided Jacobi on G, U <- U J ----
    // One thread per element (i, j).  It needs the rotation of i's pair (row operation) and of j's pair (column
    // operation) and computes both itself from the pivot entries -- 256-fold redundant arithmetic, but the
    // round is then a single chain [12 LDS reads -> two interleaved rotations -> update -> write] with ONE
    // barrier (G and U are double buffered), instead of [8 lanes rotate] barrier [256 lanes apply] barrier.
    unsigned livemask = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) livemask |= (unsigned)(q < 8 ? q < it.ni : q - 8 < it.nj) << q;
    int cur = 0;
    for (int sw = 0; sw < inner; ++sw) {
        for (int r = 0; r < nrounds; ++r) {
            const double2* Gc = GU + cur * 16 * JG_LD;
            const double2* Uc = GU + (2 + cur) * 16 * JG_LD;
            double2* Gn = GU + (cur ^ 1) * 16 * JG_LD;
            double2* Un = GU + (2 + (cur ^ 1)) * 16 * JG_LD;
            const int ib = jg_partner(ei, r, mode), jb = jg_partner(ej, r, mode);
            const int ilo = ei < ib ? ei : ib, ihi = ei < ib ? ib : ei;
            const int jlo = ej < jb ? ej : jb, jhi = ej < jb ? jb : ej;
            const double iaa = Gc[ilo * JG_LD + ilo].x, ibb = Gc[ihi * JG_LD + ihi].x;
            const double2 ig = Gc[ilo * JG_LD + ihi];                  // conj(a_lo) . a_hi
            const double jaa = Gc[jlo * JG_LD + jlo].x, jbb = Gc[jhi * JG_LD + jhi].x;
            const double2 jg = Gc[jlo * JG_LD + jhi];
            const double2 gij = Gc[ei * JG_LD + ej], gijb = Gc[ei * JG_LD + jb];
            const double2 gibj = Gc[ib * JG_LD + ej], gibjb = Gc[ib * JG_LD + jb];
            const double2 uij = Uc[ei * JG_LD + ej], uijb = Uc[ei * JG_LD + jb];
            // J = [[c, c q g], [-c q conj(g), c]] on (lo, hi), q = tan / |g|: real diagonal, no phase division.
            // Both rotations are computed branch-free (jg_rotation) so that their two dependent chains interleave.
            double ci, cqi, dqi, cj, cqj, dqj;
            const bool oni = jg_rotation(iaa, ibb, ig, ((livemask >> ilo) & (livemask >> ihi) & 1u) != 0, z2, tol2, ci, cqi, dqi);
            jg_rotation(jaa, jbb, jg, ((livemask >> jlo) & (livemask >> jhi) & 1u) != 0, z2, tol2, cj, cqj, dqj);
            // column j of J: J[j][j] = c, J[jb][j] = -cq conj(g) if j is the lower index, +cq g if the upper
            const double2 jpj = ej == jlo ? make_double2(-cqj * jg.x, cqj * jg.y) : make_double2(cqj * jg.x, cqj * jg.y);
            const double2 ipi = ei == ilo ? make_double2(-cqi * ig.x, cqi * ig.y) : make_double2(cqi * ig.x, cqi * ig.y);
            double2 t2 = cmul(gijb, jpj);
            const double2 Tij = make_double2(fma(cj, gij.x, t2.x), fma(cj, gij.y, t2.y));
            t2 = cmul(gibjb, jpj);
            const double2 Tibj = make_double2(fma(cj, gibj.x, t2.x), fma(cj, gibj.y, t2.y));
            t2 = cmulc(ipi, Tibj);
            double2 gn = make_double2(fma(ci, Tij.x, t2.x), fma(ci, Tij.y, t2.y));
            {               // pivot block by Rutishauser's formulas: a' = a - t|g|, b' = b + t|g|, off-diagonal 0
                const double dg = ei == ilo ? iaa - dqi : ibb + dqi;
                const bool pd = oni && ej == ei, po = oni && ej == ib;
                gn.x = pd ? dg : (po ? 0.0 : gn.x);
                gn.y = (pd || po) ? 0.0 : gn.y;
            }
            t2 = cmul(uijb, jpj);
            Gn[ei * JG_LD + ej] = gn;
            Un[ei * JG_LD + ej] = make_double2(fma(cj, uij.x, t2.x), fma(cj, uij.y, t2.y));
            __syncthreads();
            cur ^= 1;
        }
    }
    // ---- (3) P <- P U, transposed product so that a lane's 16 neighbours write 256 contiguous bytes:
    //      D[b][i] = sum_a U[a][b] P[i][a]:  A operand U[4 kk + l4][l15], B operand P[i0 + l15][4 kk + l4] ----
    {
        const double2* Uc = GU + (2 + cur) * 16 * JG_LD;
        double ur[4], ui[4], nui[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double2 u = Uc[(4 * kk + l4) * JG_LD + l15];
            ur[kk] = u.x;
            ui[kk] = u.y;
            nui[kk] = -u.y;
        }
        int colg[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = l4 + 4 * r;
            colg[r] = b < 8 ? (b < it.ni ? it.ci + b : -1) : (b - 8 < it.nj ? it.cj + (b - 8) : -1);
        }
        const int nrt = mp >> 4;
        const double2* pb = P + l4 * ldp + l15;
        double2 pn[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) pn[kk] = wave < nrt ? pb[4 * kk * ldp + wave * 16] : make_double2(0.0, 0.0);
        for (int rt = wave; rt < nrt; rt += 4) {
            const int i0 = rt * 16;
            double2 p[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) p[kk] = pn[kk];
            if (rt + 4 < nrt) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) pn[kk] = pb[4 * kk * ldp + i0 + 64];
            }
            d4 ar = {0.0, 0.0, 0.0, 0.0}, ai = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                ar = __builtin_amdgcn_mfma_f64_16x16x4f64(ur[kk], p[kk].x, ar, 0, 0, 0);
                ai = __builtin_amdgcn_mfma_f64_16x16x4f64(ur[kk], p[kk].y, ai, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f64_16x16x4f64(nui[kk], p[kk].y, ar, 0, 0, 0);
                ai = __builtin_amdgcn_mfma_f64_16x16x4f64(ui[kk], p[kk].x, ai, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (colg[r] >= 0) X[(int64_t)colg[r] * mp + i0 + l15] = make_double2(ar[r], ai[r]);
        }
    }
    if (tid == 0) atomicMax(&ratio_bits[it.blk], first_bits);
}

// after every outer sweep: convergence bookkeeping of the large blocks ON THE DEVICE (one thread per block), so
// the host never has to answer before the next sweep can start.  active_out is host-pinned: the host reads it
// one sweep late (the next sweep is already enqueued; if everything had converged its visits return at once).
__global__ void k_jacobi_check(unsigned long long* __restrict__ ratio_bits, int* __restrict__ done,
                               int* __restrict__ sweeps, int nl, double thr, int* __restrict__ active_out) {
    __shared__ int s_active;
    if (threadIdx.x == 0) s_active = 0;
    __syncthreads();
    for (int li = threadIdx.x; li < nl; li += blockDim.x) {
        if (!done[li]) {
            const double mx = __longlong_as_double((long long)ratio_bits[li]);
            sweeps[li] += 1;
            if (mx <= thr) done[li] = 1;
            else atomicAdd(&s_active, 1);
        }
        ratio_bits[li] = 0ull;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *active_out = s_active;
        __threadfence_system();
    }
}

__global__ __launch_bounds__(JAC_THREADS) void k_jacobi_finish(double2* __restrict__ G, const double2* __restrict__ Vj,
                                                               double* __restrict__ S,
                                                               const htn_svd_block* __restrict__ desc,
                                                               const int* __restrict__ large_ids,
                                                               const int* __restrict__ perm,
                                                               const int* __restrict__ sweeps,
                                                               const int* __restrict__ done, int* __restrict__ info) {
    const int b = large_ids[blockIdx.x];
    const htn_svd_block D = desc[b];
    const int m = D.m, n = D.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gsx = m <= 16 * JAC_MAXEL ? 16 : (m <= 32 * JAC_MAXEL ? 32 : 64);
    const int mp = gsx * ((m + gsx - 1) / gsx);
    const double2* __restrict__ X = Vj + D.v_off;
    double2* __restrict__ g = G + D.g_off;
    const int* __restrict__ pc = perm + blockIdx.x * 64 * JAC_MAXEL;
    for (int j = wave; j < n; j += JAC_THREADS / 64) {
        double sn = 0.0;
        for (int i = lane; i < m; i += 64) {
            const double2 x = X[(int64_t)j * mp + i];
            sn += x.x * x.x + x.y * x.y;
            g[(int64_t)j * m + pc[i]] = x;          // undo the pivoting: row k of X is row perm[k] of the result
        }
        sn = wave_sum(sn);
        if (lane == 0) S[D.s_off + j] = sqrt(sn);
    }
    if (tid == 0) info[b] = done[blockIdx.x] ? sweeps[blockIdx.x] : -sweeps[blockIdx.x];
}

// =====================================================================================================================
// Ring block Jacobi: the large blocks' sweeps in ONE launch
// =====================================================================================================================
// The multi-launch path above pays, per tournament round, a kernel boundary plus a visit that re-derives everything from
// global memory (columns -> LDS, Gram, P U back): 15 us per round, ~210 dependent rounds per centre bond at chi = 1024, with
// a tenth of the chip busy.  Here a block of n columns is given P workgroups (one per CU, 1024 threads) that stay resident
// for ALL sweeps.  The columns are cut into 2 P panels of w columns; every workgroup keeps two panels in LDS (w * mp <= 4608
// elements each) and plays the classic round-robin ("caterpillar") tournament on panels: in each of the 2 P - 1 rounds of a
// sweep it rotates its w x w cross pairs directly on the LDS-resident columns (plain one-sided rotations, one group of GS
// lanes per pair, the panel-T column of a group kept in registers for the whole round), then the panels move one position:
// top panels to the next workgroup, bottom panels to the previous one (workgroup 0 keeps its top, the last one turns its
// top into its bottom).  Once per sweep the pairs INSIDE the resident panels are rotated.  A sweep costs
// (2 P - 1) x (w steps of ~0.5 us + one panel exchange) instead of ~(n / 8) x 15 us.
//
// Hand-off between workgroups inside the launch (cdna_hip_programming.md section 6, Guideline 16, form R1 / first table row of
// MI355X_MICROARCH.md "visibility"): the payload goes out as 16-byte WRITE-THROUGH (sc1) stores into the receiver's mailbox,
// every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane stores the round's epoch into the receiver's
// flag with an agent-scope atomic store; the receiver polls that one word with relaxed agent-scope loads from ONE lane,
// workgroup barrier, then EVERY load of the payload is an sc1 load.  No placement assumption: correctness does not depend on
// which CU or XCD a workgroup landed on.  Mailbox slots alternate by epoch parity; neighbours exchange in both directions in
// every round, so a sender can be at most one round ahead of its receiver and never overwrites a slot that is still unread.
// All workgroups of a launch are co-resident (the host caps the grid at the CU count; LDS use forces one per CU); every poll
// is bounded by wall-clock time and a shared failure word, so a launch always drains.
#define RING_THREADS 512
#define RING_PANEL_ELEMS 4608         // complex128 elements of one resident panel (2 panels = 144 KiB of LDS)
#define RING_MAX_P 64
struct RingItem {
    int32_t li;        // large-block index (large_ids[li] = block in desc)
    int32_t k, P;      // CU slot of this workgroup, CU slots of the block
    int32_t w;         // panel width: panel q = columns [q w, min((q + 1) w, n))
    int32_t n;         // columns taking part (the rank the QR found, <= desc.n)
    int32_t g0;        // grid index of the block's slot 0 (flag addressing)
    int64_t mbox;      // element offset of the block's mailboxes: slot k owns 4 slots [role][parity] of w * mp elements
};                     // 32 bytes
typedef unsigned int ring_u4 __attribute__((ext_vector_type(4)));
// Lanes per column (GS) and elements per lane (E).  A rotation step is a dependent chain (LDS read -> dots -> DPP sums ->
// rotation arithmetic -> update -> LDS write -> barrier) and every VALU instruction of a wave costs 4 clocks whatever it does,
// so the sums and the scalar arithmetic of a rotation must be amortised over many elements per lane.  Measured per step of
// 15-16 pairs on a 202 x 202 block: 64 lanes x 4 elements (one pair per wave, 16 waves) 1.86 us -- 260 wave instructions per
// pair, 64 of them useful, VALU-issue bound; 16 lanes x 14 elements (FOUR pairs per wave, the sums stay inside one DPP row)
// 1.06 us; 32 lanes x 7 elements (two pairs per wave, two busy waves per SIMD) 1.12 us.  Beyond 256 rows a lane cannot hold
// a column in 16 lanes' registers: 64 lanes x <= 8 elements.
__host__ __device__ __forceinline__ int ring_gs(int m) { return m <= 256 ? 16 : 64; }
__host__ __device__ __forceinline__ int ring_e(int m) {
    const int g = ring_gs(m);
    return (m + g - 1) / g;
}
// one lane polls one word (relaxed, agent scope); false: timed out or another workgroup reported failure
__device__ __forceinline__ bool ring_wait_ge(unsigned* flag, unsigned want, unsigned* fail) {
    const long long t0 = wall_clock64();
    for (unsigned spins = 1;; ++spins) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 0u) {
            if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (wall_clock64() - t0 > 300000000ll) {             // 3 s of the 100 MHz constant clock
                __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

// LDS panel (cnt contiguous elements) -> mailbox slot, 16-byte stores (the caller drains and raises the flag): sc1
// (write-through to memory, any placement) or, when the block's workgroups have FOUND themselves on one XCD (`local`, see
// ring_run), plain stores that stay in that XCD's L2 -- the reader's sc1 loads are served from the same L2
__device__ __forceinline__ void ring_send(const double2* __restrict__ src, double2* slot, int cnt, int tid, bool local) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slot, 0, cnt * 16, 0x00020000);
    if (local) {
        for (int idx = tid; idx < cnt; idx += RING_THREADS) {
            const double2 v = src[idx];
            ring_u4 u;
            __builtin_memcpy(&u, &v, 16);
            __builtin_amdgcn_raw_buffer_store_b128(u, rs, idx * 16, 0, 0);
        }
    } else {
        for (int idx = tid; idx < cnt; idx += RING_THREADS) {
            const double2 v = src[idx];
            ring_u4 u;
            __builtin_memcpy(&u, &v, 16);
            __builtin_amdgcn_raw_buffer_store_b128(u, rs, idx * 16, 0, 16);       // aux 16 = sc1
        }
    }
}
// the epoch word that tells a neighbour its panel has arrived (after the drain and the barrier): agent scope, or -- all
// workgroups of the block on one XCD -- a store that stays in the shared L2 (the poll is an agent-scope load either way)
__device__ __forceinline__ void ring_flag(unsigned* flag, unsigned epoch, bool local) {
    if (local) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// mailbox slot -> LDS panel, every load sc1; all loads of a thread in flight together (cnt <= 9 * 512)
__device__ __forceinline__ void ring_recv(double2* dst, const double2* slot, int cnt, int tid) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slot, 0, cnt * 16, 0x00020000);
    // (unconditional loads: the descriptor's range check returns zeros beyond cnt, and a conditionally filled register
    // array makes the compiler carry -- and spill -- the whole array as one tuple)
#define RING_LD(q) const ring_u4 u##q = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + q * RING_THREADS) * 16, 0, 16)
    RING_LD(0);
    RING_LD(1);
    RING_LD(2);
    RING_LD(3);
    RING_LD(4);
    RING_LD(5);
    RING_LD(6);
    RING_LD(7);
    RING_LD(8);
#undef RING_LD
    auto put = [&](int idx, ring_u4 u) {
        if (idx < cnt) {
            double2 v;
            __builtin_memcpy(&v, &u, 16);
            dst[idx] = v;
        }
    };
    put(tid + 0 * RING_THREADS, u0);
    put(tid + 1 * RING_THREADS, u1);
    put(tid + 2 * RING_THREADS, u2);
    put(tid + 3 * RING_THREADS, u3);
    put(tid + 4 * RING_THREADS, u4);
    put(tid + 5 * RING_THREADS, u5);
    put(tid + 6 * RING_THREADS, u6);
    put(tid + 7 * RING_THREADS, u7);
    put(tid + 8 * RING_THREADS, u8);
}

// Rotation of a column pair held in registers; returns the squared cosine seen.  With g = a^H b, h = |b|^2 - |a|^2:
//   q = sign(h) 2 / (|h| + sqrt(h^2 + 4 |g|^2)) = tan / |g|,  c = 1 / sqrt(1 + q^2 |g|^2),
//   [a' b'] = [a b] [[c, c q g], [-c q conj(g), c]]
// -- real diagonal, so neither |g| nor the phase g / |g| is ever formed (three reciprocal-type operations instead of
// five); against the textbook form b' carries an extra unit phase, which a one-sided Jacobi SVD is free to choose.
// NORMS = false: aa / bb come in as the tracked squared norms of the two columns and go out updated by Rutishauser's
// identities |a'|^2 = |a|^2 - q |g|^2, |b'|^2 = |b|^2 + q |g|^2 (as LAPACK's xGESVJ tracks them); they only steer the
// rotation angle and the stopping test and are recomputed from the columns at the start of every round, i.e. after at
// most w updates.  Halves the dot products and the reductions of a rotation.
template <int GS, int E, bool NORMS>
__device__ __forceinline__ double ring_rotate(double2 (&a)[E], const double2 (&b)[E], double2* b_out, double& aa, double& bb,
                                              double tol2, double zero2) {
    double gr = 0.0, gi = 0.0;
    if (NORMS) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            sa = fma(a[e].x, a[e].x, fma(a[e].y, a[e].y, sa));
            sb = fma(b[e].x, b[e].x, fma(b[e].y, b[e].y, sb));
        }
        aa = group_sum<GS>(sa);
        bb = group_sum<GS>(sb);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        gr = fma(a[e].x, b[e].x, fma(a[e].y, b[e].y, gr));      // conj(a) * b
        gi = fma(a[e].x, b[e].y, fma(-a[e].y, b[e].x, gi));
    }
    gr = group_sum<GS>(gr);
    gi = group_sum<GS>(gi);
    if (aa <= zero2 || bb <= zero2) return 0.0;
    const double g2 = fma(gr, gr, gi * gi);
    const double ab = aa * bb;
    if (g2 == 0.0) return 0.0;
    // the squared cosine only steers the stopping test: the hardware reciprocal (no Newton step) is plenty
    const double ratio2 = g2 * __builtin_amdgcn_rcp(ab);
    if (g2 <= tol2 * ab) return ratio2;
    // ANY real q gives an exactly unitary rotation once c = 1 / sqrt(1 + q^2 |g|^2) is accurate; q itself only sets the
    // angle, so the hardware rsq / rcp seeds (relative error ~1e-8: the pair is left with a cosine 1e-8 times the one it
    // had, below what the next sweep's quadratic convergence leaves anyway) do without their Newton steps
    const double h = bb - aa;
    const double w2 = fma(h, h, 4.0 * g2);
    const double w = w2 * __builtin_amdgcn_rsq(w2);
    double q = 2.0 * __builtin_amdgcn_rcp(fabs(h) + w);
    q = h >= 0.0 ? q : -q;
    const double c = fast_rsq(fma(q * q, g2, 1.0));
    const double cq = c * q;
    const double sr = cq * gr, si = -cq * gi;                   // sig = c q conj(g):  a' = c a - sig b,  b' = conj(sig) a + c b
    aa = fma(-q, g2, aa);
    bb = fma(q, g2, bb);
    // b' goes straight to its column in LDS (only a rotated column is written back), a is updated IN PLACE (scale, then
    // accumulate): neither needs a register copy where the rotated and the skipped path meet
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double2 nb;
        nb.x = fma(si, a[e].y, fma(sr, a[e].x, c * b[e].x));
        nb.y = fma(-si, a[e].x, fma(sr, a[e].y, c * b[e].y));
        b_out[GS * e] = nb;
        double x = a[e].x * c, y = a[e].y * c;
        x = fma(-sr, b[e].x, x);
        y = fma(-sr, b[e].y, y);
        a[e].x = fma(si, b[e].y, x);
        a[e].y = fma(-si, b[e].x, y);
    }
    return ratio2;
}

// The same rotation in SCALED ("fast") form for the cross rounds: a column is kept as (scale, stored vector) with true column =
// scale * stored.  With the true g = sa sb g_st the update  [a' b'] = [a b] [[c, c q g], [-c q conj(g), c]]  becomes
//   a_st' = a_st - (q sb^2 conj(g_st)) b_st,   b_st' = b_st + (q sa^2 g_st) a_st,   sa' = c sa,   sb' = c sb
// -- 8 real multiply-adds per element pair instead of 12 (the factor c moves into the two scales).  Scales start at 1 in
// every round and are applied when the round ends (at most w factors c >= 1/sqrt 2 accumulate: no range issue).
template <int GS, int E>
__device__ __forceinline__ double ring_rotate_scaled(double2 (&a)[E], const double2 (&b)[E], double2* b_out, double& aa, double& bb,
                                                     double& sa, double& sb, double tol2, double zero2) {
    // conj(a_st) * b_st in FOUR independent chains (even / odd elements x re / im): a dependent f64 multiply-add cannot issue
    // back to back, and with one busy wave per SIMD nothing else fills the slots
    double gr = 0.0, gi = 0.0, gr1 = 0.0, gi1 = 0.0;
#pragma unroll
    for (int e = 0; e + 1 < E; e += 2) {
        gr = fma(a[e].x, b[e].x, fma(a[e].y, b[e].y, gr));
        gi = fma(a[e].x, b[e].y, fma(-a[e].y, b[e].x, gi));
        gr1 = fma(a[e + 1].x, b[e + 1].x, fma(a[e + 1].y, b[e + 1].y, gr1));
        gi1 = fma(a[e + 1].x, b[e + 1].y, fma(-a[e + 1].y, b[e + 1].x, gi1));
    }
    if (E & 1) {
        gr = fma(a[E - 1].x, b[E - 1].x, fma(a[E - 1].y, b[E - 1].y, gr));
        gi = fma(a[E - 1].x, b[E - 1].y, fma(-a[E - 1].y, b[E - 1].x, gi));
    }
    gr = group_sum<GS>(gr + gr1);
    gi = group_sum<GS>(gi + gi1);
    if (aa <= zero2 || bb <= zero2) return 0.0;
    const double ss = sa * sb;
    const double g2 = ss * ss * fma(gr, gr, gi * gi);           // |g|^2 of the true columns
    const double ab = aa * bb;
    if (g2 == 0.0) return 0.0;
    const double ratio2 = g2 * __builtin_amdgcn_rcp(ab);
    if (g2 <= tol2 * ab) return ratio2;
    const double h = bb - aa;
    const double w2 = fma(h, h, 4.0 * g2);
    const double w = w2 * __builtin_amdgcn_rsq(w2);
    double q = 2.0 * __builtin_amdgcn_rcp(fabs(h) + w);
    q = h >= 0.0 ? q : -q;
    const double c = fast_rsq(fma(q * q, g2, 1.0));
    aa = fma(-q, g2, aa);
    bb = fma(q, g2, bb);
    const double qb = q * sb * sb, qa = q * sa * sa;
    const double mur = qb * gr, mui = -qb * gi;                 // mu = q sb^2 conj(g_st)
    const double nur = qa * gr, nui = qa * gi;                  // nu = q sa^2 g_st
    sa *= c;
    sb *= c;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double2 nb;
        nb.x = fma(-nui, a[e].y, fma(nur, a[e].x, b[e].x));
        nb.y = fma(nui, a[e].x, fma(nur, a[e].y, b[e].y));
        b_out[GS * e] = nb;
        double x = fma(-mur, b[e].x, a[e].x), y = fma(-mur, b[e].y, a[e].y);
        a[e].x = fma(mui, b[e].y, x);
        a[e].y = fma(-mui, b[e].x, y);
    }
    return ratio2;
}

// all nt x nb cross pairs of the two resident panels: group g owns T column g in registers (and its tracked norm), B columns
// pass through LDS, their tracked norms through bnorm[]
template <int GS, int E>
__device__ __forceinline__ double ring_cross(double2* T, double2* B, int nt, int nb, int tid, double tol2, double zero2, double* bnorm,
                                             double* bscale) {
    constexpr int mp = GS * E;
    const int grp = tid / GS, sub = tid % GS;
    const int wm = nt > nb ? nt : nb;
    double ratio = 0.0;
    if (nt == 0 || nb == 0) return ratio;                      // (uniform over the workgroup)
    const bool own = grp < nt;
    double2 a[E];
    double aa = 0.0, sa = 1.0;
    if (own) {
#pragma unroll
        for (int e = 0; e < E; ++e) a[e] = T[grp * mp + sub + GS * e];
        double t = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) t = fma(a[e].x, a[e].x, fma(a[e].y, a[e].y, t));
        aa = group_sum<GS>(t);
    }
    if (grp < nb) {            // exact squared norm of B column grp; its scale starts at 1
        const double2* bc = B + grp * mp + sub;
        double sb = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double2 v = bc[GS * e];
            sb = fma(v.x, v.x, fma(v.y, v.y, sb));
        }
        sb = group_sum<GS>(sb);
        if (sub == 0) {
            bnorm[grp] = sb;
            bscale[grp] = 1.0;
        }
    }
    __syncthreads();
    for (int s = 0; s < wm; ++s) {
        int j = grp + s;
        j = j >= wm ? j - wm : j;
        if (own && j < nb) {
            double2 b[E];
            double2* bc = B + j * mp + sub;
#pragma unroll
            for (int e = 0; e < E; ++e) b[e] = bc[GS * e];
            double bb = bnorm[j], sb = bscale[j];
            const double rr = ring_rotate_scaled<GS, E>(a, b, bc, aa, bb, sa, sb, tol2, zero2);
            ratio = rr > ratio ? rr : ratio;
            if (sub == 0) {
                bnorm[j] = bb;
                bscale[j] = sb;
            }
        }
        __syncthreads();
    }
    if (own) {                 // the scales come out with the columns: T from registers, B in place
#pragma unroll
        for (int e = 0; e < E; ++e) T[grp * mp + sub + GS * e] = make_double2(a[e].x * sa, a[e].y * sa);
    }
    if (grp < nb) {
        const double sb = bscale[grp];
        if (sb != 1.0) {
            double2* bc = B + grp * mp + sub;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double2 v = bc[GS * e];
                bc[GS * e] = make_double2(v.x * sb, v.y * sb);
            }
        }
    }
    __syncthreads();
    return ratio;
}

// the pairs inside each resident panel (circle tournament per panel, both panels side by side)
template <int GS, int E>
__device__ __forceinline__ double ring_intra(double2* T, int nt, double2* B, int nb, int tid, double tol2, double zero2) {
    constexpr int mp = GS * E;
    const int grp = tid / GS, sub = tid % GS;
    const int npT = nt + (nt & 1), npB = nb + (nb & 1), hT = npT >> 1, hB = npB >> 1;
    const int steps = (npT > npB ? npT : npB) - 1;
    double ratio = 0.0;
    for (int r = 0; r < steps; ++r) {
        double2* base = nullptr;
        int np = 0, pn = 0, p = 0;
        if (grp < hT) base = T, np = npT, pn = nt, p = grp;
        else if (grp < hT + hB) base = B, np = npB, pn = nb, p = grp - hT;
        if (base && r < np - 1) {
            const int md = np - 1;
            int i, j;
            if (p == 0) i = np - 1, j = r;
            else {
                i = r + p;
                i = i >= md ? i - md : i;
                j = r + md - p;
                j = j >= md ? j - md : j;
            }
            if (i < pn && j < pn) {
                const int lo = i < j ? i : j, hi = i < j ? j : i;
                const double rr = jacobi_pair_pad<GS, E>(base + lo * mp, base + hi * mp, sub, tol2, zero2);
                ratio = rr > ratio ? rr : ratio;
            }
        }
        __syncthreads();
    }
    return ratio;
}

#ifdef HTN_RING_PROF        // diagnostic build only (tools/ring_prof.py): per-workgroup 100 MHz tick sums per phase
__device__ long long g_ring_prof[256 * 8];
extern "C" int htn_ring_prof_dump(long long* out) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ring_prof), sizeof(long long) * 256 * 8));
    return 0;
}
#define RING_T(var) const long long var = wall_clock64()
#define RING_ACC(slot, a, b) prof[slot] += (b) - (a)
#else
#define RING_T(var)
#define RING_ACC(slot, a, b)
#endif
struct RingArgs {
    double2* Vj;
    double2* G;
    double* S;
    const htn_svd_block* desc;
    const int* large_ids;
    const RingItem* items;
    const int* perm;
    const double* zero2;
    double2* mbox;
    unsigned* sync;              // [flags: 2 per workgroup | arrive: nl * max_sweeps | fail | pad | conv (u64): nl * max_sweeps]
    int arrive_off, fail_off, conv_off;      // in 32-bit words (conv_off even)
    int xcc_off;                 // 32-bit words: one per workgroup, XCD id + 1 once the workgroup has started
    int max_sweeps;
    double tol;
    int* info;
    int* sweeps_out;             // host-mapped, [nl]: outer sweeps of each block (0 = failed)
};

template <int GS, int E>
__device__ __forceinline__ void ring_run(const RingArgs A, const RingItem it, const htn_svd_block D, double2* lds, int* s_top, int* s_bot,
                         unsigned long long* s_rbits, int* s_ok, double* s_bnorm, int* s_local) {
    constexpr int mp = GS * E;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = it.P, k = it.k, w = it.w, n = it.n, m = D.m;
    double2* __restrict__ X = A.Vj + D.v_off;
    const int slot_elems = w * mp;
    double2* bufT = lds;
    double2* bufB = lds + slot_elems;
    unsigned* flags = A.sync;
    unsigned* fail = A.sync + A.fail_off;
    const double zero2 = A.zero2[it.li];
    const double tol2 = A.tol * A.tol, thr = fmax(tol2, 0.1 * A.tol);
    auto ncols = [&](int q) {
        const int c = n - q * w;
        return c < 0 ? 0 : (c > w ? w : c);
    };
    if (tid < P) {
        s_top[tid] = 2 * tid;
        s_bot[tid] = 2 * tid + 1;
    }
    if (tid == 0) {
        *s_ok = 1;
        // Which XCD is this?  The host places a block's workgroups at grid positions that the dispatcher has been SEEN to
        // deal to one XCD (position mod 8); nothing promises that, so every workgroup publishes the XCD id it reads from
        // the hardware register and the block takes the cheaper hand-off (plain stores kept in the shared L2) only if ALL
        // of its workgroups report the same id -- otherwise the placement-independent sc1 form, as before.  Every
        // workgroup of the block reads the same P words, so all take the same decision.
        int local = 0;
        if (P > 1 && A.xcc_off >= 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc = (xcc & 0xfu) + 1u;
            unsigned* xw = A.sync + A.xcc_off + it.g0;
            __hip_atomic_store(xw + k, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            local = 1;
            for (int q = 0; q < P && local >= 0; ++q) {
                if (!ring_wait_ge(xw + q, 1u, fail)) local = -1;
                else if (__hip_atomic_load(xw + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc) local = local > 0 ? 0 : local;
            }
            if (local < 0) *s_ok = 0;
        }
        *s_local = local > 0 ? 1 : 0;
    }
    __syncthreads();
    const bool local = *s_local != 0;
    {   // resident panels <- X (written by the QR kernel before this launch: plain loads), re-padded from X's leading
        // dimension (the rule of the one-workgroup kernel) to mp rows; rows >= m are zero in both
        const int gsx = m <= 16 * JAC_MAXEL ? 16 : (m <= 32 * JAC_MAXEL ? 32 : 64);
        const int ldx = gsx * ((m + gsx - 1) / gsx);
        for (int side = 0; side < 2; ++side) {
            const int q = side ? s_bot[k] : s_top[k];
            double2* buf = side ? bufB : bufT;
            const int cnt = ncols(q) * mp;
            for (int idx = tid; idx < cnt; idx += RING_THREADS) {
                const int c = idx / mp, i = idx - c * mp;
                buf[idx] = i < m ? X[(int64_t)(q * w + c) * ldx + i] : make_double2(0.0, 0.0);
            }
        }
    }
    __syncthreads();
#ifdef HTN_RING_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // cross | send+drain | flags+wait | recv | intra | conv | total | sweeps
    const long long prof_t0 = wall_clock64();
#endif
    int sweeps = 0;
    bool done = n < 2, ok = *s_ok != 0;
    const int rounds = 2 * P - 1;
    while (!done && ok && sweeps < A.max_sweeps) {
        double ratio = 0.0;
        for (int r = 0; r < rounds && ok; ++r) {
            const unsigned epoch = (unsigned)(sweeps * rounds + r + 1);
            const int par = (int)(epoch & 1u);
            const int nt = ncols(s_top[k]), nb = ncols(s_bot[k]);
            RING_T(p0);
            const double rr = ring_cross<GS, E>(bufT, bufB, nt, nb, tid, tol2, zero2, s_bnorm, s_bnorm + RING_THREADS / 16);
            ratio = rr > ratio ? rr : ratio;
            RING_T(p1);
            RING_ACC(0, p0, p1);
            if (P < 2) continue;
            // ---- panels move one position: send, drain, raise the flags ----
            double2* box = A.mbox + it.mbox;                    // slot of workgroup kd, role, parity: ((kd * 2 + role) * 2 + par)
            if (k == 0) ring_send(bufB, box + (int64_t)((1 * 2 + 0) * 2 + par) * slot_elems, nb * mp, tid, local);            // bottom -> top of 1
            else {
                if (k < P - 1) ring_send(bufT, box + (int64_t)(((k + 1) * 2 + 0) * 2 + par) * slot_elems, nt * mp, tid, local);   // top -> top of k + 1
                ring_send(bufB, box + (int64_t)(((k - 1) * 2 + 1) * 2 + par) * slot_elems, nb * mp, tid, local);                  // bottom -> bottom of k - 1
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            RING_T(p2);
            RING_ACC(1, p1, p2);
            if (tid == 0) {
                if (k == 0) ring_flag(flags + (it.g0 + 1) * 2 + 0, epoch, local);
                else {
                    if (k < P - 1) ring_flag(flags + (it.g0 + k + 1) * 2 + 0, epoch, local);
                    ring_flag(flags + (it.g0 + k - 1) * 2 + 1, epoch, local);
                }
                // the panel ids follow the same permutation in every workgroup of the block
                const int t_last = s_top[P - 1], b0 = s_bot[0];
                for (int q = P - 1; q >= 2; --q) s_top[q] = s_top[q - 1];
                s_top[1] = b0;
                for (int q = 0; q + 1 < P; ++q) s_bot[q] = s_bot[q + 1];
                s_bot[P - 1] = t_last;
                // ---- wait for what comes in ----
                bool good = true;
                if (k > 0) good = ring_wait_ge(flags + (it.g0 + k) * 2 + 0, epoch, fail);
                if (good && k < P - 1) good = ring_wait_ge(flags + (it.g0 + k) * 2 + 1, epoch, fail);
                if (!good) *s_ok = 0;
            }
            if (k == P - 1) {           // the last workgroup's top panel becomes its bottom panel: swap the roles of the buffers
                double2* t = bufT;
                bufT = bufB;
                bufB = t;
            }
            __syncthreads();
            RING_T(p3);
            RING_ACC(2, p2, p3);
            ok = *s_ok != 0;
            if (ok) {
                const double2* mine = box + (int64_t)(k * 4) * slot_elems;
                if (k > 0) ring_recv(bufT, mine + (int64_t)(0 * 2 + par) * slot_elems, ncols(s_top[k]) * mp, tid);
                if (k < P - 1) ring_recv(bufB, mine + (int64_t)(1 * 2 + par) * slot_elems, ncols(s_bot[k]) * mp, tid);
            }
            __syncthreads();
            RING_T(p4);
            RING_ACC(3, p3, p4);
        }
        if (!ok) break;
        RING_T(p5);
        {
            const double rr = ring_intra<GS, E>(bufT, ncols(s_top[k]), bufB, ncols(s_bot[k]), tid, tol2, zero2);
            ratio = rr > ratio ? rr : ratio;
        }
        RING_T(p6);
        RING_ACC(4, p5, p6);
        // ---- block-wide maximum of the squared cosines this sweep SAW: one returning atomic max + one arrival per workgroup ----
        if (tid == 0) *s_rbits = 0ull;
        __syncthreads();
        {
            const unsigned long long key = wave_max_u64((unsigned long long)__double_as_longlong(ratio));
            if (lane == 0 && key) atomicMax(s_rbits, key);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long* conv = (unsigned long long*)(A.sync + A.conv_off) + (int64_t)it.li * A.max_sweeps + sweeps;
            unsigned* arrive = A.sync + A.arrive_off + it.li * A.max_sweeps + sweeps;
            unsigned long long mx = *s_rbits;
            if (P > 1) {
                const unsigned long long old = __hip_atomic_fetch_max(conv, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the maximum has been applied before this workgroup arrives
                (void)old;
                __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ring_wait_ge(arrive, (unsigned)P, fail)) mx = __hip_atomic_load(conv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *s_ok = 0;
            }
            *s_rbits = mx;
        }
        __syncthreads();
        ok = *s_ok != 0;
        ++sweeps;
        done = __longlong_as_double((long long)*s_rbits) <= thr;
        __syncthreads();
        RING_T(p7);
        RING_ACC(5, p6, p7);
    }
#ifdef HTN_RING_PROF
    if (tid == 0 && blockIdx.x < 256) {
        prof[6] = wall_clock64() - prof_t0;
        prof[7] = (local ? 1000000 : 0) + sweeps * 1000 + P;
        for (int q = 0; q < 8; ++q) g_ring_prof[blockIdx.x * 8 + q] = prof[q];
    }
#endif
    // ---- result: column norms -> S, columns (pivoting undone) -> G; the panels are wherever the tournament left them ----
    double2* __restrict__ g = A.G + D.g_off;
    const int* __restrict__ pc = A.perm + it.li * 64 * JAC_MAXEL;
    for (int side = 0; side < 2; ++side) {
        const double2* buf = side ? bufB : bufT;
        const int q = side ? s_bot[k] : s_top[k];
        const int nc = ncols(q);
        for (int c = wave; c < nc; c += RING_THREADS / 64) {
            const int col = q * w + c;
            double sn = 0.0;
            for (int i = lane; i < m; i += 64) {
                const double2 x = buf[c * mp + i];
                sn += x.x * x.x + x.y * x.y;
                g[(int64_t)col * m + pc[i]] = x;
            }
            sn = wave_sum(sn);
            if (lane == 0) A.S[D.s_off + col] = sqrt(sn);
        }
    }
    for (int col = n + k * (RING_THREADS / 64) + wave; col < D.n; col += P * (RING_THREADS / 64)) {      // columns beyond the rank: zero
        for (int i = lane; i < m; i += 64) g[(int64_t)col * m + i] = make_double2(0.0, 0.0);
        if (lane == 0) A.S[D.s_off + col] = 0.0;
    }
    if (k == 0 && tid == 0) {
        A.info[A.large_ids[it.li]] = (done && ok) ? sweeps : -(sweeps > 0 ? sweeps : 1);
        A.sweeps_out[it.li] = ok ? sweeps + ((local || P == 1) ? 1000 : 0) : 0;      // (+ 1000: hand-offs went through one XCD's L2)
        __threadfence_system();
    }
}

__global__ __launch_bounds__(RING_THREADS) void k_jacobi_ring(RingArgs A) {
    extern __shared__ double2 g_lds[];
    __shared__ int s_top[RING_MAX_P], s_bot[RING_MAX_P];
    __shared__ unsigned long long s_rbits;
    __shared__ int s_ok, s_local;
    __shared__ double s_bnorm[2 * (RING_THREADS / 16)];      // tracked squared norms | scales of the bottom panel's columns
    const RingItem it = A.items[blockIdx.x];
    if (it.P <= 0) return;           // (a gap of the XCD-aware placement: see plan_ring)
    const htn_svd_block D = A.desc[A.large_ids[it.li]];
    const int m = D.m;
    const int gs = ring_gs(m), E = ring_e(m);
#define RING_CASE(GSV, EV) \
    case EV: ring_run<GSV, EV>(A, it, D, g_lds, s_top, s_bot, &s_rbits, &s_ok, s_bnorm, &s_local); break;
    if (gs == 16) {                  // m <= 256
        switch (E) {
            RING_CASE(16, 1) RING_CASE(16, 2) RING_CASE(16, 3) RING_CASE(16, 4) RING_CASE(16, 5) RING_CASE(16, 6) RING_CASE(16, 7)
            RING_CASE(16, 8) RING_CASE(16, 9) RING_CASE(16, 10) RING_CASE(16, 11) RING_CASE(16, 12) RING_CASE(16, 13)
            RING_CASE(16, 14) RING_CASE(16, 15)
            default: ring_run<16, 16>(A, it, D, g_lds, s_top, s_bot, &s_rbits, &s_ok, s_bnorm, &s_local);
        }
    } else {                         // 256 < m <= 512: E = 5 .. 8
        switch (E) {
            RING_CASE(64, 5) RING_CASE(64, 6) RING_CASE(64, 7)
            default: ring_run<64, 8>(A, it, D, g_lds, s_top, s_bot, &s_rbits, &s_ok, s_bnorm, &s_local);
        }
    }
#undef RING_CASE
}

// per-stream scratch of the multi-launch path (grown on demand; htn_common.h: owned by the stream's registry entry and
// released by the backend that owns the stream)
struct JacScratch {
    int device = -1;
    void* dev = nullptr;
    size_t bytes = 0;
    void* pinned = nullptr;             // host -> device staging (work lists, ids)
    size_t pinned_bytes = 0;
    int* flags = nullptr;               // device -> host: [active count per sweep | rank per large block]; coherent, its own block
    int* flags_dev = nullptr;           // device view of it
    size_t flags_elems = 0;
    void* ring_sync = nullptr;          // ring Jacobi: flags / arrival counters / maxima (zeroed before every launch)
    size_t ring_sync_bytes = 0;
    void* ring_mbox = nullptr;          // ring Jacobi: mailboxes
    size_t ring_mbox_bytes = 0;
    void* qr_box = nullptr;             // k_qr_large with helper workgroups: per-block panel basis / column map / norms (sc1 traffic)
    size_t qr_box_bytes = 0;
    void* ring_items = nullptr;         // ring Jacobi: work items (device) and their pinned staging
    void* ring_items_h = nullptr;
    size_t ring_items_cap = 0;
    int cu_count = 0;
    int xcd_local = 0;                  // 1: in the last ring launch every block's workgroups found themselves on one XCD
    hipStream_t aux = nullptr;          // forked stream: small blocks run beside the large-block pipeline
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_sweep[2] = {nullptr, nullptr};
    ~JacScratch() {
        if (device >= 0) (void)hipSetDevice(device);
        if (aux) {
            (void)hipStreamSynchronize(aux);
            (void)hipStreamDestroy(aux);
            (void)hipEventDestroy(ev_fork);
            (void)hipEventDestroy(ev_join);
            (void)hipEventDestroy(ev_sweep[0]);
            (void)hipEventDestroy(ev_sweep[1]);
        }
        if (dev) (void)hipFree(dev);
        if (pinned) (void)hipHostFree(pinned);
        if (flags) (void)hipHostFree(flags);
        if (qr_box) (void)hipFree(qr_box);
        if (ring_sync) (void)hipFree(ring_sync);
        if (ring_mbox) (void)hipFree(ring_mbox);
        if (ring_items) (void)hipFree(ring_items);
        if (ring_items_h) (void)hipHostFree(ring_items_h);
    }
};
static std::mutex g_js_mu;
static std::map<hipStream_t, std::unique_ptr<JacScratch>> g_js_res;

void htn_svd_release_stream(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_js_mu);
    g_js_res.erase(st);
}

static int js_get(hipStream_t st, JacScratch** out) {
    std::lock_guard<std::mutex> lk(g_js_mu);
    auto& slot = g_js_res[st];
    if (!slot) {
        slot = std::make_unique<JacScratch>();
        HIP_TRY(hipGetDevice(&slot->device));
        HIP_TRY(hipDeviceGetAttribute(&slot->cu_count, hipDeviceAttributeMultiprocessorCount, slot->device));
    }
    *out = slot.get();
    return 0;
}

static int js_reserve(JacScratch& g_js, size_t dev_bytes, size_t pin_bytes, size_t flag_elems) {
    if (dev_bytes > g_js.bytes) {
        if (g_js.dev) HIP_TRY(hipFree(g_js.dev));
        g_js.dev = nullptr, g_js.bytes = 0;
        HIP_TRY(hipMalloc(&g_js.dev, dev_bytes * 2));
        if (htn_debug_poison()) HIP_TRY(hipMemset(g_js.dev, 0xFF, dev_bytes * 2));
        g_js.bytes = dev_bytes * 2;
    }
    if (pin_bytes > g_js.pinned_bytes) {
        if (g_js.pinned) HIP_TRY(hipHostFree(g_js.pinned));
        g_js.pinned = nullptr, g_js.pinned_bytes = 0;
        HIP_TRY(hipHostMalloc(&g_js.pinned, pin_bytes * 2, hipHostMallocDefault));
        g_js.pinned_bytes = pin_bytes * 2;
    }
    if (flag_elems > g_js.flags_elems) {
        if (g_js.flags) HIP_TRY(hipHostFree(g_js.flags));
        g_js.flags = nullptr, g_js.flags_elems = 0;
        HIP_TRY(hipHostMalloc((void**)&g_js.flags, sizeof(int) * flag_elems * 2, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer((void**)&g_js.flags_dev, g_js.flags, 0));
        g_js.flags_elems = flag_elems * 2;
    }
    if (!g_js.aux) {
        HIP_TRY(hipStreamCreateWithFlags(&g_js.aux, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g_js.ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g_js.ev_join, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g_js.ev_sweep[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g_js.ev_sweep[1], hipEventDisableTiming));
    }
    return 0;
}

extern "C" int htn_jacobi_svd_z(void* G, void* Vj, double* S, const htn_svd_block* desc,
                                const htn_svd_block* desc_host, int32_t n_blocks, int32_t max_m_host,
                                int32_t max_sweeps, double tol, int32_t* info_dev, const htn_svd_opts* opts,
                                void* stream) {
    if (n_blocks <= 0) return 0;
    // per-call settings (ABI 2): nothing process-wide is read or written here
    const int g_jac_split = opts && opts->split_elems > 0 ? opts->split_elems : 0;
    const double g_jac_cut = opts && opts->rank_cut > 0.0 ? opts->rank_cut : 0.0;
    if (max_m_host > 64 * JAC_MAXEL) return fail_msg("htn_jacobi_svd_z: block taller than 512 rows");
    hipStream_t st = (hipStream_t)stream;
    // dynamic LDS window for the matrix: 144 KiB leaves room for the static shared variables
    const int lds_elems = 9216;     // complex128 elements = 144 KiB
    // one-time kernel attributes: per device and thread safe (the attribute lives with the device's code object)
    {
        static std::mutex attr_mu;
        static bool attr_set[64] = {};
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(attr_mu);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_jacobi_svd, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        lds_elems * (int)sizeof(double2)));
            HIP_TRY(hipFuncSetAttribute((const void*)k_qr_large, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        16 * (64 * JAC_MAXEL + 1) * (int)sizeof(double2)));
            HIP_TRY(hipFuncSetAttribute((const void*)k_jacobi_pairs_gram, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (16 * (64 * JAC_MAXEL + 1) + JG_GU_ELEMS) * (int)sizeof(double2)));
            HIP_TRY(hipFuncSetAttribute((const void*)k_jacobi_ring, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        2 * RING_PANEL_ELEMS * (int)sizeof(double2)));
            attr_set[dev] = true;
        }
    }
    // blocks that do not fit one CU's LDS go to the multi-launch block-Jacobi path (needs the host copy of desc)
    const int large_min = g_jac_split > 0 ? std::min(g_jac_split, lds_elems) : lds_elems;
    std::vector<int> large;
    if (desc_host)
        for (int b = 0; b < n_blocks; ++b) {
            const htn_svd_block& D = desc_host[b];
            if (!(D.flags & HTN_SVD_QRCP) || D.n < 2) continue;
            const int gsx = D.m <= 16 * JAC_MAXEL ? 16 : (D.m <= 32 * JAC_MAXEL ? 32 : 64);
            const int mp = gsx * ((D.m + gsx - 1) / gsx);
            if ((int64_t)mp * D.n > large_min) large.push_back(b);
        }
    if (opts && opts->sweeps_used) *opts->sweeps_used = 0;
    if (large.empty()) {
        hipLaunchKernelGGL(k_jacobi_svd, dim3(n_blocks), dim3(JAC_THREADS), lds_elems * sizeof(double2), st, (double2*)G,
                           (double2*)Vj, S, desc, max_sweeps, tol, info_dev, lds_elems, (const int*)nullptr,
                           g_jac_cut * g_jac_cut);
        HIP_TRY(hipGetLastError());
        return 0;
    }

    const int nl = (int)large.size();
    // panel-pair work lists, one per round of the round-robin tournament over the column panels of each block;
    // built AFTER the QR has been enqueued (and, with a rank cut, after it has reported the ranks)
    std::vector<std::vector<JacPairItem>> rounds;
    std::vector<JacPairItem> intra;
    int max_mp = 0;
    size_t n_items_max = 0;
    for (int li = 0; li < nl; ++li) {
        const htn_svd_block& D = desc_host[large[li]];
        const int gsx = D.m <= 16 * JAC_MAXEL ? 16 : (D.m <= 32 * JAC_MAXEL ? 32 : 64);
        max_mp = std::max(max_mp, gsx * ((D.m + gsx - 1) / gsx));
        const int nb = (D.n + JAC_PANEL - 1) / JAC_PANEL, nbp = nb + (nb & 1);
        n_items_max += (size_t)(nbp - 1) * (nbp / 2) + (nb + 1) / 2;
    }
    auto build_rounds = [&](const int* n_eff) {
        const int w = JAC_PANEL;
        for (int li = 0; li < nl; ++li) {
            const int n = n_eff[li];
            if (n < 1) continue;
            const int nb = (n + w - 1) / w;
            const int nbp = nb + (nb & 1);
            if ((int)rounds.size() < nbp - 1) rounds.resize(nbp - 1);
            for (int r = 0; r < nbp - 1; ++r)
                for (int p = 0; p < nbp / 2; ++p) {
                    int a = p == 0 ? nbp - 1 : (r + p) % (nbp - 1);
                    int c = p == 0 ? r : (r + nbp - 1 - p) % (nbp - 1);
                    if (a >= nb || c >= nb) continue;
                    if (a > c) std::swap(a, c);
                    JacPairItem it = {li, a * w, std::min(w, n - a * w), c * w, std::min(w, n - c * w), {0, 0, 0}};
                    rounds[r].push_back(it);
                }
            // the pairs inside each panel, two panels per workgroup, once per outer sweep
            for (int a = 0; a < nb; a += 2) {
                const int c = a + 1;
                JacPairItem it = {li, a * w, std::min(w, n - a * w), c < nb ? c * w : 0,
                                  c < nb ? std::min(w, n - c * w) : 0, {1, 0, 0}};
                intra.push_back(it);
            }
        }
        // the intra-panel visit closes the sweep: one outer sweep fewer than with it in front (measured on graded
        // spectra and in the DMRG sweep; the cross visits leave the panels' own pairs slightly non-orthogonal)
        rounds.push_back(intra);
    };
    // device scratch layout: [large_ids | slot of every block | perm | zero2 | ratio | done | sweeps | items]
    const size_t off_ids = 0, off_slot = off_ids + sizeof(int) * nl, off_perm = off_slot + sizeof(int) * n_blocks;
    const size_t off_zero = off_perm + sizeof(int) * nl * 64 * JAC_MAXEL;
    const size_t off_ratio = (off_zero + sizeof(double) * nl + 7) / 8 * 8, off_done = off_ratio + 8 * nl;
    const size_t off_sw = off_done + sizeof(int) * nl, off_qsync = off_sw + sizeof(int) * nl;      // qsync: QR_SYNC_WORDS per block + failure word
    const size_t off_items = (off_qsync + sizeof(unsigned) * (QR_SYNC_WORDS * (size_t)nl + 8) + 31) / 32 * 32;
    const size_t dev_bytes = off_items + sizeof(JacPairItem) * n_items_max;
    JacScratch* jsp = nullptr;
    if (js_get(st, &jsp)) return 1;
    JacScratch& g_js = *jsp;
    if (js_reserve(g_js, dev_bytes, sizeof(JacPairItem) * n_items_max + 4 * (size_t)(nl + n_blocks) + 128,
                   (size_t)(max_sweeps + 1) + 2 * (size_t)nl + 16))
        return 1;
    char* d = (char*)g_js.dev;
    int* d_ids = (int*)(d + off_ids);
    int* d_slot = (int*)(d + off_slot);
    int* d_perm = (int*)(d + off_perm);
    double* d_zero = (double*)(d + off_zero);
    unsigned long long* d_ratio = (unsigned long long*)(d + off_ratio);
    int* d_done = (int*)(d + off_done);
    int* d_sw = (int*)(d + off_sw);
    unsigned* d_qsync = (unsigned*)(d + off_qsync);
    JacPairItem* d_items = (JacPairItem*)(d + off_items);
    // pinned staging (host -> device): [items | ids | slot of every block]; device -> host flags (their own coherent
    // block, read by the host only behind a completed event): [active count per sweep | rank per large block]
    char* h = (char*)g_js.pinned;
    JacPairItem* h_items = (JacPairItem*)h;
    int* h_ids = (int*)(h + sizeof(JacPairItem) * n_items_max);
    int* h_slot = h_ids + nl;                        // [ids | slot] contiguous like the device copy: ONE upload
    volatile int* h_active = (volatile int*)g_js.flags;
    volatile int* h_rank = h_active + (max_sweeps + 1);
    int* d_active = g_js.flags_dev;
    int* d_rank = d_active + (max_sweeps + 1);
    volatile int* h_ring_sw = h_rank + nl;           // ring path: outer sweeps per large block (0: the launch failed)
    int* d_ring_sw = d_rank + nl;
    for (int b = 0; b < n_blocks; ++b) h_slot[b] = -1;
    for (int li = 0; li < nl; ++li) h_slot[large[li]] = li;
    for (int li = 0; li < nl; ++li) h_ids[li] = large[li];
    for (int k = 0; k <= max_sweeps; ++k) h_active[k] = 1;
    // every small copy / fill is a separate blit launch on the stream: they are enqueued BEFORE the QR (nothing here
    // depends on it unless a rank cut sizes the tournament), merged where the regions are contiguous
    HIP_TRY(hipMemcpyAsync(d_ids, h_ids, sizeof(int) * (nl + n_blocks), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_ratio, 0, 16 * (size_t)nl + sizeof(unsigned) * (QR_SYNC_WORDS * (size_t)nl + 8), st));      // ratio (8 nl) | done (4 nl) | sweeps (4 nl) | qsync
    const double cut2 = g_jac_cut * g_jac_cut;
    std::vector<int> n_eff(nl);
    for (int li = 0; li < nl; ++li) n_eff[li] = desc_host[large[li]].n;
    std::vector<size_t> r_off;
    auto upload_rounds = [&]() -> int {
        build_rounds(n_eff.data());
        r_off.resize(rounds.size());
        size_t pos = 0;
        for (size_t r = 0; r < rounds.size(); ++r) {
            r_off[r] = pos;
            for (auto& it : rounds[r]) h_items[pos++] = it;
        }
        if (pos) HIP_TRY(hipMemcpyAsync(d_items, h_items, sizeof(JacPairItem) * pos, hipMemcpyHostToDevice, st));
        return 0;
    };
    // Ring path (default): all sweeps of the large blocks in one launch per batch of <= #CU workgroups.  The multi-launch
    // pair-visit path stays for blocks the ring cannot take (more CU slots than the chip has) and behind HTN_SVD_PAIRS=1.
    static const bool force_pairs = htn_env_flag("HTN_SVD_PAIRS");
    std::vector<RingItem> ring_items;                         // in GRID order of their launch (gaps: P = 0)
    std::vector<std::pair<int, int>> ring_batches;            // (first item, grid size) of each launch
    int ring_wgs = 0;                                         // workgroups with work (flag / id words are indexed densely)
    int64_t ring_mbox_elems = 0;
    // XCD-aware placement: a block's workgroups go to grid positions 8 s + x with ONE x -- the dispatcher has been seen to
    // deal position p to XCD p mod 8 (tools/gemm_prof.py), so they share an L2 and the kernel, after CHECKING the XCD ids it
    // reads at run time, hands panels over through that L2.  Speed only: a block whose workgroups find themselves on
    // different XCDs uses the placement-independent hand-off.  HTN_RING_NO_XCD=1: dense placement, as before.
    static const bool no_xcd = htn_env_flag("HTN_RING_NO_XCD");
    const int n_xcd = (!no_xcd && g_js.cu_count > 0 && g_js.cu_count % 8 == 0) ? 8 : 1;
    const int ring_cap = std::max(1, std::min(g_js.cu_count > 0 ? g_js.cu_count : 256, 256));
    // CU slots and panel width of every large block; batches of <= #CU workgroups (all workgroups of a launch must be
    // co-resident: they wait for each other).  false: some block needs more slots than the ring supports.
    auto plan_ring = [&]() -> bool {
        ring_items.clear();
        ring_batches.clear();
        ring_mbox_elems = 0;
        ring_wgs = 0;
        struct Blk {
            int li, P, w, mp;
        };
        std::vector<Blk> blks;
        for (int li = 0; li < nl; ++li) {
            const htn_svd_block& D = desc_host[large[li]];
            const int gsx = ring_gs(D.m);
            const int mp = gsx * ring_e(D.m);
            int wcap = std::min(RING_THREADS / gsx, RING_PANEL_ELEMS / mp);
            if (gsx == 16 && wcap > 16 && wcap < 32) wcap = 16;      // 16 pairs = one busy wave per SIMD; 17..31 would put two on one
            if (g_jac_split > 0) wcap = std::min(wcap, 3);       // test mode: small blocks still get several CU slots
            const int n = std::max(n_eff[li], 1);
            const int P = std::max(1, (n + 2 * wcap - 1) / (2 * wcap));
            if (wcap < 1 || P > RING_MAX_P || P > ring_cap) return false;
            blks.push_back({li, P, (n + 2 * P - 1) / (2 * P), mp});
        }
        std::stable_sort(blks.begin(), blks.end(), [](const Blk& a, const Blk& b) { return a.P > b.P; });
        std::vector<char> placed(blks.size(), 0);
        size_t left = blks.size();
        int nx = n_xcd;                                       // (a block with more slots than one XCD has CUs: dense placement)
        for (const Blk& B : blks)
            if (B.P > ring_cap / nx) nx = 1;
        const int lane_cap = ring_cap / nx;                     // CU slots of one XCD (of the chip when nx = 1)
        while (left) {
            const int first = (int)ring_items.size();
            std::vector<int> lane_used((size_t)nx, 0);
            struct Put {
                int q, x, s0;
            };
            std::vector<Put> puts;
            for (size_t q = 0; q < blks.size(); ++q) {
                if (placed[q]) continue;
                int x = 0;
                for (int y = 1; y < nx; ++y)
                    if (lane_used[y] < lane_used[x]) x = y;
                if (lane_used[x] + blks[q].P > lane_cap) continue;
                puts.push_back({(int)q, x, lane_used[x]});
                lane_used[x] += blks[q].P;
                placed[q] = 1;
                --left;
            }
            int depth = 0;
            for (int y = 0; y < nx; ++y) depth = std::max(depth, lane_used[y]);
            const int grid = depth * nx;
            ring_items.resize((size_t)first + grid, RingItem{0, 0, 0, 0, 0, 0, 0});
            for (const Put& pt : puts) {
                const Blk& B = blks[pt.q];
                for (int k = 0; k < B.P; ++k)
                    ring_items[(size_t)first + (size_t)(pt.s0 + k) * nx + pt.x] = {B.li, k, B.P, B.w, n_eff[B.li], ring_wgs, ring_mbox_elems};
                ring_mbox_elems += (int64_t)B.P * 4 * B.w * B.mp;
                ring_wgs += B.P;
            }
            ring_batches.push_back({first, grid});
        }
        return true;
    };
    bool use_ring = !force_pairs && plan_ring();
    if (cut2 <= 0.0 && !use_ring && upload_rounds()) return 1;
    // large blocks: pivoted QR on this stream, then the sweeps; the small blocks run their whole SVD beside
    // them on the forked stream and join before the call returns
    HIP_TRY(hipEventRecord(g_js.ev_fork, st));
    HIP_TRY(hipStreamWaitEvent(g_js.aux, g_js.ev_fork, 0));
    hipLaunchKernelGGL(k_jacobi_svd, dim3(n_blocks), dim3(JAC_THREADS), lds_elems * sizeof(double2), g_js.aux,
                       (double2*)G, (double2*)Vj, S, desc, max_sweeps, tol, info_dev, lds_elems, (const int*)d_slot,
                       g_jac_cut * g_jac_cut);
    HIP_TRY(hipEventRecord(g_js.ev_join, g_js.aux));
    {
        int max_m0 = 0;
        for (int li = 0; li < nl; ++li) max_m0 = std::max(max_m0, (int)desc_host[large[li]].pad);
        const size_t qr_panel_elems = (size_t)16 * (((max_m0 + 15) & ~15) + 1);
        int min_m0 = 1 << 30;
        for (int li = 0; li < nl; ++li) min_m0 = std::min(min_m0, (int)desc_host[large[li]].pad);
        // + 64 KiB for the partial tiles of the cooperative trailing update, when some block is small enough to use them and the
        // panel of the largest leaves the room
        const bool coop = ((min_m0 + 15) & ~15) <= 256 && qr_panel_elems * sizeof(double2) + 65536 + 30720 <= 163840;
        const size_t qr_lds = qr_panel_elems * sizeof(double2) + (coop ? 65536 : 0);
        // workgroups per block: the master + helpers for the trailing update (all co-resident: they wait for each other)
        int max_n0 = 0;
        for (int li = 0; li < nl; ++li) max_n0 = std::max(max_n0, (int)desc_host[large[li]].m);
        static const bool qr_single = htn_env_flag("HTN_QR_SINGLE");
        // Measured (tools/ring_prof.py, HTN_QR_PROF): in the placement-independent form every shared byte goes to memory and
        // comes back from memory (sc1), so a chunk's operand loads wait ~2 us each instead of an L2 hit: 202 x 202 blocks LOSE
        // (trailing 483 -> 655 us with four workgroups), 400 x 400 blocks gain (3.0 -> 2.3 ms).
        static const int env_nw = getenv("HTN_QR_NW") ? atoi(getenv("HTN_QR_NW")) : 0;      // (experiments: helpers at any size)
        // Helpers (profiles/r03_ring_qr_phase_times.txt): from 160 columns on always (202 x 202: 855 us alone, 697 with four
        // workgroups through memory, 623 through one XCD's L2; 400 x 400: 3.9 -> 2.5 ms); below, only while the kernels keep
        // finding a block's workgroups on one XCD (100 x 100: 266 -> 253 us through the L2).
        int NW = qr_single ? 1 : ((max_n0 >= 160 || g_js.xcd_local) ? 4 : 1);
        if (env_nw > 0 && !qr_single) NW = std::min(env_nw, 4);
        // placement: the workgroups of a block at grid positions of one residue mod 8 (see k_qr_large); the gaps count
        // against the co-residency bound like everything else
        const int cus = std::max(1, std::min(g_js.cu_count > 0 ? g_js.cu_count : 256, 256));
        const int qnx = (NW > 1 && n_xcd > 1) ? n_xcd : 1;
        const int nl_pad = (nl + qnx - 1) / qnx * qnx;
        NW = std::max(1, std::min(NW, cus / std::max(nl_pad, 1)));
        if (NW > 1) {
            const size_t need = (size_t)nl * QR_BOX_BYTES;
            if (need > g_js.qr_box_bytes) {
                if (g_js.qr_box) HIP_TRY(hipFree(g_js.qr_box));
                g_js.qr_box = nullptr, g_js.qr_box_bytes = 0;
                HIP_TRY(hipMalloc(&g_js.qr_box, need * 2));
                g_js.qr_box_bytes = need * 2;
                if (htn_debug_poison()) HIP_TRY(hipMemset(g_js.qr_box, 0xFF, need * 2));
            }
        }
        const int qgrid_nx = NW > 1 ? qnx : 1;
        hipLaunchKernelGGL(k_qr_large, dim3((NW > 1 ? nl_pad : nl) * NW), dim3(JAC_THREADS), qr_lds, st, (double2*)G, (double2*)Vj, desc, d_ids,
                           d_perm, d_zero, cut2, d_rank, NW, (char*)g_js.qr_box, d_qsync, nl, qgrid_nx, coop ? (int)qr_panel_elems : 0);
    }
    if (cut2 > 0.0) {        // the tournament is sized by the ranks the QR found: wait for them (one sync per call)
        HIP_TRY(hipEventRecord(g_js.ev_sweep[0], st));
        HIP_TRY(htn_event_spin(g_js.ev_sweep[0]));
        for (int li = 0; li < nl; ++li) n_eff[li] = std::min(n_eff[li], (int)h_rank[li]);
        if (use_ring) use_ring = plan_ring();
        if (!use_ring && upload_rounds()) return 1;
    }
    if (use_ring) {
        const int n_wg = ring_wgs, n_items = (int)ring_items.size();
        // sync block (32-bit words): [flags: 2 per workgroup | arrivals: nl x max_sweeps | failure word | XCD ids: 1 per
        // workgroup | pad] then the 64-bit maxima, nl x max_sweeps; zeroed as ONE block that starts its allocation and is a
        // multiple of 16 bytes
        const int arrive_off = 2 * n_wg, fail_off = arrive_off + nl * max_sweeps, xcc_off = fail_off + 1;
        const int conv_off = (xcc_off + n_wg + 3) & ~3;
        const size_t sync_bytes = ((size_t)conv_off * 4 + (size_t)nl * max_sweeps * 8 + 15) & ~(size_t)15;
        if (sync_bytes > g_js.ring_sync_bytes) {
            if (g_js.ring_sync) HIP_TRY(hipFree(g_js.ring_sync));
            g_js.ring_sync = nullptr, g_js.ring_sync_bytes = 0;
            HIP_TRY(hipMalloc(&g_js.ring_sync, sync_bytes * 2));
            g_js.ring_sync_bytes = sync_bytes * 2;
        }
        const size_t mbox_bytes = sizeof(double2) * (size_t)std::max<int64_t>(ring_mbox_elems, 1);
        if (mbox_bytes > g_js.ring_mbox_bytes) {
            if (g_js.ring_mbox) HIP_TRY(hipFree(g_js.ring_mbox));
            g_js.ring_mbox = nullptr, g_js.ring_mbox_bytes = 0;
            HIP_TRY(hipMalloc(&g_js.ring_mbox, mbox_bytes + mbox_bytes / 2));
            g_js.ring_mbox_bytes = mbox_bytes + mbox_bytes / 2;
            if (htn_debug_poison()) HIP_TRY(hipMemset(g_js.ring_mbox, 0xFF, g_js.ring_mbox_bytes));
        }
        if ((size_t)n_items > g_js.ring_items_cap) {
            if (g_js.ring_items) HIP_TRY(hipFree(g_js.ring_items));
            if (g_js.ring_items_h) HIP_TRY(hipHostFree(g_js.ring_items_h));
            g_js.ring_items = g_js.ring_items_h = nullptr, g_js.ring_items_cap = 0;
            HIP_TRY(hipMalloc(&g_js.ring_items, sizeof(RingItem) * 2 * n_items));
            HIP_TRY(hipHostMalloc(&g_js.ring_items_h, sizeof(RingItem) * 2 * n_items, hipHostMallocDefault));
            g_js.ring_items_cap = 2 * (size_t)n_items;
        }
        memcpy(g_js.ring_items_h, ring_items.data(), sizeof(RingItem) * n_items);
        for (int li = 0; li < nl; ++li) h_ring_sw[li] = 0;
        HIP_TRY(hipMemcpyAsync(g_js.ring_items, g_js.ring_items_h, sizeof(RingItem) * n_items, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(g_js.ring_sync, 0, sync_bytes, st));
        RingArgs ra;
        ra.Vj = (double2*)Vj, ra.G = (double2*)G, ra.S = S, ra.desc = desc, ra.large_ids = d_ids;
        ra.perm = d_perm, ra.zero2 = d_zero, ra.mbox = (double2*)g_js.ring_mbox, ra.sync = (unsigned*)g_js.ring_sync;
        ra.arrive_off = arrive_off, ra.fail_off = fail_off, ra.conv_off = conv_off, ra.max_sweeps = max_sweeps, ra.tol = tol;
        ra.xcc_off = xcc_off;
        ra.info = info_dev, ra.sweeps_out = d_ring_sw;
        for (auto& bt : ring_batches) {
            ra.items = (const RingItem*)g_js.ring_items + bt.first;
            hipLaunchKernelGGL(k_jacobi_ring, dim3((unsigned)bt.second), dim3(RING_THREADS), 2 * RING_PANEL_ELEMS * sizeof(double2), st, ra);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamWaitEvent(st, g_js.ev_join, 0));
        HIP_TRY(htn_stream_spin(st));      // the staging blocks are reused by the next call; the sweep counts are read below
        int used = 0;
        for (int li = 0; li < nl; ++li) {
            if (h_rank[li] < 0) return fail_msg("htn_jacobi_svd_z: a hand-off of the multi-workgroup QR timed out");
            if (h_ring_sw[li] <= 0 && n_eff[li] >= 2) return fail_msg("htn_jacobi_svd_z: a hand-off of the ring Jacobi kernel timed out");
            used = std::max(used, (int)h_ring_sw[li] % 1000);
        }
        {   // what the kernel saw of the placement steers the NEXT call's choice of QR helpers (below 160 columns they only pay
            // when the block's workgroups share an L2)
            int all_local = n_xcd > 1 ? 1 : 0;
            for (int li = 0; li < nl; ++li)
                if (n_eff[li] >= 2 && h_ring_sw[li] < 1000) all_local = 0;
            g_js.xcd_local = all_local;
        }
        if (opts && opts->sweeps_used) *opts->sweeps_used = used;
        return 0;
    }
    const size_t gram_lds_bytes = (size_t)(16 * (max_mp + 1) + JG_GU_ELEMS) * sizeof(double2);
    // sweeps are enqueued one ahead of the host's knowledge (depth-1 pipeline, like htn_lanczos_z): the device
    // decides convergence itself (k_jacobi_check), the host only learns when to stop enqueuing
    const double thr = std::max(tol * tol, 0.1 * tol);          // quadratic convergence, see jacobi_sweeps
    auto enqueue_sweep = [&](int sweep) {
        for (size_t r = 0; r < rounds.size(); ++r)
            if (!rounds[r].empty())
                hipLaunchKernelGGL(k_jacobi_pairs_gram, dim3((unsigned)rounds[r].size()), dim3(256), gram_lds_bytes, st,
                                   (double2*)Vj, desc, d_ids, d_items + r_off[r], d_zero, d_ratio, d_done, tol, 1);
        hipLaunchKernelGGL(k_jacobi_check, dim3(1), dim3(64), 0, st, d_ratio, d_done, d_sw, nl, thr,
                           d_active + sweep);
        return hipEventRecord(g_js.ev_sweep[sweep & 1], st);
    };
    // Speculation is bounded by the caller's expectation (htn_svd_opts.sweeps_hint, normally what the previous update of
    // the same bond needed): the sweep expected to be the last is NOT followed by a speculative one -- an outer sweep that
    // finds every block done still costs its launches (26 x 4.5 us at chi = 1024).  A wrong hint costs one host round trip
    // per extra sweep instead.
    const int hint = opts && opts->sweeps_hint > 0 ? opts->sweeps_hint : 0;
    int enq = 0, used = 0;
    if (max_sweeps > 0) {
        HIP_TRY(enqueue_sweep(0));
        enq = 1;
    }
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        const bool expect_last = hint > 0 && sweep + 1 >= hint;
        if (sweep + 1 < max_sweeps && !expect_last && enq == sweep + 1) {
            HIP_TRY(enqueue_sweep(sweep + 1));
            ++enq;
        }
        HIP_TRY(htn_event_spin(g_js.ev_sweep[sweep & 1]));
        used = sweep + 1;
        if (h_active[sweep] == 0) break;
        if (sweep + 1 < max_sweeps && enq == sweep + 1) {
            HIP_TRY(enqueue_sweep(sweep + 1));
            ++enq;
        }
    }
    if (opts && opts->sweeps_used) *opts->sweeps_used = used;
    HIP_TRY(hipStreamWaitEvent(st, g_js.ev_join, 0));
    hipLaunchKernelGGL(k_jacobi_finish, dim3(nl), dim3(JAC_THREADS), 0, st, (double2*)G, (const double2*)Vj, S, desc,
                       d_ids, d_perm, d_sw, d_done, info_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(htn_stream_spin(st));      // the pinned staging block is reused by the next call
    return 0;
}

// ----------------------------------------------------------------------------------------------
// batched strided copy / gather / scale
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_batched_copy(double2* __restrict__ dst, const double2* __restrict__ src,
                                                      const int32_t* __restrict__ idx,
                                                      const double* __restrict__ scl,
                                                      const htn_copy_item* __restrict__ items, double gscale) {
    // grid = (items, COPY_SPLIT): the largest item (a 200 x 250 block) would otherwise occupy ONE workgroup for
    // ~80 us while the other CUs idle; the element range of every item is dealt over gridDim.y workgroups
    const htn_copy_item I = items[blockIdx.x];
    const int total = I.rows * I.cols;
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < total; e += blockDim.x * gridDim.y) {
        const int i = e % I.rows, j = e / I.rows;
        // gathered source index along gather_dim of dst
        int gi = i, gj = j;
        if (I.idx_off >= 0) {
            if (I.gather_dim == 0) gi = idx[I.idx_off + i];
            else gj = idx[I.idx_off + j];
        }
        double2 v;
        if (I.op == HTN_OP_N) v = src[I.src_off + (int64_t)gi + (int64_t)gj * I.lds];
        else {
            v = src[I.src_off + (int64_t)gj + (int64_t)gi * I.lds];   // dst(i,j) = conj(src(gj, gi))
            v.y = -v.y;
        }
        double f = gscale;
        if (I.scale_dim >= 0 && I.scl_off >= 0) {
            const double sv = scl[I.scl_off + (I.scale_dim == 0 ? gi : gj)];
            f = I.inv_norm ? (sv > 0.0 ? f / sv : 0.0) : f * sv;
        }
        dst[I.dst_off + (int64_t)i + (int64_t)j * I.ldd] = make_double2(v.x * f, v.y * f);
    }
}

extern "C" int htn_batched_copy_z(void* dst, const void* src, const int32_t* idx, const double* scl,
                                  const htn_copy_item* items, int32_t n_items, double global_scale,
                                  void* stream) {
    if (n_items <= 0) return 0;
    hipLaunchKernelGGL(k_batched_copy, dim3(n_items, 16), dim3(256), 0, (hipStream_t)stream, (double2*)dst,
                       (const double2*)src, idx, scl, items, global_scale);
    HIP_TRY(hipGetLastError());
    return 0;
}
