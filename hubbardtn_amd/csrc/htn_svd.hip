// batched Jacobi SVD + strided copy kernels (libhubbardtn_hip.so)
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
//   * sweeps stop early through the quadratic convergence of cyclic Jacobi: once the largest
//     |a.b| / (|a||b|) seen in a sweep is below 1e-8, ONE more sweep brings it below 1e-16, so the final
//     "checking" sweep of the textbook loop is skipped.
#define JAC_THREADS 1024
#define JAC_MAXEL 8           // column elements per lane kept in registers: m <= GS * JAC_MAXEL

template <int GS>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = GS / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int GS>
__device__ __forceinline__ double jacobi_pair(double2* ga, double2* gb, double2* __restrict__ va,
                                              double2* __restrict__ vb, int m, int n, int sub, double tol) {
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
    const double g = sqrt(gr * gr + gi * gi);
    const double den = sqrt(aa * bb);
    if (g == 0.0 || g <= tol * den) return den > 0.0 ? g / den : 0.0;
    const double zeta = (bb - aa) / (2.0 * g);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t);
    const double s = c * t;
    // b~ = exp(-i phi) b with exp(i phi) = gamma / |gamma|;  a' = c a - s b~ ; b' = s a + c b~
    const double pr = gr / g, pi = -gi / g;   // exp(-i phi)
#pragma unroll
    for (int e = 0; e < JAC_MAXEL; ++e) {
        const int i = sub + GS * e;
        if (i < m) {
            const double2 bt = make_double2(pr * b[e].x - pi * b[e].y, pr * b[e].y + pi * b[e].x);
            ga[i] = make_double2(c * a[e].x - s * bt.x, c * a[e].y - s * bt.y);
            gb[i] = make_double2(s * a[e].x + c * bt.x, s * a[e].y + c * bt.y);
        }
    }
    for (int i = sub; i < n; i += GS) {
        const double2 x = va[i], y = vb[i];
        const double2 yt = make_double2(pr * y.x - pi * y.y, pr * y.y + pi * y.x);
        va[i] = make_double2(c * x.x - s * yt.x, c * x.y - s * yt.y);
        vb[i] = make_double2(s * x.x + c * yt.x, s * x.y + c * yt.y);
    }
    return g / den;
}

template <int GS>
__device__ __forceinline__ int jacobi_sweeps(double2* g, double2* __restrict__ v, int m, int n, int max_sweeps,
                                             double tol, double* s_ratio, int tid) {
    const int grp = tid / GS, sub = tid % GS;
    const int ngroups = JAC_THREADS / GS;
    const int np = n + (n & 1);      // padded to even; index np-1 == n is a bye when n is odd
    int sweeps = 0;
    bool done = (n < 2);
    bool last = false;
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
                                                      v + (int64_t)hi * n, m, n, sub, tol);
                    ratio = rr > ratio ? rr : ratio;
                }
            }
            __syncthreads();
        }
        // workgroup max of the non-negative ratio: doubles order like their bit patterns
        if (sub == 0 && ratio > 0.0) atomicMax((unsigned long long*)s_ratio, (unsigned long long)__double_as_longlong(ratio));
        __syncthreads();
        const double mx = *s_ratio;
        ++sweeps;
        done = last || mx <= tol;
        last = mx < 1e-8;            // quadratic convergence: the next sweep is the final one
        __syncthreads();
    }
    return done ? sweeps : -sweeps;
}

__global__ __launch_bounds__(JAC_THREADS) void k_jacobi_svd(double2* __restrict__ G, double2* __restrict__ Vj,
                                                            double* __restrict__ S,
                                                            const htn_svd_block* __restrict__ desc,
                                                            int max_sweeps, double tol, int* __restrict__ info,
                                                            int lds_elems) {
    extern __shared__ double2 g_lds[];
    __shared__ double s_ratio;
    const htn_svd_block D = desc[blockIdx.x];
    const int m = D.m, n = D.n;
    double2* __restrict__ gglob = G + D.g_off;
    double2* __restrict__ v = Vj + D.v_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = JAC_THREADS / 64;
    const bool in_lds = (int64_t)m * n <= lds_elems;
    // V = identity ; stage G into LDS when it fits
    for (int idx = tid; idx < n * n; idx += JAC_THREADS) {
        const int i = idx % n, j = idx / n;
        v[idx] = make_double2(i == j ? 1.0 : 0.0, 0.0);
    }
    if (in_lds)
        for (int idx = tid; idx < m * n; idx += JAC_THREADS) g_lds[idx] = gglob[idx];
    __syncthreads();
    double2* g = in_lds ? (double2*)g_lds : gglob;
    int sw;
    if (m <= 16 * JAC_MAXEL) sw = jacobi_sweeps<16>(g, v, m, n, max_sweeps, tol, &s_ratio, tid);
    else if (m <= 32 * JAC_MAXEL) sw = jacobi_sweeps<32>(g, v, m, n, max_sweeps, tol, &s_ratio, tid);
    else sw = jacobi_sweeps<64>(g, v, m, n, max_sweeps, tol, &s_ratio, tid);
    __syncthreads();
    // column norms (+ write back)
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

extern "C" int htn_jacobi_svd_z(void* G, void* Vj, double* S, const htn_svd_block* desc, int32_t n_blocks,
                                int32_t max_m_host, int32_t max_sweeps, double tol, int32_t* info_dev,
                                void* stream) {
    if (n_blocks <= 0) return 0;
    if (max_m_host > 64 * JAC_MAXEL) return fail_msg("htn_jacobi_svd_z: block taller than 512 rows");
    // dynamic LDS window for the matrix: 144 KiB leaves room for the static shared variables
    const int lds_elems = 9216;     // complex128 elements = 144 KiB
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_jacobi_svd, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    lds_elems * (int)sizeof(double2)));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_jacobi_svd, dim3(n_blocks), dim3(JAC_THREADS), lds_elems * sizeof(double2),
                       (hipStream_t)stream, (double2*)G, (double2*)Vj, S, desc, max_sweeps, tol, info_dev, lds_elems);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ----------------------------------------------------------------------------------------------
// batched strided copy / gather / scale
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_batched_copy(double2* __restrict__ dst, const double2* __restrict__ src,
                                                      const int32_t* __restrict__ idx,
                                                      const double* __restrict__ scl,
                                                      const htn_copy_item* __restrict__ items, double gscale) {
    const htn_copy_item I = items[blockIdx.x];
    const int total = I.rows * I.cols;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
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
    hipLaunchKernelGGL(k_batched_copy, dim3(n_items), dim3(256), 0, (hipStream_t)stream, (double2*)dst,
                       (const double2*)src, idx, scl, items, global_scale);
    HIP_TRY(hipGetLastError());
    return 0;
}
