// libhubbardtn_hip.so -- hand-written CDNA4 (gfx950) kernels behind the C ABI of
// include/hubbardtn_hip.h.  No CPU path exists in this library: every entry point launches a
// HIP kernel.  See DESIGN.md for the data layout and the roofline that bounds each kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "hubbardtn_hip.h"

// ----------------------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return 1;
}
static int fail_msg(const char* what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return 1;
}
#define HIP_TRY(x)                                   \
    do {                                             \
        hipError_t _e = (x);                         \
        if (_e != hipSuccess) return fail(#x, _e);   \
    } while (0)

extern "C" const char* htn_last_error(void) { return g_err; }
extern "C" int htn_abi_version(void) { return HTN_ABI_VERSION; }

extern "C" int htn_device_init(int device, char* name_host, int* cu_count_host) {
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name_host) {
        strncpy(name_host, p.gcnArchName, 255);
        name_host[255] = 0;
    }
    if (cu_count_host) *cu_count_host = p.multiProcessorCount;
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail_msg("htn_device_init: this library is built for gfx950 (MI355X) only");
    return 0;
}

// ----------------------------------------------------------------------------------------------
// grouped, segmented complex128 GEMM on v_mfma_f64_16x16x4_f64
// ----------------------------------------------------------------------------------------------
// One workgroup (4 waves) owns one <=32x32 output tile and walks the tile's segment list.  Each
// segment contributes alpha * op(A) * op(B); alpha is folded into A while staging, so all segments
// accumulate into the same MFMA accumulators.  K is consumed in slabs of 16 staged through LDS as
// separate re / im planes.  Each wave owns a 16x16 quadrant.  The MFMA computes the TRANSPOSED
// quadrant (operand roles swapped) so that consecutive lanes hold consecutive rows of the
// column-major output and the epilogue stores are 256-byte runs.
//
// f64 MFMA lane maps (cdna_hip_programming.md section 3):
//   A operand: lane l holds Aop[i = l & 15][k = l >> 4]     B operand: Bop[k = l >> 4][j = l & 15]
//   D: lane l, reg r holds D[row = (l >> 4) + 4 r][col = l & 15]
// With Aop[i][k] = B[k][c0 + i] and Bop[k][j] = A[r0 + j][k]:  D[i][j] = C[r0 + j][c0 + i].
typedef double d4 __attribute__((ext_vector_type(4)));

struct BufTable {
    double2* p[HTN_MAX_BUFS];
};

#define KB 16         // K slab
#define LDS_LD 48     // padded leading dimension (doubles) of a 32-wide slab row: 48 = 16 mod 32 keeps
                      // the two k-rows read by one 32-lane group on disjoint banks (ds_read_b64)
// Row kk is additionally rotated by kk inside its 32 doubles: a k-contiguous staging pass (16 lanes
// writing 16 different kk at one idx) then hits 16 different banks instead of one.
#define LDS_AT(kk, idx) ((kk) * LDS_LD + (((idx) + (kk)) & 31))

__device__ __forceinline__ double2 ld_a(const double2* __restrict__ A, int op, int lda, int r, int kk) {
    // element (r, kk) of op(A)
    if (op == HTN_OP_N) return A[(int64_t)r + (int64_t)kk * lda];
    double2 v = A[(int64_t)kk + (int64_t)r * lda];
    if (op == HTN_OP_C) v.y = -v.y;
    return v;
}

__global__ __launch_bounds__(256) void k_grouped_gemm_z(BufTable bufs, const htn_tile* __restrict__ tiles,
                                                        const htn_seg* __restrict__ segs) {
    __shared__ double sA_re[KB * LDS_LD];
    __shared__ double sA_im[KB * LDS_LD];
    __shared__ double sB_re[KB * LDS_LD];
    __shared__ double sB_im[KB * LDS_LD];

    const htn_tile T = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    d4 acc_re = {0.0, 0.0, 0.0, 0.0};
    d4 acc_im = {0.0, 0.0, 0.0, 0.0};
    // COPY segments are added outside the MFMA accumulators, in the epilogue layout:
    // this lane's outputs are C[r0 + l15][c0 + l4 + 4 r], r = 0..3
    const int orow = wr * 16 + l15;
    double cp_re[4] = {0.0, 0.0, 0.0, 0.0};
    double cp_im[4] = {0.0, 0.0, 0.0, 0.0};

    for (int s = 0; s < T.seg_count; ++s) {
        const htn_seg S = segs[T.seg_begin + s];
        const double2* __restrict__ Bp = bufs.p[S.buf_b] + S.b_off;
        if (S.type == HTN_SEG_COPY) {
            if (orow < T.m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ocol = wc * 16 + l4 + 4 * r;
                    if (ocol < T.n) {
                        const double2 v = Bp[(int64_t)(T.row0 + orow) + (int64_t)(T.col0 + ocol) * S.ldb];
                        cp_re[r] += S.alpha_re * v.x - S.alpha_im * v.y;
                        cp_im[r] += S.alpha_re * v.y + S.alpha_im * v.x;
                    }
                }
            }
            continue;
        }
        const double2* __restrict__ Ap = bufs.p[S.buf_a] + S.a_off;
        const int K = S.k;
        for (int k0 = 0; k0 < K; k0 += KB) {
            // ---- stage A slab: rows T.row0 .. +32, k0 .. +16 ; 512 elements, 2 per thread ----
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                int r, kk;
                if (S.op_a == HTN_OP_N) {       // rows are contiguous in memory
                    r = tid & 31;
                    kk = (tid >> 5) + 8 * it;
                } else {                        // k is contiguous in memory
                    kk = tid & 15;
                    r = (tid >> 4) + 16 * it;
                }
                double2 v = make_double2(0.0, 0.0);
                if (r < T.m && k0 + kk < K) {
                    const double2 a = ld_a(Ap, S.op_a, S.lda, T.row0 + r, k0 + kk);
                    v.x = S.alpha_re * a.x - S.alpha_im * a.y;
                    v.y = S.alpha_re * a.y + S.alpha_im * a.x;
                }
                sA_re[LDS_AT(kk, r)] = v.x;
                sA_im[LDS_AT(kk, r)] = v.y;
            }
            // ---- stage B slab: k0 .. +16, cols T.col0 .. +32 ----
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                int c, kk;
                if (S.op_b == HTN_OP_N) {       // k contiguous
                    kk = tid & 15;
                    c = (tid >> 4) + 16 * it;
                } else {                        // columns contiguous
                    c = tid & 31;
                    kk = (tid >> 5) + 8 * it;
                }
                double2 v = make_double2(0.0, 0.0);
                if (c < T.n && k0 + kk < K) {
                    if (S.op_b == HTN_OP_N) {
                        v = Bp[(int64_t)(k0 + kk) + (int64_t)(T.col0 + c) * S.ldb];
                    } else {
                        v = Bp[(int64_t)(T.col0 + c) + (int64_t)(k0 + kk) * S.ldb];
                        if (S.op_b == HTN_OP_C) v.y = -v.y;
                    }
                }
                sB_re[LDS_AT(kk, c)] = v.x;
                sB_im[LDS_AT(kk, c)] = v.y;
            }
            __syncthreads();
            const int kleft = K - k0;
            const int ksteps = kleft >= KB ? KB / 4 : (kleft + 3) >> 2;
            for (int ks = 0; ks < ksteps; ++ks) {
                const int kk = ks * 4 + l4;
                const double b_re = sB_re[LDS_AT(kk, wc * 16 + l15)];   // Aop[i][k] = B[k][c0+i]
                const double b_im = sB_im[LDS_AT(kk, wc * 16 + l15)];
                const double a_re = sA_re[LDS_AT(kk, wr * 16 + l15)];   // Bop[k][j] = A[r0+j][k]
                const double a_im = sA_im[LDS_AT(kk, wr * 16 + l15)];
                acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(b_re, a_re, acc_re, 0, 0, 0);
                acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(-b_im, a_im, acc_re, 0, 0, 0);
                acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(b_re, a_im, acc_im, 0, 0, 0);
                acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(b_im, a_re, acc_im, 0, 0, 0);
            }
            __syncthreads();
        }
    }
    // ---- epilogue: lane holds C[r0 + l15][c0 + l4 + 4 r] ----
    if (orow < T.m) {
        double2* __restrict__ Cp = bufs.p[T.buf_c] + T.c_off;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ocol = wc * 16 + l4 + 4 * r;
            if (ocol < T.n)
                Cp[(int64_t)(T.row0 + orow) + (int64_t)(T.col0 + ocol) * T.ldc] =
                    make_double2(acc_re[r] + cp_re[r], acc_im[r] + cp_im[r]);
        }
    }
}

extern "C" int htn_grouped_gemm_z(const void* const* bufs_host, const htn_tile* tiles, int32_t n_tiles,
                                  const htn_seg* segs, void* stream) {
    if (n_tiles <= 0) return 0;
    BufTable bt;
    for (int i = 0; i < HTN_MAX_BUFS; ++i) bt.p[i] = (double2*)bufs_host[i];
    hipLaunchKernelGGL(k_grouped_gemm_z, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, bt, tiles, segs);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ----------------------------------------------------------------------------------------------
// Krylov vector algebra (HBM / L2 bound)
// ----------------------------------------------------------------------------------------------
#define DOT_BLOCKS 128
#define DOT_THREADS 256

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// partial[i * DOT_BLOCKS + b] = sum over this block's slice of conj(V_i) * w
__global__ __launch_bounds__(DOT_THREADS) void k_dots_partial(const double2* __restrict__ V, int64_t ldv,
                                                              int nvec, const double2* __restrict__ w,
                                                              int64_t n, double2* __restrict__ partial) {
    __shared__ double red[2][DOT_THREADS / 64];
    const int tid = threadIdx.x;
    const int64_t per = (n + DOT_BLOCKS - 1) / DOT_BLOCKS;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    for (int i = 0; i < nvec; ++i) {
        const double2* __restrict__ Vi = V + (int64_t)i * ldv;
        double sr = 0.0, si = 0.0;
        for (int64_t j = lo + tid; j < hi; j += DOT_THREADS) {
            const double2 a = Vi[j], b = w[j];
            sr += a.x * b.x + a.y * b.y;
            si += a.x * b.y - a.y * b.x;
        }
        sr = wave_sum(sr);
        si = wave_sum(si);
        if ((tid & 63) == 0) {
            red[0][tid >> 6] = sr;
            red[1][tid >> 6] = si;
        }
        __syncthreads();
        if (tid == 0) {
            double tr = 0.0, ti = 0.0;
            for (int q = 0; q < DOT_THREADS / 64; ++q) {
                tr += red[0][q];
                ti += red[1][q];
            }
            partial[(int64_t)i * DOT_BLOCKS + blockIdx.x] = make_double2(tr, ti);
        }
        __syncthreads();
    }
}

__global__ void k_dots_reduce(const double2* __restrict__ partial, int nvec, double2* __restrict__ out) {
    const int i = blockIdx.x;
    const int lane = threadIdx.x;   // 64 threads
    double sr = 0.0, si = 0.0;
    for (int b = lane; b < DOT_BLOCKS; b += 64) {
        const double2 p = partial[(int64_t)i * DOT_BLOCKS + b];
        sr += p.x;
        si += p.y;
    }
    sr = wave_sum(sr);
    si = wave_sum(si);
    if (lane == 0) out[i] = make_double2(sr, si);
}

extern "C" int64_t htn_dots_scratch_elems(int32_t nvec) { return (int64_t)nvec * DOT_BLOCKS; }

extern "C" int htn_dots_z(const void* V, int64_t ldv, int32_t nvec, const void* w, int64_t n, void* out,
                          void* scratch, void* stream) {
    if (nvec <= 0) return 0;
    hipLaunchKernelGGL(k_dots_partial, dim3(DOT_BLOCKS), dim3(DOT_THREADS), 0, (hipStream_t)stream,
                       (const double2*)V, ldv, nvec, (const double2*)w, n, (double2*)scratch);
    hipLaunchKernelGGL(k_dots_reduce, dim3(nvec), dim3(64), 0, (hipStream_t)stream, (const double2*)scratch,
                       nvec, (double2*)out);
    HIP_TRY(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void k_axpys(double2* __restrict__ w, const double2* __restrict__ V,
                                               int64_t ldv, int nvec, const double2* __restrict__ coef,
                                               double sign, int64_t n) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        double sr = 0.0, si = 0.0;
        for (int i = 0; i < nvec; ++i) {
            const double2 c = coef[i];
            const double2 v = V[(int64_t)i * ldv + j];
            sr += c.x * v.x - c.y * v.y;
            si += c.x * v.y + c.y * v.x;
        }
        double2 x = w[j];
        x.x += sign * sr;
        x.y += sign * si;
        w[j] = x;
    }
}

extern "C" int htn_axpys_z(void* w, const void* V, int64_t ldv, int32_t nvec, const void* coef, double sign,
                           int64_t n, void* stream) {
    if (nvec <= 0 || n <= 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_axpys, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (double2*)w, (const double2*)V,
                       ldv, nvec, (const double2*)coef, sign, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void k_scale_inv_sqrt(double2* __restrict__ dst, const double2* __restrict__ src,
                                                        const double2* __restrict__ nrm2, int64_t n) {
    const double s = 1.0 / sqrt(nrm2[0].x);
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = src[j];
        dst[j] = make_double2(v.x * s, v.y * s);
    }
}

extern "C" int htn_scale_inv_sqrt_z(void* dst, const void* src, const void* nrm2, int64_t n, void* stream) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_scale_inv_sqrt, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (double2*)dst,
                       (const double2*)src, (const double2*)nrm2, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ----------------------------------------------------------------------------------------------
// batched one-sided (Hestenes) Jacobi SVD, one workgroup per coupled-sector block
// ----------------------------------------------------------------------------------------------
// Columns live in global memory (L2 resident: the largest block is a few hundred KB); one wave
// owns one column pair per pass and keeps both columns in registers while it rotates them.
// Round-robin (circle) ordering gives n-1 rounds of n/2 disjoint pairs per sweep; rounds are
// separated by a workgroup barrier.
#define JAC_THREADS 1024
#define JAC_MAXEL 8           // column elements per lane kept in registers: m <= 64 * JAC_MAXEL = 512

__device__ __forceinline__ void jacobi_pair(double2* __restrict__ ga, double2* __restrict__ gb,
                                            double2* __restrict__ va, double2* __restrict__ vb, int m, int n,
                                            int lane, double tol, int* rotated) {
    double2 a[JAC_MAXEL], b[JAC_MAXEL];
    double aa = 0.0, bb = 0.0, gr = 0.0, gi = 0.0;
#pragma unroll
    for (int e = 0; e < JAC_MAXEL; ++e) {
        const int i = lane + 64 * e;
        if (i < m) {
            a[e] = ga[i];
            b[e] = gb[i];
            aa += a[e].x * a[e].x + a[e].y * a[e].y;
            bb += b[e].x * b[e].x + b[e].y * b[e].y;
            gr += a[e].x * b[e].x + a[e].y * b[e].y;      // conj(a) * b
            gi += a[e].x * b[e].y - a[e].y * b[e].x;
        }
    }
    aa = wave_sum(aa);
    bb = wave_sum(bb);
    gr = wave_sum(gr);
    gi = wave_sum(gi);
    const double g = sqrt(gr * gr + gi * gi);
    if (g <= tol * sqrt(aa * bb) || g == 0.0) return;
    *rotated = 1;
    const double zeta = (bb - aa) / (2.0 * g);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t);
    const double s = c * t;
    // b~ = exp(-i phi) b with exp(i phi) = gamma / |gamma|;  a' = c a - s b~ ; b' = s a + c b~
    const double pr = gr / g, pi = -gi / g;   // exp(-i phi)
#pragma unroll
    for (int e = 0; e < JAC_MAXEL; ++e) {
        const int i = lane + 64 * e;
        if (i < m) {
            const double2 bt = make_double2(pr * b[e].x - pi * b[e].y, pr * b[e].y + pi * b[e].x);
            ga[i] = make_double2(c * a[e].x - s * bt.x, c * a[e].y - s * bt.y);
            gb[i] = make_double2(s * a[e].x + c * bt.x, s * a[e].y + c * bt.y);
        }
    }
    for (int i = lane; i < n; i += 64) {
        const double2 x = va[i], y = vb[i];
        const double2 yt = make_double2(pr * y.x - pi * y.y, pr * y.y + pi * y.x);
        va[i] = make_double2(c * x.x - s * yt.x, c * x.y - s * yt.y);
        vb[i] = make_double2(s * x.x + c * yt.x, s * x.y + c * yt.y);
    }
}

__global__ __launch_bounds__(JAC_THREADS) void k_jacobi_svd(double2* __restrict__ G, double2* __restrict__ Vj,
                                                            double* __restrict__ S,
                                                            const htn_svd_block* __restrict__ desc,
                                                            int max_sweeps, double tol, int* __restrict__ info) {
    __shared__ int any_rot;
    const htn_svd_block D = desc[blockIdx.x];
    const int m = D.m, n = D.n;
    double2* __restrict__ g = G + D.g_off;
    double2* __restrict__ v = Vj + D.v_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = JAC_THREADS / 64;
    // V = identity
    for (int idx = tid; idx < n * n; idx += JAC_THREADS) {
        const int i = idx % n, j = idx / n;
        v[idx] = make_double2(i == j ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    const int np = n + (n & 1);      // padded to even; index np-1 == n is a bye when n is odd
    int sweeps = 0;
    bool done = (n < 2);
    while (!done && sweeps < max_sweeps) {
        if (tid == 0) any_rot = 0;
        __syncthreads();
        int rotated = 0;
        for (int r = 0; r < np - 1; ++r) {
            for (int p = wave; p < np / 2; p += nwaves) {
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
                    jacobi_pair(g + (int64_t)lo * m, g + (int64_t)hi * m, v + (int64_t)lo * n,
                                v + (int64_t)hi * n, m, n, lane, tol, &rotated);
                }
            }
            __syncthreads();
        }
        if (rotated && lane == 0) any_rot = 1;
        __syncthreads();
        done = (any_rot == 0);
        ++sweeps;
        __syncthreads();
    }
    // column norms
    for (int j = wave; j < n; j += nwaves) {
        double s = 0.0;
        for (int i = lane; i < m; i += 64) {
            const double2 x = g[(int64_t)j * m + i];
            s += x.x * x.x + x.y * x.y;
        }
        s = wave_sum(s);
        if (lane == 0) S[D.s_off + j] = sqrt(s);
    }
    if (tid == 0) info[blockIdx.x] = done ? sweeps : -sweeps;
}

extern "C" int htn_jacobi_svd_z(void* G, void* Vj, double* S, const htn_svd_block* desc, int32_t n_blocks,
                                int32_t max_m_host, int32_t max_sweeps, double tol, int32_t* info_dev,
                                void* stream) {
    if (n_blocks <= 0) return 0;
    if (max_m_host > 64 * JAC_MAXEL) return fail_msg("htn_jacobi_svd_z: block taller than 512 rows");
    hipLaunchKernelGGL(k_jacobi_svd, dim3(n_blocks), dim3(JAC_THREADS), 0, (hipStream_t)stream, (double2*)G,
                       (double2*)Vj, S, desc, max_sweeps, tol, info_dev);
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
