/* hubbardtn_hip.h -- C ABI of libhubbardtn_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the two-site DMRG hot path that HubbardTN reaches through
 *     find_groundstate(psi0, H, IDMRG2(...))            src/HubbardFunctions.jl:1010
 * i.e. the work MPSKit 0.13.1 / TensorKit 0.14.6 / KrylovKit 0.9.5 do per bond update
 * (SURVEY.md section 8a rows a7-a10).  Those packages are not vendored under /root/reference;
 * each entry point below names the dependency routine it stands in for and the reference call
 * site that reaches it.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; sizes are element counts
 *   - complex128 = interleaved (re, im) doubles, exactly Julia's Vector{ComplexF64}
 *     (src/HubbardFunctions.jl:264 ff. build every TensorMap as ComplexF64)
 *   - matrices are column-major with an explicit leading dimension (TensorKit block layout)
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued, nothing synchronises
 *     unless stated
 *   - return value 0 = ok, otherwise htn_last_error() (thread-local) describes the failure
 *   - no exceptions cross the boundary; no global mutable state except the error string
 */
#ifndef HUBBARDTN_HIP_H
#define HUBBARDTN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HTN_ABI_VERSION 1
#define HTN_MAX_BUFS 8
#define HTN_TILE 32              /* output tile edge of the grouped GEMM */

/* operand ops */
#define HTN_OP_N 0               /* as stored                        */
#define HTN_OP_T 1               /* transposed                       */
#define HTN_OP_C 2               /* conjugate transposed             */
/* segment types */
#define HTN_SEG_GEMM 0           /* C += alpha * op(A) * op(B)       */
#define HTN_SEG_COPY 1           /* C += alpha * B  (B is m x n)     */

/* One output tile (<= 32 x 32) of an output block.  The tile owns segs[seg_begin .. +seg_count)
 * and is WRITTEN (not accumulated): a tile with no segments stores zeros.  Within that range all
 * GEMM segments come first and the last pad[0] entries are the COPY segments.  pad[1] = 1 promises that every
 * GEMM segment of the tile has k <= 16 (one K slab): the kernel then walks the list without reading it. */
typedef struct {
    int64_t c_off;      /* element offset of the BLOCK origin inside bufs[buf_c]   */
    int32_t buf_c;      /* index into the buffer table                              */
    int32_t ldc;
    int32_t m, n;       /* extent of this tile (1..32)                              */
    int32_t row0, col0; /* tile origin inside the block                             */
    int32_t seg_begin, seg_count;
    int32_t pad[2];
} htn_tile;             /* 48 bytes */

/* One contribution  alpha * op(A)[block rows, 0:k] * op(B)[0:k, block cols]  to an output block.
 * op(A) is (block rows x k), op(B) is (k x block cols); offsets address element (0,0) of op(.). */
typedef struct {
    int64_t a_off, b_off;
    int32_t buf_a, buf_b;
    int32_t lda, ldb;
    int32_t k;
    int32_t op_a, op_b;
    int32_t type;
    double alpha_re, alpha_im;
} htn_seg;              /* 64 bytes */

const char* htn_last_error(void);
int htn_abi_version(void);

/* Binds the calling thread to a device and returns its properties (name buffer >= 256 bytes). */
int htn_device_init(int device, char* name_host, int* cu_count_host);

/* Grouped, segmented complex128 GEMM on v_mfma_f64_16x16x4_f64.
 * Stands in for: TensorKit block mul! (BLAS zgemm per coupled sector) + the TensorOperations /
 * Strided permutes around it, as executed inside MPSKit's AC2 effective-Hamiltonian apply and
 * environment transfers (reached from src/HubbardFunctions.jl:1010).
 * bufs_host: HTN_MAX_BUFS device base pointers (unused entries may be NULL). */
int htn_grouped_gemm_z(const void* const* bufs_host, const htn_tile* tiles, int32_t n_tiles,
                       const htn_seg* segs, void* stream);

/* Krylov vector algebra (stands in for VectorInterface inner/add!!/scale!! on TensorMaps as used
 * by KrylovKit.eigsolve(..., Lanczos)).  The reduced data are stored in the
 * Euclidean ("tilde") normalisation, so TensorKit's dim-weighted inner product is the plain dot.
 *   out[i]  = sum_j conj(V[i*ldv + j]) * w[j]        i < nvec      (deterministic 2-stage reduce)
 *   w[j]   += sign * sum_i coef[i] * V[i*ldv + j]
 * scratch must hold htn_dots_scratch_elems(nvec) complex128. */
int64_t htn_dots_scratch_elems(int32_t nvec);
int htn_dots_z(const void* V, int64_t ldv, int32_t nvec, const void* w, int64_t n,
               void* out, void* scratch, void* stream);
int htn_axpys_z(void* w, const void* V, int64_t ldv, int32_t nvec, const void* coef,
                double sign, int64_t n, void* stream);
/* dst[j] = src[j] / sqrt(Re(nrm2[0]))  (nrm2 on device; dst may alias src) */
int htn_scale_inv_sqrt_z(void* dst, const void* src, const void* nrm2, int64_t n, void* stream);

/* Device-resident Lanczos: lowest eigenpair of the Hermitian map y = H_eff x given as a short sequence of
 * grouped-GEMM launches (stage list; the buffer-table entries x_slot / y_slot are replaced by the current
 * Krylov vector / the output vector for every matvec).
 * Stands in for: KrylovKit.eigsolve(H_eff, x0, 1, :SR, Lanczos(krylovdim, tol, eager=true)) as called by
 * MPSKit's two-site update (reached from src/HubbardFunctions.jl:1010): orthonormal Krylov
 * basis kept in full, explicit two-pass reorthogonalisation, tridiagonal problem solved after every expansion,
 * stop when |beta_j y_j| < tol, restart from the Ritz vector at krylovdim.
 * V: (krylovdim + 2) * n complex128; on entry V[0:n] = start vector, on exit V[0:n] = normalised Ritz vector.
 * exchange (may be NULL): called on the host after each matvec has been ENQUEUED, with the device pointer of y;
 * the multi-GPU host uses it to enqueue an RCCL all-reduce of y on the same stream (zero_y = 1 then clears y
 * before the local tiles are written).  The call synchronises the stream once per iteration (it needs
 * alpha_j, beta_j on the host). */
typedef struct {
    const void* bufs[HTN_MAX_BUFS];
    const htn_tile* tiles;
    const htn_seg* segs;
    int32_t n_tiles;
    int32_t pad;
} htn_gemm_launch;
typedef void (*htn_exchange_fn)(void* y_dev, int64_t n, void* user);
int64_t htn_lanczos_scratch_elems(int32_t krylovdim);
int htn_lanczos_z(const htn_gemm_launch* stages_host, int32_t n_stages, int32_t x_slot, int32_t y_slot,
                  void* V, int64_t n, int32_t krylovdim, double tol, int32_t max_restart, void* scratch,
                  int32_t zero_y, htn_exchange_fn exchange, void* user,
                  double* eig_host, int32_t* n_matvec_host, double* residual_host,
                  double* matvec_ms_host /* NULL, or receives the HIP-event time of all matvec launches */,
                  void* stream);

/* Batched one-sided Jacobi SVD of the coupled-sector blocks of a two-site tensor.
 * Stands in for: TensorKit tsvd!(t; alg=SVD()) -> LAPACK zgesvd per block (SURVEY.md 8a a9,
 * scheme chosen at src/HubbardFunctions.jl:1010 and :1363-1365).
 * For block i the m_i x n_i (column-major, ld = m_i) matrix at G + g_off[i] is overwritten by G*J
 * (mutually orthogonal columns = sigma_j u_j, UNSORTED; surplus columns of a wide block converge to 0),
 * the column norms go to S + s_off[i] and, if flags & HTN_SVD_ACCUMULATE, the n_i x n_i rotation J to
 * Vj + v_off[i] (ld = n_i).  The caller stages M or M^H (htn_batched_copy_z) so that the isometry the
 * sweep direction needs is the normalised G*J itself; the other factor is then a plain GEMM with M.
 * desc: device array of htn_svd_block, desc_host: the same array in host memory (may be NULL: then every block
 * runs on ONE CU; with it, QRCP blocks larger than one CU's LDS take the large-block path: panel-blocked pivoted QR,
 * then block Jacobi whose panel-pair visits run on many CUs, one kernel launch per tournament round, convergence
 * decided on the device; the small blocks run beside it on an internal second stream that joins `stream` before the
 * call returns; the call blocks the host until the sweeps have converged, results stay on the device);
 * max_m_host = max_i max(m_i, pad_i) (<= 512 in this version).
 * info_dev[i] receives the sweep count (<0: not converged).  A sweep ends the iteration when the largest squared
 * cosine it SAW before rotating was <= max(tol^2, 0.1 tol) (quadratic convergence: it leaves < tol^2 behind). */
#define HTN_SVD_ACCUMULATE 1      /* flags: also accumulate the rotation J (else Vj is not touched) */
#define HTN_SVD_QRCP 2            /* flags: the block at g_off is G0 (pad x m, ld = pad); pivoted-QR precondition it,
                                     run Jacobi on R^H (m x n, n = min(pad, m)) and write the m x n result
                                     (rows un-pivoted, ld = m) back to g_off; the v_off region (>= m*n) is scratch */
typedef struct {
    int64_t g_off, v_off, s_off;
    int32_t m, n;
    int32_t flags, pad;
} htn_svd_block;        /* 40 bytes */
int htn_jacobi_svd_z(void* G, void* Vj, double* S, const htn_svd_block* desc, const htn_svd_block* desc_host,
                     int32_t n_blocks, int32_t max_m_host, int32_t max_sweeps, double tol, int32_t* info_dev,
                     void* stream);

/* Blocks whose R^H (roundup(m) x n elements) exceeds `elems` leave the one-workgroup kernel for the large-block
 * path (k_qr_large: panel-blocked pivoted QR; k_jacobi_pairs_gram: block Jacobi over many CUs).  elems <= 0 restores
 * the default ("does not fit one CU's LDS window", 9216 elements); values above the default are clamped to it.
 * Returns the previous setting.  Exists so that small problems can exercise the large-block path (tests); results
 * agree to the Jacobi tolerance either way.  Process-wide, not thread safe. */
int32_t htn_jacobi_set_split(int32_t elems);

/* Rank-revealing stop of the large blocks' pivoted QR: once the squared Frobenius norm of the part not yet
 * factorised is below abs_cut^2 (it bounds every remaining singular value), the remaining rows of R are dropped, the
 * Jacobi tournament runs on the rank found, and the dropped singular values are reported as 0.  abs_cut <= 0 = off
 * (default).  A caller that truncates anyway (truncdim / truncbelow) passes a small fraction of its cut: kept singular
 * values then move by at most abs_cut^2 / (2 sigma) (interlacing).  Returns the previous setting.  Process-wide. */
double htn_jacobi_set_rank_cut(double abs_cut);

/* dst(r x c, ldd) = op(src)(.., lds) with optional per-row / per-column real scaling:
 * generic batched strided copy used to (a) stage M or M^H into the Jacobi workspace and
 * (b) write the truncated isometries U[:, keep] / V^H[keep, :] and the centre S*V^H / U*S.
 * For each item: dst[i + j*ldd] = scale * f(src element), where
 *   op N: src[(i) + perm(j)*lds]      op C: conj(src[perm(j)... see htn_copy_item]). */
typedef struct {
    int64_t dst_off, src_off, idx_off, scl_off; /* idx/scl offsets into int32 / double side arrays; -1 = none */
    int32_t rows, cols;          /* extent of dst                                   */
    int32_t ldd, lds;
    int32_t op;                  /* HTN_OP_N: dst(i,j) = src(i, J) ; HTN_OP_C: dst(i,j) = conj(src(J', i))
                                    where the gathered index (idx side array) applies to dst's
                                    `gather_dim` (0 = rows, 1 = cols)                 */
    int32_t gather_dim;
    int32_t scale_dim;           /* 0: scale by scl[i], 1: scale by scl[j], -1: none  */
    int32_t inv_norm;            /* 1: divide by the gathered scale value instead     */
} htn_copy_item;        /* 64 bytes */
int htn_batched_copy_z(void* dst, const void* src, const int32_t* idx, const double* scl,
                       const htn_copy_item* items, int32_t n_items, double global_scale,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif
