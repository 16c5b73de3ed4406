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

#define HTN_ABI_VERSION 2
#define HTN_MAX_BUFS 8
#define HTN_TILE 32              /* output tile edge of the grouped GEMM */
/* grouped GEMM: one 4-wave workgroup per tile, several co-resident per CU; a tile of q = 1, 2 or 4 quadrants (16 x 16)
 * splits its K slabs 4 / q ways over its waves (htn_gemm.hip). */

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
    /* split-K across workgroups (nparts > 1; 0 or 1 = the record is the whole tile): this record is part `part` of
     * `nparts` records that share one output tile and own consecutive ranges of its segment list (COPY segments in the
     * last part).  The parts meet through bufs[HTN_BUF_WS]: int32 tickets[HTN_WS_TICKET_ELEMS * 4] (zero before the
     * launch, reset by the kernel) followed by 32 x 32 complex128 slabs; the part drawing ticket nparts - 1 adds the
     * slabs ws_slot .. ws_slot + nparts - 1 in part order and writes the tile. */
    int32_t part, nparts;
    int32_t ws_slot;    /* first slab of this tile                                   */
    int32_t ticket;     /* index of this tile's ticket counter                       */
} htn_tile;             /* 64 bytes */
#define HTN_BUF_WS 7                 /* buffer-table slot of the split-K workspace   */
#define HTN_WS_TICKET_ELEMS 4096     /* complex128 elements reserved for the tickets (64 KiB = 16384 counters) */
/* workspace size in complex128 elements for a task list using n_slots slabs */
#define HTN_WS_ELEMS(n_slots) (HTN_WS_TICKET_ELEMS + (int64_t)(n_slots) * HTN_TILE * HTN_TILE)

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
 * stop when |beta_j y_j| < tol, restart from the Ritz vector at krylovdim (2 <= krylovdim <= 31).
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
typedef int (*htn_exchange2_fn)(void* y_dev, int64_t n, void* user);    /* != 0: failure, the solve is aborted */
int64_t htn_lanczos_scratch_elems(int32_t krylovdim);
int htn_lanczos_z(const htn_gemm_launch* stages_host, int32_t n_stages, int32_t x_slot, int32_t y_slot,
                  void* V, int64_t n, int32_t krylovdim, double tol, int32_t max_restart, void* scratch,
                  int32_t zero_y, htn_exchange2_fn exchange, void* user,
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
 * runs on ONE CU; with it, QRCP blocks larger than one CU's LDS take the large-block path: panel-blocked pivoted QR
 * (helper workgroups share its trailing update from 160 columns on), then the ring block Jacobi -- all sweeps in ONE launch,
 * every block on several CUs whose LDS-resident column panels are handed round inside the launch (write-through stores +
 * epoch flags -- or stores kept in one XCD's L2 when all workgroups of the block read the same XCD id at run time; bounded
 * waits: a time-out is reported as an error of this call), convergence decided on the device; the
 * small blocks run beside it on an internal second stream that joins `stream` before the call returns; the call blocks
 * the host until everything has converged, results stay on the device.  Blocks the ring cannot take fall back to the
 * pair-visit block Jacobi of ABI 2 (one launch per tournament round; sweeps_hint bounds its speculative enqueue));
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
/* Per-call options of htn_jacobi_svd_z (NULL = defaults); replaces the process-wide setters of ABI version 1 so
 * that two contexts on two host threads cannot see each other's settings.
 *   split_elems : blocks whose R^H (roundup(m) x n elements) exceeds it leave the one-workgroup kernel for the
 *                 large-block path (k_qr_large: panel-blocked pivoted QR; k_jacobi_pairs_gram: block Jacobi over many
 *                 CUs).  <= 0: the default ("does not fit one CU's LDS window", 9216 elements); larger values are
 *                 clamped to it.  Exists so that small problems can exercise the large-block path (tests); results
 *                 agree to the Jacobi tolerance either way.
 *   rank_cut    : rank-revealing stop of the large blocks' pivoted QR: once the squared Frobenius norm of the part
 *                 not yet factorised is below rank_cut^2 (it bounds every remaining singular value), the remaining
 *                 rows of R are dropped, the tournament runs on the rank found, and the dropped singular values are
 *                 reported as 0.  <= 0 = off (default).  Kept singular values then move by at most
 *                 rank_cut^2 / (2 sigma) (interlacing).
 *   sweeps_hint : outer sweeps the large-block path is expected to need (normally `*sweeps_used` of the previous call
 *                 for the same bond); the sweep expected to be the last is not followed by a speculatively enqueued
 *                 one.  <= 0: unknown, every sweep is enqueued one ahead of the host's knowledge.  Results do not
 *                 depend on it.
 *   sweeps_used : host pointer or NULL; receives the number of outer sweeps of the large-block path after which every
 *                 large block had converged (0 when no block took that path). */
typedef struct {
    int32_t split_elems;
    int32_t sweeps_hint;
    double rank_cut;
    int32_t* sweeps_used;
} htn_svd_opts;         /* 24 bytes */
int htn_jacobi_svd_z(void* G, void* Vj, double* S, const htn_svd_block* desc, const htn_svd_block* desc_host,
                     int32_t n_blocks, int32_t max_m_host, int32_t max_sweeps, double tol, int32_t* info_dev,
                     const htn_svd_opts* opts, void* stream);

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


/* =====================================================================================================
 * Bond-update / sweep level (ABI 2): what sits behind
 *     find_groundstate(psi0, H, IDMRG2(; trscheme, tol))          src/HubbardFunctions.jl:1010
 * i.e. MPSKit's two-site sweep body.  The Hamiltonian builder
 * (hamiltonian(), src:386-472, 811-910) and initialize_mps (src:917-959) stay on the host-language side, exactly as
 * in the reference; they hand over
 *   - the MPO as tables: per site the (dN, 2k) labels of its left / right virtual levels and a list of
 *     (left level, right level, site-operator id, coefficient) entries; site operators as reduced matrix elements;
 *   - the MPS as TensorKit-shaped data: one flat ComplexF64 vector per site + a table of sub-blocks
 *     (left sector, site multiplet, right sector, offset, leading dimension) -- TensorKit's per-fusion-tree views --
 *     and per bond the int32 sector labels (N, 2S) with their int32 multiplet counts (htn_sector).
 * Everything below that -- sector layouts, recoupling coefficients (closed-form 9j), task lists, theta formation,
 * Lanczos, per-sector SVD, the global truncation rule, write-back, environment transfer, the sweep loop -- runs
 * inside the library (C++ planner + HIP kernels; one host thread per context).
 * Host pointers end in _host; everything is copied at the call, the caller keeps ownership of its buffers.
 * ===================================================================================================== */
typedef struct htn_ctx htn_ctx;
typedef struct htn_mpo htn_mpo;
typedef struct htn_mps htn_mps;

/* Symmetry of the reduced tensors (src:245-382).  Sector label = (N, j):
 *   HTN_SYM_SU2_U1 : fZ2 x SU(2) x U(1) (src:250): N = particle number (parity = N mod 2), j = 2S, spins couple
 *   HTN_SYM_U1_U1  : fZ2 x U(1) x U(1)  (src:247): N = particle number, j = 2 Sz (additive, may be negative)
 *   HTN_SYM_SU2    : fZ2 x SU(2)        (src:341): N = parity (mod 2), j = 2S
 * site_N / site_j: labels of the n_site site multiplets (3 or 4). */
#define HTN_SYM_SU2_U1 0
#define HTN_SYM_U1_U1 1
#define HTN_SYM_SU2 2
#define HTN_MAX_SITE 4
typedef struct {
    int32_t kind;
    int32_t n_site;
    int32_t site_N[HTN_MAX_SITE];
    int32_t site_j[HTN_MAX_SITE];
} htn_symmetry;

/* reduced site operator: irreducible tensor of rank k/2 (SU(2) kinds; U1_U1: k = change of 2Sz) changing N by dN;
 * red[out * HTN_MAX_SITE + in] = reduced matrix element <out || O || in> in the convention of src:281-293
 * (isometric fusion tensors: c+ has 1 and sqrt(2)). */
typedef struct {
    int32_t k, dN;
    double red[HTN_MAX_SITE * HTN_MAX_SITE];
} htn_site_op;

typedef struct {
    int32_t wl, wr;          /* level on the left / right MPO bond of this site */
    int32_t op;              /* index into the site-operator table             */
    int32_t pad;
    double coef_re, coef_im;
} htn_mpo_entry;

/* one sub-block of a site tensor inside the caller's flat vector: A[(l, s) ; r] as an n_l x n_r column-major
 * matrix at data[off] with leading dimension ld (TensorKit fusion-tree view) */
typedef struct {
    int32_t lN, lj, s, rN, rj;
    int32_t ld;
    int64_t off;
} htn_subblock;

typedef struct {
    int32_t N, j, count;
} htn_sector;

/* truncation + solver settings of one bond update / sweep */
typedef struct {
    int32_t chi_full;          /* truncdim(D), src:1363-1365, in TensorKit dim units (sum (2S+1) n); <= 0: off      */
    int32_t weighting;         /* order at the truncdim cut: 0 = sqrt(2S+1) x Schmidt value, 1 = Schmidt value       */
    double cutoff;             /* truncbelow(eta), src:1007-1010: keep Schmidt values > eta; 0: off                  */
    int32_t krylovdim;         /* 30 (KrylovKit / MPSKit default)                                                    */
    int32_t maxrestart;
    double lanczos_tol;
    double jacobi_tol;         /* 1e-14 */
    int32_t jacobi_max_sweeps; /* 40 */
    int32_t svd_split_elems;   /* htn_svd_opts.split_elems; 0 = default */
    double rank_cut;           /* 0 = off (parity-exact); see DESIGN.md section 4 */
    int32_t profile;           /* 1: synchronise around the stages so that the t_* fields are GPU-inclusive          */
    int32_t pad;
} htn_sweep_opts;

typedef struct {
    int32_t bond, direction;
    int32_t n_matvec, jacobi_sweeps;
    int32_t chi_full, multiplets;
    int32_t n_tiles, n_segs;
    int64_t theta_size;
    int64_t apply_flops, apply_bytes, svd_flops;
    double energy, residual, trunc_weight;
    double t_plan, t_lanczos, t_svd, t_env, t_total;   /* host wall seconds per stage */
    double matvec_ms;                                    /* HIP-event time of the H_eff apply launches (if timed) */
} htn_bond_stats;

/* context: one device, one stream (NULL: the library creates its own), one host thread at a time.
 * backend: HTN_BACKEND_HIP is the product.  libhubbardtn_cpu.so (built from oracle/cpu_backend, the CPU baseline of
 * SURVEY 8d) exports the same ABI with HTN_BACKEND_CPU; neither library contains the other's backend and nothing
 * falls back. */
#define HTN_BACKEND_CPU 0
#define HTN_BACKEND_HIP 1
int htn_ctx_create(int32_t backend, int32_t device, void* stream, htn_ctx** out);
void htn_ctx_destroy(htn_ctx* ctx);
int htn_ctx_backend(const htn_ctx* ctx);
/* record HIP events around every H_eff apply launch (htn_bond_stats.matvec_ms); costs two event records per matvec */
int htn_ctx_set_timing(htn_ctx* ctx, int32_t on);

/* sector-parallel effective-H apply: rank r of `world` computes output tiles r, r + world, ... of every
 * apply and the partial results are summed by ONE all-reduce per matvec on the context's stream.
 *  - htn_comm_unique_id + htn_ctx_set_comm: RCCL over xGMI, called from inside the library (HIP backend);
 *    the 128-byte id is created on rank 0 and distributed by the caller (torch.distributed, MPI, a file, ...)
 *  - htn_ctx_set_exchange: caller-supplied reduction (host callback, invoked after each matvec has been enqueued
 *    with the device pointer of y); a non-zero return aborts the solve.  Used by the gloo/CPU rehearsal tests. */
#define HTN_COMM_ID_BYTES 128
int htn_comm_unique_id(void* id_host);
int htn_ctx_set_comm(htn_ctx* ctx, int32_t rank, int32_t world, const void* id_host);
int htn_ctx_set_exchange(htn_ctx* ctx, int32_t rank, int32_t world, htn_exchange2_fn fn, void* user);

/* MPO of a chain of nsites sites.  level_ptr[nsites + 2]: levels (dN, k pairs) of bond b (between site b-1 and b,
 * b = 0 .. nsites) are levels[2 * level_ptr[b] .. 2 * level_ptr[b + 1]); entry_ptr[nsites + 1]: entries of site i.
 * Level 0 of an interior bond is the identity "nothing applied yet", the last level "term complete" (Jordan form,
 * SURVEY App. A.3); the boundary bonds hold exactly one level. */
int htn_mpo_create(htn_ctx* ctx, const htn_symmetry* sym, int32_t nsites, const htn_site_op* ops_host, int32_t n_ops,
                   const int32_t* level_ptr_host, const int32_t* levels_host, const int32_t* entry_ptr_host,
                   const htn_mpo_entry* entries_host, htn_mpo** out);
void htn_mpo_destroy(htn_mpo* mpo);

/* MPS + environments of a finite chain (or of an iDMRG window between two blocks).  bond_ptr[nsites + 2]: sectors of
 * bond b are sectors[bond_ptr[b] .. bond_ptr[b + 1]).  Site tensors must be right-canonical for sites >= 1 (site 0
 * carries the centre), "tilde" normalised.  sub_ptr[nsites + 1] indexes `subs`; data_ptr[nsites
 * + 1] (element offsets) indexes data_host, sub-block offsets are relative to their site's start.
 * left_env / right_env (may be NULL = open end): the boundary environments in the library's block order
 * (htn_mps_env_size / htn_mps_get_env of the engine they come from). */
int htn_mps_create(htn_ctx* ctx, const htn_mpo* mpo, int32_t nsites, const int32_t* bond_ptr_host,
                   const htn_sector* sectors_host, const int32_t* sub_ptr_host, const htn_subblock* subs_host,
                   const int64_t* data_ptr_host, const void* data_host, const void* left_env_host,
                   const void* right_env_host, htn_mps** out);
void htn_mps_destroy(htn_mps* mps);

/* one two-site update of sites (i, i+1) (0-based): form theta, lowest eigenpair of H_eff (optimise = 1) or
 * <theta|H|theta> only (optimise = 0: the centre is moved without optimisation), per-sector SVD + global truncation,
 * write back (placement 0 'right': A_i = U, centre S V^H on i+1; 1 'left': centre U S on i, B_{i+1} = V^H) and
 * move the environment.  stats may be NULL. */
int htn_bond_update(htn_mps* mps, int32_t i, int32_t direction, int32_t placement, int32_t optimise,
                    const htn_sweep_opts* opts, htn_bond_stats* stats_host);
/* one sweep in MPSKit's DMRG2 order: bonds 1..L-1 rightwards, L-2..1 leftwards (2L-3 updates);
 * stats_host: 2L-3 records or NULL; energy_host receives the last eigenvalue */
int htn_dmrg2_sweep(htn_mps* mps, const htn_sweep_opts* opts, htn_bond_stats* stats_host, double* energy_host);

/* y = H_eff(bond i, i+1) x on host vectors in the library's theta layout (size htn_mps_theta_size); tests and
 * Hermiticity checks.  x and y are complex128 host arrays. */
int64_t htn_mps_theta_size(htn_mps* mps, int32_t i);
int htn_heff2_apply(htn_mps* mps, int32_t i, const void* x_host, void* y_host);
/* theta of sites (i, i+1) as currently stored (host, theta layout) */
int htn_mps_get_theta(htn_mps* mps, int32_t i, void* theta_host);

/* queries (two-call pattern: a NULL output pointer returns the count only) */
int32_t htn_mps_nsites(const htn_mps* mps);
int32_t htn_mps_bond(const htn_mps* mps, int32_t b, htn_sector* sectors_host);              /* -> number of sectors  */
int64_t htn_mps_spectrum(const htn_mps* mps, int32_t b, htn_sector* sectors_host, double* values_host);
                                        /* Schmidt values of the last update of bond b, per sector descending;
                                           sectors[k].count values each; returns the total number of values */
int64_t htn_mps_site_size(const htn_mps* mps, int32_t i, int32_t* kind_host);               /* kind: 'L' or 'R'      */
int32_t htn_mps_get_site(const htn_mps* mps, int32_t i, htn_subblock* subs_host, void* data_host);  /* -> #sub-blocks */
int64_t htn_mps_env_size(const htn_mps* mps, int32_t side, int32_t b);                      /* side 0 = left, 1 = right */
int htn_mps_get_env(const htn_mps* mps, int32_t side, int32_t b, void* data_host);
/* block table of an environment (htn_mps_env_blocks): one record per stored block */
typedef struct {
    int32_t aN, aj, w, bN, bj;     /* left env: (bra, w, ket) stored [n_bra x n_ket]; right env: (ket, w, bra) [n_ket x n_bra] */
    int32_t rows, cols;
    int32_t pad;
    int64_t off;
} htn_env_block;
int32_t htn_mps_env_blocks(const htn_mps* mps, int32_t side, int32_t b, htn_env_block* blocks_host);
/* the bond table an environment was built on.  It is the table of bond b AT THE TIME the environment was formed: the
 * left environment of bond b dates from the last rightward update of that bond, the right one from the last leftward
 * update, and a truncation by dimension may have kept different counts in between.  -> number of sectors */
int32_t htn_mps_env_bond(const htn_mps* mps, int32_t side, int32_t b, htn_sector* sectors_host);
/* planner introspection (tests): the task lists of the H_eff apply on bond i as the kernels receive them.
 * stage 0 = Z stage (may be empty), 1 = Y stage.  Returns counts through n_tiles / n_segs; tiles / segs may be NULL. */
int htn_plan_apply_dump(htn_mps* mps, int32_t i, int32_t stage, int32_t* n_tiles, htn_tile* tiles_host, int32_t* n_segs,
                        htn_seg* segs_host, int64_t* z_size, int64_t* flops);
/* the launch load-balancing pass the engine applies to every task list before upload (HIP backend): long tiles of 2 or 4
 * quadrants (16 x 16) are cut into one-quadrant tiles sharing their segment list, tiles still longer than a CU's fair
 * share into split-K parts (htn_tile.part / nparts / ws_slot / ticket); the records are then dealt in layers of n_cus,
 * longest to the least loaded CU, XCD-aware.  Returns the number of workspace slabs the balanced list needs (bufs[HTN_BUF_WS]
 * must then hold HTN_WS_ELEMS(slabs) elements with the ticket region zeroed), -1 on error; *n_out = number of records
 * (out may be NULL to query it). */
int32_t htn_balance_tiles(const htn_tile* tiles_host, int32_t n_tiles, int32_t n_cus, htn_tile* out_host, int32_t out_cap,
                          int32_t* n_out);
/* plan-cache statistics: hits, misses (host planner invocations) */
int htn_mps_cache_stats(const htn_mps* mps, int64_t* hits, int64_t* misses);

#ifdef __cplusplus
}
#endif
#endif
