// grouped, segmented complex128 GEMM on v_mfma_f64_16x16x4_f64 (libhubbardtn_hip.so)
//
// Regime: one H_eff apply at chi = 512..2048 is 10^7..10^9 flops spread over a few
// hundred ragged output tiles -- far too little to fill 256 CUs by output tiles alone, and the f64
// MFMA is slow (64 clk per 16x16x4), so the critical path is ONE tile's K loop.  The kernel therefore
// spends a whole 16-wave workgroup on each <=32x32 tile: 4 wave-quads take the tile's K slabs
// round-robin (intra-workgroup split-K), each quad owns private LDS slab buffers, the next slab's
// global loads are issued before the current slab's MFMAs (register prefetch), and the 4 partial
// accumulators are summed through LDS in a fixed order (deterministic).
//
// Each segment contributes alpha * op(A) * op(B); alpha is folded into A while staging, so every
// segment accumulates into the same MFMA accumulators.  K slabs are 16 deep, staged as separate
// re / im planes.  Each wave owns a 16x16 quadrant and computes it TRANSPOSED (operand roles swapped)
// so that consecutive lanes hold consecutive rows of the column-major output (256-byte store runs).
//
// f64 MFMA lane maps (cdna_hip_programming.md section 3):
//   A operand: lane l holds Aop[i = l & 15][k = l >> 4]     B operand: Bop[k = l >> 4][j = l & 15]
//   D: lane l, reg r holds D[row = (l >> 4) + 4 r][col = l & 15]
// With Aop[i][k] = B[k][c0 + i] and Bop[k][j] = A[r0 + j][k]:  D[i][j] = C[r0 + j][c0 + i].
#include "htn_common.h"

struct BufTable {
    double2* p[HTN_MAX_BUFS];
};

#define KB 16         // K slab depth
#define NQ HTN_GEMM_QUADS   // wave-quads per workgroup (intra-workgroup split-K ways)
#define LDS_LD 48     // padded leading dimension (doubles) of a 32-wide slab row: 48 = 16 mod 32 keeps
                      // the two k-rows read by one 32-lane group on disjoint banks (ds_read_b64)
// Row kk is additionally rotated by kk inside its 32 doubles: a k-contiguous staging pass (16 lanes
// writing 16 different kk at one idx) then hits 16 different banks instead of one.
#define LDS_AT(kk, idx) ((kk) * LDS_LD + (((idx) + (kk)) & 31))
#define SLAB (KB * LDS_LD)

struct Slab {             // what one thread stages for one K slab: 2 elements of A, 2 of B
    double2 a[2], b[2];
};

// cursor of one quad over the tile's GEMM segments: (segment index, k offset)
struct Cursor {
    int s, k0;
};

// move the cursor forward by `nslabs` K slabs over the flat slab sequence of the tile's GEMM segments
__device__ __forceinline__ void advance(Cursor& c, int nslabs, const htn_seg* __restrict__ segs, int seg_begin,
                                        int n_gemm, bool presplit) {
    if (presplit) {          // every GEMM segment is one slab (k <= 16): no descriptor reads needed to walk
        c.s += nslabs;
        return;
    }
    while (nslabs > 0 && c.s < n_gemm) {
        const int K = segs[seg_begin + c.s].k;
        const int rem = (K - c.k0 + KB - 1) / KB;
        if (nslabs < rem) {
            c.k0 += nslabs * KB;
            nslabs = 0;
        } else {
            nslabs -= rem;
            ++c.s;
            c.k0 = 0;
        }
    }
}

__device__ __forceinline__ void load_slab(Slab& r, const BufTable& bufs, const htn_tile& T, const htn_seg& S,
                                          const Cursor& c, bool valid, int tq) {
    r.a[0] = r.a[1] = r.b[0] = r.b[1] = make_double2(0.0, 0.0);
    if (!valid) return;
    const double2* __restrict__ Ap = bufs.p[S.buf_a] + S.a_off;
    const double2* __restrict__ Bp = bufs.p[S.buf_b] + S.b_off;
    const int K = S.k;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int r_, kk;
        if (S.op_a == HTN_OP_N) {       // rows contiguous in memory
            r_ = tq & 31;
            kk = (tq >> 5) + 8 * it;
        } else {                        // k contiguous in memory
            kk = tq & 15;
            r_ = (tq >> 4) + 16 * it;
        }
        if (r_ < T.m && c.k0 + kk < K) {
            double2 a;
            if (S.op_a == HTN_OP_N) a = Ap[(int64_t)(T.row0 + r_) + (int64_t)(c.k0 + kk) * S.lda];
            else {
                a = Ap[(int64_t)(c.k0 + kk) + (int64_t)(T.row0 + r_) * S.lda];
                if (S.op_a == HTN_OP_C) a.y = -a.y;
            }
            r.a[it] = make_double2(S.alpha_re * a.x - S.alpha_im * a.y, S.alpha_re * a.y + S.alpha_im * a.x);
        }
        int c_, kb;
        if (S.op_b == HTN_OP_N) {       // k contiguous
            kb = tq & 15;
            c_ = (tq >> 4) + 16 * it;
        } else {                        // columns contiguous
            c_ = tq & 31;
            kb = (tq >> 5) + 8 * it;
        }
        if (c_ < T.n && c.k0 + kb < K) {
            double2 b;
            if (S.op_b == HTN_OP_N) b = Bp[(int64_t)(c.k0 + kb) + (int64_t)(T.col0 + c_) * S.ldb];
            else {
                b = Bp[(int64_t)(T.col0 + c_) + (int64_t)(c.k0 + kb) * S.ldb];
                if (S.op_b == HTN_OP_C) b.y = -b.y;
            }
            r.b[it] = b;
        }
    }
}

__device__ __forceinline__ void store_slab(const Slab& r, double* __restrict__ lds, int op_a, int op_b, int tq) {
    // lds: [A_re | A_im | B_re | B_im] of this quad
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int r_, kk;
        if (op_a == HTN_OP_N) {
            r_ = tq & 31;
            kk = (tq >> 5) + 8 * it;
        } else {
            kk = tq & 15;
            r_ = (tq >> 4) + 16 * it;
        }
        lds[0 * SLAB + LDS_AT(kk, r_)] = r.a[it].x;
        lds[1 * SLAB + LDS_AT(kk, r_)] = r.a[it].y;
        int c_, kb;
        if (op_b == HTN_OP_N) {
            kb = tq & 15;
            c_ = (tq >> 4) + 16 * it;
        } else {
            c_ = tq & 31;
            kb = (tq >> 5) + 8 * it;
        }
        lds[2 * SLAB + LDS_AT(kb, c_)] = r.b[it].x;
        lds[3 * SLAB + LDS_AT(kb, c_)] = r.b[it].y;
    }
}

__global__ __launch_bounds__(256 * NQ) void k_grouped_gemm_z(BufTable bufs, const htn_tile* __restrict__ tiles,
                                                             const htn_seg* __restrict__ segs) {
    __shared__ double lds_all[NQ * 4 * SLAB];        // 4 quads x (A_re, A_im, B_re, B_im) = 96 KiB

    const htn_tile T = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    const int q = tid >> 8;               // wave-quad
    const int tq = tid & 255;             // thread within quad
    const int lane = tid & 63;
    const int wave = tq >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    double* __restrict__ lds = lds_all + q * 4 * SLAB;
    const int n_gemm = T.seg_count - T.pad[0];     // GEMM segments first, then pad[0] COPY segments

    d4 acc_re = {0.0, 0.0, 0.0, 0.0};
    d4 acc_im = {0.0, 0.0, 0.0, 0.0};

    const bool presplit = T.pad[1] != 0;
    if (presplit) {
        // ---- pre-split tiles (what the planner emits): every GEMM segment IS one K slab, so quad q's slab t is
        // segment q + NQ t and no cursor walk is needed.  Two slabs are in flight in registers (loaded two rounds
        // before they are staged): the operand fetch -- L2 / Infinity-Cache latency, ~2-3 us under load -- hides
        // behind TWO rounds of MFMAs instead of one (SQ_VALU_MFMA_BUSY was 28 % of the busy CU time with one).
        const htn_seg* __restrict__ sg = segs + T.seg_begin;
        const Cursor c0 = {0, 0};
        int j = q;                                         // segment of the slab staged next
        htn_seg D = {};
        Slab r0, r1;
        bool v0 = j < n_gemm, v1 = j + NQ < n_gemm, vn = j + 2 * NQ < n_gemm;
        int oa0 = 0, ob0 = 0, k0 = 0, oa1 = 0, ob1 = 0, k1 = 0;
        if (v0) {
            D = sg[j];
            oa0 = D.op_a, ob0 = D.op_b, k0 = D.k;
        }
        load_slab(r0, bufs, T, D, c0, v0, tq);
        if (v1) {
            D = sg[j + NQ];
            oa1 = D.op_a, ob1 = D.op_b, k1 = D.k;
        }
        load_slab(r1, bufs, T, D, c0, v1, tq);
        if (vn) D = sg[j + 2 * NQ];                          // descriptor of the slab loaded in the first round
        bool any = n_gemm > 0;
#define HTN_ROUND(RS, OA, OB, KK, VV)                                                                        \
        {                                                                                                    \
            store_slab(RS, lds, VV ? OA : 0, VV ? OB : 0, tq);                                               \
            __syncthreads();                                                                                 \
            const bool cur_valid = VV;                                                                       \
            const int kleft = VV ? KK : 0;                                                                   \
            /* the register slab just staged is free: fetch the slab two rounds ahead into it */            \
            VV = vn;                                                                                         \
            if (vn) OA = D.op_a, OB = D.op_b, KK = D.k;                                                      \
            load_slab(RS, bufs, T, D, c0, vn, tq);                                                           \
            j += NQ;                                                                                         \
            vn = j + 2 * NQ < n_gemm;                                                                        \
            if (cur_valid) {                                                                                 \
                const int ksteps = kleft >= KB ? KB / 4 : (kleft + 3) >> 2;                                  \
                for (int ks = 0; ks < ksteps; ++ks) {                                                        \
                    const int kk = ks * 4 + l4;                                                              \
                    const double b_re = lds[2 * SLAB + LDS_AT(kk, wc * 16 + l15)];                           \
                    const double b_im = lds[3 * SLAB + LDS_AT(kk, wc * 16 + l15)];                           \
                    const double a_re = lds[0 * SLAB + LDS_AT(kk, wr * 16 + l15)];                           \
                    const double a_im = lds[1 * SLAB + LDS_AT(kk, wr * 16 + l15)];                           \
                    acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(b_re, a_re, acc_re, 0, 0, 0);              \
                    acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(-b_im, a_im, acc_re, 0, 0, 0);             \
                    acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(b_re, a_im, acc_im, 0, 0, 0);              \
                    acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(b_im, a_re, acc_im, 0, 0, 0);              \
                }                                                                                            \
            }                                                                                                \
            /* descriptor of the slab the NEXT round loads: after the MFMA phase (scalar loads share lgkmcnt with  \
               the LDS operand reads), its latency hides behind the barrier and the next staging */         \
            if (vn) D = sg[j + 2 * NQ];                                                                      \
        }
        while (any) {
            HTN_ROUND(r0, oa0, ob0, k0, v0)
            any = __syncthreads_or(v1 ? 1 : 0) != 0;          // is there a slab left to stage? (also fences LDS reuse)
            if (!any) break;
            HTN_ROUND(r1, oa1, ob1, k1, v1)
            any = __syncthreads_or(v0 ? 1 : 0) != 0;
        }
#undef HTN_ROUND
    } else {
        // Cursor of this quad over the tile's flat slab sequence (quad q takes slabs q, q+4, q+8, ...).  The
        // segment DESCRIPTOR of the slab after the one being fetched is loaded one round ahead, so the data loads
        // of a round never wait for a descriptor (a cold 64-byte scalar load costs ~0.6 us on the critical path).
        Cursor cur = {0, 0};
        advance(cur, q, segs, T.seg_begin, n_gemm, presplit);
        bool valid = cur.s < n_gemm;
        htn_seg Sd = {};            // never read a descriptor that does not exist (a tile may own zero segments)
        if (valid) Sd = segs[T.seg_begin + cur.s];
        Slab regs;
        load_slab(regs, bufs, T, Sd, cur, valid, tq);
        Cursor nxt = cur;
        bool nvalid = false;
        if (valid) {
            advance(nxt, NQ, segs, T.seg_begin, n_gemm, presplit);
            nvalid = nxt.s < n_gemm;
        }
        htn_seg Sn = {};
        if (nvalid) Sn = segs[T.seg_begin + nxt.s];
        bool any = n_gemm > 0;
        while (any) {
            const int op_a = valid ? Sd.op_a : 0, op_b = valid ? Sd.op_b : 0;
            const int kleft = valid ? Sd.k - cur.k0 : 0;
            store_slab(regs, lds, op_a, op_b, tq);
            __syncthreads();
            const bool cur_valid = valid;
            // the prefetched descriptor becomes current: issue the next slab's global loads before the MFMAs
            cur = nxt;
            valid = nvalid;
            Sd = Sn;
            load_slab(regs, bufs, T, Sd, cur, valid, tq);
            nvalid = false;
            if (valid) {
                nxt = cur;
                advance(nxt, NQ, segs, T.seg_begin, n_gemm, presplit);
                nvalid = nxt.s < n_gemm;
            }
            if (cur_valid) {
                const int ksteps = kleft >= KB ? KB / 4 : (kleft + 3) >> 2;
                for (int ks = 0; ks < ksteps; ++ks) {
                    const int kk = ks * 4 + l4;
                    const double b_re = lds[2 * SLAB + LDS_AT(kk, wc * 16 + l15)];   // Aop[i][k] = B[k][c0+i]
                    const double b_im = lds[3 * SLAB + LDS_AT(kk, wc * 16 + l15)];
                    const double a_re = lds[0 * SLAB + LDS_AT(kk, wr * 16 + l15)];   // Bop[k][j] = A[r0+j][k]
                    const double a_im = lds[1 * SLAB + LDS_AT(kk, wr * 16 + l15)];
                    acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(b_re, a_re, acc_re, 0, 0, 0);
                    acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(-b_im, a_im, acc_re, 0, 0, 0);
                    acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(b_re, a_im, acc_im, 0, 0, 0);
                    acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(b_im, a_re, acc_im, 0, 0, 0);
                }
            }
            // descriptor of the slab after next: issued AFTER the MFMA phase -- scalar loads share lgkmcnt with the LDS
            // reads of the MFMA operands, so an earlier issue would stall the first MFMA on this cold load; here its
            // latency hides behind the barrier and the next round's LDS staging
            if (nvalid) Sn = segs[T.seg_begin + nxt.s];
            any = __syncthreads_or(valid ? 1 : 0) != 0;     // also fences LDS reuse
        }
    }
    // ---- COPY segments: C += alpha * X tile, spread over the quads (each adds into its partial) ----
    // this lane's outputs are C[r0 + l15][c0 + l4 + 4 r], r = 0..3
    const int orow = wr * 16 + l15;
    for (int s = n_gemm + q; s < T.seg_count; s += NQ) {
        const htn_seg S = segs[T.seg_begin + s];
        const double2* __restrict__ Bp = bufs.p[S.buf_b] + S.b_off;
        if (orow < T.m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ocol = wc * 16 + l4 + 4 * r;
                if (ocol < T.n) {
                    const double2 v = Bp[(int64_t)(T.row0 + orow) + (int64_t)(T.col0 + ocol) * S.ldb];
                    acc_re[r] += S.alpha_re * v.x - S.alpha_im * v.y;
                    acc_im[r] += S.alpha_re * v.y + S.alpha_im * v.x;
                }
            }
        }
    }
    // ---- reduce the 4 partial accumulators through LDS (fixed order => deterministic) ----
    __syncthreads();
    double* red = lds_all;                      // [3][256][8] doubles = 48 KiB
    if (q > 0) {
        double* dst = red + ((q - 1) * 256 + tq) * 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dst[r] = acc_re[r];
            dst[4 + r] = acc_im[r];
        }
    }
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int p = 0; p < NQ - 1; ++p) {
            const double* src = red + (p * 256 + tq) * 8;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc_re[r] += src[r];
                acc_im[r] += src[4 + r];
            }
        }
    }
    // ---- split-K ACROSS workgroups (nparts > 1): the tile's segment list was cut into nparts consecutive ranges, one
    // workgroup each (the planner does this to the longest tiles: one apply is bound by its longest tile's dependent
    // K loop, not by flops).  Every part publishes its 32 x 32 partial sum as a 16 KiB slab in the workspace
    // (bufs.p[HTN_BUF_WS]); the part whose ticket draw is nparts - 1 adds the slabs IN PART ORDER (bit-reproducible
    // whatever the arrival order) and writes the tile.  In-launch hand-off across CUs / XCDs: plain slab stores,
    // vmcnt drain, workgroup barrier, ONE agent-scope release + relaxed agent ticket add; the reducer: ONE agent-scope
    // acquire, then plain loads (cdna_hip_programming.md, "in-launch split-K reduction").  The ticket is reset by the
    // reducer: it is zero before every launch (zero-initialised once by the owner of the workspace). ----
    if (T.nparts > 1) {
        double2* __restrict__ ws = bufs.p[HTN_BUF_WS];
        int* __restrict__ tickets = (int*)ws;
        double2* __restrict__ slab = ws + HTN_WS_TICKET_ELEMS + (int64_t)(T.ws_slot + T.part) * (HTN_TILE * HTN_TILE);
        if (q == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[tq * 4 + r] = make_double2(acc_re[r], acc_im[r]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = (int*)lds_all;               // (the LDS planes are dead now; ONE __shared__ object in this kernel)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            flag[0] = __hip_atomic_fetch_add(&tickets[T.ticket], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (flag[0] != T.nparts - 1) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&tickets[T.ticket], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (q == 0) {
            const double2* __restrict__ s0 = ws + HTN_WS_TICKET_ELEMS + (int64_t)T.ws_slot * (HTN_TILE * HTN_TILE) + tq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_re[r] = acc_im[r] = 0.0;
            for (int p = 0; p < T.nparts; ++p) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 v = s0[(int64_t)p * (HTN_TILE * HTN_TILE) + r];
                    acc_re[r] += v.x;
                    acc_im[r] += v.y;
                }
            }
        }
    }
    if (q == 0 && orow < T.m) {
        double2* __restrict__ Cp = bufs.p[T.buf_c] + T.c_off;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ocol = wc * 16 + l4 + 4 * r;
            if (ocol < T.n)
                Cp[(int64_t)(T.row0 + orow) + (int64_t)(T.col0 + ocol) * T.ldc] =
                    make_double2(acc_re[r], acc_im[r]);
        }
    }
}

extern "C" int htn_grouped_gemm_z(const void* const* bufs_host, const htn_tile* tiles, int32_t n_tiles,
                                  const htn_seg* segs, void* stream) {
    if (n_tiles <= 0) return 0;
    BufTable bt;
    for (int i = 0; i < HTN_MAX_BUFS; ++i) bt.p[i] = (double2*)bufs_host[i];
    hipLaunchKernelGGL(k_grouped_gemm_z, dim3(n_tiles), dim3(256 * NQ), 0, (hipStream_t)stream, bt, tiles, segs);
    HIP_TRY(hipGetLastError());
    return 0;
}
