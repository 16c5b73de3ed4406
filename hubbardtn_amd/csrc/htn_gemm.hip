// grouped, segmented complex128 GEMM on v_mfma_f64_16x16x4_f64 (libhubbardtn_hip.so)
//
// Regime: one H_eff apply at chi = 512..2048 is 10^8..10^9 flops spread over a few hundred ragged
// output tiles (<= 32 x 32; mean 18 x 18 at chi = 1024: the sector blocks are 12, 23, 24, 53 ... wide) whose K loops are
// lists of 16-deep slabs gathered from different operand blocks.  The f64 MFMA is slow (64 clk per 16x16x4), so what
// bounds a launch is MFMA ISSUE per CU plus whatever latency is not hidden behind it.
//
// One 4-wave workgroup per tile, several workgroups resident per CU (no LDS staging, ~12 KiB LDS for the epilogue):
//   * a wave feeds its MFMAs STRAIGHT from global memory / L2: lane (l15, l4) of the A operand loads element
//     [row l15][k = 4 ks + l4] of the slab, i.e. exactly the register the MFMA wants -- no LDS round trip, no
//     workgroup barrier inside the K loop; the next slab's 8 loads are in flight during the current slab's 16 MFMAs and
//     the other resident waves of the SIMD cover the rest of the latency;
//   * only the 16 x 16 quadrants a tile really has are computed: a tile with m, n <= 16 is ONE quadrant and its four
//     waves split the K slabs four ways; 16 < m <= 32, n <= 16 (or the transpose): two quadrants x two-way split; a full
//     tile: four quadrants, every wave walks all slabs.  (The previous kernel issued the full 32 x 32 x 16 MFMA set for
//     every slab: 40 % of the issued MFMAs were useful at chi = 1024.)
//   * partial accumulators of the K groups are summed through LDS in a fixed order (deterministic).
//
// Each segment contributes alpha * op(A) * op(B); alpha and the conjugations are applied to the operand registers just
// before the MFMAs (the loads stay in flight until then).  Each wave computes its quadrant TRANSPOSED (operand roles
// swapped) so that consecutive lanes hold consecutive rows of the column-major output (256-byte store runs).
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

// cursor of one wave over the tile's GEMM segments: (segment index, k offset)
struct Cursor {
    int s, k0;
};

// move the cursor forward by `nslabs` K slabs over the flat slab sequence of the tile's GEMM segments
__device__ __forceinline__ void advance(Cursor& c, int nslabs, const htn_seg* __restrict__ sg, int n_gemm, bool presplit) {
    if (presplit) {          // every GEMM segment is one slab (k <= 16): no descriptor reads needed to walk
        c.s += nslabs;
        return;
    }
    while (nslabs > 0 && c.s < n_gemm) {
        const int K = sg[c.s].k;
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

// the MFMA operands of one wave for one K slab, as loaded (alpha / conjugation / K mask still to be applied).  Everything
// but a[] and b[] is wave-uniform and lives in SGPRs.
struct Ops {
    double2 a[4], b[4];       // [k step]: A[row l15][k = 4 ks + l4], B[k = 4 ks + l4][col l15]
    double c1, c2, c3, c4;    // alpha * op(a):  re = c1 a.x + c2 a.y,  im = c3 a.y + c4 a.x
    int ksteps;               // 0: past the last slab
    int kleft;                // k extent of the slab (1..16): lanes with 4 ks + l4 >= kleft contribute zero
    unsigned bflip;           // 0x80000000 if op(B) conjugates (xor-ed into the sign of b.y)
    bool real_alpha;          // alpha_im == 0: two multiplications instead of four per element
};

__device__ __forceinline__ double sflip(double x, bool neg) {
    return __longlong_as_double(__double_as_longlong(x) ^ (neg ? (long long)0x8000000000000000ull : 0ll));
}

// Loads are branch-free and always in bounds: rows / columns beyond the tile edge re-read the edge (their products land in
// accumulator rows / columns that are never stored), k beyond the slab re-reads its last k (masked to zero at use).
// Addressing: scalar base (buffer + block offset + slab start) + ONE 32-bit lane offset per load.
__device__ __forceinline__ void load_ops(Ops& r, const BufTable& bufs, const htn_seg& S, const Cursor& c,
                                         bool valid, int arow, int bcol, int l4) {
    // NO control flow around the loads: exactly 8 per call, so that the compiler can count them (s_waitcnt vmcnt(8 (D - 1))
    // in front of a slab's MFMAs instead of vmcnt(0), which would wait for the prefetches just issued as well).  An invalid
    // call (past the last slab) re-reads the last valid slab and marks the set unused (ksteps = 0).
    const int k0 = valid ? c.k0 : 0;
    int kleft = S.k - k0 < KB ? S.k - k0 : KB;
    kleft = kleft > 0 ? kleft : 1;
    r.kleft = kleft;
    r.ksteps = valid ? (kleft + 3) >> 2 : 0;
    const bool an = S.op_a == HTN_OP_N, bn = S.op_b == HTN_OP_N, ca = S.op_a == HTN_OP_C;
    r.bflip = S.op_b == HTN_OP_C ? 0x80000000u : 0u;
    r.real_alpha = ((unsigned long long)__double_as_longlong(S.alpha_im) << 1) == 0ull;      // (integer test: stays on the SALU)
    r.c1 = S.alpha_re;                               // (sign flips on the bit pattern: SALU, the values stay in SGPRs)
    r.c2 = sflip(S.alpha_im, !ca);
    r.c3 = sflip(S.alpha_re, ca);
    r.c4 = S.alpha_im;
    const int a_sr = an ? 1 : S.lda, a_sk = an ? S.lda : 1;           // element strides of op(A): row, k
    const int b_sc = bn ? S.ldb : 1, b_sk = bn ? 1 : S.ldb;           //                  op(B): column, k
    const char* __restrict__ Ab = (const char*)(bufs.p[S.buf_a] + S.a_off + (int64_t)k0 * a_sk);
    const char* __restrict__ Bb = (const char*)(bufs.p[S.buf_b] + S.b_off + (int64_t)k0 * b_sk);
    // (24-bit multiplies: v_mul_lo_u32 is quarter rate; rows, strides and k are far below 2^24, products below 2^31)
    const unsigned ar = __umul24((unsigned)arow, (unsigned)a_sr), bc = __umul24((unsigned)bcol, (unsigned)b_sc);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned kl = (unsigned)min(4 * ks + l4, kleft - 1);
        r.a[ks] = *(const double2*)(Ab + (size_t)((ar + __umul24(kl, (unsigned)a_sk)) << 4));
        r.b[ks] = *(const double2*)(Bb + (size_t)((bc + __umul24(kl, (unsigned)b_sk)) << 4));
    }
}

__device__ __forceinline__ void mfma_ops(const Ops& r, d4& acc_re, d4& acc_im, int l4) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ks < r.ksteps) {                         // wave-uniform
            double ax = r.a[ks].x, ay = r.a[ks].y;
            if (4 * ks + 4 > r.kleft) {              // the slab's last, partial k step (wave-uniform): lanes beyond k give 0
                const bool live = 4 * ks + l4 < r.kleft;
                ax = live ? ax : 0.0;
                ay = live ? ay : 0.0;
            }
            double a_re, a_im;
            if (r.real_alpha) {                      // wave-uniform
                a_re = r.c1 * ax;
                a_im = r.c3 * ay;
            } else {
                a_re = fma(r.c1, ax, r.c2 * ay);
                a_im = fma(r.c3, ay, r.c4 * ax);
            }
            const double bx = r.b[ks].x;
            const double by = __hiloint2double(__double2hiint(r.b[ks].y) ^ (int)r.bflip, __double2loint(r.b[ks].y));
            // re / im alternate: an MFMA never depends on the one issued just before it
            acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(bx, a_re, acc_re, 0, 0, 0);
            acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(bx, a_im, acc_im, 0, 0, 0);
            acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(-by, a_im, acc_re, 0, 0, 0);
            acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(by, a_re, acc_im, 0, 0, 0);
        }
    }
}

// Segment descriptors are staged ONCE per workgroup in LDS (coalesced vector loads) and read from there per slab: a scalar
// load of a cold 64-byte descriptor costs ~0.6 us and, SMEM returning out of order, cannot be kept more than one deep --
// it was the pace-maker of the K loop (measured: the kernel with loads and MFMAs compiled out still took 11..18 us).
#define GEMM_DESC_MAX 64       // descriptors staged per tile (4 KiB); longer lists read the rest through the scalar cache
__device__ __forceinline__ htn_seg read_seg(const htn_seg* s_desc, const htn_seg* __restrict__ sg, int idx, int n_staged) {
    if (idx >= n_staged) return sg[idx];             // wave-uniform
    const uint4* p = (const uint4*)(s_desc + idx);
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    auto u = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    auto i64 = [&](unsigned lo, unsigned hi) { return (int64_t)(((unsigned long long)u(hi) << 32) | u(lo)); };
    htn_seg S;
    S.a_off = i64(q0.x, q0.y);
    S.b_off = i64(q0.z, q0.w);
    S.buf_a = (int)u(q1.x), S.buf_b = (int)u(q1.y), S.lda = (int)u(q1.z), S.ldb = (int)u(q1.w);
    S.k = (int)u(q2.x), S.op_a = (int)u(q2.y), S.op_b = (int)u(q2.z), S.type = 0;
    S.alpha_re = __longlong_as_double(i64(q3.x, q3.y));
    S.alpha_im = __longlong_as_double(i64(q3.z, q3.w));
    return S;
}

#define GEMM_WAVES 4
#ifdef HTN_GEMM_PROF       // diagnostic build only (tools/gemm_prof.py): per-workgroup start / end stamps and placement
__device__ long long g_gemm_prof[8192 * 6];
extern "C" int htn_gemm_prof_dump(long long* out, int n) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_prof), sizeof(long long) * 6 * (size_t)(n < 8192 ? n : 8192)));
    return 0;
}
#endif
#ifndef GEMM_DEPTH
#define GEMM_DEPTH 1     // K slabs in registers per wave (32 VGPRs each).  Measured: 1 slab x 6 resident workgroups per CU
                         // beats 2 x 4 by 5 % at chi = 1024 (latency is hidden across waves rather than inside one)
#endif
#ifndef GEMM_MINOCC
#define GEMM_MINOCC 6     // workgroups (one wave per SIMD each) co-resident per CU
#endif
__global__ __launch_bounds__(64 * GEMM_WAVES, GEMM_MINOCC) void k_grouped_gemm_z(BufTable bufs, const htn_tile* __restrict__ tiles,
                                                                       const htn_seg* __restrict__ segs, HtnGemmPublish pub) {
    __shared__ double red[3 * 64 * 8];      // partial accumulators of the K groups 1..3 (12 KiB); also the split-K flag
    __shared__ htn_seg s_desc[GEMM_DESC_MAX];

#ifdef HTN_GEMM_PROF
    const long long prof_t0 = wall_clock64();
    const long long prof_c0 = (long long)__builtin_amdgcn_s_memtime();
#endif
    const htn_tile T = tiles[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    if (pub.rec_out && blockIdx.x == 0 && tid < 64) {       // (wave-uniform) the Lanczos driver's record of the previous step
        double t = 0.0;
#pragma unroll
        for (int b = 0; b < HTN_DOT_BLOCKS / 64; ++b) t += pub.norm_partial[tid + 64 * b];      // same order as k_scale_by_norm
        t = wave_sum(t);
        if (tid == 0) {
            const unsigned long long w0 = (unsigned long long)__double_as_longlong(pub.c1[0].x);
            const unsigned long long w1 = (unsigned long long)__double_as_longlong(pub.c2[0].x);
            const unsigned long long w2 = (unsigned long long)__double_as_longlong(t);
            ulonglong2* out = (ulonglong2*)pub.rec_out;
            out[0] = make_ulonglong2(w0, w1);
            out[1] = make_ulonglong2(w2, lan_check(w0, w1, w2, pub.serial));
            __threadfence_system();
        }
    }
    // (readfirstlane: the wave index, hence the K group, the cursor and every segment descriptor field, lives in SGPRs)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // quadrants the tile has, K groups the waves form: wave = grp * nquad + quad
    const int qr = T.m > 16 ? 2 : 1, qc = T.n > 16 ? 2 : 1, nquad = qr * qc, ngrp = GEMM_WAVES / nquad;
    const int quad = wave % nquad, grp = wave / nquad;
    const int wr = qr == 2 ? (qc == 2 ? quad >> 1 : quad) : 0, wc = qc == 2 ? (quad & 1) : 0;
    const int n_gemm = T.seg_count - T.pad[0];     // GEMM segments first, then pad[0] COPY segments
    const bool presplit = T.pad[1] != 0;
    const htn_seg* __restrict__ sg = segs + T.seg_begin;
    const int n_staged = n_gemm < GEMM_DESC_MAX ? n_gemm : GEMM_DESC_MAX;
    const int orow = wr * 16 + l15;                // this lane's outputs are C[r0 + orow][c0 + wc 16 + l4 + 4 r]
    // operand row / column of this lane, clamped to the tile edge (see load_ops)
    const int arow = T.row0 + (orow < T.m ? orow : T.m - 1), bcol = T.col0 + (wc * 16 + l15 < T.n ? wc * 16 + l15 : T.n - 1);

    d4 acc_re = {0.0, 0.0, 0.0, 0.0};
    d4 acc_im = {0.0, 0.0, 0.0, 0.0};

    // Start-up chain: tile record -> segment descriptors -> operands are three dependent memory round trips (~1.3 us each)
    // during which every resident workgroup of the launch idles at once.  The descriptor staging (vector loads -> LDS) is
    // issued first, but the wave's FIRST descriptor comes straight through the scalar cache in parallel with it and the first
    // slab's operand loads are in flight before the staging barrier; only the later descriptors wait for the LDS copy.
    for (int i = tid; i < 4 * n_staged; i += 64 * GEMM_WAVES) ((uint4*)s_desc)[i] = ((const uint4*)sg)[i];
    Cursor cn = {0, 0};
    bool vn = false;
    Ops r[GEMM_DEPTH];
    if (n_gemm > 0) {
        advance(cn, grp, sg, n_gemm, presplit);
        vn = cn.s < n_gemm;
        const htn_seg S0 = sg[vn ? cn.s : n_gemm - 1];      // (scalar loads; always a real descriptor: load_ops loads from it)
        load_ops(r[0], bufs, S0, cn, vn, arow, bcol, l4);
    }
    __syncthreads();

    // ---- K loop: group g takes slabs g, g + ngrp, ... of the tile's flat slab sequence.  GEMM_DEPTH register sets
    // rotate: the loads of the next GEMM_DEPTH - 1 slabs are in flight while the current slab's 16 MFMAs issue (an
    // operand fetch out of L2 / Infinity Cache takes 1.5-2 us, a slab's MFMAs 0.43 us) ----
    if (n_gemm > 0) {
        // The segment DESCRIPTOR of the slab after the one being fetched is read (from LDS) one step ahead (Sn).
        htn_seg Sn = read_seg(s_desc, sg, vn ? cn.s : n_gemm - 1, n_staged);
        if (vn) {
            advance(cn, ngrp, sg, n_gemm, presplit);
            vn = cn.s < n_gemm;
            if (vn) Sn = read_seg(s_desc, sg, cn.s, n_staged);
        }
#pragma unroll
        for (int d = 1; d < GEMM_DEPTH; ++d) {
            load_ops(r[d], bufs, Sn, cn, vn, arow, bcol, l4);
            if (vn) {
                advance(cn, ngrp, sg, n_gemm, presplit);
                vn = cn.s < n_gemm;
                if (vn) Sn = read_seg(s_desc, sg, cn.s, n_staged);
            }
        }
        bool more = true;
        while (more) {
#pragma unroll
            for (int d = 0; d < GEMM_DEPTH; ++d) {
                if (r[d].ksteps == 0) {              // past the last slab of this group (wave-uniform)
                    more = false;
                    break;
                }
                mfma_ops(r[d], acc_re, acc_im, l4);
                load_ops(r[d], bufs, Sn, cn, vn, arow, bcol, l4);
                if (vn) {
                    advance(cn, ngrp, sg, n_gemm, presplit);
                    vn = cn.s < n_gemm;
                    if (vn) Sn = read_seg(s_desc, sg, cn.s, n_staged);
                }
            }
        }
    }
#ifdef HTN_GEMM_PROF
    const long long prof_t1 = wall_clock64();
    const long long prof_c1 = (long long)__builtin_amdgcn_s_memtime();
#endif
    // ---- COPY segments: C += alpha * X tile, spread over the K groups (each adds into its partial) ----
    for (int s = n_gemm + grp; s < T.seg_count; s += ngrp) {
        const htn_seg S = sg[s];
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
    // ---- reduce the partial accumulators of the K groups through LDS (fixed order => deterministic) ----
    if (ngrp > 1) {
        if (grp > 0) {
            double* dst = red + (((grp - 1) * nquad + quad) * 64 + lane) * 8;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dst[r] = acc_re[r];
                dst[4 + r] = acc_im[r];
            }
        }
        __syncthreads();
        if (grp == 0) {
            for (int p = 0; p < ngrp - 1; ++p) {
                const double* src = red + ((p * nquad + quad) * 64 + lane) * 8;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc_re[r] += src[r];
                    acc_im[r] += src[4 + r];
                }
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
    // reducer: it is zero before every launch (zero-initialised once by the owner of the workspace).  Slab layout:
    // [quadrant wr * 2 + wc][lane][reg] -- every part of a tile has the same quadrants. ----
    if (T.nparts > 1) {
        double2* __restrict__ ws = bufs.p[HTN_BUF_WS];
        int* __restrict__ tickets = (int*)ws;
        const int sidx = ((wr * 2 + wc) * 64 + lane) * 4;
        double2* __restrict__ slab = ws + HTN_WS_TICKET_ELEMS + (int64_t)(T.ws_slot + T.part) * (HTN_TILE * HTN_TILE);
        if (grp == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[sidx + r] = make_double2(acc_re[r], acc_im[r]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = (int*)red;                   // (the partials are dead now)
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
        if (grp == 0) {
            const double2* __restrict__ s0 = ws + HTN_WS_TICKET_ELEMS + (int64_t)T.ws_slot * (HTN_TILE * HTN_TILE) + sidx;
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
    if (grp == 0 && orow < T.m) {
        double2* __restrict__ Cp = bufs.p[T.buf_c] + T.c_off;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ocol = wc * 16 + l4 + 4 * r;
            if (ocol < T.n)
                Cp[(int64_t)(T.row0 + orow) + (int64_t)(T.col0 + ocol) * T.ldc] =
                    make_double2(acc_re[r], acc_im[r]);
        }
    }
#ifdef HTN_GEMM_PROF
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        long long* o = g_gemm_prof + 6 * (size_t)blockIdx.x;
        o[4] = prof_c1 - prof_c0;
        o[5] = n_gemm * 256 + nquad;
        o[0] = prof_t0;
        o[1] = prof_t1;
        o[2] = wall_clock64();
        o[3] = ((long long)(xcc & 0xf) << 32) | hwid;
    }
#endif
}

int htn_grouped_gemm_launch(const void* const* bufs_host, const htn_tile* tiles, int32_t n_tiles, const htn_seg* segs,
                            const HtnGemmPublish* pub, hipStream_t stream) {
    if (n_tiles <= 0) return pub && pub->rec_out ? fail_msg("htn_grouped_gemm_launch: a record to publish but no tiles") : 0;
    BufTable bt;
    for (int i = 0; i < HTN_MAX_BUFS; ++i) bt.p[i] = (double2*)bufs_host[i];
    HtnGemmPublish p = {nullptr, nullptr, nullptr, nullptr, 0ull};
    if (pub) p = *pub;
    hipLaunchKernelGGL(k_grouped_gemm_z, dim3(n_tiles), dim3(64 * GEMM_WAVES), 0, stream, bt, tiles, segs, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int htn_grouped_gemm_z(const void* const* bufs_host, const htn_tile* tiles, int32_t n_tiles,
                                  const htn_seg* segs, void* stream) {
    return htn_grouped_gemm_launch(bufs_host, tiles, n_tiles, segs, nullptr, (hipStream_t)stream);
}
