// Krylov vector algebra + Lanczos driver (libhubbardtn_hip.so).
//
// Stands in for KrylovKit.eigsolve(f, x0, 1, :SR, Lanczos(krylovdim, tol, eager)) as MPSKit calls it
// on the AC2 effective Hamiltonian (reached from src/HubbardFunctions.jl:1010).
// All vectors live in HBM; per iteration the host sees only the new tridiagonal coefficients
// (alpha_j, beta_j) -- a 32-byte record in coherent pinned memory -- and solves the <= krylovdim x krylovdim tridiagonal
// eigenproblem itself.  The vector kernels are HBM / Infinity-Cache bandwidth bound for many rows (roofline "hbm") and
// launch-bound for few.
//
// Reductions are two-stage with a fixed summation order (per-block partials, then one wave sums the
// 64 partials with a butterfly), so results are bit-reproducible run to run and rank to rank.
#include <math.h>

#include <chrono>

#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "htn_common.h"

#define DOT_BLOCKS HTN_DOT_BLOCKS   // partial sums per vector = slices of the vector (4 per lane of the reducing wave)
#define DOT_THREADS 256       // threads of the axpy / scale kernels
#define DOT_CHUNK 32          // vectors handled per pass over the slice of w (krylovdim + 1 <= 31 by default)

// vectors per pass are a compile-time count (all loads of an element in flight at once); rows past nvec are clamped re-reads
// of the last row, which cost real cache traffic -- so the count comes in steps of four (nvec = 17 under a 32-wide
// instantiation took 22 us where 16 rows took 15).  Threads per workgroup (TH) are a template parameter too, but only 256 is
// dispatched: 1024 threads (a slice of 667 elements at chi = 1024 in one step per thread instead of three) changed no
// kernel's duration -- the 4.5-5 us of the small-j kernels are launch + one round trip + the end-of-kernel write-back,
// whatever the thread count (measured, round 3).
#define HTN_CH_DISPATCH(nvec, per, LAUNCH)          \
    do {                                            \
        if ((nvec) <= 4) LAUNCH(4, 256);            \
        else if ((nvec) <= 8) LAUNCH(8, 256);       \
        else if ((nvec) <= 12) LAUNCH(12, 256);     \
        else if ((nvec) <= 16) LAUNCH(16, 256);     \
        else if ((nvec) <= 20) LAUNCH(20, 256);     \
        else if ((nvec) <= 24) LAUNCH(24, 256);     \
        else if ((nvec) <= 28) LAUNCH(28, 256);     \
        else LAUNCH(32, 256);                       \
    } while (0)
static inline int64_t dot_slice(int64_t n) { return (n + DOT_BLOCKS - 1) / DOT_BLOCKS; }

// sum of the DOT_BLOCKS partials of one vector by one wave, fixed order
__device__ __forceinline__ double2 reduce_partials(const double2* __restrict__ p, int lane) {
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int b = 0; b < DOT_BLOCKS / 64; ++b) {
        const double2 v = p[lane + 64 * b];
        sr += v.x;
        si += v.y;
    }
    return make_double2(wave_sum(sr), wave_sum(si));
}

// partial[i * DOT_BLOCKS + b] = sum over slice b of conj(V_i) * w.  Every lane issues all its loads back to back (latency,
// not bandwidth, bounds this kernel at |theta| ~ 10^5); the reductions are the in-wave butterfly plus one fixed-order sum over
// the four waves of a slice.  CH = vectors per pass (compile time, so the loads are unconditional and can all be in flight;
// indices past nvec are clamped and their sums discarded).
template <int CH, int TH>
__global__ __launch_bounds__(TH) void k_dots_partial(const double2* __restrict__ V, int64_t ldv, int nvec,
                                                      const double2* __restrict__ w, int64_t n,
                                                      double2* __restrict__ partial) {
    // TH / 64 waves per slice: a wave keeps CH + 1 loads in flight per lane and the kernel is bound by the latency of that
    // one batch -- 256 waves on 256 CUs moved 4.7 TB/s; the waves' shares are summed in fixed wave order
    __shared__ double red[TH / 64][CH][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t per = (n + DOT_BLOCKS - 1) / DOT_BLOCKS;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    for (int i0 = 0; i0 < nvec; i0 += CH) {
        double sr[CH], si[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) sr[c] = si[c] = 0.0;
        for (int64_t j = lo + threadIdx.x; j < hi; j += TH) {
            const double2 b = w[j];
            double2 a[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int i = i0 + c < nvec ? i0 + c : nvec - 1;
                a[c] = V[(int64_t)i * ldv + j];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                sr[c] += a[c].x * b.x + a[c].y * b.y;
                si[c] += a[c].x * b.y - a[c].y * b.x;
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const double r = wave_sum(sr[c]);
            const double m = wave_sum(si[c]);
            if (lane == 0) {
                red[wave][c][0] = r;
                red[wave][c][1] = m;
            }
        }
        __syncthreads();
        if (threadIdx.x < CH && i0 + (int)threadIdx.x < nvec) {
            const int c = threadIdx.x;
            double r = 0.0, m = 0.0;
#pragma unroll
            for (int q = 0; q < TH / 64; ++q) {
                r += red[q][c][0];
                m += red[q][c][1];
            }
            partial[(int64_t)(i0 + c) * DOT_BLOCKS + blockIdx.x] = make_double2(r, m);
        }
        __syncthreads();
    }
}

static void launch_dots_partial(const double2* V, int64_t ldv, int nvec, const double2* w, int64_t n, double2* partial,
                                hipStream_t st) {
#define HTN_DP(CHV, THV) hipLaunchKernelGGL((k_dots_partial<CHV, THV>), dim3(DOT_BLOCKS), dim3(THV), 0, st, V, ldv, nvec, w, n, partial)
    HTN_CH_DISPATCH(nvec, dot_slice(n), HTN_DP);
#undef HTN_DP
}

// one wave per vector: out[i] = sum_b partial[i][b]
__global__ void k_dots_reduce(const double2* __restrict__ partial, int nvec, double2* __restrict__ out) {
    const int i = blockIdx.x;
    const double2 r = reduce_partials(partial + (int64_t)i * DOT_BLOCKS, threadIdx.x);
    if (threadIdx.x == 0) out[i] = r;
}

// fused: c = reduce(partial); w += sign * V c ; norm_partial[b] = |w_new slice|^2 ; block 0 writes c[c_index]
// to *c_out (a device scalar: the next matvec launch hands it to the host inside the step record).  CH = vectors per pass
// (compile time): all basis loads of an element are in flight together (the earlier version walked them eight at a time, one
// memory round trip per group of eight).
template <int CH, int TH>
__global__ __launch_bounds__(TH) void k_axpy_norm(double2* __restrict__ w, const double2* __restrict__ V,
                                                           int64_t ldv, int nvec,
                                                           const double2* __restrict__ partial,
                                                           double2* __restrict__ c_out, int c_index, double sign,
                                                           int64_t n, double* __restrict__ norm_partial) {
    __shared__ double cs[64][2];
    __shared__ double red[TH / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < nvec; i += TH / 64) {
        const double2 r = reduce_partials(partial + (int64_t)i * DOT_BLOCKS, lane);
        if (lane == 0) {
            cs[i][0] = r.x;
            cs[i][1] = r.y;
            if (blockIdx.x == 0 && i == c_index) *c_out = r;
        }
    }
    if (tid >= nvec && tid < 64) cs[tid][0] = cs[tid][1] = 0.0;      // padding coefficients for the unrolled loop
    __syncthreads();
    const int64_t per = (n + DOT_BLOCKS - 1) / DOT_BLOCKS;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    double nn = 0.0;
    for (int64_t j = lo + tid; j < hi; j += TH) {
        double2 v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = c < nvec ? c : nvec - 1;
            v[c] = V[(int64_t)i * ldv + j];
        }
        double2 x = w[j];
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {                    // cs[c] = 0 beyond nvec
            sr += cs[c][0] * v[c].x - cs[c][1] * v[c].y;
            si += cs[c][0] * v[c].y + cs[c][1] * v[c].x;
        }
        x.x += sign * sr;
        x.y += sign * si;
        w[j] = x;
        nn += x.x * x.x + x.y * x.y;
    }
    nn = wave_sum(nn);
    if (lane == 0) red[wave] = nn;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < TH / 64; ++q) t += red[q];
        norm_partial[blockIdx.x] = t;
    }
}

static void launch_axpy_norm(double2* w, const double2* V, int64_t ldv, int nvec, const double2* partial, double2* c_out, int c_index,
                             double sign, int64_t n, double* norm_partial, hipStream_t st) {
#define HTN_AN(CHV, THV)                                                                                                    \
    hipLaunchKernelGGL((k_axpy_norm<CHV, THV>), dim3(DOT_BLOCKS), dim3(THV), 0, st, w, V, ldv, nvec, partial, c_out, c_index, \
                       sign, n, norm_partial)
    HTN_CH_DISPATCH(nvec, dot_slice(n), HTN_AN);
#undef HTN_AN
}

// first Gram-Schmidt update FUSED with the second pass's dots: c = reduce(partial_in); w += sign * V c; and, while the
// basis values of the element are still in registers, partial_out[i][b] = sum over slice b of conj(V_i) * w_new.
// One read of the Krylov basis instead of two (the basis is the traffic of a Lanczos step: (j+1) x |theta|).
// 256 threads = one wave per SIMD, so the CH basis values + 2 CH accumulators per thread fit the register file.
// DEFERRED NORMALISATION (norm_prev != nullptr): the newest basis row V[nvec-1] is still the raw, unnormalised vector the
// previous step's second update left behind (|raw|^2 = sum norm_prev = beta^2), and w = H raw = beta H v.  This kernel reads
// that row anyway, so it does the previous step's normalisation on the way: s = 1 / beta, the row is written back as
// v = s raw, w is taken as s w, and the dots of K2 (taken with the raw row and the raw w) become c_i = s P_i (i < nvec-1),
// c_last = s^2 P_last.  The separate scale kernel of every step is gone; its record is published by the next matvec launch.
template <int CH, int TH>
__global__ __launch_bounds__(TH) void k_axpy_dots(double2* __restrict__ w, double2* __restrict__ V,
                                                           int64_t ldv, int nvec,
                                                           const double2* __restrict__ partial_in,
                                                           double2* __restrict__ c_out, int c_index, double sign,
                                                           int64_t n, double2* __restrict__ partial_out,
                                                           const double* __restrict__ norm_prev, int upd0) {
    __shared__ double cs[64][2];
    __shared__ double red[TH / 64][CH][2];
    __shared__ double s_scale;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int last = nvec - 1;
    if (tid < 64) {                                   // (wave 0) scale of the deferred normalisation, same order as every norm sum
        double t = 1.0;
        if (norm_prev) {
            t = 0.0;
#pragma unroll
            for (int b = 0; b < DOT_BLOCKS / 64; ++b) t += norm_prev[tid + 64 * b];
            t = wave_sum(t);
            t = t > 0.0 ? 1.0 / sqrt(t) : 0.0;
        }
        if (tid == 0) s_scale = t;
    }
    __syncthreads();
    const double s = s_scale;
    // partial_in holds the dots of rows upd0 .. nvec-1 only (the three-term first pass: upd0 = nvec - 2); the rows below
    // take no part in this update (coefficient 0) but do in the dots taken on the way
    for (int i = upd0 + wave; i < nvec; i += TH / 64) {
        double2 r = reduce_partials(partial_in + (int64_t)(i - upd0) * DOT_BLOCKS, lane);
        const bool raw_row = norm_prev && i == last;   // the row this coefficient multiplies is still unnormalised
        const double f = raw_row ? s * s : s;
        r.x *= f;
        r.y *= f;
        if (lane == 0) {
            // the update below uses the rows as stored: the raw row's coefficient carries its factor s
            cs[i][0] = raw_row ? r.x * s : r.x;
            cs[i][1] = raw_row ? r.y * s : r.y;
            if (blockIdx.x == 0 && i == c_index) *c_out = r;
        }
    }
    if ((tid >= nvec || tid < upd0) && tid < 64) cs[tid][0] = cs[tid][1] = 0.0;
    __syncthreads();
    const bool fix = norm_prev != nullptr;
    const int64_t per = (n + DOT_BLOCKS - 1) / DOT_BLOCKS;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    double ar[CH], ai[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) ar[c] = ai[c] = 0.0;
    for (int64_t j = lo + tid; j < hi; j += TH) {
        double2 v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = c < nvec ? c : nvec - 1;
            v[c] = V[(int64_t)i * ldv + j];
        }
        if (fix) {                                    // (uniform) the newest row goes back normalised; v[] keeps it RAW: its factor s
            const double2 vl = V[(int64_t)last * ldv + j];      // sits in its coefficient (above) and in its dot (below) -- no per-row select
            V[(int64_t)last * ldv + j] = make_double2(vl.x * s, vl.y * s);
        }
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {                // cs[c] = 0 beyond nvec
            sr += cs[c][0] * v[c].x - cs[c][1] * v[c].y;
            si += cs[c][0] * v[c].y + cs[c][1] * v[c].x;
        }
        double2 x = w[j];
        x.x = fma(x.x, s, sign * sr);
        x.y = fma(x.y, s, sign * si);
        w[j] = x;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            ar[c] += v[c].x * x.x + v[c].y * x.y;     // conj(v) * w_new
            ai[c] += v[c].x * x.y - v[c].y * x.x;
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const double r = wave_sum(ar[c]);
        const double m = wave_sum(ai[c]);
        if (lane == 0) {
            red[wave][c][0] = r;
            red[wave][c][1] = m;
        }
    }
    __syncthreads();
    if (tid < CH && tid < nvec) {                     // fixed order over the 4 waves: deterministic
        double r = 0.0, m = 0.0;
#pragma unroll
        for (int q = 0; q < TH / 64; ++q) {
            r += red[q][tid][0];
            m += red[q][tid][1];
        }
        if (fix && tid == last) {                     // <s raw, w> = s <raw, w>
            r *= s;
            m *= s;
        }
        partial_out[(int64_t)tid * DOT_BLOCKS + blockIdx.x] = make_double2(r, m);
    }
}

static void launch_axpy_dots(double2* w, double2* V, int64_t ldv, int nvec, const double2* partial_in, double2* c_out,
                             int c_index, double sign, int64_t n, double2* partial_out, const double* norm_prev, int upd0,
                             hipStream_t st) {
#define HTN_AD(CHV, THV)                                                                                           \
    hipLaunchKernelGGL((k_axpy_dots<CHV, THV>), dim3(DOT_BLOCKS), dim3(THV), 0, st, w, V, ldv, nvec, partial_in, c_out, \
                       c_index, sign, n, partial_out, norm_prev, upd0)
    HTN_CH_DISPATCH(nvec, dot_slice(n), HTN_AD);
#undef HTN_AD
}

// dst = src / sqrt(sum norm_partial); with rec_out, block 0 publishes the step record {c1[0].re, c2[0].re, |w|^2}
// (c1 / c2: device scalars written by the EARLIER kernels of the step -- same stream, hence complete and visible)
__global__ __launch_bounds__(DOT_THREADS) void k_scale_by_norm(double2* __restrict__ dst, const double2* __restrict__ src,
                                                               const double* __restrict__ norm_partial, int64_t n,
                                                               LanRecord* __restrict__ rec_out,
                                                               const double2* __restrict__ c1, const double2* __restrict__ c2,
                                                               unsigned long long serial) {
    __shared__ double s_inv;
    const int tid = threadIdx.x;
    if (tid < 64) {
        double t = 0.0;
#pragma unroll
        for (int b = 0; b < DOT_BLOCKS / 64; ++b) t += norm_partial[tid + 64 * b];
        t = wave_sum(t);
        if (tid == 0) {
            s_inv = t > 0.0 ? 1.0 / sqrt(t) : 0.0;
            if (blockIdx.x == 0 && rec_out) {
                const unsigned long long w0 = (unsigned long long)__double_as_longlong(c1[0].x);
                const unsigned long long w1 = (unsigned long long)__double_as_longlong(c2[0].x);
                const unsigned long long w2 = (unsigned long long)__double_as_longlong(t);
                ulonglong2* out = (ulonglong2*)rec_out;
                out[0] = make_ulonglong2(w0, w1);
                out[1] = make_ulonglong2(w2, lan_check(w0, w1, w2, serial));
                __threadfence_system();
            }
        }
    }
    __syncthreads();
    const double s = s_inv;
    for (int64_t j = (int64_t)blockIdx.x * DOT_THREADS + tid; j < n; j += (int64_t)gridDim.x * DOT_THREADS) {
        const double2 v = src[j];
        dst[j] = make_double2(v.x * s, v.y * s);
    }
}

// the record of a step that no further matvec launch follows (the last step of a restart cycle): one wave
__global__ __launch_bounds__(64) void k_publish_record(const double* __restrict__ norm_partial, const double2* __restrict__ c1,
                                                       const double2* __restrict__ c2, LanRecord* __restrict__ rec_out,
                                                       unsigned long long serial) {
    const int tid = threadIdx.x;
    double t = 0.0;
#pragma unroll
    for (int b = 0; b < DOT_BLOCKS / 64; ++b) t += norm_partial[tid + 64 * b];
    t = wave_sum(t);
    if (tid == 0) {
        const unsigned long long w0 = (unsigned long long)__double_as_longlong(c1[0].x);
        const unsigned long long w1 = (unsigned long long)__double_as_longlong(c2[0].x);
        const unsigned long long w2 = (unsigned long long)__double_as_longlong(t);
        ulonglong2* out = (ulonglong2*)rec_out;
        out[0] = make_ulonglong2(w0, w1);
        out[1] = make_ulonglong2(w2, lan_check(w0, w1, w2, serial));
        __threadfence_system();
    }
}

// |w|^2 partials only (start-vector normalisation)
__global__ __launch_bounds__(DOT_THREADS) void k_norm_partial(const double2* __restrict__ w, int64_t n,
                                                              double* __restrict__ norm_partial) {
    __shared__ double red[DOT_THREADS / 64];
    const int tid = threadIdx.x;
    const int64_t per = (n + DOT_BLOCKS - 1) / DOT_BLOCKS;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    double nn = 0.0;
    for (int64_t j = lo + tid; j < hi; j += DOT_THREADS) {
        const double2 x = w[j];
        nn += x.x * x.x + x.y * x.y;
    }
    nn = wave_sum(nn);
    if ((tid & 63) == 0) red[tid >> 6] = nn;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int q = 0; q < DOT_THREADS / 64; ++q) t += red[q];
        norm_partial[blockIdx.x] = t;
    }
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

__global__ __launch_bounds__(256) void k_scale_inv_sqrt(double2* __restrict__ dst, const double2* __restrict__ src,
                                                        const double2* __restrict__ nrm2, int64_t n) {
    const double s = 1.0 / sqrt(nrm2[0].x);
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = src[j];
        dst[j] = make_double2(v.x * s, v.y * s);
    }
}

static inline int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" int64_t htn_dots_scratch_elems(int32_t nvec) { return (int64_t)nvec * DOT_BLOCKS; }

extern "C" int htn_dots_z(const void* V, int64_t ldv, int32_t nvec, const void* w, int64_t n, void* out,
                          void* scratch, void* stream) {
    if (nvec <= 0) return 0;
    if (nvec > 64) return fail_msg("htn_dots_z: nvec > 64");
    launch_dots_partial((const double2*)V, ldv, nvec, (const double2*)w, n, (double2*)scratch, (hipStream_t)stream);
    hipLaunchKernelGGL(k_dots_reduce, dim3(nvec), dim3(64), 0, (hipStream_t)stream, (const double2*)scratch,
                       nvec, (double2*)out);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int htn_axpys_z(void* w, const void* V, int64_t ldv, int32_t nvec, const void* coef, double sign,
                           int64_t n, void* stream) {
    if (nvec <= 0 || n <= 0) return 0;
    hipLaunchKernelGGL(k_axpys, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (double2*)w, (const double2*)V,
                       ldv, nvec, (const double2*)coef, sign, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int htn_scale_inv_sqrt_z(void* dst, const void* src, const void* nrm2, int64_t n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_scale_inv_sqrt, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (double2*)dst,
                       (const double2*)src, (const double2*)nrm2, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ----------------------------------------------------------------------------------------------
// host side of the Lanczos driver
// ----------------------------------------------------------------------------------------------
// lowest eigenpair of the symmetric tridiagonal (alpha, beta): bisection on the Sturm sequence for the
// eigenvalue, then inverse iteration with a shift just below it (T - mu I is positive definite, so the
// LDL^T tridiagonal solve needs no pivoting).  O(k) per step: microseconds for k <= 64.
static int sturm_count(const std::vector<double>& a, const std::vector<double>& b, double x) {
    int cnt = 0;
    double d = 1.0;
    const int k = (int)a.size();
    for (int i = 0; i < k; ++i) {
        const double off = i ? b[i - 1] * b[i - 1] : 0.0;
        d = (a[i] - x) - (i ? off / d : 0.0);
        if (d == 0.0) d = 1e-300;
        if (d < 0.0) ++cnt;
    }
    return cnt;        // number of eigenvalues < x
}

static void tridiag_lowest(const std::vector<double>& alpha, const std::vector<double>& beta, double* eig,
                           std::vector<double>& vec) {
    const int k = (int)alpha.size();
    double lo = alpha[0], hi = alpha[0];
    for (int i = 0; i < k; ++i) {
        const double r = (i ? fabs(beta[i - 1]) : 0.0) + (i + 1 < k ? fabs(beta[i]) : 0.0);
        lo = fmin(lo, alpha[i] - r);
        hi = fmax(hi, alpha[i] + r);
    }
    const double scale = fmax(fabs(lo), fabs(hi)) + 1e-300;
    hi = hi + 1e-12 * scale;
    for (int it = 0; it < 200 && hi - lo > 4e-16 * scale; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (sturm_count(alpha, beta, mid) >= 1) hi = mid;
        else lo = mid;
    }
    const double lam = 0.5 * (lo + hi);
    *eig = lam;
    vec.assign(k, 1.0 / sqrt((double)k));
    if (k == 1) {
        vec[0] = 1.0;
        return;
    }
    const double mu = lam - 1e-10 * scale;
    std::vector<double> d(k), l(k), z(k);
    d[0] = alpha[0] - mu;
    for (int i = 1; i < k; ++i) {
        l[i] = beta[i - 1] / d[i - 1];
        d[i] = (alpha[i] - mu) - l[i] * beta[i - 1];
        if (d[i] == 0.0) d[i] = 1e-300;
    }
    for (int i = 0; i < k; ++i) vec[i] = (i & 1) ? -0.7 : 1.0;   // not orthogonal to the lowest vector in practice
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
}

extern "C" int64_t htn_lanczos_scratch_elems(int32_t krylovdim) {
    // 2 x partial sums (krylovdim+1 vectors) + c1 + c2 + y (each krylovdim+1) + norm partials (as doubles)
    return 2 * (int64_t)(krylovdim + 1) * DOT_BLOCKS + 3 * (krylovdim + 1) + DOT_BLOCKS + 8;
}

// ---- per-stream resources of the driver (htn_common.h: owned by the stream's registry entry) ----------------------------
#define LAN_SLOTS 64
namespace {
struct LanRes {
    int device = -1;
    LanRecord* h_rec = nullptr;      // [LAN_SLOTS] host view  \ hipHostMallocMapped | hipHostMallocCoherent: written by the
    LanRecord* d_rec = nullptr;      //             device view / device only, read by the host only
    double2* h_y = nullptr;          // Ritz coefficients, host -> device staging (a SEPARATE allocation)
    hipEvent_t ev_done[LAN_SLOTS], ev_mv0[LAN_SLOTS], ev_mv1[LAN_SLOTS];
    bool have_events = false;
    unsigned long long serial = 0;   // last step serial handed out; never reused within the life of the entry
    unsigned sample = 0;             // phase of the 1-in-8 matvec timing sample
    ~LanRes() {
        if (device >= 0) (void)hipSetDevice(device);
        if (h_rec) (void)hipHostFree(h_rec);
        if (h_y) (void)hipHostFree(h_y);
        if (have_events)
            for (int i = 0; i < LAN_SLOTS; ++i) {
                (void)hipEventDestroy(ev_done[i]);
                (void)hipEventDestroy(ev_mv0[i]);
                (void)hipEventDestroy(ev_mv1[i]);
            }
    }
};
std::mutex g_lan_mu;
std::map<hipStream_t, std::unique_ptr<LanRes>> g_lan_res;

int lan_res_get(hipStream_t st, LanRes** out) {
    std::lock_guard<std::mutex> lk(g_lan_mu);
    auto& slot = g_lan_res[st];
    if (!slot) {
        auto r = std::make_unique<LanRes>();
        HIP_TRY(hipGetDevice(&r->device));
        HIP_TRY(hipHostMalloc((void**)&r->h_rec, sizeof(LanRecord) * LAN_SLOTS, hipHostMallocMapped | hipHostMallocCoherent));
        memset(r->h_rec, 0, sizeof(LanRecord) * LAN_SLOTS);      // (before any device work can see the block)
        HIP_TRY(hipHostGetDevicePointer((void**)&r->d_rec, r->h_rec, 0));
        HIP_TRY(hipHostMalloc((void**)&r->h_y, sizeof(double2) * LAN_SLOTS, hipHostMallocDefault));
        for (int i = 0; i < LAN_SLOTS; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&r->ev_done[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreate(&r->ev_mv0[i]));
            HIP_TRY(hipEventCreate(&r->ev_mv1[i]));
        }
        r->have_events = true;
        slot = std::move(r);
    }
    *out = slot.get();
    return 0;
}
}  // namespace

void htn_krylov_release_stream(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_lan_mu);
    g_lan_res.erase(st);
}

extern "C" int htn_lanczos_z(const htn_gemm_launch* stages, int32_t n_stages, int32_t x_slot, int32_t y_slot,
                             void* Vv, int64_t n, int32_t krylovdim, double tol, int32_t max_restart,
                             void* scratch, int32_t zero_y, htn_exchange2_fn exchange, void* user,
                             double* eig_host, int32_t* n_matvec_host, double* residual_host,
                             double* matvec_ms_host, void* stream_v) {
    hipStream_t st = (hipStream_t)stream_v;
    const int kd = krylovdim;
    // (k_axpy_dots keeps one basis value per Krylov vector in registers: at most DOT_CHUNK = 32 vectors per step)
    if (kd < 2 || kd + 1 > DOT_CHUNK) return fail_msg("htn_lanczos_z: krylovdim must be in 2..31");
    double2* V = (double2*)Vv;
    double2* partial = (double2*)scratch;
    double2* partial2 = partial + (int64_t)(kd + 1) * DOT_BLOCKS;
    double2* c1 = partial2 + (int64_t)(kd + 1) * DOT_BLOCKS;      // device scalars <v_j, w>: first / second pass
    double2* c2 = c1 + (kd + 1);
    double2* ycoef = c2 + (kd + 1);
    double* norm_partial = (double*)(ycoef + (kd + 1));

    LanRes* R = nullptr;
    if (lan_res_get(st, &R)) return 1;
    const bool event_waits = htn_debug_event_waits();
    static const bool full_first_pass = htn_env_flag("HTN_LANCZOS_FULL_FIRST_PASS");

    bool timed_step[LAN_SLOTS] = {false};
    unsigned long long step_serial[LAN_SLOTS] = {0};
    auto matvec = [&](double2* x, double2* y, const HtnGemmPublish* pub) -> int {
        if (zero_y) HIP_TRY(hipMemsetAsync(y, 0, sizeof(double2) * n, st));
        bool published = pub == nullptr;
        for (int s = 0; s < n_stages; ++s) {
            const void* bufs[HTN_MAX_BUFS];
            for (int b = 0; b < HTN_MAX_BUFS; ++b) bufs[b] = stages[s].bufs[b];
            bufs[x_slot] = x;
            bufs[y_slot] = y;
            const bool here = !published && stages[s].n_tiles > 0;        // the first launch of the matvec carries the record
            if (htn_grouped_gemm_launch(bufs, stages[s].tiles, stages[s].n_tiles, stages[s].segs, here ? pub : nullptr, st)) return 1;
            published |= here;
        }
        if (!published)         // (a matvec without a single tile: publish by the one-wave kernel)
            hipLaunchKernelGGL(k_publish_record, dim3(1), dim3(64), 0, st, pub->norm_partial, pub->c1, pub->c2, pub->rec_out, pub->serial);
        if (exchange && exchange(y, n, user)) return fail_msg("htn_lanczos_z: the exchange hook reported a failure");
        return 0;
    };
    // One Lanczos step, fully enqueued: w = H V[j]; two Gram-Schmidt passes against V[0..j]; the result stays UNNORMALISED in
    // V[j+1] with its squared norm in norm_partial.  For j > 0 (inside a cycle) V[j] itself is still the raw vector of step
    // j - 1: its normalisation is folded into this step's first update (k_axpy_dots, norm_prev) and the record of step j - 1 is
    // published by this step's first matvec launch -- four kernels per step instead of five.  The last step of a cycle
    // publishes its own record (nothing follows it).  `first` = V[j] is already normalised (start / Ritz vector).
    auto enqueue_step = [&](int j, bool first) -> int {
        double2* vj = V + (int64_t)j * n;
        double2* w = V + (int64_t)(j + 1) * n;
        // HIP events around a SAMPLE of the matvec launches (every 8th): an event is a marker packet that costs ~5 us
        // of pipeline bubble on this part, three of them per step were 15 us of a ~105 us Lanczos step
        const bool timed = matvec_ms_host && (R->sample % 8) == 0;
        timed_step[j] = timed;
        ++R->sample;
        step_serial[j] = ++R->serial;
        HtnGemmPublish pub = {nullptr, nullptr, nullptr, nullptr, 0ull};
        if (!first) pub = {norm_partial, (const double2*)(c1 + (j - 1)), (const double2*)(c2 + (j - 1)), R->d_rec + (j - 1), step_serial[j - 1]};
        if (timed) HIP_TRY(hipEventRecord(R->ev_mv0[j], st));
        if (matvec(vj, w, first ? nullptr : &pub)) return 1;
        if (timed) HIP_TRY(hipEventRecord(R->ev_mv1[j], st));
        if (!first && event_waits) HIP_TRY(hipEventRecord(R->ev_done[j - 1], st));
        // Orthogonalisation in three kernels and TWO passes over the basis.  First the three-term part: dots with V[j-1]
        // and V[j] only, w' = w - V[j-1] c - V[j] c (against a basis that is orthonormal to rounding every other component
        // of H v_j is O(eps |H|): <v_i, H v_j> = conj(<v_j, H v_i>) and H v_i has no component along v_j for j > i + 1);
        // the kernel that makes w' takes the dots of w' with ALL rows on the way, and the last kernel subtracts those --
        // a full classical Gram-Schmidt pass whose coefficients are O(eps |H|) and whose result is orthogonal to eps |w'|,
        // which is what the second pass of a two-pass scheme delivers.  (Until round 3 the first pass was a full one as well:
        // one more read of the whole basis per step; HTN_LANCZOS_FULL_FIRST_PASS=1 brings it back for comparison.)
        const int upd0 = full_first_pass || j < 2 ? 0 : j - 1;
        launch_dots_partial(V + (int64_t)upd0 * n, n, j + 1 - upd0, w, n, partial, st);
        launch_axpy_dots(w, V, n, j + 1, partial, c1 + j, j, -1.0, n, partial2, first ? (const double*)nullptr : norm_partial, upd0, st);
        launch_axpy_norm(w, V, n, j + 1, partial2, c2 + j, j, -1.0, n, norm_partial, st);
        if (j == kd - 1) {
            hipLaunchKernelGGL(k_publish_record, dim3(1), dim3(64), 0, st, (const double*)norm_partial, (const double2*)(c1 + j),
                               (const double2*)(c2 + j), R->d_rec + j, step_serial[j]);
            if (event_waits) HIP_TRY(hipEventRecord(R->ev_done[j], st));
        }
        return 0;
    };
    // read slot j if it holds the record of THIS step (see LanRecord): true = accepted
    auto read_record = [&](int j, double* a1, double* a2, double* nn) -> bool {
        const volatile unsigned long long* p = R->h_rec[j].w;
        const unsigned long long w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
        if (w3 != lan_check(w0, w1, w2, step_serial[j])) return false;
        memcpy(a1, &w0, 8);
        memcpy(a2, &w1, 8);
        memcpy(nn, &w2, 8);
        return true;
    };
    // wait for step j's record.  Product path: poll the slot (the host wakes within the PCIe latency of the store; a
    // sleeping wait costs tens of us per step).  A stream that has drained without the record arriving means the step never
    // ran to its end (a fault): report instead of spinning forever.  HTN_DEBUG_EVENT_WAITS: wait for the step's event, after
    // which the record MUST validate.
    auto wait_step = [&](int j, double* a1, double* a2, double* nn) -> int {
        if (event_waits) {
            HIP_TRY(hipEventSynchronize(R->ev_done[j]));
            if (!read_record(j, a1, a2, nn)) return fail_msg("htn_lanczos_z: the step's event completed but its record does not validate");
            return 0;
        }
        for (unsigned spins = 1;; ++spins) {
            if (read_record(j, a1, a2, nn)) return 0;
            __builtin_ia32_pause();
            if ((spins & 0xffff) == 0) {
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess)
                    return read_record(j, a1, a2, nn) ? 0 : fail_msg("htn_lanczos_z: the stream drained without producing the step's record");
                if (q != hipErrorNotReady) return fail("hipStreamQuery", q);
            }
        }
    };

    // normalise the start vector
    hipLaunchKernelGGL(k_norm_partial, dim3(DOT_BLOCKS), dim3(DOT_THREADS), 0, st, V, n, norm_partial);
    hipLaunchKernelGGL(k_scale_by_norm, dim3(grid_for(n)), dim3(DOT_THREADS), 0, st, V, V, norm_partial, n,
                       (LanRecord*)nullptr, (const double2*)nullptr, (const double2*)nullptr, 0ull);
    double mv_ms = 0.0;
    int nmv = 0, n_timed = 0;
    static const bool dbg_timers = getenv("HTN_DEBUG_HOST_TIMERS") != nullptr;
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_enq = 0.0, t_wait = 0.0, t_host = 0.0;
    const double t_begin = now_us();
    double theta = 0.0, res = 0.0, beta = 0.0, amax = 0.0;
    std::vector<double> y;
    for (int restart = 0; restart <= max_restart; ++restart) {
        std::vector<double> alphas, betas;
        // Software pipeline of depth 1: step j+1 is enqueued BEFORE the host waits for step j's record, so the
        // GPU never idles during the host's convergence test.  If step j converges, step j+1 was speculative:
        // it only wrote Krylov row j+2, device scalars j+1 and record slot j+1, which nothing reads afterwards.
        double tq = now_us();
        if (enqueue_step(0, true)) return 1;
        t_enq += now_us() - tq;
        ++nmv;
        for (int j = 0; j < kd; ++j) {
            if (j + 1 < kd) {
                tq = now_us();
                if (enqueue_step(j + 1, false)) return 1;
                t_enq += now_us() - tq;
                ++nmv;
            }
            double a1 = 0.0, a2 = 0.0, nn = 0.0;
            tq = now_us();
            if (wait_step(j, &a1, &a2, &nn)) return 1;
            t_wait += now_us() - tq;
            tq = now_us();
            if (timed_step[j]) {
                float ms = 0.f;
                HIP_TRY(htn_event_spin(R->ev_mv1[j]));
                HIP_TRY(hipEventElapsedTime(&ms, R->ev_mv0[j], R->ev_mv1[j]));
                mv_ms += ms;
                ++n_timed;
            }
            const double alpha = a1 + a2;
            if (!(alpha == alpha) || !(nn == nn)) return fail_msg("htn_lanczos_z: NaN in the tridiagonal coefficients (operator or start vector not finite)");
            beta = sqrt(nn > 0.0 ? nn : 0.0);
            alphas.push_back(alpha);
            tridiag_lowest(alphas, betas, &theta, y);
            res = fabs(beta * y.back());
            amax = std::max(amax, std::max(fabs(alpha), beta));
            // invariant subspace: beta negligible RELATIVE to the scale of the tridiagonal matrix
            t_host += now_us() - tq;
            if (res < tol || beta < 1e-14 * std::max(amax, 1e-300) || j == kd - 1) break;
            betas.push_back(beta);
        }
        HIP_TRY(htn_stream_spin(st));       // drain the speculative step before rows are reused
        // x = sum_i y_i V_i  -> scratch row kd+1, normalised into row 0
        const int k = (int)y.size();
        for (int i = 0; i < k; ++i) R->h_y[i] = make_double2(y[i], 0.0);
        HIP_TRY(hipMemcpyAsync(ycoef, R->h_y, sizeof(double2) * k, hipMemcpyHostToDevice, st));
        double2* xrow = V + (int64_t)(kd + 1) * n;
        HIP_TRY(hipMemsetAsync(xrow, 0, sizeof(double2) * n, st));
        hipLaunchKernelGGL(k_axpys, dim3(grid_for(n)), dim3(256), 0, st, xrow, V, n, k, ycoef, 1.0, n);
        hipLaunchKernelGGL(k_norm_partial, dim3(DOT_BLOCKS), dim3(DOT_THREADS), 0, st, xrow, n, norm_partial);
        hipLaunchKernelGGL(k_scale_by_norm, dim3(grid_for(n)), dim3(DOT_THREADS), 0, st, V, xrow, norm_partial, n,
                           (LanRecord*)nullptr, (const double2*)nullptr, (const double2*)nullptr, 0ull);
        HIP_TRY(htn_stream_spin(st));      // h_y is reused by the next restart / call
        if (res < tol || beta < 1e-14 * std::max(amax, 1e-300)) break;
    }
    HIP_TRY(hipGetLastError());
    if (dbg_timers)
        fprintf(stderr, "lanczos n=%lld: %d matvecs in %.1f us: enqueue %.1f, waiting for records %.1f, tridiagonal %.1f\n", (long long)n, nmv,
                now_us() - t_begin, t_enq, t_wait, t_host);
    *eig_host = theta;
    *n_matvec_host = nmv;
    *residual_host = res;
    // sampled launches scaled to all launches of this solve (the next solves continue the 1-in-8 sampling phase)
    if (matvec_ms_host) *matvec_ms_host = n_timed ? mv_ms * nmv / n_timed : -1.0;
    return 0;
}
