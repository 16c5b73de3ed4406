// shared plumbing of libhubbardtn_hip.so (gfx950 only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hubbardtn_hip.h"

char* htn_err_buf();          // thread-local, 512 bytes (htn_abi.hip)

static inline int htn_fail(const char* what, hipError_t e) {
    snprintf(htn_err_buf(), 512, "%s: %s", what, hipGetErrorString(e));
    return 1;
}
static inline int htn_fail_msg(const char* what) {
    snprintf(htn_err_buf(), 512, "%s", what);
    return 1;
}
#define fail htn_fail
#define fail_msg htn_fail_msg
#define HIP_TRY(x)                                       \
    do {                                                 \
        hipError_t _e = (x);                             \
        if (_e != hipSuccess) return htn_fail(#x, _e);   \
    } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

// ---- host waits on the latency-critical path: poll, do not sleep ------------------------------------------------------
// hipStreamSynchronize / hipEventSynchronize put the calling thread to sleep and wake it by interrupt (tens of us per wait,
// several waits per bond update); the sweep driver's waits are short, so it polls instead.
static inline hipError_t htn_event_spin(hipEvent_t ev) {
    hipError_t e;
    while ((e = hipEventQuery(ev)) == hipErrorNotReady) __builtin_ia32_pause();
    return e;
}
static inline hipError_t htn_stream_spin(hipStream_t st) {
    static thread_local hipEvent_t evs[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipStreamSynchronize(st);
    if (!evs[dev] && (e = hipEventCreateWithFlags(&evs[dev], hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventRecord(evs[dev], st)) != hipSuccess) return e;
    return htn_event_spin(evs[dev]);
}


// ---- debug switches (test infrastructure; read once per process from the environment) ----------------------------------
//   HTN_DEBUG_POISON=1       HipBackend::alloc fills every block it hands out with 0xFF bytes (NaN as doubles, -1 as
//                            integers): one run exposes any kernel that consumes memory nobody wrote.
//   HTN_DEBUG_EVENT_WAITS=1  every host wait on device results is a completed HIP event (hipEventSynchronize) instead of a
//                            poll of host-mapped memory; the polled record is still validated.
static inline bool htn_env_flag(const char* name) {
    const char* v = getenv(name);
    return v && v[0] && v[0] != '0';
}
static inline bool htn_debug_poison() {
    static const bool on = htn_env_flag("HTN_DEBUG_POISON");
    return on;
}
static inline bool htn_debug_event_waits() {
    static const bool on = htn_env_flag("HTN_DEBUG_EVENT_WAITS");
    return on;
}

// ---- Lanczos step record (htn_krylov.hip), published either by k_scale_by_norm / k_publish_record or by workgroup 0 of the
// NEXT step's first grouped-GEMM launch (htn_gemm.hip) ------------------------------------------------------------------------
// One 32-byte slot per Lanczos step in host-mapped COHERENT pinned memory, written by ONE lane as two 16-byte stores.  It
// validates itself: word 3 = serial ^ mix(words 0..2), where `serial` is a number the host chose for exactly this step and
// never reuses.  The host accepts a slot only when the check reproduces the serial it expects, so neither a slot left over
// from an earlier solve, nor a half-arrived record, nor any reordering of the two stores on their way to host memory can be
// taken for the step's result; and the CPU never writes into this block (no sentinel), so no CPU store can share a cache line
// with a device store.
struct LanRecord {
    unsigned long long w[4];      // bits of <v_j,w> pass 1 (re), pass 2 (re), |w|^2 after orthogonalisation, check
};
__host__ __device__ __forceinline__ unsigned long long lan_rotl(unsigned long long x, int r) { return (x << r) | (x >> (64 - r)); }
__host__ __device__ __forceinline__ unsigned long long lan_check(unsigned long long w0, unsigned long long w1, unsigned long long w2,
                                                                 unsigned long long serial) {
    return serial ^ lan_rotl(w0, 13) ^ lan_rotl(w1, 29) ^ lan_rotl(w2, 47) ^ 0x9E3779B97F4A7C15ull;
}
#define HTN_DOT_BLOCKS 256        // partial sums per Krylov vector / per norm (htn_krylov.hip: DOT_BLOCKS)
// optional side job of a grouped-GEMM launch: workgroup 0 reduces the HTN_DOT_BLOCKS partial sums of |w|^2 the previous
// Lanczos step left behind and publishes that step's record {c1[0].re, c2[0].re, |w|^2} -- which saves the step a kernel of
// its own (the normalisation itself is folded into the next step's Gram-Schmidt pass).  rec_out == nullptr: nothing to do.
struct HtnGemmPublish {
    const double* norm_partial;
    const double2* c1;
    const double2* c2;
    LanRecord* rec_out;
    unsigned long long serial;
};
int htn_grouped_gemm_launch(const void* const* bufs_host, const htn_tile* tiles, int32_t n_tiles, const htn_seg* segs,
                            const HtnGemmPublish* pub, hipStream_t stream);

// per-stream scratch of the multi-launch drivers (htn_krylov.hip, htn_svd.hip): owned by the stream's registry entry,
// released by the backend that owns the stream (HipBackend::~HipBackend) -- nothing thread-local, nothing shared between
// two contexts.  A caller of the kernel-level ABI that brings its own stream keeps its entry until the process ends.
void htn_krylov_release_stream(hipStream_t st);
void htn_svd_release_stream(hipStream_t st);

// ---- cross-lane sums -------------------------------------------------------------------------------
// Within a row of 16 lanes the butterfly runs on DPP (pure VALU, no LDS crossbar round trip):
// quad_perm xor-1, quad_perm xor-2, row_half_mirror, row_mirror.  Across rows ds_bpermute (__shfl_xor).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double row_sum16(double v) {      // all-reduce inside each aligned 16-lane row
    v += dpp_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);    // row_half_mirror
    v += dpp_f64<0x140>(v);    // row_mirror
    return v;
}

// value of lane L (compile-time constant) as a wave-uniform scalar: two v_readlane_b32, no LDS crossbar trip
template <int L>
__device__ __forceinline__ double lane_bcast(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), L);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), L);
    return __hiloint2double(hi, lo);
}

// all-reduce over the 64 lanes (call with the whole wave active).  The four row sums are combined as
// (r0 + r1) + (r2 + r3) -- the order of the xor-16 / xor-32 butterfly this replaces, so results are
// bit-identical to it -- but read with v_readlane instead of ds_bpermute: a bpermute round trip costs
// ~120 cycles and a double needs two of them per butterfly step.
__device__ __forceinline__ double wave_sum(double v) {
    v = row_sum16(v);
    const double a = lane_bcast<0>(v) + lane_bcast<16>(v);
    const double b = lane_bcast<32>(v) + lane_bcast<48>(v);
    return a + b;
}

// ---- cross-lane max of 64-bit keys (order-preserving bit patterns of non-negative doubles | tag) -------
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned long long row_max16_u64(unsigned long long v) {
    v = umax64(v, dpp_u64<0xB1>(v));
    v = umax64(v, dpp_u64<0x4E>(v));
    v = umax64(v, dpp_u64<0x141>(v));
    v = umax64(v, dpp_u64<0x140>(v));
    return v;
}
template <int L>
__device__ __forceinline__ unsigned long long lane_bcast_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, L);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), L);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = row_max16_u64(v);
    return umax64(umax64(lane_bcast_u64<0>(v), lane_bcast_u64<16>(v)),
                  umax64(lane_bcast_u64<32>(v), lane_bcast_u64<48>(v)));
}

// ---- fast reciprocal / reciprocal square root: hardware seed + 2 Newton steps (full f64 accuracy for
// normal-range arguments; the precise library versions expand to 20-40 instructions each) ------------
__device__ __forceinline__ double fast_rcp(double a) {
    double x = __builtin_amdgcn_rcp(a);
    x = fma(fma(-a, x, 1.0), x, x);
    x = fma(fma(-a, x, 1.0), x, x);
    return x;
}
__device__ __forceinline__ double fast_rsq(double a) {
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}
