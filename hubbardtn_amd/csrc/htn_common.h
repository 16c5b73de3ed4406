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
