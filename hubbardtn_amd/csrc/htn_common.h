// shared plumbing of libhubbardtn_hip.so (gfx950 only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
