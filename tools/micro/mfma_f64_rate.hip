// microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (cycles per instruction per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, int iters, long long* cyc) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int waves_per_block, int blocks) {
    double* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * 1024 * 1024);
    (void)hipMalloc(&cyc, sizeof(long long) * 4096);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 64 * waves_per_block>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 64 * waves_per_block>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[1]; hipMemcpy(h, cyc, sizeof(long long), hipMemcpyDeviceToHost);
    double n_mfma = (double)iters * NACC;
    double flops = n_mfma * 2048.0 * waves_per_block * blocks;
    printf("NACC=%d waves/block=%d blocks=%d: %.1f clk per MFMA per wave (s_memtime), chip %.2f TFLOP/s, %.3f ms\n", NACC,
           waves_per_block, blocks, (double)h[0] / n_mfma, flops / (ms * 1e-3) / 1e12, ms);
}
int main() {
    run<1>(1, 1); run<4>(1, 1); run<4>(4, 1); run<4>(8, 1); run<4>(16, 1);
    run<4>(4, 256); run<4>(8, 256); run<4>(16, 256);
    return 0;
}
