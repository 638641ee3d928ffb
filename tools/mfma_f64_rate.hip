// Developer microbenchmark: issue rate of the two fp64 matrix-core instructions of gfx950.
//   v_mfma_f64_16x16x4_f64   : one 16 x 16 x 4 product   (1024 multiply-adds)
//   v_mfma_f64_4x4x4_4b_f64  : four 4 x 4 x 4 products   ( 256 multiply-adds)
// Question: is the upper triangle of a 16 x 16 Gram tile as ten 4 x 4 blocks (2.5 instructions of the
// second kind) cheaper than one instruction of the first kind?
// build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_rate tools/mfma_f64_rate.hip && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(double *out, int iters, unsigned long long *cyc)
{
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else {
            d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d0 + d1 + d2 + d3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    double *out;
    unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 1024 * sizeof(double));
    hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int kind = 0; kind < 2; ++kind) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            // one workgroup of `waves` x 4 wavefronts per CU (256 CUs): waves wavefronts per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) rate_kernel<0><<<256 * waves, 256>>>(out, iters, cyc);
                else rate_kernel<1><<<256 * waves, 256>>>(out, iters, cyc);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double per = (double)h / (4.0 * iters);
            const double macs = kind == 0 ? 1024.0 : 256.0;
            printf("%s, %d wavefront(s) per SIMD: %.1f shader-clock ticks per instruction of one wavefront, %.3f ms, "
                   "%.1f TFLOP/s chip-wide\n", kind == 0 ? "16x16x4   " : "4x4x4 (4b)", waves, per, ms,
                   2.0 * macs * 4.0 * iters * 1024.0 * waves / (ms * 1e-3) / 1e12);
        }
    return 0;
}
