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

// A mix like the fused 16-lane hull kernel's: per iteration ONE large matrix-core instruction (64 cycles of the pipe) and NV
// independent fp64 vector FMAs (4 issue cycles each) -- NV = 32 keeps the matrix pipe busy about a third of the time.
// Question: does the clock fall under such a mix as it does under a pure stream of the large instruction?
template <int NV, bool SMALL>
__global__ __launch_bounds__(256) void mix_kernel(double *out, int iters, unsigned long long *cyc)
{
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    f64x4 c0 = {0, 0, 0, 0};
    double d0 = 0;
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (SMALL) {   // the same multiply-adds as four small instructions
#pragma unroll
            for (int r = 0; r < 4; ++r) d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_fma(v[k & 7], b, a);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double sum = c0[0] + c0[1] + d0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
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
    // the mix, three wavefronts per SIMD (as the fused kernel runs): effective clock = counter ticks / elapsed time
    for (int kind = 0; kind < 3; ++kind) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int it2 = 40000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) mix_kernel<32, false><<<256 * 3, 256>>>(out, it2, cyc);
            else if (kind == 1) mix_kernel<32, true><<<256 * 3, 256>>>(out, it2, cyc);
            else mix_kernel<0, false><<<256 * 3, 256>>>(out, it2, cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%s, 3 wavefronts per SIMD: %.0f ticks per iteration of one wavefront, %.3f ms, effective clock %.2f GHz\n",
               kind == 0 ? "mix: 1 large MFMA + 32 fp64 FMAs" : (kind == 1 ? "mix: 4 small MFMAs + 32 fp64 FMAs" : "pure large MFMA, one chain     "),
               (double)h / it2, ms, (double)h / (ms * 1e-3) / 1e9);
    }
    return 0;
}
