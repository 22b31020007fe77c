// hipcc --offload-arch=gfx950 -O3 -o mfma_rate microbench_mfma_f64_rate.hip && ./mfma_rate   (output of a run on MI355X: microbench_mfma_f64_rate.txt)
// Issue rate of v_mfma_f64_16x16x4 from registers: 26.9 ns per SIMD = 77.9 TFLOP/s, with one or two waves per SIMD
// and one to four accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(512) k(double *out, int iters, double a0, double b0)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, int wgs, const char *name)
{
    double *out;
    hipMalloc(&out, sizeof(double) * 512 * 4096);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC><<<wgs, threads>>>(out, 100, 1.0, 2.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<wgs, threads>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 64.0 / 4.0;
    const double mfma_per_simd = iters * 16.0 * waves_per_simd;
    printf("%s: %d threads x %d wgs, %d accumulators: %.2f ms, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", name, threads, wgs, NACC, ms,
           ms * 1e6 / mfma_per_simd, (double)wgs * (threads / 64) * iters * 16.0 * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main()
{
    run<4>(256, 256, "one wave per SIMD");
    run<2>(256, 256, "one wave per SIMD");
    run<1>(256, 256, "one wave per SIMD");
    run<4>(512, 256, "two waves per SIMD");
    run<2>(512, 256, "two waves per SIMD");
    run<4>(1024, 256, "four waves per SIMD");
    return 0;
}
