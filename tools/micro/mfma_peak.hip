// Stand-alone probe: sustained v_mfma_f32_16x16x4_f32 rate on gfx950 under the forward kernel's instruction mix.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/micro/mfma_peak.hip && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int WITH_LDS>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1.0f / (1 + i);
    __syncthreads();
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b[NACC];
    for (int n = 0; n < NACC; n++) b[n] = 1.f + n;
    const int r16 = threadIdx.x & 15;
    for (int it = 0; it < iters; it++) {
        if (WITH_LDS) {
            const float4 v = *reinterpret_cast<const float4 *>(lds + ((it * 64 + 4 * r16) & 8191));
            b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; n++) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int WITH_LDS>
void run(const char *name, int wgs_per_cu) {
    float *out;
    const int grid = 256 * wgs_per_cu, iters = 20000;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, WITH_LDS><<<grid, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, WITH_LDS><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * NACC * 2048.0;
    printf("%-40s %d WG/CU: %.3f ms  %.1f TFLOP/s\n", name, wgs_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<12, 0>("12 independent accumulators", 1);
    run<12, 0>("12 independent accumulators", 2);
    run<12, 0>("12 independent accumulators", 3);
    run<6, 0>("6 independent accumulators", 1);
    run<6, 0>("6 independent accumulators", 2);
    run<4, 0>("4 independent accumulators", 2);
    run<2, 0>("2 independent accumulators", 2);
    run<2, 0>("2 independent accumulators", 4);
    run<12, 1>("12 acc + ds_read_b128 per step", 2);
    run<6, 1>("6 acc + ds_read_b128 per step", 2);
    return 0;
}
