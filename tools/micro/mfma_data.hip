// Stand-alone probe: does the DATA fed to v_mfma_f32_16x16x4_f32 change the sustained rate (power -> clock)?
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_data tools/micro/mfma_data.hip && gpurun_out/mfma_data
// Same instruction stream in every variant (6 or 12 independent accumulators, B from LDS by ds_read_b128 each step,
// A changes every step), ~0.3 s per run so that the power controller settles:
//   DATA 0: zeros   1: small integers (few mantissa bits toggle)   2: random fp32 in (-1, 1) (all mantissa bits toggle)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float rnd(unsigned &h) {
    h = h * 1664525u + 1013904223u;
    return __uint_as_float(0x3f800000u | (h >> 9)) * 2.0f - 3.0f;  // uniform in [-1, 1)
}

template <int NACC, int DATA>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    __shared__ float lds[8192];
    unsigned h = blockIdx.x * 2654435761u + threadIdx.x * 40503u + 17u;
    for (int i = threadIdx.x; i < 8192; i += 256) {
        unsigned hh = i * 2246822519u + 3266489917u;
        lds[i] = DATA == 0 ? 0.f : (DATA == 1 ? (float)(i & 3) : rnd(hh));
    }
    __syncthreads();
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a4[4];
    for (int e = 0; e < 4; e++) a4[e] = DATA == 0 ? 0.f : (DATA == 1 ? (float)((threadIdx.x + e) & 3) : rnd(h));
    float b[NACC];
    for (int n = 0; n < NACC; n++) b[n] = 0.f;
    const int r16 = threadIdx.x & 15, g = (threadIdx.x >> 4) & 3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < NACC / 4; q++) {
            const float4 v = *reinterpret_cast<const float4 *>(lds + ((it * 256 + g * 1024 + q * 64 + 4 * r16) & 8191));
            b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
        }
        if (NACC % 4) {
            const float2 v = *reinterpret_cast<const float2 *>(lds + ((it * 256 + g * 1024 + 192 + 2 * r16) & 8191));
            b[NACC - 2] = v.x; b[NACC - 1] = v.y;
        }
        __builtin_amdgcn_sched_barrier(0);
        const float a = a4[it & 3];
#pragma unroll
        for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; n++) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int DATA>
void run(const char *name, int wgs_per_cu, int iters) {
    float *out;
    const int grid = 256 * wgs_per_cu;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, DATA><<<grid, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, DATA><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * NACC * 2048.0;
    printf("%-34s NACC=%2d %d WG/CU: %8.2f ms  %6.1f TFLOP/s  -> implied MFMA clock %.2f GHz (if the pipe never idles)\n", name, NACC,
           wgs_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 2.4);
    hipFree(out);
}

int main() {
    for (int rep = 0; rep < 2; rep++) {
        run<6, 0>("zeros", 2, 400000);
        run<6, 1>("small integers", 2, 400000);
        run<6, 2>("random fp32 in (-1,1)", 2, 400000);
        run<12, 0>("zeros", 2, 200000);
        run<12, 2>("random fp32 in (-1,1)", 2, 200000);
        run<6, 2>("random fp32 in (-1,1)", 1, 400000);
        run<6, 2>("random fp32 in (-1,1), short run", 2, 10000);
        run<6, 0>("zeros, short run", 2, 10000);
    }
    return 0;
}
