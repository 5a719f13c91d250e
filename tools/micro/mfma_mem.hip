// Stand-alone probe (kernel work, not product): what do vector-memory instructions, LDS-DMA pieces and workgroup
// barriers cost a wave that is otherwise issuing v_mfma_f32_16x16x4_f32 back to back, at 1-3 workgroups per CU?
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_mem tools/micro/mfma_mem.hip && gpurun_out/mfma_mem
// One iteration = 24 MFMAs (6 column tiles x 4 steps, B operands from LDS as in the conv tile kernel) preceded by
//   G  row gathers   (global_load_dwordx4: 16 rows x 64 B per wave-instruction, consumed one iteration later)
//   D  LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB per wave-instruction, into the weight buffer)
//   BAR a workgroup barrier (raw s_barrier behind vmcnt(0) when D > 0 else plain)
// The table rows come from a window of `rows` rows (small = L2 resident, large = beyond the Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int G, int D, int BAR, int PAT = 0>
__global__ __launch_bounds__(256) void k(const float *__restrict__ table, unsigned row_mask, const float *__restrict__ wsrc,
                                         float *out, int iters) {
    extern __shared__ float lds[];  // [2][3072] weight slices (32 planes x 96 columns)
    for (int i = threadIdx.x; i < 2 * 3072; i += 256) lds[i] = 1.0f / (1 + (i & 1023));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    f32x4 acc[6];
    for (int n = 0; n < 6; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a_c = make_float4(1.f, 2.f, 3.f, 4.f), a_n[G > 0 ? G : 1];
    for (int q = 0; q < (G > 0 ? G : 1); q++) a_n[q] = a_c;
    unsigned h = blockIdx.x * 2654435761u + wave * 40503u + r16 * 9973u;
    for (int it = 0; it < iters; it++) {
        if (G > 0) {
            a_c = a_n[0];
#pragma unroll
            for (int q = 1; q < G; q++) asm volatile("" ::"v"(a_n[q].x), "v"(a_n[q].y), "v"(a_n[q].z), "v"(a_n[q].w));
        }
        if (BAR) {
            if (D > 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < D; q++) {
            const float *gp = wsrc + ((size_t)((it * 4 + wave) * D + q) % 1024) * 256 + lane * 4;
            float *lp = lds + ((it + 1) & 1) * 3072 + ((wave * D + q) % 12) * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                             (__attribute__((address_space(3))) void *)lp, 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < G; q++) {
            h = h * 1664525u + 1013904223u;
            if (PAT == 0) {         // 16 rows x 64 B per wave-instruction (the conv kernels' A gather)
                const unsigned row = (h >> 8) & row_mask;
                a_n[q] = *reinterpret_cast<const float4 *>(table + (size_t)row * 96 + 16 * (q % 6) + 4 * g);
            } else if (PAT == 1) {  // 4 rows x 256 B per wave-instruction
                const unsigned row = (__shfl(h, g * 16) >> 8) & row_mask;
                a_n[q] = *reinterpret_cast<const float4 *>(table + (size_t)row * 96 + 4 * r16);
            } else {                // 1 KiB contiguous per wave-instruction (rows of 96 floats: spans 2.7 rows)
                unsigned row = (__shfl(h, 0) >> 8) & row_mask;
                row = row > 8 ? row - 8 : 0;   // the 1-KiB read spans 2.7 rows: stay inside the table
                a_n[q] = *reinterpret_cast<const float4 *>(table + (size_t)row * 96 + 4 * lane);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const float *wb = lds + (it & 1) * 3072;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float4 b4 = *reinterpret_cast<const float4 *>(wb + (4 * g + e) * 96 + 4 * r16);
            const float2 b2 = *reinterpret_cast<const float2 *>(wb + (4 * g + e) * 96 + 64 + 2 * r16);
            const float av = e == 0 ? a_c.x : e == 1 ? a_c.y : e == 2 ? a_c.z : a_c.w;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b4.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b4.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b4.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b4.w, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2.x, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2.y, acc[5], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int n = 0; n < 6; n++) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float *g_table, *g_w, *g_out;

template <int G, int D, int BAR, int PAT = 0>
void run(int wgs_per_cu, unsigned row_mask, const char *where) {
    const int grid = 256 * wgs_per_cu, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<G, D, BAR, PAT><<<grid, 256, 2 * 3072 * 4>>>(g_table, row_mask, g_w, g_out, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<G, D, BAR, PAT><<<grid, 256, 2 * 3072 * 4>>>(g_table, row_mask, g_w, g_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 24 * 2048.0;
    // SIMD-cycles per iteration at the nominal 2.4 GHz (the chip's real clock under load is lower): each SIMD runs
    // wgs_per_cu waves, so one wave-iteration gets (ms * 2.4e6 cycles) / (iters * wgs_per_cu)
    printf("G=%d D=%d BAR=%d PAT=%d %d WG/CU rows %-8s: %8.3f ms  %6.1f TFLOP/s  %7.0f nominal SIMD-cycles per wave-iteration (24 MFMA = 768)\n",
           G, D, BAR, PAT, wgs_per_cu, where, ms, flops / ms / 1e9, ms * 2.4e6 / iters / wgs_per_cu);
}

int main() {
    const size_t rows = 1u << 22;  // 4 M rows x 384 B = 1.6 GB
    hipMalloc(&g_table, rows * 96 * 4);
    hipMemset(g_table, 0, rows * 96 * 4);
    hipMalloc(&g_w, 1024 * 256 * 4);
    hipMemset(g_w, 0, 1024 * 256 * 4);
    hipMalloc(&g_out, 256 * 8 * 256 * 4);
    const unsigned l2 = (1u << 12) - 1, big = (1u << 22) - 1;  // 1.5 MB window | the whole table
    for (int w = 1; w <= 3; w++) {
        run<0, 0, 0>(w, l2, "-");
        run<0, 0, 1>(w, l2, "-");
        run<1, 0, 0>(w, l2, "L2"); run<2, 0, 0>(w, l2, "L2"); run<4, 0, 0>(w, l2, "L2"); run<6, 0, 0>(w, l2, "L2"); run<8, 0, 0>(w, l2, "L2");
        run<2, 0, 0>(w, big, "HBM"); run<4, 0, 0>(w, big, "HBM");
        run<0, 1, 1>(w, l2, "-"); run<0, 3, 1>(w, l2, "-");
        run<2, 1, 1>(w, l2, "L2"); run<2, 3, 1>(w, l2, "L2"); run<2, 3, 1>(w, big, "HBM");
        run<4, 3, 1>(w, big, "HBM");
    }
    // gather shape: 16 x 64 B (PAT 0) vs 4 x 256 B (PAT 1) vs 1 KiB contiguous (PAT 2) per wave-instruction
    run<6, 0, 0, 1>(3, l2, "L2"); run<8, 0, 0, 1>(3, l2, "L2"); run<6, 0, 0, 2>(3, l2, "L2"); run<8, 0, 0, 2>(3, l2, "L2");
    run<4, 0, 0, 1>(3, big, "HBM"); run<4, 0, 0, 2>(3, big, "HBM");
    return 0;
}
