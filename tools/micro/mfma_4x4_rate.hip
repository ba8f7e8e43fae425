// Micro-benchmark (gfx950): issue rate of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products, 256 MACs) against
// v_mfma_f32_16x16x4_f32 (1024 MACs), with NACC independent accumulators round-robin, one or two waves per SIMD, and with
// K ds_read_b32 per MFMA (the 4x4x1 form needs 4x the operand registers per MAC).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_4x4_rate.hip -o ab/mfma_4x4_rate && ab/mfma_4x4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FORM, int NACC, int K>   // FORM 0: 16x16x4, 1: 4x4x1
__global__ void __launch_bounds__(512) kern(float* out, unsigned long long* clk, int iters) {
    __shared__ float lds[2048];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if constexpr (FORM == 0) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % NACC], 0, 0, 0);
            else acc[m % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m % NACC], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("ds_read_b32 %0, %1" : "=v"(v[(m * K + k) & 7]) : "v"((int)(threadIdx.x * 4)));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int FORM, int NACC, int K>
void run(int threads) {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&clk, 16 * 8);
    const int iters = 64;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((kern<FORM, NACC, K>), dim3(1), dim3(threads), 0, 0, out, clk, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[16]; (void)hipMemcpy(h, clk, 16 * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    unsigned long long lo = h[0], hi = h[1];
    for (int w = 0; w < nw; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
    const double per = (double)(hi - lo) / (iters * 16 * (threads / 256));
    printf("%-8s acc=%d ds_read/mfma=%d waves/SIMD=%d : %6.1f cycles per MFMA per SIMD = %5.1f MACs/cycle/SIMD\n", FORM ? "4x4x1" : "16x16x4", NACC, K,
           threads / 256, per, (FORM ? 256.0 : 1024.0) / per);
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    for (int threads : {256, 512}) {
        run<0, 4, 0>(threads); run<0, 1, 0>(threads);
        run<1, 1, 0>(threads); run<1, 2, 0>(threads); run<1, 4, 0>(threads); run<1, 8, 0>(threads);
        run<1, 4, 1>(threads); run<1, 4, 2>(threads); run<0, 4, 2>(threads);
    }
    return 0;
}
