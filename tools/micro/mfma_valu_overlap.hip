// Micro-benchmark (gfx950): does a wave's own VALU / transcendental work execute in the shadow of its in-flight MFMAs?
// One workgroup of 4 waves (one per SIMD) or 8 waves (two per SIMD); each wave runs 256 x { v_mfma_f32_16x16x4_f32 ; K x VALU }.
// Prints shader cycles per MFMA (per SIMD) for K filler instructions of each kind after every MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o gpurun_out/mfma_valu && gpurun_out/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int K, int KIND, bool BF16 = false>   // KIND 0: v_fma_f32 (independent chains), 1: v_exp_f32, 2: ds_read_b32
__global__ void __launch_bounds__(512) kern(float* out, unsigned long long* clk, int iters) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if constexpr (BF16) {
                bf16x8 ha, hb;
                for (int q = 0; q < 8; ++q) { ha[q] = (__bf16)a; hb[q] = (__bf16)b; }
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[m & 3], 0, 0, 0);
            } else {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float& x = v[(m * K + k) & 7];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"((int)(threadIdx.x * 4)));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int K, int KIND, bool BF16 = false>
void run(const char* name, int threads) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 512 * 4); hipMalloc(&clk, 16 * 8);
    const int iters = 64;
    hipLaunchKernelGGL((kern<K, KIND, BF16>), dim3(1), dim3(threads), 0, 0, out, clk, iters);
    hipLaunchKernelGGL((kern<K, KIND, BF16>), dim3(1), dim3(threads), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    unsigned long long h[16]; hipMemcpy(h, clk, 16 * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    unsigned long long lo = h[0], hi = h[1];
    for (int w = 0; w < nw; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
    // all waves, first start to last end, per MFMA of ONE SIMD's waves (2 waves per SIMD issue 2 x the MFMAs)
    printf("%-10s waves/SIMD=%d  K=%d : %6.1f cycles per MFMA per SIMD  (wave 0 alone: %6.1f)\n", name, threads / 256, K,
           (double)(hi - lo) / (iters * 8 * (threads / 256)), (double)(h[1] - h[0]) / (iters * 8));
    hipFree(out); hipFree(clk);
}

int main() {
    for (int threads : {256, 512}) {
        run<0, 0>("none", threads);
        run<1, 0>("v_fma", threads); run<2, 0>("v_fma", threads); run<4, 0>("v_fma", threads); run<6, 0>("v_fma", threads);
        run<1, 1>("v_exp", threads); run<2, 1>("v_exp", threads); run<3, 1>("v_exp", threads);
        run<1, 2>("ds_read", threads); run<2, 2>("ds_read", threads); run<4, 2>("ds_read", threads);
    }
    printf("-- the same with v_mfma_f32_16x16x32_bf16 (16x the flops per instruction) --\n");
    for (int threads : {256, 512}) {
        run<0, 0, true>("none", threads);
        run<1, 0, true>("v_fma", threads); run<2, 0, true>("v_fma", threads); run<4, 0, true>("v_fma", threads);
        run<1, 1, true>("v_exp", threads); run<2, 1, true>("v_exp", threads);
    }
    return 0;
}
