// Which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13] on gfx9)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/wave_simd_map.hip -o gpurun_out/wave_simd_map && gpurun_out/wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(unsigned* out) {
    extern __shared__ float smem[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    smem[threadIdx.x] = 0;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 130 * 1024, 0, d);
    unsigned h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
    for (int w = 0; w < 8; ++w) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 3; ++b) { printf("wg %d:", b); for (int w = 0; w < 8; ++w) printf(" [simd %u wave_id %u cu %u]", (h[b*8+w] >> 4) & 3, h[b*8+w] & 15, (h[b*8+w] >> 8) & 15); printf("\n"); }
    }
    return 0;
}
