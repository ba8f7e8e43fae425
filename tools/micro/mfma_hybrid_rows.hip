// Micro-benchmark (gfx950): would the second row tile of a 24-row token GEMM (rows 16-23: 8 of 16 rows are padding) pay on
// v_mfma_f32_4x4x1_16B_f32 next to a 16x16x4 wave on the same SIMD?  One workgroup of eight waves, two per SIMD.  A "job" is one
// [rows x 36] . [36 x 48] product (K = 36, three column tiles), operands from LDS as in fe_frame8_kernel:
//   role A (16x16x4, 16 rows): 27 MFMAs, per k-step one A read + three B reads = 36 ds_read_b32
//   role B (4x4x1, 8 rows = 2 row groups x 8 column groups per instruction): 36 + 18 = 54 MFMAs (third tile K-split over the
//           two lane halves), one A read per k + one B read per MFMA = 90 ds_read_b32   (RB4 = 0: operands from registers)
// baseline: both waves of a SIMD run role A (what the kernel does today, 54 MFMAs of 16x16x4 per SIMD and job pair)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_hybrid_rows.hip -o ab/mfma_hybrid_rows && ab/mfma_hybrid_rows
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DSR(dst, addr) asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr))

template <int HYBRID, int READS, int BCAST>
__global__ void __launch_bounds__(512) kern(float* out, unsigned long long* clk, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int addr = (BCAST ? (lane & 31) : lane) * 4 + wave * 256;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b[3] = {1.0f, 2.0f, 3.0f};
    unsigned long long t0 = __builtin_readcyclecounter();
    if (!HYBRID || wave < 4) {
        float ra[2] = {a, a}, rb[2][3] = {{b[0], b[1], b[2]}, {b[0], b[1], b[2]}};
        if (READS) { DSR(ra[0], addr); DSR(rb[0][0], addr + 1024); DSR(rb[0][1], addr + 2048); DSR(rb[0][2], addr + 3072); }
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) {      // (ten k-steps of one read set each: an even count keeps the double buffer's parity static)
                const int c = ks & 1, n = c ^ 1;
                if (READS) { DSR(ra[n], addr); DSR(rb[n][0], addr + 1024); DSR(rb[n][1], addr + 2048); DSR(rb[n][2], addr + 3072); asm volatile("s_waitcnt lgkmcnt(4)"); }
                if (ks < 9) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[c], rb[c][j], acc[j], 0, 0, 0);
                }
            }
        }
        a = ra[0] + ra[1]; b[0] = rb[0][0] + rb[1][0]; b[1] = rb[0][1] + rb[1][1]; b[2] = rb[0][2] + rb[1][2];
    } else {
        float ra[2][2] = {{a, a}, {a, a}}, rb[2][3] = {{b[0], b[1], b[2]}, {b[0], b[1], b[2]}};
        if (READS) { DSR(ra[0][0], addr); DSR(ra[0][1], addr + 512); DSR(rb[0][0], addr + 1024); DSR(rb[0][1], addr + 2048); DSR(rb[0][2], addr + 3072); }
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 18; ++kk) {      // two k per pass: 2 x (tiles 0-1) + 1 x (tile 2, K-split)
                const int c = kk & 1, n = c ^ 1;
                if (READS) { DSR(ra[n][0], addr); DSR(ra[n][1], addr + 512); DSR(rb[n][0], addr + 1024); DSR(rb[n][1], addr + 2048); DSR(rb[n][2], addr + 3072); asm volatile("s_waitcnt lgkmcnt(5)"); }
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ra[c][0], rb[c][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ra[c][1], rb[c][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ra[c][0], rb[c][2], acc[2], 0, 0, 0);
            }
        }
        a = ra[0][0] + ra[1][1]; b[0] = rb[0][0] + rb[1][0]; b[1] = rb[0][1] + rb[1][1]; b[2] = rb[0][2] + rb[1][2];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = a + b[0] + b[1] + b[2];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x] = s;
    if (lane == 0) { clk[2 * wave] = t0; clk[2 * wave + 1] = t1; }
}

template <int HYBRID, int READS, int BCAST>
void run(const char* what) {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&clk, 16 * 8);
    const int iters = 64;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((kern<HYBRID, READS, BCAST>), dim3(1), dim3(512), 0, 0, out, clk, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[16]; (void)hipMemcpy(h, clk, 16 * 8, hipMemcpyDeviceToHost);
    unsigned long long lo = h[0], hi = h[1], hiA = 0, hiB = 0;
    for (int w = 0; w < 8; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi;
        if (w < 4) hiA = h[2 * w + 1] > hiA ? h[2 * w + 1] : hiA; else hiB = h[2 * w + 1] > hiB ? h[2 * w + 1] : hiB; }
    printf("%-58s : %7.1f cycles per job pair (waves 0-3 done at %7.1f, waves 4-7 at %7.1f)\n", what, (double)(hi - lo) / iters, (double)(hiA - lo) / iters, (double)(hiB - lo) / iters);
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    run<0, 0, 0>("baseline 16x16x4 + 16x16x4, register operands");
    run<0, 1, 0>("baseline 16x16x4 + 16x16x4, LDS operands");
    run<1, 0, 0>("hybrid   16x16x4 + 4x4x1 (rows 16-23), register operands");
    run<1, 1, 0>("hybrid   16x16x4 + 4x4x1 (rows 16-23), LDS operands");
    run<1, 1, 1>("hybrid   16x16x4 + 4x4x1, LDS operands, 32 distinct addrs");
    return 0;
}
