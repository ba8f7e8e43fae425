// Micro-benchmark (gfx950): how fast ONE wave issues vector-ALU work on its SIMD, and what a second wave on the same SIMD adds - the numbers
// a latency-chain kernel (BSRNN's band recurrence: one stream per CU, no MFMA) is built on.
//   kinds: v_fma_f32 / v_pk_fma_f32 with NCH independent accumulator chains (NCH = 1: a dependent chain), v_exp_f32, v_rcp_f32,
//          v_mov_b32 DPP (quad_perm), ds_read_b128 (broadcast address) + s_waitcnt, v_readlane_b32
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue_rate.hip -o ab/valu_issue_rate && ab/valu_issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NCH>
__global__ void __launch_bounds__(512) kern(float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    lds[threadIdx.x] = 1.0f + threadIdx.x * 1e-3f;
    __syncthreads();
    float a[8], w = 1.0001f + threadIdx.x * 1e-6f;
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0.5f + i + threadIdx.x * 1e-3f; p[i] = f32x2{a[i], a[i] + 1.0f}; }
    const f32x2 w2 = {w, w};
    float4 q[4];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const int c = m % NCH;
            if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[c]) : "v"(w));
            else if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[c]) : "v"(w2));
            else if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[c]));
            else if constexpr (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[c]));
            else if constexpr (KIND == 4) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[c]));
            else if constexpr (KIND == 5) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(q[m & 3]) : "v"((m & 7) * 16));
                if ((m & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)");
            }
            else if constexpr (KIND == 6) { int s; asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s) : "v"(a[c]), "n"(m)); asm volatile("" :: "s"(s)); }
            else if constexpr (KIND == 7) {      // dependent: ds_write -> ds_read (same wave, in order) -> use
                asm volatile("ds_write_b32 %1, %0\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[0]) : "v"((int)(threadIdx.x * 4)));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i) s += q[i].x + q[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int KIND, int NCH>
void run(const char* name, int threads) {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&clk, 16 * 8);
    const int iters = 64;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((kern<KIND, NCH>), dim3(1), dim3(threads), 0, 0, out, clk, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[16]; (void)hipMemcpy(h, clk, 16 * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    unsigned long long lo = h[0], hi = h[1];
    for (int w = 0; w < nw; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
    const double per_wave = (double)(hi - lo) / (iters * 32);
    printf("%-28s chains=%d waves/SIMD=%d : %6.2f cycles per instruction per wave, %6.2f per instruction per SIMD\n", name, NCH, threads / 256, per_wave,
           per_wave / (threads / 256));
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    for (int threads : {64, 256, 512, 1024}) {
        if (threads == 1024) continue;
        run<0, 8>("v_fma_f32", threads); run<0, 2>("v_fma_f32", threads); run<0, 1>("v_fma_f32 (dependent)", threads);
        run<1, 8>("v_pk_fma_f32", threads); run<1, 2>("v_pk_fma_f32", threads); run<1, 1>("v_pk_fma_f32 (dependent)", threads);
        run<2, 8>("v_exp_f32", threads); run<2, 1>("v_exp_f32 (dependent)", threads);
        run<3, 8>("v_rcp_f32", threads); run<3, 1>("v_rcp_f32 (dependent)", threads);
        run<4, 8>("v_mov_b32 dpp quad_perm", threads); run<4, 1>("v_mov_b32 dpp (dependent)", threads);
        run<5, 1>("ds_read_b128 x8 + wait", threads);
        run<6, 8>("v_readlane_b32", threads);
        run<7, 1>("ds_write->ds_read->wait", threads);
    }
    return 0;
}
