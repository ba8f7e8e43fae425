// Micro-benchmark (gfx950): cost of L2-resident buffer loads per CU, by width.  One workgroup of 4 or 8 waves; each wave issues
// NL loads (all lanes, coalesced 64 x W bytes, distinct addresses per load, the whole set L2-resident after a warm-up pass)
// and waits for them; reports cycles per wave-load as seen by the CU (all waves, first start to last end).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/vmem_issue_rate.hip -o ab/vmem_issue_rate && ab/vmem_issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int W>   // dwords per lane: 1, 2 or 4
__global__ void __launch_bounds__(512) kern(const float* src, float* out, unsigned long long* clk, int iters) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 24, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.0f;
    constexpr int NL = 32;
    unsigned long long t0 = 0, t1 = 0;
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            f32x4 v[NL];
            int itv = it & 3;                                          // (opaque: keeps the loads inside the loop) 4 x 256 KiB regions
            asm volatile("" : "+s"(itv));
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int soff = itv * (1 << 18) + ((wave * NL + i) * 64 * W) * 4;      // bytes; a wave's loads are consecutive W x 256-byte blocks
                if constexpr (W == 1) v[i] = f32x4{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, soff, 0)), 0, 0, 0};
                else if constexpr (W == 2) { auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8, soff, 0); v[i] = f32x4{__builtin_bit_cast(float, t[0]), __builtin_bit_cast(float, t[1]), 0, 0}; }
                else v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, soff, 0));
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) acc += v[i][0] + v[i][3];
        }
        t1 = __builtin_readcyclecounter();
    }
    out[threadIdx.x] = acc;
    if (lane == 0) { clk[2 * wave] = t0; clk[2 * wave + 1] = t1; }
}

template <int W>
void run(int threads, const float* src) {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&clk, 16 * 8);
    const int iters = 16;
    hipLaunchKernelGGL((kern<W>), dim3(1), dim3(threads), 0, 0, src, out, clk, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[16]; (void)hipMemcpy(h, clk, 16 * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    unsigned long long lo = h[0], hi = h[1];
    for (int w = 0; w < nw; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
    const double per = (double)(hi - lo) / (iters * 32 * nw);
    printf("buffer_load_dword%s waves=%d : %6.1f cycles per wave-load per CU = %5.1f bytes/cycle/CU\n", W == 1 ? "  " : (W == 2 ? "x2" : "x4"), nw, per, 256.0 * W / per);
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    float* src; (void)hipMalloc(&src, 1 << 24); (void)hipMemset(src, 0, 1 << 24);
    for (int threads : {256, 512}) { run<1>(threads, src); run<2>(threads, src); run<4>(threads, src); }
    return 0;
}
