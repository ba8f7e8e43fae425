// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access patterns the product kernels use
// (/opt/skills/guides/MI355X_MICROARCH.md "HBM": "calibrate on a known byte count in your own access pattern").
//   rd_dword     4 B per lane, coalesced (global_load_dword): state rows / fragment loads as dwords
//   rd_x4        16 B per lane, coalesced (global_load_dwordx4): state rows as whole lines, weight rows
//   rd_buf_x4    16 B per lane through a buffer resource (buffer_load_dwordx4): the k4-regrouped weight streams, LDS staging pieces
//   rd_buf_dword 4 B per lane through a buffer resource (buffer_load_dword): MFMA fragment loads
//   wr_dword     4 B per lane, coalesced stores
//   wr_x4        16 B per lane, coalesced stores (state rows written at the end of the frame)
//   wr_piece64   64-byte pieces (16 lanes x 4 B) of rows 144 B apart: accumulator-layout state stores (partial lines)
// Each kernel touches every byte of its range exactly once.  Two sizes: 1 GiB (past the 256 MiB Infinity Cache: every byte comes from /
// goes to HBM) and 8 MiB launched repeatedly (the product's regime: a launch's working set sits in the Infinity Cache / L2 from the
// previous launch).  Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE  (separate passes); tools/hbm_calib.sh
// turns the CSVs into bytes-per-counter-unit factors.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_counter_calib.hip -o ab/hbm_counter_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) rd_dword(const float* __restrict__ p, size_t n, float* sink) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) rd_x4(const f32x4* __restrict__ p, size_t n4, float* sink) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}
// buffer loads: the range is walked in 1 GiB-or-less windows of one resource (32-bit offsets)
__global__ void __launch_bounds__(256) rd_buf_x4(const float* p, size_t n4, float* sink) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(n4 * 16 > 0x7fffffffu ? 0x7fffffff : n4 * 16), 0x00020000);
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        acc += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i * 16), 0, 0));
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}
__global__ void __launch_bounds__(256) rd_buf_dword(const float* p, size_t n, float* sink) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(n * 4 > 0x7fffffffu ? 0x7fffffff : n * 4), 0x00020000);
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(i * 4), 0, 0));
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) wr_dword(float* __restrict__ p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ void __launch_bounds__(256) wr_x4(f32x4* __restrict__ p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = f32x4{v, v, v, v};
}
// rows of 36 floats (144 B); a wave's store instruction writes columns [16 c, 16 c + 16) x 4 consecutive rows... as the accumulator
// layout does: lane (li, lg) -> row 4 q + lg, column 16 c + li (c = 0, 1; the last 4 columns by a third, masked, store).  Every float of
// every row is written once: bytes = rows * 144.
__global__ void __launch_bounds__(256) wr_piece64(float* __restrict__ p, size_t rows, float v) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256) >> 6;
    for (size_t q = wave; q * 4 < rows; q += nw) {
        const size_t row = 4 * q + lg;
        if (row < rows) {
            p[row * 36 + li] = v;
            p[row * 36 + 16 + li] = v;
            if (li < 4) p[row * 36 + 32 + li] = v;
        }
    }
}

int main(int argc, char** argv) {
    const size_t big = (size_t)1 << 30, small = (size_t)8 << 20;
    float *buf, *sink;
    CK(hipMalloc(&buf, big));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, big));
    const int grid = 256 * 8;
    for (int pass = 0; pass < 2; ++pass) {
        const size_t bytes = pass == 0 ? big : small;
        const int reps = pass == 0 ? 3 : 20;
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(rd_dword, dim3(grid), dim3(256), 0, 0, buf, bytes / 4, sink);
            hipLaunchKernelGGL(rd_x4, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const f32x4*>(buf), bytes / 16, sink);
            hipLaunchKernelGGL(rd_buf_x4, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, sink);
            hipLaunchKernelGGL(rd_buf_dword, dim3(grid), dim3(256), 0, 0, buf, bytes / 4, sink);
            hipLaunchKernelGGL(wr_dword, dim3(grid), dim3(256), 0, 0, buf, bytes / 4, 1.0f);
            hipLaunchKernelGGL(wr_x4, dim3(grid), dim3(256), 0, 0, reinterpret_cast<f32x4*>(buf), bytes / 16, 2.0f);
            hipLaunchKernelGGL(wr_piece64, dim3(grid), dim3(256), 0, 0, buf, bytes / 144, 3.0f);
            CK(hipDeviceSynchronize());
        }
    }
    printf("bytes per launch: pass 0 (first 3 launches of each kernel) %zu, pass 1 (next 20) %zu; wr_piece64 writes rows * 144 = %zu / %zu\n", big, small,
           (big / 144) * 144, (small / 144) * 144);
    return 0;
}
