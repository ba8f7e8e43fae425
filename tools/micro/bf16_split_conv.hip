// Micro-benchmark (gfx950): one FastEnhancer_B-sized conv phase (48 -> 48 channels, 3 taps over 64 frequency rows: a
// [64 x 144] . [144 x 48] GEMM + bias + SiLU, output = the next layer's input in LDS, one workgroup of 4 waves per CU) as
//   mode 0: fp32 on v_mfma_f32_16x16x4_f32                                 (108 MFMAs per wave and layer; what the library runs)
//   mode 2: operands split into 2 bf16 terms, 3 products  hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16   (45 per wave)
//   mode 3: operands split into 3 bf16 terms, 6 products  (all pairs of order <= 2)                            (90 per wave)
// fp32 accumulation in all modes; the splitting of the activations is part of the epilogue (and of the measured time), the
// weights are split once on the host.  Prints shader cycles per layer and the error of an 8-layer chain against a float64
// CPU evaluation of the same chain.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bf16_split_conv.hip -o gpurun_out/bf16_split_conv && gpurun_out/bf16_split_conv
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int C = 48, F = 64, TAPS = 3, K = C * TAPS;     // 144
constexpr int LAYERS = 8;
// fp32 layout: act[c][LDF], frequency f at column f + 1 (zero halo columns 0 and F + 1)
constexpr int LDF = 69;
// bf16 layout: one plane per term, plane[f + 1][LDR] (zero halo rows 0 and F + 1), channels contiguous
constexpr int LDR = 56;
constexpr int KS32 = 5;                                   // k-steps of 32 (18 chunks of 8 channels, padded to 20)

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }

// ------------------------------------------------------------------------------------------------ fp32
// wave w owns frequency rows 16 w .. 16 w + 15 (A operand = activations), the 3 column tiles are the 48 output channels
// (B operand = weights, packed [layer][ks][nt][lane] and staged in LDS)
__global__ void __launch_bounds__(256) conv_f32(const float* __restrict__ x0, const float* __restrict__ wpk, const float* __restrict__ bias,
                                                float* __restrict__ out, unsigned long long* clk, int reps) {
    __shared__ float act[2][C * LDF];
    __shared__ float wl[36 * 3 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * C * LDF; i += 256) (&act[0][0])[i] = 0.0f;
    for (int i = tid; i < 36 * 3 * 64; i += 256) wl[i] = wpk[i];
    __syncthreads();
    for (int i = tid; i < C * F; i += 256) act[0][(i / F) * LDF + (i % F) + 1] = x0[i];
    __syncthreads();
    float b[3];
    for (int nt = 0; nt < 3; ++nt) b[nt] = bias[nt * 16 + (lane & 15)];
    unsigned long long tsum = 0;
    for (int rep = 0; rep < reps + 1; ++rep) {
        if (rep >= 1) {           // restart the chain from the input (values stay O(1))
            for (int i = tid; i < C * F; i += 256) act[0][(i / F) * LDF + (i % F) + 1] = x0[i];
            __syncthreads();
        }
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int l = 0; l < LAYERS; ++l) {
            const float* src = act[l & 1];
            float* dst = act[(l & 1) ^ 1];
            f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const int f = wave * 16 + (lane & 15);
            float av[3], bv[3][3];
            auto load32 = [&](int ks, float& a, float* b3) {
                const int kk = ks * 4;                            // + lane / 16: the tap kk / 48 is wave-uniform
                a = src[(kk % C + (lane >> 4)) * LDF + f + kk / C];
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) b3[nt] = wl[(ks * 3 + nt) * 64 + lane];
            };
            load32(0, av[0], bv[0]); load32(1, av[1], bv[1]);
#pragma unroll
            for (int ks = 0; ks < 36; ++ks) {
                if (ks + 2 < 36) load32(ks + 2, av[(ks + 2) % 3], bv[(ks + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks % 3], bv[ks % 3][nt], acc[nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[(nt * 16 + (lane & 15)) * LDF + wave * 16 + 4 * (lane >> 4) + i + 1] = silu(acc[nt][i] + b[nt]);
            __syncthreads();
        }
        if (rep >= 1) tsum += __builtin_readcyclecounter() - t0;
    }
    if (tid == 0) clk[blockIdx.x] = tsum;
    if (blockIdx.x == 0)
        for (int i = tid; i < C * F; i += 256) out[i] = act[LAYERS & 1][(i / F) * LDF + (i % F) + 1];
}

// ------------------------------------------------------------------------------------------------ split bf16
// wave w owns frequency columns 16 w .. 16 w + 15 (B operand = activations, 8 consecutive channels of one tap per lane), the
// 3 row tiles are the 48 output channels (A operand = weights, packed [mt][ks][term][lane][8], staged in LDS); the
// accumulator lane holds 4 consecutive output channels of one frequency: one 8-byte store per term and row tile
// VAR (timing breakdown only, results wrong): 1 = the weights stay in registers over the layers, 2 = no SiLU / split in the epilogue,
// 3 = no MFMAs
template <int NT, int VAR = 0>   // NT = 2: hi, lo (3 products); NT = 3: hi, mid, lo (6 products)
__global__ void __launch_bounds__(256) conv_bf16(const float* __restrict__ x0, const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                 float* __restrict__ out, unsigned long long* clk, int reps) {
    constexpr int PL = (F + 2) * LDR;                     // one plane, in bf16 elements
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* wl = reinterpret_cast<uint4*>(smem);
    __bf16 (*act)[NT][PL] = reinterpret_cast<__bf16 (*)[NT][PL]>(smem + (size_t)3 * KS32 * NT * 64 * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * NT * PL; i += 256) (&act[0][0][0])[i] = (__bf16)0.0f;
    for (int i = tid; i < 3 * KS32 * NT * 64; i += 256) wl[i] = wpk[i];
    __syncthreads();
    auto put = [&](__bf16 (*p)[PL], int f, int c, float v) {
        float r = v;
#pragma unroll
        for (int t = 0; t < NT; ++t) { __bf16 h = (__bf16)r; p[t][(f + 1) * LDR + c] = h; r -= (float)h; }
    };
    for (int i = tid; i < C * F; i += 256) put(act[0], i % F, i / F, x0[i]);
    __syncthreads();
    float b[3][4];
    for (int mt = 0; mt < 3; ++mt)
        for (int i = 0; i < 4; ++i) b[mt][i] = bias[mt * 16 + 4 * (lane >> 4) + i];
    bf16x8 wreg[VAR == 1 ? 3 * KS32 : 1][NT];
    if constexpr (VAR == 1)
        for (int g = 0; g < 3 * KS32; ++g)
            for (int t = 0; t < NT; ++t) {
                const uint4 u = wl[(((g % 3) * KS32 + g / 3) * NT + t) * 64 + lane];
                __builtin_memcpy(&wreg[g][t], &u, 16);
            }
    unsigned long long tsum = 0;
    for (int rep = 0; rep < reps + 1; ++rep) {
        if (rep >= 1) {
            for (int i = tid; i < C * F; i += 256) put(act[0], i % F, i / F, x0[i]);
            __syncthreads();
        }
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int l = 0; l < LAYERS; ++l) {
            const __bf16 (*src)[PL] = act[l & 1];
            __bf16 (*dst)[PL] = act[(l & 1) ^ 1];
            f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const int f = wave * 16 + (lane & 15);
            // software pipeline over the 15 (k-step, row tile) groups: the operands of group g + 2 are in flight while group g
            // multiplies; the products of a group alternate between two accumulators (no back-to-back dependent MFMAs)
            bf16x8 wb[3][NT], xb[2][NT];
            f32x4 acc2[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            auto load_w = [&](int g, bf16x8* w) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const uint4 u = wl[(((g % 3) * KS32 + g / 3) * NT + t) * 64 + lane];
                    __builtin_memcpy(&w[t], &u, 16);
                }
            };
            auto load_x = [&](int ks, bf16x8* x) {
                int q = ks * 4 + (lane >> 4);                     // chunk of 8 channels: tap q / 6, channels 8 (q % 6) ..
                q = q < 18 ? q : 17;                              // padded chunks: their weights are zero, any finite operand does
                const int tap = q / 6, c0 = (q % 6) * 8;
#pragma unroll
                for (int t = 0; t < NT; ++t) x[t] = *reinterpret_cast<const bf16x8*>(&src[t][(f + tap) * LDR + c0]);
            };
            load_x(0, xb[0]);
            if constexpr (VAR != 1) { load_w(0, wb[0]); load_w(1, wb[1]); }
#pragma unroll
            for (int g = 0; g < 3 * KS32; ++g) {
                if (g + 2 < 3 * KS32) {
                    if ((g + 2) % 3 == 0) load_x((g + 2) / 3, xb[((g + 2) / 3) & 1]);
                    if constexpr (VAR != 1) load_w(g + 2, wb[(g + 2) % 3]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int mt = g % 3;
                int n = 0;
                // smallest terms first
#pragma unroll
                for (int o = 2 * (NT - 1); o >= 0; --o)
#pragma unroll
                    for (int ta = 0; ta < NT; ++ta) {
                        const int tb = o - ta;
                        if (tb < 0 || tb >= NT || o > NT - 1) continue;     // products of order <= NT - 1
                        const bf16x8 w = VAR == 1 ? wreg[g][ta] : wb[g % 3][ta];
                        f32x4& a = (n++ & 1) ? acc2[mt] : acc[mt];
                        if constexpr (VAR == 3) { a[0] += (float)w[0] * (float)xb[(g / 3) & 1][tb][0]; continue; }
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xb[(g / 3) & 1][tb], a, 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) acc[mt] += acc2[mt];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                float r[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = VAR == 2 ? acc[mt][i] : silu(acc[mt][i] + b[mt][i]);
#pragma unroll
                for (int t = 0; t < (VAR == 2 ? 1 : NT); ++t) {
                    bf16x4 h;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { h[i] = (__bf16)r[i]; r[i] -= (float)h[i]; }
                    *reinterpret_cast<bf16x4*>(&dst[t][(f + 1) * LDR + mt * 16 + 4 * (lane >> 4)]) = h;
                }
            }
            __syncthreads();
        }
        if (rep >= 1) tsum += __builtin_readcyclecounter() - t0;
    }
    if (tid == 0) clk[blockIdx.x] = tsum;
    if (blockIdx.x == 0)
        for (int i = tid; i < C * F; i += 256) {
            float s = 0;
            for (int t = NT - 1; t >= 0; --t) s += (float)act[LAYERS & 1][t][((i % F) + 1) * LDR + i / F];
            out[i] = s;
        }
}

// ------------------------------------------------------------------------------------------------ host
static uint16_t bf16_rne(float v) {
    uint32_t u; std::memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float v; std::memcpy(&v, &u, 4); return v; }

int main() {
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    std::vector<float> W((size_t)C * K), bias(C), x0((size_t)C * F);      // W[m][kk], kk = tap * 48 + c
    for (auto& v : W) v = nd(rng) * 1.6f / std::sqrt((float)K);
    for (auto& v : bias) v = nd(rng) * 0.1f;
    for (auto& v : x0) v = nd(rng);
    // float64 chain
    std::vector<double> cur(x0.begin(), x0.end()), nxt((size_t)C * F);
    for (int l = 0; l < LAYERS; ++l) {
        for (int m = 0; m < C; ++m)
            for (int f = 0; f < F; ++f) {
                double s = bias[m];
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int ff = f + tap - 1;
                    if (ff < 0 || ff >= F) continue;
                    for (int c = 0; c < C; ++c) s += (double)W[(size_t)m * K + tap * C + c] * cur[(size_t)c * F + ff];
                }
                nxt[(size_t)m * F + f] = s / (1.0 + std::exp(-s));
            }
        cur.swap(nxt);
    }
    double ref_rms = 0;
    for (double v : cur) ref_rms += v * v;
    ref_rms = std::sqrt(ref_rms / cur.size());

    // fp32 pack: [ks][nt][lane] = W[n = nt*16 + lane%16][kk = 4 ks + lane/16]
    std::vector<float> wf(36 * 3 * 64);
    for (int ks = 0; ks < 36; ++ks)
        for (int nt = 0; nt < 3; ++nt)
            for (int l = 0; l < 64; ++l) wf[(ks * 3 + nt) * 64 + l] = W[(size_t)(nt * 16 + l % 16) * K + ks * 4 + l / 16];
    auto pack_bf16 = [&](int NT) {
        std::vector<uint16_t> p((size_t)3 * KS32 * NT * 64 * 8, 0);
        for (int mt = 0; mt < 3; ++mt)
            for (int ks = 0; ks < KS32; ++ks)
                for (int l = 0; l < 64; ++l) {
                    const int q = ks * 4 + l / 16;
                    for (int j = 0; j < 8; ++j) {
                        float r = q < 18 ? W[(size_t)(mt * 16 + l % 16) * K + q * 8 + j] : 0.0f;
                        for (int t = 0; t < NT; ++t) {
                            const uint16_t h = bf16_rne(r);
                            p[((((size_t)mt * KS32 + ks) * NT + t) * 64 + l) * 8 + j] = h;
                            r -= bf16_to_f(h);
                        }
                    }
                }
        return p;
    };
    float *dx, *dwf, *db, *dout; unsigned long long* dclk; void *dw2, *dw3;
    const int NWG = 256, REPS = 50;
    hipMalloc(&dx, x0.size() * 4); hipMalloc(&dwf, wf.size() * 4); hipMalloc(&db, C * 4); hipMalloc(&dout, C * F * 4); hipMalloc(&dclk, NWG * 8);
    auto w2 = pack_bf16(2), w3 = pack_bf16(3);
    hipMalloc(&dw2, w2.size() * 2); hipMalloc(&dw3, w3.size() * 2);
    hipMemcpy(dx, x0.data(), x0.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dwf, wf.data(), wf.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bias.data(), C * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw2, w2.data(), w2.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw3, w3.data(), w3.size() * 2, hipMemcpyHostToDevice);
    auto report = [&](const char* name, int mfma) {
        hipDeviceSynchronize();
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return; }
        std::vector<unsigned long long> clk(NWG); std::vector<float> o((size_t)C * F);
        hipMemcpy(clk.data(), dclk, NWG * 8, hipMemcpyDeviceToHost); hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        std::sort(clk.begin(), clk.end());
        double err2 = 0, emax = 0;
        for (size_t i = 0; i < o.size(); ++i) { const double d = o[i] - cur[i]; err2 += d * d; emax = std::max(emax, std::fabs(d)); }
        const double per_layer = (double)clk[NWG / 2] / (REPS * LAYERS);
        printf("%-34s %7.0f cycles / layer (median of %d workgroups; %d MFMAs per wave: %.1f cycles each)   rms err / rms %.3g   max abs err %.3g\n",
               name, per_layer, NWG, mfma, per_layer / mfma, std::sqrt(err2 / o.size()) / ref_rms, emax);
    };
    auto lds_bytes = [](int NT) { return (size_t)3 * KS32 * NT * 64 * 16 + (size_t)2 * NT * (F + 2) * LDR * 2; };
    const size_t lds2 = lds_bytes(2), lds3 = lds_bytes(3);
    hipFuncSetAttribute((const void*)conv_bf16<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipFuncSetAttribute((const void*)conv_bf16<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
    for (int it = 0; it < 2; ++it) {      // first pass warms the clocks
        hipLaunchKernelGGL(conv_f32, dim3(NWG), dim3(256), 0, 0, dx, dwf, db, dout, dclk, REPS);
        if (it) report("fp32 16x16x4", 108); else hipDeviceSynchronize();
        hipLaunchKernelGGL(conv_bf16<2>, dim3(NWG), dim3(256), lds2, 0, dx, (const uint4*)dw2, db, dout, dclk, REPS);
        if (it) report("bf16 x 2 terms, 3 products", 45); else hipDeviceSynchronize();
        hipLaunchKernelGGL(conv_bf16<3>, dim3(NWG), dim3(256), lds3, 0, dx, (const uint4*)dw3, db, dout, dclk, REPS);
        if (it) report("bf16 x 3 terms, 6 products", 90); else hipDeviceSynchronize();
        if (it) {
            hipFuncSetAttribute((const void*)conv_bf16<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            hipFuncSetAttribute((const void*)conv_bf16<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            hipFuncSetAttribute((const void*)conv_bf16<3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            hipLaunchKernelGGL((conv_bf16<3, 1>), dim3(NWG), dim3(256), lds3, 0, dx, (const uint4*)dw3, db, dout, dclk, REPS);
            report("  [timing only] weights in VGPRs", 90);
            hipLaunchKernelGGL((conv_bf16<3, 2>), dim3(NWG), dim3(256), lds3, 0, dx, (const uint4*)dw3, db, dout, dclk, REPS);
            report("  [timing only] bare epilogue", 90);
            hipLaunchKernelGGL((conv_bf16<3, 3>), dim3(NWG), dim3(256), lds3, 0, dx, (const uint4*)dw3, db, dout, dclk, REPS);
            report("  [timing only] no MFMAs", 90);
        }
    }
    printf("reference rms %.4f (float64 chain of %d layers)\n", ref_rms, LAYERS);
    return 0;
}
