// Micro-benchmark (gfx950): one step of BSRNN-xt's band-LSTM recurrence (hidden 32, 128 gate rows, both directions, 31 steps, six layers) as a
// latency chain, in the forms considered for the per-stream frame kernel (csrc/bsrnn_kernels.hip.h):
//   V0   the r4 kernel's form: two waves per direction, one gate row per thread, h through LDS (8 x ds_read_b128 broadcasts), DPP quad
//        exchange of the activations, one workgroup barrier per step
//   S1   ONE wave per direction, two gate rows per lane (lanes 0-31: i, g of unit lane; lanes 32-63: f, o), W_hh rows in registers (64),
//        h through LDS without a barrier (the LDS operations of one wave execute in order), one v_permlane32_swap per step; packed FMAs
//   S1F  the same with plain v_fma_f32
//   S4   ONE wave per direction, h broadcast through SGPRs (32 x v_readlane_b32), no LDS in the chain
// Gate rows are pre-scaled (sigma(v) = rcp(1 + exp2(pre)), tanh(v) = 2 rcp(1 + exp2(pre)) - 1) like the packed weights of the product.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/micro/lstm_step.hip -o ab/lstm_step && ab/lstm_step
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HH = 32, G4 = 128, NB = 31, NLAY = 6;
constexpr int LDP = G4 + 2, LDY = 2 * HH + 2;
constexpr float K2 = -2.8853900817779268f;

__device__ __forceinline__ float sig2(float pre) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre)); }

// Wc[d][g][u][k] canonical; XPc[d][band][g][u]; out Y[layer-sum][band][d][u]
template <int VAR, int ABL = 0>
__global__ void __launch_bounds__(256) kern(const float* __restrict__ Wc, const float* __restrict__ XPc, float* __restrict__ Yout, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float XP[2 * 32 * LDP];
    __shared__ __attribute__((aligned(16))) float XP2[2 * 32 * 128];      // S1 / S4: [d][band][lane][2]
    __shared__ __attribute__((aligned(16))) float Yf[32 * LDY];
    __shared__ __attribute__((aligned(16))) float Hb[2 * 2 * HH + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * NB * G4; i += 256) {
        const int d = i / (NB * G4), r = i % (NB * G4), band = r / G4, gu = r % G4, g = gu / HH, u = gu % HH;
        XP[(d * 32 + band) * LDP + gu] = XPc[i];
        // S1 layout: lane l = (half, u): half 0 rows (i, g) = gates (0, 2), half 1 rows (f, o) = gates (1, 3)
        const int half = g & 1, slot = g >> 1;
        XP2[((d * 32 + band) * 64 + half * 32 + u) * 2 + slot] = XPc[i];
    }
    for (int i = tid; i < 32 * LDY; i += 256) Yf[i] = 0.0f;
    __syncthreads();
    float ysum = 0.0f;
    unsigned long long t0 = 0, t1 = 0;

    if constexpr (VAR == 0) {
        const int rd = wave >> 1, rq = tid & 127, rgate = rq & 3, u = rq >> 2;
        float Whh[HH];
#pragma unroll
        for (int k = 0; k < HH; ++k) Whh[k] = Wc[((rd * 4 + rgate) * HH + u) * HH + k];
        const float act_m = rgate == 2 ? 2.0f : 1.0f, act_a = rgate == 2 ? -1.0f : 0.0f;
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int l = 0; l < NLAY; ++l) {
            if (tid < 4 * HH) Hb[tid] = 0.0f;
            __syncthreads();
            float cstate = 0.0f;
            const int band0 = rd == 0 ? 0 : NB - 1;
            int xo = (rd * 32 + band0) * LDP + rgate * HH + u, yo = band0 * LDY + rd * HH + u;
            const int xd = rd == 0 ? LDP : -LDP, yd = rd == 0 ? LDY : -LDY;
            float xp_next = XP[xo];
#pragma unroll 1
            for (int s = 0; s < NB; ++s) {
                const int par = s & 1;
                const float4* hp4 = reinterpret_cast<const float4*>(Hb + (rd * 2 + par) * HH);
                float* hnext = Hb + (rd * 2 + (par ^ 1)) * HH;
                const float xp_cur = xp_next;
                if (s + 1 < NB) { xo += xd; xp_next = XP[xo]; }
                float4 hq[HH / 4];
#pragma unroll
                for (int k = 0; k < HH / 4; ++k) {
                    if constexpr (ABL & 1) { hq[k] = float4{cstate, xp_cur, cstate, xp_cur}; asm volatile("" : "+v"(hq[k].x), "+v"(hq[k].y), "+v"(hq[k].z), "+v"(hq[k].w)); }
                    else hq[k] = hp4[k];
                }
                f32x2 p0 = {0.0f, 0.0f}, p1 = {0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < ((ABL & 8) ? 1 : HH / 4); ++k) {
                    p0 += f32x2{Whh[4 * k], Whh[4 * k + 1]} * f32x2{hq[k].x, hq[k].y};
                    p1 += f32x2{Whh[4 * k + 2], Whh[4 * k + 3]} * f32x2{hq[k].z, hq[k].w};
                }
                if constexpr (ABL & 8) { for (int k = 1; k < HH / 4; ++k) p0.x += hq[k].x * 0.0f; }
                const f32x2 ps = p0 + p1;
                const float pre = ps.x + ps.y + xp_cur;
                if constexpr (ABL & 4) {
                    const float hn = pre * 0.01f;
                    cstate = hn;
                    hnext[u] = hn;
                    Yf[yo] = hn;
                    yo += yd;
                    if constexpr (!(ABL & 2)) __syncthreads();
                    continue;
                }
                const float act = __builtin_fmaf(sig2(pre), act_m, act_a);
                const int ai = __builtin_bit_cast(int, act);
                const float ig = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0x00, 0xf, 0xf, true));
                const float fg = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0x55, 0xf, 0xf, true));
                const float gg = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0xaa, 0xf, 0xf, true));
                const float og = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0xff, 0xf, 0xf, true));
                const float cn = fg * cstate + ig * gg;
                cstate = cn;
                const float hn = og * (2.0f * sig2(K2 * cn) - 1.0f);
                hnext[u] = hn;
                Yf[yo] = hn;
                yo += yd;
                if constexpr (!(ABL & 2)) __syncthreads();
            }
            for (int i = tid; i < NB * 2 * HH; i += 256) ysum += Yf[(i / (2 * HH)) * LDY + i % (2 * HH)] * (1.0f + 0.001f * (i % 7));
            __syncthreads();
        }
        t1 = __builtin_readcyclecounter();
    } else if constexpr (VAR == 5) {
        // VQ: two waves per direction; a quad of lanes = one hidden unit; lane (p = gate pair, kh = k half): two gate rows over 16 k.
        // kh = 0 holds (row A, row B) = p ? (f, o) : (i, g) over k < 16; kh = 1 holds them SWAPPED (row B, row A) over k >= 16, so that
        // "own first sum + partner's second sum" is gate A's total in lane kh = 0 and gate B's in lane kh = 1: quad lanes = [i, g, f, o]
        const int rd = wave >> 1, rq = tid & 127, ql = rq & 3, u = rq >> 2, kh = ql & 1, pp = ql >> 1;
        const int gA = pp ? 1 : 0, gB = pp ? 3 : 2;                 // canonical gate indices (i f g o) = (0 1 2 3)
        const int g0 = kh ? gB : gA, g1 = kh ? gA : gB;
        float W0[16], W1[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            W0[k] = Wc[((rd * 4 + g0) * HH + u) * HH + kh * 16 + k];
            W1[k] = Wc[((rd * 4 + g1) * HH + u) * HH + kh * 16 + k];
        }
        // lane's own gate after the exchange: ql 0: i, 1: g, 2: f, 3: o.  g: tanh = 2 s - 1, scaled by K2 so that the cell state is kept as K2 * c
        const float act_m = ql == 1 ? 2.0f * K2 : 1.0f, act_a = ql == 1 ? -K2 : 0.0f;
        const int mygate = ql == 0 ? 0 : (ql == 1 ? 2 : (ql == 2 ? 1 : 3));
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int l = 0; l < NLAY; ++l) {
            if (tid < 4 * HH) Hb[tid] = 0.0f;
            __syncthreads();
            float cs = 0.0f;                                     // K2 * c
            const int band0 = rd == 0 ? 0 : NB - 1;
            int xo = (rd * 32 + band0) * LDP + mygate * HH + u, yo = band0 * LDY + rd * HH + u;
            const int xd = rd == 0 ? LDP : -LDP, yd = rd == 0 ? LDY : -LDY;
            float xp_next = XP[xo];
#pragma unroll 1
            for (int s = 0; s < NB; ++s) {
                const int par = s & 1;
                const float4* hp4 = reinterpret_cast<const float4*>(Hb + (rd * 2 + par) * HH + kh * 16);
                float* hnext = Hb + (rd * 2 + (par ^ 1)) * HH;
                const float xp_cur = xp_next;
                if (s + 1 < NB) { xo += xd; xp_next = XP[xo]; }
                float4 hq[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) hq[k] = hp4[k];
                f32x2 p0 = {xp_cur, 0.0f}, p1 = {0.0f, 0.0f};    // (the input projection of the lane's own gate rides in its first sum)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    p0 += f32x2{W0[4 * k], W0[4 * k + 1]} * f32x2{hq[k].x, hq[k].y};
                    p1 += f32x2{W1[4 * k], W1[4 * k + 1]} * f32x2{hq[k].x, hq[k].y};
                    p0 += f32x2{W0[4 * k + 2], W0[4 * k + 3]} * f32x2{hq[k].z, hq[k].w};
                    p1 += f32x2{W1[4 * k + 2], W1[4 * k + 3]} * f32x2{hq[k].z, hq[k].w};
                }
                const float t0s = p0.x + p0.y, t1s = p1.x + p1.y;
                const float pre = t0s + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t1s), 0xB1, 0xf, 0xf, true));   // + partner's second sum
                const float act = __builtin_fmaf(sig2(pre), act_m, act_a);      // ql 0: i, 1: K2 * g, 2: f, 3: o
                const int ai = __builtin_bit_cast(int, act);
                const float gg = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0x55, 0xf, 0xf, true));       // lane 1
                const float ig = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0x00, 0xf, 0xf, true));
                const float fg = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0xaa, 0xf, 0xf, true));
                const float og = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0xff, 0xf, 0xf, true));
                cs = __builtin_fmaf(fg, cs, ig * gg);
                const float th = __builtin_fmaf(sig2(cs), 2.0f, -1.0f);
                const float hn = og * th;
                hnext[u] = hn;
                Yf[yo] = hn;
                yo += yd;
                __syncthreads();
            }
            for (int i = tid; i < NB * 2 * HH; i += 256) ysum += Yf[(i / (2 * HH)) * LDY + i % (2 * HH)] * (1.0f + 0.001f * (i % 7));
            __syncthreads();
        }
        t1 = __builtin_readcyclecounter();
    } else if constexpr (VAR == 6) {
        // S1O: S1 hand-scheduled - h reads as one asm block, FMAs released read by read (lgkmcnt 7 .. 0), four accumulator chains, the
        // cell state kept as K2 * c (one multiply less on the chain), h = fma(2 o, r, -o)
        const int d = wave, u = lane & 31, half = lane >> 5;
        float W0[HH], W1[HH];
        if (wave < 2) {
#pragma unroll
            for (int k = 0; k < HH; ++k) {
                W0[k] = Wc[((d * 4 + half) * HH + u) * HH + k];
                W1[k] = Wc[((d * 4 + 2 + half) * HH + u) * HH + k];
            }
        }
        const float act_m = half == 0 ? 2.0f * K2 : 1.0f, act_a = half == 0 ? -K2 : 0.0f;      // low: K2 * tanh(g), high: o
        __syncthreads();
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int l = 0; l < NLAY; ++l) {
            if (wave < 2) {
                float* hb = Hb + d * (2 * HH);
                if (lane < 2 * HH) hb[lane] = 0.0f;
                float cst = 0.0f;
                const int band0 = d == 0 ? 0 : NB - 1;
                int xo = ((d * 32 + band0) * 64 + lane) * 2, yo = band0 * LDY + d * HH + u;
                const int xd = d == 0 ? 128 : -128, yd = d == 0 ? LDY : -LDY;
                float* ydump = Hb + 4 * HH + u;
                f32x2 xp_cur = *reinterpret_cast<const f32x2*>(XP2 + xo);
                const int hb_addr = (int)(size_t)hb;       // LDS byte address
#pragma unroll 1
                for (int s = 0; s < NB; ++s) {
                    const int par = s & 1;
                    f32x4 hq[8];
                    f32x2 xp_next;
                    xo += (s + 1 < NB) ? xd : 0;
                    const int ha = hb_addr + par * (HH * 4), xa = (int)(size_t)(XP2 + xo);
                    asm volatile("ds_read_b128 %0, %9\n\tds_read_b128 %1, %9 offset:16\n\tds_read_b128 %2, %9 offset:32\n\tds_read_b128 %3, %9 offset:48\n\t"
                                 "ds_read_b128 %4, %9 offset:64\n\tds_read_b128 %5, %9 offset:80\n\tds_read_b128 %6, %9 offset:96\n\tds_read_b128 %7, %9 offset:112\n\t"
                                 "ds_read_b64 %8, %10"
                                 : "=&v"(hq[0]), "=&v"(hq[1]), "=&v"(hq[2]), "=&v"(hq[3]), "=&v"(hq[4]), "=&v"(hq[5]), "=&v"(hq[6]), "=&v"(hq[7]), "=&v"(xp_next)
                                 : "v"(ha), "v"(xa) : "memory");
                    f32x2 p0 = {xp_cur.x, 0.0f}, p1 = {xp_cur.y, 0.0f}, p2 = {0.0f, 0.0f}, p3 = {0.0f, 0.0f};
#define LSTM_STEP_K(k, N) \
                    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(hq[k])); \
                    p0 += f32x2{W0[4 * k], W0[4 * k + 1]} * f32x2{hq[k][0], hq[k][1]}; \
                    p1 += f32x2{W1[4 * k], W1[4 * k + 1]} * f32x2{hq[k][0], hq[k][1]}; \
                    p2 += f32x2{W0[4 * k + 2], W0[4 * k + 3]} * f32x2{hq[k][2], hq[k][3]}; \
                    p3 += f32x2{W1[4 * k + 2], W1[4 * k + 3]} * f32x2{hq[k][2], hq[k][3]}; \
                    __builtin_amdgcn_sched_barrier(0);
                    LSTM_STEP_K(0, 8) LSTM_STEP_K(1, 7) LSTM_STEP_K(2, 6) LSTM_STEP_K(3, 5) LSTM_STEP_K(4, 4) LSTM_STEP_K(5, 3) LSTM_STEP_K(6, 2) LSTM_STEP_K(7, 1)
#undef LSTM_STEP_K
                    const f32x2 q0 = p0 + p2, q1 = p1 + p3;
                    const float a0 = q0.x + q0.y, a1 = q1.x + q1.y;
                    const float s0 = sig2(a0);                                       // low: i, high: f
                    const float r1 = sig2(a1);
                    const float s1 = __builtin_fmaf(r1, act_m, act_a);               // low: K2 * g, high: o
                    const float o2 = s1 + s1, on = -s1;
                    float x = s0, y = s0 * s1;                                       // low y: K2 * i * g
                    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
                    const float cn = __builtin_fmaf(s0, cst, x);                     // high: f * (K2 c) + K2 i g
                    cst = cn;
                    const float hn = __builtin_fmaf(o2, sig2(cn), on);               // high: o * (2 r - 1)
                    *(half ? hb + (par ^ 1) * HH + u : ydump) = hn;
                    *(half ? Yf + yo : ydump) = hn;
                    yo += yd;
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xp_next));            // (the two stores are younger)
                    xp_cur = xp_next;
                }
            }
            __syncthreads();
            for (int i = tid; i < NB * 2 * HH; i += 256) ysum += Yf[(i / (2 * HH)) * LDY + i % (2 * HH)] * (1.0f + 0.001f * (i % 7));
            __syncthreads();
        }
        t1 = __builtin_readcyclecounter();
    } else if constexpr (VAR == 7) {
        // S1K: ONE wave per direction, lane = (unit, k half): all four gate rows over 16 k (4 x ds_read_b128 instead of 8, four independent
        // accumulator chains); the halves are joined with two v_permlane32_swap (+ add): low lanes get the totals of (i, g), high lanes of (f, o)
        const int d = wave, u = lane & 31, half = lane >> 5;
        float W[4][16];
        if (wave < 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < 16; ++k) W[g][k] = Wc[((d * 4 + g) * HH + u) * HH + half * 16 + k];
        }
        const float act_m = half == 0 ? 2.0f * K2 : 1.0f, act_a = half == 0 ? -K2 : 0.0f;      // low: K2 * tanh(g), high: o
        __syncthreads();
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int l = 0; l < NLAY; ++l) {
            if (wave < 2) {
                float* hb = Hb + d * (2 * HH);
                if (lane < 2 * HH) hb[lane] = 0.0f;
                float cst = 0.0f;
                const int band0 = d == 0 ? 0 : NB - 1;
                int xo = ((d * 32 + band0) * 64 + lane) * 2, yo = band0 * LDY + d * HH + u;
                const int xd = d == 0 ? 128 : -128, yd = d == 0 ? LDY : -LDY;
                float* ydump = Hb + 4 * HH + u;
                f32x2 xp_next = *reinterpret_cast<const f32x2*>(XP2 + xo);
#pragma unroll 1
                for (int s = 0; s < NB; ++s) {
                    const int par = s & 1;
                    const f32x2 xp_cur = xp_next;            // low: (i, g), high: (f, o)
                    if (s + 1 < NB) { xo += xd; xp_next = *reinterpret_cast<const f32x2*>(XP2 + xo); }
                    const float4* hp4 = reinterpret_cast<const float4*>(hb + par * HH + half * 16);
                    float4 hq[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) hq[k] = hp4[k];
                    // gate order i f g o; the lane's own two gates start from their input projections
                    f32x2 p[4] = {{half ? 0.0f : xp_cur.x, 0.0f}, {half ? xp_cur.x : 0.0f, 0.0f}, {half ? 0.0f : xp_cur.y, 0.0f}, {half ? xp_cur.y : 0.0f, 0.0f}};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            p[g] += f32x2{W[g][4 * k], W[g][4 * k + 1]} * f32x2{hq[k].x, hq[k].y};
                            p[g] += f32x2{W[g][4 * k + 2], W[g][4 * k + 3]} * f32x2{hq[k].z, hq[k].w};
                        }
                    float ti = p[0].x + p[0].y, tf = p[1].x + p[1].y, tg = p[2].x + p[2].y, to = p[3].x + p[3].y;
                    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ti), "+v"(tf));      // ti = (L_i | L_f), tf = (H_i | H_f)
                    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(tg), "+v"(to));
                    const float a0 = ti + tf, a1 = tg + to;                          // low: (i, g) totals, high: (f, o)
                    const float s0 = sig2(a0);
                    const float s1 = __builtin_fmaf(sig2(a1), act_m, act_a);         // low: K2 * g, high: o
                    const float o2 = s1 + s1, on = -s1;
                    float x = s0, y = s0 * s1;
                    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
                    const float cn = __builtin_fmaf(s0, cst, x);
                    cst = cn;
                    const float hn = __builtin_fmaf(o2, sig2(cn), on);
                    *(half ? hb + (par ^ 1) * HH + u : ydump) = hn;
                    *(half ? Yf + yo : ydump) = hn;
                    yo += yd;
                }
            }
            __syncthreads();
            for (int i = tid; i < NB * 2 * HH; i += 256) ysum += Yf[(i / (2 * HH)) * LDY + i % (2 * HH)] * (1.0f + 0.001f * (i % 7));
            __syncthreads();
        }
        t1 = __builtin_readcyclecounter();
    } else {
        // ---------------- one wave per direction
        const int d = wave, u = lane & 31, half = lane >> 5;
        float W0[HH], W1[HH];
        if (wave < 2) {
#pragma unroll
            for (int k = 0; k < HH; ++k) {
                W0[k] = Wc[((d * 4 + half) * HH + u) * HH + k];            // half 0: i (gate 0), half 1: f (gate 1)
                W1[k] = Wc[((d * 4 + 2 + half) * HH + u) * HH + k];        // half 0: g (gate 2), half 1: o (gate 3)
            }
        }
        const float act_m = half == 0 ? 2.0f : 1.0f, act_a = half == 0 ? -1.0f : 0.0f;      // second row: g (tanh) in the low half, o in the high half
        __syncthreads();
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int l = 0; l < NLAY; ++l) {
            if (wave < 2) {
                float* hb = Hb + d * (2 * HH);                                  // [2 buffers][HH]
                if (lane < 2 * HH) hb[lane] = 0.0f;
                float cst = 0.0f, hreg = 0.0f;
                const int band0 = d == 0 ? 0 : NB - 1;
                int xo = ((d * 32 + band0) * 64 + lane) * 2, yo = band0 * LDY + d * HH + u;
                const int xd = d == 0 ? 128 : -128, yd = d == 0 ? LDY : -LDY;
                // low lanes "store" to a dump slot (no exec-masked region): per-lane addresses
                float* ydump = Hb + 4 * HH + u;
                f32x2 xp_next = *reinterpret_cast<const f32x2*>(XP2 + xo);
#pragma unroll 1
                for (int s = 0; s < NB; ++s) {
                    const int par = s & 1;
                    const f32x2 xp_cur = xp_next;
                    if (s + 1 < NB) { xo += xd; xp_next = *reinterpret_cast<const f32x2*>(XP2 + xo); }
                    float a0, a1;
                    if constexpr (VAR == 1 || VAR == 2) {
                        const float4* hp4 = reinterpret_cast<const float4*>(hb + par * HH);
                        float4 hq[HH / 4];
#pragma unroll
                        for (int k = 0; k < HH / 4; ++k) hq[k] = hp4[k];
                        if constexpr (VAR == 1) {
                            f32x2 p0 = {xp_cur.x, 0.0f}, p1 = {xp_cur.y, 0.0f};
#pragma unroll
                            for (int k = 0; k < HH / 4; ++k) {
                                p0 += f32x2{W0[4 * k], W0[4 * k + 1]} * f32x2{hq[k].x, hq[k].y};
                                p1 += f32x2{W1[4 * k], W1[4 * k + 1]} * f32x2{hq[k].x, hq[k].y};
                                p0 += f32x2{W0[4 * k + 2], W0[4 * k + 3]} * f32x2{hq[k].z, hq[k].w};
                                p1 += f32x2{W1[4 * k + 2], W1[4 * k + 3]} * f32x2{hq[k].z, hq[k].w};
                            }
                            a0 = p0.x + p0.y; a1 = p1.x + p1.y;
                        } else {
                            float q0 = xp_cur.x, q1 = xp_cur.y, q2 = 0.0f, q3 = 0.0f;
#pragma unroll
                            for (int k = 0; k < HH / 4; ++k) {
                                q0 = fmaf(W0[4 * k], hq[k].x, q0); q1 = fmaf(W1[4 * k], hq[k].x, q1);
                                q2 = fmaf(W0[4 * k + 1], hq[k].y, q2); q3 = fmaf(W1[4 * k + 1], hq[k].y, q3);
                                q0 = fmaf(W0[4 * k + 2], hq[k].z, q0); q1 = fmaf(W1[4 * k + 2], hq[k].z, q1);
                                q2 = fmaf(W0[4 * k + 3], hq[k].w, q2); q3 = fmaf(W1[4 * k + 3], hq[k].w, q3);
                            }
                            a0 = q0 + q2; a1 = q1 + q3;
                        }
                    } else {
                        // S4: h[k] of the previous step sits in lane 32 + k: 32 readlanes -> SGPRs
                        float q0 = xp_cur.x, q1 = xp_cur.y, q2 = 0.0f, q3 = 0.0f;
                        const int hi_ = __builtin_bit_cast(int, hreg);
#pragma unroll
                        for (int k = 0; k < HH; k += 2) {
                            const float h0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(hi_, 32 + k));
                            const float h1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(hi_, 33 + k));
                            q0 = fmaf(W0[k], h0, q0); q1 = fmaf(W1[k], h0, q1);
                            q2 = fmaf(W0[k + 1], h1, q2); q3 = fmaf(W1[k + 1], h1, q3);
                        }
                        a0 = q0 + q2; a1 = q1 + q3;
                    }
                    const float s0 = sig2(a0);                                       // low: i, high: f
                    const float s1 = __builtin_fmaf(sig2(a1), act_m, act_a);         // low: g (tanh), high: o
                    float x = s0, y = s0 * s1;                                       // low y: i * g
                    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));       // x.hi <- y.lo: high lanes now hold i * g in x
                    const float cn = __builtin_fmaf(s0, cst, x);                     // high: f * c + i * g   (low lanes: bounded garbage)
                    cst = cn;
                    const float hn = s1 * (2.0f * sig2(K2 * cn) - 1.0f);             // high: o * tanh(c)
                    hreg = hn;
                    if constexpr (VAR != 3) *(half ? hb + (par ^ 1) * HH + u : ydump) = hn;
                    *(half ? Yf + yo : ydump) = hn;
                    yo += yd;
                }
            }
            __syncthreads();
            for (int i = tid; i < NB * 2 * HH; i += 256) ysum += Yf[(i / (2 * HH)) * LDY + i % (2 * HH)] * (1.0f + 0.001f * (i % 7));
            __syncthreads();
        }
        t1 = __builtin_readcyclecounter();
    }
    Yout[(size_t)blockIdx.x * 256 + tid] = ysum;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t0; clk[1] = t1; }
    if (blockIdx.x == 0) for (int i = tid; i < NB * 2 * HH; i += 256) Yout[(size_t)gridDim.x * 256 + i] = Yf[(i / (2 * HH)) * LDY + i % (2 * HH)];
}

static void reference(const std::vector<float>& W, const std::vector<float>& XP, std::vector<double>& Y) {
    Y.assign(NB * 2 * HH, 0.0);
    for (int d = 0; d < 2; ++d) {
        double h[HH] = {0}, c[HH] = {0};
        for (int s = 0; s < NB; ++s) {
            const int band = d == 0 ? s : NB - 1 - s;
            double a[4][HH];
            for (int g = 0; g < 4; ++g)
                for (int u = 0; u < HH; ++u) {
                    double acc = XP[((d * NB + band) * 4 + g) * HH + u];
                    for (int k = 0; k < HH; ++k) acc += (double)W[((d * 4 + g) * HH + u) * HH + k] * h[k];
                    a[g][u] = 1.0 / (1.0 + std::exp2(acc));
                }
            for (int u = 0; u < HH; ++u) {
                c[u] = a[1][u] * c[u] + a[0][u] * (2.0 * a[2][u] - 1.0);
                h[u] = a[3][u] * (2.0 / (1.0 + std::exp2((double)K2 * c[u])) - 1.0);
                Y[(band * 2 + d) * HH + u] = h[u];
            }
        }
    }
}

template <int VAR, int ABL = 0>
static void run(const char* name, const float* Wd, const float* XPd, const std::vector<double>& ref, int blocks) {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, ((size_t)blocks * 256 + NB * 2 * HH) * 4); (void)hipMalloc(&clk, 16);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((kern<VAR, ABL>), dim3(blocks), dim3(256), 0, 0, Wd, XPd, out, clk);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((kern<VAR, ABL>), dim3(blocks), dim3(256), 0, 0, Wd, XPd, out, clk);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    std::vector<float> y(NB * 2 * HH);
    (void)hipMemcpy(y.data(), out + (size_t)blocks * 256, y.size() * 4, hipMemcpyDeviceToHost);
    double err = 0, nrm = 0;
    for (size_t i = 0; i < y.size(); ++i) { err += (y[i] - ref[i]) * (y[i] - ref[i]); nrm += ref[i] * ref[i]; }
    printf("%-22s blocks=%4d  %7.1f cycles per step (both directions; %llu cycles for %d layers x %d steps incl. per-layer read-out)   kernel %.2f us   rel rms err vs f64 %.2e\n",
           name, blocks, (double)(h[1] - h[0]) / (NLAY * NB), h[1] - h[0], NLAY, NB, ms * 1e3 / 20, std::sqrt(err / nrm));
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    std::vector<float> W(2 * 4 * HH * HH), XP(2 * NB * 4 * HH);
    srand(7);
    for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
    for (auto& v : XP) v = (rand() / (float)RAND_MAX - 0.5f) * 3.0f;
    std::vector<double> ref;
    reference(W, XP, ref);
    float *Wd, *XPd;
    (void)hipMalloc(&Wd, W.size() * 4); (void)hipMalloc(&XPd, XP.size() * 4);
    (void)hipMemcpy(Wd, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(XPd, XP.data(), XP.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {1, 256}) {
        run<0>("V0", Wd, XPd, ref, blocks);
        run<1>("S1", Wd, XPd, ref, blocks);
        run<2>("S1F", Wd, XPd, ref, blocks);
        run<3>("S4", Wd, XPd, ref, blocks);
        run<5>("VQ", Wd, XPd, ref, blocks);
        run<6>("S1O", Wd, XPd, ref, blocks);
        run<7>("S1K", Wd, XPd, ref, blocks);
        if (blocks == 256 && getenv("LSTM_ABL")) {      // ablations of V0 (results are wrong by construction: timing only)
            run<0, 1>("V0 -hreads", Wd, XPd, ref, blocks);
            run<0, 2>("V0 -barrier", Wd, XPd, ref, blocks);
            run<0, 4>("V0 -activations", Wd, XPd, ref, blocks);
            run<0, 8>("V0 -fmas", Wd, XPd, ref, blocks);
            run<0, 3>("V0 -hreads -barrier", Wd, XPd, ref, blocks);
            run<0, 9>("V0 -hreads -fmas", Wd, XPd, ref, blocks);
            run<0, 13>("V0 -hreads -fmas -act", Wd, XPd, ref, blocks);
            run<0, 15>("V0 -all", Wd, XPd, ref, blocks);
        }
    }
    return 0;
}
