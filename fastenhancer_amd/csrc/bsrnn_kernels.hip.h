// bsrnn_kernels.hip.h — BSRNN (models/bsrnn/model.py of the reference) streaming / offline forward for gfx950.
//
// Same execution model as fe_kernels.hip.h: one workgroup (256 threads) owns one stream and runs the whole frame
//   STFT -> compress (all 257 bins) -> band split (31 bands) -> L x [time-LSTM, bidirectional band-LSTM]
//        -> per-band mask/residual MLPs (GLU) -> complex mask + residual -> un-compress -> iSTFT
// with every activation in LDS.  The batched contractions (LSTM gate pre-activations of the 31 bands, the fc layers)
// run on the fp32 matrix cores; the band-LSTM recurrence (31 sequential steps per direction, a 1 x 2C by 2C x 8C
// product each) runs on the vector ALUs with one gate row per thread whose W_hh row stays in registers for the
// whole layer, the four gates of a hidden unit sitting in adjacent lanes (quad shuffles, no LDS round trip).
#pragma once
#include <atomic>

#include "fe_kernels.hip.h"

namespace fe {

constexpr int kBands = 31;
constexpr int kBins = 257;
__device__ __constant__ const int c_sub[kBands] = {2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 16, 16, 16, 16, 16, 16, 16, 17};

// compile-time shape: C = num_channels, NLAY = num_layers (n_fft = 512 is fixed by the model: models/bsrnn/model.py:112)
template <int C_, int NLAY_, int HOP_>
struct BShape {
    static constexpr int C = C_, NLAY = NLAY_, HOP = HOP_, NFFT = 512, LOG2N = 9;
    static constexpr int HH = 2 * C;            // LSTM hidden size
    static constexpr int G4 = 4 * HH;           // gate rows
    static constexpr int OVL = NFFT - HOP;
    static constexpr int LDX = C + 2, LDH = HH + 2, LDY = 2 * HH + 2, LDP = G4 + 2;
    static constexpr int NCT = HH / 16;         // hidden-unit tiles
    static constexpr int NTC = C / 16;          // channel tiles
    static constexpr int RPT = (2 * G4) / kThreads;   // gate rows per thread in the band-LSTM recurrence (both directions)
    static_assert(C % 16 == 0 && (RPT == 1 || RPT == 2), "num_channels must be 16 or 32");
};

// offsets (floats) into the packed weight buffer; filled by the host packer (fe_api.hip)
struct BOffsets {
    int bs_w[kBands];          // band split: [2*sub][C]  (k-major)
    int bs_b;                  // [31][C]
    int t_w[8], t_b[8];        // time LSTM: B fragments, K = C + HH (x rows then h rows), N = 4*HH ; bias b_ih + b_hh
    int tfc_w[8], tfc_b[8];    // fc_time: B fragments K = HH, N = C
    int f_wih[8][2], f_b[8][2], f_whh[8][2];   // band LSTM per direction: B fragments K = C, N = 4HH; bias; W_hh as [rr][k][q] (thread order)
    int ffc_w[8], ffc_b[8];    // fc_freq: B fragments K = 2HH, N = C
    int m_w1[2], m_b1[2];      // mask decoder layer 1 per kind: [31][C][4C] (k-major), [31][4C]
    int m_w2[2], m_b2[2];      // layer 2 per kind: [4C][1028] (k-major over the global row index), [1028]
    int window, window_istft, twiddle;
    int total;
};

struct BArgs {
    const float* wp;
    BOffsets off;
    const float* wav_in;
    float* wav_out;
    size_t in_stride, out_stride;
    float* cache_stft;
    float* cache_istft;
    float* lstm;              // [2*NLAY][B*31][HH]  (h0, c0, h1, c1, ...)
    const float* spec_in;     // spec mode [B][257][T][2]
    float* spec_out;
    int B, T, mode, Tw;
    float compression;
    unsigned long long* clk;  // fe_profile_step: cycle probes of workgroup 0 (PROF instantiation only)
};

template <class S>
struct BLds {
    static constexpr int SP = 0;                              // compressed spectrum [257][2]
    static constexpr int TW = SP + 2 * kBins + 2;             // twiddles
    static constexpr int FA = TW + S::NFFT;                   // FFT ping-pong
    static constexpr int FB = FA + 2 * S::NFFT;
    static constexpr int X = FB + 2 * S::NFFT;                // [32][LDX] band features
    static constexpr int HS = X + 32 * S::LDX;                // [32][LDH] time-LSTM h (A operand)
    static constexpr int HN = HS + 32 * S::LDH;               // [32][LDH] new h (A operand of fc_time)
    static constexpr int XP = HN + 32 * S::LDH;               // [2][32][LDP] band-LSTM input projections
    static constexpr int YF = XP + 2 * 32 * S::LDP;           // [32][LDY] band-LSTM outputs (fwd | bwd)
    static constexpr int HB = (YF + 32 * S::LDY + 3) / 4 * 4;   // [2 dirs][2 buffers][HH], 16-byte aligned (float4 broadcast reads)
    static constexpr int H1 = HB + 4 * S::HH;                 // [2 kinds][31][4C]
    static constexpr int TOTAL = H1 + 2 * kBands * 4 * S::C;
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
};

// HOT: the per-hop streaming step (mode and T = 1 are compile-time facts: no frame loop, no spec / offline branches)
#define BE_CLK(i) do { if constexpr (PROF) { if (blockIdx.x == 0 && threadIdx.x == 0) a.clk[(i)] = __builtin_readcyclecounter(); } } while (0)
// PROF: cycle probes per phase (fe_profile_step, tools/gpu_phases_bsrnn.py)
template <class S, bool HOT, bool PROF>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) bsrnn_frame_kernel(BArgs a) {
    const int aT = HOT ? 1 : a.T;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = BLds<S>;
    constexpr int N = S::NFFT, H = S::HOP, OVL = S::OVL, C = S::C, HH = S::HH, G4 = S::G4;
    constexpr int LDX = S::LDX, LDH = S::LDH, LDY = S::LDY, LDP = S::LDP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x;
    const float* __restrict__ wp = a.wp;
    const BOffsets& o = a.off;
    WSrc<false> wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, o.total * 4, 0x00020000);
    wb.lane4 = lane * 4;
    wb.li4 = li * 4;
    wb.lds = nullptr;
    wb.base = 0;

    float* sp = smem + L::SP;
    float2* tw = reinterpret_cast<float2*>(smem + L::TW);
    float2* fa = reinterpret_cast<float2*>(smem + L::FA);
    float2* fb = reinterpret_cast<float2*>(smem + L::FB);
    float* X = smem + L::X;
    float* Hs = smem + L::HS;
    float* Hn = smem + L::HN;
    float* XP = smem + L::XP;
    float* Yf = smem + L::YF;
    float* Hb = smem + L::HB;
    float* H1 = smem + L::H1;
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = reinterpret_cast<const float2*>(wp + o.twiddle)[i];
    __syncthreads();

    float* cst = a.cache_stft + (size_t)b * OVL;
    float* cis = a.cache_istft + (size_t)b * OVL;
    const int mode = HOT ? FE_MODE_STREAM : a.mode;
    // band of each bin (for the mask decoder's row bookkeeping): bin f -> band start / width
    auto band_of = [&](int f, int& start, int& sub) {
        int s0 = 0;
#pragma unroll 1
        for (int bb = 0; bb < kBands; ++bb) {
            const int sb = c_sub[bb];
            if (f < s0 + sb) { start = s0; sub = sb; return bb; }
            s0 += sb;
        }
        start = 0; sub = 1;
        return 0;
    };

#pragma unroll 1
    for (int t = 0; t < aT; ++t) {
        BE_CLK(0);
        // ============================ STFT + compress (all 257 bins; models/bsrnn/model.py:430-436) ============================
        if (mode != FE_MODE_SPEC) {
            const float* win = wp + o.window;
            if (mode == FE_MODE_STREAM) {
                const float* xin = a.wav_in + (size_t)b * a.in_stride + (size_t)t * H;
                for (int n = tid; n < N; n += kThreads) {
                    float v = (n < OVL) ? cst[n] : xin[n - OVL];
                    fb[n] = make_float2(v, 0.0f);
                    fa[n] = make_float2(v * win[n], 0.0f);
                }
            } else {
                const float* xin = a.wav_in + (size_t)b * a.in_stride;
                for (int n = tid; n < N; n += kThreads) {
                    int idx = t * H + n - N / 2;
                    idx = idx < 0 ? -idx : idx;
                    idx = idx >= a.Tw ? 2 * (a.Tw - 1) - idx : idx;
                    fa[n] = make_float2(xin[idx] * win[n], 0.0f);
                }
            }
            __syncthreads();
            if (mode == FE_MODE_STREAM) {
                for (int m = tid; m < OVL; m += kThreads) cst[m] = fb[m + H].x;
                __syncthreads();
            }
            float2* Xf = fft_lds<S, false>(fa, fb, tw);
            for (int f = tid; f < kBins; f += kThreads) {
                const float re = Xf[f].x, im = Xf[f].y;
                const float g = pow_f(fmaxf(sqrtf(re * re + im * im), 1.0e-5f), a.compression - 1.0f);
                sp[2 * f] = re * g;
                sp[2 * f + 1] = im * g;
            }
        } else {
            const float* si = a.spec_in + (size_t)b * kBins * aT * 2;
            for (int f = tid; f < kBins; f += kThreads) {
                const float re = si[((size_t)f * aT + t) * 2], im = si[((size_t)f * aT + t) * 2 + 1];
                const float g = pow_f(fmaxf(sqrtf(re * re + im * im), 1.0e-5f), a.compression - 1.0f);
                sp[2 * f] = re * g;
                sp[2 * f + 1] = im * g;
            }
        }
        __syncthreads();

        BE_CLK(1);
        // ============================ band split (BandSplit.forward, :136-153; BN folded) ============================
        for (int i = tid; i < kBands * C; i += kThreads) {
            const int bb = i / C, c = i - bb * C;
            int s0 = 0;
            for (int q = 0; q < bb; ++q) s0 += c_sub[q];
            const int k2 = 2 * c_sub[bb];
            const float* w = wp + o.bs_w[bb] + c;
            float acc = wp[o.bs_b + bb * C + c];
            for (int k = 0; k < k2; ++k) acc += w[k * C] * sp[2 * s0 + k];      // input index f*2 + ri
            X[bb * LDX + c] = acc;
        }
        __syncthreads();

#pragma unroll 1
        for (int l = 0; l < S::NLAY; ++l) {
            float* hg = a.lstm + ((size_t)(2 * l) * a.B + b) * (kBands * HH);
            float* cg = a.lstm + ((size_t)(2 * l + 1) * a.B + b) * (kBands * HH);
            if (l == 0) BE_CLK(2);
            // ---------------- time LSTM (LSTMCell over the 31 bands; :371-381): h -> LDS
            for (int i = tid; i < kBands * HH; i += kThreads) { const int r = i / HH; Hs[r * LDH + (i - r * HH)] = hg[i]; }
            __syncthreads();
            {
                // items (m-tile, hidden tile): 4 gate accumulators each; gates fused into the epilogue (order i,f,g,o)
                constexpr int NITEM = 2 * S::NCT;
#pragma unroll 1
                for (int it = wave; it < NITEM; it += kWaves) {
                    const int mt = it / S::NCT, ct = it - mt * S::NCT;
                    f32x4 acc[1][4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float bv = wb.at16_g(o.t_b[l] + g * HH + ct * 16);
                        acc[0][g] = f32x4{bv, bv, bv, bv};
                    }
                    const float* xa = X + (16 * mt + li) * LDX + lg;
                    const float* ha = Hs + (16 * mt + li) * LDH + lg;
                    constexpr int KSX = C / 4, KSH = HH / 4;
                    mma_panel<1, 4, KSX + KSH>(
                        acc, [&](int, int ks) { return ks < KSX ? xa[4 * ks] : ha[4 * (ks - KSX)]; },
                        [&](int g, int ks) { return wb.at_g(o.t_w[l] + ((g * S::NCT + ct) * (KSX + KSH) + ks) * 64); }, NoSide{});
                    const int j = 16 * ct + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * lg + r;
                        if (row < kBands) {
                            const float ig = sigmoid_f(acc[0][0][r]), fg = sigmoid_f(acc[0][1][r]);
                            const float gg = tanh_f(acc[0][2][r]), og = sigmoid_f(acc[0][3][r]);
                            const float cn = fg * cg[row * HH + j] + ig * gg;
                            const float hn = og * tanh_f(cn);
                            cg[row * HH + j] = cn;
                            hg[row * HH + j] = hn;
                            Hn[row * LDH + j] = hn;
                        }
                    }
                }
            }
            __syncthreads();
            if (l == 0) BE_CLK(3);
            {
                // fc_time + residual (:382-384): X += Hn W^T + b
                constexpr int NITEM = 2 * S::NTC;
#pragma unroll 1
                for (int it = wave; it < NITEM; it += kWaves) {
                    const int mt = it / S::NTC, nt = it - mt * S::NTC;
                    f32x4 acc[1][1];
                    const float bv = wb.at16_g(o.tfc_b[l] + nt * 16);
                    acc[0][0] = f32x4{bv, bv, bv, bv};
                    const float* ha = Hn + (16 * mt + li) * LDH + lg;
                    mma_panel<1, 1, HH / 4>(acc, [&](int, int ks) { return ha[4 * ks]; },
                                            [&](int, int ks) { return wb.at_g(o.tfc_w[l] + (nt * (HH / 4) + ks) * 64); }, NoSide{});
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * lg + r;
                        if (row < kBands) X[row * LDX + 16 * nt + li] += acc[0][0][r];
                    }
                }
            }
            __syncthreads();
            {
                if (l == 0) BE_CLK(4);
                // ---------------- band LSTM (:386-390): input projections of all 31 bands, both directions
                constexpr int NTP = 2 * (G4 / 16);       // n-tiles: direction-major
#pragma unroll 1
                for (int nt = wave; nt < NTP; nt += kWaves) {
                    const int d = nt / (G4 / 16), ntd = nt - d * (G4 / 16);
                    f32x4 acc[2][1];
                    const float bv = wb.at16_g(o.f_b[l][d] + ntd * 16);
                    acc[0][0] = f32x4{bv, bv, bv, bv};
                    acc[1][0] = acc[0][0];
                    mma_panel<2, 1, C / 4>(acc, [&](int i, int ks) { return X[(16 * i + li) * LDX + lg + 4 * ks]; },
                                           [&](int, int ks) { return wb.at_g(o.f_wih[l][d] + (ntd * (C / 4) + ks) * 64); }, NoSide{});
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) XP[(d * 32 + 16 * i + 4 * lg + r) * LDP + 16 * ntd + li] = acc[i][0][r];
                }
            }
            // recurrence: thread <-> (direction, hidden unit j, gate) with the 4 gates of a unit in adjacent lanes
            const int d = tid / (kThreads / 2);
            const int q = tid - d * (kThreads / 2);             // 0..127
            const int gate = q & 3;
            float wrow[S::RPT][HH];
            float cstate[S::RPT];
#pragma unroll
            for (int rr = 0; rr < S::RPT; ++rr) {
                const int j = (q >> 2) + 32 * rr;
                const float* wr = wp + o.f_whh[l][d] + rr * HH * (kThreads / 2) + q;   // [rr][k][q]: coalesced over the threads
#pragma unroll
                for (int k = 0; k < HH; ++k) wrow[rr][k] = wr[k * (kThreads / 2)];
                (void)j;
                cstate[rr] = 0.0f;
            }
            if (tid < 4 * HH) Hb[tid] = 0.0f;                    // h = 0 for both directions, both buffers
            __syncthreads();
            if (l == 0) BE_CLK(5);
#pragma unroll 1
            for (int s = 0; s < kBands; ++s) {
                const int band = d == 0 ? s : kBands - 1 - s;
                const float* hprev = Hb + (d * 2 + (s & 1)) * HH;
                float* hnext = Hb + (d * 2 + ((s + 1) & 1)) * HH;
#pragma unroll
                for (int rr = 0; rr < S::RPT; ++rr) {
                    const int j = (q >> 2) + 32 * rr;
                    const float4* hp4 = reinterpret_cast<const float4*>(hprev);      // broadcast reads, 16 B each
                    // four partial sums: one serial chain of HH dependent FMAs per step is the recurrence's critical path
                    float p0 = XP[(d * 32 + band) * LDP + gate * HH + j], p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
                    for (int k = 0; k < HH / 4; ++k) {
                        const float4 hv = hp4[k];
                        p0 += wrow[rr][4 * k] * hv.x;
                        p1 += wrow[rr][4 * k + 1] * hv.y;
                        p2 += wrow[rr][4 * k + 2] * hv.z;
                        p3 += wrow[rr][4 * k + 3] * hv.w;
                    }
                    const float pre = (p0 + p1) + (p2 + p3);
                    // one exp + one rcp for either activation: tanh(x) = 2 s(2x) - 1  (a select between tanh_f and sigmoid_f
                    // evaluates both: four quarter-rate transcendentals on the recurrence's critical path)
                    const bool is_g = gate == 2;
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre * (is_g ? -2.8853900817779268f : -1.4426950408889634f)));
                    const float act = is_g ? 2.0f * sg - 1.0f : sg;
                    // the 4 gates of a unit sit in one quad: DPP quad_perm broadcasts (no LDS crossbar)
                    const int ai = __builtin_bit_cast(int, act);
                    const float ig = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0x00, 0xf, 0xf, true));
                    const float fg = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0x55, 0xf, 0xf, true));
                    const float gg = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0xaa, 0xf, 0xf, true));
                    const float og = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0xff, 0xf, 0xf, true));
                    const float cn = fg * cstate[rr] + ig * gg;
                    cstate[rr] = cn;
                    if (gate == 0) {
                        const float hn = og * tanh_f(cn);
                        hnext[j] = hn;
                        Yf[band * LDY + d * HH + j] = hn;
                    }
                }
                __syncthreads();
            }
            if (l == 0) BE_CLK(6);
            {
                // fc_freq + residual: X += Yf W^T + b
                constexpr int NITEM = 2 * S::NTC;
#pragma unroll 1
                for (int it = wave; it < NITEM; it += kWaves) {
                    const int mt = it / S::NTC, nt = it - mt * S::NTC;
                    f32x4 acc[1][1];
                    const float bv = wb.at16_g(o.ffc_b[l] + nt * 16);
                    acc[0][0] = f32x4{bv, bv, bv, bv};
                    const float* ya = Yf + (16 * mt + li) * LDY + lg;
                    mma_panel<1, 1, 2 * HH / 4>(acc, [&](int, int ks) { return ya[4 * ks]; },
                                                [&](int, int ks) { return wb.at_g(o.ffc_w[l] + (nt * (2 * HH / 4) + ks) * 64); }, NoSide{});
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * lg + r;
                        if (row < kBands) X[row * LDX + 16 * nt + li] += acc[0][0][r];
                    }
                }
            }
            __syncthreads();
            if (l == 0) BE_CLK(7);
        }

        BE_CLK(8);
        // ============================ mask decoder (MaskDecoder.forward, :225-246) ============================
        for (int i = tid; i < 2 * kBands * 4 * C; i += kThreads) {
            const int kind = i / (kBands * 4 * C), rem = i - kind * (kBands * 4 * C);
            const int bb = rem / (4 * C), oo = rem - bb * (4 * C);
            const float* w = wp + o.m_w1[kind] + (bb * C) * (4 * C) + oo;
            float acc = wp[o.m_b1[kind] + bb * 4 * C + oo];
#pragma unroll 16
            for (int k = 0; k < C; ++k) acc += w[k * 4 * C] * X[bb * LDX + k];
            H1[i] = tanh_f(acc);
        }
        __syncthreads();
        BE_CLK(9);
        {
            constexpr int R = 4 * kBins;            // rows of the second layers (1028)
            // item = (bin f, kind): GLU outputs (re, im) of that MLP for that bin -> MR[f][kind*2 + ri]  (reuses XP)
            float* MR = XP;
            for (int it = tid; it < 2 * kBins; it += kThreads) {
                const int kind = it / kBins, f = it - kind * kBins;
                int s0, sub;
                const int bb = band_of(f, s0, sub);
                const float* h1 = H1 + (kind * kBands + bb) * 4 * C;
                const float* w2 = wp + o.m_w2[kind];
                const int rowA0 = 4 * s0 + (f - s0) * 2, rowB0 = rowA0 + 2 * sub;
                float va0 = wp[o.m_b2[kind] + rowA0], va1 = wp[o.m_b2[kind] + rowA0 + 1];
                float vb0 = wp[o.m_b2[kind] + rowB0], vb1 = wp[o.m_b2[kind] + rowB0 + 1];
#pragma unroll 16
                for (int k = 0; k < 4 * C; ++k) {
                    const float hv = h1[k];
                    va0 += w2[k * R + rowA0] * hv;
                    va1 += w2[k * R + rowA0 + 1] * hv;
                    vb0 += w2[k * R + rowB0] * hv;
                    vb1 += w2[k * R + rowB0 + 1] * hv;
                }
                MR[f * 4 + kind * 2 + 0] = va0 * (1.0f / (1.0f + expf(-vb0)));      // GLU(dim=1)
                MR[f * 4 + kind * 2 + 1] = va1 * (1.0f / (1.0f + expf(-vb1)));
            }
            __syncthreads();
            float* spo = mode == FE_MODE_SPEC ? a.spec_out + (size_t)b * kBins * aT * 2 : nullptr;
            float* sph = mode == FE_MODE_OFFLINE ? a.spec_out + (size_t)b * kBins * aT * 2 : nullptr;
            for (int f = tid; f < kBins; f += kThreads) {
                const float xr = sp[2 * f], xi = sp[2 * f + 1];
                const float* mr = MR + f * 4;
                float yr = xr * mr[0] - xi * mr[1] + mr[2];             // spec * mask + residual (:393-401)
                float yi = xr * mr[1] + xi * mr[0] + mr[3];
                if (sph != nullptr) { sph[((size_t)f * aT + t) * 2] = yr; sph[((size_t)f * aT + t) * 2 + 1] = yi; }
                const float g = pow_f(sqrtf(yr * yr + yi * yi), 1.0f / a.compression - 1.0f);
                yr *= g;
                yi *= g;
                if (mode == FE_MODE_SPEC) {
                    spo[((size_t)f * aT + t) * 2] = yr;
                    spo[((size_t)f * aT + t) * 2 + 1] = yi;
                } else if (f == 0) {
                    fa[0] = make_float2(yr, 0.0f);
                } else if (f == N / 2) {
                    fa[N / 2] = make_float2(yr, 0.0f);        // irfft keeps only Re X[N/2]
                } else {
                    fa[f] = make_float2(yr, yi);
                    fa[N - f] = make_float2(yr, -yi);
                }
            }
        }
        __syncthreads();

        BE_CLK(10);
        // ============================ iSTFT (functional/audio_modules.py:259-303 / torch.istft) ============================
        if (mode != FE_MODE_SPEC) {
            float2* y = fft_lds<S, true>(fa, fb, tw);
            float2* spare = (y == fa) ? fb : fa;
            const float* wi = wp + (mode == FE_MODE_STREAM ? o.window_istft : o.window);
            float* xo = reinterpret_cast<float*>(spare);
            const float invN = 1.0f / (float)N;
            for (int n = tid; n < N; n += kThreads) {
                float v = y[n].x * invN * wi[n];
                if (n < OVL) v += cis[n];
                xo[n] = v;
            }
            __syncthreads();
            if (mode == FE_MODE_STREAM) {
                float* out = a.wav_out + (size_t)b * a.out_stride + (size_t)t * H;
                for (int n = tid; n < H; n += kThreads) out[n] = xo[n];
            } else {
                const float* w = wp + o.window;
                const int n_out = H * (aT - 1);
                const int emit = (t == aT - 1) ? N : H;
                float* out = a.wav_out + (size_t)b * a.out_stride;
                for (int j = tid; j < emit; j += kThreads) {
                    const int n = t * H + j, pos = n - N / 2;
                    if (pos >= 0 && pos < n_out) {
                        int t_lo = (n - N + H) / H;
                        t_lo = t_lo < 0 ? 0 : t_lo;
                        int t_hi = n / H;
                        t_hi = t_hi > aT - 1 ? aT - 1 : t_hi;
                        float env = 0.0f;
                        for (int tt = t_lo; tt <= t_hi; ++tt) { const float wv = w[n - tt * H]; env += wv * wv; }
                        out[pos] = xo[j] / env;
                    }
                }
            }
            for (int m = tid; m < OVL; m += kThreads) cis[m] = xo[m + H];
            __syncthreads();
        }
        BE_CLK(11);
    }
}

struct BImpl {
    int C, NLAY, HOP;
    size_t lds_bytes;
    void (*launch)(const BArgs&, hipStream_t, hipError_t*);
};

template <class S, bool HOT, bool PROF>
void blaunch_one(const BArgs& a, hipStream_t st, hipError_t* err) {
    static std::atomic<bool> attr_set[64];           // per device (see fe_impl.h::launch_one)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bsrnn_frame_kernel<S, HOT, PROF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)BLds<S>::BYTES);
        if (e != hipSuccess) { *err = e; return; }
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((bsrnn_frame_kernel<S, HOT, PROF>), dim3(a.B), dim3(kThreads), BLds<S>::BYTES, st, a);
    *err = hipGetLastError();
}

template <class S>
void blaunch_impl(const BArgs& a, hipStream_t st, hipError_t* err) {
    if (a.clk != nullptr) {                                                          // fe_profile_step
        if (a.mode == FE_MODE_STREAM && a.T == 1) blaunch_one<S, true, true>(a, st, err);
        else blaunch_one<S, false, true>(a, st, err);
    }
    else if (a.mode == FE_MODE_STREAM && a.T == 1) blaunch_one<S, true, false>(a, st, err);
    else blaunch_one<S, false, false>(a, st, err);
}

template <class S>
BImpl make_bimpl() { return BImpl{S::C, S::NLAY, S::HOP, BLds<S>::BYTES, &blaunch_impl<S>}; }

}  // namespace fe
