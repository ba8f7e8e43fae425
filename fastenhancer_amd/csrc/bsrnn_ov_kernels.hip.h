// bsrnn_ov_kernels.hip.h — PART 1 of BSRNN's per-hop step (STFT -> compress -> band split -> L x [time LSTM, band LSTM]) for
// num_channels = 16 (xt / xxt) at ONE stream per CU, with the layers' matrix-core work taken off the recurrence's critical path.
//
// bsrnn_frame_kernel runs a layer as five barrier-separated phases on all four waves: time-LSTM gates, fc_time, the band LSTM's input
// projections, the band recurrence (31 dependent steps per direction), fc_freq - 7.9 k + 21.1 k cycles per layer
// (profiles/r5a_phases_bsrnn_xt.txt).  What the next layer needs of band j - x[j] after fc_freq - exists as soon as BOTH scans have passed
// band j, i.e. after step max(j, 30 - j): the middle bands long before the recurrence ends.  Here the waves have ROLES:
//   waves 0, 1   the forward / backward scan, one wave per direction (two gate rows per lane, W_hh rows in registers, h through LDS with
//                no barrier: a wave's LDS operations execute in order; tools/micro/lstm_step.hip S1: 569 cycles per step against 585-680
//                with a workgroup barrier per step) - nothing else
//   waves 2, 3   everything on the matrix cores, for two row tiles of bands that are cut by readiness, not by index:
//                tile A = bands 8..22 (ready after step 22), tile B = bands 0..7 and 23..30 (ready after step 30).  Per tile the chain
//                fc_freq (layer l) -> x -> time-LSTM gates (layer l + 1) -> fc_time -> input projections of layer l + 1; tile A's chain
//                runs under steps 23..30 of layer l's scans, only tile B's is serial with them.  The h half of the time-LSTM gates
//                (W_hh h_{t-1}: known since the last frame) is accumulated under the early steps.
// Synchronisation is by monotonic counters in LDS (scan progress per direction, a two-wave rendezvous inside a chain, "projections of
// layer l complete"), polled; the input projections are double-buffered by layer parity.  The two helper waves split the gate GEMM by
// hidden tile and the projections by direction, and compute the narrow fc layers (N = 16) redundantly - identical values written to the
// same LDS words - so that a chain needs ONE rendezvous (the new time-LSTM h, whose two column halves meet in fc_time).
// Reference: models/bsrnn/model.py:367-390 (the layer loop), :249-257 (ONNXLSTM), :136-153 (BandSplit).
#pragma once

namespace fe {

template <class S>
struct BOvLds {
    static constexpr int SP = 0;                              // compressed spectrum [257][2]
    static constexpr int TW = SP + 2 * kBins + 2;             // twiddles
    static constexpr int FA = TW + S::NFFT;                   // FFT ping-pong
    static constexpr int FB = FA + 2 * S::NFFT;
    static constexpr int XB = FB + 2 * S::NFFT;               // (FA: windowed frame, FB: spectrum {Re, Im} + Nyquist)   [32][LDX] band features before a layer's time LSTM (after fc_freq)
    static constexpr int XA = XB + 32 * S::LDX;               // [32][LDX] ... after fc_time (input of the projections and of fc_freq's residual)
    static constexpr int HN = XA + 32 * S::LDX;               // [32][LDH] the time LSTM's new h (A operand of fc_time)
    static constexpr int YF = HN + 32 * S::LDH;               // [32][LDY] band-LSTM outputs (fwd | bwd)
    static constexpr int HB = (YF + 32 * S::LDY + 3) / 4 * 4; // [2 dirs][2 buffers][HH] + [HH] dump slots of the low lanes
    static constexpr int FLG = HB + 5 * S::HH;                // counters (ints): scan progress fwd / bwd, helper rendezvous x 2, projections ready
    static constexpr int XP = FLG + 16;                       // [2 layer parities][2 dirs][32 bands][64 lanes][2]: gate rows in the scan's lane order
    static constexpr int XPBUF = 2 * 32 * 128;
    static constexpr int TOTAL = XP + 2 * XPBUF;
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
    static_assert(XP % 2 == 0 && HB % 4 == 0, "aligned f32x2 / float4 LDS reads");
    static_assert(BYTES <= 160 * 1024, "BSRNN role-split LDS plan exceeds 160 KiB");
};

// monotonic LDS counters: the writer's earlier LDS writes are complete before the counter moves (lgkmcnt(0), and a wave's LDS
// operations execute in issue order); the reader polls, then reads the data
__device__ __forceinline__ void ov_signal(int* flag, int v) {
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
__device__ __forceinline__ void ov_wait(int* flag, int v) {
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < v) {
        if (++spins > (1 << 22)) __builtin_trap();        // (a partner that never arrives: fail the launch loudly)
    }
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// band of row i of a tile: tile 0 (A) = bands 8..22 (+ one pad row), tile 1 (B) = bands 0..7, 23..30
template <int T>
__device__ __forceinline__ int ov_band(int i) {
    if constexpr (T == 0) return 8 + (i < 15 ? i : 14);
    else return i < 8 ? i : i + 15;
}

// PROF: cycle probes of workgroup 0, sixteen per wave (clk[16 wave + i]; tools/gpu_phases_bsrnn_ov.py)
#define OV_CLK(i) do { if constexpr (PROF) { if (blockIdx.x == 0 && lane == 0) a.clk[16 * wave + (i)] = __builtin_readcyclecounter(); } } while (0)
template <class S, bool PROF = false>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) bsrnn_ov_kernel(BArgs a) {
    static_assert(S::C == 16 && S::HH == 32 && S::NFFT == 512, "the role-split kernel is built for num_channels = 16");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = BOvLds<S>;
    constexpr int N = S::NFFT, H = S::HOP, OVL = S::OVL, C = S::C, HH = S::HH;
    constexpr int LDX = S::LDX, LDH = S::LDH, LDY = S::LDY;
    constexpr int KSC = S::KSC, KSH = S::KSH, KS1 = S::KS1;      // 4, 8, 12
    constexpr float K2 = -2.8853900817779268f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const float* __restrict__ wp = a.wp;
    const BOffsets& o = a.off;
    WSrc<false> wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, o.total * 4, 0x00020000);
    wb.lane4 = lane * 4;
    wb.li4 = li * 4;
    wb.lds = nullptr;
    wb.base = 0;
    wb.k4d = 0;

    float* sp = smem + L::SP;
    float2* tw = reinterpret_cast<float2*>(smem + L::TW);
    float2* fa = reinterpret_cast<float2*>(smem + L::FA);
    float2* fb = reinterpret_cast<float2*>(smem + L::FB);
    float* XB = smem + L::XB;
    float* XA = smem + L::XA;
    float* Hn = smem + L::HN;
    float* Yf = smem + L::YF;
    float* Hb = smem + L::HB;
    int* flg = reinterpret_cast<int*>(smem + L::FLG);
    float* XPs = smem + L::XP;
    const int b = (int)blockIdx.x;
    float* cst = a.cache_stft + (size_t)b * OVL;
    const size_t lsz = (size_t)kBands * HH;                      // one (h or c) tensor of a layer and stream

    OV_CLK(0);
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = reinterpret_cast<const float2*>(wp + o.twiddle)[i];
    if (tid < 16) flg[tid] = 0;

    // ---------------- the roles' register sets
    // scans (waves 0, 1): lane = (half, unit): half 0 holds gate rows (i, g), half 1 (f, o) of unit u over the whole K
    const int d = wave & 1, u = lane & 31, half = lane >> 5;
    float W0[HH], W1[HH];
    const int lane16 = lane * 16;                                   // bytes: this lane's 16-byte piece of a regrouped fragment set
    auto ld4 = [&](int off_floats) { return wb.at_gv4(off_floats, lane16); };
    auto load_whh = [&](int l) {                                    // BOffsets::ov_hh: [row set][k / 4][lane][4]
#pragma unroll
        for (int q = 0; q < HH / 4; ++q) {
            const f32x4 v0 = ld4(o.ov_hh[l][d] + q * 256), v1 = ld4(o.ov_hh[l][d] + (HH / 4 + q) * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) { W0[4 * q + j] = v0[j]; W1[4 * q + j] = v1[j]; }
        }
    };
    // matrix-core work: hidden tile ct of the time-LSTM gates, direction ct of the projections (a pair of waves = ct 0, 1)
    const int ct = wave & 1;
    float Wt[4][KS1], Wtb[4], Wf1[KSH], Wf1b = 0.0f, Wf2[2 * KSH], Wf2b = 0.0f, Wip[8][KSC], Wipb[8];
    f32x4 hh[2][4];             // the gates' h half (+ bias) per tile
    float cprev[2][4];          // previous cell state of this lane's outputs per tile
    f32x4 xr[2];                // the tiles' band features in accumulator layout
    auto load_proj = [&](int l, auto J0_, auto NJ_) {               // projection column-tile pairs (j, j + 4), j = J0 .. J0 + NJ - 1
        constexpr int J0 = decltype(J0_)::value, NJ = decltype(NJ_)::value;
        static_assert(KSC == 4, "one 16-byte fetch per column tile");
#pragma unroll
        for (int jj = 0; jj < 2 * NJ; ++jj) {
            const int j = J0 + (jj < NJ ? jj : jj - NJ + 4);
            const f32x4 v = ld4(o.ov_ip[l][ct] + j * 256);
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) Wip[j][ks] = v[ks];
        }
        const f32x4 b0 = wb.at_gv4(o.ov_ipb[l][ct], li * 32), b1 = wb.at_gv4(o.ov_ipb[l][ct] + 4, li * 32);
#pragma unroll
        for (int jj = 0; jj < 2 * NJ; ++jj) {
            const int j = J0 + (jj < NJ ? jj : jj - NJ + 4);
            Wipb[j] = j < 4 ? b0[j & 3] : b1[j & 3];
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    auto load_time = [&](int l) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < KS1 / 4; ++q) {
                const f32x4 v = ld4(o.ov_t[l] + ((ct * 4 + g) * (KS1 / 4) + q) * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j) Wt[g][4 * q + j] = v[j];
            }
        {
            const f32x4 v = wb.at_gv4(o.ov_tb[l] + ct * 64, li * 16);
#pragma unroll
            for (int g = 0; g < 4; ++g) Wtb[g] = v[g];
        }
#pragma unroll
        for (int q = 0; q < KSH / 4; ++q) {
            const f32x4 v = ld4(o.ov_f1[l] + q * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) Wf1[4 * q + j] = v[j];
        }
        Wf1b = wb.at16_g(o.tfc_b[l]);
        load_proj(l, I0{}, I4{});
    };
    auto load_ffc = [&](int l) {
#pragma unroll
        for (int q = 0; q < 2 * KSH / 4; ++q) {
            const f32x4 v = ld4(o.ov_f2[l] + q * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) Wf2[4 * q + j] = v[j];
        }
        Wf2b = wb.at16_g(o.ffc_b[l]);
    };
    // the h half of layer l's time-LSTM gates for tile T: state fragments straight from the state tensors (pre_load), the products once
    // the layer's weights are there (pre_mma)
    auto pre_load = [&](int l, auto T_, float (&af)[KSH]) {
        constexpr int T = decltype(T_)::value;
        const float* hg = a.lstm + ((size_t)(2 * l) * a.B + b) * lsz;
        const float* cg = hg + (size_t)a.B * lsz;
        const float* hr = hg + ov_band<T>(li) * HH + lg;
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) af[ks] = hr[4 * ks];
#pragma unroll
        for (int r = 0; r < 4; ++r) cprev[T][r] = cg[ov_band<T>(4 * lg + r) * HH + 16 * ct + li];
    };
    auto pre_mma = [&](auto T_, const float (&af)[KSH]) {
        constexpr int T = decltype(T_)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) hh[T][g] = f32x4{Wtb[g], Wtb[g], Wtb[g], Wtb[g]};
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g) hh[T][g] = FE_MFMA(af[ks], Wt[g][KSC + ks], hh[T][g]);
    };
    auto pre_gates = [&](int l, auto T_) {
        float af[KSH];
        pre_load(l, T_, af);
        pre_mma(T_, af);
    };
    using TA = std::integral_constant<int, 0>;
    using TB = std::integral_constant<int, 1>;
    __syncthreads();

    constexpr int BSI = (kBands * C + kThreads - 1) / kThreads;      // band-split outputs per thread
    float4 bsw[BSI][kBsKP / 4];
    float bsb[BSI];
    float af0[KSH];
    // ============================ STFT + compress (all 257 bins; models/bsrnn/model.py:430-436) ============================
    // The transform runs on the matrix cores (fe::Dft: two chained GEMM stages per wave, one barrier) instead of nine radix-2 passes
    {
        using D = Dft<S, 3>;
        float* xw = smem + L::FA;                  // windowed frame [N]
        float* Xs = smem + L::FB;                  // {Re[N/2], Im[N/2]}, then the Nyquist bin {Re, Im}
        const float* win = wp + o.window;
        const float* xin = a.wav_in + (size_t)b * a.in_stride;
        constexpr int NPT = N / kThreads;
        float fv[NPT], fw[NPT];
#pragma unroll
        for (int q = 0; q < NPT; ++q) { const int n = tid + q * kThreads; fv[q] = (n < OVL) ? cst[n] : xin[n - OVL]; fw[q] = win[n]; }
        typename D::FwdConst dc;
        D::load(dc, wb, o, wave);
        if (wave < 2) pre_load(0, TA{}, af0);      // layer 0's state fragments and the band split's weight rows: independent of the frame
        else pre_load(0, TB{}, af0);
#pragma unroll
        for (int it = 0; it < BSI; ++it) {
            const int i = tid + it * kThreads, ic = i < kBands * C ? i : kBands * C - 1;
            const float4* w4 = reinterpret_cast<const float4*>(wp + o.bs_w) + ic;      // [k/4][band * C + c] float4: coalesced over the threads
#pragma unroll
            for (int k = 0; k < kBsKP / 4; ++k) bsw[it][k] = w4[k * (kBands * C)];
            bsb[it] = wp[o.bs_b + ic];
        }
#pragma unroll
        for (int q = 0; q < NPT; ++q) { const int n = tid + q * kThreads; xw[n] = fv[q] * fw[q]; }
        __syncthreads();                           // (every read of the old cache has landed)
        if (wave == 0) OV_CLK(12);
#pragma unroll
        for (int q = 0; q < NPT; ++q) { const int n = tid + q * kThreads; if (n >= H) cst[n - H] = fv[q]; }      // cache' = frame[H:]
        // layer 0's weights: requested behind the frame, in flight across the transform and the band split
        if (wave < 2) load_whh(0);
        load_time(0);
        D::template forward<true>(xw, Xs, tw, dc, wave, lane, Xs + N);
        if (wave == 0) OV_CLK(13);
        for (int f = tid; f < kBins; f += kThreads) {
            const float re = f < N / 2 ? Xs[f] : Xs[N], im = f < N / 2 ? Xs[N / 2 + f] : Xs[N + 1];
            const float g = pow_f(fmaxf(sqrtf(re * re + im * im), 1.0e-5f), a.compression - 1.0f);
            sp[2 * f] = re * g;
            sp[2 * f + 1] = im * g;
        }
    }
    __syncthreads();
    if (wave == 0) OV_CLK(14);
    // ============================ band split (BandSplit.forward, :136-153; BN folded) ============================
    // thread <-> (band, channel): its zero-padded weight row of kBsKP floats was fetched at the top of the kernel
#pragma unroll
    for (int it = 0; it < BSI; ++it) {
        const int i = tid + it * kThreads;
        if (i < kBands * C) {
            const int bb = i / C;
            const int s0 = bb == 0 ? 0 : (bb <= 10 ? 3 * bb - 1 : (bb <= 22 ? 8 * bb - 56 : 16 * bb - 240));
            const float* s = sp + 2 * s0;
            float a0 = bsb[it], a1 = 0.0f;
#pragma unroll
            for (int k = 0; k < kBsKP / 4; ++k) {
                a0 += bsw[it][k].x * s[4 * k] + bsw[it][k].z * s[4 * k + 2];
                a1 += bsw[it][k].y * s[4 * k + 1] + bsw[it][k].w * s[4 * k + 3];
            }
            XB[bb * LDX + (i - bb * C)] = a0 + a1;
        }
    }
    if (wave == 0) OV_CLK(15);
    // layer 0's time part has nothing to hide under: tile A on waves 0, 1, tile B on waves 2, 3
    if (wave < 2) pre_mma(TA{}, af0);
    else pre_mma(TB{}, af0);
    __syncthreads();
    OV_CLK(1);

    int* const f_prog = flg;            // [2]: 32 l + steps done, per direction
    int* const f_hs = flg + 2;          // [2 pairs][2]: chains passed, per wave of a pair
    int* const f_xa = flg + 6;          // [2]: helper ct has stored tile B's x of layer f_xa (input of the scans' share of the projections)
    int* const f_xpd = flg + 8;         // [2]: layers whose tile-B projections helper ct has completed (its share)
    int* const f_xpa = flg + 10;        // [2]: layers whose tile-A projections helper ct has completed

    // projection column-tile pairs (j, j + 4), j = J0 .. J0 + NJ - 1, of direction ct for tile T: 16-row tiles of gate rows, stored in the
    // scan's lane order [band][half * 32 + unit][slot] - gates (i, g) / (f, o) of a unit are the two slots of one lane (:386-388)
    auto proj = [&](auto T_, auto J0_, auto NJ_, float* xpn) {
        constexpr int T = decltype(T_)::value, J0 = decltype(J0_)::value, NJ = decltype(NJ_)::value;
        const float* xa = XA + ov_band<T>(li) * LDX + lg;
        float af[KSC];
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) af[ks] = xa[4 * ks];
        f32x4 pa[2 * NJ];
#pragma unroll
        for (int jj = 0; jj < 2 * NJ; ++jj) { const float bv = Wipb[J0 + (jj < NJ ? jj : jj - NJ + 4)]; pa[jj] = f32x4{bv, bv, bv, bv}; }
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
            for (int jj = 0; jj < 2 * NJ; ++jj) pa[jj] = FE_MFMA(af[ks], Wip[J0 + (jj < NJ ? jj : jj - NJ + 4)][ks], pa[jj]);
        float* dst = xpn + ((ct * 32) * 64 + li) * 2;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = J0 + jj, g = j >> 1, uh = j & 1;           // column tile j = 2 gate + unit half
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (T == 1 || r < 3 || 4 * lg + 3 < 15)
                    *reinterpret_cast<f32x2*>(dst + (ov_band<T>(4 * lg + r) * 64 + g * 32 + 16 * uh) * 2) = f32x2{pa[jj][r], pa[jj + NJ][r]};
        }
    };

    int nsync = 0;
    // One tile's chain on a pair of waves.  FREQ: starts with fc_freq of the finished layer; TIME: continues with layer l's time part;
    // PSPLIT: the scans take half of the tile's projections (tile B in the steady state: the scans have ended, their waves are free)
    auto chain = [&](auto T_, auto FREQ_, auto TIME_, auto PSPLIT_, auto DOPROJ_, int l, float* xpn, int pair) {
        constexpr int T = decltype(T_)::value;
        constexpr bool FREQ = decltype(FREQ_)::value, TIME = decltype(TIME_)::value, PSPLIT = decltype(PSPLIT_)::value, DOPROJ = decltype(DOPROJ_)::value;
        // (PROF: the inside of tile B's chain under layer 0's scans, wave 2: slots 44-47, 60-63)
        auto cclk = [&](int j) {
            if constexpr (PROF && T == 1 && FREQ && TIME) {
                if (blockIdx.x == 0 && lane == 0 && wave == 2 && l == 1) a.clk[(j < 4 ? 44 : 56) + j] = __builtin_readcyclecounter();
            }
        };
        cclk(0);
        const int arow = ov_band<T>(li);
        int crow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) crow[r] = ov_band<T>(4 * lg + r);
        const bool cok3 = T == 1 || 4 * lg + 3 < 15;                  // (tile A: its last row is a pad row)
        f32x4 x = xr[T];
        if constexpr (FREQ) {
            // fc_freq + residual (:389-390): x += Yf W^T + b
            const float* ya = Yf + arow * LDY + lg;
            f32x4 c0 = x + f32x4{Wf2b, Wf2b, Wf2b, Wf2b}, c1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            float af[2 * KSH];
#pragma unroll
            for (int ks = 0; ks < 2 * KSH; ++ks) af[ks] = ya[4 * ks];
#pragma unroll
            for (int ks = 0; ks < 2 * KSH; ks += 2) { c0 = FE_MFMA(af[ks], Wf2[ks], c0); c1 = FE_MFMA(af[ks + 1], Wf2[ks + 1], c1); }
            x = c0 + c1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < 3 || cok3) XB[crow[r] * LDX + li] = x[r];
        }
        cclk(1);
        if constexpr (!TIME) { xr[T] = x; return; }
        else {
        // time-LSTM gates (LSTMCell over the bands; :371-381): the x half on top of the accumulated h half, gate math in the epilogue
        f32x4 acc[4];
        {
            const float* xa = XB + arow * LDX + lg;
            float af[KSC];
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) af[ks] = xa[4 * ks];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = hh[T][g];
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = FE_MFMA(af[ks], Wt[g][ks], acc[g]);
        }
        cclk(2);
        float hn[4], cn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0][r]));
            const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[1][r]));
            const float gg = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[2][r])) - 1.0f;
            const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[3][r]));
            cn[r] = fg * cprev[T][r] + ig * gg;
            hn[r] = og * (2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(K2 * cn[r])) - 1.0f);
            if (r < 3 || cok3) Hn[crow[r] * LDH + 16 * ct + li] = hn[r];
        }
        cclk(3);
        // rendezvous: both column halves of the new h are in LDS (and the partner has consumed its fragments of the old state)
        ++nsync;
        if (lane == 0) ov_signal(f_hs + 2 * pair + ct, nsync);
        ov_wait(f_hs + 2 * pair + (ct ^ 1), nsync);
        cclk(4);
        {
            // the layer's new (h, c) -> the state tensors (after the rendezvous: the partner's fragments of the old state are consumed)
            float* hg = a.lstm + ((size_t)(2 * l) * a.B + b) * lsz;
            float* cg = hg + (size_t)a.B * lsz;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < 3 || cok3) { hg[crow[r] * HH + 16 * ct + li] = hn[r]; cg[crow[r] * HH + 16 * ct + li] = cn[r]; }
        }
        // fc_time + residual (:382-384): x += h' W^T + b
        {
            const float* ha = Hn + arow * LDH + lg;
            float af[KSH];
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) af[ks] = ha[4 * ks];
            f32x4 c0 = x + f32x4{Wf1b, Wf1b, Wf1b, Wf1b}, c1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < KSH; ks += 2) { c0 = FE_MFMA(af[ks], Wf1[ks], c0); c1 = FE_MFMA(af[ks + 1], Wf1[ks + 1], c1); }
            x = c0 + c1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < 3 || cok3) XA[crow[r] * LDX + li] = x[r];
            xr[T] = x;
        }
        if constexpr (PSPLIT) { if (lane == 0) ov_signal(f_xa + ct, l); }
        cclk(5);
        if constexpr (DOPROJ) {
            if constexpr (PSPLIT) proj(T_, I2{}, I2{}, xpn);
            else proj(T_, I0{}, I4{}, xpn);
        }
        cclk(6);
        cclk(7);
        }
    };

    if (wave < 2) {
        // ================================================ the scans ================================================
        float* hb = Hb + d * (2 * HH);                                   // [2 buffers][HH]
        float* ydump = Hb + 4 * HH + u;                                  // where the low lanes' (unused) h goes: no exec-masked region
        // second row: g in the low half - tanh scaled by K2 = -2 log2 e, so that the cell state is carried as K2 c and tanh(c) is one exp2 + rcp
        // of it with no multiply on the step's chain -, o in the high half
        const float act_m = half == 0 ? 2.0f * K2 : 1.0f, act_a = half == 0 ? -K2 : 0.0f;
        const int band0 = d == 0 ? 0 : kBands - 1;
        const int xd = d == 0 ? 128 : -128, yd = d == 0 ? LDY : -LDY;
        // layer 0, tile A
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[0][r] = XB[ov_band<0>(4 * lg + r) * LDX + li];
        chain(TA{}, std::false_type{}, std::true_type{}, std::false_type{}, std::true_type{}, 0, XPs, 1);
#pragma unroll 1
        for (int l = 0; l < S::NLAY; ++l) {
            ov_wait(f_xpd + d, l + 1);
            if (l == 0) OV_CLK(2);
            if (l == 1) OV_CLK(5);
            if (l == S::NLAY - 1) OV_CLK(7);
            const float* xp = XPs + (l & 1) * L::XPBUF;
            hb[lane] = 0.0f;                                             // h = 0, both buffers
            float cs = 0.0f;
            int xo = ((d * 32 + band0) * 64 + lane) * 2, yo = band0 * LDY + d * HH + u;
            f32x2 xp_next = *reinterpret_cast<const f32x2*>(xp + xo);
            auto steps = [&](int s0, int s1) {
#pragma unroll 1
                for (int s = s0; s < s1; ++s) {
                    const int par = s & 1;
                    const f32x2 xp_cur = xp_next;
                    if (s + 1 < kBands) { xo += xd; xp_next = *reinterpret_cast<const f32x2*>(xp + xo); }
                    const float4* hp4 = reinterpret_cast<const float4*>(hb + par * HH);
                    float4 hq[HH / 4];
#pragma unroll
                    for (int k = 0; k < HH / 4; ++k) hq[k] = hp4[k];
                    f32x2 p0 = {xp_cur.x, 0.0f}, p1 = {xp_cur.y, 0.0f};
#pragma unroll
                    for (int k = 0; k < HH / 4; ++k) {
                        p0 += f32x2{W0[4 * k], W0[4 * k + 1]} * f32x2{hq[k].x, hq[k].y};
                        p1 += f32x2{W1[4 * k], W1[4 * k + 1]} * f32x2{hq[k].x, hq[k].y};
                        p0 += f32x2{W0[4 * k + 2], W0[4 * k + 3]} * f32x2{hq[k].z, hq[k].w};
                        p1 += f32x2{W1[4 * k + 2], W1[4 * k + 3]} * f32x2{hq[k].z, hq[k].w};
                    }
                    const float a0 = p0.x + p0.y, a1 = p1.x + p1.y;
                    const float s0v = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a0));                                   // low: i, high: f
                    const float s1v = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a1)), act_m, act_a);      // low: g, high: o
                    float x = s0v, y = s0v * s1v;                                    // low y: K2 i g
                    const float o2 = 2.0f * s1v, on = -s1v;                          // (off the chain)
                    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));       // x.hi <- y.lo: the high lanes hold K2 i g in x
                    const float cn = __builtin_fmaf(s0v, cs, x);                     // high: K2 (f c + i g)   (low lanes: bounded garbage)
                    cs = cn;
                    const float hn = __builtin_fmaf(o2, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cn)), on);       // high: o (2 r - 1) = o tanh(c)
                    *(half ? hb + (par ^ 1) * HH + u : ydump) = hn;
                    *(half ? Yf + yo : ydump) = hn;
                    yo += yd;
                }
            };
            steps(0, 8);
            if (l > 0) ov_wait(f_xpa + d, l + 1);                       // bands 8..22 come next: tile A's projections (normally long there)
            steps(8, 23);
            if (lane == 0) ov_signal(f_prog + d, 32 * l + 23);
            if (l == 0) OV_CLK(3);
            steps(23, kBands);
            if (lane == 0) ov_signal(f_prog + d, 32 * l + 31);
            if (l == 0) OV_CLK(4);
            if (l == 1) OV_CLK(6);
            if (l == S::NLAY - 1) OV_CLK(8);
            if (l + 1 < S::NLAY) {
                // the helpers are on tile B's chain: fetch the next layer's W_hh and this wave's half of tile B's projections
                // (column-tile pairs 0, 1 of its own direction), computed as soon as helper d has stored the tile's x
                load_whh(l + 1);
                load_proj(l + 1, I0{}, I2{});
                ov_wait(f_xa + d, l + 1);
                proj(TB{}, I0{}, I2{}, XPs + ((l + 1) & 1) * L::XPBUF);
            }
        }
    } else {
        // ================================================ the matrix-core work ================================================
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[1][r] = XB[ov_band<1>(4 * lg + r) * LDX + li];
        chain(TB{}, std::false_type{}, std::true_type{}, std::false_type{}, std::true_type{}, 0, XPs, 0);
        if (lane == 0) ov_signal(f_xpd + ct, 1);
        OV_CLK(2);
#pragma unroll 1
        for (int l = 0; l < S::NLAY; ++l) {
            const bool more = l + 1 < S::NLAY;
            float* xpn = XPs + ((l + 1) & 1) * L::XPBUF;
            // under the early steps of layer l's scans: this layer's fc_freq, the next layer's time part and the h half of its gates
            load_ffc(l);
            if (more) { load_time(l + 1); pre_gates(l + 1, TA{}); pre_gates(l + 1, TB{}); }
            if (l == 0) OV_CLK(3);
            ov_wait(f_prog, 32 * l + 23);
            ov_wait(f_prog + 1, 32 * l + 23);
            if (l == 0) {         // (tile A's x after layer 0's time part was computed by the scan waves)
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[0][r] = XA[ov_band<0>(4 * lg + r) * LDX + li];
            }
            if (l == 0) OV_CLK(4);
            // tile A up to its x after fc_time: its projections are not needed before step 8 of the next layer's scans and wait until
            // tile B's chain - the only serial piece - is through
            if (more) chain(TA{}, std::true_type{}, std::true_type{}, std::false_type{}, std::false_type{}, l + 1, xpn, 0);
            else chain(TA{}, std::true_type{}, std::false_type{}, std::false_type{}, std::false_type{}, l + 1, xpn, 0);
            if (l == 0) OV_CLK(5);
            ov_wait(f_prog, 32 * l + 31);
            ov_wait(f_prog + 1, 32 * l + 31);
            if (l == 0) OV_CLK(6);
            if (l == S::NLAY - 1) OV_CLK(8);
            if (more) chain(TB{}, std::true_type{}, std::true_type{}, std::true_type{}, std::true_type{}, l + 1, xpn, 0);
            else chain(TB{}, std::true_type{}, std::false_type{}, std::false_type{}, std::false_type{}, l + 1, xpn, 0);
            if (more && lane == 0) ov_signal(f_xpd + ct, l + 2);
            if (more) {
                proj(TA{}, I0{}, I4{}, xpn);                  // under steps 0..7 of layer l + 1's scans
                if (lane == 0) ov_signal(f_xpa + ct, l + 2);
            }
            if (l == 0) OV_CLK(7);
            if (l == S::NLAY - 1) OV_CLK(9);
        }
    }
    __syncthreads();
    OV_CLK(10);
    // hand-over to bsrnn_mlp_kernel / the PART 2 launch: band features after the last layer, the compressed spectrum
    {
        float* xg = a.mlp_x + (size_t)b * (kBands * C);
        for (int i = tid; i < kBands * C; i += kThreads) { const int bb = i / C; xg[i] = XB[bb * LDX + (i - bb * C)]; }
        float* sg = a.mlp_sp + (size_t)b * (2 * kBins);
        for (int i = tid; i < 2 * kBins; i += kThreads) sg[i] = sp[i];
    }
    OV_CLK(11);
}
#undef OV_CLK

template <class S, bool PROF = false>
void blaunch_ov(const BArgs& a, int grid, hipStream_t st, hipError_t* err) {
    if constexpr (S::C == 16) {
        auto* fn = &bsrnn_ov_kernel<S, PROF>;
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!attr_set[dev].load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BOvLds<S>::BYTES);
            if (e != hipSuccess) { *err = e; return; }
            attr_set[dev].store(true, std::memory_order_relaxed);
        }
        hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads), BOvLds<S>::BYTES, st, a);
        *err = hipGetLastError();
    } else {
        *err = hipErrorInvalidValue;
    }
}

}  // namespace fe
