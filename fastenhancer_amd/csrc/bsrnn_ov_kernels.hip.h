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
//                fc_freq (layer l) -> x -> time-LSTM gates (layer l + 1) -> fc_time -> input projections of layer l + 1.  Tile A's chain
//                up to its x runs under steps 23..30 of layer l's scans and its projections under steps 0..6 of layer l + 1's (bands
//                8..22 are not fetched before step 7: a step fetches the next step's projections); only tile B's chain is serial with the scans, and the scan waves - idle by then -
//                take half of its projections.  The h half of the time-LSTM gates (W_hh h_{t-1}: known since the last frame) is
//                accumulated under the early steps.  Layer 0's time part runs on all four waves (tile A on waves 0, 1).
// Every product of a chain is computed TRANSPOSED, the sixteen bands of a tile as its N: the accumulator fragment of one product is the B
// operand of the next (see the register sets below) - a chain touches LDS for the scans' outputs, the exchange of the new h between the two
// hidden tiles and the projections it hands over.
// Synchronisation is by monotonic counters in LDS, polled: scan progress per direction, a two-wave rendezvous inside a chain, "tile B's x
// stored", "projections of layer l complete" per tile and direction; the projections are double-buffered by layer parity.  The two waves
// of a pair split the gate GEMM by hidden tile and the projections by direction, and both compute the narrow fc layers (N = 16) - IN THE
// SAME SUMMATION ORDER: either wave's copy of x is read by other waves, so the two must agree bit for bit (else a run is not reproducible).
// Reference: models/bsrnn/model.py:367-390 (the layer loop), :249-257 (ONNXLSTM), :136-153 (BandSplit).
#pragma once

namespace fe {

template <class S>
struct BOvLds {
    static constexpr int SP = 0;                              // compressed spectrum [257][2]
    static constexpr int TW = SP + 2 * kBins + 2;             // twiddles
    static constexpr int FA = TW + S::NFFT;                   // windowed frame [N] (the radix-2 kernels' ping-pong size is kept: 2 N floats each)
    static constexpr int FB = FA + 2 * S::NFFT;
    static constexpr int XB = FB + 2 * S::NFFT;               // (FB: spectrum {Re[N/2], Im[N/2]} + the Nyquist bin)   [32][LDX] band features: the band split's output, the last layer's
    static constexpr int XT = (XB + 32 * S::LDX + 3) / 4 * 4; // [2 tiles][64 lanes][4]: a tile's x after fc_time as accumulator fragments (for the waves that did not compute it)
    static constexpr int HX = XT + 2 * 256;                   // [2 pairs][2 chain parities][2 ct][64 lanes][4]: the new time-LSTM h of a wave's hidden tile, for its partner
    static constexpr int YF = HX + 2 * 2 * 2 * 256;           // [32][LDY] band-LSTM outputs (fwd | bwd)
    static constexpr int HB = (YF + 32 * S::LDY + 3) / 4 * 4; // [2 dirs][2 buffers][HH] + [HH] dump slots of the low lanes
    static constexpr int FLG = HB + 5 * S::HH;                // counters (ints): scan progress fwd / bwd, helper rendezvous, projections ready
    static constexpr int XP = FLG + 16;                       // [2 layer parities][2 dirs][32 bands][XPLD]: gate rows in the scan's lane order [64 lanes][2]
    static constexpr int XPLD = 132;                          // (+ 4: the chains store a band's row from sixteen lanes = sixteen bands at once)
    static constexpr int XPBUF = 2 * 32 * XPLD;
    static constexpr int TOTAL = XP + 2 * XPBUF;
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
    static_assert(XP % 4 == 0 && HB % 4 == 0 && XT % 4 == 0 && XPLD % 4 == 0, "aligned f32x2 / float4 LDS accesses");
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
        // a partner that never arrives fails the launch instead of hanging the device - but only after ~2^30 polls (minutes): a debugger, a
        // page-fault / XNACK stall on first touch of the weights or a preempted queue must not trip it (2^22 polls were 0.1-0.2 s)
        if (++spins > (1 << 30)) __builtin_trap();
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
// Arrival at / wait for a barrier over the (up to) sixteen workgroups of one sixteen-stream tile of a FUSED launch.  A monotonic counter in global
// memory per tile and barrier: every barrier instance advances it by exactly 16 (the first workgroup of a short last tile arrives for the absent
// ones), so the instance a workgroup takes part in is old / 16 and it is over at (old / 16 + 1) * 16 - no reset, no generation word, no epoch in the
// kernel arguments (a captured HIP graph replays the same arguments), unsigned wrap-around included.  Data handed across the barrier goes through
// agent-scope stores drained with s_waitcnt vmcnt(0) before the arrival, and agent-scope loads after it (the protocol of the time-pipelined
// kernels: no release / acquire fences, which write back / invalidate whole caches on this part).  All workgroups of the launch are
// co-resident: one per CU, at most #CUs of them, launched cooperatively.
__device__ __forceinline__ void ov_group_barrier(unsigned int* cnt, unsigned int arrive) {
    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): this thread's stores have left the CU
    __syncthreads();
#ifdef FE_EXP_FUS_NOBAR      // (timing experiment: no waiting for the other workgroups)
    return;
#endif
    if (threadIdx.x == 0) {
        const unsigned int old = __hip_atomic_fetch_add(cnt, arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int target = (old / 16u + 1u) * 16u;
        unsigned int spins = 0;
        while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 28)) __builtin_trap();        // (a workgroup of the tile that never arrives: fail the launch loudly)
        }
    }
    __syncthreads();
}

// FUSED (r6): the whole per-hop step in ONE launch - after the layers the workgroups of a sixteen-stream tile meet at a barrier, run the mask
// decoder's two MLP layers for their sixteen streams on the matrix cores (one (kind, band) job per WAVE: 62 jobs + the widest band split in two
// = the tile's 64 waves), meet again, and every workgroup finishes its own stream (GLU, mask, un-compress, inverse transform, overlap-add).
// The three-launch step pays ~4.8 us per kernel boundary (an empty bsrnn_mlp_kernel: profiles/r3zz_*) twice in a 77-us step - but a barrier
// over sixteen workgroups on eight XCDs measured 5.9 us, and a cooperative launch ~21 us more than a plain one: NEGATIVE (81 / 102 us),
// profiles/r6_bsrnn_fused_step.txt.  Kept behind fe_set_option("bsrnn_fused_step", 1), bit-identical to the three launches.
template <class S, bool PROF = false, bool FUSED = false>
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
    float* XT = smem + L::XT;
    float* HX = smem + L::HX;
    float* Yf = smem + L::YF;
    float* Hb = smem + L::HB;
    int* flg = reinterpret_cast<int*>(smem + L::FLG);
    float* XPs = smem + L::XP;
    const int b = (int)blockIdx.x;
    float* cst = a.cache_stft + (size_t)b * OVL;
    const size_t lsz = (size_t)kBands * HH;                      // one (h or c) tensor of a layer and stream

    OV_CLK(0);
    // (the twiddles are requested here and parked in LDS together with the windowed frame: ONE cold round trip to memory for both)
    static_assert(N / 2 == kThreads, "one twiddle per thread");
    const float2 twv = reinterpret_cast<const float2*>(wp + o.twiddle)[tid];
    if (tid < 16) flg[tid] = 0;
    if (tid < 2) sp[2 * kBins + tid] = 0.0f;      // the band split reads the last band's row zero-padded to kBsKP floats: two words past the spectrum meet zero weights - they must not be NaN / inf leftovers

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
    // matrix-core work: hidden tile ct of the time-LSTM gates, direction ct of the projections (a pair of waves = ct 0, 1).
    // Every product is computed TRANSPOSED, the sixteen bands of a tile as its N: out^T [rows x bands] = W [rows x K] . in^T [K x bands].  The
    // accumulator fragment of one product - lane (li = band, lg) holds rows 4 lg + r - IS the B operand of the next one, k-step r carrying row
    // 4 lg + r; the A operand then is the weight matrix row-major, W[row li][4 lg + r] (BOffsets::ov_tx / ov_f1t / ov_ipt).  A tile's chain
    // touches LDS for the scans' outputs, the exchange of the new h between the two hidden tiles, and the projections it hands to the scans.
    const int ct = wave & 1;
    auto g4 = [&](int off_floats) { return *reinterpret_cast<const f32x4*>(wp + off_floats); };
    float Wf2[2 * KSH];                          // fc_freq [C x 2 HH]: A fragments (k = 4 ks + lg over the scans' outputs, read from LDS)
    f32x4 Wf2b, Wf1b;                            // biases of this lane's rows 4 lg + r
    f32x4 Wtx[4], Wtb[4];                        // time LSTM, x rows of gate g: W[16 ct + li][4 lg + r]; bias of units 16 ct + 4 lg + r
    float Wth[4][KSH];                           // ... h rows: A fragments (k = 4 ks + lg over the state, read from the state tensor)
    f32x4 Wf1lo, Wf1hi;                          // fc_time [C x HH]: W[li][4 lg + r] (units 0..15), W[li][16 + 4 lg + r] (units 16..31)
    f32x4 Wip[8], Wipb[8];                       // projections of direction ct, column tile j: W[16 j + li][4 lg + r], bias of rows 16 j + 4 lg + r
    f32x4 hh[2][4];                              // the gates' h half (+ bias) per tile
    f32x4 cprev[2];                              // previous cell state of this lane's units per tile
    f32x4 xr[2];                                 // the tiles' band features x^T: channels 4 lg + r of band li
    auto load_proj = [&](int l, auto J0_, auto NJ_) {               // projection column-tile pairs (j, j + 4), j = J0 .. J0 + NJ - 1
        constexpr int J0 = decltype(J0_)::value, NJ = decltype(NJ_)::value;
#pragma unroll
        for (int jj = 0; jj < 2 * NJ; ++jj) {
            const int j = J0 + (jj < NJ ? jj : jj - NJ + 4);
            Wip[j] = g4(o.ov_ipt[l][ct] + (16 * j + li) * C + 4 * lg);
            Wipb[j] = g4(o.f_b[l][ct] + 16 * j + 4 * lg);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    auto load_time = [&](int l) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            Wtx[g] = g4(o.ov_tx[l] + ((ct * 4 + g) * 16 + li) * C + 4 * lg);
            Wtb[g] = g4(o.t_b[l] + g * HH + 16 * ct + 4 * lg);
#pragma unroll
            for (int q = 0; q < KSH / 4; ++q) {
                const f32x4 v = ld4(o.ov_t[l] + ((ct * 4 + g) * (KS1 / 4) + KSC / 4 + q) * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j) Wth[g][4 * q + j] = v[j];
            }
        }
        Wf1lo = g4(o.ov_f1t[l] + li * HH + 4 * lg);
        Wf1hi = g4(o.ov_f1t[l] + li * HH + 16 + 4 * lg);
        Wf1b = g4(o.tfc_b[l] + 4 * lg);
        load_proj(l, I0{}, I4{});
    };
    auto load_ffc = [&](int l) {
#pragma unroll
        for (int q = 0; q < 2 * KSH / 4; ++q) {
            const f32x4 v = ld4(o.ov_f2[l] + q * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) Wf2[4 * q + j] = v[j];
        }
        Wf2b = g4(o.ffc_b[l] + 4 * lg);
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
        cprev[T] = *reinterpret_cast<const f32x4*>(cg + ov_band<T>(li) * HH + 16 * ct + 4 * lg);
    };
    auto pre_mma = [&](auto T_, const float (&af)[KSH]) {
        constexpr int T = decltype(T_)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) hh[T][g] = Wtb[g];
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g) hh[T][g] = FE_MFMA(Wth[g][ks], af[ks], hh[T][g]);
    };
    auto pre_gates = [&](int l, auto T_) {
        float af[KSH];
        pre_load(l, T_, af);
        pre_mma(T_, af);
    };
    using TA = std::integral_constant<int, 0>;
    using TB = std::integral_constant<int, 1>;

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
        tw[tid] = twv;
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

    // projection column-tile pairs (j, j + 4), j = J0 .. J0 + NJ - 1, of direction ct for tile T from its x^T fragment, stored in the scan's
    // lane order [band][half * 32 + unit][slot] - gates (i, g) / (f, o) of a unit are the two slots of one lane (:386-388): a lane holds units
    // 4 lg + r of column tile j = 2 gate + unit half, i.e. eight consecutive floats of its band's row per pair
    auto proj = [&](auto T_, auto J0_, auto NJ_, float* xpn, const f32x4& x) {
        constexpr int T = decltype(T_)::value, J0 = decltype(J0_)::value, NJ = decltype(NJ_)::value;
        f32x4 pa[2 * NJ];
#pragma unroll
        for (int jj = 0; jj < 2 * NJ; ++jj) pa[jj] = Wipb[J0 + (jj < NJ ? jj : jj - NJ + 4)];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int jj = 0; jj < 2 * NJ; ++jj) pa[jj] = FE_MFMA(Wip[J0 + (jj < NJ ? jj : jj - NJ + 4)][r], x[r], pa[jj]);
        if (T == 1 || li < 15) {
            float* dst = xpn + (ct * 32 + ov_band<T>(li)) * L::XPLD + 8 * lg;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = J0 + jj, g = j >> 1, uh = j & 1;
                *reinterpret_cast<f32x4*>(dst + (g * 32 + 16 * uh) * 2) = f32x4{pa[jj][0], pa[jj + NJ][0], pa[jj][1], pa[jj + NJ][1]};
                *reinterpret_cast<f32x4*>(dst + (g * 32 + 16 * uh) * 2 + 4) = f32x4{pa[jj][2], pa[jj + NJ][2], pa[jj][3], pa[jj + NJ][3]};
            }
        }
    };

    int nsync = 0;
    // One tile's chain on a pair of waves.  FREQ: starts with fc_freq of the finished layer; TIME: continues with layer l's time part;
    // PSPLIT: the scans take half of the tile's projections (tile B in the steady state: the scans have ended, their waves are free);
    // DOPROJ: the projections follow at once (tile A's wait until tile B's chain is through)
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
        const int band = ov_band<T>(li);
        const bool valid = T == 1 || li < 15;                             // (tile A: its last column is a pad band)
        f32x4 x = xr[T];
        if constexpr (FREQ) {
            // fc_freq + residual (:389-390): x^T += W Yf^T + b
            const float* ya = Yf + band * LDY + lg;
            float yf[2 * KSH];
#pragma unroll
            for (int ks = 0; ks < 2 * KSH; ++ks) yf[ks] = ya[4 * ks];
            // (r6: four accumulator chains of four instead of two of eight - a lone wave's dependent 16x16x4 chain runs at ~64 cycles per MFMA, and
            // tile B's chain is serial with the scans; the order is the same on both waves of a pair, whose x must agree bit for bit)
            f32x4 c0 = x + Wf2b, c1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, c2 = c1, c3 = c1;
#pragma unroll
            for (int ks = 0; ks < 2 * KSH; ks += 4) {
                c0 = FE_MFMA(Wf2[ks], yf[ks], c0); c1 = FE_MFMA(Wf2[ks + 1], yf[ks + 1], c1);
                c2 = FE_MFMA(Wf2[ks + 2], yf[ks + 2], c2); c3 = FE_MFMA(Wf2[ks + 3], yf[ks + 3], c3);
            }
            x = (c0 + c1) + (c2 + c3);
        }
        cclk(1);
        if constexpr (!TIME) {
            xr[T] = x;
            if (valid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) XB[band * LDX + 4 * lg + r] = x[r];      // the last layer's features, for the hand-over
            }
            return;
        }
        else {
        // time-LSTM gates (LSTMCell over the bands; :371-381): the x half on top of the accumulated h half, gate math in the epilogue
        f32x4 acc[4], acc2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { acc[g] = hh[T][g]; acc2[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
        for (int r = 0; r < 4; r += 2)
#pragma unroll
            for (int g = 0; g < 4; ++g) { acc[g] = FE_MFMA(Wtx[g][r], x[r], acc[g]); acc2[g] = FE_MFMA(Wtx[g][r + 1], x[r + 1], acc2[g]); }
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] += acc2[g];
        cclk(2);
        f32x4 hn, cn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0][r]));
            const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[1][r]));
            const float gg = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[2][r])) - 1.0f;
            const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[3][r]));
            cn[r] = fg * cprev[T][r] + ig * gg;
            hn[r] = og * (2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(K2 * cn[r])) - 1.0f);
        }
        // the two hidden tiles meet in fc_time: this wave's new h goes to its partner as a fragment (the same lane reads it back), double-
        // buffered by chain parity; the rendezvous also says that the partner has consumed its fragments of the old state
        ++nsync;
        float* hx = HX + ((pair * 2 + (nsync & 1)) * 2) * 256 + lane * 4;
        *reinterpret_cast<f32x4*>(hx + ct * 256) = hn;
        cclk(3);
        if (lane == 0) ov_signal(f_hs + 2 * pair + ct, nsync);
        ov_wait(f_hs + 2 * pair + (ct ^ 1), nsync);
        const f32x4 hp = *reinterpret_cast<const f32x4*>(hx + (ct ^ 1) * 256);
        cclk(4);
        if (valid) {
            // the layer's new (h, c) -> the state tensors: units 16 ct + 4 lg + r of band li, 16 bytes each
            float* hg = a.lstm + ((size_t)(2 * l) * a.B + b) * lsz + band * HH + 16 * ct + 4 * lg;
            *reinterpret_cast<f32x4*>(hg) = hn;
            *reinterpret_cast<f32x4*>(hg + (size_t)a.B * lsz) = cn;
        }
        // fc_time + residual (:382-384): x^T += W h'^T + b.  Both waves of the pair compute it, in the SAME order (units 0..15 into c0,
        // 16..31 into c1, whoever produced them): their x must agree bit for bit - either one's copy is read by other waves
        {
            const f32x4 hlo = ct ? hp : hn, hhi = ct ? hn : hp;
            f32x4 c0 = x + Wf1b, c1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, c2 = c1, c3 = c1;
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                c0 = FE_MFMA(Wf1lo[r], hlo[r], c0); c1 = FE_MFMA(Wf1hi[r], hhi[r], c1);
                c2 = FE_MFMA(Wf1lo[r + 1], hlo[r + 1], c2); c3 = FE_MFMA(Wf1hi[r + 1], hhi[r + 1], c3);
            }
            x = (c0 + c2) + (c1 + c3);
            xr[T] = x;
        }
        *reinterpret_cast<f32x4*>(XT + (T * 64 + lane) * 4) = x;          // (for the waves that take part of this tile's work without having computed x)
        if constexpr (PSPLIT) { if (lane == 0) ov_signal(f_xa + ct, l); }
        cclk(5);
        if constexpr (DOPROJ) {
            if constexpr (PSPLIT) proj(T_, I2{}, I2{}, xpn, x);
            else proj(T_, I0{}, I4{}, xpn, x);
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
        const int xd = d == 0 ? L::XPLD : -L::XPLD, yd = d == 0 ? LDY : -LDY;
        // layer 0, tile A
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[0][r] = XB[ov_band<0>(li) * LDX + 4 * lg + r];
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
            int xo = (d * 32 + band0) * L::XPLD + lane * 2, yo = band0 * LDY + d * HH + u;
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
            // bands 8..22 belong to tile A, whose projections have a ready counter of their own.  A step FETCHES the next step's projections
            // (xp_next), so the wait sits before step 7 - the first one that touches a tile A row - not before step 8 (with the wait
            // after step 7 its fetch of band 8 / 22 was ordered before the wait and could read the buffer's contents of two layers ago)
            steps(0, 7);
            if (l > 0) ov_wait(f_xpa + d, l + 1);                       // (normally long there: the helpers compute them under steps 0..6)
            steps(7, 23);
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
                proj(TB{}, I0{}, I2{}, XPs + ((l + 1) & 1) * L::XPBUF, *reinterpret_cast<const f32x4*>(XT + (64 + lane) * 4));
            }
        }
    } else {
        // ================================================ the matrix-core work ================================================
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[1][r] = XB[ov_band<1>(li) * LDX + 4 * lg + r];
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
            if (l == 0) xr[0] = *reinterpret_cast<const f32x4*>(XT + lane * 4);      // (tile A's x after layer 0's time part was computed by the scan waves)
            if (l == 0) OV_CLK(4);
            // tile A up to its x after fc_time: its projections are not fetched before step 7 of the next layer's scans and wait until
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
                proj(TA{}, I0{}, I4{}, xpn, xr[0]);            // under steps 0..6 of layer l + 1's scans
                if (lane == 0) ov_signal(f_xpa + ct, l + 2);
            }
            if (l == 0) OV_CLK(7);
            if (l == S::NLAY - 1) OV_CLK(9);
        }
    }
    __syncthreads();
    OV_CLK(10);
    if constexpr (!FUSED) {
        // hand-over to bsrnn_mlp_kernel / the PART 2 launch: band features after the last layer, the compressed spectrum
        float* xg = a.mlp_x + (size_t)b * (kBands * C);
        for (int i = tid; i < kBands * C; i += kThreads) { const int bb = i / C; xg[i] = XB[bb * LDX + (i - bb * C)]; }
        float* sg = a.mlp_sp + (size_t)b * (2 * kBins);
        for (int i = tid; i < 2 * kBins; i += kThreads) sg[i] = sp[i];
        OV_CLK(11);
    } else {
        // ============================ mask decoder for the sixteen-stream tile (MaskDecoder.forward, :225-246) ============================
        const int tile = b >> 4, r16 = b & 15;
        const int members = a.B - 16 * tile < 16 ? a.B - 16 * tile : 16;       // workgroups of this tile (a short last tile)
        unsigned int* gcnt = a.gsync + 2 * tile;
        const unsigned int arrive = r16 == 0 ? (unsigned int)(16 - members + 1) : 1u;
        {
            float* xg = a.mlp_x + (size_t)b * (kBands * C);
            for (int i = tid; i < kBands * C; i += kThreads) {
                const int bb = i / C;
                __hip_atomic_store(xg + i, XB[bb * LDX + (i - bb * C)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        ov_group_barrier(gcnt, arrive);
        // the tail's constants, requested now: the inverse transform's operands, the synthesis window, the old overlap tail, the bins' row tables
        using DI = Dft<S, 3>;
        typename DI::InvConst idc;
        DI::load(idc, wb, o, wave);
        constexpr int OPT = (N + kThreads - 1) / kThreads, FPT = (kBins + kThreads - 1) / kThreads;
        float* cis = a.cache_istft + (size_t)b * OVL;
        float ocis[OPT], owin[OPT];
        int bra[FPT], brg[FPT];
#pragma unroll
        for (int q = 0; q < OPT; ++q) {
            const int n = tid + q * kThreads;
            ocis[q] = n < OVL ? cis[n] : 0.0f;
            owin[q] = wp[o.window_istft + (n < N ? n : N - 1)];
        }
        {
            const int* bin_row = reinterpret_cast<const int*>(wp + o.bin_row);
            const int* bin_2sub = reinterpret_cast<const int*>(wp + o.bin_2sub);
#pragma unroll
            for (int q = 0; q < FPT; ++q) {
                const int f = tid + q * kThreads < kBins ? tid + q * kThreads : kBins - 1;
                bra[q] = bin_row[f];
                brg[q] = bin_2sub[f];
            }
        }
        // unit u of the tile's 64: (kind, band) jobs in cost order - bands 30 (five layer-2 items, two waves: items 0-2 / 3-4), 29 .. 23 (four),
        // 22 .. 11 (two), 10 .. 0 (one) - dealt round-robin over the workgroups (unit u on wave u / 16 of workgroup u % 16); a workgroup of a
        // short tile also takes the units of the absent ones
        {
            float* h1 = smem + L::XP + wave * (16 * BMlpLds<S>::LDH);          // (the projection buffers are dead)
            static_assert(kWaves * 16 * BMlpLds<S>::LDH + 2 * kMlpRows <= 2 * L::XPBUF, "MLP hidden tiles + pre-activations alias the projection buffers");
#pragma unroll 1
            for (int vr = r16; vr < 16; vr += members) {
                const int u = 16 * wave + vr;
                int kind, band, it0, nitw;
                if (u < 4) { kind = u >> 1; band = 30; it0 = (u & 1) ? 3 : 0; nitw = (u & 1) ? 2 : 3; }
                else {
                    const int j = u - 4;                       // 60 single-wave jobs: bands 29 .. 0 of kind j & 1
                    kind = j & 1; band = 29 - (j >> 1); it0 = 0;
                    nitw = (4 * bsrnn_band_sub(band) + 15) >> 4;
                }
#ifndef FE_EXP_FUS_NOMLP      // (timing experiment: the fused step without its mask decoder)
                bsrnn_mlp_wave<S, true>(a, h1, kind, band, tile, 1, it0, 1, nitw, lane);
#endif
            }
        }
        ov_group_barrier(gcnt + 1, arrive);
        // ============================ GLU, mask, un-compress (:393-401), iSTFT (functional/audio_modules.py:259-303) ============================
        float* PRE = smem + L::XP + kWaves * 16 * BMlpLds<S>::LDH;
        {
            const float* pg = a.mlp_pre + (size_t)b * (2 * kMlpRows);
            for (int i = tid; i < 2 * kMlpRows; i += kThreads) PRE[i] = __hip_atomic_load(pg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        float* Ys = smem + L::FA;                  // {Re[N/2], Im[N/2]}, then Re of the Nyquist bin
#pragma unroll
        for (int q = 0; q < FPT; ++q) {
            const int f = tid + q * kThreads;
            if (f >= kBins) break;
            const int ra = bra[q], rg = ra + brg[q];
            float mr[4];
#pragma unroll
            for (int kind = 0; kind < 2; ++kind)
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
                    mr[kind * 2 + ri] = PRE[kind * kMlpRows + ra + ri] * sigmoid_f(PRE[kind * kMlpRows + rg + ri]);
            const float xr_ = sp[2 * f], xi_ = sp[2 * f + 1];
            float yr = xr_ * mr[0] - xi_ * mr[1] + mr[2];
            float yi = xr_ * mr[1] + xi_ * mr[0] + mr[3];
            const float g = pow_f(sqrtf(yr * yr + yi * yi), 1.0f / a.compression - 1.0f);
            yr *= g;
            yi *= g;
            if (f < N / 2) { Ys[f] = yr; Ys[N / 2 + f] = yi; }
            else Ys[N] = yr;
        }
        __syncthreads();
        {
            // irfft keeps Re X[N/2] only and ignores Im X[0]: the transform covers bins 0 .. N/2 - 1, the Nyquist bin is (-1)^n X[N/2] / N
            float* P0 = smem + L::FB;
            float* P1 = P0 + N;
            DI::template inverse<WSrc<false>, true>(Ys, P0, P1, tw, idc, wb, o, wave, lane);
            const float nyq = Ys[N] * (1.0f / (float)N);
            float* out = a.wav_out + (size_t)b * a.out_stride;
            float vo[OPT];
#pragma unroll
            for (int q = 0; q < OPT; ++q) {
                const int n = tid + q * kThreads, nc = n < N ? n : N - 1;
                const int pi = DI::pidx(nc & (DI::N1 - 1), nc / DI::N1);
                vo[q] = (P0[pi] + P1[pi] + ((nc & 1) ? -nyq : nyq)) * owin[q] + ocis[q];
            }
#pragma unroll
            for (int q = 0; q < OPT; ++q) {
                const int n = tid + q * kThreads;
                if (n < H) out[n] = vo[q];
                else if (n < N) cis[n - H] = vo[q];
            }
        }
    }
}
#undef OV_CLK

template <class S, bool PROF = false, bool FUSED = false>
void blaunch_ov(const BArgs& a, int grid, hipStream_t st, hipError_t* err) {
    if constexpr (S::C == 16) {
        auto* fn = &bsrnn_ov_kernel<S, PROF, FUSED>;
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!attr_set[dev].load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BOvLds<S>::BYTES);
            if (e != hipSuccess) { *err = e; return; }
            attr_set[dev].store(true, std::memory_order_relaxed);
        }
        if constexpr (FUSED) {
            // the workgroups of a sixteen-stream tile wait for each other: a cooperative launch (co-residency guaranteed, or refused)
            static const bool plain = [] { const char* e = getenv("FE_EXP_FUSED_PLAIN"); return e && e[0] == '1'; }();      // (timing experiment)
            if (plain) {
                hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads), BOvLds<S>::BYTES, st, a);
                *err = hipGetLastError();
                note_kernel("bsrnn_ov_kernel<fused step, plain launch>");
                return;
            }
            BArgs args = a;
            void* kargs[] = {&args};
            *err = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(fn), dim3(grid), dim3(kThreads), kargs, (unsigned int)BOvLds<S>::BYTES, st);
            if (*err == hipSuccess) note_kernel("bsrnn_ov_kernel<fused step>");
            else (void)hipGetLastError();
            return;
        }
        note_kernel(PROF ? "bsrnn_ov_kernel<profile>" : "bsrnn_ov_kernel");
        hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads), BOvLds<S>::BYTES, st, a);
        *err = hipGetLastError();
    } else {
        *err = hipErrorInvalidValue;
    }
}

}  // namespace fe
