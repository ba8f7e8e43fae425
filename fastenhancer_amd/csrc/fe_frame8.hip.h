// fe_frame8.hip.h — the per-hop streaming step (scripts/export_onnx.py:48-58, one hop per launch) on a 512-THREAD workgroup:
// eight wavefronts per stream, TWO per SIMD of the CU.
//
// fe_frame_kernel (fe_kernels.hip.h) runs a stream on four waves, one per SIMD: a frame is a chain of ~38 barrier-bounded
// phases, and with a lone wave per SIMD nothing hides a wave's own latencies - the pipeline fill after each barrier, the LDS
// write -> read turn-around, the dependent transcendental chains of the epilogues (52 % MFMA-busy, 42 % of the wave cycles
// waiting, profiles/r3k_*).  Here every phase's tiles are split once more so that each SIMD holds two waves (w and w + 4)
// that issue into each other's stalls:
//   conv-type GEMMs  [F1 x C1] (M = 4 row tiles, N = NTC channel tiles): wave (ws, wh) = row tile ws, channel tiles
//                    [0, NTA) for wh = 0 and [NTA, NTC) for wh = 1 - unequal halves (2 : 1 for FastEnhancer_B) on purpose:
//                    the lighter wave reaches its epilogue while the heavier one still feeds the matrix pipe
//   token GEMMs      [F2P x C2] (M = 2 row tiles): wave (ws, wh) = row tile wh x the column tiles ws, ws + 4 - the
//                    register-resident weight fragments are fetched by both waves of a pair (L1 / L2 hits)
//   attention        wave (ws, wh) = head ws, query tile wh
//   element-wise     512 threads
//   DFT / iDFT       the four-wave transform of Dft<S> on waves 0-3 (one per SIMD), the other four run the
//                    element-wise work next to it (cache shift, constant prefetch)
// Same LDS plan (Lds<S>), same packed weights (Pack<S>), same state and the same arithmetic per output element as
// fe_frame_kernel: the two kernels agree to fp32 rounding (the order of the additions inside a GEMM is the same; tests
// compare both with the oracle and with each other).
// Reference: models/fastenhancer/default/model.py:266-290 (RNNFormer block), 620-675 (model_forward), 677-710 (ONNXModel.forward),
// functional/audio_modules.py:243-303 (ONNXSTFT).
#pragma once
#include "fe_kernels.hip.h"

namespace fe {

// FE_WG8_HPRE=1 (r5; MEASURED NEGATIVE, off by default - profiles/r5_headline_hpre.txt): the GRU's hidden halves
// W_hh h_{t-1} of ALL blocks - h_{t-1} is known when the frame starts - accumulated by waves 4-7 in the front of the frame (next to the DFT of
// waves 0-3 and in enc_pre), the GRU jobs moved to the waves that hold the sums (wave 4, 5: channel group 0 + the mixed tile of a row tile;
// wave 6, 7: channel group 1).  The products removed altogether (wrong results) are worth 31.64 -> 30.09 us; moved, the frame is 33.5 us: the
// front grows by 7.4 k cycles (operand fetches of three blocks: +4.9 k before the first phase, DFT phase +1.3 k, enc_pre +1.2 k) and a
// block's GRU phase shrinks by 0.67 k instead of 1.15 k - its GEMM, now on ONE wave per SIMD, runs at 57 cycles per MFMA (LDS operand
// latency that the second wave of a SIMD used to cover).
// (FE_WG8_HPRE defaults to 0 in fe_kernels.hip.h: the packed section u8_gh4 exists in such builds only)
#ifndef FE_WG8_MIXSPLIT
#define FE_WG8_MIXSPLIT 1
#endif
// r6: dec_post's 1x1 conv and the transposed conv behind it in ONE phase (one stream per workgroup).  The transposed conv of row tile ws reads the 1x1's
// outputs of that row tile only - the tiles of waves (ws, 0) and (ws, 1), the two waves of SIMD ws - so the pair hands over through an LDS counter instead of
// a workgroup barrier, and wave (ws, 0) runs the 12 MFMAs with the weight fragments it fetched into registers at the top of the phase (the phase of its own
// cost ~1.4 k cycles for 0.4 k of matrix-pipe time).  Same chain, same summation order: bit-identical.  MEASURED NEGATIVE (30.86 -> 30.96 us, same box, parity
// green): the 12 dependent MFMAs now trail the HEAVY wave of each SIMD alone (a lone wave's dependent 16x16x4 chain runs at ~64 cycles per MFMA) - off by default.
#ifndef FE_WG8_FUSEPOST
#define FE_WG8_FUSEPOST 0
#endif
constexpr int kThreads8 = 512;
constexpr int kWaves8 = 8;

template <class S>
struct Wg8 {
    using L = Lds<S>;
    static constexpr bool MDFT = (S::NFFT == 512) || (S::C1 < 128);
    // built for the B-type plan: staged conv weights, LDS-resident skips, register-resident block weights with flat GRU gates
    static constexpr bool OK0 = L::STAGED && L::SKIPS_LDS && S::GFLAT && S::KT == 1 && S::LOW == 0 && !S::FRNN && !S::TATT && !S::LN && !S::BIDIR &&
                               S::G8P && S::MTC == 4 && S::MT2 == 2 && MDFT && !L::PERHEAD && S::NFFT == kThreads8 && S::NTPW2 == 1;
    static constexpr int NTA = (S::NTC + 1) / 2, NTB = S::NTC - NTA;        // conv channel tiles of the wh = 0 / wh = 1 wave
    static constexpr int N2A = (S::NT2 + 1) / 2, N2B = S::NT2 - N2A;        // rf_post's token-channel tiles likewise
    static constexpr int NPW = ceil_div(ceil_div(Pack<S>::umax(), 256), kWaves8);
    static constexpr int HPT = ceil_div(S::F2 * S::C2, kThreads8);          // hidden-state elements per thread
    // Weight staging region: FOUR slots of U8_SLOT floats.  The conv units use them as the two buffers of fe_frame_kernel (WB0 = slots
    // 0-1, WB1 = slots 2-3); during the RNNFormer blocks - where fe_frame_kernel's staging buffers sit idle and its waves fetch their
    // private weight fragments from L2 - the BLOCK weights go through the four slots as well: eight waves would fetch every fragment
    // twice (the two waves of a SIMD work on the same columns), 240 wave-level loads per block at ~16 cycles of the CU's vector-memory
    // path each (measured: 3.5 us of the 32.7 us frame, profiles/r4a_wg8_steps.txt); staged, a block is 64 one-KiB pieces.
    //   slot 1: GRU input weights   slot 3: GRU hidden weights   slot 0: rnn_fc, then attn_fc   slot 2: qkv
    static constexpr int SLOT = S::U8_SLOT;
    static constexpr int WB0 = L::NOSTAGE_TOTAL, WB1 = WB0 + 2 * SLOT;
    // The new GRU states of the KB blocks wait in LDS for the end of the frame and leave as whole lines (HST, when it fits): stored from the GRU
    // epilogues - 64-byte pieces of [F2][C2] rows - their acknowledgements were charged to the next vmcnt wait of the weight staging
    // (vmcnt counts loads and stores in order): 0.6 us of the 31.6 us frame (same-box timing experiment, profiles/r4a_wg8_steps.txt)
    // r6 (FE_WG8_MIXSPLIT): the mixed gate tile's x half and h half are computed by different waves (4 / 5: x, 6 / 7: h - 63 MFMAs per SIMD instead
    // of 72 / 72 / 54 / 54); both write their raw sums here, [half][row tile][16 rows][LDM], and the h-half wave finishes the tile's gate math
    // one element per lane.  + 4 words: the x-half waves' "written" counters per row tile (monotonic: frame * KB + block + 1)
    static constexpr int LDM = 13;
    static constexpr int MXB = L::NOSTAGE_TOTAL + 4 * SLOT, MX_FLOATS = round_up(4 * 16 * LDM + 8, 4), MXF = MXB + 4 * 16 * LDM;      // (+ 4 words: FE_WG8_FUSEPOST's counters per SIMD pair)
    static constexpr int HST = MXB + MX_FLOATS;
    static constexpr bool HSTASH = (size_t)(HST + S::KB * S::F2 * S::C2) * 4 <= 160 * 1024;
    static constexpr size_t BYTES = (size_t)(HST + (HSTASH ? S::KB * S::F2 * S::C2 : 0)) * 4;
    static constexpr int NPWB = ceil_div(ceil_div(SLOT, 256), kWaves8);      // pieces per wave of a block unit
    static constexpr bool PLAN_OK = 2 * SLOT >= Pack<S>::umax() && BYTES <= 160 * 1024 && (S::NU & 1) == 0;
    static constexpr bool OK = OK0 && PLAN_OK;
};

// One conv-layout GEMM job of a wave: row tile mt x the NT channel tiles J0 .. J0 + NT - 1, K = 4 KS.
//   af(ks) -> A fragment element, bf(j, ks) -> B fragment element of ABSOLUTE tile j, bias(j) -> accumulator start.
// Epilogue: optional SiLU (scaled trunk), store to out[(row0 + m)][col] for col < NCOLS.
template <class S, int NT, int J0, int KS, int NCOLS, int LDO, bool ACT, class AF, class BF, class BI, class SIDE>
__device__ __forceinline__ void conv8(AF&& af, BF&& bf, BI&& bias, const SIDE& side, float* out, int row0, int mt, int lane) {
    if constexpr (NT > 0) {
        const int li = lane & 15, lg = lane >> 4;
        f32x4 acc[1][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[0][j] = bias(J0 + j);
        mma_panel<1, NT, KS, Lds<S>::PDK>(acc, [&](int, int ks) { return af(ks); }, [&](int j, int ks) { return bf(J0 + j, ks); }, side);
        side.commit();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = 16 * (J0 + j) + li;
            if (col < NCOLS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * mt + 4 * lg + r;
                    float v = acc[0][j][r];
                    if (ACT) v = silu_scaled_f(v);
                    out[(row0 + m) * LDO + col] = v;
                }
            }
        }
    } else {
        // a wave without a tile still moves its share of the next weight unit
        constexpr int NSG = (KS + 3) / 4;
#pragma unroll
        for (int g = 0; g < NSG; ++g) side(g, NSG);
        side.commit();
    }
}
// the pair split: wh = 0 takes the tiles [0, NA), wh = 1 the tiles [NA, NA + NB)
template <class S, int NA, int NB, int KS, int NCOLS, int LDO, bool ACT, class AF, class BF, class BI, class SIDE>
__device__ __forceinline__ void conv8_pair(int wh, AF&& af, BF&& bf, BI&& bias, const SIDE& side, float* out, int row0, int mt, int lane) {
    if (wh == 0) conv8<S, NA, 0, KS, NCOLS, LDO, ACT>(af, bf, bias, side, out, row0, mt, lane);
    else conv8<S, NB, NA, KS, NCOLS, LDO, ACT>(af, bf, bias, side, out, row0, mt, lane);
}

// DBG: per-stage dumps (fe_debug_step) and the phase cycle probes (fe_profile_step).  PERSIST: more streams than CUs,
// each workgroup walks the streams blockIdx.x, blockIdx.x + gridDim.x, ...
template <class S, bool DBG, bool PERSIST>
__global__ void __launch_bounds__(kThreads8) __attribute__((amdgpu_waves_per_eu(2, 2))) fe_frame8_kernel(FrameArgs a_in) {
    static_assert(Wg8<S>::OK, "fe_frame8_kernel: shape outside the 512-thread kernel's plan");
    FrameArgs a = a_in;
#ifdef FE_PROBE_HOT
    if constexpr (!DBG) a.dbg = nullptr;
#else
    if constexpr (!DBG) { a.dbg = nullptr; a.clk = nullptr; }
#endif
    FE_CLK(62);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = Lds<S>;
    using W8 = Wg8<S>;
    constexpr int N = S::NFFT, H = S::HOP, OVL = S::OVL, F0 = S::F0, F1 = S::F1;
    constexpr int C1 = S::C1, C2 = S::C2, F2 = S::F2, HD = S::HD;
    constexpr int LDC = S::LDC, LDX = S::LDX, LDG = L::LDGX;
    constexpr int NTH = kThreads8;
#ifndef FE_WG8_PDK
#define FE_WG8_PDK 3      // software-pipeline depth of the token GEMMs (A and B both from LDS)
#endif
    constexpr int PDK = FE_WG8_PDK;

    const int tid0 = threadIdx.x;
    const int tid = tid0;
    const int lane = tid & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave0;
#ifdef FE_WG8_PRIO
    if (wave0 >= 4) __builtin_amdgcn_s_setprio(FE_WG8_PRIO);      // experiment: the second wave of a SIMD runs ahead of the first
#endif
    const float* __restrict__ wp = a.wp;
    WSrc<true> wb;
    constexpr PackedOffsets o = Pack<S>::v;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, o.total * 4, 0x00020000);
    wb.lane4 = lane * 4;
    wb.li4 = (lane & 15) * 4;
    wb.lds = nullptr;
    wb.base = 0;
    wb.k4d = 0;

    // ---- one-time: the zero halos (compressed spectrum, skip buffers), twiddles, weight unit 0.
    // !PERSIST (one stream per workgroup): everything the front of the frame waits for is requested HERE, in one memory round trip
    // with the prologue's own loads - the frame and its window, the DFT constants, and weight unit 1 as well (both staging
    // buffers are free at this point), so that enc_pre, a 12-MFMA phase, does not wait for 28 KiB of the next layer's weights.
    float* sc = smem + L::SC;
    float* Ebuf = smem + L::E;
    float* W0 = smem + L::W0;
    float* W1 = smem + L::W1;
    float2* tw = reinterpret_cast<float2*>(smem + L::TW);
    constexpr int NPW = W8::NPW;
    DmaJobT<NPW> job;
    job.l = smem + W8::WB0;
    job.rsrc = wb.rsrc;
    job.soff = (o.u_off[0] + wave * 256) * 4;
    job.wave = wave;
    job.lane = lane;
    typename Dft<S>::FwdConst dc;
    float fv = 0.0f, fw = 0.0f;
    {
        const StageSide<NPW, o.u_size[0] / 256, kWaves8> st0{&job};
        st0(0, 1);
        float2 twv = make_float2(0.0f, 0.0f);
        if (tid < N / 2) twv = reinterpret_cast<const float2*>(wp + o.twiddle)[tid];
        // (r6: committing unit 1 after the forward DFT instead of here - vmcnt completes in order, the first barrier waits for its 28 KiB - measured neutral: 30.76 / 30.76 us)
        DmaJobT<NPW> job1 = job;
        job1.l = smem + W8::WB1;
        job1.soff = (o.u_off[1] + wave * 256) * 4;
        const StageSide<NPW, PERSIST ? 0 : o.u_size[1] / 256, kWaves8> st1{&job1};
        if constexpr (!PERSIST) {
            const int b0 = (int)blockIdx.x;
            fv = (tid < OVL) ? a.cache_stft[(size_t)b0 * OVL + tid] : a.wav_in[(size_t)b0 * a.in_stride + tid - OVL];
            fw = wp[o.window + tid];
            if (wave < 4) Dft<S>::load(dc, wb, o, wave);
            st1(0, 1);
        }
        for (int i = tid; i < 2 * S::LDS_S; i += NTH) smem[L::SC + i] = 0.0f;
        for (int i = tid; i < (S::NL + 1) * 2 * LDC; i += NTH) {
            const int e = i / (2 * LDC), q = i - e * (2 * LDC);
            smem[L::E + e * S::ACT + (q >= LDC ? (F1 + 1) * LDC + (q - LDC) : q)] = 0.0f;
        }
        if (tid < N / 2) tw[tid] = twv;
        if (tid < 8) smem[W8::MXF + tid] = 0.0f;          // (the hand-over counters of the mixed tile / the dec_post pairs: ints, 0)
        st0.commit();
        if constexpr (!PERSIST) {
            st1.commit();
            smem[L::FFT_A + tid] = fv * fw;
        }
    }
    __syncthreads();

    int b = (int)blockIdx.x;
    int fc = 0;
#pragma unroll 1
    do {
        // PERSIST: a loop-variant zero keeps the wave-uniform / per-lane offsets of a frame from being hoisted out of the stream loop
        // (hoisted, they stay live through the whole frame and spill - see fe_frame_kernel)
        int lz = 0, lzv = 0;
        if constexpr (PERSIST) { asm volatile("" : "+s"(lz)); asm volatile("" : "+v"(lzv)); }
        const int wave = wave0 + lz;
        const int tid = tid0 + lzv;
        const int lane = tid & 63;
        const int ws = wave & 3, wh = wave >> 2;          // SIMD slot, half
        const int li = lane & 15, lg = lane >> 4;
        float* cst = a.cache_stft + (size_t)b * OVL;
        float* cis = a.cache_istft + (size_t)b * OVL;
        const int fpar = (S::NU & 1) ? (fc & 1) : 0;
#define FE8_BEGIN_UNIT(U)                                                                          \
        constexpr int fe_un_ = ((U) + 1 == S::NU) ? 0 : (U) + 1;                                   \
        {                                                                                          \
            const int slot_ = ((U) & 1) ^ fpar;                                                    \
            job.l = smem + (slot_ ? W8::WB0 : W8::WB1);                                            \
            job.soff = (o.u_off[fe_un_] + wave * 256) * 4;                                         \
            wb.lds = smem + (slot_ ? W8::WB1 : W8::WB0);                                           \
            wb.base = o.u_off[(U)];                                                                \
        }                                                                                          \
        const StageSide<NPW, ((!PERSIST && ((U) + 1 == S::NU || (U) == 0)) || (U) == S::U_RFPRE + 1) ? 0 : o.u_size[fe_un_] / 256, kWaves8> stage{&job}
#if FE_WG8_HPRE
        // ---- W_hh h_{t-1} of every block, on waves 4-7: job = (channel group hcg, row tile hrt) [+ the mixed tile on waves 4, 5]
        constexpr int HK2 = S::KS_2, HNQ = HK2 / 4, HKR = HK2 % 4, HTS = HNQ * 256 + HKR * 64;
        static_assert(HKR <= 1, "one plain k-step after the 16-byte groups");
        const int hcg = (wave - 4) >> 1, hrt = wave & 1;
        f32x4 hp[S::KB][4];                     // [block]: r, z, n (with b_hn) of the group; [3]: the mixed tile (waves 4, 5)
        float hpa[S::KB][HK2];                  // per block: A fragments (rows of h), B fragments of the job's tiles, start values - all requested
        f32x4 hpb[S::KB][4][HNQ > 0 ? HNQ : 1];    // at the top of the frame (dead once the block's products are issued)
        float hpr[S::KB][4], hpbn[S::KB], hpbm[S::KB];
        auto hpre_load = [&](auto k_) {
            constexpr int k = decltype(k_)::value;
            if (wave >= 4) {
                const int row = 16 * hrt + (lane & 15);
                const float* hrow = a.h + ((size_t)k * a.B + b) * (S::F2 * S::C2) + (row < S::F2 ? row : S::F2 - 1) * S::C2 + (lane >> 4);
#pragma unroll
                for (int ks = 0; ks < HK2; ++ks) hpa[k][ks] = hrow[4 * ks];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = j < 3 ? 3 * hcg + j : 3 * S::G8_NG;              // (j = 3: the mixed tile; fetched by waves 6, 7 too - unused there)
#pragma unroll
                    for (int q = 0; q < HNQ; ++q) hpb[k][j][q] = wb.at_gv4(o.u8_gh4[k] + t * HTS + q * 256, wb.lane4 * 4);
                    if constexpr (HKR == 1) hpr[k][j] = wb.at_g(o.u8_gh4[k] + t * HTS + HNQ * 256);
                }
                hpbn[k] = wb.at16_g(o.u8_gh[k] + S::G8_NT * HK2 * 64 + (3 * hcg + 2) * 16);      // b_hn of the group's n tile
                hpbm[k] = wb.at16_g(o.u8_gh[k] + S::G8_NT * HK2 * 64 + (3 * S::G8_NG) * 16);      // the mixed tile's h-side start values
            }
        };
        auto hpre_mma = [&](auto k_) {
            constexpr int k = decltype(k_)::value;
            if (wave >= 4) {
                hp[k][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                hp[k][1] = hp[k][0];
                hp[k][2] = f32x4{hpbn[k], hpbn[k], hpbn[k], hpbn[k]};
                hp[k][3] = f32x4{hpbm[k], hpbm[k], hpbm[k], hpbm[k]};
                auto bfr = [&](int j, int ks) { return ks < 4 * HNQ ? hpb[k][j][ks >> 2][ks & 3] : hpr[k][j]; };
                if (wave < 6) {
#pragma unroll
                    for (int ks = 0; ks < HK2; ++ks)
#pragma unroll
                        for (int j = 0; j < 4; ++j) hp[k][j] = FE_MFMA(hpa[k][ks], bfr(j, ks), hp[k][j]);
                } else {
#pragma unroll
                    for (int ks = 0; ks < HK2; ++ks)
#pragma unroll
                        for (int j = 0; j < 3; ++j) hp[k][j] = FE_MFMA(hpa[k][ks], bfr(j, ks), hp[k][j]);
                }
            }
        };
        using HK0 = std::integral_constant<int, 0>;
        using HK1 = std::integral_constant<int, (S::KB > 1 ? 1 : 0)>;
        using HK2_ = std::integral_constant<int, (S::KB > 2 ? 2 : 0)>;
        static_assert(S::KB <= 3, "FE_WG8_HPRE schedules three blocks");
        hpre_load(HK0{});
        if constexpr (S::KB > 1) hpre_load(HK1{});
        if constexpr (S::KB > 2) hpre_load(HK2_{});
#endif
        FE_CLK(0);
        // =========================== STFT (a1-a3) ===========================
        // LDS quarters of the FFT arena: q0 windowed frame / iSTFT partial sums, q1 iSTFT partial sums, q3 spectrum {Re[N/2], Im[N/2]}
        float* q0 = smem + L::FFT_A;
        float* q1 = q0 + N;
        float* q3 = smem + L::FFT_B + N;
        {
            if constexpr (PERSIST) {
                if (wave < 4) Dft<S>::load(dc, wb, o, wave);             // in flight while the frame is fetched
                const float* xin = a.wav_in + (size_t)b * a.in_stride;
                fv = (tid < OVL) ? cst[tid] : xin[tid - OVL];            // one sample per thread
                fw = wp[o.window + tid];
                q0[tid] = fv * fw;
                __syncthreads();
            }
            // cache' = frame[H:], straight from the registers (every load of the old cache has landed: its value went to LDS above)
            if (tid >= H) cst[tid - H] = fv;
            FE_CLK(1);
            float* nyq = a.dbg ? a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(0) + 2 * F0 : nullptr;
            if (wave < 4) Dft<S>::template forward<false>(q0, q3, tw, dc, wave, lane, nyq);
#if FE_WG8_HPRE
            else {
                hpre_mma(HK0{});
                if constexpr (S::KB > 1) hpre_mma(HK1{});
            }
#endif
            __syncthreads();
            FE_CLK(2);
            const float* Xr = q3;
            const float* Xi = q3 + N / 2;
            if (a.dbg) {
                float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(0);
                for (int f = tid; f < F0; f += NTH) { dst[2 * f] = Xr[f]; dst[2 * f + 1] = Xi[f]; }
            }
            // =========================== compress (a4) ===========================
            for (int f = tid; f < F0; f += NTH) {
                const float re = Xr[f], im = Xi[f];
                const float mag = fmaxf(sqrtf(re * re + im * im), 1.0e-5f);
                const float g = pow_f(mag, a.compression - 1.0f);
                sc[2 + f] = re * g;
                sc[S::LDS_S + 2 + f] = im * g;
            }
        }
        __syncthreads();
        if (a.dbg) {
            float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(1);
            for (int f = tid; f < F0; f += NTH) { dst[2 * f] = sc[2 + f]; dst[2 * f + 1] = sc[S::LDS_S + 2 + f]; }
        }

        FE_CLK(3);
        // =========================== enc_pre (a5): strided conv as a K = 16 GEMM ===========================
        {
            FE8_BEGIN_UNIT(0);
            conv8_pair<S, W8::NTA, W8::NTB, 4, C1, LDC, true>(
                wh,
                [&](int ks) {
                    const int kk = 4 * ks + lg;
                    const int c = kk & 1, s = (kk >> 1) & 3, tp = kk >> 3;
                    return sc[c * S::LDS_S + 4 * (16 * ws + li + tp) + s];
                },
                [&](int j, int ks) { return wb.at(o.enc_pre_w + (j * 4 + ks) * 64); },
                [&](int j) { return wb.at16x4(o.enc_pre_b + j * 64); }, stage, Ebuf, 1, ws, lane);
#if FE_WG8_HPRE
            if constexpr (S::KB > 2) hpre_mma(HK2_{});        // (waves 4-7: the lighter half of this 12-MFMA phase)
#endif
        }
        __syncthreads();
        dbg_dump<S, NTH>(a, b, 2, Ebuf + LDC, LDC);

        FE_CLK(4);
        // =========================== encoder (a6): k = 3 convs ===========================
        static_for<S::NL>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            const float* in = Ebuf + l * S::ACT;
            float* out = Ebuf + (l + 1) * S::ACT;
            if (l == 0) FE_CLK(40);
            {
                FE8_BEGIN_UNIT(S::U_ENC + l);
                const float* const t0 = in + (16 * ws + li) * LDC + lg;
                conv8_pair<S, W8::NTA, W8::NTB, 3 * S::KS_C, C1, LDC, true>(
                    wh, [&](int ks) { return t0[(ks / S::KS_C) * LDC + 4 * (ks % S::KS_C)]; },
                    [&](int j, int ks) { return wb.at(o.enc_w[l] + (j * (3 * S::KS_C) + ks) * 64); },
                    [&](int j) { return wb.at16x4(o.enc_b[l] + j * 64); }, stage, out, 1, ws, lane);
            }
            if (l == 0) FE_CLK(42);
            __syncthreads();
            if (l == 0) FE_CLK(43);
            dbg_dump<S, NTH>(a, b, 3 + l, out + LDC, LDC);
        });

        float* Xb = smem + L::X;
        float* Hl = smem + L::HL;
        float* Hs = smem + L::HS;
        float* Gi = smem + L::GI;
        float* Y1 = smem + L::Y1;
        float* Y2 = smem + L::Y2;

        FE_CLK(5);
        // =========================== rf_pre (a7) ===========================
        constexpr int NTPW3 = S::NTPW3;
        constexpr int HPT = W8::HPT;
        constexpr int K2 = S::KS_2;
        f32x4 xr;                                      // residual stream x: row tile wh, channel tile ws
        // GRU: (channel group, row tile) jobs - waves 0-3 a 16-channel group x a row tile (three gate tiles: 54 MFMAs), waves 4 and 5
        // (SIMDs 0 and 1) the mixed tile of the left-over channels x a row tile (18 MFMAs), waves 6 and 7 none: 72 / 72 / 54 / 54
        // MFMAs per SIMD, and the three gates of a (row, channel) meet in one lane
        const int g_rt = wave & 1;                                         // row tile of this wave's GRU job
        const int g_t0 = wave < 4 ? 3 * (wave >> 1) : 3 * S::G8_NG;        // its first gate tile
        // block weights in LDS (W8 slots): B fragment (tile, k-step) of a unit at u[(tile * K2 + ks) * 64 + lane], start value of a tile's
        // column at u[NT * K2 * 64 + tile * 16 + li]
#ifdef FE_EXP_BREG      // timing experiment (wrong results): the blocks' B operands as register values, no block-weight staging - the upper bound of register-resident block weights
        float breg0 = 0.001f * (float)lane, breg1 = 0.002f, breg2 = -0.001f, breg3 = 0.0005f;
        asm volatile("" : "+v"(breg0), "+v"(breg1), "+v"(breg2), "+v"(breg3));
#define FE8_B(expr, ks) (((ks) & 3) == 0 ? breg0 : ((ks) & 3) == 1 ? breg1 : ((ks) & 3) == 2 ? breg2 : breg3)
#define FE8_STAGE_ON 0
#else
#define FE8_B(expr, ks) (expr)
#define FE8_STAGE_ON 1
#endif
#ifdef FE_EXP_AREG      // ... and their A operands too (no LDS operand reads in the blocks' GEMMs at all)
        float areg0 = 0.003f * (float)(lane & 15), areg1 = 0.001f, areg2 = -0.002f;
        asm volatile("" : "+v"(areg0), "+v"(areg1), "+v"(areg2));
#define FE8_A(expr, ks) ((ks) % 3 == 0 ? areg0 : (ks) % 3 == 1 ? areg1 : areg2)
#else
#define FE8_A(expr, ks) (expr)
#endif
        const float* const sGx = smem + W8::WB0 + 1 * W8::SLOT;
        const float* const sGh = smem + W8::WB0 + 3 * W8::SLOT;
        const float* const sF = smem + W8::WB0 + 0 * W8::SLOT;
        const float* const sQ = smem + W8::WB0 + 2 * W8::SLOT;
        constexpr int NPWB = W8::NPWB;
        DmaJobT<NPWB> jb1, jb2;                        // staging jobs of the block units (two may ride in one GEMM)
        jb1.rsrc = wb.rsrc; jb1.wave = wave; jb1.lane = lane; jb1.l = smem; jb1.soff = 0;
        jb2 = jb1;
        auto stage_to = [&](DmaJobT<NPWB>& j, int src_floats, int dst_floats) { j.l = smem + dst_floats; j.soff = (src_floats + wave * 256) * 4; };
        constexpr int u8_stride = S::KB > 1 ? o.u8_gx[1] - o.u8_gx[0] : 0;
        using StG = StageSide<NPWB, FE8_STAGE_ON * (S::U8_G / 256), kWaves8>;
        using StF = StageSide<NPWB, FE8_STAGE_ON * (S::U8_F / 256), kWaves8>;
        using StQ = StageSide<NPWB, FE8_STAGE_ON * (S::U8_Q / 256), kWaves8>;
        float pe_r[4];
        // unpredicated epilogue stores into the [F2P][C2 + 2] token buffers (pad rows are real rows, lanes past C2 aim at the pad column)
        auto tok_dst = [&](float* base) {
            const int col = 16 * ws + li;
            return base + (16 * wh + 4 * lg) * LDX + (col < C2 ? col : C2);
        };
        int hs_off[HPT];
#pragma unroll
        for (int q = 0; q < HPT; ++q) {
            const int i = tid + q * NTH, f = i / C2;
            hs_off[q] = i < F2 * C2 ? f * LDX + (i - f * C2) : C2;
        }
        {
            // Y1[f2][c1] = sum_f1 Wf[f2][f1] E[f1][c1]      (A = packed filterbank rows of tile wh, B = LDS, channel tile ws)
            constexpr int KS = F1 / 4;
            const float* Ein = Ebuf + S::NL * S::ACT + LDC;   // row 0 = bin 0
            FE8_BEGIN_UNIT(S::U_RFPRE);
            stage_to(jb1, o.u8_gx[0], W8::WB0 + 1 * W8::SLOT);          // block 0's GRU input weights -> slot 1
            const StG stg{&jb1};
            f32x4 acc[1][1];
            acc_init_zero<1, 1>(acc);
            const int nt = ws < S::NTC ? ws : S::NTC - 1;
            mma_panel<1, 1, KS, PDK>(
                acc, [&](int, int ks) { return wb.at(o.rfpre_lin + (wh * KS + ks) * 64); },
                [&](int, int ks) { return Ein[(4 * ks + lg) * LDC + 16 * nt + li]; }, side2(stage, stg));
            stage.commit();
            stg.commit();
            const int col = 16 * ws + li;
            if (ws < S::NTC && col < C1 && 16 * wh + 4 * lg < F2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Y1[(16 * wh + 4 * lg + r) * LDC + col] = acc[0][0][r];
            }
        }
        __syncthreads();
        {
            // X[f2][c2] = Y1[f2][:] . Wc[c2][:] + b
            FE8_BEGIN_UNIT(S::U_RFPRE + 1);
#if FE_WG8_HPRE
            const NoSide stg{};                                         // (the hidden weights never pass through LDS)
#else
            stage_to(jb1, o.u8_gh[0], W8::WB0 + 3 * W8::SLOT);          // ... and the hidden weights -> slot 3
            const StG stg{&jb1};
#endif
            float hpre[HPT];
            const float* hg0 = a.h + (size_t)b * (F2 * C2);
#pragma unroll
            for (int q = 0; q < HPT; ++q) { const int i = tid + q * NTH; hpre[q] = hg0[i < F2 * C2 ? i : F2 * C2 - 1]; }
            f32x4 acc[1][1];
            const int nt = ws < S::NT2 ? ws : S::NT2 - 1;
            acc[0][0] = wb.at16x4(o.rfpre_b + nt * 64);
            const float* ya = Y1 + (16 * wh + li) * LDC + lg;
            mma_panel<1, 1, S::KS_C, PDK>(
                acc, [&](int, int ks) { return ya[4 * ks]; },
                [&](int, int ks) { return wb.at(o.rfpre_w + (nt * S::KS_C + ks) * 64); }, stg);
            (void)stage;
#if !FE_WG8_HPRE
            stg.commit();
#endif
            xr = acc[0][0];
            float* xd = tok_dst(Xb);
#pragma unroll
            for (int r = 0; r < 4; ++r) xd[r * LDX] = acc[0][0][r];
#pragma unroll
            for (int q = 0; q < HPT; ++q) Hs[hs_off[q]] = hpre[q];
        }
        __syncthreads();
        dbg_dump<S, NTH>(a, b, 3 + S::NL, Xb, LDX);

        FE_CLK(6);
        // =========================== RNNFormer blocks (a9-a11) ===========================
#pragma unroll
        for (int k = 0; k < S::KB; ++k) {
            float* hg = a.h + ((size_t)k * a.B + b) * (F2 * C2);
            const int ub = k * u8_stride;
            if (k == 0) FE_CLK(20);
            {
                // GRU (nn.GRU gate order r, z, n; model.py:187, 271) over (channel group, row tile) jobs, the gate math in the GEMM
                // epilogue - r, z, n of a (row, channel) sit in one lane: no exchange through LDS, one barrier for the phase.
                // Staged meanwhile: this block's rnn_fc -> slot 0 and qkv -> slot 2.
                stage_to(jb1, o.u8_f1[0] + ub, W8::WB0 + 0 * W8::SLOT);
                stage_to(jb2, o.u8_q[0] + ub, W8::WB0 + 2 * W8::SLOT);
                const StF st1{&jb1};
                const StQ st2{&jb2};
                const auto sideg = side2(st1, st2);
                if (k == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * wh + 4 * lg + r, col = 16 * ws + li;
                        pe_r[r] = wb.gather_g(o.blk_pe + (row < F2 ? row : F2 - 1) * C2 + (col < C2 ? col : C2 - 1));
                    }
                }
                const float* xa = Xb + (16 * g_rt + li) * LDX + lg;
                const float* ha = Hs + (16 * g_rt + li) * LDX + lg;
                const float* wx = sGx + g_t0 * (K2 * 64) + lane;
                const float* wh_ = sGh + g_t0 * (K2 * 64) + lane;
                const float* bx = sGx + S::G8_NT * K2 * 64 + g_t0 * 16 + li;
                const float* bh = sGh + S::G8_NT * K2 * 64 + g_t0 * 16 + li;
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0) FE_CLK(45);
#if FE_WG8_HPRE
                if (wave >= 4) {
                    // x halves on top of the hidden halves accumulated in the front: the group's tiles r, z, n (+ the mixed tile on waves 4, 5:
                    // it shares the row tile's A fragments)
                    const int ch = 16 * hcg + li;
                    constexpr int R = S::G8_R;
                    const int chm = 16 * S::G8_NG + (li < R ? li : 0);
                    float hprev[4], hprevm[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        hprev[r] = Hs[(16 * hrt + 4 * lg + r) * LDX + ch];
                        hprevm[r] = Hs[(16 * hrt + 4 * lg + r) * LDX + chm];
                    }
                    const float* xa8 = Xb + (16 * hrt + li) * LDX + lg;
                    const float* wx8 = sGx + (3 * hcg) * (K2 * 64) + lane;
                    const float* wxm = sGx + (3 * S::G8_NG) * (K2 * 64) + lane;
                    const float* bx8 = sGx + S::G8_NT * K2 * 64 + (3 * hcg) * 16 + li;
                    const float b0 = bx8[0], b1 = bx8[16], b2 = bx8[32], bm = sGx[S::G8_NT * K2 * 64 + (3 * S::G8_NG) * 16 + li];
                    f32x4 ar, az, anx = f32x4{b2, b2, b2, b2}, anh, ax = f32x4{bm, bm, bm, bm}, ah;
                    static_for<S::KB>([&](auto kk_) {         // (hp is indexed at compile time: the block loop is unrolled)
                        constexpr int kk = decltype(kk_)::value;
                        if (k == kk) { ar = hp[kk][0] + f32x4{b0, b0, b0, b0}; az = hp[kk][1] + f32x4{b1, b1, b1, b1}; anh = hp[kk][2]; ah = hp[kk][3]; }
                    });
                    if (wave < 6) {
                        mma_panel_sel<1, 4, K2, PDK>(
                            [&](int, int j, int) -> f32x4& { return j == 0 ? ar : (j == 1 ? az : (j == 2 ? anx : ax)); },
                            [&](int, int ks) { return xa8[4 * ks]; },
                            [&](int j, int ks) { return j < 3 ? wx8[(j * K2 + ks) * 64] : wxm[ks * 64]; }, sideg);
                    } else {
                        mma_panel_sel<1, 3, K2, PDK>(
                            [&](int, int j, int) -> f32x4& { return j == 0 ? ar : (j == 1 ? az : anx); },
                            [&](int, int ks) { return xa8[4 * ks]; },
                            [&](int j, int ks) { return wx8[(j * K2 + ks) * 64]; }, sideg);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (k == 0) FE_CLK(46);
                    float hn[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float rr = sigmoid_pre(ar[r]);        // (the packer scales the gate rows: -log2 e / 2 log2 e)
                        const float zz = sigmoid_pre(az[r]);
                        const float nn = tanh_pre(__builtin_fmaf(rr, anh[r], anx[r]));
                        hn[r] = __builtin_fmaf(zz, hprev[r] - nn, nn);          // (1 - z) n + z h
                    }
                    if (16 * hrt + 4 * lg < F2) {       // (F2 % 4 == 0: the four rows of a lane are valid together)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * hrt + 4 * lg + r;
                            Hl[row * LDX + ch] = hn[r];
                            if constexpr (W8::HSTASH) smem[W8::HST + k * (F2 * C2) + row * C2 + ch] = hn[r];
                            else hg[row * C2 + ch] = hn[r];
                        }
                    }
                    if (wave < 6) {
                        // the mixed tile: lanes li < R hold r, R .. 2 R - 1 z, 2 R .. 3 R - 1 n of channel 16 NG + li % R; the z and n values
                        // move down to the r lanes (DPP row shifts), which finish the R channels
                        auto shl = [](float v, auto n_) {      // lane i <- lane i + n of its 16-lane row
                            constexpr int n = decltype(n_)::value;
                            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + n, 0xf, 0xf, true));
                        };
                        float hm[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sx = ax[r], sh = ah[r], sm = sx + sh;
                            const float rr = sigmoid_pre(sm);
                            const float zz = sigmoid_pre(shl(sm, std::integral_constant<int, R>{}));
                            const float nn = tanh_pre(__builtin_fmaf(rr, shl(sh, std::integral_constant<int, 2 * R>{}), shl(sx, std::integral_constant<int, 2 * R>{})));
                            hm[r] = __builtin_fmaf(zz, hprevm[r] - nn, nn);
                        }
                        if (li < R && 16 * hrt + 4 * lg < F2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * hrt + 4 * lg + r;
                                Hl[row * LDX + chm] = hm[r];
                                if constexpr (W8::HSTASH) smem[W8::HST + k * (F2 * C2) + row * C2 + chm] = hm[r];
                                else hg[row * C2 + chm] = hm[r];
                            }
                        }
                    }
                } else {
                    constexpr int NSG = (K2 + 3) / 4;      // waves 0-3: their share of the staging only
#pragma unroll
                    for (int g = 0; g < NSG; ++g) sideg(g, NSG);
                }
#else
                if (wave < 4) {
                    // a 16-channel group: tiles r, z (x and h halves in one accumulator), n (separate halves)
                    const int ch = 16 * (wave >> 1) + li;
                    float hprev[4];                      // previous state of this lane's outputs: in flight under the GEMM
#pragma unroll
                    for (int r = 0; r < 4; ++r) hprev[r] = Hs[(16 * g_rt + 4 * lg + r) * LDX + ch];
                    const float b0 = bx[0], b1 = bx[16], b2 = bx[32], b3 = bh[32];
                    f32x4 ar = f32x4{b0, b0, b0, b0}, az = f32x4{b1, b1, b1, b1};
                    f32x4 anx = f32x4{b2, b2, b2, b2}, anh = f32x4{b3, b3, b3, b3};
#ifdef FE_EXP_NOHH      // timing experiment (wrong results): the GRU phases without their W_hh h products
                    constexpr int KSG = K2;
#else
                    constexpr int KSG = 2 * K2;
#endif
                    mma_panel_sel<1, 3, KSG, PDK>(
                        [&](int, int j, int ks) -> f32x4& { return j == 0 ? ar : (j == 1 ? az : (ks < K2 ? anx : anh)); },
                        [&](int, int ks) { return FE8_A(ks < K2 ? xa[4 * ks] : ha[4 * (ks - K2)], ks); },
                        [&](int j, int ks) { return FE8_B(ks < K2 ? wx[(j * K2 + ks) * 64] : wh_[(j * K2 + (ks - K2)) * 64], ks + j); }, sideg);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k == 0) FE_CLK(46);
                    float hn[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float rr = sigmoid_pre(ar[r]);        // (the packer scales the gate rows: -log2 e / 2 log2 e)
                        const float zz = sigmoid_pre(az[r]);
                        const float nn = tanh_pre(__builtin_fmaf(rr, anh[r], anx[r]));
                        hn[r] = __builtin_fmaf(zz, hprev[r] - nn, nn);          // (1 - z) n + z h
                    }
                    if (16 * g_rt + 4 * lg < F2) {       // (F2 % 4 == 0: the four rows of a lane are valid together)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * g_rt + 4 * lg + r;
                            Hl[row * LDX + ch] = hn[r];
                            if constexpr (W8::HSTASH) smem[W8::HST + k * (F2 * C2) + row * C2 + ch] = hn[r];
                            else hg[row * C2 + ch] = hn[r];
                        }
                    }
                }
#if FE_WG8_MIXSPLIT
                else {
                    // r6: the mixed tile (the left-over R channels' r | z | n columns) of row tile g_rt, x half on waves 4 / 5, h half on waves
                    // 6 / 7: 9 MFMAs each next to the 54 of their SIMD's channel-group job - 63 MFMAs per SIMD.  Each half goes to LDS as a
                    // [16 rows][3 R columns] matrix; the h-half wave waits for its partner's counter (a wave's LDS operations execute in order)
                    // and finishes the R channels one (row, channel) per lane.  Same chains, same summation order, same gate arithmetic as the
                    // one-wave form (bit-identical results).
                    constexpr int R = S::G8_R, LDM = W8::LDM;
                    const bool hhalf = wave >= 6;
                    float* mx = smem + W8::MXB + ((hhalf ? 2 : 0) + g_rt) * (16 * LDM);
                    int* mflag = reinterpret_cast<int*>(smem + W8::MXF) + g_rt;
                    const int seq = fc * S::KB + k + 1;
                    f32x4 s0, s1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    __builtin_amdgcn_s_setprio(3);
                    if (!hhalf) {
                        const float b0 = bx[0];
                        s0 = f32x4{b0, b0, b0, b0};
                        mma_panel_sel<1, 1, K2, PDK>([&](int, int, int ks) -> f32x4& { return (ks & 1) ? s1 : s0; },
                                                     [&](int, int ks) { return FE8_A(xa[4 * ks], ks); }, [&](int, int ks) { return FE8_B(wx[ks * 64], ks); }, sideg);
                    } else {
                        const float b1 = bh[0];
                        s0 = f32x4{b1, b1, b1, b1};
                        // (the one-wave form ran x and h as ONE 2 K2-step panel with chains by the parity of the global k-step: K2 is odd here,
                        // so the biased chain took the h half's odd local steps)
                        mma_panel_sel<1, 1, K2, PDK>([&](int, int, int ks) -> f32x4& { return ((ks + K2) & 1) ? s1 : s0; },
                                                     [&](int, int ks) { return FE8_A(ha[4 * ks], ks); }, [&](int, int ks) { return FE8_B(wh_[ks * 64], ks); }, sideg);
                    }
                    // (the priority stays raised through the hand-over and the gate math: at priority 0 every instruction of this dependent chain
                    // waited for a gap between the MFMAs of the SIMD's 54-MFMA job - 1.5 k cycles for ~35 instructions, and this wave arrived last)
                    s0 += s1;
                    __builtin_amdgcn_sched_barrier(0);
                    if (k == 0) FE_CLK(46);
                    if (li < 3 * R) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx[(4 * lg + r) * LDM + li] = s0[r];
                    }
                    if (!hhalf) {
                        __atomic_signal_fence(__ATOMIC_SEQ_CST);
                        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the writes above are in LDS
                        if (lane == 0) __hip_atomic_store(mflag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    } else {
                        const int row = lane >> 2, c = lane & 3;
                        static_assert(R == 4, "one (row, channel) per lane: sixteen rows x four channels");
                        const int ch = 16 * S::G8_NG + c;
                        const bool live = 16 * g_rt + row < F2;
                        const float hprev = Hs[(16 * g_rt + row) * LDX + ch];
                        __atomic_signal_fence(__ATOMIC_SEQ_CST);
                        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(mflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < seq) {}
                        __atomic_signal_fence(__ATOMIC_SEQ_CST);
                        const float* mxx = smem + W8::MXB + g_rt * (16 * LDM) + row * LDM + c;
                        const float* mxh = mx + row * LDM + c;
                        const float sxr = mxx[0], sxz = mxx[R], sxn = mxx[2 * R];
                        const float shr = mxh[0], shz = mxh[R], shn = mxh[2 * R];
                        const float rr = sigmoid_pre(sxr + shr);
                        const float zz = sigmoid_pre(sxz + shz);
                        const float nn = tanh_pre(__builtin_fmaf(rr, shn, sxn));
                        const float hn = __builtin_fmaf(zz, hprev - nn, nn);          // (1 - z) n + z h
                        if (live) {
                            Hl[(16 * g_rt + row) * LDX + ch] = hn;
                            if constexpr (W8::HSTASH) smem[W8::HST + k * (F2 * C2) + (16 * g_rt + row) * C2 + ch] = hn;
                            else hg[(16 * g_rt + row) * C2 + ch] = hn;
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
#else
                else if (wave < 6) {
                    // the mixed tile: lanes li < R hold r, R .. 2 R - 1 z, 2 R .. 3 R - 1 n of channel 16 NG + li % R; the z and n values
                    // move down to the r lanes (DPP row shifts), which finish the R channels.  (18 dependent MFMAs next to the 54
                    // independent ones of this SIMD's big job: two chains per half, and a raised priority - arbitrated oldest-first,
                    // this wave got a matrix-pipe slot only when the other one stalled and the whole workgroup waited for it)
                    constexpr int R = S::G8_R;
                    const int ch = 16 * S::G8_NG + (li < R ? li : 0);
                    float hprev[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) hprev[r] = Hs[(16 * g_rt + 4 * lg + r) * LDX + ch];
                    const float b0 = bx[0], b1 = bh[0];
                    f32x4 ax = f32x4{b0, b0, b0, b0}, ah = f32x4{b1, b1, b1, b1};
                    f32x4 ax1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, ah1 = ax1;
#ifdef FE_WG8_MIX4      // four accumulator chains per half instead of two: the dependent chain of the small job ends earlier
                    f32x4 ax2 = ax1, ax3 = ax1, ah2 = ax1, ah3 = ax1;
#endif
                    __builtin_amdgcn_s_setprio(3);
#ifdef FE_EXP_NOHH
                    constexpr int KSG = K2;
#else
                    constexpr int KSG = 2 * K2;
#endif
                    mma_panel_sel<1, 1, KSG, PDK>(
#ifdef FE_WG8_MIX4
                        [&](int, int, int ks) -> f32x4& { return ks < K2 ? ((ks & 3) == 0 ? ax : (ks & 3) == 1 ? ax1 : (ks & 3) == 2 ? ax2 : ax3)
                                                                         : ((ks & 3) == 0 ? ah : (ks & 3) == 1 ? ah1 : (ks & 3) == 2 ? ah2 : ah3); },
#else
                        [&](int, int, int ks) -> f32x4& { return ks < K2 ? ((ks & 1) ? ax1 : ax) : ((ks & 1) ? ah1 : ah); },
#endif
                        [&](int, int ks) { return FE8_A(ks < K2 ? xa[4 * ks] : ha[4 * (ks - K2)], ks); },
                        [&](int, int ks) { return FE8_B(ks < K2 ? wx[ks * 64] : wh_[(ks - K2) * 64], ks); }, sideg);
                    __builtin_amdgcn_s_setprio(0);
#ifdef FE_WG8_MIX4
                    ax = (ax + ax1) + (ax2 + ax3);
                    ah = (ah + ah1) + (ah2 + ah3);
#else
                    ax += ax1;
                    ah += ah1;
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    if (k == 0) FE_CLK(46);
                    auto shl = [](float v, auto n_) {      // lane i <- lane i + n of its 16-lane row
                        constexpr int n = decltype(n_)::value;
                        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + n, 0xf, 0xf, true));
                    };
                    float hn[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sx = ax[r], sh = ah[r], sm = sx + sh;
                        const float rr = sigmoid_pre(sm);
                        const float zz = sigmoid_pre(shl(sm, std::integral_constant<int, R>{}));
                        const float nn = tanh_pre(__builtin_fmaf(rr, shl(sh, std::integral_constant<int, 2 * R>{}), shl(sx, std::integral_constant<int, 2 * R>{})));
                        hn[r] = __builtin_fmaf(zz, hprev[r] - nn, nn);          // (1 - z) n + z h
                    }
                    if (li < R && 16 * g_rt + 4 * lg < F2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * g_rt + 4 * lg + r;
                            Hl[row * LDX + ch] = hn[r];
                            if constexpr (W8::HSTASH) smem[W8::HST + k * (F2 * C2) + row * C2 + ch] = hn[r];
                            else hg[row * C2 + ch] = hn[r];
                        }
                    }
                } else {
                    constexpr int NSG = (2 * K2 + 3) / 4;      // no GRU job: this wave's share of the staging only
#pragma unroll
                    for (int g = 0; g < NSG; ++g) sideg(g, NSG);
                }
#endif
#endif
                st1.commit();
                st2.commit();
            }
            if (k == 0) FE_CLK(47);
            __syncthreads();
            if (k == 0) FE_CLK(21);
            if (k == 0) FE_CLK(22);
            const int ntf = ws < S::NT2 ? ws : S::NT2 - 1;           // this wave's channel tile of the fc layers (ws = 3: a shadow of the last)
            {
                // x += rnn_fc(h') (+ pe in block 0); staged meanwhile: the next block's GRU input weights -> slot 1
                stage_to(jb1, o.u8_gx[0] + ub + u8_stride, W8::WB0 + 1 * W8::SLOT);
                const StG stn{&jb1};
                f32x4 acc[1][1];
                const float bj = sF[S::NT2 * K2 * 64 + ntf * 16 + li];
                acc[0][0] = f32x4{bj, bj, bj, bj};
                const float* hla = Hl + (16 * wh + li) * LDX + lg;
                const float* wf = sF + ntf * (K2 * 64) + lane;
                if (k + 1 < S::KB) {
                    mma_panel<1, 1, K2, PDK>(acc, [&](int, int ks) { return FE8_A(hla[4 * ks], ks); }, [&](int, int ks) { return FE8_B(wf[ks * 64], ks); }, stn);
                    stn.commit();
                } else
                    mma_panel<1, 1, K2, PDK>(acc, [&](int, int ks) { return FE8_A(hla[4 * ks], ks); }, [&](int, int ks) { return FE8_B(wf[ks * 64], ks); }, NoSide{});
                float* xd = tok_dst(Xb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[0][0][r] + xr[r];
                    if (k == 0) v += pe_r[r];
                    xr[r] = v;
                    xd[r * LDX] = v;
                }
            }
            __syncthreads();
            dbg_dump<S, NTH>(a, b, 4 + S::NL + 2 * k, Xb, LDX);
            if (k == 0) FE_CLK(23);
            {
                // qkv = x W_qkv^T -> Gi (columns per head interleaved [h][q|k|v][hd]); staged meanwhile: attn_fc -> slot 0
                stage_to(jb1, o.u8_f2[0] + ub, W8::WB0 + 0 * W8::SLOT);
                const StF st1{&jb1};
                f32x4 acc[1][NTPW3];
                acc_init_zero<1, NTPW3>(acc);
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0) FE_CLK(50);
                const float* xa = Xb + (16 * wh + li) * LDX + lg;
                mma_panel<1, NTPW3, K2, PDK>(acc, [&](int, int ks) { return FE8_A(xa[4 * ks], ks); },
                                             [&](int j, int ks) {
                                                 int nt = ws + 4 * j;
                                                 nt = nt < S::NT3 ? nt : S::NT3 - 1;
                                                 return FE8_B(sQ[(nt * K2 + ks) * 64 + lane], ks + j);
                                             }, st1);
                st1.commit();
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0) FE_CLK(51);
                float* gdst = Gi + (16 * wh + 4 * lg) * LDG + 16 * ws + li;
#pragma unroll
                for (int j = 0; j < NTPW3; ++j)
                    if (ws + 4 * j < S::NT3) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) gdst[r * LDG + 64 * j] = acc[0][j][r];
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0) FE_CLK(52);
            }
            __syncthreads();
            if (k == 0) FE_CLK(24);
            // attention: wave (ws, wh) = head ws, query tile wh
            attention_head<S, 1, LDG>(Gi, Hl, ws * 3 * HD, ws, wh, 1, lane);
            __syncthreads();
            if (k == 0) FE_CLK(25);
            {
                // x += attn_fc(o); staged meanwhile: the next block's GRU hidden weights -> slot 3 (last block: rf_post's filterbank -> WB1);
                // the next block's hidden state is fetched now and parked after the GEMM
                float hpre[HPT];
                if (k + 1 < S::KB) {
                    const float* hgn = hg + (size_t)a.B * (F2 * C2);
#pragma unroll
                    for (int q = 0; q < HPT; ++q) { const int i = tid + q * NTH; hpre[q] = hgn[i < F2 * C2 ? i : F2 * C2 - 1]; }
                }
                f32x4 acc[1][1];
                const float bj = sF[S::NT2 * K2 * 64 + ntf * 16 + li];
                acc[0][0] = f32x4{bj, bj, bj, bj};
                const float* hla = Hl + (16 * wh + li) * LDX + lg;
                const float* wf = sF + ntf * (K2 * 64) + lane;
                if (k + 1 < S::KB) {
#if FE_WG8_HPRE
                    mma_panel<1, 1, K2, PDK>(acc, [&](int, int ks) { return FE8_A(hla[4 * ks], ks); }, [&](int, int ks) { return FE8_B(wf[ks * 64], ks); }, NoSide{});
#else
                    stage_to(jb1, o.u8_gh[0] + ub + u8_stride, W8::WB0 + 3 * W8::SLOT);
                    const StG stn{&jb1};
                    mma_panel<1, 1, K2, PDK>(acc, [&](int, int ks) { return FE8_A(hla[4 * ks], ks); }, [&](int, int ks) { return FE8_B(wf[ks * 64], ks); }, stn);
                    stn.commit();
#endif
                } else {
                    stage_to(jb1, o.u_off[S::U_RFPOST], W8::WB1);
                    const StageSide<NPWB, o.u_size[S::U_RFPOST] / 256, kWaves8> stn{&jb1};
                    mma_panel<1, 1, K2, PDK>(acc, [&](int, int ks) { return FE8_A(hla[4 * ks], ks); }, [&](int, int ks) { return FE8_B(wf[ks * 64], ks); }, stn);
                    stn.commit();
                }
                if (k + 1 < S::KB) {
#pragma unroll
                    for (int q = 0; q < HPT; ++q) Hs[hs_off[q]] = hpre[q];
                }
                float* xd = tok_dst(Xb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[0][0][r] + xr[r];
                    xr[r] = v;
                    xd[r * LDX] = v;
                }
            }
            __syncthreads();
            if (k == 0) FE_CLK(26);
            dbg_dump<S, NTH>(a, b, 5 + S::NL + 2 * k, Xb, LDX);
        }

        FE_CLK(7);
        // =========================== rf_post (a13) ===========================
        {
            // Y2[f1][c2] = sum_f2 Wp[f1][f2] X[f2][c2]      (A = packed filterbank rows of tile ws, B = LDS tokens)
            constexpr int KS = F2 / 4;
            FE8_BEGIN_UNIT(S::U_RFPOST);
            conv8_pair<S, W8::N2A, W8::N2B, KS, C2, LDX, false>(
                wh, [&](int ks) { return wb.at(o.rfpost_lin + (ws * KS + ks) * 64); },
                [&](int j, int ks) { return Xb[(4 * ks + lg) * LDX + 16 * j + li]; },
                [&](int) { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }, stage, Y2, 0, ws, lane);
        }
        __syncthreads();
        // rf_post's 1x1 conv is folded into decoder layer 0's 1x1 on the host (fe_api.hip::pack_weights): that layer reads Y2 as its
        // first K-segment.  Wy (= W0) takes the 1x1 outputs, Wx (= W1, whose first rows Y2 occupies until layer 0 has consumed it)
        // the k = 3 outputs.
        float* const Wx = W1;
        float* const Wy = W0;
        for (int i = tid; i < 2 * LDC; i += NTH) {                // Wy lay under the token arena: zero its halo rows 0 and F1 + 1
            const int r = i / LDC, c = i - r * LDC;
            Wy[(r ? F1 + 1 : 0) * LDC + c] = 0.0f;
        }
        if (a.dbg != nullptr) {                                   // the rf_post stage no longer exists: recompute it for the dump
            float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(4 + S::NL + 2 * S::KB);
            for (int i = tid; i < F1 * C1; i += NTH) {
                const int f = i / C1, n = i - f * C1;
                float v = wp[o.rfpost_b + n];
                for (int kk = 0; kk < C2; ++kk) v += Y2[f * LDX + kk] * wp[o.rfpost_w + n * C2 + kk];
                dst[i] = v;
            }
        }

        FE_CLK(8);
        // =========================== decoder (a14) ===========================
        static_for<S::NL>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            const float* skip = Ebuf + (S::NL - l) * S::ACT;
            {
                // 1x1 conv on cat([x, skip]): two K-segments, never materialised
                FE8_BEGIN_UNIT(S::U_DEC + l * 2);
                constexpr bool FOLD0 = (l == 0);
                constexpr int K0 = FOLD0 ? S::KS_2 : S::KS_C;
                const float* xa = FOLD0 ? Y2 + (16 * ws + li) * LDX + lg : Wx + (16 * ws + li + 1) * LDC + lg;
                const float* sk = skip + (16 * ws + li + 1) * LDC + lg;
                conv8_pair<S, W8::NTA, W8::NTB, K0 + S::KS_C, C1, LDC, true>(
                    wh, [&](int ks) { return ks < K0 ? xa[4 * ks] : sk[4 * (ks - K0)]; },
                    [&](int j, int ks) { return wb.at(o.dec1_w[l] + (j * (K0 + S::KS_C) + ks) * 64); },
                    [&](int j) { return wb.at16x4(o.dec1_b[l] + j * 64); }, stage, Wy, 1, ws, lane);
            }
            __syncthreads();
            {
                FE8_BEGIN_UNIT(S::U_DEC + l * 2 + 1);
                const float* const t0 = Wy + (16 * ws + li) * LDC + lg;
                conv8_pair<S, W8::NTA, W8::NTB, 3 * S::KS_C, C1, LDC, true>(
                    wh, [&](int ks) { return t0[(ks / S::KS_C) * LDC + 4 * (ks % S::KS_C)]; },
                    [&](int j, int ks) { return wb.at(o.dec3_w[l] + (j * (3 * S::KS_C) + ks) * 64); },
                    [&](int j) { return wb.at16x4(o.dec3_b[l] + j * 64); }, stage, Wx, 1, ws, lane);
            }
            __syncthreads();
            dbg_dump<S, NTH>(a, b, 5 + S::NL + 2 * S::KB + l, Wx + LDC, LDC);
        });

        FE_CLK(9);
        // =========================== dec_post (a15) ===========================
        float* PT = smem + L::PT;
        typename Dft<S>::InvConst idc;
        constexpr bool FUSEP = FE_WG8_FUSEPOST && !PERSIST;
        float mb0 = 0.0f, mb1 = 0.0f;                  // the mask's two biases (FUSEP: the transposed conv's unit is not staged)
        if constexpr (FUSEP) {
            FE8_BEGIN_UNIT(S::U_POST);
            (void)stage;
            float wt[S::KS_C];
            if (wh == 0) {
#pragma unroll
                for (int ks = 0; ks < S::KS_C; ++ks) wt[ks] = wb.at_g(o.post_t_w + ks * 64);
            }
            mb0 = wp[o.post_t_b];
            mb1 = wp[o.post_t_b + 1];
            {
                const float* xa = Wx + (16 * ws + li + 1) * LDC + lg;
                const float* sk = Ebuf + (16 * ws + li + 1) * LDC + lg;
                conv8_pair<S, W8::NTA, W8::NTB, 2 * S::KS_C, C1, LDC, true>(
                    wh, [&](int ks) { return ks < S::KS_C ? xa[4 * ks] : sk[4 * (ks - S::KS_C)]; },
                    [&](int j, int ks) { return wb.at(o.post1_w + (j * (2 * S::KS_C) + ks) * 64); },
                    [&](int j) { return wb.at16x4(o.post1_b + j * 64); }, NoSide{}, Wy, 1, ws, lane);
            }
            int* pflag = reinterpret_cast<int*>(smem + W8::MXF) + 4 + ws;
            const int seq = fc + 1;
            if (wh == 1) {
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's tile is in LDS
                if (lane == 0) __hip_atomic_store(pflag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
            } else {
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(pflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < seq) {}
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                // transposed conv as a GEMM: P[i][co * 8 + j] = sum_ci x[i][ci] w[ci][co][j]   (rows of tile ws; a wave's own LDS writes are visible to it in order)
                const float* xa = Wy + (16 * ws + li + 1) * LDC + lg;
                float av[S::KS_C];
#pragma unroll
                for (int ks = 0; ks < S::KS_C; ++ks) av[ks] = xa[4 * ks];
                f32x4 pacc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int ks = 0; ks < S::KS_C; ++ks) pacc = FE_MFMA(av[ks], wt[ks], pacc);
#pragma unroll
                for (int r = 0; r < 4; ++r) PT[(16 * ws + 4 * lg + r) * S::LDP + li] = pacc[r];
                Dft<S>::load(idc, wb, o, wave);               // iSTFT constants, in flight during the mask phase
            }
        } else {
        {
            FE8_BEGIN_UNIT(S::U_POST);
            const float* xa = Wx + (16 * ws + li + 1) * LDC + lg;
            const float* sk = Ebuf + (16 * ws + li + 1) * LDC + lg;
            conv8_pair<S, W8::NTA, W8::NTB, 2 * S::KS_C, C1, LDC, true>(
                wh, [&](int ks) { return ks < S::KS_C ? xa[4 * ks] : sk[4 * (ks - S::KS_C)]; },
                [&](int j, int ks) { return wb.at(o.post1_w + (j * (2 * S::KS_C) + ks) * 64); },
                [&](int j) { return wb.at16x4(o.post1_b + j * 64); }, stage, Wy, 1, ws, lane);
        }
        __syncthreads();
        {
            // transposed conv as a GEMM: P[i][co * 8 + j] = sum_ci x[i][ci] w[ci][co][j]   (one channel tile: the wh = 0 waves)
            FE8_BEGIN_UNIT(S::U_POST + 1);
            const float* xa = Wy + (16 * ws + li + 1) * LDC + lg;
            conv8_pair<S, 1, 0, S::KS_C, 16, S::LDP, false>(
                wh, [&](int ks) { return xa[4 * ks]; }, [&](int j, int ks) { return wb.at(o.post_t_w + (j * S::KS_C + ks) * 64); },
                [&](int) { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }, stage, PT, 0, ws, lane);
            if (wave < 4) Dft<S>::load(idc, wb, o, wave);         // iSTFT constants, in flight during the mask phase
            mb0 = wb.scalar(o.post_t_b);
            mb1 = wb.scalar(o.post_t_b + 1);
        }
        }
        __syncthreads();

        FE_CLK(10);
        // =========================== mask, un-compress (a16, a17) ===========================
        {
            const float b0 = mb0, b1 = mb1;
            for (int f = tid; f < F0; f += NTH) {
                const int q = f + 2, j1 = q & 3, i1 = q >> 2;
                float m0 = b0, m1 = b1;
                if (i1 < F1) { m0 += PT[i1 * S::LDP + j1]; m1 += PT[i1 * S::LDP + 8 + j1]; }
                if (i1 >= 1) { m0 += PT[(i1 - 1) * S::LDP + j1 + 4]; m1 += PT[(i1 - 1) * S::LDP + 8 + j1 + 4]; }
                const float xr_ = sc[2 + f], xi_ = sc[S::LDS_S + 2 + f];
                float yr = xr_ * m0 - xi_ * m1;
                float yi = xr_ * m1 + xi_ * m0;
                if (a.dbg) {
                    float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(5 + 2 * S::NL + 2 * S::KB);
                    dst[2 * f] = m0; dst[2 * f + 1] = m1;
                }
                const float mag = sqrtf(yr * yr + yi * yi);
                const float g = pow_f(mag, 1.0f / a.compression - 1.0f);
                yr *= g; yi *= g;
                if (a.dbg) {
                    float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(6 + 2 * S::NL + 2 * S::KB);
                    dst[2 * f] = yr; dst[2 * f + 1] = yi;
                    if (f == 0) { dst[2 * F0] = 0.0f; dst[2 * F0 + 1] = 0.0f; }
                }
                q3[f] = yr;
                q3[N / 2 + f] = yi;
            }
        }
        __syncthreads();

        FE_CLK(11);
        // =========================== iSTFT (a18) ===========================
        {
            // synthesis window w / sum_k w^2 and the overlap tail: fetched across the inverse DFT
            const float ow = wp[o.window_istft + tid];
            const float oc = tid < OVL ? cis[tid] : 0.0f;
            if (wave < 4) Dft<S>::template inverse<WSrc<true>, false>(q3, q0, q1, tw, idc, wb, o, wave, lane);
            __syncthreads();
            FE_CLK(12);
            const int pi = Dft<S>::pidx(tid & (Dft<S>::N1 - 1), tid / Dft<S>::N1);
            const float xo = (q0[pi] + q1[pi]) * ow + oc;
            // one sample per thread: the first H go out, the rest is the new overlap tail (the old tail was read before the barrier)
            if (tid < H) a.wav_out[(size_t)b * a.out_stride + tid] = xo;
            else cis[tid - H] = xo;
            if constexpr (W8::HSTASH) {       // the frame's new GRU states: [KB][B][F2 * C2], 16 bytes per thread and store
                constexpr int N4 = F2 * C2 / 4;
                static_assert((F2 * C2) % 4 == 0 && W8::HST % 4 == 0, "16-byte state rows");
                const f32x4* hst = reinterpret_cast<const f32x4*>(smem + W8::HST);
#pragma unroll
                for (int k = 0; k < S::KB; ++k) {
                    f32x4* dst = reinterpret_cast<f32x4*>(a.h + ((size_t)k * a.B + b) * (F2 * C2));
                    for (int e = tid; e < N4; e += NTH) dst[e] = hst[k * N4 + e];
                }
            }
            if constexpr (PERSIST) __syncthreads();               // (the next stream's frame load reuses q0)
        }
        FE_CLK(13);
        b += (int)gridDim.x;
        ++fc;
    } while (PERSIST && b < a.B);
#undef FE8_BEGIN_UNIT
#undef FE8_B
#undef FE8_A
#undef FE8_STAGE_ON
    FE_CLK(63);
}

}  // namespace fe
