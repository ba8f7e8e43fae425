// lisennet_kernels.hip.h — the LiSenNet baseline model (models/lisennet/model.py of the reference, configs/others/lisennet.yaml) as
// one fused per-frame kernel for gfx950: STFT -> compress -> magnitude / group-delay / instantaneous-frequency-deviation features
// -> encoder (1x1 + three causal two-frame DSConvs, each with a (channel, freq) LayerNorm and PReLU) -> 2 x DPR (LayerNorm,
// bidirectional GRU over the 32 sub-bands, LayerNorm, GRU over time, ConvolutionalGLU with a causal three-frame depthwise conv and
// Mish) -> MaskDecoder (three sub-pixel up-convolutions with skips, causal two-frame mask conv, LayerNorm, PReLU, learnable sigmoid)
// -> complex mask -> un-compress -> iSTFT.  SURVEY.md §8(f) rank 4.
//
// Like FSPEN it is a tiny M = 1 model (one workgroup per stream, activations in LDS, VALU dot products over k-major packed
// weights, a wave per direction for the intra GRU) - latency-bound, no MFMA.  The frame caches of the causal convs, the inter-GRU
// states and the previous frame's phase are the model's caches, laid out per cache tensor as the reference's
// (ONNXModel.initialize_cache, :380-396, sized for B streams).
#pragma once
#include "fspen_kernels.hip.h"

namespace fe {

template <int HOP_>
struct LShape {
    static constexpr int HOP = HOP_, NFFT = 512, LOG2N = 9, OVL = NFFT - HOP, BINS = 257;
    static constexpr int C = 16, C1 = 4, C2 = 8, C3 = 12, NB = 2, HD = 24, HI = 12, NF = 32;     // channels, blocks, GRU sizes, sub-bands
    // per-stream cache sizes in the order of the reference's cache list
    static constexpr int K_PHA = 257, K_E2 = C1 * 257, K_E3 = C2 * 128, K_E4 = C3 * 64, K_H = NF * HD, K_GLU = 2 * C * 2 * NF, K_DEC = C1 * 256;
    static constexpr int CACHE_FLOATS = K_PHA + K_E2 + K_E3 + K_E4 + NB * (K_H + K_GLU) + K_DEC;
};

// ---- packed weights (floats); conv weights k-major: [(c, dt, df)][out]
struct LPk {
    static constexpr int WINDOW = 0, WINDOW_I = 512, TW = 1024;
    static constexpr int C1_W = 1536;                    // conv_1: [c < 3][4]
    static constexpr int C1_B = C1_W + 12, C1_G = C1_B + 4, C1_BE = C1_G + 260, C1_P = C1_BE + 260;      // bias, gamma[257], beta[257], PReLU[4]
    // DSConv i (cin -> cout, half = output bins per branch): low [(c*2+dt)*3+df][cout], high [(c*2+dt)*5+k][cout], biases, gamma / beta [2 half], PReLU
    static constexpr int D2 = C1_P + 4;
    static constexpr int D2_LO = D2, D2_HI = D2_LO + 4 * 6 * 8, D2_BL = D2_HI + 4 * 10 * 8, D2_BH = D2_BL + 8, D2_G = D2_BH + 8, D2_BE = D2_G + 128, D2_P = D2_BE + 128;
    static constexpr int D3 = D2_P + 8;
    static constexpr int D3_LO = D3, D3_HI = D3_LO + 8 * 6 * 12, D3_BL = D3_HI + 8 * 10 * 12, D3_BH = D3_BL + 12, D3_G = D3_BH + 12, D3_BE = D3_G + 64, D3_P = D3_BE + 64;
    static constexpr int D4 = D3_P + 12;
    static constexpr int D4_LO = D4, D4_HI = D4_LO + 12 * 6 * 16, D4_BL = D4_HI + 12 * 10 * 16, D4_BH = D4_BL + 16, D4_G = D4_BH + 16, D4_BE = D4_G + 32, D4_P = D4_BE + 32;
    static constexpr int BLK = D4_P + 16;
    // per DPR block
    static constexpr int B_N1W = 0, B_N1B = 512;                                 // intra_norm [f][d]
    static constexpr int B_IH = 1024;                                            // intra W_ih^T [dir][k < 16][36]
    static constexpr int B_GB = B_IH + 2 * 16 * 36;                              // [dir][36]: b_ih + (b_hh for r, z)
    static constexpr int B_HH = B_GB + 72;                                       // intra W_hh [dir][gate][k < 12][12 units]
    static constexpr int B_HN = B_HH + 2 * 3 * 12 * 12;                          // [dir][12] b_hh of n
    static constexpr int B_D1W = B_HN + 24, B_D1B = B_D1W + 24 * 16;             // intra dense^T [k < 24][16]
    static constexpr int B_N2W = B_D1B + 16, B_N2B = B_N2W + 512;                // inter_norm
    static constexpr int B_XIH = B_N2B + 512;                                    // inter W_ih^T [k < 16][72]
    static constexpr int B_XHH = B_XIH + 16 * 72;                                // inter W_hh^T [k < 24][72]
    static constexpr int B_XGB = B_XHH + 24 * 72, B_XHN = B_XGB + 72;            // [72], [24]
    static constexpr int B_D2W = B_XHN + 24, B_D2B = B_D2W + 24 * 16;            // inter dense^T [k < 24][16]
    static constexpr int B_GG = B_D2B + 16, B_GBE = B_GG + 512;                  // conv_glu norm gamma / beta [c][f]
    static constexpr int B_F1W = B_GBE + 512, B_F1B = B_F1W + 16 * 64;           // fc1^T [d < 16][64]
    static constexpr int B_DW = B_F1B + 64, B_DWB = B_DW + 9 * 32;               // dwconv [(dt*3+df)][32 ch]
    static constexpr int B_F2W = B_DWB + 32, B_F2B = B_F2W + 32 * 16;            // fc2^T [ch < 32][16]
    static constexpr int B_SIZE = (B_F2B + 16 + 3) / 4 * 4;
    // decoder: USConv i (cin -> cout): low [(c*3+df)][cout], high [(c*3+df)][3*cout], biases
    static constexpr int U1 = BLK + 2 * B_SIZE;
    static constexpr int U1_LO = U1, U1_HI = U1_LO + 32 * 3 * 12, U1_BL = U1_HI + 32 * 3 * 36, U1_BH = U1_BL + 12;
    static constexpr int U2 = U1_BH + 36;
    static constexpr int U2_LO = U2, U2_HI = U2_LO + 24 * 3 * 8, U2_BL = U2_HI + 24 * 3 * 24, U2_BH = U2_BL + 8;
    static constexpr int U3 = U2_BH + 24;
    static constexpr int U3_LO = U3, U3_HI = U3_LO + 16 * 3 * 4, U3_BL = U3_HI + 16 * 3 * 12, U3_BH = U3_BL + 4;
    static constexpr int M0_W = U3_BH + 12;                                      // mask_conv.0 [(c*2+dt)*2+df][2]
    static constexpr int M0_B = M0_W + 32, M_G = M0_B + 2, M_BE = M_G + 260, M_P = M_BE + 260;
    static constexpr int M3_W = M_P + 2, M3_B = M3_W + 4, SLOPE = M3_B + 2;       // mask_conv.3 [c][o], bias, lsigmoid slope [257]
    static constexpr int TOTAL = (SLOPE + 260 + 3) / 4 * 4;
};

struct LArgs {
    const float* wp;
    const float* wav_in;
    float* wav_out;
    size_t in_stride, out_stride;
    float* cache_stft;
    float* cache_istft;
    float* cache;             // the model caches, cache-major: [pha B x 257][enc2 B x 4 x 257][enc3 B x 8 x 128][enc4 B x 12 x 64]
                              //   { [h B*32 x 24][glu B x 32 x 2 x 32] } x 2  [dec B x 4 x 256]
    const float* spec_in;
    float* spec_out;
    float* dbg;
    size_t dbg_stride;
    int B, T, mode, Tw;
    float compression;
    unsigned long long* clk;
    // time-pipelined offline launch (PIPE instantiation): pipe_p workgroups per utterance, workgroup p runs frames p, p + pipe_p, ...
    unsigned int* pipe_flags; // [B][5 + 2 n_blocks]: per cache, the number of frames that have published it
    float* ring;              // [B][pipe_p + 2][CACHE_FLOATS]: the caches of the frames in flight (one slot per frame, cache layout)
    float* frames;            // [B][T][N] windowed output frames (summed / envelope-normalised by istft_ola_kernel)
    int pipe_p;
    float* carry;             // r6, the three-launch per-hop step of large batches (lisennet_sb_kernels.hip.h): [tiles][LCarry::TILE] [B][LCarry::SP]
};

// debug stages: 0 spec_in [257][2], 1 compressed [257][2], 2 features [3][257], 3 encoder.conv_1 [4][257], 4 encoder.conv_2 [8][128],
// 5 encoder.conv_3 [12][64], 6 encoder.conv_4 [16][32], 7+3b blocks.b.intra [32][16], 8+3b blocks.b.inter [32][16], 9+3b blocks.b [16][32],
// 13 decoder.up3 [4][256], 14 mask [257][2], 15 spec_out [257][2]
struct LDebugLayout {
    static constexpr int n_stages = 16;
    __host__ __device__ static constexpr int rows(int s) {
        return (s <= 1 || s >= 14) ? 257 : s == 2 ? 3 : s == 3 ? 4 : s == 4 ? 8 : s == 5 ? 12 : s == 6 ? 16 : s == 13 ? 4 : ((s - 7) % 3 == 2 ? 16 : 32);
    }
    __host__ __device__ static constexpr int cols(int s) {
        return (s <= 1 || s >= 14) ? 2 : (s == 2 || s == 3) ? 257 : s == 4 ? 128 : s == 5 ? 64 : s == 6 ? 32 : s == 13 ? 256 : ((s - 7) % 3 == 2 ? 32 : 16);
    }
    __host__ __device__ static constexpr size_t offset(int s) {
        size_t o = 0;
        for (int i = 0; i < s; ++i) o += (size_t)rows(i) * cols(i);
        return o;
    }
    __host__ __device__ static constexpr size_t total() { return offset(n_stages); }
};

// PART (r6): the plan of lisennet_frame_kernel<.., PART>: 0 = the whole frame; the two per-stream parts of the three-launch step keep only what they
// touch (PART 1: STFT + features - 16.5 KB, eight workgroups per CU; PART 2: mask + iSTFT - 14.4 KB)
template <int PART>
struct LLdsT {
    static constexpr int SP = 0;                    // compressed spectrum [257][2]
    static constexpr int TW = SP + 516;
    static constexpr int X2 = TW + 512;             // encoder.conv_2 out [8][128]   (skip of up3)
    static constexpr int X3 = X2 + (PART == 0 ? 1024 : 0);            // encoder.conv_3 out [12][64]   (skip of up2)
    static constexpr int X4 = X3 + (PART == 0 ? 768 : 0);             // encoder.conv_4 out [16][32]   (skip of up1)
    static constexpr int RED = X4 + (PART == 0 ? 512 : 0);            // block-reduction slots [16]
    static constexpr int SB = RED + 16;             // ---- phase scratch
    static constexpr int FA = SB, FB = SB + 1024;
    static constexpr int FEAT = SB + 2048;          // features [3][260]  /  phases [260] behind them
    static constexpr int PHA = FEAT + 780;
    static constexpr int X1 = SB;                   // conv_1 out, current frame [4][260] (after the FFT buffers are dead)
    static constexpr int X1P = SB + 1040;           // previous frame (cache)
    static constexpr int XP = SB + (PART == 1 ? 2080 : 3100);            // previous-frame copy of the current DSConv's input [<= 12 x 64 .. 8 x 128 = 1024]
    static constexpr int Y = SB + 4200;             // DSConv pre-norm output [<= 1024]
    // DPR blocks
    static constexpr int XT = SB;                   // tokens [32 f][16 d]
    static constexpr int YN = SB + 512;             // normalised tokens
    static constexpr int GI = SB + 1024;            // intra input projections [2][32][36]
    static constexpr int HSEQ = SB + 3328;          // [32][24]
    static constexpr int HP = SB + 4096;            // inter GRU state [32][24]
    static constexpr int HN = SB + 4864;
    static constexpr int Z = SB + 512;              // conv_glu: normalised [16][32]  (aliases YN)
    static constexpr int XX = SB + 1024;            // fc1 first half, three frames [3][32 ch][32 f]
    static constexpr int V = SB + 4096;             // fc1 second half [32][32]
    static constexpr int G = SB + 5120;             // gated [32][32]
    static constexpr int YD = SB + 6144;            // block output [16 d][32 f]
    // decoder
    static constexpr int U1 = SB;                   // [12][64]
    static constexpr int U2 = SB + 768;             // [8][128]
    static constexpr int U3 = SB + (PART == 2 ? 0 : 1792);            // [4][256]
    static constexpr int U3P = SB + (PART == 2 ? 1024 : 2816);           // previous frame (cache) [4][256]
    static constexpr int MY = SB + (PART == 2 ? 2048 : 3840);            // mask conv out [2][260]  (PART 2: unused)
    static constexpr int MK = SB + (PART == 2 ? 2048 : 4360);            // mask [2][260]
    static constexpr int WST = SB + (PART == 1 ? 3104 : 6656);           // weight staging area of the conv phases (one layer's weights at a time; PART 1: conv_2's 792 floats)
    static constexpr int WST_SIZE = 4672;
    static constexpr int TOTAL = PART == 1 ? PHA + 264 : PART == 2 ? MK + 520 : WST + WST_SIZE;
    static_assert(SB % 2 == 0 && TW % 2 == 0, "float2 alignment");
    static_assert((size_t)TOTAL * 4 <= 64 * 1024, "static LDS");
};
using LLds = LLdsT<0>;

__device__ __forceinline__ float mish_f(float x) {
    const float sp_ = x > 20.0f ? x : log1pf(__expf(x));       // softplus (torch's threshold 20)
    return x * tanh_f(sp_);
}

#define LS_CLK(i) do { if constexpr (PROF) { if (blockIdx.x == 0 && threadIdx.x == 0) a.clk[(i)] = __builtin_readcyclecounter(); } } while (0)

}  // namespace fe
#include "lisennet_sb_kernels.hip.h"
namespace fe {

// PIPE: time pipelining of an offline launch.  A frame needs the previous frame's value of NINE caches (the phase, three encoder frames,
// per block the GRU state and the ConvGLU's two frames, the decoder frame) - each is read and replaced at ONE place of the frame.  The
// frames in flight keep their caches in a ring of pipe_p + 2 slots per utterance (slot = frame mod RS, cache layout); a site waits for
// the previous frame's count of that cache, reads slot t - 1 (the ConvGLU: t - 2 and t - 1), writes slot t and counts it - agent-scope
// accesses on both sides, one counter per (utterance, cache).
// PART (r6): 0 = the whole frame; 1 = STFT .. encoder.conv_2 of a per-hop step whose middle runs batched over the streams (lisennet_sb_kernel): x2, its
// cached frame and the compressed spectrum go to the carry; 2 = that step's tail (decoder cache, mask conv .. iSTFT) from the carry's up3 output.
template <class S, bool PROF, bool DBG, bool PIPE = false, int PART = 0>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu((PART == 0 || DBG) ? 2 : 8, (PART == 0 || DBG) ? 2 : 8))) lisennet_frame_kernel(LArgs a) {
    static_assert(!PIPE || (!PROF && !DBG), "the time-pipelined instantiation is the plain offline kernel");
    static_assert(PART == 0 || !PIPE, "the split step is a streaming step");
    __shared__ __attribute__((aligned(16))) float smem[LLdsT<PART>::TOTAL];
    using L = LLdsT<PART>;
    using P = LPk;
    constexpr int N = S::NFFT, H = S::HOP, OVL = S::OVL, BINS = S::BINS;
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const float* __restrict__ wp0 = a.wp;
    const int mode = a.mode, aT = a.T;
    float* sp = smem + L::SP;
    float2* tw = reinterpret_cast<float2*>(smem + L::TW);
    float2* fa = reinterpret_cast<float2*>(smem + L::FA);
    float2* fb = reinterpret_cast<float2*>(smem + L::FB);
    float* x2 = smem + L::X2;
    float* x3 = smem + L::X3;
    float* x4 = smem + L::X4;
    float* red = smem + L::RED;
    for (int i = tid0; i < N / 2; i += kThreads) tw[i] = reinterpret_cast<const float2*>(wp0 + P::TW)[i];
    __syncthreads();

    int b = PIPE ? (int)blockIdx.x / a.pipe_p : (int)blockIdx.x;
    const int t_first = PIPE ? (int)blockIdx.x - b * a.pipe_p : 0, t_step = PIPE ? a.pipe_p : 1;
    constexpr int NSITE = 5 + 2 * S::NB;
    auto ld_state = [](const float* p) -> float {
        if constexpr (PIPE) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *p;
    };
    auto st_state = [](float* p, float v) {
        if constexpr (PIPE) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = v;
    };
#pragma unroll 1
    do {
    float* dbg = DBG ? a.dbg + (size_t)b * a.dbg_stride : nullptr;
    auto dump = [&](int stage, auto&& f) {
        if constexpr (DBG) {
            const int rows = LDebugLayout::rows(stage), cols = LDebugLayout::cols(stage), n = rows * cols;
            float* dst = dbg + LDebugLayout::offset(stage);
            // full trip counts and a clamped index: only the store is predicated (no long partially-executed loop bodies)
            for (int i0_ = 0; i0_ < n; i0_ += kThreads) {
                const int i = i0_ + tid0, ic = i < n ? i : n - 1;
                const int r = ic / cols, c = ic - r * cols;
                const float v = f(r, c);
                if (i < n) dst[i] = v;
            }
        }
    };
    // per-stream cache pointers (cache-major state)
    auto cache_ptr = [&](int off_floats_per_stream_sum, int per_stream) -> float* {
        return a.cache + (size_t)off_floats_per_stream_sum * a.B + (size_t)b * per_stream;
    };

#pragma unroll 1
    for (int t = t_first; t < aT; t += t_step) {
        // PIPE: the ring slots of this utterance; cprev(off, back) = cache `off` as frame t - back left it, ccur(off) = where frame t leaves it
        const int RS = PIPE ? a.pipe_p + 2 : 1;
        float* const ringb = PIPE ? a.ring + (size_t)b * RS * S::CACHE_FLOATS : nullptr;
        auto cprev = [&](int off, int back) -> const float* { return ringb + (size_t)((t - back < 0 ? 0 : t - back) % RS) * S::CACHE_FLOATS + off; };
        auto ccur = [&](int off) -> float* { return ringb + (size_t)(t % RS) * S::CACHE_FLOATS + off; };
        unsigned int* const pflag = PIPE ? a.pipe_flags + (size_t)b * NSITE : nullptr;
        auto cwait = [&](int site) {          // frame t - 1 has left cache `site` (all threads call)
            if constexpr (PIPE) {
                if (t > 0) {
                    if (threadIdx.x == 0) {
                        int spins = 0;
                        while (__hip_atomic_load(pflag + site, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)t && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
                        if (spins >= (1 << 24)) __builtin_trap();      // (a producer that never shows up: fail the launch loudly, never run on stale state)
                    }
                    __syncthreads();
                }
            }
        };
        auto cpub = [&](int site) {           // this frame's value of cache `site` is in its slot (includes a barrier)
            if constexpr (PIPE) {
                __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0)
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(pflag + site, (unsigned int)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        // loop-variant zeros on the weight pointer and the thread index (see fspen_kernels.hip.h)
        int lz = 0, lzv = 0;
        asm volatile("" : "+s"(lz));
        asm volatile("" : "+v"(lzv));
        const float* __restrict__ wp = wp0 + lz;
        const int tid = tid0 + lzv;
        const int lane = tid & 63;
        int red_slot = 0;
        // sum over the workgroup (every thread gets it); two alternating slot sets, one barrier per call
        auto block_sum = [&](float v) -> float {
            v = wave_sum(v);
            float* r = red + 4 * red_slot;
            if (lane == 0) r[wave] = v;
            __syncthreads();
            red_slot ^= 1;
            return (r[0] + r[1]) + (r[2] + r[3]);
        };
        // LayerNorm statistics over n values, each thread contributing `cnt` values v[]: returns (mean, 1 / sqrt(var + eps)), two-pass
        auto ln_stats = [&](const float* v, int cnt, float inv_n, float& mean, float& rstd) {
            float s_ = 0.0f;
            for (int i = 0; i < cnt; ++i) s_ += v[i];
            mean = block_sum(s_) * inv_n;
            float q_ = 0.0f;
            for (int i = 0; i < cnt; ++i) { const float d = v[i] - mean; q_ = fmaf(d, d, q_); }
            rstd = 1.0f / sqrtf(block_sum(q_) * inv_n + 1.0e-5f);
        };

        // a conv layer's weights (a contiguous block of the packed buffer) -> LDS in one coalesced burst: the layers then read them
        // with LDS latency instead of an L2 round trip per tap (all threads call; ends with a barrier)
        float* wst = smem + L::WST;
        auto stage = [&](int base, int n) {
            for (int i = tid; i < n; i += kThreads) wst[i] = wp[base + i];
            __syncthreads();
        };

        LS_CLK(0);
        // ============================ STFT + compress + phase features (models/lisennet/model.py:441-456 / :512-524) ============================
        float* feat = smem + L::FEAT;
        float* phs = smem + L::PHA;
        float* x1 = smem + L::X1;
        float* x1p = smem + L::X1P;
        if constexpr (PART != 2) {
        if (mode != FE_MODE_SPEC) {
            const float* win = wp + P::WINDOW;
            float* cst = a.cache_stft + (size_t)b * OVL;
            if (mode == FE_MODE_STREAM) {
                const float* xin = a.wav_in + (size_t)b * a.in_stride + (size_t)t * H;
                for (int n = tid; n < N; n += kThreads) {
                    const float v = (n < OVL) ? cst[n] : xin[n - OVL];
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
            if constexpr (DBG) { for (int f = tid; f < BINS; f += kThreads) { dbg[2 * f] = Xf[f].x; dbg[2 * f + 1] = Xf[f].y; } }
            for (int f = tid; f < BINS; f += kThreads) {
                float re = Xf[f].x + 0.0f, im = Xf[f].y + 0.0f;           // (+ 0.0f: a -0.0 would turn atan2(0, -0.0) into pi)
                const float g = pow_f(fmaxf(sqrtf(re * re + im * im), 1.0e-5f), a.compression - 1.0f);
                re *= g; im *= g;
                sp[2 * f] = re; sp[2 * f + 1] = im;
            }
        } else {
            const float* si = a.spec_in + (size_t)b * BINS * aT * 2;
            for (int f = tid; f < BINS; f += kThreads) {
                float re = si[((size_t)f * aT + t) * 2], im = si[((size_t)f * aT + t) * 2 + 1];
                if constexpr (DBG) { dbg[2 * f] = re; dbg[2 * f + 1] = im; }
                const float g = pow_f(fmaxf(sqrtf(re * re + im * im), 1.0e-5f), a.compression - 1.0f);
                re *= g; im *= g;
                sp[2 * f] = re; sp[2 * f + 1] = im;
            }
        }
        __syncthreads();
        dump(1, [&](int r, int c) { return sp[2 * r + c]; });
        {
            float* cpha = cache_ptr(0, S::K_PHA);
            // offline: the previous frame's phase of frame 0 is the prepended zero (torch.diff(prepend = 0), :506-507); the work buffer is zeroed
            for (int f = tid; f < BINS; f += kThreads) phs[f] = atan2f(sp[2 * f + 1], sp[2 * f]);
            __syncthreads();
            const float sgn = mode == FE_MODE_OFFLINE ? 1.0f : -1.0f;        // Model: current - previous (torch.diff); ONNXModel: previous - current
            constexpr float kInvPi = 0.31830988618379067f;
            cwait(0);
            for (int f = tid; f < BINS; f += kThreads) {
                const float re = sp[2 * f], im = sp[2 * f + 1], ph = phs[f];
                const float dgd = sgn * (ph - (f > 0 ? phs[f - 1] : 0.0f));
                float pprev;
                if constexpr (PIPE) pprev = t > 0 ? ld_state(cprev(0, 1) + f) : 0.0f; else pprev = cpha[f];
                const float dif = sgn * (ph - pprev) - 6.283185307179586f * ((float)H / (float)N) * (float)f;
                feat[f] = sqrtf(re * re + im * im);
                feat[260 + f] = atan2f(sinf(dgd), cosf(dgd)) * kInvPi;
                feat[520 + f] = atan2f(sinf(dif), cosf(dif)) * kInvPi;
                if constexpr (PIPE) st_state(ccur(0) + f, ph); else cpha[f] = ph;
            }
        }
        if constexpr (PIPE) cpub(0); else
        __syncthreads();
        dump(2, [&](int r, int c) { return feat[r * 260 + c]; });

        LS_CLK(1);
        // ============================ encoder (Encoder.forward, :269-274) ============================
        if constexpr (PART == 0)
        {   // conv_1: 1x1 (3 -> 4), LayerNorm over (channel, freq) with a per-frequency affine, PReLU   (FFT buffers are dead: x1 aliases them)
            float v[5];
            int cnt = 0;
            for (int i = tid; i < 4 * BINS; i += kThreads, ++cnt) {
                const int o = i / BINS, f = i - o * BINS;
                float acc = wp[P::C1_B + o];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc = fmaf(wp[P::C1_W + c * 4 + o], feat[c * 260 + f], acc);
                v[cnt] = acc;
            }
            float mean, rstd;
            ln_stats(v, cnt, 1.0f / (4.0f * BINS), mean, rstd);
            float* c2 = cache_ptr(S::K_PHA, S::K_E2);
            cnt = 0;
            cwait(1);
            for (int i = tid; i < 4 * BINS; i += kThreads, ++cnt) {
                const int o = i / BINS, f = i - o * BINS;
                float y = (v[cnt] - mean) * rstd * wp[P::C1_G + f] + wp[P::C1_BE + f];
                y = y >= 0.0f ? y : y * wp[P::C1_P + o];
                if constexpr (PIPE) {
                    x1p[o * 260 + f] = t > 0 ? ld_state(cprev(S::K_PHA, 1) + i) : 0.0f;
                    st_state(ccur(S::K_PHA) + i, y);
                } else {
                x1p[o * 260 + f] = c2[i];           // previous frame (the cache) in, this frame out
                c2[i] = y;
                }
                x1[o * 260 + f] = y;
            }
        }
        if constexpr (PART == 0) {
        if constexpr (PIPE) cpub(1); else
        __syncthreads();
        dump(3, [&](int r, int c) { return x1[r * 260 + c]; });
        }
        }   // PART != 2
        // DSConv (:190-208): causal two-frame conv, the bins split into a low quarter (k 3, stride 1) and the rest (k 5, stride 3), both
        // zero padded by one bin AFTER the split; LayerNorm over (channel, freq), per-frequency affine, PReLU
        auto dsconv = [&](auto CIN_, auto COUT_, auto FIN_, const float* cur, const float* prev, int ld_in, float* out, float* cache_io, float* prev_out,
                          int w_lo, int w_hi, int b_lo, int b_hi, int g_, int be_, int p_, int site = -1, int site_off = 0) {
            constexpr int CIN = decltype(CIN_)::value, COUT = decltype(COUT_)::value, FIN = decltype(FIN_)::value;
            constexpr int LOWF = FIN / 4, HALF = LOWF, FO = 2 * HALF, NOUT = COUT * FO, PER = (NOUT + kThreads - 1) / kThreads;
            stage(w_lo, p_ + COUT - w_lo);                      // [low | high | biases | gamma | beta | PReLU] of this layer
            const float* ws = wst - w_lo;
            float v[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int i = tid + q * kThreads;
                const int o = i % COUT, f = i / COUT;            // consecutive threads: consecutive output channels (weights coalesced)
                float acc = 0.0f;
                if (i < NOUT) {
                    if (f < HALF) {
                        acc = ws[b_lo + o];
#pragma unroll 4
                        for (int c = 0; c < CIN; ++c)
#pragma unroll
                            for (int df = 0; df < 3; ++df) {
                                const int fi = f + df - 1;
                                const bool ok = fi >= 0 && fi < LOWF;
                                const int fc = ok ? fi : 0;
                                const float w0 = ws[w_lo + ((c * 2 + 0) * 3 + df) * COUT + o], w1 = ws[w_lo + ((c * 2 + 1) * 3 + df) * COUT + o];
                                const float s_ = fmaf(w0, prev[c * ld_in + fc], w1 * cur[c * ld_in + fc]);
                                acc += ok ? s_ : 0.0f;
                            }
                    } else {
                        const int j = f - HALF;
                        acc = ws[b_hi + o];
#pragma unroll 4
                        for (int c = 0; c < CIN; ++c)
#pragma unroll
                            for (int k = 0; k < 5; ++k) {
                                const int fi = 3 * j + k - 1;                 // index into the high slice [LOWF, FIN)
                                const bool ok = fi >= 0 && fi < FIN - LOWF;
                                const int fc = LOWF + (ok ? fi : 0);
                                const float w0 = ws[w_hi + ((c * 2 + 0) * 5 + k) * COUT + o], w1 = ws[w_hi + ((c * 2 + 1) * 5 + k) * COUT + o];
                                const float s_ = fmaf(w0, prev[c * ld_in + fc], w1 * cur[c * ld_in + fc]);
                                acc += ok ? s_ : 0.0f;
                            }
                    }
                }
                v[q] = acc;
            }
            float mean, rstd;
            ln_stats(v, PER, 1.0f / (float)NOUT, mean, rstd);          // (threads past NOUT contribute zeros: corrected below)
            constexpr int PADN = PER * kThreads - NOUT;                  // zero contributions: mean is exact, the variance needs - PADN * mean^2
            if constexpr (PADN > 0) {
                const float var = 1.0f / (rstd * rstd) - 1.0e-5f - (float)PADN / (float)NOUT * mean * mean;
                rstd = 1.0f / sqrtf(var + 1.0e-5f);
            }
            if (cache_io) cwait(site);
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int i = tid + q * kThreads;
                if (i < NOUT) {
                    const int o = i % COUT, f = i / COUT;
                    float y = (v[q] - mean) * rstd * ws[g_ + f] + ws[be_ + f];
                    y = y >= 0.0f ? y : y * ws[p_ + o];
                    out[o * FO + f] = y;
                    if (cache_io) {
                        if constexpr (PIPE) {
                            prev_out[o * FO + f] = t > 0 ? ld_state(cprev(site_off, 1) + o * FO + f) : 0.0f;
                            st_state(ccur(site_off) + o * FO + f, y);
                        } else { prev_out[o * FO + f] = cache_io[o * FO + f]; cache_io[o * FO + f] = y; }
                    }
                }
            }
            if (PIPE && cache_io) cpub(site); else
            __syncthreads();
        };
        using I4 = std::integral_constant<int, 4>;
        using I8 = std::integral_constant<int, 8>;
        using I12 = std::integral_constant<int, 12>;
        using I16 = std::integral_constant<int, 16>;
        using I64 = std::integral_constant<int, 64>;
        using I128 = std::integral_constant<int, 128>;
        using I257 = std::integral_constant<int, 257>;
        float* xp = smem + L::XP;
        constexpr int OFF_BLK = S::K_PHA + S::K_E2 + S::K_E3 + S::K_E4;
        if constexpr (PART == 0) {
        dsconv(I4{}, I8{}, I257{}, x1, x1p, 260, x2, cache_ptr(S::K_PHA + S::K_E2, S::K_E3), xp, P::D2_LO, P::D2_HI, P::D2_BL, P::D2_BH, P::D2_G, P::D2_BE, P::D2_P, 2, S::K_PHA + S::K_E2);
        dump(4, [&](int r, int c) { return x2[r * 128 + c]; });
        }
        if constexpr (PART == 1) {
            // the input features [3][260] and the compressed spectrum, per stream as they stand (coalesced): the middle's prologue regroups the features for its
            // sixteen streams and runs the encoder from conv_1 on (r6, first version: conv_1 / conv_2 here and x2 written scattered into the tiles' layout -
            // 3 072 partial-line writes per stream, 100 of this launch's 163 us at 4096 streams; then coalesced: 86 us; without the two convolutions: see DESIGN 3e)
            using A = LCarry;
            float* fn = a.carry + A::feat(a.B) + (size_t)b * A::FEAT;
            for (int i = tid; i < 3 * 260; i += kThreads) fn[i] = feat[i];
            float* spc = a.carry + (size_t)((a.B + 15) >> 4) * A::TILE + (size_t)b * A::SP;
            for (int i = tid; i < 2 * BINS; i += kThreads) spc[i] = sp[i];
        }
        if constexpr (PART == 0) {
        {
            float* xq = smem + L::Y;       // previous frame of x3's input is in xp (x2's cache); x3's own cache lands in xq
            dsconv(I8{}, I12{}, I128{}, x2, xp, 128, x3, cache_ptr(S::K_PHA + S::K_E2 + S::K_E3, S::K_E4), xq, P::D3_LO, P::D3_HI, P::D3_BL, P::D3_BH, P::D3_G, P::D3_BE, P::D3_P, 3, S::K_PHA + S::K_E2 + S::K_E3);
            dump(5, [&](int r, int c) { return x3[r * 64 + c]; });
            dsconv(I12{}, I16{}, I64{}, x3, xq, 64, x4, nullptr, nullptr, P::D4_LO, P::D4_HI, P::D4_BL, P::D4_BH, P::D4_G, P::D4_BE, P::D4_P);
            dump(6, [&](int r, int c) { return x4[r * 32 + c]; });
        }

        LS_CLK(2);
        // ============================ 2 x DPR (DPR.forward, :151-159) ============================
        float* xt = smem + L::XT;
        float* yn = smem + L::YN;
        float* gi = smem + L::GI;
        float* hseq = smem + L::HSEQ;
        float* hp = smem + L::HP;
        float* hn = smem + L::HN;
        float* yd = smem + L::YD;
        // tokens [f][d] of the block input (b, d, t, f) -> (b, t, f, d)
        for (int q = 0; q < 2; ++q) { const int i = tid + 256 * q, f = i >> 4, d = i & 15; xt[i] = x4[d * 32 + f]; }
        __syncthreads();
#pragma unroll 1
        for (int blk = 0; blk < S::NB; ++blk) {
            const float* wd = wp + P::BLK + blk * P::B_SIZE;
            float* ch = cache_ptr(OFF_BLK + blk * (S::K_H + S::K_GLU), S::K_H);                  // [32][24] rows b*32 + f
            float* cg = cache_ptr(OFF_BLK + blk * (S::K_H + S::K_GLU) + S::K_H, S::K_GLU);       // [32 ch][2][32 f]
            // inter GRU state -> LDS (in flight across the intra path)
            float hpre[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) hpre[q] = PIPE ? 0.0f : ch[tid + 256 * q];
            // ---- intra_norm: nn.LayerNorm((32, 16)) over the whole token matrix
            {
                float v[2] = {xt[tid], xt[tid + 256]};
                float mean, rstd;
                ln_stats(v, 2, 1.0f / 512.0f, mean, rstd);
#pragma unroll
                for (int q = 0; q < 2; ++q) { const int i = tid + 256 * q; yn[i] = (v[q] - mean) * rstd * wd[P::B_N1W + i] + wd[P::B_N1B + i]; }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) hp[tid + 256 * q] = hpre[q];
            __syncthreads();
            // ---- intra GRU input projections gi[dir][f][36]: thread (gate row g36, f group): its 2 x 16 weights in registers, 5 rows of tokens
            {
                const int g36 = tid % 36, fq = tid / 36;                 // 252 threads, fq < 7
                float wq[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) wq[k] = wd[P::B_IH + k * 36 + g36];
                const float bq0 = wd[P::B_GB + g36], bq1 = wd[P::B_GB + 36 + g36];
                const float gsc = g36 < 24 ? kGateRZ : kGateN;          // (scaled pre-activations: fspen_kernels.hip.h, row_dot)
                // all 256 threads run all five rows: a row index past the end is clamped to row 31 and threads 252.. repeat rows of group 0
                // (identical values stored twice) - no partially-executed region around register-heavy code (a VGPR spilled and
                // reloaded inside one loses its inactive lanes: see DESIGN.md)
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const int f = (fq + 7 * r) < 32 ? (fq + 7 * r) : 31;
                    float a0 = bq0, a1 = bq1;
#pragma unroll
                    for (int k = 0; k < 16; ++k) { const float xv = yn[f * 16 + k]; a0 = fmaf(wq[k], xv, a0); a1 = fmaf(wq[16 + k], xv, a1); }
                    gi[f * 36 + g36] = a0 * gsc;
                    gi[(32 + f) * 36 + g36] = a1 * gsc;
                }
            }
            // lane = (gate row = lane / 16: r, z, n, (r again), hidden unit c = lane % 16, 12 used): one gate row of 12 weights per lane
            float wg_[12];
            const int g_row = (lane >> 4) < 3 ? (lane >> 4) : 0;
            const int c12 = (lane & 15) < 12 ? (lane & 15) : 0;
            {
                const int dsel = wave & 1;
#pragma unroll
                for (int k = 0; k < 12; ++k) wg_[k] = wd[P::B_HH + ((dsel * 3 + g_row) * 12 + k) * 12 + c12] * (g_row == 2 ? kGateN : kGateRZ);
            }
            const float bhn = (lane >> 4) == 2 ? wd[P::B_HN + (wave & 1) * 12 + c12] * kGateN : 0.0f;
            __syncthreads();
            if (blk == 0) LS_CLK(6);
            if (wave < 2) {      // (gates one per lane, DPP row broadcasts of h, row-swap gather: see fspen_kernels.hip.h)
                const int d = wave;
                const bool is_n = (lane >> 4) == 2;
                float h = 0.0f;
                // running LDS offsets of the walk, the next step's x side fetched a step ahead (fspen_kernels.hip.h)
                const int f0 = d ? 31 : 0;
                int go = (d * 32 + f0) * 36 + g_row * 12 + c12, gn = (d * 32 + f0) * 36 + 24 + c12, ho = f0 * 24 + d * 12 + c12;
                const int gstep = d ? -36 : 36, hstep = d ? -24 : 24;
                float g_own = gi[go], g_n = gi[gn];
#pragma unroll 1
                for (int s_ = 0; s_ < 32; ++s_) {
                    go += gstep; gn += gstep;
                    const float n_own = gi[go], n_n = gi[gn];       // (after the last step: the other direction's rows, unused)
                    const float acc = row_dot(wg_, h, bhn);
                    const float x_ = is_n ? acc : sigmoid_pre(g_own + acc);
                    float r, z, pn;
                    rows_gather3(x_, r, z, pn);
                    const float n = tanh_pre(__builtin_fmaf(r, pn, g_n));
                    h = __builtin_fmaf(z, h - n, n);
                    hseq[ho] = h;                                   // (lanes 12..15 of a row shadow unit 0; the four rows hold the same h)
                    ho += hstep;
                    g_own = n_own; g_n = n_n;
                }
            }
            __syncthreads();
            if (blk == 0) LS_CLK(7);
            // ---- intra dense (24 -> 16) + residual
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = tid + 256 * q, f = i >> 4, d = i & 15;
                float acc = wd[P::B_D1B + d];
#pragma unroll
                for (int k = 0; k < 24; ++k) acc = fmaf(wd[P::B_D1W + k * 16 + d], hseq[f * 24 + k], acc);
                xt[i] += acc;
            }
            __syncthreads();
            dump(7 + 3 * blk, [&](int r, int c) { return xt[r * 16 + c]; });
            if (blk == 0) LS_CLK(8);
            // ---- inter_norm + inter GRU over time (one step; state [32][24]) + dense + residual
            {
                float v[2] = {xt[tid], xt[tid + 256]};
                float mean, rstd;
                ln_stats(v, 2, 1.0f / 512.0f, mean, rstd);
#pragma unroll
                for (int q = 0; q < 2; ++q) { const int i = tid + 256 * q; yn[i] = (v[q] - mean) * rstd * wd[P::B_N2W + i] + wd[P::B_N2B + i]; }
            }
            __syncthreads();
            if constexpr (PIPE) {       // frame t - 1's GRU state of this block -> hp
                cwait(4 + 2 * blk);
                const int hoff = OFF_BLK + blk * (S::K_H + S::K_GLU);
#pragma unroll
                for (int q = 0; q < 3; ++q) hp[tid + 256 * q] = t > 0 ? ld_state(cprev(hoff, 1) + tid + 256 * q) : 0.0f;
                __syncthreads();
            }
            {   // ONE GRU for all sub-bands: thread (hidden unit c, row group fg) keeps the unit's three gate rows (3 x (16 + 24) weights)
                // in registers and walks the rows f = fg, fg + 10, ...
                const int c = tid % 24, fg = tid / 24;                  // 240 threads, fg < 10
                float wi[48], wh[72];
#pragma unroll
                for (int k = 0; k < 16; ++k) { wi[k] = wd[P::B_XIH + k * 72 + c]; wi[16 + k] = wd[P::B_XIH + k * 72 + 24 + c]; wi[32 + k] = wd[P::B_XIH + k * 72 + 48 + c]; }
#pragma unroll
                for (int k = 0; k < 24; ++k) { wh[k] = wd[P::B_XHH + k * 72 + c]; wh[24 + k] = wd[P::B_XHH + k * 72 + 24 + c]; wh[48 + k] = wd[P::B_XHH + k * 72 + 48 + c]; }
                const float b_r = wd[P::B_XGB + c], b_z = wd[P::B_XGB + 24 + c], b_n = wd[P::B_XGB + 48 + c], b_hn = wd[P::B_XHN + c];
                // (uniform trip count, clamped row; threads 240.. repeat rows of group 0: identical values stored twice - see above)
#pragma unroll 1
                for (int r = 0; r < 4; ++r) {
                    const int f = (fg + 10 * r) < 32 ? (fg + 10 * r) : 31;
                    float ir = b_r, iz = b_z, in_ = b_n, hr = 0.0f, hz = 0.0f, hnn = b_hn;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float xv = yn[f * 16 + k];
                        ir = fmaf(wi[k], xv, ir);
                        iz = fmaf(wi[16 + k], xv, iz);
                        in_ = fmaf(wi[32 + k], xv, in_);
                    }
#pragma unroll
                    for (int k = 0; k < 24; ++k) {
                        const float hv = hp[f * 24 + k];
                        hr = fmaf(wh[k], hv, hr);
                        hz = fmaf(wh[24 + k], hv, hz);
                        hnn = fmaf(wh[48 + k], hv, hnn);
                    }
                    const float rg = sigmoid_f(ir + hr);
                    const float z = sigmoid_f(iz + hz);
                    const float n = tanh_f(in_ + rg * hnn);
                    const float hnew = (1.0f - z) * n + z * hp[f * 24 + c];
                    hn[f * 24 + c] = hnew;
                    if constexpr (PIPE) st_state(ccur(OFF_BLK + blk * (S::K_H + S::K_GLU)) + f * 24 + c, hnew); else
                    ch[f * 24 + c] = hnew;
                }
            }
            if constexpr (PIPE) cpub(4 + 2 * blk); else
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = tid + 256 * q, f = i >> 4, d = i & 15;
                float acc = wd[P::B_D2B + d];
#pragma unroll
                for (int k = 0; k < 24; ++k) acc = fmaf(wd[P::B_D2W + k * 16 + d], hn[f * 24 + k], acc);
                xt[i] += acc;
            }
            __syncthreads();
            dump(8 + 3 * blk, [&](int r, int c) { return xt[r * 16 + c]; });
            if (blk == 0) LS_CLK(9);
            // ---- ConvolutionalGLU (:120-136) on (b, d, t, f): CustomLayerNorm over (d, f) with gamma / beta [d][f]
            float* z = smem + L::Z;
            float* xx = smem + L::XX;
            float* vv = smem + L::V;
            float* gg = smem + L::G;
            {
                float v[2] = {xt[tid], xt[tid + 256]};
                float mean, rstd;
                ln_stats(v, 2, 1.0f / 512.0f, mean, rstd);
                // the previous two frames of fc1's first half (the cache) -> xx[0], xx[1]; in flight across fc1
                float cpre[8];
                if constexpr (PIPE) {       // the ConvGLU's frames t - 2 and t - 1: each frame leaves ITS frame in the first half of its slot's cache
                    cwait(5 + 2 * blk);
                    const int goff = OFF_BLK + blk * (S::K_H + S::K_GLU) + S::K_H;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int i = tid + 256 * q, chn = i >> 6, dt = (i >> 5) & 1, f = i & 31;
                        const int back = 2 - dt;                                    // dt = 0: frame t - 2, dt = 1: frame t - 1
                        cpre[q] = t - back >= 0 ? ld_state(cprev(goff, back) + chn * 32 + f) : 0.0f;
                    }
                } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) cpre[q] = cg[tid + 256 * q];
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = tid + 256 * q, f = i >> 4, d = i & 15;
                    z[d * 32 + f] = (v[q] - mean) * rstd * wd[P::B_GG + d * 32 + f] + wd[P::B_GBE + d * 32 + f];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int i = tid + 256 * q, chn = i >> 6, dt = (i >> 5) & 1, f = i & 31; xx[(dt * 32 + chn) * 32 + f] = cpre[q]; }
            }
            __syncthreads();
            // fc1: 1x1 (16 -> 64): channels 0..31 -> xx[2] (and the new cache), 32..63 -> v
#pragma unroll 2
            for (int q = 0; q < 8; ++q) {
                const int i = tid + 256 * q, o = i & 63, f = i >> 6;
                float acc = wd[P::B_F1B + o];
#pragma unroll
                for (int d = 0; d < 16; ++d) acc = fmaf(wd[P::B_F1W + d * 64 + o], z[d * 32 + f], acc);
                if (o < 32) xx[(64 + o) * 32 + f] = acc; else vv[(o - 32) * 32 + f] = acc;
            }
            __syncthreads();
            if (blk == 0) LS_CLK(10);
            // new cache = frames (t-1, t); depthwise 3 x 3 over (time, freq), Mish, gate
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                const int i = tid + 256 * q, chn = i >> 5, f = i & 31;
                float acc = wd[P::B_DWB + chn];
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                    for (int df = 0; df < 3; ++df) {
                        const int fi = f + df - 1;
                        const bool ok = fi >= 0 && fi < 32;
                        const float xv = xx[(dt * 32 + chn) * 32 + (ok ? fi : 0)];
                        acc = fmaf(wd[P::B_DW + (dt * 3 + df) * 32 + chn], ok ? xv : 0.0f, acc);
                    }
                gg[i] = mish_f(acc) * vv[i];
                if constexpr (PIPE) st_state(ccur(OFF_BLK + blk * (S::K_H + S::K_GLU) + S::K_H) + chn * 32 + f, xx[(64 + chn) * 32 + f]); else {
                cg[(chn * 2 + 0) * 32 + f] = xx[(32 + chn) * 32 + f];
                cg[(chn * 2 + 1) * 32 + f] = xx[(64 + chn) * 32 + f];
                }
            }
            if constexpr (PIPE) cpub(5 + 2 * blk); else
            __syncthreads();
            // fc2: 1x1 (32 -> 16) + residual -> block output (b, d, t, f) and the next block's tokens
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = tid + 256 * q, f = i >> 4, d = i & 15;
                float acc = wd[P::B_F2B + d];
#pragma unroll
                for (int c = 0; c < 32; ++c) acc = fmaf(wd[P::B_F2W + c * 16 + d], gg[c * 32 + f], acc);
                acc += xt[i];
                xt[i] = acc;
                yd[d * 32 + f] = acc;
            }
            __syncthreads();
            dump(9 + 3 * blk, [&](int r, int c) { return yd[r * 32 + c]; });
            if (blk == 0) LS_CLK(11);
        }

        LS_CLK(3);
        // ============================ MaskDecoder (:295-309) ============================
        // USConv (:218-226): input = cat(x, skip) over channels; low half of the bins: conv k 3; high half: conv k 3 to 3 x cout
        // channels, pixel-shuffled over frequency (SPConvTranspose2d, :240-246): out[c][3 w + r] = conv[r * cout + c][w]
        auto usconv = [&](auto CX_, auto COUT_, auto FIN_, const float* xa, const float* skip, float* out, int w_lo, int w_hi, int b_lo, int b_hi) {
            constexpr int CX = decltype(CX_)::value, COUT = decltype(COUT_)::value, FIN = decltype(FIN_)::value;
            constexpr int LOWF = FIN / 2, FO = LOWF + 3 * LOWF, NOUT = COUT * FO, CIN = 2 * CX;
            stage(w_lo, b_hi + 3 * COUT - w_lo);                // [low | high | low bias | high bias]
            const float* ws = wst - w_lo;
            for (int i = tid; i < NOUT; i += kThreads) {
                const int c = i % COUT, fo = i / COUT;
                const bool low = fo < LOWF;
                const int w = low ? fo : (fo - LOWF) / 3, r = low ? 0 : (fo - LOWF) - 3 * w;
                const int oc = low ? c : r * COUT + c, nout = low ? COUT : 3 * COUT;
                const int wofs = low ? w_lo : w_hi, base = low ? 0 : LOWF;
                float acc = ws[(low ? b_lo : b_hi) + oc];
#pragma unroll 8
                for (int ci = 0; ci < CIN; ++ci) {
                    const float* src = ci < CX ? xa + ci * FIN : skip + (ci - CX) * FIN;
#pragma unroll
                    for (int df = 0; df < 3; ++df) {
                        const int fi = w + df - 1;
                        const bool ok = fi >= 0 && fi < LOWF;
                        const float xv = src[base + (ok ? fi : 0)];
                        acc = fmaf(ws[wofs + (ci * 3 + df) * nout + oc], ok ? xv : 0.0f, acc);
                    }
                }
                out[c * FO + fo] = acc;
            }
            __syncthreads();
        };
        float* u1 = smem + L::U1;
        float* u2 = smem + L::U2;
        {   // yd lives at SB + 6144: u1 / u2 / u3 below it
            using I32 = std::integral_constant<int, 32>;
            usconv(I16{}, I12{}, I32{}, yd, x4, u1, P::U1_LO, P::U1_HI, P::U1_BL, P::U1_BH);
            usconv(I12{}, I8{}, I64{}, u1, x3, u2, P::U2_LO, P::U2_HI, P::U2_BL, P::U2_BH);
            usconv(I8{}, I4{}, I128{}, u2, x2, smem + L::U3, P::U3_LO, P::U3_HI, P::U3_BL, P::U3_BH);
        }
        dump(13, [&](int r, int c) { return smem[L::U3 + r * 256 + c]; });
        }   // PART == 0
        float* u3 = smem + L::U3;
        float* u3p = smem + L::U3P;
        float* my = smem + L::MY;
        float* mk = smem + L::MK;
        if constexpr (PART == 2) {      // the stream-batched middle's mask [257 f][16 n][2] and PART 1's compressed spectrum
            using A = LCarry;
            const float* cm = a.carry + (size_t)(b >> 4) * A::TILE + A::MK + (b & 15) * 2;
            for (int f = tid; f < BINS; f += kThreads) {
                const float2 v = *reinterpret_cast<const float2*>(cm + f * 32);
                mk[f] = v.x; mk[260 + f] = v.y;
            }
            const float* spc = a.carry + (size_t)((a.B + 15) >> 4) * A::TILE + (size_t)b * A::SP;
            for (int i = tid; i < 2 * BINS; i += kThreads) sp[i] = spc[i];
            __syncthreads();
        }
        if constexpr (PART != 1) {
        if constexpr (PART == 0)
        {
            float* cd = cache_ptr(OFF_BLK + S::NB * (S::K_H + S::K_GLU), S::K_DEC);
            if constexpr (PIPE) {
                cwait(4 + 2 * S::NB);
                const int doff = OFF_BLK + S::NB * (S::K_H + S::K_GLU);
                for (int i = tid; i < 4 * 256; i += kThreads) { u3p[i] = t > 0 ? ld_state(cprev(doff, 1) + i) : 0.0f; st_state(ccur(doff) + i, u3[i]); }
                cpub(4 + 2 * S::NB);
            } else {
            for (int i = tid; i < 4 * 256; i += kThreads) { u3p[i] = cd[i]; cd[i] = u3[i]; }
            __syncthreads();
            }
            // mask_conv.0: Conv2d(4 -> 2, (2, 2), padding (0, 1)) over (previous, current) frame: 257 output bins
            float v[3];
            int cnt = 0;
            for (int i = tid; i < 2 * BINS; i += kThreads, ++cnt) {
                const int o = i & 1, f = i >> 1;
                float acc = wp[P::M0_B + o];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int df = 0; df < 2; ++df) {
                        const int fi = f + df - 1;
                        const bool ok = fi >= 0 && fi < 256;
                        const int fc = ok ? fi : 0;
                        const float s_ = fmaf(wp[P::M0_W + ((c * 2 + 0) * 2 + df) * 2 + o], u3p[c * 256 + fc], wp[P::M0_W + ((c * 2 + 1) * 2 + df) * 2 + o] * u3[c * 256 + fc]);
                        acc += ok ? s_ : 0.0f;
                    }
                v[cnt] = acc;
            }
            float mean, rstd;
            ln_stats(v, cnt, 1.0f / (2.0f * BINS), mean, rstd);
            cnt = 0;
            for (int i = tid; i < 2 * BINS; i += kThreads, ++cnt) {
                const int o = i & 1, f = i >> 1;
                float y = (v[cnt] - mean) * rstd * wp[P::M_G + f] + wp[P::M_BE + f];
                my[o * 260 + f] = y >= 0.0f ? y : y * wp[P::M_P + o];
            }
            __syncthreads();
            for (int i = tid; i < 2 * BINS; i += kThreads) {
                const int o = i & 1, f = i >> 1;
                const float y = wp[P::M3_B + o] + wp[P::M3_W + 0 * 2 + o] * my[f] + wp[P::M3_W + 1 * 2 + o] * my[260 + f];
                mk[o * 260 + f] = sigmoid_f(wp[P::SLOPE + f] * y);
            }
            __syncthreads();
        }
        dump(14, [&](int r, int c) { return mk[c * 260 + r]; });

        LS_CLK(4);
        // ============================ mask, un-compress, iSTFT ============================
        {
            float* spo = mode == FE_MODE_STREAM ? nullptr : a.spec_out + (size_t)b * BINS * aT * 2;
            for (int f = tid; f < BINS; f += kThreads) {
                const float sr = sp[2 * f], si = sp[2 * f + 1], mr = mk[f], mi = mk[260 + f];
                float yr = sr * mr - si * mi, yi = sr * mi + si * mr;
                if (mode == FE_MODE_OFFLINE) {          // Model.forward returns the compressed spec_hat
                    spo[((size_t)f * aT + t) * 2] = yr;
                    spo[((size_t)f * aT + t) * 2 + 1] = yi;
                }
                const float g = pow_f(sqrtf(yr * yr + yi * yi), 1.0f / a.compression - 1.0f);
                yr *= g; yi *= g;
                if constexpr (DBG) { float* d = dbg + LDebugLayout::offset(15); d[2 * f] = yr; d[2 * f + 1] = yi; }
                if (mode == FE_MODE_SPEC) {
                    spo[((size_t)f * aT + t) * 2] = yr;
                    spo[((size_t)f * aT + t) * 2 + 1] = yi;
                } else if (f == 0) {
                    fa[0] = make_float2(yr, 0.0f);
                } else if (f == N / 2) {
                    fa[N / 2] = make_float2(yr, 0.0f);
                } else {
                    fa[f] = make_float2(yr, yi);
                    fa[N - f] = make_float2(yr, -yi);
                }
            }
        }
        __syncthreads();
        if (mode != FE_MODE_SPEC) {
            float2* yv = fft_lds<S, true>(fa, fb, tw);
            float2* spare = (yv == fa) ? fb : fa;
            float* cis = a.cache_istft + (size_t)b * OVL;
            const float* wi = wp + (mode == FE_MODE_STREAM ? P::WINDOW_I : P::WINDOW);
            float* xo = reinterpret_cast<float*>(spare);
            const float invN = 1.0f / (float)N;
            if constexpr (PIPE) {
                float* fr = a.frames + ((size_t)b * aT + t) * N;
                for (int n = tid; n < N; n += kThreads) fr[n] = yv[n].x * invN * wi[n];
                __syncthreads();
            } else {
            for (int n = tid; n < N; n += kThreads) {
                float v = yv[n].x * invN * wi[n];
                if (n < OVL) v += cis[n];
                xo[n] = v;
            }
            __syncthreads();
            if (mode == FE_MODE_STREAM) {
                float* out = a.wav_out + (size_t)b * a.out_stride + (size_t)t * H;
                for (int n = tid; n < H; n += kThreads) out[n] = xo[n];
            } else {
                const float* w = wp + P::WINDOW;
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
        }
        }   // PART != 1
        LS_CLK(5);
    }
    if constexpr (PIPE) break;
    b += gridDim.x;
    } while (b < a.B);
}

struct LImpl {
    int HOP;
    size_t lds_bytes;
    size_t dbg_floats;
    int dbg_stages;
    size_t packed_floats;
    size_t cache_floats;      // model caches per stream
    void (*launch)(const LArgs&, int max_wgs, hipStream_t, hipError_t*);
    void (*dbg_stage)(int, int*, int*, size_t*);
    void (*launch_pipe)(const LArgs&, hipStream_t, hipError_t*);       // time-pipelined offline launch (cooperative: B * pipe_p workgroups)
    int occ;                  // workgroups per CU
    int nsite;                // caches handed from frame to frame (counters per stream)
    void (*launch_sb)(const LArgs&, int max_wgs, hipStream_t, hipError_t*);      // r6: the per-hop step of a large batch in three launches, the middle batched over the streams
};

template <class S>
void llaunch_pipe_impl(const LArgs& a, hipStream_t st, hipError_t* err) {
    LArgs args = a;
    void* kargs[] = {&args};
    note_kernel("lisennet_frame_kernel<time-pipelined>");
    *err = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&lisennet_frame_kernel<S, false, false, true>), dim3(a.B * a.pipe_p), dim3(kThreads), kargs, 0, st);
}

template <class S>
void llaunch_impl(const LArgs& a, int max_wgs, hipStream_t st, hipError_t* err) {
    constexpr int OCC_LDS = (160 * 1024) / (LLds::TOTAL * 4);
    constexpr int OCC = OCC_LDS < 2 ? OCC_LDS : 2;                 // (256 VGPRs per wave: two workgroups per CU)
    const int slots = max_wgs * OCC;
    const int grid = a.B < slots ? a.B : slots;
    note_kernel(a.dbg != nullptr ? "lisennet_frame_kernel<debug>" : a.clk != nullptr ? "lisennet_frame_kernel<profile>" : "lisennet_frame_kernel");
    if (a.dbg != nullptr) hipLaunchKernelGGL((lisennet_frame_kernel<S, false, true>), dim3(grid), dim3(kThreads), 0, st, a);
    else if (a.clk != nullptr) hipLaunchKernelGGL((lisennet_frame_kernel<S, true, false>), dim3(grid), dim3(kThreads), 0, st, a);
    else hipLaunchKernelGGL((lisennet_frame_kernel<S, false, false>), dim3(grid), dim3(kThreads), 0, st, a);
    *err = hipGetLastError();
}

// the per-hop step of a LARGE batch: front per stream (PART 1), conv_3 .. up3 for sixteen streams per workgroup on the matrix cores
// (lisennet_sb_kernel), tail per stream (PART 2).  fe_debug_step: the same three launches with per-stage dumps; fe_profile_step: the middle's counters.
template <class S>
void llaunch_sb_impl(const LArgs& a, int max_wgs, hipStream_t st, hipError_t* err) {
    constexpr int OCC1 = 8, OCC2 = 8;                              // (16.5 / 14.4 KB of LDS, <= 64 VGPRs: the parts run eight workgroups per CU)
    static_assert(LLdsT<1>::TOTAL * 4 * OCC1 <= 160 * 1024 && LLdsT<2>::TOTAL * 4 * OCC2 <= 160 * 1024, "LDS of the parts");
    const int grid = a.B < max_wgs * OCC1 ? a.B : max_wgs * OCC1;          // (one workgroup per stream instead of persistent ones: measured neutral, 328 / 334 us at 4096 streams)
    const int grid2 = a.B < max_wgs * OCC2 ? a.B : max_wgs * OCC2;
    note_kernel(a.dbg != nullptr ? "lisennet_frame_kernel<PART 1, debug>" : "lisennet_frame_kernel<PART 1>");
    if (a.dbg != nullptr) hipLaunchKernelGGL((lisennet_frame_kernel<S, false, true, false, 1>), dim3(grid), dim3(kThreads), 0, st, a);
    else hipLaunchKernelGGL((lisennet_frame_kernel<S, false, false, false, 1>), dim3(grid), dim3(kThreads), 0, st, a);
    *err = hipGetLastError();
    if (*err != hipSuccess) return;
    LSbArgs sa{};
    sa.wp = a.wp; sa.wp_floats = LPk::TOTAL + LSbPk::TOTAL; sa.carry = a.carry; sa.cache = a.cache; sa.dbg = a.dbg; sa.dbg_stride = a.dbg_stride; sa.B = a.B; sa.clk = a.clk;
    *err = lisennet_sb_launch<S>(sa, st);
    if (*err != hipSuccess) return;
    note_kernel(a.dbg != nullptr ? "lisennet_frame_kernel<PART 2, debug>" : "lisennet_frame_kernel<PART 2>");
    if (a.dbg != nullptr) hipLaunchKernelGGL((lisennet_frame_kernel<S, false, true, false, 2>), dim3(grid2), dim3(kThreads), 0, st, a);
    else hipLaunchKernelGGL((lisennet_frame_kernel<S, false, false, false, 2>), dim3(grid2), dim3(kThreads), 0, st, a);
    *err = hipGetLastError();
}

inline void ldbg_stage_impl(int s, int* rows, int* cols, size_t* off) {
    *rows = LDebugLayout::rows(s);
    *cols = LDebugLayout::cols(s);
    *off = LDebugLayout::offset(s);
}

template <class S>
LImpl make_limpl() {
    constexpr int OCC_LDS = (160 * 1024) / (LLds::TOTAL * 4);
    return LImpl{S::HOP, (size_t)LLds::TOTAL * 4, LDebugLayout::total(), LDebugLayout::n_stages, (size_t)LPk::TOTAL + LSbPk::TOTAL, (size_t)S::CACHE_FLOATS,
                 &llaunch_impl<S>, &ldbg_stage_impl, &llaunch_pipe_impl<S>, OCC_LDS < 2 ? OCC_LDS : 2, 5 + 2 * S::NB, &llaunch_sb_impl<S>};
}

}  // namespace fe
