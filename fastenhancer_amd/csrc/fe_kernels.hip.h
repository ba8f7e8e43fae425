// fe_kernels.hip.h — device code of the FastEnhancer streaming forward path for gfx950 (MI355X).
//
// One workgroup (256 threads = 4 wavefronts, one per SIMD of a CU) owns one stream and runs the
// WHOLE frame pipeline of scripts/export_onnx.py:48-58 for it, frame after frame:
//   STFT (window + FFT)           functional/audio_modules.py:243-257
//   compress, encoder, RNNFormer, decoder, mask, un-compress
//                                 models/fastenhancer/default/model.py:620-710
//   iSTFT (iFFT + synthesis window + overlap-add)   functional/audio_modules.py:259-303
// All activations of a frame live in LDS; every contraction (frequency-axis convs as
// shifted GEMMs, filterbank, GRU gates, qkv / fc, attention) runs on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32, 157 TFLOP/s peak on MI355X).  Weights are read from a
// handle-owned buffer that the host pre-packed in MFMA B-fragment order, so every weight load
// is one fully coalesced 256-byte wave read served by L2.
//
// Activation layouts in LDS (row-major, "row = position along frequency"):
//   conv activations  act[F1+2][LDC]   row r <-> frequency bin r-1 (rows 0 and F1+1 are the zero
//                                     halo of the k=3 convs), column = channel
//   token activations x[F2P][LDX]      row = sub-band, column = RNNFormer channel
// Row strides are 2*odd floats so that the 16x4 A-fragment read (16 rows x 4 k-groups) of
// ds_read_b32 hits 32 distinct banks per half-wave.
#pragma once
#include <hip/hip_runtime.h>

namespace fe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kThreads = 256;
constexpr int kWaves = 4;

__host__ __device__ constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ constexpr int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ------------------------------------------------------------------------------------------
// Compile-time shape of one model (the yaml model_kwargs).  NL = len(kernel_size)-1.
template <int C1_, int NL_, int C2_, int F2_, int KB_, int NFFT_, int HOP_>
struct Shape {
    static constexpr int C1 = C1_, NL = NL_, C2 = C2_, F2 = F2_, KB = KB_, NFFT = NFFT_, HOP = HOP_;
    static constexpr int NH = 4;
    static constexpr int HD = C2 / NH;
    static constexpr int F0 = NFFT / 2;
    static constexpr int F1 = F0 / 4;
    static constexpr int OVL = NFFT - HOP;        // cache length N-H
    static constexpr int LOG2N = (NFFT == 512) ? 9 : (NFFT == 1024 ? 10 : -1);
    // tiles
    static constexpr int MTC = F1 / 16;           // conv m-tiles (4 or 8)
    static constexpr int MTPW = MTC / kWaves;     // conv m-tiles per wave (1 or 2)
    static constexpr int NTC = ceil_div(C1, 16);  // conv n-tiles
    static constexpr int F2P = round_up(F2, 16);
    static constexpr int MT2 = F2P / 16;          // token m-tiles
    static constexpr int NT2 = ceil_div(C2, 16);
    static constexpr int N3 = 3 * C2;
    static constexpr int NT3 = ceil_div(N3, 16);
    // LDS strides (floats)
    static constexpr int LDC = C1 + 2;
    static constexpr int LDX = C2 + 2;
    static constexpr int LDG = NT3 * 16 + 2;
    static constexpr int LDP = 18;                // transposed-conv partials [F1][16]
    static constexpr int LDS_S = F0 + 4 + 2;      // compressed spectrum rows (2-bin zero halo each side)
    static constexpr int ACT = (F1 + 2) * LDC;    // one conv activation buffer
    // state layout (floats, for B streams): [stft B*OVL][istft B*OVL][h KB*B*F2*C2]
    // packed-weight sizes (floats)
    static constexpr int KS_C = C1 / 4;           // k-steps over C1
    static constexpr int KS_2 = C2 / 4;           // k-steps over C2
    static_assert(C1 % 4 == 0 && C2 % 4 == 0 && F2 % 4 == 0, "channel counts must be multiples of 4");
    static_assert(C2 % NH == 0, "C2 must be divisible by the 4 heads");
    static_assert(F1 % 64 == 0, "F1 must be a multiple of 64");
    static_assert(LOG2N > 0, "n_fft must be 512 or 1024");
};

// Offsets (floats) of the packed weights inside the handle's device buffer.  Filled by the host
// packer (fe_api.hip) with the same formulas.
struct PackedOffsets {
    int enc_pre_w, enc_pre_b;
    int enc_w[8], enc_b[8];
    int rfpre_lin, rfpre_w, rfpre_b;
    int blk_pe;                        // block 0 only, [F2][C2]
    int blk_wih[8], blk_bih[8], blk_whh[8], blk_bhh[8];
    int blk_fc1_w[8], blk_fc1_b[8], blk_qkv[8], blk_fc2_w[8], blk_fc2_b[8];
    int rfpost_lin, rfpost_w, rfpost_b;
    int dec1_w[8], dec1_b[8], dec3_w[8], dec3_b[8];
    int post1_w, post1_b, post_t_w, post_t_b;
    int window, window_istft, twiddle;  // [N], [N], [N/2] float2
    int total;
};

struct FrameArgs {
    const float* wp;          // packed weights + tables
    PackedOffsets off;
    const float* wav_in;      // [b*in_stride + t*H + n]
    float* wav_out;
    size_t in_stride, out_stride;
    float* cache_stft;        // [B][OVL]
    float* cache_istft;       // [B][OVL]
    float* h;                 // [KB][B*F2][C2]
    const float* spec_in;     // spec mode: [B][F0+1][T][2]
    float* spec_out;
    float* dbg;               // debug dumps or nullptr
    size_t dbg_stride;        // floats per stream
    int B, T;
    float compression;
};

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// One MFMA panel: acc[MTP][NTP] += A-frags x B-frags over KS k-steps.
template <int MTP, int NTP, int KS, typename AF, typename BF>
__device__ __forceinline__ void mma_panel(f32x4 (&acc)[MTP][NTP], AF&& af, BF&& bf) {
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
        float a[MTP], b[NTP];
#pragma unroll
        for (int i = 0; i < MTP; ++i) a[i] = af(i, ks);
#pragma unroll
        for (int j = 0; j < NTP; ++j) b[j] = bf(j, ks);
#pragma unroll
        for (int i = 0; i < MTP; ++i)
#pragma unroll
            for (int j = 0; j < NTP; ++j) acc[i][j] = FE_MFMA(a[i], b[j], acc[i][j]);
    }
}

template <int MTP, int NTP>
__device__ __forceinline__ void acc_init_bias(f32x4 (&acc)[MTP][NTP], const float* __restrict__ bias_lane, int nt0, int nt_stride, int nt_max) {
#pragma unroll
    for (int j = 0; j < NTP; ++j) {
        int nt = nt0 + j * nt_stride;
        nt = nt < nt_max ? nt : nt_max - 1;
        float b = bias_lane ? bias_lane[nt * 16] : 0.0f;
#pragma unroll
        for (int i = 0; i < MTP; ++i) acc[i][j] = f32x4{b, b, b, b};
    }
}

// ------------------------------------------------------------------------------------------
// Debug stage table (shared by host and device).
template <class S>
struct DebugLayout {
    // order: spec_in, compressed, enc_pre, encoder.i..., rf_pre, (blk.k.rnn, blk.k)..., rf_post, decoder.i..., mask, spec_out
    static constexpr int n_stages = 2 + 1 + S::NL + 1 + 2 * S::KB + 1 + S::NL + 2;
    __host__ __device__ static constexpr int rows(int s) {
        if (s == 0) return S::F0 + 1;
        if (s == 1) return S::F0;
        if (s < 3 + S::NL) return S::F1;
        if (s < 4 + S::NL + 2 * S::KB) return S::F2;
        if (s < 5 + 2 * S::NL + 2 * S::KB) return S::F1;
        if (s == 5 + 2 * S::NL + 2 * S::KB) return S::F0;
        return S::F0 + 1;
    }
    __host__ __device__ static constexpr int cols(int s) {
        if (s <= 1) return 2;
        if (s < 3 + S::NL) return S::C1;
        if (s < 4 + S::NL + 2 * S::KB) return S::C2;
        if (s < 5 + 2 * S::NL + 2 * S::KB) return S::C1;
        return 2;
    }
    __host__ __device__ static constexpr size_t offset(int s) {
        size_t o = 0;
        for (int i = 0; i < s; ++i) o += (size_t)rows(i) * cols(i);
        return o;
    }
    __host__ __device__ static constexpr size_t total() { return offset(n_stages); }
};

template <class S>
__device__ __forceinline__ void dbg_dump(const FrameArgs& a, int b, int stage, const float* src, int ld) {
    if (a.dbg == nullptr) return;
    using D = DebugLayout<S>;
    const int rows = D::rows(stage), cols = D::cols(stage);
    float* dst = a.dbg + (size_t)b * a.dbg_stride + D::offset(stage);
    for (int i = threadIdx.x; i < rows * cols; i += kThreads) {
        int r = i / cols, c = i - r * cols;
        dst[i] = src[r * ld + c];
    }
}

// ------------------------------------------------------------------------------------------
// LDS plan (floats).  "skips" E[0..NL] stay alive from the encoder to the decoder; the rest is a
// scratch arena whose sub-buffers are reused by the phases of a frame.
template <class S>
struct Lds {
    static constexpr int SC = 0;                                  // compressed spectrum [2][LDS_S]
    static constexpr int TW = SC + 2 * S::LDS_S;                  // twiddles float2[N/2]
    static constexpr int E = TW + S::NFFT;                        // skips: (NL+1) x ACT
    static constexpr int ARENA = E + (S::NL + 1) * S::ACT;
    // phase A (STFT / iSTFT): two complex ping-pong buffers
    static constexpr int FFT_A = ARENA;
    static constexpr int FFT_B = FFT_A + 2 * S::NFFT;
    static constexpr int END_FFT = FFT_B + 2 * S::NFFT;
    // phase B (decoder): two conv work buffers + transposed-conv partials
    static constexpr int W0 = ARENA;
    static constexpr int W1 = W0 + S::ACT;
    static constexpr int PT = W1 + S::ACT;                        // [F1][LDP]
    static constexpr int END_CONV = PT + S::F1 * S::LDP;
    // phase C (RNNFormer): placed after W0 so rf_post can write W0 while X is alive
    static constexpr int X = W0 + S::ACT;                         // [F2P][LDX]
    static constexpr int HL = X + S::F2P * S::LDX;                // hidden state / attention out
    static constexpr int GI = HL + S::F2P * S::LDX;               // [F2P][LDG]  (also qkv)
    static constexpr int GH = GI + S::F2P * S::LDG;               // [F2P][LDG]
    static constexpr int Y1 = GH + S::F2P * S::LDG;               // rf_pre: [F2P][LDC]; rf_post: [F1][LDX]
    static constexpr int Y1_SIZE = (S::F2P * S::LDC > S::F1 * S::LDX) ? S::F2P * S::LDC : S::F1 * S::LDX;
    static constexpr int END_RF = Y1 + Y1_SIZE;
    static constexpr int TOTAL_ = (END_FFT > END_CONV ? END_FFT : END_CONV);
    static constexpr int TOTAL = (TOTAL_ > END_RF ? TOTAL_ : END_RF);
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
    static_assert(2 * S::ACT >= 4 * S::NFFT, "FFT ping-pong buffers must not reach the transposed-conv partials");
};

// ------------------------------------------------------------------------------------------
// Complex radix-2 Stockham FFT of NFFT points over two LDS buffers; result ends in `dst`
// returned pointer.  tw[k] = exp(-2*pi*i*k/N); inverse uses the conjugate.
template <class S, bool INVERSE>
__device__ __forceinline__ float2* fft_lds(float2* x, float2* y, const float2* tw) {
    constexpr int N = S::NFFT;
#pragma unroll 1
    for (int s = 0; s < S::LOG2N; ++s) {
        const int Ns = 1 << s;
        for (int j = threadIdx.x; j < N / 2; j += kThreads) {
            const int k = j & (Ns - 1);
            float2 u0 = x[j];
            float2 u1 = x[j + N / 2];
            float2 w = tw[k << (S::LOG2N - 1 - s)];
            if (INVERSE) w.y = -w.y;
            float2 t = make_float2(u1.x * w.x - u1.y * w.y, u1.x * w.y + u1.y * w.x);
            const int j0 = ((j - k) << 1) + k;
            y[j0] = make_float2(u0.x + t.x, u0.y + t.y);
            y[j0 + Ns] = make_float2(u0.x - t.x, u0.y - t.y);
        }
        __syncthreads();
        float2* tmp = x; x = y; y = tmp;
    }
    return x;
}

// ------------------------------------------------------------------------------------------
// conv-layout GEMM segment: this wave's m-tiles (wave + 4*i) x all NT n-tiles, K = 4*KS.
//   a_lane : LDS pointer to A[(16*wave + (lane&15)) rows][(lane>>4) col] of the segment
//   w_lane : packed weights + lane, at k-step 0 of this segment;  KS_TOT = k-steps per n-tile
template <class S, int NT, int KS, int KS_TOT, int LDA>
__device__ __forceinline__ void conv_seg(f32x4 (&acc)[S::MTPW][NT], const float* a_lane, const float* __restrict__ w_lane) {
    mma_panel<S::MTPW, NT, KS>(
        acc,
        [&](int i, int ks) { return a_lane[(64 * i) * LDA + 4 * ks]; },
        [&](int j, int ks) { return w_lane[(j * KS_TOT + ks) * 64]; });
}

// Epilogue of a conv-layout GEMM: optional SiLU, store to out[(row0 + m)][col] for col < NCOLS.
template <class S, int NT, int NCOLS, int LDO, bool ACT>
__device__ __forceinline__ void conv_store(const f32x4 (&acc)[S::MTPW][NT], float* out, int row0, int wave, int lane) {
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int i = 0; i < S::MTPW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = 16 * j + li;
            if (col < NCOLS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * (wave + 4 * i) + 4 * lg + r;
                    float v = acc[i][j][r];
                    if (ACT) v = silu_f(v);
                    out[(row0 + m) * LDO + col] = v;
                }
            }
        }
}

// token-layout GEMM: all MT2 m-tiles x this wave's n-tiles (wave + 4*j), A from LDS, B packed.
template <class S, int NTPW, int KS, int LDA>
__device__ __forceinline__ void tok_gemm(f32x4 (&acc)[S::MT2][NTPW], const float* a_lane, const float* __restrict__ w_lane, int NT, int wave) {
    mma_panel<S::MT2, NTPW, KS>(
        acc,
        [&](int i, int ks) { return a_lane[(16 * i) * LDA + 4 * ks]; },
        [&](int j, int ks) {
            int nt = wave + 4 * j;
            nt = nt < NT ? nt : NT - 1;
            return w_lane[(nt * KS + ks) * 64];
        });
}

// ------------------------------------------------------------------------------------------
template <class S, bool SPEC_MODE>
__global__ void __launch_bounds__(kThreads, 1) fe_frame_kernel(FrameArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = Lds<S>;
    constexpr int N = S::NFFT, H = S::HOP, OVL = S::OVL, F0 = S::F0, F1 = S::F1;
    constexpr int C1 = S::C1, C2 = S::C2, F2 = S::F2, HD = S::HD;
    constexpr int LDC = S::LDC, LDX = S::LDX, LDG = S::LDG;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x;
    const float* __restrict__ wp = a.wp;
    const PackedOffsets& o = a.off;

    // ---- one-time: zero LDS (halo rows, pad rows), load twiddles
    for (int i = tid; i < L::TOTAL; i += kThreads) smem[i] = 0.0f;
    __syncthreads();
    float2* tw = reinterpret_cast<float2*>(smem + L::TW);
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = reinterpret_cast<const float2*>(wp + o.twiddle)[i];
    float* sc = smem + L::SC;
    float2* fa = reinterpret_cast<float2*>(smem + L::FFT_A);
    float2* fb = reinterpret_cast<float2*>(smem + L::FFT_B);
    float* Ebuf = smem + L::E;
    __syncthreads();

    float* cst = a.cache_stft + (size_t)b * OVL;
    float* cis = a.cache_istft + (size_t)b * OVL;

#pragma unroll 1
    for (int t = 0; t < a.T; ++t) {
        // =========================== STFT (a3) ===========================
        if (!SPEC_MODE) {
            const float* win = wp + o.window;
            const float* xin = a.wav_in + (size_t)b * a.in_stride + (size_t)t * H;
            for (int n = tid; n < N; n += kThreads) {
                float v = (n < OVL) ? cst[n] : xin[n - OVL];
                fb[n] = make_float2(v, 0.0f);          // raw frame kept in fb.x for the cache shift
                fa[n] = make_float2(v * win[n], 0.0f);
            }
            __syncthreads();
            for (int m = tid; m < OVL; m += kThreads) cst[m] = fb[m + H].x;   // cache' = frame[H:]
            __syncthreads();
            float2* X = fft_lds<S, false>(fa, fb, tw);
            // spectrum bins 0..F0 (F0 = Nyquist, dropped by the model)
            if (a.dbg) {
                float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(0);
                for (int f = tid; f <= F0; f += kThreads) { dst[2 * f] = X[f].x; dst[2 * f + 1] = X[f].y; }
            }
            // =========================== compress (a4) ===========================
            for (int f = tid; f < F0; f += kThreads) {
                float re = X[f].x, im = X[f].y;
                float mag = fmaxf(sqrtf(re * re + im * im), 1.0e-5f);
                float g = powf(mag, a.compression - 1.0f);
                sc[2 + f] = re * g;
                sc[S::LDS_S + 2 + f] = im * g;
            }
        } else {
            const float* sp = a.spec_in + (size_t)b * (F0 + 1) * a.T * 2;
            for (int f = tid; f < F0; f += kThreads) {
                float re = sp[((size_t)f * a.T + t) * 2], im = sp[((size_t)f * a.T + t) * 2 + 1];
                float mag = fmaxf(sqrtf(re * re + im * im), 1.0e-5f);
                float g = powf(mag, a.compression - 1.0f);
                sc[2 + f] = re * g;
                sc[S::LDS_S + 2 + f] = im * g;
            }
        }
        __syncthreads();
        if (a.dbg) {
            float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(1);
            for (int f = tid; f < F0; f += kThreads) { dst[2 * f] = sc[2 + f]; dst[2 * f + 1] = sc[S::LDS_S + 2 + f]; }
        }

        // =========================== enc_pre (a5): strided conv as K=16 GEMM ===========================
        {
            f32x4 acc[S::MTPW][S::NTC];
            acc_init_bias<S::MTPW, S::NTC>(acc, wp + o.enc_pre_b + li, 0, 1, S::NTC);
            const float* wl = wp + o.enc_pre_w + lane;
            // k = t*8 + s*2 + c  (weight (C1, 8, 2): channel index s*2+c, tap t);  A[m][k] = xpad[c][4(m+t)+s]
            mma_panel<S::MTPW, S::NTC, 4>(
                acc,
                [&](int i, int ks) {
                    const int kk = 4 * ks + lg;
                    const int c = kk & 1, s = (kk >> 1) & 3, tp = kk >> 3;
                    const int m = 16 * (wave + 4 * i) + li;
                    return sc[c * S::LDS_S + 4 * (m + tp) + s];
                },
                [&](int j, int ks) { return wl[(j * 4 + ks) * 64]; });
            conv_store<S, S::NTC, C1, LDC, true>(acc, Ebuf, 1, wave, lane);
        }
        __syncthreads();
        dbg_dump<S>(a, b, 2, Ebuf + LDC, LDC);

        // =========================== encoder (a6): k=3 convs ===========================
#pragma unroll
        for (int l = 0; l < S::NL; ++l) {
            const float* in = Ebuf + l * S::ACT;
            float* out = Ebuf + (l + 1) * S::ACT;
            f32x4 acc[S::MTPW][S::NTC];
            acc_init_bias<S::MTPW, S::NTC>(acc, wp + o.enc_b[l] + li, 0, 1, S::NTC);
            const float* wl = wp + o.enc_w[l] + lane;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
                conv_seg<S, S::NTC, S::KS_C, 3 * S::KS_C, LDC>(acc, in + (16 * wave + li + tap) * LDC + lg, wl + tap * S::KS_C * 64);
            conv_store<S, S::NTC, C1, LDC, true>(acc, out, 1, wave, lane);
            __syncthreads();
            dbg_dump<S>(a, b, 3 + l, out + LDC, LDC);
        }

        float* Xb = smem + L::X;
        float* Hl = smem + L::HL;
        float* Gi = smem + L::GI;
        float* Gh = smem + L::GH;
        float* Y1 = smem + L::Y1;

        // =========================== rf_pre (a7) ===========================
        {
            // Y1[f2][c1] = sum_f1 Wf[f2][f1] * E[f1][c1]      (A = packed filterbank, B = LDS)
            constexpr int NTPW = ceil_div(S::NTC, kWaves);
            constexpr int KS = F1 / 4;
            const float* Ein = Ebuf + S::NL * S::ACT + LDC;   // row 0 = bin 0
            f32x4 acc[S::MT2][NTPW];
            acc_init_bias<S::MT2, NTPW>(acc, nullptr, 0, 1, 1);
            const float* al = wp + o.rfpre_lin + lane;
            mma_panel<S::MT2, NTPW, KS>(
                acc,
                [&](int i, int ks) { return al[(i * KS + ks) * 64]; },
                [&](int j, int ks) {
                    int nt = wave + 4 * j;
                    nt = nt < S::NTC ? nt : S::NTC - 1;
                    return Ein[(4 * ks + lg) * LDC + 16 * nt + li];
                });
#pragma unroll
            for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                for (int j = 0; j < NTPW; ++j) {
                    const int nt = wave + 4 * j;
                    const int col = 16 * nt + li;
                    if (nt < S::NTC && col < C1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * i + 4 * lg + r;
                            if (row < F2) Y1[row * LDC + col] = acc[i][j][r];
                        }
                    }
                }
        }
        __syncthreads();
        {
            // X[f2][c2] = Y1[f2][:] . Wc[c2][:] + b
            constexpr int NTPW = ceil_div(S::NT2, kWaves);
            f32x4 acc[S::MT2][NTPW];
            acc_init_bias<S::MT2, NTPW>(acc, wp + o.rfpre_b + li, wave, 4, S::NT2);
            tok_gemm<S, NTPW, S::KS_C, LDC>(acc, Y1 + li * LDC + lg, wp + o.rfpre_w + lane, S::NT2, wave);
#pragma unroll
            for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                for (int j = 0; j < NTPW; ++j) {
                    const int nt = wave + 4 * j;
                    const int col = 16 * nt + li;
                    if (nt < S::NT2 && col < C2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * i + 4 * lg + r;
                            if (row < F2) Xb[row * LDX + col] = acc[i][j][r];
                        }
                    }
                }
        }
        __syncthreads();
        dbg_dump<S>(a, b, 3 + S::NL, Xb, LDX);

        // =========================== RNNFormer blocks (a9-a11) ===========================
#pragma unroll 1
        for (int k = 0; k < S::KB; ++k) {
            float* hg = a.h + ((size_t)k * a.B + b) * (F2 * C2);
            for (int i = tid; i < F2 * C2; i += kThreads) {
                int f = i / C2, c = i - f * C2;
                Hl[f * LDX + c] = hg[i];
            }
            __syncthreads();
            {
                // gi = x W_ih^T + b_ih ; gh = h W_hh^T + b_hh
                constexpr int NTPW = ceil_div(S::NT3, kWaves);
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    f32x4 acc[S::MT2][NTPW];
                    const float* bias = wp + (which == 0 ? o.blk_bih[k] : o.blk_bhh[k]) + li;
                    const float* wsrc = wp + (which == 0 ? o.blk_wih[k] : o.blk_whh[k]) + lane;
                    const float* asrc = (which == 0 ? Xb : Hl) + li * LDX + lg;
                    float* dst = which == 0 ? Gi : Gh;
                    acc_init_bias<S::MT2, NTPW>(acc, bias, wave, 4, S::NT3);
                    tok_gemm<S, NTPW, S::KS_2, LDX>(acc, asrc, wsrc, S::NT3, wave);
#pragma unroll
                    for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                        for (int j = 0; j < NTPW; ++j) {
                            const int nt = wave + 4 * j;
                            if (nt < S::NT3) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int row = 16 * i + 4 * lg + r;
                                    if (row < F2) dst[row * LDG + 16 * nt + li] = acc[i][j][r];
                                }
                            }
                        }
                }
            }
            __syncthreads();
            // gates (PyTorch order r,z,n) and state update
            for (int i = tid; i < F2 * C2; i += kThreads) {
                int f = i / C2, c = i - f * C2;
                const float* gi = Gi + f * LDG;
                const float* gh = Gh + f * LDG;
                float r = sigmoid_f(gi[c] + gh[c]);
                float z = sigmoid_f(gi[C2 + c] + gh[C2 + c]);
                float n = tanhf(gi[2 * C2 + c] + r * gh[2 * C2 + c]);
                float hp = Hl[f * LDX + c];
                float hn = (1.0f - z) * n + z * hp;
                Hl[f * LDX + c] = hn;
                hg[i] = hn;
            }
            __syncthreads();
            {
                // x += rnn_fc(h') (+ pe in block 0)
                constexpr int NTPW = ceil_div(S::NT2, kWaves);
                f32x4 acc[S::MT2][NTPW];
                acc_init_bias<S::MT2, NTPW>(acc, wp + o.blk_fc1_b[k] + li, wave, 4, S::NT2);
                tok_gemm<S, NTPW, S::KS_2, LDX>(acc, Hl + li * LDX + lg, wp + o.blk_fc1_w[k] + lane, S::NT2, wave);
                const float* pe = wp + o.blk_pe;
#pragma unroll
                for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                    for (int j = 0; j < NTPW; ++j) {
                        const int nt = wave + 4 * j;
                        const int col = 16 * nt + li;
                        if (nt < S::NT2 && col < C2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * i + 4 * lg + r;
                                if (row < F2) {
                                    float v = acc[i][j][r] + Xb[row * LDX + col];
                                    if (k == 0) v += pe[row * C2 + col];
                                    Xb[row * LDX + col] = v;
                                }
                            }
                        }
                    }
            }
            __syncthreads();
            dbg_dump<S>(a, b, 4 + S::NL + 2 * k, Xb, LDX);
            {
                // qkv = x W_qkv^T  -> Gi (rows per head interleaved [h][q|k|v][hd])
                constexpr int NTPW = ceil_div(S::NT3, kWaves);
                f32x4 acc[S::MT2][NTPW];
                acc_init_bias<S::MT2, NTPW>(acc, nullptr, 0, 1, 1);
                tok_gemm<S, NTPW, S::KS_2, LDX>(acc, Xb + li * LDX + lg, wp + o.blk_qkv[k] + lane, S::NT3, wave);
#pragma unroll
                for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                    for (int j = 0; j < NTPW; ++j) {
                        const int nt = wave + 4 * j;
                        if (nt < S::NT3) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * i + 4 * lg + r;
                                if (row < F2) Gi[row * LDG + 16 * nt + li] = acc[i][j][r];
                            }
                        }
                    }
            }
            __syncthreads();
            {
                // attention: wave = head.  S^T[key][query] = K Q^T, softmax over keys, O^T = V^T P^T
                const int hoff = wave * 3 * HD;
                constexpr int KSD = ceil_div(HD, 4);
                constexpr int MTD = ceil_div(HD, 16);
                f32x4 sacc[S::MT2][S::MT2];
                acc_init_bias<S::MT2, S::MT2>(sacc, nullptr, 0, 1, 1);
                mma_panel<S::MT2, S::MT2, KSD>(
                    sacc,
                    [&](int i, int ks) {
                        const int d = 4 * ks + lg;
                        float v = Gi[(16 * i + li) * LDG + hoff + HD + (d < HD ? d : HD - 1)];
                        return d < HD ? v : 0.0f;
                    },
                    [&](int j, int ks) {
                        const int d = 4 * ks + lg;
                        float v = Gi[(16 * j + li) * LDG + hoff + (d < HD ? d : HD - 1)];
                        return d < HD ? v : 0.0f;
                    });
                const float scale = rsqrtf((float)HD);
#pragma unroll
                for (int j = 0; j < S::MT2; ++j) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = 16 * i + 4 * lg + r;
                            float s = sacc[i][j][r] * scale;
                            s = key < F2 ? s : -INFINITY;
                            sacc[i][j][r] = s;
                            mx = fmaxf(mx, s);
                        }
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    float sum = 0.0f;
#pragma unroll
                    for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float p = __expf(sacc[i][j][r] - mx);
                            sacc[i][j][r] = p;
                            sum += p;
                        }
                    sum += __shfl_xor(sum, 16);
                    sum += __shfl_xor(sum, 32);
                    const float inv = 1.0f / sum;
#pragma unroll
                    for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc[i][j][r] *= inv;
                }
                f32x4 oacc[MTD][S::MT2];
                acc_init_bias<MTD, S::MT2>(oacc, nullptr, 0, 1, 1);
                // k-step (i, r): lane group lg supplies key = 16 i + 4 lg + r  (matches the C/D row map)
#pragma unroll
                for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int key = 16 * i + 4 * lg + r;
                        key = key < F2 ? key : F2 - 1;
                        float av[MTD];
#pragma unroll
                        for (int md = 0; md < MTD; ++md) {
                            int d = 16 * md + li;
                            d = d < HD ? d : HD - 1;
                            av[md] = Gi[key * LDG + hoff + 2 * HD + d];
                        }
#pragma unroll
                        for (int md = 0; md < MTD; ++md)
#pragma unroll
                            for (int j = 0; j < S::MT2; ++j) oacc[md][j] = FE_MFMA(av[md], sacc[i][j][r], oacc[md][j]);
                    }
                // O[query][h*HD + d]  (into Hl, dead after rnn_fc)
#pragma unroll
                for (int md = 0; md < MTD; ++md)
#pragma unroll
                    for (int j = 0; j < S::MT2; ++j) {
                        const int q = 16 * j + li;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int d = 16 * md + 4 * lg + r;
                            if (d < HD && q < F2) Hl[q * LDX + wave * HD + d] = oacc[md][j][r];
                        }
                    }
            }
            __syncthreads();
            {
                // x += attn_fc(o)
                constexpr int NTPW = ceil_div(S::NT2, kWaves);
                f32x4 acc[S::MT2][NTPW];
                acc_init_bias<S::MT2, NTPW>(acc, wp + o.blk_fc2_b[k] + li, wave, 4, S::NT2);
                tok_gemm<S, NTPW, S::KS_2, LDX>(acc, Hl + li * LDX + lg, wp + o.blk_fc2_w[k] + lane, S::NT2, wave);
#pragma unroll
                for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                    for (int j = 0; j < NTPW; ++j) {
                        const int nt = wave + 4 * j;
                        const int col = 16 * nt + li;
                        if (nt < S::NT2 && col < C2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * i + 4 * lg + r;
                                if (row < F2) Xb[row * LDX + col] += acc[i][j][r];
                            }
                        }
                    }
            }
            __syncthreads();
            dbg_dump<S>(a, b, 5 + S::NL + 2 * k, Xb, LDX);
        }

        // =========================== rf_post (a13) ===========================
        float* W0 = smem + L::W0;
        float* W1 = smem + L::W1;
        {
            // Y2[f1][c2] = sum_f2 Wp[f1][f2] X[f2][c2]      (A packed, B = LDS tokens)
            constexpr int KS = F2 / 4;
            f32x4 acc[S::MTPW][S::NT2];
            acc_init_bias<S::MTPW, S::NT2>(acc, nullptr, 0, 1, 1);
            const float* al = wp + o.rfpost_lin + lane;
            mma_panel<S::MTPW, S::NT2, KS>(
                acc,
                [&](int i, int ks) { return al[((wave + 4 * i) * KS + ks) * 64]; },
                [&](int j, int ks) { return Xb[(4 * ks + lg) * LDX + 16 * j + li]; });
            conv_store<S, S::NT2, C2, LDX, false>(acc, Y1, 0, wave, lane);
        }
        __syncthreads();
        {
            f32x4 acc[S::MTPW][S::NTC];
            acc_init_bias<S::MTPW, S::NTC>(acc, wp + o.rfpost_b + li, 0, 1, S::NTC);
            conv_seg<S, S::NTC, S::KS_2, S::KS_2, LDX>(acc, Y1 + (16 * wave + li) * LDX + lg, wp + o.rfpost_w + lane);
            conv_store<S, S::NTC, C1, LDC, false>(acc, W0, 1, wave, lane);
        }
        __syncthreads();
        // W1 overlaps the token arena: re-zero its halo rows (row 0 and F1+1)
        for (int i = tid; i < 2 * LDC; i += kThreads) {
            int r = i / LDC, c = i - r * LDC;
            W1[(r ? F1 + 1 : 0) * LDC + c] = 0.0f;
        }
        dbg_dump<S>(a, b, 4 + S::NL + 2 * S::KB, W0 + LDC, LDC);

        // =========================== decoder (a14) ===========================
#pragma unroll
        for (int l = 0; l < S::NL; ++l) {
            const float* skip = Ebuf + (S::NL - l) * S::ACT;
            {
                f32x4 acc[S::MTPW][S::NTC];
                acc_init_bias<S::MTPW, S::NTC>(acc, wp + o.dec1_b[l] + li, 0, 1, S::NTC);
                const float* wl = wp + o.dec1_w[l] + lane;
                conv_seg<S, S::NTC, S::KS_C, 2 * S::KS_C, LDC>(acc, W0 + (16 * wave + li + 1) * LDC + lg, wl);
                conv_seg<S, S::NTC, S::KS_C, 2 * S::KS_C, LDC>(acc, skip + (16 * wave + li + 1) * LDC + lg, wl + S::KS_C * 64);
                conv_store<S, S::NTC, C1, LDC, true>(acc, W1, 1, wave, lane);
            }
            __syncthreads();
            {
                f32x4 acc[S::MTPW][S::NTC];
                acc_init_bias<S::MTPW, S::NTC>(acc, wp + o.dec3_b[l] + li, 0, 1, S::NTC);
                const float* wl = wp + o.dec3_w[l] + lane;
#pragma unroll
                for (int tap = 0; tap < 3; ++tap)
                    conv_seg<S, S::NTC, S::KS_C, 3 * S::KS_C, LDC>(acc, W1 + (16 * wave + li + tap) * LDC + lg, wl + tap * S::KS_C * 64);
                conv_store<S, S::NTC, C1, LDC, true>(acc, W0, 1, wave, lane);   // W0 was fully consumed before the barrier above
            }
            __syncthreads();
            dbg_dump<S>(a, b, 5 + S::NL + 2 * S::KB + l, W0 + LDC, LDC);
        }

        // =========================== dec_post (a15) ===========================
        float* PT = smem + L::PT;
        {
            f32x4 acc[S::MTPW][S::NTC];
            acc_init_bias<S::MTPW, S::NTC>(acc, wp + o.post1_b + li, 0, 1, S::NTC);
            const float* wl = wp + o.post1_w + lane;
            conv_seg<S, S::NTC, S::KS_C, 2 * S::KS_C, LDC>(acc, W0 + (16 * wave + li + 1) * LDC + lg, wl);
            conv_seg<S, S::NTC, S::KS_C, 2 * S::KS_C, LDC>(acc, Ebuf + (16 * wave + li + 1) * LDC + lg, wl + S::KS_C * 64);
            conv_store<S, S::NTC, C1, LDC, true>(acc, W1, 1, wave, lane);
        }
        __syncthreads();
        {
            // transposed conv as GEMM: P[i][co*8+j] = sum_ci x[i][ci] w[ci][co][j]
            f32x4 acc[S::MTPW][1];
            acc_init_bias<S::MTPW, 1>(acc, nullptr, 0, 1, 1);
            conv_seg<S, 1, S::KS_C, S::KS_C, LDC>(acc, W1 + (16 * wave + li + 1) * LDC + lg, wp + o.post_t_w + lane);
            conv_store<S, 1, 16, S::LDP, false>(acc, PT, 0, wave, lane);
        }
        __syncthreads();

        // =========================== mask, un-compress (a16, a17), Hermitian spectrum ===========================
        {
            const float b0 = wp[o.post_t_b], b1 = wp[o.post_t_b + 1];
            float* spo = SPEC_MODE ? a.spec_out + (size_t)b * (F0 + 1) * a.T * 2 : nullptr;
            for (int f = tid; f < F0; f += kThreads) {
                const int q = f + 2, j1 = q & 3, i1 = q >> 2;
                float m0 = b0, m1 = b1;
                if (i1 < F1) { m0 += PT[i1 * S::LDP + j1]; m1 += PT[i1 * S::LDP + 8 + j1]; }
                if (i1 >= 1) { m0 += PT[(i1 - 1) * S::LDP + j1 + 4]; m1 += PT[(i1 - 1) * S::LDP + 8 + j1 + 4]; }
                const float xr = sc[2 + f], xi = sc[S::LDS_S + 2 + f];
                float yr = xr * m0 - xi * m1;
                float yi = xr * m1 + xi * m0;
                if (a.dbg) {
                    float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(5 + 2 * S::NL + 2 * S::KB);
                    dst[2 * f] = m0; dst[2 * f + 1] = m1;
                }
                const float mag = sqrtf(yr * yr + yi * yi);
                const float g = powf(mag, 1.0f / a.compression - 1.0f);
                yr *= g; yi *= g;
                if (a.dbg) {
                    float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(6 + 2 * S::NL + 2 * S::KB);
                    dst[2 * f] = yr; dst[2 * f + 1] = yi;
                    if (f == 0) { dst[2 * F0] = 0.0f; dst[2 * F0 + 1] = 0.0f; }
                }
                if (SPEC_MODE) {
                    spo[((size_t)f * a.T + t) * 2] = yr;
                    spo[((size_t)f * a.T + t) * 2 + 1] = yi;
                    if (f == 0) { spo[((size_t)F0 * a.T + t) * 2] = 0.0f; spo[((size_t)F0 * a.T + t) * 2 + 1] = 0.0f; }
                } else {
                    if (f == 0) {
                        fa[0] = make_float2(yr, 0.0f);       // irfft ignores Im X[0]
                        fa[F0] = make_float2(0.0f, 0.0f);    // zero-padded Nyquist bin
                    } else {
                        fa[f] = make_float2(yr, yi);
                        fa[N - f] = make_float2(yr, -yi);
                    }
                }
            }
        }
        __syncthreads();

        // =========================== iSTFT (a18) ===========================
        if (!SPEC_MODE) {
            float2* y = fft_lds<S, true>(fa, fb, tw);
            float2* spare = (y == fa) ? fb : fa;
            const float* wi = wp + o.window_istft;
            float* xo = reinterpret_cast<float*>(spare);
            const float invN = 1.0f / (float)N;
            for (int n = tid; n < N; n += kThreads) {
                float v = y[n].x * invN * wi[n];
                if (n < OVL) v += cis[n];
                xo[n] = v;
            }
            __syncthreads();
            float* out = a.wav_out + (size_t)b * a.out_stride + (size_t)t * H;
            for (int n = tid; n < H; n += kThreads) out[n] = xo[n];
            for (int m = tid; m < OVL; m += kThreads) cis[m] = xo[m + H];
            __syncthreads();
        }
    }
}

}  // namespace fe
