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

#include <type_traits>
#include <utility>

#ifndef FE_OCC_SMALL
#define FE_OCC_SMALL 3
#endif
namespace fe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int FE_MODE_STREAM = 0, FE_MODE_SPEC = 1, FE_MODE_OFFLINE = 2;
constexpr int kThreads = 256;
constexpr int kWaves = 4;

// FE_WG8_HPRE=1: a measured-negative variant of the 512-thread per-hop kernel (fe_frame8.hip.h, profiles/r5_headline_hpre.txt); its extra packed
// section (PackedOffsets::u8_gh4) exists in such builds only
#ifndef FE_WG8_HPRE
#define FE_WG8_HPRE 0
#endif

// fe_last_step_kernel (C ABI, r6): every host-side launcher names the kernel it enqueues (a string literal: family + instantiation);
// the compute entry points of fe_api.hip collect the names of one call in the handle.  Defined in fe_api.hip (thread-local log).
void note_kernel(const char* name);

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__host__ __device__ constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ constexpr int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ------------------------------------------------------------------------------------------
// Compile-time shape of one model (the yaml model_kwargs).  NL = len(kernel_size)-1.  KT = kernel_size_time of the
// `fastenhancer.time_kernel` variant (models/fastenhancer/time_kernel/model.py: the k = 3 convs are causal Conv2d with KT
// taps over time); KT = 1 is the default model.
// LOW = 1: the low-LDS build of the same model (weights read from L2 instead of staged through LDS) whose plan fits twice
// per CU: the companion kernel for batches with more streams than CUs, where two workgroups per CU fill each other's stalls.
// FR = 1: the `fastenhancer.dprnn` variant (models/fastenhancer/dprnn/model.py:135-247): the block's attention is a bidirectional
// GRU over the F2 sub-bands, C2 / 2 hidden units per direction, zero initial state every frame; no positional embedding.
// TA = L > 0: the `fastenhancer.dptransformer` variant (models/fastenhancer/dptransformer/model.py:175-236): the block's time GRU is
// a causal attention over the last L frames (K / V caches [F2][NH][L][HD] per block and stream in the state) with a learned
// positional bias [NH][L + 1].
// LN = 1: the `fastenhancer.ln` variant (models/fastenhancer/ln/model.py): GroupNorm(1, C) after every conv (statistics over the
// channels and sub-bands of the frame), the reference's LayerNorm over (F2, C2) after the blocks' fc layers; nothing folds.
// BD = 1: the `fastenhancer.noncausal` variant (models/fastenhancer/noncausal/model.py:186-187): the blocks' time GRU is
// bidirectional and rnn_fc maps 2 C2 -> C2; offline only, run by the time-batched engine (tb_kernels.hip.h) alone.
template <int C1_, int NL_, int C2_, int F2_, int KB_, int NFFT_, int HOP_, int KT_ = 1, int LOW_ = 0, int FR_ = 0, int TA_ = 0, int LN_ = 0, int BD_ = 0>
struct Shape {
    static constexpr bool LN = LN_ != 0;
    static constexpr bool BIDIR = BD_ != 0;
    static constexpr int ND = BD_ != 0 ? 2 : 1;      // GRU directions over time
    // shapes the time-batched (layer-by-layer) engine runs: the default model and the noncausal variant
    static constexpr bool TB = KT_ == 1 && LOW_ == 0 && FR_ == 0 && TA_ == 0 && LN_ == 0;
    static constexpr int LN_SITES = 4 + 3 * NL_ + 2 * KB_;      // enc_pre, encoder.i, rf_pre, (rnn, attn) per block, rf_post, (1x1, k3) per decoder layer, dec_post
    static constexpr int C1 = C1_, NL = NL_, C2 = C2_, F2 = F2_, KB = KB_, NFFT = NFFT_, HOP = HOP_, KT = KT_, LOW = LOW_;
    static constexpr bool FRNN = FR_ != 0;
    static constexpr bool TATT = TA_ != 0;
    static constexpr int LB = TA_;                // dptransformer: lookbehind
    static constexpr int HSTATE = TATT ? 2 * LB * F2_ * C2_ : F2_ * C2_;      // model-state floats per block and stream
    static constexpr int HF = C2_ / 2;            // dprnn: hidden units per direction of the sub-band GRU
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
    static constexpr int NU = 6 + NL * (2 * KT + 1) + (LN ? 1 : 0);   // LDS-staged weight units (Pack<S>): a k = 3 conv is one unit per time tap; ln: + rf_post's 1x1
    // unit indices in consumption order
    static constexpr int U_ENC = 1;                                  // + l * KT + tap
    static constexpr int U_RFPRE = 1 + NL * KT;                      // filterbank, then the 1x1
    static constexpr int U_RFPOST = 3 + NL * KT;
    static constexpr int U_DEC = 4 + NL * KT + (LN ? 1 : 0);         // + l * (KT + 1): the 1x1, then + 1 + tap: the k = 3 conv  (ln: rf_post's 1x1 is unit U_RFPOST + 1)
    static constexpr int U_POST = 4 + NL * (2 * KT + 1) + (LN ? 1 : 0);   // dec_post 1x1, then the transposed conv
    static constexpr int TKQ = (KT - 1) * (F0 / 4) * C1;             // floats of one conv's frame cache per stream: [KT-1][F1][C1]
    // RNNFormer-block weight fragments held in registers per wave (this wave's column tiles)
    static constexpr int NTPW2 = ceil_div(NT2, kWaves), NTPW3 = ceil_div(NT3, kWaves);
    // "flat" GRU gate GEMM: C2 not a multiple of 16 (T: 20, B: 36) pads every gate to whole column tiles (B: 3 x 48
    // columns = 9 tiles, one wave idle); the gates as ONE 3 C2-column matrix need ceil(3 C2 / 16) tiles (B: 7) spread
    // like the qkv GEMM.  r, z, n of a channel then sit in different lanes: the pre-activations cross through LDS
    // and the gate math runs element-per-thread over all 256 threads.  (Register-resident weights only.)
    static constexpr bool GFLAT = (C2 % 16 != 0) && (3 * NTPW3 + 2 * NTPW2) * KS_2 <= 160;
    // block weights register-resident in the 256-thread per-hop kernel (fetched one phase ahead).  Those shapes read a second
    // copy of the block-weight region (PackedOffsets::k4_delta) whose tiles are regrouped four k-steps per lane: one 16-byte
    // load per four fragments (a wave-level load costs the CU's vector-memory path 12 - 17 cycles whatever its width)
    static constexpr bool REGW = GFLAT || (NTPW2 * 6 + NTPW3 + 2 * NTPW2) * KS_2 <= 160;
    // "channel-grouped" GRU gate packing of the 512-thread per-hop kernel (fe_frame8.hip.h): the r, z, n columns of 16 channels as
    // three tiles of ONE wave (the gates of a (row, channel) meet in a lane: gate math in the GEMM epilogue, no exchange through
    // LDS), the C2 % 16 left-over channels' r | z | n in one mixed tile.  A packing predicate only: it must not depend on LOW
    // (a low-LDS companion shares the packed buffer of its shape).
    static constexpr int G8_NG = C2 / 16, G8_R = C2 % 16;
    static constexpr bool G8P = GFLAT && MT2 == 2 && G8_NG == 2 && G8_R > 0 && 3 * G8_R <= 16 && KT_ == 1 && FR_ == 0 && TA_ == 0 && LN_ == 0 && BD_ == 0;
    static constexpr int G8_NT = 3 * G8_NG + 1;      // tiles: (group, gate) ..., the mixed tile
    // 512-thread per-hop kernel: its block weights are staged through LDS like the conv weights, as units of whole 1-KiB pieces:
    // B fragments [tile][k-step][64] followed by the tiles' start values [tile][16] (a token GEMM's bias)
    static constexpr int U8_G = round_up(G8_NT * (C2 / 4) * 64 + G8_NT * 16, 256);                     // GRU input / hidden weights: the channel-grouped gate tiles
    static constexpr int U8_GH4 = round_up(G8_NT * (((C2 / 4) / 4) * 256 + ((C2 / 4) % 4) * 64), 256);      // the hidden weights regrouped for 16-byte fetches (PackedOffsets::u8_gh4)
    static constexpr int U8_F = round_up(ceil_div(C2, 16) * (C2 / 4) * 64 + ceil_div(C2, 16) * 16, 256);   // rnn_fc / attn_fc
    static constexpr int U8_Q = round_up(ceil_div(3 * C2, 16) * (C2 / 4) * 64, 256);                   // qkv (no bias)
    static constexpr int U8_SLOT = U8_G > U8_Q ? U8_G : U8_Q;                                        // one of the four staging slots
    static_assert(C1 % 4 == 0 && C2 % 4 == 0 && F2 % 4 == 0, "channel counts must be multiples of 4");
    static_assert(C2 % NH == 0, "C2 must be divisible by the 4 heads");
    static_assert(F1 % 64 == 0, "F1 must be a multiple of 64");
    static_assert(LOG2N > 0, "n_fft must be 512 or 1024");
};

// Offsets (floats) of the packed weights inside the handle's device buffer.  The layout is a pure
// function of the shape, evaluated at compile time and shared by the host packer (fe_api.hip) and the
// kernel, where every offset folds into an instruction immediate / one SGPR add.
#ifndef FE_FBAL
#define FE_FBAL 1           // r5: rnn_fc / attn_fc of the shapes with more than four column tiles over balanced (column tile, row tile) jobs (fe_frame_kernel; 0: whole column tiles, for A/B runs)
#endif
#ifndef FE_QBAL
#define FE_QBAL 1           // r5: qkv's last two column tiles (NT3 % 4 == 2: L) split by row halves over the wave pairs (0: whole column tiles, for A/B runs)
#endif
#ifndef FE_K4_STREAM
#define FE_K4_STREAM 1      // r4x: shapes that stream their block weights from L2 inside the GEMMs (M, L, their 48 kHz / variant shapes) fetch four k-steps per 16-byte load too
#endif
struct PackedOffsets {
    int enc_pre_w, enc_pre_b;
    int enc_w[16], enc_b[8];            // enc_w / dec3_w: [layer * KT + tap], taps in consumption order (tap 0 = the current frame)
    int rfpre_lin, rfpre_w, rfpre_b;
    int blk_pe;                        // block 0 only, [F2][C2]
    int blk_stride;                    // offset of block k+1's arrays minus block k's
    int blk_wih[8], blk_bih[8], blk_whh[8], blk_bhh[8];
    int blk_fc1_w[8], blk_fc1_b[8], blk_qkv[8], blk_fc2_w[8], blk_fc2_b[8];
    // dprnn variant: blk_qkv holds the sub-band GRU's input weights of both directions as one (3 C2)-column matrix (columns
    // [direction][gate r|z|n][unit]), blk_qkv_b its bias (b_ih, plus b_hh for r and z), blk_fhh the hidden weights
    // [direction][j][gate][unit] (a lane = a unit reads consecutive floats), blk_fbhn b_hn [direction][unit]
    int blk_qkv_b[8], blk_fhh[8], blk_fbhn[8];
    // dptransformer variant: the time attention's qkv weights per block, the model's positional bias [NH][32] (slot L = current frame)
    int blk_tqkv[8], tpe;
    int blk_end, k4_delta;      // end of the block-weight region; distance to its k4-regrouped copy (Shape::REGW, else 0)
    int conv_k4_delta;          // time-batched engine (Shape::TB): distance from the conv units to their k4-regrouped copy (r4w), else 0
    // ln variant: rf_post's 1x1 conv as a staged unit (B fragments + bias), gain / bias of every norm site ([channel])
    int rfpost1_w, rfpost1_b, ln_g[48], ln_b[48];
    int rfpost_lin, rfpost_w, rfpost_b;
    int dec1_w[8], dec1_b[8], dec3_w[16], dec3_b[8];
    int post1_w, post1_b, post_t_w, post_t_b;
    int window, window_istft, twiddle;  // [N], [N], [N/2] float2
    int dft1, dft2, dft3, dft4;         // constant operands of the matrix-core DFT (see Dft<S>)
    int gru_flat;                       // 1: GRU weights / biases packed as one (3 C2)-column matrix (Shape::GFLAT)
    // time-batched engine (tb_kernels.hip.h; Shape::TB), per block and direction: the GRU input weights as ONE flat (3 C2)-column B
    // operand (rows r | z | n as stored by nn.GRU) with the bias b_ih (+ b_hh for r, z) [3 C2]; the hidden weights per gate
    // (tile = gate * NT2 + channel tile, so that r, z, n of a (row, channel) meet in one lane) and b_hn [NT2 * 16];
    // noncausal: rnn_fc over 2 C2 input channels
    int tb_wih[8][2], tb_bx[8][2], tb_whh[8][2], tb_bhn[8][2], tb_fc1_w[8];
    // 512-thread per-hop kernel (Shape::G8P), per block: the block weights as LDS-staged units (Shape::U8_*): B fragments
    // [tile][k-step][64], then the start values [tile][16].  gx / gh: input / hidden weights of the channel-grouped GRU gate tiles
    // (start values: b_ih, plus b_hh on pure r / z tiles whose x and h halves share an accumulator / b_hh on the n tiles and the mixed
    // tile, else 0); f1, q, f2: rnn_fc, qkv, attn_fc in their plain column order.
    int u8_gx[8], u8_gh[8], u8_f1[8], u8_q[8], u8_f2[8];
    int u8_gh4[8];                      // r5: the hidden-weight tiles once more, [tile][k-step / 4][lane][4] + the k-steps % 4 plain: fetched by the waves that
                                        // accumulate W_hh h of all blocks in the front of the frame (fe_frame8.hip.h, FE_WG8_HPRE)
    int total;
    // LDS-staged weight "units" in consumption order (one per conv-type GEMM phase): [weights | bias],
    // 256-float aligned and padded, so that a unit is staged by whole 1-KiB global_load_lds pieces.
    int n_units;
    int u_off[32], u_size[32];
};

template <class S>
struct Pack {
    static constexpr int szB(int K, int N) { return ceil_div(N, 16) * (K / 4) * 64; }   // B fragments
    static constexpr int szA(int M, int K) { return ceil_div(M, 16) * (K / 4) * 64; }   // A fragments
    static constexpr int szBias(int n) { return round_up(n, 16); }
    static constexpr PackedOffsets make() {
        PackedOffsets o{};
        int cur = 0, nu = 0;
        auto alloc = [&](int n) { cur = round_up(cur, 64); const int at = cur; cur += n; return at; };
        auto ubegin = [&]() { cur = round_up(cur, 256); o.u_off[nu] = cur; };
        auto uend = [&]() { cur = round_up(cur, 256); o.u_size[nu] = cur - o.u_off[nu]; ++nu; };
        constexpr int C1 = S::C1, C2 = S::C2, F1 = S::F1, F2 = S::F2;
        // conv-type biases are stored 4x replicated ([channel][4]): one 16-byte read initialises the four accumulator
        // rows of a lane - replicating in registers costs three v_mov per tile, and VALU work is never hidden here
        ubegin(); o.enc_pre_w = alloc(szB(16, C1)); o.enc_pre_b = alloc(4 * szBias(C1)); uend();
        for (int l = 0; l < S::NL; ++l)
            for (int tp = 0; tp < S::KT; ++tp) {
                ubegin(); o.enc_w[l * S::KT + tp] = alloc(szB(3 * C1, C1));
                if (tp == 0) o.enc_b[l] = alloc(4 * szBias(C1));
                uend();
            }
        ubegin(); o.rfpre_lin = alloc(szA(F2, F1)); uend();
        ubegin(); o.rfpre_w = alloc(szB(C1, C2)); o.rfpre_b = alloc(4 * szBias(C2)); uend();
        ubegin(); o.rfpost_lin = alloc(szA(F1, F2)); uend();
        if (S::LN) { ubegin(); o.rfpost1_w = alloc(szB(C2, C1)); o.rfpost1_b = alloc(4 * szBias(C1)); uend(); }
        // (rf_post's 1x1 conv has no unit: the host folds it into decoder layer 0's 1x1, whose first K-segment is the
        //  filterbank output - dec1_w[0] holds (C2 + C1) x C1; the plain copy below only serves the debug dump)
        for (int l = 0; l < S::NL; ++l) {
            ubegin(); o.dec1_w[l] = alloc(szB(2 * C1, C1)); o.dec1_b[l] = alloc(4 * szBias(C1)); uend();
            for (int tp = 0; tp < S::KT; ++tp) {
                ubegin(); o.dec3_w[l * S::KT + tp] = alloc(szB(3 * C1, C1));
                if (tp == 0) o.dec3_b[l] = alloc(4 * szBias(C1));
                uend();
            }
        }
        ubegin(); o.post1_w = alloc(szB(2 * C1, C1)); o.post1_b = alloc(4 * szBias(C1)); uend();
        ubegin(); o.post_t_w = alloc(szB(C1, 16)); o.post_t_b = alloc(szBias(2)); uend();
        o.n_units = nu;
        o.rfpost_w = alloc(C1 * C2); o.rfpost_b = alloc(szBias(C1));      // plain [C1][C2] / [C1], true scale (debug only)
        // RNNFormer-block weights: read by each wave straight into registers (not staged)
        o.blk_pe = alloc(F2 * C2);
        for (int k = 0; k < S::KB; ++k) {      // identical sizes per block: the offsets advance by blk_stride
            // GRU weights packed per gate (r, z, n): tile index = gate * NT2 + channel-tile, so that the three
            // gate pre-activations of one (row, channel) land in the same lane and the gates fuse into the epilogue
            // (GFLAT shapes: one flat (3 C2)-column matrix in the same, larger, allocation)
            o.blk_wih[k] = alloc(3 * szB(C2, C2)); o.blk_whh[k] = alloc(3 * szB(C2, C2));
            o.blk_bih[k] = alloc(3 * szBias(C2)); o.blk_bhh[k] = alloc(3 * szBias(C2));
            o.blk_fc1_w[k] = alloc(szB(C2, C2)); o.blk_fc1_b[k] = alloc(szBias(C2));
            o.blk_qkv[k] = alloc(szB(C2, 3 * C2));
            o.blk_fc2_w[k] = alloc(szB(C2, C2)); o.blk_fc2_b[k] = alloc(szBias(C2));
            if (S::FRNN) {
                o.blk_qkv_b[k] = alloc(szBias(3 * C2)); o.blk_fhh[k] = alloc(2 * S::HF * 3 * S::HF); o.blk_fbhn[k] = alloc(2 * S::HF);
            }
            if (S::TATT) o.blk_tqkv[k] = alloc(szB(C2, 3 * C2));
        }
        o.blk_end = round_up(cur, 64);
        if (S::TATT) o.tpe = alloc(S::NH * 32);
        if (S::LN)
            for (int q = 0; q < S::LN_SITES; ++q) { o.ln_g[q] = alloc(szBias(C1 > C2 ? C1 : C2)); o.ln_b[q] = alloc(szBias(C1 > C2 ? C1 : C2)); }
        o.blk_stride = S::KB > 1 ? o.blk_wih[1] - o.blk_wih[0] : 0;
        o.gru_flat = S::GFLAT ? 1 : 0;
        o.k4_delta = 0;
        if (S::REGW || FE_K4_STREAM) {      // k4-regrouped copy of [blk_wih[0], blk_end): the register-resident fetches (TokW) and, r4x, the STREAMED block weights of the big shapes
            const int n = o.blk_end - o.blk_wih[0];
            o.k4_delta = alloc(n) - o.blk_wih[0];
        }
        o.window = alloc(S::NFFT); o.window_istft = alloc(S::NFFT); o.twiddle = alloc(S::NFFT);
        {
            constexpr int N1 = S::NFFT / 32, KC = N1 / 2, MT = N1 / 16;
            o.dft1 = alloc(2 * 2 * 8 * 64); o.dft2 = alloc(2 * KC * 64); o.dft3 = alloc(2 * MT * KC * 64); o.dft4 = alloc(2 * 2 * 8 * 64);
        }
        // (allocated for the low-LDS companions too - S::TB without its LOW term: a companion shares its shape's buffer and reads the conv k4 copy behind this)
        if (S::KT == 1 && !S::FRNN && !S::TATT && !S::LN)
            for (int k = 0; k < S::KB; ++k) {
                for (int d = 0; d < S::ND; ++d) {
                    o.tb_wih[k][d] = alloc(szB(C2, 3 * C2)); o.tb_bx[k][d] = alloc(szBias(3 * C2));
                    o.tb_whh[k][d] = alloc(3 * szB(C2, C2)); o.tb_bhn[k][d] = alloc(szBias(C2));
                }
                if (S::BIDIR) o.tb_fc1_w[k] = alloc(szB(2 * C2, C2));
            }
        if (S::G8P)      // (allocated last: every other offset is the same with and without it)
            for (int k = 0; k < S::KB; ++k) {
                cur = round_up(cur, 256); o.u8_gx[k] = cur; cur += S::U8_G;
                o.u8_gh[k] = cur; cur += S::U8_G;
                o.u8_f1[k] = cur; cur += S::U8_F;
                o.u8_q[k] = cur; cur += S::U8_Q;
                o.u8_f2[k] = cur; cur += S::U8_F;
            }
#if FE_WG8_HPRE
        if (S::G8P)
            for (int k = 0; k < S::KB; ++k) { cur = round_up(cur, 256); o.u8_gh4[k] = cur; cur += S::U8_GH4; }
#endif
        o.conv_k4_delta = 0;
        if (S::KT == 1 && !S::LN) {     // (allocated after everything else: every other offset is the same with and without it; NOT a function of LOW: a companion shares its shape's buffer)
            // r4w: the time-batched engine's conv GEMMs stream their weight fragments from L2 - one wave-level load per (tile, k-step) kept the
            // CU's vector-memory path as busy as its matrix pipes (539 loads for 930 MFMAs per tile of tb_dec<B>).  A copy of the conv units
            // [u_off[0], end of the last unit) whose weight tiles are regrouped four k-steps per lane ([ks / 4][lane][4], the ks % 4 remainder
            // plain; the host packer does the shuffle) lets conv_gemm fetch four fragments per 16-byte load.
            const int ubeg = o.u_off[0], uend_ = o.u_off[o.n_units - 1] + o.u_size[o.n_units - 1];
            o.conv_k4_delta = alloc(uend_ - ubeg) - ubeg;
        }
        o.total = round_up(cur, 64);
        return o;
    }
    static constexpr PackedOffsets v = make();
    static constexpr int umax() { int m = 0; for (int u = 0; u < v.n_units; ++u) m = v.u_size[u] > m ? v.u_size[u] : m; return m; }
};

struct FrameArgs {
    const float* wp;          // packed weights + tables (layout: Pack<S>::v)
    const float* wav_in;      // [b*in_stride + t*H + n]
    float* wav_out;
    size_t in_stride, out_stride;
    float* cache_stft;        // [B][OVL]
    float* cache_istft;       // [B][OVL]
    float* h;                 // [KB][B*F2][C2]
    float* tk;                // time_kernel variant: the causal convs' frame caches [2 NL][B][KT-1][F1][C1] (encoder layers, then decoder)
    float* skip;              // global skip scratch [B][(NL+1)][F1*C1] (A-fragment order), shapes with !Lds::SKIPS_LDS
    const float* spec_in;     // spec mode: [B][F0+1][T][2]
    float* spec_out;
    float* dbg;               // debug dumps or nullptr
    unsigned long long* clk;  // phase cycle counters (block 0, thread 0) or nullptr
    size_t dbg_stride;        // floats per stream
    int B, T;
    int mode;                 // FE_MODE_*
    int Tw;                   // offline: samples per stream
    float compression;
    float rf_eps;             // ln variant: eps of the blocks' LayerNorms (rnnformer_kwargs.eps)
    // time-pipelined launches (PIPE instantiation): P workgroups per stream, workgroup p runs frames p, p + P, ...
    unsigned int* pipe_flags; // [B][KB (+ 2 NL, time_kernel variant)]: number of frames whose block-k GRU state (whose input of time conv
                              // j) has been published (zeroed before the launch)
    float* frames;            // offline: [B][T][N] windowed output frames (overlap-added by istft_ola_kernel afterwards)
    int pipe_p;
    int tatt_base;            // dptransformer, time-pipelined launch: 0 = offline (frame t lives in ring slot t mod RS, frames before the utterance are
                              // masked); LB = spec -> spec step with carried caches (r4v): ring indices 0 .. LB - 1 hold the caller's caches, oldest
                              // first, frame t lives at index LB + t, and only slots marked +inf (a cache-less start) are masked
    int tk_base;              // time_kernel variant, time-pipelined launch: 0 = offline (frames before the utterance read as zero); KT - 1 = spec -> spec
                              // step with carried caches (r4v): ring indices 0 .. KT - 2 hold the caller's cache slots, frame t lives at index KT - 1 + t
    int step_kernel;          // host side only (fe_impl.h::launch_impl): FE_STEP_KERNEL_* of the handle (fe_set_step_kernel)
};

// ------------------------------------------------------------------------------------------
#ifndef FE_PROBE_TID
#define FE_PROBE_TID 0      // the thread whose clock the phase probes record (measurement builds: -DFE_PROBE_TID=256 = wave 4 of the 512-thread kernel)
#endif
#define FE_CLK(i) do { if (a.clk != nullptr && blockIdx.x == 0 && threadIdx.x == FE_PROBE_TID) a.clk[(i)] = __builtin_readcyclecounter(); } while (0)

// v_exp_f32 / v_rcp_f32 are 1-ulp hardware ops; the resulting activations are accurate to a few
// 1e-7 (measured against the oracle per stage), far inside the 1e-4 waveform budget.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// The conv trunk (enc_pre .. encoder, rf_post .. decoder .. dec_post) carries its activations SCALED by
// kSiluScale = -log2(e): with u = kSiluScale * v the SiLU is  kSiluScale * silu(v) = u / (1 + 2^u)  - one VALU multiply
// less per element than x * rcp(1 + exp2(x * -log2 e)), and VALU instructions are not hidden by anything on this path
// (DESIGN.md §3).  A conv fed by scaled activations produces scaled pre-activations from UNCHANGED weights; the host
// packer (fe_api.hip) scales the biases, the first layer's weights (enc_pre, rf_post's 1x1) and un-scales at the two
// exits (rf_pre's 1x1, the transposed conv).  Debug dumps of those stages are un-scaled on the way out.
constexpr float kSiluScale = -1.4426950408889634f;
__device__ __forceinline__ float silu_scaled_f(float u) { return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u)); }
// x^p for x > 0 through the 1-ulp hardware log2 / exp2 (v_log_f32, v_exp_f32); pow_f(0, p > 0) = 0
__device__ __forceinline__ float pow_f(float x, float p) { return x > 0.0f ? __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x)) : 0.0f; }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
// GRU gate pre-activations carried SCALED on the latency chains (the time-batched scans, the FSPEN / LiSenNet recurrences): r, z by
// -log2 e, the n parts by 2 log2 e - the scale sits in the weights / at the place the x side is produced -, so that sigmoid and tanh are
// exp2 + rcp with no multiply on the step's chain (a lone wave issues one instruction every ~6 cycles whatever its kind)
constexpr float kGateRZ = -1.4426950408889634f, kGateN = 2.8853900817779268f;
__device__ __forceinline__ float sigmoid_pre(float u) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u)); }          // u = -log2e x
__device__ __forceinline__ float tanh_pre(float u) { return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u)), 1.0f); }   // u = 2 log2e x

// Weights / tables are read through ONE buffer resource: the per-lane part of every address is the
// single VGPR `lane*4`; the section / tile / k-step part is a scalar (SGPR or immediate) offset.
template <bool STAGED>
struct WSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    int lane4;          // (threadIdx & 63) * 4  bytes
    int li4;            // (lane & 15) * 4 bytes
    const float* lds;   // STAGED: LDS copy of the current weight unit
    int base;           // STAGED: absolute offset (floats) of the current unit in the packed buffer
    int k4d;            // distance (floats) from the block-weight region to its k4-regrouped copy (Shape::REGW), else 0
    __device__ __forceinline__ float at(int off_floats) const {          // + lane
        if constexpr (STAGED) return lds[off_floats - base + (lane4 >> 2)];
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane4, off_floats * 4, 0));
    }
    __device__ __forceinline__ f32x4 at16x4(int off_floats) const {      // 4x replicated table: + 4 (lane & 15), 16 bytes
        if constexpr (STAGED) return *reinterpret_cast<const f32x4*>(lds + (off_floats - base) + li4);
        else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, li4 * 4, off_floats * 4, 0));
    }
    __device__ __forceinline__ float at16(int off_floats) const {        // + (lane & 15)
        if constexpr (STAGED) return lds[off_floats - base + (li4 >> 2)];
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, li4, off_floats * 4, 0));
    }
    // always from global/L2 (weights that are not LDS-staged)
    __device__ __forceinline__ float at_g(int off_floats) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane4, off_floats * 4, 0));
    }
    __device__ __forceinline__ float at16_g(int off_floats) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, li4, off_floats * 4, 0));
    }
    // explicit per-lane byte offset (an offset beyond the buffer reads 0 without touching memory)
    __device__ __forceinline__ float at_gv(int off_floats, int voff_bytes) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_bytes, off_floats * 4, 0));
    }
    __device__ __forceinline__ f32x4 at_gv4(int off_floats, int voff_bytes) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, off_floats * 4, 0));
    }
    __device__ __forceinline__ float gather_g(int off_floats_per_lane) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off_floats_per_lane * 4, 0, 0));
    }
    __device__ __forceinline__ float gather(int off_floats_per_lane) const {   // arbitrary per-lane offset
        if constexpr (STAGED) return lds[off_floats_per_lane - base];
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off_floats_per_lane * 4, 0, 0));
    }
    __device__ __forceinline__ float scalar(int off_floats) const {
        if constexpr (STAGED) return lds[off_floats - base];
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, 0, off_floats * 4, 0));
    }
};

// Staging of one weight unit (n floats, multiple of 256) from L2 into an LDS buffer, one GEMM phase ahead, in 1-KiB
// pieces (wave w moves pieces w, w+4, ...).  Each piece is a buffer_load_dwordx4 into registers issued when the
// previous phase starts (stage_begin) and a ds_write_b128 issued after that phase's GEMM (dma_rest): the L2 latency
// hides behind the GEMM and the two instructions cost ~25 issue cycles per piece.  (The LDS-DMA form,
// global_load_lds_dwordx4, was measured at ~130 issue cycles per piece inside the MFMA loop - 12 % of the frame -
// and makes every __syncthreads() drain it.)
template <int NPW>
struct DmaJobT {
    f32x4 r[NPW];     // this wave's pieces in flight
    float* l;         // LDS destination of the unit
    __amdgpu_buffer_rsrc_t rsrc;
    int soff;         // byte offset of this wave's piece 0 in the packed weights
    int wave, lane;
};
// The loads of a unit of NP pieces ride in the software pipeline of the phase's GEMM as its side job, spread
// over the k-groups (issued back to back they stall the wave: the vector-memory path takes 64 B/clk per CU, i.e.
// 64 cycles per round of four 1-KiB pieces); commit() writes them to LDS after the GEMM.
template <int NPW, int NP, int NWV = kWaves>      // NWV: waves of the workgroup (8 in the 512-thread per-hop kernel, fe_frame8.hip.h)
struct StageSide {
    DmaJobT<NPW>* j;
    static constexpr int NPI = (NP + NWV - 1) / NWV;     // pieces per wave
    // piece i rides in k-group slot(i, ng): spread over the first 2/3 of the GEMM so that the last one has landed by commit()
    static constexpr int slot(int i, int ng) { const int span = (2 * ng + 2) / 3 > 0 ? (2 * ng + 2) / 3 : 1; return i * span / NPI; }
    static constexpr int loads(int g, int ng) {
        int n = 0;
        for (int i = 0; i < NPI; ++i) n += (slot(i, ng) == g) ? 1 : 0;
        return n;
    }
    __device__ __forceinline__ void load(int i) const {
        int voff = j->lane * 16;
        if (NWV * i + NWV - 1 >= NP) voff = (j->wave + NWV * i < NP) ? voff : 0x40000000;     // out of range: reads 0, no traffic
        j->r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(j->rsrc, voff, j->soff + i * (NWV * 1024), 0));
    }
    __device__ __forceinline__ void operator()(int g, int ng) const {
#pragma unroll
        for (int i = 0; i < NPI; ++i)
            if (slot(i, ng) == g) load(i);
    }
    __device__ __forceinline__ void commit() const {
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int p = j->wave + NWV * i;
            if (NWV * i + NWV - 1 < NP || p < NP) *reinterpret_cast<f32x4*>(j->l + p * 256 + j->lane * 4) = j->r[i];
        }
    }
};
// `side(g, NG)`: a job run once per k-group g of a GEMM's software pipeline, in the matrix pipe's shadow
struct NoSide {
    static constexpr int loads(int, int) { return 0; }
    __device__ __forceinline__ void operator()(int, int) const {}
    __device__ __forceinline__ void commit() const {}
};
template <class A, class B>
struct Side2 {
    A a; B b;
    static constexpr int loads(int g, int ng) { return A::loads(g, ng) + B::loads(g, ng); }
    __device__ __forceinline__ void operator()(int g, int ng) const { a(g, ng); b(g, ng); }
};
template <class A, class B>
__device__ __forceinline__ Side2<A, B> side2(const A& a, const B& b) { return Side2<A, B>{a, b}; }
// fetch a slice of one / two TokW register sets per k-group (the next phase's weights)
template <class TW>
struct FetchSide {
    TW* w;
    static constexpr int loads(int g, int ng) { return TW::part_count(g, ng); }
    __device__ __forceinline__ void operator()(int g, int ng) const { w->fetch_part(g, ng); }
};
template <class TA, class TB>
struct FetchSide2 {
    TA* a; TB* b;
    static constexpr int loads(int g, int ng) { return TA::part_count(g, ng) + TB::part_count(g, ng); }
    __device__ __forceinline__ void operator()(int g, int ng) const { a->fetch_part(g, ng); b->fetch_part(g, ng); }
};

// Software-pipelined MFMA panel.  Operands of k-step ks+PD are fetched while k-step ks is multiplied (ring of PD
// k-steps in registers), and the instruction stream is pinned with sched_group_barrier so that the fetches issue
// one or two at a time right AFTER an MFMA - in the 32-cycle shadow of the matrix pipe - instead of in blocks
// between MFMA groups (measured on the k=3 conv: 47 -> ~34 cycles per MFMA).  Left alone, hipcc sinks every
// load next to its consumer and the loop runs at operand latency.
// The fine-grained form is used for panels of up to 128 MFMAs (the scheduler's group solver makes hipcc's
// compile time explode beyond that: FastEnhancer_L did not finish in an hour); larger panels use the group-blocked
// form (operand groups two ahead, pinned with sched_barrier(0)).
// accsel(i, j, ks) -> the accumulator of tile (i, j) at k-step ks (a compile-time choice once unrolled): lets ONE
// pipeline run two GEMMs back to back into different accumulator sets (the GRU's x and h halves).
template <int MTP, int NTP, int KS, int PDK = 8, typename ACCSEL, typename AF, typename BF, typename SIDE>
__device__ __forceinline__ void mma_panel_sel(ACCSEL&& accsel, AF&& af, BF&& bf, SIDE&& side) {
  using SIDE_T = std::remove_cv_t<std::remove_reference_t<SIDE>>;
  if constexpr (KS * MTP * NTP <= 128) {
    constexpr int PD = KS < PDK ? KS : PDK;          // prefetch distance in k-steps (PDK: 8 when operands stream from L2,
                                                     // 3 when they all sit in LDS / registers - Lds<S>::PDK)
    constexpr int NSG = (KS + 3) / 4;                // side-job slots (one per 4 k-steps)
    float a[PD][MTP], b[PD][NTP];
#pragma unroll
    for (int ks = 0; ks < PD; ++ks) {
#pragma unroll
        for (int i = 0; i < MTP; ++i) a[ks][i] = af(i, ks);
#pragma unroll
        for (int j = 0; j < NTP; ++j) b[ks][j] = bf(j, ks);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float av[MTP], bv[NTP];
#pragma unroll
        for (int i = 0; i < MTP; ++i) av[i] = a[ks % PD][i];
#pragma unroll
        for (int j = 0; j < NTP; ++j) bv[j] = b[ks % PD][j];
        if (ks + PD < KS) {
#pragma unroll
            for (int i = 0; i < MTP; ++i) a[ks % PD][i] = af(i, ks + PD);
#pragma unroll
            for (int j = 0; j < NTP; ++j) b[ks % PD][j] = bf(j, ks + PD);
        }
        if ((ks & 3) == 0) side(ks >> 2, NSG);
#pragma unroll
        for (int i = 0; i < MTP; ++i)
#pragma unroll
            for (int j = 0; j < NTP; ++j) { f32x4& c = accsel(i, j, ks); c = FE_MFMA(av[i], bv[j], c); }
    }
    // instruction-stream shape: [prologue fetches] then per k-step { MFMA, fetch, fetch, MFMA, fetch, ... }
    constexpr int LOADS = 0x100 | 0x020;             // DS read | VMEM read
    constexpr int NM = MTP * NTP, NL = MTP + NTP;
    __builtin_amdgcn_sched_group_barrier(LOADS, PD * NL, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            // distribute this k-step's NL fetches over its NM MFMAs
            const int lo = (m * NL) / NM, hi = ((m + 1) * NL) / NM;
            if (ks + PD < KS) {
#pragma unroll
                for (int q = lo; q < hi; ++q) __builtin_amdgcn_sched_group_barrier(LOADS, 1, 0);
            }
            // the side job's loads of this k-group, one per MFMA
            {
                const int g = ks >> 2, npos = (KS - 4 * g < 4 ? KS - 4 * g : 4) * NM, pos = (ks & 3) * NM + m;
                const int sl = SIDE_T::loads(g, NSG);
#pragma unroll
                for (int q = (pos * sl) / npos; q < ((pos + 1) * sl) / npos; ++q) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    }
  } else {
    constexpr int G = (MTP * NTP >= 6) ? 2 : 4;      // k-steps per group
    constexpr int D = 2;                             // prefetch distance in groups
    constexpr int NG = (KS + G - 1) / G;
    float a[D + 1][G][MTP], b[D + 1][G][NTP];
    auto load_group = [&](int g, int slot) {
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
            const int ks = g * G + kk;
            if (ks < KS) {
#pragma unroll
                for (int i = 0; i < MTP; ++i) a[slot][kk][i] = af(i, ks);
#pragma unroll
                for (int j = 0; j < NTP; ++j) b[slot][kk][j] = bf(j, ks);
            }
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < NG) load_group(d, d);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + D < NG) load_group(g + D, (g + D) % (D + 1));
        side(g, NG);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
            if (g * G + kk < KS) {
#pragma unroll
                for (int i = 0; i < MTP; ++i)
#pragma unroll
                    for (int j = 0; j < NTP; ++j)
                    { f32x4& c = accsel(i, j, g * G + kk); c = FE_MFMA(a[g % (D + 1)][kk][i], b[g % (D + 1)][kk][j], c); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int MTP, int NTP, int KS, int PDK = 8, typename AF, typename BF, typename SIDE>
__device__ __forceinline__ void mma_panel(f32x4 (&acc)[MTP][NTP], AF&& af, BF&& bf, SIDE&& side) {
    mma_panel_sel<MTP, NTP, KS, PDK>([&](int i, int j, int) -> f32x4& { return acc[i][j]; }, af, bf, side);
}

template <int MTP, int NTP>
__device__ __forceinline__ void acc_init_zero(f32x4 (&acc)[MTP][NTP]) {
#pragma unroll
    for (int j = 0; j < NTP; ++j)
#pragma unroll
        for (int i = 0; i < MTP; ++i) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// acc[.][j] = bias[16 * nt_j + (lane & 15)] (4x replicated table),  nt_j = min(nt0 + j * nt_stride, nt_max - 1)
template <int MTP, int NTP, class WS>
__device__ __forceinline__ void acc_init_bias(f32x4 (&acc)[MTP][NTP], const WS& w, int bias_off, int nt0, int nt_stride, int nt_max) {
#pragma unroll
    for (int j = 0; j < NTP; ++j) {
        int nt = nt0 + j * nt_stride;
        nt = nt < nt_max ? nt : nt_max - 1;
        const f32x4 b = w.at16x4(bias_off + nt * 64);
#pragma unroll
        for (int i = 0; i < MTP; ++i) acc[i][j] = b;
    }
}

// All-reduce over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with the gfx950 row-swap VALU ops
// (v_permlane32_swap / v_permlane16_swap) instead of ds_bpermute round trips through the LDS pipeline.
// (inline asm: this hipcc's __builtin_amdgcn_permlane{16,32}_swap returns its first result twice; the s_nop covers
//  the VALU-write -> permlane-swap read hazard the compiler would otherwise pad for)
template <class OP>
__device__ __forceinline__ float rows_allreduce(float x, OP op) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a = {lo|lo}, b = {hi|hi}
    float c = op(a, b), d = c;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));      // c = even rows twice, d = odd rows twice
    return op(c, d);
}

// All-reduce over the 16 lanes of each DPP row (rotate by 8, 4, 2, 1 - every lane ends with the row's result).
template <class OP>
__device__ __forceinline__ float row16_allreduce(float x, OP op) {
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false)));   // row_ror:8
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false)));   // row_ror:4
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false)));   // row_ror:2
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false)));   // row_ror:1
    return x;
}

// ln variant: one norm site.  buf holds the [ROWS x COLS] pre-norm values of the frame (leading dimension LD); the statistics are
// over ALL of them (nn.GroupNorm(1, C) on [C, F] / the reference's LayerNorm over (F, C)): per-thread partial sums, DPP / row-swap
// wave sums, four partials through LDS, one barrier.  FC = false: buf <- act(xhat * g[c] + b[c]).  FC = true (the blocks'
// LayerNorm AS WRITTEN in models/fastenhancer/ln/model.py:31-34 - `diff.addcmul(inv_std * weight, bias)`, i.e. the centred value
// plus inv_std * weight * bias): xres[r][c] += (v - mean) + inv_std * g[c] * b[c] (+ pe[r][c]).  The caller barriers afterwards.
template <int ROWS, int COLS, int LD, bool ACT, bool FC>
__device__ __forceinline__ void ln_pass(float* buf, float* red, const float* g, const float* bt, float eps, float* xres = nullptr, int ldx = 0,
                                        const float* pe = nullptr) {
    constexpr int N = ROWS * COLS, PER = ceil_div(N, kThreads);
    const int tid = threadIdx.x;
    float v[PER], gc[PER], bc[PER], s = 0.0f, q = 0.0f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + kThreads * i, ec = e < N ? e : N - 1, r = ec / COLS, c = ec - r * COLS;
        v[i] = buf[r * LD + c];
        gc[i] = g[c]; bc[i] = bt[c];          // (L2 loads: in flight under the reduction and its barrier)
        const float m = e < N ? v[i] : 0.0f;
        s += m; q = fmaf(m, m, q);
    }
    auto add = [](float x, float y) { return x + y; };
    s = rows_allreduce(row16_allreduce(s, add), add);
    q = rows_allreduce(row16_allreduce(q, add), add);
    if ((tid & 63) == 0) { red[tid >> 6] = s; red[4 + (tid >> 6)] = q; }
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) * (1.0f / N);
    const float var = fmaxf((red[4] + red[5] + red[6] + red[7]) * (1.0f / N) - mean * mean, 0.0f);
    const float inv = __builtin_amdgcn_rsqf(var + eps);
    // (threads past the end of the last round redo element N - 1 and store to a dummy slot: no partially executed region)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + kThreads * i, ec = e < N ? e : N - 1, r = ec / COLS, c = ec - r * COLS;
        if constexpr (FC) {
            float x = xres[r * ldx + c] + (v[i] - mean) + inv * gc[i] * bc[i];
            if (pe != nullptr) x += pe[r * COLS + c];
            (e < N ? xres + r * ldx + c : red + 8)[0] = x;
        } else {
            float y = fmaf((v[i] - mean) * inv, gc[i], bc[i]);
            if (ACT) y = silu_f(y);
            (e < N ? buf + r * LD + c : red + 8)[0] = y;
        }
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

template <class S, int NTH = kThreads>
__device__ __forceinline__ void dbg_dump(const FrameArgs& a, int b, int stage, const float* src, int ld) {
    if (a.dbg == nullptr) return;
    using D = DebugLayout<S>;
    const int rows = D::rows(stage), cols = D::cols(stage);
    float* dst = a.dbg + (size_t)b * a.dbg_stride + D::offset(stage);
    // conv-trunk stages (enc_pre, encoder.i, rf_post, decoder.i) live scaled by kSiluScale
    const bool trunk = (stage >= 2 && stage < 3 + S::NL) || (stage >= 4 + S::NL + 2 * S::KB && stage < 5 + 2 * S::NL + 2 * S::KB);
    const float sc = (trunk && !S::LN) ? 1.0f / kSiluScale : 1.0f;      // (the ln variant's trunk is not scaled)
    for (int i = threadIdx.x; i < rows * cols; i += NTH) {
        int r = i / cols, c = i - r * cols;
        dst[i] = src[r * ld + c] * sc;
    }
}

// ------------------------------------------------------------------------------------------
// LDS plan (floats).  "skips" E[0..NL] stay alive from the encoder to the decoder; the rest is a
// scratch arena whose sub-buffers are reused by the phases of a frame.
template <class S>
struct Lds {
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int SC = 0;                                  // compressed spectrum [2][LDS_S]
    static constexpr int TW = SC + 2 * S::LDS_S;                  // twiddles float2[N/2]
    static constexpr int E = TW + S::NFFT;                        // skips: (NL+1) x ACT (when they fit)
    // Encoder outputs ("skips") stay in LDS from the encoder to the decoder when the plan fits in 160 KiB;
    // otherwise they live in a per-stream global scratch (L2-resident) in MFMA A-fragment order and the
    // encoder ping-pongs through W0/W1 of the arena.
    static constexpr int ARENA_SIZE_SKIPS_LDS =
        cmax(cmax(4 * S::NFFT, 3 * S::F2P * S::LDX + S::F2P * S::LDG),
             cmax(2 * S::ACT + S::F1 * S::LDP, cmax(S::F2P * S::LDX, S::ACT) + S::F1 * S::LDX));
    // Preference: skips in LDS + staged weights; else - for shapes that cannot use the column-split conv GEMMs (three
    // channel tiles: 48 kHz B) - skips in the global scratch so that the weights can be staged; else skips in LDS with
    // streamed weights; else everything global / streamed.
    static constexpr bool FITS_SKIPS = (size_t)(E + (S::NL + 1) * S::ACT + ARENA_SIZE_SKIPS_LDS) * 4 <= 160 * 1024;
    static constexpr bool FITS_SKIPS_STAGED = (size_t)(E + (S::NL + 1) * S::ACT + ARENA_SIZE_SKIPS_LDS + 2 * Pack<S>::umax()) * 4 <= 160 * 1024;
    static constexpr int ARENA_SIZE_SKIPS_GLOBAL =
        cmax(cmax(4 * S::NFFT, cmax(3 * S::F2P * S::LDX, S::ACT) + S::F2P * S::LDG),
             cmax(2 * S::ACT + S::F1 * S::LDP, cmax(S::F2P * S::LDX, S::ACT) + S::F1 * S::LDX));
    static constexpr bool FITS_GLOBAL_STAGED = (size_t)(E + ARENA_SIZE_SKIPS_GLOBAL + 2 * Pack<S>::umax()) * 4 <= 160 * 1024;
    static constexpr bool SKIPS_LDS = S::LOW != 2 && FITS_SKIPS && (S::LOW == 1 || FITS_SKIPS_STAGED || !(FITS_GLOBAL_STAGED && S::NTC % 4 != 0 && S::NTC % 2 != 0));
    static constexpr int ARENA = E + (SKIPS_LDS ? (S::NL + 1) * S::ACT : 0);
    // The arena is re-used by the phases of a frame (offsets relative to ARENA):
    //   STFT / iSTFT : FFT_A, FFT_B                       (complex ping-pong)
    //   rf_pre       : Y1 (aliases the qkv buffer), X
    //   blocks       : X, HL, HS, qkv
    //   rf_post      : X (read) -> Y2 -> W0               (Y2 must not overlap X nor W0)
    //   decoder      : W0, W1, PT
    static constexpr int FFT_A = ARENA;
    static constexpr int FFT_B = FFT_A + 2 * S::NFFT;
    static constexpr int END_FFT = 4 * S::NFFT;
    static constexpr int X = ARENA;                               // [F2P][LDX]
    static constexpr int HL = X + S::F2P * S::LDX;                // hidden state / attention out
    static constexpr int HS = HL + S::F2P * S::LDX;               // GRU hidden state of the current block [F2P][LDX]
    // qkv [F2P][LDG]; with global skips the last encoder output sits in W0 while rf_pre writes Y1 (= this buffer).
    // PERHEAD (48 kHz L: 96 tokens x 288 qkv columns = 111 KB): not even the global-skip plan fits with a full qkv
    // buffer, so qkv is computed and consumed one head at a time ([F2P][3 hd + 2], all four waves on the same head) and
    // the rf_pre intermediate sits behind W0, over the not yet live HL / HS.
    static constexpr bool PERHEAD = (size_t)(E + ARENA_SIZE_SKIPS_GLOBAL) * 4 > 160 * 1024;
    static constexpr int LDGX = PERHEAD ? 3 * S::HD + 2 : S::LDG;
    static constexpr int GI_OFF = SKIPS_LDS ? 3 * S::F2P * S::LDX : cmax(3 * S::F2P * S::LDX, S::ACT);
    static constexpr int GI = ARENA + GI_OFF;
    static constexpr int Y1_OFF = PERHEAD ? cmax(S::F2P * S::LDX, S::ACT) : GI_OFF;
    static constexpr int Y1 = ARENA + Y1_OFF;                     // rf_pre intermediate [F2P][LDC]
    static constexpr int END_RF = cmax(GI_OFF + S::F2P * LDGX, Y1_OFF + S::F2P * S::LDC);
    static constexpr int W0 = ARENA;
    static constexpr int W1 = W0 + S::ACT;
    static constexpr int PT = W1 + S::ACT;                        // [F1][LDP]
    static constexpr int END_CONV = 2 * S::ACT + S::F1 * S::LDP;
    static constexpr int Y2_OFF = cmax(S::F2P * S::LDX, S::ACT);   // after X, after W0
    static constexpr int Y2 = ARENA + Y2_OFF;                     // rf_post intermediate [F1][LDX]
    static constexpr int END_Y2 = Y2_OFF + S::F1 * S::LDX;
    static constexpr int ARENA_SIZE = cmax(cmax(END_FFT, END_RF), cmax(END_CONV, END_Y2));
    static constexpr int NOSTAGE_TOTAL = ARENA + ARENA_SIZE;
    // weights are staged through two LDS buffers (one GEMM phase ahead) whenever they fit
    static constexpr bool STAGED = !S::LOW && (size_t)(NOSTAGE_TOTAL + 2 * Pack<S>::umax()) * 4 <= 160 * 1024;
    static constexpr int WB0 = NOSTAGE_TOTAL;
    static constexpr int WB1 = WB0 + Pack<S>::umax();
    static constexpr int TOTAL_KT1 = STAGED ? NOSTAGE_TOTAL + 2 * Pack<S>::umax() : NOSTAGE_TOTAL;
    // time_kernel variant: the previous KT-1 input frames of the conv being computed, activation layout with halo rows
    static constexpr int CA = TOTAL_KT1;
    static constexpr int LNS = TOTAL_KT1 + (S::KT - 1) * S::ACT;      // ln variant: 16 floats for the block-wide sums
    static constexpr int TOTAL = LNS + (S::LN ? 16 : 0);
    static_assert(S::KT == 1 || (STAGED && SKIPS_LDS), "the time_kernel variant is built for shapes with staged weights and LDS-resident skips");
    static_assert((size_t)TOTAL * 4 <= 160 * 1024, "LDS plan exceeds 160 KiB");
    // software-pipeline depth of the fine-grained MFMA panels: with staged conv weights (and then register-resident
    // block weights) every operand comes from LDS / registers, ~130 cycles away: 3 k-steps ahead is enough and a
    // shorter pipeline fill after each barrier is worth 4.7 % on FastEnhancer_B (8 / 6 / 4 / 3 / 2 measured);
    // operands streamed from L2 need the full 8.
    // low-LDS companions (two workgroups per CU share the latency hiding; r2 same-box A/B at 512 / 1024 streams: B 8 -> 5:
    // +5 % / -0.6 %, S and 48 kHz B 8 -> 4: +2.6 % / +5 % at 512)
    static constexpr int PDK = STAGED ? 3 : (S::LOW == 1 ? 5 : (S::LOW == 2 ? 4 : 8));
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
    // workgroups per CU: small shapes (FastEnhancer_T: 62 KiB) fit twice - with more streams than CUs two workgroups share a
    // CU and fill each other's barrier / latency stalls (one wave per SIMD each); everything else owns its CU
    // r6: the low-LDS companions of the SMALLEST shapes (C1 <= 32: T, 48 kHz T - ~45 KB without the staging buffers, < 170 registers per lane) fit three
    // times (FE_OCC_SMALL): three one-stream workgroups per CU, three waves per SIMD
    static constexpr int OCC_FIT = (int)((160 * 1024) / BYTES);
    static constexpr int OCC = (S::LOW != 0 && S::C1 <= 32 && OCC_FIT > 2) ? (OCC_FIT < FE_OCC_SMALL ? OCC_FIT : FE_OCC_SMALL) : ((2 * BYTES <= 160 * 1024) ? 2 : 1);
    // a companion's PERSISTENT instantiation (more streams than 2 x #CUs) is only used where it beats the shape's own
    // kernel: S spills 92 VGPRs there (3.54 M frames/s at 1024 streams against 3.65 M), B / 48 kHz B gain 8 % / 5 %
    static constexpr bool MANY_PERSIST = !(S::LOW == 2 && S::C1 >= 64);
    // ... and the dptransformer B companion only THERE: its one-stream-per-workgroup instantiation spills (the K / V window in registers next to the streamed weight
    // fragments: 384 / 512 streams 167 / 183 us against the shape's own 106 / 124), the persistent one does not (1024 / 2048 streams 249 -> 216 / 483 -> 410 us)
    static constexpr bool MANY_ONE_ROUND = !(S::TATT && S::C1 >= 48);
    static_assert(2 * S::ACT >= 4 * S::NFFT, "the FFT buffers must not reach the transposed-conv partials");
    static_assert(PERHEAD || S::F2P * S::LDC <= S::F2P * S::LDG, "rf_pre intermediate must fit in the qkv buffer");
    static_assert(!PERHEAD || !SKIPS_LDS, "per-head qkv implies global skips");
};

// ------------------------------------------------------------------------------------------
// Complex radix-2 Stockham FFT of NFFT points over two LDS buffers (autosort, no bit reversal); returns the
// buffer that holds the result.  tw[k] = exp(-2*pi*i*k/N); INVERSE uses the conjugate.  (A radix-4 variant was
// measured slower: its stride-4p stores are 16-way bank-conflicted for p = 1, 4 - each radix-2 stage costs
// ~360 cycles, mostly barrier + LDS latency.)
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
// Real DFT / inverse real DFT of one frame on the matrix cores.  N = N1 * 32 (N1 = 16 or 32); with
// n = n1 + N1 n2 and k = 32 k1 + k2 the transform factors (Cooley-Tukey, decimation in frequency) into
//   forward:  G[n1][k2] = sum_n2 x[n1 + N1 n2] W_32^(n2 k2)              (N1 x 32) . (32 x 32)   real x complex
//             X[32 k1 + k2] = sum_n1 W_N1^(n1 k1) (W_N^(n1 k2) G[n1][k2])   (16 x N1) . (N1 x 32)   complex x complex
//   inverse:  H[n1][k2] = sum_k1 conj(W_N1^(n1 k1)) Y[32 k1 + k2]         (Y Hermitian-extended past N/2)
//             y[n1 + N1 n2] = Re sum_k2 (conj(W_N^(n1 k2)) H[n1][k2]) conj(W_32^(n2 k2)) / N
// Complex products are real GEMMs with the real and imaginary parts stacked along K.  Wave w owns the column
// tile k2 in [16 jt, 16 jt + 16), jt = w & 1, and p = w >> 1 selects its output half.  The two stages of a
// transform chain IN REGISTERS: the first stage's C/D fragments (row 4 lg + r, column li) are used directly as
// the second stage's B (forward) / A (inverse, first stage computed transposed) fragments - a k-step then carries
// the rows {r, 4 + r, 8 + r, 12 + r} instead of four consecutive ones, which only permutes the K order of the
// constant operand (packed accordingly by fe_api.hip into Pack<S>::dft1..4).  The twiddle multiply happens on
// those registers, so each wave computes both the real and the imaginary half of the first stage for its
// column tile (2x redundant, 16 MFMAs) and there is no LDS exchange or barrier between the stages: one barrier
// per transform instead of the log2 N of a radix-2 pass (each ~350 cycles at this size).
template <class S, int PDK = Lds<S>::PDK>      // (PDK given explicitly by shapes that have no Lds<S> plan: BSRNN)
struct Dft {
    static constexpr int N = S::NFFT, N1 = N / 32, MT = N1 / 16, KC = N1 / 2;
    static constexpr bool PRELOAD3 = (N1 == 16);      // inverse first-stage constants in registers (else streamed from L2)
    static_assert(N1 == 16 || N1 == 32, "DFT factorisation: N = 512 or 1024");
    // y partial sums, index of sample n = n1 + N1 n2 (row n2, rotated by n2: the producer writes one row per lane)
    __device__ static __forceinline__ int pidx(int n1, int n2) { return n2 * N1 + ((n1 + n2) & (N1 - 1)); }
    __device__ static __forceinline__ float2 twid(const float2* tw, int idx) {     // W_N^idx, idx < N
        float2 t = tw[idx & (N / 2 - 1)];
        if (idx >= N / 2) { t.x = -t.x; t.y = -t.y; }
        return t;
    }

    struct FwdConst { float c1[2][8], c2[KC]; };
    struct InvConst { float c3[PRELOAD3 ? 2 * MT : 1][PRELOAD3 ? KC : 1], c4[8]; };
    template <class WS, class OFF>      // (OFF: any offsets struct with dft1 .. dft4 - PackedOffsets, BSRNN's BOffsets)
    __device__ static __forceinline__ void load(FwdConst& c, const WS& wb, const OFF& o, int wave) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) c.c1[a][ks] = wb.at_g(o.dft1 + ((a * 2 + (wave & 1)) * 8 + ks) * 64);
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) c.c2[ks] = wb.at_g(o.dft2 + ((wave >> 1) * KC + ks) * 64);
    }
    template <class WS, class OFF>
    __device__ static __forceinline__ void load(InvConst& c, const WS& wb, const OFF& o, int wave) {
        if constexpr (PRELOAD3) {
#pragma unroll
            for (int j = 0; j < 2 * MT; ++j)
#pragma unroll
                for (int ks = 0; ks < KC; ++ks) c.c3[j][ks] = wb.at_g(o.dft3 + (j * KC + ks) * 64);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) c.c4[ks] = wb.at_g(o.dft4 + (wave * 8 + ks) * 64);
    }

    // xw[N] (windowed frame) -> X = {Re[N/2], Im[N/2]} (bins 0 .. N/2-1), followed by a barrier.
    // nyq != nullptr: also store bin N/2 there (debug dump only; for N1 = 32 it costs an extra reduction).
    template <bool BAR = true>      // BAR = false: the caller barriers (the 512-thread kernel runs the transform on four of its eight waves)
    __device__ static __forceinline__ void forward(const float* xw, float* X, const float2* tw, const FwdConst& c,
                                                   int wave, int lane, float* nyq) {
        const int li = lane & 15, lg = lane >> 4, p = wave >> 1, jt = wave & 1, k2 = 16 * jt + li;
        f32x4 g[MT][2];                                  // [.][0] = Re G, [.][1] = Im G, rows 16 i + 4 lg + r, column k2
        acc_init_zero<MT, 2>(g);
        mma_panel<MT, 2, 8, PDK>(g, [&](int i, int ks) { return xw[16 * i + li + N1 * (4 * ks + lg)]; },
                            [&](int a2, int ks) { return c.c1[a2][ks]; }, NoSide{});
        float gq[KC];                                    // k-step a * 4 MT + 4 i + r of the second stage
        float2 tq[MT][4];                                // (twiddles fetched together, ahead of the panel's results)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) tq[i][r] = twid(tw, (16 * i + 4 * lg + r) * k2);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 t = tq[i][r];
                const float gr = g[i][0][r], gi = g[i][1][r];
                gq[4 * i + r] = gr * t.x - gi * t.y;
                gq[4 * MT + 4 * i + r] = gi * t.x + gr * t.y;
            }
        f32x4 e0 = {0.0f, 0.0f, 0.0f, 0.0f}, e1 = e0;    // two chains (dependent-MFMA latency is 40 cycles, issue 32)
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            e0 = FE_MFMA(c.c2[ks], gq[ks], e0);
            e1 = FE_MFMA(c.c2[KC / 2 + ks], gq[KC / 2 + ks], e1);
        }
        float* Xp = X + p * (N / 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k1 = 4 * lg + r;
            if (k1 < N1 / 2) Xp[32 * k1 + k2] = e0[r] + e1[r];
        }
        if (nyq != nullptr && wave == 0) {
            if constexpr (N1 == 16) {
                if (lane == 32) { nyq[0] = e0[0] + e1[0]; nyq[1] = 0.0f; }       // k1 = 8, k2 = 0
            } else {                                     // X[N/2] = sum (-1)^n x[n]
                float sgn = 0.0f;
                for (int n = lane; n < N; n += 64) sgn += (n & 1) ? -xw[n] : xw[n];
                for (int d = 32; d >= 1; d >>= 1) sgn += __shfl_xor(sgn, d);
                if (lane == 0) { nyq[0] = sgn; nyq[1] = 0.0f; }
            }
        }
        if constexpr (BAR) __syncthreads();
    }

    // Y = {Re[N/2], Im[N/2]} (bins 0 .. N/2-1, bin N/2 = 0, Im Y[0] ignored) -> y[n] = P0[pidx] + P1[pidx]
    // (P_jt = the partial sum over this wave pair's k2 tile), followed by a barrier.
    template <class WS, bool BAR = true, class OFF = PackedOffsets>
    __device__ static __forceinline__ void inverse(const float* Y, float* P0, float* P1, const float2* tw, const InvConst& c,
                                                   const WS& wb, const OFF& o, int wave, int lane) {
        const int li = lane & 15, lg = lane >> 4, p = wave >> 1, jt = wave & 1, k2 = 16 * jt + li;
        // first stage, transposed: H^T[k2][n1] = sum Y^T[k2][(b, k1)] C[(b, k1)][n1], both halves (columns j = q * MT + i)
        float yq[KC];
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
            const int kl = 4 * ks + lg, b = kl / N1, k1 = kl % N1;
            const int kk = 32 * k1 + k2;
            const int m = kk <= N / 2 ? kk : N - kk;
            const bool ok = m < N / 2;
            const int mm = ok ? m : 0;
            float v = Y[b * (N / 2) + mm];
            if (b == 1) v = (kk > N / 2) ? -v : ((m == 0) ? 0.0f : v);
            yq[ks] = ok ? v : 0.0f;
        }
        f32x4 h[1][2 * MT];
        acc_init_zero<1, 2 * MT>(h);
        mma_panel<1, 2 * MT, KC, PDK>(h, [&](int, int ks) { return yq[ks]; },
                                 [&](int j, int ks) {
                                     if constexpr (PRELOAD3) return c.c3[j][ks];
                                     else return wb.at_g(o.dft3 + (j * KC + ks) * 64);
                                 }, NoSide{});
        // twiddle: element (k2' = 16 jt + 4 lg + r, n1 = 16 i + li) times conj(W_N^(n1 k2'))
        float hq[MT][2][4];
        float2 tq[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) tq[i][r] = twid(tw, (16 * i + li) * (16 * jt + 4 * lg + r));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 t = tq[i][r];
                const float hr = h[0][i][r], hi = h[0][MT + i][r];
                hq[i][0][r] = hr * t.x + hi * t.y;
                hq[i][1][r] = hi * t.x - hr * t.y;
            }
        // second stage: this wave's k2 tile (both halves, k-step a * 4 + r) x the n2 tile p
        float* P = jt ? P1 : P0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f32x4 e0 = {0.0f, 0.0f, 0.0f, 0.0f}, e1 = e0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e0 = FE_MFMA(hq[i][0][r], c.c4[r], e0);
                e1 = FE_MFMA(hq[i][1][r], c.c4[4 + r], e1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) P[pidx(16 * i + 4 * lg + r, 16 * p + li)] = e0[r] + e1[r];
        }
        if constexpr (BAR) __syncthreads();
    }
};

// ------------------------------------------------------------------------------------------
// B fragment (tile j, k-step ks) of a conv unit's weights at w_off (KS_TOT k-steps per tile).  Staged shapes read the LDS copy.  Shapes that
// STREAM their conv weights from L2 (the low-LDS companions, the variants' big shapes) fetch four k-steps per 16-byte load from the k4-regrouped
// copy of the conv units (r4x: a wave-level load costs the vector-memory path the same whatever its width; the GEMM pipelines ask for (j, ks)
// in rising ks per tile, so the load rides on the first k-step of every group of four; the KS_TOT % 4 left-over k-steps are plain in the copy).
template <class S, int NT, int KS_TOT, class WS>
struct ConvB {
    const WS& w;
    int w_off;
    mutable f32x4 cur[NT];
    static constexpr bool K4 = FE_K4_STREAM && !std::is_same_v<std::remove_cv_t<WS>, WSrc<true>> && KS_TOT >= 4;
    __device__ __forceinline__ float operator()(int j, int ks) const {
        constexpr int CK4 = Pack<S>::v.conv_k4_delta;
        if constexpr (K4 && CK4 != 0) {
            const int base = w_off + CK4 + j * (KS_TOT * 64);
            if (ks >= 4 * (KS_TOT / 4)) return w.at_g(base + ks * 64);
            if ((ks & 3) == 0) cur[j] = w.at_gv4(base + (ks >> 2) * 256, w.lane4 * 4);
            return cur[j][ks & 3];
        } else return w.at(w_off + (j * KS_TOT + ks) * 64);
    }
};

// conv-layout GEMM segment: this wave's m-tiles (wave + 4*i) x all NT n-tiles, K = 4*KS.
//   a_lane : LDS pointer to A[(16*wave + (lane&15)) rows][(lane>>4) col] of the segment
//   w_lane : packed weights + lane, at k-step 0 of this segment;  KS_TOT = k-steps per n-tile
template <class S, int NT, int KS, int KS_TOT, int LDA, class WS, class SIDE>
__device__ __forceinline__ void conv_seg(f32x4 (&acc)[S::MTPW][NT], const float* a_lane, const WS& w, int w_off, const SIDE& side) {
    mma_panel<S::MTPW, NT, KS, Lds<S>::PDK>(
        acc,
        [&](int i, int ks) { return a_lane[(64 * i) * LDA + 4 * ks]; },
        ConvB<S, NT, KS_TOT, WS>{w, w_off}, side);
}

// Several K-segments (conv taps / concatenated inputs) accumulated in ONE software pipeline:
// segment s reads its A rows through a_lane[s]; the packed weights hold the segments' k-steps
// back to back (k-step index = s * KS_SEG + ks).
template <class S, int NT, int NSEG, int KS_SEG, int LDA, class WS, class SIDE>
__device__ __forceinline__ void conv_multi(f32x4 (&acc)[S::MTPW][NT], const float* const (&a_lane)[NSEG], const WS& w, int w_off, const SIDE& side) {
    mma_panel<S::MTPW, NT, NSEG * KS_SEG, Lds<S>::PDK>(
        acc,
        [&](int i, int ks) { return a_lane[ks / KS_SEG][(64 * i) * LDA + 4 * (ks % KS_SEG)]; },
        ConvB<S, NT, NSEG * KS_SEG, WS>{w, w_off}, side);
}

// Epilogue of a conv-layout GEMM: optional SiLU, store to out[(row0 + m)][col] for col < NCOLS.
// gskip != nullptr: also store the value into a global skip buffer in A-fragment order
//   gskip[((mt * KS_C + col/4) * 64 + (col%4) * 16 + m%16)],   mt = m / 16
// (what the decoder's 1x1 conv later reads back as coalesced 256-byte A fragments).
template <class S, int NT, int NCOLS, int LDO, bool ACT>
__device__ __forceinline__ void conv_store(const f32x4 (&acc)[S::MTPW][NT], float* out, int row0, int wave, int lane,
                                           float* gskip = nullptr) {
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
                    if (ACT) v = silu_scaled_f(v);
                    out[(row0 + m) * LDO + col] = v;
                    if (gskip != nullptr) gskip[((wave + 4 * i) * S::KS_C + (col >> 2)) * 64 + (col & 3) * 16 + 4 * lg + r] = v;
                }
            }
        }
}

// Column-split conv GEMM for shapes whose conv weights are NOT staged in LDS and whose channel-tile count is a multiple
// of 4 (S, L): this wave's NTC/4 column tiles x ALL row tiles.  Row-split, every wave streams every weight fragment
// from L2 (4x redundant, 8 vector-memory loads per k-step for L); column-split, a wave's B fragments are private and the
// shared operand is the activation tile in LDS (4 LDS + 2 L2 loads per k-step).  af(i, ks): A fragment of row tile i.
// NS = 4: columns over the four waves; NS = 2 (6 channel tiles: M): a 2 x 2 split - wave (wm, wn) takes half the row
// tiles x half the column tiles, weights 2x instead of 4x redundant.
template <class S, int NS, int KS, int NCOLS, int LDO, bool ACT, class AF, class WS, class SIDE>
__device__ __forceinline__ void conv_nsplit(AF&& af, const WS& w, int w_off, int bias_off, const SIDE& side, float* out, int row0,
                                            int wave, int lane, float* gskip) {
    constexpr int MS = kWaves / NS, MT = S::MTC / MS, NTW = S::NTC / NS;
    const int li = lane & 15, lg = lane >> 4;
    const int wn = NS == kWaves ? wave : wave % NS, m0 = NS == kWaves ? 0 : (wave / NS) * MT;   // (literals keep the offsets immediates)
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const f32x4 bj = w.at16x4(bias_off + (wn * NTW + j) * 64);
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][j] = bj;
    }
    constexpr int CK4 = Pack<S>::v.conv_k4_delta;
    if constexpr (FE_K4_STREAM && CK4 != 0 && KS >= 4 && !std::is_same_v<std::remove_cv_t<WS>, WSrc<true>>) {
        // r4x: the streamed conv weights four k-steps per 16-byte load, from the k4-regrouped copy of the conv units (the one the time-batched
        // engine reads; its KS % 4 left-over k-steps are plain)
        f32x4 cur[NTW];
        mma_panel<MT, NTW, KS, Lds<S>::PDK>(acc, [&](int i, int ks) { return af(m0 + i, ks); },
                               [&](int j, int ks) {
                                   const int base = w_off + CK4 + (wn * NTW + j) * (KS * 64);
                                   if (ks >= 4 * (KS / 4)) return w.at_g(base + ks * 64);
                                   if ((ks & 3) == 0) cur[j] = w.at_gv4(base + (ks >> 2) * 256, w.lane4 * 4);
                                   return cur[j][ks & 3];
                               }, side);
    } else
    mma_panel<MT, NTW, KS, Lds<S>::PDK>(acc, [&](int i, int ks) { return af(m0 + i, ks); },
                           [&](int j, int ks) { return w.at(w_off + ((wn * NTW + j) * KS + ks) * 64); }, side);
    side.commit();
#pragma unroll
    for (int ii = 0; ii < MT; ++ii)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int i = m0 + ii;
            const int col = 16 * (wn * NTW + j) + li;
            if (col < NCOLS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * i + 4 * lg + r;
                    float v = acc[ii][j][r];
                    if (ACT) v = silu_scaled_f(v);
                    out[(row0 + m) * LDO + col] = v;
                    if (gskip != nullptr) gskip[(i * S::KS_C + (col >> 2)) * 64 + (col & 3) * 16 + 4 * lg + r] = v;
                }
            }
        }
}

// token-layout GEMM: all MT2 m-tiles x this wave's n-tiles (wave + 4*j), A from LDS, B packed.
template <class S, int NTPW, int KS, int LDA, class WS, class SIDE>
__device__ __forceinline__ void tok_gemm(f32x4 (&acc)[S::MT2][NTPW], const float* a_lane, const WS& w, int w_off, int NT, int wave, const SIDE& side) {
    mma_panel<S::MT2, NTPW, KS, Lds<S>::PDK>(
        acc,
        [&](int i, int ks) { return a_lane[(16 * i) * LDA + 4 * ks]; },
        [&](int j, int ks) {
            int nt = wave + 4 * j;
            nt = nt < NT ? nt : NT - 1;
            return w.at(w_off + (nt * KS + ks) * 64);
        }, side);
}

// Token GEMMs split the OUTPUT columns over the waves, so a wave's B fragments (weights) are private to
// it.  TokW holds the fragments of this wave's column tiles ct = wave + 4j (NG column blocks per tile:
// 3 for the per-gate packed GRU matrices, else 1; fragment tile index = g * NT + ct):
//   REG = true : fetched from L2 into registers by fetch(), one phase before the GEMM that uses them
//   REG = false: (shapes whose fragments do not fit in registers) fetched inside the GEMM's software pipeline
template <int NTPW, int KS, int NG, bool REG, class WS>
struct TokW {
    float w[REG ? NTPW : 1][NG][REG ? KS : 1];
    float bv[REG ? NTPW : 1][NG];
    const WS* src;
    int w_off, b_off, NT, wave;
    int kill;          // 0x40000000: every load of this set is suppressed (reads 0), 0: normal
    // column tiles beyond NT (waves left over when NT is not a multiple of 4) load zeros through an
    // out-of-range buffer offset: no L2 / vector-memory traffic for work whose results are discarded
    __device__ __forceinline__ int oob(int j) const { return wave + 4 * j < NT ? kill : 0x40000000; }
    __device__ __forceinline__ int tile(int j, int g) const {
        int ct = wave + 4 * j;
        ct = ct < NT ? ct : NT - 1;
        return g * NT + ct;
    }
    __device__ __forceinline__ void bind(const WS& s, int w_off_, int b_off_, int NT_, int wave_, bool live = true) {
        src = &s; w_off = w_off_ + (REG ? s.k4d : 0); b_off = b_off_; NT = NT_; wave = wave_; kill = live ? 0 : 0x40000000;
    }
    // Register-resident sets read the k4-regrouped copy of the weights: per tile [KS / 4][lane][4] (four consecutive k-steps of
    // a lane in 16 bytes) then the KS % 4 left-over k-steps as plain [ks][lane].  One fetch element = one load:
    // per (tile, gate) NF 16-byte loads, REM dword loads and the bias dword.
    static constexpr int NF = KS / 4, REM = KS % 4, EPT = NF + REM + 1;
    __device__ __forceinline__ void fetch_elem(int e) {
        const int j = e / (NG * EPT), r = e - j * (NG * EPT);
        const int g = r / EPT, q = r - g * EPT;
        const int t0 = w_off + tile(j, g) * (KS * 64);
        if (q < NF) {
            const f32x4 v = src->at_gv4(t0 + q * 256, src->lane4 * 4 + oob(j));
            w[j][g][4 * q] = v[0]; w[j][g][4 * q + 1] = v[1]; w[j][g][4 * q + 2] = v[2]; w[j][g][4 * q + 3] = v[3];
        } else if (q < NF + REM) {
            const int ks = 4 * NF + (q - NF);
            w[j][g][ks] = src->at_gv(t0 + ks * 64, src->lane4 + oob(j));
        } else {
            bv[j][g] = b_off >= 0 ? src->at_gv(b_off + tile(j, g) * 16, src->li4 + oob(j)) : 0.0f;
        }
    }
    // number of loads fetch_part(part, parts) issues
    static constexpr int part_count(int part, int parts) {
        if (!REG) return 0;
        const int TOT = NTPW * NG * EPT, per = (TOT + parts - 1) / parts;
        int lo = part * per, hi = (part + 1) * per;
        lo = lo < TOT ? lo : TOT;
        hi = hi < TOT ? hi : TOT;
        return hi - lo;
    }
    // slice `part` of `parts` (all indices are compile-time constants once the caller's loops are unrolled)
    __device__ __forceinline__ void fetch_part(int part, int parts) {
        if constexpr (REG) {
            constexpr int TOT = NTPW * NG * EPT;
            const int per = (TOT + parts - 1) / parts;
#pragma unroll
            for (int q = 0; q < TOT; ++q)
                if (q >= part * per && q < (part + 1) * per) fetch_elem(q);
        }
    }
    __device__ __forceinline__ void fetch(const WS& s, int w_off_, int b_off_, int NT_, int wave_) {
        bind(s, w_off_, b_off_, NT_, wave_);
        fetch_part(0, 1);
    }
    // streamed (REG = false): four k-steps of a tile per 16-byte load from the k4 copy - the GEMM pipelines ask for (j, ks) in rising ks per tile, the load
    // rides on the first k-step of every group of four (r4x: a wave-level load costs the vector-memory path the same whatever its width, and the big
    // shapes' block GEMMs issued one per two to four MFMAs)
    mutable f32x4 cur[REG ? 1 : NTPW][REG ? 1 : NG];
    __device__ __forceinline__ float get(int j, int g, int ks) const {
        if constexpr (REG) return w[j][g][ks];
        else if constexpr (FE_K4_STREAM) {
            if (ks >= 4 * (KS / 4)) return src->at_gv(w_off + src->k4d + (tile(j, g) * KS + ks) * 64, src->lane4 + oob(j));
            if ((ks & 3) == 0) cur[j][g] = src->at_gv4(w_off + src->k4d + tile(j, g) * (KS * 64) + (ks >> 2) * 256, src->lane4 * 4 + oob(j));
            return cur[j][g][ks & 3];
        } else return src->at_gv(w_off + (tile(j, g) * KS + ks) * 64, src->lane4 + oob(j));
    }
    __device__ __forceinline__ float bias(int j, int g) const {
        if constexpr (REG) return bv[j][g];
        else return b_off >= 0 ? src->at_gv(b_off + tile(j, g) * 16, src->li4 + oob(j)) : 0.0f;
    }
};

// acc = bias + A(LDS tokens) x W  for this wave's column tiles
template <class S, int NTPW, int KS, int LDA, class TW, class SIDE = NoSide>
__device__ __forceinline__ void tok_gemm_w(f32x4 (&acc)[S::MT2][NTPW], const float* a_lane, const TW& W, SIDE side = SIDE{}) {
#pragma unroll
    for (int j = 0; j < NTPW; ++j) {
        const float bj = W.bias(j, 0);
#pragma unroll
        for (int i = 0; i < S::MT2; ++i) acc[i][j] = f32x4{bj, bj, bj, bj};
    }
    mma_panel<S::MT2, NTPW, KS, Lds<S>::PDK>(
        acc, [&](int i, int ks) { return a_lane[(16 * i) * LDA + 4 * ks]; }, [&](int j, int ks) { return W.get(j, 0, ks); }, side);
}

// Attention of one head for the query tiles q0, q0 + qstride, ... (NQ of them; tiles >= MT2 are skipped):
//   S^T[key][query] = K Q^T, softmax over keys in registers, O^T = V^T P^T with the C/D row map as the k-permutation.
// G holds q | k | v of the head at columns hoff, hoff + HD, hoff + 2 HD (row stride LDG); O -> Hl[query][head * HD + d].
template <class S, int NQ, int LDG, int PDK = Lds<S>::PDK>
__device__ __forceinline__ void attention_head(const float* G, float* Hl, int hoff, int head, int q0, int qstride, int lane) {
    constexpr int HD = S::HD, F2 = S::F2, LDX = S::LDX;
    constexpr int KSD = ceil_div(HD, 4);
    constexpr int MTD = ceil_div(HD, 16);
    const int li = lane & 15, lg = lane >> 4;
    int qt[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) { qt[j] = q0 + qstride * j; qt[j] = qt[j] < S::MT2 ? qt[j] : S::MT2 - 1; }
    f32x4 sacc[S::MT2][NQ];
    acc_init_zero<S::MT2, NQ>(sacc);
    mma_panel<S::MT2, NQ, KSD, PDK>(
        sacc,
        [&](int i, int ks) {
            // (K side unmasked: for d >= HD it reads the head's first v columns - finite values that meet the Q side's zeros)
            return G[(16 * i + li) * LDG + hoff + HD + 4 * ks + lg];
        },
        [&](int j, int ks) {
            const int d = 4 * ks + lg;
            float v = G[(16 * qt[j] + li) * LDG + hoff + (d < HD ? d : HD - 1)];
            return d < HD ? v : 0.0f;
        }, NoSide{});
    // the V fragments of the P V product: requested now, in flight under the softmax (they do not depend on it)
    float av[S::MT2][4][MTD];
#pragma unroll
    for (int i = 0; i < S::MT2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int key = 16 * i + 4 * lg + r;
            key = key < F2 ? key : F2 - 1;
#pragma unroll
            for (int md = 0; md < MTD; ++md) {
                int d = 16 * md + li;
                d = d < HD ? d : HD - 1;
                av[i][r][md] = G[key * LDG + hoff + 2 * HD + d];
            }
        }
    const float scale = rsqrtf((float)HD) * 1.4426950408889634f;      // 1/sqrt(hd) * log2(e): softmax through exp2
    float inv_sum[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < S::MT2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * i + 4 * lg + r;
                float sv = sacc[i][j][r];                                    // raw score: the (positive) scale commutes with the max
                if (16 * i + 15 >= F2) sv = key < F2 ? sv : -INFINITY;      // (only the last key tile has padding rows)
                sacc[i][j][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = rows_allreduce(mx, [](float p, float q) { return fmaxf(p, q); }) * scale;
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < S::MT2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[i][j][r], scale, -mx));
                sacc[i][j][r] = p;
                sum += p;
            }
        sum = rows_allreduce(sum, [](float p, float q) { return p + q; });
        inv_sum[j] = __builtin_amdgcn_rcpf(sum);      // applied to O below: the P V MFMAs need not wait for it
    }
    f32x4 oacc[MTD][NQ];
    acc_init_zero<MTD, NQ>(oacc);
    // k-step (i, r): lane group lg supplies key = 16 i + 4 lg + r  (matches the C/D row map)
#pragma unroll
    for (int i = 0; i < S::MT2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int md = 0; md < MTD; ++md)
#pragma unroll
                for (int j = 0; j < NQ; ++j) oacc[md][j] = FE_MFMA(av[i][r][md], sacc[i][j][r], oacc[md][j]);
    // O[query][head*HD + d]  (into Hl, dead after rnn_fc)
#pragma unroll
    for (int md = 0; md < MTD; ++md)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int q = 16 * qt[j] + li;
            const bool live = q0 + qstride * j < S::MT2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * md + 4 * lg + r;
                if (live && d < HD && q < F2) Hl[q * LDX + head * HD + d] = oacc[md][j][r] * inv_sum[j];
            }
        }
}

// ------------------------------------------------------------------------------------------
// a.mode (wave-uniform): FE_MODE_STREAM  wav->wav streaming step (scripts/export_onnx.py:48-58)
//                        FE_MODE_SPEC    spec->spec step (model.py:677-710)
//                        FE_MODE_OFFLINE Model.forward (model.py:728-735): centered STFT of the whole signal,
//                                        zero initial GRU state, torch.istft-style normalised overlap-add
// DBG = false (the production instantiation): the per-stage debug dumps and cycle probes are compiled out - they
// cost a scalar test + branch each (~40 per frame) and keep their pointers alive in SGPRs for the whole kernel.
// T1: a.T == 1 is a compile-time fact.  MODE >= 0: the mode is a compile-time constant (the streaming step gets its own instantiation, without the
// spec / offline branches and their arguments); MODE = -1: a.mode decides at run time.
// PERSIST: the grid is smaller than the batch (more streams than CUs): each workgroup walks the streams blockIdx.x,
// blockIdx.x + gridDim.x, ... and keeps what does not depend on the stream - twiddles, zeroed halos, the staged weight
// pipeline (the last phase of a stream's last frame stages unit 0 for the next stream) - instead of paying the kernel
// prologue and a workgroup launch per stream.  PERSIST = false: one stream per workgroup, no stream loop.
// PIPE (offline / spec -> spec with T >> 1): the frames of ONE stream are spread over P co-resident workgroups (one per
// CU), workgroup p running frames p, p + P, ...  Everything in a frame but the GRU state is independent of the other
// frames, so the workgroups run their frames concurrently, staggered by the one true dependency: frame t's GRU in block
// k needs h_k(t-1).  The producer publishes it through global memory: agent-scope (sc1) stores of the state, drained with
// s_waitcnt vmcnt(0) + the phase barrier, then a RELAXED agent-scope store of a per-(stream, block) frame counter; the consumer polls
// the counter (relaxed, agent scope) and then fetches the state with agent-scope loads.  INVARIANT: every datum handed from one
// workgroup to another goes through st_state / ld_state - a plain load or store added to a ring or state would be a silent stale read
// (no release / acquire fence covers it).  The serial chain is
// T x (state round trip + one GRU phase) instead of T x (whole frame): ~12 frames in flight for FastEnhancer_B.
template <class S, bool DBG, int MODE, bool T1, bool PERSIST, bool PIPE = false>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(Lds<S>::OCC, Lds<S>::OCC))) fe_frame_kernel(FrameArgs a_in) {
    FrameArgs a = a_in;
#ifdef FE_PROBE_HOT          // measurement builds: the production instantiations keep the cycle probes (tools/gpu_phases.py ... 1)
    if constexpr (!DBG) a.dbg = nullptr;
#else
    if constexpr (!DBG) { a.dbg = nullptr; a.clk = nullptr; }
#endif
    if constexpr (MODE >= 0) a.mode = MODE;
    if constexpr (MODE == FE_MODE_STREAM) { a.spec_in = nullptr; a.spec_out = nullptr; a.Tw = 0; }
    FE_CLK(62);                                      // kernel entry (tools/gpu_phases.py: prologue = probe 0 - probe 62)
    if constexpr (T1) a.T = 1;                       // one hop per launch (the per-hop driver loop): no frame loop, and the
                                                     // last GEMM phase does not stage weights for a next frame
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = Lds<S>;
    constexpr int N = S::NFFT, H = S::HOP, OVL = S::OVL, F0 = S::F0, F1 = S::F1;
    constexpr int C1 = S::C1, C2 = S::C2, F2 = S::F2, HD = S::HD;
    constexpr int LDC = S::LDC, LDX = S::LDX, LDG = L::LDGX;

    const int tid0 = threadIdx.x;
    const int tid = tid0;
    const int lane0 = tid & 63;
    const int lane = lane0;
    const int wave0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave0;
    const int li = lane & 15, lg = lane >> 4;
    const float* __restrict__ wp = a.wp;
    WSrc<Lds<S>::STAGED> wb;
    constexpr PackedOffsets o = Pack<S>::v;
    static_assert(o.n_units == S::NU, "unit count");
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, o.total * 4, 0x00020000);
    wb.lane4 = lane * 4;
    wb.li4 = (lane & 15) * 4;
    wb.lds = nullptr;
    wb.base = 0;
    wb.k4d = o.k4_delta;

    // ---- one-time: zero what must be zero - the 2-bin halo of the compressed spectrum and the halo rows of the
    // LDS-resident skip buffers (the work buffers' halos are re-zeroed per frame).  Everything else in LDS is either
    // written before it is read, or only ever feeds MFMA rows / columns whose results are discarded (pad rows of the
    // token buffers, pad columns), so it may start out as garbage.
    for (int i = tid; i < 2 * S::LDS_S; i += kThreads) smem[L::SC + i] = 0.0f;
    if constexpr (L::SKIPS_LDS) {
        for (int i = tid; i < (S::NL + 1) * 2 * LDC; i += kThreads) {
            const int e = i / (2 * LDC), q = i - e * (2 * LDC);
            smem[L::E + e * S::ACT + (q >= LDC ? (F1 + 1) * LDC + (q - LDC) : q)] = 0.0f;
        }
    }
    if constexpr (S::KT > 1) {                       // frame-cache arena: its halo rows must read as zero
        for (int i = tid; i < (S::KT - 1) * S::ACT; i += kThreads) smem[L::CA + i] = 0.0f;
    }
    float2* tw = reinterpret_cast<float2*>(smem + L::TW);
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = reinterpret_cast<const float2*>(wp + o.twiddle)[i];
    float* sc = smem + L::SC;
    float2* fa = reinterpret_cast<float2*>(smem + L::FFT_A);
    float2* fb = reinterpret_cast<float2*>(smem + L::FFT_B);
    float* Ebuf = smem + L::E;
    constexpr bool SG = !L::SKIPS_LDS;               // skips in the global scratch
    // column-split (NS = 4) or 2 x 2-split (NS = 2) conv GEMMs for the shapes whose conv weights are not staged (see conv_nsplit)
    constexpr int NS = L::STAGED ? 1 : (S::NTC % 4 == 0 ? 4 : ((S::NTC % 2 == 0 && S::MTC % 2 == 0) ? 2 : 1));
    constexpr bool NSPLIT = NS > 1;
    constexpr int SKIP_FLOATS = F1 * C1;             // one skip tensor in A-fragment order
    float* W0 = smem + L::W0;
    float* W1 = smem + L::W1;
    float* skipg = SG ? a.skip + (size_t)blockIdx.x * ((S::NL + 1) * SKIP_FLOATS) : nullptr;      // per workgroup, not per stream
    WSrc<false> skb;                                 // the same scratch as a buffer resource (coalesced fragment reads)
    skb.rsrc = __builtin_amdgcn_make_buffer_rsrc(SG ? skipg : const_cast<float*>(a.wp), 0, SG ? (S::NL + 1) * SKIP_FLOATS * 4 : 4, 0x00020000);
    skb.lane4 = lane * 4;
    skb.li4 = (lane & 15) * 4;
    skb.lds = nullptr;
    skb.base = 0;
    // LDS buffer that holds the output of enc_pre (l = 0) / encoder layer l-1; with global skips the encoder
    // ping-pongs through W0/W1 such that the last output lands in W0
    auto encbuf = [&](int l) -> float* {
        if constexpr (!SG) return Ebuf + l * S::ACT;
        else return ((l + (S::NL & 1)) & 1) ? W1 : W0;
    };
    // ln variant: norm site `site` (Shape::LN_SITES order) over a [ROWS x COLS] LDS tile, GroupNorm form
    float* const lnred = smem + L::LNS;
#define FE_LN_SITE(ROWS, COLS, LD, ACT, buf, site) ln_pass<ROWS, COLS, LD, ACT, false>(buf, lnred, wp + lz + o.ln_g[site], wp + lz + o.ln_b[site], 1.0e-5f)   /* (+ lz: not hoisted out of the frame loop) */
    static_assert(!S::LN || (L::STAGED && L::SKIPS_LDS && S::KT == 1 && !S::FRNN && !S::TATT), "ln variant: built for the B-type plan (staged weights, LDS skips)");
    // weight units: unit U of frame t is consumed from LDS buffer ((U + t*NU) & 1) while the next streams in
    constexpr int NPW = L::STAGED ? ceil_div(ceil_div(Pack<S>::umax(), 256), kWaves) : 1;
    DmaJobT<NPW> job;
    job.l = smem;
    job.rsrc = wb.rsrc;
    job.soff = 0;
    job.wave = wave;
    job.lane = lane;
    if constexpr (L::STAGED) {
        job.l = smem + L::WB0;
        job.soff = (o.u_off[0] + wave * 256) * 4;
        const StageSide<NPW, o.u_size[0] / 256> st0{&job};
        st0(0, 1);
        st0.commit();
    }
    __syncthreads();

    int b = PIPE ? (int)blockIdx.x / a.pipe_p : (int)blockIdx.x;
    const int t_first = PIPE ? (int)blockIdx.x - b * a.pipe_p : 0, t_step = PIPE ? a.pipe_p : 1;
    int fc = 0;                                      // frames done by this workgroup (staging-buffer parity)
    // GRU state exchange between the workgroups of a stream (PIPE): agent-scope accesses (not served from a stale L1 / L2 line)
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
    float* cst = a.cache_stft + (size_t)b * OVL;
    float* cis = a.cache_istft + (size_t)b * OVL;
    constexpr int NFLAG = S::KB + (S::KT > 1 ? 2 * S::NL : 0);       // counters per stream
    unsigned int* pflag = PIPE ? a.pipe_flags + (size_t)b * NFLAG : nullptr;
    // wait until the block-k state of frame t-1 is published (all threads call; thread 0 polls)
    auto pipe_wait = [&](int k, int t) {
        if constexpr (PIPE) {
            if (t > 0) {
                if (tid == 0) {
                    while (__hip_atomic_load(pflag + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)t) __builtin_amdgcn_s_sleep(1);
                }
                __syncthreads();
            }
        }
    };
    // publish frame t's block-k state: every thread's state stores have left the CU, then one release store of the counter
    auto pipe_publish = [&](int k, int t) {
        if constexpr (PIPE) {
            __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0) (gfx9 encoding: lgkmcnt / expcnt left alone)
            __syncthreads();
            if (tid == 0) __hip_atomic_store(pflag + k, (unsigned int)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    int ring_head0 = 0;                              // dptransformer: slot of the oldest cached frame when this launch starts
    if constexpr (S::TATT && !PIPE) ring_head0 = (int)a.h[(size_t)a.B * S::KB * S::HSTATE + b];
    // dptransformer, per-hop launch with one stream per workgroup (r4): the K / V window of a block - 31 cached frames x [F2][C2] x 2,
    // 214 KB per stream and block for B, the whole HBM traffic of this HBM-bound step - is fetched into REGISTERS (4 HD floats per lane
    // and round of sixteen (sub-band, head) pairs: 216 of the 342 the lone wave per SIMD has left) in pieces issued at the END of the
    // phases that precede its use: block 0's under STFT / encoder / rf_pre, block k + 1's under block k's fc / qkv / attention phases,
    // piece by piece so that a phase's own weight fetches (older in the in-order vmcnt queue) never wait for more than one piece.
    // The time attention itself then runs from registers; what stays exposed is the part of the stream that does not fit under the
    // compute in between (kvw_issue / FE_KVW below, phase B of the blocks).
#ifndef FE_KVW_PREFETCH
#define FE_KVW_PREFETCH 1
#endif
    constexpr int W_PAIRS = F2 * S::NH;
    constexpr bool WPF = FE_KVW_PREFETCH && S::TATT && !PIPE && !DBG && T1 && !PERSIST && W_PAIRS % 16 == 0 && (W_PAIRS / 16) * 4 * HD <= 224;
    constexpr int W_NIT = WPF ? W_PAIRS / 16 : 1;
#ifndef FE_KVW_NB
#define FE_KVW_NB 2
#endif
#ifndef FE_KVW_FRONT
#define FE_KVW_FRONT 0
#endif
    constexpr int W_NB = FE_KVW_NB;                  // pieces of the next block's window issued inside phase B itself
    // front schedule of block 0's window: piece issued after {frame load, DFT, compress, enc_pre}, the encoder layers take the next NL
    constexpr int W_F0 = 0, W_F1 = FE_KVW_FRONT == 0 ? 1 : -1, W_F2 = FE_KVW_FRONT == 0 ? 2 : 1, W_F3 = FE_KVW_FRONT == 0 ? 3 : 2;
    constexpr int W_FE = W_F3 + 1;
    float kvw[W_NIT][WPF ? 4 * HD : 1];
#pragma unroll 1
    for (int t = t_first; t < a.T; t += t_step, ++fc) {
        // A loop-variant zero keeps the (many) wave-uniform offsets of a frame from being hoisted out of the frame
        // loop: hoisted, they sit in SGPRs for the whole kernel and spill to VGPR lanes by the hundred.
        // (Measured on the kernels with a frame loop: FastEnhancer_B 46.0 -> 44.1 us per frame, T 21.2 -> 20.5 us,
        //  S +29 %, M +1 %; L - whose frames re-derive far more offsets than it has SGPRs to save - loses 7 %.)
        int lz = 0;
        // r4: the big shapes (C1 > 96: L, 48 kHz L, dprnn L) get it too in the instantiations that HAVE a frame / stream loop (persistent,
        // generic, time-pipelined) - r1's "L loses 7 %" was measured when one instantiation served everything.  FastEnhancer_L:
        // SGPR spills 2322 / 2371 / 2380 -> 37 / 93 / 119, VGPR spills 114 / 205 / 75 -> 0 (with the per-lane zero below);
        // 512 / 1024 streams 60.2 / 60.8 % -> 65.3 / 65.4 % of the fp32 peak, 16 x 4 s offline (frame walk) 16.1 -> 15.5 ms.
#ifndef FE_LZ_BIG
#define FE_LZ_BIG 1      // (0: r3's behaviour, for A/B builds)
#endif
#ifndef FE_LZ_T1
#define FE_LZ_T1 1      // r4w: the big shapes' per-hop instantiation too - it has no loop, but its offsets were all derived at kernel entry and
#endif                  // parked in spilled SGPRs (FastEnhancer_L: 1022 SGPR spills -> 0; 256 streams 377 -> 364 us, 0.645 -> 0.669).  0: r4d's behaviour
        if constexpr (S::C1 <= 96 || (FE_LZ_BIG && (PERSIST || PIPE || !T1 || FE_LZ_T1))) asm volatile("" : "+s"(lz));
        const int wave = wave0 + lz;
        // LOW = 2 companions (256 VGPRs, operands streamed from L2): the same for the per-lane offsets - hoisted out of the
        // stream loop they stay live through the whole frame and the persistent instantiation spills (S: 167 -> 92 VGPRs,
        // 48 kHz B: 73 -> 14; +12 % / +15 % at 1024 streams).  Kernels that fit anyway pay for the re-derived offsets
        // (B companion -3 %, T -7 %): not applied there.
        int lzv = 0;
        if constexpr ((PERSIST && S::LOW == 2) || (FE_LZ_BIG && S::C1 > 96 && (PERSIST || PIPE || !T1))) asm volatile("" : "+v"(lzv));
        const int tid = tid0 + lzv;
        const int lane = lane0 + lzv;
        const int li = lane & 15, lg = lane >> 4;
        // (WPF) this lane's two window slots - positions l16 and l16 + 16 of the ring, oldest first - as float offsets into the
        // [pair][LB][HD] cache of a block; kvw_issue(block, piece): k0 | k1 | v0 | v1 of the pairs 16 piece + (tid >> 4)
        int w_off0 = 0, w_off1 = 0;
        if constexpr (WPF) {
            const int head = (ring_head0 + t) % S::LB;
            const int l16 = tid & 15, j1 = l16 + 16;
            int sl0 = head + l16, sl1 = head + (j1 < S::LB ? j1 : S::LB - 1);
            sl0 = sl0 >= S::LB ? sl0 - S::LB : sl0;
            sl1 = sl1 >= S::LB ? sl1 - S::LB : sl1;
            w_off0 = ((tid >> 4) * S::LB + sl0) * HD;
            w_off1 = ((tid >> 4) * S::LB + sl1) * HD;
        }
        auto kvw_issue = [&](int kblk, auto it_) {
            constexpr int it = decltype(it_)::value;
            if constexpr (WPF && it < W_NIT) {
                if (kblk < S::KB) {
                    const size_t cstride = (size_t)F2 * C2 * S::LB;
                    const float* kc = a.h + ((size_t)(2 * kblk) * a.B + b) * cstride + (size_t)(16 * it) * (S::LB * HD);
                    const float* vc = kc + (size_t)a.B * cstride;
                    // (plain loads: the rings of 256 streams - 164 MB for B - live in the 256 MiB Infinity Cache from launch to launch; non-temporal
                    //  loads measured 52.5 -> 76.6 us, profiles/r4u_*)
#pragma unroll
                    for (int d = 0; d < HD; ++d) kvw[it][d] = kc[w_off0 + d];
#pragma unroll
                    for (int d = 0; d < HD; ++d) kvw[it][HD + d] = kc[w_off1 + d];
#pragma unroll
                    for (int d = 0; d < HD; ++d) kvw[it][2 * HD + d] = vc[w_off0 + d];
#pragma unroll
                    for (int d = 0; d < HD; ++d) kvw[it][3 * HD + d] = vc[w_off1 + d];
                }
            }
        };
#define FE_KVW(KBLK, IT) kvw_issue((KBLK), std::integral_constant<int, (IT)>{})
        // pieces FROM .. W_NIT - 1 (the last issue point before a window is used takes whatever is left)
        auto kvw_issue_rest = [&](int kblk, auto from_) {
            static_for<W_NIT>([&](auto c_) {
                if constexpr (decltype(c_)::value >= decltype(from_)::value) kvw_issue(kblk, c_);
            });
        };
        // begin_unit(U): called right after the barrier that precedes the GEMM phase of staged unit U:
        // selects the LDS copy of this phase's weights and sets up the DMA job of the next unit.
        const int fpar = (S::NU & 1) ? (((PERSIST || PIPE) ? fc : t) & 1) : 0;
#define FE_BEGIN_UNIT(U)                                                                           \
        constexpr int fe_un_ = ((U) + 1 == S::NU) ? 0 : (U) + 1;                                   \
        if constexpr (L::STAGED) {                                                                 \
            const int slot_ = ((U) & 1) ^ fpar;                                                    \
            job.l = smem + (slot_ ? L::WB0 : L::WB1);                                              \
            job.soff = (o.u_off[fe_un_] + wave * 256) * 4;                                         \
            wb.lds = smem + (slot_ ? L::WB1 : L::WB0);                                             \
            wb.base = o.u_off[(U)];                                                                \
        }                                                                                          \
        const StageSide<NPW, (L::STAGED && !(T1 && !PERSIST && !PIPE && (U) + 1 == S::NU)) ? o.u_size[fe_un_] / 256 : 0> stage{&job}
        FE_CLK(0);
        // =========================== STFT (a3) ===========================
        const int mode = a.mode;
        // LDS quarters of the FFT arena (N floats each): q0 windowed frame, q1 raw frame (cache shift), q3 spectrum
        // {Re[N/2], Im[N/2]};  iSTFT: q3 spectrum -> q0, q1 partial sums of y -> q2 windowed / overlap-added frame
        float* q0 = reinterpret_cast<float*>(fa);
        float* q1 = q0 + N;
        float* q2 = reinterpret_cast<float*>(fb);
        float* q3 = q2 + N;
        // matrix-core DFT (N = 512, and N = 1024 for all but the largest shape; r2 same-box A/B against the radix-2 LDS FFT
        // with its 20 barrier-separated stages per frame: 48 kHz T +9 %, B +3 % (+4 % at hop 480), S +2.8 %, M +0.6 %,
        // L -0.2 % - there the 64 + 32 constant fragments per wave meet a kernel that already spills.  In r1, before the
        // per-hop instantiation freed its registers, the same switch had cost 48 kHz B 82 -> 91 us.)
        constexpr bool MDFT = (N == 512) || (S::C1 < 128);
        constexpr int XS = MDFT ? 1 : 2;                 // element stride of the spectrum arrays below
        const float* Xr = nullptr;
        const float* Xi = nullptr;
        if (mode != FE_MODE_SPEC) {
            typename Dft<S>::FwdConst dc;
            if constexpr (MDFT) Dft<S>::load(dc, wb, o, wave);            // in flight while the frame is fetched
            const float* win = wp + o.window;
            constexpr int NPT = N / kThreads;              // samples per thread; all global loads are issued before the
            float fv[NPT], fw[NPT];                        // first LDS store (one memory round trip, not NPT)
            if (mode == FE_MODE_STREAM) {
                const float* xin = a.wav_in + (size_t)b * a.in_stride + (size_t)t * H;
#pragma unroll
                for (int q = 0; q < NPT; ++q) {
                    const int n = tid + q * kThreads;
                    fv[q] = (n < OVL) ? cst[n] : xin[n - OVL];
                    fw[q] = win[n];
                }
#pragma unroll
                for (int q = 0; q < NPT; ++q) {
                    const int n = tid + q * kThreads;
                    if constexpr (MDFT) {
                        q1[n] = fv[q];                     // raw frame kept for the cache shift
                        q0[n] = fv[q] * fw[q];
                    } else {
                        fb[n] = make_float2(fv[q], 0.0f);
                        fa[n] = make_float2(fv[q] * fw[q], 0.0f);
                    }
                }
            } else {
                // torch.stft(center=True, pad_mode="reflect") (functional/audio_modules.py:78-80): frame t covers
                // xp[tH : tH+N], xp = reflect_pad(x, N/2)
                const float* xin = a.wav_in + (size_t)b * a.in_stride;
#pragma unroll
                for (int q = 0; q < NPT; ++q) {
                    const int n = tid + q * kThreads;
                    int idx = t * H + n - N / 2;
                    idx = idx < 0 ? -idx : idx;
                    idx = idx >= a.Tw ? 2 * (a.Tw - 1) - idx : idx;
                    fv[q] = xin[idx];
                    fw[q] = win[n];
                }
#pragma unroll
                for (int q = 0; q < NPT; ++q) {
                    if constexpr (MDFT) q0[tid + q * kThreads] = fv[q] * fw[q];
                    else fa[tid + q * kThreads] = make_float2(fv[q] * fw[q], 0.0f);
                }
            }
            __syncthreads();
            FE_KVW(0, W_F0);
            if (mode == FE_MODE_STREAM) {
                for (int m = tid; m < OVL; m += kThreads) cst[m] = MDFT ? q1[m + H] : fb[m + H].x;   // cache' = frame[H:]
                if constexpr (!MDFT) __syncthreads();       // (the FFT's first stage overwrites fb)
            }
            FE_CLK(1);
            if constexpr (MDFT) {
                float* nyq = a.dbg ? a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(0) + 2 * F0 : nullptr;
                Dft<S>::forward(q0, q3, tw, dc, wave, lane, nyq);
                Xr = q3;
                Xi = q3 + N / 2;
            } else {
                const float2* X = fft_lds<S, false>(fa, fb, tw);
                Xr = &X[0].x;
                Xi = &X[0].y;
            }
            FE_CLK(2);
            if constexpr (W_F1 >= 0) FE_KVW(0, W_F1);
            // spectrum bins 0..F0 (F0 = Nyquist, dropped by the model)
            if (a.dbg) {
                float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(0);
                for (int f = tid; f < F0 + (MDFT ? 0 : 1); f += kThreads) { dst[2 * f] = Xr[f * XS]; dst[2 * f + 1] = Xi[f * XS]; }
            }
            // =========================== compress (a4) ===========================
            for (int f = tid; f < F0; f += kThreads) {
                float re = Xr[f * XS], im = Xi[f * XS];
                float mag = fmaxf(sqrtf(re * re + im * im), 1.0e-5f);
                float g = pow_f(mag, a.compression - 1.0f);
                sc[2 + f] = re * g;
                sc[S::LDS_S + 2 + f] = im * g;
            }
        } else {
            const float* sp = a.spec_in + (size_t)b * (F0 + 1) * a.T * 2;
            FE_KVW(0, W_F0);
            if constexpr (W_F1 >= 0) FE_KVW(0, W_F1);
            for (int f = tid; f < F0; f += kThreads) {
                float re = sp[((size_t)f * a.T + t) * 2], im = sp[((size_t)f * a.T + t) * 2 + 1];
                float mag = fmaxf(sqrtf(re * re + im * im), 1.0e-5f);
                float g = pow_f(mag, a.compression - 1.0f);
                sc[2 + f] = re * g;
                sc[S::LDS_S + 2 + f] = im * g;
            }
        }
        FE_KVW(0, W_F2);
        __syncthreads();
        if (a.dbg) {
            float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(1);
            for (int f = tid; f < F0; f += kThreads) { dst[2 * f] = sc[2 + f]; dst[2 * f + 1] = sc[S::LDS_S + 2 + f]; }
        }

        FE_CLK(3);
        // =========================== enc_pre (a5): strided conv as K=16 GEMM ===========================
        {
            FE_BEGIN_UNIT(0);
            f32x4 acc[S::MTPW][S::NTC];
            acc_init_bias<S::MTPW, S::NTC>(acc, wb, o.enc_pre_b, 0, 1, S::NTC);
            // k = t*8 + s*2 + c  (weight (C1, 8, 2): channel index s*2+c, tap t);  A[m][k] = xpad[c][4(m+t)+s]
            mma_panel<S::MTPW, S::NTC, 4, Lds<S>::PDK>(
                acc,
                [&](int i, int ks) {
                    const int kk = 4 * ks + lg;
                    const int c = kk & 1, s = (kk >> 1) & 3, tp = kk >> 3;
                    const int m = 16 * (wave + 4 * i) + li;
                    return sc[c * S::LDS_S + 4 * (m + tp) + s];
                },
                ConvB<S, S::NTC, 4, decltype(wb)>{wb, o.enc_pre_w}, stage);
            stage.commit();
            if constexpr (SG) {   // the arena was used by the FFT: restore the zero halo rows of both ping-pong buffers
                for (int i = tid; i < 4 * LDC; i += kThreads) {
                    const int q = i / LDC, c = i - q * LDC;
                    ((q & 2) ? W1 : W0)[((q & 1) ? F1 + 1 : 0) * LDC + c] = 0.0f;
                }
            }
            conv_store<S, S::NTC, C1, LDC, !S::LN>(acc, encbuf(0), 1, wave, lane, SG ? skipg : nullptr);
        }
        FE_KVW(0, W_F3);
        __syncthreads();
        if constexpr (S::LN) { FE_LN_SITE(F1, C1, LDC, true, encbuf(0) + LDC, 0); __syncthreads(); }
        dbg_dump<S>(a, b, 2, encbuf(0) + LDC, LDC);

        // ---- time_kernel variant (CausalConv2d, models/fastenhancer/time_kernel/model.py:119-148): a k = 3 conv over frequency
        // with KT taps over time = KT staged sub-phases accumulating into the same registers.  Tap 0 reads the current
        // frame (`in`, already in LDS); meanwhile the conv's cache - its input of the previous KT-1 frames, [KT-1][F1][C1]
        // per stream in the state - streams from HBM into registers (in flight under tap 0's MFMAs), is parked in the
        // cache arena (activation layout) and feeds taps 1 .. KT-1.  The new cache (old slots shifted by one, this
        // frame's input appended) is written back from the same registers / from `in`.
        auto k3_time = [&](auto u0_, const float* in, float* out, int lidx, const int* w_off, int b_off) {
            if constexpr (S::KT > 1) {
            constexpr int U0 = decltype(u0_)::value;
            constexpr int KT = S::KT, Q4 = F1 * C1 / 4;          // float4s per cache slot
            constexpr int CPT = (KT - 1) * Q4 / kThreads, NPT = Q4 / kThreads;
            static_assert(Q4 % kThreads == 0, "cache slots are moved by whole float4 rounds");
            float* CAb = smem + L::CA;
            float4* tkg = reinterpret_cast<float4*>(a.tk + ((size_t)lidx * a.B + b) * S::TKQ);
            // PIPE (time-pipelined offline launch): the conv's input of every frame goes through a RING of pipe_p + KT - 1 slots per
            // (conv, stream) in a.tk instead of the two-slot cache - frame t publishes its input in slot t mod RS and counts it in
            // pflag[KB + conv]; frames t + 1 .. t + KT - 1 wait for that count and fetch the slot (agent-scope accesses on both sides).
            // Slot t mod RS is overwritten by frame t + RS, run by the workgroup of frame t + KT - 1 after that frame: every reader is done.
            const int RS = PIPE ? a.pipe_p + KT - 1 : 1;
            float* ringb = a.tk + (((size_t)lidx * a.B + b) * RS) * (size_t)(F1 * C1);
            f32x4 acc[S::MTPW][S::NTC];
            float4 creg[CPT];
            static_for<KT>([&](auto s_) {
                constexpr int sub = decltype(s_)::value;
                FE_BEGIN_UNIT(U0 + sub);
                if constexpr (sub == 0) {
                    acc_init_bias<S::MTPW, S::NTC>(acc, wb, b_off, 0, 1, S::NTC);
                    if constexpr (PIPE) {
                        if (t > 0) {
                            if (tid == 0) {
                                while (__hip_atomic_load(pflag + S::KB + lidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)t) __builtin_amdgcn_s_sleep(1);
                            }
                            __syncthreads();
                        }
#pragma unroll
                        for (int q = 0; q < CPT; ++q) {
                            const int e4 = tid + q * kThreads, slot = e4 / Q4, r = e4 - slot * Q4;
                            const int fr = t + a.tk_base - (KT - 1) + slot;        // slot 0 = the oldest frame of the window (ring index)
                            const float* src = ringb + (size_t)((fr < 0 ? 0 : fr) % RS) * (F1 * C1) + 4 * r;
                            float4 v;
                            v.x = ld_state(src); v.y = ld_state(src + 1); v.z = ld_state(src + 2); v.w = ld_state(src + 3);
                            creg[q] = fr < 0 ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : v;      // (before the utterance: the zero cache of Model.forward)
                        }
                    } else {
#pragma unroll
                    for (int q = 0; q < CPT; ++q) creg[q] = tkg[tid + q * kThreads];
                    }
                }
                const float* src = sub == 0 ? in : CAb + (KT - 1 - sub) * S::ACT;     // tap 1 <-> frame t-1 = the newest slot
                const float* const taps[3] = {src + (16 * wave + li + 0) * LDC + lg, src + (16 * wave + li + 1) * LDC + lg,
                                              src + (16 * wave + li + 2) * LDC + lg};
                conv_multi<S, S::NTC, 3, S::KS_C, LDC>(acc, taps, wb, w_off[sub], stage);
                stage.commit();
                if constexpr (sub == 0) {
#pragma unroll
                    for (int q = 0; q < CPT; ++q) {
                        const int e4 = tid + q * kThreads, slot = e4 / Q4, r = e4 - slot * Q4;
                        const int f = (4 * r) / C1, c = 4 * r - f * C1;
                        float2* d2 = reinterpret_cast<float2*>(CAb + slot * S::ACT + (f + 1) * LDC + c);      // (rows are 8-byte aligned)
                        // (the state holds the reference's activations; the conv trunk works on them scaled by kSiluScale)
                        d2[0] = make_float2(creg[q].x * kSiluScale, creg[q].y * kSiluScale);
                        d2[1] = make_float2(creg[q].z * kSiluScale, creg[q].w * kSiluScale);
                        if constexpr (!PIPE) { if (slot >= 1) tkg[e4 - Q4] = creg[q]; }      // shift: slot j <- old slot j + 1
                    }
#pragma unroll
                    for (int q = 0; q < NPT; ++q) {                                 // newest slot <- this frame's input
                        const int r = tid + q * kThreads, f = (4 * r) / C1, c = 4 * r - f * C1;
                        const float2* s2 = reinterpret_cast<const float2*>(in + (f + 1) * LDC + c);
                        const float2 v0 = s2[0], v1 = s2[1];
                        constexpr float un = 1.0f / kSiluScale;
                        if constexpr (PIPE) {
                            float* dst = ringb + (size_t)((t + a.tk_base) % RS) * (F1 * C1) + 4 * r;
                            st_state(dst, v0.x * un); st_state(dst + 1, v0.y * un); st_state(dst + 2, v1.x * un); st_state(dst + 3, v1.y * un);
                        } else
                        tkg[(KT - 2) * Q4 + r] = make_float4(v0.x * un, v0.y * un, v1.x * un, v1.y * un);
                    }
                    if constexpr (PIPE) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this thread's slot stores have left the CU
                }
                if constexpr (sub + 1 < KT) __syncthreads();
                if constexpr (PIPE && sub == 0) {
                    if (tid == 0) __hip_atomic_store(pflag + S::KB + lidx, (unsigned int)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            });
            conv_store<S, S::NTC, C1, LDC, true>(acc, out, 1, wave, lane);
            }
        };

        FE_CLK(4);
        // =========================== encoder (a6): k=3 convs ===========================
        static_for<S::NL>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            const float* in = encbuf(l);
            float* out = encbuf(l + 1);
            if (l == 0) FE_CLK(40);
            if constexpr (S::KT > 1) {
                k3_time(std::integral_constant<int, S::U_ENC + l * S::KT>{}, in, out, l, &o.enc_w[l * S::KT], o.enc_b[l]);
            } else {
            FE_BEGIN_UNIT(S::U_ENC + l);
            if constexpr (NSPLIT) {
                const float* a0 = in + li * LDC + lg;
                conv_nsplit<S, NS, 3 * S::KS_C, C1, LDC, true>(
                    [&](int i, int ks) { return a0[(16 * i + ks / S::KS_C) * LDC + 4 * (ks % S::KS_C)]; }, wb, o.enc_w[l], o.enc_b[l], stage,
                    out, 1, wave, lane, SG ? skipg + (l + 1) * SKIP_FLOATS : nullptr);
                if (l == 0) FE_CLK(41);
            } else {
            f32x4 acc[S::MTPW][S::NTC];
            acc_init_bias<S::MTPW, S::NTC>(acc, wb, o.enc_b[l], 0, 1, S::NTC);
            const float* const taps[3] = {in + (16 * wave + li + 0) * LDC + lg, in + (16 * wave + li + 1) * LDC + lg,
                                          in + (16 * wave + li + 2) * LDC + lg};
            conv_multi<S, S::NTC, 3, S::KS_C, LDC>(acc, taps, wb, o.enc_w[l], stage);
            __builtin_amdgcn_sched_barrier(0);
            if (l == 0) FE_CLK(41);
            stage.commit();
            conv_store<S, S::NTC, C1, LDC, !S::LN>(acc, out, 1, wave, lane, SG ? skipg + (l + 1) * SKIP_FLOATS : nullptr);
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (l == 0) FE_CLK(42);
            FE_KVW(0, W_FE + l);
            __syncthreads();
            if constexpr (S::LN) { FE_LN_SITE(F1, C1, LDC, true, out + LDC, 1 + l); __syncthreads(); }
            if (l == 0) FE_CLK(43);
            dbg_dump<S>(a, b, 3 + l, out + LDC, LDC);
        });

        float* Xb = smem + L::X;
        float* Hl = smem + L::HL;
        float* Hs = smem + L::HS;
        float* Gi = smem + L::GI;
        float* Y1 = smem + L::Y1;
        float* Y2 = smem + L::Y2;

        FE_CLK(5);
        // =========================== rf_pre (a7) ===========================
        constexpr int NTPW2 = ceil_div(S::NT2, kWaves);
        constexpr int HPT = ceil_div(F2 * C2, kThreads);   // hidden-state elements per thread
        f32x4 xr[S::MT2][NTPW2];                           // residual stream x, this wave's output tiles
        // weights of the RNNFormer-block GEMMs (this wave's output columns): register-resident and prefetched
        // one phase ahead when they fit (REGW), else streamed from L2 inside the GEMM pipeline
        constexpr int NTPW3 = ceil_div(S::NT3, kWaves);
        constexpr bool GFLAT = S::GFLAT;
        constexpr bool REGW = S::REGW;
        using WS = WSrc<Lds<S>::STAGED>;
        // GRU gates (r,z,n), input and hidden matrices: three column blocks per channel tile, or (GFLAT) one flat matrix
        constexpr int GNT = GFLAT ? S::NT3 : S::NT2;
        std::conditional_t<GFLAT, TokW<NTPW3, S::KS_2, 1, REGW, WS>, TokW<NTPW2, S::KS_2, 3, REGW, WS>> Wgi, Wgh;
        float hkeep[GFLAT ? HPT : 1];                   // GFLAT: previous hidden state of this thread's gate elements
        TokW<NTPW2, S::KS_2, 1, REGW, WS> Wf1, Wf2;     // rnn_fc, attn_fc
        TokW<NTPW3, S::KS_2, 1, REGW, WS> Wq;           // qkv
        TokW<NTPW3, S::KS_2, 1, REGW, WS> Wtq;          // dptransformer: the time attention's qkv (prefetched like the GRU weights it replaces)
        float pe_r[S::MT2][NTPW2][4];                                                  // positional embedding (block 0)
        // Unpredicated epilogue stores into the [F2P][C2 + 2] token buffers: pad rows are real rows, and lanes whose
        // column lies beyond C2 (the last channel tile; the wave without a tile) aim at the pad column - one select
        // per lane instead of an exec-masked block with its own address arithmetic per store group.
        auto tok_dst = [&](float* base, int j) {
            const int col = 16 * (wave + 4 * j) + li;
            return base + (4 * lg) * LDX + (col < C2 ? col : C2);
        };
        // r5, FBAL: shapes with more than four column tiles and streamed block weights (M: 5 tiles, L: 6) gave whole column tiles of rnn_fc / attn_fc
        // to the waves - 2:1:1:1 / 2:2:1:1.  Now a wave owns column tile `wave` (all row tiles: slot 0 of xr / pe_r) and the extra tiles' (row tile,
        // column tile) items q = wave + 4 x, x < NXI (slot 1, index x): L 6 + 6 + 6 + 6 tile-rows instead of 8 + 8 + 4 + 4, M 4 + 4 + 4 + 3 instead of
        // 6 + 3 + 3 + 3.  (The GRU phase of these shapes has been running on such jobs since r2: GBAL.)
        // (three items per wave in three different row tiles - 48 kHz L, six row tiles - measured -0.3 %: no A fragment is shared any more; left as it was)
        constexpr bool FBAL = FE_FBAL && !REGW && S::NT2 > 4 && S::NT2 <= 8 && !S::LN &&
                              ((kWaves % S::MT2) == 0 || ceil_div((S::NT2 - 4) * S::MT2, kWaves) <= 2);
        constexpr int NE = FBAL ? S::NT2 - 4 : 1, NXQ = NE * S::MT2, NXI = FBAL ? ceil_div(NXQ, kWaves) : 1;
        static_assert(!FBAL || (NXI <= S::MT2 && NTPW2 == 2), "FBAL: the extra items live in slot 1 of the residual registers");
        constexpr bool XSAME = (kWaves % S::MT2) == 0;      // every extra item of a wave lies in the same row tile: one A fragment serves them
        auto x_ok = [&](int x) { return wave + kWaves * x < NXQ; };
        auto x_ct = [&](int x) { int q = wave + kWaves * x; q = q < NXQ ? q : NXQ - 1; return 4 + q / S::MT2; };
        auto x_rt = [&](int x) { int q = wave + kWaves * x; q = q < NXQ ? q : NXQ - 1; return q % S::MT2; };
        auto x_dst = [&](float* base, int x) {               // like tok_dst: rows 4 lg + r of the item's row tile, its column (pad column past C2)
            const int col = 16 * x_ct(x) + li;
            return base + (16 * x_rt(x) + 4 * lg) * LDX + (col < C2 ? col : C2);
        };
        // acc = bias + A(LDS tokens) x W for the wave's FBAL jobs: accb = column tile `wave`, accx[x] = extra item x
        auto fc_gemm_bal = [&](f32x4 (&accb)[S::MT2][1], f32x4 (&accx)[1][NXI], const float* abase, int w_off, int b_off) {
            if constexpr (FBAL) {
                const int wk4 = w_off + wb.k4d;
                constexpr int KS = S::KS_2;
                f32x4 cache[NXI + 1];
                auto wget = [&](int slot, int t, int ks) -> float {      // TokW::get for an explicit column tile (streamed, k4 copy)
                    if (ks >= 4 * (KS / 4)) return wb.at_gv(wk4 + (t * KS + ks) * 64, wb.lane4);
                    if ((ks & 3) == 0) cache[slot] = wb.at_gv4(wk4 + t * (KS * 64) + (ks >> 2) * 256, wb.lane4 * 4);
                    return cache[slot][ks & 3];
                };
                {
                    const float bj = wb.at_gv(b_off + wave * 16, wb.li4);
#pragma unroll
                    for (int i = 0; i < S::MT2; ++i) accb[i][0] = f32x4{bj, bj, bj, bj};
                    const float* a_lane = abase + li * LDX + lg;
                    mma_panel<S::MT2, 1, KS, Lds<S>::PDK>(accb, [&](int i, int ks) { return a_lane[(16 * i) * LDX + 4 * ks]; },
                                                          [&](int, int ks) { return wget(0, wave, ks); }, NoSide{});
                }
#pragma unroll
                for (int x = 0; x < NXI; ++x) { const float bj = wb.at_gv(b_off + x_ct(x) * 16, wb.li4); accx[0][x] = f32x4{bj, bj, bj, bj}; }
                if constexpr (XSAME) {
                    const float* a_lane = abase + (16 * x_rt(0) + li) * LDX + lg;
                    mma_panel<1, NXI, KS, Lds<S>::PDK>(accx, [&](int, int ks) { return a_lane[4 * ks]; },
                                                       [&](int x, int ks) { return wget(1 + x, x_ct(x), ks); }, NoSide{});
                } else {
                    // the items lie in different row tiles: NXI independent (A row tile, B column tile) chains through one software pipeline
                    constexpr int PD = KS < Lds<S>::PDK ? KS : Lds<S>::PDK;
                    const float* a_lane[NXI];
#pragma unroll
                    for (int x = 0; x < NXI; ++x) a_lane[x] = abase + (16 * x_rt(x) + li) * LDX + lg;
                    float av[PD][NXI], bv[PD][NXI];
#pragma unroll
                    for (int ks = 0; ks < PD; ++ks)
#pragma unroll
                        for (int x = 0; x < NXI; ++x) { av[ks][x] = a_lane[x][4 * ks]; bv[ks][x] = wget(1 + x, x_ct(x), ks); }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        float a1[NXI], b1[NXI];
#pragma unroll
                        for (int x = 0; x < NXI; ++x) { a1[x] = av[ks % PD][x]; b1[x] = bv[ks % PD][x]; }
                        if (ks + PD < KS) {
#pragma unroll
                            for (int x = 0; x < NXI; ++x) { av[ks % PD][x] = a_lane[x][4 * (ks + PD)]; bv[ks % PD][x] = wget(1 + x, x_ct(x), ks + PD); }
                        }
#pragma unroll
                        for (int x = 0; x < NXI; ++x) accx[0][x] = FE_MFMA(a1[x], b1[x], accx[0][x]);
                    }
                }
            }
        };
        // x += acc (+ pe) for the FBAL jobs, into the residual registers and the token buffer
        auto fc_epi_bal = [&](const f32x4 (&accb)[S::MT2][1], const f32x4 (&accx)[1][NXI], bool add_pe) {
            if constexpr (FBAL) {
                float* xd = tok_dst(Xb, 0);
#pragma unroll
                for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = accb[i][0][r] + xr[i][0][r];
                        if (add_pe) v += pe_r[i][0][r];
                        xr[i][0][r] = v;
                        xd[(16 * i + r) * LDX] = v;
                    }
#pragma unroll
                for (int x = 0; x < NXI; ++x)
                    if (x_ok(x)) {
                        float* xe = x_dst(Xb, x);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = accx[0][x][r] + xr[x][1][r];
                            if (add_pe) v += pe_r[x][1][r];
                            xr[x][1][r] = v;
                            xe[r * LDX] = v;
                        }
                    }
            }
        };
        auto pe_load = [&]() {
#pragma unroll
            for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                for (int j = 0; j < NTPW2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int row = 16 * i + 4 * lg + r, col = 16 * (wave + 4 * j) + li;
                        if constexpr (FBAL) {
                            if (j == 1) { row = 16 * x_rt(i < NXI ? i : 0) + 4 * lg + r; col = 16 * x_ct(i < NXI ? i : 0) + li; }
                        }
                        pe_r[i][j][r] = wb.gather_g(o.blk_pe + (row < F2 ? row : F2 - 1) * C2 + (col < C2 ? col : C2 - 1));   // (pad rows / columns are never stored)
                    }
        };
        // LDS slot of hidden-state element tid + q * 256 (elements past F2 * C2 park in row 0's pad column)
        int hs_off[HPT];
#pragma unroll
        for (int q = 0; q < HPT; ++q) {
            const int i = tid + q * kThreads, f = i / C2;
            hs_off[q] = i < F2 * C2 ? f * LDX + (i - f * C2) : C2;
        }
        {
            // Y1[f2][c1] = sum_f1 Wf[f2][f1] * E[f1][c1]      (A = packed filterbank, B = LDS)
            constexpr int NTPW = ceil_div(S::NTC, kWaves);
            constexpr int KS = F1 / 4;
            const float* Ein = encbuf(S::NL) + LDC;   // row 0 = bin 0
            FE_BEGIN_UNIT(S::U_RFPRE);
            Wgi.bind(wb, o.blk_wih[0], o.blk_bih[0], GNT, wave, !S::TATT);      // block 0's GRU input weights ride in this GEMM
            f32x4 acc[S::MT2][NTPW];
            acc_init_zero<S::MT2, NTPW>(acc);
            mma_panel<S::MT2, NTPW, KS, Lds<S>::PDK>(
                acc,
                [&](int i, int ks) { return wb.at(o.rfpre_lin + (i * KS + ks) * 64); },
                [&](int j, int ks) {
                    int nt = wave + 4 * j;
                    nt = nt < S::NTC ? nt : S::NTC - 1;
                    return Ein[(4 * ks + lg) * LDC + 16 * nt + li];
                }, side2(stage, FetchSide<decltype(Wgi)>{&Wgi}));
            stage.commit();
#pragma unroll
            for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                for (int j = 0; j < NTPW; ++j) {
                    const int nt = wave + 4 * j;
                    const int col = 16 * nt + li;
                    if (nt < S::NTC && col < C1 && 16 * i + 4 * lg < F2) {   // F2 % 4 == 0: the 4 rows of a lane are valid together
#pragma unroll
                        for (int r = 0; r < 4; ++r) Y1[(16 * i + 4 * lg + r) * LDC + col] = acc[i][j][r];
                    }
                }
        }
        kvw_issue_rest(0, std::integral_constant<int, W_FE + S::NL>{});
        __syncthreads();
        {
            // X[f2][c2] = Y1[f2][:] . Wc[c2][:] + b
            constexpr int NTPW = ceil_div(S::NT2, kWaves);
            FE_BEGIN_UNIT(S::U_RFPRE + 1);
            Wgh.bind(wb, o.blk_whh[0], o.blk_bhh[0], GNT, wave, !S::TATT);      // ... and the hidden weights in this one
            // hidden state of block 0: fetched now, parked in LDS after the GEMM
            float hpre[HPT];
            if constexpr (!PIPE && !S::TATT) {     // (PIPE: the state is fetched as late as possible, inside the GRU phase)
                const float* hg0 = a.h + (size_t)b * (F2 * C2);
#pragma unroll
                for (int q = 0; q < HPT; ++q) { const int i = tid + q * kThreads; hpre[q] = hg0[i < F2 * C2 ? i : F2 * C2 - 1]; }   // (clamped, not predicated: no branch per load)
            }
            f32x4 acc[S::MT2][NTPW];
            acc_init_bias<S::MT2, NTPW>(acc, wb, o.rfpre_b, wave, 4, S::NT2);
            if constexpr (S::TATT) {
                Wtq.bind(wb, o.blk_tqkv[0], -1, S::NT3, wave);               // block 0's time-attention weights ride in this GEMM
                tok_gemm<S, NTPW, S::KS_C, LDC>(acc, Y1 + li * LDC + lg, wb, o.rfpre_w, S::NT2, wave, side2(stage, FetchSide<decltype(Wtq)>{&Wtq}));
            } else
            tok_gemm<S, NTPW, S::KS_C, LDC>(acc, Y1 + li * LDC + lg, wb, o.rfpre_w, S::NT2, wave, side2(stage, FetchSide<decltype(Wgh)>{&Wgh}));
            stage.commit();
#pragma unroll
            for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                for (int j = 0; j < NTPW; ++j) {
                    xr[i][j] = acc[i][j];
                    float* xd = tok_dst(Xb, j);
#pragma unroll
                    for (int r = 0; r < 4; ++r) xd[(16 * i + r) * LDX] = acc[i][j][r];
                }
            if constexpr (L::PERHEAD) __syncthreads();       // (there Y1, still being read by slower waves, lies over HS)
            if constexpr (!PIPE && !S::TATT) {
#pragma unroll
            for (int q = 0; q < HPT; ++q) {
                Hs[hs_off[q]] = hpre[q];
                if constexpr (GFLAT) hkeep[q] = hpre[q];
            }
            }
        }
        __syncthreads();
        // reload of the residual registers from the token buffer (ln variant: the norm passes update x in LDS)
        auto xr_reload = [&]() {
            if constexpr (FBAL) {
                const float* xd = tok_dst(Xb, 0);
#pragma unroll
                for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xr[i][0][r] = xd[(16 * i + r) * LDX];
#pragma unroll
                for (int x = 0; x < NXI; ++x) {
                    const float* xe = x_dst(Xb, x);
#pragma unroll
                    for (int r = 0; r < 4; ++r) xr[x][1][r] = xe[r * LDX];
                }
            } else {
#pragma unroll
            for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                for (int j = 0; j < NTPW2; ++j) {
                    const float* xd = tok_dst(Xb, j);
#pragma unroll
                    for (int r = 0; r < 4; ++r) xr[i][j][r] = xd[(16 * i + r) * LDX];
                }
            }
        };
        if constexpr (S::LN) {
            FE_LN_SITE(F2, C2, LDX, false, Xb, 1 + S::NL);
            __syncthreads();
            xr_reload();
        } else if constexpr (FBAL) xr_reload();         // (rf_pre's GEMM hands out whole column tiles: pick the residual up under the fc layers' ownership)
        dbg_dump<S>(a, b, 3 + S::NL, Xb, LDX);

        FE_CLK(6);
        // =========================== RNNFormer blocks (a9-a11) ===========================
        // small shapes: unrolled (block-0 special cases and every weight offset fold to constants; the per-hop
        // FastEnhancer_B kernel drops from 256 to 174 VGPRs); big shapes: rolled, for register pressure and code size
#pragma unroll (S::C1 <= 48 ? S::KB : 1)
        for (int k = 0; k < S::KB; ++k) {
            float* hg = a.h + ((size_t)k * a.B + b) * (F2 * C2);
            const int kb = k * o.blk_stride;
            if (k == 0) FE_CLK(20);
            if constexpr (S::TATT) {
                // dptransformer variant (models/fastenhancer/dptransformer/model.py:200-236, 378-389): causal attention over time per
                // sub-band and head.  Phase A: q | k | v of the frame = x W^T -> Gi (the layout of the sub-band attention's qkv).
                static_assert(!L::PERHEAD && S::LB == 31, "dptransformer: full qkv buffer, lookbehind 31");
                // (PIPE: a frame writes its K / V slot before any wait; re-use of that slot is ordered through the block-0 wait of the
                //  same frame one block later - which needs a second block)
                static_assert(!PIPE || S::KB >= 2, "dptransformer time pipeline: slot re-use is ordered through the next block's wait");
                // (Wtq was fetched inside the previous phase's GEMM: rf_pre's for block 0, the previous block's attn_fc after that;
                //  shapes that stream their block weights re-bind here)
                if constexpr (!REGW) Wtq.bind(wb, (o.blk_tqkv[0] + kb), -1, S::NT3, wave);
                Wf1.bind(wb, (o.blk_fc1_w[0] + kb), (o.blk_fc1_b[0] + kb), S::NT2, wave);   // fetched inside the GEMM below
                if constexpr (GFLAT) Wq.bind(wb, (o.blk_qkv[0] + kb), -1, S::NT3, wave);
                if (k == 0) pe_load();
                {
                    f32x4 acc[S::MT2][NTPW3];
                    if constexpr (GFLAT) tok_gemm_w<S, NTPW3, S::KS_2, LDX>(acc, Xb + li * LDX + lg, Wtq, FetchSide2<decltype(Wf1), decltype(Wq)>{&Wf1, &Wq});
                    else tok_gemm_w<S, NTPW3, S::KS_2, LDX>(acc, Xb + li * LDX + lg, Wtq, FetchSide<decltype(Wf1)>{&Wf1});
                    float* gdst = Gi + (4 * lg) * LDG + 16 * wave + li;
#pragma unroll
                    for (int j = 0; j < NTPW3; ++j)
                        if (wave + 4 * j < S::NT3) {
#pragma unroll
                            for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                                for (int r = 0; r < 4; ++r) gdst[(16 * i + r) * LDG + 64 * j] = acc[i][j][r];
                        }
                }
                __syncthreads();
                {
                    // Phase B: 16 lanes per (sub-band, head); lane l owns window positions l and l + 16 of the 32 (position 31 = this
                    // frame, read from Gi; the others = the cached frames, oldest first).  Scores / softmax / weighted values reduce over
                    // the 16-lane DPP row.  Each cache is a RING over its 31 slots: window position j lives in slot (head + j) mod 31,
                    // the frame's k / v overwrite slot `head` (the oldest), then head advances - one slot written per pair and frame
                    // instead of the reference's shift of all 31 (k[:, :, -L:]), which would double the HBM traffic of this HBM-bound
                    // step.  With head = 0 the ring IS the reference's cache tensor.  A slot whose first K element is +inf is masked
                    // (how a cache-less run marks frames before the start), as are, in offline mode, the slots older than the utterance.
                    if constexpr (PIPE) {
                    // time-pipelined offline launch: the K / V "caches" are per-frame rings of L + pipe_p slots per pair in the work buffer
                    // ([pair][RS][hd]; frame fr lives in slot fr mod RS).  The frame's own k / v go out FIRST and are counted in pflag[k] as
                    // soon as frame t - 1 has counted its own - the next frame waits for that, not for this frame's attention -, then the
                    // window t - 31 .. t - 1 is fetched
                    // (agent-scope accesses on both sides; slots older than the utterance are masked by frame index and read as zero).
                    constexpr int LBK = S::LB, PAIRS = F2 * S::NH;
                    const int RS = LBK + a.pipe_p;
                    const size_t cstride = (size_t)F2 * C2 * RS;
                    float* kc = a.h + ((size_t)(2 * k) * a.B + b) * cstride;
                    float* vc = a.h + ((size_t)(2 * k + 1) * a.B + b) * cstride;
                    const int wbase = a.tatt_base;
                    {
                        const int slot = (t + wbase) % RS;
                        for (int e = tid; e < PAIRS * HD; e += kThreads) {
                            const int p = e / HD, d = e - p * HD, f = p / S::NH, hh = p - f * S::NH;
                            const float* qk = Gi + f * LDG + hh * 3 * HD;
                            st_state(kc + ((size_t)p * RS + slot) * HD + d, qk[HD + d]);
                            st_state(vc + ((size_t)p * RS + slot) * HD + d, qk[2 * HD + d]);
                        }
                    }
                    // (wait first, then count: the counter must never step back - frame t + 1 may get here before frame t)
                    pipe_wait(k, t);
                    pipe_publish(k, t);
                    const int grp = tid >> 4, l16 = tid & 15;
                    const int mask_lo = wbase ? 0 : (LBK - t > 0 ? LBK - t : 0);
                    const float sc = __builtin_amdgcn_rsqf((float)HD);
                    const int j1 = l16 + 16;                                        // position 31 = the current frame
                    const bool mf0 = l16 < mask_lo, mf1 = j1 < LBK && j1 < mask_lo;
                    int fr0 = t + wbase - LBK + l16, fr1 = t + wbase - LBK + (j1 < LBK ? j1 : 0);
                    fr0 = fr0 < 0 ? 0 : fr0; fr1 = fr1 < 0 ? 0 : fr1;
                    const int sl0 = fr0 % RS, sl1 = fr1 % RS;
                    constexpr int NIT = ceil_div(PAIRS, 16);
                    static_assert(S::NH == 4, "head of a pair = its index mod 4");
                    const float tpe0 = wb.gather_g(o.tpe + (grp & 3) * 32 + l16), tpe1 = wb.gather_g(o.tpe + (grp & 3) * 32 + j1);
#pragma unroll 1
                    for (int it = 0; it < NIT; ++it) {
                        int p = grp + 16 * it;
                        const bool live = p < PAIRS;
                        p = live ? p : PAIRS - 1;
                        const int f = p / S::NH, hh = p - f * S::NH;
                        const float* qk = Gi + f * LDG + hh * 3 * HD;
                        const float* kp = kc + (size_t)p * (RS * HD);
                        const float* vp = vc + (size_t)p * (RS * HD);
                        float k0[HD], k1[HD], v0[HD], v1[HD];
#pragma unroll
                        for (int d = 0; d < HD; ++d) k0[d] = ld_state(kp + sl0 * HD + d);
#pragma unroll
                        for (int d = 0; d < HD; ++d) k1[d] = ld_state(kp + sl1 * HD + d);
#pragma unroll
                        for (int d = 0; d < HD; ++d) v0[d] = ld_state(vp + sl0 * HD + d);
#pragma unroll
                        for (int d = 0; d < HD; ++d) v1[d] = ld_state(vp + sl1 * HD + d);
                        // (carried caches: a slot whose first K element is +inf marks a frame before a cache-less start)
                        const bool m0 = mf0 || (wbase != 0 && k0[0] == __builtin_inff());
                        const bool m1 = mf1 || (wbase != 0 && j1 < LBK && k1[0] == __builtin_inff());
                        if (j1 == LBK) {
#pragma unroll
                            for (int d = 0; d < HD; ++d) { k1[d] = qk[HD + d]; v1[d] = qk[2 * HD + d]; }
                        }
#pragma unroll
                        for (int d = 0; d < HD; ++d) {          // (slots before the utterance hold whatever the buffer held: keep it out of the sums)
                            v0[d] = m0 ? 0.0f : v0[d];
                            v1[d] = m1 ? 0.0f : v1[d];
                        }
                        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                        for (int d = 0; d < HD; ++d) { const float qd = qk[d]; s0 = fmaf(qd, k0[d], s0); s1 = fmaf(qd, k1[d], s1); }
                        const float ninf = -__builtin_inff();
                        s0 = m0 ? ninf : fmaf(sc, s0, tpe0);
                        s1 = m1 ? ninf : fmaf(sc, s1, tpe1);
                        const float mx = row16_allreduce(fmaxf(s0, s1), [](float x, float y) { return fmaxf(x, y); });
                        const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx);
                        const float inv = __builtin_amdgcn_rcpf(row16_allreduce(e0 + e1, [](float x, float y) { return x + y; }));
                        float od = 0.0f, od2 = 0.0f;
#pragma unroll
                        for (int d = 0; d < HD; ++d) {
                            const float sum = row16_allreduce(e0 * v0[d] + e1 * v1[d], [](float x, float y) { return x + y; });
                            if (d < 16) od = (l16 == d) ? sum : od; else od2 = (l16 == d - 16) ? sum : od2;
                        }
                        if (live) {
                            if (l16 < HD) Hl[f * LDX + hh * HD + l16] = od * inv;
                            if (HD > 16 && l16 + 16 < HD) Hl[f * LDX + hh * HD + l16 + 16] = od2 * inv;
                        }
                    }
                    } else if constexpr (WPF) {
                    // per-hop launch: the window is in registers (kvw, fetched piece by piece since the previous use - see the top of the
                    // kernel); the same arithmetic per pair as the branch below.  After round `it` the registers of piece `it` are free:
                    // the first W_NB pieces of the NEXT block's window are requested right here, the others at the end of the block's
                    // later phases.
                    constexpr int LBK = S::LB;
                    const size_t cstride = (size_t)F2 * C2 * LBK;
                    float* kc = a.h + ((size_t)(2 * k) * a.B + b) * cstride;
                    float* vc = a.h + ((size_t)(2 * k + 1) * a.B + b) * cstride;
                    const int grp = tid >> 4, l16 = tid & 15;
                    const int mask_lo = (a.mode == FE_MODE_OFFLINE) ? (LBK - t > 0 ? LBK - t : 0) : 0;
                    const float sc = __builtin_amdgcn_rsqf((float)HD);
                    const int head = (ring_head0 + t) % LBK;
                    const int j1 = l16 + 16;
                    static_assert(S::NH == 4, "head of a pair = its index mod 4");
                    const float tpe0 = wb.gather_g(o.tpe + (grp & 3) * 32 + l16), tpe1 = wb.gather_g(o.tpe + (grp & 3) * 32 + j1);
                    static_for<W_NIT>([&](auto it_) {
                        constexpr int it = decltype(it_)::value;
                        const int p = grp + 16 * it;
                        const int f = p / S::NH, hh = p - f * S::NH;
                        const float* qk = Gi + f * LDG + hh * 3 * HD;
                        float* kp = kc + (size_t)p * (LBK * HD);
                        float* vp = vc + (size_t)p * (LBK * HD);
                        float k0[HD], k1[HD], v0[HD], v1[HD];
#pragma unroll
                        for (int d = 0; d < HD; ++d) { k0[d] = kvw[it][d]; k1[d] = kvw[it][HD + d]; v0[d] = kvw[it][2 * HD + d]; v1[d] = kvw[it][3 * HD + d]; }
                        // (opaque copies: with the conditional overwrite below the optimiser would otherwise select between the ADDRESSES -
                        //  LDS or the register array - and load once, which pins the whole array in scratch memory)
#pragma unroll
                        for (int d = 0; d < HD; ++d) { asm("" : "+v"(k1[d])); asm("" : "+v"(v1[d])); }
                        if (j1 == LBK) {
#pragma unroll
                            for (int d = 0; d < HD; ++d) { k1[d] = qk[HD + d]; v1[d] = qk[2 * HD + d]; }
                        }
                        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                        for (int d = 0; d < HD; ++d) { const float qd = qk[d]; s0 = fmaf(qd, k0[d], s0); s1 = fmaf(qd, k1[d], s1); }
                        const float ninf = -__builtin_inff();
                        s0 = (l16 < mask_lo || k0[0] == __builtin_inff()) ? ninf : fmaf(sc, s0, tpe0);
                        s1 = (j1 < mask_lo || (j1 < LBK && k1[0] == __builtin_inff())) ? ninf : fmaf(sc, s1, tpe1);
                        const float mx = row16_allreduce(fmaxf(s0, s1), [](float x, float y) { return fmaxf(x, y); });
                        const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx);
                        const float inv = __builtin_amdgcn_rcpf(row16_allreduce(e0 + e1, [](float x, float y) { return x + y; }));
                        float od = 0.0f;
                        float od2 = 0.0f;
#pragma unroll
                        for (int d = 0; d < HD; ++d) {
                            const float sum = row16_allreduce(e0 * v0[d] + e1 * v1[d], [](float x, float y) { return x + y; });
                            if (d < 16) od = (l16 == d) ? sum : od; else od2 = (l16 == d - 16) ? sum : od2;
                        }
                        if (l16 < HD) Hl[f * LDX + hh * HD + l16] = od * inv;
                        if (HD > 16 && l16 + 16 < HD) Hl[f * LDX + hh * HD + l16 + 16] = od2 * inv;
                        if (j1 == LBK) {                                  // the lane that holds the frame's k / v: over the oldest slot
#pragma unroll
                            for (int d = 0; d < HD; ++d) { kp[head * HD + d] = k1[d]; vp[head * HD + d] = v1[d]; }
                        }
                        if constexpr (it < W_NB) kvw_issue(k + 1, it_);
                    });
                    if (k == S::KB - 1 && tid == 0) a.h[(size_t)a.B * S::KB * S::HSTATE + b] = (float)((head + 1) % LBK);
                    } else {
                    constexpr int LBK = S::LB, PAIRS = F2 * S::NH;
                    const size_t cstride = (size_t)F2 * C2 * LBK;                    // one cache tensor of one stream: [F2][NH][L][HD]
                    float* kc = a.h + ((size_t)(2 * k) * a.B + b) * cstride;
                    float* vc = a.h + ((size_t)(2 * k + 1) * a.B + b) * cstride;
                    const int grp = tid >> 4, l16 = tid & 15;
                    const int mask_lo = (a.mode == FE_MODE_OFFLINE) ? (LBK - t > 0 ? LBK - t : 0) : 0;
                    const float sc = __builtin_amdgcn_rsqf((float)HD);               // (hd)^-0.5
                    const int head = (ring_head0 + t) % LBK;
                    const int j1 = l16 + 16;                                        // position 31 = the current frame
                    int sl0 = head + l16, sl1 = head + (j1 < LBK ? j1 : LBK - 1);
                    sl0 = sl0 >= LBK ? sl0 - LBK : sl0;
                    sl1 = sl1 >= LBK ? sl1 - LBK : sl1;
                    constexpr int NIT = ceil_div(PAIRS, 16);
                    // the positional bias of this lane's two window positions: the head of pair grp + 16 it is grp mod 4 in every round
                    static_assert(S::NH == 4, "head of a pair = its index mod 4");
                    const float tpe0 = wb.gather_g(o.tpe + (grp & 3) * 32 + l16), tpe1 = wb.gather_g(o.tpe + (grp & 3) * 32 + j1);
#pragma unroll 1
                    for (int it = 0; it < NIT; ++it) {
                        int p = grp + 16 * it;
                        const bool live = p < PAIRS;
                        p = live ? p : PAIRS - 1;                                   // (idle groups of the last round shadow the last pair, stores predicated)
                        const int f = p / S::NH, hh = p - f * S::NH;
                        const float* qk = Gi + f * LDG + hh * 3 * HD;
                        float* kp = kc + (size_t)p * (LBK * HD);
                        float* vp = vc + (size_t)p * (LBK * HD);
                        float k0[HD], k1[HD], v0[HD], v1[HD];
#pragma unroll
                        for (int d = 0; d < HD; ++d) k0[d] = kp[sl0 * HD + d];
#pragma unroll
                        for (int d = 0; d < HD; ++d) k1[d] = kp[sl1 * HD + d];
#pragma unroll
                        for (int d = 0; d < HD; ++d) v0[d] = vp[sl0 * HD + d];
#pragma unroll
                        for (int d = 0; d < HD; ++d) v1[d] = vp[sl1 * HD + d];
                        if (j1 == LBK) {
#pragma unroll
                            for (int d = 0; d < HD; ++d) { k1[d] = qk[HD + d]; v1[d] = qk[2 * HD + d]; }
                        }
                        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                        for (int d = 0; d < HD; ++d) { const float qd = qk[d]; s0 = fmaf(qd, k0[d], s0); s1 = fmaf(qd, k1[d], s1); }
                        const float ninf = -__builtin_inff();
                        s0 = (l16 < mask_lo || k0[0] == __builtin_inff()) ? ninf : fmaf(sc, s0, tpe0);
                        s1 = (j1 < mask_lo || (j1 < LBK && k1[0] == __builtin_inff())) ? ninf : fmaf(sc, s1, tpe1);
                        const float mx = row16_allreduce(fmaxf(s0, s1), [](float x, float y) { return fmaxf(x, y); });
                        const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx);
                        const float inv = __builtin_amdgcn_rcpf(row16_allreduce(e0 + e1, [](float x, float y) { return x + y; }));
                        float od = 0.0f;       // output element d = l16 (+ 16)
                        float od2 = 0.0f;
#pragma unroll
                        for (int d = 0; d < HD; ++d) {
                            const float sum = row16_allreduce(e0 * v0[d] + e1 * v1[d], [](float x, float y) { return x + y; });
                            if (d < 16) od = (l16 == d) ? sum : od; else od2 = (l16 == d - 16) ? sum : od2;
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every cache load of this pair has landed before the oldest slot is overwritten
                        if (live) {
                            if (l16 < HD) Hl[f * LDX + hh * HD + l16] = od * inv;
                            if (HD > 16 && l16 + 16 < HD) Hl[f * LDX + hh * HD + l16 + 16] = od2 * inv;
                            if (j1 == LBK) {                                  // the lane that holds the frame's k / v
#pragma unroll
                                for (int d = 0; d < HD; ++d) { kp[head * HD + d] = k1[d]; vp[head * HD + d] = v1[d]; }
                            }
                        }
                    }
                    if (k == S::KB - 1 && tid == 0) a.h[(size_t)a.B * S::KB * S::HSTATE + b] = (float)((head + 1) % LBK);
                    }
                }
            } else
            {
                // GRU (nn.GRU gate order r,z,n; model.py:187,271), gates fused into the GEMM epilogue:
                //   ax[.][g] = x W_i{g}^T + b_i{g},  ah[.][g] = h W_h{g}^T + b_h{g}   for this wave's channel tiles
                //   r = s(ax0+ah0), z = s(ax1+ah1), n = tanh(ax2 + r*ah2), h' = (1-z) n + z h
                // prefetch for the next phase: rnn_fc weights (+ the positional embedding in block 0)
                Wf1.bind(wb, (o.blk_fc1_w[0] + kb), (o.blk_fc1_b[0] + kb), S::NT2, wave);   // fetched inside the GEMM below
                // GFLAT: the qkv weights ride in this (long) GEMM too, attn_fc's in rnn_fc's - fetched one short phase
                // ahead they were still in flight when their GEMM started
                if constexpr (GFLAT) Wq.bind(wb, (o.blk_qkv[0] + kb), (S::FRNN ? o.blk_qkv_b[0] + kb : -1), S::NT3, wave);
                if (k == 0) pe_load();
                // Shapes with more than four column tiles and streamed weights (M: 5 tiles, L: 6): handing whole column
                // tiles to the waves leaves them 2:1:1:1 / 2:2:1:1 loaded.  Their GRU phase - it writes only to LDS and
                // the state, no register-resident residual - runs over (column tile, row-tile group) jobs instead,
                // round-robin: 15 jobs -> 4:4:4:3, 12 jobs -> 3:3:3:3.
                constexpr bool GBAL = !REGW && S::NT2 > 4;
                // PIPE: wait for frame t-1's state of this block, fetch it, park it in Hs (one barrier) - here, at the last
                // moment, so that the hand-off chain from frame to frame is wait -> fetch -> h-half GEMM -> gates -> publish
                auto pipe_fetch_state = [&]() {
                    if constexpr (PIPE) {
                        float hq_[HPT];
                        pipe_wait(k, t);
#pragma unroll
                        for (int q = 0; q < HPT; ++q) { const int i = tid + q * kThreads; hq_[q] = ld_state(hg + (i < F2 * C2 ? i : F2 * C2 - 1)); }
#pragma unroll
                        for (int q = 0; q < HPT; ++q) {
                            Hs[hs_off[q]] = hq_[q];
                            if constexpr (GFLAT) hkeep[q] = hq_[q];
                        }
                        __syncthreads();
                    }
                };
                if constexpr (!(GFLAT && PIPE)) pipe_fetch_state();
                if constexpr (GFLAT) {
                    // gates as one (3 C2)-column GEMM over this wave's flat column tiles (see Shape::GFLAT); the
                    // pre-activations cross through LDS: r | z | n_x -> Gi[row][0 .. 3 C2), n_h -> Hl[row][c]
                    constexpr int K2 = S::KS_2;
                    // column tiles 4 j .. 4 j + 3 (this j of the four waves) that hold only r / z columns sum x and h
                    // parts in ONE accumulator
                    auto pure_rz = [](int j) constexpr { return 16 * (4 * j + 4) <= 2 * C2; };
                    f32x4 ax[S::MT2][NTPW3], ah[S::MT2][NTPW3];
#pragma unroll
                    for (int j = 0; j < NTPW3; ++j) {
                        const float bi = Wgi.bias(j, 0), bh = Wgh.bias(j, 0);
                        const float b0 = pure_rz(j) ? bi + bh : bi;
#pragma unroll
                        for (int i = 0; i < S::MT2; ++i) { ax[i][j] = f32x4{b0, b0, b0, b0}; ah[i][j] = f32x4{bh, bh, bh, bh}; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (k == 0) FE_CLK(45);
                    if constexpr (PIPE) {
                        // the x half does not need the previous frame: it runs before the wait
                        mma_panel_sel<S::MT2, NTPW3, K2, Lds<S>::PDK>(
                            [&](int i, int j, int) -> f32x4& { return ax[i][j]; },
                            [&](int i, int ks) { return Xb[(16 * i + li) * LDX + lg + 4 * ks]; },
                            [&](int j, int ks) { return Wgi.get(j, 0, ks); }, FetchSide2<decltype(Wf1), decltype(Wq)>{&Wf1, &Wq});
                        pipe_fetch_state();
                        mma_panel_sel<S::MT2, NTPW3, K2, Lds<S>::PDK>(
                            [&](int i, int j, int) -> f32x4& { return pure_rz(j) ? ax[i][j] : ah[i][j]; },
                            [&](int i, int ks) { return Hs[(16 * i + li) * LDX + lg + 4 * ks]; },
                            [&](int j, int ks) { return Wgh.get(j, 0, ks); }, NoSide{});
                    } else
                    mma_panel_sel<S::MT2, NTPW3, 2 * K2, Lds<S>::PDK>(
                        [&](int i, int j, int ks) -> f32x4& { return ks < K2 || pure_rz(j) ? ax[i][j] : ah[i][j]; },
                        [&](int i, int ks) { return ks < K2 ? Xb[(16 * i + li) * LDX + lg + 4 * ks] : Hs[(16 * i + li) * LDX + lg + 4 * (ks - K2)]; },
                        [&](int j, int ks) { return ks < K2 ? Wgi.get(j, 0, ks) : Wgh.get(j, 0, ks - K2); },
                        FetchSide2<decltype(Wf1), decltype(Wq)>{&Wf1, &Wq});
                    __builtin_amdgcn_sched_barrier(0);
                    if (k == 0) FE_CLK(46);
                    // (pad rows / columns of Gi are written too - they are never read; Hl's rows are narrower)
                    {
                        float* gdst = Gi + (4 * lg) * LDG + 16 * wave + li;
#pragma unroll
                        for (int j = 0; j < NTPW3; ++j) {
                            const int g = 16 * (wave + 4 * j) + li;
                            const bool rz = g < 2 * C2, nh = !rz && g < 3 * C2;
                            // lanes without an n_h value store into Hl's pad column: no predicate on the stores
                            float* hdst = Hl + (4 * lg) * LDX + (nh ? g - 2 * C2 : C2);
                            if (pure_rz(j)) {
#pragma unroll
                                for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) gdst[(16 * i + r) * LDG + 64 * j] = ax[i][j][r];
                            } else if (wave + 4 * j < S::NT3) {
#pragma unroll
                                for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const float sx = ax[i][j][r], sh = ah[i][j][r];
                                        gdst[(16 * i + r) * LDG + 64 * j] = rz ? sx + sh : sx;
                                        hdst[(16 * i + r) * LDX] = sh;
                                    }
                            }
                        }
                    }
                    if (k == 0) FE_CLK(48);
                    __syncthreads();
                    if (k == 0) FE_CLK(49);
                    {
                        // element e = row * C2 + c per thread, HPT rounds; loads first, the gate chains interleave
                        float gr[HPT], gz[HPT], gn[HPT], gh[HPT];
                        int eo[HPT];
#pragma unroll
                        for (int q = 0; q < HPT; ++q) {
                            int e = tid + q * kThreads;
                            e = e < F2 * C2 ? e : F2 * C2 - 1;
                            const int row = e / C2, c = e - row * C2;
                            eo[q] = row * LDX + c;
                            gr[q] = Gi[row * LDG + c];
                            gz[q] = Gi[row * LDG + C2 + c];
                            gn[q] = Gi[row * LDG + 2 * C2 + c];
                            gh[q] = Hl[eo[q]];
                        }
                        float hn[HPT];
#pragma unroll
                        for (int q = 0; q < HPT; ++q) {
                            const float rr = sigmoid_f(gr[q]);
                            const float zz = sigmoid_f(gz[q]);
                            const float nn = tanh_f(gn[q] + rr * gh[q]);
                            hn[q] = (1.0f - zz) * nn + zz * hkeep[q];
                        }
#pragma unroll
                        for (int q = 0; q < HPT; ++q) asm volatile("" : "+v"(hn[q]));   // (keeps the chains out of the store predicate)
#pragma unroll
                        for (int q = 0; q < HPT; ++q) {
                            const int e = tid + q * kThreads;
                            if ((q + 1) * kThreads <= F2 * C2 || e < F2 * C2) { Hl[eo[q]] = hn[q]; st_state(hg + e, hn[q]); }
                        }
                    }
                } else if constexpr (GBAL) {
                    constexpr int MG = (S::MT2 % 2 == 0 && (S::NT2 * (S::MT2 / 2)) % kWaves == 0) ? 2 : 1;   // row tiles per job
                    constexpr int NMG = S::MT2 / MG, NJ = S::NT2 * NMG;
                    const int wih = o.blk_wih[0] + kb, whh = o.blk_whh[0] + kb, bih = o.blk_bih[0] + kb, bhh = o.blk_bhh[0] + kb;
#pragma unroll 1
                    for (int q = wave; q < NJ; q += kWaves) {
                        const int ct = q % S::NT2, m0 = (q / S::NT2) * MG;
                        f32x4 ax[MG][3], ah[MG][3];
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            const float bi = wb.at_gv(bih + (g * S::NT2 + ct) * 16, wb.li4), bh = wb.at_gv(bhh + (g * S::NT2 + ct) * 16, wb.li4);
#pragma unroll
                            for (int i = 0; i < MG; ++i) { ax[i][g] = f32x4{bi, bi, bi, bi}; ah[i][g] = f32x4{bh, bh, bh, bh}; }
                        }
                        // x half then h half in ONE software pipeline (k-steps KS_2 .. 2 KS_2 - 1 accumulate into ah)
                        constexpr int K2 = S::KS_2;
                        f32x4 gcur[3];
                        mma_panel_sel<MG, 3, 2 * K2, Lds<S>::PDK>(
                            [&](int i, int g, int ks) -> f32x4& { return ks < K2 ? ax[i][g] : ah[i][g]; },
                            [&](int i, int ks) {
                                return ks < K2 ? Xb[(16 * (m0 + i) + li) * LDX + lg + 4 * ks] : Hs[(16 * (m0 + i) + li) * LDX + lg + 4 * (ks - K2)];
                            },
                            [&](int g, int ks) {
                                const int k = ks < K2 ? ks : ks - K2;
                                const int base = (ks < K2 ? wih : whh) + (g * S::NT2 + ct) * (K2 * 64);
                                if constexpr (FE_K4_STREAM) {       // four k-steps per 16-byte load (the k4 copy; its K2 % 4 left-over k-steps are plain)
                                    if (k >= 4 * (K2 / 4)) return wb.at_g(base + wb.k4d + k * 64);
                                    if ((k & 3) == 0) gcur[g] = wb.at_gv4(base + wb.k4d + (k >> 2) * 256, wb.lane4 * 4);
                                    return gcur[g][k & 3];
                                } else return wb.at_g(base + k * 64);
                            }, NoSide{});
                        const int c = 16 * ct + li;
                        if (c < C2) {
#pragma unroll
                            for (int i = 0; i < MG; ++i)
                                if (16 * (m0 + i) + 4 * lg < F2) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const int row = 16 * (m0 + i) + 4 * lg + r;
                                        const float rr = sigmoid_f(ax[i][0][r] + ah[i][0][r]);
                                        const float zz = sigmoid_f(ax[i][1][r] + ah[i][1][r]);
                                        const float nn = tanh_f(ax[i][2][r] + rr * ah[i][2][r]);
                                        const float hp = Hs[row * LDX + c];
                                        const float hn = (1.0f - zz) * nn + zz * hp;
                                        Hl[row * LDX + c] = hn;
                                        st_state(hg + row * C2 + c, hn);
                                    }
                                }
                        }
                    }
                } else
#pragma unroll
                for (int j = 0; j < NTPW2; ++j) {
                    const int ct = wave + 4 * j;
                    f32x4 ax[S::MT2][3], ah[S::MT2][3];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const float bi = Wgi.bias(j, g), bh = Wgh.bias(j, g);
#pragma unroll
                        for (int i = 0; i < S::MT2; ++i) { ax[i][g] = f32x4{bi, bi, bi, bi}; ah[i][g] = f32x4{bh, bh, bh, bh}; }
                    }
                    // previous hidden state of this lane's outputs: read now, in flight under the GEMM (read in the epilogue,
                    // each LDS round trip would be exposed: 8 outputs x ~120 cycles)
                    float hprev[S::MT2][4];
                    {
                        const int cc = 16 * ct + li < C2 ? 16 * ct + li : C2 - 1;
#pragma unroll
                        for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) hprev[i][r] = Hs[(16 * i + 4 * lg + r) * LDX + cc];
                    }
                    // x half then h half in ONE software pipeline (k-steps KS_2 .. 2 KS_2 - 1 accumulate into ah)
                    constexpr int K2 = S::KS_2;
                    __builtin_amdgcn_sched_barrier(0);
                    if (k == 0) FE_CLK(45);
                    mma_panel_sel<S::MT2, 3, 2 * K2, Lds<S>::PDK>(
                        [&](int i, int g, int ks) -> f32x4& { return ks < K2 ? ax[i][g] : ah[i][g]; },
                        [&](int i, int ks) { return ks < K2 ? Xb[(16 * i + li) * LDX + lg + 4 * ks] : Hs[(16 * i + li) * LDX + lg + 4 * (ks - K2)]; },
                        [&](int g, int ks) { return ks < K2 ? Wgi.get(j, g, ks) : Wgh.get(j, g, ks - K2); }, FetchSide<decltype(Wf1)>{&Wf1});
                    __builtin_amdgcn_sched_barrier(0);
                    if (k == 0) FE_CLK(46);
                    const int c = 16 * ct + li;
                    if (ct < S::NT2 && c < C2) {
#pragma unroll
                        for (int i = 0; i < S::MT2; ++i)
                            if (16 * i + 4 * lg < F2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * i + 4 * lg + r;
                                {
                                    const float rr = sigmoid_f(ax[i][0][r] + ah[i][0][r]);
                                    const float zz = sigmoid_f(ax[i][1][r] + ah[i][1][r]);
                                    const float nn = tanh_f(ax[i][2][r] + rr * ah[i][2][r]);
                                    const float hn = (1.0f - zz) * nn + zz * hprev[i][r];
                                    Hl[row * LDX + c] = hn;
                                    st_state(hg + row * C2 + c, hn);
                                }
                            }
                            }
                    }
                }
            }
            if (k == 0) FE_CLK(47);
            if constexpr (PIPE && !S::TATT) pipe_publish(k, t);      // (its barrier is this phase's barrier; dptransformer: published above)
            else __syncthreads();
            if (k == 0) FE_CLK(21);
            if (k == 0) FE_CLK(22);
            {
                // x += rnn_fc(h') (+ pe in block 0)
                constexpr int NTPW = NTPW2;
                f32x4 acc[S::MT2][NTPW];
                if constexpr (FBAL) {
                    Wq.bind(wb, (o.blk_qkv[0] + kb), (S::FRNN ? o.blk_qkv_b[0] + kb : -1), S::NT3, wave);
                    f32x4 accb[S::MT2][1], accx[1][NXI];
                    fc_gemm_bal(accb, accx, Hl, (o.blk_fc1_w[0] + kb), (o.blk_fc1_b[0] + kb));
                    fc_epi_bal(accb, accx, k == 0);
                } else
                if constexpr (GFLAT) {
                    Wf2.bind(wb, (o.blk_fc2_w[0] + kb), (o.blk_fc2_b[0] + kb), S::NT2, wave);
                    tok_gemm_w<S, NTPW, S::KS_2, LDX>(acc, Hl + li * LDX + lg, Wf1, FetchSide<decltype(Wf2)>{&Wf2});
                } else {
                    Wq.bind(wb, (o.blk_qkv[0] + kb), (S::FRNN ? o.blk_qkv_b[0] + kb : -1), S::NT3, wave);      // for the next phase, fetched inside the GEMM
                    tok_gemm_w<S, NTPW, S::KS_2, LDX>(acc, Hl + li * LDX + lg, Wf1, FetchSide<decltype(Wq)>{&Wq});
                }
                if constexpr (S::LN) {
                    // raw fc output -> the (free) Gi buffer; the LayerNorm pass adds its result (and the positional embedding) to x
                    float* td = Gi + (4 * lg) * LDG + 16 * wave + li;
#pragma unroll
                    for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                        for (int j = 0; j < NTPW; ++j)
                            if (wave + 4 * j < S::NT2) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) td[(16 * i + r) * LDG + 64 * j] = acc[i][j][r];
                            }
                } else if constexpr (!FBAL) {
#pragma unroll
                for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                    for (int j = 0; j < NTPW; ++j) {
                        float* xd = tok_dst(Xb, j);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[i][j][r] + xr[i][j][r];
                            if (k == 0) v += pe_r[i][j][r];
                            xr[i][j][r] = v;
                            xd[(16 * i + r) * LDX] = v;
                        }
                    }
                }
            }
            FE_KVW(k + 1, W_NB);
            __syncthreads();
            if constexpr (S::LN) {
                ln_pass<F2, C2, LDG, false, true>(Gi, lnred, wp + lz + o.ln_g[2 + S::NL + 2 * k], wp + lz + o.ln_b[2 + S::NL + 2 * k], a.rf_eps, Xb, LDX,
                                                  k == 0 ? wp + lz + o.blk_pe : nullptr);
                __syncthreads();
                xr_reload();
            }
            dbg_dump<S>(a, b, 4 + S::NL + 2 * k, Xb, LDX);
            if (k == 0) FE_CLK(23);
            {
                // qkv = x W_qkv^T  -> Gi (rows per head interleaved [h][q|k|v][hd])
                constexpr int NTPW = NTPW3;
                // fetched inside the GEMM: attn_fc weights and the next block's GRU input weights
                if constexpr (!GFLAT) Wf2.bind(wb, (o.blk_fc2_w[0] + kb), (o.blk_fc2_b[0] + kb), S::NT2, wave);
                Wgi.bind(wb, (o.blk_wih[0] + kb + o.blk_stride), (o.blk_bih[0] + kb + o.blk_stride), GNT, wave, k + 1 < S::KB && !S::TATT);
                // r5, QBAL: eighteen column tiles (L) are 5 : 5 : 4 : 4 over the waves; the last two tiles' row tiles go half and half to the wave pairs
                // (0, 2) / (1, 3) instead: 4.5 tiles each.  (Gi is LDS only: no register-resident ownership to move.)
                constexpr bool QBAL = FE_QBAL && !REGW && !L::PERHEAD && (S::NT3 % 4) == 2 && (S::MT2 % 2) == 0 && NTPW3 >= 2;
                if constexpr (QBAL) {
                    constexpr int NQ = NTPW - 1, MH = S::MT2 / 2, KS = S::KS_2;
                    f32x4 acc[S::MT2][NQ];
                    tok_gemm_w<S, NQ, KS, LDX>(acc, Xb + li * LDX + lg, Wq, NoSide{});
                    {
                        float* gdst = Gi + (4 * lg) * LDG + 16 * wave + li;
#pragma unroll
                        for (int j = 0; j < NQ; ++j)
#pragma unroll
                            for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                                for (int r = 0; r < 4; ++r) gdst[(16 * i + r) * LDG + 64 * j] = acc[i][j][r];
                    }
                    const int xt = S::NT3 - 2 + (wave & 1), xr0 = MH * (wave >> 1);
                    const int wk4 = Wq.w_off + wb.k4d;
                    f32x4 cache;
                    f32x4 accx[MH][1];
                    {
                        const float bj = Wq.b_off >= 0 ? wb.at_gv(Wq.b_off + xt * 16, wb.li4) : 0.0f;
#pragma unroll
                        for (int i = 0; i < MH; ++i) accx[i][0] = f32x4{bj, bj, bj, bj};
                    }
                    const float* a_lane = Xb + (16 * xr0 + li) * LDX + lg;
                    mma_panel<MH, 1, KS, Lds<S>::PDK>(accx, [&](int i, int ks) { return a_lane[(16 * i) * LDX + 4 * ks]; },
                                                       [&](int, int ks) -> float {
                                                           if (ks >= 4 * (KS / 4)) return wb.at_gv(wk4 + (xt * KS + ks) * 64, wb.lane4);
                                                           if ((ks & 3) == 0) cache = wb.at_gv4(wk4 + xt * (KS * 64) + (ks >> 2) * 256, wb.lane4 * 4);
                                                           return cache[ks & 3];
                                                       }, NoSide{});
                    float* gx = Gi + (16 * xr0 + 4 * lg) * LDG + 16 * xt + li;
#pragma unroll
                    for (int i = 0; i < MH; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) gx[(16 * i + r) * LDG] = accx[i][0][r];
                } else
                if constexpr (!L::PERHEAD) {
                f32x4 acc[S::MT2][NTPW];
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0) FE_CLK(50);
                if constexpr (GFLAT) tok_gemm_w<S, NTPW, S::KS_2, LDX>(acc, Xb + li * LDX + lg, Wq, FetchSide<decltype(Wgi)>{&Wgi});
                else tok_gemm_w<S, NTPW, S::KS_2, LDX>(acc, Xb + li * LDX + lg, Wq, FetchSide2<decltype(Wf2), decltype(Wgi)>{&Wf2, &Wgi});
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0) FE_CLK(51);
                {
                    // (pad rows of Gi are written too - never read; one base address, immediate offsets, no predicates)
                    float* gdst = Gi + (4 * lg) * LDG + 16 * wave + li;
#pragma unroll
                    for (int j = 0; j < NTPW; ++j)
                        if (wave + 4 * j < S::NT3) {
#pragma unroll
                            for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                                for (int r = 0; r < 4; ++r) gdst[(16 * i + r) * LDG + 64 * j] = acc[i][j][r];
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0) FE_CLK(52);
                }
            }
            FE_KVW(k + 1, W_NB + 1);
            if constexpr (S::FRNN) {
                // dprnn variant (models/fastenhancer/dprnn/model.py:239-241): bidirectional GRU over the F2 sub-bands, zero initial
                // state.  Gi holds the input pre-activations [f][direction][r|z|n][unit] (+ b_ih, + b_hh for r, z); wave = direction,
                // lane = hidden unit with its 3 H hidden weights in registers; the step's h goes to Hl[f][direction * H + unit] -
                // the row the next step reads back as broadcasts (LDS operations of one wave execute in order) and frnn_fc's input.
                static_assert(!L::PERHEAD && L::HL % 2 == 0 && S::LDX % 2 == 0 && S::HF % 2 == 0 && S::HF <= 64, "dprnn: sub-band GRU layout");
                constexpr int H = S::HF;
                // 16 <= H <= 32: the two 32-lane halves of the wave split the sum over j (half 0: j < HA, half 1: the rest, zero-padded
                // to HA), one v_permlane32_swap joins the partial sums; both halves then do the same gate math and store the same h
                // (256 streams: dprnn B 49.2 -> 47.5 us, S 98.9 -> 89.9; T, H = 10, loses 1 % and keeps the one-unit-per-lane form)
                constexpr bool SPLIT = H <= 32 && H >= 16;
                constexpr int HA = SPLIT ? round_up(H / 2, 2) : H;
                const int half = SPLIT ? (lane >> 5) : 0;
                const int ul = SPLIT ? (lane & 31) : lane;
                const int u = ul < H ? ul : H - 1;               // lanes past H shadow unit H - 1 (same loads, same stores: no partial-exec region)
                const int dirw = wave & 1;
                const int j0 = half * HA;
                float fw[3][HA], fbn;
                {
                    const int base = o.blk_fhh[0] + kb + dirw * (H * 3 * H);
#pragma unroll
                    for (int j = 0; j < HA; ++j)
#pragma unroll
                        for (int g = 0; g < 3; ++g)           // (j0 + j >= H: an out-of-range offset reads 0)
                            fw[g][j] = wb.at_gv(base + (j * 3 + g) * H, (j0 + j < H ? (j0 * 3 * H + u) * 4 : 0x40000000));
                    fbn = wb.at_gv(o.blk_fbhn[0] + kb + dirw * H, u * 4);
                }
                __syncthreads();
                if (k == 0) FE_CLK(24);
                if (wave < 2) {
                    const float* gcol = Gi + dirw * (3 * H) + u;
                    float* hrow = Hl + dirw * H;
                    int f = dirw ? F2 - 1 : 0;
                    const int df = dirw ? -1 : 1;
                    float gr = gcol[f * LDG], gz = gcol[f * LDG + H], gn = gcol[f * LDG + 2 * H];
                    float hcur = 0.0f;
#pragma unroll 1
                    for (int st = 0; st < F2; ++st) {
                        const int fn = st + 1 < F2 ? f + df : f;
                        const float gr1 = gcol[fn * LDG], gz1 = gcol[fn * LDG + H], gn1 = gcol[fn * LDG + 2 * H];
                        float ar = half ? 0.0f : gr, az = half ? 0.0f : gz, an = half ? 0.0f : fbn;
                        if (st > 0) {
                            // (plain FMAs: the packed form, two j per v_pk_fma_f32, measured 4 % / 11 % slower on dprnn B / L)
                            const float* hp = hrow + (f - df) * LDX + j0;
#pragma unroll
                            for (int j = 0; j < HA; j += 2) {
                                // (the padded tail of half 1 re-reads the row's last pair: its weights are zero)
                                const float2 hv = *reinterpret_cast<const float2*>(j0 + j < H ? hp + j : hp + (H - 2 - j0));
                                ar = fmaf(fw[0][j], hv.x, ar); az = fmaf(fw[1][j], hv.x, az); an = fmaf(fw[2][j], hv.x, an);
                                ar = fmaf(fw[0][j + 1], hv.y, ar); az = fmaf(fw[1][j + 1], hv.y, az); an = fmaf(fw[2][j + 1], hv.y, an);
                            }
                        }
                        if constexpr (SPLIT) {
                            auto join = [](float x) { float p = x, q = x; asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q)); return p + q; };
                            ar = join(ar); az = join(az); an = join(an);
                        }
                        const float r = sigmoid_f(ar), z = sigmoid_f(az);
                        const float nn = tanh_f(gn + r * an);
                        hcur = (1.0f - z) * nn + z * hcur;
                        hrow[f * LDX + u] = hcur;
                        f += df; gr = gr1; gz = gz1; gn = gn1;
                    }
                }
            } else if constexpr (!L::PERHEAD) {
                __syncthreads();
                if (k == 0) FE_CLK(24);
                // attention: wave = head
                attention_head<S, S::MT2, LDG>(Gi, Hl, wave * 3 * HD, wave, 0, 1, lane);
            } else {
                // one head at a time, all four waves on it: its 3 hd qkv columns (column tiles t0 .. t1 of the packed
                // weight, (tile, row-tile group) jobs round-robin) -> Gi[F2P][3 hd + 2], then its attention with the
                // query tiles split over the waves
                constexpr int MGq = S::MT2 % 2 == 0 ? 2 : 1, NMGq = S::MT2 / MGq;
                constexpr int NQW = ceil_div(S::MT2, kWaves);
#pragma unroll 1
                for (int hh = 0; hh < S::NH; ++hh) {
                    const int c_lo = 3 * HD * hh, t0 = c_lo / 16, nth = (c_lo + 3 * HD - 1) / 16 - t0 + 1;
                    const int wq = o.blk_qkv[0] + kb;
#pragma unroll 1
                    for (int q = wave; q < nth * NMGq; q += kWaves) {
                        const int ct = t0 + q % nth, m0 = (q / nth) * MGq;
                        f32x4 acc[MGq][1];
                        acc_init_zero<MGq, 1>(acc);
                        mma_panel<MGq, 1, S::KS_2, Lds<S>::PDK>(
                            acc, [&](int i, int ks) { return Xb[(16 * (m0 + i) + li) * LDX + lg + 4 * ks]; },
                            [&](int, int ks) { return wb.at_g(wq + (ct * S::KS_2 + ks) * 64); }, NoSide{});
                        const int cl = 16 * ct + li - c_lo;
                        if (cl >= 0 && cl < 3 * HD) {
#pragma unroll
                            for (int i = 0; i < MGq; ++i)
                                if (16 * (m0 + i) + 4 * lg < F2) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) Gi[(16 * (m0 + i) + 4 * lg + r) * LDG + cl] = acc[i][0][r];
                                }
                        }
                    }
                    __syncthreads();
                    attention_head<S, NQW, LDG>(Gi, Hl, 0, hh, wave, kWaves, lane);
                    if (hh + 1 < S::NH) __syncthreads();           // (the next head overwrites Gi)
                }
            }
            FE_KVW(k + 1, W_NB + 2);
            __syncthreads();
            if (k == 0) FE_CLK(25);
            {
                // x += attn_fc(o)
                constexpr int NTPW = ceil_div(S::NT2, kWaves);
                float hpre[HPT];
                // next block: GRU hidden weights into registers inside the GEMM; hidden state fetched now / parked after it
                Wgh.bind(wb, (o.blk_whh[0] + kb + o.blk_stride), (o.blk_bhh[0] + kb + o.blk_stride), GNT, wave, k + 1 < S::KB && !S::TATT);
                if (!PIPE && !S::TATT && k + 1 < S::KB) {
                    const float* hgn = hg + (size_t)a.B * (F2 * C2);
#pragma unroll
                    for (int q = 0; q < HPT; ++q) { const int i = tid + q * kThreads; hpre[q] = hgn[i < F2 * C2 ? i : F2 * C2 - 1]; }
                }
                f32x4 acc[S::MT2][NTPW];
                f32x4 accb[S::MT2][1], accx[1][NXI];
                if constexpr (FBAL) {
                    if constexpr (S::TATT) Wtq.bind(wb, (o.blk_tqkv[0] + kb + o.blk_stride), -1, S::NT3, wave, k + 1 < S::KB);
                    fc_gemm_bal(accb, accx, Hl, (o.blk_fc2_w[0] + kb), (o.blk_fc2_b[0] + kb));
                } else
                if constexpr (S::TATT) {
                    Wtq.bind(wb, (o.blk_tqkv[0] + kb + o.blk_stride), -1, S::NT3, wave, k + 1 < S::KB);      // the next block's
                    tok_gemm_w<S, NTPW, S::KS_2, LDX>(acc, Hl + li * LDX + lg, Wf2, FetchSide<decltype(Wtq)>{&Wtq});
                } else
                tok_gemm_w<S, NTPW, S::KS_2, LDX>(acc, Hl + li * LDX + lg, Wf2, FetchSide<decltype(Wgh)>{&Wgh});
                if (!PIPE && !S::TATT && k + 1 < S::KB) {
#pragma unroll
                    for (int q = 0; q < HPT; ++q) {
                        Hs[hs_off[q]] = hpre[q];
                        if constexpr (GFLAT) hkeep[q] = hpre[q];
                    }
                }
                if constexpr (S::LN) {
                    float* td = Gi + (4 * lg) * LDG + 16 * wave + li;      // (qkv is dead: the attention has consumed it)
#pragma unroll
                    for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                        for (int j = 0; j < NTPW; ++j)
                            if (wave + 4 * j < S::NT2) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) td[(16 * i + r) * LDG + 64 * j] = acc[i][j][r];
                            }
                } else if constexpr (FBAL) {
                    fc_epi_bal(accb, accx, false);
                } else {
#pragma unroll
                for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                    for (int j = 0; j < NTPW; ++j) {
                        float* xd = tok_dst(Xb, j);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[i][j][r] + xr[i][j][r];
                            xr[i][j][r] = v;
                            xd[(16 * i + r) * LDX] = v;
                        }
                    }
                }
            }
            kvw_issue_rest(k + 1, std::integral_constant<int, W_NB + 3>{});
            __syncthreads();
            if constexpr (S::LN) {
                ln_pass<F2, C2, LDG, false, true>(Gi, lnred, wp + lz + o.ln_g[3 + S::NL + 2 * k], wp + lz + o.ln_b[3 + S::NL + 2 * k], a.rf_eps, Xb, LDX);
                __syncthreads();
                xr_reload();
            }
            if (k == 0) FE_CLK(26);
            dbg_dump<S>(a, b, 5 + S::NL + 2 * k, Xb, LDX);
        }

        FE_CLK(7);
        // =========================== rf_post (a13) ===========================
        {
            // Y2[f1][c2] = sum_f2 Wp[f1][f2] X[f2][c2]      (A packed, B = LDS tokens)
            constexpr int KS = F2 / 4;
            FE_BEGIN_UNIT(S::U_RFPOST);
            f32x4 acc[S::MTPW][S::NT2];
            acc_init_zero<S::MTPW, S::NT2>(acc);
            mma_panel<S::MTPW, S::NT2, KS, Lds<S>::PDK>(
                acc,
                [&](int i, int ks) { return wb.at(o.rfpost_lin + ((wave + 4 * i) * KS + ks) * 64); },
                [&](int j, int ks) { return Xb[(4 * ks + lg) * LDX + 16 * j + li]; }, stage);
            stage.commit();
            conv_store<S, S::NT2, C2, LDX, false>(acc, Y2, 0, wave, lane);
        }
        __syncthreads();
        // rf_post's 1x1 conv (C2 -> C1, no activation) is folded into decoder layer 0's 1x1 on the host: that layer
        // reads the filterbank output Y2 as its first K-segment (one GEMM phase and 3 C2 / 4 k-steps less per frame).
        // Buffer roles from here on: Wy (= W0) takes the 1x1 outputs - the k=3 convs read it with its zero halo rows -
        // and Wx (= W1, whose first rows Y2 occupies until layer 0's 1x1 has consumed it) the k=3 outputs.
        float* const Wx = W1;
        float* const Wy = W0;
        for (int i = tid; i < 2 * LDC; i += kThreads) {          // Wy lay under the token arena: zero its halo rows 0 and F1+1
            int r = i / LDC, c = i - r * LDC;
            Wy[(r ? F1 + 1 : 0) * LDC + c] = 0.0f;
        }
        if constexpr (S::LN) {
            // ln variant: rf_post's 1x1 conv (C2 -> C1) is a phase of its own again - a GroupNorm sits between it and decoder
            // layer 0: Z = GN(conv(Y2) + b) -> Wy (rows 1 .. F1), which layer 0's 1x1 then reads as its x segment (in place:
            // a wave reads and writes its own rows only)
            __syncthreads();                                      // (Wy's halo rows above)
            {
                FE_BEGIN_UNIT(S::U_RFPOST + 1);
                f32x4 acc[S::MTPW][S::NTC];
                acc_init_bias<S::MTPW, S::NTC>(acc, wb, o.rfpost1_b, 0, 1, S::NTC);
                const float* xa = Y2 + (16 * wave + li) * LDX + lg;
                mma_panel<S::MTPW, S::NTC, S::KS_2, Lds<S>::PDK>(
                    acc, [&](int i, int ks) { return xa[(64 * i) * LDX + 4 * ks]; },
                    [&](int j, int ks) { return wb.at(o.rfpost1_w + (j * S::KS_2 + ks) * 64); }, stage);
                stage.commit();
                conv_store<S, S::NTC, C1, LDC, false>(acc, Wy, 1, wave, lane);
            }
            __syncthreads();
            FE_LN_SITE(F1, C1, LDC, false, Wy + LDC, 2 + S::NL + 2 * S::KB);
            __syncthreads();
            dbg_dump<S>(a, b, 4 + S::NL + 2 * S::KB, Wy + LDC, LDC);
        } else
        if (a.dbg != nullptr) {                                  // the rf_post stage no longer exists: recompute it for the dump
            float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(4 + S::NL + 2 * S::KB);
            for (int i = tid; i < F1 * C1; i += kThreads) {
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
            const float* skip = Ebuf + (S::NL - l) * S::ACT;   // (LDS-resident skips)
            {
                // 1x1 conv on cat([x, skip]): two K-segments, never materialised.  Layer 0: x = rf_post's filterbank
                // output Y2 [F1][C2] with rf_post's 1x1 folded into this layer's weights; later layers: x = Wx [F1][C1].
                FE_BEGIN_UNIT(S::U_DEC + l * (S::KT + 1));
                constexpr bool FOLD0 = (l == 0) && !S::LN;             // (ln variant: layer 0 is a layer like the others, x = Z in Wy)
                constexpr int K0 = FOLD0 ? S::KS_2 : S::KS_C;          // k-steps of the x segment
                constexpr int LD0 = FOLD0 ? LDX : LDC;
                if constexpr (NSPLIT) {
                    const float* xa0 = (l == 0) ? Y2 + li * LDX + lg : Wx + (li + 1) * LDC + lg;
                    const float* sk0 = skip + (li + 1) * LDC + lg;
                    conv_nsplit<S, NS, K0 + S::KS_C, C1, LDC, true>(
                        [&](int i, int ks) {
                            if (ks < K0) return xa0[(16 * i) * LD0 + 4 * ks];
                            if constexpr (SG) return skb.at_g((S::NL - l) * SKIP_FLOATS + (i * S::KS_C + (ks - K0)) * 64);
                            else return sk0[(16 * i) * LDC + 4 * (ks - K0)];
                        }, wb, o.dec1_w[l], o.dec1_b[l], stage, Wy, 1, wave, lane, nullptr);
                } else {
                    f32x4 acc[S::MTPW][S::NTC];
                    acc_init_bias<S::MTPW, S::NTC>(acc, wb, o.dec1_b[l], 0, 1, S::NTC);
                    const float* xa = FOLD0 ? Y2 + (16 * wave + li) * LDX + lg : ((l == 0 ? Wy : Wx) + (16 * wave + li + 1) * LDC + lg);
                    const float* sk = skip + (16 * wave + li + 1) * LDC + lg;
                    // (SG: the skip comes back from the global scratch as A fragments, 8 k-steps ahead)
                    mma_panel<S::MTPW, S::NTC, K0 + S::KS_C, SG ? 8 : Lds<S>::PDK>(
                        acc,
                        [&](int i, int ks) {
                            if (ks < K0) return xa[(64 * i) * LD0 + 4 * ks];
                            if constexpr (SG) return skb.at_g((S::NL - l) * SKIP_FLOATS + ((wave + 4 * i) * S::KS_C + (ks - K0)) * 64);
                            else return sk[(64 * i) * LDC + 4 * (ks - K0)];
                        },
                        ConvB<S, S::NTC, K0 + S::KS_C, decltype(wb)>{wb, o.dec1_w[l]}, stage);
                    stage.commit();
                    conv_store<S, S::NTC, C1, LDC, !S::LN>(acc, Wy, 1, wave, lane);
                }
            }
            __syncthreads();
            if constexpr (S::LN) { FE_LN_SITE(F1, C1, LDC, true, Wy + LDC, 3 + S::NL + 2 * S::KB + 2 * l); __syncthreads(); }
            if constexpr (S::KT > 1) {
                k3_time(std::integral_constant<int, S::U_DEC + l * (S::KT + 1) + 1>{}, Wy, Wx, S::NL + l, &o.dec3_w[l * S::KT], o.dec3_b[l]);
            } else {
                FE_BEGIN_UNIT(S::U_DEC + l * (S::KT + 1) + 1);
                if constexpr (NSPLIT) {
                    const float* a0 = Wy + li * LDC + lg;
                    conv_nsplit<S, NS, 3 * S::KS_C, C1, LDC, true>(
                        [&](int i, int ks) { return a0[(16 * i + ks / S::KS_C) * LDC + 4 * (ks % S::KS_C)]; }, wb, o.dec3_w[l], o.dec3_b[l], stage,
                        Wx, 1, wave, lane, nullptr);
                } else {
                f32x4 acc[S::MTPW][S::NTC];
                acc_init_bias<S::MTPW, S::NTC>(acc, wb, o.dec3_b[l], 0, 1, S::NTC);
                const float* const taps[3] = {Wy + (16 * wave + li + 0) * LDC + lg, Wy + (16 * wave + li + 1) * LDC + lg,
                                              Wy + (16 * wave + li + 2) * LDC + lg};
                conv_multi<S, S::NTC, 3, S::KS_C, LDC>(acc, taps, wb, o.dec3_w[l], stage);
                stage.commit();
                conv_store<S, S::NTC, C1, LDC, !S::LN>(acc, Wx, 1, wave, lane);   // Wx (and Y2 under it) was fully consumed before the barrier above
                }
            }
            __syncthreads();
            if constexpr (S::LN) { FE_LN_SITE(F1, C1, LDC, true, Wx + LDC, 4 + S::NL + 2 * S::KB + 2 * l); __syncthreads(); }
            dbg_dump<S>(a, b, 5 + S::NL + 2 * S::KB + l, Wx + LDC, LDC);
        });

        FE_CLK(9);
        // =========================== dec_post (a15) ===========================
        float* PT = smem + L::PT;
        {
            FE_BEGIN_UNIT(S::U_POST);
            if constexpr (NSPLIT) {
                const float* xa0 = Wx + (li + 1) * LDC + lg;
                const float* sk0 = Ebuf + (li + 1) * LDC + lg;
                conv_nsplit<S, NS, 2 * S::KS_C, C1, LDC, true>(
                    [&](int i, int ks) {
                        if (ks < S::KS_C) return xa0[(16 * i) * LDC + 4 * ks];
                        if constexpr (SG) return skb.at_g((i * S::KS_C + (ks - S::KS_C)) * 64);
                        else return sk0[(16 * i) * LDC + 4 * (ks - S::KS_C)];
                    }, wb, o.post1_w, o.post1_b, stage, Wy, 1, wave, lane, nullptr);
            } else {
            f32x4 acc[S::MTPW][S::NTC];
            acc_init_bias<S::MTPW, S::NTC>(acc, wb, o.post1_b, 0, 1, S::NTC);
            if constexpr (!SG) {
                const float* const segs[2] = {Wx + (16 * wave + li + 1) * LDC + lg, Ebuf + (16 * wave + li + 1) * LDC + lg};
                conv_multi<S, S::NTC, 2, S::KS_C, LDC>(acc, segs, wb, o.post1_w, stage);
            } else {
                const float* xa = Wx + (16 * wave + li + 1) * LDC + lg;
                mma_panel<S::MTPW, S::NTC, 2 * S::KS_C, 8>(
                    acc,
                    [&](int i, int ks) {
                        return ks < S::KS_C ? xa[(64 * i) * LDC + 4 * ks] : skb.at_g(((wave + 4 * i) * S::KS_C + (ks - S::KS_C)) * 64);
                    },
                    ConvB<S, S::NTC, 2 * S::KS_C, decltype(wb)>{wb, o.post1_w}, stage);
            }
            stage.commit();
            conv_store<S, S::NTC, C1, LDC, !S::LN>(acc, Wy, 1, wave, lane);
            }
        }
        __syncthreads();
        if constexpr (S::LN) { FE_LN_SITE(F1, C1, LDC, true, Wy + LDC, 3 + 3 * S::NL + 2 * S::KB); __syncthreads(); }
        {
            // transposed conv as GEMM: P[i][co*8+j] = sum_ci x[i][ci] w[ci][co][j]
            FE_BEGIN_UNIT(S::U_POST + 1);
            f32x4 acc[S::MTPW][1];
            acc_init_zero<S::MTPW, 1>(acc);
            conv_seg<S, 1, S::KS_C, S::KS_C, LDC>(acc, Wy + (16 * wave + li + 1) * LDC + lg, wb, o.post_t_w, stage);
            stage.commit();
            conv_store<S, 1, 16, S::LDP, false>(acc, PT, 0, wave, lane);
        }
        __syncthreads();

        FE_CLK(10);
        typename Dft<S>::InvConst idc;
        if constexpr (MDFT) Dft<S>::load(idc, wb, o, wave);        // iSTFT constants, in flight during the mask phase
        // =========================== mask, un-compress (a16, a17), Hermitian spectrum ===========================
        {
            const float b0 = wb.scalar(o.post_t_b), b1 = wb.scalar(o.post_t_b + 1);
            float* spo = mode == FE_MODE_SPEC ? a.spec_out + (size_t)b * (F0 + 1) * a.T * 2 : nullptr;
            float* sph = mode == FE_MODE_OFFLINE ? a.spec_out + (size_t)b * F0 * a.T * 2 : nullptr;   // spec_hat [B,F0,T,2]
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
                if (sph != nullptr) {   // Model.forward returns the masked spectrum in the compressed domain
                    sph[((size_t)f * a.T + t) * 2] = yr;
                    sph[((size_t)f * a.T + t) * 2 + 1] = yi;
                }
                const float mag = sqrtf(yr * yr + yi * yi);
                const float g = pow_f(mag, 1.0f / a.compression - 1.0f);
                yr *= g; yi *= g;
                if (a.dbg) {
                    float* dst = a.dbg + (size_t)b * a.dbg_stride + DebugLayout<S>::offset(6 + 2 * S::NL + 2 * S::KB);
                    dst[2 * f] = yr; dst[2 * f + 1] = yi;
                    if (f == 0) { dst[2 * F0] = 0.0f; dst[2 * F0 + 1] = 0.0f; }
                }
                if (mode == FE_MODE_SPEC) {
                    spo[((size_t)f * a.T + t) * 2] = yr;
                    spo[((size_t)f * a.T + t) * 2 + 1] = yi;
                    if (f == 0) { spo[((size_t)F0 * a.T + t) * 2] = 0.0f; spo[((size_t)F0 * a.T + t) * 2 + 1] = 0.0f; }
                } else {
                    if constexpr (MDFT) {
                        q3[f] = yr;                  // (irfft ignores Im X[0]; the zero-padded Nyquist bin and the Hermitian
                        q3[N / 2 + f] = yi;          //  upper half are implied by Dft<S>::inverse)
                    } else if (f == 0) {
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

        FE_CLK(11);
        // =========================== iSTFT (a18) ===========================
        if (mode != FE_MODE_SPEC) {
            // streaming: synthesis window w / sum_k w^2 (steady state); offline: plain w, normalised below
            const float* wi = wp + (mode == FE_MODE_STREAM ? o.window_istft : o.window);
            constexpr int NPT = N / kThreads;
            float ow[NPT], oc[NPT];                        // window / overlap tail: fetched across the inverse DFT
#pragma unroll
            for (int q = 0; q < NPT; ++q) {
                const int n = tid + q * kThreads;
                ow[q] = wi[n];
                oc[q] = (!PIPE && n < OVL) ? cis[n] : 0.0f;
            }
            float* xo;
            if constexpr (MDFT) {
                Dft<S>::inverse(q3, q0, q1, tw, idc, wb, o, wave, lane);
                FE_CLK(12);
                xo = q2;
#pragma unroll
                for (int q = 0; q < NPT; ++q) {
                    const int n = tid + q * kThreads;
                    const int pi = Dft<S>::pidx(n & (Dft<S>::N1 - 1), n / Dft<S>::N1);
                    xo[n] = (q0[pi] + q1[pi]) * ow[q] + oc[q];
                }
            } else {
                const float2* y = fft_lds<S, true>(fa, fb, tw);
                FE_CLK(12);
                xo = reinterpret_cast<float*>((y == fa) ? fb : fa);
                const float invN = 1.0f / (float)N;
#pragma unroll
                for (int q = 0; q < NPT; ++q) {
                    const int n = tid + q * kThreads;
                    xo[n] = y[n].x * invN * ow[q] + oc[q];
                }
            }
            __syncthreads();
            if constexpr (PIPE) {
                // the frames overlap-add in a separate launch (istft_ola_kernel): no tail is carried from frame to frame
                float* fr = a.frames + ((size_t)b * a.T + t) * N;
                for (int n = tid; n < N; n += kThreads) fr[n] = xo[n];
            } else if (mode == FE_MODE_STREAM) {
                float* out = a.wav_out + (size_t)b * a.out_stride + (size_t)t * H;
                for (int n = tid; n < H; n += kThreads) out[n] = xo[n];
            } else {
                // torch.istft(center=True) (functional/audio_modules.py:117-119): y = OLA / sum_t w^2, trimmed by N/2.
                // After frame t the samples [tH, tH+H) of the overlap-add are final (all of [tH, tH+N) after the last frame).
                const float* w = wp + o.window;
                const int n_out = H * (a.T - 1);
                const int emit = (t == a.T - 1) ? N : H;
                float* out = a.wav_out + (size_t)b * a.out_stride;
                for (int j = tid; j < emit; j += kThreads) {
                    const int n = t * H + j;             // position in the un-trimmed overlap-add
                    const int pos = n - N / 2;
                    if (pos >= 0 && pos < n_out) {
                        int t_lo = (n - N + H) / H;      // ceil((n - N + 1) / H)
                        t_lo = t_lo < 0 ? 0 : t_lo;
                        int t_hi = n / H;
                        t_hi = t_hi > a.T - 1 ? a.T - 1 : t_hi;
                        float env = 0.0f;
                        for (int tt = t_lo; tt <= t_hi; ++tt) { const float wv = w[n - tt * H]; env += wv * wv; }
                        out[pos] = xo[j] / env;
                    }
                }
            }
            if constexpr (!PIPE) { for (int m = tid; m < OVL; m += kThreads) cis[m] = xo[m + H]; }
            __syncthreads();
        }
        FE_CLK(13);
    }
    b += PIPE ? a.B : (int)gridDim.x;
    } while (PERSIST && b < a.B);
    FE_CLK(63);
}

}  // namespace fe
