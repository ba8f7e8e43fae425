// lisennet_sb_kernels.hip.h — the MIDDLE of LiSenNet BATCHED OVER THE STREAMS on the fp32 matrix cores (gfx950), for the per-hop step of large
// batches (r6): encoder.conv_3 / conv_4 (DSConv, models/lisennet/model.py:173-207), the two DPR blocks (:61-159: LayerNorm + bidirectional
// GRU over the 32 sub-bands + dense, LayerNorm + one GRU step over time + dense, ConvolutionalGLU) and the decoder's three sub-pixel
// up-convolutions (USConv, :210-246, :295-298) - 0.86 of the model's 0.90 MMAC per frame.
//
// lisennet_frame_kernel gives a stream a workgroup: every product is then M = 1 - vector FMAs over k-major weights, 2.8 % (256 streams) to
// 5.1 % (4096) of the fp32 matrix peak.  Here a 512-thread workgroup takes SIXTEEN streams and every product is a matrix-core GEMM with the
// streams as the N dimension, computed TRANSPOSED (as in fspen_sb_kernels.hip.h / bsrnn_sb_kernels.hip.h):
//     out^T [rows x 16 streams] = W [rows x K] . in^T [K x 16 streams]
// A = weights, fetched as 16-byte fragments in "k4" order ([tile][quad][lane][4]: four consecutive k-steps of a lane per load),
// B = activations: lane (li, lg) holds one k index of stream li per k-step, C/D = lane (li, lg) holds rows 4 lg .. 4 lg + 3 of stream li.
//   * Convolutions (DSConv / USConv): one 16-row tile per output position (rows = output channels, or (sub-pixel phase, channel) for the
//     pixel-shuffled high halves), K = (source, tap, channel).  Their inputs live in the tile's CARRY in global memory (L2 / Infinity Cache)
//     as [position][channel group][16 streams][4 channels]: a lane's accumulator register quadruple - four consecutive output channels of one
//     position and stream - IS one 16-byte element of that layout (stored as it stands), and a 16-byte load of it is the B operand of four
//     consecutive k-steps (k-step j of quad m, lane group lg <-> pair 4 m + lg = (source, tap, channel group), channel 4 g + j).  The zero
//     padding AFTER the low / high split of the bins is a zero row on either side of each slice ("halo": written once per launch); tensors that
//     two layers slice differently (x2, x3: DSConv quarters, USConv halves) are stored once per slicing.
//   * The DPR blocks follow FSPEN's DPE recipe: wave w owns sub-bands 4 w .. 4 w + 3 for everything but the intra recurrence, the residual
//     stream stays in registers (register r of lane group lg <-> channel 4 lg + r), every product's output feeds the next one's B operand from
//     registers.  Intra GRU: wave (direction, q) owns hidden units 3 q .. 3 q + 2, its one tile is (unit, gate r / z / n_x / n_h) x K = 16 + 12,
//     the four gate values of a (stream, unit) meet in one lane, the new h goes to the h sequence in LDS, one barrier per step for both directions.
//     ConvolutionalGLU: the causal 3 x 3 depthwise conv runs on the fc1 accumulators (a lane owns eight channels x four positions); the two columns
//     next to a wave's positions come from the neighbour waves through LDS (current frame) and from the cache tensor (the two older frames).
// The step is three launches: lisennet_frame_kernel<PART 1> (STFT, features, conv_1, conv_2 per stream; x2 and its cached frame go to the carry
// in the layout above), this kernel, lisennet_frame_kernel<PART 2> (mask conv, LayerNorm, sigmoid, mask, iSTFT per stream).
// (included by lisennet_kernels.hip.h, after LShape / LPk)
#pragma once
#include <atomic>

namespace fe {

constexpr int kLsbStreams = 16;
constexpr int kLsbThreads = 512;

// a sliced activation tensor of a stream tile in the carry: [rows][G channel groups][16 streams][4], the low slice's NLO positions and the high
// slice's NHI positions each between two zero rows
template <int G_, int NLO_, int NHI_, int BASE_>
struct LSl {
    static constexpr int G = G_, NLO = NLO_, NHI = NHI_, BASE = BASE_;
    static constexpr int ROW = G * 64;
    static constexpr int ROWS = NLO + NHI + 4;
    static constexpr int LO = BASE;                        // the low slice's halo row (position -1)
    static constexpr int HI = BASE + (NLO + 2) * ROW;      // the high slice's
    static constexpr int END = BASE + ROWS * ROW;
    __host__ __device__ static constexpr int row(int P) { return P < NLO ? LO + (1 + P) * ROW : HI + (1 + P - NLO) * ROW; }
    __host__ __device__ static constexpr int halo(int i) { return i == 0 ? LO : i == 1 ? LO + (NLO + 1) * ROW : i == 2 ? HI : HI + (NHI + 1) * ROW; }
};

struct LCarry {
    using X1C = LSl<1, 64, 193, 0>;            // conv_1 output [4][257], this frame, in conv_2's slicing
    using X1P = LSl<1, 64, 193, X1C::END>;     // ... the cached frame
    using X2C = LSl<2, 32, 96, X1P::END>;      // conv_2 output [8][128], this frame, in conv_3's slicing
    using X2P = LSl<2, 32, 96, X2C::END>;      // ... the cached frame
    using X2S = LSl<2, 64, 64, X2P::END>;      // ... this frame in up3's slicing (skip)
    using X3C = LSl<3, 16, 48, X2S::END>;      // conv_3 output [12][64] in conv_4's slicing
    using X3P = LSl<3, 16, 48, X3C::END>;      // ... the cached frame
    using X3S = LSl<3, 32, 32, X3P::END>;      // ... in up2's slicing
    using X4S = LSl<4, 16, 16, X3S::END>;      // conv_4 output [16][32] in up1's slicing
    using XD = LSl<4, 16, 16, X4S::END>;       // output of the DPR blocks
    using U1 = LSl<3, 32, 32, XD::END>;        // up1 output [12][64] in up2's slicing
    using U2 = LSl<2, 64, 64, U1::END>;        // up2 output [8][128] in up3's slicing
    using U3S = LSl<1, 256, 0, U2::END>;       // up3 output [4][256] in mask_conv's "slicing" (one slice: zero rows at -1 and 256)
    using U3P = LSl<1, 256, 0, U3S::END>;      // ... the cached frame
    static constexpr int MK = U3P::END;        // the mask [257 f][16 n][2] (read by PART 2)
    static constexpr int TILE = MK + 258 * 32;
    static constexpr int SP = 516;             // per stream, behind the tiles: the compressed spectrum [257][2] (PART 1 -> PART 2)
    static constexpr int FEAT = 784;           // ... and behind those, per stream: the input features [3][260] (PART 1 -> the middle's conv_1)
    __host__ __device__ static constexpr size_t feat(int B) { return (size_t)((B + 15) / 16) * TILE + (size_t)((B + 15) / 16) * 16 * SP; }
    __host__ __device__ static constexpr size_t floats(int B) { return feat(B) + (size_t)((B + 15) / 16) * 16 * FEAT; }
};

// packed weights of the stream-batched kernel (floats, relative to LPk::TOTAL); A fragments in k4 order [tile][quad][lane][4]
struct LSbPk {
    static constexpr int C1_W = 0, C1_G = 16, C1_BE = C1_G + 260, C1_P = C1_BE + 260;             // conv_1: [c < 3][4 o] | bias [4], gamma / beta [257], PReLU [4]
    static constexpr int C2_LO = C1_P + 4, C2_HI = C2_LO + 2 * 256, C2_BL = C2_HI + 3 * 256, C2_BH = C2_BL + 16, C2_G = C2_BH + 16, C2_BE = C2_G + 128, C2_P = C2_BE + 128;
    static constexpr int C3_LO = C2_P + 16, C3_HI = C3_LO + 3 * 256, C3_BL = C3_HI + 5 * 256, C3_BH = C3_BL + 16, C3_G = C3_BH + 16, C3_BE = C3_G + 64, C3_P = C3_BE + 64;
    static constexpr int C4_LO = C3_P + 16, C4_HI = C4_LO + 5 * 256, C4_BL = C4_HI + 8 * 256, C4_BH = C4_BL + 16, C4_G = C4_BH + 16, C4_BE = C4_G + 32, C4_P = C4_BE + 32;
    static constexpr int U1_LO = C4_P + 16, U1_HI = U1_LO + 6 * 256, U1_BL = U1_HI + 3 * 6 * 256, U1_BH = U1_BL + 16;
    static constexpr int U2_LO = U1_BH + 48, U2_HI = U2_LO + 5 * 256, U2_BL = U2_HI + 2 * 5 * 256, U2_BH = U2_BL + 16;
    static constexpr int U3_LO = U2_BH + 32, U3_HI = U3_LO + 3 * 256, U3_BL = U3_HI + 3 * 256, U3_BH = U3_BL + 16;
    // mask_conv.0 (2, 4, 2, 2) as a DSConv-like tile | bias; LayerNorm affine [257]; PReLU [2]; mask_conv.3 [o][c] | bias; learnable sigmoid slope [257]
    static constexpr int M0_W = U3_BH + 16, M0_B = M0_W + 256, M_G = M0_B + 16, M_BE = M_G + 260, M_SL = M_BE + 260, M_P = M_SL + 260, M3 = M_P + 4;
    static constexpr int BLK = M3 + 8;
    // per DPR block
    static constexpr int N1W = 0, N1B = 512;                                  // intra_norm [f][d]
    static constexpr int I_W = 1024, I_B = I_W + 8 * 2 * 256;                 // intra GRU: [wave = 4 d + q][quad x | h][256], start values [wave][16]
    static constexpr int D1_W = I_B + 8 * 16, D1_B = D1_W + 2 * 256;          // intra dense: [quad = direction][256], bias [16]
    static constexpr int N2W = D1_B + 16, N2B = N2W + 512;                    // inter_norm [f][d]
    static constexpr int XR = N2B + 512, XZ = XR + 2 * 3 * 256, XNX = XZ + 2 * 3 * 256, XNH = XNX + 2 * 256;      // inter GRU gate tiles [tile][quad][256]
    static constexpr int XB = XNH + 2 * 2 * 256;                              // start values [gate r, z, n_x, n_h][tile][16]
    static constexpr int D2_W = XB + 4 * 2 * 16, D2_B = D2_W + 2 * 256;       // inter dense
    static constexpr int GG = D2_B + 16, GBE = GG + 512;                      // conv_glu norm gamma / beta, TRANSPOSED to [f][d]
    static constexpr int F1_W = GBE + 512, F1_B = F1_W + 4 * 256;             // fc1 [tile][256], bias [4][16]
    static constexpr int DW = F1_B + 64, DWB = DW + 2 * 4 * 3 * 16;          // dwconv [t][r][q < 3][lg][e]: tap dt * 3 + df = 4 q + e of channel 16 t + 4 lg + r; bias [t][lg][r]
    static constexpr int F2_W = DWB + 32, F2_B = F2_W + 2 * 256;              // fc2 [quad][256], bias [16]
    static constexpr int B_SIZE = F2_B + 16;
    static constexpr int TOTAL = BLK + 2 * B_SIZE;
    static_assert(BLK % 4 == 0 && B_SIZE % 4 == 0 && I_W % 4 == 0 && XR % 4 == 0 && F1_W % 4 == 0 && DW % 4 == 0 && C2_LO % 4 == 0 && C3_LO % 4 == 0 && C1_P % 4 == 0, "16-byte fragment loads");
};

struct LSbLds {
    static constexpr int X = 0;                        // tokens [32 f][16 slots][16 n]: slot 4 r + lg <-> channel 4 lg + r
    static constexpr int HS = X + 32 * 16 * 16;        // intra h sequences [2 d][32 f][16 slots (12 units)][16 n]; conv_glu: the waves' edge columns [8][2][32][16]
    // prologue: [channel][16 n][positions + pad] staging of the tensors that arrive per stream, two areas (features / x1 [4][16][261] | cached x1; then cached x2
    // [8][16][129] | cached x3 [12][16][65]; later the new cache frames on their way out)
    static constexpr int TA = 0, TB = TA + 4 * 16 * 261, TEND = TB + 4 * 16 * 261;
    static constexpr int T3 = TA;                      // conv_3's new cache frame [12][16][65]
    static constexpr int RED = (HS + 2 * 32 * 16 * 16 > TEND ? HS + 2 * 32 * 16 * 16 : TEND);  // [2][8 waves][16 n]
    static constexpr int TOTAL = RED + 256;
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
};

// r6d experiment (FE_LSB_NT=1): the per-stream cache tensors - read once and written once per step - as non-temporal accesses, so that they do not displace the carry in L2
#ifndef FE_LSB_NT
#define FE_LSB_NT 0
#endif
__device__ __forceinline__ f32x4 lsb_ld_state4(const float* p) {
#if FE_LSB_NT
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
    return *reinterpret_cast<const f32x4*>(p);
#endif
}
__device__ __forceinline__ void lsb_st_state4(float* p, f32x4 v) {
#if FE_LSB_NT
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}

struct LSbArgs {
    const float* wp;          // the packed buffer of lisennet_frame_kernel; the stream-batched region starts at LPk::TOTAL
    int wp_floats;
    float* carry;             // [tiles][LCarry::TILE] [B][LCarry::SP]
    float* cache;             // the model caches (LArgs::cache)
    float* dbg;               // per-stage dumps (fe_debug_step) or nullptr; LDebugLayout
    size_t dbg_stride;
    int B;
    unsigned long long* clk;  // fe_profile_step: cycle counters of workgroup 0 (slots 32 ..), else null
};

__device__ __forceinline__ float lsb_sig(float pre) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre)); }                              // pre = -log2e x
__device__ __forceinline__ float lsb_tanh(float pre) { return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre)), 1.0f); }   // pre = 2 log2e x
// mish(x) = x tanh(softplus(x)) with tanh(ln(1 + n)) = n (n + 2) / (n (n + 2) + 2), n = e^x: one exponential and one reciprocal (lisennet_frame_kernel's
// mish_f goes through log1p, exp and tanh: ~8 x the vector instructions, and the conv_glu of sixteen streams is 16 384 evaluations per block);
// no cancellation for x -> -inf (n (n + 2) -> 2 n), x > 20: the factor is 1 to fp32 (and n^2 would overflow from x = 44)
__device__ __forceinline__ float lsb_mish(float x) {
    const float n = __builtin_amdgcn_exp2f(1.4426950408889634f * fminf(x, 20.0f));
    const float t = n * (n + 2.0f);
    float y = x * (t * __builtin_amdgcn_rcpf(t + 2.0f));
    asm volatile("" : "+v"(y));          // (keeps the evaluations in program order: scheduled freely, the conv_glu passes - straight-line code - spilled 58 registers)
    return y;
}
__device__ __forceinline__ float lsb_sum_lg(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// per-lane source offsets (floats, relative to the tile's carry) of a convolution's quads at output position 0: pair 4 m + lg = (source, tap, group)
template <int G, int TAPS, int NQ>
__device__ __forceinline__ void lsb_src(int (&off)[NQ], int s0, int s1, int lg, int li) {
    constexpr int NPAIR = 2 * TAPS * G;
    static_assert(4 * NQ >= NPAIR, "quads cover the pairs");
#pragma unroll
    for (int m = 0; m < NQ; ++m) {
        int pi = 4 * m + lg;
        pi = pi < NPAIR ? pi : NPAIR - 1;              // (pairs past the end carry zero weights: any valid address)
        const int s = pi / (TAPS * G), rem = pi - s * (TAPS * G), df = rem / G, g = rem - df * G;
        off[m] = (s ? s1 : s0) + (df * G + g) * 64 + li * 4;
    }
}

template <class S>      // (S = LShape<HOP>: the middle does not depend on it - a template so that the kernel is emitted by the translation unit that launches it)
__global__ void __launch_bounds__(kLsbThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) lisennet_sb_kernel(LSbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = LSbLds;
    using Q = LSbPk;
    using A = LCarry;
    constexpr int SB = LPk::TOTAL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int b0 = blockIdx.x * kLsbStreams;
    const bool live = b0 + li < a.B;
    const int bn = live ? b0 + li : a.B - 1;                       // this lane's stream (tiles past the batch repeat the last stream; state stores predicated)
    float* ct = a.carry + (size_t)blockIdx.x * A::TILE;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, a.wp_floats * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(ct, 0, A::TILE * 4, 0x00020000);
    auto ldw4 = [&](int off_floats, int voff_bytes) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff_bytes, off_floats * 4, 0)); };
    auto frag = [&](int off_floats) { return ldw4(off_floats, lane * 16); };                  // one A-fragment quad
    auto row4 = [&](int off_floats) { return ldw4(off_floats, lg * 16); };                    // start values / per-channel parameters [lg][r]
    auto ldc4 = [&](int off_floats) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, off_floats * 4, 0, 0)); };
#define LSB_CLK(i) do { if (a.clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.clk[32 + (i)] = __builtin_readcyclecounter(); } while (0)
    LSB_CLK(0);
    float* X = smem + L::X;
    float* HS = smem + L::HS;
    float* red = smem + L::RED;
    float* dbg = (a.dbg && live) ? a.dbg + (size_t)bn * a.dbg_stride : nullptr;      // (per lane: the idle columns of the last tile dump nothing)
    auto cache_ptr = [&](int off_sum, int per_stream) -> float* { return a.cache + (size_t)off_sum * a.B + (size_t)bn * per_stream; };
    // sum over a stream's values held by the lanes li, li + 16, .. of all eight waves (one barrier; slot 0 / 1 alternate)
    auto tile_sum = [&](float v, int slot) {
        v = lsb_sum_lg(v);
        if (lg == 0) red[slot * 128 + wave * 16 + li] = v;
        __syncthreads();
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[slot * 128 + w * 16 + li];
        return t;
    };

    // ---------------- halos of the sliced tensors; what arrives per stream - the input features (PART 1), the cached conv_1 / conv_2 / conv_3 frames (the cache
    // tensors) - regrouped for the sixteen streams through LDS: coalesced reads along f -> [channel][n][f (+ pad)] -> one 16-byte element (four channels of a
    // position and stream) per thread, whole 256-byte pieces per sixteen lanes; encoder.conv_1 (1x1, 3 -> 4: vector FMAs) + LayerNorm + PReLU on the way ----------------
        auto zero_halo = [&](auto sl) {
            using SL = decltype(sl);
            for (int i = tid; i < 4 * (SL::ROW / 4); i += kLsbThreads) {
                const int h = i / (SL::ROW / 4), q = i - h * (SL::ROW / 4);
                *reinterpret_cast<f32x4*>(ct + SL::halo(h) + 4 * q) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        };
        zero_halo(A::X1C{}); zero_halo(A::X1P{}); zero_halo(A::X2C{}); zero_halo(A::X2P{}); zero_halo(A::X2S{});
        zero_halo(A::X3C{}); zero_halo(A::X3P{}); zero_halo(A::X3S{}); zero_halo(A::X4S{}); zero_halo(A::XD{}); zero_halo(A::U1{}); zero_halo(A::U2{}); zero_halo(A::U3S{}); zero_halo(A::U3P{});
        float* TA = smem + L::TA;
        float* TB = smem + L::TB;
        // a tensor [NCH][NF] per stream (NCH * NF floats at stride `per`, a multiple of 4) of the sixteen streams -> T[(c * 16 + n) * LDT + f]: the stream's tensor read
        // as a flat run of 16-byte pieces (a piece may straddle two channels: NF = 257).  Two steps - ALL four tensors are requested at the top of the kernel
        // (30 pieces per thread in flight: one memory round trip instead of four), each goes to LDS when its staging area is free
        auto stage_load = [&](auto& v, const float* src, size_t per, auto ne_) {
            constexpr int NE4 = decltype(ne_)::value / 4, TOT = 16 * NE4, NV = (TOT + kLsbThreads - 1) / kLsbThreads;
            static_assert(decltype(ne_)::value % 4 == 0 && sizeof(v) == NV * sizeof(f32x4), "whole 16-byte pieces per stream");
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = tid + k * kLsbThreads, ic = i < TOT ? i : TOT - 1, n = ic / NE4, e = (ic - n * NE4) * 4;
                const int bs = b0 + n < a.B ? b0 + n : a.B - 1;
                v[k] = lsb_ld_state4(src + (size_t)bs * per + e);
            }
        };
        auto stage_put = [&](float* T, const auto& v, auto nch_, auto nf_, auto ldt_) {
            constexpr int NCH = decltype(nch_)::value, NF = decltype(nf_)::value, LDT = decltype(ldt_)::value, NE4 = NCH * NF / 4, TOT = 16 * NE4, NV = (TOT + kLsbThreads - 1) / kLsbThreads;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = tid + k * kLsbThreads, n = i / NE4, e = (i - n * NE4) * 4;
                if (i < TOT) {      // (one division per piece: element j sits at f0 + j of channel c0, or - NF = 257 only - wraps into the next channel's row)
                    const int c0 = e / NF, f0 = e - c0 * NF;
                    float* t0 = T + (c0 * 16 + n) * LDT + f0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) t0[(NF % 4 != 0 && f0 + j >= NF) ? j + 16 * LDT - NF : j] = v[k][j];
                }
            }
        };
        // T -> a sliced tensor of the carry: NF positions x G groups x 16 streams
        auto emit = [&](auto sl, const float* T, auto nf_, auto ldt_) {
            using SL = decltype(sl);
            constexpr int NF = decltype(nf_)::value, LDT = decltype(ldt_)::value, G = SL::G, TOT = NF * G * 16;
#pragma unroll 1
            for (int i0 = 0; i0 < TOT; i0 += kLsbThreads) {
                const int i = i0 + tid, ic = i < TOT ? i : TOT - 1, n = ic & 15, gf = ic >> 4, f = gf / G, g = gf - G * f;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = T[((4 * g + j) * 16 + n) * LDT + f];
                if (i < TOT) *reinterpret_cast<f32x4*>(ct + SL::row(f) + g * 64 + n * 4) = o;
            }
        };
        // T -> a cache tensor [NCH][NF] per stream, as a flat run of 16-byte pieces (coalesced)
        auto unstage = [&](float* dst, size_t per, const float* T, auto nch_, auto nf_, auto ldt_) {
            constexpr int NCH = decltype(nch_)::value, NF = decltype(nf_)::value, LDT = decltype(ldt_)::value, NE = NCH * NF, NE4 = NE / 4, TOT = 16 * NE4;
            static_assert(NE % 4 == 0, "whole 16-byte pieces per stream");
#pragma unroll 1
            for (int i0 = 0; i0 < TOT; i0 += kLsbThreads) {
                const int i = i0 + tid, ic = i < TOT ? i : TOT - 1, n = ic / NE4, e = (ic - n * NE4) * 4;
                const int c0 = e / NF, f0 = e - c0 * NF;
                const float* t0 = T + (c0 * 16 + n) * LDT + f0;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = t0[(NF % 4 != 0 && f0 + j >= NF) ? j + 16 * LDT - NF : j];
                if (i < TOT && b0 + n < a.B) lsb_st_state4(dst + (size_t)(b0 + n) * per + e, o);
            }
        };
        using I3 = std::integral_constant<int, 3>;
        using I4 = std::integral_constant<int, 4>;
        using I8 = std::integral_constant<int, 8>;
        using I12 = std::integral_constant<int, 12>;
        using I64 = std::integral_constant<int, 64>;
        using I65 = std::integral_constant<int, 65>;
        using I128 = std::integral_constant<int, 128>;
        using I129 = std::integral_constant<int, 129>;
        using I256 = std::integral_constant<int, 256>;
        using I257 = std::integral_constant<int, 257>;
        using I260 = std::integral_constant<int, 260>;
        using I261 = std::integral_constant<int, 261>;
        float* const c2g = a.cache + (size_t)S::K_PHA * a.B;                                   // cached conv_1 frames [B][4][257]
        float* const c3g = a.cache + (size_t)(S::K_PHA + S::K_E2) * a.B;                       // cached conv_2 frames [B][8][128]
        float* const c4g = a.cache + (size_t)(S::K_PHA + S::K_E2 + S::K_E3) * a.B;             // cached conv_3 frames [B][12][64]
        float* const cdg = a.cache + (size_t)(S::K_PHA + S::K_E2 + S::K_E3 + S::K_E4 + S::NB * (S::K_H + S::K_GLU)) * a.B;      // cached up3 frames [B][4][256]
        f32x4 vf[(16 * 780 / 4 + kLsbThreads - 1) / kLsbThreads], v1[(16 * S::K_E2 / 4 + kLsbThreads - 1) / kLsbThreads], v2[16 * S::K_E3 / 4 / kLsbThreads], v3[16 * S::K_E4 / 4 / kLsbThreads], v4[16 * S::K_DEC / 4 / kLsbThreads];
        stage_load(vf, a.carry + A::feat(a.B), A::FEAT, std::integral_constant<int, 780>{});
        stage_load(v1, c2g, S::K_E2, std::integral_constant<int, S::K_E2>{});
        stage_load(v2, c3g, S::K_E3, std::integral_constant<int, S::K_E3>{});
        stage_load(v3, c4g, S::K_E4, std::integral_constant<int, S::K_E4>{});
        stage_load(v4, cdg, S::K_DEC, std::integral_constant<int, S::K_DEC>{});
        // features [3][260] -> area A, the cached conv_1 frame -> area B
        stage_put(TA, vf, I3{}, I260{}, I261{});
        stage_put(TB, v1, I4{}, I257{}, I261{});
        __syncthreads();
        LSB_CLK(26);
        emit(A::X1P{}, TB, I257{}, I261{});
        // conv_1 + LayerNorm over (channel, freq) + per-frequency affine + PReLU: thread (n = tid % 16, p = tid / 16 + 32 k); same lane structure as the tiles
        {
            const f32x4 w0 = ldw4(SB + Q::C1_W, 0), w1 = ldw4(SB + Q::C1_W + 4, 0), w2 = ldw4(SB + Q::C1_W + 8, 0), cb = ldw4(SB + Q::C1_W + 12, 0), pr = ldw4(SB + Q::C1_P, 0);
            f32x4 v[9];
            float s0 = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int pp = (tid >> 4) + 32 * k, pc = pp < 257 ? pp : 256;
                const float f0 = TA[(0 * 16 + li) * 261 + pc], f1 = TA[(1 * 16 + li) * 261 + pc], f2 = TA[(2 * 16 + li) * 261 + pc];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[k][r] = __builtin_fmaf(w2[r], f2, __builtin_fmaf(w1[r], f1, __builtin_fmaf(w0[r], f0, cb[r])));
                if (pp < 257) s0 += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
            }
            float gam[9], bet[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int pp = (tid >> 4) + 32 * k, pc = pp < 257 ? pp : 256;
                gam[k] = a.wp[SB + Q::C1_G + pc]; bet[k] = a.wp[SB + Q::C1_BE + pc];
            }
            const float mean = tile_sum(s0, 0) * (1.0f / 1028.0f);
            float s1 = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int pp = (tid >> 4) + 32 * k;
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[k][r] -= mean; if (pp < 257) s1 = __builtin_fmaf(v[k][r], v[k][r], s1); }
            }
            const float rstd = 1.0f / sqrtf(tile_sum(s1, 1) * (1.0f / 1028.0f) + 1.0e-5f);
            __syncthreads();                        // (every thread has read the features / emitted the cached frame: the areas are free)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int pp = (tid >> 4) + 32 * k;
                if (pp < 257) {
                    const float ga = gam[k] * rstd, be = bet[k];
                    f32x4 y;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float t = __builtin_fmaf(v[k][r], ga, be); y[r] = t >= 0.0f ? t : t * pr[r]; }
                    *reinterpret_cast<f32x4*>(ct + A::X1C::row(pp) + li * 4) = y;
#pragma unroll
                    for (int r = 0; r < 4; ++r) TA[(r * 16 + li) * 261 + pp] = y[r];
                    if (dbg) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dbg[LDebugLayout::offset(3) + r * 257 + pp] = y[r];
                    }
                }
            }
        }
        LSB_CLK(27);
        stage_put(TB, v2, I8{}, I128{}, I129{});                        // the cached conv_2 frame -> area B
        __syncthreads();
        LSB_CLK(28);
        unstage(c2g, S::K_E2, TA, I4{}, I257{}, I261{});                // the new conv_1 cache frame
        emit(A::X2P{}, TB, I128{}, I129{});
        __syncthreads();
        LSB_CLK(29);
        stage_put(TA, v3, I12{}, I64{}, I65{});                         // the cached conv_3 frame -> area A
        stage_put(TB, v4, I4{}, I256{}, I257{});                        // the cached up3 frame -> area B
        __syncthreads();
        LSB_CLK(30);
        emit(A::X3P{}, TA, I64{}, I65{});
        emit(A::U3P{}, TB, I256{}, I257{});
    __syncthreads();
    LSB_CLK(1);

    // one convolution job of a wave: NP output positions (their source offsets pofs[], floats), NT tiles each
#define LSB_CONV(NQ_, NT_, NP_, acc_, W_, off_, pofs_)                                                                     \
    do {                                                                                                                   \
        f32x4 bq_[NP_][NQ_];                                                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < NP_; ++i_)                                                                 \
            _Pragma("unroll") for (int m_ = 0; m_ < NQ_; ++m_) bq_[i_][m_] = ldc4(off_[m_] + pofs_[i_]);                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        _Pragma("unroll") for (int m_ = 0; m_ < NQ_; ++m_)                                                                 \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                               \
                _Pragma("unroll") for (int i_ = 0; i_ < NP_; ++i_)                                                         \
                    _Pragma("unroll") for (int t_ = 0; t_ < NT_; ++t_)                                                     \
                        acc_[i_][t_] = FE_MFMA(W_[t_][m_][j_], bq_[i_][m_][j_], acc_[i_][t_]);                             \
    } while (0)

    // ---------------- encoder.conv_2: DSConv(4 -> 8, 257 bins -> 128), LayerNorm over (channel, freq), per-frequency affine, PReLU ----------------
    {
        int offL[2], offH[3];
        lsb_src<1, 3, 2>(offL, A::X1P::LO, A::X1C::LO, lg, li);
        lsb_src<1, 5, 3>(offH, A::X1P::HI, A::X1C::HI, lg, li);
        f32x4 WL[1][2], WH[1][3];
#pragma unroll
        for (int m = 0; m < 2; ++m) WL[0][m] = frag(SB + Q::C2_LO + m * 256);
#pragma unroll
        for (int m = 0; m < 3; ++m) WH[0][m] = frag(SB + Q::C2_HI + m * 256);
        const f32x4 bL = row4(SB + Q::C2_BL), bH = row4(SB + Q::C2_BH), pr = row4(SB + Q::C2_P);
        f32x4 accA[4][1], accB[4][1], accC[4][1], accD[4][1];           // low positions 8 w .. + 3 | .. + 7, high positions likewise
        int pA[4], pB[4], pC[4], pD[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            accA[i][0] = bL; accB[i][0] = bL; accC[i][0] = bH; accD[i][0] = bH;
            pA[i] = (8 * wave + i) * A::X1C::ROW; pB[i] = (8 * wave + 4 + i) * A::X1C::ROW;
            pC[i] = (8 * wave + i) * 3 * A::X1C::ROW; pD[i] = (8 * wave + 4 + i) * 3 * A::X1C::ROW;
        }
        LSB_CONV(2, 1, 4, accA, WL, offL, pA);
        LSB_CONV(2, 1, 4, accB, WL, offL, pB);
        LSB_CONV(3, 1, 4, accC, WH, offH, pC);
        LSB_CONV(3, 1, 4, accD, WH, offH, pD);
        const bool val = lg < 2;                                   // rows 8 .. 15 of the tile are idle
        float gam[16], bet[16];                                    // (requested before the statistics' barriers)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int P = i < 8 ? 8 * wave + i : 64 + 8 * wave + (i - 8);
            gam[i] = a.wp[SB + Q::C2_G + P]; bet[i] = a.wp[SB + Q::C2_BE + P];
        }
        auto acc_of = [&](int i) -> f32x4& { return i < 4 ? accA[i][0] : i < 8 ? accB[i - 4][0] : i < 12 ? accC[i - 8][0] : accD[i - 12][0]; };
        float s0 = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const f32x4& v = acc_of(i); s0 += (v[0] + v[1]) + (v[2] + v[3]); }
        const float mean = tile_sum(val ? s0 : 0.0f, 0) * (1.0f / 1024.0f);
        float s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4& v = acc_of(i);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] -= mean; s1 = __builtin_fmaf(v[r], v[r], s1); }
        }
        const float rstd = 1.0f / sqrtf(tile_sum(val ? s1 : 0.0f, 1) * (1.0f / 1024.0f) + 1.0e-5f);
        float* TBo = smem + L::TB;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int P = i < 8 ? 8 * wave + i : 64 + 8 * wave + (i - 8);
            const float ga = gam[i] * rstd, be = bet[i];
            const f32x4& v = acc_of(i);
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float t = __builtin_fmaf(v[r], ga, be); y[r] = t >= 0.0f ? t : t * pr[r]; }
            if (val) {
                *reinterpret_cast<f32x4*>(ct + A::X2C::row(P) + lg * 64 + li * 4) = y;
                *reinterpret_cast<f32x4*>(ct + A::X2S::row(P) + lg * 64 + li * 4) = y;
#pragma unroll
                for (int r = 0; r < 4; ++r) TBo[((4 * lg + r) * 16 + li) * 129 + P] = y[r];      // (the new cache frame: out as whole rows below)
                if (dbg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dbg[LDebugLayout::offset(4) + (4 * lg + r) * 128 + P] = y[r];
                }
            }
        }
    }
    __syncthreads();
    unstage(c3g, S::K_E3, smem + L::TB, I8{}, I128{}, I129{});          // the new conv_2 cache frame
    LSB_CLK(24);
    // ---------------- encoder.conv_3: DSConv(8 -> 12, 128 bins -> 64), LayerNorm over (channel, freq), per-frequency affine, PReLU ----------------
    {
        int offL[3], offH[5];
        lsb_src<2, 3, 3>(offL, A::X2P::LO, A::X2C::LO, lg, li);
        lsb_src<2, 5, 5>(offH, A::X2P::HI, A::X2C::HI, lg, li);
        f32x4 WL[1][3], WH[1][5];
#pragma unroll
        for (int m = 0; m < 3; ++m) WL[0][m] = frag(SB + Q::C3_LO + m * 256);
#pragma unroll
        for (int m = 0; m < 5; ++m) WH[0][m] = frag(SB + Q::C3_HI + m * 256);
        const f32x4 bL = row4(SB + Q::C3_BL), bH = row4(SB + Q::C3_BH), pr = row4(SB + Q::C3_P);
        f32x4 accL[4][1], accH[4][1];
        int pL[4], pH[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { accL[i][0] = bL; accH[i][0] = bH; pL[i] = (4 * wave + i) * A::X2C::ROW; pH[i] = (4 * wave + i) * 3 * A::X2C::ROW; }
        LSB_CONV(3, 1, 4, accL, WL, offL, pL);
        LSB_CONV(5, 1, 4, accH, WH, offH, pH);
        const bool val = lg < 3;                                   // rows 12 .. 15 of the tile are idle
        float gam[8], bet[8];                                      // the per-frequency affine of this wave's positions (requested before the statistics' barriers)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int P = i < 4 ? 4 * wave + i : 32 + 4 * wave + (i - 4);
            gam[i] = a.wp[SB + Q::C3_G + P]; bet[i] = a.wp[SB + Q::C3_BE + P];
        }
        float s0 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s0 += ((accL[i][0][0] + accL[i][0][1]) + (accL[i][0][2] + accL[i][0][3])) + ((accH[i][0][0] + accH[i][0][1]) + (accH[i][0][2] + accH[i][0][3]));
        const float mean = tile_sum(val ? s0 : 0.0f, 0) * (1.0f / 768.0f);
        float s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accL[i][0][r] -= mean; accH[i][0][r] -= mean;
                s1 = __builtin_fmaf(accL[i][0][r], accL[i][0][r], s1);
                s1 = __builtin_fmaf(accH[i][0][r], accH[i][0][r], s1);
            }
        const float rstd = 1.0f / sqrtf(tile_sum(val ? s1 : 0.0f, 1) * (1.0f / 768.0f) + 1.0e-5f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int P = i < 4 ? 4 * wave + i : 32 + 4 * wave + (i - 4);
            const float ga = gam[i] * rstd, be = bet[i];
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = __builtin_fmaf(i < 4 ? accL[i][0][r] : accH[i - 4][0][r], ga, be);
                y[r] = v >= 0.0f ? v : v * pr[r];
            }
            if (val) {
                *reinterpret_cast<f32x4*>(ct + A::X3C::row(P) + lg * 64 + li * 4) = y;
                *reinterpret_cast<f32x4*>(ct + A::X3S::row(P) + lg * 64 + li * 4) = y;
                // the new cache frame [12][64] per stream: through LDS ([channel][n][f + 1 pad], the prologue's staging area) and out as whole rows below -
                // stored from here, one float per (channel, position, stream), it was 12 288 partial-line writes per tile: 35 k of this phase's 54 k cycles
#pragma unroll
                for (int r = 0; r < 4; ++r) smem[L::T3 + ((4 * lg + r) * 16 + li) * 65 + P] = y[r];
                if (dbg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dbg[LDebugLayout::offset(5) + (4 * lg + r) * 64 + P] = y[r];
                }
            }
        }
    }
    __syncthreads();
    {
        float* c4 = a.cache + (size_t)(S::K_PHA + S::K_E2 + S::K_E3) * a.B;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = tid + k * kLsbThreads, n = i / (S::K_E4 / 4), e = (i - n * (S::K_E4 / 4)) * 4, c = e >> 6, f = e & 63;
            const float* t3 = smem + L::T3 + (c * 16 + n) * 65 + f;
            if (b0 + n < a.B) lsb_st_state4(c4 + (size_t)(b0 + n) * S::K_E4 + e, f32x4{t3[0], t3[1], t3[2], t3[3]});
        }
    }
    LSB_CLK(2);
    // ---------------- encoder.conv_4: DSConv(12 -> 16, 64 bins -> 32) -> the up1 skip and the blocks' tokens ----------------
    {
        int offL[5], offH[8];
        lsb_src<3, 3, 5>(offL, A::X3P::LO, A::X3C::LO, lg, li);
        lsb_src<3, 5, 8>(offH, A::X3P::HI, A::X3C::HI, lg, li);
        f32x4 WL[1][5], WH[1][8];
#pragma unroll
        for (int m = 0; m < 5; ++m) WL[0][m] = frag(SB + Q::C4_LO + m * 256);
#pragma unroll
        for (int m = 0; m < 8; ++m) WH[0][m] = frag(SB + Q::C4_HI + m * 256);
        const f32x4 bL = row4(SB + Q::C4_BL), bH = row4(SB + Q::C4_BH), pr = row4(SB + Q::C4_P);
        f32x4 accL[2][1], accH[2][1];
        int pL[2], pH[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { accL[i][0] = bL; accH[i][0] = bH; pL[i] = (2 * wave + i) * A::X3C::ROW; pH[i] = (2 * wave + i) * 3 * A::X3C::ROW; }
        LSB_CONV(5, 1, 2, accL, WL, offL, pL);
        LSB_CONV(8, 1, 2, accH, WH, offH, pH);
        float gam[4], bet[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int P = i < 2 ? 2 * wave + i : 16 + 2 * wave + (i - 2);
            gam[i] = a.wp[SB + Q::C4_G + P]; bet[i] = a.wp[SB + Q::C4_BE + P];
        }
        float s0 = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) s0 += ((accL[i][0][0] + accL[i][0][1]) + (accL[i][0][2] + accL[i][0][3])) + ((accH[i][0][0] + accH[i][0][1]) + (accH[i][0][2] + accH[i][0][3]));
        const float mean = tile_sum(s0, 0) * (1.0f / 512.0f);
        float s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accL[i][0][r] -= mean; accH[i][0][r] -= mean;
                s1 = __builtin_fmaf(accL[i][0][r], accL[i][0][r], s1);
                s1 = __builtin_fmaf(accH[i][0][r], accH[i][0][r], s1);
            }
        const float rstd = 1.0f / sqrtf(tile_sum(s1, 1) * (1.0f / 512.0f) + 1.0e-5f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int P = i < 2 ? 2 * wave + i : 16 + 2 * wave + (i - 2);
            const float ga = gam[i] * rstd, be = bet[i];
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = __builtin_fmaf(i < 2 ? accL[i][0][r] : accH[i - 2][0][r], ga, be);
                y[r] = v >= 0.0f ? v : v * pr[r];
            }
            *reinterpret_cast<f32x4*>(ct + A::X4S::row(P) + lg * 64 + li * 4) = y;
#pragma unroll
            for (int r = 0; r < 4; ++r) X[(P * 16 + 4 * r + lg) * 16 + li] = y[r];
            if (dbg) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dbg[LDebugLayout::offset(6) + (4 * lg + r) * 32 + P] = y[r];
            }
        }
    }
    __syncthreads();
    LSB_CLK(3);
    // ---------------- 2 x DPR: wave w owns sub-bands 4 w .. 4 w + 3; xr[fl][r] = channel 4 lg + r of sub-band 4 w + fl, stream li ----------------
    float xr[4][4];
#pragma unroll
    for (int fl = 0; fl < 4; ++fl)
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[fl][r] = X[((4 * wave + fl) * 16 + 4 * r + lg) * 16 + li];
    // LayerNorm statistics over a stream's 512 values (16 per lane): v -= mean in place, returns 1 / sqrt(var + eps)
    auto ln512 = [&](float (&v)[4][4]) {
        float s0 = 0.0f;
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) s0 += (v[fl][0] + v[fl][1]) + (v[fl][2] + v[fl][3]);
        const float mean = tile_sum(s0, 0) * (1.0f / 512.0f);
        float s1 = 0.0f;
#pragma unroll
        for (int fl = 0; fl < 4; ++fl)
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[fl][r] -= mean; s1 = __builtin_fmaf(v[fl][r], v[fl][r], s1); }
        return 1.0f / sqrtf(tile_sum(s1, 1) * (1.0f / 512.0f) + 1.0e-5f);
    };
    constexpr int OFF_BLK = S::K_PHA + S::K_E2 + S::K_E3 + S::K_E4;
    // (the block's body as a lambda of loop-variant copies of the wave / lane indices - opaque zeros added per iteration: the dozens of per-lane LDS
    //  addresses of a block are loop-invariant, and hoisted out of the block loop they were what the kernel spilled)
    auto dpr_block = [&](const int blk, const int lz, const int wave, const int li, const int lg) __attribute__((always_inline)) {
        const int d = wave >> 2, q = wave & 3;
        const int D = SB + Q::BLK + blk * Q::B_SIZE + lz;
        float* const chp = cache_ptr(OFF_BLK + blk * (S::K_H + S::K_GLU), S::K_H);                  // inter GRU state [32 f][24]
        float* const cgp = cache_ptr(OFF_BLK + blk * (S::K_H + S::K_GLU) + S::K_H, S::K_GLU);       // conv_glu frames [32 ch][2][32 f]
        // ---- intra_norm (nn.LayerNorm((32, 16))) -> normalised tokens in LDS
        {
            float v[4][4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[fl][r] = xr[fl][r];
            f32x4 nw[4], nb[4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) { nw[fl] = row4(D + Q::N1W + (4 * wave + fl) * 16); nb[fl] = row4(D + Q::N1B + (4 * wave + fl) * 16); }
            const float rstd = ln512(v);
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[((4 * wave + fl) * 16 + 4 * r + lg) * 16 + li] = __builtin_fmaf(v[fl][r] * rstd, nw[fl][r], nb[fl][r]);
        }
        float pf[3];
        const f32x4 awx = frag(D + Q::I_W + (wave * 2 + 0) * 256), awh = frag(D + Q::I_W + (wave * 2 + 1) * 256);
        const f32x4 ibias = row4(D + Q::I_B + wave * 16);
        __syncthreads();
        LSB_CLK(4 + 6 * blk);
        // ---- intra GRU: 32 steps, both directions, one barrier per step (fspen_sb_kernels.hip.h)
        {
            const int f0 = d ? 31 : 0, fstep = d ? -1 : 1;
            auto xload = [&](float (&xb)[4], int f) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) xb[ks] = X[(f * 16 + 4 * ks + lg) * 16 + li];
            };
            auto xpart = [&](const float (&xb)[4]) {
                f32x4 p = ibias;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) p = FE_MFMA(awx[ks], xb[ks], p);
                return p;
            };
            auto gates = [&](const f32x4& acc, float h) {
                const float rg = lsb_sig(acc[0]), zg = lsb_sig(acc[1]);
                const float ng = lsb_tanh(__builtin_fmaf(rg, acc[3], acc[2]));
                return __builtin_fmaf(zg, h - ng, ng);                      // (1 - z) n + z h
            };
            const int hslot = lg < 3 ? 3 * q + lg : 12 + q;                 // (lane group 3: idle rows -> the unused slots 12 .. 15)
            float xb[4];
            xload(xb, f0);
            f32x4 accx = xpart(xb);
            xload(xb, f0 + fstep);
            f32x4 accn = xpart(xb);                                         // step 1's x half
            float h = gates(accx, 0.0f);
            HS[((d * 32 + f0) * 16 + hslot) * 16 + li] = h;
            accx = accn;
            xload(xb, f0 + 2 * fstep);
            __syncthreads();
            // the block's cache tensors (inter GRU state 3 KB, conv_glu frames 8 KB per stream: first touched this step, so in HBM) are requested HERE - behind the
                // recurrence's own fragment fetches (vmcnt completes in order), one dword per 128-byte line and thread - and land in L2 under its 29 k cycles
            {
                const int pn = tid >> 5, pj = tid & 31, pbs = b0 + pn < a.B ? b0 + pn : a.B - 1;
                const float* pg = a.cache + (size_t)(OFF_BLK + blk * (S::K_H + S::K_GLU) + S::K_H) * a.B + (size_t)pbs * S::K_GLU;
                const float* ph = a.cache + (size_t)(OFF_BLK + blk * (S::K_H + S::K_GLU)) * a.B + (size_t)pbs * S::K_H;
                pf[0] = pg[(2 * pj) * 32]; pf[1] = pg[(2 * pj + 1) * 32]; pf[2] = ph[(pj < 24 ? pj : 23) * 32];
            }
            int f = f0 + fstep;
#pragma unroll 1
            for (int s = 1; s < 32; ++s) {
                float hb[3];
                const float* hq = HS + ((d * 32 + (f - fstep)) * 16) * 16;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) hb[ks] = hq[(4 * ks + lg) * 16 + li];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) accx = FE_MFMA(awh[ks], hb[ks], accx);
                accn = xpart(xb);                 // the next step's x half
                h = gates(accx, h);
                HS[((d * 32 + f) * 16 + hslot) * 16 + li] = h;
                accx = accn;
                f += fstep;
                const int fn = f + fstep;
                xload(xb, fn & 31);               // (the load after the last step is not used)
                __syncthreads();
            }
        }
        asm volatile("" :: "v"(pf[0]), "v"(pf[1]), "v"(pf[2]));         // (the touches above: consumed here so that they are not dropped)
        LSB_CLK(5 + 6 * blk);
        // the inter GRU state of this wave's sub-bands: unit 16 t + 4 lg + r (24 units: tile 1's lane groups 2, 3 are idle)
        f32x4 hp[4][2];
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) {
            hp[fl][0] = *reinterpret_cast<const f32x4*>(chp + (4 * wave + fl) * 24 + 4 * lg);
            hp[fl][1] = lg < 2 ? *reinterpret_cast<const f32x4*>(chp + (4 * wave + fl) * 24 + 16 + 4 * lg) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        // ---- intra dense (24 -> 16) + residual
        float x1[4][4];
        {
            const f32x4 fw0 = frag(D + Q::D1_W), fw1 = frag(D + Q::D1_W + 256), fb = row4(D + Q::D1_B);
            f32x4 y[4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) y[fl] = fb;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                    for (int fl = 0; fl < 4; ++fl) {
                        const float hb = HS[(((m * 32) + 4 * wave + fl) * 16 + 4 * ks + lg) * 16 + li];
                        y[fl] = FE_MFMA(m ? fw1[ks] : fw0[ks], hb, y[fl]);
                    }
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) x1[fl][r] = xr[fl][r] + y[fl][r];
        }
        if (dbg) {
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbg[LDebugLayout::offset(7 + 3 * blk) + (4 * wave + fl) * 16 + 4 * lg + r] = x1[fl][r];
        }
        // ---- inter_norm + ONE GRU step over time per sub-band + dense + residual
        float x2v[4][4];
        {
            float yn[4][4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) yn[fl][r] = x1[fl][r];
            f32x4 nw[4], nb[4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) { nw[fl] = row4(D + Q::N2W + (4 * wave + fl) * 16); nb[fl] = row4(D + Q::N2B + (4 * wave + fl) * 16); }
            const f32x4 cb = row4(D + Q::D2_B);
            const float rstd = ln512(yn);
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) yn[fl][r] = __builtin_fmaf(yn[fl][r] * rstd, nw[fl][r], nb[fl][r]);
            f32x4 o[4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) o[fl] = cb;
            // two passes, one per tile of hidden units (0 .. 15 | 16 .. 23): a pass holds its nine gate fragments, the dense layer's sum runs through both
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                f32x4 gr[3], gz[3], gnh[2];
#pragma unroll
                for (int m = 0; m < 3; ++m) { gr[m] = frag(D + Q::XR + (t * 3 + m) * 256); gz[m] = frag(D + Q::XZ + (t * 3 + m) * 256); }
                const f32x4 gnx = frag(D + Q::XNX + t * 256);
                gnh[0] = frag(D + Q::XNH + (t * 2) * 256); gnh[1] = frag(D + Q::XNH + (t * 2 + 1) * 256);
                const f32x4 cw = frag(D + Q::D2_W + t * 256);
                const f32x4 gbr = row4(D + Q::XB + (0 * 2 + t) * 16), gbz = row4(D + Q::XB + (1 * 2 + t) * 16), gbx = row4(D + Q::XB + (2 * 2 + t) * 16),
                            gbh = row4(D + Q::XB + (3 * 2 + t) * 16);
#pragma unroll
                for (int fp = 0; fp < 4; fp += 2) {         // two sub-bands at a time: eight independent accumulator chains
                    f32x4 ar[2], az[2], anx[2], anh[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) { ar[e] = gbr; az[e] = gbz; anx[e] = gbx; anh[e] = gbh; }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            ar[e] = FE_MFMA(gr[0][j], yn[fp + e][j], ar[e]);
                            az[e] = FE_MFMA(gz[0][j], yn[fp + e][j], az[e]);
                            anx[e] = FE_MFMA(gnx[j], yn[fp + e][j], anx[e]);
                        }
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                ar[e] = FE_MFMA(gr[1 + m][j], hp[fp + e][m][j], ar[e]);
                                az[e] = FE_MFMA(gz[1 + m][j], hp[fp + e][m][j], az[e]);
                                anh[e] = FE_MFMA(gnh[m][j], hp[fp + e][m][j], anh[e]);
                            }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int fl = fp + e;
                        f32x4 hn;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float rg = lsb_sig(ar[e][r]), zg = lsb_sig(az[e][r]);
                            const float ng = lsb_tanh(__builtin_fmaf(rg, anh[e][r], anx[e][r]));
                            hn[r] = __builtin_fmaf(zg, hp[fl][t][r] - ng, ng);
                        }
                        // (the old state of both tiles is in registers; no other lane reads these rows)
                        if (live && (t == 0 || lg < 2)) lsb_st_state4(chp + (4 * wave + fl) * 24 + 16 * t + 4 * lg, hn);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[fl] = FE_MFMA(cw[j], hn[j], o[fl]);
                    }
                }
            }
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) x2v[fl][r] = x1[fl][r] + o[fl][r];
        }
        if (dbg) {
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbg[LDebugLayout::offset(8 + 3 * blk) + (4 * wave + fl) * 16 + 4 * lg + r] = x2v[fl][r];
        }
        LSB_CLK(6 + 6 * blk);
        // ---- ConvolutionalGLU: CustomLayerNorm over (d, f), fc1 (16 -> 64), causal 3 x 3 depthwise conv on the first half, Mish, gate, fc2 (32 -> 16), residual
        {
            float zn[4][4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) zn[fl][r] = x2v[fl][r];
            f32x4 nw[4], nb[4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) { nw[fl] = row4(D + Q::GG + (4 * wave + fl) * 16); nb[fl] = row4(D + Q::GBE + (4 * wave + fl) * 16); }
            f32x4 f1w[2], f1b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { f1w[t] = frag(D + Q::F1_W + t * 256); f1b[t] = row4(D + Q::F1_B + t * 16); }
            const float rstd = ln512(zn);
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) zn[fl][r] = __builtin_fmaf(zn[fl][r] * rstd, nw[fl][r], nb[fl][r]);
            __builtin_amdgcn_sched_barrier(0);
            if (blk == 0) LSB_CLK(20);
            // fc1's first half (the conv input, channels 16 t + 4 lg + r) of the wave's EDGE sub-bands only, for the neighbour waves; a pass below computes
            // its channel tile for all four sub-bands again (16 more MFMAs per block: keeping both tiles across the passes is what spilled)
            f32x4 xe[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                f32x4 acc[2] = {f1b[0], f1b[1]};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = FE_MFMA(f1w[t][j], zn[3 * e][j], acc[t]);
                xe[0][e] = acc[0]; xe[1][e] = acc[1];
            }
            __builtin_amdgcn_sched_barrier(0);          // (keeps the edge-column section's loads and address arithmetic below fc1: one long basic block otherwise, scheduled into spills)
            // the three frames' edge columns of every wave's four sub-bands meet in LDS (tokens and h sequences are dead: the statistics' barriers lie
            // in between): [wave][side][frame][slot 16 t + 4 r + lg][16 n] - no wave reads another wave's columns of the cache tensor, so each
            // wave replaces its own columns as soon as it has read them
            float* halo = smem + L::X;
            static_assert(L::HS == L::X + 32 * 16 * 16 && 8 * 2 * 3 * 32 * 16 <= 3 * 32 * 16 * 16, "the edge columns alias tokens + h sequences");
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* src = cgp + (16 * t + 4 * lg + r) * 64 + 4 * wave;      // the cached frames t - 2, t - 1: this wave's four columns of channel 16 t + 4 lg + r
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(src), o1 = *reinterpret_cast<const f32x4*>(src + 32);
                    const int sl = t * 16 + 4 * r + lg;
                    halo[(((wave * 2 + 0) * 3 + 0) * 32 + sl) * 16 + li] = o0[0];
                    halo[(((wave * 2 + 1) * 3 + 0) * 32 + sl) * 16 + li] = o0[3];
                    halo[(((wave * 2 + 0) * 3 + 1) * 32 + sl) * 16 + li] = o1[0];
                    halo[(((wave * 2 + 1) * 3 + 1) * 32 + sl) * 16 + li] = o1[3];
                    halo[(((wave * 2 + 0) * 3 + 2) * 32 + sl) * 16 + li] = xe[t][0][r];
                    halo[(((wave * 2 + 1) * 3 + 2) * 32 + sl) * 16 + li] = xe[t][1][r];
                }
            f32x4 gg[2][4];
            __syncthreads();
            if (blk == 0) LSB_CLK(21);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                if (blk == 0 && t == 1) LSB_CLK(22);
                const f32x4 wbq = row4(D + Q::DWB + t * 16);
                // fc1 for this pass's channels: the conv half (tile t) and the gate half (tile 2 + t)
                f32x4 xc[4], vv[4];
                {
                    const f32x4 gw = frag(D + Q::F1_W + (2 + t) * 256), gb = row4(D + Q::F1_B + (2 + t) * 16);
#pragma unroll
                    for (int fl = 0; fl < 4; ++fl) { xc[fl] = f1b[t]; vv[fl] = gb; }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int fl = 0; fl < 4; ++fl) { xc[fl] = FE_MFMA(f1w[t][j], zn[fl][j], xc[fl]); vv[fl] = FE_MFMA(gw[j], zn[fl][j], vv[fl]); }
                }
                f32x4 old0[4], old1[4];                 // (read again: a pass keeps one channel tile's frames in registers)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* src = cgp + (16 * t + 4 * lg + r) * 64 + 4 * wave;
                    old0[r] = *reinterpret_cast<const f32x4*>(src);
                    old1[r] = *reinterpret_cast<const f32x4*>(src + 32);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int sl = t * 16 + 4 * r + lg;
                    float el[3], er[3];                 // columns 4 w - 1 / 4 w + 4 of the three frames (zero padding at the ends of the 32 sub-bands)
#pragma unroll
                    for (int fr = 0; fr < 3; ++fr) {
                        // (unconditional reads of a clamped slot, then a select: as conditional reads they became wave-uniform branches whose loads were hoisted to the top of the pass)
                        const float tl = halo[((((wave > 0 ? wave - 1 : 0) * 2 + 1) * 3 + fr) * 32 + sl) * 16 + li];
                        const float tr = halo[((((wave < 7 ? wave + 1 : 7) * 2 + 0) * 3 + fr) * 32 + sl) * 16 + li];
                        el[fr] = wave > 0 ? tl : 0.0f;
                        er[fr] = wave < 7 ? tr : 0.0f;
                    }
                    const float c0[6] = {el[0], old0[r][0], old0[r][1], old0[r][2], old0[r][3], er[0]};
                    const float c1[6] = {el[1], old1[r][0], old1[r][1], old1[r][2], old1[r][3], er[1]};
                    const float c2[6] = {el[2], xc[0][r], xc[1][r], xc[2][r], xc[3][r], er[2]};
                    f32x4 wq[3];                        // this channel's nine taps (dt * 3 + df = 4 q + e)
#pragma unroll
                    for (int q3 = 0; q3 < 3; ++q3) wq[q3] = row4(D + Q::DW + ((t * 4 + r) * 3 + q3) * 16);
#pragma unroll
                    for (int fl = 0; fl < 4; ++fl) {
                        float acc = wbq[r];
#pragma unroll
                        for (int df = 0; df < 3; ++df) {
                            acc = __builtin_fmaf(wq[df / 4][df % 4], c0[fl + df], acc);
                            acc = __builtin_fmaf(wq[(3 + df) / 4][(3 + df) % 4], c1[fl + df], acc);
                            acc = __builtin_fmaf(wq[(6 + df) / 4][(6 + df) % 4], c2[fl + df], acc);
                        }
                        gg[t][fl][r] = lsb_mish(acc) * vv[fl][r];
                    }
                    if (live) {         // the new cache: frames (t - 1, t) - this wave's own columns, which no other wave reads from the tensor
                        float* dst = cgp + (16 * t + 4 * lg + r) * 64 + 4 * wave;
                        lsb_st_state4(dst, old1[r]);
                        lsb_st_state4(dst + 32, f32x4{xc[0][r], xc[1][r], xc[2][r], xc[3][r]});
                    }
                    __builtin_amdgcn_sched_barrier(0);  // (one channel at a time: sixteen short independent Mish chains interleaved were scheduled into spills)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (blk == 0) LSB_CLK(23);
            const f32x4 f2w0 = frag(D + Q::F2_W), f2w1 = frag(D + Q::F2_W + 256), f2b = row4(D + Q::F2_B);
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) {
                f32x4 o = f2b;
#pragma unroll
                for (int j = 0; j < 4; ++j) o = FE_MFMA(f2w0[j], gg[0][fl][j], o);
#pragma unroll
                for (int j = 0; j < 4; ++j) o = FE_MFMA(f2w1[j], gg[1][fl][j], o);
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[fl][r] = x2v[fl][r] + o[r];
            }
        }
        if (dbg) {
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbg[LDebugLayout::offset(9 + 3 * blk) + (4 * lg + r) * 32 + 4 * wave + fl] = xr[fl][r];
        }
        __syncthreads();                    // (the next block overwrites the tokens / the edge columns)
        LSB_CLK(7 + 6 * blk);
    };
#pragma unroll 1
    for (int blk = 0; blk < S::NB; ++blk) {
        int lz = 0, lzv = 0;
        asm volatile("" : "+s"(lz));
        asm volatile("" : "+v"(lzv));
        dpr_block(blk, lz, wave + lz, li + lzv, lg + lzv);
    }
    // ---------------- MaskDecoder.up1 .. up3 (USConv over cat(x, skip)): low half k 3, high half k 3 to 3 x cout channels, pixel-shuffled over frequency ----------------
#pragma unroll
    for (int fl = 0; fl < 4; ++fl) *reinterpret_cast<f32x4*>(ct + A::XD::row(4 * wave + fl) + lg * 64 + li * 4) = f32x4{xr[fl][0], xr[fl][1], xr[fl][2], xr[fl][3]};
    __syncthreads();
    LSB_CLK(16);
    {   // up1: (16 + 16) -> 12, 32 bins -> 64: wave w takes low positions 2 w, 2 w + 1 and high positions 2 w, 2 w + 1 (three tiles: the sub-pixel phases)
        int offL[6], offH[6];
        lsb_src<4, 3, 6>(offL, A::XD::LO, A::X4S::LO, lg, li);
        lsb_src<4, 3, 6>(offH, A::XD::HI, A::X4S::HI, lg, li);
        int pp[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) pp[i] = (2 * wave + i) * A::XD::ROW;
        {
            f32x4 W[1][6];
#pragma unroll
            for (int m = 0; m < 6; ++m) W[0][m] = frag(SB + Q::U1_LO + m * 256);
            const f32x4 bL = row4(SB + Q::U1_BL);
            f32x4 acc[2][1] = {{bL}, {bL}};
            LSB_CONV(6, 1, 2, acc, W, offL, pp);
            if (lg < 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(ct + A::U1::row(2 * wave + i) + lg * 64 + li * 4) = acc[i][0];
            }
        }
        {
            f32x4 W[3][6];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int m = 0; m < 6; ++m) W[t][m] = frag(SB + Q::U1_HI + (t * 6 + m) * 256);
            f32x4 acc[2][3];
#pragma unroll
            for (int t = 0; t < 3; ++t) { acc[0][t] = row4(SB + Q::U1_BH + t * 16); acc[1][t] = acc[0][t]; }
            LSB_CONV(6, 3, 2, acc, W, offH, pp);
            if (lg < 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < 3; ++t) *reinterpret_cast<f32x4*>(ct + A::U1::row(16 + 3 * (2 * wave + i) + t) + lg * 64 + li * 4) = acc[i][t];
            }
        }
    }
    __syncthreads();
    LSB_CLK(17);
    {   // up2: (12 + 12) -> 8, 64 bins -> 128: wave w takes low positions 4 w .. 4 w + 3 and high positions 4 w .. 4 w + 3 (two tiles: phases (0, 1) | 2)
        int offL[5], offH[5];
        lsb_src<3, 3, 5>(offL, A::U1::LO, A::X3S::LO, lg, li);
        lsb_src<3, 3, 5>(offH, A::U1::HI, A::X3S::HI, lg, li);
        int pp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pp[i] = (4 * wave + i) * A::U1::ROW;
        {
            f32x4 W[1][5];
#pragma unroll
            for (int m = 0; m < 5; ++m) W[0][m] = frag(SB + Q::U2_LO + m * 256);
            const f32x4 bL = row4(SB + Q::U2_BL);
            f32x4 acc[4][1] = {{bL}, {bL}, {bL}, {bL}};
            LSB_CONV(5, 1, 4, acc, W, offL, pp);
            if (lg < 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(ct + A::U2::row(4 * wave + i) + lg * 64 + li * 4) = acc[i][0];
            }
        }
        {
            f32x4 W[2][5];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int m = 0; m < 5; ++m) W[t][m] = frag(SB + Q::U2_HI + (t * 5 + m) * 256);
            f32x4 acc[4][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[0][t] = row4(SB + Q::U2_BH + t * 16);
#pragma unroll
                for (int i = 1; i < 4; ++i) acc[i][t] = acc[0][t];
            }
            LSB_CONV(5, 2, 4, acc, W, offH, pp);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int P = 32 + 3 * (4 * wave + i);
                *reinterpret_cast<f32x4*>(ct + A::U2::row(P + (lg >> 1)) + (lg & 1) * 64 + li * 4) = acc[i][0];        // rows 0 .. 7: phase 0, rows 8 .. 15: phase 1
                if (lg < 2) *reinterpret_cast<f32x4*>(ct + A::U2::row(P + 2) + lg * 64 + li * 4) = acc[i][1];
            }
        }
    }
    __syncthreads();
    LSB_CLK(18);
    {   // up3: (8 + 8) -> 4, 128 bins -> 256: wave w takes low positions 8 w .. 8 w + 7 and high positions 8 w .. 8 w + 7 (one tile: rows 4 phase + c)
        int offL[3], offH[3];
        lsb_src<2, 3, 3>(offL, A::U2::LO, A::X2S::LO, lg, li);
        lsb_src<2, 3, 3>(offH, A::U2::HI, A::X2S::HI, lg, li);
        f32x4 WL[1][3], WH[1][3];
#pragma unroll
        for (int m = 0; m < 3; ++m) { WL[0][m] = frag(SB + Q::U3_LO + m * 256); WH[0][m] = frag(SB + Q::U3_HI + m * 256); }
        const f32x4 bL = row4(SB + Q::U3_BL), bH = row4(SB + Q::U3_BH);
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            int pp[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pp[i] = (8 * wave + 4 * hlf + i) * A::U2::ROW;
            f32x4 accL[4][1] = {{bL}, {bL}, {bL}, {bL}}, accH[4][1] = {{bH}, {bH}, {bH}, {bH}};
            LSB_CONV(3, 1, 4, accL, WL, offL, pp);
            LSB_CONV(3, 1, 4, accH, WH, offH, pp);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = 8 * wave + 4 * hlf + i;
                if (lg == 0) *reinterpret_cast<f32x4*>(ct + A::U3S::row(p) + li * 4) = accL[i][0];
                if (lg < 3) *reinterpret_cast<f32x4*>(ct + A::U3S::row(64 + 3 * p + lg) + li * 4) = accH[i][0];
#pragma unroll
                for (int r = 0; r < 4; ++r) {               // the new cache frame [4][256] per stream: out as whole rows below
                    if (lg == 0) smem[L::TA + (r * 16 + li) * 257 + p] = accL[i][0][r];
                    if (lg < 3) smem[L::TA + (r * 16 + li) * 257 + 64 + 3 * p + lg] = accH[i][0][r];
                }
                if (dbg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (lg == 0) dbg[LDebugLayout::offset(13) + r * 256 + p] = accL[i][0][r];
                        if (lg < 3) dbg[LDebugLayout::offset(13) + r * 256 + 64 + 3 * p + lg] = accH[i][0][r];
                    }
                }
            }
        }
    }
    __syncthreads();
    LSB_CLK(19);
    unstage(cdg, S::K_DEC, smem + L::TA, I4{}, I256{}, I257{});         // the new up3 cache frame
    // ---------------- mask_conv (:283-288, :304-308): Conv2d(4 -> 2, (2, 2), padding (0, 1)) over (cached, this) frame, LayerNorm over (channel, freq), PReLU, 1x1, learnable sigmoid ----------------
    {
        int offM[1];
        lsb_src<1, 2, 1>(offM, A::U3P::LO, A::U3S::LO, lg, li);
        f32x4 WM[1][1] = {{frag(SB + Q::M0_W)}};
        const f32x4 bM = row4(SB + Q::M0_B);
        // wave w takes bins w, w + 8, ..: 33 of them (past bin 256: clamped, not kept); rows 0, 1 of a tile = the two channels, in lane group 0 - they go to LDS
        // [channel][bin][16 n] and the rest of the head (LayerNorm, PReLU, 1x1, sigmoid) runs one (stream, bin) pair per thread like conv_1: from lane group 0
        // alone it was 33 dependent rounds of three parameter loads per wave (36 k cycles for 1028 MFMAs)
        float* M = smem + L::TB;
#pragma unroll
        for (int c8 = 0; c8 < 5; ++c8) {
            f32x4 acc[8][1];
            int pp[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int p = wave + 8 * (8 * c8 + i); acc[i][0] = bM; pp[i] = (p < 257 ? p : 256) * A::U3S::ROW; }
            if (c8 < 4 || wave == 0) {                         // (the fifth round is bin 256 alone: wave 0)
                LSB_CONV(1, 1, 8, acc, WM, offM, pp);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int p = wave + 8 * (8 * c8 + i);
                    if (lg == 0 && p < 257) { M[(0 * 257 + p) * 16 + li] = acc[i][0][0]; M[(1 * 257 + p) * 16 + li] = acc[i][0][1]; }
                }
            }
        }
        __syncthreads();
        float y0[9], y1[9], gam[9], bet[9], slo[9];
        float s0 = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int p = (tid >> 4) + 32 * k, pc = p < 257 ? p : 256;
            y0[k] = M[(0 * 257 + pc) * 16 + li]; y1[k] = M[(1 * 257 + pc) * 16 + li];
            gam[k] = a.wp[SB + Q::M_G + pc]; bet[k] = a.wp[SB + Q::M_BE + pc]; slo[k] = a.wp[SB + Q::M_SL + pc];
            if (p < 257) s0 += y0[k] + y1[k];
        }
        const float mean = tile_sum(s0, 0) * (1.0f / 514.0f);
        float s1 = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int p = (tid >> 4) + 32 * k;
            y0[k] -= mean; y1[k] -= mean;
            if (p < 257) s1 = __builtin_fmaf(y0[k], y0[k], __builtin_fmaf(y1[k], y1[k], s1));
        }
        const float rstd = 1.0f / sqrtf(tile_sum(s1, 1) * (1.0f / 514.0f) + 1.0e-5f);
        const f32x4 prm = ldw4(SB + Q::M_P, 0), m3a = ldw4(SB + Q::M3, 0), m3b = ldw4(SB + Q::M3 + 4, 0);      // PReLU [2]; W3 [o][c] (4); b3 [2]
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int p = (tid >> 4) + 32 * k;
            if (p < 257) {
                const float ga = gam[k] * rstd;
                float u0 = __builtin_fmaf(y0[k], ga, bet[k]), u1 = __builtin_fmaf(y1[k], ga, bet[k]);
                u0 = u0 >= 0.0f ? u0 : u0 * prm[0];
                u1 = u1 >= 0.0f ? u1 : u1 * prm[1];
                const float z0 = m3b[0] + m3a[0] * u0 + m3a[1] * u1, z1 = m3b[1] + m3a[2] * u0 + m3a[3] * u1;
                const float k0 = sigmoid_f(slo[k] * z0), k1 = sigmoid_f(slo[k] * z1);
                *reinterpret_cast<float2*>(ct + A::MK + (p * 16 + li) * 2) = make_float2(k0, k1);
                if (dbg) { dbg[LDebugLayout::offset(14) + 2 * p] = k0; dbg[LDebugLayout::offset(14) + 2 * p + 1] = k1; }
            }
        }
    }
    LSB_CLK(25);
#undef LSB_CONV
#undef LSB_CLK
}

template <class S>
hipError_t lisennet_sb_launch(const LSbArgs& a, hipStream_t st) {
    static std::atomic<bool> attr_set[64];          // (per device: a process may drive several)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lisennet_sb_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSbLds::BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    const int grid = (a.B + kLsbStreams - 1) / kLsbStreams;
    note_kernel("lisennet_sb_kernel");
    hipLaunchKernelGGL(lisennet_sb_kernel<S>, dim3(grid), dim3(kLsbThreads), LSbLds::BYTES, st, a);
    return hipGetLastError();
}

}  // namespace fe
