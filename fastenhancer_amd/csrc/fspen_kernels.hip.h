// fspen_kernels.hip.h — the FSPEN baseline model (models/fspen/model.py of the reference, configs/others/fspen.yaml) as one
// fused per-frame kernel for gfx950: STFT -> compress -> sub-band / full-band encoders -> feature merge -> 3 x DPE (intra
// bidirectional GRU over the 32 sub-bands, LayerNorm, 8 grouped inter GRUs over time with their states in the stream state)
// -> feature split -> sub-band / full-band decoders -> masks -> un-compress -> iSTFT.  SURVEY.md §8(f) rank 4.
//
// The model is tiny (0.9 MMAC per frame, 79 k weights) and most of it is M = 1 work per stream: one workgroup per stream,
// every activation in 54 KB of LDS (three workgroups per CU), VALU dot products with weights packed k-major by the host
// (fe_api.hip::pack_weights_fspen) so that the lanes of a wave read consecutive floats, and a wave-per-direction
// recurrence for the intra GRU (W_hh of a direction in 48 registers of lanes 0..15, h broadcast with v_readlane).
// It is latency-bound (96 dependent GRU steps per frame), not a GEMM: no MFMA here.
#pragma once
#include "fe_kernels.hip.h"

namespace fe {

// fixed architecture of configs/others/fspen.yaml (the only shipped yaml; fe_create rejects anything else)
template <int HOP_>
struct FShape {
    static constexpr int HOP = HOP_, NFFT = 512, LOG2N = 9, OVL = NFFT - HOP;
    static constexpr int BINS = 257;
    static constexpr int NB = 3, C = 16, F = 32, G = 8, FG = F / G;       // DPE: blocks, channels, sub-bands, groups
    static constexpr int NCACHE = NB * G;
    static constexpr int CACHE_FLOATS = NCACHE * FG * C;                  // per stream
};

// ---- the stream-batched DPE kernel's region of the packed weights (fspen_sb_kernels.hip.h; per DPE block, offsets in floats)
struct FSbPk {
    static constexpr int I_W = 0;                       // intra GRU: A fragments [wave = 4 d + q][k-step < 8 (x | h)][64], rows (unit 4 q + j, gate r z n_x n_h)
    static constexpr int I_B = I_W + 8 * 8 * 64;        // [wave][lg = unit 4 q + lg][gate]: b_ih + b_hh (r, z), b_in, b_hn - pre-scaled like the rows
    static constexpr int FC_W = I_B + 128;              // intra_fc: A fragments [k-step < 8 (fwd | bwd)][64], row 4 lg + r <-> channel 4 r + lg
    static constexpr int FC_B = FC_W + 512;             // [lg][r]
    static constexpr int LN_W = FC_B + 16;              // [f][lg][r]
    static constexpr int LN_B = LN_W + 512;
    static constexpr int GRP = LN_B + 512;              // 8 groups x { gate fragments [24][64]: r (x | h), z (x | h), n_x, n_h; biases [gate][lg][r]; inter_fc [4][64]; its bias [lg][r] }
    static constexpr int G_W = 0, G_B = 24 * 64, G_FCW = G_B + 64, G_FCB = G_FCW + 256, G_SIZE = G_FCB + 16;
    static constexpr int D_SIZE = GRP + 8 * G_SIZE;
    // after the three blocks: feature merge / split (row 4 lg + r <-> output 4 r + lg unless noted)
    static constexpr int FE2_W = 3 * D_SIZE;             // fullband_encoder.2: A fragments [o tile < 2][k-step < 24 = 4 tap + cq][64]; bias [tile][lg][r]
    static constexpr int FE2_B = FE2_W + 48 * 64;
    static constexpr int POST_W = FE2_B + 32;            // fullband_encoder_post: A fragments [o tile < 2][k-step < 8][64]
    static constexpr int MG1_W = POST_W + 16 * 64;       // feature_merge.0: [j tile < 2][k-step < 16][64]; k-steps 0-7 natural (input 4 ks + lg), 8 + 4 q + e <-> input 32 + 16 q + 4 lg + e
    static constexpr int MG2_W = MG1_W + 32 * 64;        // feature_merge.2: [k-step < 8][64]; bias [lg][r]
    static constexpr int MG2_B = MG2_W + 8 * 64;
    static constexpr int SP1_W = MG2_B + 16;             // feature_split.0: [ch tile < 2][k-step < 4][64]; bias [tile][lg][r]
    static constexpr int SP1_B = SP1_W + 8 * 64;
    static constexpr int SP2_W = SP1_B + 32;             // feature_split.1: [j tile < 4][k-step < 8][64]; tiles 2, 3 (sub-band half): rows in natural order
    static constexpr int FD0_W = SP2_W + 32 * 64;        // fullband_decoder.0.0 (1x1, 64 -> 32): [o tile < 2][k-step < 16][64]; k-steps 0-7 x_full, 8-15 enc_out[2]
    static constexpr int FD0T_W = FD0_W + 32 * 64;       // fullband_decoder.0.1 (ConvTranspose1d 32 -> 16, k 6, s 2): [parity < 2][tap < 3][k-step < 8][64], rows in natural order,
                                                         // tap i of parity q = kernel index q + 2 i <-> input position m + 1 - i of output positions 2 m + q
    static constexpr int FD0T_B = FD0T_W + 48 * 64;      // [16]
    static constexpr int FD1_W = FD0T_B + 16;            // fullband_decoder.1.0 (1x1, 32 -> 16): [k-step < 8][64]; k-steps 0-3: d2 channel 4 lg + ks, 4-7: enc_out[1] channel 4 (ks - 4) + lg
    static constexpr int FD1T_W = FD1_W + 8 * 64;        // fullband_decoder.1.1 (ConvTranspose1d 16 -> 4, k 8, s 2): [input position j < 5][cq < 4][64], rows 4 q + o (parity q, output o), rows 8-15 zero
    static constexpr int FD1T_B = FD1T_W + 20 * 64;      // [16] (rows 4 q + o: bias[o]; rows 8-15: 0)
    static constexpr int TOTAL = FD1T_B + 16;
};

// ---- packed weights (floats), filled by the host packer; all matrices k-major: [k][outputs]
struct FPk {
    static constexpr int WINDOW = 0, WINDOW_I = 512, TW = 1024;           // twiddles float2[256]
    static constexpr int SE_W = 1536;                                     // sub-band encoder: 5 x [K_i][32], K = 4, 7, 11, 20, 40
    static constexpr int SE_B = SE_W + 82 * 32;                           // 5 x [32]
    static constexpr int FE0_W = SE_B + 160;                              // [(c*6 + k)][4]     c < 2
    static constexpr int FE0_B = FE0_W + 48;
    static constexpr int FE1_W = FE0_B + 4;                               // [(c*8 + k)][16]    c < 4
    static constexpr int FE1_B = FE1_W + 512;
    static constexpr int FE2_W = FE1_B + 16;                              // [(c*6 + k)][32]    c < 16
    static constexpr int FE2_B = FE2_W + 3072;
    static constexpr int POST_W = FE2_B + 32;                             // [c][32]
    static constexpr int MG1_W = POST_W + 1024;                           // feature_merge.0: [i < 64][j < 32]
    static constexpr int MG2_W = MG1_W + 2048;                            // feature_merge.2: [ch < 32][c < 16]
    static constexpr int MG2_B = MG2_W + 512;
    static constexpr int DPE = MG2_B + 16;
    // per DPE block:
    static constexpr int D_IH = 0;                                        // intra W_ih^T [d][k][48]
    static constexpr int D_GB = D_IH + 2 * 768;                           // [d][48]: b_ih + (b_hh for r, z | 0 for n)
    static constexpr int D_HH = D_GB + 96;                                // intra W_hh: [d][gate][k][16 units]
    static constexpr int D_HN = D_HH + 2 * 768;                           // [d][16]: b_hh of the n gate
    static constexpr int D_FC_W = D_HN + 32;                              // intra_fc^T [k < 32][c]
    static constexpr int D_FC_B = D_FC_W + 512;
    static constexpr int D_LN_W = D_FC_B + 16;                            // [f][c]
    static constexpr int D_LN_B = D_LN_W + 512;
    static constexpr int D_G = D_LN_B + 512;                              // 8 groups x { ih [k][48], hh [k][48], gb [48], hn [16], fc^T [k][16], fcb [16] }
    static constexpr int G_IH = 0, G_HH = 768, G_GB = 1536, G_HN = 1584, G_FC_W = 1600, G_FC_B = 1856, G_SIZE = 1872;
    static constexpr int D_SIZE = D_G + 8 * G_SIZE;
    static constexpr int SP1_W = DPE + 3 * D_SIZE;                        // feature_split.0^T [c < 16][ch < 32]
    static constexpr int SP1_B = SP1_W + 512;
    static constexpr int SP2_W = SP1_B + 32;                              // feature_split.1^T [f < 32][j < 64]
    static constexpr int SD_W = SP2_W + 2048;                             // sub-band decoder, one column per bin: [k / 4 < 16][260][4] (16-byte loads: four k per load)
    static constexpr int SD_B = SD_W + 64 * 260;                          // [260]
    static constexpr int FD0_W = SD_B + 260;                              // decoder 1x1 ^T: [c < 64][32]
    static constexpr int FD0_T = FD0_W + 2048;                            // transposed conv [(c*6 + k)][16]   c < 32
    static constexpr int FD0_B = FD0_T + 3072;
    static constexpr int FD1_W = FD0_B + 16;                              // [c < 32][16]
    static constexpr int FD1_T = FD1_W + 512;                             // [(c*8 + k)][4]    c < 16
    static constexpr int FD1_B = FD1_T + 512;
    static constexpr int FD2_W = FD1_B + 4;                               // [c < 8][4]
    static constexpr int FD2_T = FD2_W + 32;                              // [(c*6 + k)][2]    c < 4
    static constexpr int FD2_B = FD2_T + 48;
    static constexpr int SB = (FD2_B + 2 + 3) / 4 * 4;                    // stream-batched DPE region (FSbPk)
    static constexpr int TOTAL = SB + FSbPk::TOTAL;
};

struct FArgs {
    const float* wp;
    const float* wav_in;
    float* wav_out;
    size_t in_stride, out_stride;
    float* cache_stft;
    float* cache_istft;
    float* gru;               // [24][B*4][16] inter-GRU states (block-major, then group): ONNXModel.initialize_cache order
    const float* spec_in;     // spec mode [B][257][T][2]
    float* spec_out;
    float* dbg;
    size_t dbg_stride;
    int B, T, mode, Tw;
    float compression;
    unsigned long long* clk;
    // time-pipelined offline launch (PIPE instantiation): pipe_p workgroups per utterance, workgroup p runs frames p, p + pipe_p, ...
    unsigned int* pipe_flags; // [B][num_blocks]: frames whose inter-GRU states of block k are in `gru`
    float* frames;            // [B][T][N] windowed output frames (summed / envelope-normalised by istft_ola_kernel)
    int pipe_p;
    // split step of large batches (PART 1 -> fspen_sb_dpe_kernel -> PART 2)
    float* tok;               // [B][2][1024] written by fspen_sb_dpe_kernel, read by PART 2: feature_split output, sub-band half [32][32] | fullband_decoder.1 output [4][128]
    float* carry;             // [B][FCarry::FLOATS] the front's LDS regions that the DPE kernel (cat) and the tail read: compressed spectrum, encoder outputs, sub-band / full-band features
};

// debug stages (fe_debug_step): name, rows, cols as dumped (row-major)
struct FDebugLayout {
    static constexpr int n_stages = 16;
    // 0 spec_in [257][2], 1 compressed [257][2], 2 subband_encoder [32 ch][32], 3 fullband_encoder.2 [32][32],
    // 4 feature_merge [16 c][32 f], 5+2b dpe.b.intra [32 f][16 c], 6+2b dpe.b.inter, 11 feature_split [32][64],
    // 12 fullband_decoder.0 [16][64], 13 fullband_decoder.1 [4][128], 14 mask [257][3] (full re, im, sub), 15 spec_out [257][2]
    __host__ __device__ static constexpr int rows(int s) {
        return (s <= 1 || s >= 14) ? 257 : (s == 2 || s == 3 || s == 11) ? 32 : s == 4 ? 16 : s == 12 ? 16 : s == 13 ? 4 : 32;
    }
    __host__ __device__ static constexpr int cols(int s) {
        return (s <= 1 || s == 15) ? 2 : s == 14 ? 3 : (s == 2 || s == 3 || s == 4) ? 32 : s == 11 ? 64 : s == 12 ? 64 : s == 13 ? 128 : 16;
    }
    __host__ __device__ static constexpr size_t offset(int s) {
        size_t o = 0;
        for (int i = 0; i < s; ++i) o += (size_t)rows(i) * cols(i);
        return o;
    }
    __host__ __device__ static constexpr size_t total() { return offset(n_stages); }
};

struct FLds {
    // live for the whole frame
    static constexpr int SP = 0;                   // compressed spectrum [257][2]
    static constexpr int TW = SP + 516;            // twiddles float2[256]
    static constexpr int E0 = TW + 512;            // fullband_encoder.0 out [4][134]  (3 zero columns either side: next conv's padding)
    static constexpr int E1 = E0 + 4 * 134;        // fullband_encoder.1 out [16][68]  (2 either side)
    static constexpr int E2 = E1 + 16 * 68;        // fullband_encoder.2 out [32][32]
    static constexpr int CAT = E2 + 1024;          // [32][64]: fullband_encoder_post | sub-band encoder
    static constexpr int SB = CAT + 2048;          // ---- phase scratch
    static constexpr int FA = SB, FB = SB + 1024;
    static constexpr int MAGP = SB + 2048;         // |X| [1 + 257 + 5] zero padded
    static constexpr int EIN = MAGP + 264;         // compressed re / im planes [2][264] (2 zero columns left, 5 right)
    static constexpr int M1 = SB;                  // feature_merge linear out [32][32]
    static constexpr int XA = SB + 1024, XB = SB + 1536;      // DPE tokens [32 f][16 c], ping-pong
    static constexpr int GI = SB + 2048;           // intra GRU input projections [2][32][48]
    static constexpr int HSEQ = SB + 5120;         // intra GRU outputs [32][32] (fwd | bwd)
    static constexpr int Y = SB + 6144;            // intra_fc out / inter scratch [32][16]
    static constexpr int HPREV = SB + 6656, HN = SB + 7168, RED = SB + 7680;
    static constexpr int S1 = SB + 2048;           // feature_split conv out [32][32]
    static constexpr int S2 = SB + 3072;           // feature_split out [32][64]
    static constexpr int MSUB = SB + 5120;         // sub-band mask [260]
    static constexpr int T2 = SB, D2 = SB + 1024;  // decoder 0: 1x1 out [32][32], transposed conv out [16][64]
    static constexpr int T1 = SB + 5632, D1 = SB + 6656, T0 = SB + 7168;
    static constexpr int MF = SB + 2048;           // full-band mask [2][258]
    static constexpr int TOTAL = SB + 7696;
    static constexpr int FRONT_TOTAL = EIN + 2 * 264;      // what PART 1 (STFT .. feature merge) touches: 34 KB, four workgroups per CU
    static_assert(SB % 2 == 0 && TW % 2 == 0, "float2 alignment");
    static_assert((size_t)TOTAL * 4 <= 64 * 1024, "static LDS");
};

// the front's LDS regions that the tail reads: [SP, TW) and [E0, SB)
struct FCarry {
    static constexpr int A0 = FLds::SP, AN = FLds::TW - FLds::SP, B0 = FLds::E0, BN = FLds::SB - FLds::E0;
    static constexpr int FLOATS = (AN + BN + 3) / 4 * 4;
    static_assert(AN % 4 == 0 && BN % 4 == 0 && A0 % 4 == 0 && B0 % 4 == 0 && FLds::E1 % 4 == 0 && FLds::E2 % 4 == 0 && FLds::CAT % 4 == 0, "16-byte copies");
};

// Read-only view of the packed weights through ONE buffer resource with 32-bit indices.  With plain pointers every far constant
// offset became its own 64-bit `base + constant` SGPR pair, hoisted to the kernel entry and spilled to VGPR lanes by the dozen; a
// buffer load takes the constant in its immediate / scalar-offset field instead.
struct WView {
    __amdgpu_buffer_rsrc_t rsrc;
    int base;                  // floats (goes into the load's vector offset together with the index: a block offset in the SCALAR offset
                               // made every vector offset loop-invariant, hoisted and kept live - measured slower)
    __device__ __forceinline__ float operator[](int i) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (base + i) * 4, 0, 0));
    }
    __device__ __forceinline__ f32x4 ld4(int i) const {         // 16 bytes at float index i (a multiple of 4)
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (base + i) * 4, 0, 0));
    }
    __device__ __forceinline__ WView operator+(int off) const { return WView{rsrc, base + off}; }
};

__device__ __forceinline__ float elu_f(float x) { return x > 0.0f ? x : __expf(x) - 1.0f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Software-pipelined dot products of the big-K phases: the weights of a thread's outputs arrive in NCH chunks of CH floats; chunk
// 0 is fetched by the caller BEFORE the barrier that ends the previous phase (wa), chunk i + 1 streams into the other register
// buffer while chunk i multiplies.  wf(ch, j): weight j of chunk ch; xf(ch, j, q): the input it multiplies for output q (ch is
// a run-time value, j / q compile-time: per-FMA LDS offsets are immediates).  2 CH + Q registers instead of K.
template <int NCH, int CH, int Q, class WF, class XF>
__device__ __forceinline__ void fs_pipe(float (&acc)[Q], float (&wa)[CH], WF&& wf, XF&& xf) {
    float wb[CH];
#pragma unroll 1
    for (int ch = 0; ch < NCH; ch += 2) {
        if (ch + 1 < NCH) {
#pragma unroll
            for (int j = 0; j < CH; ++j) wb[j] = wf(ch + 1, j);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[q] = fmaf(wa[j], xf(ch, j, q), acc[q]);
        if (ch + 1 < NCH) {
            if (ch + 2 < NCH) {
#pragma unroll
                for (int j = 0; j < CH; ++j) wa[j] = wf(ch + 2, j);
            }
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] = fmaf(wb[j], xf(ch + 1, j, q), acc[q]);
        }
    }
}

// The values of rows 0, 1, 2 of a wave (16 lanes each), per column, broadcast to all four rows: three gfx950 row-swap VALU ops
// (see rows_allreduce in fe_kernels.hip.h for the inline-asm form)
__device__ __forceinline__ void rows_gather3(float x, float& r0, float& r1, float& r2) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a = [x0 x1 x0 x1], b = [x2 x3 x2 x3]
    float c = a, d = a;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));      // c = [x0 x0 x0 x0], d = [x1 x1 x1 x1]
    float e = b, f = b;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(e), "+v"(f));      // e = [x2 x2 x2 x2]
    r0 = c; r1 = d; r2 = e;
}

// lane k of the caller's own 16-lane row, in every lane of that row (DPP row_newbcast: folds into the consuming VALU op)
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + K, 0xf, 0xf, true));
}

// acc + sum_k w[k] * (lane k of the caller's 16-lane row of h): K `v_fmac_f32_dpp` with the row broadcast on the multiplicand, in two
// chains.  Written out because hipcc does not fold row_bcast<k>() into the consuming fmac here: it emitted v_mov_b32_dpp + v_fmac_f32 +
// a hazard nop per term - 48 instructions for 16 products on the latency chain of a GRU step (a lone wave issues one every ~6 cycles).
// (`s_nop 1`: a DPP operand needs two wait states after the VALU write of its register - the recogniser does not look into the asm.)
#define FS_DF(acc, w, k) "v_fmac_f32_dpp %" #acc ", %2, %" #w " row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ float row_dot(const float (&w)[16], float h, float acc) {
    float a1 = 0.0f;
    asm("s_nop 1\n\t"
        FS_DF(0, 3, 0) FS_DF(1, 4, 1) FS_DF(0, 5, 2) FS_DF(1, 6, 3) FS_DF(0, 7, 4) FS_DF(1, 8, 5) FS_DF(0, 9, 6) FS_DF(1, 10, 7)
        FS_DF(0, 11, 8) FS_DF(1, 12, 9) FS_DF(0, 13, 10) FS_DF(1, 14, 11) FS_DF(0, 15, 12) FS_DF(1, 16, 13) FS_DF(0, 17, 14) FS_DF(1, 18, 15)
        : "+v"(acc), "+v"(a1)
        : "v"(h), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]),
          "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    return acc + a1;
}
__device__ __forceinline__ float row_dot(const float (&w)[12], float h, float acc) {
    float a1 = 0.0f;
    asm("s_nop 1\n\t"
        FS_DF(0, 3, 0) FS_DF(1, 4, 1) FS_DF(0, 5, 2) FS_DF(1, 6, 3) FS_DF(0, 7, 4) FS_DF(1, 8, 5) FS_DF(0, 9, 6) FS_DF(1, 10, 7)
        FS_DF(0, 11, 8) FS_DF(1, 12, 9) FS_DF(0, 13, 10) FS_DF(1, 14, 11)
        : "+v"(acc), "+v"(a1)
        : "v"(h), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]),
          "v"(w[11]));
    return acc + a1;
}
#undef FS_DF
#ifndef FS_WPE
#define FS_WPE 3          // waves per SIMD = workgroups per CU of the register budget (168 VGPRs: what the 54 KB LDS plan allows too)
#endif

#define FS_CLK(i) do { if constexpr (PROF) { if (blockIdx.x == 0 && threadIdx.x == 0) a.clk[(i)] = __builtin_readcyclecounter(); } } while (0)

// PIPE: time pipelining of an offline launch (as for FastEnhancer / BSRNN): the only thing a frame needs from the previous one are the
// inter-GRU states (24 x [4][16] per stream), handed over per DPE block through `gru` - agent-scope stores, drained, a counter per
// (utterance, block); the consumer polls the counter and fetches the states right before the block's inter GRUs.
// PART: 0 the whole frame; 1 the front (STFT, sub-band encoder, fullband_encoder.0-1: its LDS regions to global memory); 2 the tail
// (sub-band decoder, fullband_decoder.2, masks, iSTFT) - the per-hop step of large batches runs 1 -> fspen_sb_dpe_kernel (fullband_encoder.2,
// fullband_encoder_post, feature merge, 3 x DPE, feature split, fullband_decoder.0-1 - batched over the streams, fspen_sb_kernels.hip.h) -> 2
#ifndef FS_WPE_FRONT
#define FS_WPE_FRONT 4
#endif
template <class S, bool PROF, bool DBG, bool PIPE = false, int PART = 0>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(PART == 1 ? FS_WPE_FRONT : FS_WPE, PART == 1 ? FS_WPE_FRONT : FS_WPE))) fspen_frame_kernel(FArgs a) {
    static_assert(!PIPE || (!PROF && !DBG), "the time-pipelined instantiation is the plain offline kernel");
    static_assert(PART == 0 || (!PROF && !DBG && !PIPE), "the split step is the plain per-hop kernel");
    __shared__ __attribute__((aligned(16))) float smem[PART == 1 ? FLds::FRONT_TOTAL : FLds::TOTAL];
    using L = FLds;
    using P = FPk;
    constexpr int N = S::NFFT, H = S::HOP, OVL = S::OVL, BINS = S::BINS;
    const int tid0 = threadIdx.x;
    const int tid = tid0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ wp0 = a.wp;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, FPk::TOTAL * 4, 0x00020000);
    const int mode = a.mode, aT = a.T;

    float* sp = smem + L::SP;
    float2* tw = reinterpret_cast<float2*>(smem + L::TW);
    float2* fa = reinterpret_cast<float2*>(smem + L::FA);
    float2* fb = reinterpret_cast<float2*>(smem + L::FB);
    float* e0 = smem + L::E0;
    float* e1 = smem + L::E1;
    float* e2 = smem + L::E2;
    float* cat = smem + L::CAT;
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = reinterpret_cast<const float2*>(wp0 + P::TW)[i];
    // the zero columns of the padded encoder outputs are written once (the convs only write the interior)
    for (int i = tid; i < 4 * 134; i += kThreads) e0[i] = 0.0f;
    for (int i = tid; i < 16 * 68; i += kThreads) e1[i] = 0.0f;
    __syncthreads();

    int b = PIPE ? (int)blockIdx.x / a.pipe_p : (int)blockIdx.x;
    const int t_first = PIPE ? (int)blockIdx.x - b * a.pipe_p : 0, t_step = PIPE ? a.pipe_p : 1;
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
    // (per-stream pointers are derived from the kernel arguments where they are used, not kept live through the frame)
    float* dbg = DBG ? a.dbg + (size_t)b * a.dbg_stride : nullptr;
    // dump(stage, f): element (r, c) of the stage = f(r, c)
    auto dump = [&](int stage, auto&& f) {
        if constexpr (DBG) {
            const int rows = FDebugLayout::rows(stage), cols = FDebugLayout::cols(stage), n = rows * cols;
            float* dst = dbg + FDebugLayout::offset(stage);
            // full trip counts and a clamped index: only the store is predicated (no long partially-executed loop bodies)
            for (int i0_ = 0; i0_ < n; i0_ += kThreads) {
                const int i = i0_ + tid, ic = i < n ? i : n - 1;
                const int r = ic / cols, c = ic - r * cols;
                const float v = f(r, c);
                if (i < n) dst[i] = v;
            }
        }
    };

#pragma unroll 1
    for (int t = t_first; t < aT; t += t_step) {
        // a loop-variant zero on the weight pointer: the weights sit at compile-time offsets, and hoisted out of the frame /
        // stream loops their loads would stay live for the whole kernel (512 VGPRs, 229 spilled SGPRs without it)
        // (the same for the thread index: ~250 hoisted LDS addresses otherwise)
        int lz = 0, lzv = 0;
        asm volatile("" : "+s"(lz));
        asm volatile("" : "+v"(lzv));
        const WView wp{wrsrc, lz};
        const int tid = tid0 + lzv;
        const int lane = tid & 63;
        FS_CLK(0);
        float* x = smem + L::XA;          // tokens [f][c]
        float* xn = smem + L::XB;
        if constexpr (PART != 2) {
        // ============================ STFT + compress (models/fspen/model.py:409-417) ============================
        float* magp = smem + L::MAGP;
        float* ein = smem + L::EIN;
        if (mode != FE_MODE_SPEC) {
            const WView win = wp + P::WINDOW;
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
            for (int f = tid; f < 264; f += kThreads) {
                float re = 0.0f, im = 0.0f;
                if (f < BINS) {
                    re = Xf[f].x; im = Xf[f].y;
                    const float g = pow_f(fmaxf(sqrtf(re * re + im * im), 1.0e-5f), a.compression - 1.0f);
                    re *= g; im *= g;
                    sp[2 * f] = re; sp[2 * f + 1] = im;
                }
                // planes with 2 zero columns on the left (conv padding), |X| with 1
                if (f + 2 < 264) { ein[f + 2] = re; ein[264 + f + 2] = im; }
                if (f + 1 < 264) magp[f + 1] = sqrtf(re * re + im * im);
            }
            if (tid < 2) { ein[tid] = 0.0f; ein[264 + tid] = 0.0f; }
            if (tid == 0) magp[0] = 0.0f;
        } else {
            const float* si = a.spec_in + (size_t)b * BINS * aT * 2;
            for (int f = tid; f < 264; f += kThreads) {
                float re = 0.0f, im = 0.0f;
                if (f < BINS) {
                    re = si[((size_t)f * aT + t) * 2]; im = si[((size_t)f * aT + t) * 2 + 1];
                    if constexpr (DBG) { dbg[2 * f] = re; dbg[2 * f + 1] = im; }
                    const float g = pow_f(fmaxf(sqrtf(re * re + im * im), 1.0e-5f), a.compression - 1.0f);
                    re *= g; im *= g;
                    sp[2 * f] = re; sp[2 * f + 1] = im;
                }
                if (f + 2 < 264) { ein[f + 2] = re; ein[264 + f + 2] = im; }
                if (f + 1 < 264) magp[f + 1] = sqrtf(re * re + im * im);
            }
            if (tid < 2) { ein[tid] = 0.0f; ein[264 + tid] = 0.0f; }
            if (tid == 0) magp[0] = 0.0f;
        }
        // Weights sit in global memory (316 KB, L2-resident) at compile-time offsets, k-major.  Every phase FETCHES ALL ITS WEIGHTS
        // IN ONE BURST BEFORE THE BARRIER THAT ENDS THE PREVIOUS PHASE (FS_LDW): one memory round trip per phase, overlapped with
        // the barrier, instead of one per k-step (a lone wave per SIMD has nothing else to hide an L2 access behind).
#define FS_LDW(arr, K, base, stride)                                                                \
        float arr[K];                                                                               \
        _Pragma("unroll") for (int k_ = 0; k_ < (K); ++k_) arr[k_] = wp[(base) + k_ * (stride)]
        // sub-band encoder: thread (ch = tid & 31, position group tid >> 5); its four positions (one per q) belong to segments
        // with kernels <= 4, 11, 20, 40: fixed-size bursts, taps past the segment's kernel masked
        auto se_geom = [&](int q, int& K, int& base, int& wrow, int& seg) {
            const int pos = (tid >> 5) + 8 * q;
            int j;
            if (pos < 8) { seg = 0; j = pos; } else { seg = 1 + (pos - 8) / 6; j = (pos - 8) - (seg - 1) * 6; }
            K = seg == 0 ? 4 : seg == 1 ? 7 : seg == 2 ? 11 : seg == 3 ? 20 : 40;
            const int st = seg == 0 ? 2 : seg == 1 ? 3 : seg == 2 ? 5 : seg == 3 ? 10 : 20;
            base = (seg == 0 ? 0 : seg == 1 ? 14 : seg == 2 ? 31 : seg == 3 ? 62 : 123) + st * j;
            wrow = seg == 0 ? 0 : seg == 1 ? 4 : seg == 2 ? 11 : seg == 3 ? 22 : 42;
        };
        float se_w0[4], se_w1[11], se_w2[20], se_w3[40], se_b[4];
        {
            const int ch = tid & 31;
            int K, base, wrow, seg;
            se_geom(0, K, base, wrow, seg); se_b[0] = wp[P::SE_B + seg * 32 + ch];
#pragma unroll
            for (int k = 0; k < 4; ++k) se_w0[k] = wp[P::SE_W + (wrow + (k < K ? k : 0)) * 32 + ch];
            se_geom(1, K, base, wrow, seg); se_b[1] = wp[P::SE_B + seg * 32 + ch];
#pragma unroll
            for (int k = 0; k < 11; ++k) se_w1[k] = wp[P::SE_W + (wrow + (k < K ? k : 0)) * 32 + ch];
            se_geom(2, K, base, wrow, seg); se_b[2] = wp[P::SE_B + seg * 32 + ch];
#pragma unroll
            for (int k = 0; k < 20; ++k) se_w2[k] = wp[P::SE_W + (wrow + (k < K ? k : 0)) * 32 + ch];
            se_geom(3, K, base, wrow, seg); se_b[3] = wp[P::SE_B + seg * 32 + ch];
#pragma unroll
            for (int k = 0; k < 40; ++k) se_w3[k] = wp[P::SE_W + (wrow + (k < K ? k : 0)) * 32 + ch];
        }
        FS_LDW(fe0_w, 12, P::FE0_W + (tid & 3), 4);
        const float fe0_b = wp[P::FE0_B + (tid & 3)];
        __syncthreads();
        dump(1, [&](int r, int c) { return sp[2 * r + c]; });

        FS_CLK(1);
        // ============================ sub-band encoder (SubbandEncoder.forward, :58-66) + full-band conv 0 ============================
        {
            const int ch = tid & 31;
            auto run4 = [&]() {
                int K, base, wrow, seg;
                {   se_geom(0, K, base, wrow, seg); float acc = se_b[0];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc = fmaf(k < K ? se_w0[k] : 0.0f, magp[base + (k < K ? k : 0)], acc);
                    cat[ch * 64 + 32 + (tid >> 5)] = fmaxf(acc, 0.0f); }
                {   se_geom(1, K, base, wrow, seg); float acc = se_b[1];
#pragma unroll
                    for (int k = 0; k < 11; ++k) acc = fmaf(k < K ? se_w1[k] : 0.0f, magp[base + (k < K ? k : 0)], acc);
                    cat[ch * 64 + 32 + (tid >> 5) + 8] = fmaxf(acc, 0.0f); }
                {   se_geom(2, K, base, wrow, seg); float acc = se_b[2];
#pragma unroll
                    for (int k = 0; k < 20; ++k) acc = fmaf(k < K ? se_w2[k] : 0.0f, magp[base + (k < K ? k : 0)], acc);
                    cat[ch * 64 + 32 + (tid >> 5) + 16] = fmaxf(acc, 0.0f); }
                {   se_geom(3, K, base, wrow, seg); float acc = se_b[3];
#pragma unroll
                    for (int k = 0; k < 40; ++k) acc = fmaf(k < K ? se_w3[k] : 0.0f, magp[base + (k < K ? k : 0)], acc);
                    cat[ch * 64 + 32 + (tid >> 5) + 24] = fmaxf(acc, 0.0f); }
            };
            run4();
        }
        {   // fullband_encoder.0: Conv1d(2 -> 4, k 6, s 2, p 2) + folded BN + ELU -> e0[o][3 + j], j < 128
            const int o = tid & 3;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = (tid >> 2) + 64 * q;
                float acc = fe0_b;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int k = 0; k < 6; ++k) acc = fmaf(fe0_w[c * 6 + k], ein[c * 264 + 2 * j + k], acc);
                e0[o * 134 + 3 + j] = elu_f(acc);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        FS_LDW(fe1_w, 32, P::FE1_W + (tid & 15), 16);
        const float fe1_b = wp[P::FE1_B + (tid & 15)];
        __syncthreads();
        dump(2, [&](int r, int c) { return cat[r * 64 + 32 + c]; });
        {   // fullband_encoder.1: Conv1d(4 -> 16, k 8, s 2, p 3) -> e1[o][2 + j], j < 64
            const int o = tid & 15;
            float acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fe1_b;
#pragma unroll
            for (int ck = 0; ck < 32; ++ck) {
                const int c = ck >> 3, k = ck & 7;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(fe1_w[ck], e0[c * 134 + 2 * ((tid >> 4) + 16 * q) + k], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) e1[o * 68 + 2 + (tid >> 4) + 16 * q] = elu_f(acc[q]);
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        FS_LDW(fe2_w, 12, P::FE2_W + (tid & 31), 32);          // chunk 0 of 8 (2 channels x 6 taps each)
        const float fe2_b = wp[P::FE2_B + (tid & 31)];
        __syncthreads();
        if constexpr (PART != 1)
        {   // fullband_encoder.2: Conv1d(16 -> 32, k 6, s 2, p 2) -> e2[o][j], j < 32
            const int o = tid & 31;
            float acc[4] = {fe2_b, fe2_b, fe2_b, fe2_b};
            const WView wcol = wp + P::FE2_W + o;
            const float* xin = e1 + 2 * (tid >> 5);
            fs_pipe<8, 12, 4>(acc, fe2_w, [&](int ch, int j) { return wcol[(ch * 12 + j) * 32]; },
                              [&](int ch, int j, int q) { return xin[ch * 136 + (j / 6) * 68 + 16 * q + (j % 6)]; });
#pragma unroll
            for (int q = 0; q < 4; ++q) e2[o * 32 + (tid >> 5) + 8 * q] = elu_f(acc[q]);
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        FS_LDW(post_w, 32, P::POST_W + (tid & 31), 32);
        __syncthreads();
        dump(3, [&](int r, int c) { return e2[r * 32 + c]; });
        if constexpr (PART != 1)
        {   // fullband_encoder_post: 1x1 (32 -> 32), no bias -> cat[o][f], f < 32
            const int o = tid & 31;
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 32; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(post_w[c], e2[c * 32 + (tid >> 5) + 8 * q], acc[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) cat[o * 64 + (tid >> 5) + 8 * q] = acc[q];
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        FS_LDW(mg1_w, 16, P::MG1_W + (tid & 31), 32);          // chunk 0 of 4
        __syncthreads();

        FS_CLK(2);
        // ============================ feature merge (:246-250): Linear(64 -> 32) over the band axis, ELU, 1x1 (32 -> 16) ============================
        float* m1 = smem + L::M1;
        if constexpr (PART != 1) {
        {
            const int j = tid & 31;
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const WView wcol = wp + P::MG1_W + j;
            const float* xin = cat + (tid >> 5) * 64;
            fs_pipe<4, 16, 4>(acc, mg1_w, [&](int ch, int jj) { return wcol[(ch * 16 + jj) * 32]; },
                              [&](int ch, int jj, int q) { return xin[q * 512 + ch * 16 + jj]; });
#pragma unroll
            for (int q = 0; q < 4; ++q) m1[((tid >> 5) + 8 * q) * 32 + j] = elu_f(acc[q]);
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        FS_LDW(mg2_w, 32, P::MG2_W + (tid & 15), 16);
        const float mg2_b = wp[P::MG2_B + (tid & 15)];
        __syncthreads();
        {
            const int c = tid & 15;
            float acc[2] = {mg2_b, mg2_b};
#pragma unroll
            for (int ch = 0; ch < 32; ++ch)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[q] = fmaf(mg2_w[ch], m1[ch * 32 + (tid >> 4) + 16 * q], acc[q]);
#pragma unroll
            for (int q = 0; q < 2; ++q) x[((tid >> 4) + 16 * q) * 16 + c] = acc[q];
        }
        __syncthreads();
        dump(4, [&](int r, int c) { return x[c * 16 + r]; });
        }
        if constexpr (PART == 1) {
            // what the later kernels read, at its LDS offsets: the compressed spectrum and enc_out[0] (tail), enc_out[1] and the sub-band half of
            // cat (middle kernel; the tail's sub-band decoder reads that half too) - 12.7 KB per stream (the whole region: 20.8 KB)
            f32x4* cr = reinterpret_cast<f32x4*>(a.carry + (size_t)b * FCarry::FLOATS);
            for (int i = tid; i < FCarry::AN / 4; i += kThreads) cr[i] = reinterpret_cast<const f32x4*>(smem + FCarry::A0)[i];
            for (int i = tid; i < (L::E2 - L::E0) / 4; i += kThreads) cr[FCarry::AN / 4 + i] = reinterpret_cast<const f32x4*>(smem + FCarry::B0)[i];
            for (int i = tid; i < 256; i += kThreads) {
                const int o = (L::CAT - L::E0) / 4 + (i >> 3) * 16 + 8 + (i & 7);
                cr[FCarry::AN / 4 + o] = reinterpret_cast<const f32x4*>(smem + FCarry::B0)[o];
            }
            __syncthreads();
            continue;
        }
        } else {
            const f32x4* tk = reinterpret_cast<const f32x4*>(a.tok + (size_t)b * 2048);
            const f32x4* cr = reinterpret_cast<const f32x4*>(a.carry + (size_t)b * FCarry::FLOATS);
            // feature_split output, sub-band half -> s2[ch][32 ..]; fullband_decoder.1's output -> d1 [4][128]
            for (int i = tid; i < 256; i += kThreads) reinterpret_cast<f32x4*>(smem + L::S2 + (i >> 3) * 64 + 32)[i & 7] = tk[i];
            for (int i = tid; i < 128; i += kThreads) reinterpret_cast<f32x4*>(smem + L::D1)[i] = tk[256 + i];
            // of the front's regions the tail reads the compressed spectrum, enc_out[0] and the sub-band half of cat
            for (int i = tid; i < FCarry::AN / 4; i += kThreads) reinterpret_cast<f32x4*>(smem + FCarry::A0)[i] = cr[i];
            for (int i = tid; i < (L::E1 - L::E0) / 4; i += kThreads) reinterpret_cast<f32x4*>(smem + FCarry::B0)[i] = cr[FCarry::AN / 4 + i];
            for (int i = tid; i < 256; i += kThreads) {
                const int o = (L::CAT - L::E0) / 4 + (i >> 3) * 16 + 8 + (i & 7);
                reinterpret_cast<f32x4*>(smem + FCarry::B0)[o] = cr[FCarry::AN / 4 + o];
            }
            __syncthreads();
        }

        FS_CLK(3);
        if constexpr (PART == 0) {
        // ============================ 3 x DPE (DPE.forward, :172-189) ============================
        float* gi = smem + L::GI;
        float* hseq = smem + L::HSEQ;
        float* hprev = smem + L::HPREV;
        float* hn = smem + L::HN;
        float* red = smem + L::RED;
#pragma unroll 1
        for (int blk = 0; blk < S::NB; ++blk) {
            const WView wd = wp + (P::DPE + blk * P::D_SIZE);
            // inter-GRU states of this block: [g][B*4][16] -> hprev[f][16]   (in flight across the intra GRU)
            float hp[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int f = (tid >> 4) + 16 * q, g = f >> 2;
                hp[q] = PIPE ? 0.0f : a.gru[(((size_t)(blk * S::G + g) * a.B + b) * S::FG + (f & 3)) * S::C + (tid & 15)];
            }
            // ---- intra GRU input projections: gi[d][f][g48] = x[f] . W_ih^T + bias
            if (tid < 192) {
                const int g48 = tid % 48, fq = tid / 48;
                float w[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) w[k] = wd[P::D_IH + k * 48 + g48];
                const float bias0 = wd[P::D_GB + g48], bias1 = wd[P::D_GB + 48 + g48];
                const float gsc = g48 < 32 ? kGateRZ : kGateN;        // (the recurrence works on scaled pre-activations: row_dot's note)
#pragma unroll 2
                for (int r = 0; r < 8; ++r) {
                    const int f = fq + 4 * r;
                    float a0 = bias0, a1 = bias1;
#pragma unroll
                    for (int k = 0; k < 16; ++k) { const float xv = x[f * 16 + k]; a0 = fmaf(w[k], xv, a0); a1 = fmaf(w[16 + k], xv, a1); }
                    gi[f * 48 + g48] = a0 * gsc;
                    gi[(32 + f) * 48 + g48] = a1 * gsc;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) hprev[((tid >> 4) + 16 * q) * 16 + (tid & 15)] = hp[q];
            __builtin_amdgcn_sched_barrier(0);
            // recurrence weights of this wave's direction (waves 0, 1) and everybody's intra_fc / LayerNorm weights: fetched before the barrier
            // (fetched by all four waves - waves 2 / 3 get a copy of direction 0 / 1: a conditional definition would be carried
            //  around the block loop as 49 live registers)
            // lane = (gate row = lane / 16: r, z, n, (r again), hidden unit c = lane % 16): ONE gate row of 16 weights per lane
            float wg_[16];
            const int g_row = (lane >> 4) < 3 ? (lane >> 4) : 0;
            {
                const int c = lane & 15, dsel = wave & 1;
#pragma unroll
                for (int k = 0; k < 16; ++k) wg_[k] = wd[P::D_HH + ((dsel * 3 + g_row) * 16 + k) * 16 + c] * (g_row == 2 ? kGateN : kGateRZ);
            }
            const float bhn = (lane >> 4) == 2 ? wd[P::D_HN + (wave & 1) * 16 + (lane & 15)] * kGateN : 0.0f;
            FS_LDW(fc_w, 32, (P::DPE + blk * P::D_SIZE) + P::D_FC_W + (tid & 15), 16);
            const float fc_b = wd[P::D_FC_B + (tid & 15)];
            float ln_w[2], ln_b[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = ((tid >> 4) + 16 * q) * 16 + (tid & 15);
                ln_w[q] = wd[P::D_LN_W + i]; ln_b[q] = wd[P::D_LN_B + i];
            }
            __syncthreads();
            if (blk == 0) FS_CLK(8);
            // ---- the recurrence: wave d walks direction d.  The 48 gate rows of a step sit one per lane (rows of 16 lanes = gates r, z,
            // n; the fourth row repeats r): 16 FMAs per lane with h broadcast inside each row by DPP (h is kept replicated in all rows),
            // then the three gate values of a unit are gathered to every row with three row-swap ops and every row updates its copy
            // of h.  (First version: lane c computed all three gates of unit c - 48 FMAs, 16 v_readlane + 17 hazard nops per step:
            // 620 cycles per step.)
            if (wave < 2) {
                const int d = wave, c = lane & 15;
                const bool is_n = (lane >> 4) == 2;
                float h = 0.0f;
                // running LDS offsets of the walk (sub-band 0 -> 31 forward, 31 -> 0 backward): one add each per step
                const int f0 = d ? 31 : 0;
                int go = (d * 32 + f0) * 48 + g_row * 16 + c, gn = (d * 32 + f0) * 48 + 32 + c, ho = f0 * 32 + d * 16 + c;
                const int gstep = d ? -48 : 48, hstep = d ? -32 : 32;
                float g_own = gi[go], g_n = gi[gn];
#pragma unroll 1
                for (int s_ = 0; s_ < 32; ++s_) {
                    go += gstep; gn += gstep;
                    // next step's x side (the read after the last step lands in the other direction's rows: unused)
                    const float n_own = gi[go], n_n = gi[gn];
                    const float acc = row_dot(wg_, h, bhn);
                    const float x_ = is_n ? acc : sigmoid_pre(g_own + acc);
                    float r, z, pn;
                    rows_gather3(x_, r, z, pn);
                    const float n = tanh_pre(__builtin_fmaf(r, pn, g_n));
                    h = __builtin_fmaf(z, h - n, n);                      // (1 - z) n + z h
                    hseq[ho] = h;                                        // (the four rows hold the same h: same value to the same address)
                    ho += hstep;
                    g_own = n_own; g_n = n_n;
                }
            }
            __syncthreads();
            if (blk == 0) FS_CLK(9);
            // ---- intra_fc + LayerNorm([F, C]) + residual
            float yv[2] = {fc_b, fc_b};
#pragma unroll
            for (int k = 0; k < 32; ++k)
#pragma unroll
                for (int q = 0; q < 2; ++q) yv[q] = fmaf(fc_w[k], hseq[((tid >> 4) + 16 * q) * 32 + k], yv[q]);
            __builtin_amdgcn_sched_barrier(0);
            // inter GRUs: thread (group g = tid / 16 < 8, hidden unit c) keeps the unit's three gate rows of its group's GRU (96 weights) in
            // registers - fetched here, in flight across the LayerNorm - and walks the group's 4 sub-band rows (threads 128.. idle)
            const int ig_ = (tid >> 4) & 7;
            const WView wgi = wd + P::D_G + ig_ * P::G_SIZE;
            float wi[48], wh[48];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                wi[k] = wgi[P::G_IH + k * 48 + (tid & 15)]; wi[16 + k] = wgi[P::G_IH + k * 48 + 16 + (tid & 15)]; wi[32 + k] = wgi[P::G_IH + k * 48 + 32 + (tid & 15)];
                wh[k] = wgi[P::G_HH + k * 48 + (tid & 15)]; wh[16 + k] = wgi[P::G_HH + k * 48 + 16 + (tid & 15)]; wh[32 + k] = wgi[P::G_HH + k * 48 + 32 + (tid & 15)];
            }
            const float gb_r = wgi[P::G_GB + (tid & 15)], gb_z = wgi[P::G_GB + 16 + (tid & 15)], gb_n = wgi[P::G_GB + 32 + (tid & 15)], gb_hn = wgi[P::G_HN + (tid & 15)];
            float s0 = wave_sum(yv[0] + yv[1]);
            if (lane == 0) red[wave] = s0;
            __syncthreads();
            const float mean = (red[0] + red[1] + red[2] + red[3]) * (1.0f / 512.0f);
            const float d0 = yv[0] - mean, d1 = yv[1] - mean;
            float s1 = wave_sum(d0 * d0 + d1 * d1);
            if (lane == 0) red[4 + wave] = s1;
            __syncthreads();
            const float inv_std = __builtin_amdgcn_rsqf((red[4] + red[5] + red[6] + red[7]) * (1.0f / 512.0f) + 1.0e-5f);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = ((tid >> 4) + 16 * q) * 16 + (tid & 15);
                xn[i] = (q ? d1 : d0) * inv_std * ln_w[q] + ln_b[q] + x[i];
            }
            __syncthreads();
            dump(5 + 2 * blk, [&](int r, int c) { return xn[r * 16 + c]; });
            if (blk == 0) FS_CLK(10);
            if constexpr (PIPE) {
                // frame t - 1's inter-GRU states of this block: wait for them, fetch them into hprev
                if (t > 0) {
                    if (tid == 0) {
                        int spins = 0;
                        while (__hip_atomic_load(a.pipe_flags + (size_t)b * S::NB + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)t && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
                        if (spins >= (1 << 24)) __builtin_trap();      // (a producer that never shows up: fail the launch loudly, never run on stale state)
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int f = (tid >> 4) + 16 * q, g = f >> 2;
                    hprev[f * 16 + (tid & 15)] = ld_state(a.gru + (((size_t)(blk * S::G + g) * a.B + b) * S::FG + (f & 3)) * S::C + (tid & 15));
                }
                __syncthreads();
            }
            // ---- inter path (InterRNNPathExtension.forward, :122-138): group g = f / 4 has its own GRU (one step per frame) and fc
            if (tid < 128) {
                const int c = tid & 15;
#pragma unroll 1
                for (int fl = 0; fl < S::FG; ++fl) {
                    const int f = ig_ * S::FG + fl;
                    float ir = gb_r, iz = gb_z, in_ = gb_n, hr = 0.0f, hz = 0.0f, hnn = gb_hn;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float xv = xn[f * 16 + k], hv = hprev[f * 16 + k];
                        ir = fmaf(wi[k], xv, ir);
                        iz = fmaf(wi[16 + k], xv, iz);
                        in_ = fmaf(wi[32 + k], xv, in_);
                        hr = fmaf(wh[k], hv, hr);
                        hz = fmaf(wh[16 + k], hv, hz);
                        hnn = fmaf(wh[32 + k], hv, hnn);
                    }
                    const float r = sigmoid_f(ir + hr);
                    const float z = sigmoid_f(iz + hz);
                    const float n = tanh_f(in_ + r * hnn);
                    const float hnew = (1.0f - z) * n + z * hprev[f * 16 + c];
                    hn[f * 16 + c] = hnew;
                    st_state(a.gru + (((size_t)(blk * S::G + ig_) * a.B + b) * S::FG + fl) * S::C + c, hnew);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            float ifc_w[2][16], ifc_b[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const WView wg = wd + P::D_G + (((tid >> 4) + 16 * q) >> 2) * P::G_SIZE;
#pragma unroll
                for (int k = 0; k < 16; ++k) ifc_w[q][k] = wg[P::G_FC_W + k * 16 + (tid & 15)];
                ifc_b[q] = wg[P::G_FC_B + (tid & 15)];
            }
            if constexpr (PIPE) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this thread's state stores have left the CU
            __syncthreads();
            if constexpr (PIPE) {
                if (tid == 0) __hip_atomic_store(a.pipe_flags + (size_t)b * S::NB + blk, (unsigned int)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int f = (tid >> 4) + 16 * q, c = tid & 15;
                float acc = ifc_b[q];
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = fmaf(ifc_w[q][k], hn[f * 16 + k], acc);
                x[f * 16 + c] = acc + 2.0f * xn[f * 16 + c];          // + x_in inside the path extension, + x_in again in DPE.forward
            }
            __syncthreads();
            dump(6 + 2 * blk, [&](int r, int c) { return x[r * 16 + c]; });
            if (blk == 0) FS_CLK(11);
        }
        }

        FS_CLK(4);
        // ============================ feature split (:256-260): 1x1 (16 -> 32), Linear(32 -> 64) over the band axis, ELU ============================
        float* s1 = smem + L::S1;
        float* s2 = smem + L::S2;
        if constexpr (PART != 2) {
        {
            const int ch = tid & 31;
            FS_LDW(w, 16, P::SP1_W + ch, 32);
            float acc[4];
            const float bias = wp[P::SP1_B + ch];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = bias;
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(w[c], x[((tid >> 5) + 8 * q) * 16 + c], acc[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) s1[ch * 32 + (tid >> 5) + 8 * q] = acc[q];
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        FS_LDW(sp2_w, 32, P::SP2_W + (tid & 63), 64);
        __syncthreads();
        {
            const int j = tid & 63;
            float acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = 0.0f;
#pragma unroll
            for (int f = 0; f < 32; ++f)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(sp2_w[f], s1[((tid >> 6) + 4 * q) * 32 + f], acc[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) s2[((tid >> 6) + 4 * q) * 64 + j] = elu_f(acc[q]);
        }
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        // sub-band decoder column of bin = tid: 64 weights as sixteen 16-byte loads, the first eight before the barrier.  (The wave-level load
        // is the unit the CU's vector-memory path charges for - with three / four of these workgroups per CU that path, not the ALU, bounds the tail.)
        f32x4 sd_w[8];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) sd_w[k4] = wp.ld4(P::SD_W + (k4 * 260 + tid) * 4);
        const float sd_b = wp[P::SD_B + tid];
        __syncthreads();
        dump(11, [&](int r, int c) { return s2[r * 64 + c]; });

        FS_CLK(5);
        // ============================ sub-band decoder (SubbandDecoder.forward, :83-95): one output per bin ============================
        float* msub = smem + L::MSUB;
        // bin -> (linear layer, flattened index) -> input row of the [32 + zero row][64] matrix
        auto sd_row = [](int bin) {
            if (bin < 16) return bin / 2;
            if (bin < 32) return 8 + (bin - 16 + 1) / 3;
            if (bin < 64) return 13 + (bin - 32 + 4) / 5;
            if (bin < 128) return 19 + (bin - 64 + 8) / 10;
            return 25 + (bin - 128 + 16) / 20;
        };
        {
            const int row = sd_row(tid);
            const bool zero_row = row >= 32;                      // lin5's F.pad row: ReLU(bias)
            const int rc = zero_row ? 0 : row;
            // k < 32: x_sub1 (cat[k][32 + row]); k >= 32: x_sub2 (s2[k - 32][32 + row]); both are [32][64] arrays
            f32x4 sd_w2[8];
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) sd_w2[k4] = wp.ld4(P::SD_W + ((8 + k4) * 260 + tid) * 4);
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc0 = fmaf(sd_w[k4][e], cat[(4 * k4 + e) * 64 + 32 + rc], acc0);
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1 = fmaf(sd_w2[k4][e], s2[(4 * k4 + e) * 64 + 32 + rc], acc1);
            msub[tid] = fmaxf(sd_b + (zero_row ? 0.0f : acc0 + acc1), 0.0f);
            if (tid == 0) msub[256] = fmaxf(wp[P::SD_B + 256], 0.0f);          // bin 256 reads the zero row too
        }
        __builtin_amdgcn_sched_barrier(0);
        // ============================ full-band decoder (:262-277, :397-400) ============================
        float* t2 = smem + L::T2;
        if constexpr (PART != 2)
        {   // decoder 0: 1x1 over cat(x_full, enc_out[2]) (64 -> 32): chunks 0, 1 from s2[c][f] (row stride 64), 2, 3 from e2[c][f] (32)
            const int o = tid & 31;
            FS_LDW(w0, 16, P::FD0_W + o, 32);
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const WView wcol = wp + P::FD0_W + o;
            const int f0 = tid >> 5;
            fs_pipe<4, 16, 4>(acc, w0, [&](int ch, int j) { return wcol[(ch * 16 + j) * 32]; },
                              [&](int ch, int j, int q) {
                                  return ch < 2 ? s2[(ch * 16 + j) * 64 + f0 + 8 * q] : e2[((ch - 2) * 16 + j) * 32 + f0 + 8 * q];
                              });
#pragma unroll
            for (int q = 0; q < 4; ++q) t2[o * 32 + f0 + 8 * q] = acc[q];
        }
        __builtin_amdgcn_sched_barrier(0);
        // transposed convs: output p takes the taps k = k0, k0 + 2, ... (k0 = parity of p + padding), input f = (p + pad - k) / 2; all
        // of a thread's outputs share the parity, hence the taps' weights; out-of-range inputs are clamped and multiplied by 0
        float* d2 = smem + L::D2;
        if constexpr (PART != 2) {
        float fd0_t[12];                                         // chunk 0 of 8: 4 channels x 3 taps
        {
            const int k0 = (tid >> 4) & 1;
#pragma unroll
            for (int j = 0; j < 12; ++j) fd0_t[j] = wp[P::FD0_T + ((j / 3) * 6 + k0 + 2 * (j % 3)) * 16 + (tid & 15)];
        }
        const float fd0_b = wp[P::FD0_B + (tid & 15)];
        __syncthreads();
        {   // ConvTranspose1d(32 -> 16, k 6, s 2, p 2) + folded BN + ELU: y[o][p] += x[c][f] w[c][o][k], p = 2 f + k - 2
            const int o = tid & 15, k0 = (tid >> 4) & 1;
            // per (output q, tap i): clamped input column and its 0 / 1 mask
            int fcol[4][3];
            float fm[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int f = ((tid >> 4) + 16 * q + 2 - k0 - 2 * i) >> 1;
                    const bool ok = f >= 0 && f < 32;
                    fcol[q][i] = ok ? f : 0;
                    fm[q][i] = ok ? 1.0f : 0.0f;
                }
            float acc3[12];                                       // [q][tap] partial sums (masked at the end)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc3[i] = 0.0f;
            const WView wcol = wp + P::FD0_T + o;
            // Q = 12 "outputs" (q, tap): weight j = (channel j / 3 of the chunk, tap j % 3) only feeds the outputs of its own tap
            float wb[12];
#pragma unroll 1
            for (int ch = 0; ch < 8; ch += 2) {
#pragma unroll
                for (int j = 0; j < 12; ++j) wb[j] = wcol[(((ch + 1) * 4 + j / 3) * 6 + k0 + 2 * (j % 3)) * 16];
#pragma unroll
                for (int j = 0; j < 12; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc3[q * 3 + j % 3] = fmaf(fd0_t[j], t2[(ch * 4 + j / 3) * 32 + fcol[q][j % 3]], acc3[q * 3 + j % 3]);
                if (ch + 2 < 8) {
#pragma unroll
                    for (int j = 0; j < 12; ++j) fd0_t[j] = wcol[(((ch + 2) * 4 + j / 3) * 6 + k0 + 2 * (j % 3)) * 16];
                }
#pragma unroll
                for (int j = 0; j < 12; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc3[q * 3 + j % 3] = fmaf(wb[j], t2[((ch + 1) * 4 + j / 3) * 32 + fcol[q][j % 3]], acc3[q * 3 + j % 3]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float acc = fd0_b;
#pragma unroll
                for (int i = 0; i < 3; ++i) acc = fmaf(fm[q][i], acc3[q * 3 + i], acc);
                d2[o * 64 + (tid >> 4) + 16 * q] = elu_f(acc);
            }
        }
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        FS_LDW(fd1_w, 32, P::FD1_W + (tid & 15), 16);
        __syncthreads();
        dump(12, [&](int r, int c) { return d2[r * 64 + c]; });
        float* t1 = smem + L::T1;
        float* d1 = smem + L::D1;
        if constexpr (PART != 2) {
        {   // decoder 1: 1x1 over cat(d2, enc_out[1]) (32 -> 16)
            const int o = tid & 15;
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = (tid >> 4) + 16 * q;
                    acc[q] = fmaf(fd1_w[c], d2[c * 64 + f], acc[q]);
                    acc[q] = fmaf(fd1_w[16 + c], e1[c * 68 + 2 + f], acc[q]);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) t1[o * 64 + (tid >> 4) + 16 * q] = acc[q];
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        float fd1_t[16];                                         // chunk 0 of 4: 4 channels x 4 taps
        {
            const int k0 = ((tid >> 2) + 1) & 1;
#pragma unroll
            for (int j = 0; j < 16; ++j) fd1_t[j] = wp[P::FD1_T + ((j / 4) * 8 + k0 + 2 * (j % 4)) * 4 + (tid & 3)];
        }
        const float fd1_b = wp[P::FD1_B + (tid & 3)];
        __syncthreads();
        {   // ConvTranspose1d(16 -> 4, k 8, s 2, p 3) + folded BN + ELU: p = 2 f + k - 3
            const int o = tid & 3, k0 = ((tid >> 2) + 1) & 1;
            int fcol[2][4];
            float fm[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = ((tid >> 2) + 64 * q + 3 - k0 - 2 * i) >> 1;
                    const bool ok = f >= 0 && f < 64;
                    fcol[q][i] = ok ? f : 0;
                    fm[q][i] = ok ? 1.0f : 0.0f;
                }
            float acc4[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc4[i] = 0.0f;
            const WView wcol = wp + P::FD1_T + o;
            float wb[16];
#pragma unroll 1
            for (int ch = 0; ch < 4; ch += 2) {
#pragma unroll
                for (int j = 0; j < 16; ++j) wb[j] = wcol[(((ch + 1) * 4 + j / 4) * 8 + k0 + 2 * (j % 4)) * 4];
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc4[q * 4 + j % 4] = fmaf(fd1_t[j], t1[(ch * 4 + j / 4) * 64 + fcol[q][j % 4]], acc4[q * 4 + j % 4]);
                if (ch + 2 < 4) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) fd1_t[j] = wcol[(((ch + 2) * 4 + j / 4) * 8 + k0 + 2 * (j % 4)) * 4];
                }
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc4[q * 4 + j % 4] = fmaf(wb[j], t1[((ch + 1) * 4 + j / 4) * 64 + fcol[q][j % 4]], acc4[q * 4 + j % 4]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float acc = fd1_b;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = fmaf(fm[q][i], acc4[q * 4 + i], acc);
                d1[o * 128 + (tid >> 2) + 64 * q] = elu_f(acc);
            }
        }
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        FS_LDW(fd2_w, 8, P::FD2_W + (tid & 3), 4);
        __syncthreads();
        dump(13, [&](int r, int c) { return d1[r * 128 + c]; });
        float* t0 = smem + L::T0;
        {   // decoder 2: 1x1 over cat(d1, enc_out[0]) (8 -> 4)
            const int o = tid & 3;
            float acc[2] = {0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int f = (tid >> 2) + 64 * q;
                    acc[q] = fmaf(fd2_w[c], d1[c * 128 + f], acc[q]);
                    acc[q] = fmaf(fd2_w[4 + c], e0[c * 134 + 3 + f], acc[q]);
                }
#pragma unroll
            for (int q = 0; q < 2; ++q) t0[o * 128 + (tid >> 2) + 64 * q] = acc[q];
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next phase's weight burst from being hoisted above this phase's arithmetic)
        float fd2_t[12];
        {
            const int k0 = (tid >> 1) & 1;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 3; ++j) fd2_t[c * 3 + j] = wp[P::FD2_T + (c * 6 + k0 + 2 * j) * 2 + (tid & 1)];
        }
        const float fd2_b = wp[P::FD2_B + (tid & 1)];
        __syncthreads();
        float* mf = smem + L::MF;
        for (int i = tid; i < 2 * BINS; i += kThreads) {
            // ConvTranspose1d(4 -> 2, k 6, s 2, p 2, output_padding 1) + bias: p = 2 f + k - 2, p < 257   (i = tid + 256 q: same o, same parity)
            const int o = i & 1, p = i >> 1, k0 = p & 1;
            float acc = fd2_b;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int f = (p + 2 - k0 - 2 * j) >> 1;
                const bool ok = f >= 0 && f < 128;
                const int fc = ok ? f : 0;
                float part = 0.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) part = fmaf(fd2_t[c * 3 + j], t0[c * 128 + fc], part);
                acc += ok ? part : 0.0f;
            }
            mf[o * 258 + p] = acc;
        }
        __syncthreads();
        dump(14, [&](int r, int c) { return c < 2 ? mf[c * 258 + r] : msub[r]; });

        FS_CLK(6);
        // ============================ masks (:401-407), un-compress (:422-428), iSTFT ============================
        {
            float* spo = mode == FE_MODE_SPEC ? a.spec_out + (size_t)b * BINS * aT * 2
                                              : (mode == FE_MODE_OFFLINE ? a.spec_out + (size_t)b * BINS * aT * 2 : nullptr);
            for (int f = tid; f < BINS; f += kThreads) {
                const float sr = sp[2 * f], si = sp[2 * f + 1], mr = mf[f], mi = mf[258 + f];
                const float o_r = sr * mr - si * mi, o_i = sr * mi + si * mr;
                const float mfm = sqrtf(mr * mr + mi * mi);
                const float mm = (msub[f] + mfm) * 0.5f;
                float yr = o_r / mfm * mm, yi = o_i / mfm * mm;
                if (mode == FE_MODE_OFFLINE) {          // Model.forward returns the COMPRESSED spectrum (spec_out of model_forward)
                    spo[((size_t)f * aT + t) * 2] = yr;
                    spo[((size_t)f * aT + t) * 2 + 1] = yi;
                }
                const float g = pow_f(sqrtf(yr * yr + yi * yi), 1.0f / a.compression - 1.0f);
                yr *= g; yi *= g;
                if constexpr (DBG) { float* d = dbg + FDebugLayout::offset(15); d[2 * f] = yr; d[2 * f + 1] = yi; }
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
        if (mode != FE_MODE_SPEC) {
            float2* yv = fft_lds<S, true>(fa, fb, tw);
            float2* spare = (yv == fa) ? fb : fa;
            float* cis = a.cache_istft + (size_t)b * OVL;
            const WView wi = wp + (mode == FE_MODE_STREAM ? P::WINDOW_I : P::WINDOW);
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
                const WView w = wp + P::WINDOW;
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
        FS_CLK(7);
    }
    if constexpr (PIPE) break;
    b += gridDim.x;
    } while (b < a.B);
}

#undef FS_LDW

}  // namespace fe
#include "fspen_sb_kernels.hip.h"
namespace fe {

struct FImpl {
    int HOP;
    size_t lds_bytes;
    size_t dbg_floats;
    int dbg_stages;
    size_t packed_floats;
    void (*launch)(const FArgs&, int max_wgs, hipStream_t, hipError_t*);
    void (*dbg_stage)(int, int*, int*, size_t*);
    void (*launch_pipe)(const FArgs&, hipStream_t, hipError_t*);       // time-pipelined offline launch (cooperative: B * pipe_p workgroups)
    int occ;                  // workgroups per CU
    int num_blocks;
    // per-hop step of large batches: front -> DPE blocks batched over the streams (fspen_sb_kernels.hip.h) -> tail; a.tok / a.carry set
    void (*launch_sb)(const FArgs&, int max_wgs, hipStream_t, hipError_t*);
    size_t split_floats_per_stream;      // tok + carry
};

template <class S>
void flaunch_pipe_impl(const FArgs& a, hipStream_t st, hipError_t* err) {
    FArgs args = a;
    void* kargs[] = {&args};
    note_kernel("fspen_frame_kernel<time-pipelined>");
    *err = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&fspen_frame_kernel<S, false, false, true>), dim3(a.B * a.pipe_p), dim3(kThreads), kargs, 0, st);
}

template <class S>
void flaunch_impl(const FArgs& a, int max_wgs, hipStream_t st, hipError_t* err) {
    constexpr int OCC_LDS = (160 * 1024) / (FLds::TOTAL * 4);      // workgroups per CU: LDS- and register-limited
    constexpr int OCC = OCC_LDS < FS_WPE ? OCC_LDS : FS_WPE;
    const int slots = max_wgs * OCC;
    const int grid = a.B < slots ? a.B : slots;                    // more streams than slots: persistent workgroups walk b, b + grid, ...
    note_kernel(a.dbg != nullptr ? "fspen_frame_kernel<debug>" : a.clk != nullptr ? "fspen_frame_kernel<profile>" : "fspen_frame_kernel");
    if (a.dbg != nullptr) hipLaunchKernelGGL((fspen_frame_kernel<S, false, true>), dim3(grid), dim3(kThreads), 0, st, a);
    else if (a.clk != nullptr) hipLaunchKernelGGL((fspen_frame_kernel<S, true, false>), dim3(grid), dim3(kThreads), 0, st, a);
    else hipLaunchKernelGGL((fspen_frame_kernel<S, false, false>), dim3(grid), dim3(kThreads), 0, st, a);
    *err = hipGetLastError();
}

template <class S>
void flaunch_sb_impl(const FArgs& a, int max_wgs, hipStream_t st, hipError_t* err) {
    constexpr int OCC_LDS = (160 * 1024) / (FLds::TOTAL * 4);
    constexpr int OCC = OCC_LDS < FS_WPE ? OCC_LDS : FS_WPE;
    const int slots = max_wgs * OCC;
    const int grid = a.B < slots ? a.B : slots;
    constexpr int OCC_F_LDS = (160 * 1024) / (FLds::FRONT_TOTAL * 4);
    const int slots_f = max_wgs * (OCC_F_LDS < FS_WPE_FRONT ? OCC_F_LDS : FS_WPE_FRONT);
    note_kernel("fspen_frame_kernel<PART 1>");
    hipLaunchKernelGGL((fspen_frame_kernel<S, false, false, false, 1>), dim3(a.B < slots_f ? a.B : slots_f), dim3(kThreads), 0, st, a);
    FSbArgs sa{a.wp, a.carry, a.tok, a.carry + (size_t)a.B * FCarry::FLOATS, a.gru, a.B, a.clk};
    *err = fspen_sb_launch<S>(sa, st);
    if (*err != hipSuccess) return;
    note_kernel("fspen_frame_kernel<PART 2>");
    hipLaunchKernelGGL((fspen_frame_kernel<S, false, false, false, 2>), dim3(grid), dim3(kThreads), 0, st, a);
    *err = hipGetLastError();
}

inline void fdbg_stage_impl(int s, int* rows, int* cols, size_t* off) {
    *rows = FDebugLayout::rows(s);
    *cols = FDebugLayout::cols(s);
    *off = FDebugLayout::offset(s);
}

template <class S>
FImpl make_fimpl() {
    constexpr int OCC_LDS = (160 * 1024) / (FLds::TOTAL * 4);
    return FImpl{S::HOP, (size_t)FLds::TOTAL * 4, FDebugLayout::total(), FDebugLayout::n_stages, (size_t)FPk::TOTAL, &flaunch_impl<S>, &fdbg_stage_impl,
                 &flaunch_pipe_impl<S>, OCC_LDS < FS_WPE ? OCC_LDS : FS_WPE, S::NB, &flaunch_sb_impl<S>, (size_t)2048 + FCarry::FLOATS + 1024};
}

}  // namespace fe
