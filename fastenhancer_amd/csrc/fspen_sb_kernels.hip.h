// fspen_sb_kernels.hip.h — the MIDDLE of FSPEN BATCHED OVER THE STREAMS on the fp32 matrix cores (gfx950), for the per-hop step of large
// batches: fullband_encoder.2, fullband_encoder_post, feature merge, the three DPE blocks, feature split and fullband_decoder.0-1 (models/fspen/model.py:
// 244-264 around 122-189: per block an intra bidirectional GRU over the 32 sub-bands + intra_fc + LayerNorm + residual, then eight
// grouped inter GRUs over time + inter_fc + residuals) - 0.97 of the model's 1.04 MMAC per frame.
//
// fspen_frame_kernel gives a stream a workgroup: every product is then M = 1 (vector FMAs), and the intra GRU is a chain of 96
// dependent steps per frame on two of its four waves.  Here a workgroup takes SIXTEEN streams and every product is a matrix-core
// GEMM with the streams as the N dimension, computed TRANSPOSED (as in bsrnn_sb_kernels.hip.h):
//     out^T [rows x 16 streams] = W [rows x K] . in^T [K x 16 streams]
// A = weights (fragments in registers), B = activations: lane (li, lg) holds feature 4 ks + lg of stream li for k-step ks,
// C/D = lane (li, lg) holds rows 4 lg + r of stream li.  The ROW ORDER of every A operand is chosen by the host packer:
//   * intra GRU step: all eight waves take part - wave (direction d = wave / 4, quarter q = wave % 4) owns hidden units 4 q .. 4 q + 3 of
//     its direction; its ONE 16-row tile is (unit 4 q + j, gate g) at row 4 j + g with the gates r, z, n_x, n_h: K = 32 (the x half of
//     n_x and the h half of n_h only - zero fragments elsewhere).  The four gate values of a (stream, unit) are the four accumulator
//     registers of one lane: one gate evaluation per lane and step (6 transcendentals), the new h goes to the h sequence in LDS -
//     [direction][sub-band][unit][16 streams] - which is at once the exchange buffer of the next step (ONE barrier per step for both
//     directions) and the input of intra_fc.  The x half of the next step is issued one step ahead.
//   * intra_fc, the inter GRUs' gate tiles and inter_fc: row 4 lg + r <-> feature 4 r + lg.  A lane's accumulator register r is then
//     feature 4 r + lg = what the B operand of k-step r carries: LayerNorm output, new inter state and block output feed the next product
//     from registers, and their LDS stores ([feature][16 streams], 64 consecutive floats per register) are conflict-free.
// Wave w owns sub-bands 4 w .. 4 w + 3 = inter group w for intra_fc / LayerNorm / inter GRU / inter_fc: no barrier between those
// phases but the two of the LayerNorm statistics (sums over a stream's 512 values: lanes li, li + 16, .. of eight waves).
// The layers around the blocks contract alternately over channels and over positions: a product stores its C/D tile TRANSPOSED into one
// of two [32 rows][32 + 1 pad][16 streams] LDS matrices (R0 / R1, aliasing the tokens / h sequences of the DPE) and the next product reads
// its B operands from there; operands that the front left in global memory (enc_out[2], the sub-band features) are read straight from
// there, 16 bytes (four positions or four k-steps) per lane; fullband_decoder.0's transposed convolution is, per pair of output
// positions 2 m / 2 m + 1, two 16-row tiles over K = 3 taps x 32 channels of the positions m + 1, m, m - 1.
// LDS: 136 KB: one workgroup per CU, 4096 streams fill the chip.
// (included by fspen_kernels.hip.h, after FShape / FPk / FLds / FCarry)
#pragma once
#include <atomic>

namespace fe {

constexpr int kFsbStreams = 16;
constexpr int kFsbThreads = 512;

struct FSbLds {
    static constexpr int X = 0;                       // [32 f][16 c][16 n]
    static constexpr int HS = X + 32 * 16 * 16;       // [2 d][32 f][16 u][16 n]
    // before / after the DPE blocks the same space holds two [32 rows][32 + 1 pad][16 n] matrices (R0 over X and the head of HS, R1 behind it)
    static constexpr int MS = 33 * 16;                // their row stride: 528 = 16 mod 64 - the C/D-layout stores of four lane groups hit four bank quarters
    static constexpr int R0 = 0, R1 = 32 * MS;
    static constexpr int E1S = 69;                    // fullband_encoder.1's output in R1: [16 c][68 + 1 pad positions][16 n] (channel stride 69 x 16 = 16 mod 64)
    static constexpr int RED = R1 + 16 * E1S * 16;    // [2][8 waves][16 n]
    static_assert(16 * E1S * 16 >= 32 * MS, "R1 holds either");
    static_assert(RED >= HS + 2 * 32 * 16 * 16, "the h sequences end before the reduction slots");
    static constexpr int WX = RED + 2 * 8 * 16;       // fullband_decoder.1's fragments and bias (FSbPk::FD1_W ..): a region of their own - R0 / R1 are busy by then
    static constexpr int TOTAL = WX + (FSbPk::FD1T_B + 16 - FSbPk::FD1_W);
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
};

struct FSbArgs {
    const float* wp;          // the packed buffer of fspen_frame_kernel; the stream-batched region starts at FPk::SB
    float* carry;             // [B][FCarry::FLOATS] the front's LDS regions (fspen_frame_kernel PART 1): enc_out[1] and the sub-band features are read here,
                              // enc_out[2] is written (and read back by fullband_decoder.0 at the end)
    float* s2;                // [B][2][1024] for the tail (fspen_frame_kernel PART 2): feature_split output, sub-band half [32][32] | fullband_decoder.0 output [16][64]
    float* e2x;               // [ceil(B / 16)][8 waves][8][64 lanes][4] enc_out[2] of the step, lane-private: written after fullband_encoder.2, read back by fullband_decoder.0
    float* gru;               // [24][B * 4][16] inter-GRU states
    int B;
    unsigned long long* clk;  // fe_profile_step: cycle counters of workgroup 0 (slots 32 ..), else null
};

__device__ __forceinline__ float fsb_sig(float pre) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre)); }                            // pre = -log2e x
__device__ __forceinline__ float fsb_tanh(float pre) { return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre)), 1.0f); }   // pre = 2 log2e x

// sum over the four lane groups (lanes li, li + 16, li + 32, li + 48)
__device__ __forceinline__ float fsb_sum_lg(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <class S>      // (S = FShape<HOP>: the DPE does not depend on it - a template so that the kernel is emitted by the translation unit that launches it)
__global__ void __launch_bounds__(kFsbThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) fspen_sb_dpe_kernel(FSbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = FSbLds;
    using Q = FSbPk;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, FPk::TOTAL * 4, 0x00020000);
    auto ldw = [&](int off_floats, int voff_bytes) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_bytes, off_floats * 4, 0));
    };
    auto ldw4 = [&](int off_floats, int voff_bytes) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, off_floats * 4, 0));
    };
#define FSB_CLK(i) do { if (a.clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.clk[32 + (i)] = __builtin_readcyclecounter(); } while (0)
    FSB_CLK(0);
    float* X = smem + L::X;
    float* HS = smem + L::HS;
    float* red = smem + L::RED;
    const int b0 = blockIdx.x * kFsbStreams;
    const bool live = b0 + li < a.B;
    const int bn = live ? b0 + li : a.B - 1;                     // this lane's stream (tiles past the batch repeat the last stream; stores predicated)
    // ---------------- fullband_encoder_post (1x1, 32 -> 32, :244) + feature merge (:246-250): Linear(64 -> 32) over the band axis, ELU, 1x1 (32 -> 16) ----
    float* cr = a.carry + (size_t)bn * FCarry::FLOATS + FCarry::AN;                // the front's LDS region [E0, SB) of this lane's stream
    // enc_out[2] between fullband_encoder.2 and fullband_decoder.0: the lane that produces (channel 16 ot + 4 r + lg, positions 4 w .. 4 w + 3) of
    // stream li is the lane that consumes it - 16 bytes per (ot, r), whole 1-KB lines per wave instruction (as [32 o][32 f] rows per stream the
    // same stores were 64 partial lines per instruction and took ~12 k cycles to drain)
    f32x4* e2x = reinterpret_cast<f32x4*>(a.e2x) + ((size_t)blockIdx.x * 8 + wave) * 8 * 64 + lane;
    // ---------------- fullband_encoder.2 (Conv1d 16 -> 32, k 6, s 2, p 2, + folded BN + ELU, :238-242) ----------------
    // enc_out[1] [16 c][68 positions, zero padded] of the sixteen streams -> LDS, stream-minor; then per output position j one product
    // e2^T [32 o x 16 n] = W [32 x (6 taps x 16 c)] . patch_j^T, k-step 4 tap + cq <-> (tap, channel 4 cq + lg) at position 2 j + tap
    // The weight fragments of these layers are the same for all eight waves: fetched by every wave they are 1.3 k wave-level loads per CU at
    // ~16 cycles of its vector-memory path each (20 k cycles measured).  The workgroup copies them ONCE into LDS (R0 is free until the post
    // layer writes there) with 16-byte loads, and every wave picks its register copy from there.
    constexpr int WS0 = Q::FE2_W, WS0_N = Q::MG2_B + 16 - Q::FE2_W;        // fullband_encoder.2 .. feature_merge.2, contiguous in the packed buffer
    static_assert(WS0_N % 4 == 0 && WS0_N <= 32 * L::MS, "start weights fit R0");
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.wp + FPk::SB + WS0);
        f32x4* dst = reinterpret_cast<f32x4*>(smem + L::R0);
        for (int i = tid; i < WS0_N / 4; i += kFsbThreads) dst[i] = src[i];
    }
    f32x4 cv[4][2];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) cv[c4][q4] = *reinterpret_cast<const f32x4*>(cr + (FLds::CAT - FLds::E0) + (4 * wave + c4) * 64 + 32 + 16 * q4 + 4 * lg);
    auto lw0 = [&](int off) { return smem[L::R0 + (off - WS0) + lane]; };                                                 // fragment at packed offset `off`
    auto lw0_4 = [&](int off) { return *reinterpret_cast<const f32x4*>(smem + L::R0 + (off - WS0) + 4 * lg); };            // bias [lg][r]
    float w1[2][16], w2m[8], wp_[2][8], wf[2][24];
    f32x4 b2, bf[2];
    float e2r[4][2][4];                  // this wave's positions 4 w .. 4 w + 3: e2r[fl][ot][r] = channel 16 ot + 4 r + lg = the B operand of k-step 4 ot + r below
    {
        float* E1L = smem + L::R1;
        {
            // thread (stream n = tid % 16, j = tid / 16) takes the 16-byte pieces j, j + 32, .. of its stream's [16][68] rows (17 pieces per row):
            // four consecutive lanes read 64 contiguous bytes
            const int n = tid & 15, j = tid >> 4;
            const int bs = b0 + n < a.B ? b0 + n : a.B - 1;
            const f32x4* src = reinterpret_cast<const f32x4*>(a.carry + (size_t)bs * FCarry::FLOATS + FCarry::AN + (FLds::E1 - FLds::E0));
            f32x4 v[9];
#pragma unroll
            for (int q9 = 0; q9 < 9; ++q9) { const int i4 = j + 32 * q9; v[q9] = src[i4 < 272 ? i4 : 271]; }
#pragma unroll
            for (int q9 = 0; q9 < 9; ++q9) {
                const int i4 = j + 32 * q9, c = i4 / 17, p0 = (i4 - 17 * c) * 4;
                if (i4 < 272) {
                    float* dst = E1L + (c * L::E1S + p0) * 16 + n;
                    dst[0] = v[q9][0]; dst[16] = v[q9][1]; dst[32] = v[q9][2]; dst[48] = v[q9][3];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int ks = 0; ks < 24; ++ks) wf[ot][ks] = lw0(Q::FE2_W + (ot * 24 + ks) * 64);
        bf[0] = lw0_4(Q::FE2_B); bf[1] = lw0_4(Q::FE2_B + 16);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) wp_[ot][ks] = lw0(Q::POST_W + (ot * 8 + ks) * 64);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) w1[jt][ks] = lw0(Q::MG1_W + (jt * 16 + ks) * 64);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) w2m[ks] = lw0(Q::MG2_W + ks * 64);
        b2 = lw0_4(Q::MG2_B);
        __syncthreads();                 // (every wave has its copy: R0 may be overwritten)
        FSB_CLK(15);
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) {
            f32x4 acc[2] = {bf[0], bf[1]};
            const float* xp = E1L + (lg * L::E1S + 2 * (4 * wave + fl)) * 16 + li;
            float xb[24];        // (all B operands of the position first: left alone, hipcc sinks each LDS read next to its MFMA and the loop runs at LDS latency)
#pragma unroll
            for (int ks = 0; ks < 24; ++ks) xb[ks] = xp[(4 * (ks & 3) * L::E1S + (ks >> 2)) * 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 24; ++ks)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) acc[ot] = FE_MFMA(wf[ot][ks], xb[ks], acc[ot]);
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) e2r[fl][ot][r] = elu_f(acc[ot][r]);
        }
    }
    float* P = smem + L::R0;             // post output [32 o][32 f + pad][16 n]
    float* M1 = smem + L::R1;            // merge Linear output [32 j][32 ch + pad][16 n]
    FSB_CLK(16);
    {   // wave w takes positions f = 4 w .. 4 w + 3: post^T [32 o x 16 n] = Wp [32 x 32] . e2[:, f]^T, B = the accumulators above
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) {
            f32x4 acc[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) acc[ot] = FE_MFMA(wp_[ot][ks], e2r[fl][ks >> 2][ks & 3], acc[ot]);
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[(16 * ot + 4 * r + lg) * L::MS + (4 * wave + fl) * 16 + li] = acc[ot][r];
        }
    }
    __syncthreads();
    FSB_CLK(17);
    {   // wave w takes channels 4 w .. 4 w + 3 of cat = post | sub-band features [32 ch][64]: m1^T [32 j x 16 n] = W1 [32 x 64] . cat[ch]^T; the
        // full-band half from LDS, the sub-band half straight from global memory (k-step 8 + 4 q + e <-> input 32 + 16 q + 4 lg + e)
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            f32x4 acc[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
            float pb[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) pb[ks] = P[(4 * wave + c4) * L::MS + (4 * ks + lg) * 16 + li];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) acc[jt] = FE_MFMA(w1[jt][ks], pb[ks], acc[jt]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) acc[jt] = FE_MFMA(w1[jt][8 + ks], cv[c4][ks >> 2][ks & 3], acc[jt]);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) M1[(16 * jt + 4 * r + lg) * L::MS + (4 * wave + c4) * 16 + li] = elu_f(acc[jt][r]);
        }
    }
    __syncthreads();
    FSB_CLK(18);
    // 1x1 (32 -> 16): wave w owns sub-bands 4 w .. 4 w + 3 from here on - the residual stream in registers, xr[fl][r] = channel 4 r + lg of stream li
    float xr[4][4];
    {
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) {
            f32x4 acc = b2;
            float mb[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) mb[ks] = M1[(4 * wave + fl) * L::MS + (4 * ks + lg) * 16 + li];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = FE_MFMA(w2m[ks], mb[ks], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xr[fl][r] = acc[r];
                X[((4 * wave + fl) * 16 + 4 * r + lg) * 16 + li] = acc[r];
            }
        }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) e2x[(4 * ot + r) * 64] = f32x4{e2r[0][ot][r], e2r[1][ot][r], e2r[2][ot][r], e2r[3][ot][r]};
    __syncthreads();
    FSB_CLK(1);
    const int d = wave >> 2, q = wave & 3;
#pragma unroll 1
    for (int blk = 0; blk < 3; ++blk) {
        int lz = 0;
        asm volatile("" : "+s"(lz));
        const int D = FPk::SB + blk * Q::D_SIZE + lz;
        // every weight of the block and the inter-GRU states are requested here: they land under the first recurrence steps
        float aw[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) aw[ks] = ldw(D + Q::I_W + (wave * 8 + ks) * 64, lane * 4);
        const f32x4 bias = ldw4(D + Q::I_B + (wave * 4) * 4, lg * 16);
        float fw[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) fw[ks] = ldw(D + Q::FC_W + ks * 64, lane * 4);
        const f32x4 fb = ldw4(D + Q::FC_B, lg * 16);
        f32x4 lw[4], lb[4];
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) {
            lw[fl] = ldw4(D + Q::LN_W + (4 * wave + fl) * 16, lg * 16);
            lb[fl] = ldw4(D + Q::LN_B + (4 * wave + fl) * 16, lg * 16);
        }
        const int G = D + Q::GRP + wave * Q::G_SIZE;
        float gw[24];                                       // r: 8 k-steps (x | h), z: 8, n_x: 4, n_h: 4
#pragma unroll
        for (int i = 0; i < 24; ++i) gw[i] = ldw(G + Q::G_W + i * 64, lane * 4);
        float cw[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cw[ks] = ldw(G + Q::G_FCW + ks * 64, lane * 4);
        const f32x4 b_r = ldw4(G + Q::G_B, lg * 16), b_z = ldw4(G + Q::G_B + 16, lg * 16), b_nx = ldw4(G + Q::G_B + 32, lg * 16),
                    b_nh = ldw4(G + Q::G_B + 48, lg * 16), b_fc = ldw4(G + Q::G_FCB, lg * 16);
        float* const st0 = a.gru + (((size_t)(blk * 8 + wave) * a.B + bn) * 4) * 16;
        float hp[4][4];
#pragma unroll
        for (int fl = 0; fl < 4; ++fl)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) hp[fl][ks] = st0[fl * 16 + 4 * ks + lg];
        // ---------------- intra GRU: 32 steps, both directions, one barrier per step ----------------
        FSB_CLK(2 + 4 * blk);
        {
            const int f0 = d ? 31 : 0, fstep = d ? -1 : 1;
            auto xload = [&](float (&xb)[4], int f) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) xb[ks] = X[(f * 16 + 4 * ks + lg) * 16 + li];
            };
            // the x half of a step - always issued one step ahead, off the h chain
            auto xpart = [&](const float (&xb)[4]) {
                f32x4 p = bias;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) p = FE_MFMA(aw[ks], xb[ks], p);
                return p;
            };
            auto gates = [&](const f32x4& acc, float h) {
                const float rg = fsb_sig(acc[0]), zg = fsb_sig(acc[1]);
                const float ng = fsb_tanh(__builtin_fmaf(rg, acc[3], acc[2]));
                return __builtin_fmaf(zg, h - ng, ng);                      // (1 - z) n + z h
            };
            // One workgroup barrier per step for both directions.  On gfx950 the fp32 MFMAs and the vector ALU share a SIMD's datapath:
            // a step costs its 16 MFMAs (two waves x 8, 512 cycles) PLUS its vector instructions (timing experiments, cycles per step of
            // 1.19 k: no x MFMAs -325, no h MFMAs -272, no barrier -210, no transcendentals -160) - no ordering of the two overlaps
            // them (x half before / after / interleaved with the gate math: 36.0 / 38.2 / 38.2 k cycles per block), and neither does
            // decoupling the directions (each direction's four waves on their own LDS counter, release add / acquire poll: 40.5 k).
            float xb[4];
            xload(xb, f0);
            f32x4 accx = xpart(xb);
            xload(xb, f0 + fstep);
            // step 0: h = 0
            f32x4 accn = xpart(xb);                                         // step 1's x half
            float h = gates(accx, 0.0f);
            HS[((d * 32 + f0) * 16 + 4 * q + lg) * 16 + li] = h;
            accx = accn;
            xload(xb, f0 + 2 * fstep);
            __syncthreads();
            int f = f0 + fstep;
#pragma unroll 1
            for (int s = 1; s < 32; ++s) {
                float hb[4];
                const float* hq = HS + ((d * 32 + (f - fstep)) * 16) * 16;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) hb[ks] = hq[(4 * ks + lg) * 16 + li];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) accx = FE_MFMA(aw[4 + ks], hb[ks], accx);
                accn = xpart(xb);                 // the next step's x half
                h = gates(accx, h);
                HS[((d * 32 + f) * 16 + 4 * q + lg) * 16 + li] = h;
                accx = accn;
                f += fstep;
                const int fn = f + fstep;
                xload(xb, fn & 31);               // (the load after the last step is not used)
                __syncthreads();
            }
        }
        // the h half of the inter GRUs' gates does not wait for the LayerNorm: issued here, under the statistics' barriers
        f32x4 ar[4], az[4], anx[4], anh[4];
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) { ar[fl] = b_r; az[fl] = b_z; anx[fl] = b_nx; anh[fl] = b_nh; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) {
                ar[fl] = FE_MFMA(gw[4 + ks], hp[fl][ks], ar[fl]);
                az[fl] = FE_MFMA(gw[12 + ks], hp[fl][ks], az[fl]);
                anh[fl] = FE_MFMA(gw[20 + ks], hp[fl][ks], anh[fl]);
            }
        // ---------------- intra_fc + LayerNorm([F, C]) + residual: wave w owns sub-bands 4 w .. 4 w + 3 ----------------
        FSB_CLK(3 + 4 * blk);
        float xn[4][4];
        {
            f32x4 y[4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) y[fl] = fb;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int fl = 0; fl < 4; ++fl) {
                    const float hb = HS[(((ks >> 2) * 32 + 4 * wave + fl) * 16 + 4 * (ks & 3) + lg) * 16 + li];
                    y[fl] = FE_MFMA(fw[ks], hb, y[fl]);
                }
            float s0 = 0.0f;
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) s0 += (y[fl][0] + y[fl][1]) + (y[fl][2] + y[fl][3]);
            s0 = fsb_sum_lg(s0);
            if (lg == 0) red[wave * 16 + li] = s0;
            __syncthreads();
            float mean = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) mean += red[w * 16 + li];
            mean *= (1.0f / 512.0f);
            float s1 = 0.0f;
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) { y[fl][r] -= mean; s1 = __builtin_fmaf(y[fl][r], y[fl][r], s1); }
            s1 = fsb_sum_lg(s1);
            if (lg == 0) red[128 + wave * 16 + li] = s1;
            __syncthreads();
            float var = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) var += red[128 + w * 16 + li];
            const float inv_std = __builtin_amdgcn_rsqf(var * (1.0f / 512.0f) + 1.0e-5f);
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) xn[fl][r] = __builtin_fmaf(y[fl][r] * inv_std, lw[fl][r], lb[fl][r]) + xr[fl][r];
        }
        // ---------------- inter path: group g = wave, one GRU step per sub-band row, inter_fc, residuals ----------------
        FSB_CLK(4 + 4 * blk);
        {
            // the four rows' gate products interleaved: four independent chains per gate
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int fl = 0; fl < 4; ++fl) {
                    ar[fl] = FE_MFMA(gw[ks], xn[fl][ks], ar[fl]);
                    az[fl] = FE_MFMA(gw[8 + ks], xn[fl][ks], az[fl]);
                    anx[fl] = FE_MFMA(gw[16 + ks], xn[fl][ks], anx[fl]);
                }
            float hn[4][4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float rg = fsb_sig(ar[fl][r]), zg = fsb_sig(az[fl][r]);
                    const float ng = fsb_tanh(__builtin_fmaf(rg, anh[fl][r], anx[fl][r]));
                    hn[fl][r] = __builtin_fmaf(zg, hp[fl][r] - ng, ng);
                    if (live) st0[fl * 16 + 4 * r + lg] = hn[fl][r];
                }
            f32x4 o[4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) o[fl] = b_fc;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int fl = 0; fl < 4; ++fl) o[fl] = FE_MFMA(cw[ks], hn[fl][ks], o[fl]);
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xr[fl][r] = __builtin_fmaf(2.0f, xn[fl][r], o[fl][r]);     // + x_in inside the path extension, + x_in again in DPE.forward
                    X[((4 * wave + fl) * 16 + 4 * r + lg) * 16 + li] = xr[fl][r];
                }
        }
        __syncthreads();
        FSB_CLK(5 + 4 * blk);
    }
    // ---------------- feature split (:256-260): 1x1 (16 -> 32), Linear(32 -> 64) over the band axis, ELU; fullband_decoder.0's 1x1 (:262-264) ----------------
    {
        float* S1 = smem + L::R1;            // [32 ch][32 f + pad][16 n]
        float* S2T = smem + L::R0;           // full-band half of the split output, transposed: [32 f][32 ch + pad][16 n]
        // the section's weight fragments through LDS, like the start section's (R0: the tokens and h sequences are dead)
        constexpr int WS1 = Q::SP1_W, WS1_N = Q::FD0T_B + 16 - Q::SP1_W;
        static_assert(WS1_N % 4 == 0 && WS1_N <= 32 * L::MS, "end weights fit R0");
        {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.wp + FPk::SB + WS1);
            f32x4* dst = reinterpret_cast<f32x4*>(smem + L::R0);
            for (int i = tid; i < WS1_N / 4; i += kFsbThreads) dst[i] = src[i];
        }
        {
            constexpr int WX_N = Q::FD1T_B + 16 - Q::FD1_W;
            static_assert(WX_N % 4 == 0, "16-byte copy");
            const f32x4* src = reinterpret_cast<const f32x4*>(a.wp + FPk::SB + Q::FD1_W);
            f32x4* dst = reinterpret_cast<f32x4*>(smem + L::WX);
            for (int i = tid; i < WX_N / 4; i += kFsbThreads) dst[i] = src[i];
        }
        auto lwx = [&](int off) { return smem[L::WX + (off - Q::FD1_W) + lane]; };
        auto lw1 = [&](int off) { return smem[L::R0 + (off - WS1) + lane]; };
        auto lw1_4 = [&](int off) { return *reinterpret_cast<const f32x4*>(smem + L::R0 + (off - WS1) + 4 * lg); };
        __syncthreads();
        float ws[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ws[ct][ks] = lw1(Q::SP1_W + (ct * 4 + ks) * 64);
        const f32x4 bs[2] = {lw1_4(Q::SP1_B), lw1_4(Q::SP1_B + 16)};
        float w2[4][8];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) w2[jt][ks] = lw1(Q::SP2_W + (jt * 8 + ks) * 64);
        float wd[2][16];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) wd[ot][ks] = lw1(Q::FD0_W + (ot * 16 + ks) * 64);
        f32x4 ev[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) ev[ks] = e2x[ks * 64];
        // (all of the section's loads before its first global store: see the note at the enc_out[2] stores)
        float wt[2][3][8];
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) wt[par][t][ks] = lw1(Q::FD0T_W + ((par * 3 + t) * 8 + ks) * 64);
        const f32x4 bt = lw1_4(Q::FD0T_B);
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) {
            f32x4 acc[2] = {bs[0], bs[1]};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ct] = FE_MFMA(ws[ct][ks], xr[fl][ks], acc[ct]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) S1[(16 * ct + 4 * r + lg) * L::MS + (4 * wave + fl) * 16 + li] = acc[ct][r];
        }
        __syncthreads();
        // wave w takes channels 4 w .. 4 w + 3: s2^T [64 j x 16 n] = W2 [64 x 32] . s1[ch]^T.  Full-band half (j < 32, rows 4 lg + r <-> j 4 r + lg):
        // to LDS, transposed, for the decoder's 1x1; sub-band half (rows in natural order: a lane's four rows = 16 bytes): to the tail
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int ch = 4 * wave + c4;
            f32x4 acc[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) acc[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            float sb[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) sb[ks] = S1[ch * L::MS + (4 * ks + lg) * 16 + li];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) acc[jt] = FE_MFMA(w2[jt][ks], sb[ks], acc[jt]);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) S2T[(16 * jt + 4 * r + lg) * L::MS + ch * 16 + li] = elu_f(acc[jt][r]);
            if (live) {
#pragma unroll
                for (int jt = 2; jt < 4; ++jt) {
                    const f32x4 o = {elu_f(acc[jt][0]), elu_f(acc[jt][1]), elu_f(acc[jt][2]), elu_f(acc[jt][3])};
                    *reinterpret_cast<f32x4*>(a.s2 + (size_t)bn * 2048 + ch * 32 + 16 * (jt - 2) + 4 * lg) = o;
                }
            }
        }
        // fullband_decoder.0, 1x1 over cat(x_full, enc_out[2]) (64 -> 32, no bias): wave w takes positions f = 4 w .. 4 w + 3
        __syncthreads();
        float* T2L = smem + L::R1;           // the 1x1's output, position-major: [32 f][32 c + pad][16 n] (S1 is dead)
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) {
            f32x4 t2v[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
            float sb[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) sb[ks] = S2T[(4 * wave + fl) * L::MS + (4 * ks + lg) * 16 + li];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) t2v[ot] = FE_MFMA(wd[ot][ks], sb[ks], t2v[ot]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) t2v[ot] = FE_MFMA(wd[ot][8 + ks], ev[ks][fl], t2v[ot]);
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) T2L[(4 * wave + fl) * L::MS + (16 * ot + 4 * r + lg) * 16 + li] = t2v[ot][r];
        }
        // fullband_decoder.0's ConvTranspose1d(32 -> 16, k 6, s 2, p 2) + folded BN + ELU: output positions 2 m + q (q = 0, 1) take kernel
        // indices q + 2 i from input positions m + 1 - i (i = 0, 1, 2).  Wave w takes m = 4 w .. 4 w + 3: per m and parity one 16-row tile, K = 3 x 32
        __syncthreads();
        f32x4 dv[4][2];
#pragma unroll
        for (int ml = 0; ml < 4; ++ml) {
            const int m = 4 * wave + ml;
            dv[ml][0] = bt; dv[ml][1] = bt;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int f = m + 1 - t;
                if (f >= 0 && f < 32) {          // (wave-uniform)
                    float tb[8];
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) tb[ks] = T2L[f * L::MS + (4 * ks + lg) * 16 + li];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        dv[ml][0] = FE_MFMA(wt[0][t][ks], tb[ks], dv[ml][0]);
                        dv[ml][1] = FE_MFMA(wt[1][t][ks], tb[ks], dv[ml][1]);
                    }
                }
            }
        }
        // ---------------- fullband_decoder.1 (:266-270): 1x1 over cat(d2, enc_out[1]) (32 -> 16, no bias), ConvTranspose1d(16 -> 4, k 8, s 2, p 3) + folded BN + ELU ----------------
        // the 1x1 per output position p = 8 w .. 8 w + 7 of this wave: the d2 half of K straight from the accumulators above (register r of a lane =
        // channel 4 lg + r = k-step r of the "4 lg + ks" k-order), the enc_out[1] half from global memory (8 bytes = two positions per load)
        float* T1L = smem + L::R0;           // its output, position-major: [64 p][16 c][16 n] (the split output there is dead)
        {
            float wa[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) wa[ks] = lwx(Q::FD1_W + ks * 64);
            float2 e1v[4][4];
            const float* e1g = cr + (FLds::E1 - FLds::E0) + 2 + 8 * wave;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int q2 = 0; q2 < 4; ++q2) e1v[ks][q2] = *reinterpret_cast<const float2*>(e1g + (4 * ks + lg) * 68 + 2 * q2);
#pragma unroll
            for (int ml = 0; ml < 4; ++ml)
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = FE_MFMA(wa[ks], elu_f(dv[ml][par][ks]), acc);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = FE_MFMA(wa[4 + ks], par ? e1v[ks][ml].y : e1v[ks][ml].x, acc);
                    const int p = 8 * wave + 2 * ml + par;
#pragma unroll
                    for (int r = 0; r < 4; ++r) T1L[(p * 16 + 4 * r + lg) * 16 + li] = acc[r];
                }
        }
        float wq[5][4];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) wq[j][cq] = lwx(Q::FD1T_W + (j * 4 + cq) * 64);
        const f32x4 bq = *reinterpret_cast<const f32x4*>(smem + L::WX + (Q::FD1T_B - Q::FD1_W) + 4 * lg);
        __syncthreads();
        // the transposed convolution: output positions 2 m + q take kernel indices 7 - 2 j (q = 0) / 8 - 2 j (q = 1) from input positions m - 2 + j, j < 5.
        // Per m ONE 16-row tile (rows 4 q + o, q < 2, o < 4; rows 8 .. 15 idle) over K = 5 positions x 16 channels; wave w takes m = 8 w .. 8 w + 7
        {
            f32x4 d1v[8];
#pragma unroll
            for (int ml = 0; ml < 8; ++ml) {
                const int m = 8 * wave + ml;
                d1v[ml] = bq;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int f = m - 2 + j;
                    if (f >= 0 && f < 64) {          // (wave-uniform)
                        float tb[4];
#pragma unroll
                        for (int cq = 0; cq < 4; ++cq) tb[cq] = T1L[(f * 16 + 4 * cq + lg) * 16 + li];
#pragma unroll
                        for (int cq = 0; cq < 4; ++cq) d1v[ml] = FE_MFMA(wq[j][cq], tb[cq], d1v[ml]);
                    }
                }
            }
            // d1 [4 o][128 p] for the tail: lane group lg < 2 holds parity lg of output o = r; the parities of a position pair meet in the lg = 0 lanes
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float own[8], oth[8];
#pragma unroll
                for (int ml = 0; ml < 8; ++ml) { own[ml] = elu_f(d1v[ml][r]); oth[ml] = __shfl_xor(own[ml], 16, 64); }
                if (live && lg == 0) {
                    float* dst = a.s2 + (size_t)bn * 2048 + 1024 + r * 128 + 16 * wave;
#pragma unroll
                    for (int h4 = 0; h4 < 4; ++h4)
                        *reinterpret_cast<f32x4*>(dst + 4 * h4) = f32x4{own[2 * h4], oth[2 * h4], own[2 * h4 + 1], oth[2 * h4 + 1]};
                }
            }
        }
    }
    FSB_CLK(14);
#undef FSB_CLK
}

template <class S>
hipError_t fspen_sb_launch(const FSbArgs& a, hipStream_t st) {
    static std::atomic<bool> attr_set[64];          // (per device: a process may drive several)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fspen_sb_dpe_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FSbLds::BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    const int grid = (a.B + kFsbStreams - 1) / kFsbStreams;
    note_kernel("fspen_sb_dpe_kernel");
    hipLaunchKernelGGL(fspen_sb_dpe_kernel<S>, dim3(grid), dim3(kFsbThreads), FSbLds::BYTES, st, a);
    return hipGetLastError();
}

}  // namespace fe
