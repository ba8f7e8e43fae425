// tb_kernels.hip.h — the TIME-BATCHED (layer-by-layer) engine of the FastEnhancer forward path for gfx950 (MI355X).
//
// The reference's offline / chunk forward runs all T frames of an utterance as ONE batch
// (models/fastenhancer/default/model.py:620-675, 728-735; noncausal/model.py:578-635): every layer but the blocks' time GRU
// is independent from frame to frame.  The per-hop kernel (fe_kernels.hip.h) walks the frames of a stream one after the
// other; here the network is cut at the GRUs instead and every piece runs over ALL frames of ALL utterances:
//
//   tb_enc_kernel   per tile of FT frames: STFT, compress, enc_pre, encoder, rf_pre, and the x half of block 0's GRU gates
//                   (W_ih x + b: it does not depend on the previous frame)                      model.py:628-650, 266-271
//   tb_scan_kernel  per block: the recurrence itself, h_t = GRU(gx_t, h_{t-1}), 16 (utterance, sub-band) rows per
//                   workgroup, only W_hh h serial; forward and - noncausal - reverse in time     model.py:271 / noncausal :186,271
//   tb_blk_kernel   per tile of frames: rnn_fc + residual (+ pe), qkv, attention, attn_fc + residual, and the x half of the
//                   NEXT block's gates                                                          model.py:273-290
//   tb_dec_kernel   per tile of frames: rf_post, decoder, dec_post, mask, un-compress, inverse DFT, synthesis window
//                   (the overlap-add is istft_ola_kernel's)                                     model.py:654-674, 694-709
//
// Between the launches the activations that cross a GRU live in HBM (288 GB: the whole batch fits): the compressed
// spectrum, the encoder outputs (skips, in MFMA A-fragment order so that the decoder reads them back as coalesced 256-byte
// fragments), the token stream x [frame][F2][C2], the gate pre-activations gx and the GRU outputs hs.  Inside a launch a tile of
// FT consecutive frames is ONE GEMM problem per layer: M = FT * F1 conv rows / FT * F2 token rows (no 24 -> 32 padding of the token
// rows at FT = 2 for FastEnhancer_B), one barrier-bounded phase per layer and TILE instead of per layer and frame.
// All contractions run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) through the same software-pipelined panels as
// the per-hop kernel; weights are read from the same packed buffer (fe::Pack<S>, fragment order, L2-resident).
#pragma once
#include <atomic>
#include <cstdlib>

#include "fe_kernels.hip.h"

namespace fe {
constexpr int kMaxDevices = 64;
namespace tb {

constexpr int kPD = 8;          // software-pipeline depth of the MFMA panels: weight fragments stream from L2

struct TbArgs {
    const float* wp;            // packed weights + tables (Pack<S>::v)
    const float* wav_in;        // offline: [b * in_stride + n], n < Tw
    size_t in_stride;
    const float* spec_in;       // spec mode: [B][F0+1][T][2]
    float* spec_out;            // spec mode: [B][F0+1][T][2]; offline: spec_hat [B][F0][T][2] (compressed domain)
    float* frames;              // offline: [B][T][N] windowed output frames (summed by istft_ola_kernel)
    float* xc;                  // [NF][2][F0]  compressed spectrum (Re plane, Im plane)
    float* skip;                // [NF][NL+1][F1*C1]  encoder outputs, A-fragment order
    // token-layout buffers are CHANNEL-MAJOR per frame, [frame][channel][sub-band]: the four rows an accumulator lane holds (C/D layout:
    // rows 4 lg .. 4 lg + 3 of a 16-row tile = four consecutive sub-bands, F2 % 4 == 0) are ONE 16-byte access, in the GEMM epilogues
    // that write them and in the scan that reads gx / writes hs per (row, channel)
    float* x;                   // [NF][C2][F2]  token stream between the blocks
    float* gx;                  // [ND][NF][3 C2][F2]  W_ih x + b_ih (+ b_hh for r, z)
    float* hs;                  // [NF][ND*C2][F2]  GRU outputs
    float* hstate;              // [KB][B*F2][C2] carried GRU state (spec -> spec with caches), or nullptr: zero initial state
    int B, T, NF;               // the NODE's utterances, frames per utterance and B * T: the work buffers are indexed by the node-local frame b * T + t
    int b0, t0;                 // the node's first utterance / first frame within the whole call (input / output addressing, carried state)
    int Bfull, Tfull;           // utterances and frames per utterance of the whole call
    int h_init;                 // tb_scan_kernel: 1 = start from hstate (a later time chunk, or caches handed in), 0 = zero initial state
    int Tw;                     // offline: samples per utterance
    const int* Tw_b;            // offline, ragged batch (fe_offline_ragged): samples of utterance b, [Bfull], every one <= Tw (the batch is laid out for
                                // Tw: frames past an utterance's own 1 + Tw_b / H are computed on clamped input and never reach its output); nullptr: all Tw
    int mode;                   // FE_MODE_OFFLINE / FE_MODE_SPEC
    int k;                      // block index (tb_scan_kernel, tb_blk_kernel)
    float compression;
    unsigned int* prog;         // fused stage (tb_stage_kernel): frames finished by each 16-row scan workgroup of this block [n scan workgroups]
    int nscan;                  // fused stage: workgroups 0 .. nscan - 1 of the launch run the scan, the others the block's tiles
    unsigned long long* probe;  // FE_TB_PROBE builds (tools/gpu_tb_phases.py): cycles per phase of workgroup 0, [stage][kProbeSlots]; else nullptr
};

constexpr int kProbeSlots = 32;
enum { TB_ENC = 0, TB_SCAN = 1, TB_BLK = 2, TB_DEC = 3, TB_STAGE = 4 };     // (TB_STAGE: scan + block tiles of one block in one launch)
#ifdef FE_TB_PROBE
// phase clocks: workgroup 0's thread 0 adds the cycles since the previous mark to slot i (s_memtime, after the phase's barrier)
#define TB_PROBE_INIT(stage) unsigned long long* const tb_pr_ = (a.probe != nullptr && blockIdx.x == 0 && threadIdx.x == 0) ? a.probe + (stage) * kProbeSlots : nullptr; \
                             unsigned long long tb_pt_ = __builtin_amdgcn_s_memtime()
#define TB_MARK(i) do { if (tb_pr_ != nullptr) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tb_pr_[i] += n_ - tb_pt_; tb_pt_ = n_; } } while (0)
#else
#define TB_PROBE_INIT(stage) do {} while (0)
#define TB_MARK(i) do {} while (0)
#endif

// A buffer resource over [p, p + bytes): stores / loads beyond the range are dropped / return 0 in hardware - the last, partial tile of a
// launch needs no per-element predicate (hipcc turns those into one divergent block per store)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t range_rsrc(const float* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(bytes < 0x7fffffffu ? bytes : 0x7fffffffu), 0x00020000);
}
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, const f32x4& v, int voff_bytes) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), r, voff_bytes, 0, 0);
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, int voff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, 0, 0));
}
// float offset, within a tile's [frame][NCH][F2] region, of the four sub-bands that are tile rows row0 .. row0 + 3 (row0 % 4 == 0) of channel col
template <class S>
__device__ __forceinline__ int cm_off(int row0, int col, int nch) {
    const int fr = row0 / S::F2, f0 = row0 - fr * S::F2;
    return (fr * nch + col) * S::F2 + f0;
}
// a tile's tokens [frame][NCH][F2] (global, frames g0 .. g0 + nv - 1 valid: later frames repeat the last) -> LDS rows [frame * F2 + f][LD], 16-byte loads
template <class S, int FRAMES, int NCH, int LD>
__device__ __forceinline__ void load_tokens(const float* src, int nv, float* dst, int tid) {
    constexpr int F4 = S::F2 / 4, PER = NCH * F4;
    for (int i = tid; i < FRAMES * PER; i += kThreads) {
        const int fr = i / PER, q = i - fr * PER, c = q / F4, f4 = q - c * F4, fs = fr < nv ? fr : nv - 1;
        const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)fs * NCH + c) * S::F2 + 4 * f4);
        float* d = dst + (fr * S::F2 + 4 * f4) * LD + c;
        d[0] = v.x; d[LD] = v.y; d[2 * LD] = v.z; d[3 * LD] = v.w;
    }
}

template <class S>
__device__ __forceinline__ WSrc<false> make_wsrc(const float* wp, int lane) {
    WSrc<false> wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, Pack<S>::v.total * 4, 0x00020000);
    wb.lane4 = lane * 4;
    wb.li4 = (lane & 15) * 4;
    wb.lds = nullptr;
    wb.base = 0;
    wb.k4d = 0;
    return wb;
}

// ------------------------------------------------------------------------------------------ conv-layout GEMMs over FT frames
// Wave tiling: NS column groups x MS = 4 / NS row groups.  A wave's B fragments (weights, from L2) are private when NS = 4
// (column split: every weight fragment is fetched once per workgroup); with channel-tile counts that do not divide by 4
// (T, B: 3 tiles; M: 6) the rows are split instead and the waves re-read the (few) weight fragments.
// Row tile rt of the tile of frames = (frame rt / MTC, m-tile rt % MTC); wave (wm, wn) owns rt = wm + MS * i, i < MT.
template <class S, int FT>
struct ConvTiling {
    static constexpr int NS = (S::NTC % 4 == 0) ? 4 : ((S::NTC % 2 == 0) ? 2 : 1);
    static constexpr int MS = kWaves / NS;
    static constexpr int MT = FT * S::MTC / MS;          // row tiles per wave
    static constexpr int NTW = S::NTC / NS;              // column tiles per wave
    static_assert(S::MTC % MS == 0, "row groups must divide the m-tiles of a frame");
    __host__ __device__ static constexpr int frame(int i) { return (MS * i) / S::MTC; }
    __host__ __device__ static constexpr int mtile0(int i) { return (MS * i) % S::MTC; }      // + wm
};

// acc[i][j] = bias[column tile j of this wave]; A(i, ks) from `af`, B from the packed weights at w_off (KS k-steps per tile);
// epilogue: optional SiLU (scaled trunk), store to the activation buffers out[f] (row 1 + m: row 0 is the halo) and - gskip -
// to the frames' global skip slots in A-fragment order (one 16-byte store per lane and tile).
template <class S, int FT, int KS, bool ACT, int NCOLS, int LDO, int OSTR, int ROW0, class AF>
__device__ __forceinline__ void conv_gemm(AF&& af, const WSrc<false>& wb, int w_off, int b_off, float* out, float* gskip, size_t gskip_fstride,
                                          int nvalid, int wave, int lane) {
    using CT = ConvTiling<S, FT>;
    constexpr int MT = CT::MT, NTW = CT::NTW, NTALL = (NCOLS + 15) / 16;
    const int li = lane & 15, lg = lane >> 4;
    const int wn = CT::NS == 1 ? 0 : wave % CT::NS, wm = CT::NS == kWaves ? 0 : wave / CT::NS;
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        int nt = wn * NTW + j;
        nt = nt < NTALL ? nt : NTALL - 1;
        const f32x4 bj = b_off >= 0 ? wb.at16x4(b_off + nt * 64) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][j] = bj;
    }
    const int wbase = w_off + wn * NTW * KS * 64;
    constexpr int K4D = Pack<S>::v.conv_k4_delta;
    if constexpr (K4D != 0 && KS >= 4 && S::NTC >= 3) {       // (two channel tiles - T: measured -1.7 %, kept on the plain fetches)
        // four k-steps of a tile per 16-byte load from the k4-regrouped copy of the conv units (mma_panel asks for (j, ks) in rising ks per
        // tile: the load rides on the first k-step of each group of four, the other three take their lane's components)
        f32x4 cur[NTW];
        mma_panel<MT, NTW, KS, kPD>(acc, af, [&](int j, int ks) {
            int nt = j;                                                    // (clamped for the 16-column transposed-conv GEMM: NTALL = 1)
            if (NTALL < CT::NS * NTW) nt = (wn * NTW + j < NTALL) ? j : NTALL - 1 - wn * NTW;
            if (ks >= 4 * (KS / 4)) return wb.at_g(wbase + K4D + (nt * KS + ks) * 64);          // the ks % 4 remainder, plain
            if ((ks & 3) == 0) cur[j] = wb.at_gv4(wbase + K4D + nt * KS * 64 + (ks >> 2) * 256, wb.lane4 * 4);
            return cur[j][ks & 3];
        }, NoSide{});
    } else
    mma_panel<MT, NTW, KS, kPD>(acc, af, [&](int j, int ks) {
        int nt = j;                                                        // (clamped for the 16-column transposed-conv GEMM: NTALL = 1)
        if (NTALL < CT::NS * NTW) nt = (wn * NTW + j < NTALL) ? j : NTALL - 1 - wn * NTW;
        return wb.at_g(wbase + (nt * KS + ks) * 64);
    }, NoSide{});
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int f = CT::frame(i), mt = CT::mtile0(i) + wm;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int ntg = wn * NTW + j, col = 16 * ntg + li;
            if (ntg < NTALL && col < NCOLS) {
                f32x4 v = acc[i][j];
                if (ACT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = silu_scaled_f(v[r]);
                }
                float* od = out + f * OSTR + (ROW0 + 16 * mt + 4 * lg) * LDO + col;
#pragma unroll
                for (int r = 0; r < 4; ++r) od[r * LDO] = v[r];
                if (gskip != nullptr && f < nvalid)
                    *reinterpret_cast<f32x4*>(gskip + (size_t)f * gskip_fstride + (mt * S::KS_C + (col >> 2)) * 64 + (col & 3) * 16 + 4 * lg) = v;
            }
        }
    }
}

// A-fragment source of a k = 3 conv over the activation buffers `in` (halo rows 0 and F1 + 1): k-step ks = tap * KS_C + c4
template <class S, int FT>
struct K3Src {
    const float* base;          // in + (16 wm + li) * LDC + lg
    __device__ __forceinline__ float operator()(int i, int ks) const {
        using CT = ConvTiling<S, FT>;
        return base[CT::frame(i) * S::ACT + (16 * CT::mtile0(i) + ks / S::KS_C) * S::LDC + 4 * (ks % S::KS_C)];
    }
};

// ------------------------------------------------------------------------------------------ token-layout GEMMs over FT frames
// Rows = the FT * F2 tokens of the tile, dense (frame f, sub-band r -> row f * F2 + r): MTT = ceil(FT F2 / 16) row tiles, no
// per-frame padding.  Columns split over the waves (tile wave + 4 j): private weight fragments.
template <class S, int FT>
struct TokTiling {
    static constexpr int ROWS = FT * S::F2;
    static constexpr int MTT = (ROWS + 15) / 16;
    static constexpr int ROWS_P = MTT * 16 + 16;        // allocated rows: whole tiles + one tile of slack for the attention's padded key reads
};

// acc[i][j] += A (LDS rows, leading dimension LDA) x W[:, tile wave + 4 j]   (tiles beyond NT read a clamped tile: discarded)
// K4D != 0 (r4x; w_off inside the block-weight region, whose k4-regrouped copy sits K4D = PackedOffsets::k4_delta floats behind it): four k-steps per 16-byte load
template <int MTT, int NTPW, int KS, int LDA, int K4D = 0>
__device__ __forceinline__ void tok_panel(f32x4 (&acc)[MTT][NTPW], const float* a_lane, const WSrc<false>& wb, int w_off, int NT, int wave) {
    f32x4 cur[NTPW];
    mma_panel<MTT, NTPW, KS, kPD>(
        acc, [&](int i, int ks) { return a_lane[(16 * i) * LDA + 4 * ks]; },
        [&](int j, int ks) {
            int nt = wave + 4 * j;
            nt = nt < NT ? nt : NT - 1;
            if constexpr (K4D != 0 && FE_K4_STREAM && KS >= 4) {
                const int base = w_off + K4D + nt * (KS * 64);
                if (ks >= 4 * (KS / 4)) return wb.at_g(base + ks * 64);
                if ((ks & 3) == 0) cur[j] = wb.at_gv4(base + (ks >> 2) * 256, wb.lane4 * 4);
                return cur[j][ks & 3];
            } else return wb.at_g(w_off + (nt * KS + ks) * 64);
        }, NoSide{});
}

// The same panel with the B operand (weights) REGISTER-RESIDENT - w[j][ks], loaded once per launch: nothing but the A fragments
// moves during the GEMM (LDS reads, ring of 3 k-steps), no L2 round trip at the head of the phase.
template <int MTP, int NTP, int KS, int LDA, typename WF>
__device__ __forceinline__ void tok_panel_rb(f32x4 (&acc)[MTP][NTP], const float* a_lane, WF&& wf) {
    constexpr int PD = KS < 3 ? KS : 3;
    float a[PD][MTP];
#pragma unroll
    for (int ks = 0; ks < PD; ++ks)
#pragma unroll
        for (int i = 0; i < MTP; ++i) a[ks][i] = a_lane[(16 * i) * LDA + 4 * ks];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float av[MTP];
#pragma unroll
        for (int i = 0; i < MTP; ++i) av[i] = a[ks % PD][i];
        if (ks + PD < KS) {
#pragma unroll
            for (int i = 0; i < MTP; ++i) a[ks % PD][i] = a_lane[(16 * i) * LDA + 4 * (ks + PD)];
        }
#pragma unroll
        for (int i = 0; i < MTP; ++i)
#pragma unroll
            for (int j = 0; j < NTP; ++j) acc[i][j] = FE_MFMA(av[i], wf(j, ks), acc[i][j]);
    }
    constexpr int NM = MTP * NTP;
    __builtin_amdgcn_sched_group_barrier(0x100, PD * MTP, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (ks + PD < KS) {
#pragma unroll
                for (int q = (m * MTP) / NM; q < ((m + 1) * MTP) / NM; ++q) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
}

// gx is stored SCALED for the scans (fe_kernels.hip.h, kGateRZ / kGateN): the r and z columns by -log2 e, the n columns by 2 log2 e
template <class S>
__device__ __forceinline__ float gx_scale(int col) { return col < 2 * S::C2 ? kGateRZ : kGateN; }

// gx = (x W_ih^T + b) * gx_scale for one block and direction -> global [frame][3 C2][F2]; X: the tile's tokens in LDS
template <class S, int FT>
__device__ __forceinline__ void gx_gemm(const float* Xb, const WSrc<false>& wb, int w_off, int b_off, float* gx_tile, int rows_valid, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t gr = range_rsrc(gx_tile, (size_t)rows_valid * S::N3 * 4);
    using TT = TokTiling<S, FT>;
    constexpr int MTT = TT::MTT, NT3 = S::NT3, N3 = S::N3;
    constexpr int NTPW = ceil_div(NT3, kWaves);
    // column tiles in chunks so that the accumulators stay within ~32 tiles per wave
    constexpr int CH = (MTT * NTPW <= 32) ? NTPW : (32 / MTT > 0 ? 32 / MTT : 1);
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll 1
    for (int j0 = 0; j0 < NTPW; j0 += CH) {
        f32x4 acc[MTT][CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            int nt = wave + 4 * (j0 + j);
            nt = nt < NT3 ? nt : NT3 - 1;
            const float bj = wb.at16_g(b_off + nt * 16);
#pragma unroll
            for (int i = 0; i < MTT; ++i) acc[i][j] = f32x4{bj, bj, bj, bj};
        }
        mma_panel<MTT, CH, S::KS_2, kPD>(
            acc, [&](int i, int ks) { return Xb[(16 * i + li) * S::LDX + lg + 4 * ks]; },
            [&](int j, int ks) {
                int nt = wave + 4 * (j0 + j);
                nt = nt < NT3 ? nt : NT3 - 1;
                return wb.at_g(w_off + (nt * S::KS_2 + ks) * 64);
            }, NoSide{});
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int nt = wave + 4 * (j0 + j), col = 16 * nt + li;
            if (nt < NT3 && col < N3) {
                const float gsc = gx_scale<S>(col);
#pragma unroll
                for (int i = 0; i < MTT; ++i) bstore4(gr, acc[i][j] * gsc, cm_off<S>(16 * i + 4 * lg, col, N3) * 4);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ encoder segment
template <class S, int FT>
struct EncLds {
    using TT = TokTiling<S, FT>;
    static constexpr int TW = 0;                                   // twiddles float2[N/2]
    static constexpr int SC = TW + S::NFFT;                        // compressed spectrum [FT][2][LDS_S]
    static constexpr int A0 = SC + FT * 2 * S::LDS_S;              // activation ping-pong [FT][ACT] x 2
    static constexpr int A1 = A0 + FT * S::ACT;
    static constexpr int TOTAL = A1 + FT * S::ACT;
    // aliases: the FFT scratch (windowed frame | spectrum, 2 N floats) lies in A1 (not live before encoder layer 0); the rf_pre
    // intermediate Y1 [FT F2][LDC] in the ping-pong buffer the last encoder layer did NOT write, the tokens X [16 MTT][LDX] in the other
    // one (its content - the last encoder output - is dead once the filterbank has run)
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
    static constexpr bool OK = 2 * S::NFFT <= FT * S::ACT && TT::MTT * 16 * S::LDX <= FT * S::ACT && TT::MTT * 16 * S::LDC <= FT * S::ACT && BYTES <= 160 * 1024;
};

template <class S, int FT>
__global__ void __launch_bounds__(kThreads) tb_enc_kernel(TbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = EncLds<S, FT>;
    using CT = ConvTiling<S, FT>;
    using TT = TokTiling<S, FT>;
    using D = Dft<S, kPD>;
    constexpr int N = S::NFFT, H = S::HOP, F0 = S::F0, F1 = S::F1, C1 = S::C1, C2 = S::C2, F2 = S::F2;
    constexpr int LDC = S::LDC, LDX = S::LDX;
    constexpr PackedOffsets o = Pack<S>::v;
    const int tid = threadIdx.x, lane = tid & 63, wave_k = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const WSrc<false> wb = make_wsrc<S>(a.wp, lane);
    const int wave = wave_k, wm = CT::NS == kWaves ? 0 : wave / CT::NS;

    float2* tw = reinterpret_cast<float2*>(smem + L::TW);
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = reinterpret_cast<const float2*>(a.wp + o.twiddle)[i];
    for (int i = tid; i < FT * 2 * S::LDS_S; i += kThreads) smem[L::SC + i] = 0.0f;       // (the 2-bin halos stay zero)
    float* const A0 = smem + L::A0;
    float* const A1 = smem + L::A1;
    float* const q0 = A1;             // windowed frame
    float* const q3 = A1 + N;         // spectrum {Re[N/2], Im[N/2]}
    typename D::FwdConst dc;
    D::load(dc, wb, o, wave);
    const int ntiles = (a.NF + FT - 1) / FT;
    // analysis window in registers; the samples of the tile's frames are fetched one tile ahead (fetch_frames)
    constexpr int NPT = N / kThreads;
    constexpr bool PARF = FT * 2 * N <= FT * S::ACT;          // every frame of the tile has its own transform scratch in A1
    float fw[NPT], fv[FT][NPT];
#pragma unroll
    for (int q = 0; q < NPT; ++q) fw[q] = (a.wp + o.window)[tid + q * kThreads];
    auto fetch_frames = [&](int tl) {
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            int g = tl * FT + f;
            g = g < a.NF ? g : a.NF - 1;
            const int bl = g / a.T, t = a.t0 + (g - bl * a.T), b = a.b0 + bl;
            const float* xin = a.wav_in + (size_t)b * a.in_stride;
            const int Twb = a.Tw_b != nullptr ? a.Tw_b[b] : a.Tw;
#pragma unroll
            for (int q = 0; q < NPT; ++q) {
                const int n = tid + q * kThreads;
                int idx = t * H + n - N / 2;
                idx = idx < 0 ? -idx : idx;
                idx = idx >= Twb ? 2 * (Twb - 1) - idx : idx;
                idx = idx < 0 ? 0 : idx;             // (ragged batch: a frame past the utterance's end)
                fv[f][q] = xin[idx];
            }
        }
    };
    if (PARF && a.mode != FE_MODE_SPEC && (int)blockIdx.x < ntiles) fetch_frames(blockIdx.x);
    __syncthreads();
    TB_PROBE_INIT(TB_ENC);
    TB_MARK(0);                     // prologue

#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // big shapes (C1 > 96): a loop-variant zero in the wave index keeps the (hundreds of) wave-uniform weight offsets of a tile from
        // being hoisted out of the tile loop, where they sat in SGPRs for the whole kernel and spilled to VGPR lanes (tb_enc_kernel<L>:
        // 961 SGPR spills - v_writelane / v_readlane traffic on the datapath the fp32 MFMAs share with the vector ALUs)
        int lz = 0;
        if constexpr (S::C1 >= 96) asm volatile("" : "+s"(lz));
        const int wave = wave_k + lz, wm = CT::NS == kWaves ? 0 : wave / CT::NS;
        const int g0 = tile * FT;
        const int nvalid = a.NF - g0 < FT ? a.NF - g0 : FT;
        // ---- STFT (functional/audio_modules.py:78-80: center = True, reflect padding) + compress (model.py:684-690).  The tile's samples
        // were fetched while the previous tile computed (fv), the window sits in registers; the frames' transforms run back to back,
        // each in its own scratch (windowed frame | spectrum, 2 N floats per frame in A1)
        if (a.mode != FE_MODE_SPEC) {
            if constexpr (PARF) {
#pragma unroll
                for (int f = 0; f < FT; ++f)
#pragma unroll
                    for (int q = 0; q < NPT; ++q) A1[f * 2 * N + tid + q * kThreads] = fv[f][q] * fw[q];
                __syncthreads();
                if (tile + (int)gridDim.x < ntiles) fetch_frames(tile + gridDim.x);       // in flight under this tile's GEMMs
#pragma unroll 1
                for (int f = 0; f < FT; ++f) D::forward(A1 + f * 2 * N, A1 + f * 2 * N + N, tw, dc, wave, lane, nullptr);          // (each ends with a barrier)
                for (int i = tid; i < FT * F0; i += kThreads) {
                    const int f = i / F0, fb = i - f * F0;
                    const float* q3 = A1 + f * 2 * N + N;
                    float* sc = smem + L::SC + f * 2 * S::LDS_S;
                    const float re = q3[fb], im = q3[N / 2 + fb];
                    const float mag = fmaxf(sqrtf(re * re + im * im), 1.0e-5f);
                    const float gn = pow_f(mag, a.compression - 1.0f);
                    sc[2 + fb] = re * gn;
                    sc[S::LDS_S + 2 + fb] = im * gn;
                    if (f < nvalid) {
                        float* xcg = a.xc + (size_t)(g0 + f) * (2 * F0);
                        xcg[fb] = re * gn; xcg[F0 + fb] = im * gn;
                    }
                }
                __syncthreads();                                               // (encoder layer 0 writes A1)
            } else {
#pragma unroll 1
                for (int f = 0; f < FT; ++f) {
                    const int g = g0 + f < a.NF ? g0 + f : a.NF - 1;
                    const int bl = g / a.T, t = a.t0 + (g - bl * a.T), b = a.b0 + bl;
                    float* sc = smem + L::SC + f * 2 * S::LDS_S;
                    float* xcg = a.xc + (size_t)g * (2 * F0);
                    const float* xin = a.wav_in + (size_t)b * a.in_stride;
                    const int Twb = a.Tw_b != nullptr ? a.Tw_b[b] : a.Tw;
                    float fr[NPT];
#pragma unroll
                    for (int q = 0; q < NPT; ++q) {
                        const int n = tid + q * kThreads;
                        int idx = t * H + n - N / 2;
                        idx = idx < 0 ? -idx : idx;
                        idx = idx >= Twb ? 2 * (Twb - 1) - idx : idx;
                        idx = idx < 0 ? 0 : idx;
                        fr[q] = xin[idx];
                    }
#pragma unroll
                    for (int q = 0; q < NPT; ++q) q0[tid + q * kThreads] = fr[q] * fw[q];
                    __syncthreads();
                    D::forward(q0, q3, tw, dc, wave, lane, nullptr);          // (ends with a barrier)
                    for (int fb = tid; fb < F0; fb += kThreads) {
                        const float re = q3[fb], im = q3[N / 2 + fb];
                        const float mag = fmaxf(sqrtf(re * re + im * im), 1.0e-5f);
                        const float gn = pow_f(mag, a.compression - 1.0f);
                        sc[2 + fb] = re * gn;
                        sc[S::LDS_S + 2 + fb] = im * gn;
                        if (f < nvalid) { xcg[fb] = re * gn; xcg[F0 + fb] = im * gn; }
                    }
                    __syncthreads();                                           // (q0 / q3 are re-used by the next frame)
                }
            }
        } else {
#pragma unroll 1
            for (int f = 0; f < FT; ++f) {
                const int g = g0 + f < a.NF ? g0 + f : a.NF - 1;
                const int bl = g / a.T, t = a.t0 + (g - bl * a.T), b = a.b0 + bl;
                float* sc = smem + L::SC + f * 2 * S::LDS_S;
                float* xcg = a.xc + (size_t)g * (2 * F0);
                const float* sp = a.spec_in + (size_t)b * (F0 + 1) * a.Tfull * 2;
                for (int fb = tid; fb < F0; fb += kThreads) {
                    const float re = sp[((size_t)fb * a.Tfull + t) * 2], im = sp[((size_t)fb * a.Tfull + t) * 2 + 1];
                    const float mag = fmaxf(sqrtf(re * re + im * im), 1.0e-5f);
                    const float gn = pow_f(mag, a.compression - 1.0f);
                    sc[2 + fb] = re * gn;
                    sc[S::LDS_S + 2 + fb] = im * gn;
                    if (f < nvalid) { xcg[fb] = re * gn; xcg[F0 + fb] = im * gn; }
                }
            }
        }
        // zero halo rows of both ping-pong buffers (A1 held the FFT scratch, A0 the previous tile's tokens)
        for (int i = tid; i < FT * 4 * LDC; i += kThreads) {
            const int f = i / (4 * LDC), q = (i / LDC) & 3, c = i % LDC;
            ((q & 2) ? A1 : A0)[f * S::ACT + ((q & 1) ? F1 + 1 : 0) * LDC + c] = 0.0f;
        }
        __syncthreads();
        TB_MARK(1);                 // STFT + compress
        float* const skip_tile = a.skip + (size_t)g0 * ((S::NL + 1) * F1 * C1);
        constexpr size_t SKF = (size_t)(S::NL + 1) * F1 * C1;
        // ---- enc_pre (model.py:436-443): strided conv as a K = 16 GEMM over the 2-bin-haloed spectrum
        {
            const float* scb = smem + L::SC;
            conv_gemm<S, FT, 4, true, C1, LDC, S::ACT, 1>(
                [&](int i, int ks) {
                    const int kk = 4 * ks + lg, c = kk & 1, s = (kk >> 1) & 3, tp = kk >> 3;
                    const int m = 16 * (CT::mtile0(i) + wm) + li;
                    return scb[CT::frame(i) * 2 * S::LDS_S + c * S::LDS_S + 4 * (m + tp) + s];
                }, wb, o.enc_pre_w, o.enc_pre_b, A0, skip_tile, SKF, nvalid, wave, lane);
        }
        __syncthreads();
        TB_MARK(2);                 // enc_pre
        // ---- encoder (model.py:446-456): k = 3 convs, ping-pong A0 <-> A1
        static_for<S::NL>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            const float* in = (l & 1) ? A1 : A0;
            float* out = (l & 1) ? A0 : A1;
            conv_gemm<S, FT, 3 * S::KS_C, true, C1, LDC, S::ACT, 1>(K3Src<S, FT>{in + (16 * wm + li) * LDC + lg}, wb, o.enc_w[l], o.enc_b[l], out,
                                                                     skip_tile + (size_t)(l + 1) * F1 * C1, SKF, nvalid, wave, lane);
            __syncthreads();
            TB_MARK(3 + l);         // encoder layer l
        });
        float* const Elast = (S::NL & 1) ? A1 : A0;      // last encoder output
        float* const Y1 = (S::NL & 1) ? A0 : A1;         // rf_pre intermediate [FT F2][LDC] (dense rows)
        float* const Xb = Elast;                         // tokens [ROWS_P][LDX] (after the filterbank)
        // ---- rf_pre (model.py:458-465): Linear over the frequency axis - A = the packed filterbank [F2][F1], B = the frames' last
        // encoder outputs, the frames of the tile side by side along N - then the 1x1 conv over the dense token rows
        {
            constexpr int NTPW = ceil_div(S::NTC, kWaves), KS = F1 / 4;
            f32x4 acc[S::MT2][FT * NTPW];
            acc_init_zero<S::MT2, FT * NTPW>(acc);
            const float* Ein = Elast + LDC;                  // row 0 = bin 0
            mma_panel<S::MT2, FT * NTPW, KS, kPD>(
                acc, [&](int i, int ks) { return wb.at_g(o.rfpre_lin + (i * KS + ks) * 64); },
                [&](int j, int ks) {
                    int nt = wave + 4 * (j % NTPW);
                    nt = nt < S::NTC ? nt : S::NTC - 1;
                    return Ein[(j / NTPW) * S::ACT + (4 * ks + lg) * LDC + 16 * nt + li];
                }, NoSide{});
#pragma unroll
            for (int i = 0; i < S::MT2; ++i)
#pragma unroll
                for (int j = 0; j < FT * NTPW; ++j) {
                    const int nt = wave + 4 * (j % NTPW), col = 16 * nt + li, f = j / NTPW;
                    if (nt < S::NTC && col < C1 && 16 * i + 4 * lg < F2) {        // (F2 % 4 == 0: a lane's four rows are valid together)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Y1[(f * F2 + 16 * i + 4 * lg + r) * LDC + col] = acc[i][j][r];
                    }
                }
        }
        __syncthreads();
        TB_MARK(12);                // rf_pre filterbank
        {
            constexpr int NTPW = ceil_div(S::NT2, kWaves);
            f32x4 acc[TT::MTT][NTPW];
            acc_init_bias<TT::MTT, NTPW>(acc, wb, o.rfpre_b, wave, 4, S::NT2);
            tok_panel<TT::MTT, NTPW, S::KS_C, LDC>(acc, Y1 + li * LDC + lg, wb, o.rfpre_w, S::NT2, wave);
            const __amdgpu_buffer_rsrc_t xr = range_rsrc(a.x + (size_t)g0 * (F2 * C2), (size_t)nvalid * F2 * C2 * 4);
#pragma unroll
            for (int i = 0; i < TT::MTT; ++i)
#pragma unroll
                for (int j = 0; j < NTPW; ++j) {
                    const int nt = wave + 4 * j, col = 16 * nt + li;
                    if (nt < S::NT2 && col < C2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Xb[(16 * i + 4 * lg + r) * LDX + col] = acc[i][j][r];
                        bstore4(xr, acc[i][j], cm_off<S>(16 * i + 4 * lg, col, C2) * 4);
                    }
                }
        }
        __syncthreads();
        TB_MARK(13);                // rf_pre 1x1
        // ---- the x half of block 0's GRU gates, every direction
#pragma unroll 1
        for (int d = 0; d < S::ND; ++d)
            gx_gemm<S, FT>(Xb, wb, o.tb_wih[0][0] + d * (o.tb_wih[0][S::ND - 1] - o.tb_wih[0][0]), o.tb_bx[0][0] + d * (o.tb_bx[0][S::ND - 1] - o.tb_bx[0][0]),
                           a.gx + ((size_t)d * a.NF + g0) * (F2 * S::N3), nvalid * F2, wave, lane);
        __syncthreads();                                     // (the next tile's FFT overwrites A1, its encoder A0)
        TB_MARK(14);                // gx of block 0
    }
}

// ------------------------------------------------------------------------------------------ GRU scan over time
// One workgroup = 16 (utterance, sub-band) rows of one direction; wave w owns the channel tiles w, w + 4, ... with the three
// gates' hidden weights of those channels in registers for the whole scan.  Per step only h W_hh^T is computed (the x half,
// gx, was batched over all frames): A = h_{t-1} from LDS (double-buffered: one barrier per step), r / z / n of a (row, channel)
// land in the same lane, the gate math runs in the epilogue, h_t goes to LDS (next step's A operand), to the registers (next
// step's z h term) and to hs in global memory.  gx of step t + 1 is fetched while step t computes.
// PUB > 0 (the fused stage): every PUB steps - and after the last one - the workgroup publishes how many frames of its 16 rows are in hs.
// hs is written with agent-scope stores (they go to the coherence point, not into this XCD's L2); every wave drains its stores
// (vmcnt(0)), the step's barrier, then one agent-scope store of the counter.  The consumer polls the counter and reads hs with
// agent-scope loads as well.  No release / acquire FENCE on either side: measured, one L2 write-back per publish and one cache
// invalidate per tile made the fused stage 3x slower than the two launches it replaces (profiles/r3g_tb_fused_stage.txt).
constexpr int kScanPub = 32;
constexpr int kProgStride = 32;      // ints between two scan workgroups' counters: a 128-byte line each (pollers and publisher meet on one line only)
template <class S, int PUB>
__device__ __forceinline__ void scan_role(const TbArgs& a, int rg, int dir, float (*hbuf)[16 * S::LDX]) {
    constexpr int C2 = S::C2, F2 = S::F2, KS = S::KS_2, NT2 = S::NT2, LDX = S::LDX, N3 = S::N3;
    constexpr int NTPW = ceil_div(NT2, kWaves);
    constexpr int HW = S::ND * C2;                           // hs row width
    constexpr PackedOffsets o = Pack<S>::v;
    __builtin_amdgcn_s_setprio(3);      // a latency chain next to GEMM passes: its few instructions go first
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const WSrc<false> wb = make_wsrc<S>(a.wp, lane);
    const int k = a.k;
    const int R = a.B * F2, r0 = rg * 16;
    const int w_off = o.tb_whh[0][0] + k * (S::KB > 1 ? o.tb_whh[1][0] - o.tb_whh[0][0] : 0) + dir * (o.tb_whh[0][S::ND - 1] - o.tb_whh[0][0]);
    const int bn_off = o.tb_bhn[0][0] + k * (S::KB > 1 ? o.tb_bhn[1][0] - o.tb_bhn[0][0] : 0) + dir * (o.tb_bhn[0][S::ND - 1] - o.tb_bhn[0][0]);
    // hidden weights: tile (gate g, channel tile ct) at (g * NT2 + ct) * KS fragments
    float whh[NTPW][3][KS], bhn[NTPW];
    bool live[NTPW];
#pragma unroll
    for (int j = 0; j < NTPW; ++j) {
        const int ct = wave + 4 * j;
        live[j] = ct < NT2;
        const int ctc = live[j] ? ct : NT2 - 1;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) whh[j][g][ks] = wb.at_g(w_off + ((g * NT2 + ctc) * KS + ks) * 64) * (g < 2 ? kGateRZ : kGateN);
        bhn[j] = wb.at16_g(bn_off + ctc * 16) * kGateN;          // (scaled pre-activations, like gx: see gx_scale)
    }
    // this lane's four rows (C/D layout: rows 4 lg + r - four consecutive sub-bands of one utterance: R % 4 == 0) and columns 16 ct + li
    bool rok[4];
    int tb_last = 0x7fffffff;      // (noncausal, ragged batch) last frame of this lane's utterance
    size_t grow, hrow;             // element offsets of (the four rows, t = 0, channel 0) in gx / hs
    {
        int row = r0 + 4 * lg;
        const bool ok = row < R;
        row = ok ? row : R - 4;
        const int b = row / F2, f = row - b * F2;
        grow = (size_t)b * a.T * N3 * F2 + f;
        hrow = ((size_t)b * a.T * HW + dir * C2) * F2 + f;
#pragma unroll
        for (int r = 0; r < 4; ++r) rok[r] = ok;
        // ragged batch, reverse direction: the utterance of this lane's rows ends at frame tb_last - the steps before it (t > tb_last)
        // leave h at its zero initial state
        if constexpr (S::BIDIR) { if (a.Tw_b != nullptr) tb_last = a.Tw_b[a.b0 + b] / S::HOP; }
    }
    const float* gxd = a.gx + (size_t)dir * a.NF * F2 * N3;
    // carried state [KB][ND][Bfull * F2][C2]: this node's rows start at utterance b0
    float* hst = a.hstate ? a.hstate + (((size_t)(k * S::ND + dir) * a.Bfull + a.b0) * F2) * C2 : nullptr;
    const bool hinit = a.h_init != 0;
    float hprev[NTPW][4];
#pragma unroll
    for (int j = 0; j < NTPW; ++j) {
        const int col = 16 * (wave + 4 * j) + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + 4 * lg + r;
            float v = 0.0f;
            if (hst != nullptr && hinit && live[j] && col < C2 && rok[r]) v = hst[(size_t)row * C2 + col];
            hprev[j][r] = v;
            if (live[j] && col < C2) hbuf[0][(4 * lg + r) * LDX + col] = v;
        }
    }
    // gx of the steps ahead: a ring of GD steps in registers.  gx does not fit in L2 for a real batch (10 KB per frame): with ONE step of
    // prefetch every step waited for an HBM round trip - all scan variants ran at the same ~0.9 us per step whatever their arithmetic.
    // The step is a latency chain on a lone wave per SIMD - it costs its instruction count - so: the r / z parts of gx are the MFMA
    // accumulators' initial values (no add), everything is pre-scaled for exp2 (no multiply), a slot is refilled AFTER its step has
    // used it (no register copies), gx / hs are walked with running pointers, h goes to LDS unpredicated (idle lanes aim at the pad column).
    constexpr int GD = (NTPW * 12 <= 24) ? 4 : 2;
    const int t_first = dir ? a.T - 1 : 0, dt = dir ? -1 : 1;
    f32x4 gxv[GD][NTPW][3];
    const float* gxp[NTPW];
    float* hsp[NTPW];
    int hcol[NTPW];
    bool cokj[NTPW];
#pragma unroll
    for (int j = 0; j < NTPW; ++j) {
        const int col = 16 * (wave + 4 * j) + li;
        cokj[j] = live[j] && col < C2;
        hcol[j] = cokj[j] ? col : C2;                                           // LDS column (C2 = the pad column of the h tile)
        gxp[j] = gxd + grow + (size_t)t_first * F2 * N3 + (size_t)(cokj[j] ? col : 0) * F2;
        hsp[j] = a.hs + hrow + (size_t)t_first * F2 * HW + (size_t)(cokj[j] ? col : 0) * F2;
    }
    const ptrdiff_t gstep = (ptrdiff_t)dt * F2 * N3, hstep = (ptrdiff_t)dt * F2 * HW;
    auto load_gx = [&](int slot) {             // the next unfetched step of every column tile; the pointers move on
#pragma unroll
        for (int j = 0; j < NTPW; ++j) {
#pragma unroll
            for (int g = 0; g < 3; ++g) gxv[slot][j][g] = *reinterpret_cast<const f32x4*>(gxp[j] + (size_t)g * C2 * F2);
            gxp[j] += gstep;
        }
    };
#pragma unroll
    for (int d = 0; d < GD; ++d)
        if (d < a.T) load_gx(d);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int st0 = 0; st0 < a.T; st0 += GD) {
#pragma unroll
    for (int d = 0; d < GD; ++d) {
        const int st = st0 + d;
        if (st >= a.T) break;
        const float* hc = hbuf[cur] + li * LDX + lg;
        float av[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) av[ks] = hc[4 * ks];
        f32x4 acc[NTPW][3];
#pragma unroll
        for (int j = 0; j < NTPW; ++j) {
            acc[j][0] = gxv[d][j][0];
            acc[j][1] = gxv[d][j][1];
            acc[j][2] = f32x4{bhn[j], bhn[j], bhn[j], bhn[j]};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < NTPW; ++j)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[j][g] = FE_MFMA(av[ks], whh[j][g][ks], acc[j][g]);
        float* hn = hbuf[cur ^ 1];
#pragma unroll
        for (int j = 0; j < NTPW; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rr = sigmoid_pre(acc[j][0][r]);
                const float zz = sigmoid_pre(acc[j][1][r]);
                const float nn = tanh_pre(__builtin_fmaf(rr, acc[j][2][r], gxv[d][j][2][r]));
                float hv = __builtin_fmaf(zz, hprev[j][r] - nn, nn);                // (1 - z) n + z h
                if constexpr (S::BIDIR) { if (dir && t_first - st > tb_last) hv = 0.0f; }
                hprev[j][r] = hv;
                hn[(4 * lg + r) * LDX + hcol[j]] = hv;
            }
            if (cokj[j] && rok[0]) {
                float* hd = hsp[j];
                if constexpr (PUB > 0) {            // (read by other workgroups of the SAME launch: agent-scope accesses on both sides, no cache in between)
                    // one 16-byte store with the system-coherent bits (what a relaxed agent-scope atomic store of each float would set)
                    const f32x4 hv4 = {hprev[j][0], hprev[j][1], hprev[j][2], hprev[j][3]};
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(hd), "v"(hv4) : "memory");
                } else
                *reinterpret_cast<float4*>(hd) = make_float4(hprev[j][0], hprev[j][1], hprev[j][2], hprev[j][3]);
            }
            hsp[j] += hstep;
        }
        if (st + GD < a.T) load_gx(d);          // the x half GD steps ahead, into the slot this step has just used up
        cur ^= 1;
        if constexpr (PUB > 0) {
            const bool pub = ((st + 1) % PUB == 0) || st + 1 == a.T;
            if (pub) __builtin_amdgcn_s_waitcnt(0x0f70);         // vmcnt(0): this wave's hs stores have left the CU
            __syncthreads();
            if (pub && tid == 0) __hip_atomic_store(a.prog + rg * kProgStride, (unsigned int)(st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
        __syncthreads();
    }
    }
    if (hst != nullptr) {
#pragma unroll
        for (int j = 0; j < NTPW; ++j) {
            const int col = 16 * (wave + 4 * j) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (live[j] && col < C2 && rok[r]) hst[(size_t)(r0 + 4 * lg + r) * C2 + col] = hprev[j][r];
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

template <class S>
__global__ void __launch_bounds__(kThreads) tb_scan_kernel(TbArgs a) {
    __shared__ float hbuf[2][16 * S::LDX];
    scan_role<S, 0>(a, blockIdx.x, blockIdx.y, hbuf);
}

// The same recurrence for FEW rows (a handful of utterances: 16 rows per workgroup leave most of the chip idle and each step is a
// chain of 27+ dependent 32-cycle MFMAs): FOUR (utterance, sub-band) rows per workgroup on v_mfma_f32_4x4x1_16B_f32 - sixteen
// independent 4 x 4 outer products per instruction, 8 cycles.  A wave owns C2 / 4 channels of all three gates; its 16 blocks are
// (channel block cb) x (K quarter q): lane = 16 cb + 4 q + j supplies h[row j][k] as A and W_g[k][channel 4 cb + j] as B for the k
// of quarter q, so a gate is C2 / 4 instructions deep instead of C2 / 4 x 4 MFMA tiles wide.  The four quarter sums of a (row,
// channel) sit in the four lanes q of a DPP row: two rotate-adds leave every lane with all four rows' totals, lane q keeps row q -
// the gate math then runs on ONE (row, channel) per lane instead of four.  4x the workgroups, ~2.3x shorter steps.
template <class S>
__global__ void __launch_bounds__(kThreads) tb_scan4_kernel(TbArgs a) {
    constexpr int C2 = S::C2, F2 = S::F2, KS = S::KS_2, NT2 = S::NT2, N3 = S::N3;
    constexpr int HW = S::ND * C2;
    constexpr int KQ = C2 / 4;                               // instructions per gate: k of one quarter
    constexpr int KP = (KQ + 3) & ~3;                        // LDS stride of a quarter (16-byte reads)
    constexpr int LDR = 4 * KP + 4;                          // LDS row stride
    constexpr int CW = C2 / kWaves;                          // channels per wave
    constexpr int NCB = ceil_div(CW, 4), NSET = ceil_div(NCB, 4);
    static_assert(C2 % kWaves == 0, "tb_scan4_kernel: rnnformer channels must be a multiple of 4");
    constexpr PackedOffsets o = Pack<S>::v;
    __shared__ __attribute__((aligned(16))) float hbuf[2][4 * LDR];
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = lane >> 4, q = (lane >> 2) & 3, j = lane & 3;
    const WSrc<false> wb = make_wsrc<S>(a.wp, lane);
    const int dir = blockIdx.y, k = a.k;
    const int gsz = NT2 * KS * 64;
    const int w_off = o.tb_whh[0][0] + k * (S::KB > 1 ? o.tb_whh[1][0] - o.tb_whh[0][0] : 0) + dir * (o.tb_whh[0][S::ND - 1] - o.tb_whh[0][0]);
    const int bn_off = o.tb_bhn[0][0] + k * (S::KB > 1 ? o.tb_bhn[1][0] - o.tb_bhn[0][0] : 0) + dir * (o.tb_bhn[0][S::ND - 1] - o.tb_bhn[0][0]);
    // this lane's channel per set, its hidden weights of the quarter's k (gathered out of the 16x16x4 fragment order, once) and b_hn
    int ch[NSET];
    bool ok[NSET];
    float w[NSET][3][KQ], bhn[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
        const int cl = 4 * (4 * s + cb) + j;
        ok[s] = cl < CW;
        ch[s] = wave * CW + (ok[s] ? cl : 0);
        const int c = ch[s];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int kx = q * KQ + kk;
                w[s][g][kk] = wb.gather_g(w_off + g * gsz + ((c >> 4) * KS + (kx >> 2)) * 64 + (kx & 3) * 16 + (c & 15)) * (g < 2 ? kGateRZ : kGateN);
            }
        bhn[s] = wb.gather_g(bn_off + c) * kGateN;               // (scaled pre-activations, like gx: see gx_scale)
    }
    // this lane's (row, channel): row = 4 blockIdx.x + q
    const int row = blockIdx.x * 4 + q;
    const int b = row / F2, f = row - b * F2;
    const size_t grow = (size_t)b * a.T * N3 * F2 + f;                          // gx / hs element offsets of (row, t = 0, channel 0)
    const size_t hrow = ((size_t)b * a.T * HW + dir * C2) * F2 + f;
    const float* gxd = a.gx + (size_t)dir * a.NF * F2 * N3;
    float* hst = a.hstate ? a.hstate + (((size_t)(k * S::ND + dir) * a.Bfull + a.b0) * F2) * C2 : nullptr;
    int tb_last = 0x7fffffff;      // (noncausal, ragged batch) last frame of this row's utterance: see tb_scan_kernel
    if constexpr (S::BIDIR) { if (a.Tw_b != nullptr) tb_last = a.Tw_b[a.b0 + b] / S::HOP; }
    float hprev[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
        float v = 0.0f;
        if (hst != nullptr && a.h_init != 0 && ok[s]) v = hst[(size_t)row * C2 + ch[s]];
        hprev[s] = v;
        if (ok[s]) hbuf[0][q * LDR + (ch[s] / KQ) * KP + ch[s] % KQ] = v;
    }
    // (the step costs its instruction count - see scan_role: running pointers, scaled pre-activations, the slot refilled after its use)
    const int t_first = dir ? a.T - 1 : 0, dt = dir ? -1 : 1;
    float gxv[NSET][3];
    const float* gxp[NSET];
    float* hsp[NSET];
    int hoff[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
        gxp[s] = gxd + grow + (size_t)t_first * F2 * N3 + (size_t)ch[s] * F2;
        hsp[s] = a.hs + hrow + (size_t)t_first * F2 * HW + (size_t)ch[s] * F2;
        hoff[s] = q * LDR + (ch[s] / KQ) * KP + ch[s] % KQ;
    }
    const ptrdiff_t gstep = (ptrdiff_t)dt * F2 * N3, hstep = (ptrdiff_t)dt * F2 * HW;
    auto load_gx = [&]() {
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
#pragma unroll
            for (int g = 0; g < 3; ++g) gxv[s][g] = gxp[s][(size_t)g * C2 * F2];
            gxp[s] += gstep;
        }
    };
    load_gx();
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int st = 0; st < a.T; ++st) {
        // A fragments: h[row j][quarter q]
        float ha[KP];
        {
            const float* hc = hbuf[cur] + j * LDR + q * KP;
#pragma unroll
            for (int i = 0; i < KP / 4; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(hc + 4 * i);
                ha[4 * i] = v.x; ha[4 * i + 1] = v.y; ha[4 * i + 2] = v.z; ha[4 * i + 3] = v.w;
            }
        }
        f32x4 acc[NSET][3];
#pragma unroll
        for (int s = 0; s < NSET; ++s)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[s][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk)
#pragma unroll
            for (int s = 0; s < NSET; ++s)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[s][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(ha[kk], w[s][g][kk], acc[s][g], 0, 0, 0);
        float* hn = hbuf[cur ^ 1];
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
            float tot[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = acc[s][g][i];
                    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));   // row_ror:8
                    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));   // row_ror:4
                    u[i] = x;
                }
                tot[g] = q == 0 ? u[0] : (q == 1 ? u[1] : (q == 2 ? u[2] : u[3]));
            }
            const float rr = sigmoid_pre(gxv[s][0] + tot[0]);
            const float zz = sigmoid_pre(gxv[s][1] + tot[1]);
            const float nn = tanh_pre(__builtin_fmaf(rr, tot[2] + bhn[s], gxv[s][2]));
            float hv = __builtin_fmaf(zz, hprev[s] - nn, nn);
            if constexpr (S::BIDIR) { if (dir && t_first - st > tb_last) hv = 0.0f; }
            hprev[s] = hv;
            if (ok[s]) { hn[hoff[s]] = hv; *hsp[s] = hv; }
            hsp[s] += hstep;
        }
        if (st + 1 < a.T) load_gx();                         // next step's x half: in flight under the next step's MFMAs
        cur ^= 1;
        __syncthreads();
    }
    if (hst != nullptr) {
#pragma unroll
        for (int s = 0; s < NSET; ++s)
            if (ok[s]) hst[(size_t)row * C2 + ch[s]] = hprev[s];
    }
}

// ------------------------------------------------------------------------------------------ RNNFormer block segment
template <class S, int FT>
struct BlkLds {
    using TT = TokTiling<S, FT>;
    static constexpr int LDH = S::ND * S::C2 + 2;                  // GRU-output rows
    static constexpr int X = 0;                                    // tokens [ROWS_P][LDX]
    static constexpr int U = X + TT::ROWS_P * S::LDX;              // union: { HS [ROWS_P][LDH] }  |  { HL [ROWS_P][LDX], GI [ROWS_P][LDG] }
    static constexpr int HS = U;
    static constexpr int HL = U;
    static constexpr int GI = HL + TT::ROWS_P * S::LDX;
    static constexpr int cmax(int p, int q) { return p > q ? p : q; }
    static constexpr int FULL = U + cmax(TT::ROWS_P * LDH, TT::ROWS_P * (S::LDX + S::LDG));
    // per-head qkv (C2 = 128, 48 kHz L): the full [rows][3 C2] buffer does not fit; q | k | v of ONE head at a time ([rows][3 hd + 2])
    static constexpr bool PERHEAD = (size_t)FULL * 4 > 160 * 1024;
    static constexpr int LDGX = PERHEAD ? 3 * S::HD + 2 : S::LDG;
    static constexpr int TOTAL = U + cmax(TT::ROWS_P * LDH, TT::ROWS_P * (S::LDX + LDGX));
    static constexpr size_t BYTES = (size_t)TOTAL * 4 + 16;      // (+ one word: the fused stage's "next tile is ready" flag)
    static constexpr bool OK = BYTES <= 160 * 1024;
    static constexpr int OCC = 2 * BYTES <= 160 * 1024 ? 2 : 1;   // workgroups per CU the plan allows: the register budget follows it
};

// FUSED (tb_stage_kernel): the tiles are walked TIME-major (tile n = frames tt FT .. of utterance n % B, tt = n / B) and a tile
// waits until the scan workgroups of its utterance's rows have published its frames (a.prog; one lane polls with agent-scope loads,
// the barrier hands the result to the workgroup; the tile's hs then comes in with agent-scope loads, x - written by the previous
// launch - with plain ones).
template <class S, int FT, bool FUSED>
__device__ __forceinline__ void blk_body(const TbArgs& a, float* smem, const int wgid, const int nwg) {
    using L = BlkLds<S, FT>;
    using TT = TokTiling<S, FT>;
    constexpr int C2 = S::C2, F2 = S::F2, LDX = S::LDX, LDH = L::LDH, LDG = L::LDGX, HD = S::HD, HW = S::ND * C2;
    constexpr int MTT = TT::MTT, NTPW2 = ceil_div(S::NT2, kWaves), NTPW3 = ceil_div(S::NT3, kWaves);
    constexpr PackedOffsets o = Pack<S>::v;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const WSrc<false> wb = make_wsrc<S>(a.wp, lane);
    const int k = a.k, kb = k * o.blk_stride;
    float* const Xb = smem + L::X;
    float* const Hs = smem + L::HS;
    float* const Hl = smem + L::HL;
    float* const Gi = smem + L::GI;
    const int TPU = (a.T + FT - 1) / FT;                      // FUSED: tiles per utterance
    const int ntiles = FUSED ? a.B * TPU : (a.NF + FT - 1) / FT;
    // tile -> its first frame (node-local index b * T + t) and the number of valid frames
    auto tile_frames = [&](int tl, int& nv) -> int {
        if constexpr (FUSED) {
            const int tt = tl / a.B, b = tl - tt * a.B, t0 = tt * FT;
            nv = a.T - t0 < FT ? a.T - t0 : FT;
            return b * a.T + t0;
        } else {
            const int g0 = tl * FT;
            nv = a.NF - g0 < FT ? a.NF - g0 : FT;
            return g0;
        }
    };
    // FUSED: has the scan published the tile's frames for every row of its utterance?  (thread 0; wait = poll until it has)
    auto tile_ready = [&](int tl, bool wait) -> bool {
        const int tt = tl / a.B, b = tl - tt * a.B;
        const unsigned int need = (unsigned int)((tt + 1) * FT < a.T ? (tt + 1) * FT : a.T);
        const int rg0 = (b * S::F2) / 16, rg1 = (b * S::F2 + S::F2 - 1) / 16;
        bool ok = true;
        for (int rg = rg0; rg <= rg1; ++rg) {
            int spins = 0;
            while (__hip_atomic_load(a.prog + rg * kProgStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                if (!wait || ++spins > (1 << 20)) { ok = false; break; }
                __builtin_amdgcn_s_sleep(16);
            }
        }
        return ok;
    };
    auto wait_tile = [&](int tl) {
        if constexpr (FUSED) {
            if (threadIdx.x == 0) (void)tile_ready(tl, true);
            __syncthreads();
        }
    };
    TB_PROBE_INIT(TB_BLK);
    // Small blocks (T, B, S): this wave's column tiles of EVERY weight matrix of the block stay in registers for the whole launch,
    // with the biases and - block 0 - the positional embedding of the lane's elements (the rows of a tile always start at
    // sub-band 0, so a lane meets the same pe entries in every tile).  A token GEMM then moves nothing but its A fragments
    // (LDS), the residual is the accumulators' initial value (read with the first A fragments instead of after the GEMM), and
    // the NEXT tile's tokens / GRU outputs are in flight from HBM while this tile computes.
    constexpr int KSF = S::ND * S::KS_2;
    constexpr int WREGS = NTPW2 * KSF + NTPW3 * S::KS_2 + NTPW2 * S::KS_2 + S::ND * NTPW3 * S::KS_2;
    constexpr bool REGW = !L::PERHEAD && WREGS <= 100 && MTT * NTPW3 <= 8;
    if constexpr (REGW) {
        float w1[NTPW2][KSF], wq[NTPW3][S::KS_2], w2[NTPW2][S::KS_2], wg[S::ND][NTPW3][S::KS_2];
        float b1[NTPW2], b2[NTPW2], bg[S::ND][NTPW3], pe[MTT][NTPW2][4];
        const bool more = k + 1 < S::KB;
        {
            const int w1o = S::BIDIR ? o.tb_fc1_w[0] + k * (S::KB > 1 ? o.tb_fc1_w[1] - o.tb_fc1_w[0] : 0) : o.blk_fc1_w[0] + kb;
#pragma unroll
            for (int j = 0; j < NTPW2; ++j) {
                int nt = wave + 4 * j;
                nt = nt < S::NT2 ? nt : S::NT2 - 1;
#pragma unroll
                for (int ks = 0; ks < KSF; ++ks) w1[j][ks] = wb.at_g(w1o + (nt * KSF + ks) * 64);
#pragma unroll
                for (int ks = 0; ks < S::KS_2; ++ks) w2[j][ks] = wb.at_g(o.blk_fc2_w[0] + kb + (nt * S::KS_2 + ks) * 64);
                b1[j] = wb.at16_g(o.blk_fc1_b[0] + kb + nt * 16);
                b2[j] = wb.at16_g(o.blk_fc2_b[0] + kb + nt * 16);
                const int col = 16 * nt + li;
#pragma unroll
                for (int i = 0; i < MTT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * i + 4 * lg + r;
                        pe[i][j][r] = (k == 0 && col < C2) ? wb.gather_g(o.blk_pe + (row % F2) * C2 + col) : 0.0f;
                    }
            }
            const int dw = o.tb_wih[0][S::ND - 1] - o.tb_wih[0][0], db = o.tb_bx[0][S::ND - 1] - o.tb_bx[0][0];
            const int kw = (k + 1) * (S::KB > 1 ? o.tb_wih[1][0] - o.tb_wih[0][0] : 0), kbx = (k + 1) * (S::KB > 1 ? o.tb_bx[1][0] - o.tb_bx[0][0] : 0);
#pragma unroll
            for (int j = 0; j < NTPW3; ++j) {
                int nt = wave + 4 * j;
                nt = nt < S::NT3 ? nt : S::NT3 - 1;
#pragma unroll
                for (int ks = 0; ks < S::KS_2; ++ks) wq[j][ks] = wb.at_g(o.blk_qkv[0] + kb + (nt * S::KS_2 + ks) * 64);
                const float gsc = gx_scale<S>(16 * nt + li);        // gx is stored SCALED (gx_scale): free here, the weights are loaded once
#pragma unroll
                for (int d = 0; d < S::ND; ++d) {
#pragma unroll
                    for (int ks = 0; ks < S::KS_2; ++ks) wg[d][j][ks] = more ? wb.at_g(o.tb_wih[0][0] + kw + d * dw + (nt * S::KS_2 + ks) * 64) * gsc : 0.0f;
                    bg[d][j] = more ? wb.at16_g(o.tb_bx[0][0] + kbx + d * db + nt * 16) * gsc : 0.0f;
                }
            }
        }
        // the next tile's tokens / GRU outputs: [frame][channel][F2] in global memory, one 16-byte load = four sub-bands of a channel
        constexpr int F4 = F2 / 4, XQ = FT * C2 * F4, HQ = FT * HW * F4;
        constexpr int NX = ceil_div(XQ, kThreads), NH = ceil_div(HQ, kThreads);
        float4 px[NX], ph[NH];
        auto fetch_tile = [&](int tl) {
            int nv;
            const int g0 = tile_frames(tl, nv);
            const float* xg = a.x + (size_t)g0 * (F2 * C2);
            const float* hg = a.hs + (size_t)g0 * (F2 * HW);
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                int i = tid + q * kThreads;
                i = i < XQ ? i : XQ - 1;
                const int fr = i / (C2 * F4), fs = fr < nv ? fr : nv - 1;
                px[q] = *reinterpret_cast<const float4*>(xg + (size_t)(i - fr * (C2 * F4) + fs * (C2 * F4)) * 4);
            }
#pragma unroll
            for (int q = 0; q < NH; ++q) {
                int i = tid + q * kThreads;
                i = i < HQ ? i : HQ - 1;
                const int fr = i / (HW * F4), fs = fr < nv ? fr : nv - 1;
                const float* hp = hg + (size_t)(i - fr * (HW * F4) + fs * (HW * F4)) * 4;
                if constexpr (FUSED) {       // (written by the scan workgroups of this launch: agent-scope loads)
                    ph[q].x = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ph[q].y = __hip_atomic_load(hp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ph[q].z = __hip_atomic_load(hp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ph[q].w = __hip_atomic_load(hp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else
                ph[q] = *reinterpret_cast<const float4*>(hp);
            }
        };
        auto park_tile = [&]() {
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                int i = tid + q * kThreads;
                i = i < XQ ? i : XQ - 1;                                  // (the threads past the end store the last element again)
                const int fr = i / (C2 * F4), qq = i - fr * (C2 * F4), c = qq / F4, f4 = qq - c * F4;
                float* d = Xb + (fr * F2 + 4 * f4) * LDX + c;
                d[0] = px[q].x; d[LDX] = px[q].y; d[2 * LDX] = px[q].z; d[3 * LDX] = px[q].w;
            }
#pragma unroll
            for (int q = 0; q < NH; ++q) {
                int i = tid + q * kThreads;
                i = i < HQ ? i : HQ - 1;
                const int fr = i / (HW * F4), qq = i - fr * (HW * F4), c = qq / F4, f4 = qq - c * F4;
                float* d = Hs + (fr * F2 + 4 * f4) * LDH + c;
                d[0] = ph[q].x; d[LDH] = ph[q].y; d[2 * LDH] = ph[q].z; d[3 * LDH] = ph[q].w;
            }
        };
        if (wgid < ntiles) { wait_tile(wgid); fetch_tile(wgid); park_tile(); }
        __syncthreads();
        TB_MARK(0);
#pragma unroll 1
        for (int tile = wgid; tile < ntiles; tile += nwg) {
            int nvalid;
            const int g0 = tile_frames(tile, nvalid);
            const int rows_valid = nvalid * F2;
            const int nxt = tile + nwg;
            // the next tile rides under this one's phases - FUSED: if the scan is already past it (asked once, by thread 0; the answer
            // crosses the first phase's barrier in LDS), else it is fetched after this tile
            bool pre = !FUSED;
            if constexpr (FUSED) {
                if (nxt < ntiles && threadIdx.x == 0) smem[L::TOTAL] = tile_ready(nxt, false) ? 1.0f : 0.0f;
            } else if (nxt < ntiles) fetch_tile(nxt);
            // ---- x += rnn_fc(h) (+ pe in block 0)   (model.py:273-280; noncausal: K = 2 C2)
            {
                f32x4 acc[MTT][NTPW2];
#pragma unroll
                for (int i = 0; i < MTT; ++i)
#pragma unroll
                    for (int j = 0; j < NTPW2; ++j) {
                        int nt = wave + 4 * j;
                        nt = nt < S::NT2 ? nt : S::NT2 - 1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] = Xb[(16 * i + 4 * lg + r) * LDX + 16 * nt + li] + (b1[j] + pe[i][j][r]);
                    }
                tok_panel_rb<MTT, NTPW2, KSF, LDH>(acc, Hs + li * LDH + lg, [&](int j, int ks) { return w1[j][ks]; });
#pragma unroll
                for (int i = 0; i < MTT; ++i)
#pragma unroll
                    for (int j = 0; j < NTPW2; ++j) {
                        const int nt = wave + 4 * j, col = 16 * nt + li;
                        if (nt < S::NT2 && col < C2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) Xb[(16 * i + 4 * lg + r) * LDX + col] = acc[i][j][r];
                        }
                    }
            }
            __syncthreads();
            TB_MARK(1);
            if constexpr (FUSED) {
                if (nxt < ntiles) { pre = smem[L::TOTAL] != 0.0f; if (pre) fetch_tile(nxt); }
            }
            // ---- qkv = x W_qkv^T  (rows per head interleaved [h][q|k|v][hd], model.py:142-146)
            {
                f32x4 acc[MTT][NTPW3];
                acc_init_zero<MTT, NTPW3>(acc);
                tok_panel_rb<MTT, NTPW3, S::KS_2, LDX>(acc, Xb + li * LDX + lg, [&](int j, int ks) { return wq[j][ks]; });
#pragma unroll
                for (int j = 0; j < NTPW3; ++j) {
                    const int nt = wave + 4 * j;
                    if (nt < S::NT3) {
#pragma unroll
                        for (int i = 0; i < MTT; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) Gi[(16 * i + 4 * lg + r) * LDG + 16 * nt + li] = acc[i][j][r];
                    }
                }
            }
            __syncthreads();
            TB_MARK(2);
            // ---- attention over the F2 tokens of each frame, wave = head; O -> Hl[token][head * hd + d]
#pragma unroll
            for (int f = 0; f < FT; ++f)
                attention_head<S, S::MT2, LDG, 3>(Gi + f * F2 * LDG, Hl + f * F2 * LDX, wave * 3 * HD, wave, 0, 1, lane);
            __syncthreads();
            TB_MARK(3);
            // ---- x += attn_fc(o)   (model.py:282-290) -> LDS and the token stream in global memory
            {
                f32x4 acc[MTT][NTPW2];
#pragma unroll
                for (int i = 0; i < MTT; ++i)
#pragma unroll
                    for (int j = 0; j < NTPW2; ++j) {
                        int nt = wave + 4 * j;
                        nt = nt < S::NT2 ? nt : S::NT2 - 1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] = Xb[(16 * i + 4 * lg + r) * LDX + 16 * nt + li] + b2[j];
                    }
                tok_panel_rb<MTT, NTPW2, S::KS_2, LDX>(acc, Hl + li * LDX + lg, [&](int j, int ks) { return w2[j][ks]; });
                const __amdgpu_buffer_rsrc_t xr = range_rsrc(a.x + (size_t)g0 * (F2 * C2), (size_t)rows_valid * C2 * 4);
#pragma unroll
                for (int i = 0; i < MTT; ++i)
#pragma unroll
                    for (int j = 0; j < NTPW2; ++j) {
                        const int nt = wave + 4 * j, col = 16 * nt + li;
                        if (nt < S::NT2 && col < C2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) Xb[(16 * i + 4 * lg + r) * LDX + col] = acc[i][j][r];
                            bstore4(xr, acc[i][j], cm_off<S>(16 * i + 4 * lg, col, C2) * 4);
                        }
                    }
            }
            __syncthreads();
            TB_MARK(4);
            // ---- the x half of the next block's GRU gates
            if (more) {
#pragma unroll
                for (int d = 0; d < S::ND; ++d) {
                    f32x4 acc[MTT][NTPW3];
#pragma unroll
                    for (int j = 0; j < NTPW3; ++j)
#pragma unroll
                        for (int i = 0; i < MTT; ++i) acc[i][j] = f32x4{bg[d][j], bg[d][j], bg[d][j], bg[d][j]};
                    tok_panel_rb<MTT, NTPW3, S::KS_2, LDX>(acc, Xb + li * LDX + lg, [&](int j, int ks) { return wg[d][j][ks]; });
                    const __amdgpu_buffer_rsrc_t gr = range_rsrc(a.gx + ((size_t)d * a.NF + g0) * (F2 * S::N3), (size_t)rows_valid * S::N3 * 4);
#pragma unroll
                    for (int j = 0; j < NTPW3; ++j) {
                        const int nt = wave + 4 * j, col = 16 * nt + li;
                        if (nt < S::NT3 && col < S::N3) {
#pragma unroll
                            for (int i = 0; i < MTT; ++i) bstore4(gr, acc[i][j], cm_off<S>(16 * i + 4 * lg, col, S::N3) * 4);
                        }
                    }
                }
            }
            __syncthreads();                                     // (every wave is done with Xb / Hl / Gi)
            TB_MARK(5);
            if (nxt < ntiles) {
                if (!pre) { wait_tile(nxt); fetch_tile(nxt); }
                park_tile();
            }
            __syncthreads();
            TB_MARK(0);
        }
        return;
    }
    static_assert(!FUSED || REGW, "the fused stage is built on the register-resident block path");
    if constexpr (!FUSED) {
    const int wave_outer = wave;
#pragma unroll 1
    for (int tile = wgid; tile < ntiles; tile += nwg) {
        // (streamed block weights - M, L, the noncausal model: the loop-variant zero of tb_enc / tb_dec keeps the tile's wave-uniform weight
        //  offsets out of the loop pre-header - tb_blk_kernel<L>: 299 SGPR spills, <M>: 184)
        int lz = 0;
        asm volatile("" : "+s"(lz));
        const int wave = wave_outer + lz;
        const int g0 = tile * FT;
        const int nvalid = a.NF - g0 < FT ? a.NF - g0 : FT;
        const int rows_valid = nvalid * F2;
        // ---- the tile's tokens and GRU outputs: [frame][channel][F2] in global memory, 16-byte loads (frames past the batch: clamped)
        load_tokens<S, FT, C2, LDX>(a.x + (size_t)g0 * (F2 * C2), nvalid, Xb, tid);
        load_tokens<S, FT, HW, LDH>(a.hs + (size_t)g0 * (F2 * HW), nvalid, Hs, tid);
        __syncthreads();
        TB_MARK(0);                 // loads of x, hs
        // ---- x += rnn_fc(h) (+ pe in block 0)   (model.py:273-280; noncausal: K = 2 C2)
        {
            f32x4 acc[MTT][NTPW2];
#pragma unroll
            for (int j = 0; j < NTPW2; ++j) {
                int nt = wave + 4 * j;
                nt = nt < S::NT2 ? nt : S::NT2 - 1;
                const float bj = wb.at16_g(o.blk_fc1_b[0] + kb + nt * 16);
#pragma unroll
                for (int i = 0; i < MTT; ++i) acc[i][j] = f32x4{bj, bj, bj, bj};
            }
            const int w_off = S::BIDIR ? o.tb_fc1_w[0] + k * (S::KB > 1 ? o.tb_fc1_w[1] - o.tb_fc1_w[0] : 0) : o.blk_fc1_w[0] + kb;
            tok_panel<MTT, NTPW2, S::ND * S::KS_2, LDH, S::BIDIR ? 0 : o.k4_delta>(acc, Hs + li * LDH + lg, wb, w_off, S::NT2, wave);
#pragma unroll
            for (int i = 0; i < MTT; ++i)
#pragma unroll
                for (int j = 0; j < NTPW2; ++j) {
                    const int nt = wave + 4 * j, col = 16 * nt + li;
                    if (nt < S::NT2 && col < C2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * i + 4 * lg + r;
                            float v = acc[i][j][r] + Xb[row * LDX + col];
                            if (k == 0) v += wb.gather_g(o.blk_pe + (row % F2) * C2 + col);
                            Xb[row * LDX + col] = v;
                        }
                    }
                }
        }
        __syncthreads();
        TB_MARK(1);                 // rnn_fc
        // ---- qkv = x W_qkv^T (rows per head interleaved [h][q|k|v][hd], model.py:142-146) and the attention over the F2 tokens of
        // each frame; O -> Hl[token][head * hd + d]
        if constexpr (!L::PERHEAD) {
            constexpr int CH = (MTT * NTPW3 <= 32) ? NTPW3 : (32 / MTT > 0 ? 32 / MTT : 1);
#pragma unroll 1
            for (int j0 = 0; j0 < NTPW3; j0 += CH) {
                f32x4 acc[MTT][CH];
                acc_init_zero<MTT, CH>(acc);
                f32x4 qcur[CH];
                mma_panel<MTT, CH, S::KS_2, kPD>(
                    acc, [&](int i, int ks) { return Xb[(16 * i + li) * LDX + lg + 4 * ks]; },
                    [&](int j, int ks) {
                        int nt = wave + 4 * (j0 + j);
                        nt = nt < S::NT3 ? nt : S::NT3 - 1;
                        if constexpr (FE_K4_STREAM && S::KS_2 >= 4 && o.k4_delta != 0) {       // (r4x: four k-steps per 16-byte load from the k4 copy)
                            const int base = o.blk_qkv[0] + kb + o.k4_delta + nt * (S::KS_2 * 64);
                            if (ks >= 4 * (S::KS_2 / 4)) return wb.at_g(base + ks * 64);
                            if ((ks & 3) == 0) qcur[j] = wb.at_gv4(base + (ks >> 2) * 256, wb.lane4 * 4);
                            return qcur[j][ks & 3];
                        } else return wb.at_g(o.blk_qkv[0] + kb + (nt * S::KS_2 + ks) * 64);
                    }, NoSide{});
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int nt = wave + 4 * (j0 + j);
                    if (nt < S::NT3) {
#pragma unroll
                        for (int i = 0; i < MTT; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) Gi[(16 * i + 4 * lg + r) * LDG + 16 * nt + li] = acc[i][j][r];
                    }
                }
            }
            __syncthreads();
            TB_MARK(2);             // qkv
            // wave = head, frame after frame
#pragma unroll 1
            for (int f = 0; f < FT; ++f)
                attention_head<S, S::MT2, LDG, kPD>(Gi + f * F2 * LDG, Hl + f * F2 * LDX, wave * 3 * HD, wave, 0, 1, lane);
        } else {
            // one head at a time, all four waves on it: its 3 hd qkv columns as (column tile, row tile) jobs -> Gi[row][3 hd + 2],
            // then its attention with the (frame, query tile) pairs split over the waves
#pragma unroll 1
            for (int hh = 0; hh < S::NH; ++hh) {
                // the head's 3 hd columns [c_lo, c_lo + 3 hd) of the packed qkv weight lie in column tiles t0 .. t0 + nth - 1
                const int c_lo = 3 * HD * hh, t0 = c_lo / 16, nth = (c_lo + 3 * HD - 1) / 16 - t0 + 1;
#pragma unroll 1
                for (int q = wave; q < nth * MTT; q += kWaves) {
                    const int ct = t0 + q % nth, m0 = q / nth;
                    f32x4 acc[1][1];
                    acc_init_zero<1, 1>(acc);
                    const int wq = o.blk_qkv[0] + kb + (ct * S::KS_2) * 64;
                    mma_panel<1, 1, S::KS_2, kPD>(
                        acc, [&](int, int ks) { return Xb[(16 * m0 + li) * LDX + lg + 4 * ks]; },
                        [&](int, int ks) { return wb.at_g(wq + ks * 64); }, NoSide{});
                    const int cl = 16 * ct + li - c_lo;
                    if (cl >= 0 && cl < 3 * HD) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Gi[(16 * m0 + 4 * lg + r) * LDG + cl] = acc[0][0][r];
                    }
                }
                __syncthreads();
                constexpr int NQW = ceil_div(S::MT2, kWaves);
#pragma unroll 1
                for (int f = 0; f < FT; ++f)
                    attention_head<S, NQW, LDG, kPD>(Gi + f * F2 * LDG, Hl + f * F2 * LDX, 0, hh, wave, kWaves, lane);
                __syncthreads();                             // (the next head overwrites Gi)
            }
        }
        if constexpr (!L::PERHEAD) __syncthreads();
        TB_MARK(3);                 // attention (per-head shapes: qkv + attention)
        // ---- x += attn_fc(o)   (model.py:282-290) -> LDS and the token stream in global memory
        {
            f32x4 acc[MTT][NTPW2];
#pragma unroll
            for (int j = 0; j < NTPW2; ++j) {
                int nt = wave + 4 * j;
                nt = nt < S::NT2 ? nt : S::NT2 - 1;
                const float bj = wb.at16_g(o.blk_fc2_b[0] + kb + nt * 16);
#pragma unroll
                for (int i = 0; i < MTT; ++i) acc[i][j] = f32x4{bj, bj, bj, bj};
            }
            tok_panel<MTT, NTPW2, S::KS_2, LDX, o.k4_delta>(acc, Hl + li * LDX + lg, wb, o.blk_fc2_w[0] + kb, S::NT2, wave);
            const __amdgpu_buffer_rsrc_t xr = range_rsrc(a.x + (size_t)g0 * (F2 * C2), (size_t)rows_valid * C2 * 4);
#pragma unroll
            for (int i = 0; i < MTT; ++i)
#pragma unroll
                for (int j = 0; j < NTPW2; ++j) {
                    const int nt = wave + 4 * j, col = 16 * nt + li;
                    if (nt < S::NT2 && col < C2) {
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * i + 4 * lg + r;
                            v[r] = acc[i][j][r] + Xb[row * LDX + col];
                            Xb[row * LDX + col] = v[r];
                        }
                        bstore4(xr, v, cm_off<S>(16 * i + 4 * lg, col, C2) * 4);
                    }
                }
        }
        __syncthreads();
        TB_MARK(4);                 // attn_fc
        // ---- the x half of the next block's GRU gates
        if (k + 1 < S::KB) {
            const int dw = o.tb_wih[0][S::ND - 1] - o.tb_wih[0][0], db = o.tb_bx[0][S::ND - 1] - o.tb_bx[0][0];
            const int kw = (k + 1) * (S::KB > 1 ? o.tb_wih[1][0] - o.tb_wih[0][0] : 0), kbx = (k + 1) * (S::KB > 1 ? o.tb_bx[1][0] - o.tb_bx[0][0] : 0);
#pragma unroll 1
            for (int d = 0; d < S::ND; ++d)
                gx_gemm<S, FT>(Xb, wb, o.tb_wih[0][0] + kw + d * dw, o.tb_bx[0][0] + kbx + d * db, a.gx + ((size_t)d * a.NF + g0) * (F2 * S::N3), rows_valid, wave, lane);
        }
        __syncthreads();
        TB_MARK(5);                 // gx of the next block
    }
    }
}

template <class S, int FT>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(BlkLds<S, FT>::OCC, BlkLds<S, FT>::OCC))) tb_blk_kernel(TbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    blk_body<S, FT, false>(a, smem, blockIdx.x, gridDim.x);
}

// One block stage in ONE launch: workgroups 0 .. nscan - 1 run the time scan of block k (16 rows each), the others the block's tiles
// behind it.  The scan is a latency chain that leaves most of the chip idle and the tile pass cannot start before it - unless the
// tiles follow the scan frame by frame: a stage then takes about as long as its scan alone.  All workgroups must be co-resident (the
// tile workgroups spin on the scan's counters): a cooperative launch, sized by tb_launch.
template <class S, int FT>
constexpr bool stage_fusable() {
    constexpr int NTPW2 = ceil_div(S::NT2, kWaves), NTPW3 = ceil_div(S::NT3, kWaves);
    constexpr int WREGS = NTPW2 * S::ND * S::KS_2 + NTPW3 * S::KS_2 + NTPW2 * S::KS_2 + S::ND * NTPW3 * S::KS_2;
    return !S::BIDIR && !BlkLds<S, FT>::PERHEAD && WREGS <= 100 && TokTiling<S, FT>::MTT * NTPW3 <= 8 && BlkLds<S, FT>::OCC == 2;
}

template <class S, int FT>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) tb_stage_kernel(TbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (stage_fusable<S, FT>()) {
        __shared__ float hbuf[2][16 * S::LDX];
        if ((int)blockIdx.x < a.nscan) scan_role<S, kScanPub>(a, blockIdx.x, 0, hbuf);
        else blk_body<S, FT, true>(a, smem, (int)blockIdx.x - a.nscan, (int)gridDim.x - a.nscan);
    }
}

// ------------------------------------------------------------------------------------------ decoder segment
template <class S, int FT>
struct DecLds {
    static constexpr int TW = 0;                                   // twiddles
    static constexpr int WY = TW + S::NFFT;                        // [FT][ACT]: 1x1 outputs (its first rows hold the tokens X until the filterbank has run)
    static constexpr int WX = WY + FT * S::ACT;                    // [FT][ACT]: k = 3 outputs (first Y2 = the filterbank output [FT][F1][LDX]; at the end the iDFT scratch)
    static constexpr int PT = WX + FT * S::ACT;                    // transposed-conv partials [FT][F1][LDP]
    static constexpr int TOTAL = PT + FT * S::F1 * S::LDP;
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
    static constexpr bool OK = S::F2 * S::LDX <= S::ACT && S::F1 * S::LDX <= S::ACT && 4 * S::NFFT <= FT * S::ACT && BYTES <= 160 * 1024;
};

template <class S, int FT>
__global__ void __launch_bounds__(kThreads) tb_dec_kernel(TbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = DecLds<S, FT>;
    using CT = ConvTiling<S, FT>;
    using D = Dft<S, kPD>;
    constexpr int N = S::NFFT, F0 = S::F0, F1 = S::F1, C1 = S::C1, C2 = S::C2, F2 = S::F2;
    constexpr int LDC = S::LDC, LDX = S::LDX;
    constexpr PackedOffsets o = Pack<S>::v;
    const int tid = threadIdx.x, lane = tid & 63, wave_k = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const WSrc<false> wb = make_wsrc<S>(a.wp, lane);
    const int wave = wave_k, wm = CT::NS == kWaves ? 0 : wave / CT::NS;
    float2* tw = reinterpret_cast<float2*>(smem + L::TW);
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = reinterpret_cast<const float2*>(a.wp + o.twiddle)[i];
    float* const Wy = smem + L::WY;
    float* const Wx = smem + L::WX;
    float* const PT = smem + L::PT;
    float* const Xb = Wy;             // tokens [FT F2][LDX], dense rows
    float* const Y2 = Wx;             // filterbank output [FT][F1][LDX]
    typename D::InvConst idc;
    D::load(idc, wb, o, wave);
    const int ntiles = (a.NF + FT - 1) / FT;
    constexpr size_t SKF = (size_t)(S::NL + 1) * F1 * C1;
    constexpr int NPT = N / kThreads, NB = FT * F0 / kThreads;
    constexpr bool PARD = FT * 3 * N <= FT * S::ACT && F0 % kThreads == 0;       // every frame of the tile has its own inverse-transform scratch in Wx
    float ow[NPT], xcr[NB][2];
#pragma unroll
    for (int q = 0; q < NPT; ++q) ow[q] = (a.wp + o.window)[tid + q * kThreads];
    __syncthreads();
    TB_PROBE_INIT(TB_DEC);
    TB_MARK(0);                     // prologue
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int lz = 0;                       // (big shapes: weight offsets not hoisted out of the tile loop - see tb_enc_kernel; tb_dec_kernel<L>: 1395 SGPR spills)
        if constexpr (S::C1 >= 96) asm volatile("" : "+s"(lz));
        const int wave = wave_k + lz, wm = CT::NS == kWaves ? 0 : wave / CT::NS;
        const int g0 = tile * FT;
        const int nvalid = a.NF - g0 < FT ? a.NF - g0 : FT;
        const int rows_valid = nvalid * F2;
        load_tokens<S, FT, C2, LDX>(a.x + (size_t)g0 * (F2 * C2), nvalid, Xb, tid);
        if constexpr (PARD) {       // the frames' compressed input spectra (the mask multiplies them at the end of the tile): in flight under the GEMMs
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int i = tid + q * kThreads, f = i / F0, fb = i - f * F0;
                const float* xcg = a.xc + (size_t)(g0 + (f < nvalid ? f : nvalid - 1)) * (2 * F0);
                xcr[q][0] = xcg[fb]; xcr[q][1] = xcg[F0 + fb];
            }
        }
        __syncthreads();
        TB_MARK(1);                 // load of x
        // the tile's encoder outputs as a buffer resource (coalesced A-fragment reads)
        WSrc<false> skb;
        {
            const int gl = g0 + FT <= a.NF ? FT : a.NF - g0;
            skb.rsrc = __builtin_amdgcn_make_buffer_rsrc(a.skip + (size_t)g0 * SKF, 0, (int)(gl * SKF * 4), 0x00020000);
            skb.lane4 = lane * 4; skb.li4 = (lane & 15) * 4; skb.lds = nullptr; skb.base = 0;
        }
        // ---- rf_post (model.py:485-490): Linear over the sub-band axis, A = packed [F1][F2], B = the frames' tokens side by side
        {
            constexpr int KS = F2 / 4;
            f32x4 acc[S::MTPW][FT * S::NT2];
            acc_init_zero<S::MTPW, FT * S::NT2>(acc);
            mma_panel<S::MTPW, FT * S::NT2, KS, kPD>(
                acc, [&](int i, int ks) { return wb.at_g(o.rfpost_lin + ((wave + 4 * i) * KS + ks) * 64); },
                [&](int j, int ks) { return Xb[((j / S::NT2) * F2 + 4 * ks + lg) * LDX + 16 * (j % S::NT2) + li]; }, NoSide{});
            __syncthreads();                                 // (Y2 does not overlap X, but Wy's halo rows - zeroed below - do)
#pragma unroll
            for (int i = 0; i < S::MTPW; ++i)
#pragma unroll
                for (int j = 0; j < FT * S::NT2; ++j) {
                    const int col = 16 * (j % S::NT2) + li;
                    if (col < C2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Y2[(j / S::NT2) * (F1 * LDX) + (16 * (wave + 4 * i) + 4 * lg + r) * LDX + col] = acc[i][j][r];
                    }
                }
            for (int i = tid; i < FT * 2 * LDC; i += kThreads) {
                const int f = i / (2 * LDC), q = i - f * (2 * LDC);
                Wy[f * S::ACT + (q >= LDC ? (F1 + 1) * LDC + (q - LDC) : q)] = 0.0f;
            }
        }
        __syncthreads();
        TB_MARK(2);                 // rf_post filterbank
        // ---- decoder (model.py:492-506): 1x1 on cat([x, skip]) as two K segments (layer 0: x = Y2 with rf_post's 1x1 folded into
        // the weights on the host, fe_api.hip), then the k = 3 conv
        static_for<S::NL>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            constexpr int K0 = (l == 0) ? S::KS_2 : S::KS_C;
            {
                const float* x0 = (l == 0) ? Y2 + (16 * wm + li) * LDX + lg : Wx + (16 * wm + li + 1) * LDC + lg;
                const int sk_off = (S::NL - l) * F1 * C1;
                conv_gemm<S, FT, K0 + S::KS_C, true, C1, LDC, S::ACT, 1>(
                    [&](int i, int ks) {
                        const int f = CT::frame(i), mt0 = CT::mtile0(i);
                        if (ks < K0) return (l == 0) ? x0[f * (F1 * LDX) + (16 * mt0) * LDX + 4 * ks] : x0[f * S::ACT + (16 * mt0) * LDC + 4 * ks];
                        return skb.at_g(f * (int)SKF + sk_off + ((mt0 + wm) * S::KS_C + (ks - K0)) * 64);
                    }, wb, o.dec1_w[l], o.dec1_b[l], Wy, nullptr, 0, nvalid, wave, lane);
            }
            __syncthreads();
            TB_MARK(3 + 2 * l);     // decoder layer l, 1x1
            conv_gemm<S, FT, 3 * S::KS_C, true, C1, LDC, S::ACT, 1>(K3Src<S, FT>{Wy + (16 * wm + li) * LDC + lg}, wb, o.dec3_w[l], o.dec3_b[l], Wx,
                                                                     nullptr, 0, nvalid, wave, lane);
            __syncthreads();
            TB_MARK(4 + 2 * l);     // decoder layer l, k = 3
        });
        // ---- dec_post (model.py:508-521): 1x1 on cat([x, enc_pre output]), then the transposed conv as a [F1 x C1].[C1 x 16] GEMM
        {
            const float* x0 = Wx + (16 * wm + li + 1) * LDC + lg;
            conv_gemm<S, FT, 2 * S::KS_C, true, C1, LDC, S::ACT, 1>(
                [&](int i, int ks) {
                    const int f = CT::frame(i), mt0 = CT::mtile0(i);
                    if (ks < S::KS_C) return x0[f * S::ACT + (16 * mt0) * LDC + 4 * ks];
                    return skb.at_g(f * (int)SKF + ((mt0 + wm) * S::KS_C + (ks - S::KS_C)) * 64);
                }, wb, o.post1_w, o.post1_b, Wy, nullptr, 0, nvalid, wave, lane);
        }
        __syncthreads();
        TB_MARK(20);                // dec_post 1x1
        {
            const float* x0 = Wy + (16 * wm + li + 1) * LDC + lg;
            conv_gemm<S, FT, S::KS_C, false, 16, S::LDP, F1 * S::LDP, 0>(
                [&](int i, int ks) { return x0[CT::frame(i) * S::ACT + (16 * CT::mtile0(i)) * LDC + 4 * ks]; }, wb, o.post_t_w, -1, PT, nullptr, 0, nvalid, wave, lane);
        }
        __syncthreads();
        TB_MARK(21);                // transposed conv
        // ---- mask, un-compress (model.py:694-709), inverse DFT + synthesis window (functional/audio_modules.py:117-119)
        const float b0 = wb.scalar(o.post_t_b), b1 = wb.scalar(o.post_t_b + 1);
        if constexpr (PARD) {
            // all frames of the tile at once: the compressed input spectrum was fetched at the top of the tile (xcr), every frame has its own
            // transform scratch (spectrum | two partial outputs, 3 N floats in Wx), the transforms run back to back
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int i = tid + q * kThreads, f = i / F0, fb = i - f * F0;
                const int g = g0 + (f < nvalid ? f : nvalid - 1), bl = g / a.T, t = a.t0 + (g - bl * a.T), b = a.b0 + bl, TF = a.Tfull;
                const float* PTf = PT + f * (F1 * S::LDP);
                const int qq = fb + 2, j1 = qq & 3, i1 = qq >> 2;
                float m0 = b0, m1 = b1;
                if (i1 < F1) { m0 += PTf[i1 * S::LDP + j1]; m1 += PTf[i1 * S::LDP + 8 + j1]; }
                if (i1 >= 1) { m0 += PTf[(i1 - 1) * S::LDP + j1 + 4]; m1 += PTf[(i1 - 1) * S::LDP + 8 + j1 + 4]; }
                const float xr = xcr[q][0], xi = xcr[q][1];
                float yr = xr * m0 - xi * m1, yi = xr * m1 + xi * m0;
                if (a.mode == FE_MODE_OFFLINE && f < nvalid) {
                    float* sph = a.spec_out + (size_t)b * F0 * TF * 2;
                    *reinterpret_cast<float2*>(sph + ((size_t)fb * TF + t) * 2) = make_float2(yr, yi);
                }
                const float mag = sqrtf(yr * yr + yi * yi);
                const float gn = pow_f(mag, 1.0f / a.compression - 1.0f);
                yr *= gn; yi *= gn;
                if (a.mode == FE_MODE_SPEC) {
                    if (f < nvalid) {
                        float* spo = a.spec_out + (size_t)b * (F0 + 1) * TF * 2;
                        *reinterpret_cast<float2*>(spo + ((size_t)fb * TF + t) * 2) = make_float2(yr, yi);
                        if (fb == 0) *reinterpret_cast<float2*>(spo + ((size_t)F0 * TF + t) * 2) = make_float2(0.0f, 0.0f);
                    }
                } else {
                    float* q3 = Wx + f * 3 * N + 2 * N;
                    q3[fb] = yr;
                    q3[N / 2 + fb] = yi;
                }
            }
            if (a.mode != FE_MODE_SPEC) {
                __syncthreads();
#pragma unroll 1
                for (int f = 0; f < FT; ++f) D::inverse(Wx + f * 3 * N + 2 * N, Wx + f * 3 * N, Wx + f * 3 * N + N, tw, idc, wb, o, wave, lane);       // (each ends with a barrier)
#pragma unroll 1
                for (int f = 0; f < nvalid; ++f) {
                    const int g = g0 + f, bl = g / a.T, t = a.t0 + (g - bl * a.T), b = a.b0 + bl;
                    float* fr = a.frames + ((size_t)b * a.Tfull + t) * N;
                    const float* q0 = Wx + f * 3 * N;
#pragma unroll
                    for (int q = 0; q < NPT; ++q) {
                        const int n = tid + q * kThreads;
                        const int pi = D::pidx(n & (D::N1 - 1), n / D::N1);
                        fr[n] = (q0[pi] + q0[N + pi]) * ow[q];
                    }
                }
            }
        } else {
        float* const q0 = Wx;
        float* const q1 = Wx + N;
        float* const q3 = Wx + 3 * N;
#pragma unroll 1
        for (int f = 0; f < nvalid; ++f) {
            const int g = g0 + f, bl = g / a.T, t = a.t0 + (g - bl * a.T), b = a.b0 + bl;
            const int TF = a.Tfull;
            const float* xcg = a.xc + (size_t)g * (2 * F0);
            const float* PTf = PT + f * (F1 * S::LDP);
            float* spo = a.mode == FE_MODE_SPEC ? a.spec_out + (size_t)b * (F0 + 1) * TF * 2 : nullptr;
            float* sph = a.mode == FE_MODE_OFFLINE ? a.spec_out + (size_t)b * F0 * TF * 2 : nullptr;
            for (int fb = tid; fb < F0; fb += kThreads) {
                const int q = fb + 2, j1 = q & 3, i1 = q >> 2;
                float m0 = b0, m1 = b1;
                if (i1 < F1) { m0 += PTf[i1 * S::LDP + j1]; m1 += PTf[i1 * S::LDP + 8 + j1]; }
                if (i1 >= 1) { m0 += PTf[(i1 - 1) * S::LDP + j1 + 4]; m1 += PTf[(i1 - 1) * S::LDP + 8 + j1 + 4]; }
                const float xr = xcg[fb], xi = xcg[F0 + fb];
                float yr = xr * m0 - xi * m1, yi = xr * m1 + xi * m0;
                if (sph != nullptr) { sph[((size_t)fb * TF + t) * 2] = yr; sph[((size_t)fb * TF + t) * 2 + 1] = yi; }
                const float mag = sqrtf(yr * yr + yi * yi);
                const float gn = pow_f(mag, 1.0f / a.compression - 1.0f);
                yr *= gn; yi *= gn;
                if (spo != nullptr) {
                    spo[((size_t)fb * TF + t) * 2] = yr;
                    spo[((size_t)fb * TF + t) * 2 + 1] = yi;
                    if (fb == 0) { spo[((size_t)F0 * TF + t) * 2] = 0.0f; spo[((size_t)F0 * TF + t) * 2 + 1] = 0.0f; }
                } else {
                    q3[fb] = yr;
                    q3[N / 2 + fb] = yi;
                }
            }
            if (a.mode != FE_MODE_SPEC) {
                __syncthreads();
                D::inverse(q3, q0, q1, tw, idc, wb, o, wave, lane);       // (ends with a barrier)
                float* fr = a.frames + ((size_t)b * TF + t) * N;
#pragma unroll
                for (int q = 0; q < NPT; ++q) {
                    const int n = tid + q * kThreads;
                    const int pi = D::pidx(n & (D::N1 - 1), n / D::N1);
                    fr[n] = (q0[pi] + q1[pi]) * ow[q];
                }
                __syncthreads();
            }
        }
        }
        __syncthreads();
        TB_MARK(22);                // mask, un-compress, inverse DFT, window
    }
}

// ------------------------------------------------------------------------------------------ launch glue
// frames per tile of a segment: 2 where the plan fits and the shape is small enough for the tile's accumulators
template <template <class, int> class P, class S>
constexpr int pick_ft() {
    if constexpr (S::C1 <= 64 && P<S, 2>::OK) return 2;
    else return 1;
}

template <class S>
struct TbCfg {
    static constexpr int FT_E = pick_ft<EncLds, S>(), FT_B = pick_ft<BlkLds, S>(), FT_D = pick_ft<DecLds, S>();
    static_assert(EncLds<S, FT_E>::OK && BlkLds<S, FT_B>::OK && DecLds<S, FT_D>::OK, "time-batched engine: an LDS plan does not fit");
};

struct TbImpl {
    int ft[4];                  // frames per tile of the stage (scan: 0)
    size_t lds[4];              // dynamic LDS bytes
    void (*launch)(int stage, const TbArgs&, int max_wgs, hipStream_t, hipError_t*);
};

template <class K>
inline void set_lds(K* kern, size_t bytes, hipError_t* err) {
    static std::atomic<bool> done[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    if (!done[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { *err = e; return; }
        done[dev].store(true, std::memory_order_relaxed);
    }
}

// persistent grids: as many workgroups as fit on the chip at once (LDS-limited), each walking tiles blockIdx.x, + gridDim.x, ...
template <class S>
void tb_launch(int stage, const TbArgs& a, int max_wgs, hipStream_t st, hipError_t* err) {
    using C = TbCfg<S>;
    *err = hipSuccess;
    auto grid_for = [&](int ft, size_t lds) {
        const int ntiles = (a.NF + ft - 1) / ft;
        int per_cu = (int)((160 * 1024) / (lds > 0 ? lds : 1));
        per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
        static const int cap = [] { const char* v = std::getenv("FE_TB_PER_CU"); return v ? std::atoi(v) : 0; }();      // (measurement: one workgroup per CU)
        if (cap > 0 && per_cu > cap) per_cu = cap;
        const int slots = max_wgs * per_cu;
        return ntiles < slots ? ntiles : slots;
    };
    if (stage == TB_ENC) {
        auto* k = &tb_enc_kernel<S, C::FT_E>;
        constexpr size_t lds = EncLds<S, C::FT_E>::BYTES;
        set_lds(k, lds, err);
        if (*err != hipSuccess) return;
        note_kernel("tb_enc_kernel");
        hipLaunchKernelGGL(k, dim3(grid_for(C::FT_E, lds)), dim3(kThreads), lds, st, a);
    } else if (stage == TB_SCAN) {
        // 16 rows per workgroup when that fills the chip (the MFMA throughput of the 16x16x4 tiles is what counts then), else 4
        static const int force = [] { const char* v = std::getenv("FE_TB_SCAN_ROWS"); return v ? std::atoi(v) : 0; }();
        const int wg16 = ((a.B * S::F2 + 15) / 16) * S::ND;
        const int wg4 = (a.B * S::F2 / 4) * S::ND;
        (void)wg16;
        const bool rows16 = force == 16 || (force != 4 && 4 * wg4 > 5 * max_wgs);      // (two 4-row workgroups on a CU take turns on its SIMDs: no gain beyond ~1 per CU)
        note_kernel(rows16 ? "tb_scan_kernel" : "tb_scan4_kernel");
        if (rows16)
            hipLaunchKernelGGL((tb_scan_kernel<S>), dim3((a.B * S::F2 + 15) / 16, S::ND), dim3(kThreads), 0, st, a);
        else
            hipLaunchKernelGGL((tb_scan4_kernel<S>), dim3(a.B * S::F2 / 4, S::ND), dim3(kThreads), 0, st, a);
    } else if (stage == TB_STAGE) {
        // the fused stage: only where the 16-row scan is the one in use, its workgroups leave room for tile workgroups on every CU, and the
        // block runs on the register-resident path; anything else is refused (the caller launches scan and tiles one after the other)
        if constexpr (!stage_fusable<S, C::FT_B>()) { *err = hipErrorNotSupported; return; } else {
            const int nscan = (a.B * S::F2 + 15) / 16, wg4 = a.B * S::F2 / 4;
            const int TPU = (a.T + C::FT_B - 1) / C::FT_B, ntiles = a.B * TPU;
            const int room = 2 * max_wgs - nscan - 16;           // (a margin: the co-residency the runtime grants is sized by its own occupancy estimate)
            if (!(4 * wg4 > 5 * max_wgs) || nscan > max_wgs || room < max_wgs / 2 || a.prog == nullptr) { *err = hipErrorNotSupported; return; }
            auto* k = &tb_stage_kernel<S, C::FT_B>;
            constexpr size_t lds = BlkLds<S, C::FT_B>::BYTES;
            set_lds(k, lds, err);
            if (*err != hipSuccess) return;
            TbArgs args = a;
            args.nscan = nscan;
            void* kargs[] = {&args};
            int grid = nscan + (ntiles < room ? ntiles : room);
            static const int dbg = [] { const char* v = std::getenv("FE_TB_FUSE_DEBUG"); return v ? std::atoi(v) : 0; }();
            if (dbg == 1) grid = nscan;          // (timing experiment: the scan role alone, with its agent-scope stores and publishes; results are garbage)
            *err = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k), dim3(grid), dim3(kThreads), kargs, (unsigned int)lds, st);
            if (*err != hipSuccess) (void)hipGetLastError();
            else note_kernel("tb_stage_kernel");
            return;
        }
    } else if (stage == TB_BLK) {
        auto* k = &tb_blk_kernel<S, C::FT_B>;
        constexpr size_t lds = BlkLds<S, C::FT_B>::BYTES;
        set_lds(k, lds, err);
        if (*err != hipSuccess) return;
        note_kernel("tb_blk_kernel");
        hipLaunchKernelGGL(k, dim3(grid_for(C::FT_B, lds)), dim3(kThreads), lds, st, a);
    } else {
        auto* k = &tb_dec_kernel<S, C::FT_D>;
        constexpr size_t lds = DecLds<S, C::FT_D>::BYTES;
        set_lds(k, lds, err);
        if (*err != hipSuccess) return;
        note_kernel("tb_dec_kernel");
        hipLaunchKernelGGL(k, dim3(grid_for(C::FT_D, lds)), dim3(kThreads), lds, st, a);
    }
    *err = hipGetLastError();
}

template <class S>
TbImpl make_tb_impl() {
    using C = TbCfg<S>;
    return TbImpl{{C::FT_E, 0, C::FT_B, C::FT_D}, {EncLds<S, C::FT_E>::BYTES, 0, BlkLds<S, C::FT_B>::BYTES, DecLds<S, C::FT_D>::BYTES}, &tb_launch<S>};
}

}  // namespace tb
}  // namespace fe
