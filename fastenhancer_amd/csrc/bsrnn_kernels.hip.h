// bsrnn_kernels.hip.h — BSRNN (models/bsrnn/model.py of the reference) streaming / offline forward for gfx950.
//
// Same execution model as fe_kernels.hip.h: one workgroup (256 threads) owns one stream at a time and runs the whole frame
//   STFT -> compress (all 257 bins) -> band split (31 bands) -> L x [time-LSTM, bidirectional band-LSTM]
//        -> per-band mask/residual MLPs (GLU) -> complex mask + residual -> un-compress -> iSTFT
// with every activation in LDS.
//  * The batched contractions (time-LSTM gate pre-activations of the 31 bands, the band-LSTM input projections, the fc
//    layers) run on the fp32 matrix cores.  For num_channels = 16 (xt / xxt) a wave's weight fragments of a whole layer
//    (130 registers) are prefetched from L2 while the previous layer's band recurrence - a latency chain that leaves the
//    vector-memory path idle - is running; the GEMM phases then feed from registers and LDS only.
//  * The band-LSTM recurrence (31 sequential steps per direction, a 1 x 2C by 2C x 8C product each) runs on the vector
//    ALUs with one gate row per thread whose W_hh row stays in registers for the whole layer (C <= 32; C = 64 streams
//    it from L2), packed fp32 FMAs, the four gates of a hidden unit in adjacent lanes (DPP quad broadcasts).  Gate rows
//    are packed pre-scaled by -log2(e) (i, f, o) / -2 log2(e) (g), so every activation is rcp(1 + exp2(pre)).
//  * Band split and the per-band MLPs (every band has its own weights: M = 1, each weight is used once per frame) are
//    vector-ALU dot products over 16-byte weight loads, software-pipelined one row ahead: these phases are bound by the
//    L2 -> CU path (780 KB of MLP weights per frame for xt), not by arithmetic.
#pragma once
#include <atomic>

#include "fe_kernels.hip.h"

namespace fe {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kBands = 31;
constexpr int kBins = 257;
constexpr int kMlpRows = 4 * kBins;       // rows of the second MLP layers over all bands (1028)
constexpr int kBsKP = 36;                 // band-split K (2 * sub <= 34) padded to whole 16-byte loads
__device__ __constant__ const int c_sub[kBands] = {2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 16, 16, 16, 16, 16, 16, 16, 17};
__device__ __constant__ const int c_start[kBands] = {0, 2, 5, 8, 11, 14, 17, 20, 23, 26, 29, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120,
                                                     128, 144, 160, 176, 192, 208, 224, 240};

// compile-time shape: C = num_channels, NLAY = num_layers (n_fft = 512 is fixed by the model: models/bsrnn/model.py:112)
template <int C_, int NLAY_, int HOP_>
struct BShape {
    static constexpr int C = C_, NLAY = NLAY_, HOP = HOP_, NFFT = 512, LOG2N = 9;
    static constexpr int HH = 2 * C;            // LSTM hidden size
    static constexpr int G4 = 4 * HH;           // gate rows
    static constexpr int OVL = NFFT - HOP;
    static constexpr int LDX = C + 2, LDH = HH + 2, LDY = 2 * HH + 2, LDP = G4 + 2;
    static constexpr int LDH1 = 4 * C + 4;      // MLP hidden rows (16-byte aligned, bands 4 banks apart)
    static constexpr int NCT = HH / 16;         // hidden-unit tiles
    static constexpr int NTC = C / 16;          // channel tiles
    static constexpr int KSC = C / 4, KSH = HH / 4, KS1 = KSC + KSH;
    // C = 64: W_hh of a direction is 256 KB - streamed from L2 every step it made the recurrence 87 % of the frame (32 k cycles
    // per step, the L2 -> CU path saturated by 256 workgroups).  SEQD: the two directions run one after the other on all 256
    // threads, each with its W_hh register-resident (256 floats per thread, fetched once per layer and direction: 31 x less
    // L2 traffic, 62 instead of 31 barrier steps per layer).
    static constexpr bool SEQD = (C == 64);
    static constexpr int NTD = SEQD ? 256 : 128;          // threads of a direction
    static constexpr int UPP = NTD / 4;                   // hidden units per pass (a quad of lanes per unit)
    static constexpr int RPT = HH / UPP;                  // units (gate-row sets) per thread in the band recurrence
    static constexpr bool REGW = (C == 16);     // a layer's GEMM weight fragments live in registers (prefetched a layer ahead)
    static constexpr bool WREG = true;          // recurrence weights register-resident (C = 64: one direction at a time, SEQD)
    static constexpr bool XPG = (C > 32);       // band-LSTM input projections in a global scratch (do not fit in LDS)
    // band recurrence, work split of a quad of lanes (one hidden unit): KSPLIT = false: lane g holds gate row g over the
    // whole K = HH (reads all of h); KSPLIT = true: lane q holds a quarter of K for all four gate rows (reads HH/4 of h,
    // two DPP adds per gate to sum the quarters).  A lone wave per SIMD issues an instruction every ~5.5 cycles, so a step
    // costs its instruction count: at HH = 32 the extra 11 reduction instructions outweigh 6 fewer LDS reads (measured
    // 881 vs 813 cycles per step), at HH = 64 the 12 fewer LDS reads and no second pass win (1535 vs 1746).
    static constexpr bool KSPLIT = (C >= 32);
    static constexpr int NIPW = (2 * G4 / 16) / kWaves;   // input-projection column tiles per wave
    static_assert(C == 16 || C == 32 || C == 64, "num_channels must be 16, 32 or 64");
};

// offsets (floats) into the packed weight buffer; filled by the host packer (fe_api.hip)
struct BOffsets {
    int bs_w, bs_b;            // band split: [kBsKP/4][31*C] float4 (k-major, rows zero-padded to kBsKP), [31][C]
    int t_w[8], t_b[8];        // time LSTM: B fragments, K = C + HH (x rows then h rows), N = 4*HH, tile (gate*NCT + ct); bias b_ih + b_hh [4][HH]
    int tfc_w[8], tfc_b[8];    // fc_time: B fragments K = HH, N = C
    int f_wih[8][2], f_b[8][2], f_whh[8][2];   // band LSTM per direction: B fragments K = C, N = 4HH; bias; W_hh in thread order:
                               // KSPLIT: [rr][gate * HH/4 + kk][thread 4u + q] = W_hh[gate * HH + u + 32 rr][q * HH/4 + kk]
                               // else:   [rr][k][thread 4u + gate]            = W_hh[gate * HH + u + 32 rr][k]     (streamed shapes: k in float4 groups)
    int ffc_w[8], ffc_b[8];    // fc_freq: B fragments K = 2HH, N = C
    int m_w1[2], m_b1[2];      // mask decoder layer 1 per kind: [31][C/4][4C] float4 (per band k-major: coalesced over the outputs), [31][4C]
    int m_w2[2], m_b2[2];      // layer 2 per kind: [4C/4][1028] float4 (k-major over the global row index), [1028]
    int row_band;              // int[1028]: band of each layer-2 row
    int bin_row, bin_2sub;     // int[257]: first "a" row of a bin (its (re, im) pair), 2 * sub of its band (distance to the gate rows)
    int window, window_istft, twiddle;
    int dft1, dft2, dft3, dft4;   // constant operands of the matrix-core DFT (fe::Dft, N = 512; r5: the role-split PART 1's STFT)
    // r5, num_channels = 16: a layer's fragments regrouped for 16-byte fetches by the role-split PART 1 (bsrnn_ov_kernels.hip.h) - a
    // wave-level load costs the vector-memory path ~16 cycles whatever its width, and a layer is 182 dword fragments per wave
    int ov_t[8];               // time LSTM: [ct][gate][k-step / 4][lane][4]
    int ov_f2[8];              // fc_freq: [k-step / 4][lane][4]
    int ov_hh[8][2];           // W_hh in the scan's lane order: [row set (2)][k / 4][lane = half * 32 + unit][4]
    // ... and row-major pieces for the TRANSPOSED chains (bands as the N of every product: an accumulator fragment - rows 4 lg + r of lane
    // (li, lg) - is the next product's B operand as it stands, k-step r carrying row 4 lg + r; the A operand then is W[row li][4 lg + r]):
    int ov_tx[8];              // time LSTM, x rows: [ct][gate][16 gate rows][C] (scaled like t_w)
    int ov_f1t[8];             // fc_time: [C][HH]
    int ov_ipt[8][2];          // input projections per direction: [4 HH][C] (scaled like f_wih)
    int total;
};

struct BArgs {
    const float* wp;
    BOffsets off;
    const float* wav_in;
    float* wav_out;
    size_t in_stride, out_stride;
    float* cache_stft;
    float* cache_istft;
    float* lstm;              // [2*NLAY][B*31][HH]  (h0, c0, h1, c1, ...)
    const float* spec_in;     // spec mode [B][257][T][2]
    float* spec_out;
    float* xp_scratch;        // XPG shapes: [grid][2][32][G4]
    float* dbg;               // per-stage dumps (DBG instantiation) or nullptr
    size_t dbg_stride;
    int B, T, mode, Tw;
    float compression;
    unsigned long long* clk;  // fe_profile_step: cycle probes of workgroup 0 (PROF instantiation only)
    // time-pipelined offline launch (PIPE instantiation): pipe_p workgroups per utterance, workgroup p runs frames p, p + pipe_p, ...
    unsigned int* pipe_flags; // [B][NLAY]: frames whose layer-l time-LSTM state (h, c) is in `lstm`
    float* frames;            // [B][T][N] windowed output frames (summed / envelope-normalised by istft_ola_kernel)
    int pipe_p;
    // the split per-hop step (PART 1 -> bsrnn_mlp_kernel -> PART 2): band features after the last layer [B][31][C], the compressed
    // spectrum [B][257][2], the mask decoder's layer-2 pre-activations [B][2][1028]
    float* mlp_x;
    float* mlp_sp;
    float* mlp_pre;
    int mlp_tpw;              // bsrnn_mlp_kernel: sixteen-stream tiles per wave (> 1 only where a band's layer-2 weights all sit in the register ring: C = 16)
    float* sb_y;              // stream-batched layers (bsrnn_sb_kernels.hip.h): the band LSTM's outputs of the running layer [B][2][31][HH]
    int ov_off;               // fe_set_step_kernel(FE_STEP_KERNEL_WAVES4): PART 1 on the phase-by-phase kernel instead of the role-split one
    // r6, the fused per-hop step (bsrnn_ov_kernel<FUSED>): barrier counters of the sixteen-stream tiles [tiles][2] (handle-owned, zeroed once,
    // monotonic), nullptr = the three-launch step
    unsigned int* gsync;
};

// debug stage table: spec_in, compressed, band_split, (layer.l.time, layer.l.freq)..., mask_mlp, spec_out
template <class S>
struct BDebugLayout {
    static constexpr int n_stages = 3 + 2 * S::NLAY + 2;
    __host__ __device__ static constexpr int rows(int s) { return (s <= 1 || s >= 3 + 2 * S::NLAY) ? kBins : kBands; }
    __host__ __device__ static constexpr int cols(int s) { return s <= 1 ? 2 : (s < 3 + 2 * S::NLAY ? S::C : (s == 3 + 2 * S::NLAY ? 4 : 2)); }
    __host__ __device__ static constexpr size_t offset(int s) {
        size_t o = 0;
        for (int i = 0; i < s; ++i) o += (size_t)rows(i) * cols(i);
        return o;
    }
    __host__ __device__ static constexpr size_t total() { return offset(n_stages); }
};

template <class S, int PART = 0>
struct BLds {
    // PART (r6): the per-stream front (3) and tail (2) of the split per-hop steps keep only what they touch - 14.6 / 20.5 KB (xt) instead of the whole
    // frame's plan: four workgroups per CU (they are memory round trips and barriers, not throughput)
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int SP = 0;                              // compressed spectrum [257][2]
    static constexpr int TW = SP + 2 * kBins + 2;             // twiddles
    static constexpr int FA = TW + S::NFFT;                   // FFT ping-pong
    static constexpr int FB = FA + 2 * S::NFFT;
    static constexpr int X = FB + 2 * S::NFFT;                // [32][LDX] band features
    static constexpr int HS = X + 32 * S::LDX;                // [32][LDH] time-LSTM h (A operand)
    static constexpr int HN = HS + 32 * S::LDH;               // [32][LDH] new h (A operand of fc_time)
    static constexpr int YF = HN + 32 * S::LDH;               // [32][LDY] band-LSTM outputs (fwd | bwd)
    static constexpr int HB = (YF + 32 * S::LDY + 3) / 4 * 4;   // [2 dirs][2 buffers][HH], 16-byte aligned (float4 broadcast reads)
    static constexpr int XP = HB + 4 * S::HH;                 // [2][32][LDP] band-LSTM input projections (LDS-resident shapes)
    static constexpr int XP_SIZE = S::XPG ? 0 : 2 * 32 * S::LDP;
    // after the layers the HS / HN / YF / XP region is dead: the MLP hidden layer and the layer-2 pre-activations alias it
    static constexpr int H1 = (HS + 3) / 4 * 4;               // [2 kinds][31][LDH1]
    static constexpr int PRE = PART == 2 ? X : H1 + 2 * kBands * S::LDH1;     // [2][1028]
    static constexpr int TOTAL = PART == 3 ? X + 32 * S::LDX : PART == 2 ? PRE + 2 * kMlpRows : cmax(XP + XP_SIZE, PRE + 2 * kMlpRows);
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
    static_assert(BYTES <= 160 * 1024, "BSRNN LDS plan exceeds 160 KiB");
};

#define BE_CLK(i) do { if constexpr (PROF) { if (blockIdx.x == 0 && threadIdx.x == 0) a.clk[(i)] = __builtin_readcyclecounter(); } } while (0)

// HOT: the per-hop streaming step (mode and T = 1 are compile-time facts: no frame loop, no spec / offline branches)
// PROF: cycle probes per phase (fe_profile_step, tools/gpu_phases_bsrnn.py);  DBG: per-stage dumps (fe_debug_step)
// OCC2: the two-workgroups-per-CU build for batches with more streams than CUs (shapes whose LDS plan fits twice: xt /
// xxt).  A frame is a latency chain (186 barrier-separated recurrence steps); a second workgroup on the CU runs its own
// chain in the gaps.  It has to live in 256 registers per wave: no register-resident layer weights (streamed inside the
// GEMM pipelines like the larger shapes), shorter MLP weight rings - the other workgroup covers those latencies.
// PIPE: time pipelining of an offline launch (the per-hop kernel's scheme, fe_kernels.hip.h): everything in a frame but the time-LSTM's
// (h, c) per layer is independent of the other frames, so the frames of an utterance are spread over pipe_p CO-RESIDENT workgroups
// (cooperative launch) and the state is handed from frame t - 1 to frame t through `lstm` - agent-scope stores, drained, then an
// agent-scope store of a per-(utterance, layer) frame counter; the consumer polls the counter and fetches the state with agent-scope
// loads right before the layer's gate GEMM (state AND counter through agent-scope accesses: no release / acquire fence, which on this
// part writes back / invalidates whole caches).  The serial chain per frame and layer is wait -> fetch -> gates -> publish; the band
// recurrence, the fc layers and the mask MLPs - 9/10 of a frame - run in parallel across the frames in flight.
// PART (the per-hop step split in three launches, blaunch_split): every workgroup of the fused kernel streams the mask decoder's 780 KB
// (xt) of weights from L2 once per frame - each weight is used ONCE per stream (M = 1) - and that is 40 k of a frame's 240 k cycles at
// ~20 B/clk per CU.  Batched over the streams the same products are a small MFMA GEMM whose weights are read once per 64 streams:
// PART = 1 runs the frame up to the last layer and leaves the band features and the compressed spectrum in global memory,
// bsrnn_mlp_kernel computes the two MLP layers for all streams, PART = 2 applies GLU / mask / residual and runs the iSTFT.
template <class S, bool HOT, bool PROF, bool DBG, bool OCC2 = false, bool PIPE = false, int PART = 0>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, PART == 2 ? 6 : PART == 3 ? 4 : OCC2 ? 2 : 1))) bsrnn_frame_kernel(BArgs a) {
    static_assert(!PIPE || (!HOT && !PROF && !DBG), "the time-pipelined instantiation is the plain offline kernel");
    static_assert(PART == 0 || (HOT && !PROF && !DBG && !PIPE), "the split step is the per-hop streaming step");
    // PART = 3: the front of the frame alone (STFT, compress, band split -> mlp_x / mlp_sp): the stream-batched step
    // (bsrnn_sb_kernels.hip.h) runs the layers for sixteen streams per workgroup on the matrix cores
    const int aT = HOT ? 1 : a.T;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = BLds<S, (PART == 2 || PART == 3) ? PART : 0>;
    constexpr int N = S::NFFT, H = S::HOP, OVL = S::OVL, C = S::C, HH = S::HH, G4 = S::G4;
    constexpr int LDX = S::LDX, LDH = S::LDH, LDY = S::LDY, LDP = S::LDP, LDH1 = S::LDH1;
    constexpr int KSC = S::KSC, KSH = S::KSH, KS1 = S::KS1;
    constexpr bool REGW = S::REGW && !OCC2 && PART != 3;
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
    float* X = smem + L::X;
    float* Hs = smem + L::HS;
    float* Hn = smem + L::HN;
    float* Cn = smem + L::YF;         // the time-LSTM's new cell state on its way to the state buffer (Yf is not live yet in that phase)
    float* Yf = smem + L::YF;
    float* Hb = smem + L::HB;
    float* XPl = smem + L::XP;
    float* H1 = smem + L::H1;
    float* PRE = smem + L::PRE;
    float* XPg = S::XPG ? a.xp_scratch + (size_t)blockIdx.x * (2 * 32 * G4) : nullptr;
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = reinterpret_cast<const float2*>(wp + o.twiddle)[i];
    if (tid < 2) sp[2 * kBins + tid] = 0.0f;      // (r5) the band split reads the last band's row zero-padded to kBsKP floats: the two words past the spectrum meet zero
                                                  // weights - leftovers of another kernel there could be NaN / inf (seen once: a whole run non-finite right after process start)
    __syncthreads();

    const int mode = HOT ? FE_MODE_STREAM : a.mode;
    // band recurrence: a quad of lanes <-> (direction, hidden unit); lane rgate of the quad holds a quarter of K for all four gates
    const int rd0 = wave >> 1;               // direction (wave-uniform: waves 0, 1 forward, waves 2, 3 backward; SEQD: a loop over both)
    const int rq = tid & (S::NTD - 1);       // thread within the direction's group
    const int rgate = rq & 3;

    // ---- a layer's register-resident weight set (REGW): this wave's fragments / this thread's recurrence row
    constexpr int NFC1 = REGW ? KSH : 1, NFC2 = REGW ? 2 * KSH : 1, NIP = S::NIPW;
    float Wt[REGW ? 4 : 1][REGW ? KS1 : 1], Wtb[REGW ? 4 : 1];
    float Wf1[NFC1], Wf1b, Wf2[NFC2], Wf2b, Wf2n[NFC2], Wf2bn;
    float Wip[REGW ? NIP : 1][REGW ? KSC : 1], Wipb[REGW ? NIP : 1];
    float Whh[S::WREG ? S::RPT : 1][S::WREG ? HH : 1], Whhn[REGW ? HH : 1];
    Wf1b = Wf2b = Wf2bn = 0.0f;
    // this wave's time-LSTM item (REGW: exactly one per wave)
    const int t_mt = wave / S::NCT, t_ct = wave % S::NCT;
    // element e of layer l's prefetch list (compile-time e): Wt, Wtb, Wf1, Wip, Wipb go in place (dead during the
    // recurrence that hides the fetch), W_hh / fc_freq into the *n set (still in use; handed over at the layer boundary)
    constexpr int E_WT = 0, E_WTB = E_WT + 4 * KS1, E_F1 = E_WTB + 4, E_F1B = E_F1 + KSH, E_IP = E_F1B + 1, E_IPB = E_IP + NIP * KSC,
                  E_HH = E_IPB + NIP, E_F2 = E_HH + HH, E_F2B = E_F2 + 2 * KSH, E_END = E_F2B + 1;
    auto fetch_elem = [&](auto e_, int l, auto first_) {
        constexpr int e = decltype(e_)::value;
        constexpr bool first = decltype(first_)::value;
        if constexpr (!REGW) return;
        else if constexpr (e < E_WTB) { constexpr int g = e / KS1, ks = e % KS1; Wt[g][ks] = wb.at_g(o.t_w[l] + ((g * S::NCT + t_ct) * KS1 + ks) * 64); }
        else if constexpr (e < E_F1) { constexpr int g = e - E_WTB; Wtb[g] = wb.at16_g(o.t_b[l] + g * HH + t_ct * 16); }
        else if constexpr (e < E_F1B) { constexpr int ks = e - E_F1; Wf1[ks] = wb.at_g(o.tfc_w[l] + ks * 64); }
        else if constexpr (e < E_IP) Wf1b = wb.at16_g(o.tfc_b[l]);
        else if constexpr (e < E_IPB) {
            constexpr int jj = (e - E_IP) / KSC, ks = (e - E_IP) % KSC;
            const int nt = wave + kWaves * jj, d = nt / (G4 / 16), ntd = nt - d * (G4 / 16);
            Wip[jj][ks] = wb.at_g(o.f_wih[l][d] + (ntd * KSC + ks) * 64);
        }
        else if constexpr (e < E_HH) {
            constexpr int jj = e - E_IPB;
            const int nt = wave + kWaves * jj, d = nt / (G4 / 16), ntd = nt - d * (G4 / 16);
            Wipb[jj] = wb.at16_g(o.f_b[l][d] + ntd * 16);
        }
        else if constexpr (e < E_F2) {
            constexpr int k = e - E_HH;
            const float v = wp[o.f_whh[l][rd0] + k * 128 + rq];
            if constexpr (first) Whh[0][k] = v; else Whhn[k] = v;
        }
        else if constexpr (e < E_F2B) {
            constexpr int ks = e - E_F2;
            const float v = wb.at_g(o.ffc_w[l] + ks * 64);
            if constexpr (first) Wf2[ks] = v; else Wf2n[ks] = v;
        }
        else { const float v = wb.at16_g(o.ffc_b[l]); if constexpr (first) Wf2b = v; else Wf2bn = v; }
    };
    // slice `part` of `parts` of the list
    auto fetch_part = [&](auto part_, auto parts_, int l, auto first_) {
        constexpr int part = decltype(part_)::value, parts = decltype(parts_)::value;
        constexpr int per = (E_END + parts - 1) / parts, lo = part * per, hi = (lo + per < E_END) ? lo + per : E_END;
        static_for<(hi > lo ? hi - lo : 0)>([&](auto i_) { fetch_elem(std::integral_constant<int, lo + decltype(i_)::value>{}, l, first_); });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

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
    // An opaque zero per stream: the per-thread weight addresses of the phases far down the frame (band split, mask MLPs) do not depend on the
    // stream, so hipcc computed them above this loop and kept them alive through the whole frame - 29 to 35 spilled registers whose scratch
    // stores, made once per workgroup, were 40 % of the kernel's HBM writes (profiles/pmc_r2_bsrnn_xt.json).  Made loop-variant they are
    // computed where they are used.
    int oz = 0;
    asm volatile("" : "+v"(oz));
    const int tidv = tid + oz;
    float* cst = a.cache_stft + (size_t)b * OVL;
    float* cis = a.cache_istft + (size_t)b * OVL;
    float* dbg = DBG ? a.dbg + (size_t)b * a.dbg_stride : nullptr;
    auto dump = [&](int stage, const float* src, int ld) {
        if constexpr (DBG) {
            using D = BDebugLayout<S>;
            const int rows = D::rows(stage), cols = D::cols(stage);
            float* dst = dbg + D::offset(stage);
            for (int i = tid; i < rows * cols; i += kThreads) { const int r = i / cols, c = i - r * cols; dst[i] = src[r * ld + c]; }
        }
    };

    unsigned int* pflag = PIPE ? a.pipe_flags + (size_t)b * S::NLAY : nullptr;
#pragma unroll 1
    for (int t = t_first; t < aT; t += t_step) {
        BE_CLK(0);
        // r5, PART 2 (the per-hop tail): the inverse transform on the matrix cores (fe::Dft: one barrier instead of nine radix-2 passes);
        // its constants and the old overlap tail are requested here, ahead of the GLU
        constexpr bool MIDFT = (PART == 2);
        using DI = Dft<S, 3>;
        typename DI::InvConst idc;
        constexpr int OPT = (N + kThreads - 1) / kThreads;      // output samples per thread
        float ocis[OPT], owin[OPT];
        if constexpr (MIDFT) {
            DI::load(idc, wb, o, wave);
#pragma unroll
            for (int q = 0; q < OPT; ++q) {
                const int n = tid + q * kThreads;
                ocis[q] = n < OVL ? cis[n] : 0.0f;
                owin[q] = wp[o.window_istft + (n < N ? n : N - 1)];
            }
        }
        if constexpr (PART != 2) {
        // ============================ STFT + compress (all 257 bins; models/bsrnn/model.py:430-436) ============================
        if (mode != FE_MODE_SPEC) {
            const float* win = wp + o.window;
            if (mode == FE_MODE_STREAM) {
                const float* xin = a.wav_in + (size_t)b * a.in_stride + (size_t)t * H;
                for (int n = tid; n < N; n += kThreads) {
                    float v = (n < OVL) ? cst[n] : xin[n - OVL];
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
            if constexpr (REGW) fetch_part(I0{}, I3{}, 0, std::true_type{});   // layer 0's weights: in flight across the FFT (three bursts:
                                                                                 // a wave keeps at most 63 vector-memory loads in flight)
            if (mode == FE_MODE_STREAM) {
                for (int m = tid; m < OVL; m += kThreads) cst[m] = fb[m + H].x;
                __syncthreads();
            }
            float2* Xf = fft_lds<S, false>(fa, fb, tw);
            if constexpr (REGW) fetch_part(I1{}, I3{}, 0, std::true_type{});
            if constexpr (DBG) { for (int f = tid; f < kBins; f += kThreads) { dbg[2 * f] = Xf[f].x; dbg[2 * f + 1] = Xf[f].y; } }
            for (int f = tid; f < kBins; f += kThreads) {
                const float re = Xf[f].x, im = Xf[f].y;
                const float g = pow_f(fmaxf(sqrtf(re * re + im * im), 1.0e-5f), a.compression - 1.0f);
                sp[2 * f] = re * g;
                sp[2 * f + 1] = im * g;
            }
        } else {
            if constexpr (REGW) { fetch_part(I0{}, I3{}, 0, std::true_type{}); fetch_part(I1{}, I3{}, 0, std::true_type{}); }
            const float* si = a.spec_in + (size_t)b * kBins * aT * 2;
            for (int f = tid; f < kBins; f += kThreads) {
                const float re = si[((size_t)f * aT + t) * 2], im = si[((size_t)f * aT + t) * 2 + 1];
                if constexpr (DBG) { dbg[2 * f] = re; dbg[2 * f + 1] = im; }
                const float g = pow_f(fmaxf(sqrtf(re * re + im * im), 1.0e-5f), a.compression - 1.0f);
                sp[2 * f] = re * g;
                sp[2 * f + 1] = im * g;
            }
        }
        __syncthreads();
        if constexpr (REGW) fetch_part(I2{}, I3{}, 0, std::true_type{});
        dump(1, sp, 2);

        BE_CLK(1);
        // ============================ band split (BandSplit.forward, :136-153; BN folded) ============================
        // thread <-> (band, channel): its zero-padded weight row of kBsKP floats as nine 16-byte loads, all in flight at once
        for (int i = tid; i < kBands * C; i += kThreads) {
            const int bb = i / C;
            const float4* w4 = reinterpret_cast<const float4*>(wp + o.bs_w) + i;      // [k/4][band * C + c] float4: coalesced over the threads
            float4 wv[kBsKP / 4];
#pragma unroll
            for (int k = 0; k < kBsKP / 4; ++k) wv[k] = w4[k * (kBands * C)];
            // first bin of the band (2, 10 x 3, 12 x 8, 7 x 16, 17 bins), computed - a table lookup would be a dependent load
            const int s0 = bb == 0 ? 0 : (bb <= 10 ? 3 * bb - 1 : (bb <= 22 ? 8 * bb - 56 : 16 * bb - 240));
            const float* s = sp + 2 * s0;                      // input index f*2 + ri
            float a0 = wp[o.bs_b + i], a1 = 0.0f;
#pragma unroll
            for (int k = 0; k < kBsKP / 4; ++k) {
                a0 += wv[k].x * s[4 * k] + wv[k].z * s[4 * k + 2];
                a1 += wv[k].y * s[4 * k + 1] + wv[k].w * s[4 * k + 3];
            }
            X[bb * LDX + (i - bb * C)] = a0 + a1;
        }
        // time-LSTM state of layer 0 (the later layers' is fetched under the previous layer's recurrence)
        constexpr int HPT = (kBands * HH + kThreads - 1) / kThreads;
        float hpre[HPT];
        if constexpr (!PIPE && PART != 3) {
            const float* hg0 = a.lstm + (size_t)b * (kBands * HH);
#pragma unroll
            for (int q = 0; q < HPT; ++q) { const int i = tid + q * kThreads; hpre[q] = hg0[i < kBands * HH ? i : kBands * HH - 1]; }
#pragma unroll
            for (int q = 0; q < HPT; ++q) {
                const int i = tid + q * kThreads, r = i / HH;
                if (i < kBands * HH) Hs[r * LDH + (i - r * HH)] = hpre[q];
            }
        }
        __syncthreads();
        dump(2, X, LDX);

#pragma unroll 1
        for (int l = 0; l < (PART == 3 ? 0 : S::NLAY); ++l) {
            float* hg = a.lstm + ((size_t)(2 * l) * a.B + b) * (kBands * HH);
            float* cg = a.lstm + ((size_t)(2 * l + 1) * a.B + b) * (kBands * HH);
            if (l == 0) BE_CLK(2);
            if constexpr (PIPE) {
                // frame t - 1's state of this layer: wait for it, fetch h into the GEMM's A operand buffer (c is fetched in the epilogue's place)
                if (t > 0) {
                    if (tid == 0) {
                        int spins = 0;
                        while (__hip_atomic_load(pflag + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)t && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
                        if (spins >= (1 << 24)) __builtin_trap();      // (a producer that never shows up: fail the launch loudly, never run on stale state)
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int q = 0; q < HPT; ++q) { const int i = tid + q * kThreads; hpre[q] = ld_state(hg + (i < kBands * HH ? i : kBands * HH - 1)); }
#pragma unroll
                for (int q = 0; q < HPT; ++q) {
                    const int i = tid + q * kThreads, r = i / HH;
                    if (i < kBands * HH) Hs[r * LDH + (i - r * HH)] = hpre[q];
                }
                __syncthreads();
            }
            // ---------------- time LSTM (LSTMCell over the 31 bands; :371-381)
            // items (m-tile, hidden tile): 4 gate accumulators each; gates fused into the epilogue (order i,f,g,o);
            // gate rows are packed pre-scaled, so sigma(v) = rcp(1 + exp2(pre)) and tanh(v) = 2 rcp(1 + exp2(pre)) - 1
            {
                constexpr int NITEM = 2 * S::NCT;
                auto item = [&](int mt, int ct, auto wtf, auto wtbf) {
                    // previous cell state of this lane's outputs: in flight under the GEMM
                    float cprev[4];
                    const int j = 16 * ct + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int row = 16 * mt + 4 * lg + r; cprev[r] = ld_state(cg + (row < kBands ? row : kBands - 1) * HH + j); }
                    f32x4 acc[1][4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) { const float bv = wtbf(g); acc[0][g] = f32x4{bv, bv, bv, bv}; }
                    const float* xa = X + (16 * mt + li) * LDX + lg;
                    const float* ha = Hs + (16 * mt + li) * LDH + lg;
                    mma_panel<1, 4, KS1, REGW ? 3 : 8>(
                        acc, [&](int, int ks) { return ks < KSC ? xa[4 * ks] : ha[4 * (ks - KSC)]; }, wtf, NoSide{});
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * lg + r;
                        const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0][0][r]));
                        const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0][1][r]));
                        const float gg = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0][2][r])) - 1.0f;
                        const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0][3][r]));
                        const float cn = fg * cprev[r] + ig * gg;
                        const float hn = og * tanh_f(cn);
                        if (row < kBands) {        // (the state goes out after the barrier, coalesced: an accumulator lane's four rows are 4-byte
                            Cn[row * LDH + j] = cn;    //  pieces of four different 128-byte lines - written from here, WRITE_SIZE was 1.7x the state)
                            Hn[row * LDH + j] = hn;
                        }
                    }
                };
                if constexpr (REGW) {
                    static_assert(!REGW || NITEM == kWaves, "one time-LSTM item per wave");
                    item(t_mt, t_ct, [&](int g, int ks) { return Wt[g][ks]; }, [&](int g) { return Wtb[g]; });
                } else {
#pragma unroll 1
                    for (int it = wave; it < NITEM; it += kWaves) {
                        const int mt = it / S::NCT, ct = it - mt * S::NCT;
                        item(mt, ct, [&](int g, int ks) { return wb.at_g(o.t_w[l] + ((g * S::NCT + ct) * KS1 + ks) * 64); },
                             [&](int g) { return wb.at16_g(o.t_b[l] + g * HH + ct * 16); });
                    }
                }
            }
            __syncthreads();
            {
                // the layer's new (h, c) -> the state, consecutive threads = consecutive floats of the [31][HH] tensors
#pragma unroll
                for (int q = 0; q < HPT; ++q) {
                    const int i = tid + q * kThreads, ic = i < kBands * HH ? i : kBands * HH - 1, r = ic / HH, c = ic - r * HH;
                    const float hv = Hn[r * LDH + c], cv = Cn[r * LDH + c];
                    if (i < kBands * HH) { st_state(hg + i, hv); st_state(cg + i, cv); }
                }
                if constexpr (PIPE) {      // publish: every thread's state stores have left the CU, then one store of the counter
                    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0)
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(pflag + l, (unsigned int)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (l == 0) BE_CLK(3);
            {
                // fc_time + residual (:382-384): X += Hn W^T + b
                constexpr int NITEM = 2 * S::NTC;
#pragma unroll 1
                for (int it = wave; it < NITEM; it += kWaves) {
                    const int mt = it / S::NTC, nt = it - mt * S::NTC;
                    f32x4 acc[1][1];
                    float bv;
                    if constexpr (REGW) bv = Wf1b; else bv = wb.at16_g(o.tfc_b[l] + nt * 16);
                    acc[0][0] = f32x4{bv, bv, bv, bv};
                    const float* ha = Hn + (16 * mt + li) * LDH + lg;
                    mma_panel<1, 1, KSH, REGW ? 3 : 8>(acc, [&](int, int ks) { return ha[4 * ks]; },
                                        [&](int, int ks) {
                                            if constexpr (REGW) return Wf1[ks];
                                            else return wb.at_g(o.tfc_w[l] + (nt * KSH + ks) * 64);
                                        }, NoSide{});
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * lg + r;
                        if (row < kBands) X[row * LDX + 16 * nt + li] += acc[0][0][r];
                    }
                }
            }
            __syncthreads();
            dump(3 + 2 * l, X, LDX);
            {
                if (l == 0) BE_CLK(4);
                // ---------------- band LSTM (:386-390): input projections of all 31 bands, both directions
                constexpr int NTP = 2 * (G4 / 16);       // n-tiles: direction-major
                auto proj = [&](int nt, auto wf, float bv) {
                    const int d = nt / (G4 / 16), ntd = nt - d * (G4 / 16);
                    f32x4 acc[2][1];
                    acc[0][0] = f32x4{bv, bv, bv, bv};
                    acc[1][0] = acc[0][0];
                    mma_panel<2, 1, KSC, REGW ? 3 : 8>(acc, [&](int i, int ks) { return X[(16 * i + li) * LDX + lg + 4 * ks]; }, wf, NoSide{});
                    float* dst = S::XPG ? XPg : XPl;
                    constexpr int ld = S::XPG ? G4 : LDP;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) dst[(d * 32 + 16 * i + 4 * lg + r) * ld + 16 * ntd + li] = acc[i][0][r];
                };
                if constexpr (REGW) {
                    static_for<S::NIPW>([&](auto jj_) {
                        constexpr int jj = decltype(jj_)::value;
                        proj(wave + kWaves * jj, [&](int, int ks) { return Wip[jj][ks]; }, Wipb[jj]);
                    });
                } else {
#pragma unroll 1
                    for (int nt = wave; nt < NTP; nt += kWaves) {
                        const int d = nt / (G4 / 16), ntd = nt - d * (G4 / 16);
                        proj(nt, [&](int, int ks) { return wb.at_g(o.f_wih[l][d] + (ntd * KSC + ks) * 64); }, wb.at16_g(o.f_b[l][d] + ntd * 16));
                    }
                }
            }
            // recurrence weights (register shapes without the layer-ahead prefetch: loaded here)
            float cstate[S::RPT];
            int rd = rd0;                                           // direction of the scan this thread works on
            auto load_whh = [&](int d) {
#pragma unroll
                for (int rr = 0; rr < S::RPT; ++rr) {
                    const float* wr = wp + o.f_whh[l][d] + rr * HH * S::NTD + rq;   // [rr][k][thread]: coalesced over the threads
#pragma unroll
                    for (int k = 0; k < HH; ++k) Whh[rr][k] = wr[k * S::NTD];
                }
            };
            if constexpr (S::WREG && !REGW && !S::SEQD) load_whh(rd);
            if (tid < 4 * HH) Hb[tid] = 0.0f;                    // h = 0 for both directions, both buffers
            if constexpr (4 * HH > kThreads) { if (tid + kThreads < 4 * HH) Hb[tid + kThreads] = 0.0f; }
            if constexpr (S::XPG) __threadfence_block();
            __syncthreads();
            if (l == 0) BE_CLK(5);
            // next layer's time-LSTM hidden state: fetched now, parked in Hs after the recurrence
            if (!PIPE && l + 1 < S::NLAY) {
                const float* hgn = a.lstm + ((size_t)(2 * l + 2) * a.B + b) * (kBands * HH);
#pragma unroll
                for (int q = 0; q < HPT; ++q) { const int i = tid + q * kThreads; hpre[q] = hgn[i < kBands * HH ? i : kBands * HH - 1]; }
            }
            // The walk of a direction (band = s forward, kBands - 1 - s backward) as RUNNING offsets - one add per step for the
            // projections and one for the outputs instead of a select and two multiplies - and the h double buffer's parity as a
            // compile-time fact (steps come in pairs): a lone wave per SIMD issues an instruction every ~5.5 cycles whatever its
            // kind, and the step's ~22 scalar bookkeeping instructions were a seventh of it.
            constexpr int XS = S::XPG ? G4 : LDP;
            int xo = 0, xd = 0, yob = 0, ydb = 0;           // (yob, ydb: bytes)
            const float act_m = rgate == 2 ? 2.0f : 1.0f, act_a = rgate == 2 ? -1.0f : 0.0f;
            auto walk_init = [&]() {
                const int band0 = rd == 0 ? 0 : kBands - 1;
                xo = (rd * 32 + band0) * XS + rgate * HH + (rq >> 2);
                yob = 4 * (band0 * LDY + rd * HH + (rq >> 2));
                xd = rd == 0 ? XS : -XS;
                ydb = rd == 0 ? 4 * LDY : -4 * LDY;
                asm volatile("" : "+v"(yob), "+v"(ydb));          // (per-lane values: the running output address is ONE vector add per step)
            };
            auto xp_ld = [&](int rr) -> float {
                if constexpr (S::XPG) return XPg[xo + S::UPP * rr];
                else return XPl[xo + S::UPP * rr];
            };
            float xp_next[S::RPT];
            // One step of both scans (work split: see BShape::KSPLIT).  With KSPLIT a lane reads HH/4 floats of h from LDS
            // instead of all HH; the quarters are summed across the quad with two DPP adds per gate; lane rgate then finishes
            // gate rgate.  Either way the four activations of a unit are exchanged with DPP quad broadcasts, ALL four lanes
            // compute the (identical) cell / hidden update and store it - no exec-masked block, no branch.
            // PAR: parity of the step (which half of the h double buffer is read); LAST: no projection fetch for a next step.
            auto rec_step = [&](auto PAR, auto LAST) {
                constexpr int par = decltype(PAR)::value;
                constexpr int Q = S::KSPLIT ? HH / 4 : HH;   // floats of h per lane
                const float4* hp4 = reinterpret_cast<const float4*>(Hb + (rd * 2 + par) * HH + (S::KSPLIT ? rgate * Q : 0));
                float* hnext = Hb + (rd * 2 + (par ^ 1)) * HH;
                float xp_cur[S::RPT];
#pragma unroll
                for (int rr = 0; rr < S::RPT; ++rr) xp_cur[rr] = xp_next[rr];
                if constexpr (!decltype(LAST)::value) {
                    xo += xd;
#pragma unroll
                    for (int rr = 0; rr < S::RPT; ++rr) xp_next[rr] = xp_ld(rr);
                }
                float4 hq[Q / 4];
#pragma unroll
                for (int k = 0; k < Q / 4; ++k) hq[k] = hp4[k];
#pragma unroll
                for (int rr = 0; rr < S::RPT; ++rr) {
                    const int j = (rq >> 2) + S::UPP * rr;
                    float mine;
                    if constexpr (!S::KSPLIT) {
                        // this lane's gate row over the whole K: two chains of packed FMAs
                        f32x2 p0 = {0.0f, 0.0f}, p1 = {0.0f, 0.0f};
                        if constexpr (S::WREG) {
#pragma unroll
                            for (int k = 0; k < Q / 4; ++k) {
                                p0 += f32x2{Whh[rr][4 * k], Whh[rr][4 * k + 1]} * f32x2{hq[k].x, hq[k].y};
                                p1 += f32x2{Whh[rr][4 * k + 2], Whh[rr][4 * k + 3]} * f32x2{hq[k].z, hq[k].w};
                            }
                        } else {
                            const float4* w4 = reinterpret_cast<const float4*>(wp + o.f_whh[l][rd]) + (size_t)rr * (Q / 4) * S::NTD + rq;
#pragma unroll 8
                            for (int k = 0; k < Q / 4; ++k) {
                                const float4 wv = w4[(size_t)k * S::NTD];
                                p0 += f32x2{wv.x, wv.y} * f32x2{hq[k].x, hq[k].y};
                                p1 += f32x2{wv.z, wv.w} * f32x2{hq[k].z, hq[k].w};
                            }
                        }
                        const f32x2 ps = p0 + p1;
                        mine = ps.x + ps.y;
                    } else {
                    f32x2 p[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) p[g] = f32x2{0.0f, 0.0f};
                    if constexpr (S::WREG) {
#pragma unroll
                        for (int k = 0; k < Q / 4; ++k)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                p[g] += f32x2{Whh[rr][g * Q + 4 * k], Whh[rr][g * Q + 4 * k + 1]} * f32x2{hq[k].x, hq[k].y};
                                p[g] += f32x2{Whh[rr][g * Q + 4 * k + 2], Whh[rr][g * Q + 4 * k + 3]} * f32x2{hq[k].z, hq[k].w};
                            }
                    } else {
                        // streamed: [rr][gate][k/4][128 threads] float4
                        const float4* w4 = reinterpret_cast<const float4*>(wp + o.f_whh[l][rd]) + (size_t)rr * 4 * (Q / 4) * S::NTD + rq;
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int k = 0; k < Q / 4; ++k) {
                                const float4 wv = w4[(size_t)(g * (Q / 4) + k) * S::NTD];
                                p[g] += f32x2{wv.x, wv.y} * f32x2{hq[k].x, hq[k].y};
                                p[g] += f32x2{wv.z, wv.w} * f32x2{hq[k].z, hq[k].w};
                            }
                    }
                    if constexpr (S::SEQD) {      // (C = 64: 256 register-resident weights per thread; the exchange below costs it 6 %)
                        float sg4[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float v = p[g].x + p[g].y;
                            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
                            sg4[g] = v;
                        }
                        mine = rgate == 0 ? sg4[0] : (rgate == 1 ? sg4[1] : (rgate == 2 ? sg4[2] : sg4[3]));
                    } else {
                    // lane q of the quad wants the quad-wide sum of gate q: a 4 x 4 transpose-reduce in two exchange steps (three DPP
                    // adds) instead of four butterflies and a four-way select (which hipcc lowered to a divergent switch: ~30
                    // instructions of exec-mask juggling per unit and step)
                    const float s0 = p[0].x + p[0].y, s1 = p[1].x + p[1].y, s2 = p[2].x + p[2].y, s3 = p[3].x + p[3].y;
                    const bool q1 = (rgate & 1) != 0, q2 = (rgate & 2) != 0;
                    float a_m = q1 ? s1 : s0, b_m = q1 ? s3 : s2;
                    const float a_o = q1 ? s0 : s1, b_o = q1 ? s2 : s3;
                    a_m += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a_o), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                    b_m += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b_o), 0xB1, 0xf, 0xf, true));
                    const float m_ = q2 ? b_m : a_m, o_ = q2 ? a_m : b_m;
                    mine = m_ + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o_), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
                    }
                    }
                    const float pre = mine + xp_cur[rr];
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre));
                    const float act = __builtin_fmaf(sg, act_m, act_a);          // g: tanh = 2 s - 1, the others: s
                    const int ai = __builtin_bit_cast(int, act);
                    const float ig = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0x00, 0xf, 0xf, true));
                    const float fg = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0x55, 0xf, 0xf, true));
                    const float gg = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0xaa, 0xf, 0xf, true));
                    const float og = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0xff, 0xf, 0xf, true));
                    const float cn = fg * cstate[rr] + ig * gg;
                    cstate[rr] = cn;
                    const float hn = og * (2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * cn)) - 1.0f);
                    hnext[j] = hn;                                  // (the four lanes of the quad store the same value)
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(Yf) + yob + 4 * S::UPP * rr) = hn;
                }
                yob += ydb;
                __syncthreads();
            };
            using P0 = std::integral_constant<int, 0>;
            using P1 = std::integral_constant<int, 1>;
            auto rec_pairs = [&](int n) {          // 2 n steps, starting on an even one
#pragma unroll 1
                for (int s = 0; s < n; ++s) { rec_step(P0{}, std::false_type{}); rec_step(P1{}, std::false_type{}); }
            };
#pragma unroll 1
            for (int dd = 0; dd < (S::SEQD ? 2 : 1); ++dd) {
            if constexpr (S::SEQD) { rd = dd; load_whh(dd); }      // (the previous direction's last step ended with a barrier)
#pragma unroll
            for (int rr = 0; rr < S::RPT; ++rr) cstate[rr] = 0.0f;
            walk_init();
#pragma unroll
            for (int rr = 0; rr < S::RPT; ++rr) xp_next[rr] = xp_ld(rr);
            static_assert(kBands == 31, "the recurrence is walked as 15 pairs of steps and a last one");
            if constexpr (REGW) {
                // the next layer's weights ride under this latency chain, in three bursts (a wave keeps at most 63 loads in flight)
                const bool more = l + 1 < S::NLAY;
                const int ln = more ? l + 1 : l;
                if (more) fetch_part(I0{}, I3{}, ln, std::false_type{});
                rec_pairs(5);
                if (more) fetch_part(I1{}, I3{}, ln, std::false_type{});
                rec_pairs(5);
                if (more) fetch_part(I2{}, I3{}, ln, std::false_type{});
                rec_pairs(5);
            } else {
                rec_pairs(15);
            }
            rec_step(P0{}, std::true_type{});
            }
            if (l == 0) BE_CLK(6);
            {
                // fc_freq + residual: X += Yf W^T + b
                constexpr int NITEM = 2 * S::NTC;
#pragma unroll 1
                for (int it = wave; it < NITEM; it += kWaves) {
                    const int mt = it / S::NTC, nt = it - mt * S::NTC;
                    f32x4 acc[1][1];
                    float bv;
                    if constexpr (REGW) bv = Wf2b; else bv = wb.at16_g(o.ffc_b[l] + nt * 16);
                    acc[0][0] = f32x4{bv, bv, bv, bv};
                    const float* ya = Yf + (16 * mt + li) * LDY + lg;
                    mma_panel<1, 1, 2 * KSH, REGW ? 3 : 8>(acc, [&](int, int ks) { return ya[4 * ks]; },
                                            [&](int, int ks) {
                                                if constexpr (REGW) return Wf2[ks];
                                                else return wb.at_g(o.ffc_w[l] + (nt * (2 * KSH) + ks) * 64);
                                            }, NoSide{});
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * lg + r;
                        if (row < kBands) X[row * LDX + 16 * nt + li] += acc[0][0][r];
                    }
                }
            }
            if (l + 1 < S::NLAY) {
                // park the next layer's hidden state / hand the prefetched W_hh and fc_freq sets over
                if constexpr (!PIPE) {
#pragma unroll
                for (int q = 0; q < HPT; ++q) {
                    const int i = tid + q * kThreads, r = i / HH;
                    if (i < kBands * HH) Hs[r * LDH + (i - r * HH)] = hpre[q];
                }
                }
                if constexpr (REGW) {
#pragma unroll
                    for (int k = 0; k < HH; ++k) Whh[0][k] = Whhn[k];
#pragma unroll
                    for (int k = 0; k < NFC2; ++k) Wf2[k] = Wf2n[k];
                    Wf2b = Wf2bn;
                }
            }
            __syncthreads();
            dump(4 + 2 * l, X, LDX);
            if (l == 0) BE_CLK(7);
        }

        }      // (PART != 2)
        BE_CLK(8);
        if constexpr (PART == 1 || PART == 3) {
            // hand-over to bsrnn_mlp_kernel / the PART 2 launch (PART 3: to bsrnn_sb_layers_kernel first)
            float* xg = a.mlp_x + (size_t)b * (kBands * C);
            for (int i = tid; i < kBands * C; i += kThreads) { const int bb = i / C; xg[i] = X[bb * LDX + (i - bb * C)]; }
            float* sg = a.mlp_sp + (size_t)b * (2 * kBins);
            for (int i = tid; i < 2 * kBins; i += kThreads) sg[i] = sp[i];
            __syncthreads();          // (a persistent workgroup's next stream overwrites X / sp)
        } else {
        // ============================ mask decoder (MaskDecoder.forward, :225-246) ============================
        // Every workgroup streams the same 780 KB (xt) of MLP weights once per frame.  Started together, the workgroups of
        // an XCD would all ask its L2 for the same lines at the same time (one channel busy, fifteen idle): each workgroup
        // therefore walks the rows in its own rotation.  Rows are software-pipelined D rows ahead of the FMAs (register ring).
        // layer 1: thread <-> output (kind, band, o): a wave covers 64 consecutive outputs of ONE band (4C >= 64), so the
        // X reads are LDS broadcasts
        if constexpr (PART == 0) {
            constexpr int NOUT = 2 * kBands * 4 * C, ROUNDS = (NOUT + kThreads - 1) / kThreads, R4 = C / 4;
            constexpr int D = OCC2 ? 2 : (R4 <= 4 ? 8 : (R4 <= 8 ? 4 : 2));      // rows in flight: under load the L2 round trip is ~1 us, the ring holds 32 KB per wave
            const int rot = (int)(blockIdx.x >> 3) % ROUNDS;
            auto out_of = [&](int r) { int rr = r + rot; rr = rr >= ROUNDS ? rr - ROUNDS : rr; return tidv + rr * kThreads; };
            auto load_row = [&](int r, float4 (&wv)[R4], float& bias) {
                if (r < ROUNDS) {
                    const int i = out_of(r), ii = i < NOUT ? i : NOUT - 1;
                    const int kind = ii / (kBands * 4 * C), rem = ii - kind * (kBands * 4 * C);
                    const int bb = rem / (4 * C), oo = rem - bb * (4 * C);
                    const float4* w4 = reinterpret_cast<const float4*>(wp + o.m_w1[kind]) + (size_t)bb * R4 * (4 * C) + oo;   // [band][k/4][o] float4
#pragma unroll
                    for (int k = 0; k < R4; ++k) wv[k] = w4[k * (4 * C)];
                    bias = wp[o.m_b1[kind] + rem];
                }
            };
            float4 ring[D][R4];
            float rb[D];
#pragma unroll
            for (int jj = 0; jj < D; ++jj) { rb[jj] = 0.0f; load_row(jj, ring[jj], rb[jj]); }
#pragma unroll 1
            for (int g = 0; g * D < ROUNDS; ++g) {
#pragma unroll
                for (int jj = 0; jj < D; ++jj) {
                    const int r = g * D + jj;
                    if (r < ROUNDS) {
                        const int i = out_of(r), ii = i < NOUT ? i : NOUT - 1;
                        const int kind = ii / (kBands * 4 * C), rem = ii - kind * (kBands * 4 * C);
                        const int bb = rem / (4 * C), oo = rem - bb * (4 * C);
                        const float* xr = X + bb * LDX;
                        float acc0 = rb[jj], acc1 = 0.0f;
#pragma unroll
                        for (int k = 0; k < R4; ++k) {
                            acc0 += ring[jj][k].x * xr[4 * k] + ring[jj][k].z * xr[4 * k + 2];
                            acc1 += ring[jj][k].y * xr[4 * k + 1] + ring[jj][k].w * xr[4 * k + 3];
                        }
                        if (i < NOUT) H1[(kind * kBands + bb) * LDH1 + oo] = tanh_f(acc0 + acc1);
                    }
                    load_row(r + D, ring[jj], rb[jj]);
                }
            }
        }
        __syncthreads();
        BE_CLK(9);
        {
            // layer 2: thread <-> output row (kind, r), K = 4C in chunks of 16 sixteen-byte loads; pipeline items are (row, chunk)
            // pairs, two items ahead; the hidden vector of the row's band is read from LDS (bands sit 4 banks apart)
            constexpr int K4 = 4 * C / 4;          // 16-byte loads per row
            constexpr int CH = K4 < 16 ? K4 : 16;  // loads per chunk (C = 64: a row is 4 chunks)
            constexpr int NCH = K4 / CH;
            constexpr int NROW = 2 * kMlpRows, ROUNDS = (NROW + kThreads - 1) / kThreads, NIT = ROUNDS * NCH, D = OCC2 ? 1 : 3;
            const int* row_band = reinterpret_cast<const int*>(wp + o.row_band);
            const int rot = (int)(blockIdx.x >> 3) % ROUNDS;
            auto out_of = [&](int r) { int rr = r + rot; rr = rr >= ROUNDS ? rr - ROUNDS : rr; return tidv + rr * kThreads; };
            // the bins' row tables for the GLU below: fetched now (a dependent load there would be exposed)
            constexpr int FPT = (kBins + kThreads - 1) / kThreads;
            int bra[FPT], brg[FPT];
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
            if constexpr (PART == 2) {
                const float* pg = a.mlp_pre + (size_t)b * (2 * kMlpRows);
                for (int i = tid; i < 2 * kMlpRows; i += kThreads) PRE[i] = pg[i];
                const float* sg = a.mlp_sp + (size_t)b * (2 * kBins);
                for (int i = tid; i < 2 * kBins; i += kThreads) sp[i] = sg[i];
            } else {
            auto load_item = [&](int it, float4 (&wv)[CH], int& band, float& bias) {
                if (it < NIT) {
                    const int r = it / NCH, ch = it - r * NCH;
                    const int i = out_of(r), ii = i < NROW ? i : NROW - 1;
                    const int kind = ii / kMlpRows, row = ii - kind * kMlpRows;
                    const float4* w4 = reinterpret_cast<const float4*>(wp + o.m_w2[kind]) + (size_t)(ch * CH) * kMlpRows + row;   // [k/4][row] float4
#pragma unroll
                    for (int k = 0; k < CH; ++k) wv[k] = w4[(size_t)k * kMlpRows];
                    band = row_band[row];                  // (the row's band and bias ride along: no dependent loads at compute time)
                    bias = wp[o.m_b2[kind] + row];
                }
            };
            float4 ring[D][CH];
            int rbnd[D];
            float rbias[D];
#pragma unroll
            for (int jj = 0; jj < D; ++jj) { rbnd[jj] = 0; rbias[jj] = 0.0f; load_item(jj, ring[jj], rbnd[jj], rbias[jj]); }
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll 1
            for (int g = 0; g * D < NIT; ++g) {
#pragma unroll
                for (int jj = 0; jj < D; ++jj) {
                    const int it = g * D + jj;
                    if (it < NIT) {
                        const int r = it / NCH, ch = it - r * NCH;
                        const int i = out_of(r), ii = i < NROW ? i : NROW - 1;
                        const int kind = ii / kMlpRows, row = ii - kind * kMlpRows;
                        const float4* h4 = reinterpret_cast<const float4*>(H1 + (kind * kBands + rbnd[jj]) * LDH1) + ch * CH;
                        if (ch == 0) { acc0 = rbias[jj]; acc1 = 0.0f; }
#pragma unroll
                        for (int k = 0; k < CH; ++k) {
                            const float4 hv = h4[k];
                            acc0 += ring[jj][k].x * hv.x + ring[jj][k].z * hv.z;
                            acc1 += ring[jj][k].y * hv.y + ring[jj][k].w * hv.w;
                        }
                        if (ch == NCH - 1 && i < NROW) PRE[i] = acc0 + acc1;
                    }
                    load_item(it + D, ring[jj], rbnd[jj], rbias[jj]);
                }
            }
            }      // (PART != 2)
            __syncthreads();
            // GLU(dim=1) per band: rows [0, 2 sub) are values, rows [2 sub, 4 sub) their gates; then spec * mask + residual (:393-401)
            float* spo = mode == FE_MODE_SPEC ? a.spec_out + (size_t)b * kBins * aT * 2 : nullptr;
            float* sph = mode == FE_MODE_OFFLINE ? a.spec_out + (size_t)b * kBins * aT * 2 : nullptr;
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
                if constexpr (DBG) {
                    float* dst = dbg + BDebugLayout<S>::offset(3 + 2 * S::NLAY) + 4 * f;
                    dst[0] = mr[0]; dst[1] = mr[1]; dst[2] = mr[2]; dst[3] = mr[3];
                }
                const float xr = sp[2 * f], xi = sp[2 * f + 1];
                float yr = xr * mr[0] - xi * mr[1] + mr[2];
                float yi = xr * mr[1] + xi * mr[0] + mr[3];
                if (sph != nullptr) { sph[((size_t)f * aT + t) * 2] = yr; sph[((size_t)f * aT + t) * 2 + 1] = yi; }
                const float g = pow_f(sqrtf(yr * yr + yi * yi), 1.0f / a.compression - 1.0f);
                yr *= g;
                yi *= g;
                if constexpr (DBG) {
                    float* dst = dbg + BDebugLayout<S>::offset(4 + 2 * S::NLAY) + 2 * f;
                    dst[0] = yr; dst[1] = yi;
                }
                if constexpr (MIDFT) {
                    float* Ys = reinterpret_cast<float*>(fa);            // {Re[N/2], Im[N/2]}, then Re of the Nyquist bin
                    if (f < N / 2) { Ys[f] = yr; Ys[N / 2 + f] = yi; }
                    else Ys[N] = yr;
                } else if (mode == FE_MODE_SPEC) {
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

        BE_CLK(10);
        // ============================ iSTFT (functional/audio_modules.py:259-303 / torch.istft) ============================
        if constexpr (PART == 2) {
            // irfft keeps Re X[N/2] only and ignores Im X[0]: the transform covers bins 0 .. N/2 - 1, the Nyquist bin is (-1)^n X[N/2] / N
            const float* Ys = reinterpret_cast<const float*>(fa);
            float* P0 = reinterpret_cast<float*>(fb);
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
            // (every read of the old overlap tail landed before the GLU's barrier)
#pragma unroll
            for (int q = 0; q < OPT; ++q) {
                const int n = tid + q * kThreads;
                if (n < H) out[n] = vo[q];
                else if (n < N) cis[n - H] = vo[q];
            }
        } else
        if (mode != FE_MODE_SPEC) {
            float2* y = fft_lds<S, true>(fa, fb, tw);
            float2* spare = (y == fa) ? fb : fa;
            const float* wi = wp + (mode == FE_MODE_STREAM ? o.window_istft : o.window);
            float* xo = reinterpret_cast<float*>(spare);
            const float invN = 1.0f / (float)N;
            if constexpr (PIPE) {
                float* fr = a.frames + ((size_t)b * aT + t) * N;
                for (int n = tid; n < N; n += kThreads) fr[n] = y[n].x * invN * wi[n];
                __syncthreads();               // (the next frame's FFT re-uses the buffers)
            } else {
            for (int n = tid; n < N; n += kThreads) {
                float v = y[n].x * invN * wi[n];
                if (n < OVL) v += cis[n];
                xo[n] = v;
            }
            __syncthreads();
            if (mode == FE_MODE_STREAM) {
                float* out = a.wav_out + (size_t)b * a.out_stride + (size_t)t * H;
                for (int n = tid; n < H; n += kThreads) out[n] = xo[n];
            } else {
                const float* w = wp + o.window;
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
        }      // (PART != 1)
        BE_CLK(11);
    }
    if constexpr (PIPE) break;
    b += gridDim.x;
    } while (b < a.B);
}

// The mask decoder's two MLP layers (MaskDecoder.forward, models/bsrnn/model.py:225-246) batched over the streams: a wave = 16 streams x one
// (kind, band); its four waves give a workgroup 64 streams of the same (kind, band), so the band's weights are read from L2 once per 64 streams
// (the fused kernel: once per stream).  Layer 1 [16 x C] x [C x 4C] + tanh -> a wave-private LDS tile (C/D layout -> A operand) -> layer 2
// [16 x 4C] x [4C x 4 sub] for the band's rows -> PRE[stream][kind][row] in global memory (GLU / mask in the PART 2 launch).  The B
// fragments come straight out of the VALU path's k-major float4 layouts ([band][k/4][o] / [k/4][row]): lane (n = li, k = 4 ks + lg) is
// element lg of float4 (ks, n).  No barrier: nothing is shared between the waves.
__host__ __device__ constexpr int bsrnn_band_sub(int b) { return b == 0 ? 2 : (b <= 10 ? 3 : (b <= 22 ? 8 : (b <= 29 ? 16 : 17))); }
__host__ __device__ constexpr int bsrnn_band_bin0(int b) { return b == 0 ? 0 : (b <= 10 ? 2 + 3 * (b - 1) : (b <= 22 ? 32 + 8 * (b - 11) : (b <= 29 ? 128 + 16 * (b - 23) : 240))); }
static_assert(bsrnn_band_bin0(30) + bsrnn_band_sub(30) == kBins, "band table");
template <class S>
struct BMlpLds {
    static constexpr int LDH = 4 * S::C + 2;                   // hidden rows: 2 x odd floats (conflict-free A-fragment reads)
    static constexpr size_t BYTES = (size_t)kWaves * 16 * LDH * 4;
};
// One wave's share of the mask decoder for sixteen streams (s0 .. s0 + 15) of one (kind, band): layer 1, then the layer-2 items
// it0, it0 + its, ... (nitw of them; an item = a column tile of the band's rows x a burst of KB k-steps) of `tpw` consecutive stream tiles.
// AGENT (r6, the fused per-hop step: bsrnn_ov_kernels.hip.h): the band features were written, and the pre-activations will be read, by
// OTHER workgroups of the same launch - agent-scope loads / stores (the hand-over protocol of the time-pipelined kernels: no fences).
template <class S, bool AGENT>
__device__ __forceinline__ void bsrnn_mlp_wave(const BArgs& a, float* h1, int kind, int band, int tile0, int tpw, int it0, int its, int nitw, int lane) {
    constexpr int C = S::C, O1 = 4 * C, R4 = C / 4, NT1 = O1 / 16, KS2 = O1 / 4, LDH = BMlpLds<S>::LDH;
    const int li = lane & 15, lg = lane >> 4;
    const float* __restrict__ wp = a.wp;
    const BOffsets& o = a.off;
    const int n2 = 4 * bsrnn_band_sub(band), row0 = 4 * bsrnn_band_bin0(band);
    constexpr int KB = KS2 < 16 ? KS2 : 16, NB = KS2 / KB, D = NB == 1 ? 5 : 8;   // (C = 16: a band has at most five items - all in flight at once)
    const int nit = nitw;
    auto row_of = [&](int nt) { const int c = 16 * nt + li; return row0 + (c < n2 ? c : n2 - 1); };
    // layer 2's work list: the items' weights ride D items ahead of the MFMAs in a register ring, and the first D are requested before layer 1
    // (a wave per SIMD: nothing else hides the L2 round trips)
    auto load_item = [&](int itl, float (&wv)[KB], float& bias) {
        if (itl < nit) {
            const int it = it0 + its * itl;
            const int nt = it / NB, kbi = it - nt * NB, row = row_of(nt);
            const float* w2 = wp + o.m_w2[kind] + (size_t)row * 4 + lg + (size_t)(kbi * KB) * kMlpRows * 4;
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) wv[ks] = w2[(size_t)ks * kMlpRows * 4];
            bias = wp[o.m_b2[kind] + row];
        }
    };
    float ring[D][KB], rbias[D];
#pragma unroll
    for (int jj = 0; jj < D; ++jj) { rbias[jj] = 0.0f; load_item(jj, ring[jj], rbias[jj]); }
#pragma unroll 1
    for (int tl = 0; tl < tpw; ++tl) {
    const int s0 = (tile0 + tl) * 16;
    if (s0 >= a.B) break;
    // ---- layer 1
    {
        const int srow = s0 + li < a.B ? s0 + li : a.B - 1;   // (rows past the batch shadow its last stream; their results are not stored)
        const float* xa = a.mlp_x + ((size_t)srow * kBands + band) * C + lg;
        float av[R4];
#pragma unroll
        for (int ks = 0; ks < R4; ++ks) {
            if constexpr (AGENT) av[ks] = __hip_atomic_load(xa + 4 * ks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else av[ks] = xa[4 * ks];
        }
        const float* w1 = wp + o.m_w1[kind] + (size_t)band * R4 * O1 * 4 + li * 4 + lg;
        const float* b1 = wp + o.m_b1[kind] + band * O1 + li;
#pragma unroll
        for (int nt0 = 0; nt0 < NT1; nt0 += 4) {              // four column tiles at a time (C = 64: 16 tiles)
            f32x4 acc[4];
            float wv[4][R4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bj = b1[16 * (nt0 + j)];
                acc[j] = f32x4{bj, bj, bj, bj};
#pragma unroll
                for (int ks = 0; ks < R4; ++ks) wv[j][ks] = w1[((size_t)ks * O1 + 16 * (nt0 + j)) * 4];
            }
#pragma unroll
            for (int ks = 0; ks < R4; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = FE_MFMA(av[ks], wv[j][ks], acc[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[(4 * lg + r) * LDH + 16 * (nt0 + j) + li] = tanh_f(acc[j][r]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- layer 2: the band's 4 sub rows (value rows, then gate rows)
    {
        float av[KS2];
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) av[ks] = h1[li * LDH + 4 * ks + lg];
        float* pre = a.mlp_pre + (size_t)kind * kMlpRows;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int g = 0; g * D < nit; ++g) {
#pragma unroll
            for (int jj = 0; jj < D; ++jj) {
                const int itl = g * D + jj;
                if (itl < nit) {
                    const int it = it0 + its * itl;
                    const int nt = it / NB, kbi = it - nt * NB;
                    if (kbi == 0) acc = f32x4{rbias[jj], rbias[jj], rbias[jj], rbias[jj]};
                    // (av is indexed by compile-time k-steps: one unrolled body per burst position)
                    static_for<NB>([&](auto kb_) {
                        constexpr int kbc = decltype(kb_)::value;
                        if (NB == 1 || kbi == kbc) {
#pragma unroll
                            for (int ks = 0; ks < KB; ++ks) acc = FE_MFMA(av[kbc * KB + ks], ring[jj][ks], acc);
                        }
                    });
                    if (kbi == NB - 1) {
                        const bool cok = 16 * nt + li < n2;
                        const int row = row_of(nt);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int st = s0 + 4 * lg + r;
                            if (cok && st < a.B) {
                                if constexpr (AGENT) __hip_atomic_store(pre + (size_t)st * (2 * kMlpRows) + row, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                else pre[(size_t)st * (2 * kMlpRows) + row] = acc[r];
                            }
                        }
                    }
                }
                load_item(itl + D, ring[jj], rbias[jj]);
            }
        }
    }
    }
}

// r6: the walk of SEVERAL stream tiles by one wave (a.mlp_tpw > 1: C = 16, large batches) - a function of its own: folded into bsrnn_mlp_wave as a compile-time
// variant it still changed the register allocation of the one-tile instantiations (num_channels = 64: 270 registers -> 512 + 198 spilled; bsrnn_s 610 -> 671 us)
template <class S, bool AGENT, bool MULTI = true>
__device__ __forceinline__ void bsrnn_mlp_wave_multi(const BArgs& a, float* h1, int kind, int band, int tile0, int tpw, int it0, int its, int nitw, int lane) {
    constexpr int C = S::C, O1 = 4 * C, R4 = C / 4, NT1 = O1 / 16, KS2 = O1 / 4, LDH = BMlpLds<S>::LDH;
    const int li = lane & 15, lg = lane >> 4;
    const float* __restrict__ wp = a.wp;
    const BOffsets& o = a.off;
    const int n2 = 4 * bsrnn_band_sub(band), row0 = 4 * bsrnn_band_bin0(band);
    constexpr int KB = KS2 < 16 ? KS2 : 16, NB = KS2 / KB, D = NB == 1 ? 5 : 8;   // (C = 16: a band has at most five items - all in flight at once)
    const int nit = nitw;
    auto row_of = [&](int nt) { const int c = 16 * nt + li; return row0 + (c < n2 ? c : n2 - 1); };
    // layer 2's work list: the items' weights ride D items ahead of the MFMAs in a register ring, and the first D are requested before layer 1
    // (a wave per SIMD: nothing else hides the L2 round trips)
    auto load_item = [&](int itl, float (&wv)[KB], float& bias) {
        if (itl < nit) {
            const int it = it0 + its * itl;
            const int nt = it / NB, kbi = it - nt * NB, row = row_of(nt);
            const float* w2 = wp + o.m_w2[kind] + (size_t)row * 4 + lg + (size_t)(kbi * KB) * kMlpRows * 4;
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) wv[ks] = w2[(size_t)ks * kMlpRows * 4];
            bias = wp[o.m_b2[kind] + row];
        }
    };
    float ring[D][KB], rbias[D];
#pragma unroll
    for (int jj = 0; jj < D; ++jj) { rbias[jj] = 0.0f; load_item(jj, ring[jj], rbias[jj]); }
    // r6: a wave that walks several stream tiles (C = 16, large batches) keeps layer 1's fragments in registers too and requests the NEXT tile's rows of x before
    // it stores this tile's pre-activations: vmcnt completes in order, so every tile's fetches - 20 of them - used to wait for the previous tile's stores to drain
    constexpr bool HOIST1 = MULTI && (NT1 * R4 <= 16);        // (MULTI: the instantiation of the waves that walk several tiles; a one-tile wave keeps its fetch order: x first)
    const float* w1 = wp + o.m_w1[kind] + (size_t)band * R4 * O1 * 4 + li * 4 + lg;
    const float* b1 = wp + o.m_b1[kind] + band * O1 + li;
    float wvh[HOIST1 ? NT1 : 1][HOIST1 ? R4 : 1], bjh[HOIST1 ? NT1 : 1];
    if constexpr (HOIST1) {
#pragma unroll
        for (int j = 0; j < NT1; ++j) {
            bjh[j] = b1[16 * j];
#pragma unroll
            for (int ks = 0; ks < R4; ++ks) wvh[j][ks] = w1[((size_t)ks * O1 + 16 * j) * 4];
        }
    }
    auto load_x = [&](float (&av)[R4], int s0) {
        const int srow = s0 + li < a.B ? s0 + li : a.B - 1;   // (rows past the batch shadow its last stream; their results are not stored)
        const float* xa = a.mlp_x + ((size_t)srow * kBands + band) * C + lg;
#pragma unroll
        for (int ks = 0; ks < R4; ++ks) {
            if constexpr (AGENT) av[ks] = __hip_atomic_load(xa + 4 * ks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else av[ks] = xa[4 * ks];
        }
    };
    float avn[R4];
    if constexpr (MULTI) load_x(avn, tile0 * 16 < a.B ? tile0 * 16 : 0);
#pragma unroll 1
    for (int tl = 0; tl < tpw; ++tl) {
    const int s0 = (tile0 + tl) * 16;
    if (s0 >= a.B) break;
    // ---- layer 1
    {
        float av[R4];
        if constexpr (MULTI) {
#pragma unroll
            for (int ks = 0; ks < R4; ++ks) av[ks] = avn[ks];
            if (tl + 1 < tpw && s0 + 16 < a.B) load_x(avn, s0 + 16);      // the next tile's rows: in flight under this tile's MFMAs, ahead of its stores
        } else load_x(av, s0);
#pragma unroll
        for (int nt0 = 0; nt0 < NT1; nt0 += 4) {              // four column tiles at a time (C = 64: 16 tiles)
            f32x4 acc[4];
            float wv[4][R4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bj = HOIST1 ? bjh[HOIST1 ? nt0 + j : 0] : b1[16 * (nt0 + j)];
                acc[j] = f32x4{bj, bj, bj, bj};
#pragma unroll
                for (int ks = 0; ks < R4; ++ks) wv[j][ks] = HOIST1 ? wvh[HOIST1 ? nt0 + j : 0][HOIST1 ? ks : 0] : w1[((size_t)ks * O1 + 16 * (nt0 + j)) * 4];
            }
#pragma unroll
            for (int ks = 0; ks < R4; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = FE_MFMA(av[ks], wv[j][ks], acc[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[(4 * lg + r) * LDH + 16 * (nt0 + j) + li] = tanh_f(acc[j][r]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- layer 2: the band's 4 sub rows (value rows, then gate rows)
    {
        float av[KS2];
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) av[ks] = h1[li * LDH + 4 * ks + lg];
        float* pre = a.mlp_pre + (size_t)kind * kMlpRows;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int g = 0; g * D < nit; ++g) {
#pragma unroll
            for (int jj = 0; jj < D; ++jj) {
                const int itl = g * D + jj;
                if (itl < nit) {
                    const int it = it0 + its * itl;
                    const int nt = it / NB, kbi = it - nt * NB;
                    if (kbi == 0) acc = f32x4{rbias[jj], rbias[jj], rbias[jj], rbias[jj]};
                    // (av is indexed by compile-time k-steps: one unrolled body per burst position)
                    static_for<NB>([&](auto kb_) {
                        constexpr int kbc = decltype(kb_)::value;
                        if (NB == 1 || kbi == kbc) {
#pragma unroll
                            for (int ks = 0; ks < KB; ++ks) acc = FE_MFMA(av[kbc * KB + ks], ring[jj][ks], acc);
                        }
                    });
                    if (kbi == NB - 1) {
                        const bool cok = 16 * nt + li < n2;
                        const int row = row_of(nt);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int st = s0 + 4 * lg + r;
                            if (cok && st < a.B) {
                                if constexpr (AGENT) __hip_atomic_store(pre + (size_t)st * (2 * kMlpRows) + row, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                else pre[(size_t)st * (2 * kMlpRows) + row] = acc[r];
                            }
                        }
                    }
                }
                load_item(itl + D, ring[jj], rbias[jj]);
            }
        }
    }
    }
}

// MULTI (r6): the instantiation for waves that walk several stream tiles (a.mlp_tpw > 1) - a kernel of its own: its 188 registers in one kernel with the one-tile
// path (142) cost that path a wave per SIMD (BASELINE config 5: 77.3 -> 79.2 us)
template <class S, bool MULTI = false>
__global__ void __launch_bounds__(kThreads) bsrnn_mlp_kernel(BArgs a) {
    constexpr int O1 = 4 * S::C, KS2 = O1 / 4, LDH = BMlpLds<S>::LDH;
    constexpr int KB = KS2 < 16 ? KS2 : 16, NB = KS2 / KB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = (int)blockIdx.x % (2 * kBands), tg = (int)blockIdx.x / (2 * kBands);
    const int kind = kb / kBands, band = kb - kind * kBands;
    // a wave takes `tpw` sixteen-stream tiles of its band, one after the other (large batches of the C = 16 shapes: the band's weights - all of
    // them in registers there - are fetched once per wave instead of once per tile: the kernel ran at a third of its MFMA time, on weight loads)
    // mlp_tpw = 0 (r5, small batches - fewer 64-stream groups than CUs): the four waves of a workgroup share ONE sixteen-stream tile and split
    // the band's layer-2 column tiles between them (layer 1 - 16 MFMAs - is computed by each): a wide band's five dependent 16-MFMA items were
    // the kernel's critical path
    const bool split = a.mlp_tpw == 0;
    const int tpw = a.mlp_tpw > 1 ? a.mlp_tpw : 1;
    const int tile0 = split ? tg : (tg * kWaves + wave) * tpw;
    if (tile0 * 16 >= a.B) return;                            // (wave-uniform; the kernel has no barrier)
    const int n2 = 4 * bsrnn_band_sub(band);
    const int nit_all = ((n2 + 15) >> 4) * NB;
    // (split: this wave's items are wave, wave + 4, ... of the band's list - whole column tiles: NB = 1 there)
    const bool wsplit = split && NB == 1;
    const int nit = wsplit ? (nit_all - wave + kWaves - 1) / kWaves : nit_all;
    if (wsplit && nit <= 0) return;                           // (a narrow band: fewer column tiles than waves)
    if constexpr (MULTI) bsrnn_mlp_wave_multi<S, false>(a, smem + wave * (16 * LDH), kind, band, tile0, tpw, 0, 1, nit, lane);
    else bsrnn_mlp_wave<S, false>(a, smem + wave * (16 * LDH), kind, band, tile0, tpw, wsplit ? wave : 0, wsplit ? kWaves : 1, nit, lane);
}

struct SbOffsets;
struct BImpl {
    int C, NLAY, HOP;
    size_t lds_bytes;
    size_t xp_floats;         // global scratch per workgroup (0: the input projections stay in LDS)
    size_t dbg_floats;
    int dbg_stages;
    bool whh_regs;            // W_hh packing: [rr][k'][128] floats (register shapes) or [rr][k'/4][128] float4 (streamed)
    bool ksplit;              // thread <-> (unit, K quarter) holding all four gates (BShape::KSPLIT) or (unit, gate) over the whole K
    int rec_threads;          // threads of one direction of the band recurrence (BShape::NTD)
    void (*launch)(const BArgs&, int max_wgs, hipStream_t, hipError_t*);
    void (*dbg_stage)(int, int*, int*, size_t*);
    void (*launch_pipe)(const BArgs&, hipStream_t, hipError_t*);       // time-pipelined offline launch (cooperative: B * pipe_p workgroups)
    int occ;                  // workgroups per CU the LDS plan allows
    void (*launch_split)(const BArgs&, int max_wgs, hipStream_t, hipError_t*);   // the per-hop step as three launches (mlp_x / mlp_sp / mlp_pre set)
    // the per-hop step with the layers batched over the streams (bsrnn_sb_kernels.hip.h): front (PART 3) -> layers -> mask-decoder MLP -> tail
    // (PART 2); nullptr where the stream-batched layers are not built (num_channels > 16)
    void (*launch_sb)(const BArgs&, const SbOffsets&, int total_floats, int max_wgs, hipStream_t, hipError_t*);
    const char* name = nullptr;      // the line of fe_bsrnn_shapes.def (fe_bsrnn_shape.hip.in): part of fe_last_step_kernel's answer
};

template <class S, bool HOT, bool PROF, bool DBG, bool OCC2 = false>
void blaunch_one(const BArgs& a, int grid, hipStream_t st, hipError_t* err) {
    static std::atomic<bool> attr_set[64];           // per device (see fe_impl.h::launch_one)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bsrnn_frame_kernel<S, HOT, PROF, DBG, OCC2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)BLds<S>::BYTES);
        if (e != hipSuccess) { *err = e; return; }
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    note_kernel(DBG ? "bsrnn_frame_kernel<debug>" : PROF ? "bsrnn_frame_kernel<profile>" : HOT ? (OCC2 ? "bsrnn_frame_kernel<per-hop, two workgroups per CU>" : "bsrnn_frame_kernel<per-hop>")
                    : (OCC2 ? "bsrnn_frame_kernel<generic, two workgroups per CU>" : "bsrnn_frame_kernel<generic>"));
    hipLaunchKernelGGL((bsrnn_frame_kernel<S, HOT, PROF, DBG, OCC2>), dim3(grid), dim3(kThreads), BLds<S>::BYTES, st, a);
    *err = hipGetLastError();
}

template <class S>
void blaunch_impl(const BArgs& a, int max_wgs, hipStream_t st, hipError_t* err) {
    constexpr bool FITS2 = 2 * BLds<S>::BYTES <= 160 * 1024 && !S::XPG;
    if (FITS2 && a.B > max_wgs && a.dbg == nullptr && a.clk == nullptr) {      // two workgroups per CU, persistent above 2 x #CUs streams
        const int grid2 = a.B < 2 * max_wgs ? a.B : 2 * max_wgs;
        if (a.mode == FE_MODE_STREAM && a.T == 1) blaunch_one<S, true, false, false, FITS2>(a, grid2, st, err);
        else blaunch_one<S, false, false, false, FITS2>(a, grid2, st, err);
        return;
    }
    const int grid = a.B < max_wgs ? a.B : max_wgs;      // more streams than CUs: persistent workgroups walk b, b + grid, ...
    if (a.dbg != nullptr) blaunch_one<S, false, false, true>(a, grid, st, err);                 // fe_debug_step
    else if (a.clk != nullptr) {                                                                // fe_profile_step
        if (a.mode == FE_MODE_STREAM && a.T == 1) blaunch_one<S, true, true, false>(a, grid, st, err);
        else blaunch_one<S, false, true, false>(a, grid, st, err);
    }
    else if (a.mode == FE_MODE_STREAM && a.T == 1) blaunch_one<S, true, false, false>(a, grid, st, err);
    else blaunch_one<S, false, false, false>(a, grid, st, err);
}

template <class S, bool OCC2, int PART>
void blaunch_part(const BArgs& a, int grid, hipStream_t st, hipError_t* err) {
    auto* fn = &bsrnn_frame_kernel<S, true, false, false, OCC2, false, PART>;
    using LP = BLds<S, (PART == 2 || PART == 3) ? PART : 0>;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LP::BYTES);
        if (e != hipSuccess) { *err = e; return; }
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    note_kernel(PART == 1 ? (OCC2 ? "bsrnn_frame_kernel<PART 1, two workgroups per CU>" : "bsrnn_frame_kernel<PART 1>")
                : PART == 2 ? "bsrnn_frame_kernel<PART 2>" : "bsrnn_frame_kernel<PART 3>");      // (r6: their own LDS plans, up to four workgroups per CU)
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads), LP::BYTES, st, a);
    *err = hipGetLastError();
}

// grid of a per-stream front / tail launch (PART 3 / 2): their own LDS plans fit four times per CU
template <class S, int PART>
int bpart_grid(int B, int max_wgs) {
    constexpr int OCC = PART == 2 ? 6 : 4;          // (tail: 82 VGPRs, 20.5 KB (xt); front: 106 VGPRs, 14.6 KB)
    static_assert(OCC * BLds<S, PART>::BYTES <= 160 * 1024, "front / tail workgroups per CU");
    return B < OCC * max_wgs ? B : OCC * max_wgs;
}

}  // namespace fe
#include "bsrnn_sb_kernels.hip.h"
#include "bsrnn_ov_kernels.hip.h"
namespace fe {

// the per-hop streaming step (mode = stream, T = 1) as head -> batched mask-decoder MLP -> tail
template <class S>
void blaunch_split_impl(const BArgs& a, int max_wgs, hipStream_t st, hipError_t* err) {
    constexpr bool FITS2 = 2 * BLds<S>::BYTES <= 160 * 1024 && !S::XPG;
    // r5: at most one stream per CU and num_channels = 16 - the role-split PART 1 (bsrnn_ov_kernels.hip.h: the scans alone on two
    // waves, the layers' matrix-core work under them on the other two).  a.ov_off (fe_set_step_kernel(WAVES4) / fe_set_option("bsrnn_role_split", 0)):
    // the phase-by-phase kernel.
    if (S::C == 16 && !a.ov_off && a.B <= max_wgs && a.clk != nullptr) blaunch_ov<S, true>(a, a.B, st, err);      // (fe_profile_step with fe_set_option("bsrnn_ov_profile", 1))
    else if (S::C == 16 && !a.ov_off && a.B <= max_wgs && a.gsync != nullptr) {
        // r6: the whole step in ONE cooperative launch (fe_set_option("bsrnn_fused_step", 1); MEASURED NEGATIVE - 81 us against 77, and the
        // cooperative launch adds ~21 us on top: profiles/r6_bsrnn_fused_step.txt - hence off by default); a launch the runtime refuses
        // (co-residency not guaranteed) falls through to the three launches
        blaunch_ov<S, false, true>(a, a.B, st, err);
        if (*err == hipSuccess) return;
        *err = hipSuccess;
        blaunch_ov<S>(a, a.B, st, err);
    }
    else if (S::C == 16 && !a.ov_off && a.B <= max_wgs) blaunch_ov<S>(a, a.B, st, err);
    else if (FITS2 && a.B > max_wgs) blaunch_part<S, FITS2, 1>(a, a.B < 2 * max_wgs ? a.B : 2 * max_wgs, st, err);
    else blaunch_part<S, false, 1>(a, a.B < max_wgs ? a.B : max_wgs, st, err);
    if (*err != hipSuccess) return;
    {
        auto* fn = &bsrnn_mlp_kernel<S>;
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!attr_set[dev].load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BMlpLds<S>::BYTES);
            if (e != hipSuccess) { *err = e; return; }
            attr_set[dev].store(true, std::memory_order_relaxed);
        }
        BArgs am = a;
        const bool msplit = S::C == 16 && a.B <= max_wgs;      // (fewer 64-stream groups than CUs: one sixteen-stream tile per workgroup, column tiles over its waves)
        am.mlp_tpw = msplit ? 0 : 1;
        const int groups = msplit ? (a.B + 15) / 16 : (a.B + 16 * kWaves - 1) / (16 * kWaves);
        note_kernel(msplit ? "bsrnn_mlp_kernel<one 16-stream tile per workgroup>" : "bsrnn_mlp_kernel");
        hipLaunchKernelGGL(fn, dim3(2 * kBands * groups), dim3(kThreads), BMlpLds<S>::BYTES, st, am);
        *err = hipGetLastError();
        if (*err != hipSuccess) return;
    }
    if (FITS2 && a.B > max_wgs) blaunch_part<S, FITS2, 2>(a, bpart_grid<S, 2>(a.B, max_wgs), st, err);
    else blaunch_part<S, false, 2>(a, bpart_grid<S, 2>(a.B, max_wgs), st, err);
}

template <class S, bool MULTI>
void blaunch_mlp_one(const BArgs& am, int groups, hipStream_t st, hipError_t* err) {
    auto* fn = &bsrnn_mlp_kernel<S, MULTI>;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BMlpLds<S>::BYTES);
        if (e != hipSuccess) { *err = e; return; }
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(fn, dim3(2 * kBands * groups), dim3(kThreads), BMlpLds<S>::BYTES, st, am);
    *err = hipGetLastError();
}

template <class S>
void blaunch_mlp(const BArgs& a, hipStream_t st, hipError_t* err) {
    BArgs am = a;
    am.mlp_tpw = (S::C == 16 && a.B >= 2048) ? 4 : 1;       // (C = 16: KS2 = 16 k-steps = one burst per item, at most five items = the ring)
    const int groups = (a.B + 16 * kWaves * am.mlp_tpw - 1) / (16 * kWaves * am.mlp_tpw);
    note_kernel(am.mlp_tpw == 4 ? "bsrnn_mlp_kernel<four tiles per wave>" : "bsrnn_mlp_kernel");
    if constexpr (S::C == 16) {         // (the multi-tile instantiation exists for the shapes that use it)
        if (am.mlp_tpw > 1) { blaunch_mlp_one<S, true>(am, groups, st, err); return; }
    }
    blaunch_mlp_one<S, false>(am, groups, st, err);
}

// the per-hop step of a LARGE batch: front per stream (PART 3), the LSTM layers for sixteen streams per workgroup on the matrix
// cores (bsrnn_sb_layers_kernel), the batched mask-decoder MLP, the tail per stream (PART 2)
template <class S>
void blaunch_sb_impl(const BArgs& a, const SbOffsets& so, int total_floats, int max_wgs, hipStream_t st, hipError_t* err) {
    constexpr bool FITS2 = 2 * BLds<S>::BYTES <= 160 * 1024 && !S::XPG;
    if (FITS2 && a.B > max_wgs) blaunch_part<S, FITS2, 3>(a, bpart_grid<S, 3>(a.B, max_wgs), st, err);
    else blaunch_part<S, false, 3>(a, bpart_grid<S, 3>(a.B, max_wgs), st, err);
    if (*err != hipSuccess) return;
    SbArgs sa{};
    sa.wp = a.wp; sa.off = so; sa.x = a.mlp_x; sa.lstm = a.lstm; sa.y = a.sb_y; sa.B = a.B; sa.total = total_floats;
    sb_launch_layers<S>(sa, st, err);
    if (*err != hipSuccess) return;
    blaunch_mlp<S>(a, st, err);
    if (*err != hipSuccess) return;
    if (FITS2 && a.B > max_wgs) blaunch_part<S, FITS2, 2>(a, bpart_grid<S, 2>(a.B, max_wgs), st, err);
    else blaunch_part<S, false, 2>(a, bpart_grid<S, 2>(a.B, max_wgs), st, err);
}

template <class S>
void blaunch_pipe_impl(const BArgs& a, hipStream_t st, hipError_t* err) {
    auto* fn = &bsrnn_frame_kernel<S, false, false, false, false, true>;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BLds<S>::BYTES);
        if (e != hipSuccess) { *err = e; return; }
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    BArgs args = a;
    void* kargs[] = {&args};
    note_kernel("bsrnn_frame_kernel<time-pipelined>");
    *err = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(fn), dim3(a.B * a.pipe_p), dim3(kThreads), kargs, (unsigned int)BLds<S>::BYTES, st);
}

template <class S>
void bdbg_stage_impl(int s, int* rows, int* cols, size_t* off) {
    *rows = BDebugLayout<S>::rows(s);
    *cols = BDebugLayout<S>::cols(s);
    *off = BDebugLayout<S>::offset(s);
}

template <class S>
BImpl make_bimpl() {
    return BImpl{S::C, S::NLAY, S::HOP, BLds<S>::BYTES, S::XPG ? (size_t)2 * 32 * S::G4 : (size_t)0,
                 BDebugLayout<S>::total(), BDebugLayout<S>::n_stages, S::WREG, S::KSPLIT, S::NTD, &blaunch_impl<S>, &bdbg_stage_impl<S>,
                 &blaunch_pipe_impl<S>, 1,       // (the PIPE instantiation is compiled for one workgroup per CU: waves_per_eu(1, 1))
                 &blaunch_split_impl<S>, (SbLds<S>::FITS || Sb64Lds<S>::FITS) ? &blaunch_sb_impl<S> : nullptr};
}

}  // namespace fe
