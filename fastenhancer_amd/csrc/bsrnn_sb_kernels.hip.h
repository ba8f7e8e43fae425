// bsrnn_sb_kernels.hip.h — BSRNN's LSTM layers BATCHED OVER THE STREAMS on the fp32 matrix cores (gfx950), for the per-hop step of
// large batches (models/bsrnn/model.py:367-390: per layer a time-LSTM cell over the 31 bands + fc_time, a bidirectional LSTM over
// the bands + fc_freq, both with residuals).
//
// bsrnn_frame_kernel gives a stream a workgroup: every product of the band recurrence is then M = 1 - 186 dependent steps of vector
// FMAs per frame, a latency chain (13 % of the fp32 peak at one stream per CU, 22 % with two workgroups per CU).  Here a workgroup
// takes SIXTEEN streams and every product is a matrix-core GEMM with the streams as the N dimension, computed TRANSPOSED:
//     gates^T [4 HH x 16 streams] = W [4 HH x K] . in^T [K x 16 streams]
// A = the weights (fragments in registers for a whole layer and direction), B = the activations: lane (li, lg) of a wave holds
// feature 4 ks + lg of stream li for k-step ks, C/D = lane (li, lg) holds rows 4 lg + r of stream li.  The weight rows are packed in
// the order that makes the two coincide - row 4 lg + r of output tile t = (hidden unit KSH lg + t, gate r) - so that
//   * the four gates of a (stream, hidden unit) are the four accumulator registers of ONE lane: the cell update is lane-local,
//   * the new h of unit KSH lg + t in lane (li, lg) IS the B operand of k-step t of the next product (the feature a k-step's lane
//     group carries is a free choice as long as the A operand uses the same one: feature KSH lg + ks): h feeds the next step, fc_time
//     and fc_freq without a transpose,
//   * a lane's hidden units are CONTIGUOUS in the reference's state tensors ([.., HH], units KSH lg .. KSH lg + KSH - 1): the
//     time-LSTM's (h, c) - the streaming caches - move as 16-byte loads / stores of whole 128-byte lines per stream.
// Time LSTM + fc_time: the bands are a batch - wave w takes bands w, w + 4, ... and runs each band's 16 streams through all
// 4 HH gate rows with no barrier, the next band's state in flight under the current band's MFMAs.  Band LSTM: sequential over the
// bands; both directions run TOGETHER - waves 0-3 forward, waves 4-7 backward, two waves per SIMD - each direction's four waves split
// the gate rows (hidden units) of a step and exchange the new h through a 2 KiB LDS buffer (one barrier per step for both
// directions), the x half of the NEXT step's gates is issued before that barrier; the outputs y go to a global scratch (a CU's own
// lines) and fc_freq + residual runs after the 31 steps, batched over the bands.  The band features x of the 16 streams
// (31 x C x 16 floats) live in LDS (36 KiB in all: two workgroups per CU).  Built for num_channels = 16 (xt, xxt); the per-stream
// kernel keeps the other sizes.  (First version, r4: four waves, the directions one after the other, y in LDS: 406 us for the layers
// of 4096 streams, a band step ~2 k cycles for 24 MFMAs per wave.)
// (included by bsrnn_kernels.hip.h, after BShape / kBands / the fe_kernels.hip.h helpers)
#pragma once

namespace fe {

constexpr int kSbStreams = 16;      // streams per workgroup = the N dimension of a 16x16x4 tile
constexpr int kSbThreads = 512;     // eight waves: two per SIMD - in the band LSTM waves 0-3 run the forward direction, waves 4-7 the backward one
constexpr int kSbWaves = 8;

template <class S>
struct SbLds {
    // r6: num_channels = 32 (bsrnn_t).  A layer's time-LSTM fragments - 4 HH x (C + HH) = 96 KiB - fit neither a wave's registers (384 per lane) nor
    // the LDS next to the band features: TSPLIT - the eight waves split the GATE TILES (hidden units) of every band instead of the bands, and the
    // new h of a group of GB bands meets in LDS for fc_time (see the kernel).
    static constexpr bool TSPLIT = S::C > 16;
    static constexpr int GB = 8;                                          // bands per group of the TSPLIT time part (one fc_time band per wave)
    static constexpr int XS = 0;                                         // [31][C][16]   band features, position p = 4 ks + lg <-> channel KSC lg + ks
    static constexpr int HB = XS + kBands * S::C * kSbStreams;            // [2 directions][2][HH][16]   h of the running step (double buffer)
                                                                          // TSPLIT, time part: [GB][HH][16] new h of a band group (position 4 t + lg <-> unit KSH lg + t)
    static constexpr int HB_N = (TSPLIT && GB > 4 ? GB : 4) * S::HH * kSbStreams;
    // the time LSTM's and fc_time's fragments + start values of the running layer - the same for all eight waves: copied once per workgroup
    // (each wave fetching its own copy: 1.1 k wave-level loads per workgroup and layer on the CU's vector-memory path)
    static constexpr int WT = HB + HB_N;
    static constexpr int WT_W = 0, WT_FC = WT_W + (S::HH / 4) * (S::C / 4 + S::HH / 4) * 64, WT_B = WT_FC + (S::C / 16) * (S::HH / 4) * 64,
                         WT_FCB = WT_B + (S::HH / 4) * 16, WT_N = WT_FCB + (S::C / 16) * 16;
    static constexpr int TOTAL = WT + (TSPLIT ? 0 : WT_N);
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
    // (num_channels = 64: the band features of sixteen streams alone are 127 KiB and a direction's gate fragments 384 registers per lane - the per-stream kernel keeps it)
    static constexpr bool FITS = (S::C == 16 || S::C == 32) && BYTES <= 160 * 1024;     // (num_channels = 64: bsrnn_sb64_layers_kernel below, Sb64Lds)
};

// offsets (floats) of the stream-batched layer weights inside the packed buffer (host: fe_api.hip::pack_weights_bsrnn)
struct SbOffsets {
    int t_w[8], t_b[8];           // time LSTM: A fragments [tile t < KSH][k-step < KSC + KSH][64]; start values [t][lg][r] (b_ih + b_hh, pre-scaled)
    int tfc_w[8], tfc_b[8];       // fc_time: A fragments [tile < C / 16][k-step < KSH][64]; bias [tile][lg][r]
    int f_w[8][2], f_b[8][2];     // band LSTM per direction: as the time LSTM's
    int ffc_w[8][2], ffc_b[8];    // fc_freq per direction half: A fragments [tile][k-step < KSH][64]; bias
    // num_channels = 64 (bsrnn_sb64_layers_kernel): every matrix once more in k4 order [tile][k-step / 4][lane][4] - 16-byte fragment fetches
    int f_wx4[8][2], f_wh4[8][2];  // band LSTM: the x k-steps (streamed every step), the h k-steps
    int t_w4[8], tfc_w4[8], ffc_w4[8][2];
};

struct SbArgs {
    const float* wp;
    SbOffsets off;
    float* x;                     // [B][31][C] band features, in place (bsrnn_frame_kernel PART 3 -> this kernel -> bsrnn_mlp_kernel)
    float* lstm;                  // [2 NLAY][B * 31][HH] time-LSTM caches (h0, c0, h1, c1, ...), the reference's tensors
    float* y;                     // [B][2][31][HH] scratch: the band LSTM's outputs of the running layer (written by the steps, read by fc_freq)
    int B;
    int total;                    // floats of the packed buffer
};

// sigma / tanh of pre-scaled pre-activations (the packer multiplies the i, f, o rows by -log2 e and the g rows by -2 log2 e)
__device__ __forceinline__ float sb_sig(float pre) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre)); }
__device__ __forceinline__ float sb_tanh_pre(float pre) { return __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre)), -1.0f); }

template <class S>
__global__ void __launch_bounds__(kSbThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) bsrnn_sb_layers_kernel(SbArgs a) {
    static_assert(SbLds<S>::FITS, "stream-batched BSRNN layers: built for num_channels = 16 / 32");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = SbLds<S>;
    constexpr int C = S::C, HH = S::HH, KSC = S::KSC, KSH = S::KSH, KS1 = KSC + KSH, NS = kSbStreams;
    constexpr int NTO = C / 16;                 // output tiles of the fc layers
    constexpr int NTW = KSH / 4;                // gate tiles (of 4 hidden units) per wave in the band recurrence (four waves per direction)
    static_assert(KSH % 4 == 0 && KSC % 4 == 0, "tiles per wave / 16-byte state rows");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const SbOffsets& o = a.off;
    WSrc<false> wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, a.total * 4, 0x00020000);
    wb.lane4 = lane * 4;
    wb.li4 = li * 4;
    wb.lds = nullptr;
    wb.base = 0;
    wb.k4d = 0;
    float* xs = smem + L::XS;
    const int b0 = (int)blockIdx.x * NS;
    const int bs = b0 + li < a.B ? b0 + li : a.B - 1;          // this lane's stream (the last tile's idle columns shadow the last stream)
    const bool sok = b0 + li < a.B;

    // ---- the tile's band features: global [b][31][C] -> LDS [31][p][16], p = 4 ks + lg <-> channel KSC lg + ks
    for (int i = tid; i < NS * kBands * (C / 4); i += kSbThreads) {
        const int s = i / (kBands * (C / 4)), q = i - s * (kBands * (C / 4)), j = q / (C / 4), c4 = q - j * (C / 4);
        const int b = b0 + s < a.B ? b0 + s : a.B - 1;
        const float4 v = *reinterpret_cast<const float4*>(a.x + ((size_t)b * kBands + j) * C + 4 * c4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * c4 + e, lgc = c / KSC, ks = c - lgc * KSC;
            xs[(j * C + 4 * ks + lgc) * NS + s] = vv[e];
        }
    }
    __syncthreads();

    // fragment helpers: A fragment (tile, k-step) of a packed matrix with KST k-steps per tile; start values [tile][lg][r]
    auto afrag = [&](int off, int tile, int kst, int ks) { return wb.at_g(off + (tile * kst + ks) * 64); };
    auto bias4 = [&](int off, int tile) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wb.rsrc, lg * 16, (off + tile * 16) * 4, 0)); };
    // this lane's KSH hidden units of (stream bs, band j): contiguous in the state tensors / in the y scratch
    auto soff = [&](int j) { return ((size_t)bs * kBands + j) * HH + KSH * lg; };
    auto yoff = [&](int d, int j) { return (((size_t)bs * 2 + d) * kBands + j) * HH + KSH * lg; };

#pragma unroll 1
    for (int l = 0; l < S::NLAY; ++l) {
        // ======================================= time LSTM + fc_time: bands over the waves, no barrier =======================================
        if constexpr (!L::TSPLIT)
        {
            float* wtl = smem + L::WT;
            {
                auto copy4 = [&](int dst, int off, int n) {
                    const f32x4* src = reinterpret_cast<const f32x4*>(a.wp + off);
                    for (int i = tid; i < n / 4; i += kSbThreads) reinterpret_cast<f32x4*>(wtl + dst)[i] = src[i];
                };
                copy4(L::WT_W, o.t_w[l], KSH * KS1 * 64);
                copy4(L::WT_FC, o.tfc_w[l], NTO * KSH * 64);
                copy4(L::WT_B, o.t_b[l], KSH * 16);
                copy4(L::WT_FCB, o.tfc_b[l], NTO * 16);
            }
            __syncthreads();
            float Wt[KSH][KS1], Wf1[NTO][KSH];
            f32x4 Wf1b[NTO];
#pragma unroll
            for (int t = 0; t < KSH; ++t)
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) Wt[t][ks] = wtl[L::WT_W + (t * KS1 + ks) * 64 + lane];
#pragma unroll
            for (int to = 0; to < NTO; ++to) {
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) Wf1[to][ks] = wtl[L::WT_FC + (to * KSH + ks) * 64 + lane];
                Wf1b[to] = *reinterpret_cast<const f32x4*>(wtl + L::WT_FCB + to * 16 + 4 * lg);
            }
            float* hg = a.lstm + (size_t)(2 * l) * a.B * (kBands * HH);
            float* cg = a.lstm + (size_t)(2 * l + 1) * a.B * (kBands * HH);
            f32x4 hq[KSH / 4], cq[KSH / 4];
            auto fetch_state = [&](int j) {
#pragma unroll
                for (int q = 0; q < KSH / 4; ++q) {
                    hq[q] = *reinterpret_cast<const f32x4*>(hg + soff(j) + 4 * q);
                    cq[q] = *reinterpret_cast<const f32x4*>(cg + soff(j) + 4 * q);
                }
            };
            if (wave < kBands) fetch_state(wave);
#pragma unroll 1
            for (int j = wave; j < kBands; j += kSbWaves) {
                float hp[KSH], cp[KSH], xb[KSC];
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) { hp[ks] = hq[ks / 4][ks % 4]; cp[ks] = cq[ks / 4][ks % 4]; }
                if (j + kSbWaves < kBands) fetch_state(j + kSbWaves);      // the next band's state: in flight under this band's MFMAs
#pragma unroll
                for (int ks = 0; ks < KSC; ++ks) xb[ks] = xs[(j * C + 4 * ks + lg) * NS + li];
                f32x4 acc[KSH];
#pragma unroll
                for (int t = 0; t < KSH; ++t) acc[t] = *reinterpret_cast<const f32x4*>(wtl + L::WT_B + t * 16 + 4 * lg);     // (held in registers they would not fit next to the fragments)
#pragma unroll
                for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
                    for (int t = 0; t < KSH; ++t) acc[t] = FE_MFMA(Wt[t][ks], xb[ks], acc[t]);
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks)
#pragma unroll
                    for (int t = 0; t < KSH; ++t) acc[t] = FE_MFMA(Wt[t][KSC + ks], hp[ks], acc[t]);
                float hn[KSH], cn[KSH];
#pragma unroll
                for (int t = 0; t < KSH; ++t) {                              // gate order i, f, g, o (nn.LSTMCell)
                    const float ig = sb_sig(acc[t][0]), fg = sb_sig(acc[t][1]), gg = sb_tanh_pre(acc[t][2]), og = sb_sig(acc[t][3]);
                    cn[t] = fg * cp[t] + ig * gg;
                    hn[t] = og * tanh_f(cn[t]);
                }
                if (sok) {
#pragma unroll
                    for (int q = 0; q < KSH / 4; ++q) {
                        *reinterpret_cast<f32x4*>(hg + soff(j) + 4 * q) = f32x4{hn[4 * q], hn[4 * q + 1], hn[4 * q + 2], hn[4 * q + 3]};
                        *reinterpret_cast<f32x4*>(cg + soff(j) + 4 * q) = f32x4{cn[4 * q], cn[4 * q + 1], cn[4 * q + 2], cn[4 * q + 3]};
                    }
                }
                // fc_time + residual: output tile `to`, register r <-> channel KSC lg + 4 to + r = the B operand slot of k-step 4 to + r
#pragma unroll
                for (int to = 0; to < NTO; ++to) {
                    f32x4 a2 = Wf1b[to];
#pragma unroll
                    for (int ks = 0; ks < KSH; ++ks) a2 = FE_MFMA(Wf1[to][ks], hn[ks], a2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) xs[(j * C + 4 * (4 * to + r) + lg) * NS + li] = xb[4 * to + r] + a2[r];
                }
            }
        }
        else {
            // ======================================= TSPLIT (num_channels = 32): gate tiles over the waves, bands in groups of GB =======================================
            // wave w owns gate tiles t = TPW w .. TPW w + TPW - 1 (hidden units KSH lg + t of lane group lg) of EVERY band: its fragments are TPW x KS1
            // registers.  Per group of GB bands: every wave runs its tiles of the group's bands (h_{t-1} of a band straight from the state tensor - a lane's
            // KSH units are contiguous there -, the next band's in flight under this band's MFMAs; c stays with the owning lane), the new h goes to LDS
            // [band][position 4 t + lg][16 streams]; barrier; wave w then takes band g0 + w: reads the band's whole new h back as B operands (k-step t <-> unit
            // KSH lg + t), writes it to the state tensor as 16-byte pieces (only now: the other waves have read h_{t-1} of this band before the barrier) and
            // runs fc_time + the residual; barrier (the next group overwrites the h buffer).
            constexpr int TPW = KSH / kSbWaves, GB = L::GB;
            static_assert(KSH % kSbWaves == 0 && (TPW == 1 || TPW == 2 || TPW == 4), "gate tiles per wave");
            float* hnl = smem + L::HB;
            float Wt[TPW][KS1], Wf1[NTO][KSH];
            f32x4 Wtb[TPW], Wf1b[NTO];
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) {
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) Wt[tt][ks] = afrag(o.t_w[l], wave * TPW + tt, KS1, ks);
                Wtb[tt] = bias4(o.t_b[l], wave * TPW + tt);
            }
#pragma unroll
            for (int to = 0; to < NTO; ++to) {
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) Wf1[to][ks] = afrag(o.tfc_w[l], to, KSH, ks);
                Wf1b[to] = bias4(o.tfc_b[l], to);
            }
            float* hg = a.lstm + (size_t)(2 * l) * a.B * (kBands * HH);
            float* cg = a.lstm + (size_t)(2 * l + 1) * a.B * (kBands * HH);
            f32x4 hq[KSH / 4];
            float cq[TPW];
            auto fetch_state = [&](int j) {
#pragma unroll
                for (int q = 0; q < KSH / 4; ++q) hq[q] = *reinterpret_cast<const f32x4*>(hg + soff(j) + 4 * q);
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) cq[tt] = cg[soff(j) + wave * TPW + tt];
            };
#pragma unroll 1
            for (int g0 = 0; g0 < kBands; g0 += GB) {
                const int nb = kBands - g0 < GB ? kBands - g0 : GB;
                fetch_state(g0);
#pragma unroll 1
                for (int jl = 0; jl < nb; ++jl) {
                    const int j = g0 + jl;
                    float hp[KSH], cp[TPW], xb[KSC];
#pragma unroll
                    for (int ks = 0; ks < KSH; ++ks) hp[ks] = hq[ks / 4][ks % 4];
#pragma unroll
                    for (int tt = 0; tt < TPW; ++tt) cp[tt] = cq[tt];
                    if (jl + 1 < nb) fetch_state(j + 1);
#pragma unroll
                    for (int ks = 0; ks < KSC; ++ks) xb[ks] = xs[(j * C + 4 * ks + lg) * NS + li];
                    f32x4 acc[TPW];
#pragma unroll
                    for (int tt = 0; tt < TPW; ++tt) acc[tt] = Wtb[tt];
#pragma unroll
                    for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
                        for (int tt = 0; tt < TPW; ++tt) acc[tt] = FE_MFMA(Wt[tt][ks], xb[ks], acc[tt]);
#pragma unroll
                    for (int ks = 0; ks < KSH; ++ks)
#pragma unroll
                        for (int tt = 0; tt < TPW; ++tt) acc[tt] = FE_MFMA(Wt[tt][KSC + ks], hp[ks], acc[tt]);
#pragma unroll
                    for (int tt = 0; tt < TPW; ++tt) {                           // gate order i, f, g, o (nn.LSTMCell)
                        const float ig = sb_sig(acc[tt][0]), fg = sb_sig(acc[tt][1]), gg = sb_tanh_pre(acc[tt][2]), og = sb_sig(acc[tt][3]);
                        const float cn = fg * cp[tt] + ig * gg;
                        const float hn = og * tanh_f(cn);
                        if (sok) cg[soff(j) + wave * TPW + tt] = cn;
                        hnl[((jl * HH) + 4 * (wave * TPW + tt) + lg) * NS + li] = hn;
                    }
                }
                __syncthreads();
                if (wave < nb) {
                    const int j = g0 + wave;
                    float hn[KSH], xb[KSC];
#pragma unroll
                    for (int ks = 0; ks < KSH; ++ks) hn[ks] = hnl[((wave * HH) + 4 * ks + lg) * NS + li];
#pragma unroll
                    for (int ks = 0; ks < KSC; ++ks) xb[ks] = xs[(j * C + 4 * ks + lg) * NS + li];
                    if (sok) {
#pragma unroll
                        for (int q = 0; q < KSH / 4; ++q)
                            *reinterpret_cast<f32x4*>(hg + soff(j) + 4 * q) = f32x4{hn[4 * q], hn[4 * q + 1], hn[4 * q + 2], hn[4 * q + 3]};
                    }
#pragma unroll
                    for (int to = 0; to < NTO; ++to) {
                        f32x4 a2 = Wf1b[to];
#pragma unroll
                        for (int ks = 0; ks < KSH; ++ks) a2 = FE_MFMA(Wf1[to][ks], hn[ks], a2);
#pragma unroll
                        for (int r = 0; r < 4; ++r) xs[(j * C + 4 * (4 * to + r) + lg) * NS + li] = xb[4 * to + r] + a2[r];
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
        // ======================================= band LSTM: waves 0-3 forward, waves 4-7 backward, 31 steps together =======================================
        {
            const int d = wave >> 2, wq = wave & 3;         // direction, this wave's quarter of the gate rows
            float* hb = smem + L::HB + d * (2 * HH * NS);
            float Wb[NTW][KS1];
            f32x4 Wbb[NTW];
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) {
                const int t = wq + 4 * tt;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) Wb[tt][ks] = afrag(o.f_w[l][d], t, KS1, ks);
                Wbb[tt] = bias4(o.f_b[l][d], t);
            }
            for (int i = tid & 255; i < HH * NS; i += 256) hb[i] = 0.0f;         // h = 0 (buffer 0 of this direction)
            float cst[NTW];
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) cst[tt] = 0.0f;
            f32x4 accx[NTW];
            auto xpart = [&](int j) {                         // the x half of a step's gates
                float xb[KSC];
#pragma unroll
                for (int ks = 0; ks < KSC; ++ks) xb[ks] = xs[(j * C + 4 * ks + lg) * NS + li];
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) accx[tt] = Wbb[tt];
#pragma unroll
                for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
                    for (int tt = 0; tt < NTW; ++tt) accx[tt] = FE_MFMA(Wb[tt][ks], xb[ks], accx[tt]);
            };
            xpart(d ? kBands - 1 : 0);
            __syncthreads();
            int cur = 0;
#pragma unroll 1
            for (int s = 0; s < kBands; ++s) {
                const int j = d ? kBands - 1 - s : s;
                const float* hc = hb + cur * (HH * NS);
                float hf[KSH];
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) hf[ks] = hc[(4 * ks + lg) * NS + li];
                f32x4 acc[NTW];
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) acc[tt] = accx[tt];
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks)
#pragma unroll
                    for (int tt = 0; tt < NTW; ++tt) acc[tt] = FE_MFMA(Wb[tt][KSC + ks], hf[ks], acc[tt]);
                float* hnx = hb + (cur ^ 1) * (HH * NS);
                float* yg = a.y + yoff(d, j);
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) {
                    const float ig = sb_sig(acc[tt][0]), fg = sb_sig(acc[tt][1]), gg = sb_tanh_pre(acc[tt][2]), og = sb_sig(acc[tt][3]);
                    cst[tt] = fg * cst[tt] + ig * gg;
                    const float hv = og * tanh_f(cst[tt]);
                    const int t = wq + 4 * tt;
                    hnx[(4 * t + lg) * NS + li] = hv;                       // unit KSH lg + t <-> position 4 t + lg
                    yg[t] = hv;                                             // (the last tile's idle columns shadow the last stream: same value)
                }
                if (s + 1 < kBands) xpart(d ? kBands - 2 - s : s + 1);      // the x half of the next step: before the barrier
                __syncthreads();
                cur ^= 1;
            }
        }
        // ======================================= fc_freq + residual, batched over the bands: x += b + W [y_fwd | y_bwd] =======================================
        {
            float Wf2[2][NTO][KSH];
            f32x4 Wf2b[NTO];
#pragma unroll
            for (int to = 0; to < NTO; ++to) {
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int ks = 0; ks < KSH; ++ks) Wf2[dd][to][ks] = afrag(o.ffc_w[l][dd], to, KSH, ks);
                Wf2b[to] = bias4(o.ffc_b[l], to);
            }
            // (the y scratch was written by other waves of this workgroup before the barrier above: workgroup-scope visibility through the CU's L1 is
            //  what the barrier's release / acquire gives; the lines are this CU's own)
#pragma unroll 1
            for (int j = wave; j < kBands; j += kSbWaves) {
                f32x4 yq[2][KSH / 4];
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int q = 0; q < KSH / 4; ++q) yq[dd][q] = *reinterpret_cast<const f32x4*>(a.y + yoff(dd, j) + 4 * q);
#pragma unroll
                for (int to = 0; to < NTO; ++to) {
                    f32x4 a2 = Wf2b[to];
#pragma unroll
                    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                        for (int ks = 0; ks < KSH; ++ks) a2 = FE_MFMA(Wf2[dd][to][ks], yq[dd][ks / 4][ks % 4], a2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) xs[(j * C + 4 * (4 * to + r) + lg) * NS + li] += a2[r];
                }
            }
        }
        __syncthreads();
    }
    // ---- back to global [b][31][C]
    for (int i = tid; i < NS * kBands * (C / 4); i += kSbThreads) {
        const int s = i / (kBands * (C / 4)), q = i - s * (kBands * (C / 4)), j = q / (C / 4), c4 = q - j * (C / 4);
        if (b0 + s < a.B) {
            float vv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * c4 + e, lgc = c / KSC, ks = c - lgc * KSC;
                vv[e] = xs[(j * C + 4 * ks + lgc) * NS + s];
            }
            *reinterpret_cast<float4*>(a.x + ((size_t)(b0 + s) * kBands + j) * C + 4 * c4) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        }
    }
}


// ======================================================================== num_channels = 64 (bsrnn_s), r6
// The band features of sixteen streams are 127 KiB and a direction's gate fragments 384 registers per lane: neither the LDS-resident features nor
// the register-resident direction of the kernel above carry over.  What does: the transposed products and the packer's row / k orders.
//   * the band features stay in global memory ([B][31][C], L2): a lane's B operands of a band - channels KSC lg .. KSC lg + 15 - are 64 contiguous
//     bytes there (four 16-byte loads), and an fc layer's accumulator quadruple (channels KSC lg + 4 to + r) is one 16-byte store: no regrouping at all
//   * time LSTM: wave w owns gate tiles 4 w .. 4 w + 3 of every band, two at a time (96 fragment registers); bands in two groups of sixteen, the new
//     h of a group in LDS (128 KiB) for fc_time, whose (band, output tile) jobs are spread over the waves
//   * band LSTM: ONE direction at a time on all eight waves (62 barrier steps per layer): a wave's four gate tiles keep their h fragments in registers
//     (128); the x fragments (64) are streamed from L2 every step as 16-byte pieces of a k4-ordered copy, for the NEXT step's x half
//   * fc_freq: (band, output tile) jobs over the waves, y through the global scratch
template <class S>
struct Sb64Lds {
    static constexpr int GB = 16;
    static constexpr int HN = 0;                                         // time part: [GB][HH][16] new h of a band group (position 4 t + lg <-> unit KSH lg + t)
    static constexpr int HB = 0;                                         // band LSTM: [2][HH][16] h of the running step (double buffer), start values [KSH][16] behind it
    static constexpr int BB = HB + 2 * S::HH * kSbStreams;
    static constexpr int TOTAL = GB * S::HH * kSbStreams;
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
    static constexpr bool FITS = S::C == 64 && BYTES <= 160 * 1024;
};

template <class S>
__global__ void __launch_bounds__(kSbThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) bsrnn_sb64_layers_kernel(SbArgs a) {
    static_assert(S::C == 64, "the num_channels = 64 plan");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = Sb64Lds<S>;
    constexpr int C = S::C, HH = S::HH, KSC = S::KSC, KSH = S::KSH, KS1 = KSC + KSH, NS = kSbStreams, GB = L::GB;
    static_assert(KSC == 16 && KSH == 32, "tile plans below");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const SbOffsets& o = a.off;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, a.total * 4, 0x00020000);
    // (a section's base is the instruction's scalar offset; tile and k-step go into the vector offset, whose constant part the instruction carries:
    //  with everything in the scalar offset every fragment had a scalar register of its own - 148 of them spilled)
    //  `lzv`: an opaque zero, renewed before every fetch burst - the vector offsets' bases are loop-invariant, and hoisted out of the layer loop they
    //  stayed live through the whole kernel: 205 spilled registers.  Fragments come as 16-byte pieces of the k4-ordered copies: a quarter of the fetches.)
    int lzv = 0;
    auto frag4 = [&](int off, int tile, int nq, int q) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (lane + lzv) * 16 + (tile * nq + q) * 1024, off * 4, 0)); };
    auto bias4 = [&](int off, int tile) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (lg + lzv) * 16 + tile * 64, off * 4, 0)); };
#define SB64_RENEW() asm volatile("" : "+v"(lzv))
    const int b0 = (int)blockIdx.x * NS;
    const int bs = b0 + li < a.B ? b0 + li : a.B - 1;          // this lane's stream (the last tile's idle columns shadow the last stream)
    const bool sok = b0 + li < a.B;
    float* const xg = a.x + (size_t)bs * kBands * C + KSC * lg;                    // + j * C: this lane's sixteen channels of band j
    auto soff = [&](int j) { return ((size_t)bs * kBands + j) * HH + KSH * lg; };
    auto yoff = [&](int d, int j) { return (((size_t)bs * 2 + d) * kBands + j) * HH + KSH * lg; };
    const int to = wave & 3, jw = wave >> 2;                   // fc layers: this wave's output tile, its bands jw, jw + 2, ..

#pragma unroll 1
    for (int l = 0; l < S::NLAY; ++l) {
        float* hg = a.lstm + (size_t)(2 * l) * a.B * (kBands * HH);
        float* cg = a.lstm + (size_t)(2 * l + 1) * a.B * (kBands * HH);
        // ======================================= time LSTM + fc_time =======================================
        {
            float* hnl = smem + L::HN;
#pragma unroll 1
            for (int g0 = 0; g0 < kBands; g0 += GB) {
                const int nb = kBands - g0 < GB ? kBands - g0 : GB;
#pragma unroll 1
                for (int sp = 0; sp < 2; ++sp) {
                    const int t0 = wave * 4 + sp * 2;
                    SB64_RENEW();
                    f32x4 Wt[2][KS1 / 4];
                    f32x4 Wtb[2];
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                        for (int q = 0; q < KS1 / 4; ++q) Wt[tt][q] = frag4(o.t_w4[l], t0 + tt, KS1 / 4, q);
                        Wtb[tt] = bias4(o.t_b[l], t0 + tt);
                    }
                    f32x4 xq[KSC / 4], hq[KSH / 4];
                    float cq[2];
                    auto fetch = [&](int j) {
#pragma unroll
                        for (int q = 0; q < KSC / 4; ++q) xq[q] = *reinterpret_cast<const f32x4*>(xg + j * C + 4 * q);
#pragma unroll
                        for (int q = 0; q < KSH / 4; ++q) hq[q] = *reinterpret_cast<const f32x4*>(hg + soff(j) + 4 * q);
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) cq[tt] = cg[soff(j) + t0 + tt];
                    };
                    fetch(g0);
#pragma unroll 1
                    for (int jl = 0; jl < nb; ++jl) {
                        const int j = g0 + jl;
                        float xb[KSC], hp[KSH], cp[2];
#pragma unroll
                        for (int ks = 0; ks < KSC; ++ks) xb[ks] = xq[ks / 4][ks % 4];
#pragma unroll
                        for (int ks = 0; ks < KSH; ++ks) hp[ks] = hq[ks / 4][ks % 4];
                        cp[0] = cq[0]; cp[1] = cq[1];
                        if (jl + 1 < nb) fetch(j + 1);                  // the next band's operands: in flight under this band's MFMAs
                        f32x4 acc[2] = {Wtb[0], Wtb[1]};
#pragma unroll
                        for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt) acc[tt] = FE_MFMA(Wt[tt][ks / 4][ks % 4], xb[ks], acc[tt]);
#pragma unroll
                        for (int ks = 0; ks < KSH; ++ks)
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt) acc[tt] = FE_MFMA(Wt[tt][(KSC + ks) / 4][ks % 4], hp[ks], acc[tt]);
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {                   // gate order i, f, g, o (nn.LSTMCell)
                            const float ig = sb_sig(acc[tt][0]), fg = sb_sig(acc[tt][1]), gg = sb_tanh_pre(acc[tt][2]), og = sb_sig(acc[tt][3]);
                            const float cn = fg * cp[tt] + ig * gg;
                            const float hn = og * tanh_f(cn);
                            if (sok) cg[soff(j) + t0 + tt] = cn;
                            hnl[((jl * HH) + 4 * (t0 + tt) + lg) * NS + li] = hn;
                        }
                    }
                }
                // fc_time's fragments of this wave's output tile (fetched per group: next to the gate fragments they would not fit)
                SB64_RENEW();
                f32x4 Wf1[KSH / 4];
#pragma unroll
                for (int q = 0; q < KSH / 4; ++q) Wf1[q] = frag4(o.tfc_w4[l], to, KSH / 4, q);
                const f32x4 Wf1b = bias4(o.tfc_b[l], to);
                __syncthreads();
                // fc_time + residual: (band, output tile) jobs; the tile-0 wave of a band writes the band's new h to the state tensor (only now: every
                // wave has read h_{t-1} of this group before the barrier)
#pragma unroll 1
                for (int jl = jw; jl < nb; jl += 2) {
                    const int j = g0 + jl;
                    float hn[KSH];
#pragma unroll
                    for (int ks = 0; ks < KSH; ++ks) hn[ks] = hnl[((jl * HH) + 4 * ks + lg) * NS + li];
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xg + j * C + 4 * to);
                    if (sok && to == 0) {
#pragma unroll
                        for (int q = 0; q < KSH / 4; ++q)
                            *reinterpret_cast<f32x4*>(hg + soff(j) + 4 * q) = f32x4{hn[4 * q], hn[4 * q + 1], hn[4 * q + 2], hn[4 * q + 3]};
                    }
                    f32x4 a2 = Wf1b;
#pragma unroll
                    for (int ks = 0; ks < KSH; ++ks) a2 = FE_MFMA(Wf1[ks / 4][ks % 4], hn[ks], a2);
                    if (sok) *reinterpret_cast<f32x4*>(xg + j * C + 4 * to) = f32x4{xv[0] + a2[0], xv[1] + a2[1], xv[2] + a2[2], xv[3] + a2[3]};
                }
                __syncthreads();
            }
        }
        // ======================================= band LSTM: one direction at a time, all eight waves, 31 steps each =======================================
#pragma unroll 1
        for (int d = 0; d < 2; ++d) {
            float* hb = smem + L::HB;
            float* bb = smem + L::BB;
            SB64_RENEW();
            f32x4 Wh[4][KSH / 4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int q = 0; q < KSH / 4; ++q) Wh[tt][q] = frag4(o.f_wh4[l][d], wave * 4 + tt, KSH / 4, q);
            // (start values [tile][lg][r] -> LDS: they would not fit next to the fragments)
            for (int i = tid; i < KSH * 16; i += kSbThreads) bb[i] = a.wp[o.f_b[l][d] + i];
            for (int i = tid; i < HH * NS; i += kSbThreads) hb[i] = 0.0f;          // h = 0 (buffer 0)
            float cst[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            f32x4 accx[4];
            const int wx = o.f_wx4[l][d] + (wave * 4) * (KSC / 4) * 256;
            __syncthreads();
            auto xpart = [&](int j) {                         // the x half of a step's gates: the fragments streamed from L2 (k4 order)
                SB64_RENEW();
                f32x4 xq[KSC / 4];
#pragma unroll
                for (int q = 0; q < KSC / 4; ++q) xq[q] = *reinterpret_cast<const f32x4*>(xg + j * C + 4 * q);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) accx[tt] = *reinterpret_cast<const f32x4*>(bb + (wave * 4 + tt) * 16 + 4 * lg);
#pragma unroll
                for (int q = 0; q < KSC / 4; ++q) {
                    f32x4 w4[4];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) w4[tt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (lane + lzv) * 16 + (tt * (KSC / 4) + q) * 1024, wx * 4, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) accx[tt] = FE_MFMA(w4[tt][e], xq[q][e], accx[tt]);
                    __builtin_amdgcn_sched_barrier(0);          // (four fragment quadruples in flight at a time: all sixteen hoisted were 64 registers)
                }
            };
            xpart(d ? kBands - 1 : 0);
            int cur = 0;
#pragma unroll 1
            for (int s = 0; s < kBands; ++s) {
                const int j = d ? kBands - 1 - s : s;
                const float* hc = hb + cur * (HH * NS);
                f32x4 acc[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[tt] = accx[tt];
#pragma unroll
                for (int hf2 = 0; hf2 < 2; ++hf2) {
                    float hf[KSH / 2];
#pragma unroll
                    for (int ks = 0; ks < KSH / 2; ++ks) hf[ks] = hc[(4 * (16 * hf2 + ks) + lg) * NS + li];
#pragma unroll
                    for (int ks = 0; ks < KSH / 2; ++ks)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) acc[tt] = FE_MFMA(Wh[tt][(16 * hf2 + ks) / 4][ks % 4], hf[ks], acc[tt]);
                }
                float* hnx = hb + (cur ^ 1) * (HH * NS);
                f32x4 yv;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const float ig = sb_sig(acc[tt][0]), fg = sb_sig(acc[tt][1]), gg = sb_tanh_pre(acc[tt][2]), og = sb_sig(acc[tt][3]);
                    cst[tt] = fg * cst[tt] + ig * gg;
                    const float hv = og * tanh_f(cst[tt]);
                    hnx[(4 * (wave * 4 + tt) + lg) * NS + li] = hv;         // unit KSH lg + t <-> position 4 t + lg
                    yv[tt] = hv;
                }
                *reinterpret_cast<f32x4*>(a.y + yoff(d, j) + 4 * wave) = yv;      // (the last tile's idle columns shadow the last stream: same value)
                if (s + 1 < kBands) xpart(d ? kBands - 2 - s : s + 1);      // the x half of the next step: before the barrier
                __syncthreads();
                cur ^= 1;
            }
        }
        // ======================================= fc_freq + residual: x += b + W [y_fwd | y_bwd], (band, output tile) jobs =======================================
        {
            SB64_RENEW();
            f32x4 Wf2[2][KSH / 4];
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                for (int q = 0; q < KSH / 4; ++q) Wf2[dd][q] = frag4(o.ffc_w4[l][dd], to, KSH / 4, q);
            const f32x4 Wf2b = bias4(o.ffc_b[l], to);
#pragma unroll 1
            for (int j = jw; j < kBands; j += 2) {
                f32x4 yq[2][KSH / 4];
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int q = 0; q < KSH / 4; ++q) yq[dd][q] = *reinterpret_cast<const f32x4*>(a.y + yoff(dd, j) + 4 * q);
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xg + j * C + 4 * to);
                f32x4 a2 = Wf2b;
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int ks = 0; ks < KSH; ++ks) a2 = FE_MFMA(Wf2[dd][ks / 4][ks % 4], yq[dd][ks / 4][ks % 4], a2);
                if (sok) *reinterpret_cast<f32x4*>(xg + j * C + 4 * to) = f32x4{xv[0] + a2[0], xv[1] + a2[1], xv[2] + a2[2], xv[3] + a2[3]};
            }
        }
        __syncthreads();
    }
#undef SB64_RENEW
}

template <class S>
void sb_launch_layers(const SbArgs& a, hipStream_t st, hipError_t* err) {
    static std::atomic<bool> attr_set[64];      // (more than 64 KB of dynamic LDS; per device: a process may drive several)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if constexpr (SbLds<S>::FITS) {
        auto* fn = &bsrnn_sb_layers_kernel<S>;
        if (!attr_set[dev].load(std::memory_order_relaxed)) {
            *err = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SbLds<S>::BYTES);
            if (*err != hipSuccess) return;
            attr_set[dev].store(true, std::memory_order_relaxed);
        }
        note_kernel("bsrnn_sb_layers_kernel");
        hipLaunchKernelGGL(fn, dim3((a.B + kSbStreams - 1) / kSbStreams), dim3(kSbThreads), SbLds<S>::BYTES, st, a);
        *err = hipGetLastError();
    } else if constexpr (Sb64Lds<S>::FITS) {
        auto* fn = &bsrnn_sb64_layers_kernel<S>;
        if (!attr_set[dev].load(std::memory_order_relaxed)) {
            *err = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Sb64Lds<S>::BYTES);
            if (*err != hipSuccess) return;
            attr_set[dev].store(true, std::memory_order_relaxed);
        }
        note_kernel("bsrnn_sb64_layers_kernel");
        hipLaunchKernelGGL(fn, dim3((a.B + kSbStreams - 1) / kSbStreams), dim3(kSbThreads), Sb64Lds<S>::BYTES, st, a);
        *err = hipGetLastError();
    } else {
        *err = hipErrorNotSupported;
    }
}

}  // namespace fe
