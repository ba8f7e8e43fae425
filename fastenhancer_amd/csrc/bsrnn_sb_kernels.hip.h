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
    static constexpr bool FITS = (S::C == 16 || S::C == 32) && BYTES <= 160 * 1024;
};

// offsets (floats) of the stream-batched layer weights inside the packed buffer (host: fe_api.hip::pack_weights_bsrnn)
struct SbOffsets {
    int t_w[8], t_b[8];           // time LSTM: A fragments [tile t < KSH][k-step < KSC + KSH][64]; start values [t][lg][r] (b_ih + b_hh, pre-scaled)
    int tfc_w[8], tfc_b[8];       // fc_time: A fragments [tile < C / 16][k-step < KSH][64]; bias [tile][lg][r]
    int f_w[8][2], f_b[8][2];     // band LSTM per direction: as the time LSTM's
    int ffc_w[8][2], ffc_b[8];    // fc_freq per direction half: A fragments [tile][k-step < KSH][64]; bias
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

template <class S>
void sb_launch_layers(const SbArgs& a, hipStream_t st, hipError_t* err) {
    if constexpr (SbLds<S>::FITS) {
        auto* fn = &bsrnn_sb_layers_kernel<S>;
        static std::atomic<bool> attr_set[64];      // (more than 64 KB of dynamic LDS; per device: a process may drive several)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!attr_set[dev].load(std::memory_order_relaxed)) {
            *err = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SbLds<S>::BYTES);
            if (*err != hipSuccess) return;
            attr_set[dev].store(true, std::memory_order_relaxed);
        }
        note_kernel("bsrnn_sb_layers_kernel");
        hipLaunchKernelGGL(fn, dim3((a.B + kSbStreams - 1) / kSbStreams), dim3(kSbThreads), SbLds<S>::BYTES, st, a);
        *err = hipGetLastError();
    } else {
        *err = hipErrorNotSupported;
    }
}

}  // namespace fe
