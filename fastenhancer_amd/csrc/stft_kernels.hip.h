// stft_kernels.hip.h — the STFT front / back ends of the path as stand-alone launches (gfx950).
//
// Inside fe_step / fe_offline the transforms are fused into the frame kernel (fe_kernels.hip.h).  The reference also
// exposes them as modules of their own, which scripts/export_onnx.py:55-57 composes line by line:
//   ONNXSTFT.forward(x, cache) / .inverse(spec, cache)        functional/audio_modules.py:243-303   (streaming, one hop)
//   CompressedSTFT.forward(x) / .inverse(spec)                functional/audio_modules.py:124-164   (offline, centered)
// These kernels back the mirrors of those methods (fe_stft_step, fe_istft_step, fe_stft_offline, fe_istft_offline) and
// the time-parallel overlap-add of fe_offline.  One workgroup = one frame of one stream; radix-2 Stockham FFT in LDS
// (fe::fft_lds).  They are HBM / latency-bound element-wise work around a 512- or 1024-point FFT, not GEMMs.
#pragma once
#include "fe_kernels.hip.h"

namespace fe {

template <int N_>
struct FftShape {
    static constexpr int NFFT = N_;
    static constexpr int LOG2N = (N_ == 512) ? 9 : (N_ == 1024 ? 10 : -1);
    static_assert(LOG2N > 0, "n_fft must be 512 or 1024");
};

// tables: [window N | window_istft N | twiddle N (N/2 float2)]
struct StftArgs {
    const float* tables;
    const float* wav_in;      // streaming: [b*in_stride + n], n < H;  offline: [b*in_stride + n], n < Tw
    size_t in_stride;
    const float* cache_in;    // [B][N-H]
    float* cache_out;         // [B][N-H] (may alias cache_in)
    float* spec;              // streaming: [B][N/2+1][1][2];  offline: [B][F][T][2]
    float* wav_out;           // streaming: [b*out_stride + n], n < H
    size_t out_stride;
    float* frames;            // offline inverse: [B][T][N] windowed frames before the overlap-add
    int B, T, H, Tw;
    int F;                    // offline: number of bins kept (N/2 with discard_last_freq_bin, else N/2+1)
    float compression;        // offline: |X|^(c-1) (forward) / |X|^(1/c-1) (inverse); 1 = none
    float eps;
};

// ONNXSTFT.forward (functional/audio_modules.py:243-257): x = cat(cache, new); cache' = x[-(N-H):]; rfft(x * window)
template <int N>
__global__ void __launch_bounds__(kThreads) stft_step_kernel(StftArgs a) {
    using S = FftShape<N>;
    __shared__ float2 fa[N], fb[N], tw[N / 2];
    const int b = blockIdx.x, tid = threadIdx.x, H = a.H, OVL = N - H;
    const float* win = a.tables;
    const float2* twg = reinterpret_cast<const float2*>(a.tables + 2 * N);
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = twg[i];
    const float* cin = a.cache_in + (size_t)b * OVL;
    const float* xin = a.wav_in + (size_t)b * a.in_stride;
    for (int n = tid; n < N; n += kThreads) {
        const float v = n < OVL ? cin[n] : xin[n - OVL];
        fb[n] = make_float2(v, 0.0f);
        fa[n] = make_float2(v * win[n], 0.0f);
    }
    __syncthreads();
    float* cout = a.cache_out + (size_t)b * OVL;
    for (int m = tid; m < OVL; m += kThreads) cout[m] = fb[m + H].x;
    __syncthreads();
    const float2* X = fft_lds<S, false>(fa, fb, tw);
    float2* out = reinterpret_cast<float2*>(a.spec) + (size_t)b * (N / 2 + 1);
    for (int f = tid; f <= N / 2; f += kThreads) out[f] = X[f];
}

// ONNXSTFT.inverse (functional/audio_modules.py:259-303): irfft (Im X[0], Im X[N/2] ignored), * window_istft,
// x[:N-H] += cache, out = x[:H], cache' = x[H:]
template <int N>
__global__ void __launch_bounds__(kThreads) istft_step_kernel(StftArgs a) {
    using S = FftShape<N>;
    __shared__ float2 fa[N], fb[N], tw[N / 2];
    const int b = blockIdx.x, tid = threadIdx.x, H = a.H, OVL = N - H;
    const float* wi = a.tables + N;
    const float2* twg = reinterpret_cast<const float2*>(a.tables + 2 * N);
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = twg[i];
    const float2* X = reinterpret_cast<const float2*>(a.spec) + (size_t)b * (N / 2 + 1);
    for (int f = tid; f <= N / 2; f += kThreads) {
        const float2 v = X[f];
        if (f == 0 || f == N / 2) fa[f] = make_float2(v.x, 0.0f);
        else { fa[f] = v; fa[N - f] = make_float2(v.x, -v.y); }
    }
    __syncthreads();
    const float2* y = fft_lds<S, true>(fa, fb, tw);
    float* xo = reinterpret_cast<float*>((y == fa) ? fb : fa);
    const float* cin = a.cache_in + (size_t)b * OVL;
    const float invN = 1.0f / (float)N;
    for (int n = tid; n < N; n += kThreads) {
        float v = y[n].x * invN * wi[n];
        if (n < OVL) v += cin[n];
        xo[n] = v;
    }
    __syncthreads();
    float* out = a.wav_out + (size_t)b * a.out_stride;
    for (int n = tid; n < H; n += kThreads) out[n] = xo[n];
    float* cout = a.cache_out + (size_t)b * OVL;
    for (int m = tid; m < OVL; m += kThreads) cout[m] = xo[m + H];
}

// CompressedSTFT.forward (functional/audio_modules.py:146-155 over STFT.forward :70-95): torch.stft(center=True,
// pad_mode="reflect"), keep F bins, X *= max(|X|, eps)^(c-1).  grid (T, B).
template <int N>
__global__ void __launch_bounds__(kThreads) stft_frames_kernel(StftArgs a) {
    using S = FftShape<N>;
    __shared__ float2 fa[N], fb[N], tw[N / 2];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* win = a.tables;
    const float2* twg = reinterpret_cast<const float2*>(a.tables + 2 * N);
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = twg[i];
    const float* xin = a.wav_in + (size_t)b * a.in_stride;
    for (int n = tid; n < N; n += kThreads) {
        int idx = t * a.H + n - N / 2;
        idx = idx < 0 ? -idx : idx;
        idx = idx >= a.Tw ? 2 * (a.Tw - 1) - idx : idx;
        fa[n] = make_float2(xin[idx] * win[n], 0.0f);
    }
    __syncthreads();
    const float2* X = fft_lds<S, false>(fa, fb, tw);
    float2* out = reinterpret_cast<float2*>(a.spec) + (size_t)b * a.F * a.T;
    for (int f = tid; f < a.F; f += kThreads) {
        float2 v = X[f];
        if (a.compression != 1.0f) {
            const float g = pow_f(fmaxf(sqrtf(v.x * v.x + v.y * v.y), a.eps), a.compression - 1.0f);
            v.x *= g; v.y *= g;
        }
        out[(size_t)f * a.T + t] = v;
    }
}

// CompressedSTFT.inverse (functional/audio_modules.py:157-164), first half: un-compress, zero-pad the dropped bin,
// irfft of frame t, * window -> frames[b][t][N].  grid (T, B).
template <int N>
__global__ void __launch_bounds__(kThreads) istft_frames_kernel(StftArgs a) {
    using S = FftShape<N>;
    __shared__ float2 fa[N], fb[N], tw[N / 2];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* win = a.tables;
    const float2* twg = reinterpret_cast<const float2*>(a.tables + 2 * N);
    for (int i = tid; i < N / 2; i += kThreads) tw[i] = twg[i];
    const float2* X = reinterpret_cast<const float2*>(a.spec) + (size_t)b * a.F * a.T;
    for (int f = tid; f <= N / 2; f += kThreads) {
        float2 v = f < a.F ? X[(size_t)f * a.T + t] : make_float2(0.0f, 0.0f);
        if (a.compression != 1.0f) {
            const float g = pow_f(sqrtf(v.x * v.x + v.y * v.y), 1.0f / a.compression - 1.0f);
            v.x *= g; v.y *= g;
        }
        if (f == 0 || f == N / 2) fa[f] = make_float2(v.x, 0.0f);
        else { fa[f] = v; fa[N - f] = make_float2(v.x, -v.y); }
    }
    __syncthreads();
    const float2* y = fft_lds<S, true>(fa, fb, tw);
    float* fr = a.frames + ((size_t)b * a.T + t) * N;
    const float invN = 1.0f / (float)N;
    for (int n = tid; n < N; n += kThreads) fr[n] = y[n].x * invN * win[n];
}

// torch.istft(center=True) tail (functional/audio_modules.py:117-119): y[pos] = sum_t frames[t][n - tH] / sum_t w^2[n - tH],
// n = pos + N/2, for pos < H (T-1).  One thread per output sample; frames stay L2-resident between the two launches.
// Tw_b != nullptr (ragged batch, fe_offline_ragged): utterance b has Tw_b[b] samples = Tb = 1 + Tw_b[b] / H frames of the T the batch is
// laid out for; its output is H (Tb - 1) samples from its own frames only (the rest of its row is left untouched).
__global__ void __launch_bounds__(kThreads) istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ win,
                                                            float* __restrict__ wav_out, size_t out_stride, int N, int H, int T,
                                                            const int* __restrict__ Tw_b = nullptr) {
    const int b = blockIdx.y;
    const int pos = blockIdx.x * kThreads + threadIdx.x;
    const int Tb = Tw_b != nullptr ? 1 + Tw_b[b] / H : T;
    const int n_out = H * (Tb - 1);
    if (pos >= n_out) return;
    const int n = pos + N / 2;
    int t_lo = (n - N + H) / H;            // ceil((n - N + 1) / H)
    t_lo = t_lo < 0 ? 0 : t_lo;
    int t_hi = n / H;
    t_hi = t_hi > Tb - 1 ? Tb - 1 : t_hi;
    const float* fr = frames + (size_t)b * T * N;
    float acc = 0.0f, env = 0.0f;
    for (int tt = t_lo; tt <= t_hi; ++tt) {
        const float wv = win[n - tt * H];
        acc += fr[(size_t)tt * N + (n - tt * H)];
        env += wv * wv;
    }
    wav_out[(size_t)b * out_stride + pos] = acc / env;
}

}  // namespace fe
