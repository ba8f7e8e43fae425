/*
 * fe_oracle.c — plain-C restatement of the FastEnhancer wav->wav streaming step (TEST INFRASTRUCTURE ONLY).
 *
 * Second, independent CPU oracle next to oracle/fe_oracle.py: used by tests/ as a cross-check and by the
 * `cpu_baseline` leg of bench.py (OpenMP over the independent streams).  The product path never links it.
 * It follows the reference file:line given at each function (paths relative to the reference checkout) and
 * is pinned on the golden vectors produced by the imported reference (tests/test_oracle_golden.py).
 *
 * Layout: activations are [position][channel] (channel innermost, so the inner loops vectorise);
 * weights are passed in the FUSED reference layouts and re-arranged once by feo_create().
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int C1, NL, C2, F2, KB, NH, N, H;   /* shape */
    float compression;
} feo_shape;

typedef struct {
    feo_shape s;
    int F0, F1, HD, OVL;
    float *window, *window_istft, *tw_re, *tw_im;
    float *enc_pre_w, *enc_pre_b;             /* [16][C1]  (k = t*8 + s*2 + c) */
    float **enc_w, **enc_b;                   /* [3][C1][C1] (tap, ci, co) */
    float *rfpre_lin, *rfpre_w, *rfpre_b;     /* [F2][F1], [C1][C2], [C2] */
    float **wih, **whh, **bih, **bhh;         /* [C2][3C2] (ci, gate*C2+co), [3C2] */
    float **fc1_w, **fc1_b, **qkv, **fc2_w, **fc2_b, *pe;   /* [C2][C2], [C2], [C2][3C2] */
    float *rfpost_lin, *rfpost_w, *rfpost_b;  /* [F1][F2], [C2][C1], [C1] */
    float **dec1_w, **dec1_b, **dec3_w, **dec3_b;   /* [2C1][C1], [3][C1][C1] */
    float *post1_w, *post1_b, *post_t_w, *post_t_b; /* [2C1][C1], [C1][16], [2] */
} feo_model;

static float *dupf(const float *src, size_t n) {
    float *p = (float *)malloc(n * sizeof(float));
    memcpy(p, src, n * sizeof(float));
    return p;
}
/* (Co, Ci, k) -> [k][Ci][Co] */
static float *conv_w(const float *w, int Co, int Ci, int k) {
    float *p = (float *)malloc((size_t)Co * Ci * k * sizeof(float));
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int t = 0; t < k; ++t) p[((size_t)t * Ci + ci) * Co + co] = w[((size_t)co * Ci + ci) * k + t];
    return p;
}

static inline float silu(float x) { return x / (1.0f + expf(-x)); }
static inline float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

/* iterative radix-2 complex FFT, N power of two; inverse=1 -> unscaled inverse */
static void fft(float *re, float *im, int N, const float *tw_re, const float *tw_im, int inverse) {
    for (int i = 1, j = 0; i < N; ++i) {
        int bit = N >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { float t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= N; len <<= 1) {
        int step = N / len;
        for (int i = 0; i < N; i += len)
            for (int k = 0; k < len / 2; ++k) {
                float wr = tw_re[k * step], wi = inverse ? -tw_im[k * step] : tw_im[k * step];
                float ur = re[i + k], ui = im[i + k];
                float vr = re[i + k + len / 2] * wr - im[i + k + len / 2] * wi;
                float vi = re[i + k + len / 2] * wi + im[i + k + len / 2] * wr;
                re[i + k] = ur + vr; im[i + k] = ui + vi;
                re[i + k + len / 2] = ur - vr; im[i + k + len / 2] = ui - vi;
            }
    }
}

/* y[F][Co] = act(b + sum_t sum_ci x[f+t-pad][ci] w[t][ci][co]) ; x rows outside [0,F) are zero */
static void conv_fc(const float *x, int F, int Ci, const float *w, const float *b, int k, int Co, float *y, int act) {
    const int pad = (k - 1) / 2;
    for (int f = 0; f < F; ++f) {
        float *yo = y + (size_t)f * Co;
        for (int co = 0; co < Co; ++co) yo[co] = b ? b[co] : 0.0f;
        for (int t = 0; t < k; ++t) {
            int fi = f + t - pad;
            if (fi < 0 || fi >= F) continue;
            const float *xi = x + (size_t)fi * Ci;
            const float *wt = w + (size_t)t * Ci * Co;
            for (int ci = 0; ci < Ci; ++ci) {
                const float xv = xi[ci];
                const float *wr = wt + (size_t)ci * Co;
                for (int co = 0; co < Co; ++co) yo[co] += xv * wr[co];
            }
        }
        if (act) for (int co = 0; co < Co; ++co) yo[co] = silu(yo[co]);
    }
}

feo_model *feo_create(const feo_shape *s, const float *const *t /* fused tensors, order below */) {
    feo_model *m = (feo_model *)calloc(1, sizeof(feo_model));
    m->s = *s;
    const int C1 = s->C1, C2 = s->C2, N = s->N, H = s->H, NL = s->NL, KB = s->KB, F2 = s->F2;
    const int F0 = N / 2, F1 = F0 / 4;
    m->F0 = F0; m->F1 = F1; m->HD = C2 / s->NH; m->OVL = N - H;
    /* windows: ONNXSTFT.__init__, functional/audio_modules.py:207-235 */
    m->window = (float *)malloc(N * sizeof(float));
    m->window_istft = (float *)malloc(N * sizeof(float));
    for (int i = 0; i < N; ++i) m->window[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / N));
    {
        int K = (N + H - 1) / H, L = H * (2 * K - 1) + (N - H);
        float *acc = (float *)calloc(L, sizeof(float));
        for (int j = 0; j < 2 * K - 1; ++j)
            for (int n = 0; n < N; ++n) acc[j * H + n] += m->window[n] * m->window[n];
        for (int n = 0; n < N; ++n) m->window_istft[n] = m->window[n] / acc[(K - 1) * H + n];
        free(acc);
    }
    m->tw_re = (float *)malloc(N / 2 * sizeof(float));
    m->tw_im = (float *)malloc(N / 2 * sizeof(float));
    for (int k = 0; k < N / 2; ++k) { m->tw_re[k] = (float)cos(-2.0 * M_PI * k / N); m->tw_im[k] = (float)sin(-2.0 * M_PI * k / N); }
    int q = 0;
    /* tensor order = section order of fe_weight_section (fused state_dict) */
    {   /* enc_pre.0.weight (C1, 8, 2) -> [k = t*8 + ch][co] */
        const float *w = t[q++];
        m->enc_pre_w = (float *)malloc(16 * C1 * sizeof(float));
        for (int co = 0; co < C1; ++co)
            for (int ch = 0; ch < 8; ++ch)
                for (int tp = 0; tp < 2; ++tp) m->enc_pre_w[(tp * 8 + ch) * C1 + co] = w[(co * 8 + ch) * 2 + tp];
        m->enc_pre_b = dupf(t[q++], C1);
    }
    m->enc_w = (float **)malloc(NL * sizeof(float *)); m->enc_b = (float **)malloc(NL * sizeof(float *));
    for (int l = 0; l < NL; ++l) { m->enc_w[l] = conv_w(t[q++], C1, C1, 3); m->enc_b[l] = dupf(t[q++], C1); }
    m->rfpre_lin = dupf(t[q++], (size_t)F2 * F1);
    m->rfpre_w = conv_w(t[q++], C2, C1, 1); m->rfpre_b = dupf(t[q++], C2);
#define ARR(name) m->name = (float **)malloc(KB * sizeof(float *))
    ARR(wih); ARR(whh); ARR(bih); ARR(bhh); ARR(fc1_w); ARR(fc1_b); ARR(qkv); ARR(fc2_w); ARR(fc2_b);
    for (int k = 0; k < KB; ++k) {
        if (k == 0) m->pe = dupf(t[q++], (size_t)F2 * C2);
        m->wih[k] = conv_w(t[q++], 3 * C2, C2, 1); m->whh[k] = conv_w(t[q++], 3 * C2, C2, 1);
        m->bih[k] = dupf(t[q++], 3 * C2); m->bhh[k] = dupf(t[q++], 3 * C2);
        m->fc1_w[k] = conv_w(t[q++], C2, C2, 1); m->fc1_b[k] = dupf(t[q++], C2);
        m->qkv[k] = conv_w(t[q++], 3 * C2, C2, 1);
        m->fc2_w[k] = conv_w(t[q++], C2, C2, 1); m->fc2_b[k] = dupf(t[q++], C2);
    }
    m->rfpost_lin = dupf(t[q++], (size_t)F1 * F2);
    m->rfpost_w = conv_w(t[q++], C1, C2, 1); m->rfpost_b = dupf(t[q++], C1);
    m->dec1_w = (float **)malloc(NL * sizeof(float *)); m->dec1_b = (float **)malloc(NL * sizeof(float *));
    m->dec3_w = (float **)malloc(NL * sizeof(float *)); m->dec3_b = (float **)malloc(NL * sizeof(float *));
    for (int l = 0; l < NL; ++l) {
        m->dec1_w[l] = conv_w(t[q++], C1, 2 * C1, 1); m->dec1_b[l] = dupf(t[q++], C1);
        m->dec3_w[l] = conv_w(t[q++], C1, C1, 3); m->dec3_b[l] = dupf(t[q++], C1);
    }
    m->post1_w = conv_w(t[q++], C1, 2 * C1, 1); m->post1_b = dupf(t[q++], C1);
    m->post_t_w = dupf(t[q++], (size_t)C1 * 16);   /* (C1, 2, 8) = [ci][co*8+j] already */
    m->post_t_b = dupf(t[q++], 2);
    return m;
}

void feo_destroy(feo_model *m) { free(m); /* test infrastructure: leaks the tables on purpose (process-lifetime) */ }

size_t feo_scratch_floats(const feo_model *m) {
    const feo_shape *s = &m->s;
    size_t act = (size_t)m->F1 * s->C1, tok = (size_t)s->F2 * s->C2;
    return 4 * (size_t)s->N + 2 * (size_t)m->F0 + (s->NL + 3) * act + 2 * act + 8 * tok + (size_t)s->F2 * 3 * s->C2 * 2 +
           (size_t)m->F1 * s->C2 + (size_t)m->F1 * 16 + (size_t)s->F2 * s->F2 + 1024;
}

/* One hop of ONE stream: scripts/export_onnx.py:48-58.
 * wav_in[H], cache_stft[N-H], cache_istft[N-H], h[KB][F2*C2] (this stream's rows), wav_out[H]; ws = scratch */
void feo_step_stream(const feo_model *m, const float *wav_in, float *cache_stft, float *cache_istft, float **h, float *wav_out,
                     float *ws) {
    const feo_shape *s = &m->s;
    const int C1 = s->C1, C2 = s->C2, N = s->N, H = s->H, NL = s->NL, KB = s->KB, F2 = s->F2, NH = s->NH;
    const int F0 = m->F0, F1 = m->F1, HD = m->HD, OVL = m->OVL;
    float *re = ws, *im = re + N, *xr = im + N, *xi = xr + F0;       /* spectrum, compressed spectrum */
    float *frame = xi + F0;                                            /* [N] */
    float *skips = frame + N;                                          /* (NL+1) x [F1][C1] */
    float *w0 = skips + (size_t)(NL + 1) * F1 * C1, *w1 = w0 + (size_t)F1 * 2 * C1;   /* w0 holds the cat [F1][2C1] */
    float *x = w1 + (size_t)F1 * C1, *y = x + (size_t)F2 * C2, *o = y + (size_t)F2 * C2;
    float *gi = o + (size_t)F2 * C2, *gh = gi + (size_t)F2 * 3 * C2, *y1 = gh + (size_t)F2 * 3 * C2;
    float *pt = y1 + (size_t)F1 * C1 + (size_t)F1 * C2, *sc = pt + (size_t)F1 * 16;

    /* ONNXSTFT.forward, functional/audio_modules.py:243-257 */
    memcpy(frame, cache_stft, OVL * sizeof(float));
    memcpy(frame + OVL, wav_in, H * sizeof(float));
    memcpy(cache_stft, frame + H, OVL * sizeof(float));
    for (int n = 0; n < N; ++n) { re[n] = frame[n] * m->window[n]; im[n] = 0.0f; }
    fft(re, im, N, m->tw_re, m->tw_im, 0);
    /* compress, model.py:684-690 */
    for (int f = 0; f < F0; ++f) {
        float mag = fmaxf(sqrtf(re[f] * re[f] + im[f] * im[f]), 1.0e-5f);
        float g = powf(mag, s->compression - 1.0f);
        xr[f] = re[f] * g; xi[f] = im[f] * g;
    }
    /* enc_pre: StridedConv1d (model.py:51-59) as an 8-tap stride-4 conv over (re, im), + SiLU */
    float *e0 = skips;
    for (int i = 0; i < F1; ++i) {
        float *yo = e0 + (size_t)i * C1;
        for (int co = 0; co < C1; ++co) yo[co] = m->enc_pre_b[co];
        for (int kk = 0; kk < 16; ++kk) {
            int c = kk & 1, sft = (kk >> 1) & 3, tp = kk >> 3;
            int f = 4 * (i + tp) + sft - 2;
            if (f < 0 || f >= F0) continue;
            float xv = c ? xi[f] : xr[f];
            const float *wr = m->enc_pre_w + (size_t)kk * C1;
            for (int co = 0; co < C1; ++co) yo[co] += xv * wr[co];
        }
        for (int co = 0; co < C1; ++co) yo[co] = silu(yo[co]);
    }
    /* encoder, model.py:637-642 */
    for (int l = 0; l < NL; ++l)
        conv_fc(skips + (size_t)l * F1 * C1, F1, C1, m->enc_w[l], m->enc_b[l], 3, C1, skips + (size_t)(l + 1) * F1 * C1, 1);
    /* rf_pre, model.py:646: Linear over freq then 1x1 */
    {
        const float *e = skips + (size_t)NL * F1 * C1;
        for (int f2 = 0; f2 < F2; ++f2) {
            float *yo = y1 + (size_t)f2 * C1;
            for (int c = 0; c < C1; ++c) yo[c] = 0.0f;
            for (int f1 = 0; f1 < F1; ++f1) {
                float wv = m->rfpre_lin[(size_t)f2 * F1 + f1];
                if (wv == 0.0f) continue;
                const float *ei = e + (size_t)f1 * C1;
                for (int c = 0; c < C1; ++c) yo[c] += wv * ei[c];
            }
        }
        conv_fc(y1, F2, C1, m->rfpre_w, m->rfpre_b, 1, C2, x, 0);
    }
    /* RNNFormer blocks, model.py:266-291 */
    for (int k = 0; k < KB; ++k) {
        float *hk = h[k];
        conv_fc(x, F2, C2, m->wih[k], m->bih[k], 1, 3 * C2, gi, 0);
        conv_fc(hk, F2, C2, m->whh[k], m->bhh[k], 1, 3 * C2, gh, 0);
        for (int f = 0; f < F2; ++f)
            for (int c = 0; c < C2; ++c) {
                const float *a = gi + (size_t)f * 3 * C2, *b = gh + (size_t)f * 3 * C2;
                float r = sigm(a[c] + b[c]), z = sigm(a[C2 + c] + b[C2 + c]);
                float n = tanhf(a[2 * C2 + c] + r * b[2 * C2 + c]);
                hk[(size_t)f * C2 + c] = (1.0f - z) * n + z * hk[(size_t)f * C2 + c];
            }
        conv_fc(hk, F2, C2, m->fc1_w[k], m->fc1_b[k], 1, C2, y, 0);
        for (int i = 0; i < F2 * C2; ++i) x[i] += y[i] + (k == 0 ? m->pe[i] : 0.0f);
        conv_fc(x, F2, C2, m->qkv[k], NULL, 1, 3 * C2, gi, 0);      /* rows [head][q|k|v][hd] */
        const float scale = 1.0f / sqrtf((float)HD);
        for (int hh = 0; hh < NH; ++hh)
            for (int qf = 0; qf < F2; ++qf) {
                const float *qv = gi + (size_t)qf * 3 * C2 + hh * 3 * HD;
                float mx = -INFINITY;
                for (int kf = 0; kf < F2; ++kf) {
                    const float *kv = gi + (size_t)kf * 3 * C2 + hh * 3 * HD + HD;
                    float d = 0.0f;
                    for (int e = 0; e < HD; ++e) d += qv[e] * kv[e];
                    sc[kf] = d * scale;
                    mx = fmaxf(mx, sc[kf]);
                }
                float sum = 0.0f;
                for (int kf = 0; kf < F2; ++kf) { sc[kf] = expf(sc[kf] - mx); sum += sc[kf]; }
                float *oo = o + (size_t)qf * C2 + hh * HD;
                for (int e = 0; e < HD; ++e) oo[e] = 0.0f;
                for (int kf = 0; kf < F2; ++kf) {
                    const float p = sc[kf] / sum;
                    const float *vv = gi + (size_t)kf * 3 * C2 + hh * 3 * HD + 2 * HD;
                    for (int e = 0; e < HD; ++e) oo[e] += p * vv[e];
                }
            }
        conv_fc(o, F2, C2, m->fc2_w[k], m->fc2_b[k], 1, C2, y, 0);
        for (int i = 0; i < F2 * C2; ++i) x[i] += y[i];
    }
    /* rf_post, model.py:654-656 */
    {
        float *y2 = y1 + (size_t)F1 * C1;
        for (int f1 = 0; f1 < F1; ++f1) {
            float *yo = y2 + (size_t)f1 * C2;
            for (int c = 0; c < C2; ++c) yo[c] = 0.0f;
            for (int f2 = 0; f2 < F2; ++f2) {
                float wv = m->rfpost_lin[(size_t)f1 * F2 + f2];
                if (wv == 0.0f) continue;
                const float *xi2 = x + (size_t)f2 * C2;
                for (int c = 0; c < C2; ++c) yo[c] += wv * xi2[c];
            }
        }
        conv_fc(y2, F1, C2, m->rfpost_w, m->rfpost_b, 1, C1, w1, 0);
    }
    /* decoder + dec_post 1x1, model.py:661-671: cat([x, skip]) -> 1x1 -> k3 */
    for (int l = 0; l <= NL; ++l) {
        const float *skip = skips + (size_t)(NL - l) * F1 * C1;
        for (int f = 0; f < F1; ++f) {
            memcpy(w0 + (size_t)f * 2 * C1, w1 + (size_t)f * C1, C1 * sizeof(float));
            memcpy(w0 + (size_t)f * 2 * C1 + C1, skip + (size_t)f * C1, C1 * sizeof(float));
        }
        if (l < NL) {
            conv_fc(w0, F1, 2 * C1, m->dec1_w[l], m->dec1_b[l], 1, C1, y1, 1);
            conv_fc(y1, F1, C1, m->dec3_w[l], m->dec3_b[l], 3, C1, w1, 1);
        } else {
            conv_fc(w0, F1, 2 * C1, m->post1_w, m->post1_b, 1, C1, y1, 1);
        }
    }
    /* ConvTranspose1d(C1 -> 2, k=8, s=4, p=2), model.py:91-95 */
    for (int i = 0; i < F1; ++i) {
        float *po = pt + (size_t)i * 16;
        for (int n = 0; n < 16; ++n) po[n] = 0.0f;
        for (int ci = 0; ci < C1; ++ci) {
            float xv = y1[(size_t)i * C1 + ci];
            const float *wr = m->post_t_w + (size_t)ci * 16;
            for (int n = 0; n < 16; ++n) po[n] += xv * wr[n];
        }
    }
    /* mask, un-compress (model.py:694-709), irfft + synthesis window + overlap-add (audio_modules.py:259-303) */
    for (int n = 0; n < N; ++n) { re[n] = 0.0f; im[n] = 0.0f; }
    for (int f = 0; f < F0; ++f) {
        int q2 = f + 2, j1 = q2 & 3, i1 = q2 >> 2;
        float m0 = m->post_t_b[0], m1 = m->post_t_b[1];
        if (i1 < F1) { m0 += pt[i1 * 16 + j1]; m1 += pt[i1 * 16 + 8 + j1]; }
        if (i1 >= 1) { m0 += pt[(i1 - 1) * 16 + j1 + 4]; m1 += pt[(i1 - 1) * 16 + 8 + j1 + 4]; }
        float yr = xr[f] * m0 - xi[f] * m1, yi = xr[f] * m1 + xi[f] * m0;
        float g = powf(sqrtf(yr * yr + yi * yi), 1.0f / s->compression - 1.0f);
        yr *= g; yi *= g;
        if (f == 0) { re[0] = yr; im[0] = 0.0f; }
        else { re[f] = yr; im[f] = yi; re[N - f] = yr; im[N - f] = -yi; }
    }
    fft(re, im, N, m->tw_re, m->tw_im, 1);
    for (int n = 0; n < N; ++n) {
        float v = re[n] / (float)N * m->window_istft[n];
        if (n < OVL) v += cache_istft[n];
        frame[n] = v;
    }
    memcpy(wav_out, frame, H * sizeof(float));
    memcpy(cache_istft, frame + H, OVL * sizeof(float));
}

/* B streams, one hop each, OpenMP over streams.  Layouts as the reference's tensors:
 * wav_in/out [B][H], cache_* [B][N-H], h [KB][B*F2][C2]. */
void feo_step(const feo_model *m, const float *wav_in, float *cache_stft, float *cache_istft, float *h, float *wav_out, int B,
              float *scratch /* B_threads * feo_scratch_floats */, int n_threads) {
    const feo_shape *s = &m->s;
    const size_t per = feo_scratch_floats(m);
    const size_t hs = (size_t)s->F2 * s->C2;
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int b = 0; b < B; ++b) {
        int tid = 0;
#ifdef _OPENMP
        extern int omp_get_thread_num(void);
        tid = omp_get_thread_num();
#endif
        float *hp[16];
        for (int k = 0; k < s->KB; ++k) hp[k] = h + ((size_t)k * B + b) * hs;
        feo_step_stream(m, wav_in + (size_t)b * s->H, cache_stft + (size_t)b * m->OVL, cache_istft + (size_t)b * m->OVL, hp,
                        wav_out + (size_t)b * s->H, scratch + (size_t)tid * per);
    }
}
