// fe_api.hip — C ABI (include/fastenhancer_hip.h), handle management, weight packing and
// kernel dispatch for the FastEnhancer forward path on gfx950.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/fastenhancer_hip.h"
#include "fe_impl.h"
#include "bsrnn_kernels.hip.h"
#include "fspen_kernels.hip.h"
#include "lisennet_kernels.hip.h"
#include "stft_kernels.hip.h"

// fe_last_step_kernel: the launchers (per-shape translation units included) name what they enqueue; a compute entry point
// clears the log when it starts and leaves it in its handle when it returns (KernelLogScope)
namespace fe {
namespace {
struct KernelLog { const char* name[16]; int n; };
thread_local KernelLog g_klog{};
}
void note_kernel(const char* name) {
    if (g_klog.n < 16) g_klog.name[g_klog.n] = name;
    if (g_klog.n < 17) ++g_klog.n;           // (17: more than the log holds)
}
}  // namespace fe

namespace {

thread_local std::string g_err;

// fe_set_option / fe_get_option.  `env`: the environment variable that sets the value NEW handles start with (A/B scripts under tools/);
// out-of-range or non-numeric values are ignored.
enum { OPT_BSRNN_ROLE_SPLIT = 0, OPT_BSRNN_SB_MIN, OPT_BSRNN_THREE_LAUNCH, OPT_BSRNN_OV_PROFILE, OPT_FSPEN_SB_MIN, OPT_LOW_LDS_COMPANION, OPT_BSRNN_FUSED, OPT_LISENNET_SB_MIN, OPT_COUNT };
struct OptionDef { const char* name; const char* env; int dflt, lo, hi; };
constexpr OptionDef kOptions[OPT_COUNT] = {
    {"bsrnn_role_split", "FE_BSRNN_OV", 1, 0, 1},
    {"bsrnn_stream_batch_min", "FE_BSRNN_SB", 2048, 0, 1 << 24},
    {"bsrnn_three_launch_step", "FE_BSRNN_SPLIT", 1, 0, 1},
    {"bsrnn_ov_profile", "FE_BSRNN_OV_PROF", 0, 0, 1},
    {"fspen_stream_batch_min", "FE_FSPEN_SB", 1536, 0, 1 << 24},
    {"low_lds_companion", "FE_LOWLDS", 1, 0, 1},
    {"bsrnn_fused_step", "FE_BSRNN_FUSED", 0, 0, 1},      // (measured negative: profiles/r6_bsrnn_fused_step.txt)
    {"lisennet_stream_batch_min", "FE_LISENNET_SB", 513, 0, 1 << 24},      // (2 x 256 + 1: from where the per-stream kernel needs a second round of workgroups - 219 us against the tiles' 193)
};
int env_int(const char* name, int dflt, int lo, int hi) {
    const char* e = std::getenv(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const long v = std::strtol(e, &end, 10);
    return (end && *end == 0 && v >= lo && v <= hi) ? (int)v : dflt;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define FE_HIP_CHECK(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return fail(FE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

struct Section {
    std::string name;
    size_t offset, count;
    std::vector<int> shape;
};

struct Dims {
    int C1, NL, C2, F2, KB, NFFT, HOP, F0, F1, HD;
    int ks[FE_MAX_KERNELS];
    int KT = 1;                // kernel_size_time (time_kernel variant)
    int FR = 0;                // 1: dprnn variant (sub-band GRU of C2 / 2 hidden units per direction instead of the attention)
    int LN = 0;                // 1: ln variant (GroupNorm after every conv, the reference's LayerNorm after the blocks' fc layers)
    int TA = 0;                // > 0: dptransformer variant (causal attention over the last TA frames instead of the time GRU)
    int BD = 0;                // 1: noncausal variant (bidirectional time GRU, rnn_fc over 2 C2; offline only, no caches)
    // model-state floats per stream: KB GRU states [F2][C2], or (dptransformer) 2 KB caches [F2][NH][TA][HD] + the ring head
    size_t hstate() const { return BD ? 0 : (size_t)KB * F2 * C2 * (TA ? 2 * TA : 1) + (TA ? 1 : 0); }
};

// ---------------------------------------------------------------------------- dispatch table
#define X(name, ...) extern "C" const fe::Impl* fe_impl_##name();
#ifdef FE_SHAPES_DEF         // a side build with a short shape list (build.py)
#include FE_SHAPES_DEF
#else
#include "fe_shapes.def"
#ifdef FE_LOCAL_DEF          // shapes added with `python -m fastenhancer_amd.build --add-shape ...`
#include FE_LOCAL_DEF
#endif
#endif
#undef X

#define XB(name, ...) extern "C" const fe::BImpl* fe_bimpl_##name();
#include "fe_bsrnn_shapes.def"
#undef XB
extern "C" const fe::FImpl* fe_fimpl_h256();
extern "C" const fe::LImpl* fe_limpl_h256();

const std::vector<const fe::BImpl*>& bimpls() {
    static const std::vector<const fe::BImpl*> v = {
#define XB(name, ...) fe_bimpl_##name(),
#include "fe_bsrnn_shapes.def"
#undef XB
    };
    return v;
}

const std::vector<const fe::Impl*>& impls() {
    static const std::vector<const fe::Impl*> v = {
#define X(name, ...) fe_impl_##name(),
#ifdef FE_SHAPES_DEF
#include FE_SHAPES_DEF
#else
#include "fe_shapes.def"
#ifdef FE_LOCAL_DEF
#include FE_LOCAL_DEF
#endif
#endif
#undef X
    };
    return v;
}

}  // namespace

struct fe_handle {
    fe_config cfg;
    Dims d;
    const fe::Impl* impl = nullptr;
    const fe::Impl* impl_many = nullptr;  // low-LDS companion (two workgroups per CU) for batches above #CUs streams, if compiled
    const fe::BImpl* bimpl = nullptr;     // arch == FE_ARCH_BSRNN
    const fe::FImpl* fimpl = nullptr;     // arch == FE_ARCH_FSPEN
    const fe::LImpl* limpl = nullptr;     // arch == FE_ARCH_LISENNET
    fe::BOffsets boff{};
    fe::SbOffsets sboff{};                    // stream-batched BSRNN layers (bsrnn_sb_kernels.hip.h), where built
    int packed_floats = 0;                    // floats of packed_dev (BSRNN)
    int device = 0;
    int max_wgs = 256;             // CUs of the device: one resident workgroup per CU (persistent grid above that)
    int pipe_frames = -1;          // fe_set_time_pipeline (-1: chosen from the model size)
    int offline_engine = FE_OFFLINE_AUTO;     // fe_set_offline_engine
    // fe_set_step_kernel / fe_set_option: which kernel a step dispatches.  The environment variables named in kOptions set the values NEW handles start
    // with (A/B runs of tools/ab_*.sh; a value outside the option's range, garbage included, is ignored) - nothing else in the library reads them.
    int step_kernel = FE_STEP_KERNEL_WG8;
    int opt[8] = {};                          // kOptions order
    const char* last_kernels[16] = {};        // fe_last_step_kernel: what the last compute call enqueued (string literals of the launchers)
    int n_last_kernels = 0;
    const char* last_shape = nullptr;         // ... and the compiled shape record that launched it
    mutable std::string last_text;
    fe_handle() {
        for (int i = 0; i < OPT_COUNT; ++i) opt[i] = env_int(kOptions[i].env, kOptions[i].dflt, kOptions[i].lo, kOptions[i].hi);
        if (std::getenv("FE_NO_LOWLDS")) opt[OPT_LOW_LDS_COMPANION] = 0;      // (the older spelling tools/ab_lowlds.sh uses)
        step_kernel = env_int("FE_WG8", FE_STEP_KERNEL_WG8, FE_STEP_KERNEL_WAVES4, FE_STEP_KERNEL_WG8_PERSIST);
    }
    unsigned int* pipe_flags_dev = nullptr;   // fe_spec_step's frame counters [max_wgs][KB] (fe_offline keeps its own in the work buffer)
    std::vector<hipStream_t> tb_streams;      // time-batched engine: the streams its nodes are spread over (lazy; tb_run)
    std::vector<hipEvent_t> tb_events;        // ... and its event pool
    unsigned long long* tb_probe_dev = nullptr;   // FE_TB_PROBE builds: phase clocks [4][kProbeSlots]
    hipStream_t host_streams[2] = {nullptr, nullptr};     // fe_step_host: copy-in / copy-out streams (lazy)
    hipEvent_t host_events[7] = {};                       // ... and its events
    float* tb_work_dev = nullptr;             // fe_spec_step on the time-batched engine: grow-only work buffer
    float* spec_ring_dev = nullptr;           // fe_spec_step, dptransformer, time-pipelined: the K / V rings of TA + P slots per pair (grow-only)
    size_t spec_ring_floats = 0;
    unsigned int* bsync_dev = nullptr;        // BSRNN fused per-hop step: barrier counters of the sixteen-stream tiles [kSyncTiles][2] (zeroed once; monotonic)
    float* bsplit_dev = nullptr;              // BSRNN per-hop step in three launches: band features | compressed spectrum | MLP pre-activations
    int bsplit_streams = 0;                   // (grow-only, sized by fe_state_init / the first step of a larger batch)
    size_t tb_work_floats = 0;
    unsigned int* tb_prog_dev = nullptr;      // fused stages: the scan workgroups' frame counters [KB][2 * max_wgs]
    std::vector<Section> sections;
    size_t blob_floats = 0;
    float* packed_dev = nullptr;
    float* tables_dev = nullptr;   // [window | window_istft | twiddle] for the stand-alone STFT launches (lazy)
    float* skip_dev = nullptr;     // scratch for shapes whose skips do not fit in LDS
    int skip_streams = 0;
    bool loaded = false;
    std::vector<float> window, window_istft, twiddle;
};

namespace {

// what a compute entry point enqueues ends up in its handle (fe_last_step_kernel)
struct KernelLogScope {
    fe_handle* h;
    explicit KernelLogScope(fe_handle* h_) : h(h_) {
        fe::g_klog.n = 0;
        h->last_shape = h->impl ? h->impl->name : h->bimpl ? h->bimpl->name : h->fimpl ? "fspen" : h->limpl ? "lisennet" : nullptr;
    }
    ~KernelLogScope() {
        if (!h) return;
        h->n_last_kernels = fe::g_klog.n;
        std::memcpy(h->last_kernels, fe::g_klog.name, sizeof(h->last_kernels));
    }
};

void add_section(fe_handle* h, const std::string& name, std::vector<int> shape) {
    size_t n = 1;
    for (int s : shape) n *= (size_t)s;
    size_t off = (h->blob_floats + 3) & ~(size_t)3;
    h->sections.push_back(Section{name, off, n, shape});
    h->blob_floats = off + n;
}

// the ln variant's fused state_dict (models/fastenhancer/ln/model.py:416-518 after remove_weight_reparameterizations; the final
// conv as dec_post.2 with its scale folded in, like the default model's): convs with their own biases, every norm layer kept
void build_sections_ln(fe_handle* h) {
    const Dims& d = h->d;
    char nm[128];
    auto wb = [&](const std::string& p, std::vector<int> wshape, bool bias = true) {
        add_section(h, p + ".weight", wshape);
        if (bias) add_section(h, p + ".bias", {wshape[0]});
    };
    wb("enc_pre.0", {d.C1, 8, 2}); wb("enc_pre.1", {d.C1});
    for (int i = 0; i < d.NL; ++i) {
        snprintf(nm, sizeof nm, "encoder.%d.0", i); wb(nm, {d.C1, d.C1, 3});
        snprintf(nm, sizeof nm, "encoder.%d.1", i); wb(nm, {d.C1});
    }
    add_section(h, "rf_pre.0.weight", {d.F2, d.F1});
    wb("rf_pre.1", {d.C2, d.C1, 1}); wb("rf_pre.2", {d.C2});
    for (int k = 0; k < d.KB; ++k) {
        auto key = [&](const char* s) { snprintf(nm, sizeof nm, "rf_block.%d.%s", k, s); return std::string(nm); };
        if (k == 0) add_section(h, key("pe"), {d.F2, d.C2});
        add_section(h, key("rnn.weight_ih_l0"), {3 * d.C2, d.C2});
        add_section(h, key("rnn.weight_hh_l0"), {3 * d.C2, d.C2});
        add_section(h, key("rnn.bias_ih_l0"), {3 * d.C2});
        add_section(h, key("rnn.bias_hh_l0"), {3 * d.C2});
        add_section(h, key("rnn_fc.weight"), {d.C2, d.C2});
        wb(key("rnn_post_norm"), {d.C2});
        add_section(h, key("attn.qkv.weight"), {3 * d.C2, d.C2});
        add_section(h, key("attn_fc.weight"), {d.C2, d.C2});
        wb(key("attn_post_norm"), {d.C2});
    }
    add_section(h, "rf_post.0.weight", {d.F1, d.F2});
    wb("rf_post.1", {d.C1, d.C2, 1}); wb("rf_post.2", {d.C1});
    for (int i = 0; i < d.NL; ++i) {
        snprintf(nm, sizeof nm, "decoder.%d.0", i); wb(nm, {d.C1, 2 * d.C1, 1});
        snprintf(nm, sizeof nm, "decoder.%d.1", i); wb(nm, {d.C1});
        snprintf(nm, sizeof nm, "decoder.%d.3", i); wb(nm, {d.C1, d.C1, 3}, false);
        snprintf(nm, sizeof nm, "decoder.%d.4", i); wb(nm, {d.C1});
    }
    wb("dec_post.0", {d.C1, 2 * d.C1, 1}, false); wb("dec_post.1", {d.C1});
    add_section(h, "dec_post.2.weight", {d.C1, 2, 8});
    add_section(h, "dec_post.2.bias", {2});
}

void build_sections(fe_handle* h) {
    const Dims& d = h->d;
    if (d.LN) { build_sections_ln(h); return; }
    char nm[128];
    add_section(h, "enc_pre.0.weight", {d.C1, 8, 2});
    add_section(h, "enc_pre.0.bias", {d.C1});
    for (int i = 0; i < d.NL; ++i) {
        snprintf(nm, sizeof nm, "encoder.%d.0.weight", i);
        if (d.KT > 1) add_section(h, nm, {d.C1, d.C1, d.KT, 3}); else add_section(h, nm, {d.C1, d.C1, 3});
        snprintf(nm, sizeof nm, "encoder.%d.0.bias", i); add_section(h, nm, {d.C1});
    }
    add_section(h, "rf_pre.0.weight", {d.F2, d.F1});
    add_section(h, "rf_pre.1.weight", {d.C2, d.C1, 1});
    add_section(h, "rf_pre.1.bias", {d.C2});
    if (d.TA) add_section(h, "time_pe", {4, d.TA + 1});      // the dptransformer model's positional bias (its `pe` parameter)
    for (int k = 0; k < d.KB; ++k) {
        auto key = [&](const char* s) { snprintf(nm, sizeof nm, "rf_block.%d.%s", k, s); return std::string(nm); };
        if (k == 0 && !d.FR) add_section(h, key("pe"), {d.F2, d.C2});
        if (d.TA) add_section(h, key("time_attn.qkv.weight"), {3 * d.C2, d.C2});
        else {
        for (const char* sfx : {"", "_reverse"}) {
            if (sfx[0] && !d.BD) break;
            add_section(h, key((std::string("rnn.weight_ih_l0") + sfx).c_str()), {3 * d.C2, d.C2});
            add_section(h, key((std::string("rnn.weight_hh_l0") + sfx).c_str()), {3 * d.C2, d.C2});
            add_section(h, key((std::string("rnn.bias_ih_l0") + sfx).c_str()), {3 * d.C2});
            add_section(h, key((std::string("rnn.bias_hh_l0") + sfx).c_str()), {3 * d.C2});
        }
        }
        add_section(h, key("rnn_fc.weight"), {d.C2, d.BD ? 2 * d.C2 : d.C2});
        add_section(h, key("rnn_fc.bias"), {d.C2});
        if (d.FR) {     // DPRNN's fused state_dict (models/fastenhancer/dprnn/model.py:159-161), module names as the default model's
            const int H = d.C2 / 2;
            for (const char* sfx : {"", "_reverse"}) {
                add_section(h, key((std::string("frnn.weight_ih_l0") + sfx).c_str()), {3 * H, d.C2});
                add_section(h, key((std::string("frnn.weight_hh_l0") + sfx).c_str()), {3 * H, H});
                add_section(h, key((std::string("frnn.bias_ih_l0") + sfx).c_str()), {3 * H});
                add_section(h, key((std::string("frnn.bias_hh_l0") + sfx).c_str()), {3 * H});
            }
            add_section(h, key("frnn_fc.weight"), {d.C2, d.C2});
            add_section(h, key("frnn_fc.bias"), {d.C2});
            continue;
        }
        add_section(h, key("attn.qkv.weight"), {3 * d.C2, d.C2});
        add_section(h, key("attn_fc.weight"), {d.C2, d.C2});
        add_section(h, key("attn_fc.bias"), {d.C2});
    }
    add_section(h, "rf_post.0.weight", {d.F1, d.F2});
    add_section(h, "rf_post.1.weight", {d.C1, d.C2, 1});
    add_section(h, "rf_post.1.bias", {d.C1});
    for (int i = 0; i < d.NL; ++i) {
        snprintf(nm, sizeof nm, "decoder.%d.0.weight", i); add_section(h, nm, {d.C1, 2 * d.C1, 1});
        snprintf(nm, sizeof nm, "decoder.%d.0.bias", i); add_section(h, nm, {d.C1});
        snprintf(nm, sizeof nm, "decoder.%d.2.weight", i);
        if (d.KT > 1) add_section(h, nm, {d.C1, d.C1, d.KT, 3}); else add_section(h, nm, {d.C1, d.C1, 3});
        snprintf(nm, sizeof nm, "decoder.%d.2.bias", i); add_section(h, nm, {d.C1});
    }
    add_section(h, "dec_post.0.weight", {d.C1, 2 * d.C1, 1});
    add_section(h, "dec_post.0.bias", {d.C1});
    add_section(h, "dec_post.2.weight", {d.C1, 2, 8});
    add_section(h, "dec_post.2.bias", {2});
}

// Window tables, ONNXSTFT.__init__ (functional/audio_modules.py:207-235).
void build_tables(fe_handle* h) {
    const int N = h->cfg.n_fft, H = h->cfg.hop_size, W = h->cfg.win_size;
    std::vector<float> win(N, 0.0f);
    const int pad = N - W;
    for (int i = 0; i < W; ++i) {   // torch.hann_window(W), periodic
        double v = 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)i / (double)W);
        win[pad / 2 + i] = (float)v;
    }
    const int K = (N + H - 1) / H;
    const int L = H * (2 * K - 1) + (N - H);
    std::vector<float> acc(L, 0.0f);
    for (int j = 0; j < 2 * K - 1; ++j)
        for (int n = 0; n < N; ++n) acc[j * H + n] += win[n] * win[n];
    std::vector<float> wi(N);
    for (int n = 0; n < N; ++n) wi[n] = win[n] / acc[(K - 1) * H + n];
    h->window = win;
    h->window_istft = wi;
    h->twiddle.resize(N);   // N/2 complex
    for (int k = 0; k < N / 2; ++k) {
        double ang = -2.0 * M_PI * (double)k / (double)N;
        h->twiddle[2 * k] = (float)std::cos(ang);
        h->twiddle[2 * k + 1] = (float)std::sin(ang);
    }
}

// Writes fragments at the compile-time offsets of fe::Pack<S>::v (the kernel uses the same table).
struct Packer {
    std::vector<float> buf;
    // B operand in fragment order: dst[(nt*KS + ks)*64 + lane] = B(k = 4ks + lane/16, n = 16nt + lane%16)
    void pack_b(int off, int K, int Ncols, const std::function<float(int, int)>& Bkn) {
        const int KS = K / 4, NT = (Ncols + 15) / 16;
        for (int nt = 0; nt < NT; ++nt)
            for (int ks = 0; ks < KS; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    int k = 4 * ks + lane / 16, n = 16 * nt + lane % 16;
                    buf[off + ((size_t)nt * KS + ks) * 64 + lane] = n < Ncols ? Bkn(k, n) : 0.0f;
                }
    }
    // A operand: dst[(mt*KS + ks)*64 + lane] = A(m = 16mt + lane%16, k = 4ks + lane/16)
    void pack_a(int off, int Mrows, int K, const std::function<float(int, int)>& Amk) {
        const int KS = K / 4, MT = (Mrows + 15) / 16;
        for (int mt = 0; mt < MT; ++mt)
            for (int ks = 0; ks < KS; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    int m = 16 * mt + lane % 16, k = 4 * ks + lane / 16;
                    buf[off + ((size_t)mt * KS + ks) * 64 + lane] = m < Mrows ? Amk(m, k) : 0.0f;
                }
    }
    void raw(int off, size_t n, const float* src) { memcpy(&buf[off], src, n * sizeof(float)); }
    void rep4(int off, size_t n, const float* src) {   // [n][4]: each value four times (16-byte accumulator initialisers)
        for (size_t i = 0; i < n; ++i)
            for (int r = 0; r < 4; ++r) buf[(size_t)off + 4 * i + r] = src[i];
    }
};

// constant operands of the matrix-core DFT (fe::Dft in fe_kernels.hip.h), N = N1 * 32, at the four given offsets of p.buf
void pack_dft_constants(Packer& p, int N, int dft1, int dft2, int dft3, int dft4) {
    {
        const int N1 = N / 32, KC = N1 / 2, MT = N1 / 16;
        auto c32 = [](int a, int b) { return std::cos(2.0 * M_PI * (double)((a * b) % 32) / 32.0); };
        auto s32 = [](int a, int b) { return std::sin(2.0 * M_PI * (double)((a * b) % 32) / 32.0); };
        auto c1 = [&](int a, int b) { return std::cos(2.0 * M_PI * (double)((a * b) % N1) / (double)N1); };
        auto s1 = [&](int a, int b) { return std::sin(2.0 * M_PI * (double)((a * b) % N1) / (double)N1); };
        for (int q = 0; q < 2; ++q) {
            // forward, 32-point stage: B[k = n2][n = k2] = Re (q = 0) / Im (q = 1) of W_32^(n2 k2)
            p.pack_b(dft1 + q * (2 * 8 * 64), 32, 32, [&](int k, int n) { return (float)(q == 0 ? c32(k, n) : -s32(k, n)); });
            // forward, N1-point stage, output half q: A[m = k1][k-step a*4MT + 4i + r, lane group lg]  <->  half a of
            // G', row n1 = 16 i + 4 lg + r:   Re X: [cos | sin],  Im X: [-sin | cos]
            p.pack_a(dft2 + q * (KC * 64), 16, 2 * N1, [&](int m, int k) {
                const int ks = k / 4, lg = k % 4, a = ks / (4 * MT), i = (ks % (4 * MT)) / 4, r = ks % 4;
                const int n1 = 16 * i + 4 * lg + r;
                if (q == 0) return (float)(a ? s1(m, n1) : c1(m, n1));
                return (float)(a ? c1(m, n1) : -s1(m, n1));
            });
            // inverse, N1-point stage (transposed), half q of H: B[k = (b, k1)][n = n1]:  Re H: [cos | -sin],  Im H: [sin | cos]
            p.pack_b(dft3 + q * (MT * KC * 64), 2 * N1, N1, [&](int k, int n) {
                const int b = k / N1, k1 = k % N1;
                if (q == 0) return (float)(b ? -s1(n, k1) : c1(n, k1));
                return (float)(b ? c1(n, k1) : s1(n, k1));
            });
            // inverse, 32-point stage (with the 1/N of irfft), per wave (p = q, jt): B[k-step a*4 + r, lane group lg][n = li]
            // <-> half a of H', k2 = 16 jt + 4 lg + r, output sample column n2 = 16 p + li:  cos / N (a = 0), -sin / N (a = 1)
            for (int jt = 0; jt < 2; ++jt)
                p.pack_b(dft4 + (q * 2 + jt) * (8 * 64), 32, 16, [&](int k, int n) {
                    const int ks = k / 4, lg = k % 4, a = ks / 4, r = ks % 4;
                    const int k2 = 16 * jt + 4 * lg + r, n2 = 16 * q + n;
                    return (float)((a ? -s32(k2, n2) : c32(k2, n2)) / (double)N);
                });
        }
    }
}

const float* sec(const fe_handle* h, const std::vector<float>& blob, const std::string& name) {
    for (const Section& s : h->sections)
        if (s.name == name) return blob.data() + s.offset;
    return nullptr;
}

int pack_weights(fe_handle* h, const std::vector<float>& blob, std::vector<float>* out) {
    const Dims& d = h->d;
    const int C1 = d.C1, C2 = d.C2, F1 = d.F1, F2 = d.F2;
    const fe::PackedOffsets& o = *h->impl->off;
    Packer p;
    p.buf.assign((size_t)o.total, 0.0f);
    char nm[128];
    auto S = [&](const std::string& n) { return sec(h, blob, n); };
    // (Co, Ci, 3), or (Co, Ci, KT, 3) for the time_kernel variant: one B operand per time tap, k = freq_tap*Ci + ci.
    // Units are in consumption order: unit 0 multiplies the CURRENT frame = time index KT-1 of the causal kernel, unit j the
    // frame j steps back = time index KT-1-j
    const int KT = d.KT;
    auto pack_k3 = [&](const int* offs, const float* w) {
        for (int j = 0; j < KT; ++j) {
            const int dt = KT - 1 - j;
            p.pack_b(offs[j], 3 * C1, C1, [&](int k, int n) { return w[((size_t)(n * C1 + (k % C1)) * KT + dt) * 3 + (k / C1)]; });
        }
    };
    auto pack_1x1 = [&](int off, const float* w, int Ci, int Co) {   // (Co, Ci[,1]): B[k=ci][n=co]
        p.pack_b(off, Ci, Co, [&](int k, int n) { return w[n * Ci + k]; });
    };
    {   // enc_pre: weight (C1, 8, 2): B[k = t*8 + ch][n = co] = W[co][ch][t]
        const float* w = S("enc_pre.0.weight");
        p.pack_b(o.enc_pre_w, 16, C1, [&](int k, int n) { return w[(n * 8 + (k & 7)) * 2 + (k >> 3)]; });
        p.rep4(o.enc_pre_b, C1, S("enc_pre.0.bias"));
    }
    for (int i = 0; i < d.NL; ++i) {
        snprintf(nm, sizeof nm, "encoder.%d.0.weight", i); pack_k3(&o.enc_w[i * KT], S(nm));
        snprintf(nm, sizeof nm, "encoder.%d.0.bias", i); p.rep4(o.enc_b[i], C1, S(nm));
    }
    {   // rf_pre: Linear (F2, F1) as A operand, then 1x1 conv (C2, C1)
        const float* w = S("rf_pre.0.weight");
        p.pack_a(o.rfpre_lin, F2, F1, [&](int m, int k) { return w[m * F1 + k]; });
        pack_1x1(o.rfpre_w, S("rf_pre.1.weight"), C1, C2);
        p.rep4(o.rfpre_b, C2, S("rf_pre.1.bias"));
    }
    if (d.TA) {     // [NH][L + 1] -> [NH][32]
        const float* tp = S("time_pe");
        for (int hh = 0; hh < 4; ++hh)
            for (int j = 0; j <= d.TA; ++j) p.buf[o.tpe + hh * 32 + j] = tp[hh * (d.TA + 1) + j];
    }
    for (int k = 0; k < d.KB; ++k) {
        auto key = [&](const char* s) { snprintf(nm, sizeof nm, "rf_block.%d.%s", k, s); return std::string(nm); };
        if (d.TA) pack_1x1(o.blk_tqkv[k], S(key("time_attn.qkv.weight")), C2, 3 * C2);
        else if (!d.BD)
        {   // GRU (3*C2, C2), gate order r,z,n: one padded column block per gate
            const int gsz = fe::ceil_div(C2, 16) * (C2 / 4) * 64, bsz = fe::round_up(C2, 16);
            const float* wih = S(key("rnn.weight_ih_l0"));
            const float* whh = S(key("rnn.weight_hh_l0"));
            const float* bih = S(key("rnn.bias_ih_l0"));
            const float* bhh = S(key("rnn.bias_hh_l0"));
            if (o.gru_flat) {   // one (3 C2)-column matrix, rows r | z | n as stored by nn.GRU
                pack_1x1(o.blk_wih[k], wih, C2, 3 * C2);
                pack_1x1(o.blk_whh[k], whh, C2, 3 * C2);
                p.raw(o.blk_bih[k], 3 * C2, bih);
                p.raw(o.blk_bhh[k], 3 * C2, bhh);
            } else
            for (int g = 0; g < 3; ++g) {
                pack_1x1(o.blk_wih[k] + g * gsz, wih + (size_t)g * C2 * C2, C2, C2);
                pack_1x1(o.blk_whh[k] + g * gsz, whh + (size_t)g * C2 * C2, C2, C2);
                p.raw(o.blk_bih[k] + g * bsz, C2, bih + g * C2);
                p.raw(o.blk_bhh[k] + g * bsz, C2, bhh + g * C2);
            }
        }
        if (o.u8_gx[k] != 0) {
            // 512-thread per-hop kernel (fe_frame8.hip.h, Shape::G8P): the block weights as LDS-staged units (PackedOffsets::u8_*):
            // B fragments [tile][k-step][64] followed by the tiles' start values [tile][16].
            const float* wih = S(key("rnn.weight_ih_l0"));
            const float* whh = S(key("rnn.weight_hh_l0"));
            const float* bih = S(key("rnn.bias_ih_l0"));
            const float* bhh = S(key("rnn.bias_hh_l0"));
            const int NG = C2 / 16, R = C2 % 16, NT = 3 * NG + 1, KS = C2 / 4;
            // tile t, column c -> row of the (rows, C2) weight matrix (-1: padding); bias(t, c)
            // (rscale(row): the gate rows carry -log2 e (r, z) / 2 log2 e (n) so that sigma / tanh are one exp2 + rcp of the accumulator)
            auto pack_u8 = [&](int off, int ntiles, const float* w, const std::function<int(int, int)>& wrow, const std::function<float(int, int)>& bias,
                               const std::function<float(int)>& rscale = [](int) { return 1.0f; }) {
                for (int t = 0; t < ntiles; ++t) {
                    for (int ks = 0; ks < KS; ++ks)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int row = wrow(t, lane % 16);
                            p.buf[(size_t)off + ((size_t)t * KS + ks) * 64 + lane] = row >= 0 ? w[(size_t)row * C2 + 4 * ks + lane / 16] * rscale(row) : 0.0f;
                        }
                    for (int c = 0; c < 16; ++c) p.buf[(size_t)off + (size_t)ntiles * KS * 64 + t * 16 + c] = wrow(t, c) >= 0 ? bias(t, c) * rscale(wrow(t, c)) : 0.0f;
                }
            };
            auto gscale = [&](int row) { return row < 2 * C2 ? fe::kGateRZ : fe::kGateN; };
            // channel-grouped gate tiles: tile 3 G + gate: column c <-> channel 16 G + c of that gate; the last tile: columns [0, R) r,
            // [R, 2 R) z, [2 R, 3 R) n of the R = C2 % 16 left-over channels
            auto grow = [&](int t, int c) -> int {
                if (t < 3 * NG) return (t % 3) * C2 + 16 * (t / 3) + c;
                return c < 3 * R ? (c / R) * C2 + 16 * NG + c % R : -1;
            };
            auto shared = [&](int t) { return t < 3 * NG && t % 3 < 2; };      // pure r / z tile: x and h halves in one accumulator
            pack_u8(o.u8_gx[k], NT, wih, grow, [&](int t, int c) { const int r = grow(t, c); return bih[r] + (shared(t) ? bhh[r] : 0.0f); }, gscale);
            pack_u8(o.u8_gh[k], NT, whh, grow, [&](int t, int c) { const int r = grow(t, c); return shared(t) ? 0.0f : bhh[r]; }, gscale);
#if FE_WG8_HPRE
            {   // ... and regrouped four k-steps per lane for the front-of-frame W_hh h products (PackedOffsets::u8_gh4)
                const int NQ = KS / 4, KR = KS % 4, TS = NQ * 256 + KR * 64;
                for (int t = 0; t < NT; ++t)
                    for (int lane = 0; lane < 64; ++lane) {
                        for (int q = 0; q < NQ; ++q)
                            for (int j = 0; j < 4; ++j)
                                p.buf[(size_t)o.u8_gh4[k] + (size_t)t * TS + q * 256 + lane * 4 + j] = p.buf[(size_t)o.u8_gh[k] + ((size_t)t * KS + 4 * q + j) * 64 + lane];
                        for (int r = 0; r < KR; ++r)
                            p.buf[(size_t)o.u8_gh4[k] + (size_t)t * TS + NQ * 256 + r * 64 + lane] = p.buf[(size_t)o.u8_gh[k] + ((size_t)t * KS + 4 * NQ + r) * 64 + lane];
                    }
            }
#endif
            auto plain = [&](int ncols) { return [ncols](int t, int c) { return 16 * t + c < ncols ? 16 * t + c : -1; }; };
            const float* f1b = S(key("rnn_fc.bias"));
            const float* f2b = S(key("attn_fc.bias"));
            pack_u8(o.u8_f1[k], fe::ceil_div(C2, 16), S(key("rnn_fc.weight")), plain(C2), [&](int t, int c) { return f1b[16 * t + c]; });
            {   // qkv: fragments only (the unit has no start values)
                const float* wq = S(key("attn.qkv.weight"));
                const int ntq = fe::ceil_div(3 * C2, 16);
                for (int t = 0; t < ntq; ++t)
                    for (int ks = 0; ks < KS; ++ks)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int row = 16 * t + lane % 16;
                            p.buf[(size_t)o.u8_q[k] + ((size_t)t * KS + ks) * 64 + lane] = row < 3 * C2 ? wq[(size_t)row * C2 + 4 * ks + lane / 16] : 0.0f;
                        }
            }
            pack_u8(o.u8_f2[k], fe::ceil_div(C2, 16), S(key("attn_fc.weight")), plain(C2), [&](int t, int c) { return f2b[16 * t + c]; });
        }
        if (h->impl->tb && !d.TA) {
            // time-batched engine (tb_kernels.hip.h): per direction the input weights as ONE flat (3 C2)-column matrix (rows r | z | n as
            // stored by nn.GRU) with the bias b_ih + (r, z) b_hh, the hidden weights per gate, b_hn
            const int gsz = fe::ceil_div(C2, 16) * (C2 / 4) * 64;
            for (int dir = 0; dir < (d.BD ? 2 : 1); ++dir) {
                const char* sfx = dir ? "_reverse" : "";
                const float* wih = S(key((std::string("rnn.weight_ih_l0") + sfx).c_str()));
                const float* whh = S(key((std::string("rnn.weight_hh_l0") + sfx).c_str()));
                const float* bih = S(key((std::string("rnn.bias_ih_l0") + sfx).c_str()));
                const float* bhh = S(key((std::string("rnn.bias_hh_l0") + sfx).c_str()));
                pack_1x1(o.tb_wih[k][dir], wih, C2, 3 * C2);
                for (int c = 0; c < 3 * C2; ++c) p.buf[(size_t)o.tb_bx[k][dir] + c] = bih[c] + (c < 2 * C2 ? bhh[c] : 0.0f);
                for (int g = 0; g < 3; ++g) pack_1x1(o.tb_whh[k][dir] + g * gsz, whh + (size_t)g * C2 * C2, C2, C2);
                p.raw(o.tb_bhn[k][dir], C2, bhh + 2 * C2);
            }
            if (d.BD) pack_1x1(o.tb_fc1_w[k], S(key("rnn_fc.weight")), 2 * C2, C2);
        }
        if (!d.BD) pack_1x1(o.blk_fc1_w[k], S(key("rnn_fc.weight")), C2, C2);
        if (!d.LN) p.raw(o.blk_fc1_b[k], C2, S(key("rnn_fc.bias")));      // (ln variant: the fc layers have no bias, their LayerNorm's is in the ln tables)
        if (d.FR) {
            // sub-band GRU: the input weights of both directions as one (3 C2)-column matrix [direction][r|z|n][unit] in the qkv
            // slot, its bias = b_ih + (r, z only) b_hh; hidden weights [direction][j][gate][unit]; b_hn.  (The block has no
            // positional embedding: blk_pe stays zero.)
            const int H = C2 / 2;
            std::vector<float> wih((size_t)3 * C2 * C2), bi((size_t)3 * C2);
            for (int dir = 0; dir < 2; ++dir) {
                const char* sfx = dir ? "_reverse" : "";
                const float* w = S(key((std::string("frnn.weight_ih_l0") + sfx).c_str()));
                const float* whh = S(key((std::string("frnn.weight_hh_l0") + sfx).c_str()));
                const float* bih = S(key((std::string("frnn.bias_ih_l0") + sfx).c_str()));
                const float* bhh = S(key((std::string("frnn.bias_hh_l0") + sfx).c_str()));
                std::copy(w, w + (size_t)3 * H * C2, wih.begin() + (size_t)dir * 3 * H * C2);
                for (int r = 0; r < 3 * H; ++r) bi[(size_t)dir * 3 * H + r] = bih[r] + (r < 2 * H ? bhh[r] : 0.0f);
                float* dst = p.buf.data() + o.blk_fhh[k] + (size_t)dir * H * 3 * H;
                for (int j = 0; j < H; ++j)
                    for (int g = 0; g < 3; ++g)
                        for (int u = 0; u < H; ++u) dst[((size_t)j * 3 + g) * H + u] = whh[((size_t)g * H + u) * H + j];
                for (int u = 0; u < H; ++u) p.buf[o.blk_fbhn[k] + (size_t)dir * H + u] = bhh[2 * H + u];
            }
            pack_1x1(o.blk_qkv[k], wih.data(), C2, 3 * C2);
            p.raw(o.blk_qkv_b[k], 3 * C2, bi.data());
            pack_1x1(o.blk_fc2_w[k], S(key("frnn_fc.weight")), C2, C2);
            p.raw(o.blk_fc2_b[k], C2, S(key("frnn_fc.bias")));
            continue;
        }
        if (k == 0) p.raw(o.blk_pe, (size_t)F2 * C2, S(key("pe")));
        pack_1x1(o.blk_qkv[k], S(key("attn.qkv.weight")), C2, 3 * C2);
        pack_1x1(o.blk_fc2_w[k], S(key("attn_fc.weight")), C2, C2);
        if (!d.LN) p.raw(o.blk_fc2_b[k], C2, S(key("attn_fc.bias")));
    }
    {
        const float* w = S("rf_post.0.weight");   // (F1, F2)
        p.pack_a(o.rfpost_lin, F1, F2, [&](int m, int k) { return w[m * F2 + k]; });
        p.raw(o.rfpost_w, (size_t)C1 * C2, S("rf_post.1.weight"));      // plain copies: the debug dump of the rf_post stage
        p.raw(o.rfpost_b, C1, S("rf_post.1.bias"));
        if (d.LN) {     // its own staged unit: a GroupNorm sits between it and decoder layer 0
            pack_1x1(o.rfpost1_w, S("rf_post.1.weight"), C2, C1);
            p.rep4(o.rfpost1_b, C1, S("rf_post.1.bias"));
        }
    }
    const std::vector<float> zeros_c1((size_t)C1, 0.0f);
    for (int i = 0; i < d.NL; ++i) {
        if (i == 0 && !d.LN) {
            // rf_post's 1x1 conv (C2 -> C1, affine, no activation) feeds only this layer's x half: folded in.
            //   v = Wd_x (Wrp y + brp) + Wd_s skip + bd  =  (Wd_x Wrp) y + Wd_s skip + (bd + Wd_x brp)
            // K = C2 (filterbank output, true scale: weights carry kSiluScale) + C1 (skip, already scaled)
            const float* wrp = S("rf_post.1.weight");     // (C1, C2)
            const float* brp = S("rf_post.1.bias");
            const float* wd = S("decoder.0.0.weight");    // (C1, 2 C1): [n][0 .. C1) x, [n][C1 .. 2 C1) skip
            const float* bd = S("decoder.0.0.bias");
            std::vector<float> wf((size_t)C1 * C2), bf(C1);
            for (int n = 0; n < C1; ++n) {
                for (int k = 0; k < C2; ++k) {
                    double acc = 0.0;
                    for (int m = 0; m < C1; ++m) acc += (double)wd[(size_t)n * 2 * C1 + m] * wrp[(size_t)m * C2 + k];
                    wf[(size_t)n * C2 + k] = (float)(acc * fe::kSiluScale);
                }
                double acc = bd[n];
                for (int m = 0; m < C1; ++m) acc += (double)wd[(size_t)n * 2 * C1 + m] * brp[m];
                bf[n] = (float)acc;
            }
            p.pack_b(o.dec1_w[0], C2 + C1, C1, [&](int k, int n) { return k < C2 ? wf[(size_t)n * C2 + k] : wd[(size_t)n * 2 * C1 + C1 + (k - C2)]; });
            p.rep4(o.dec1_b[0], C1, bf.data());
        } else {
        snprintf(nm, sizeof nm, "decoder.%d.0.weight", i); pack_1x1(o.dec1_w[i], S(nm), 2 * C1, C1);
        snprintf(nm, sizeof nm, "decoder.%d.0.bias", i); p.rep4(o.dec1_b[i], C1, S(nm));
        }
        if (d.LN) {     // (the k = 3 conv is module 3 there and has no bias)
            snprintf(nm, sizeof nm, "decoder.%d.3.weight", i); pack_k3(&o.dec3_w[i * KT], S(nm));
            p.rep4(o.dec3_b[i], C1, zeros_c1.data());
            continue;
        }
        snprintf(nm, sizeof nm, "decoder.%d.2.weight", i); pack_k3(&o.dec3_w[i * KT], S(nm));
        snprintf(nm, sizeof nm, "decoder.%d.2.bias", i); p.rep4(o.dec3_b[i], C1, S(nm));
    }
    pack_1x1(o.post1_w, S("dec_post.0.weight"), 2 * C1, C1);
    p.rep4(o.post1_b, C1, d.LN ? zeros_c1.data() : S("dec_post.0.bias"));
    if (d.LN) {     // gain / bias of the norm sites, Shape::LN_SITES order
        int q = 0;
        auto site = [&](const std::string& prefix, int C) {
            p.raw(o.ln_g[q], C, S(prefix + ".weight"));
            p.raw(o.ln_b[q], C, S(prefix + ".bias"));
            ++q;
        };
        site("enc_pre.1", C1);
        for (int i = 0; i < d.NL; ++i) { snprintf(nm, sizeof nm, "encoder.%d.1", i); site(nm, C1); }
        site("rf_pre.2", C2);
        for (int k = 0; k < d.KB; ++k) {
            snprintf(nm, sizeof nm, "rf_block.%d.rnn_post_norm", k); site(nm, C2);
            snprintf(nm, sizeof nm, "rf_block.%d.attn_post_norm", k); site(nm, C2);
        }
        site("rf_post.2", C1);
        for (int i = 0; i < d.NL; ++i) {
            snprintf(nm, sizeof nm, "decoder.%d.1", i); site(nm, C1);
            snprintf(nm, sizeof nm, "decoder.%d.4", i); site(nm, C1);
        }
        site("dec_post.1", C1);
    }
    {   // transposed conv weight (C1, 2, 8): B[k = ci][n = co*8 + j]
        const float* w = S("dec_post.2.weight");
        p.pack_b(o.post_t_w, C1, 16, [&](int k, int n) { return w[k * 16 + n]; });
        p.raw(o.post_t_b, 2, S("dec_post.2.bias"));
    }
    {   // scaled conv trunk (fe_kernels.hip.h, kSiluScale): biases of the SiLU layers, entry weights, exit weights
        const float c = d.LN ? 1.0f : fe::kSiluScale;      // (ln variant: no scaling - a norm layer follows every conv)
        auto szB = [](int K, int N) { return (size_t)fe::ceil_div(N, 16) * (K / 4) * 64; };
        auto scale = [&](int off, size_t n, float f) { for (size_t i = 0; i < n; ++i) p.buf[(size_t)off + i] *= f; };
        scale(o.enc_pre_w, szB(16, C1), c); scale(o.enc_pre_b, 4 * C1, c);       // (biases: 4x replicated tables)
        for (int i = 0; i < d.NL; ++i) { scale(o.enc_b[i], 4 * C1, c); scale(o.dec1_b[i], 4 * C1, c); scale(o.dec3_b[i], 4 * C1, c); }
        scale(o.rfpre_w, szB(C1, C2), 1.0f / c);                                  // encoder -> RNNFormer: back to true scale
        // (RNNFormer -> decoder: rf_post's 1x1 is folded into decoder.0.0 above, with the scale)
        scale(o.post1_b, 4 * C1, c);
        scale(o.post_t_w, szB(C1, 16), 1.0f / c);                                 // transposed conv: true-scale mask
    }
    p.raw(o.window, h->window.size(), h->window.data());
    p.raw(o.window_istft, h->window_istft.size(), h->window_istft.data());
    p.raw(o.twiddle, h->twiddle.size(), h->twiddle.data());
    pack_dft_constants(p, h->cfg.n_fft, o.dft1, o.dft2, o.dft3, o.dft4);
    if (o.k4_delta != 0) {
        // Register-resident block weights (fe::Shape::REGW): a second copy of the block-weight region whose B-operand tiles are
        // regrouped four k-steps per lane ([ks / 4][lane][4], the ks % 4 remainder plain) for 16-byte fetches - fe::TokW
        const int KS = C2 / 4, NF4 = KS / 4, tile_floats = KS * 64;
        std::copy(p.buf.begin() + o.blk_wih[0], p.buf.begin() + o.blk_end, p.buf.begin() + o.blk_wih[0] + o.k4_delta);
        std::vector<float> t((size_t)tile_floats);
        auto regroup = [&](int off, int floats) {
            for (int tl = 0; tl < floats / tile_floats; ++tl) {
                float* dst = &p.buf[(size_t)off + o.k4_delta + (size_t)tl * tile_floats];
                std::copy(dst, dst + tile_floats, t.begin());
                for (int ks = 0; ks < 4 * NF4; ++ks)
                    for (int ln = 0; ln < 64; ++ln) dst[(ks / 4) * 256 + ln * 4 + (ks % 4)] = t[(size_t)ks * 64 + ln];
            }
        };
        const int szCC = fe::ceil_div(C2, 16) * tile_floats, szC3 = fe::ceil_div(3 * C2, 16) * tile_floats;
        for (int k = 0; k < d.KB; ++k) {
            regroup(o.blk_wih[k], 3 * szCC); regroup(o.blk_whh[k], 3 * szCC);
            regroup(o.blk_fc1_w[k], szCC); regroup(o.blk_qkv[k], szC3); regroup(o.blk_fc2_w[k], szCC);
            if (d.TA) regroup(o.blk_tqkv[k], szC3);
        }
    }
    if (o.conv_k4_delta != 0) {
        // Time-batched engine (r4w): a copy of the conv units whose weight tiles are regrouped four k-steps per lane for 16-byte fetches
        // (tb_kernels.hip.h::conv_gemm; the biases and the filterbanks in the copy stay as they are).  Must run AFTER every conv weight is packed.
        const int ubeg = o.u_off[0], uend = o.u_off[o.n_units - 1] + o.u_size[o.n_units - 1];
        std::copy(p.buf.begin() + ubeg, p.buf.begin() + uend, p.buf.begin() + ubeg + o.conv_k4_delta);
        std::vector<float> t;
        auto regroup_c = [&](int off, int tiles, int KS) {
            const int tile_floats = KS * 64, NF4 = KS / 4;
            t.resize((size_t)tile_floats);
            for (int tl = 0; tl < tiles; ++tl) {
                float* dst = &p.buf[(size_t)off + o.conv_k4_delta + (size_t)tl * tile_floats];
                std::copy(dst, dst + tile_floats, t.begin());
                for (int ks = 0; ks < 4 * NF4; ++ks)
                    for (int ln = 0; ln < 64; ++ln) dst[(ks / 4) * 256 + ln * 4 + (ks % 4)] = t[(size_t)ks * 64 + ln];
            }
        };
        const int NTC = fe::ceil_div(C1, 16), KSC = C1 / 4, KS2 = C2 / 4;
        regroup_c(o.enc_pre_w, NTC, 4);
        for (int l = 0; l < d.NL; ++l) {
            regroup_c(o.enc_w[l], NTC, 3 * KSC);
            regroup_c(o.dec1_w[l], NTC, ((l == 0 && !d.LN) ? KS2 : KSC) + KSC);
            regroup_c(o.dec3_w[l], NTC, 3 * KSC);
        }
        regroup_c(o.post1_w, NTC, 2 * KSC);
        regroup_c(o.post_t_w, 1, KSC);
    }
    *out = std::move(p.buf);
    return FE_OK;
}


// the baseline families' host sides (weight sections, handle creation, packers): one file each
#include "fe_api_bsrnn.inc"
#include "fe_api_fspen.inc"
#include "fe_api_lisennet.inc"

size_t bsrnn_lstm_floats(const fe_handle* h, int B) { return (size_t)2 * h->cfg.rf_blocks * B * 31 * 2 * h->cfg.channels; }

fe::BArgs bsrnn_args(fe_handle* h, int B, int T) {
    fe::BArgs a{};
    a.xp_scratch = h->skip_dev;
    a.wp = h->packed_dev;
    a.off = h->boff;
    a.B = B;
    a.T = T;
    a.compression = h->cfg.input_compression;
    return a;
}

// The per-hop BSRNN step runs as three launches (bsrnn_kernels.hip.h, PART): per stream 31 C floats of band features, 514 of compressed
// spectrum and 2056 of MLP pre-activations pass through this scratch.  Grow-only; fe_state_init sizes it for its batch, so that a
// steady-state step allocates nothing (FE_BSRNN_SPLIT=0: the fused kernel, for A/B measurements).
constexpr int kSyncTiles = 64;       // sixteen-stream tiles of a fused BSRNN step (one workgroup per CU: 1024 CUs)
size_t bsplit_floats_per_stream(const fe_handle* h) {
    return (size_t)31 * h->cfg.channels + 2 * 257 + 2 * 1028 + (h->bimpl->launch_sb ? (size_t)2 * 31 * 2 * h->cfg.channels : 0);      // (+ the stream-batched layers' y scratch)
}
int ensure_bsplit(fe_handle* h, int B) {
    if (!h->bimpl || B <= h->bsplit_streams) return FE_OK;
    if (!h->opt[OPT_BSRNN_THREE_LAUNCH]) return FE_OK;
    if (h->bsplit_dev) { FE_HIP_CHECK(hipFree(h->bsplit_dev)); h->bsplit_dev = nullptr; h->bsplit_streams = 0; }
    FE_HIP_CHECK(hipMalloc(&h->bsplit_dev, (size_t)B * bsplit_floats_per_stream(h) * sizeof(float)));
    h->bsplit_streams = B;
    if (!h->bsync_dev) {
        FE_HIP_CHECK(hipMalloc(&h->bsync_dev, kSyncTiles * 2 * sizeof(unsigned int)));
        FE_HIP_CHECK(hipMemset(h->bsync_dev, 0, kSyncTiles * 2 * sizeof(unsigned int)));
    }
    return FE_OK;
}

int launch_bsrnn(fe_handle* h, const fe::BArgs& a_in, void* stream) {
    hipError_t e = hipSuccess;
    fe::BArgs a = a_in;
    a.ov_off = (h->step_kernel == FE_STEP_KERNEL_WAVES4 || !h->opt[OPT_BSRNN_ROLE_SPLIT]) ? 1 : 0;
    // (fe_set_option("bsrnn_ov_profile", 1): fe_profile_step probes the role-split PART 1 of the three-launch step instead of the fused kernel's phases)
    const bool ov_prof = h->opt[OPT_BSRNN_OV_PROFILE] != 0;
    h->last_shape = h->bimpl->name;
    if (a.mode == fe::FE_MODE_STREAM && a.T == 1 && a.dbg == nullptr && (a.clk == nullptr || (ov_prof && h->cfg.channels == 16 && a.B <= h->max_wgs))) {
        const int rc = ensure_bsplit(h, a.B);
        if (rc != FE_OK) return rc;
        if (h->bsplit_dev && a.B <= h->bsplit_streams) {
            a.mlp_x = h->bsplit_dev;
            a.mlp_sp = a.mlp_x + (size_t)a.B * 31 * h->cfg.channels;
            a.mlp_pre = a.mlp_sp + (size_t)a.B * 2 * 257;
            a.sb_y = a.mlp_pre + (size_t)a.B * 2 * 1028;
            a.gsync = (h->opt[OPT_BSRNN_FUSED] && a.clk == nullptr && (a.B + 15) / 16 <= kSyncTiles) ? h->bsync_dev : nullptr;
            // large batches: the LSTM layers batched over the streams on the matrix cores (sixteen streams per workgroup) - from the batch
            // size where sixteen-stream workgroups fill the chip better than one stream per workgroup (FE_BSRNN_SB: that threshold; 0 = never)
            // (default 2048; measured crossover on 256 CUs: ~1900 streams, profiles/r4c_bsrnn_stream_batched.txt.  num_channels = 64 (r6): a sixteen-stream tile takes 5.8 ms
            //  whatever the batch and the per-stream kernel 2.3 us per stream - crossover at ~2700 streams: the threshold counts 11 / 8 there)
            const int sb_opt = h->opt[OPT_BSRNN_SB_MIN];
            const int sb_min = h->cfg.channels == 64 ? (int)((long long)sb_opt * 11 / 8) : sb_opt;
            if (h->bimpl->launch_sb && sb_min > 0 && a.B >= sb_min) h->bimpl->launch_sb(a, h->sboff, h->packed_floats, h->max_wgs, (hipStream_t)stream, &e);
            else
            h->bimpl->launch_split(a, h->max_wgs, (hipStream_t)stream, &e);
            if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
            return FE_OK;
        }
    }
    h->bimpl->launch(a, h->max_wgs, (hipStream_t)stream, &e);
    if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return FE_OK;
}

int check_ready(const fe_handle* h) {
    if (!h) return fail(FE_ERR_INVALID_ARG, "null handle");
    if (!h->loaded) return fail(FE_ERR_NO_WEIGHTS, "fe_load_weights has not been called");
    return FE_OK;
}

// Scratch for the encoder outputs of shapes that keep them in global memory: one slot per resident WORKGROUP (the
// grid never exceeds max_wgs), allocated once by fe_load_weights - nothing is allocated, freed or synchronised inside
// the compute calls (they can be captured into HIP graphs).  The slots belong to the handle: launches of one handle
// must be stream-ordered (one stream, or event-ordered streams); concurrent launches need one handle each.
int ensure_scratch(fe_handle* h, int) {
    // (BSRNN: band-LSTM input projections of the C = 64 shape; FastEnhancer: the larger of the shape's own plan and its
    // low-LDS companion's, which runs two workgroups per CU)
    size_t floats = 0;
    if (h->fimpl || h->limpl) floats = 0;
    else if (h->bimpl) floats = (size_t)h->max_wgs * h->bimpl->xp_floats;
    else {
        floats = (size_t)h->max_wgs * h->impl->occ * h->impl->skip_floats;
        if (h->impl_many) floats = std::max(floats, (size_t)h->max_wgs * h->impl_many->occ * h->impl_many->skip_floats);
    }
    if (floats == 0 || h->skip_dev) return FE_OK;
    FE_HIP_CHECK(hipMalloc(&h->skip_dev, floats * sizeof(float)));
    FE_HIP_CHECK(hipMemset(h->skip_dev, 0, floats * sizeof(float)));
    h->skip_streams = h->max_wgs;
    return FE_OK;
}

fe::FrameArgs base_args(fe_handle* h, int B, int T) {
    fe::FrameArgs a{};
    a.wp = h->packed_dev;
    a.B = B;
    a.T = T;
    a.compression = h->cfg.input_compression;
    a.rf_eps = h->cfg.rf_eps > 0.0f ? h->cfg.rf_eps : 1.0e-5f;
    a.skip = h->skip_dev;
    a.step_kernel = h->step_kernel;
    return a;
}

}  // namespace

static int ensure_tables(fe_handle* h, hipStream_t st);

// fe_spec_step on the time pipeline for the variants whose frames exchange more than the GRU state (r4v).  dptransformer: the caller's K / V
// caches - per (cache, stream, pair) a ring of L slots whose oldest frame sits in slot `head` of the stream - are laid out as the pipeline's
// rings of RS = L + P slots (ring index j = window position j, oldest first; frame t of the chunk lives at index L + t), and after the
// launch the chunk's last L frames (ring indices T .. T + L - 1) go back to the caller's caches in the reference's order (head = 0): what
// ONNXModel.forward returns (dptransformer/model.py:231-232).  time_kernel: the same for the causal convs' input caches (L = KT - 1 slots of
// [F1][C1] per (conv, stream), no head; time_kernel/model.py:119-148).
// caches [ROWS][L][SZ], rings [ROWS][RS][SZ], ROWS = caches x streams x PAIRS; heads [B] (floats) or nullptr; one thread per float.
__global__ void ring_in_kernel(const float* __restrict__ cache, const float* __restrict__ heads, float* __restrict__ ring, size_t n, int B, int PAIRS,
                               int SZ, int L, int RS) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int dd = (int)(i % SZ);
    const int j = (int)((i / SZ) % L);                // window position, oldest first
    const size_t row = i / ((size_t)SZ * L);
    int slot = j;
    if (heads != nullptr) {
        slot += (int)heads[(row / PAIRS) % B];
        slot = slot >= L ? slot - L : slot;
    }
    ring[(row * RS + j) * SZ + dd] = cache[(row * L + slot) * SZ + dd];
}

__global__ void ring_out_kernel(const float* __restrict__ ring, float* __restrict__ cache, float* __restrict__ heads, size_t n, int B, int SZ, int L, int RS,
                                int T) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (heads != nullptr && i < (size_t)B) heads[i] = 0.0f;
    if (i >= n) return;
    const int dd = (int)(i % SZ);
    const int j = (int)((i / SZ) % L);
    const size_t row = i / ((size_t)SZ * L);
    cache[(row * L + j) * SZ + dd] = ring[(row * RS + (size_t)((T + j) % RS)) * SZ + dd];
}

// fe_debug_poison_lds: every workgroup fills 160 KiB of LDS (= one workgroup per CU at a time) with quiet-NaN patterns; 16 x #CUs workgroups so that every CU
// gets at least one whatever the dispatch order
__global__ void __launch_bounds__(256) poison_lds_kernel(unsigned int* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned int pl[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) pl[i] = 0x7fc00000u | (unsigned int)i;
    __syncthreads();
    if (pl[(threadIdx.x * 97 + blockIdx.x) % (160 * 1024 / 4)] == 0x12345u && sink) sink[0] = 1u;      // (keeps the stores alive)
}

extern "C" {

int fe_debug_poison_lds(void* stream) {
    int dev = 0, cus = 0;
    FE_HIP_CHECK(hipGetDevice(&dev));
    FE_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    FE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(poison_lds_kernel, dim3(16 * (cus > 0 ? cus : 256)), dim3(256), 160 * 1024, (hipStream_t)stream, (unsigned int*)nullptr);
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

const char* fe_last_error(void) { return g_err.c_str(); }
const char* fe_version(void) { return "fastenhancer_hip 0.1 (gfx950)"; }

int fe_create(const fe_config* cfg, fe_handle** out) {
    if (!cfg || !out) return fail(FE_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (cfg->arch == FE_ARCH_BSRNN) return create_bsrnn(cfg, out);
    if (cfg->arch == FE_ARCH_FSPEN) return create_fspen(cfg, out);
    if (cfg->arch == FE_ARCH_LISENNET) return create_lisennet(cfg, out);
    if (cfg->arch != FE_ARCH_FASTENHANCER)
        return fail(FE_ERR_UNSUPPORTED_CONFIG, "arch %d is not built into this library", cfg->arch);
    if (cfg->n_fft % 2 != 0) return fail(FE_ERR_INVALID_ARG, "`n_fft` must be an even number, but given %d.", cfg->n_fft);
    if (cfg->win_size > cfg->n_fft) return fail(FE_ERR_INVALID_ARG, "n_fft(%d) must be bigger than win_size(%d)", cfg->n_fft, cfg->win_size);
    if (cfg->hop_size <= 0 || cfg->hop_size > cfg->n_fft) return fail(FE_ERR_INVALID_ARG, "hop_size %d out of range", cfg->hop_size);
    if (cfg->stride != 4) return fail(FE_ERR_UNSUPPORTED_CONFIG, "stride %d (every shipped config uses 4)", cfg->stride);
    if (cfg->n_kernels < 2 || cfg->n_kernels > FE_MAX_KERNELS) return fail(FE_ERR_INVALID_ARG, "len(kernel_size)=%d", cfg->n_kernels);
    if (cfg->kernel_size[0] != 8) return fail(FE_ERR_UNSUPPORTED_CONFIG, "kernel_size[0]=%d (shipped: 8)", cfg->kernel_size[0]);
    for (int i = 1; i < cfg->n_kernels; ++i)
        if (cfg->kernel_size[i] != 3) return fail(FE_ERR_UNSUPPORTED_CONFIG, "kernel_size[%d]=%d (shipped: 3)", i, cfg->kernel_size[i]);
    if (cfg->rf_heads != 4) return fail(FE_ERR_UNSUPPORTED_CONFIG, "num_heads=%d (shipped: 4)", cfg->rf_heads);
    if (!(cfg->input_compression > 0.0f && cfg->input_compression <= 1.0f)) return fail(FE_ERR_INVALID_ARG, "input_compression");

    const fe::Impl* impl = nullptr;
    const int kt = cfg->kernel_size_time > 1 ? cfg->kernel_size_time : 1;
    const int fr = cfg->channels_frnn > 0 ? 1 : 0;
    if (fr && 2 * cfg->channels_frnn != cfg->rf_channels)
        return fail(FE_ERR_UNSUPPORTED_CONFIG, "channels_frnn=%d with channels=%d (the dprnn kernels are built for channels_frnn = channels / 2, every shipped yaml)",
                    cfg->channels_frnn, cfg->rf_channels);
    const int ln = cfg->ln ? 1 : 0;
    if (ln && (fr || cfg->lookbehind > 0 || kt > 1)) return fail(FE_ERR_INVALID_ARG, "ln excludes channels_frnn / lookbehind / kernel_size_time");
    const int bd = cfg->bidirectional ? 1 : 0;
    if (bd && (fr || cfg->lookbehind > 0 || kt > 1 || ln)) return fail(FE_ERR_INVALID_ARG, "bidirectional excludes channels_frnn / lookbehind / kernel_size_time / ln");
    const int ta = cfg->lookbehind > 0 ? cfg->lookbehind : 0;
    if (ta && ta != 31) return fail(FE_ERR_UNSUPPORTED_CONFIG, "lookbehind=%d (the dptransformer kernels are built for 31, every shipped yaml)", ta);
    if (ta && fr) return fail(FE_ERR_INVALID_ARG, "channels_frnn and lookbehind are exclusive");
    for (const fe::Impl* im : impls())
        if (im->C1 == cfg->channels && im->NL == cfg->n_kernels - 1 && im->C2 == cfg->rf_channels && im->F2 == cfg->rf_freq &&
            im->KB == cfg->rf_blocks && im->NFFT == cfg->n_fft && im->HOP == cfg->hop_size && im->KT == kt && im->LOW == 0 && im->FR == fr && im->TA == ta && im->LN == ln && im->BD == bd)
            impl = im;
    const fe::Impl* impl_many = nullptr;
    for (const fe::Impl* im : impls())
        if (impl && im->LOW >= 1 && im->occ >= 2 && im->C1 == impl->C1 && im->NL == impl->NL && im->C2 == impl->C2 && im->F2 == impl->F2 &&
            im->KB == impl->KB && im->NFFT == impl->NFFT && im->HOP == impl->HOP && im->KT == impl->KT && im->FR == impl->FR && im->TA == impl->TA && im->LN == impl->LN && !impl->BD)
            impl_many = im;
    // a companion reads its shape's packed buffer: every offset it uses must be the same function of the shape (fe::Pack does not depend on LOW; r4x: the
    // k4 copies sit behind the time-batched engine's and the 512-thread kernel's sections - checked here rather than trusted)
    if (impl && impl_many && (impl_many->off->k4_delta != impl->off->k4_delta || impl_many->off->conv_k4_delta != impl->off->conv_k4_delta ||
                              impl_many->off->window != impl->off->window || impl_many->off->blk_wih[0] != impl->off->blk_wih[0]))
        return fail(FE_ERR_UNSUPPORTED_CONFIG, "build error: the low-LDS companion of this shape was compiled with a different packed-weight layout (fe::Pack must not depend on LOW)");
    if (!impl)
        return fail(FE_ERR_UNSUPPORTED_CONFIG,
                    "no kernel compiled for channels=%d layers=%d rf_channels=%d rf_freq=%d rf_blocks=%d n_fft=%d hop=%d kernel_size_time=%d%s "
                    "(build it: python -m fastenhancer_amd.build --add-shape %d,%d,%d,%d,%d,%d,%d,%d%s)",
                    cfg->channels, cfg->n_kernels - 1, cfg->rf_channels, cfg->rf_freq, cfg->rf_blocks, cfg->n_fft, cfg->hop_size, kt, fr ? " dprnn" : (ta ? " dptransformer" : (ln ? " ln" : (bd ? " noncausal" : ""))),
                    cfg->channels, cfg->n_kernels - 1, cfg->rf_channels, cfg->rf_freq, cfg->rf_blocks, cfg->n_fft, cfg->hop_size, kt, fr ? ",0,1" : (ta ? ",0,0,31" : (ln ? ",0,0,0,1" : (bd ? ",0,0,0,0,1" : ""))));
    if (impl->lds_bytes > 160 * 1024)
        return fail(FE_ERR_UNSUPPORTED_CONFIG, "shape needs %zu bytes of LDS (> 160 KiB per CU)", impl->lds_bytes);
    fe_handle* h = new fe_handle();
    h->cfg = *cfg;
    h->impl = impl;
    h->impl_many = impl_many;              // (fe_set_option("low_lds_companion", 0) keeps run_step off it)
    h->d = Dims{impl->C1, impl->NL, impl->C2, impl->F2, impl->KB, impl->NFFT, impl->HOP, impl->NFFT / 2, impl->NFFT / 8, impl->C2 / 4, {0}};
    h->d.KT = impl->KT;
    h->d.FR = impl->FR;
    h->d.TA = impl->TA;
    h->d.LN = impl->LN;
    h->d.BD = impl->BD;
    for (int i = 0; i < cfg->n_kernels; ++i) h->d.ks[i] = cfg->kernel_size[i];
    if (hipGetDevice(&h->device) != hipSuccess) h->device = -1;   // no GPU: sections/tables still usable
    else {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) h->max_wgs = cus;
    }
    build_sections(h);
    build_tables(h);
    *out = h;
    return FE_OK;
}

void fe_destroy(fe_handle* h) {
    if (!h) return;
    if (h->packed_dev) (void)hipFree(h->packed_dev);
    if (h->skip_dev) (void)hipFree(h->skip_dev);
    if (h->tables_dev) (void)hipFree(h->tables_dev);
    if (h->pipe_flags_dev) (void)hipFree(h->pipe_flags_dev);
    for (hipStream_t s : h->tb_streams) (void)hipStreamDestroy(s);
    for (hipEvent_t e : h->tb_events) (void)hipEventDestroy(e);
    if (h->tb_probe_dev) (void)hipFree(h->tb_probe_dev);
    if (h->tb_prog_dev) (void)hipFree(h->tb_prog_dev);
    if (h->tb_work_dev) (void)hipFree(h->tb_work_dev);
    if (h->spec_ring_dev) (void)hipFree(h->spec_ring_dev);
    if (h->bsplit_dev) (void)hipFree(h->bsplit_dev);
    if (h->bsync_dev) (void)hipFree(h->bsync_dev);
    for (hipStream_t s : h->host_streams) if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : h->host_events) if (e) (void)hipEventDestroy(e);
    delete h;
}

#ifdef FE_TB_PROBE
// probe builds only (not part of the ABI): read and clear the time-batched engine's phase clocks, out[4 * kProbeSlots]
extern "C" int fe_tb_probe_read(fe_handle* h, unsigned long long* out) {
    if (!h || !h->tb_probe_dev) return FE_ERR_INVALID_ARG;
    FE_HIP_CHECK(hipDeviceSynchronize());
    FE_HIP_CHECK(hipMemcpy(out, h->tb_probe_dev, 4 * fe::tb::kProbeSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    FE_HIP_CHECK(hipMemset(h->tb_probe_dev, 0, 4 * fe::tb::kProbeSlots * sizeof(unsigned long long)));
    return FE_OK;
}
#endif

size_t fe_weight_floats(const fe_handle* h) { return h ? h->blob_floats : 0; }
int fe_weight_sections(const fe_handle* h) { return h ? (int)h->sections.size() : 0; }

int fe_weight_section(const fe_handle* h, int idx, const char** name, size_t* offset_floats, size_t* count_floats) {
    if (!h || idx < 0 || idx >= (int)h->sections.size()) return fail(FE_ERR_INVALID_ARG, "section index %d", idx);
    if (name) *name = h->sections[idx].name.c_str();
    if (offset_floats) *offset_floats = h->sections[idx].offset;
    if (count_floats) *count_floats = h->sections[idx].count;
    return FE_OK;
}

int fe_load_weights(fe_handle* h, const float* blob_dev, size_t nfloats, void* stream) {
    if (!h || !blob_dev) return fail(FE_ERR_INVALID_ARG, "null argument");
    if (nfloats != h->blob_floats) return fail(FE_ERR_INVALID_ARG, "blob has %zu floats, expected %zu", nfloats, h->blob_floats);
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> blob(nfloats);
    FE_HIP_CHECK(hipMemcpyAsync(blob.data(), blob_dev, nfloats * sizeof(float), hipMemcpyDeviceToHost, st));
    FE_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<float> packed;
    int rc = h->limpl ? pack_weights_lisennet(h, blob, &packed) : h->fimpl ? pack_weights_fspen(h, blob, &packed) : (h->bimpl ? pack_weights_bsrnn(h, blob, &packed) : pack_weights(h, blob, &packed));
    if (rc != FE_OK) return rc;
    if (h->packed_dev) { FE_HIP_CHECK(hipFree(h->packed_dev)); h->packed_dev = nullptr; }
    FE_HIP_CHECK(hipMalloc(&h->packed_dev, packed.size() * sizeof(float)));
    FE_HIP_CHECK(hipMemcpyAsync(h->packed_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, st));
    FE_HIP_CHECK(hipStreamSynchronize(st));
    rc = ensure_scratch(h, 0);
    if (rc != FE_OK) return rc;
    h->loaded = true;
    return FE_OK;
}

// time_kernel variant: floats of the causal convs' frame caches per stream (2 NL layers x [KT-1][F1][C1])
static size_t tk_floats(const fe_handle* h) {
    const Dims& d = h->d;
    return (h->bimpl || h->fimpl || h->limpl) ? 0 : (size_t)2 * d.NL * (d.KT - 1) * d.F1 * d.C1;
}

size_t fe_state_floats(const fe_handle* h, int B) {
    if (!h || B <= 0) return 0;
    const Dims& d = h->d;
    if (h->limpl) return (size_t)B * (2 * (size_t)(d.NFFT - d.HOP) + h->limpl->cache_floats);
    if (h->fimpl) return (size_t)B * 2 * (size_t)(d.NFFT - d.HOP) + fspen_gru_floats(B);
    if (h->bimpl) return (size_t)B * 2 * (size_t)(d.NFFT - d.HOP) + bsrnn_lstm_floats(h, B);
    return (size_t)B * (2 * (size_t)(d.NFFT - d.HOP) + d.hstate() + tk_floats(h));
}

int fe_state_init(fe_handle* h, float* state_dev, int B, void* stream) {
    if (!h || !state_dev || B <= 0) return fail(FE_ERR_INVALID_ARG, "bad argument");
    FE_HIP_CHECK(hipMemsetAsync(state_dev, 0, fe_state_floats(h, B) * sizeof(float), (hipStream_t)stream));
    if (h->bimpl) {                // (the scratch of the three-launch per-hop step: sized here, so that the steps of this batch allocate nothing)
        const int rc = ensure_bsplit(h, B);
        if (rc != FE_OK) return rc;
    }
    if (h->fimpl) {
        const int rc = ensure_fsplit(h, B);
        if (rc != FE_OK) return rc;
    }
    if (h->limpl) {
        const int rc = ensure_lsplit(h, B);
        if (rc != FE_OK) return rc;
    }
    return FE_OK;
}

static int run_step(fe_handle* h, const float* wav_in, size_t in_stride, float* state, float* wav_out, size_t out_stride,
                    int B, int T, float* dbg, unsigned long long* clk, void* stream) {
    int rc = check_ready(h);
    if (rc != FE_OK) return rc;
    KernelLogScope klog_(h);
    if (!wav_in || !state || !wav_out || B <= 0 || T <= 0) return fail(FE_ERR_INVALID_ARG, "bad argument");
    const Dims& d = h->d;
    if (in_stride < (size_t)T * d.HOP && B > 1) return fail(FE_ERR_INVALID_ARG, "in_stride %zu < T*H", in_stride);
    if (out_stride < (size_t)T * d.HOP && B > 1) return fail(FE_ERR_INVALID_ARG, "out_stride %zu < T*H", out_stride);
    if (h->limpl) {
        fe::LArgs la = lisennet_args(h, B, T);
        la.clk = clk;
        la.dbg = dbg;
        la.dbg_stride = h->limpl->dbg_floats;
        const size_t ovl_b = (size_t)(d.NFFT - d.HOP);
        la.mode = fe::FE_MODE_STREAM;
        la.wav_in = wav_in; la.wav_out = wav_out; la.in_stride = in_stride; la.out_stride = out_stride;
        la.cache_stft = state; la.cache_istft = state + (size_t)B * ovl_b; la.cache = state + 2 * (size_t)B * ovl_b;
        return launch_lisennet(h, la, stream);
    }
    if (h->fimpl) {
        fe::FArgs fa = fspen_args(h, B, T);
        fa.clk = clk;
        fa.dbg = dbg;
        fa.dbg_stride = h->fimpl->dbg_floats;
        const size_t ovl_b = (size_t)(d.NFFT - d.HOP);
        fa.mode = fe::FE_MODE_STREAM;
        fa.wav_in = wav_in; fa.wav_out = wav_out; fa.in_stride = in_stride; fa.out_stride = out_stride;
        fa.cache_stft = state; fa.cache_istft = state + (size_t)B * ovl_b; fa.gru = state + 2 * (size_t)B * ovl_b;
        return launch_fspen(h, fa, stream);
    }
    if (h->bimpl) {
        fe::BArgs ba = bsrnn_args(h, B, T);
        ba.clk = clk;
        ba.dbg = dbg;
        ba.dbg_stride = h->bimpl->dbg_floats;
        const size_t ovl_b = (size_t)(d.NFFT - d.HOP);
        ba.mode = fe::FE_MODE_STREAM;
        ba.wav_in = wav_in; ba.wav_out = wav_out; ba.in_stride = in_stride; ba.out_stride = out_stride;
        ba.cache_stft = state; ba.cache_istft = state + (size_t)B * ovl_b; ba.lstm = state + 2 * (size_t)B * ovl_b;
        return launch_bsrnn(h, ba, stream);
    }
    if (d.BD) return fail(FE_ERR_UNSUPPORTED_CONFIG, "the noncausal model has no streaming step (models/fastenhancer/noncausal/model.py has the offline Model only): use fe_offline");
    rc = ensure_scratch(h, B);
    if (rc != FE_OK) return rc;
    fe::FrameArgs a = base_args(h, B, T);
    const size_t ovl = (size_t)(d.NFFT - d.HOP);
    a.wav_in = wav_in;
    a.wav_out = wav_out;
    a.in_stride = in_stride;
    a.out_stride = out_stride;
    a.cache_stft = state;
    a.cache_istft = state + (size_t)B * ovl;
    a.h = state + 2 * (size_t)B * ovl;
    a.tk = a.h + (size_t)B * d.hstate();
    a.dbg = dbg;
    a.clk = clk;
    a.dbg_stride = h->impl->dbg_floats;
    hipError_t e = hipSuccess;
    a.mode = fe::FE_MODE_STREAM;
    // per-hop launches above the streams the shape's own plan holds at once (#CUs; 2 x #CUs for T): the low-LDS companion (two workgroups per CU, three
    // for the T shapes; same packed weights - Pack<S> does not depend on LOW), where one is compiled and measured faster
    const fe::Impl* im = h->impl;
    if (h->impl_many && h->opt[OPT_LOW_LDS_COMPANION] && T == 1 && B > h->max_wgs * h->impl->occ && (h->impl_many->many_persist || B <= h->max_wgs * h->impl_many->occ) &&
        (h->impl_many->many_one_round || B > h->max_wgs * h->impl_many->occ) && !dbg && !clk &&
        !(h->step_kernel == FE_STEP_KERNEL_WG8_PERSIST && h->impl->wg8))
        im = h->impl_many;
    h->last_shape = im->name;
    im->launch(a, h->max_wgs, (hipStream_t)stream, &e);
    if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return FE_OK;
}

int fe_step(fe_handle* h, const float* wav_in_dev, size_t in_stride, float* state_dev, float* wav_out_dev, size_t out_stride,
            int B, int T, void* stream) {
    return run_step(h, wav_in_dev, in_stride, state_dev, wav_out_dev, out_stride, B, T, nullptr, nullptr, stream);
}

int fe_step_host(fe_handle* h, const float* wav_in_host, size_t in_stride, float* state_dev, float* wav_out_host, size_t out_stride,
                 int B, int T, int n_calls, float* work_dev, void* stream) {
    int rc = check_ready(h);
    if (rc != FE_OK) return rc;
    if (!wav_in_host || !wav_out_host || !state_dev || !work_dev || B <= 0 || T <= 0 || n_calls <= 0) return fail(FE_ERR_INVALID_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (hipStream_t& s : h->host_streams)
        if (!s) FE_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (hipEvent_t& e : h->host_events)
        if (!e) FE_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipStream_t s_in = h->host_streams[0], s_out = h->host_streams[1];
    // events: [0] start, [1 + buf] block copied in, [3 + buf] block computed, [5 + buf] block copied out
    hipEvent_t* ev = h->host_events;
    const size_t row = (size_t)T * h->d.HOP, blk = (size_t)B * row;
    float* xd[2] = {work_dev, work_dev + blk};
    float* yd[2] = {work_dev + 2 * blk, work_dev + 3 * blk};
    FE_HIP_CHECK(hipEventRecord(ev[0], st));
    FE_HIP_CHECK(hipStreamWaitEvent(s_in, ev[0], 0));
    FE_HIP_CHECK(hipStreamWaitEvent(s_out, ev[0], 0));
    for (int c = 0; c < n_calls; ++c) {
        const int buf = c & 1;
        // copy-in of block c: its staging buffer was last read by the kernel of block c - 2
        if (c >= 2) FE_HIP_CHECK(hipStreamWaitEvent(s_in, ev[3 + buf], 0));
        FE_HIP_CHECK(hipMemcpy2DAsync(xd[buf], row * sizeof(float), wav_in_host + (size_t)c * row, in_stride * sizeof(float), row * sizeof(float), (size_t)B,
                                      hipMemcpyHostToDevice, s_in));
        FE_HIP_CHECK(hipEventRecord(ev[1 + buf], s_in));
        // kernel of block c: input landed, and the output staging buffer has been copied out (block c - 2)
        FE_HIP_CHECK(hipStreamWaitEvent(st, ev[1 + buf], 0));
        if (c >= 2) FE_HIP_CHECK(hipStreamWaitEvent(st, ev[5 + buf], 0));
        rc = run_step(h, xd[buf], row, state_dev, yd[buf], row, B, T, nullptr, nullptr, stream);
        if (rc != FE_OK) return rc;
        FE_HIP_CHECK(hipEventRecord(ev[3 + buf], st));
        // copy-out of block c
        FE_HIP_CHECK(hipStreamWaitEvent(s_out, ev[3 + buf], 0));
        FE_HIP_CHECK(hipMemcpy2DAsync(wav_out_host + (size_t)c * row, out_stride * sizeof(float), yd[buf], row * sizeof(float), row * sizeof(float), (size_t)B,
                                      hipMemcpyDeviceToHost, s_out));
        FE_HIP_CHECK(hipEventRecord(ev[5 + buf], s_out));
    }
    for (int buf = 0; buf < (n_calls < 2 ? n_calls : 2); ++buf) FE_HIP_CHECK(hipStreamWaitEvent(st, ev[5 + buf], 0));
    return FE_OK;
}

int fe_debug_step(fe_handle* h, const float* wav_in_dev, size_t in_stride, float* state_dev, float* wav_out_dev,
                  size_t out_stride, int B, float* dbg_dev, void* stream) {
    if (!dbg_dev) return fail(FE_ERR_INVALID_ARG, "null dbg buffer");
    return run_step(h, wav_in_dev, in_stride, state_dev, wav_out_dev, out_stride, B, 1, dbg_dev, nullptr, stream);
}

int fe_profile_step(fe_handle* h, const float* wav_in_dev, size_t in_stride, float* state_dev, float* wav_out_dev,
                    size_t out_stride, int B, int T, unsigned long long* clk_dev, void* stream) {
    if (!clk_dev) return fail(FE_ERR_INVALID_ARG, "null clock buffer");
    return run_step(h, wav_in_dev, in_stride, state_dev, wav_out_dev, out_stride, B, T, nullptr, clk_dev, stream);
}

constexpr int kMaxPipeFrames = 64;      // frames in flight per stream the work-buffer rings are sized for

// workgroups per stream of a time-pipelined launch (0: one workgroup walks the frames of a stream)
static int pipe_width(const fe_handle* h, int B, int T, bool offline = false, bool spec_rings = false) {
    if (!h->impl || h->pipe_frames == 0 || h->pipe_frames == 1 || T < 4 || 2 * B > h->max_wgs) return 0;
    // (time_kernel variant: its convs' inputs are handed from frame to frame through rings - in fe_offline's work buffer, or (r4v) the handle's
    //  for a spec -> spec step with caches)
    if (h->d.KT > 1 && !offline && !spec_rings) return 0;
    if (h->d.TA && !offline && !spec_rings) return 0;   // (dptransformer: per-frame K / V rings - fe_offline's work buffer, or the handle's for fe_spec_step, r4v)
    // automatic width: a hand-off (counter round trip + state fetch + the h half of the GRU + gates + publish) takes
    // ~2.6 us whatever the model; a frame takes ~4 us per MFLOP/frame at the measured kernel efficiency: that many frames
    // are worth having in flight (measured optimum: T 8-12, B 16, 48 kHz B 24, L > 24), more only adds pollers
    int want = h->pipe_frames;
    if (want < 0) {
        want = (int)(fe_flops_per_frame(h) / 6.0e5) + 2;
        if (h->d.TA) want *= 2;     // (dptransformer: a frame spends half its time streaming its K / V window - measured 16 -> 32 in flight: 1.37 -> 0.73 ms)
        want = want < 8 ? 8 : (want > 64 ? 64 : want);
    }
    int p = h->max_wgs / B;
    p = p < want ? p : want;
    return p < T ? p : T;
}

// BSRNN: frames in flight per utterance of a time-pipelined offline launch (0: one workgroup walks the frames).  A hand-off chain
// (wait, fetch, gate GEMM, publish) is ~1/40 of a frame, so every co-resident workgroup the batch leaves free is worth having.
static int bsrnn_pipe_width(const fe_handle* h, int B, int T) {
    if (!h->bimpl || !h->bimpl->launch_pipe || h->pipe_frames == 0 || h->pipe_frames == 1 || T < 4) return 0;
    int p = (h->max_wgs * h->bimpl->occ) / B;
    const int want = h->pipe_frames < 0 ? 64 : h->pipe_frames;
    p = p < want ? p : want;
    p = p < T ? p : T;
    return p >= 2 ? p : 0;
}

int fe_set_time_pipeline(fe_handle* h, int frames_in_flight) {
    if (!h) return fail(FE_ERR_INVALID_ARG, "bad argument");
    // The per-frame rings in work_dev (time_kernel inputs, dptransformer K / V, LiSenNet caches) are sized for at most kMaxPipeFrames
    // frames in flight (fe_offline_work_floats), and more never paid on any model: larger requests are clamped, not refused.
    h->pipe_frames = frames_in_flight > kMaxPipeFrames ? kMaxPipeFrames : frames_in_flight;
    return FE_OK;
}

static size_t tb_work_floats(const fe_handle* h, int B, int T, size_t* off);
static int tb_run(fe_handle* h, fe::tb::TbArgs a0, float* work_dev, int B, int T, hipStream_t st);
static int pipe_width(const fe_handle* h, int B, int T, bool offline, bool spec_rings);

// spec -> spec chunks on the time-batched engine: when asked for (FE_OFFLINE_TIME_BATCHED), or - AUTO - for long chunks of batches that
// the time pipeline cannot take (more than #CUs / 2 streams: each workgroup would walk its T frames alone) or that are simply large
static bool use_tb_spec(const fe_handle* h, int B, int T) {
    if (!h->impl || !h->impl->tb || h->d.BD || h->offline_engine == FE_OFFLINE_FRAME_WALK || T < 2) return false;
    if (h->offline_engine == FE_OFFLINE_TIME_BATCHED) return true;
    if (h->d.C2 >= 72) return false;
    // ("cannot take" = too many streams - NOT a caller's fe_set_time_pipeline(0 / 1), which asks for the serial walk and gets it)
    if (h->pipe_frames == 0 || h->pipe_frames == 1) return false;
    return T >= 16 && (2 * B > h->max_wgs || (long)B * T >= 2048);
}

int fe_spec_step(fe_handle* h, const float* spec_in_dev, float* h_dev, float* spec_out_dev, int B, int T, void* stream) {
    int rc = check_ready(h);
    if (rc != FE_OK) return rc;
    KernelLogScope klog_(h);
    if (!spec_in_dev || !h_dev || !spec_out_dev || B <= 0 || T <= 0) return fail(FE_ERR_INVALID_ARG, "bad argument");
    if (h->limpl) {
        fe::LArgs la = lisennet_args(h, B, T);
        la.mode = fe::FE_MODE_SPEC;
        la.spec_in = spec_in_dev; la.spec_out = spec_out_dev; la.cache = h_dev;
        return launch_lisennet(h, la, stream);
    }
    if (h->fimpl) {
        fe::FArgs fa = fspen_args(h, B, T);
        fa.mode = fe::FE_MODE_SPEC;
        fa.spec_in = spec_in_dev; fa.spec_out = spec_out_dev; fa.gru = h_dev;
        return launch_fspen(h, fa, stream);
    }
    if (h->bimpl) {
        fe::BArgs ba = bsrnn_args(h, B, T);
        ba.mode = fe::FE_MODE_SPEC;
        ba.spec_in = spec_in_dev; ba.spec_out = spec_out_dev; ba.lstm = h_dev;
        return launch_bsrnn(h, ba, stream);
    }
    if (h->d.BD) return fail(FE_ERR_UNSUPPORTED_CONFIG, "the noncausal model has no spec -> spec step with caches (models/fastenhancer/noncausal/model.py has the offline Model only): use fe_offline");
    if (use_tb_spec(h, B, T)) {
        // the chunk as one encoder pass, per block a scan that starts from the caller's GRU state and leaves the new one + a tile pass,
        // one decoder pass (fe_spec_step has no work buffer argument: the handle keeps a grow-only one - a first / larger call allocates)
        const size_t need = tb_work_floats(h, B, T, nullptr);
        if (need > h->tb_work_floats) {
            if (h->tb_work_dev) { FE_HIP_CHECK(hipFree(h->tb_work_dev)); h->tb_work_dev = nullptr; h->tb_work_floats = 0; }
            FE_HIP_CHECK(hipMalloc(&h->tb_work_dev, need * sizeof(float)));
            h->tb_work_floats = need;
        }
        fe::tb::TbArgs ta{};
        ta.wp = h->packed_dev;
        ta.spec_in = spec_in_dev; ta.spec_out = spec_out_dev;
        ta.hstate = h_dev; ta.h_init = 1;
        ta.mode = fe::FE_MODE_SPEC;
        ta.compression = h->cfg.input_compression;
        return tb_run(h, ta, h->tb_work_dev, B, T, (hipStream_t)stream);
    }
    rc = ensure_scratch(h, B);
    if (rc != FE_OK) return rc;
    fe::FrameArgs a = base_args(h, B, T);
    a.spec_in = spec_in_dev;
    a.spec_out = spec_out_dev;
    a.h = h_dev;
    a.tk = h_dev + (size_t)B * h->d.hstate();
    hipError_t e = hipSuccess;
    a.mode = fe::FE_MODE_SPEC;
    if (const int P = pipe_width(h, B, T, false, true)) {
        hipStream_t st = (hipStream_t)stream;
        const size_t per = (size_t)h->d.KB + (h->d.KT > 1 ? 2 * h->d.NL : 0);      // counters per stream (fe_kernels.hip.h: NFLAG)
        const size_t nflags = (size_t)h->max_wgs * per;
        if (!h->pipe_flags_dev) FE_HIP_CHECK(hipMalloc(&h->pipe_flags_dev, nflags * sizeof(unsigned int)));
        FE_HIP_CHECK(hipMemsetAsync(h->pipe_flags_dev, 0, (size_t)B * per * sizeof(unsigned int), st));
        a.pipe_flags = h->pipe_flags_dev;
        a.pipe_p = P;
        // dptransformer / time_kernel (r4v): the frames of a chunk in flight exchange their k / v (their convs' inputs) through rings of
        // L + P slots; the caller's caches go in before the launch and the chunk's last L frames come back after it (the handle keeps the
        // rings: grow-only, a first / wider call allocates)
        const Dims& d = h->d;
        const bool rings = d.TA != 0 || d.KT > 1;
        const int L = d.TA ? d.TA : d.KT - 1, RS = L + P;
        const int PAIRS = d.TA ? d.F2 * 4 : 1, SZ = d.TA ? d.C2 / 4 : d.F1 * d.C1;
        const size_t rows = (size_t)(d.TA ? 2 * d.KB : 2 * d.NL) * B * PAIRS, nfl = rows * L * SZ;
        float* caches = d.TA ? h_dev : h_dev + (size_t)B * d.hstate();
        float* heads = d.TA ? h_dev + (size_t)B * (d.hstate() - 1) : nullptr;     // (the ring heads sit behind the K / V caches: fe_kernels.hip.h, ring_head0)
        const unsigned cgrid = (unsigned)((nfl + 255) / 256);
        if (rings) {
            const size_t need = rows * RS * SZ;
            if (need > h->spec_ring_floats) {
                if (h->spec_ring_dev) { FE_HIP_CHECK(hipFree(h->spec_ring_dev)); h->spec_ring_dev = nullptr; h->spec_ring_floats = 0; }
                FE_HIP_CHECK(hipMalloc(&h->spec_ring_dev, need * sizeof(float)));
                h->spec_ring_floats = need;
            }
            hipLaunchKernelGGL(ring_in_kernel, dim3(cgrid), dim3(256), 0, st, caches, heads, h->spec_ring_dev, nfl, B, PAIRS, SZ, L, RS);
            if (d.TA) { a.h = h->spec_ring_dev; a.tatt_base = d.TA; }
            else { a.tk = h->spec_ring_dev; a.tk_base = d.KT - 1; }
        }
        h->impl->launch_pipe(a, st, &e);
        if (e != hipSuccess) {     // the runtime refused co-residency (GPU shared with other work): walk the frames serially
            (void)hipGetLastError();
            a.pipe_p = 0;
            a.pipe_flags = nullptr;
            a.h = h_dev;
            a.tk = h_dev + (size_t)B * d.hstate();
            a.tatt_base = 0;
            a.tk_base = 0;
            h->impl->launch(a, h->max_wgs, st, &e);
        } else if (rings) {
            hipLaunchKernelGGL(ring_out_kernel, dim3(cgrid), dim3(256), 0, st, h->spec_ring_dev, caches, heads, nfl, B, SZ, L, RS, T);
            e = hipGetLastError();
        }
    } else
    h->impl->launch(a, h->max_wgs, (hipStream_t)stream, &e);
    if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return FE_OK;
}

// floats of the time-batched engine's work buffer: xc | skip | x | gx | hs | frames | carried GRU state, each a multiple of 4 floats
static size_t tb_work_floats(const fe_handle* h, int B, int T, size_t* off /*[7] or nullptr*/) {
    const Dims& d = h->d;
    const size_t NF = (size_t)B * T, nd = d.BD ? 2 : 1;
    const size_t sz[7] = {NF * 2 * d.F0, NF * (d.NL + 1) * d.F1 * d.C1, NF * d.F2 * d.C2, nd * NF * d.F2 * 3 * d.C2, NF * d.F2 * nd * d.C2, NF * d.NFFT,
                          (size_t)d.KB * nd * B * d.F2 * d.C2};
    size_t cur = 0;
    for (int i = 0; i < 7; ++i) {
        if (off) off[i] = cur;
        cur += (sz[i] + 3) & ~(size_t)3;
    }
    return cur;
}

// The time-batched engine's schedule.  The call's B x T frames are cut into NODES - G groups of utterances x NC chunks of
// consecutive frames - and every node runs the chain  enc -> (scan k -> blk k) x KB -> dec  on its own slice of the work buffers.
// Only the scans are serial in time: scan k of chunk c starts from the state scan k of chunk c - 1 left (an event), everything
// else of a node depends on the node alone.  The nodes are spread round-robin over a few HIP streams of the handle, so that
// the latency-bound scans of one node (16 rows per workgroup, one recurrence step after the other) run UNDER the GEMM passes
// of the others instead of leaving the chip idle between them, and - a single utterance - the scans of consecutive blocks
// pipeline through the chunks.  The noncausal model's reverse scan needs all frames of an utterance: utterance groups only.
struct TbPlan { int G, NC, NS, stagger, fuse; };

static TbPlan tb_plan(const fe_handle* h, int B, int T) {
    auto env = [](const char* n, int dflt) { const char* v = std::getenv(n); return v ? std::atoi(v) : dflt; };
    TbPlan p;
    // Measured (profiles/r3d_tb_node_plans.txt): on this runtime the streams of one process share 3-4 hardware queues and kernels that
    // overlap slow each other down by what the overlap hides - every multi-node plan came out slower than ONE node.  The default is
    // therefore one node; FE_TB_NC / FE_TB_G / FE_TB_STREAMS keep the cut available (results are bit-identical by construction).
    p.NC = std::max(1, std::min(T, env("FE_TB_NC", 1)));
    if (h->d.BD) p.NC = 1;
    p.G = std::max(1, std::min(B, env("FE_TB_G", 1)));
    p.NS = std::max(1, std::min(8, env("FE_TB_STREAMS", 4)));
    p.NS = std::min(p.NS, p.G * p.NC);
    p.stagger = env("FE_TB_STAGGER", 1);
    // FE_TB_FUSE=1: scan + block tiles of a block in ONE cooperative launch (tb_stage_kernel: the tiles follow the scan frame by frame
    // behind its progress counters) where tb_launch accepts it.  Parity-green, but measured slower than the two launches it replaces
    // (profiles/r3g_tb_fused_stage.txt): off by default.
    p.fuse = env("FE_TB_FUSE", 0);
    if (p.G * p.NC > 1) p.fuse = 0;
    if (std::getenv("FE_TB_STAGES")) { p.G = p.NC = p.NS = 1; p.fuse = 0; }
    return p;
}

static int tb_run(fe_handle* h, fe::tb::TbArgs a0, float* work_dev, int B, int T, hipStream_t st) {
    const Dims& d = h->d;
    const fe::tb::TbImpl* tbi = h->impl->tb;
    const TbPlan p = tb_plan(h, B, T);
    const int nodes = p.G * p.NC;
    size_t off[7];
    tb_work_floats(h, B, T, off);
    const size_t nd = d.BD ? 2 : 1;
    a0.Bfull = B; a0.Tfull = T;
#ifdef FE_TB_PROBE
    if (!h->tb_probe_dev) {
        FE_HIP_CHECK(hipMalloc(&h->tb_probe_dev, 4 * fe::tb::kProbeSlots * sizeof(unsigned long long)));
        FE_HIP_CHECK(hipMemset(h->tb_probe_dev, 0, 4 * fe::tb::kProbeSlots * sizeof(unsigned long long)));
    }
    a0.probe = h->tb_probe_dev;
#endif
    const bool caller_state = a0.hstate != nullptr;     // caches handed in: read by the first chunk (h_init), left updated by the last
    if (!caller_state && p.NC > 1) a0.hstate = work_dev + off[6];
    a0.frames = work_dev + off[5];
    hipError_t e = hipSuccess;
    if (p.fuse) {
        const size_t nprog = (size_t)d.KB * h->max_wgs * fe::tb::kProgStride;        // (a 128-byte line per scan workgroup)
        if (!h->tb_prog_dev) FE_HIP_CHECK(hipMalloc(&h->tb_prog_dev, nprog * sizeof(unsigned int)));
        FE_HIP_CHECK(hipMemsetAsync(h->tb_prog_dev, 0, nprog * sizeof(unsigned int), st));
    }
    // (FE_TB_STAGES=n: stop after n launches - tools/gpu_tb_check.py reads the intermediate buffers out of work_dev)
    const char* lim_s = std::getenv("FE_TB_STAGES");
    int lim = lim_s ? std::atoi(lim_s) : 1 << 30;
    const bool multi = p.NS > 1;
    if (multi) {
        while ((int)h->tb_streams.size() < p.NS) {
            hipStream_t s;
            FE_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            h->tb_streams.push_back(s);
        }
        const size_t nev = 1 + (size_t)p.NS + (size_t)nodes * (d.KB + 1);
        while (h->tb_events.size() < nev) {
            hipEvent_t ev;
            FE_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            h->tb_events.push_back(ev);
        }
        FE_HIP_CHECK(hipEventRecord(h->tb_events[0], st));
        for (int s = 0; s < p.NS; ++s) FE_HIP_CHECK(hipStreamWaitEvent(h->tb_streams[s], h->tb_events[0], 0));
    }
    auto scan_event = [&](int node, int k) { return h->tb_events[1 + p.NS + (size_t)node * (d.KB + 1) + k]; };
    auto enc_event = [&](int node) { return h->tb_events[1 + p.NS + (size_t)node * (d.KB + 1) + d.KB]; };
    size_t fbase = 0;                            // frames of the nodes before this one: its offset into the work buffers
    for (int g = 0; g < p.G && e == hipSuccess; ++g) {
        const int b0 = (int)((long)B * g / p.G), b1 = (int)((long)B * (g + 1) / p.G);
        for (int c = 0; c < p.NC && e == hipSuccess; ++c) {
            const int t0 = (int)((long)T * c / p.NC), t1 = (int)((long)T * (c + 1) / p.NC);
            const int node = g * p.NC + c, sidx = node % p.NS;
            hipStream_t ns = multi ? h->tb_streams[sidx] : st;
            fe::tb::TbArgs a = a0;
            a.B = b1 - b0; a.T = t1 - t0; a.NF = a.B * a.T; a.b0 = b0; a.t0 = t0;
            a.h_init = (c > 0 || (caller_state && a0.h_init)) ? 1 : 0;
            a.xc = work_dev + off[0] + fbase * 2 * d.F0;
            a.skip = work_dev + off[1] + fbase * (size_t)(d.NL + 1) * d.F1 * d.C1;
            a.x = work_dev + off[2] + fbase * (size_t)d.F2 * d.C2;
            a.gx = work_dev + off[3] + nd * fbase * (size_t)d.F2 * 3 * d.C2;
            a.hs = work_dev + off[4] + fbase * (size_t)d.F2 * nd * d.C2;
            fbase += (size_t)a.NF;
            if (a.NF == 0) continue;
            // stagger: a node's encoder pass starts when the previous node's has finished - nodes that start together run in lockstep
            // (GEMM passes together, then all of them in their scans with the chip idle)
            if (multi && p.stagger && node > 0 && (node - 1) % p.NS != sidx) FE_HIP_CHECK(hipStreamWaitEvent(ns, enc_event(node - 1), 0));
            if (lim-- > 0) tbi->launch(fe::tb::TB_ENC, a, h->max_wgs, ns, &e);
            if (multi && p.stagger && node + 1 < nodes) FE_HIP_CHECK(hipEventRecord(enc_event(node), ns));
            for (int k = 0; k < d.KB && e == hipSuccess; ++k) {
                a.k = k;
                if (p.fuse) {
                    a.prog = h->tb_prog_dev + (size_t)k * h->max_wgs * fe::tb::kProgStride;
                    hipError_t ef = hipSuccess;
                    tbi->launch(fe::tb::TB_STAGE, a, h->max_wgs, ns, &ef);
                    if (ef == hipSuccess) continue;             // (refused: not a fusable shape / batch, or no co-residency - two launches)
                }
                if (multi && c > 0 && (node - 1) % p.NS != sidx) FE_HIP_CHECK(hipStreamWaitEvent(ns, scan_event(node - 1, k), 0));
                if (lim-- > 0) tbi->launch(fe::tb::TB_SCAN, a, h->max_wgs, ns, &e);
                if (multi && c + 1 < p.NC) FE_HIP_CHECK(hipEventRecord(scan_event(node, k), ns));
                if (e == hipSuccess && lim-- > 0) tbi->launch(fe::tb::TB_BLK, a, h->max_wgs, ns, &e);
            }
            if (e == hipSuccess && lim-- > 0) tbi->launch(fe::tb::TB_DEC, a, h->max_wgs, ns, &e);
        }
    }
    if (multi) {
        for (int s = 0; s < p.NS; ++s) {
            FE_HIP_CHECK(hipEventRecord(h->tb_events[1 + s], h->tb_streams[s]));
            FE_HIP_CHECK(hipStreamWaitEvent(st, h->tb_events[1 + s], 0));
        }
    }
    if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return FE_OK;
}

// FE_OFFLINE_AUTO: the time-batched engine, except for the big shapes (M, L and their 48 kHz forms: block weights streamed from L2,
// one frame per tile) once there are enough utterances for the time-pipelined frame walk to fill the chip on its own - measured on
// FastEnhancer_L, 4 s: 1 utterance 5.2 ms time-batched / 14.2 ms walk, 16 utterances 20.6 / 16.8 ms (profiles/r3d_tb_timing.txt)
constexpr int kAutoWalkFrom = 8;
static bool use_tb_offline(const fe_handle* h, int B) {
    if (!h->impl || !h->impl->tb) return false;
    if (h->d.BD) return true;
    if (h->offline_engine == FE_OFFLINE_AUTO) return !(h->d.C2 >= 72 && B >= kAutoWalkFrom);
    return h->offline_engine != FE_OFFLINE_FRAME_WALK;
}

int fe_set_step_kernel(fe_handle* h, int kernel) {
    if (!h || kernel < FE_STEP_KERNEL_WAVES4 || kernel > FE_STEP_KERNEL_WG8_PERSIST) return fail(FE_ERR_INVALID_ARG, "bad argument");
    h->step_kernel = kernel;
    return FE_OK;
}

int fe_set_option(fe_handle* h, const char* name, int value) {
    if (!h || !name) return fail(FE_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (std::strcmp(name, kOptions[i].name) == 0) {
            if (value < kOptions[i].lo || value > kOptions[i].hi)
                return fail(FE_ERR_INVALID_ARG, "fe_set_option(%s): %d is outside [%d, %d]", name, value, kOptions[i].lo, kOptions[i].hi);
            h->opt[i] = value;
            return FE_OK;
        }
    return fail(FE_ERR_INVALID_ARG, "fe_set_option: no option named '%s'", name);
}

int fe_get_option(const fe_handle* h, const char* name, int* value) {
    if (!h || !name || !value) return fail(FE_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (std::strcmp(name, kOptions[i].name) == 0) { *value = h->opt[i]; return FE_OK; }
    return fail(FE_ERR_INVALID_ARG, "fe_get_option: no option named '%s'", name);
}

int fe_options(void) { return OPT_COUNT; }
const char* fe_option_name(int idx) { return idx >= 0 && idx < OPT_COUNT ? kOptions[idx].name : nullptr; }

const char* fe_last_step_kernel(const fe_handle* h) {
    if (!h) return "";
    std::string& s = h->last_text;
    s.clear();
    const int n = h->n_last_kernels < 16 ? h->n_last_kernels : 16;
    for (int i = 0; i < n;) {
        int j = i;
        while (j < n && h->last_kernels[j] == h->last_kernels[i]) ++j;
        if (!s.empty()) s += " + ";
        if (j - i > 1) s += std::to_string(j - i) + " x ";
        s += h->last_kernels[i];
        i = j;
    }
    if (h->n_last_kernels > 16) s += " + ...";
    if (!s.empty() && h->last_shape) s += std::string(" [shape ") + h->last_shape + "]";
    return s.c_str();
}

int fe_set_offline_engine(fe_handle* h, int engine) {
    if (!h || engine < FE_OFFLINE_AUTO || engine > FE_OFFLINE_TIME_BATCHED) return fail(FE_ERR_INVALID_ARG, "bad argument");
    if (engine == FE_OFFLINE_TIME_BATCHED && !(h->impl && h->impl->tb)) return fail(FE_ERR_UNSUPPORTED_CONFIG, "no time-batched engine is compiled for this model");
    if (engine == FE_OFFLINE_FRAME_WALK && h->d.BD) return fail(FE_ERR_UNSUPPORTED_CONFIG, "the noncausal model runs on the time-batched engine only");
    h->offline_engine = engine;
    return FE_OK;
}

// fe_offline on the time-batched engine: encoder pass, per block (scan over time, attention pass), decoder pass, overlap-add.
// Tw_b_dev != nullptr: a ragged batch laid out for Tw (= the longest utterance), utterance b has Tw_b_dev[b] samples.
static int offline_tb(fe_handle* h, const float* noisy_dev, size_t in_stride, const int* Tw_b_dev, int Tw, int B, float* wav_hat_dev, size_t out_stride,
                      float* spec_hat_dev, float* work_dev, hipStream_t st) {
    const Dims& d = h->d;
    const int T = 1 + Tw / d.HOP;
    int rc = ensure_tables(h, st);
    if (rc != FE_OK) return rc;
    fe::tb::TbArgs a{};
    a.wp = h->packed_dev;
    a.wav_in = noisy_dev; a.in_stride = in_stride; a.Tw = Tw; a.Tw_b = Tw_b_dev;
    a.spec_out = spec_hat_dev;
    a.hstate = nullptr;                      // zero initial state (model.py:626-627)
    a.mode = fe::FE_MODE_OFFLINE;
    a.compression = h->cfg.input_compression;
    rc = tb_run(h, a, work_dev, B, T, st);
    if (rc != FE_OK) return rc;
    size_t off[7];
    tb_work_floats(h, B, T, off);
    a.frames = work_dev + off[5];
    const int n_out = d.HOP * (T - 1);
    fe::note_kernel("istft_ola_kernel");
        hipLaunchKernelGGL(fe::istft_ola_kernel, dim3((n_out + fe::kThreads - 1) / fe::kThreads, B), dim3(fe::kThreads), 0, st,
                       a.frames, h->tables_dev, wav_hat_dev, out_stride, d.NFFT, d.HOP, T, Tw_b_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return FE_OK;
}

size_t fe_offline_work_floats(const fe_handle* h, int B, int Tw) {
    if (!h || B <= 0 || Tw <= 0) return 0;
    const Dims& d = h->d;
    if (h->impl && h->impl->tb) {
        // Sized for the engine the CURRENT fe_set_offline_engine setting selects (FastEnhancer_L x 16 x 4 s: the time-batched buffers are
        // 3 GB against the frame walk's 25 MB), and MONOTONE in B under that setting: a buffer sized for B serves every batch of at most B
        // utterances of at most Tw samples.  Under AUTO the big shapes walk from 8 utterances on but take the time-batched engine below
        // that, so their size covers the time-batched need of min(B, 7) as well.  After fe_set_offline_engine: query again (header).
        const int T = 1 + Tw / d.HOP;
        const size_t walk = d.BD ? 0 : (size_t)B * ((size_t)(d.NFFT - d.HOP) + d.hstate() + tk_floats(h)) + (((size_t)B * d.KB + 3) & ~(size_t)3) + (size_t)B * T * d.NFFT;
        if (d.BD || use_tb_offline(h, B)) return std::max(walk, tb_work_floats(h, B, T, nullptr));
        if (h->offline_engine == FE_OFFLINE_AUTO) return std::max(walk, tb_work_floats(h, std::min(B, kAutoWalkFrom - 1), T, nullptr));
        return walk;
    }
    if (h->limpl) {     // tail + caches, and the time pipeline's counters, windowed frames and cache ring (widest pipeline: 64 + 2 slots)
        const int T = 1 + Tw / d.HOP;
        const size_t cf = (size_t)B * h->limpl->cache_floats;
        return (size_t)B * (size_t)(d.NFFT - d.HOP) + ((cf + 3) & ~(size_t)3) + (((size_t)B * h->limpl->nsite + 3) & ~(size_t)3) + (size_t)B * T * d.NFFT + (size_t)B * 66 * h->limpl->cache_floats;
    }
    if (h->fimpl) {     // tail + inter-GRU states, and the time pipeline's frame counters + windowed frames
        const int T = 1 + Tw / d.HOP;
        return (size_t)B * (size_t)(d.NFFT - d.HOP) + ((fspen_gru_floats(B) + 3) & ~(size_t)3) + (((size_t)B * h->fimpl->num_blocks + 3) & ~(size_t)3) + (size_t)B * T * d.NFFT;
    }
    if (h->bimpl) {     // overlap-add tail + LSTM states, and - whatever fe_set_time_pipeline says at call time - the frame counters and the windowed frames
        const int T = 1 + Tw / d.HOP;
        return (size_t)B * (size_t)(d.NFFT - d.HOP) + ((bsrnn_lstm_floats(h, B) + 3) & ~(size_t)3) + (((size_t)B * h->cfg.rf_blocks + 3) & ~(size_t)3) + (size_t)B * T * d.NFFT;
    }
    // GRU state (zero initial state, model.py:626-627) + overlap-add tail, both zeroed by fe_offline; time-pipelined
    // launches: + the frame counters [B][KB] and the windowed output frames [B][T][N]
    size_t n = (size_t)B * ((size_t)(d.NFFT - d.HOP) + d.hstate() + tk_floats(h));
    const int T = 1 + Tw / d.HOP;
    // (whatever fe_set_time_pipeline says at the time of THIS call: a buffer sized with the pipeline off must still do when it is on)
    if (h->impl && T >= 4) {
        n += (((size_t)B * (d.KB + (d.KT > 1 ? 2 * d.NL : 0)) + 3) & ~(size_t)3) + (size_t)B * T * d.NFFT;
        if (d.KT > 1) n += (size_t)B * 2 * d.NL * (64 + d.KT - 1) * d.F1 * d.C1;      // the time convs' input rings at the widest pipeline
        if (d.TA) n += (size_t)B * 2 * d.KB * d.F2 * d.C2 * (d.TA + 64);                // the K / V rings at the widest pipeline
    }
    return n;
}

int fe_offline(fe_handle* h, const float* noisy_dev, int B, int Tw, float* wav_hat_dev, float* spec_hat_dev, float* work_dev,
               void* stream) {
    int rc = check_ready(h);
    if (rc != FE_OK) return rc;
    KernelLogScope klog_(h);
    if (!noisy_dev || !wav_hat_dev || !spec_hat_dev || !work_dev || B <= 0) return fail(FE_ERR_INVALID_ARG, "bad argument");
    const Dims& d = h->d;
    if (Tw <= d.NFFT / 2)   // torch.stft reflect padding needs pad < length
        return fail(FE_ERR_INVALID_ARG, "Tw=%d: reflect padding of n_fft/2=%d needs a longer input", Tw, d.NFFT / 2);
    hipStream_t st = (hipStream_t)stream;
    const int T = 1 + Tw / d.HOP;
    if (use_tb_offline(h, B)) return offline_tb(h, noisy_dev, (size_t)Tw, nullptr, Tw, B, wav_hat_dev, (size_t)d.HOP * (T - 1), spec_hat_dev, work_dev, st);
    {   // zero the state, the tail and the frame counters (not the frames: every element is written)
        size_t nz = (size_t)B * ((size_t)(d.NFFT - d.HOP) + d.hstate() + tk_floats(h));
        if (h->bimpl) nz = (size_t)B * (size_t)(d.NFFT - d.HOP) + ((bsrnn_lstm_floats(h, B) + 3) & ~(size_t)3) + (((size_t)B * h->cfg.rf_blocks + 3) & ~(size_t)3);
        else if (h->fimpl) nz = (size_t)B * (size_t)(d.NFFT - d.HOP) + ((fspen_gru_floats(B) + 3) & ~(size_t)3) + (((size_t)B * h->fimpl->num_blocks + 3) & ~(size_t)3);
        else if (h->limpl) nz = (size_t)B * (size_t)(d.NFFT - d.HOP) + (((size_t)B * h->limpl->cache_floats + 3) & ~(size_t)3) + (((size_t)B * h->limpl->nsite + 3) & ~(size_t)3);
        else if (pipe_width(h, B, T, true)) nz += ((size_t)B * (d.KB + (d.KT > 1 ? 2 * d.NL : 0)) + 3) & ~(size_t)3;
        FE_HIP_CHECK(hipMemsetAsync(work_dev, 0, nz * sizeof(float), st));
    }
    if (h->limpl) {
        fe::LArgs la = lisennet_args(h, B, T);
        la.mode = fe::FE_MODE_OFFLINE;
        la.Tw = Tw;
        la.wav_in = noisy_dev; la.in_stride = (size_t)Tw;
        la.wav_out = wav_hat_dev; la.out_stride = (size_t)d.HOP * (T - 1);
        la.spec_out = spec_hat_dev;
        la.cache_istft = work_dev; la.cache_stft = work_dev;
        la.cache = work_dev + (size_t)B * (d.NFFT - d.HOP);
        if (h->limpl->launch_pipe && h->pipe_frames != 0 && h->pipe_frames != 1 && T >= 4) {
            // the frames of an utterance over co-resident workgroups (lisennet_kernels.hip.h, PIPE); refused co-residency: the serial walk
            int P = (h->max_wgs * h->limpl->occ) / B;
            const int want = h->pipe_frames < 0 ? 32 : h->pipe_frames;
            P = P < want ? P : want;
            P = P < T ? P : T;
            if (P >= 2) {
                rc = ensure_tables(h, st);
                if (rc != FE_OK) return rc;
                const size_t cf = (size_t)B * h->limpl->cache_floats;
                float* flags = la.cache + ((cf + 3) & ~(size_t)3);
                la.pipe_flags = reinterpret_cast<unsigned int*>(flags);
                la.frames = flags + (((size_t)B * h->limpl->nsite + 3) & ~(size_t)3);
                la.ring = la.frames + (size_t)B * T * d.NFFT;
                la.pipe_p = P;
                hipError_t e = hipSuccess;
                h->limpl->launch_pipe(la, st, &e);
                if (e == hipSuccess) {
                    const int n_out = d.HOP * (T - 1);
                    fe::note_kernel("istft_ola_kernel");
        hipLaunchKernelGGL(fe::istft_ola_kernel, dim3((n_out + fe::kThreads - 1) / fe::kThreads, B), dim3(fe::kThreads), 0, st,
                                       la.frames, h->tables_dev, wav_hat_dev, (size_t)n_out, d.NFFT, d.HOP, T);
                    e = hipGetLastError();
                    if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
                    return FE_OK;
                }
                (void)hipGetLastError();
                la.pipe_flags = nullptr; la.frames = nullptr; la.ring = nullptr; la.pipe_p = 0;
            }
        }
        return launch_lisennet(h, la, stream);
    }
    if (h->fimpl) {
        fe::FArgs fa = fspen_args(h, B, T);
        fa.mode = fe::FE_MODE_OFFLINE;
        fa.Tw = Tw;
        fa.wav_in = noisy_dev; fa.in_stride = (size_t)Tw;
        fa.wav_out = wav_hat_dev; fa.out_stride = (size_t)d.HOP * (T - 1);
        fa.spec_out = spec_hat_dev;
        fa.cache_istft = work_dev; fa.cache_stft = work_dev;
        fa.gru = work_dev + (size_t)B * (d.NFFT - d.HOP);
        if (h->fimpl->launch_pipe && h->pipe_frames != 0 && h->pipe_frames != 1 && T >= 4) {
            // the frames of an utterance over co-resident workgroups (fspen_kernels.hip.h, PIPE); refused co-residency: the serial walk
            int P = (h->max_wgs * h->fimpl->occ) / B;
            const int want = h->pipe_frames < 0 ? 32 : h->pipe_frames;
            P = P < want ? P : want;
            P = P < T ? P : T;
            if (P >= 2) {
                rc = ensure_tables(h, st);
                if (rc != FE_OK) return rc;
                float* flags = fa.gru + ((fspen_gru_floats(B) + 3) & ~(size_t)3);
                fa.pipe_flags = reinterpret_cast<unsigned int*>(flags);
                fa.frames = flags + (((size_t)B * h->fimpl->num_blocks + 3) & ~(size_t)3);
                fa.pipe_p = P;
                hipError_t e = hipSuccess;
                h->fimpl->launch_pipe(fa, st, &e);
                if (e == hipSuccess) {
                    const int n_out = d.HOP * (T - 1);
                    fe::note_kernel("istft_ola_kernel");
        hipLaunchKernelGGL(fe::istft_ola_kernel, dim3((n_out + fe::kThreads - 1) / fe::kThreads, B), dim3(fe::kThreads), 0, st,
                                       fa.frames, h->tables_dev, wav_hat_dev, (size_t)n_out, d.NFFT, d.HOP, T);
                    e = hipGetLastError();
                    if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
                    return FE_OK;
                }
                (void)hipGetLastError();
                fa.pipe_flags = nullptr; fa.frames = nullptr; fa.pipe_p = 0;
            }
        }
        return launch_fspen(h, fa, stream);
    }
    if (h->bimpl) {
        fe::BArgs ba = bsrnn_args(h, B, T);
        ba.mode = fe::FE_MODE_OFFLINE;
        ba.Tw = Tw;
        ba.wav_in = noisy_dev; ba.in_stride = (size_t)Tw;
        ba.wav_out = wav_hat_dev; ba.out_stride = (size_t)d.HOP * (T - 1);
        ba.spec_out = spec_hat_dev;
        ba.cache_istft = work_dev; ba.cache_stft = work_dev;
        ba.lstm = work_dev + (size_t)B * (d.NFFT - d.HOP);
        if (const int P = bsrnn_pipe_width(h, B, T)) {
            // the frames of an utterance over P co-resident workgroups (bsrnn_kernels.hip.h, PIPE); refused co-residency: the serial walk
            rc = ensure_tables(h, st);
            if (rc != FE_OK) return rc;
            float* flags = ba.lstm + ((bsrnn_lstm_floats(h, B) + 3) & ~(size_t)3);
            ba.pipe_flags = reinterpret_cast<unsigned int*>(flags);
            ba.frames = flags + (((size_t)B * h->cfg.rf_blocks + 3) & ~(size_t)3);
            ba.pipe_p = P;
            hipError_t e = hipSuccess;
            h->bimpl->launch_pipe(ba, st, &e);
            if (e == hipSuccess) {
                const int n_out = d.HOP * (T - 1);
                fe::note_kernel("istft_ola_kernel");
        hipLaunchKernelGGL(fe::istft_ola_kernel, dim3((n_out + fe::kThreads - 1) / fe::kThreads, B), dim3(fe::kThreads), 0, st,
                                   ba.frames, h->tables_dev, wav_hat_dev, (size_t)n_out, d.NFFT, d.HOP, T);
                e = hipGetLastError();
                if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
                return FE_OK;
            }
            (void)hipGetLastError();
            ba.pipe_flags = nullptr; ba.frames = nullptr; ba.pipe_p = 0;
        }
        return launch_bsrnn(h, ba, stream);
    }
    rc = ensure_scratch(h, B);
    if (rc != FE_OK) return rc;
    fe::FrameArgs a = base_args(h, B, T);
    a.mode = fe::FE_MODE_OFFLINE;
    a.Tw = Tw;
    a.wav_in = noisy_dev;
    a.in_stride = (size_t)Tw;
    a.wav_out = wav_hat_dev;
    a.out_stride = (size_t)d.HOP * (T - 1);
    a.spec_out = spec_hat_dev;
    a.cache_istft = work_dev;
    a.cache_stft = work_dev;   // unused in this mode
    a.h = work_dev + (size_t)B * (d.NFFT - d.HOP);
    a.tk = a.h + (size_t)B * d.hstate();
    hipError_t e = hipSuccess;
    if (const int P = pipe_width(h, B, T, true)) {
        rc = ensure_tables(h, st);
        if (rc != FE_OK) return rc;
        // work buffer: tail | GRU states | time-conv caches (serial walk) | frame counters | windowed frames | time-conv input rings
        float* const tk_serial = a.tk;
        float* flags = a.tk + (size_t)B * tk_floats(h);
        a.pipe_flags = reinterpret_cast<unsigned int*>(flags);
        a.frames = flags + (((size_t)B * (d.KB + (d.KT > 1 ? 2 * d.NL : 0)) + 3) & ~(size_t)3);
        if (d.KT > 1) a.tk = a.frames + (size_t)B * T * d.NFFT;      // (PIPE: rings of P + KT - 1 slots per conv and stream)
        float* const h_serial = a.h;
        if (d.TA) a.h = a.frames + (size_t)B * T * d.NFFT;           // (PIPE: K / V rings of L + P slots per pair)
        a.pipe_p = P;
        h->impl->launch_pipe(a, st, &e);
        if (e != hipSuccess) {     // the runtime refused co-residency (GPU shared with other work): walk the frames serially
            (void)hipGetLastError();
            a.pipe_p = 0;
            a.pipe_flags = nullptr;
            a.frames = nullptr;
            a.tk = tk_serial;
            a.h = h_serial;
            h->impl->launch(a, h->max_wgs, st, &e);
            if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
            return FE_OK;
        }
        const int n_out = d.HOP * (T - 1);
        fe::note_kernel("istft_ola_kernel");
        hipLaunchKernelGGL(fe::istft_ola_kernel, dim3((n_out + fe::kThreads - 1) / fe::kThreads, B), dim3(fe::kThreads), 0, st,
                           a.frames, h->tables_dev, wav_hat_dev, (size_t)n_out, d.NFFT, d.HOP, T);
        e = hipGetLastError();
        if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
        return FE_OK;
    }
    h->impl->launch(a, h->max_wgs, st, &e);
    if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return FE_OK;
}

// ---------------------------------------------------------------------------- stand-alone STFT / iSTFT
static int ensure_tables(fe_handle* h, hipStream_t st) {
    if (h->tables_dev) return FE_OK;
    const size_t N = (size_t)h->cfg.n_fft;
    std::vector<float> t(3 * N);
    memcpy(&t[0], h->window.data(), N * sizeof(float));
    memcpy(&t[N], h->window_istft.data(), N * sizeof(float));
    memcpy(&t[2 * N], h->twiddle.data(), N * sizeof(float));
    FE_HIP_CHECK(hipMalloc(&h->tables_dev, 3 * N * sizeof(float)));
    FE_HIP_CHECK(hipMemcpyAsync(h->tables_dev, t.data(), 3 * N * sizeof(float), hipMemcpyHostToDevice, st));
    FE_HIP_CHECK(hipStreamSynchronize(st));
    return FE_OK;
}

#define FE_STFT_LAUNCH(kernel, a, grid, st)                                                         \
    do {                                                                                            \
        if (h->cfg.n_fft == 512) hipLaunchKernelGGL((fe::kernel<512>), grid, dim3(fe::kThreads), 0, st, a);        \
        else if (h->cfg.n_fft == 1024) hipLaunchKernelGGL((fe::kernel<1024>), grid, dim3(fe::kThreads), 0, st, a); \
        else return fail(FE_ERR_UNSUPPORTED_CONFIG, "n_fft=%d (stand-alone STFT kernels: 512, 1024)", h->cfg.n_fft); \
        hipError_t e_ = hipGetLastError();                                                          \
        if (e_ != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e_));  \
    } while (0)

static fe::StftArgs stft_args(const fe_handle* h, int B) {
    fe::StftArgs a{};
    a.tables = h->tables_dev;
    a.B = B;
    a.T = 1;
    a.H = h->cfg.hop_size;
    a.compression = 1.0f;
    a.eps = 1.0e-5f;
    return a;
}

// spec_hat rows of an offline call: N/2 (FastEnhancer: the model drops the Nyquist bin) or N/2 + 1 (BSRNN / FSPEN / LiSenNet)
static int offline_spec_rows(const fe_handle* h) { return h->d.NFFT / 2 + ((h->bimpl || h->fimpl || h->limpl) ? 1 : 0); }

// A ragged batch has ONE batched form: the time-batched engine (the frame walk and its time pipeline take one length per launch).  It is
// taken whenever the model has that engine and the caller has not asked for the frame walk - also for the big shapes at 8+ utterances,
// where AUTO would walk an equal-length batch: sixteen 4 s files of FastEnhancer_L are 20.6 ms in one time-batched pass, 16 x 5.2 ms one by one.
static bool ragged_uses_tb(const fe_handle* h) {
    return h->impl && h->impl->tb && (h->d.BD || h->offline_engine != FE_OFFLINE_FRAME_WALK);
}

// scratch of the call's engine: the batched pass, or the one-by-one fallback's largest single call (fe_offline_work_floats is monotone in B)
static size_t ragged_base_floats(const fe_handle* h, int B, int Tw_max) {
    if (ragged_uses_tb(h)) return tb_work_floats(h, B, 1 + Tw_max / h->d.HOP, nullptr);
    return std::max(fe_offline_work_floats(h, B, Tw_max), fe_offline_work_floats(h, 1, Tw_max));
}

size_t fe_offline_ragged_work_floats(const fe_handle* h, int B, int Tw_max) {
    if (!h || B <= 0 || Tw_max <= 0) return 0;
    const int Tmax = 1 + Tw_max / h->d.HOP;
    // the call's scratch, the per-utterance lengths, one utterance's spec_hat (the one-by-one fallback)
    return ragged_base_floats(h, B, Tw_max) + (((size_t)B + 3) & ~(size_t)3) + (size_t)offline_spec_rows(h) * Tmax * 2;
}

int fe_offline_ragged(fe_handle* h, const float* noisy_dev, size_t in_stride, const int* Tw_host, int B, float* wav_hat_dev, size_t out_stride,
                      float* spec_hat_dev, float* work_dev, void* stream) {
    int rc = check_ready(h);
    if (rc != FE_OK) return rc;
    KernelLogScope klog_(h);
    if (!noisy_dev || !Tw_host || !wav_hat_dev || !spec_hat_dev || !work_dev || B <= 0) return fail(FE_ERR_INVALID_ARG, "bad argument");
    const Dims& d = h->d;
    int Tw_max = 0;
    for (int b = 0; b < B; ++b) {
        if (Tw_host[b] <= d.NFFT / 2) return fail(FE_ERR_INVALID_ARG, "Tw[%d]=%d: reflect padding of n_fft/2=%d needs a longer input", b, Tw_host[b], d.NFFT / 2);
        if ((size_t)Tw_host[b] > in_stride && B > 1) return fail(FE_ERR_INVALID_ARG, "Tw[%d]=%d > in_stride %zu", b, Tw_host[b], in_stride);
        Tw_max = std::max(Tw_max, Tw_host[b]);
    }
    const int Tmax = 1 + Tw_max / d.HOP;
    if (out_stride < (size_t)d.HOP * (Tmax - 1) && B > 1) return fail(FE_ERR_INVALID_ARG, "out_stride %zu < H*(Tmax-1)", out_stride);
    hipStream_t st = (hipStream_t)stream;
    const size_t base = ragged_base_floats(h, B, Tw_max);
    if (ragged_uses_tb(h)) {
        // ONE batched call laid out for the longest utterance; every utterance's frames past its own end are computed on clamped input and
        // stay out of its output (causal in time; the noncausal model's reverse scans start at each utterance's own last frame)
        int* Tw_dev = reinterpret_cast<int*>(work_dev + base);
        FE_HIP_CHECK(hipMemcpyAsync(Tw_dev, Tw_host, (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
        return offline_tb(h, noisy_dev, in_stride, Tw_dev, Tw_max, B, wav_hat_dev, out_stride, spec_hat_dev, work_dev, st);
    }
    // no batched form for this model / engine setting (the frame walk and its time pipeline take ONE length per launch): one by one
    const int F = offline_spec_rows(h);
    float* spec_tmp = work_dev + base + (((size_t)B + 3) & ~(size_t)3);
    for (int b = 0; b < B; ++b) {
        const int Tb = 1 + Tw_host[b] / d.HOP;
        float* dst = spec_hat_dev + (size_t)b * F * Tmax * 2;
        rc = fe_offline(h, noisy_dev + (size_t)b * in_stride, 1, Tw_host[b], wav_hat_dev + (size_t)b * out_stride, Tb == Tmax ? dst : spec_tmp, work_dev, stream);
        if (rc != FE_OK) return rc;
        if (Tb != Tmax)
            FE_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)Tmax * 2 * sizeof(float), spec_tmp, (size_t)Tb * 2 * sizeof(float), (size_t)Tb * 2 * sizeof(float), (size_t)F,
                                          hipMemcpyDeviceToDevice, st));
    }
    return FE_OK;
}

int fe_stft_step(fe_handle* h, const float* wav_in_dev, size_t in_stride, const float* cache_in_dev, float* cache_out_dev,
                 float* spec_out_dev, int B, void* stream) {
    if (!h || !wav_in_dev || !cache_in_dev || !cache_out_dev || !spec_out_dev || B <= 0) return fail(FE_ERR_INVALID_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_tables(h, st);
    if (rc != FE_OK) return rc;
    fe::StftArgs a = stft_args(h, B);
    a.wav_in = wav_in_dev; a.in_stride = in_stride; a.cache_in = cache_in_dev; a.cache_out = cache_out_dev; a.spec = spec_out_dev;
    FE_STFT_LAUNCH(stft_step_kernel, a, dim3(B), st);
    return FE_OK;
}

int fe_istft_step(fe_handle* h, const float* spec_in_dev, const float* cache_in_dev, float* cache_out_dev, float* wav_out_dev,
                  size_t out_stride, int B, void* stream) {
    if (!h || !spec_in_dev || !cache_in_dev || !cache_out_dev || !wav_out_dev || B <= 0) return fail(FE_ERR_INVALID_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_tables(h, st);
    if (rc != FE_OK) return rc;
    fe::StftArgs a = stft_args(h, B);
    a.spec = const_cast<float*>(spec_in_dev); a.cache_in = cache_in_dev; a.cache_out = cache_out_dev;
    a.wav_out = wav_out_dev; a.out_stride = out_stride;
    FE_STFT_LAUNCH(istft_step_kernel, a, dim3(B), st);
    return FE_OK;
}

int fe_stft_offline(fe_handle* h, const float* noisy_dev, int B, int Tw, int F, int compress, float* spec_out_dev, void* stream) {
    if (!h || !noisy_dev || !spec_out_dev || B <= 0) return fail(FE_ERR_INVALID_ARG, "bad argument");
    const int N = h->cfg.n_fft, H = h->cfg.hop_size;
    if (Tw <= N / 2) return fail(FE_ERR_INVALID_ARG, "Tw=%d: reflect padding of n_fft/2=%d needs a longer input", Tw, N / 2);
    if (F != N / 2 && F != N / 2 + 1) return fail(FE_ERR_INVALID_ARG, "F=%d (n_fft/2 or n_fft/2+1)", F);
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_tables(h, st);
    if (rc != FE_OK) return rc;
    fe::StftArgs a = stft_args(h, B);
    a.T = 1 + Tw / H; a.Tw = Tw; a.F = F;
    a.wav_in = noisy_dev; a.in_stride = (size_t)Tw; a.spec = spec_out_dev;
    a.compression = compress ? h->cfg.input_compression : 1.0f;
    FE_STFT_LAUNCH(stft_frames_kernel, a, dim3(a.T, B), st);
    return FE_OK;
}

int fe_istft_offline(fe_handle* h, const float* spec_in_dev, int B, int T, int F, int compress, float* wav_out_dev,
                     float* frames_dev, void* stream) {
    if (!h || !spec_in_dev || !wav_out_dev || !frames_dev || B <= 0 || T <= 1) return fail(FE_ERR_INVALID_ARG, "bad argument");
    const int N = h->cfg.n_fft, H = h->cfg.hop_size;
    if (F != N / 2 && F != N / 2 + 1) return fail(FE_ERR_INVALID_ARG, "F=%d (n_fft/2 or n_fft/2+1)", F);
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_tables(h, st);
    if (rc != FE_OK) return rc;
    fe::StftArgs a = stft_args(h, B);
    a.T = T; a.F = F;
    a.spec = const_cast<float*>(spec_in_dev); a.frames = frames_dev;
    a.compression = compress ? h->cfg.input_compression : 1.0f;
    FE_STFT_LAUNCH(istft_frames_kernel, a, dim3(T, B), st);
    const int n_out = H * (T - 1);
    fe::note_kernel("istft_ola_kernel");
        hipLaunchKernelGGL(fe::istft_ola_kernel, dim3((n_out + fe::kThreads - 1) / fe::kThreads, B), dim3(fe::kThreads), 0, st,
                       frames_dev, h->tables_dev, wav_out_dev, (size_t)n_out, N, H, T);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return FE_OK;
}

double fe_flops_per_frame(const fe_handle* h) {
    if (!h) return 0.0;
    const Dims& d = h->d;
    if (h->limpl) {   // models/lisennet/macs.py:8-66 with T = 1
        const double C = 16, Nb = 2, F1 = 257;
        double m = 3 * (C / 4) * F1;
        const double co[3] = {C / 2, C / 4 * 3, C}, fi[3] = {257, 128, 64};
        for (int i = 0; i < 3; ++i) {
            const double f = fi[i], fq = std::floor(f / 4), fhi = std::floor((f - fq + 2 - 5) / 3) + 1;
            m += (2 * 3 * fq + 2 * 5 * fhi) * co[i] * co[i];
        }
        auto gru = [](double i, double hd) { return (i + hd) * hd * 3 + hd * 3; };
        const double hh = 24, ff = 32;
        for (int b = 0; b < (int)Nb; ++b) {
            m += (gru(C, hh / 2) * 2 + hh * C + gru(C, hh) + hh * C) * ff;
            m += (C * C * 4 + C * 2 * 3 + C * 2 + C * 2 * C) * ff;
        }
        double c_in = C, f = 32, c_out = 0;
        for (double c_o : {C / 4 * 3, C / 2, C / 4}) { c_out = c_o; m += (3 * (f / 2) + 3 * 3 * (f / 2)) * c_in * 2 * c_out; c_in = c_out; f *= 2; }
        f += 1;
        m += (c_out * 2 * 2 * 2 + 2 * 2 + 2 * 2) * f;
        return 2.0 * m + 2.0 * 2.5 * d.NFFT * std::log2((double)d.NFFT);
    }
    if (h->fimpl) {   // models/fspen/macs.py:36-141 with T = 1 (switches as committed: conv output lengths, no BN / LN / bias terms)
        const double C1[3] = {4, 16, 32}, K[3] = {6, 8, 6}, C2 = 16;
        double F = 257, m = 0;
        for (int i = 0; i < 3; ++i) { F = std::floor(F / 2); m += (i == 0 ? 2 : C1[i - 1]) * C1[i] * F * K[i]; }
        m += 32 * 32 * F + 32 * (4 * 8 + 7 * 6 + 11 * 6 + 20 * 6 + 40 * 6) + 32 * 64 * 32 + 32 * C2 * 32;
        const double gru = (C2 + C2) * C2 * 3 + C2 * 3;
        m += 3 * (gru * 2 + 2 * C2 * C2 + C2 + gru + C2 * C2 + C2) * 32;
        m += C2 * 32 * 32 + 32 * 32 * 64 + 32 * (8 * 2 + 6 * 3 + 8 * 5 + 8 * 10 + 8 * 20);
        for (int i = 2; i >= 0; --i) { m += C1[i] * (i == 0 ? 2 : C1[i - 1]) * F * K[i]; F = i == 0 ? F * 2 + 1 : F * 2; }
        m += 257 * 8;
        return 2.0 * m + 2.0 * 2.5 * d.NFFT * std::log2((double)d.NFFT);
    }
    if (h->bimpl) {   // models/bsrnn/macs.py:18-51
        const double C = h->cfg.channels, Hh = 2 * C, Lr = h->cfg.rf_blocks;
        double m = 0;
        for (int b = 0; b < 31; ++b) m += 2 * kSub[b] * C;
        m += (C * Hh * 4 + Hh * Hh * 4 + Hh * C + (C * Hh * 4 + Hh * Hh * 4) * 2 + 2 * Hh * C) * 31 * Lr;
        for (int b = 0; b < 31; ++b) m += (C * C * 4 + 4 * C * 4 * kSub[b]) * 2;
        return 2.0 * m + 2.0 * 2.5 * d.NFFT * std::log2((double)d.NFFT);
    }
    const double C1 = d.C1, C2 = d.C2, F1 = d.F1, F2 = d.F2, K = d.KB;
    const double KT = d.KT;
    double m = 2 * C1 * 8 * F1;
    for (int i = 1; i <= d.NL; ++i) m += C1 * C1 * 3 * KT * F1;
    m += F1 * F2 * C1 + C1 * C2 * F2;
    if (d.TA) m += K * (C2 * C2 * 3 * F2 + 2 * (d.TA + 1) * C2 * F2 + C2 * C2 * F2 + C2 * C2 * 3 * F2 + 2 * F2 * C2 * F2 + C2 * C2 * F2);
    else if (d.FR) {  // time GRU + fc, then the sub-band GRU (input and hidden products of both directions) + fc
        const double H = C2 / 2;
        m += K * (C2 * C2 * 6 * F2 + C2 * C2 * F2 + 2 * 3 * H * (C2 + H) * F2 + 2 * H * C2 * F2);
    } else if (d.BD)  // both GRU directions, rnn_fc over 2 C2
    m += K * (2 * C2 * C2 * 6 * F2 + 2 * C2 * C2 * F2 + C2 * C2 * 3 * F2 + 2 * F2 * C2 * F2 + C2 * C2 * F2);
    else
    m += K * (C2 * C2 * 6 * F2 + C2 * C2 * F2 + C2 * C2 * 3 * F2 + 2 * F2 * C2 * F2 + C2 * C2 * F2);
    m += F2 * F1 * C2 + C2 * C1 * F1;
    for (int i = 1; i <= d.NL; ++i) m += 2 * C1 * C1 * F1 + C1 * C1 * 3 * KT * F1;
    m += 2 * C1 * C1 * F1 + C1 * 2 * 8 * F1;
    return 2.0 * m + 2.0 * 2.5 * d.NFFT * std::log2((double)d.NFFT);
}

int fe_debug_stages(const fe_handle* h) { return !h ? 0 : h->limpl ? h->limpl->dbg_stages : h->fimpl ? h->fimpl->dbg_stages : (h->bimpl ? h->bimpl->dbg_stages : (h->impl ? h->impl->dbg_stages : 0)); }
size_t fe_debug_floats(const fe_handle* h) { return !h ? 0 : h->limpl ? h->limpl->dbg_floats : h->fimpl ? h->fimpl->dbg_floats : (h->bimpl ? h->bimpl->dbg_floats : (h->impl ? h->impl->dbg_floats : 0)); }

int fe_debug_stage(const fe_handle* h, int idx, const char** name, int* rows, int* cols, size_t* offset_floats) {
    if (!h || idx < 0 || idx >= fe_debug_stages(h)) return fail(FE_ERR_INVALID_ARG, "stage index %d", idx);
    static thread_local std::string nm;
    if (h->limpl) {
        static const char* const names[16] = {"spec_in", "compressed", "features", "encoder.conv_1", "encoder.conv_2", "encoder.conv_3", "encoder.conv_4",
                                              "blocks.0.intra", "blocks.0.inter", "blocks.0", "blocks.1.intra", "blocks.1.inter", "blocks.1",
                                              "decoder.up3", "mask", "spec_out"};
        int r, c; size_t off;
        h->limpl->dbg_stage(idx, &r, &c, &off);
        if (name) *name = names[idx];
        if (rows) *rows = r;
        if (cols) *cols = c;
        if (offset_floats) *offset_floats = off;
        return FE_OK;
    }
    if (h->fimpl) {
        static const char* const names[16] = {"spec_in", "compressed", "subband_encoder", "fullband_encoder.2", "feature_merge", "dpe.0.intra",
                                              "dpe.0.inter", "dpe.1.intra", "dpe.1.inter", "dpe.2.intra", "dpe.2.inter", "feature_split",
                                              "fullband_decoder.0", "fullband_decoder.1", "mask", "spec_out"};
        int r, c; size_t off;
        h->fimpl->dbg_stage(idx, &r, &c, &off);
        if (name) *name = names[idx];
        if (rows) *rows = r;
        if (cols) *cols = c;
        if (offset_floats) *offset_floats = off;
        return FE_OK;
    }
    if (h->bimpl) {   // spec_in, compressed, band_split, (layer.l.time, layer.l.freq)..., mask_mlp, spec_out
        const int L = h->cfg.rf_blocks;
        char bufn[64];
        if (idx == 0) nm = "spec_in";
        else if (idx == 1) nm = "compressed";
        else if (idx == 2) nm = "band_split";
        else if (idx < 3 + 2 * L) { snprintf(bufn, sizeof bufn, (idx - 3) % 2 == 0 ? "layer.%d.time" : "layer.%d.freq", (idx - 3) / 2); nm = bufn; }
        else if (idx == 3 + 2 * L) nm = "mask_mlp";
        else nm = "spec_out";
        int r, c; size_t off;
        h->bimpl->dbg_stage(idx, &r, &c, &off);
        if (name) *name = nm.c_str();
        if (rows) *rows = r;
        if (cols) *cols = c;
        if (offset_floats) *offset_floats = off;
        return FE_OK;
    }
    const Dims& d = h->d;
    char buf[64];
    int s = idx;
    if (s == 0) nm = "spec_in";
    else if (s == 1) nm = "compressed";
    else if (s == 2) nm = "enc_pre";
    else if (s < 3 + d.NL) { snprintf(buf, sizeof buf, "encoder.%d", s - 3); nm = buf; }
    else if (s == 3 + d.NL) nm = "rf_pre";
    else if (s < 4 + d.NL + 2 * d.KB) {
        int k = (s - 4 - d.NL) / 2, w = (s - 4 - d.NL) % 2;
        snprintf(buf, sizeof buf, w == 0 ? "rf_block.%d.rnn" : "rf_block.%d", k); nm = buf;
    } else if (s == 4 + d.NL + 2 * d.KB) nm = "rf_post";
    else if (s < 5 + 2 * d.NL + 2 * d.KB) { snprintf(buf, sizeof buf, "decoder.%d", s - 5 - d.NL - 2 * d.KB); nm = buf; }
    else if (s == 5 + 2 * d.NL + 2 * d.KB) nm = "mask";
    else nm = "spec_out";
    int r, c; size_t off;
    h->impl->dbg_stage(idx, &r, &c, &off);
    if (name) *name = nm.c_str();
    if (rows) *rows = r;
    if (cols) *cols = c;
    if (offset_floats) *offset_floats = off;
    return FE_OK;
}

}  // extern "C"
