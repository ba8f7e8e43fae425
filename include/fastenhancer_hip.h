/*
 * fastenhancer_hip.h — C ABI of libfastenhancer_hip.so (MI355X / gfx950).
 *
 * The reference (aask1357/fastenhancer) has no FFI: its inference boundary is the
 * Python call surface of
 *   - scripts/export_onnx.py:38-58      class Model: forward(wav_in, cache_stft, cache_istft, *cache_model)
 *   - models/fastenhancer/default/model.py:677-710   ONNXModel.forward(spec, *cache)   (spec -> spec)
 *   - models/fastenhancer/default/model.py:614-618   ONNXModel.initialize_cache
 *   - functional/audio_modules.py:238-303            ONNXSTFT.initialize_cache / forward / inverse
 *   - models/fastenhancer/default/model.py:532-608   remove_weight_reparameterizations (done on the host,
 *                                                    fastenhancer_amd/weights.py; this library takes FUSED weights)
 * This header is what a binding for that surface calls.  Plain C types only:
 * device pointers are raw `float*`, the stream is a `hipStream_t` passed as `void*`.
 *
 * Ownership: the caller allocates and frees every device buffer.  The handle owns only its packed
 * copy of the weights and constant tables.  All compute entry points are asynchronous on the given
 * stream and return FE_OK or a negative code; fe_last_error() gives the text (thread-local).
 * A handle is bound to the device that was current at fe_create(); it is not thread-safe, and its launches must be
 * stream-ordered: the handle owns per-workgroup scratch and frame counters, so two compute calls on the same handle may
 * not run concurrently on different HIP streams (use one handle per stream).
 */
#ifndef FASTENHANCER_HIP_H
#define FASTENHANCER_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FE_OK 0
#define FE_ERR_INVALID_ARG (-1)
#define FE_ERR_UNSUPPORTED_CONFIG (-2)
#define FE_ERR_HIP (-3)
#define FE_ERR_NO_WEIGHTS (-4)

#define FE_ARCH_FASTENHANCER 0 /* models/fastenhancer/default/model.py */
#define FE_ARCH_BSRNN 1        /* models/bsrnn/model.py: channels = num_channels, rf_blocks = num_layers */
#define FE_ARCH_FSPEN 2        /* models/fspen/model.py (configs/others/fspen.yaml): channels = channels[-1], kernel_size / n_kernels / stride
                                * as in the yaml, rf_channels / rf_freq / rf_blocks / rf_heads = dpe_kwargs channels / freq / num_blocks / groups */

#define FE_ARCH_LISENNET 3     /* models/lisennet/model.py (configs/others/lisennet.yaml): channels = num_channels, rf_blocks = n_blocks */

#define FE_MAX_KERNELS 8

/* Mirror of the yaml `model_kwargs` that select the architecture
 * (configs/fastenhancer/b.yaml:2-29; defaults models/fastenhancer/default/model.py:384-403).
 * Invariants of every shipped yaml are asserted by fe_create(): activation SiLU, mask null,
 * stride 4, kernel_size[0] 8 and 3 afterwards, window hann, stft_normalized False, weight
 * reparameterisations already removed (fused weights). */
typedef struct fe_config {
    int arch;                        /* FE_ARCH_* */
    int n_fft, hop_size, win_size;   /* N, H, win (win <= N, N even) */
    int channels;                    /* C1 */
    int n_kernels;                   /* len(kernel_size) */
    int kernel_size[FE_MAX_KERNELS]; /* [8,3,3] ... */
    int stride;                      /* 4 */
    int rf_channels, rf_freq, rf_blocks, rf_heads; /* C2, F2, K, NH */
    float input_compression;         /* 0.3 */
    int kernel_size_time;            /* `model: fastenhancer.time_kernel` (models/fastenhancer/time_kernel/model.py): time taps of
                                      * the causal k = 3 Conv2d layers (3 in configs/ablation/time_kernel_b.yaml); 0 or 1 = the
                                      * default model.  The state then also holds the convs' (kernel_size_time - 1)-frame input
                                      * caches: ... | K x h | 2 (n_kernels - 1) x [B, kernel_size_time - 1, F1, C1] (encoder layers, then
                                      * decoder layers; fe_spec_step's h_dev likewise: K x h, then the caches) */
    int channels_frnn;               /* `model: fastenhancer.dprnn` (models/fastenhancer/dprnn/model.py:135-247; configs/ablation/dprnn_*.yaml):
                                      * hidden units per direction of the bidirectional sub-band GRU that replaces the attention
                                      * (must be rf_channels / 2, as in every shipped yaml); 0 = the default RNNFormer block.
                                      * rf_heads is ignored.  Weight sections: rf_block.k.frnn.{weight,bias}_{ih,hh}_l0[_reverse],
                                      * rf_block.k.frnn_fc.{weight,bias} instead of attn.qkv / attn_fc, no pe; state as the default model */
    int lookbehind;                  /* `model: fastenhancer.dptransformer` (models/fastenhancer/dptransformer/model.py; configs/ablation/dpt_*.yaml):
                                      * frames of history of the causal time attention that replaces the time GRU (31 in every shipped
                                      * yaml, the only value compiled); 0 = the default block.  Weight sections: time_pe [NH, L+1] (the model's
                                      * `pe`), rf_block.k.time_attn.qkv.weight instead of rnn.*.  State (and fe_spec_step's h_dev): per
                                      * block the K cache then the V cache, each [B*F2, NH, L, C2/NH] (DPTBlock.initialize_cache, :194-198),
                                      * then B floats `head`: every cache is a ring over its L slots - the frame that is t steps old sits
                                      * in slot (head + L - t) mod L, a step overwrites slot `head` and advances it (one slot written per
                                      * frame instead of the reference's shift of all L).  head = 0 (fe_state_init, or caches copied in from
                                      * the reference) is exactly the reference's tensor, oldest frame first; to read the caches back in
                                      * that order rotate each [.., L, ..] axis left by head.  A cache slot whose first K element is +inf
                                      * does not take part (how a run "without caches", :216-218, marks the frames before the start; zero
                                      * caches do take part) */
    int ln;                          /* `model: fastenhancer.ln` (models/fastenhancer/ln/model.py; configs/ablation/ln_b.yaml): 1 = GroupNorm(1, C)
                                      * after every conv and the reference's LayerNorm over (F2, C2) after the blocks' fc layers instead of
                                      * (folded) BatchNorms.  Weight sections = that model's state_dict after remove_weight_reparameterizations
                                      * (convs with their own biases, <conv>.1 / .2 / .4 and rnn_post_norm / attn_post_norm weight + bias),
                                      * the final conv as dec_post.2 with its scale folded in.  State as the default model */
    float rf_eps;                    /* ln: eps of the blocks' LayerNorms (rnnformer_kwargs.eps; 0 -> 1e-5) */
    int bidirectional;               /* `model: fastenhancer.noncausal` (models/fastenhancer/noncausal/model.py:186-187; configs/fastenhancer_dns/huge_noncausal*.yaml,
                                      * configs/fastenhancer_48khz/huge_noncausal.yaml): 1 = the blocks' time GRU is bidirectional and rnn_fc maps
                                      * 2 C2 -> C2.  Weight sections: + rf_block.k.rnn.{weight,bias}_{ih,hh}_l0_reverse, rnn_fc.weight [C2, 2 C2].
                                      * The reference module has the offline `Model` only (:348, :628-635): fe_offline is the one compute entry
                                      * point (no caches: fe_state_floats = the two STFT caches, fe_step / fe_spec_step return FE_ERR_UNSUPPORTED_CONFIG) */
} fe_config;

typedef struct fe_handle fe_handle;

/* Model construction: ONNXModel.__init__ (model.py:383-521) + ONNXSTFT.__init__ window tables
 * (functional/audio_modules.py:182-236).  FE_ERR_UNSUPPORTED_CONFIG if no kernel was compiled
 * for this shape (shapes of all shipped fastenhancer yamls are compiled in). */
int fe_create(const fe_config* cfg, fe_handle** out);
void fe_destroy(fe_handle* h);

/* Fused-weight blob (what an RCCL broadcast carries).  Sections are the fused state_dict tensors
 * (SURVEY.md Appendix A.1 "Fused"), fp32, reference memory layout, concatenated in the order
 * reported by fe_weight_section(); each section offset is a multiple of 4 floats. */
size_t fe_weight_floats(const fe_handle* h);
int fe_weight_sections(const fe_handle* h);
int fe_weight_section(const fe_handle* h, int idx, const char** name, size_t* offset_floats, size_t* count_floats);
/* load_state_dict (wrappers/ns.py:308-314) for already-fused weights: blob_dev is a DEVICE pointer.
 * Synchronises the stream (load-time only); repacks into MFMA fragment order inside the handle. */
int fe_load_weights(fe_handle* h, const float* blob_dev, size_t nfloats, void* stream);

/* Per-stream state = the reference's cache list, concatenated:
 *   cache_stft  [B, N-H] | cache_istft [B, N-H] | K x h [1, B*F2, C2]
 * (scripts/export_onnx.py:43-46).  fe_state_init zeroes it (initialize_cache).  The model part (what fe_spec_step takes as
 * h_dev) per architecture, every tensor sized for B streams and laid out as the reference's:
 *   FE_ARCH_FASTENHANCER  K x h [1, B*F2, C2]  (kernel_size_time > 1: + the conv caches, see fe_config)
 *   FE_ARCH_BSRNN         2 * num_layers x [B*31, 2C]: h0, c0, h1, c1, ...              (models/bsrnn/model.py:409-416)
 *   FE_ARCH_FSPEN         num_blocks * groups x [1, B*freq/groups, C] inter-GRU states  (models/fspen/model.py:293-297)
 *   FE_ARCH_LISENNET      [B,1,257] phase | [B,4,1,257] [B,8,1,128] [B,12,1,64] encoder frames | n_blocks x ([1,B*32,24] GRU,
 *                         [B,32,2,32] ConvGLU frames) | [B,4,1,256] decoder frame     (models/lisennet/model.py:380-396)
 * FE_ARCH_BSRNN: fe_state_init also sizes the handle's scratch for the per-hop step of B streams (that step runs as three launches with
 * 10.6 KB per stream between them); a step of a larger batch than any fe_state_init has seen grows it on its first call (a device
 * allocation: not inside a stream capture).  FE_ARCH_FSPEN / FE_ARCH_LISENNET likewise from their "..._stream_batch_min" streams (LiSenNet, r6: three
 * launches with 49 KB per stream between them - the activation tensors of each sixteen-stream tile in the layout the matrix-core kernel reads). */
size_t fe_state_floats(const fe_handle* h, int B);
int fe_state_init(fe_handle* h, float* state_dev, int B, void* stream);

/* The wav->wav streaming step, scripts/export_onnx.py:48-58, for B independent streams and T
 * consecutive hops per stream (T=1: one hop, as scripts/test_onnx.py:44-50 drives it).
 *   wav_in [b*in_stride + t*H + n], wav_out [b*out_stride + t*H + n],  n < H, t < T, b < B.
 * The state is updated in place. */
int fe_step(fe_handle* h, const float* wav_in_dev, size_t in_stride, float* state_dev,
            float* wav_out_dev, size_t out_stride, int B, int T, void* stream);

/* The same step for callers whose audio lives in HOST memory (the reference's scripts/test_onnx.py feeds numpy arrays hop by hop):
 * n_calls consecutive fe_step calls of T hops each, hop block c = hops c*T .. c*T+T-1 of
 *   wav_in_host [b*in_stride + t*H + n], wav_out_host [b*out_stride + t*H + n]   (page-locked for asynchronous copies; pageable works, slower)
 * The copy-in of block c + 1 and the copy-out of block c - 1 run under the kernel of block c: two copy streams of the handle next
 * to `stream`, double-buffered device staging in work_dev (4*B*T*H floats).  Asynchronous: `stream` is complete when the last block
 * is back in host memory.  (bench.py's `value` is the device-resident rate; this is the PCIe-inclusive one - DESIGN.md 1.) */
int fe_step_host(fe_handle* h, const float* wav_in_host, size_t in_stride, float* state_dev, float* wav_out_host,
                 size_t out_stride, int B, int T, int n_calls, float* work_dev, void* stream);

/* The spec->spec step, ONNXModel.forward (model.py:677-710): spec [B, N/2+1, T, 2] in and out,
 * h_dev = the K GRU caches [K][B*F2][C2] (updated in place).  Any T >= 1. */
int fe_spec_step(fe_handle* h, const float* spec_in_dev, float* h_dev, float* spec_out_dev,
                 int B, int T, void* stream);

/* Time pipelining of fe_offline / fe_spec_step (T >= 4 frames per stream, at most half as many streams as CUs): the
 * frames of a stream are spread over up to `frames_in_flight` co-resident workgroups that hand the GRU state of each
 * RNNFormer block from frame to frame through global memory (everything else in a frame is independent of the other
 * frames).  Negative (the default) = a width chosen from the model size (8 .. 64); 0 or 1 = off (one workgroup walks the
 * T frames of a stream); values above 64 are clamped to 64 (the per-frame rings in work_dev are sized for that).  Results agree to fp32 rounding.  BSRNN's fe_offline is pipelined the same way (the time-LSTM (h, c) of
 * each layer is the hand-off; up to 64 frames in flight), and so are the ln variant's, the time_kernel variant's and the
 * dptransformer variant's (the time convs' input frames / the K-V caches go through per-frame rings: in work_dev for fe_offline;
 * for fe_spec_step - ONNXModel.forward(spec, *caches) with T >> 1, time_kernel/model.py:746-806, dptransformer/model.py:194-236 - in a
 * grow-only buffer of the handle, filled from the caller's caches before the launch and read back after it: a first or wider call
 * allocates, so not inside a stream capture), FSPEN's (its inter-GRU states per DPE block) and LiSenNet's (its nine caches through
 * a ring of per-frame slots). */
int fe_set_time_pipeline(fe_handle* h, int frames_in_flight);

/* Engine of fe_offline for the default and noncausal FastEnhancer models:
 *   FE_OFFLINE_TIME_BATCHED (the default where compiled; the only engine of the noncausal model): the network is cut at the
 *     blocks' time GRUs and every piece runs over ALL frames of ALL utterances, layer by layer, as the reference's own offline
 *     forward does (model.py:620-675): encoder pass (tiles of frames as one GEMM per layer), per block a scan over time in
 *     which only W_hh h is serial (the x half of the gates is batched) + a batched attention pass, decoder pass, overlap-add
 *     (csrc/tb_kernels.hip.h).  Activations that cross a GRU live in work_dev.
 *     With few utterances the scan runs four rows per workgroup on v_mfma_f32_4x4x1 instead of sixteen on 16x16x4.
 *   FE_OFFLINE_FRAME_WALK: the per-hop kernel walking (or, fe_set_time_pipeline, pipelining) the frames of each utterance.
 *   FE_OFFLINE_AUTO: time-batched, except for the big shapes (rf_channels >= 72) with 8 or more utterances, where the pipelined
 *     walk of the per-hop kernel is the faster one.
 * fe_spec_step follows the same setting: FE_OFFLINE_TIME_BATCHED runs every chunk of T >= 2 frames on the time-batched engine (the
 * scans start from the caller's GRU caches and leave the new ones), AUTO the chunks of 16+ frames of batches too large for the time
 * pipeline (or of 2048+ frames in all); the handle then keeps a grow-only work buffer (a first / larger call allocates).
 * Results agree to fp32 rounding.  The other architectures / variants always walk.  A handle's compute calls must be stream-ordered
 * (one stream, or event-ordered streams): the engines keep per-handle scratch, counters and helper streams. */
/* Which kernel runs the per-hop streaming step (fe_step / fe_step_host with T = 1) of the default FastEnhancer model:
 *   FE_STEP_KERNEL_WAVES4: one 256-thread workgroup per stream (fe_kernels.hip.h), the kernel every other entry point uses too;
 *   FE_STEP_KERNEL_WG8 (default): where built for the shape (FastEnhancer_B), one 512-thread workgroup per stream - two waves per
 *     SIMD, GRU gates grouped by channel (fe_frame8.hip.h) - for batches of up to one stream per CU;
 *   FE_STEP_KERNEL_WG8_PERSIST: the same also above that (persistent workgroups instead of the low-LDS companion kernel).
 * The two kernels agree to fp32 rounding (a few 1e-8 on the waveform), not bit for bit: a caller that needs a chunked launch
 * (T > 1) to be bit-identical to T per-hop launches selects FE_STEP_KERNEL_WAVES4.  The environment variable FE_WG8 = 0 | 1 | 2
 * sets the value new handles start with (any other value is ignored).  The reference has one forward only (models/fastenhancer/default/model.py:677-710).
 * BSRNN with num_channels = 16 (r5): FE_STEP_KERNEL_WAVES4 runs the layers of the per-hop step phase by phase on all four waves
 * (bsrnn_frame_kernel<PART 1>), any other value the role-split kernel (bsrnn_ov_kernels.hip.h) for batches of up to one stream per CU;
 * the two agree to fp32 rounding.  (models/bsrnn/model.py:367-390) */
#define FE_STEP_KERNEL_WAVES4 0
#define FE_STEP_KERNEL_WG8 1
#define FE_STEP_KERNEL_WG8_PERSIST 2
int fe_set_step_kernel(fe_handle* h, int kernel);

/* The other kernel-selection switches of a handle, by name (r6; until r5 these were environment variables read in three files).
 * FE_ERR_INVALID_ARG for an unknown name or a value outside the option's range; fe_options() / fe_option_name(i) enumerate them.
 *   name                       range      default  meaning
 *   "bsrnn_role_split"         0 | 1      1        BSRNN, num_channels 16, up to one stream per CU: PART 1 of the per-hop step on the role-split
 *                                                  kernel (bsrnn_ov_kernels.hip.h); 0 = the phase-by-phase kernel (what fe_set_step_kernel(WAVES4) selects too)
 *   "bsrnn_stream_batch_min"   0 .. 2^24  2048     BSRNN, num_channels 16 and (r6) 32: from this many streams the layers run batched over the streams
 *                                                  (bsrnn_sb_kernels.hip.h); 0 = never
 *   "bsrnn_three_launch_step"  0 | 1      1        BSRNN per-hop step as PART 1 -> batched mask decoder -> PART 2; 0 = one fused kernel per stream
 *   "bsrnn_ov_profile"         0 | 1      0        fe_profile_step probes the role-split PART 1 instead of the fused kernel's phases
 *   "fspen_stream_batch_min"   0 .. 2^24  1536     FSPEN: from this many streams the middle of the network runs batched over the streams; 0 = never
 *   "low_lds_companion"        0 | 1      1        FastEnhancer per-hop step above the streams the shape's own LDS plan holds at once (one per CU; two for
 *                                                  FastEnhancer_T): the low-LDS companion kernel (two workgroups per CU, three for the T shapes) where one
 *                                                  is compiled; 0 = persistent workgroups of the shape's own kernel
 *   "bsrnn_fused_step"         0 | 1      0        BSRNN, num_channels 16, up to one stream per CU (r6): the whole per-hop step in ONE cooperative launch -
 *                                                  the workgroups of a sixteen-stream tile meet at a barrier after the layers, run the mask decoder for
 *                                                  their tile, meet again and finish their own streams; 0 = three launches (a launch the runtime
 *                                                  refuses falls back to them by itself).  Bit-identical results; MEASURED SLOWER (barriers across XCDs,
 *                                                  cooperative launch: profiles/r6_bsrnn_fused_step.txt), hence off by default
 *   "lisennet_stream_batch_min" 0 .. 2^24 513      LiSenNet (r6): from this many streams the per-hop step runs encoder.conv_1 .. the mask head batched over the streams on
 *                                                  the matrix cores (lisennet_sb_kernels.hip.h: STFT + features per stream, sixteen streams per workgroup, mask + iSTFT per stream); 0 = never
 * A/B scripts (tools/ab_*.sh) preset the values NEW handles start with through FE_BSRNN_OV, FE_BSRNN_SB, FE_BSRNN_SPLIT, FE_BSRNN_OV_PROF,
 * FE_FSPEN_SB, FE_LOWLDS (FE_NO_LOWLDS), FE_BSRNN_FUSED, FE_LISENNET_SB and FE_WG8 (the step kernel): read once, in fe_create, validated against the same ranges (anything else
 * is ignored).  The reference has one forward per model and nothing to select (models/fastenhancer/default/model.py:677-710). */
int fe_set_option(fe_handle* h, const char* name, int value);
int fe_get_option(const fe_handle* h, const char* name, int* value);
int fe_options(void);
const char* fe_option_name(int idx);

/* What the handle's last compute call (fe_step / fe_step_host / fe_spec_step / fe_offline / fe_offline_ragged / fe_debug_step / fe_profile_step)
 * enqueued: the kernel family and instantiation of every launch in order, as the launchers name them, " + "-joined, repeats as "n x", then the
 * compiled shape record - e.g. "fe_frame8_kernel [shape B]", "fe_frame_kernel<LOW=2, per-hop> [shape B48H480LOW]",
 * "bsrnn_ov_kernel + bsrnn_mlp_kernel<one 16-stream tile per workgroup> + bsrnn_frame_kernel<PART 2> [shape xt]".  These are the names
 * rocprofv3 --kernel-trace shows (bench.py's roofline.kernel is this string).  "" before the first call.  The pointer is valid until the
 * next call of this function on the handle. */
const char* fe_last_step_kernel(const fe_handle* h);

#define FE_OFFLINE_AUTO 0
#define FE_OFFLINE_FRAME_WALK 1
#define FE_OFFLINE_TIME_BATCHED 2
int fe_set_offline_engine(fe_handle* h, int engine);

/* Offline wav->wav, Model.forward (model.py:728-735) with CompressedSTFT
 * (functional/audio_modules.py:70-164): noisy [B, Tw] -> wav_hat [B, H*(Tw/H)], spec_hat [B, N/2, T, 2],
 * T = 1 + Tw/H.  work_dev: scratch of fe_offline_work_floats(B, Tw) floats.
 * The size is that of the engine the CURRENT fe_set_offline_engine / fe_set_time_pipeline setting selects and is monotone in B and Tw
 * under that setting: a buffer sized once for the largest batch serves every smaller one (AUTO on the big shapes covers both the frame
 * walk of 8+ utterances and the time-batched pass of fewer).  After fe_set_offline_engine the size must be queried again - the
 * time-batched engine of FastEnhancer_L needs ~100x the frame walk's scratch, and fe_offline cannot see the size of work_dev. */
size_t fe_offline_work_floats(const fe_handle* h, int B, int Tw);
int fe_offline(fe_handle* h, const float* noisy_dev, int B, int Tw, float* wav_hat_dev,
               float* spec_hat_dev, float* work_dev, void* stream);

/* The same for a batch of utterances of DIFFERENT lengths - what the reference's offline caller does file by file
 * (scripts/test_pytorch.py:28-37: one Model.forward per file of the directory): utterance b = noisy_dev[b * in_stride + n],
 * n < Tw_host[b] (a HOST array, read before the call returns); wav_hat_dev[b * out_stride + n], n < H * (Tw[b] / H), the rest of a row is
 * left untouched; spec_hat_dev [B, F, Tmax, 2] with Tmax = 1 + max(Tw) / H (F = N/2 for FastEnhancer), the frames of an utterance past
 * its own 1 + Tw[b] / H unspecified.  On the time-batched engine (the default FastEnhancer model and the noncausal one) this is ONE
 * batched pass laid out for the longest utterance - every utterance's result is bit-identical to its own fe_offline call on that engine
 * (frames are rows of the layer GEMMs, scans are per row; the reflect padding, the reverse scans of the noncausal model and the
 * overlap-add use each utterance's own length); sort a directory by length to keep the padding small.  A ragged batch takes that pass
 * under FE_OFFLINE_AUTO too, whatever the shape and B (the frame walk AUTO prefers for 8+ equal-length utterances of the big shapes has
 * no ragged form).  The other models / variants and FE_OFFLINE_FRAME_WALK have no batched form (one length per launch): the call runs
 * them one utterance after the other, each on the engine fe_offline would take for a single utterance.
 * work_dev: fe_offline_ragged_work_floats(B, max(Tw)) floats (same contract as fe_offline_work_floats: query again after
 * fe_set_offline_engine). */
size_t fe_offline_ragged_work_floats(const fe_handle* h, int B, int Tw_max);
int fe_offline_ragged(fe_handle* h, const float* noisy_dev, size_t in_stride, const int* Tw_host, int B, float* wav_hat_dev, size_t out_stride,
                      float* spec_hat_dev, float* work_dev, void* stream);

/* The STFT front / back ends as launches of their own - the modules the reference exposes as `model.stft` and that
 * scripts/export_onnx.py:55-57 composes line by line (inside fe_step / fe_offline they are fused into the frame kernel).
 * No weights needed.  Streaming, one hop (ONNXSTFT.forward / .inverse, functional/audio_modules.py:243-303):
 *   fe_stft_step : wav_in [b*in_stride + n] (n < H), cache_in [B, N-H] -> spec [B, N/2+1, 1, 2], cache_out [B, N-H]
 *   fe_istft_step: spec [B, N/2+1, 1, 2], cache_in [B, N-H] -> wav_out [b*out_stride + n] (n < H), cache_out [B, N-H]
 * Functional like the reference: cache_in is not modified unless cache_out aliases it. */
int fe_stft_step(fe_handle* h, const float* wav_in_dev, size_t in_stride, const float* cache_in_dev, float* cache_out_dev,
                 float* spec_out_dev, int B, void* stream);
int fe_istft_step(fe_handle* h, const float* spec_in_dev, const float* cache_in_dev, float* cache_out_dev,
                  float* wav_out_dev, size_t out_stride, int B, void* stream);
/* Offline, centered (CompressedSTFT.forward / .inverse, functional/audio_modules.py:124-164 over STFT :70-119), one
 * workgroup per (stream, frame):
 *   fe_stft_offline : noisy [B, Tw] -> spec [B, F, T, 2], T = 1 + Tw/H; F = N/2 (discard_last_freq_bin) or N/2+1;
 *                     compress != 0: X *= max(|X|, 1e-5)^(c-1) with c = input_compression
 *   fe_istft_offline: spec [B, F, T, 2] (complex) -> wav [B, H*(T-1)]; compress != 0: X *= |X|^(1/c-1) first;
 *                     frames_dev: scratch of B*T*N floats */
int fe_stft_offline(fe_handle* h, const float* noisy_dev, int B, int Tw, int F, int compress, float* spec_out_dev, void* stream);
int fe_istft_offline(fe_handle* h, const float* spec_in_dev, int B, int T, int F, int compress, float* wav_out_dev,
                     float* frames_dev, void* stream);

/* Analytic FLOPs of one frame (2*MACs of models/fastenhancer/default/macs.py:17-87 + FFTs). */
double fe_flops_per_frame(const fe_handle* h);

/* Debug: run ONE hop (T=1) and dump the named per-stage activations of every stream to dbg_dev.
 * fe_debug_stages() reports name / rows / cols / offset of each dump ([rows][cols] row-major per
 * stream, stream-major). Used by the parity tests to localise a mismatch. */
int fe_debug_stages(const fe_handle* h);
int fe_debug_stage(const fe_handle* h, int idx, const char** name, int* rows, int* cols, size_t* offset_floats);
size_t fe_debug_floats(const fe_handle* h);   /* per stream */
int fe_debug_step(fe_handle* h, const float* wav_in_dev, size_t in_stride, float* state_dev,
                  float* wav_out_dev, size_t out_stride, int B, float* dbg_dev, void* stream);

/* Profiling: like fe_step, and thread 0 of workgroup 0 stores the shader cycle counter (s_memtime) at
 * the phase boundaries of the LAST frame into clk_dev[0..63] (see fe_kernels.hip.h, FE_CLK; BSRNN:
 * bsrnn_kernels.hip.h, BE_CLK).  Runs the debug / profile instantiation of the kernel, not the one fe_step runs;
 * tools/gpu_phases.py and tools/gpu_phases_bsrnn.py print the table. */
int fe_profile_step(fe_handle* h, const float* wav_in_dev, size_t in_stride, float* state_dev,
                    float* wav_out_dev, size_t out_stride, int B, int T, unsigned long long* clk_dev, void* stream);

/* Test support (r5): fills the LDS of every CU of the current device with NaN bit patterns (a short launch on `stream`).  The kernels
 * read padded operands in places - zero weights against words past the end of a tensor in LDS - and whatever an earlier kernel left
 * there must not matter: the parity tests run a step after this call and require the bits of the un-poisoned run.
 * (The reference has no counterpart: PyTorch pads explicitly, models/bsrnn/model.py:136-153.) */
int fe_debug_poison_lds(void* stream);

const char* fe_last_error(void);
const char* fe_version(void);
/* r6: the digest of the sources this library was built from (fastenhancer_amd/build.py::source_key: csrc/* and this header).  The Python binding refuses an
 * in-tree library whose key is not the tree's, and __graft_entry__.build() checks it after building: a stale object cache cannot pass for the shipped sources. */
const char* fe_build_key(void);

#ifdef __cplusplus
}
#endif
#endif /* FASTENHANCER_HIP_H */
