"""Error behaviour of the C ABI (include/fastenhancer_hip.h), called through ctypes directly - no Python mirror in between.

The reference signals misuse with Python `assert`s / exceptions (models/fastenhancer/default/model.py:383-521, functional/audio_modules.py:
182-236); the C boundary returns FE_ERR_* and leaves the text in fe_last_error().  Every class of `return fail(...)` branch of
csrc/fe_api.hip is hit here: null handle / pointers, B <= 0, T <= 0, short / long blobs, out-of-range fe_set_*, calls before
fe_load_weights, strides shorter than a row, the streaming entry points on a noncausal handle, zero-length utterances.  The argument
checks that come before any device work run on CPU (`-m "not gpu"`); the ones behind check_ready() need loaded weights (`-m gpu`)."""
import ctypes
from ctypes import byref, c_char_p, c_int, c_size_t, c_void_p

import pytest
import torch

from common import MODEL_KWARGS, BSRNN_KWARGS, product_config
from fastenhancer_amd import _lib
from fastenhancer_amd.engine import Engine

FE_ERR_INVALID_ARG, FE_ERR_UNSUPPORTED_CONFIG, FE_ERR_HIP, FE_ERR_NO_WEIGHTS = -1, -2, -3, -4
P = c_void_p(0x1000)         # a non-null pointer that is never dereferenced: every call below fails before touching memory
NULL = c_void_p(0)


def _err():
    return _lib.load().fe_last_error().decode()


def _expect(rc, code, text):
    assert rc == code, (rc, _err())
    assert text in _err(), _err()


def _cfg(name="fe_b"):
    """fe_config of a shipped yaml, filled the way Engine fills it"""
    cfg = product_config(name)
    c = _lib.fe_config()
    c.arch = _lib.FE_ARCH_FASTENHANCER
    c.n_fft, c.hop_size, c.win_size = cfg.n_fft, cfg.hop_size, cfg.win_size
    c.input_compression = cfg.input_compression
    c.channels, c.n_kernels, c.stride = cfg.channels, len(cfg.kernel_size), cfg.stride
    for i, k in enumerate(cfg.kernel_size):
        c.kernel_size[i] = k
    c.rf_channels, c.rf_freq, c.rf_blocks, c.rf_heads = cfg.rf_channels, cfg.rf_freq, cfg.rf_blocks, cfg.rf_heads
    c.kernel_size_time, c.channels_frnn, c.lookbehind = cfg.kernel_size_time, cfg.channels_frnn, cfg.lookbehind
    c.ln, c.rf_eps = (1 if cfg.ln else 0), cfg.rf_eps
    c.bidirectional = 1 if getattr(cfg, "noncausal", False) else 0
    return c


def _create(c):
    lib = _lib.load()
    h = c_void_p()
    rc = lib.fe_create(byref(c), byref(h))
    return rc, h


# ------------------------------------------------------------------------------------------------ CPU: checks that precede device work
def test_fe_create_rejects_what_the_reference_constructor_rejects():
    lib = _lib.load()
    h = c_void_p()
    _expect(lib.fe_create(None, byref(h)), FE_ERR_INVALID_ARG, "null argument")
    _expect(lib.fe_create(byref(_cfg()), None), FE_ERR_INVALID_ARG, "null argument")
    cases = [
        (dict(arch=17), FE_ERR_UNSUPPORTED_CONFIG, "arch 17"),
        (dict(n_fft=511), FE_ERR_INVALID_ARG, "`n_fft` must be an even number"),              # functional/audio_modules.py:196
        (dict(win_size=1024), FE_ERR_INVALID_ARG, "must be bigger than win_size"),              # functional/audio_modules.py:194-195
        (dict(hop_size=0), FE_ERR_INVALID_ARG, "hop_size 0 out of range"),
        (dict(hop_size=513), FE_ERR_INVALID_ARG, "hop_size 513 out of range"),
        (dict(stride=2), FE_ERR_UNSUPPORTED_CONFIG, "stride 2"),
        (dict(n_kernels=1), FE_ERR_INVALID_ARG, "len(kernel_size)=1"),
        (dict(n_kernels=9), FE_ERR_INVALID_ARG, "len(kernel_size)=9"),
        (dict(rf_heads=3), FE_ERR_UNSUPPORTED_CONFIG, "num_heads=3"),
        (dict(input_compression=0.0), FE_ERR_INVALID_ARG, "input_compression"),
        (dict(input_compression=1.5), FE_ERR_INVALID_ARG, "input_compression"),
        (dict(ln=1, lookbehind=31), FE_ERR_INVALID_ARG, "ln excludes"),
        (dict(bidirectional=1, kernel_size_time=3), FE_ERR_INVALID_ARG, "bidirectional excludes"),
        (dict(lookbehind=30), FE_ERR_UNSUPPORTED_CONFIG, "lookbehind=30"),
        (dict(lookbehind=31, channels_frnn=18), FE_ERR_INVALID_ARG, "exclusive"),
        (dict(channels_frnn=7), FE_ERR_UNSUPPORTED_CONFIG, "channels_frnn=7"),
        (dict(channels=52), FE_ERR_UNSUPPORTED_CONFIG, ""),                                    # no kernel compiled for the shape: a message that names it
    ]
    for over, code, text in cases:
        c = _cfg()
        for k, v in over.items():
            setattr(c, k, v)
        rc, h = _create(c)
        _expect(rc, code, text)
        assert not h.value, over
    c = _cfg()
    c.kernel_size[0] = 6
    _expect(_create(c)[0], FE_ERR_UNSUPPORTED_CONFIG, "kernel_size[0]=6")
    c = _cfg()
    c.kernel_size[2] = 5
    _expect(_create(c)[0], FE_ERR_UNSUPPORTED_CONFIG, "kernel_size[2]=5")
    # the other architectures' own checks
    c = _lib.fe_config()
    c.arch, c.n_fft, c.hop_size, c.win_size, c.channels, c.rf_blocks, c.input_compression = _lib.FE_ARCH_BSRNN, 1024, 256, 1024, 16, 6, 0.3
    _expect(_create(c)[0], FE_ERR_INVALID_ARG, "Only n_fft=512 is supported")                  # models/bsrnn/model.py asserts the band table's n_fft
    c.n_fft, c.win_size, c.channels = 512, 512, 24
    _expect(_create(c)[0], FE_ERR_UNSUPPORTED_CONFIG, "no BSRNN kernel compiled for num_channels=24")
    c = _lib.fe_config()
    c.arch, c.n_fft, c.hop_size, c.win_size, c.channels, c.rf_blocks, c.input_compression = _lib.FE_ARCH_LISENNET, 512, 256, 512, 20, 2, 0.3
    _expect(_create(c)[0], FE_ERR_UNSUPPORTED_CONFIG, "no LiSenNet kernel compiled for num_channels=20")


def test_null_handle_is_an_error_not_a_crash():
    lib = _lib.load()
    assert lib.fe_weight_floats(NULL) == 0 and lib.fe_weight_sections(NULL) == 0 and lib.fe_state_floats(NULL, 4) == 0
    assert lib.fe_offline_work_floats(NULL, 1, 16000) == 0 and lib.fe_offline_ragged_work_floats(NULL, 1, 16000) == 0
    assert lib.fe_debug_stages(NULL) == 0 and lib.fe_debug_floats(NULL) == 0 and lib.fe_flops_per_frame(NULL) == 0.0
    lib.fe_destroy(NULL)
    name, off, cnt = c_char_p(), c_size_t(), c_size_t()
    _expect(lib.fe_weight_section(NULL, 0, byref(name), byref(off), byref(cnt)), FE_ERR_INVALID_ARG, "section index")
    _expect(lib.fe_load_weights(NULL, P, 10, NULL), FE_ERR_INVALID_ARG, "null argument")
    _expect(lib.fe_state_init(NULL, P, 1, NULL), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_step(NULL, P, 256, P, P, 256, 1, 1, NULL), FE_ERR_INVALID_ARG, "null handle")
    _expect(lib.fe_step_host(NULL, P, 256, P, P, 256, 1, 1, 1, P, NULL), FE_ERR_INVALID_ARG, "null handle")
    _expect(lib.fe_spec_step(NULL, P, P, P, 1, 1, NULL), FE_ERR_INVALID_ARG, "null handle")
    _expect(lib.fe_offline(NULL, P, 1, 16000, P, P, P, NULL), FE_ERR_INVALID_ARG, "null handle")
    tw = (c_int * 1)(16000)
    _expect(lib.fe_offline_ragged(NULL, P, 16000, tw, 1, P, 16000, P, P, NULL), FE_ERR_INVALID_ARG, "null handle")
    _expect(lib.fe_set_time_pipeline(NULL, 4), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_set_step_kernel(NULL, 0), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_set_offline_engine(NULL, 0), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_stft_step(NULL, P, 256, P, P, P, 1, NULL), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_istft_step(NULL, P, P, P, P, 256, 1, NULL), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_stft_offline(NULL, P, 1, 16000, 256, 1, P, NULL), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_istft_offline(NULL, P, 1, 10, 256, 1, P, P, NULL), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_debug_step(NULL, P, 256, P, P, 256, 1, P, NULL), FE_ERR_INVALID_ARG, "null handle")


def test_argument_checks_before_the_weights_are_loaded():
    lib = _lib.load()
    rc, h = _create(_cfg())
    assert rc == 0 and h.value
    try:
        n = lib.fe_weight_floats(h)
        name, off, cnt = c_char_p(), c_size_t(), c_size_t()
        for idx in (-1, lib.fe_weight_sections(h)):
            _expect(lib.fe_weight_section(h, idx, byref(name), byref(off), byref(cnt)), FE_ERR_INVALID_ARG, f"section index {idx}")
        # short and long blobs, null blob: rejected before any device work
        _expect(lib.fe_load_weights(h, P, n - 1, NULL), FE_ERR_INVALID_ARG, f"blob has {n - 1} floats, expected {n}")
        _expect(lib.fe_load_weights(h, P, n + 1, NULL), FE_ERR_INVALID_ARG, f"blob has {n + 1} floats, expected {n}")
        _expect(lib.fe_load_weights(h, NULL, n, NULL), FE_ERR_INVALID_ARG, "null argument")
        # setters: range
        for k in (-1, 3):
            _expect(lib.fe_set_step_kernel(h, k), FE_ERR_INVALID_ARG, "bad argument")
            _expect(lib.fe_set_offline_engine(h, k), FE_ERR_INVALID_ARG, "bad argument")
        for k in (0, 1, 2):
            assert lib.fe_set_step_kernel(h, k) == 0 and lib.fe_set_offline_engine(h, k) == 0
        assert lib.fe_set_offline_engine(h, 0) == 0 and lib.fe_set_step_kernel(h, 1) == 0
        for k in (-1, 0, 1, 7, 64, 1000):          # every width is accepted: negative = automatic, wide ones are clamped
            assert lib.fe_set_time_pipeline(h, k) == 0
        assert lib.fe_set_time_pipeline(h, -1) == 0
        # sizes of nothing
        assert lib.fe_state_floats(h, 0) == 0 and lib.fe_state_floats(h, -3) == 0
        assert lib.fe_offline_work_floats(h, 0, 16000) == 0 and lib.fe_offline_work_floats(h, 2, 0) == 0
        assert lib.fe_offline_ragged_work_floats(h, 0, 16000) == 0 and lib.fe_offline_ragged_work_floats(h, 2, -1) == 0
        # fe_offline_work_floats is monotone in B (a buffer sized once for the largest batch serves the smaller ones)
        sizes = [lib.fe_offline_work_floats(h, b, 64000) for b in range(1, 20)]
        assert sizes == sorted(sizes) and sizes[0] > 0
        # state / step: bad arguments first, then "no weights"
        _expect(lib.fe_state_init(h, NULL, 4, NULL), FE_ERR_INVALID_ARG, "bad argument")
        _expect(lib.fe_state_init(h, P, 0, NULL), FE_ERR_INVALID_ARG, "bad argument")
        _expect(lib.fe_step(h, P, 256, P, P, 256, 4, 1, NULL), FE_ERR_NO_WEIGHTS, "fe_load_weights has not been called")
        _expect(lib.fe_spec_step(h, P, P, P, 4, 1, NULL), FE_ERR_NO_WEIGHTS, "fe_load_weights has not been called")
        _expect(lib.fe_offline(h, P, 1, 16000, P, P, P, NULL), FE_ERR_NO_WEIGHTS, "fe_load_weights has not been called")
        tw = (c_int * 2)(16000, 8000)
        _expect(lib.fe_offline_ragged(h, P, 16000, tw, 2, P, 16000, P, P, NULL), FE_ERR_NO_WEIGHTS, "fe_load_weights has not been called")
        _expect(lib.fe_debug_step(h, P, 256, P, P, 256, 1, P, NULL), FE_ERR_NO_WEIGHTS, "fe_load_weights has not been called")
        # the stand-alone STFT entry points need no weights: their own argument checks
        _expect(lib.fe_stft_step(h, NULL, 256, P, P, P, 1, NULL), FE_ERR_INVALID_ARG, "bad argument")
        _expect(lib.fe_stft_step(h, P, 256, P, P, P, 0, NULL), FE_ERR_INVALID_ARG, "bad argument")
        _expect(lib.fe_istft_step(h, P, P, NULL, P, 256, 1, NULL), FE_ERR_INVALID_ARG, "bad argument")
        _expect(lib.fe_stft_offline(h, P, 1, 256, 256, 1, P, NULL), FE_ERR_INVALID_ARG, "Tw=256: reflect padding of n_fft/2=256 needs a longer input")
        _expect(lib.fe_stft_offline(h, P, 1, 16000, 255, 1, P, NULL), FE_ERR_INVALID_ARG, "F=255")
        _expect(lib.fe_istft_offline(h, P, 1, 1, 256, 1, P, P, NULL), FE_ERR_INVALID_ARG, "bad argument")         # T <= 1: nothing to overlap
        _expect(lib.fe_istft_offline(h, P, 1, 10, 300, 1, P, P, NULL), FE_ERR_INVALID_ARG, "F=300")
        # debug table
        nm, rows, cols = c_char_p(), c_int(), c_int()
        for idx in (-1, lib.fe_debug_stages(h)):
            _expect(lib.fe_debug_stage(h, idx, byref(nm), byref(rows), byref(cols), byref(off)), FE_ERR_INVALID_ARG, f"stage index {idx}")
    finally:
        lib.fe_destroy(h)


def test_engine_settings_that_a_model_does_not_have():
    lib = _lib.load()
    rc, h = _create(_cfg("fe_nc"))                         # the noncausal model: time-batched engine only
    assert rc == 0
    try:
        _expect(lib.fe_set_offline_engine(h, _lib.FE_OFFLINE_FRAME_WALK), FE_ERR_UNSUPPORTED_CONFIG, "the noncausal model runs on the time-batched engine only")
        assert lib.fe_set_offline_engine(h, _lib.FE_OFFLINE_TIME_BATCHED) == 0
    finally:
        lib.fe_destroy(h)
    rc, h = _create(_cfg("fe_tk_b"))                       # a variant without a time-batched engine
    assert rc == 0
    try:
        _expect(lib.fe_set_offline_engine(h, _lib.FE_OFFLINE_TIME_BATCHED), FE_ERR_UNSUPPORTED_CONFIG, "no time-batched engine is compiled for this model")
        assert lib.fe_set_offline_engine(h, _lib.FE_OFFLINE_FRAME_WALK) == 0
    finally:
        lib.fe_destroy(h)


def test_fe_wg8_environment_default_is_validated(monkeypatch):
    """FE_WG8 sets a new handle's step kernel; anything but 0 | 1 | 2 is ignored (it used to go through atoi unchecked).  The
    setting is not readable through the ABI, so this checks what can be seen without a GPU: creation succeeds with any value."""
    for v in ("0", "1", "2", "7", "-1", "abc", ""):
        monkeypatch.setenv("FE_WG8", v)
        rc, h = _create(_cfg())
        assert rc == 0, (v, _err())
        _lib.load().fe_destroy(h)


def test_options_are_enumerable_validated_and_preset_by_their_environment_variables(monkeypatch):
    """r6: the kernel-selection switches are ONE documented fe_set_option surface; the environment variables only preset
    the values new handles start with, validated against the option's range."""
    import os
    import re
    lib = _lib.load()
    names = [lib.fe_option_name(i).decode() for i in range(lib.fe_options())]
    assert names == ["bsrnn_role_split", "bsrnn_stream_batch_min", "bsrnn_three_launch_step", "bsrnn_ov_profile", "fspen_stream_batch_min", "low_lds_companion", "bsrnn_fused_step", "lisennet_stream_batch_min"]
    assert lib.fe_option_name(len(names)) is None and lib.fe_option_name(-1) is None
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fastenhancer_hip.h")).read()
    integration = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    for n in names:                                         # header, library and INTEGRATION.md name the same options
        assert f'"{n}"' in header, n
        assert n in integration, n
    assert "fe_last_step_kernel" in header and "fe_last_step_kernel" in integration
    defaults = {"bsrnn_role_split": 1, "bsrnn_stream_batch_min": 2048, "bsrnn_three_launch_step": 1, "bsrnn_ov_profile": 0,
                "fspen_stream_batch_min": 1536, "low_lds_companion": 1, "bsrnn_fused_step": 0, "lisennet_stream_batch_min": 513}
    for v in ("FE_BSRNN_OV", "FE_BSRNN_SB", "FE_BSRNN_SPLIT", "FE_BSRNN_OV_PROF", "FE_FSPEN_SB", "FE_LOWLDS", "FE_NO_LOWLDS", "FE_WG8", "FE_BSRNN_FUSED", "FE_LISENNET_SB"):
        monkeypatch.delenv(v, raising=False)
    rc, h = _create(_cfg())
    assert rc == 0
    try:
        val = c_int(-7)
        for n in names:
            assert lib.fe_get_option(h, n.encode(), byref(val)) == 0 and val.value == defaults[n], (n, val.value)
        assert lib.fe_set_option(h, b"bsrnn_stream_batch_min", 4096) == 0
        assert lib.fe_get_option(h, b"bsrnn_stream_batch_min", byref(val)) == 0 and val.value == 4096
        _expect(lib.fe_set_option(h, b"bsrnn_role_split", 2), FE_ERR_INVALID_ARG, "outside [0, 1]")
        _expect(lib.fe_set_option(h, b"fspen_stream_batch_min", -1), FE_ERR_INVALID_ARG, "outside [0,")
        _expect(lib.fe_set_option(h, b"no_such_switch", 1), FE_ERR_INVALID_ARG, "no option named 'no_such_switch'")
        _expect(lib.fe_get_option(h, b"no_such_switch", byref(val)), FE_ERR_INVALID_ARG, "no option named")
        _expect(lib.fe_set_option(NULL, b"bsrnn_role_split", 1), FE_ERR_INVALID_ARG, "null argument")
        _expect(lib.fe_set_option(h, None, 1), FE_ERR_INVALID_ARG, "null argument")
        assert lib.fe_last_step_kernel(h) == b"" and lib.fe_last_step_kernel(NULL) == b""          # nothing has run
    finally:
        lib.fe_destroy(h)
    # presets: a valid value is taken, garbage and out-of-range values are ignored
    for env, opt, good, bad in (("FE_BSRNN_OV", "bsrnn_role_split", "0", ("2", "x", "-1", "")), ("FE_BSRNN_SB", "bsrnn_stream_batch_min", "512", ("-5", "1e3", "99999999999")),
                                ("FE_FSPEN_SB", "fspen_stream_batch_min", "0", ("no",)), ("FE_LOWLDS", "low_lds_companion", "0", ("3",)),
                                ("FE_LISENNET_SB", "lisennet_stream_batch_min", "1", ("-2", "many"))):
        for v in (good,) + bad:
            monkeypatch.setenv(env, v)
            rc, h = _create(_cfg())
            assert rc == 0
            val = c_int(-7)
            assert lib.fe_get_option(h, opt.encode(), byref(val)) == 0
            assert val.value == (int(good) if v == good else defaults[opt]), (env, v, val.value)
            lib.fe_destroy(h)
        monkeypatch.delenv(env)
    # the library reads its selection switches from the environment in fe_create only (fe_handle's constructor): no getenv of them elsewhere
    src = ""
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastenhancer_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip", ".inc")):
            src += open(os.path.join(csrc, f)).read()
    for v in ("FE_BSRNN_OV", "FE_BSRNN_SB", "FE_BSRNN_SPLIT", "FE_FSPEN_SB", "FE_WG8"):
        assert not re.search(r'getenv\("' + v + r'"\)', src), v


# ------------------------------------------------------------------------------------------------ GPU: checks behind check_ready()
def _loaded_engine(name):
    from common import build_oracle
    import numpy as np
    cfg, sd, fused, orc = build_oracle(name)
    eng = Engine(product_config(name), torch.device("cuda:0"))
    eng.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return eng


@pytest.mark.gpu
def test_compute_calls_reject_bad_arguments_with_loaded_weights():
    lib = _lib.load()
    eng = _loaded_engine("fe_b")
    h, dev = eng._h, eng.device
    B, H = 4, eng.cfg.hop_size
    x = torch.zeros(B, 2 * H, device=dev)
    out = torch.zeros(B, 2 * H, device=dev)
    st = eng.new_state(B)
    px, po, ps = (c_void_p(t.data_ptr()) for t in (x, out, st))
    s = c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert lib.fe_step(h, px, 2 * H, ps, po, 2 * H, B, 2, s) == 0
    for args in [(NULL, 2 * H, ps, po, 2 * H, B, 2), (px, 2 * H, NULL, po, 2 * H, B, 2), (px, 2 * H, ps, NULL, 2 * H, B, 2),
                 (px, 2 * H, ps, po, 2 * H, 0, 2), (px, 2 * H, ps, po, 2 * H, -1, 2), (px, 2 * H, ps, po, 2 * H, B, 0), (px, 2 * H, ps, po, 2 * H, B, -5)]:
        _expect(lib.fe_step(h, *args, s), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_step(h, px, 2 * H - 1, ps, po, 2 * H, B, 2, s), FE_ERR_INVALID_ARG, f"in_stride {2 * H - 1} < T*H")
    _expect(lib.fe_step(h, px, 2 * H, ps, po, H, B, 2, s), FE_ERR_INVALID_ARG, f"out_stride {H} < T*H")
    _expect(lib.fe_step_host(h, px, 2 * H, ps, po, 2 * H, B, 2, 0, px, s), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_step_host(h, px, 2 * H, ps, po, 2 * H, B, 2, 1, NULL, s), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_debug_step(h, px, 2 * H, ps, po, 2 * H, B, NULL, s), FE_ERR_INVALID_ARG, "null dbg buffer")
    # spec -> spec
    spec = torch.zeros(B, eng.cfg.n_fft // 2 + 1, 3, 2, device=dev)
    hs = torch.zeros(eng.cfg.rf_blocks * B * eng.cfg.rf_freq * eng.cfg.rf_channels, device=dev)
    so = torch.zeros_like(spec)
    psp, phs, pso = (c_void_p(t.data_ptr()) for t in (spec, hs, so))
    assert lib.fe_spec_step(h, psp, phs, pso, B, 3, s) == 0
    for args in [(NULL, phs, pso, B, 3), (psp, NULL, pso, B, 3), (psp, phs, NULL, B, 3), (psp, phs, pso, 0, 3), (psp, phs, pso, B, 0)]:
        _expect(lib.fe_spec_step(h, *args, s), FE_ERR_INVALID_ARG, "bad argument")
    # offline
    Tw = 20 * H
    noisy = torch.zeros(2, Tw, device=dev)
    T = 1 + Tw // H
    wav = torch.zeros(2, H * (T - 1), device=dev)
    sh = torch.zeros(2, eng.cfg.n_fft // 2, T, 2, device=dev)
    work = torch.zeros(int(lib.fe_offline_ragged_work_floats(h, 2, Tw)), device=dev)
    pn, pw, psh, pwk = (c_void_p(t.data_ptr()) for t in (noisy, wav, sh, work))
    assert lib.fe_offline(h, pn, 2, Tw, pw, psh, pwk, s) == 0
    for args in [(NULL, 2, Tw, pw, psh, pwk), (pn, 2, Tw, NULL, psh, pwk), (pn, 2, Tw, pw, NULL, pwk), (pn, 2, Tw, pw, psh, NULL), (pn, 0, Tw, pw, psh, pwk)]:
        _expect(lib.fe_offline(h, *args, s), FE_ERR_INVALID_ARG, "bad argument")
    N = eng.cfg.n_fft
    for bad in (0, -4, N // 2):
        _expect(lib.fe_offline(h, pn, 2, bad, pw, psh, pwk, s), FE_ERR_INVALID_ARG, f"Tw={bad}: reflect padding of n_fft/2={N // 2} needs a longer input")
    # ragged: a zero-length utterance, an utterance longer than its row, an output row shorter than the longest utterance's frames
    out_stride = H * (T - 1)
    _expect(lib.fe_offline_ragged(h, pn, Tw, (c_int * 2)(Tw, 0), 2, pw, out_stride, psh, pwk, s), FE_ERR_INVALID_ARG, "Tw[1]=0: reflect padding")
    _expect(lib.fe_offline_ragged(h, pn, Tw, (c_int * 2)(Tw + 1, Tw), 2, pw, out_stride, psh, pwk, s), FE_ERR_INVALID_ARG, f"Tw[0]={Tw + 1} > in_stride {Tw}")
    _expect(lib.fe_offline_ragged(h, pn, Tw, (c_int * 2)(Tw, Tw // 2), 2, pw, out_stride - 1, psh, pwk, s), FE_ERR_INVALID_ARG, f"out_stride {out_stride - 1} < H*(Tmax-1)")
    _expect(lib.fe_offline_ragged(h, pn, Tw, None, 2, pw, out_stride, psh, pwk, s), FE_ERR_INVALID_ARG, "bad argument")
    _expect(lib.fe_offline_ragged(h, pn, Tw, (c_int * 2)(Tw, Tw), 0, pw, out_stride, psh, pwk, s), FE_ERR_INVALID_ARG, "bad argument")
    assert lib.fe_offline_ragged(h, pn, Tw, (c_int * 2)(Tw, Tw // 2), 2, pw, out_stride, psh, pwk, s) == 0
    torch.cuda.synchronize()
    assert lib.fe_last_error() is not None             # (the text of the last failure stays readable after a success)


@pytest.mark.gpu
def test_streaming_entry_points_on_a_noncausal_handle():
    """models/fastenhancer/noncausal/model.py has the offline Model only: fe_step / fe_spec_step on its handle are FE_ERR_UNSUPPORTED_CONFIG"""
    lib = _lib.load()
    eng = _loaded_engine("fe_nc")
    h, dev = eng._h, eng.device
    H = eng.cfg.hop_size
    x = torch.zeros(1, H, device=dev)
    st = torch.zeros(max(1, int(lib.fe_state_floats(h, 1))), device=dev)
    s = c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    px, ps = c_void_p(x.data_ptr()), c_void_p(st.data_ptr())
    _expect(lib.fe_step(h, px, H, ps, px, H, 1, 1, s), FE_ERR_UNSUPPORTED_CONFIG, "the noncausal model has no streaming step")
    spec = torch.zeros(1, eng.cfg.n_fft // 2 + 1, 2, 2, device=dev)
    psp = c_void_p(spec.data_ptr())
    _expect(lib.fe_spec_step(h, psp, ps, psp, 1, 2, s), FE_ERR_UNSUPPORTED_CONFIG, "the noncausal model has no spec -> spec step with caches")


@pytest.mark.gpu
def test_time_batched_engine_request_on_a_model_without_one():
    lib = _lib.load()
    from common import build_bsrnn_oracle
    from fastenhancer_amd.config import BSRNNConfig
    eng = Engine(BSRNNConfig.from_model_kwargs(**BSRNN_KWARGS["bsrnn_xt"][0]), torch.device("cuda:0"))
    _expect(lib.fe_set_offline_engine(eng._h, _lib.FE_OFFLINE_TIME_BATCHED), FE_ERR_UNSUPPORTED_CONFIG, "no time-batched engine is compiled for this model")
    x = torch.zeros(1, 256, device="cuda:0")
    p = c_void_p(x.data_ptr())
    _expect(lib.fe_step(eng._h, p, 256, p, p, 256, 1, 1, NULL), FE_ERR_NO_WEIGHTS, "fe_load_weights has not been called")


@pytest.mark.gpu
def test_last_step_kernel_names_what_was_dispatched():
    """r6: the library says which kernel family / instantiation its last compute call enqueued - bench.py's roofline.kernel
    is this string, not a guess made from flags and environment variables.  It follows fe_set_step_kernel, the batch size relative to the CU
    count, the entry point and fe_set_option."""
    dev = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    eng = _loaded_engine("fe_b")
    assert eng.last_step_kernel() == ""
    H = eng.cfg.hop_size

    def step(B, T=1):
        st = eng.new_state(B)
        eng.step(torch.zeros(B, T * H, device=dev), st, T=T)
        torch.cuda.synchronize()
        return eng.last_step_kernel()

    assert step(4) == "fe_frame8_kernel [shape B]"
    assert step(4, T=3) == "fe_frame_kernel<generic> [shape B]"                        # chunks run on the 256-thread kernel
    many = step(cus + 8)
    assert many == "fe_frame_kernel<LOW=1, per-hop> [shape BLOW]", many                 # above one stream per CU: the low-LDS companion
    eng.set_option("low_lds_companion", 0)
    assert step(cus + 8) == "fe_frame_kernel<per-hop, persistent> [shape B]"
    eng.set_option("low_lds_companion", 1)
    eng.set_step_kernel("wg8_persist")
    assert step(cus + 8) == "fe_frame8_kernel<persistent> [shape B]"
    eng.set_step_kernel("waves4")
    assert step(4) == "fe_frame_kernel<per-hop> [shape B]"
    eng.set_step_kernel("wg8")
    assert step(4) == "fe_frame8_kernel [shape B]"
    x = torch.zeros(2, 4 * 16000, device=dev)
    eng.offline(x)
    torch.cuda.synchronize()
    off = eng.last_step_kernel()
    assert off.startswith("tb_enc_kernel + ") and "tb_blk_kernel" in off and off.endswith("istft_ola_kernel [shape B]"), off
    # BSRNN: the role-split PART 1 up to one stream per CU, the phase-by-phase kernel above or on request, the stream-batched layers from the threshold
    import importlib
    import numpy as np
    from common import build_bsrnn_oracle
    kw = BSRNN_KWARGS["bsrnn_xt"][0]
    bsd = build_bsrnn_oracle("bsrnn_xt")[1]
    m = importlib.import_module("fastenhancer_amd.models.bsrnn.model").ONNXModel(**kw).to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in bsd.items()}, strict=True)
    beng = m.engine

    def bstep(B):
        st = beng.new_state(B)
        beng.step(torch.zeros(B, beng.cfg.hop_size, device=dev), st, T=1)
        torch.cuda.synchronize()
        return beng.last_step_kernel()

    assert bstep(4) == "bsrnn_ov_kernel + bsrnn_mlp_kernel<one 16-stream tile per workgroup> + bsrnn_frame_kernel<PART 2> [shape xt]"
    beng.set_option("bsrnn_fused_step", 1)                                                  # r6 (measured negative, off by default): the whole step in one cooperative launch
    assert bstep(4) == "bsrnn_ov_kernel<fused step> [shape xt]"
    beng.set_option("bsrnn_fused_step", 0)
    beng.set_option("bsrnn_role_split", 0)
    assert bstep(4) == "bsrnn_frame_kernel<PART 1> + bsrnn_mlp_kernel<one 16-stream tile per workgroup> + bsrnn_frame_kernel<PART 2> [shape xt]"
    beng.set_option("bsrnn_role_split", 1)
    assert bstep(cus + 8).startswith("bsrnn_frame_kernel<PART 1, two workgroups per CU> + bsrnn_mlp_kernel + ")
    beng.set_option("bsrnn_stream_batch_min", 32)
    assert bstep(48) == "bsrnn_frame_kernel<PART 3> + bsrnn_sb_layers_kernel + bsrnn_mlp_kernel + bsrnn_frame_kernel<PART 2> [shape xt]"
    beng.set_option("bsrnn_three_launch_step", 0)
    beng.set_option("bsrnn_stream_batch_min", 2048)
