"""Build libfastenhancer_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One translation unit per compiled shape (csrc/fe_shapes.def) plus the C-ABI unit; they are compiled
in parallel and linked into one shared library.  Objects are cached under csrc/_obj/ and rebuilt when
any source they include changes."""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# FE_BUILD_TAG=<tag>: a side build (measurement variants, A/B candidates) with its own object cache, linked to
# ab/lib_<tag>.so instead of the in-tree library (tools/ab_bench.sh, tools/gpu_phases.py swap it in on the GPU box)
_TAG = os.environ.get("FE_BUILD_TAG", "")
OBJ = os.path.join(CSRC, "_obj" + ("_" + _TAG if _TAG else ""))
LIB = os.path.join(HERE, "libfastenhancer_hip.so") if not _TAG else os.path.join(os.path.dirname(HERE), "ab", f"lib_{_TAG}.so")
FE_DEPS = ["fe_kernels.hip.h", "fe_frame8.hip.h", "fe_impl.h", "tb_kernels.hip.h"]
BSRNN_DEPS = ["fe_kernels.hip.h", "bsrnn_kernels.hip.h", "bsrnn_sb_kernels.hip.h", "bsrnn_ov_kernels.hip.h"]
FSPEN_DEPS = ["fe_kernels.hip.h", "fspen_kernels.hip.h", "fspen_sb_kernels.hip.h"]
LISENNET_DEPS = ["fe_kernels.hip.h", "fspen_kernels.hip.h", "fspen_sb_kernels.hip.h", "lisennet_kernels.hip.h", "lisennet_sb_kernels.hip.h"]
API_DEPS = ["fe_kernels.hip.h", "fe_frame8.hip.h", "fe_impl.h", "tb_kernels.hip.h", "fe_shapes.def", "bsrnn_kernels.hip.h", "bsrnn_sb_kernels.hip.h", "bsrnn_ov_kernels.hip.h", "fe_bsrnn_shapes.def", "stft_kernels.hip.h", "fspen_kernels.hip.h", "fspen_sb_kernels.hip.h", "lisennet_kernels.hip.h", "lisennet_sb_kernels.hip.h",
            "fe_api_bsrnn.inc", "fe_api_fspen.inc", "fe_api_lisennet.inc",
            os.path.join("..", "..", "include", "fastenhancer_hip.h")]
# -amdgpu-mfma-vgpr-form: MFMA accumulators in VGPRs instead of AGPRs - every epilogue read of an AGPR accumulator is a
# v_accvgpr_read, a VALU instruction that the fp32 matrix path cannot overlap (~600 of them per wave and frame on
# FastEnhancer_B: +1.8 %, 48 kHz B +2.6 %, T +2.3 %, the big shapes +1 %).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form"]
FLAGS += os.environ.get("FE_EXTRA_DEFS", "").split()   # e.g. -DFE_PROBE_HOT (tools/gpu_phases.py <shape> <streams> 1)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


# Shapes beyond the shipped yamls: `python -m fastenhancer_amd.build --add-shape C1,NL,C2,F2,KB,NFFT,HOP[,KT]` appends a line
# to this (git-ignored, optional) file and rebuilds; fe_api.hip includes it after fe_shapes.def.  FE_LOCAL_DEF overrides the path.
LOCAL_DEF = os.environ.get("FE_LOCAL_DEF") or os.path.join(CSRC, "fe_shapes_local.def")
# FE_SHAPES_DEF=<file>: a side build (FE_BUILD_TAG) with a SHORT shape list instead of csrc/fe_shapes.def - seconds instead of
# minutes while a kernel is being worked on (tools/dev_shapes.def: B, T, L, NC); the library then knows those shapes only
SHAPES_DEF = os.environ.get("FE_SHAPES_DEF", "")
if SHAPES_DEF and not _TAG:
    raise SystemExit("FE_SHAPES_DEF is for side builds: set FE_BUILD_TAG too")


def shapes(fname="fe_shapes.def", macro="X"):
    out = []
    files = [os.path.join(CSRC, fname)]
    if fname == "fe_shapes.def" and SHAPES_DEF:
        files = [os.path.abspath(SHAPES_DEF)]
    elif fname == "fe_shapes.def" and os.path.exists(LOCAL_DEF):
        files.append(LOCAL_DEF)
    for f in files:
        for line in open(f):
            m = re.match(r"\s*" + macro + r"\(\s*(\w+)\s*,(.*)\)\s*$", line)
            if m:
                out.append((m.group(1), ",".join(x.strip() for x in m.group(2).split(","))))
    return out


def add_shape(spec: str) -> str:
    """`C1,NL,C2,F2,KB,NFFT,HOP[,KT]` (channels, len(kernel_size)-1, rnnformer channels / freq / num_blocks, n_fft, hop_size,
    kernel_size_time) -> a line in the local shape list.  The kernel template's constraints are checked here with a
    readable message (hipcc would report them as failed static_asserts): stride 4 and kernel_size [8, 3, ...] are fixed."""
    v = [int(x) for x in spec.replace(" ", "").split(",")]
    if len(v) not in (7, 8, 10, 11, 12, 13):
        raise SystemExit("--add-shape wants C1,NL,C2,F2,KB,NFFT,HOP[,KT[,0,FR[,TA[,LN[,BD]]]]]  (FR = 1: the dprnn variant, TA = 31: the dptransformer variant, "
                         "LN = 1: the ln variant, BD = 1: the noncausal variant)")
    C1, NL, C2, F2, KB, NFFT, HOP = v[:7]
    KT = v[7] if len(v) >= 8 else 1
    errs = []
    if C1 % 4 or C2 % 4 or F2 % 4:
        errs.append("channels, rnnformer channels and rnnformer freq must be multiples of 4")
    if C2 % 4 or (C2 // 4) < 1:
        errs.append("rnnformer channels must be divisible by the 4 heads")
    if NFFT not in (512, 1024):
        errs.append("n_fft must be 512 or 1024")
    if not (0 < HOP <= NFFT):
        errs.append("0 < hop_size <= n_fft")
    if not (1 <= NL <= 7 and 1 <= KB <= 8 and 1 <= KT <= 4 and NL * KT <= 16):
        errs.append("1 <= layers <= 7, 1 <= num_blocks <= 8, 1 <= kernel_size_time <= 4")
    if errs:
        raise SystemExit("unsupported shape: " + "; ".join(errs))
    args = ",".join(str(x) for x in v)
    if any(a == args or (len(v) == 7 and a == args + ",1") for _, a in shapes()):
        return ""
    name = "U" + "_".join(str(x) for x in v)
    with open(LOCAL_DEF, "a") as f:
        f.write(f"X({name}, {', '.join(str(x) for x in v)})\n")
    return name


def _digest(paths, extra="") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def source_key() -> str:
    """Digest of every source the library is built from (csrc/*, the C-ABI header): libfastenhancer_hip.so carries it as fe_build_key(), and
    fastenhancer_amd._lib.load() refuses an in-tree library whose key is not the tree's - a stale build cannot pass for the shipped sources."""
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".inc", ".in", ".def")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "fastenhancer_hip.h"))
    return _digest(files)


def _compile(job):
    src, obj, defs, stamp, key = job
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
        return obj, False
    if _TAG:       # side build: a translation unit the main build already compiled with the same key is taken from its cache
        mobj, mstamp = os.path.join(CSRC, "_obj", os.path.basename(obj)), os.path.join(CSRC, "_obj", os.path.basename(stamp))
        if os.path.exists(mobj) and os.path.exists(mstamp) and open(mstamp).read() == key:
            import shutil
            shutil.copyfile(mobj, obj)
            open(stamp, "w").write(key)
            return obj, True
    cmd = [_hipcc()] + FLAGS + defs + ["-c", "-x", "hip", src, "-o", obj]
    subprocess.check_call(cmd)
    open(stamp, "w").write(key)
    return obj, True


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    common = [os.path.join(CSRC, d) for d in FE_DEPS]
    common_b = [os.path.join(CSRC, d) for d in BSRNN_DEPS]
    common_api = [os.path.join(CSRC, d) for d in API_DEPS] + ([LOCAL_DEF] if os.path.exists(LOCAL_DEF) else [])
    jobs = []
    api = os.path.join(CSRC, "fe_api.hip")
    api_defs = [f'-DFE_LOCAL_DEF="{LOCAL_DEF}"'] if os.path.exists(LOCAL_DEF) else []
    if SHAPES_DEF:
        api_defs = [f'-DFE_SHAPES_DEF="{os.path.abspath(SHAPES_DEF)}"']
        common_api.append(os.path.abspath(SHAPES_DEF))
    jobs.append((api, os.path.join(OBJ, "fe_api.o"), api_defs, os.path.join(OBJ, "fe_api.stamp"), _digest(common_api + [api], " ".join(FLAGS + api_defs))))
    tmpl = os.path.join(CSRC, "fe_shape.hip.in")
    for name, args in shapes():
        defs = [f"-DFE_SHAPE_NAME={name}", f"-DFE_SHAPE_ARGS={args}"]
        jobs.append((tmpl, os.path.join(OBJ, f"fe_shape_{name}.o"), defs, os.path.join(OBJ, f"fe_shape_{name}.stamp"),
                     _digest(common + [tmpl], " ".join(FLAGS + defs))))
    tmpl_b = os.path.join(CSRC, "fe_bsrnn_shape.hip.in")
    for name, args in shapes("fe_bsrnn_shapes.def", "XB"):
        defs = [f"-DFE_SHAPE_NAME={name}", f"-DFE_SHAPE_ARGS={args}"]
        jobs.append((tmpl_b, os.path.join(OBJ, f"fe_bsrnn_{name}.o"), defs, os.path.join(OBJ, f"fe_bsrnn_{name}.stamp"),
                     _digest(common_b + [tmpl_b], " ".join(FLAGS + defs))))
    # The two VALU-only models are compiled without the SLP vectoriser: it pairs the FMAs of two outputs into v_pk_fma_f32 whose packed
    # operand it then has to assemble with one or two v_mov_b32 each (811 moves in FSPEN's hot kernel): FSPEN +7 % at 4096 streams,
    # +2 % at 256, LiSenNet +1 % (profiles/r3z_no_slp_ab.txt; the MFMA kernels and BSRNN lose 0.2-1.3 % without it and keep it).
    valu_defs = ["-fno-slp-vectorize"]
    fsp = os.path.join(CSRC, "fe_fspen.hip")
    jobs.append((fsp, os.path.join(OBJ, "fe_fspen.o"), valu_defs, os.path.join(OBJ, "fe_fspen.stamp"),
                 _digest([os.path.join(CSRC, d) for d in FSPEN_DEPS] + [fsp], " ".join(FLAGS + valu_defs))))
    lsn = os.path.join(CSRC, "fe_lisennet.hip")
    jobs.append((lsn, os.path.join(OBJ, "fe_lisennet.o"), valu_defs, os.path.join(OBJ, "fe_lisennet.stamp"),
                 _digest([os.path.join(CSRC, d) for d in LISENNET_DEPS] + [lsn], " ".join(FLAGS + valu_defs))))
    # fe_build_key(): the digest of the sources, compiled into the library (a translation unit of its own: seconds)
    bk = os.path.join(OBJ, "fe_build_key.cpp")
    skey = source_key()
    bk_src = '#include "../../../include/fastenhancer_hip.h"\nextern "C" const char* fe_build_key(void) { return "%s"; }\n' % skey
    if not os.path.exists(bk) or open(bk).read() != bk_src:
        open(bk, "w").write(bk_src)
    jobs.append((bk, os.path.join(OBJ, "fe_build_key.o"), [], os.path.join(OBJ, "fe_build_key.stamp"), _digest([bk], " ".join(FLAGS))))
    if force:
        for j in jobs:
            if os.path.exists(j[3]):
                os.remove(j[3])
    workers = max(1, min(len(jobs), (os.cpu_count() or 2)))
    rebuilt = False
    with concurrent.futures.ThreadPoolExecutor(workers) as ex:
        for obj, did in ex.map(_compile, jobs):
            rebuilt |= did
            if verbose and did:
                print(f"[fastenhancer_amd] compiled {os.path.basename(obj)}", file=sys.stderr)
    # objects of shapes that are no longer in the .def files must not linger (nor their stamps)
    wanted = {os.path.basename(j[1]) for j in jobs} | {os.path.basename(j[3]) for j in jobs}
    for f in os.listdir(OBJ):
        if f not in wanted and (f.endswith(".o") or f.endswith(".stamp")):      # (link.key is neither)
            os.remove(os.path.join(OBJ, f))
    # relink whenever the library is missing or older than ANY object (a link that failed after the objects were
    # compiled leaves rebuilt == False on the next run), to a temporary name that replaces the library only on success;
    # the digest of all object keys is kept next to the library so that a changed object SET relinks too
    link_key = hashlib.sha256(" ".join(sorted(j[4] + os.path.basename(j[1]) for j in jobs)).encode()).hexdigest()[:16]
    key_file = os.path.join(OBJ, "link.key")
    stale = (not os.path.exists(LIB) or not os.path.exists(key_file) or open(key_file).read() != link_key
             or any(os.path.getmtime(j[1]) > os.path.getmtime(LIB) for j in jobs))
    if rebuilt or stale:
        tmp = LIB + ".tmp"
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [j[1] for j in jobs]
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
        open(key_file, "w").write(link_key)
        if verbose:
            print(f"[fastenhancer_amd] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    if "--add-shape" in sys.argv:
        added = add_shape(sys.argv[sys.argv.index("--add-shape") + 1])
        print(f"[fastenhancer_amd] {'added shape ' + added if added else 'shape already compiled'}", file=sys.stderr)
    build(force="--force" in sys.argv)
