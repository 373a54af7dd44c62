"""Build libgeoa3_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

    python -m geoa3_amd.build [--force]

The shared object lands in geoa3_amd/lib/ (git-ignored, travels to the GPU box with the tree).
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libgeoa3_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


# per-file flags.  pointnet_conv_chain.hip: no SLP vectorisation = no packed-FP32 instructions (v_pk_mul_f32 / v_pk_fma_f32 with
# SGPR-pair operands) in conv_bwd_chain_kernel, which with two wavefronts per SIMD computed wrong values in lanes 48-63
# (NOTEBOOK 5a: found with tools/ub/dtpart_pair.hip; the same source with the packed instructions: ~1e-4 of the workgroups).
# pointnet_conv_split.hip: the one-layer-per-launch kernels the chains are held to BIT FOR BIT (tests/test_gpu_pointnet.py)
# must contract their multiply-adds the same way (no cost: configs[1] does not run them, configs[3] +0.2 %).
# (pointnet_gemm.hip -- the FC heads -- loses 4 % of the iteration without SLP and is left alone.)
# pointnet2_sa.hip, pointnet2_sa2.hip: relu / max of matrix-core results without the canonicalising v_max x, x the IEEE maxnum semantics put in front
# of every one of them (a sixth of the level-1 kernels' vector instructions); NaNs still propagate through the products.
# -fno-slp-vectorize there: the operand split stays at two instructions per element (sa_split2).
# -fno-slp-vectorize: the SLP vectoriser's packed-FP32 instructions (v_pk_add / mul / fma_f32) are behind three sightings of
# wrong values on gfx950 (NOTEBOOK 5a: conv_bwd_chain_kernel at two waves per SIMD; the sampler beside sa1_fwd_kernel;
# geo_fused_kernel's long-row path at four waves per SIMD) -- every file compiles without them except the ONE where the
# packing pays (pointnet_gemm.hip: conv_cm64_kernel and the FC kernels, +4.6 % on configs[1] without it; every other file 0:
# tools/gpu_ab_noslp.sh), whose kernels stay under the replay soak's seven shapes.
_NOSLP = ["-fno-slp-vectorize", "-fno-vectorize"]   # (the loop vectoriser pairs fp32 work the same way: sa1_stage, attack_state)
SLP_FILES = ("pointnet_gemm.hip",)
FILE_FLAGS = {f: list(_NOSLP) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and f not in SLP_FILES}
for _f in ("pointnet2_sa.hip", "pointnet2_sa2.hip", "pointnet2_sa2b.hip", "geom_filter.hip", "geom_grid.hip"):
    FILE_FLAGS[_f] = ["-fno-honor-nans"] + _NOSLP


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


_HIPCC_VERSION = None


def hipcc_version() -> str:
    """`hipcc --version` (first lines): part of every object's stamp -- a compiler update rebuilds the library and so puts
    every file through the ISA guard again."""
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        r = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True)
        _HIPCC_VERSION = (r.stdout + r.stderr).strip()
    return _HIPCC_VERSION


def _digest(path: str, flags=None) -> str:
    h = hashlib.sha1()
    for p in [path] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
            os.path.join(HERE, "..", "include", "geoa3_hip.h"), os.path.join(HERE, "..", "include", "geoa3_hip_debug.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags or FLAGS).encode())
    # the compiler and the guard itself: a new hipcc, or a changed guard pattern, re-checks every object
    h.update(hipcc_version().encode())
    h.update(repr((ISA_GUARD_ALL, sorted(ISA_GUARDS.items()))).encode())
    return h.hexdigest()


def _compile(src: str, force: bool, objdir: str = OBJDIR, flags=None) -> str:
    flags_in = tuple(flags or ())
    flags = list(flags or FLAGS)
    if "--no-file-flags" in flags:     # (tools: the build WITHOUT the per-file flags, e.g. the faulty one of NOTEBOOK 5a)
        flags.remove("--no-file-flags")
    elif os.path.basename(src) in os.environ.get("GEOA3_NO_FILE_FLAGS_FOR", "").split(",") and objdir != OBJDIR:
        pass                           # (tools: a variant build without SOME files' flags)
    else:
        flags += FILE_FLAGS.get(os.path.basename(src), [])
    # (tools: GEOA3_EXTRA_FILE_FLAGS="a.hip,b.hip:-fflag" adds a flag to some files of a VARIANT build, e.g. to price
    # -fno-slp-vectorize file by file: tools/gpu_ab_noslp.sh)
    extra = os.environ.get("GEOA3_EXTRA_FILE_FLAGS", "")
    if extra and objdir != OBJDIR:
        names, _, fl = extra.partition(":")
        if os.path.basename(src) in names.split(",") and fl not in flags:
            flags.append(fl)
    obj = os.path.join(objdir, os.path.basename(src) + ".o")
    stamp = obj + ".sha1"
    dig = _digest(src, flags)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    guarded = ("--no-file-flags" not in (flags_in or ()) and
               not (objdir != OBJDIR and os.environ.get("GEOA3_NO_FILE_FLAGS_FOR")))   # (a tools variant may build the faulty form)
    import tempfile
    with tempfile.TemporaryDirectory(dir=objdir) as d:
        # ONE compilation: -save-temps=obj leaves the device assembly the object was assembled from beside it, and the ISA
        # guard reads THAT (no second `hipcc -S` per file)
        tmp_obj = os.path.join(d, os.path.basename(src) + ".o")
        cmd = [_hipcc()] + flags + (["-save-temps=obj"] if guarded else []) + ["-c", src, "-o", tmp_obj]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=d)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        if guarded:
            base = os.path.splitext(os.path.basename(src))[0]
            asm_file = os.path.join(d, "%s-hip-amdgcn-amd-amdhsa-%s.s" % (base, ARCH))
            if not os.path.exists(asm_file):
                raise RuntimeError("ISA guard: hipcc -save-temps left no device assembly for %s (looked for %s)" % (src, asm_file))
            _isa_guard_asm(src, open(asm_file).read(), flags)
        os.replace(tmp_obj, obj)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


# kernels that must not hold packed-FP32 instructions (NOTEBOOK 5a): checked on the ISA the build's own flags produce, so that a
# lost per-file flag -- or a compiler that forms v_pk_*_f32 in another pass -- fails the BUILD instead of shipping a kernel
# that is silently wrong in ~1e-4 of its workgroups
# (round 5: and the farthest-point sampler, whose SLP-packed distance update is exact alone and wrong in 1.4e-3 of its rounds
# beside sa1_fwd_kernel / sa1_bwd_kernel on the same CUs -- tools/ub/pk_fp32_coresidency.hip; EVERY instance of the template)
ISA_GUARDS = {"pointnet_conv_chain.hip": ("conv_bwd_chain_kernel", r"v_pk_(mul|fma|add)_f32"),
              "pointnet2_ops.hip": ("fps_kernel", r"v_pk_(mul|fma|add)_f32"),
              "geom_loss.hip": ("geo_fused_kernel", r"v_pk_(mul|fma|add)_f32")}


# ... and NO file, whatever its flags, may hold a packed-FP32 arithmetic instruction with an op_sel bit -- a LOW result that
# reads the HIGH half of a source register pair.  That is the form the stand-alone reproducer pins the fault on
# (tools/ub/pk_neg_mfma_min.hip: v_pk_add / mul / fma_f32 with op_sel on src1 return src1 = 0 in lanes 48-63, ~1e-8 .. 5e-7 of
# the executions, when the neighbouring wavefronts mix vector and f16 matrix instructions; op_sel_hi, neg and v_pk_mov_b32 are
# clean: NOTEBOOK 5a).  Only the SLP vectoriser forms it; every kernel that ever failed held one to eight of them.
# The pattern is deliberately wider than today's spelling (`op_sel:[0,1,0]`): any packed-FP32 arithmetic mnemonic (mul / fma /
# add, with or without an encoding suffix) whose operand-select list -- `op_sel`, NOT `op_sel_hi` -- names a high half for
# any source, whatever the separator a future disassembler prints.  tests/test_abi.py compiles a source that MUST be refused
# (isa_guard_selftest), so a compiler that forms or prints the instruction differently fails the test suite, not a user.
ISA_GUARD_ALL = r"(?i)v_pk_(mul|fma|add)_f32\w*[^\n]*\bop_sel\s*[:=]\s*\[[^\]]*1"


GUARD_SELFTEST_SOURCE = r"""
// MUST be refused by the ISA guard: packed FP32 arithmetic whose LOW result reads the HIGH half of a source register pair
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void guard_selftest_kernel(const f2* a, const f2* b, f2* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const f2 x = a[i], y = b[i];
  const f2 ys = {y.y, y.x};          // swapped halves: v_pk_fma_f32 ... op_sel:[0,1,0] op_sel_hi:[1,0,1]
  out[i] = x * ys + x;
}
"""


def isa_guard_selftest() -> str:
    """Compile a ten-line source that holds the faulty instruction form with the library's own flags and run the guard on
    it: returns the guard's refusal message, raises AssertionError when the guard lets it through (a compiler that no
    longer forms the instruction from this source, or prints it in a spelling the pattern misses)."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "guard_selftest.hip")
        with open(src, "w") as f:
            f.write(GUARD_SELFTEST_SOURCE)
        try:
            _isa_guard(src, FLAGS)
        except RuntimeError as e:
            if "ISA guard" in str(e) and "op_sel" in str(e):
                return str(e)
            raise
    raise AssertionError("the ISA guard accepted a source that holds packed FP32 arithmetic with op_sel on a source operand: "
                         "either this hipcc (%s) no longer forms v_pk_*_f32 ... op_sel from the self-test source, or it prints "
                         "it in a way ISA_GUARD_ALL misses -- look at the disassembly before trusting the build"
                         % hipcc_version().splitlines()[0])


def _isa_guard(src: str, flags) -> None:
    """The guard on a source of its own (the self-test): device assembly by `hipcc -S`, then _isa_guard_asm."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "guard.s")
        cmd = [_hipcc()] + [f for f in flags if f != "-fPIC"] + ["-S", "--cuda-device-only", "-o", out, src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("ISA guard: hipcc -S failed for %s:\n%s" % (src, r.stderr))
        asm = open(out).read()
    _isa_guard_asm(src, asm, flags)


def _isa_guard_asm(src: str, asm: str, flags) -> None:
    import re
    kernel, pattern = ISA_GUARDS.get(os.path.basename(src), (None, None))
    hit = re.search(ISA_GUARD_ALL, asm)
    if hit:
        raise RuntimeError("ISA guard: %s holds '%s' (NOTEBOOK 5a: packed FP32 with op_sel reads a zero operand in lanes 48-63 "
                           "beside matrix-core wavefronts); keep the two values apart (an opaque asm barrier) or compile the "
                           "file with %s" % (os.path.basename(src), hit.group(0).strip(), " ".join(_NOSLP)))
    if kernel is None:
        return
    found = list(re.finditer(r"^(_Z\w*%s\w*):" % kernel, asm, re.M))
    if not found:
        raise RuntimeError("ISA guard: kernel %s not found in %s" % (kernel, src))
    for m in found:   # every instance of a kernel template
        body = asm[m.end():]
        body = body[:body.index("s_endpgm")]
        hit = re.search(pattern, body)
        if hit:
            raise RuntimeError("ISA guard: %s holds %s (NOTEBOOK 5a: wrong values beside other wavefronts on its SIMDs); compile "
                               "%s with %s" % (m.group(1), hit.group(0), os.path.basename(src),
                                               FILE_FLAGS.get(os.path.basename(src))))


def build(force: bool = False, verbose: bool = True, extra_flags=(), libdir: str = LIBDIR) -> str:
    """The product library (no arguments), or a variant build for tools/ (extra -D flags, its own directory; loaded
    through GEOA3_LIB_PATH, see _lib.py)."""
    objdir, lib = os.path.join(libdir, "obj"), os.path.join(libdir, "libgeoa3_hip.so")
    flags = FLAGS + list(extra_flags)
    os.makedirs(objdir, exist_ok=True)
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, objdir, flags), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", lib, "%.0f KB" % (os.path.getsize(lib) / 1024))
    return lib


if __name__ == "__main__":
    # python -m geoa3_amd.build [--force] [--variant DIR -DX=Y ...]
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build(force="--force" in sys.argv, libdir=os.path.abspath(sys.argv[i + 1]),
              extra_flags=[f for f in sys.argv[i + 2:] if f.startswith("-") and f != "--force"])
    else:
        build(force="--force" in sys.argv)
