"""CPU-only: the C-ABI library builds, loads and exports every symbol include/geoa3_hip.h (product ABI) and
include/geoa3_hip_debug.h (diagnostics) declare; the product header holds no diagnostics and the library exports no
mutable global."""
import os
import re

import pytest
import shutil
import subprocess

from geoa3_amd import _lib

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "geoa3_hip.h")).read()
    dbg = open(os.path.join(REPO, "include", "geoa3_hip_debug.h")).read()
    decls = lambda text: set(re.findall(r"\b(geoa3_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    product, diagnostics = decls(hdr), decls(dbg)
    assert product and diagnostics, "no declarations parsed"
    # the product ABI promises "no global state": timers and single-kernel hooks live in the debug header only
    assert not [n for n in product if "debug" in n or "profile" in n]
    assert all("debug" in n or "profile" in n for n in diagnostics), diagnostics
    declared = product | diagnostics
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.geoa3_version() >= 100
    assert lib.geoa3_strerror(-1) == b"invalid argument"


def test_library_exports_no_mutable_global():
    """`no global state`: the dynamic symbol table holds functions only (no B/D/C data objects such as a tuning
    variable another caller could flip under a running loop)."""
    import __graft_entry__
    __graft_entry__.build()
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    data = [ln for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "BbDdCcGgSs"
            and not ln.split()[-1].startswith(("__hip", "_edata", "_end", "__bss_start"))]
    assert not data, data


def test_struct_layouts_match_header():
    """ctypes mirrors have one field per struct member, in order."""
    hdr = open(os.path.join(REPO, "include", "geoa3_hip.h")).read()

    def members(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part.strip())[0])
        return names

    assert members("geoa3_geo_args") == [f[0] for f in _lib.GeoArgs._fields_]
    assert members("geoa3_tnet_weights") == [f[0] for f in _lib.TnetWeights._fields_]
    assert members("geoa3_pointnet_weights") == [f[0] for f in _lib.PointNetWeights._fields_]
    assert members("geoa3_attack_state") == [f[0] for f in _lib.AttackState._fields_]
    assert members("geoa3_pn2ssg_weights") == [f[0] for f in _lib.Pn2SsgWeights._fields_]
    assert members("geoa3_sa1_weights") == [f[0] for f in _lib.Sa1Weights._fields_]


def test_abi_version_is_checked():
    """include/geoa3_hip.h, the built library and the ctypes mirror agree on GEOA3_ABI_VERSION; load() refuses any other."""
    hdr = open(os.path.join(REPO, "include", "geoa3_hip.h")).read()
    ver = int(re.search(r"#define GEOA3_ABI_VERSION (\d+)", hdr).group(1))
    assert ver == _lib.ABI_VERSION == _lib.load().geoa3_version()
    import unittest.mock as mock
    with mock.patch.object(_lib, "_lib", None), mock.patch.object(_lib, "ABI_VERSION", ver + 1):
        with pytest.raises(_lib.Geoa3Error, match="ABI version"):
            _lib.load()


def test_backward_chain_kernel_holds_no_packed_fp32():
    """NOTEBOOK 5a: with the packed-FP32 instructions the SLP vectoriser forms (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32),
    conv_bwd_chain_kernel computes wrong values in lanes 48-63 when two of its wavefronts share a SIMD.  The file is compiled
    with geoa3_amd.build.FILE_FLAGS (-fno-slp-vectorize): the kernel's ISA, produced with exactly the build's flags, must not
    contain them (a lost flag would bring the fault back at ~1e-3 of the launches, below what a short GPU test sees)."""
    import tempfile
    from geoa3_amd import build as B
    src = os.path.join(REPO, "geoa3_amd", "csrc", "pointnet_conv_chain.hip")
    flags = [f for f in B.FLAGS if f != "-fPIC"] + B.FILE_FLAGS["pointnet_conv_chain.hip"]
    assert "-fno-slp-vectorize" in flags
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "chain.s")
        subprocess.run([B._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", out, src], check=True, capture_output=True)
        asm = open(out).read()
    start = asm.index("conv_bwd_chain_kernel")
    body = asm[asm.index(":", asm.index("_ZN12_GLOBAL__N_121conv_bwd_chain_kernelE16ConvBwdChainArgs:")):]
    body = body[: body.index("s_endpgm")]
    assert start >= 0 and "v_mfma_f32_32x32x16_f16" in body
    assert not re.search(r"v_pk_(mul|fma|add)_f32", body)
    assert re.search(r"\.amdhsa_next_free_vgpr\s+2\d\d", asm)       # (two waves per SIMD: more than 128 registers is fine)


def test_sampler_and_side_queue_kernels_hold_no_packed_fp32():
    """NOTEBOOK 5a, second sighting: the farthest-point sampler with an SLP-packed distance update is exact alone and wrong in
    1.4e-3 of its rounds beside sa1_fwd_kernel (tools/ub/pk_fp32_coresidency.hip).  The kernels that run on the PointNet++
    side queue live in pointnet2_ops.hip / pointnet2_net.hip: with the build's flags none of them may hold a
    compiler-formed packed-FP32 instruction."""
    import tempfile
    from geoa3_amd import build as B
    side = {"pointnet2_ops.hip": ["fps_kernel", "ball_query_wave_kernel"],
            "pointnet2_net.hip": ["gather_rows3_kernel", "affine3_kernel", "affine3_grad_kernel", "scatter_rows3_kernel",
                                  "add_inplace_kernel"]}
    for name, kernels in side.items():
        src = os.path.join(REPO, "geoa3_amd", "csrc", name)
        flags = [f for f in B.FLAGS if f != "-fPIC"] + B.FILE_FLAGS[name]
        assert "-fno-slp-vectorize" in flags
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.run([B._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", out, src], check=True, capture_output=True)
            asm = open(out).read()
        for k in kernels:
            found = list(re.finditer(r"^(_Z\w*%s\w*):" % k, asm, re.M))
            assert found, (name, k)
            for m in found:      # every instance of a template
                body = asm[m.end():]
                body = body[:body.index("s_endpgm")]
                assert not re.search(r"v_pk_(?:mul|fma|add)_f32", body), (name, m.group(1))
    assert "fps_kernel" in B.ISA_GUARDS["pointnet2_ops.hip"][0]


def test_build_refuses_packed_fp32_with_op_sel_in_every_file():
    """NOTEBOOK 5a: the stand-alone reproducer (tools/ub/pk_neg_mfma_min.hip) pins the fault on packed FP32 arithmetic with an
    op_sel bit -- a low result reading the HIGH half of a source pair returns that operand as zero in lanes 48-63 beside
    wavefronts that mix vector and matrix instructions.  The build disassembles EVERY file after compiling it and refuses
    that form; here: the pattern itself, and that the file which keeps the SLP vectoriser (whose T-Net transform used to be
    packed that way) and the sampler's file are free of it with the build's flags."""
    import tempfile
    from geoa3_amd import build as B
    pat = re.compile(B.ISA_GUARD_ALL)
    assert pat.search("\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n")
    assert pat.search("\tv_pk_fma_f32 v[2:3], v[4:5], s[6:7], v[8:9] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n")
    assert pat.search("\tv_pk_mul_f32 v[4:5], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]\n")
    assert not pat.search("\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0] neg_lo:[0,1]\n\tv_pk_mov_b32 v[6:7], v[4:5], v[4:5] op_sel:[1,0]\n")
    for name in ("pointnet_gemm.hip", "pointnet2_ops.hip", "geom_slab.hip"):
        src = os.path.join(REPO, "geoa3_amd", "csrc", name)
        flags = [f for f in B.FLAGS if f != "-fPIC"] + B.FILE_FLAGS.get(name, [])
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.run([B._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", out, src], check=True, capture_output=True)
            asm = open(out).read()
        assert not pat.search(asm), name
        if name != "pointnet2_ops.hip":
            assert "v_pk_" in asm, name     # (their packed multiplies / adds without op_sel stay)


def test_one_instruction_reproducer_builds():
    """tools/ub/pk_neg_mfma_min.hip (NOTEBOOK 5a: the stand-alone reproducer of the op_sel fault) needs nothing but hipcc: it
    must keep compiling for gfx950, and its victim must hold the instruction it is about."""
    import tempfile
    from geoa3_amd import build as B
    src = os.path.join(REPO, "tools", "ub", "pk_neg_mfma_min.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r.s")
        subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-O3", "-w", "-S", "--cuda-device-only", "-o", out, src],
                       check=True, capture_output=True)
        asm = open(out).read()
    assert re.search(r"v_pk_add_f32 v\[\d+:\d+\], v\[\d+:\d+\], v\[100:101\] op_sel:\[0,1\]", asm)
    assert "v_mfma_f32_32x32x16_f16" in asm


def test_isa_guard_refuses_the_faulty_instruction_form():
    """The build's ISA guard (packed FP32 arithmetic with op_sel on a source: wrong values in lanes 48-63 beside matrix-core
    wavefronts, NOTEBOOK 5a) is held to a source that MUST be refused, compiled with the library's own flags by the hipcc
    of this machine: a compiler that forms or prints the instruction differently fails here instead of shipping."""
    from geoa3_amd import build as B
    msg = B.isa_guard_selftest()
    assert "op_sel" in msg and "guard_selftest.hip" in msg
    import re
    # the pattern reads the operand-select list only (op_sel_hi, v_pk_mov_b32 and an all-zero list are the clean forms)
    assert re.search(B.ISA_GUARD_ALL, "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]")
    assert re.search(B.ISA_GUARD_ALL, "v_pk_mul_f32_e64 v[0:1], v[2:3], s[4:5] op_sel = [0, 1]")
    assert not re.search(B.ISA_GUARD_ALL, "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0]")
    assert not re.search(B.ISA_GUARD_ALL, "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,0] op_sel_hi:[1,1,1]")
    assert not re.search(B.ISA_GUARD_ALL, "v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]")
    # the compiler's identity and the guard's patterns are part of every object's stamp
    d0 = B._digest(B.sources()[0])
    old = B.ISA_GUARD_ALL
    try:
        B.ISA_GUARD_ALL = old + "x"
        assert B._digest(B.sources()[0]) != d0
    finally:
        B.ISA_GUARD_ALL = old
    assert B.hipcc_version() and B._digest(B.sources()[0]) == d0
