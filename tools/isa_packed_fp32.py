#!/usr/bin/env python3
"""Which kernels of the library hold packed-FP32 instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) with an SGPR
operand, and at how many waves per SIMD (NOTEBOOK 5a: in conv_bwd_chain_kernel such instructions computed wrong values in
lanes 48-63 with two waves per SIMD; that file is compiled without the SLP vectoriser).  Compiles every csrc/*.hip to ISA
with the build's own flags (CPU only).    python tools/isa_packed_fp32.py > profiles/roundN_packed_fp32_kernels.txt"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from geoa3_amd import build as B  # noqa: E402


def main():
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for src in B.sources():
            name = os.path.basename(src)
            flags = [f for f in B.FLAGS if f != "-fPIC"] + B.FILE_FLAGS.get(name, [])
            out = os.path.join(d, name + ".s")
            r = subprocess.run([B._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
            if r.returncode != 0:
                print("!! %s: hipcc -S failed" % name)
                continue
            asm = open(out).read()
            vg = dict(re.findall(r"\.amdhsa_kernel (\S+)[\s\S]*?\.amdhsa_next_free_vgpr (\d+)", asm))
            for m in re.finditer(r"^(_Z\S+):\s*;\s*@\S+\n([\s\S]*?)s_endpgm", asm, re.M):
                kern, body = m.group(1), m.group(2)
                pk = re.findall(r"v_pk_(?:mul|fma|add)_f32 [^\n]*", body)
                pk_s = [l for l in pk if re.search(r"\bs\[\d+:\d+\]", l)]
                pk_n = [l for l in pk if re.search(r"op_sel:\[[^\]]*1", l)]   # (the form the fault is pinned on: NOTEBOOK 5a)
                if not pk:
                    continue
                nv = int(vg.get(kern, 0))
                occ = 8 if nv <= 64 else 512 // (((nv + 7) // 8) * 8) if nv else 0
                dem = subprocess.run(["c++filt", kern], capture_output=True, text=True).stdout.strip() or kern
                dem = re.sub(r"\(anonymous namespace\)::", "", dem)
                rows.append((name, re.sub(r"\(.*", "", dem)[:60], len(pk), len(pk_s), len(pk_n), nv, occ))
    print("%-28s %-60s %6s %10s %9s %6s %s" % ("file", "kernel", "v_pk_*", "with SGPR", "op_sel", "VGPRs", "waves/SIMD (register limit)"))
    for r in sorted(rows, key=lambda r: (-r[3], r[0], r[1])):
        print("%-28s %-60s %6d %10d %9d %6d %d" % r)
    print("\n%d kernels hold packed-FP32 instructions, %d of them with SGPR-pair operands at >= 2 waves per SIMD"
          % (len(rows), sum(1 for r in rows if r[3] and r[6] >= 2)))
    print("%d packed-FP32 instructions carry an op_sel bit (a low result from the high half of a pair; the build refuses any: "
          "build.py ISA_GUARD_ALL)"
          % sum(r[4] for r in rows))


if __name__ == "__main__":
    main()
