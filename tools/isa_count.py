"""Dynamic instruction count of one kernel from hipcc -S output: line ranges weighted by loop trip counts.

    python tools/isa_count.py file.s KERNEL_SUBSTRING [start:end:weight ...]   (line numbers relative to the kernel's label)

Without ranges: prints the labels / branches (to find the loops) and the static histogram by class."""
import collections
import re
import sys


def kernel_lines(path, name):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(name), l))
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or ".Lfunc_end" in lines[i])
    return lines[start:end]


def cls(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_sleep") or op.startswith("s_setprio") or op.startswith("s_sched"):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    L = kernel_lines(sys.argv[1], sys.argv[2])
    ranges = [tuple(float(x) if "." in x else int(x) for x in a.split(":")) for a in sys.argv[3:]]
    if not ranges:
        for i, l in enumerate(L):
            if re.match(r"^\.LBB", l) or "s_cbranch" in l or "s_branch" in l:
                print(i, l.strip()[:110])
        ranges = [(0, len(L), 1)]
    tot = collections.Counter()
    ops = collections.Counter()
    for a, b, w in ranges:
        for l in L[a:b]:
            m = re.match(r"^\s+([a-z_0-9]+)", l)
            if not m or l.strip().startswith(";"):
                continue
            op = m.group(1)
            tot[cls(op)] += w
            ops[op] += w
    print({k: round(v, 1) for k, v in tot.items()})
    for op, n in ops.most_common(45):
        print("%8.1f %s" % (n, op))


if __name__ == "__main__":
    main()
