#!/usr/bin/env python3
"""Print the instruction stream of ONE kernel from a gfx950 assembly listing (hipcc -S --cuda-device-only), labels and
comments stripped, so that two builds of the same kernel can be diffed:
    tools/isa_kernel.py build.s <substring of the mangled kernel name> [--stats]
--stats prints the instruction mix (MFMA / VALU / SALU / LDS / VMEM counts) and the register counts instead."""
import re
import sys


def kernel_body(text, name):
    m = re.search(r"^(\S*%s\S*):\s*(;.*)?$" % re.escape(name), text, re.M)
    if not m:
        raise SystemExit(f"no kernel matching {name!r}")
    sym = m.group(1)
    end = text.index(".Lfunc_end", m.end())
    return sym, text[m.end():end]


def main():
    path, name = sys.argv[1], sys.argv[2]
    text = open(path).read()
    sym, body = kernel_body(text, name)
    ins = []
    for line in body.splitlines():
        line = line.split(";")[0].rstrip()
        if not line or line.lstrip().startswith(".") or line.endswith(":"):
            continue
        ins.append(re.sub(r"\.LBB\d+_\d+", ".L", line.strip()))
    if "--stats" in sys.argv:
        kinds = {"mfma": 0, "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "other": 0}
        for i in ins:
            op = i.split()[0]
            k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_")
                 else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu" if op.startswith("s_")
                 else "other")
            kinds[k] += 1
        regs = re.search(r"\.name:\s+%s.*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)" % re.escape(sym), text, re.S)
        print(sym, len(ins), kinds, "sgpr/vgpr", regs.groups() if regs else "?")
    else:
        print("\n".join(ins))


if __name__ == "__main__":
    main()
