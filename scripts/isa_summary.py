"""Compiler-reported resources of every kernel of libgvk.so (hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed) and
the instruction mix of the shipped dim-128 training kernel (disassembly of the code object): VGPRs, SGPRs, scratch,
occupancy, LDS; VALU / VMEM / SALU / DPP counts.

    python scripts/isa_summary.py > profiles/r3/isa_summary.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "graphvite_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [o.replace("(anonymous namespace)::", "") for o in out]


def main():
    with tempfile.TemporaryDirectory() as tmp:
        flags = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
        rows = []
        for source in ("gvk_pairs.hip", "gvk_chains.hip", "gvk_samplers.hip", "gvk_group.hip"):
            run = subprocess.run(flags + ["-c", os.path.join(CSRC, source), "-o", os.path.join(tmp, source + ".o"),
                                          "-Rpass-analysis=kernel-resource-usage", "--save-temps=obj"], capture_output=True, text=True, cwd=tmp)
            blocks = re.split(r"Function Name: ", run.stderr)[1:]
            for b in blocks:
                name = b.split()[0]

                def field(key):
                    m = re.search(re.escape(key) + r": (\d+)", b)
                    return int(m.group(1)) if m else -1
                rows.append((name, field("VGPRs"), field("AGPRs"), field("TotalSGPRs"), field("ScratchSize [bytes/lane]"),
                             field("Occupancy [waves/SIMD]"), field("LDS Size [bytes/block]")))
        names = demangle([r[0] for r in rows])
        print("Resources per kernel (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, -O3); occupancy in waves per SIMD")
        print("%-100s %5s %5s %5s %8s %5s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
        shown = sorted(set((n.split("(")[0][:100],) + r[1:] for n, r in zip(names, rows)))
        for row in shown:
            print("%-100s %5d %5d %5d %8d %5d %6d" % row)
        # instruction mix of the shipped kernels from the device assembly
        asm = [os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f and "gvk_pairs" in f]
        if not asm:
            return
        text = open(asm[0]).read()
        print("\nInstruction mix (static, per wavefront pass through the kernel body; device assembly of gvk_pairs.hip)")
        for want in ("train_kernelILi128ELi16ELi0ELi1ELi1ELi4E", "train_runs_kernelILi128ELi16ELi0ELi1ELi1ELi4E", "probe_rows_kernelILi128ELi16E",
                     "train_kernelILi96ELi8ELi0ELi1ELi1ELi4E", "train_kernelILi64ELi16ELi0ELi1ELi1ELi4E"):
            m = re.search(r"^(_ZN\S*" + want + r"\S*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
            if not m:
                continue
            body = [line.strip() for line in m.group(2).splitlines() if line.strip() and not line.strip().startswith((";", ".", "//"))]
            ops = [line.split()[0] for line in body if re.match(r"^[a-z_0-9]+(\s|$)", line)]
            count = lambda pred: sum(1 for o in ops if pred(o))  # noqa: E731
            dpp = sum(1 for line in body if "dpp" in line or "quad_perm" in line or "row_" in line)
            print("%-60s instructions %4d | VALU %4d (of them DPP %d, v_mul_hi/v_mul_lo (Philox) %d, transcendental %d) | "
                  "global loads %d stores %d | SALU %d | s_waitcnt %d" % (
                      demangle([m.group(1)])[0].split("(")[0][:60], len(ops), count(lambda o: o.startswith("v_")), dpp,
                      count(lambda o: o.startswith(("v_mul_hi_u32", "v_mul_lo_u32", "v_mad_u64_u32"))),
                      count(lambda o: o.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq"))),
                      count(lambda o: o.startswith("global_load")), count(lambda o: o.startswith("global_store")),
                      count(lambda o: o.startswith("s_") and not o.startswith("s_waitcnt")), count(lambda o: o.startswith("s_waitcnt"))))


if __name__ == "__main__":
    main()
