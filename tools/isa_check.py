"""Checks and statistics on the EMITTED gfx950 ISA of liblgr_hip.so's translation units (llvm-objdump of the offload bundle inside
the in-tree .o files; no GPU needed).

    python tools/isa_check.py hazard          # the XDL-write -> VALU-read wait states in front of fpfh_mfma_kernel's epilogue
    python tools/isa_check.py blocks KERNEL   # per basic block of a kernel: VALU / MFMA / LDS / VMEM / SALU instruction counts

`hazard` is run by __graft_entry__.build(): hipcc's hazard recogniser does not carry the "MFMA result -> v_accvgpr_read" wait states
across the branches of the run loop of fpfh_mfma_kernel ("A hazard worth knowing"); the kernel therefore issues
`s_nop 15; s_nop 7` (24 wait states >= the 18 an 8-pass MFMA needs) in front of its first accumulator read.  A compiler bump that
drops or moves them would make rows a few ulp wrong on inputs only some builds hit -- this check fails the build instead.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lidar-global-registration_amd", "csrc")
OBJDUMP = next((p for p in (os.path.join(b, "lib", "llvm", "bin", "llvm-objdump") for b in (os.environ.get("ROCM_PATH"), "/opt/rocm") if b) if os.path.exists(p)), None) or __import__("shutil").which("llvm-objdump") or "llvm-objdump"


def disassemble(obj):
    """-> text of `llvm-objdump -d` of the gfx950 code object bundled in a hipcc .o"""
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), local)
        subprocess.check_call([OBJDUMP, "--offloading", local], cwd=td, stdout=subprocess.DEVNULL)
        cos = [f for f in os.listdir(td) if "amdgcn" in f and "gfx950" in f]
        if not cos:
            raise RuntimeError("no gfx950 code object in " + obj)
        return subprocess.check_output([OBJDUMP, "-d", os.path.join(td, cos[0])], text=True)


def functions(text):
    """{mangled name: [(address, mnemonic, operands)]}"""
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def find(funcs, needle):
    hits = [k for k in funcs if needle in k]
    if not hits:
        raise KeyError(needle)
    return hits


def check_hazard():
    funcs = functions(disassemble(os.path.join(CSRC, "lgr_features.o")))
    names = find(funcs, "fpfh_mfma_kernel")
    for name in names:
        ins = funcs[name]
        first = next((i for i, x in enumerate(ins) if x[1].startswith("v_accvgpr_read")), None)
        if first is None:
            # with a small register budget the accumulators live in architectural VGPRs and the epilogue reads them
            # directly: the read is then the first instruction behind the s_nop run that follows the last v_mfma
            last_mfma = max((i for i, x in enumerate(ins) if x[1].startswith("v_mfma")), default=None)
            if last_mfma is None:
                raise SystemExit("isa_check: %s has no v_mfma -- the check no longer matches the kernel" % name)
            nop = next((i for i in range(last_mfma + 1, len(ins)) if ins[i][1] == "s_nop" and int(ins[i][2], 0) >= 7), None)
            if nop is None:
                raise SystemExit("isa_check: %s has no s_nop run behind its last v_mfma -- the epilogue padding was dropped" % name)
            first = next(i for i in range(nop, len(ins)) if ins[i][1] != "s_nop")
        if not any(x[1].startswith("v_mfma") for x in ins[:first]):
            raise SystemExit("isa_check: no v_mfma in front of the first accumulator read of " + name)
        waits = 0
        for x in reversed(ins[max(0, first - 8):first]):     # wait states issued right in front of the read, inside its basic block
            if x[1] == "s_nop":
                waits += int(x[2], 0) + 1
            elif x[1].startswith(("v_mfma", "s_branch", "s_cbranch", "s_endpgm")):
                break
            else:
                waits += 1                                    # any other instruction between the s_nop pair and the read issues for >= 1 cycle
        if waits < 18:
            raise SystemExit("isa_check: only %d wait states in front of the first v_accvgpr_read of %s (need >= 18: XDL write -> VALU read "
                             "of an 8-pass MFMA); the s_nop pair of lgr_features.hip's epilogue was dropped or moved by the compiler" % (waits, name))
        print("isa_check hazard: %s: %d wait states in front of the first accumulator read -- ok" % (name.split("fpfh_mfma_kernel")[0] + "fpfh_mfma_kernel", waits))


def classify(mn):
    if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
        return "mfma"
    if mn.startswith("v_"):
        return "valu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith("s_waitcnt"):
        return "wait"
    if mn.startswith("s_barrier"):
        return "barrier"
    if mn.startswith("s_load") or mn.startswith("s_buffer_load"):
        return "smem"
    return "salu"


def blocks(obj, needle):
    funcs = functions(disassemble(obj))
    for name in find(funcs, needle):
        ins = funcs[name]
        base = ins[0][0]
        addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
        leaders = {0}
        for i, (a, mn, op) in enumerate(ins):
            if mn.startswith(("s_cbranch", "s_branch")):
                if i + 1 < len(ins):
                    leaders.add(i + 1)
                m = re.search(r"\+0x([0-9a-f]+)>", op)     # objdump prints the target as <func+0xoff> in the comment column only
            if mn.startswith(("s_cbranch", "s_branch")):
                off = int(op.split()[0])
                off = off - 65536 if off >= 32768 else off
                tgt = a + 4 + 4 * off
                if tgt in addr_index:
                    leaders.add(addr_index[tgt])
        order = sorted(leaders)
        print("%s: %d instructions, %d basic blocks" % (name, len(ins), len(order)))
        print("%8s %6s | %5s %5s %4s %5s %5s %5s %4s | ends with" % ("offset", "instr", "valu", "mfma", "lds", "vmem", "salu", "wait", "bar"))
        tot = {}
        for bi, st in enumerate(order):
            en = order[bi + 1] if bi + 1 < len(order) else len(ins)
            c = {}
            for (_, mn, _) in ins[st:en]:
                k = classify(mn)
                c[k] = c.get(k, 0) + 1
                tot[k] = tot.get(k, 0) + 1
            last = ins[en - 1]
            if c.get("mfma", 0) or (en - st) >= 24:
                print("%#8x %6d | %5d %5d %4d %5d %5d %5d %4d | %s %s" % (ins[st][0] - base, en - st, c.get("valu", 0), c.get("mfma", 0), c.get("lds", 0), c.get("vmem", 0),
                                                                          c.get("salu", 0) + c.get("smem", 0), c.get("wait", 0), c.get("barrier", 0), last[1], last[2][:40]))
        print("   total %6d | %5d %5d %4d %5d %5d %5d %4d | valu per mfma %.2f" % (len(ins), tot.get("valu", 0), tot.get("mfma", 0), tot.get("lds", 0), tot.get("vmem", 0),
                                                                                tot.get("salu", 0) + tot.get("smem", 0), tot.get("wait", 0), tot.get("barrier", 0),
                                                                                tot.get("valu", 0) / max(1, tot.get("mfma", 0))))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "hazard":
        check_hazard()
    elif len(sys.argv) >= 3 and sys.argv[1] == "blocks":
        obj = sys.argv[3] if len(sys.argv) > 3 else os.path.join(CSRC, "lgr_match.o")
        blocks(obj, sys.argv[2])
    else:
        raise SystemExit(__doc__)
