"""Software-wait-state audit of the EMITTED gfx950 ISA (no GPU needed), across basic-block boundaries.

    python tools/isa_hazards.py [KERNEL-SUBSTRING ...]     # default: every kernel of every csrc/*.o
    python tools/isa_hazards.py --resources                 # scratch bytes / spilled registers per kernel (code-object metadata)
    python tools/isa_hazards.py --gate                      # fail when a gated hot kernel carries scratch or spills, or any rule is violated

The hardware does not interlock the cases below: the producer's result is written by the LAST of a wave64 instruction's four
16-lane passes, and a consumer that issues too early sees the old value in the late lanes.  hipcc's hazard recogniser pads them;
this script re-derives the padding from the disassembly so that a compiler change (or an inline-asm block, which the recogniser does
not look into) cannot silently drop one.  Rules (CDNA3 ISA guide section 4.5 "manually inserted wait states" + LLVM
GCNHazardRecognizer for gfx940/gfx950); N = wait states required between producer and consumer (an instruction = 1, `s_nop k` = k + 1):

    R1  trans op (v_rcp/v_rsq/v_sqrt/v_exp/v_log/v_sin/v_cos) writes a VGPR   -> non-trans VALU reads it          1
    R2  VALU writes an SGPR / VCC (v_cmp, v_div_scale, carry out, v_readlane)  -> VALU reads that SGPR / VCC        2
    R3  VALU writes VCC                                                        -> v_div_fmas                        4
    R4  VALU writes an SGPR / VCC                                              -> v_readlane / v_writelane lane sel 4
    R5  VALU writes an SGPR                                                    -> VMEM reads that SGPR              5
    R6  VALU writes EXEC (v_cmpx)                                              -> DPP op 5, v_readlane & co.        4
    R7  VALU writes a VGPR                                                     -> DPP op reads that VGPR            2
    R8  VALU writes a VGPR                                                     -> v_readlane/v_readfirstlane src    1
    R9  SALU writes M0                                                         -> LDS-DMA / s_movrel / GWS          1
    R10 MFMA writes VGPRs / AGPRs                                              -> VALU / VMEM / LDS reads them, v_accvgpr_read     passes + 2

The walk goes backwards from every consumer through ALL predecessor blocks until the required number of wait states has been seen on
that path.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lidar-global-registration_amd", "csrc")


def _tool(name):
    for base in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if base and os.path.exists(os.path.join(base, "lib", "llvm", "bin", name)):
            return os.path.join(base, "lib", "llvm", "bin", name)
    import shutil
    p = shutil.which(name)
    if not p:
        raise SystemExit("isa_hazards: %s not found (set ROCM_PATH)" % name)
    return p


def code_object(obj, td):
    local = os.path.join(td, os.path.basename(obj))
    if not os.path.exists(local):
        os.symlink(os.path.abspath(obj), local)
    subprocess.check_call([_tool("llvm-objdump"), "--offloading", local], cwd=td, stdout=subprocess.DEVNULL)
    cos = [f for f in os.listdir(td) if f.startswith(os.path.basename(obj)) and "amdgcn" in f and "gfx950" in f]
    return os.path.join(td, cos[0]) if cos else None


def disassemble(co):
    return subprocess.check_output([_tool("llvm-objdump"), "-d", co], text=True)


def functions(text):
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


# ---------------------------------------------------------------------------------------------------------------- operands
REG = re.compile(r"(?<![\w.])(-?\|?)(v|s|a|acc)(?:(\d+)|\[(\d+):(\d+)\])(?![\w])")


def regs_of(tok):
    """register units named in one operand token: {('v', 3), ('s', 0), ('s', 1), ('vcc', 0), ...}"""
    out = set()
    t = tok.strip().lower()
    for m in REG.finditer(t):
        kind = "a" if m.group(2) in ("a", "acc") else m.group(2)
        if m.group(3) is not None:
            out.add((kind, int(m.group(3))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((kind, i))
    for name in ("vcc", "exec", "m0", "scc"):
        if re.search(r"(?<![\w])%s(_lo|_hi)?(?![\w])" % name, t):
            out.add((name, 0))
    return out


def split_ops(op):
    """top-level comma split; trailing modifiers (row_shr:1, offset:16, neg_lo:[0,1] ...) stay with the last token"""
    toks, depth, cur = [], 0, ""
    for ch in op:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            toks.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        toks.append(cur.strip())
    return toks


TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
TRANS_NOT = ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")
CARRY = ("v_add_co_", "v_sub_co_", "v_subrev_co_", "v_addc_co_", "v_subb_co_", "v_subbrev_co_", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale_")
CARRY_IN = ("v_addc_co_", "v_subb_co_", "v_subbrev_co_")
LANE_OPS = ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32")


def is_valu(mn):
    return mn.startswith("v_")


def is_mfma(mn):
    return mn.startswith("v_mfma") or mn.startswith("v_smfmac")


def is_trans(mn):
    return mn.startswith(TRANS) and not mn.startswith(TRANS_NOT)


def is_vmem(mn):
    return mn.startswith(("global_", "buffer_", "flat_", "scratch_", "tbuffer_"))


def is_dpp(mn, op):
    return "_dpp" in mn or re.search(r"\b(quad_perm|row_shl|row_shr|row_ror|wave_shl|wave_shr|wave_rol|wave_ror|row_mirror|row_half_mirror|row_bcast|row_newbcast)\b", op) is not None


def mfma_passes(mn):
    m = re.search(r"_(\d+)x(\d+)x(\d+)", mn)
    if not m:
        return 16
    a, b, k = int(m.group(1)), int(m.group(2)), int(m.group(3))
    if "f64" in mn:
        return 16
    if a == 32:
        return 8 if ("f16" in mn or "bf16" in mn or "f8" in mn or "i8" in mn) and k >= 16 else 16
    if a == 16:
        return 4 if ("f16" in mn or "bf16" in mn or "f8" in mn or "i8" in mn) and k >= 32 else 8
    return 2 if a == 4 else 8


class Ins:
    __slots__ = ("addr", "mn", "op", "defs", "uses", "toks")

    def __init__(self, addr, mn, op):
        self.addr, self.mn, self.op = addr, mn, op
        toks = split_ops(op)
        self.toks = toks
        defs, uses = set(), set()
        if is_valu(mn):
            if mn.startswith("v_cmpx"):
                defs.add(("exec", 0))
                if mn.endswith("_e64") and toks:
                    defs |= regs_of(toks[0]); rest = toks[1:]
                else:
                    rest = toks
                for t in rest:
                    uses |= regs_of(t)
            elif mn.startswith("v_cmp"):
                if mn.endswith("_e64") or (toks and regs_of(toks[0]) and not toks[0].lstrip("-|").startswith("v") and len(toks) == 3):
                    defs |= regs_of(toks[0]); rest = toks[1:]
                else:
                    defs.add(("vcc", 0)); rest = toks[1:] if toks and toks[0].strip() == "vcc" else toks
                for t in rest:
                    uses |= regs_of(t)
            elif mn.startswith("v_nop") or mn.startswith("v_accvgpr_write"):
                if toks:
                    defs |= regs_of(toks[0])
                for t in toks[1:]:
                    uses |= regs_of(t)
            else:
                if toks:
                    defs |= regs_of(toks[0])
                rest = toks[1:]
                if mn.startswith(CARRY) and rest:
                    defs |= regs_of(rest[0]); rest = rest[1:]
                for t in rest:
                    uses |= regs_of(t)
                if mn.startswith("v_div_fmas"):
                    uses.add(("vcc", 0))
                if mn.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_dot4c", "v_dot8c")) or mn.startswith("v_writelane"):
                    uses |= regs_of(toks[0])       # the destination is also a source
                if is_mfma(mn) and len(toks) >= 4:
                    pass
        elif mn.startswith("s_"):
            if mn.startswith(("s_cmp", "s_bitcmp")):
                defs.add(("scc", 0))
                for t in toks:
                    uses |= regs_of(t)
            elif mn.startswith(("s_nop", "s_waitcnt", "s_barrier", "s_endpgm", "s_branch", "s_cbranch", "s_sleep", "s_sendmsg", "s_setprio", "s_trap", "s_icache", "s_dcache")):
                pass
            elif mn.startswith(("s_store", "s_buffer_store")):
                for t in toks:
                    uses |= regs_of(t)
            else:
                if toks:
                    defs |= regs_of(toks[0])
                for t in toks[1:]:
                    uses |= regs_of(t)
                if "saveexec" in mn:
                    defs.add(("exec", 0)); uses.add(("exec", 0))
        elif mn.startswith("ds_") or is_vmem(mn):
            load = ("load" in mn or "read" in mn or "_rtn" in mn or "atomic" in mn and "glc" in op or "sc0" in op and "atomic" in mn)
            if "lds" in op.split("//")[0].split() or mn.endswith("_lds"):
                load = False
            if load and toks:
                defs |= regs_of(toks[0])
                for t in toks[1:]:
                    uses |= regs_of(t)
            else:
                for t in toks:
                    uses |= regs_of(t)
        self.defs, self.uses = defs, uses


def build_cfg(ins):
    index = {x.addr: i for i, x in enumerate(ins)}
    leaders = {0}
    edges = []   # (from instruction index, to instruction index)
    for i, x in enumerate(ins):
        if x.mn.startswith(("s_cbranch", "s_branch")):
            off = int(x.op.split()[0])
            off = off - 65536 if off >= 32768 else off
            tgt = x.addr + 4 + 4 * off
            if tgt in index:
                leaders.add(index[tgt]); edges.append((i, index[tgt]))
            if i + 1 < len(ins):
                leaders.add(i + 1)
                if x.mn.startswith("s_cbranch"):
                    edges.append((i, i + 1))
        elif x.mn.startswith(("s_endpgm", "s_setpc", "s_swappc")):
            if i + 1 < len(ins):
                leaders.add(i + 1)
    order = sorted(leaders)
    block_of = {}
    for bi, st in enumerate(order):
        en = order[bi + 1] if bi + 1 < len(order) else len(ins)
        for i in range(st, en):
            block_of[i] = bi
    preds = {bi: set() for bi in range(len(order))}
    for bi, st in enumerate(order):            # fall-through edges
        if st > 0:
            prev = ins[st - 1]
            if not prev.mn.startswith(("s_branch", "s_endpgm", "s_setpc", "s_swappc")) and not prev.mn.startswith("s_cbranch"):
                preds[bi].add(block_of[st - 1])
    for (a, b) in edges:
        preds[block_of[b]].add(block_of[a])
    ends = {bi: (order[bi + 1] if bi + 1 < len(order) else len(ins)) - 1 for bi in range(len(order))}
    return order, block_of, preds, ends


def wait_states(x):
    if x.mn == "s_nop":
        return int(x.op.split()[0], 0) + 1
    return 1


def scan(ins, name, verbose=True):
    order, block_of, preds, ends = build_cfg(ins)
    found = []

    def walk_back(i, need, is_producer, seen, acc, bi=None):
        """every path backwards from instruction i (exclusive) of block bi; acc = wait states already between the consumer and position i"""
        if bi is None:
            bi = block_of[i]
        j = i - 1
        while acc < need:
            if j < order[bi]:
                for pb in preds[bi]:
                    key = (pb, acc)
                    if key in seen:
                        continue
                    seen.add(key)
                    walk_back(ends[pb] + 1, need, is_producer, seen, acc, pb)
                return
            p = ins[j]
            if is_producer(p):
                found_at.append((p, acc))
                return
            acc += wait_states(p)
            j -= 1

    for i, c in enumerate(ins):
        rules = []
        mn = c.mn
        if is_valu(mn) and not is_mfma(mn):
            vuse = {r for r in c.uses if r[0] == "v"}
            suse = {r for r in c.uses if r[0] in ("s", "vcc")}
            if not is_trans(mn) and vuse:
                rules.append(("R1 trans->VALU", 1, lambda p, vuse=vuse: is_trans(p.mn) and p.defs & vuse))
            if suse:
                rules.append(("R2 VALU sgpr->VALU read", 2, lambda p, suse=suse: is_valu(p.mn) and p.defs & suse))
            if mn.startswith("v_div_fmas"):
                rules.append(("R3 VALU vcc->v_div_fmas", 4, lambda p: is_valu(p.mn) and ("vcc", 0) in p.defs))
            if mn.startswith(("v_readlane", "v_writelane")) and len(c.toks) >= 3:
                sel = regs_of(c.toks[2])
                if sel:
                    rules.append(("R4 VALU sgpr->lane select", 4, lambda p, sel=sel: is_valu(p.mn) and p.defs & sel))
            if mn.startswith(LANE_OPS):
                rules.append(("R6 VALU exec->lane op", 4, lambda p: is_valu(p.mn) and ("exec", 0) in p.defs))
                src = regs_of(c.toks[1]) if len(c.toks) > 1 else set()
                src = {r for r in src if r[0] == "v"}
                if src and not mn.startswith("v_writelane"):
                    rules.append(("R8 VALU vgpr->readlane", 1, lambda p, src=src: is_valu(p.mn) and p.defs & src))
            if is_dpp(mn, c.op):
                rules.append(("R6 VALU exec->DPP", 5, lambda p: is_valu(p.mn) and ("exec", 0) in p.defs))
                if vuse:
                    rules.append(("R7 VALU vgpr->DPP", 2, lambda p, vuse=vuse: is_valu(p.mn) and p.defs & vuse))
        if is_vmem(mn):
            suse = {r for r in c.uses if r[0] in ("s", "vcc")}
            if suse:
                rules.append(("R5 VALU sgpr->VMEM", 5, lambda p, suse=suse: is_valu(p.mn) and p.defs & suse))
        if mn.startswith(("s_movrel", "ds_gws", "s_sendmsg")) or (is_vmem(mn) and (" lds" in " " + c.op or mn.endswith("_lds"))):
            rules.append(("R9 SALU m0->user", 1, lambda p: p.mn.startswith("s_") and ("m0", 0) in p.defs))
        if not is_mfma(mn) and (is_valu(mn) or is_vmem(mn) or mn.startswith("ds_")):
            va = {r for r in (c.uses | (c.defs if is_valu(mn) else set())) if r[0] in ("v", "a")}
            if va:
                rules.append(("R10 MFMA->read/overwrite", None, lambda p, va=va: is_mfma(p.mn) and p.defs & va))
        for (rname, need, fn) in rules:
            found_at = []
            if need is None:
                walk_back(i, 22, fn, set(), 0)
                found_at = [(p, acc) for (p, acc) in found_at if acc < mfma_passes(p.mn) + 2]
            else:
                walk_back(i, need, fn, set(), 0)
            for (p, acc) in found_at:
                req = need if need is not None else mfma_passes(p.mn) + 2
                found.append((rname, req, acc, p, c))
    if verbose:
        for (rname, req, acc, p, c) in found:
            print("  VIOLATION %s: %d of %d wait states | %#x %s %s  ->  %#x %s %s" % (rname, acc, req, p.addr, p.mn, p.op[:60], c.addr, c.mn, c.op[:60]))
    return found


# ---------------------------------------------------------------------------------------------------------------- resources
def resources(co):
    """{kernel: dict(scratch, sgpr_spill, vgpr_spill, vgpr, sgpr, lds)} from the code object's metadata note"""
    t = subprocess.check_output([_tool("llvm-readelf"), "--notes", co], text=True)
    out = {}
    for blk in t.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        out[name] = dict(scratch=g("private_segment_fixed_size"), sgpr_spill=g("sgpr_spill_count"), vgpr_spill=g("vgpr_spill_count"),
                         vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"))
    return out


# hot kernels that must stay free of scratch memory and VGPR spills (substring of the mangled name); everything that calls lgr_svd3 is in
# here.  SGPR spills (v_writelane into a spare VGPR, no memory) are listed, not gated.  The fused match_mfma variants are NOT in the list:
# at their 128-VGPR launch bound hipcc parks three or four per-work-item values in scratch outside the K loop.
GATED = ["normals_kernel", "normals_wave_kernel", "refit_kernel", "hypotheses_kernel", "gror_umeyama_kernel", "spfh_tile_kernel", "fpfh_mfma_kernel",
         "count_kernel", "count_list_kernel", "match_sweep", "match_tiles", "rs_hyp_kernel", "rs_store_eval_kernel"]


SCRATCH_ANY = 64   # bytes of scratch memory per work item that no kernel of the library may exceed (the gated ones: none at all)


def main(argv):
    objs = sorted(glob.glob(os.path.join(CSRC, "*.o")))
    if not objs:
        raise SystemExit("isa_hazards: no object files in %s (run make first)" % CSRC)
    want = [a for a in argv if not a.startswith("--")]
    gate = "--gate" in argv
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for obj in objs:
            co = code_object(obj, td)
            if not co:
                continue
            res = resources(co)
            if "--resources" in argv or gate:
                for k, r in sorted(res.items()):
                    if "rocprim" in k:
                        continue
                    if r["scratch"] or r["sgpr_spill"] or r["vgpr_spill"]:
                        gated = any(g in k for g in GATED)
                        if "--resources" in argv or gated:
                            print("%-14s %-70s scratch %3d B  sgpr spills %3d  vgpr spills %3d  (vgpr %d, sgpr %d)%s" % (os.path.basename(obj), k[:70], r["scratch"], r["sgpr_spill"],
                                  r["vgpr_spill"], r["vgpr"], r["sgpr"], "   <-- gated" if gated else ""))
                        if gate and gated and (r["scratch"] or r["vgpr_spill"]):
                            bad += 1
                        elif gate and r["scratch"] > SCRATCH_ANY:
                            # (round 5: two counters carried across a loop made hipcc unroll box_lb_kernel into 512 VGPRs + 3.4 KB of scratch -- ten times
                            #  its time, on the matcher's critical chain, seen only in the next timeline)
                            print("%-14s %-70s scratch %d B > %d B allowed for ANY kernel" % (os.path.basename(obj), k[:70], r["scratch"], SCRATCH_ANY))
                            bad += 1
                if "--resources" in argv and not gate:
                    continue
            funcs = functions(disassemble(co))
            for name, raw in funcs.items():
                if "rocprim" in name or not raw:
                    continue
                if want and not any(w in name for w in want):
                    continue
                ins = [Ins(a, m, o) for (a, m, o) in raw]
                v = scan(ins, name, verbose=False)
                if v or want:
                    print("%s %s: %d instructions, %d violations" % (os.path.basename(obj), name[:90], len(ins), len(v)))
                    for (rname, req, acc, p, c) in v[:12]:
                        print("  VIOLATION %s: %d of %d wait states | %#x %s %s  ->  %#x %s %s" % (rname, acc, req, p.addr, p.mn, p.op[:50], c.addr, c.mn, c.op[:50]))
                bad += len(v)
    if gate and bad:
        raise SystemExit("isa_hazards: %d finding(s)" % bad)
    print("isa_hazards: %d finding(s)" % bad)


if __name__ == "__main__":
    main(sys.argv[1:])
