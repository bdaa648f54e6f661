"""Dev tool: where does a kernel hold its vector registers?  Reads the device assembly hipcc emits
(`hipcc ... --cuda-device-only -S file.hip -o file.s`), runs a classic backward liveness analysis over one kernel's
basic blocks, and prints the live-VGPR count along the instruction stream: the peak, the program points around it,
and a coarse profile per block with the memory / LDS instructions that identify the source region (BVH walk: seven
dwordx4 gathers; SD-tree walk: two; the double-precision series: v_fma_f64 ...).

    python tools/vgpr_liveness.py file.s KERNEL_SUBSTRING [--top 8]

Approximations: scalar control flow only (a write under a divergent exec mask counts as a kill, as LLVM's allocator
also assumes for structurised code); instructions whose destination is also a source (v_fmac, v_mac, v_accvgpr,
sdwa/dpp partial writes) keep the destination live.  Good enough to find WHICH phase of a fused kernel is its peak.
"""
import re
import sys
from collections import defaultdict

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
NO_DEST = ("global_store", "buffer_store", "flat_store", "ds_write", "ds_store", "scratch_store", "s_", "v_cmp", "v_cmpx",
           "global_atomic", "buffer_atomic", "ds_add", "ds_max", "ds_min", "v_nop", "buffer_wbl2", "buffer_inv", "v_readfirstlane",
           "v_readlane", "ds_bpermute_nodest")
DEST_IS_SRC = ("v_fmac", "v_mac", "v_fmaak_disabled", "v_dot2c", "v_pk_fmac", "v_writelane")


def regs_of(text):
    out = []
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out += list(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_operands(s):
    ops, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def parse(lines):
    """-> blocks: list of (label, [ (mnemonic, defs, uses, text) ]), successors by label."""
    blocks, cur, label = [], [], "entry"
    for raw in lines:
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", line)
        if m:
            if cur or label == "entry":
                blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        t = line.strip()
        if t.startswith("."):
            continue
        parts = t.split(None, 1)
        mn = parts[0]
        ops = split_operands(parts[1]) if len(parts) > 1 else []
        defs, uses = [], []
        has_dest = not mn.startswith(NO_DEST) or mn.startswith(("s_",)) is False and mn.startswith("global_atomic") and "glc" in t
        if mn.startswith("s_") or mn.startswith(NO_DEST):
            has_dest = False
        if mn.startswith(("global_atomic", "buffer_atomic", "ds_add_rtn", "ds_bpermute", "ds_swizzle")) and ("glc" in t or "rtn" in mn or mn.startswith(("ds_bpermute", "ds_swizzle"))):
            has_dest = True
        if mn.startswith(("v_readfirstlane", "v_readlane", "v_cmp")):
            has_dest = False
        if has_dest and ops:
            defs = regs_of(ops[0])
            rest = ops[1:]
            # v_div_scale / v_mad_u64 style second destination (vcc / s[...]) holds no VGPR
            for o in rest:
                uses += regs_of(o)
            if mn.startswith(DEST_IS_SRC) or "sdwa" in mn or "dpp" in mn or "op_sel" in t:
                uses += defs
        else:
            for o in ops:
                uses += regs_of(o)
        cur.append((mn, defs, uses, t))
    blocks.append((label, cur))
    return blocks


def successors(blocks):
    names = {lab: i for i, (lab, _) in enumerate(blocks)}
    succ = defaultdict(list)
    for i, (lab, ins) in enumerate(blocks):
        fall = True
        for mn, _, _, t in ins:
            if mn in ("s_branch",):
                tgt = t.split()[-1]
                if tgt in names:
                    succ[i].append(names[tgt])
                fall = False
            elif mn.startswith("s_cbranch"):
                tgt = t.split()[-1]
                if tgt in names:
                    succ[i].append(names[tgt])
                fall = True
            elif mn in ("s_endpgm", "s_setpc_b64"):
                fall = False
        if fall and i + 1 < len(blocks):
            succ[i].append(i + 1)
    return succ


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 8
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[\w$.]*" + re.escape(key) + r"[\w$.]*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    blocks = parse(lines[start + 1:end])
    succ = successors(blocks)
    n = len(blocks)
    use_b, def_b = [set() for _ in range(n)], [set() for _ in range(n)]
    for i, (_, ins) in enumerate(blocks):
        for mn, defs, uses, _ in ins:
            for u in uses:
                if u not in def_b[i]:
                    use_b[i].add(u)
            for d in defs:
                def_b[i].add(d)
    live_in, live_out = [set() for _ in range(n)], [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            inn = use_b[i] | (out - def_b[i])
            if out != live_out[i] or inn != live_in[i]:
                live_out[i], live_in[i] = out, inn
                changed = True
    # per-instruction counts
    rows = []  # (block index, instr index, live count, text)
    for i, (lab, ins) in enumerate(blocks):
        live = set(live_out[i])
        per = []
        for mn, defs, uses, t in reversed(ins):
            live -= set(defs)
            live |= set(uses)
            per.append((len(live | set(defs)), t))
        per.reverse()
        for j, (c, t) in enumerate(per):
            rows.append((i, j, c, t))
    peak = max(r[2] for r in rows)
    print(f"kernel {key}: {sum(len(b[1]) for b in blocks)} instructions in {n} blocks, peak live VGPRs (approx.) {peak}")
    # block profile: max live, instruction mix hints
    prof = []
    for i, (lab, ins) in enumerate(blocks):
        if not ins:
            continue
        mx = max(r[2] for r in rows if r[0] == i)
        g16 = sum(1 for x in ins if x[0].startswith("global_load_dwordx4"))
        g8 = sum(1 for x in ins if x[0].startswith("global_load_dwordx2"))
        g4 = sum(1 for x in ins if x[0] == "global_load_dword" or x[0].startswith("global_load_dword "))
        f64 = sum(1 for x in ins if "_f64" in x[0])
        lds = sum(1 for x in ins if x[0].startswith("ds_"))
        st = sum(1 for x in ins if x[0].startswith("global_store"))
        prof.append((mx, i, lab, len(ins), g16, g8, g4, f64, lds, st, len(live_in[i]), len(live_out[i])))
    print(f"\nblocks with the highest pressure (max live | block | instrs | x4 x2 x1 loads | f64 | lds | stores | live in/out):")
    for p in sorted(prof, reverse=True)[:top]:
        print("  %3d  #%-4d %-14s %4d   %2d %2d %2d   %3d  %3d  %2d   in %3d out %3d" % (p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11]))
    if "--explain" in sys.argv:
        # for the registers alive INTO block B: the last instruction before B (in layout order) that wrote each, and the first
        # one after B that reads it -- layout order stands in for control flow, good enough to tell what a value is
        B = int(sys.argv[sys.argv.index("--explain") + 1])
        flat = []  # (block, text, defs, uses)
        for i, (lab, ins) in enumerate(blocks):
            for mn, defs, uses, t in ins:
                flat.append((i, t, defs, uses))
        first_of = next(k for k, f in enumerate(flat) if f[0] >= B)
        last_of = max(k for k, f in enumerate(flat) if f[0] <= B)
        used_inside = set()
        for k in range(first_of, last_of + 1):
            used_inside |= set(flat[k][2]) | set(flat[k][3])
        print(f"\nregisters alive into block #{B} ({len(live_in[B])}); * = touched inside the block")
        for r in sorted(live_in[B]):
            d = next((flat[k] for k in range(first_of - 1, -1, -1) if r in flat[k][2]), None)
            u = next((flat[k] for k in range(last_of + 1, len(flat)) if r in flat[k][3]), None)
            print("  v%-3d %s def #%s: %-58s | next use #%s: %s" % (r, "*" if r in used_inside else " ", d[0] if d else "?", (d[1] if d else "?")[:58],
                                                                  u[0] if u else "?", (u[1] if u else "?")[:58]))
    print("\npressure along the kernel (every block: max live, live across = min(in, out), what it does):")
    for p in sorted(prof, key=lambda x: x[1]):
        if p[3] >= 12 or p[4] >= 2:
            tag = []
            if p[4] >= 6: tag.append("BVH node step (%d x4 gathers)" % p[4])
            elif p[4] >= 2: tag.append("%d x4 gathers" % p[4])
            if p[7] >= 8: tag.append("f64 x%d" % p[7])
            if p[9] >= 4: tag.append("stores x%d" % p[9])
            if p[8] >= 4: tag.append("lds x%d" % p[8])
            print("  #%-4d %-14s max %3d  across %3d  n=%-4d %s" % (p[1], p[2], p[0], min(p[10], p[11]), p[3], ", ".join(tag)))


if __name__ == "__main__":
    main()
