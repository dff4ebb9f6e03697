#!/usr/bin/env python3
"""Writes qampy_amd/csrc/train_seg_blocks.inc: block A of the segment trainer's hand-scheduled step (train_seg.h) for every
(taps per lane, KEEP, window pieces, padding taps) - one `asm` statement each, the wait states of the sums and of the DPP tree filled
from one list of independent instructions in a fixed priority.  Run after changing the schedule; the output is committed."""
import os

import sys

OUT = os.path.join(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qampy_amd", "csrc"), "train_seg_blocks.inc")

DPP = " row_mask:0xf bank_mask:0xf"


def dot(tpl):
    s = ["v_pk_mul_f32 v[248:249], %[x0], %[w0] op_sel_hi:[0,1]",
         "v_pk_mul_f32 v[250:251], %[x0], %[w0] op_sel:[1,0] op_sel_hi:[1,1]",
         "v_pk_mul_f32 v[252:253], %[x1], %[w1] op_sel_hi:[0,1]",
         "v_pk_mul_f32 v[254:255], %[x1], %[w1] op_sel:[1,0] op_sel_hi:[1,1]"]
    for j in range(2, tpl):
        p, r = ("v[248:249]", "v[250:251]") if j % 2 == 0 else ("v[252:253]", "v[254:255]")
        s.append("v_pk_fma_f32 %s, %%[x%d], %%[w%d], %s op_sel_hi:[0,1,1]" % (p, j, j, p))
        s.append("v_pk_fma_f32 %s, %%[x%d], %%[w%d], %s op_sel:[1,0,0] op_sel_hi:[1,1,1]" % (r, j, j, r))
    return s


SUMP = "v_pk_add_f32 v[248:249], v[248:249], v[252:253]"
SUMR = "v_pk_add_f32 v[250:251], v[250:251], v[254:255]"
SUMY = "v_pk_add_f32 v[254:255], v[248:249], v[250:251] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
KR = "v_cndmask_b32_e64 %[ebr], %[ebr], %[per], %[mk]"
KI = "v_cndmask_b32_e64 %[ebi], %[ebi], %[pei], %[mk]"


def lv(ctl, dst=("v254", "v255")):
    return ["v_add_f32_dpp %s, v254, v254 %s%s" % (dst[0], ctl, DPP), "v_add_f32_dpp %s, v255, v255 %s%s" % (dst[1], ctl, DPP)]


def lanes_of(tpl):
    """11 taps per lane is the 8-lanes-per-chain layout (41 taps x 2 input modes on 8 lanes, 3 DPP levels); 4 and 6 taps per lane the 16-lane one"""
    return 8 if tpl == 11 else 16


def block(tpl, keep, nl, nr):
    if tpl == 11:
        # a lane's window starts at 11 k samples = 88 k bytes: 8-byte aligned only - the pieces are pairs of 64-bit reads (offsets in units of 8 bytes)
        ld = ["ds_read2_b64 %%[d%d], %%[la] offset0:%%[p%d] offset1:%%[q%d]" % (k, k, k) for k in range(nl)]
    else:
        ld = ["ds_read_b128 %%[d%d], %%[la] offset:%%[o%d]" % (k, k) for k in range(nl)]
    xm = ["v_pk_mul_f32 %%[xm%d], %%[x%d], %%[tm]" % (k, tpl - 1 - k) for k in range(nr)]
    body = dot(tpl) + [SUMP, SUMR]
    if keep:
        body.append(KR)                      # one wait state between R0 + R1 and the sum that reads it
        fill = ld + [KI] + xm
    else:
        body.append(ld[0])
        fill = ld[1:] + xm
    body.append(SUMY)
    g = [fill.pop(0) for _ in range(min(2, len(fill)))]      # two wait states between y and the first DPP level
    body += g
    if len(g) == 1:
        body.append("s_nop 0")
    elif len(g) == 0:
        body.append("s_nop 1")
    if lanes_of(tpl) == 8:
        levels = [lv("quad_perm:[1,0,3,2]"), lv("quad_perm:[2,3,0,1]"), lv("row_half_mirror", ("v246", "v247"))]
    else:
        levels = [lv("quad_perm:[1,0,3,2]"), lv("quad_perm:[2,3,0,1]"), lv("row_half_mirror"), lv("row_mirror", ("v246", "v247"))]
    for i, l in enumerate(levels):
        body += l
        if i < len(levels) - 1:
            body.append(fill.pop(0) if fill else "s_nop 0")   # a DPP add reads the level before it: one more instruction in between
    body.append("v_pk_mul_f32 %[sq], v[246:247], v[246:247]")
    body += fill                                              # (what found no slot)
    return body


def block_b(tpl, d1, xname="xp"):
    """block B in front of block A of the next step (one statement: no padding between the two): c = d y into v[244:245] (the selects of the
    A part park it), its rotated copy, the two rounds of the tap update with the PREVIOUS step's samples"""
    if d1:
        s = ["v_pk_mul_f32 v[244:245], %[yp], %[dp] op_sel_hi:[1,0]", "v_pk_mul_f32 %[cr], %[yp], %[dp] op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]"]
    else:
        s = ["v_pk_mul_f32 v[244:245], %[yp], %[dp]", "v_pk_mul_f32 %[cr], %[yp], %[dp] op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]"]
    s += ["v_pk_fma_f32 %%[w%d], %%[%s%d], v[244:245], %%[w%d] op_sel_hi:[0,1,1]" % (j, xname, j, j) for j in range(tpl)]
    s += ["v_pk_fma_f32 %%[w%d], %%[%s%d], %%[cr], %%[w%d] op_sel:[1,0,0] op_sel_hi:[1,1,1]" % (j, xname, j, j) for j in range(tpl)]
    return s


def operands_ba(tpl, nl, nr):
    outs = ['[y] "={v[246:247]}"(y)', '[sq] "=&v"(sq)', '[cr] "=&v"(cr)', '[ebr] "+v"(ebr)', '[ebi] "+v"(ebi)']
    outs += ['[w%d] "+v"(w[%d])' % (j, j) for j in range(tpl)]
    outs += ['[d%d] "=&v"(d%d)' % (k, k) for k in range(nl)]
    outs += ['[xm%d] "=&v"(xm%d)' % (k, k) for k in range(nr)]
    ins = ['[x%d] "v"(x[%d])' % (j, j) for j in range(tpl)] + ['[xp%d] "v"(xp[%d])' % (j, j) for j in range(tpl)]
    ins += ['[yp] "v"(yp)', '[dp] "v"(dp)', '[la] "v"(la)', '[mk] "s"(mk)']
    if nr:
        ins.append('[tm] "v"(tm)')
    ins += imm(tpl, nl)
    return outs, ins


def imm(tpl, nl):
    if tpl == 11:
        return ['[p%d] "n"((O0 + %d) / 8)' % (k, 16 * k) for k in range(nl)] + ['[q%d] "n"((O0 + %d) / 8 + 1)' % (k, 16 * k) for k in range(nl)]
    return ['[o%d] "n"(O0 + %d)' % (k, 16 * k) for k in range(nl)]


def operands(tpl, keep, nl, nr):
    outs = ['[y] "={v[246:247]}"(y)', '[sq] "=&v"(sq)']
    if keep:
        outs += ['[ebr] "+v"(ebr)', '[ebi] "+v"(ebi)']
    outs += ['[d%d] "=&v"(d%d)' % (k, k) for k in range(nl)]
    outs += ['[xm%d] "=&v"(xm%d)' % (k, k) for k in range(nr)]
    ins = ['[x%d] "v"(x[%d])' % (j, j) for j in range(tpl)] + ['[w%d] "v"(w[%d])' % (j, j) for j in range(tpl)] + ['[la] "v"(la)']
    if keep:
        ins += ['[per] "v"(per)', '[pei] "v"(pei)', '[mk] "s"(mk)']
    if nr:
        ins.append('[tm] "v"(tm)')
    ins += imm(tpl, nl)
    return outs, ins


# ---- the wait states gfx950 wants, checked on every generated statement (the hardware does not interlock them):
#   * a register written by a packed operation (v_pk_*) is not read by the NEXT instruction;
#   * a register written by a VALU instruction is not read through DPP by either of the next TWO instructions.
def _regs(tok):
    """registers named by one operand token: v254 -> {v254}; v[248:249] -> {v248, v249}; %[w0] -> {%w0} (a compiler-assigned operand is one unit)"""
    tok = tok.strip().rstrip(",")
    if tok.startswith("%["):
        return {tok[1:].strip("[]")}
    if tok.startswith("v["):
        a, b = tok[2:-1].split(":")
        return {"v%d" % k for k in range(int(a), int(b) + 1)}
    if tok.startswith("v") and tok[1:].isdigit():
        return {tok}
    return set()


def _parse(ins):
    op, rest = ins.split(None, 1) if " " in ins else (ins, "")
    if op.startswith("s_"):
        return op, set(), set(), False
    toks = []
    for t in rest.replace(", ", ",").split(","):
        t = t.strip()
        if t.startswith(("op_sel", "neg_", "quad_perm", "row_", "bank_", "offset")):
            break
        toks.append(t.split()[0] if t else t)
    dst = _regs(toks[0]) if toks else set()
    src = set().union(*[_regs(t) for t in toks[1:]]) if len(toks) > 1 else set()
    if op.startswith("ds_read"):
        src = _regs(toks[1]) if len(toks) > 1 else set()
    if op.startswith("v_cndmask"):
        src |= dst                                            # (in-out)
    return op, dst, src, "_dpp" in op


def check(body, what):
    hist = []                                                 # (op, dst) of the instructions issued so far, s_nop N as N + 1 entries
    for ins in body:
        op, dst, src, dpp = _parse(ins)
        if op == "s_nop":
            hist += [("s_nop", set())] * (int(ins.split()[1]) + 1)
            continue
        if hist:
            pop, pdst = hist[-1]
            assert not (pop.startswith("v_pk_") and (pdst & src)), "%s: '%s' reads what the packed operation right before it wrote" % (what, ins)
        if dpp:
            for back in hist[-2:]:
                assert not (back[0].startswith("v_") and (back[1] & src)), "%s: '%s' reads through DPP what one of the two instructions before it wrote" % (what, ins)
        hist.append((op, dst))


def main():
    lines = ["// GENERATED by scripts/gen_seg_blocks.py - do not edit.  Block A of the segment trainer's step (train_seg.h: seg_block_a) for every",
             "// (taps per lane, KEEP, window pieces NL, padding taps NR): the statement's instructions in issue order, one per line.",
             "// Filling rule: R0 + R1 -> [park re | first piece] -> y; two slots behind y and one behind each of the first three DPP levels take, in this order,",
             "// the window pieces, the second select of the parked error, the padding taps' samples times the lane's 0 / 1 mask; s_nop where nothing is left."]
    first = True
    for tpl in (4, 6, 11):
        for keep in (1, 0):
            for nl in ((3, 4) if tpl == 11 else (1, 2)):
                for nr in range(0, 4):
                    body = block(tpl, keep, nl, nr)
                    check(body, "block A tpl %d keep %d nl %d nr %d" % (tpl, keep, nl, nr))
                    outs, ins = operands(tpl, keep, nl, nr)
                    cond = "TPL == %d && %sKEEP && NL == %d && NR == %d" % (tpl, "" if keep else "!", nl, nr)
                    lines.append("%sif constexpr (%s)" % ("" if first else "else ", cond))
                    first = False
                    lines.append("    asm volatile(")
                    for ins_ in body:
                        lines.append('        "%s\\n\\t"' % ins_)
                    lines.append("        : %s" % ", ".join(outs))
                    lines.append("        : %s" % ", ".join(ins))
                    lines.append('        : "memory", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");')
    lines.append('else static_assert(TPL == 4 || TPL == 6 || TPL == 11, "block A: layouts with 4, 6 or 11 taps per lane");')
    with open(OUT, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", OUT, len(lines), "lines")
    # ---- block B of step i - 1 and block A of step i in ONE statement (seg_block_ba)
    lines = ["// GENERATED by scripts/gen_seg_blocks.py - do not edit.  Block B of the previous step followed by block A of this one (train_seg.h: seg_block_ba):",
             "// one statement, so that the compiler has nothing to pad between them.  c = d y of the previous step lives in v[244:245]: the tap update reads it and the",
             "// selects of the A part park it (KEEP is implied: there is a previous step)."]
    first = True
    for tpl in (4, 6, 11):
        for d1 in (1, 0):
            for nl in ((3, 4) if tpl == 11 else (1, 2)):
                for nr in range(0, 4):
                    body = block_b(tpl, d1) + [i.replace("%[per]", "v244").replace("%[pei]", "v245") for i in block(tpl, 1, nl, nr)]
                    check(body, "blocks B + A tpl %d d1 %d nl %d nr %d" % (tpl, d1, nl, nr))
                    outs, ins = operands_ba(tpl, nl, nr)
                    cond = "TPL == %d && %sD1 && NL == %d && NR == %d" % (tpl, "" if d1 else "!", nl, nr)
                    lines.append("%sif constexpr (%s)" % ("" if first else "else ", cond))
                    first = False
                    lines.append("    asm volatile(")
                    for ins_ in body:
                        lines.append('        "%s\\n\\t"' % ins_)
                    lines.append("        : %s" % ", ".join(outs))
                    lines.append("        : %s" % ", ".join(ins))
                    lines.append('        : "memory", "v244", "v245", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");')
    lines.append('else static_assert(TPL == 4 || TPL == 6 || TPL == 11, "blocks B + A: layouts with 4, 6 or 11 taps per lane");')
    out2 = OUT.replace("train_seg_blocks.inc", "train_seg_blocks_ba.inc")
    with open(out2, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", out2, len(lines), "lines")


if __name__ == "__main__":
    main()
