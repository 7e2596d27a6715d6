#!/usr/bin/env python3
"""Generate vuke_amd/csrc/vk_field_gen.cuh: the v_mad_u64_u32 column kernels of the 256x256-bit
multiply / square for gfx950, as inline-asm statements.

Scheme (see vk_field.cuh): product scanning; column k = i + j accumulates into the register pair
(limb k, limb k+1); the multiplier's carry-out is counted into cy[k] = limb k+2, which is where
the next column of the same parity starts.  Each asm statement handles one even column and the
following odd column, interleaved, so dependent multiplies are always 3 issue slots apart; the
even column carries through vcc, the odd one through a scratch SGPR pair.  A column that continues from the counter of the
column two before it takes the pair (counter, 0) as the SOURCE addend of its first multiply-add -- the accumulator itself
is a fresh early-clobber output -- so the compiler only has to put a zero next to the counter (one plain v_mov_b32) instead
of copying a 64-bit pair into the tied accumulator (v_mov_b64, a 16-lane-class instruction, plus the s_nop the tied form
drew after every statement): +1.3 % on the Milksad sweep, same-box (profiles/archive/r02_kernel_experiments.txt).

Operand discipline of every statement emitted here (audited in round 2, profiles/archive/r02_short_fold_root_cause.txt): a
statement of more than one instruction either early-clobbers its outputs ("=&v" / "=&s") or, where an operand is
read-write ("+v"), every plain input is read only by the instruction that overwrites the register it could share (the
reduction's t[8+i] is the high half of its own accumulator pair); carries never leave a statement in vcc, only in a
declared SGPR pair; nothing inside a statement needs a software wait state (VALU carry-out -> VALU carry-in and VALU ->
SALU reads of an SGPR are interlocked on gfx9-family parts; no DPP, readlane or VMEM use of a freshly written register).
The C++ callers may alias result and operands freely: results are written only by the last statement, from temporaries.
"""
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the device header goes to the kernel sources, the host rendering (test infrastructure) under tests/: a change to the host
# backend must not alter the hash of vuke_amd/csrc that the committed PMC profiles are keyed on (bench.csrc_hash)
OUT = os.path.join(os.environ.get("VK_GEN_OUT_DIR", os.path.join(_ROOT, "vuke_amd", "csrc")), "vk_field_gen.cuh")
OUT_HOST = os.path.join(os.environ.get("VK_GEN_OUT_DIR", os.path.join(_ROOT, "tests", "host_emu")), "vk_field_gen_host.h")

# ---- second backend: the same statements as plain C++ for one lane (tests/host_emu/).  Every asm statement the generator emits is
# registered here with its operand tables; main() writes vk_field_gen_host.h = the gfx950 file with each statement replaced by
# calls into a ten-instruction emulator (tests/host_emu/vk_field_prims_host.h).  tests/test_host_field.py compiles the field layer that way with g++
# and holds it to python integers on the edge operands of the GPU field test: the arithmetic of the generated code -- carry
# accounting, flagged and exact paths, aliasing -- is checked on a machine without a GPU.  (Hardware hazards are not: those
# are what the GPU tests and the coalescer lint are for.)
# VK_GEN_PIN=1 (round 4 experiment): the carry counters of the even / odd columns live in the low halves
# of two u64 variables pinned to physical register pairs ("+{v[2:3]}" / "+{v[4:5]}"), whose high halves stay zero: the pair
# (counter, 0) a column starts from is then the pinned pair itself, and the nine `v_mov_b32 v, 0` per product that put a zero
# next to a compiler-allocated counter become two `v_mov_b64 pair, 0` at the top of the product.
PIN = os.environ.get("VK_GEN_PIN", "0") == "1"
PIN_REG = {"cpe": 2, "cpo": 4}
HOST = {}
EMU = {"v_mad_u64_u32": "emu_mad", "v_add_co_u32": "emu_add_co", "v_addc_co_u32": "emu_addc_co", "v_sub_co_u32": "emu_sub_co",
       "v_subb_co_u32": "emu_subb_co", "v_cndmask_b32": "emu_cndmask", "v_add_u32": "emu_add", "v_mul_u32_u24": "emu_mul_u24",
       "v_lshlrev_b32": "emu_lshlrev", "v_alignbit_b32": "emu_alignbit", "s_or_b64": "emu_s_or"}


def host_register(asm_text, body, exprs):
    """body: instruction lines with %N placeholders; exprs: C++ expression of operand N."""
    import re
    out = ["    { // host form of the statement (generated)", "        u64 vcc = 0, emu_dead = 0;", "        (void)vcc; (void)emu_dead;"]
    for line in body:
        mn, rest = line.split(None, 1)
        ops = [o.strip() for o in rest.split(",")]
        cpp = []
        for o in ops:
            m = re.fullmatch(r"%(\d+)", o)
            if m:
                cpp.append(exprs[int(m.group(1))])
            elif o == "vcc":
                cpp.append("vcc")
            elif o == "s[100:101]":
                cpp.append("emu_dead")
            elif o in ("v%d" % PIN_REG["cpe"], "v%d" % PIN_REG["cpo"]):   # low half of a pinned counter pair
                cpp.append("emu_lo32(%s)" % ("cpe" if o == "v%d" % PIN_REG["cpe"] else "cpo"))
            else:
                cpp.append(o + ("ull" if (mn == "v_mad_u64_u32" and o == "0") else "u"))
        out.append(f"        {EMU[mn]}({', '.join(cpp)});")
    out.append("    }")
    HOST[asm_text] = "\n".join(out) + "\n"



def col_products_mul(k):
    return [(i, k - i) for i in range(max(0, k - 7), min(k, 7) + 1)]


def col_products_sqr(k):
    return [(i, k - i) for i in range(max(0, k - 7), min(k, 7) + 1) if i < k - i]


def emit_pair(lines, ke, prods_e, prods_o, a_name, b_name, same, zero_e=False, zero_o=False, start_e=None, start_o=None, cont=None):
    """one asm statement: even column ke and odd column ke+1.  A column whose start value is zero takes the literal 0
    as the addend of its first multiply (no register to clear); the first carry of a column *sets* its counter
    (0 + 0 + carry), so the counters need no clearing either.
    cont = (counter_exists_e, counter_exists_o): the statement CONTINUES two columns an earlier statement began (the second
    product of fe_mul2_cols): the accumulators are read-write, every multiply-add -- the first included, its addend is a full
    64-bit pair -- has its carry counted, into the existing counter or, where the column had a single product so far, into a
    fresh one."""
    ko = ke + 1
    # operand table
    ops_out = []   # (constraint, expr)
    ops_in = []
    idx = {}

    def out(c, e):
        idx[e] = len(ops_out)
        ops_out.append((c, e))

    have_e, have_o = bool(prods_e), bool(prods_o)
    cnt_e, cnt_o = f"cy[{ke}]", f"cy[{ko}]"      # what the carry-count instructions of the two columns write
    assert not (cont and PIN)
    if have_e:
        out("=&v" if (zero_e or start_e) else "+v", f"acc[{ke}]")
        if cont:
            out("+v" if cont[0] else "=&v", f"cy[{ke}]")
        elif PIN and (len(prods_e) > 1 or start_e):
            out("+{v[%d:%d]}" % (PIN_REG["cpe"], PIN_REG["cpe"] + 1), "cpe")
            cnt_e = "v%d" % PIN_REG["cpe"]
        elif len(prods_e) > 1:
            out("=&v", f"cy[{ke}]")
    if have_o:
        out("=&v" if (zero_o or start_o) else "+v", f"acc[{ko}]")
        if cont:
            out("+v" if cont[1] else "=&v", f"cy[{ko}]")
        elif PIN and (len(prods_o) > 1 or start_o):
            out("+{v[%d:%d]}" % (PIN_REG["cpo"], PIN_REG["cpo"] + 1), "cpo")
            cnt_o = "v%d" % PIN_REG["cpo"]
        elif len(prods_o) > 1:
            out("=&v", f"cy[{ko}]")
        if len(prods_o) > 1 or cont:
            out("=&s", "sc")
    used = []
    for (i, j) in prods_e + prods_o:
        for nm in ((a_name, i), (a_name if same else b_name, j)):
            if nm not in used:
                used.append(nm)
    base = len(ops_out)
    for n, nm in enumerate(used):
        idx[nm] = base + n
        ops_in.append(("v", f"{nm[0]}.v[{nm[1]}]"))
    for st in (start_e, start_o):
        if st and st not in idx:      # (pinned pairs are read-write outputs already)
            idx[st] = len(ops_out) + len(ops_in)
            ops_in.append(("v", st))

    def r(e):
        return f"%{idx[e]}"

    body = []
    ne, no = len(prods_e), len(prods_o)
    for step in range(max(ne, no)):
        if step < ne:
            i, j = prods_e[step]
            add = "0" if (zero_e and step == 0) else (r(start_e) if (start_e and step == 0) else r(f'acc[{ke}]'))
            body.append(f"v_mad_u64_u32 {r(f'acc[{ke}]')}, vcc, {r((a_name, i))}, {r((a_name if same else b_name, j))}, {add}")
        if step < no:
            i, j = prods_o[step]
            sd = r("sc") if (no > 1 or cont) else "vcc"
            add = "0" if (zero_o and step == 0) else (r(start_o) if (start_o and step == 0) else r(f'acc[{ko}]'))
            body.append(f"v_mad_u64_u32 {r(f'acc[{ko}]')}, {sd}, {r((a_name, i))}, {r((a_name if same else b_name, j))}, {add}")
        ce = cnt_e if cnt_e.startswith("v") else r(cnt_e) if cnt_e in idx else None
        co = cnt_o if cnt_o.startswith("v") else r(cnt_o) if cnt_o in idx else None
        if cont:
            if step < ne:
                prev = ce if (cont[0] or step > 0) else "0"
                body.append(f"v_addc_co_u32 {ce}, vcc, 0, {prev}, vcc")
            if step < no:
                prev = co if (cont[1] or step > 0) else "0"
                body.append(f"v_addc_co_u32 {co}, {r('sc')}, 0, {prev}, {r('sc')}")
            continue
        if step < ne and step > 0:
            prev = "0" if step == 1 else ce
            body.append(f"v_addc_co_u32 {ce}, vcc, 0, {prev}, vcc")
        if step < no and step > 0:
            prev = "0" if step == 1 else co
            body.append(f"v_addc_co_u32 {co}, {r('sc')}, 0, {prev}, {r('sc')}")
    # vcc hazard: if the odd column has a single product it would write vcc between the even
    # column's multiply and its carry add -> route it through a dead SGPR pair instead
    if no == 1 and ne > 1 and not cont:
        body = [b.replace(f"{r(f'acc[{ko}]')}, vcc,", f"{r(f'acc[{ko}]')}, s[100:101],") if b.startswith(f"v_mad_u64_u32 {r(f'acc[{ko}]')}") else b for b in body]
        clob = '"vcc", "s100", "s101"'
    else:
        clob = '"vcc"'
    s = '    asm("' + '\\n\\t"\n        "'.join(body) + '"\n'
    s += "        : " + ", ".join(f'"{c}"({e})' for c, e in ops_out) + "\n"
    s += "        : " + ", ".join(f'"{c}"({e})' for c, e in ops_in) + "\n"
    s += f"        : {clob});\n"
    host_register(s, body, [e for _, e in ops_out] + [e for _, e in ops_in])
    lines.append(s)


def gen_cols(name, prods_fn, same):
    L = []
    sig = "const fe &a" if same else "const fe &a, const fe &b"
    L.append(f"// acc[k] = pair (limb k, limb k+1) of column k; returns limb 15 of the odd number\n")
    L.append(f"VK_DEV u32 {name}(u64 acc[15], {sig})\n{{\n")
    if PIN:
        L.append("    u64 cpe = 0, cpo = 0; // (carry counter, 0) of the even / odd columns, pinned to v[2:3] / v[4:5]: the high halves stay zero\n    vk_carry sc;\n    (void)sc;\n")
    else:
        L.append("    u32 cy[15];\n    vk_carry sc;\n    (void)sc;\n")
    has_cy = [len(prods_fn(k)) > 1 for k in range(15)]   # columns with one product cannot overflow
    for ke in range(0, 15, 2):
        pe, po = prods_fn(ke), (prods_fn(ke + 1) if ke + 1 < 15 else [])
        zero, start = {}, {}
        for k in (ke, ke + 1):
            if k < 15:
                zero[k] = not (k >= 2 and has_cy[k - 2])
                pin = "cpe" if k % 2 == 0 else "cpo"
                if not zero[k] and prods_fn(k):
                    # the counter pair (cy, 0) is a SOURCE of the column's first multiply-add, not its tied accumulator
                    if PIN:
                        start[k] = pin
                    else:
                        L.append(f"    const u64 cz{k} = (u64)cy[{k - 2}];\n")
                        start[k] = f"cz{k}"
                elif not (zero[k] and prods_fn(k)):
                    L.append(f"    acc[{k}] = {'0ull' if zero[k] else (pin if PIN else '(u64)cy[%d]' % (k - 2))};\n")
        if pe or po:
            emit_pair(L, ke, pe, po, "a", "b", same, zero_e=zero.get(ke, False), zero_o=zero.get(ke + 1, False),
                      start_e=start.get(ke), start_o=start.get(ke + 1))
    L.append(f"    return {('(u32)cpo' if PIN else 'cy[13]') if has_cy[13] else '0u'};\n}}\n\n")
    return "".join(L)


def gen_cols2():
    """fe_mul2_cols: the columns of a*b + c*d in ONE set of accumulators (round 5: the y-coordinate of every point addition is
    such a sum; one merge and one reduction instead of two, and no field subtraction).  Per column pair two statements (an asm
    statement takes at most 30 operands): the a*b products exactly as fe_mul_cols emits them, then the c*d products continuing
    the same accumulators with every carry counted.  The sum is below 2^513: column 14 gets a counter of its own (limb 16)."""
    L = []
    L.append("// acc[k] = pair (limb k, limb k+1) of column k of a*b + c*d; returns limb 15 of the odd number, top16 = limb 16 of the even one\n")
    L.append("VK_DEV u32 fe_mul2_cols(u64 acc[15], u32 &top16, const fe &a, const fe &b, const fe &c, const fe &d)\n{\n")
    L.append("    u32 cy[15];\n    vk_carry sc;\n    (void)sc;\n")
    for ke in range(0, 15, 2):
        cols = [k for k in (ke, ke + 1) if k < 15]
        zero, start = {}, {}
        for k in cols:
            zero[k] = k < 2
            if not zero[k]:
                L.append(f"    const u64 cz{k} = (u64)cy[{k - 2}];\n")
                start[k] = f"cz{k}"
        pe, po = col_products_mul(ke), (col_products_mul(ke + 1) if ke + 1 < 15 else [])
        emit_pair(L, ke, pe, po, "a", "b", False, zero_e=zero.get(ke, False), zero_o=zero.get(ke + 1, False),
                  start_e=start.get(ke), start_o=start.get(ke + 1))
        emit_pair(L, ke, pe, po, "c", "d", False, cont=(len(pe) > 1, len(po) > 1))
    L.append("    top16 = cy[14];\n    return cy[13];\n}\n\n")
    return "".join(L)


def gen_merge17():
    """merge_eo with the seventeenth limb of fe_mul2_cols' sum: limb 16 = the even number's counter + the chain's last carry."""
    L = []
    L.append("// t[0..16] = E + O for the columns of fe_mul2_cols (a sum below 2^513: t[16] is 0 or 1)\n")
    L.append("VK_DEV void merge_eo17(u32 t[17], const u64 acc[15], u32 top, u32 top16)\n{\n")
    L.append("    u32 e[16], o[16];\n#pragma unroll\n    for (int m = 0; m < 8; m++) {\n        e[2 * m] = (u32)acc[2 * m];\n        e[2 * m + 1] = (u32)(acc[2 * m] >> 32);\n    }\n")
    L.append("#pragma unroll\n    for (int m = 0; m < 7; m++) {\n        o[2 * m + 1] = (u32)acc[2 * m + 1];\n        o[2 * m + 2] = (u32)(acc[2 * m + 1] >> 32);\n    }\n    o[15] = top;\n    t[0] = e[0];\n    vk_carry c;\n")
    body = ["v_add_co_u32 %0, vcc, %8, %15"]
    for j in range(2, 8):
        last = j == 7
        body.append(f"v_addc_co_u32 %{j - 1}, {'%7' if last else 'vcc'}, %{7 + j}, %{14 + j}, vcc")
    s = '    asm("' + '\\n\\t"\n        "'.join(body) + '"\n'
    s += "        : " + ", ".join(f'"=&v"(t[{j}])' for j in range(1, 8)) + ', "=&s"(c)\n'
    s += "        : " + ", ".join(f'"v"(e[{j}])' for j in range(1, 8)) + ", " + ", ".join(f'"v"(o[{j}])' for j in range(1, 8)) + "\n"
    s += '        : "vcc");\n'
    host_register(s, body, [f"t[{j}]" for j in range(1, 8)] + ["c"] + [f"e[{j}]" for j in range(1, 8)] + [f"o[{j}]" for j in range(1, 8)])
    L.append(s)
    # second half: t8..t15 with the carry in %26, then limb 16 = top16 + carry
    body = []
    for j in range(8, 16):
        first = j == 8
        body.append(f"v_addc_co_u32 %{j - 8}, vcc, %{j + 1}, %{j + 9}, {'%26' if first else 'vcc'}")
    body.append("v_addc_co_u32 %8, vcc, 0, %25, vcc")
    s = '    asm("' + '\\n\\t"\n        "'.join(body) + '"\n'
    s += "        : " + ", ".join(f'"=&v"(t[{j}])' for j in range(8, 17)) + "\n"
    s += "        : " + ", ".join(f'"v"(e[{j}])' for j in range(8, 16)) + ", " + ", ".join(f'"v"(o[{j}])' for j in range(8, 16)) + ', "v"(top16), "s"(c)\n'
    s += '        : "vcc");\n'
    host_register(s, body, [f"t[{j}]" for j in range(8, 17)] + [f"e[{j}]" for j in range(8, 16)] + [f"o[{j}]" for j in range(8, 16)] + ["top16", "c"])
    L.append(s + "}\n\n")
    return "".join(L)


def gen_merge():
    L = []
    L.append("// t[0..15] = E + O: column k's pair acc[k] = (limb k, limb k+1); top = limb 15 of O\n")
    L.append("VK_DEV void merge_eo(u32 t[16], const u64 acc[15], u32 top)\n{\n")
    L.append("    u32 e[16], o[16];\n#pragma unroll\n    for (int m = 0; m < 8; m++) {\n        e[2 * m] = (u32)acc[2 * m];\n        e[2 * m + 1] = (u32)(acc[2 * m] >> 32);\n    }\n")
    L.append("#pragma unroll\n    for (int m = 0; m < 7; m++) {\n        o[2 * m + 1] = (u32)acc[2 * m + 1];\n        o[2 * m + 2] = (u32)(acc[2 * m + 1] >> 32);\n    }\n    o[15] = top;\n    t[0] = e[0];\n    vk_carry c;\n")
    # first half: t1..t7
    body = ["v_add_co_u32 %0, vcc, %8, %15"]
    for j in range(2, 8):
        last = j == 7
        body.append(f"v_addc_co_u32 %{j - 1}, {'%7' if last else 'vcc'}, %{7 + j}, %{14 + j}, vcc")
    s = '    asm("' + '\\n\\t"\n        "'.join(body) + '"\n'
    s += "        : " + ", ".join(f'"=&v"(t[{j}])' for j in range(1, 8)) + ', "=&s"(c)\n'
    s += "        : " + ", ".join(f'"v"(e[{j}])' for j in range(1, 8)) + ", " + ", ".join(f'"v"(o[{j}])' for j in range(1, 8)) + "\n"
    s += '        : "vcc");\n'
    host_register(s, body, [f"t[{j}]" for j in range(1, 8)] + ["c"] + [f"e[{j}]" for j in range(1, 8)] + [f"o[{j}]" for j in range(1, 8)])
    L.append(s)
    # second half: t8..t15, carry in %24
    body = []
    for j in range(8, 16):
        first = j == 8
        body.append(f"v_addc_co_u32 %{j - 8}, vcc, %{j}, %{j + 8}, {'%24' if first else 'vcc'}")
    s = '    asm("' + '\\n\\t"\n        "'.join(body) + '"\n'
    s += "        : " + ", ".join(f'"=&v"(t[{j}])' for j in range(8, 16)) + "\n"
    s += "        : " + ", ".join(f'"v"(e[{j}])' for j in range(8, 16)) + ", " + ", ".join(f'"v"(o[{j}])' for j in range(8, 16)) + ', "s"(c)\n'
    s += '        : "vcc");\n'
    host_register(s, body, [f"t[{j}]" for j in range(8, 16)] + [f"e[{j}]" for j in range(8, 16)] + [f"o[{j}]" for j in range(8, 16)] + ["c"])
    L.append(s + "}\n\n")
    return "".join(L)


class Asm:
    """Build one inline-asm statement from symbolic operands."""

    def __init__(self):
        self.outs, self.ins, self.body, self.clob = [], [], [], ["vcc"]
        self.map = {}

    def out(self, cons, expr):
        if expr not in self.map:
            self.map[expr] = ("o", len(self.outs))
            self.outs.append((cons, expr))
        return expr

    def inp(self, cons, expr):
        if expr not in self.map:
            self.map[expr] = ("i", len(self.ins))
            self.ins.append((cons, expr))
        return expr

    def ins_(self, text, *ops):
        self.body.append((text, ops))

    def render(self, indent="    "):
        def num(e):
            kind, n = self.map[e]
            return n if kind == "o" else len(self.outs) + n
        lines = []
        for text, ops in self.body:
            lines.append(text.format(*[f"%{num(o)}" if o in self.map else o for o in ops]))
        assert len(self.outs) + len(self.ins) <= 30, (len(self.outs), len(self.ins))
        s = indent + 'asm("' + ('\\n\\t"\n' + indent + '    "').join(lines) + '"\n'
        s += indent + "    : " + ", ".join(f'"{c}"({e})' for c, e in self.outs) + "\n"
        s += indent + "    : " + ", ".join(f'"{c}"({e})' for c, e in self.ins) + "\n"
        s += indent + "    : " + ", ".join(f'"{c}"' for c in self.clob) + ");\n"
        host_register(s, lines, [e for _, e in self.outs] + [e for _, e in self.ins])
        return s


def chain(dst, xs, ys, first_carry_in=None, carry_out=None, sub=False):
    """dst[i] = xs[i] +/- ys[i] with a vcc carry chain.  ys[i] may be the literal '0'."""
    a = Asm()
    n = len(dst)
    for d in dst:
        a.out("=&v", d)
    if carry_out:
        a.out("=&s", carry_out)
    for x in xs:
        if x != "0":
            a.inp("v", x)
    for y in ys:
        if y != "0":
            a.inp("v", y)
    if first_carry_in:
        a.inp("s", first_carry_in)
    op0, opc = ("v_sub_co_u32", "v_subb_co_u32") if sub else ("v_add_co_u32", "v_addc_co_u32")
    for i in range(n):
        sd = carry_out if (carry_out and i == n - 1) else "vcc"
        if i == 0 and not first_carry_in:
            a.ins_(op0 + " {}, {}, {}, {}", dst[i], sd, xs[i], ys[i])
        else:
            ci = first_carry_in if i == 0 else "vcc"
            a.ins_(opc + " {}, {}, {}, {}, {}", dst[i], sd, xs[i], ys[i], ci)
    return a.render()


def fold_stage():
    L = []
    # stage 3: fold o = s[8] + 2^32*c3:  o*C = s8*977 + (s8 << 32) + c3*(977 << 32) + c3 << 64
    L.append("    u64 m = mul_wide_s(s[8], VK_PC0);\n    u32 ml = (u32)m, mh = (u32)(m >> 32);\n")
    L.append("    u32 w1, w2;\n    vk_carry c4;\n")
    a = Asm()
    for e in ("w1", "w2"):
        a.out("=&v", e)
    for j in range(8):
        a.out("=&v", f"r.v[{j}]")
    a.out("=&s", "c4")
    for e in ("mh", "ml") + tuple(f"s[{j}]" for j in range(9)):
        a.inp("v", e)
    a.inp("s", "c3")
    a.inp("v", "VK_PC0")
    a.ins_("v_cndmask_b32 {}, 0, {}, {}", "w1", "VK_PC0", "c3")       # 977 * c3
    a.ins_("v_add_u32 {}, {}, {}", "w1", "w1", "mh")                    # <= 1953
    a.ins_("v_add_co_u32 {}, vcc, {}, {}", "w1", "w1", "s[8]")          # + s8, carry k in vcc
    a.ins_("v_cndmask_b32 {}, 0, 1, {}", "w2", "c3")
    a.ins_("v_addc_co_u32 {}, vcc, 0, {}, vcc", "w2", "w2")             # c3 + k  (limb 2 addend)
    a.ins_("v_add_co_u32 {}, vcc, {}, {}", "r.v[0]", "s[0]", "ml")
    a.ins_("v_addc_co_u32 {}, vcc, {}, {}, vcc", "r.v[1]", "s[1]", "w1")
    a.ins_("v_addc_co_u32 {}, vcc, {}, {}, vcc", "r.v[2]", "s[2]", "w2")
    for j in range(3, 7):
        a.ins_("v_addc_co_u32 {}, vcc, 0, {}, vcc", f"r.v[{j}]", f"s[{j}]")
    a.ins_("v_addc_co_u32 {}, {}, 0, {}, vcc", "r.v[7]", "c4", "s[7]")
    L.append(a.render())
    return "".join(L)


def gen_reduce():
    L = []
    L.append("// r = t_lo + t_hi * (2^32 + 977) folded to 8 limbs, weakly reduced (see vk_field.cuh): the form without any\n")
    L.append("// precondition on t_hi -- three carry chains; fe_reduce16 falls back to it\n")
    L.append("VK_DEV void fe_reduce16_slow(fe &r, const u32 t[16])\n{\n")
    L.append("    u32 ql[8], qh[8];\n#pragma unroll\n    for (int i = 0; i < 8; i++) {\n        u64 q = mul_wide_s(t[8 + i], VK_PC0);\n        ql[i] = (u32)q;\n        qh[i] = (u32)(q >> 32);\n    }\n")
    L.append("    u32 a[9], b[9], s[9];\n    vk_carry c1, c3;\n")
    # chain 1: a_j = t_j + ql_j, a_8 = carry
    a = Asm()
    for j in range(9):
        a.out("=&v", f"a[{j}]")
    for j in range(8):
        a.inp("v", f"t[{j}]")
    for j in range(8):
        a.inp("v", f"ql[{j}]")
    a.ins_("v_add_co_u32 {}, vcc, {}, {}", "a[0]", "t[0]", "ql[0]")
    for j in range(1, 8):
        a.ins_("v_addc_co_u32 {}, vcc, {}, {}, vcc", f"a[{j}]", f"t[{j}]", f"ql[{j}]")
    a.ins_("v_addc_co_u32 {}, vcc, 0, 0, vcc", "a[8]")
    L.append(a.render())
    # chain 2: b_j = a_j + qh_{j-1}, j = 1..8 (no carry out of limb 8)
    L.append(chain([f"b[{j}]" for j in range(1, 9)], [f"a[{j}]" for j in range(1, 9)], [f"qh[{j - 1}]" for j in range(1, 9)]))
    # chain 3: s_j = b_j + t[8+j-1], j = 1..8, carry out c3 = bit 32 of limb 8
    L.append(chain([f"s[{j}]" for j in range(1, 9)], [f"b[{j}]" for j in range(1, 9)], [f"t[{7 + j}]" for j in range(1, 9)], carry_out="c3"))
    L.append("    s[0] = a[0];\n")
    L.append(fold_stage())
    L.append("    if (__builtin_expect(c4 != 0, 0)) fe_wrap_add(r, c4); // value was 2^256 + small: add C (cannot carry again)\n}\n\n")
    return "".join(L)


def gen_addsub():
    L = []
    for name, sub in (("fe_add", False), ("fe_sub", True)):
        L.append(f"// r = a {'-' if sub else '+'} b (mod p), weakly reduced; operands may alias\n")
        L.append(f"VK_DEV void {name}(fe &r, const fe &a, const fe &b)\n{{\n    u32 d[8];\n    vk_carry c1;\n")
        L.append(chain([f"d[{j}]" for j in range(8)], [f"a.v[{j}]" for j in range(8)], [f"b.v[{j}]" for j in range(8)], carry_out="c1", sub=sub))
        # fix-up: +/- C under mask c1, limbs 0 and 1 only; a carry beyond goes through fe_ripple_*
        L.append("    u32 k0, k1;\n    vk_carry c2;\n")
        a = Asm()
        for e in ("k0", "k1"):
            a.out("=&v", e)
        for j in range(2):
            a.out("=&v", f"r.v[{j}]")
        a.out("=&s", "c2")
        for j in range(2):
            a.inp("v", f"d[{j}]")
        a.inp("s", "c1")
        a.inp("v", "VK_PC0")
        a.ins_("v_cndmask_b32 {}, 0, {}, {}", "k0", "VK_PC0", "c1")
        a.ins_("v_cndmask_b32 {}, 0, 1, {}", "k1", "c1")
        o0, oc = ("v_sub_co_u32", "v_subb_co_u32") if sub else ("v_add_co_u32", "v_addc_co_u32")
        a.ins_(o0 + " {}, vcc, {}, {}", "r.v[0]", "d[0]", "k0")
        a.ins_(oc + " {}, {}, {}, {}, vcc", "r.v[1]", "c2", "d[1]", "k1")
        L.append(a.render())
        L.append("#pragma unroll\n    for (int i = 2; i < 8; i++) r.v[i] = d[i];\n")
        fn = "fe_ripple_sub" if sub else "fe_ripple_add"
        L.append(f"    if (__builtin_expect(c2 != 0, 0)) {fn}(r, 2, c2);\n}}\n\n")
    return "".join(L)


def gen_reduce_fast():
    """r = t_lo + t_hi * C with one multiply-add per high limb: t[8+i] * 977 + (t[8+i] : t[i]) is the whole contribution of
    limb pair i at position i; it fits 64 bits unless t[8+i] >= 2^32 - 977 - 1 (one multiply in 2^22), which the
    multiplier's carry-out reports: any lane of the wave -> the three-chain form.  Otherwise one chain sums the overlapping
    halves."""
    L = []
    L.append("// r = t_lo + t_hi * (2^32 + 977), weakly reduced: 8 multiply-adds that also add t_lo and t_hi << 32, one carry chain, fold\n")
    L.append("VK_DEV void fe_reduce16(fe &r, const u32 t[16])\n{\n")
    L.append("    // with every t[8 + i] <= 2^32 - 979 no multiply-add below overflows 64 bits ((2^32 - 979)(2^32 + 977) + 2^32 - 1 < 2^64)\n    // and limb 8 of the sum stays below 2^32 (hi[7] <= t[15] + 977 <= 2^32 - 2, plus one carry)\n")
    L.append("    u32 mx = t[8];\n#pragma unroll\n    for (int i = 9; i < 16; i++) mx = mx > t[i] ? mx : t[i]; // v_max3_u32 x 3 + v_max_u32\n")
    L.append("    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx > 0xFFFFFC2Du) != 0, 0)) {\n        fe_reduce16_slow(r, t);\n    } else {\n")
    L.append("    u64 v[8];\n#pragma unroll\n    for (int i = 0; i < 8; i++) v[i] = ((u64)t[8 + i] << 32) | t[i];\n")
    for half in range(2):
        a = Asm()
        for i in range(4 * half, 4 * half + 4):
            a.out("+v", f"v[{i}]")
        for i in range(4 * half, 4 * half + 4):
            a.inp("v", f"t[{8 + i}]")
        a.inp("s", "VK_PC0")
        for i in range(4 * half, 4 * half + 4):
            a.ins_("v_mad_u64_u32 {}, vcc, {}, {}, {}", f"v[{i}]", f"t[{8 + i}]", "VK_PC0", f"v[{i}]")
        L.append(a.render())
    L.append("    u32 lo[8], hi[8], s[9];\n#pragma unroll\n    for (int i = 0; i < 8; i++) {\n        lo[i] = (u32)v[i];\n        hi[i] = (u32)(v[i] >> 32);\n    }\n")
    a = Asm()
    for j in range(1, 9):
        a.out("=&v", f"s[{j}]")
    for j in range(1, 8):
        a.inp("v", f"lo[{j}]")
    for j in range(8):
        a.inp("v", f"hi[{j}]")
    a.ins_("v_add_co_u32 {}, vcc, {}, {}", "s[1]", "lo[1]", "hi[0]")
    for j in range(2, 8):
        a.ins_("v_addc_co_u32 {}, vcc, {}, {}, vcc", f"s[{j}]", f"lo[{j}]", f"hi[{j - 1}]")
    a.ins_("v_addc_co_u32 {}, vcc, 0, {}, vcc", "s[8]", "hi[7]")
    L.append(a.render())
    L.append("    s[0] = lo[0];\n")
    # fold limb 8 (< 2^32): s8 * C = s8 * 977 + (s8 << 32)
    L.append("    u64 m = mul_wide_s(s[8], VK_PC0);\n    u32 ml = (u32)m, mh = (u32)(m >> 32);\n")
    L.append("    u32 w1, w2;\n    vk_carry c4;\n")
    a = Asm()
    for e in ("w1", "w2"):
        a.out("=&v", e)
    for j in range(8):
        a.out("=&v", f"r.v[{j}]")
    a.out("=&s", "c4")
    for e in ("mh", "ml") + tuple(f"s[{j}]" for j in range(9)):
        a.inp("v", e)
    a.ins_("v_add_co_u32 {}, vcc, {}, {}", "w1", "mh", "s[8]")          # limb 1 addend, carry k
    a.ins_("v_addc_co_u32 {}, vcc, 0, 0, vcc", "w2")                    # k: limb 2 addend
    a.ins_("v_add_co_u32 {}, vcc, {}, {}", "r.v[0]", "s[0]", "ml")
    a.ins_("v_addc_co_u32 {}, vcc, {}, {}, vcc", "r.v[1]", "s[1]", "w1")
    a.ins_("v_addc_co_u32 {}, vcc, {}, {}, vcc", "r.v[2]", "s[2]", "w2")
    for j in range(3, 7):
        a.ins_("v_addc_co_u32 {}, vcc, 0, {}, vcc", f"r.v[{j}]", f"s[{j}]")
    a.ins_("v_addc_co_u32 {}, {}, 0, {}, vcc", "r.v[7]", "c4", "s[7]")
    L.append(a.render())
    L.append("    if (__builtin_expect(c4 != 0, 0)) fe_wrap_add(r, c4); // value was 2^256 + small: add C (cannot carry again)\n    }\n}\n\n")
    return "".join(L)


def gen_sqr_tail():
    """t = 2 * od + sum a_i^2 2^(64 i): the eight diagonal products take the literal 0 as addend (no register to clear),
    the doubling and the addition run as two carry chains of eight limbs in one asm statement each."""
    L = ["// t = 2 * od + diag(a): od = the off-diagonal sum (< 2^511), see fe_sqr\n"]
    L.append("VK_DEV void fe_sqr_tail(u32 t[16], const u32 od[16], const fe &a)\n{\n    u64 d[8];\n")
    for half in range(2):
        x = Asm()
        for i in range(4 * half, 4 * half + 4):
            x.out("=&v", f"d[{i}]")
        for i in range(4 * half, 4 * half + 4):
            x.inp("v", f"a.v[{i}]")
        for i in range(4 * half, 4 * half + 4):
            x.ins_("v_mad_u64_u32 {}, vcc, {}, {}, 0", f"d[{i}]", f"a.v[{i}]", f"a.v[{i}]")
        L.append(x.render())
    L.append("    u32 dl[8], dh[8];\n#pragma unroll\n    for (int i = 0; i < 8; i++) {\n        dl[i] = (u32)d[i];\n        dh[i] = (u32)(d[i] >> 32);\n    }\n    vk_carry c;\n")
    for half in range(2):
        x = Asm()
        lo, hi = 8 * half, 8 * half + 8
        for j in range(lo, hi):
            x.out("=&v", f"t[{j}]")
        if half == 0:
            x.out("=&s", "c")
        for j in range(max(lo - 1, 0), hi):
            x.inp("v", f"od[{j}]")
        for i in range(4 * half, 4 * half + 4):
            x.inp("v", f"dl[{i}]")
            x.inp("v", f"dh[{i}]")
        if half == 1:
            x.inp("s", "c")
        # doubled limbs first (into t), then the chain adds the diagonal
        for j in range(lo, hi):
            if j == 0:
                x.ins_("v_lshlrev_b32 {}, 1, {}", "t[0]", "od[0]")
            else:
                x.ins_("v_alignbit_b32 {}, {}, {}, 31", f"t[{j}]", f"od[{j}]", f"od[{j - 1}]")
        for j in range(lo, hi):
            dsrc = f"dl[{j // 2}]" if j % 2 == 0 else f"dh[{j // 2}]"
            if j == 0:
                x.ins_("v_add_co_u32 {}, vcc, {}, {}", "t[0]", "t[0]", dsrc)
            elif j == 8:
                x.ins_("v_addc_co_u32 {}, vcc, {}, {}, {}", f"t[{j}]", f"t[{j}]", dsrc, "c")
            elif j == 7:
                x.ins_("v_addc_co_u32 {}, {}, {}, {}, vcc", f"t[{j}]", "c", f"t[{j}]", dsrc)
            else:
                x.ins_("v_addc_co_u32 {}, vcc, {}, {}, vcc", f"t[{j}]", f"t[{j}]", dsrc)
        L.append(x.render())
    L.append("}\n\n")
    return "".join(L)


def gen_sub3():
    """r = a - b - 2c: three borrow chains, the borrows counted, one fix-up of count * (2^32 + 977)."""
    L = ["// r = a - b - 2c (mod p), weakly reduced: the x-coordinate of every point addition.  r may alias a, b or c.\n"]
    L.append("VK_DEV void fe_sub3(fe &r, const fe &a, const fe &b, const fe &c)\n{\n    u32 d[8], e[8], f[8];\n    vk_carry c1, c2, c3;\n")
    A, B, C = ([f"{n}.v[{j}]" for j in range(8)] for n in "abc")
    D, E, F = ([f"{n}[{j}]" for j in range(8)] for n in "def")
    L.append(chain(D, A, B, carry_out="c1", sub=True))
    L.append(chain(E, D, C, carry_out="c2", sub=True))
    L.append(chain(F, E, C, carry_out="c3", sub=True))
    L.append("    u32 k, k0;\n    vk_carry c4;\n")
    a = Asm()
    for x in ("k", "k0"):
        a.out("=&v", x)
    for j in range(2):
        a.out("=&v", f"r.v[{j}]")
    a.out("=&s", "c4")
    for j in range(2):
        a.inp("v", F[j])
    for x in ("c1", "c2", "c3"):
        a.inp("s", x)
    a.inp("v", "VK_PC0")
    a.ins_("v_cndmask_b32 {}, 0, 1, {}", "k", "c1")
    a.ins_("v_addc_co_u32 {}, vcc, {}, 0, {}", "k", "k", "c2")
    a.ins_("v_addc_co_u32 {}, vcc, {}, 0, {}", "k", "k", "c3")
    a.ins_("v_mul_u32_u24 {}, {}, {}", "k0", "k", "VK_PC0")
    a.ins_("v_sub_co_u32 {}, vcc, {}, {}", "r.v[0]", F[0], "k0")
    a.ins_("v_subb_co_u32 {}, {}, {}, {}, vcc", "r.v[1]", "c4", F[1], "k")
    L.append(a.render())
    L.append("#pragma unroll\n    for (int i = 2; i < 8; i++) r.v[i] = f[i];\n")
    L.append("    if (__builtin_expect(c4 != 0, 0)) fe_ripple_sub(r, 2, c4); // the fix-up borrowed past limb 1\n}\n\n")
    return "".join(L)


def gen_reduce_flags(limb16=False):
    """(limb16: fe_reduce17_flags, the same reduction for the 17-limb sum of fe_mul2_cols -- limb 16, 0 or 1, is worth
    2^512 = C^2 = 2^64 + 1954 * 2^32 + 954529 (mod p) and joins limbs 0..2 of the result in a second short chain whose carry-out
    is one more flag.)
    fe_reduce16_flags: the reduction with NO precondition test in the instruction stream.  Every place where the short
    form can lose a carry -- the eight multiply-adds t[8+i] * 977 + (t[8+i] : t[i]) (overflow of 64 bits: one product in
    2^22), the carry out of limb 8 of their sum, the multiply-add s8 * 977 + (s8 : s0) that folds limb 8, and the carry out
    of limb 2 of the fold (one in 2^31) -- writes its carry-out to an SGPR pair; the pairs are OR-ed on the scalar unit
    (no VALU cycles; each OR sits a statement behind the VALU instruction that produced its operand) and returned.  A
    non-zero result (any lane of the wave) means: recompute this product exactly (fe_mul_exact / fe_sqr_exact); zero means r
    is the exact weakly reduced residue.  19 VALU instructions against 32 for fe_reduce16: the v_max3 tree, the separate
    s8 * 977 product and limbs 3..7 of the fold's carry chain are gone."""
    L = []
    L.append("// r = t_lo + t_hi * (2^32 + 977) when the return value is 0; otherwise some lane lost a carry: recompute exactly\n")
    if limb16:
        L.append("// (17 limbs: t[16] * C^2 is added to limbs 0..2 at the end)\n")
        L.append("VK_DEV vk_carry fe_reduce17_flags(fe &r, const u32 t[17])\n{\n")
    else:
        L.append("VK_DEV vk_carry fe_reduce16_flags(fe &r, const u32 t[16])\n{\n")
    L.append("    u64 v[8];\n#pragma unroll\n    for (int i = 0; i < 8; i++) v[i] = ((u64)t[8 + i] << 32) | t[i];\n")
    L.append("    vk_carry k0, k1, k2, k3, k4, k5, k6, k7, ka, kb, kc, kd, ovf;\n")
    # statement A: mads 0..3, carry-outs k0..k3
    a = Asm()
    for i in range(4):
        a.out("+v", f"v[{i}]")
    for i in range(4):
        a.out("=&s", f"k{i}")
    for i in range(4):
        a.inp("v", f"t[{8 + i}]")
    a.inp("s", "VK_PC0")
    for i in range(4):
        a.ins_("v_mad_u64_u32 {}, {}, {}, {}, {}", f"v[{i}]", f"k{i}", f"t[{8 + i}]", "VK_PC0", f"v[{i}]")
    L.append(a.render())
    # statement B: mads 4..7, carry-outs k4..k7; then the (by now finished) k0..k3 are OR-ed into ka
    a = Asm()
    a.clob = ["vcc", "scc"]
    for i in range(4, 8):
        a.out("+v", f"v[{i}]")
    for i in range(4, 8):
        a.out("=&s", f"k{i}")
    a.out("=&s", "ka")
    for i in range(4, 8):
        a.inp("v", f"t[{8 + i}]")
    a.inp("s", "VK_PC0")
    for i in range(4):
        a.inp("s", f"k{i}")
    for i in range(4, 8):
        a.ins_("v_mad_u64_u32 {}, {}, {}, {}, {}", f"v[{i}]", f"k{i}", f"t[{8 + i}]", "VK_PC0", f"v[{i}]")
    a.ins_("s_or_b64 {}, {}, {}", "ka", "k0", "k1")
    a.ins_("s_or_b64 {}, {}, {}", "ka", "ka", "k2")
    a.ins_("s_or_b64 {}, {}, {}", "ka", "ka", "k3")
    L.append(a.render())
    L.append("    u32 lo[8], hi[8], s[9];\n#pragma unroll\n    for (int i = 0; i < 8; i++) {\n        lo[i] = (u32)v[i];\n        hi[i] = (u32)(v[i] >> 32);\n    }\n")
    # chain: s1..s8, carry out of limb 8 -> kb; k4..k7 join ka behind it
    a = Asm()
    a.clob = ["vcc", "scc"]
    for j in range(1, 9):
        a.out("=&v", f"s[{j}]")
    a.out("=&s", "kb")
    a.out("+s", "ka")
    for j in range(1, 8):
        a.inp("v", f"lo[{j}]")
    for j in range(8):
        a.inp("v", f"hi[{j}]")
    for i in range(4, 8):
        a.inp("s", f"k{i}")
    a.ins_("v_add_co_u32 {}, vcc, {}, {}", "s[1]", "lo[1]", "hi[0]")
    for j in range(2, 8):
        a.ins_("v_addc_co_u32 {}, vcc, {}, {}, vcc", f"s[{j}]", f"lo[{j}]", f"hi[{j - 1}]")
    a.ins_("v_addc_co_u32 {}, {}, 0, {}, vcc", "s[8]", "kb", "hi[7]")
    for i in range(4, 8):
        a.ins_("s_or_b64 {}, {}, {}", "ka", "ka", f"k{i}")
    L.append(a.render())
    # fold limb 8 with one multiply-add: f = s8 * 977 + (s8 : s0)
    L.append("    u64 f = ((u64)s[8] << 32) | lo[0];\n")
    one = '    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(f), "=&s"(kc) : "v"(s[8]), "s"(VK_PC0));\n'
    host_register(one, ["v_mad_u64_u32 %0, %1, %2, %3, %0"], ["f", "kc", "s[8]", "VK_PC0"])
    L.append(one)
    L.append("    const u32 fl = (u32)f, fh = (u32)(f >> 32);\n")
    if limb16:
        # chain A: (r1a, r2a) = (s1 + f.hi, s2 + carry) -> flag kd; chain B: (r0, r1, r2) = (fl, r1a, r2a) + t16 * (954529, 1954, 1) -> flag ke
        L.append("    u32 q0, q1, r1a, r2a;\n    vk_carry ke;\n")
        a = Asm()
        a.clob = ["vcc", "scc"]
        for e in ("q0", "q1", "r1a", "r2a", "r.v[0]", "r.v[1]", "r.v[2]"):
            a.out("=&v", e)
        for e in ("kd", "ke", "ovf"):
            a.out("=&s", e)
        for e in ("s[1]", "s[2]", "fh", "fl", "t[16]"):
            a.inp("v", e)
        for x in ("ka", "kb", "kc"):
            a.inp("s", x)
        a.ins_("v_mul_u32_u24 {}, 0xE90A1, {}", "q0", "t[16]")              # 977^2
        a.ins_("v_mul_u32_u24 {}, 1954, {}", "q1", "t[16]")                 # 2 * 977
        a.ins_("v_add_co_u32 {}, vcc, {}, {}", "r1a", "s[1]", "fh")
        a.ins_("v_addc_co_u32 {}, {}, 0, {}, vcc", "r2a", "kd", "s[2]")
        a.ins_("v_add_co_u32 {}, vcc, {}, {}", "r.v[0]", "fl", "q0")
        a.ins_("v_addc_co_u32 {}, vcc, {}, {}, vcc", "r.v[1]", "r1a", "q1")
        a.ins_("v_addc_co_u32 {}, {}, {}, {}, vcc", "r.v[2]", "ke", "r2a", "t[16]")
        a.ins_("s_or_b64 {}, {}, {}", "ovf", "ka", "kb")
        a.ins_("s_or_b64 {}, {}, {}", "ovf", "ovf", "kc")
        a.ins_("s_or_b64 {}, {}, {}", "ovf", "ovf", "kd")
        a.ins_("s_or_b64 {}, {}, {}", "ovf", "ovf", "ke")
        L.append(a.render())
        L.append("#pragma unroll\n    for (int i = 3; i < 8; i++) r.v[i] = s[i];\n    return ovf;\n}\n\n")
        return "".join(L)
    # r1 = s1 + f.hi, r2 = s2 + carry; the carry out of limb 2 is the last flag
    a = Asm()
    a.clob = ["vcc", "scc"]
    a.out("=&v", "r.v[1]")
    a.out("=&v", "r.v[2]")
    a.out("=&s", "kd")
    a.out("=&s", "ovf")
    a.inp("v", "s[1]")
    a.inp("v", "s[2]")
    a.inp("v", "fh")
    for x in ("ka", "kb", "kc"):
        a.inp("s", x)
    a.ins_("v_add_co_u32 {}, vcc, {}, {}", "r.v[1]", "s[1]", "fh")
    a.ins_("v_addc_co_u32 {}, {}, 0, {}, vcc", "r.v[2]", "kd", "s[2]")
    a.ins_("s_or_b64 {}, {}, {}", "ovf", "ka", "kb")
    a.ins_("s_or_b64 {}, {}, {}", "ovf", "ovf", "kc")
    a.ins_("s_or_b64 {}, {}, {}", "ovf", "ovf", "kd")
    L.append(a.render())
    L.append("    r.v[0] = fl;\n#pragma unroll\n    for (int i = 3; i < 8; i++) r.v[i] = s[i];\n    return ovf;\n}\n\n")
    return "".join(L)


def main():
    hdr = "// GENERATED by tools/gen_field.py -- do not edit.  Column kernels of the 256-bit multiply / square\n"
    hdr += "// on v_mad_u64_u32 (see the scheme in vk_field.cuh).\n#pragma once\n\n"
    txt = hdr + gen_cols("fe_mul_cols", col_products_mul, False) + gen_cols("fe_sqr_cols", col_products_sqr, True) + gen_merge() + gen_sqr_tail() + gen_addsub() + gen_sub3() + gen_reduce() + gen_reduce_fast() + gen_reduce_flags() + gen_cols2() + gen_merge17() + gen_reduce_flags(limb16=True)
    open(OUT, "w").write(txt)
    print("wrote", OUT, len(txt.splitlines()), "lines")
    host = txt.replace("// GENERATED by tools/gen_field.py -- do not edit.", "// GENERATED by tools/gen_field.py -- do not edit.  HOST form: every asm statement of vuke_amd/csrc/vk_field_gen.cuh as\n// calls into the one-lane instruction emulator of vk_field_prims_host.h; test infrastructure, never part of the device build.", 1)
    n = 0
    for a, h in HOST.items():
        assert a in host, a[:200]
        n += host.count(a)
        host = host.replace(a, h)
    assert "asm(" not in host, "a statement was not registered for the host backend"
    open(OUT_HOST, "w").write(host)
    print("wrote", OUT_HOST, n, "statements translated")


if __name__ == "__main__":
    main()
