#!/usr/bin/env python3
"""Generate fp256_gen.inc: the 256x256->512 Comba product and the 512->256 fold for gfx950.

Why generated asm: on gfx950 `v_mad_u64_u32` (32x32+64->64, carry-out to an SGPR pair) issues at half
rate (profiles/r01_microbench.jsonl) and has no carry-in, and a VALU that reads an SGPR/VCC written by a
VALU needs 2 wait states in between (hipcc inserts s_nop for its own code, never inside asm).  So every
column of the product is ONE asm statement in which the carry of multiply m is consumed by a
`v_addc_co_u32` placed after multiply m+2: no bubbles, no hazards.  hipcc's own lowering of the same
C++ (u128 arithmetic) costs about 2x the issue slots.

Run:  python gen_fp256.py > fp256_gen.inc   (done by the Makefile / __graft_entry__.build()).
"""
import sys


def column_asm(products, a_name, b_name, first_col, can_carry=True):
    """products: list of (i, j). Returns C++ text of one asm statement accumulating into acc/c2.
    can_carry=False: the column provably stays below 2^64 (first / top column): no carry counting at all."""
    n = len(products)
    if not can_carry:
        a_idx = sorted({i for i, _ in products})
        b_idx = sorted({j for _, j in products})
        opn, k = {}, 1
        for i in a_idx:
            opn[("a", i)] = k
            k += 1
        for j in b_idx:
            opn[("b", j)] = k
            k += 1
        body = "\\n\\t".join("v_mad_u64_u32 %%0, vcc, %%%d, %%%d, %%0" % (opn[("a", i)], opn[("b", j)]) for i, j in products)
        ins = ", ".join(['"v"(%s[%d])' % (a_name, i) for i in a_idx] + ['"v"(%s[%d])' % (b_name, j) for j in b_idx])
        return '    asm("%s" : "+v"(acc) : %s : "vcc");\n    c2 = 0;\n' % (body, ins)
    a_idx = sorted({i for i, _ in products})
    b_idx = sorted({j for _, j in products})
    # operand numbering: 0 acc, 1 c2, 2..4 carry sgpr pairs, then a's, then b's
    opn = {}
    k = 5
    for i in a_idx:
        opn[("a", i)] = k
        k += 1
    for j in b_idx:
        opn[("b", j)] = k
        k += 1
    lines = []
    cy = [2, 3, 4]
    pend = []  # carry operand numbers waiting for their addc

    first = [True]

    def addc(c):
        # the first carry of a column also initialises the column's carry counter (no v_mov c2, 0)
        lines.append("v_addc_co_u32_e64 %%1, vcc, 0, %s, %%%d" % ("0" if first[0] else "%1", c))
        first[0] = False

    for m, (i, j) in enumerate(products):
        c = cy[m % 3]
        lines.append("v_mad_u64_u32 %%0, %%%d, %%%d, %%%d, %%0" % (c, opn[("a", i)], opn[("b", j)]))
        pend.append(c)
        if m >= 2:
            addc(pend.pop(0))
    # drain
    if n == 1:
        lines.append("s_nop 1")
    elif n == 2:
        lines.append("s_nop 0")
    while pend:
        addc(pend.pop(0))
    ins = ", ".join(['"v"(%s[%d])' % (a_name, i) for i in a_idx] + ['"v"(%s[%d])' % (b_name, j) for j in b_idx])
    txt = '    asm("%s"\n        : "+v"(acc), "=&v"(c2), "=&s"(cyA), "=&s"(cyB), "=&s"(cyC)\n        : %s : "vcc");\n' % (
        "\\n\\t".join(lines), ins)
    return txt


def column_asm_salu(products, a_name, b_name):
    """One Comba column (>= 3 products, may carry) whose carry-outs are counted by the SCALAR unit: every multiply-add leaves its carry-out as a 64-lane mask in an SGPR
    pair; the masks are compressed pairwise by a bit-sliced adder on the SALU (p0 = parity plane, k = one weight-2 carry mask per pair: 5 scalar instructions) and the
    vector unit only adds one mask per PAIR (v_addc with the mask as carry-in) plus one closing step c2 = 2*c2 + p0.  n products: n//2 + 1 vector carry steps instead of n (an odd column
    starts with a full adder of three masks).  SALU reads of VALU-written SGPRs and VALU carry-ins written by the SALU are interlocked by the hardware (no wait states in the gfx9 hazard table);
    the scalar instructions of one wave issue beside the vector instructions of the SIMD's other waves."""
    n = len(products)
    assert n >= 3
    a_idx = sorted({i for i, _ in products})
    b_idx = sorted({j for _, j in products})
    # operands: 0 acc, 1 c2, 2 mA, 3 mB, 4 p0, 5 k, 6 x, 7 t, then a's, b's
    opn, k = {}, 8
    for i in a_idx:
        opn[("a", i)] = k
        k += 1
    for j in b_idx:
        opn[("b", j)] = k
        k += 1
    L = []
    first_add = [True]

    def mad(m, dst):
        i, j = products[m]
        L.append("v_mad_u64_u32 %%0, %%%d, %%%d, %%%d, %%0" % (dst, opn[("a", i)], opn[("b", j)]))

    def addk():
        L.append("v_addc_co_u32_e64 %%1, vcc, 0, %s, %%5" % ("0" if first_add[0] else "%1"))
        first_add[0] = False

    m = 0
    pair = 0
    if n & 1:
        # an odd column starts with a TRIPLE: a full adder of three masks (p0 = parity, k = majority), so that no single mask is left over at the end
        mad(0, 2)
        mad(1, 3)
        mad(2, 6)
        L.append("s_xor_b64 %7, %2, %3")
        L.append("s_and_b64 %5, %2, %3")
        L.append("s_xor_b64 %4, %7, %6")
        L.append("s_and_b64 %7, %7, %6")
        L.append("s_or_b64 %5, %5, %7")
        addk()
        m = 3
        pair = 1
    while m + 1 < n:
        mad(m, 2)
        mad(m + 1, 3)
        if pair == 0:
            L.append("s_xor_b64 %4, %2, %3")
            L.append("s_and_b64 %5, %2, %3")
        else:
            L.append("s_xor_b64 %6, %2, %3")
            L.append("s_and_b64 %5, %2, %3")
            L.append("s_and_b64 %7, %4, %6")
            L.append("s_or_b64 %5, %5, %7")
            L.append("s_xor_b64 %4, %4, %6")
        addk()
        m += 2
        pair += 1
    assert m == n
    L.append("v_addc_co_u32_e64 %1, vcc, %1, %1, %4")
    ins = ", ".join(['"v"(%s[%d])' % (a_name, i) for i in a_idx] + ['"v"(%s[%d])' % (b_name, j) for j in b_idx])
    return '    asm("%s"\n        : "+v"(acc), "=&v"(c2), "=&s"(mA), "=&s"(mB), "=&s"(p0), "=&s"(kk), "=&s"(xx), "=&s"(tt)\n        : %s : "vcc", "scc");\n' % ("\\n\\t".join(L), ins)


def gen_mul_salu():
    out = []
    out.append("// the same product with the carry-outs of every column of three or more products counted on the scalar unit (column_asm_salu)\n")
    out.append("__device__ __forceinline__ void fe_mul512_s(u32 (&r)[16], const u32 (&a)[8], const u32 (&b)[8])\n{\n")
    out.append("    u64 acc = 0; u32 c2; u64 cyA, cyB, cyC, mA, mB, p0, kk, xx, tt;\n")
    for k in range(15):
        prods = [(i, k - i) for i in range(8) if 0 <= k - i <= 7]
        if len(prods) >= 3:
            out.append(column_asm_salu(prods, "a", "b"))
        else:
            out.append(column_asm(prods, "a", "b", k == 0, can_carry=k not in (0, 14)))
        out.append("    r[%d] = (u32)acc; acc = (acc >> 32) | ((u64)c2 << 32);\n" % k)
    out.append("    r[15] = (u32)acc;\n}\n\n")
    return "".join(out)


def gen_mul():
    out = []
    out.append("// ---- generated by gen_fp256.py: do not edit ----\n")
    out.append("__device__ __forceinline__ void fe_mul512(u32 (&r)[16], const u32 (&a)[8], const u32 (&b)[8])\n{\n")
    out.append("    u64 acc = 0; u32 c2; u64 cyA, cyB, cyC;\n")
    for k in range(15):
        prods = [(i, k - i) for i in range(8) if 0 <= k - i <= 7]
        # column 0 starts from zero and the top column cannot leave 512 bits: nothing to carry
        out.append(column_asm(prods, "a", "b", k == 0, can_carry=k not in (0, 14)))
        out.append("    r[%d] = (u32)acc; acc = (acc >> 32) | ((u64)c2 << 32);\n" % k)
    out.append("    r[15] = (u32)acc;\n}\n\n")
    return "".join(out)


def gen_sqr():
    """a*a with 36 multiplies: the 28 cross products a_i*a_j (i<j) in Comba columns, doubled by a 1-bit funnel
    shift of the whole 512-bit sum, plus the 8 diagonal squares joined by one 16-word carry chain."""
    out = []
    out.append("__device__ __forceinline__ void fe_sqr512(u32 (&r)[16], const u32 (&a)[8])\n{\n")
    out.append("    u64 acc = 0; u32 c2; u64 cyA, cyB, cyC; u32 x[16];\n    x[0] = 0;\n")
    for k in range(1, 14):
        prods = [(i, k - i) for i in range(8) if 0 <= k - i <= 7 and i < k - i]
        # columns 1 and 2 hold one product on top of < 2^32: nothing to carry
        out.append(column_asm(prods, "a", "a", k == 1, can_carry=k not in (1, 2)))
        out.append("    x[%d] = (u32)acc; acc = (acc >> 32) | ((u64)c2 << 32);\n" % k)
    out.append("    x[14] = (u32)acc; x[15] = (u32)(acc >> 32);\n")
    out.append("    u64 d[8];\n")
    for i in range(8):
        out.append('    asm("v_mad_u64_u32 %%0, vcc, %%1, %%1, 0" : "=v"(d[%d]) : "v"(a[%d]) : "vcc");\n' % (i, i))
    out.append("    u32 c = 0, co;\n")
    for k in range(16):
        dbl = "(x[0] << 1)" if k == 0 else "__builtin_amdgcn_alignbit(x[%d], x[%d], 31)" % (k, k - 1)
        dw = "(u32)d[%d]" % (k // 2) if k % 2 == 0 else "(u32)(d[%d] >> 32)" % (k // 2)
        out.append("    r[%d] = __builtin_addc(%s, %s, c, &co); c = co;\n" % (k, dbl, dw))
    out.append("}\n\n")
    return "".join(out)


def gen_sqr_lo10():
    """words 0..9 of a*a, exactly: the cross products of columns 1..9 (22 of the 28), doubled, plus the diagonal squares a0^2..a4^2.
    What the low-64-bit x coordinate needs of lambda^2 (fp256.hip.h: fe_sqr_add2_lo64): words 0, 1, 7, 8, 9 and the carries through
    the words in between."""
    out = []
    out.append("__device__ __forceinline__ void fe_sqr_lo10(u32 (&r)[10], const u32 (&a)[8])\n{\n")
    out.append("    u64 acc = 0; u32 c2; u64 cyA, cyB, cyC; u32 x[10];\n    x[0] = 0;\n")
    for k in range(1, 10):
        prods = [(i, k - i) for i in range(8) if 0 <= k - i <= 7 and i < k - i]
        out.append(column_asm(prods, "a", "a", k == 1, can_carry=k not in (1, 2)))
        if k < 9:
            out.append("    x[%d] = (u32)acc; acc = (acc >> 32) | ((u64)c2 << 32);\n" % k)
        else:
            out.append("    x[%d] = (u32)acc;\n" % k)
    out.append("    u64 d[5];\n")
    for i in range(5):
        out.append('    asm("v_mad_u64_u32 %%0, vcc, %%1, %%1, 0" : "=v"(d[%d]) : "v"(a[%d]) : "vcc");\n' % (i, i))
    out.append("    u32 c = 0, co;\n")
    for k in range(10):
        dbl = "(x[0] << 1)" if k == 0 else "__builtin_amdgcn_alignbit(x[%d], x[%d], 31)" % (k, k - 1)
        dw = "(u32)d[%d]" % (k // 2) if k % 2 == 0 else "(u32)(d[%d] >> 32)" % (k // 2)
        out.append("    r[%d] = __builtin_addc(%s, %s, c, &co); c = co;\n" % (k, dbl, dw))
    out.append("}\n\n")
    return "".join(out)


if __name__ == "__main__":
    sys.stdout.write(gen_mul())
    sys.stdout.write(gen_mul_salu())
    sys.stdout.write(gen_sqr())
    sys.stdout.write(gen_sqr_lo10())
