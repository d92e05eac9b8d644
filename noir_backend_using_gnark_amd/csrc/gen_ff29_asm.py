#!/usr/bin/env python3
"""Emits ff29_mul_gfx950.inc: one Montgomery multiplication on UNSATURATED 9 x 29-bit limbs (radix R' = 2^261).

Why a second multiplier: measured on MI355X (tools/ubench*.hip) v_mad_u64_u32 issues at ~4.6 cycles per wave -- the same
price as the v_addc_co_u32 that has to follow it in the saturated 8 x 32-bit schedule (ff_mul_gfx950.inc: 128 mad + 127 addc).
With 29-bit limbs a 64-bit column accumulator absorbs all 18 products of a column without a carry instruction:
      162 v_mad_u64_u32 + 9 v_mul_lo_u32 + 17 v_lshrrev_b64 + 17 v_and_b32  = 205 instructions  (vs ~305).
tools/u29_model.py proves the accumulators cannot overflow for the operand bounds the accumulate kernels produce.

Schedule (product scanning): accumulator = v[0:1]; m_k = (acc * -p^-1) mod 2^29 kept in the output registers until the result
limb that reuses the register is produced; modulus limbs and -p^-1 are SGPR operands (VOP3 has no literals on gfx9-family).
"""
NL, W = 9, 29


def emit(N=1):
    """N > 1: sum of N products under ONE Montgomery reduction (operands %[a<t>_<i>], %[b<t>_<i>])."""
    L = []
    if N == 1:
        a = lambda i, t=0: "%%[a%d]" % i
        b = lambda i, t=0: "%%[b%d]" % i
    else:
        a = lambda i, t: "%%[a%d_%d]" % (t, i)
        b = lambda i, t: "%%[b%d_%d]" % (t, i)
    r = lambda i: "%%[r%d]" % i
    p = lambda i: "%%[p%d]" % i
    first = [True]

    def mac(x, y):
        if first[0]:
            L.append("v_mad_u64_u32 v[0:1], vcc, %s, %s, 0" % (x, y))
            first[0] = False
        else:
            L.append("v_mad_u64_u32 v[0:1], vcc, %s, %s, v[0:1]" % (x, y))

    for k in range(NL):
        for t in range(N):
            for i in range(k + 1):
                mac(a(i, t), b(k - i, t))
        for j in range(k):
            mac(r(j), p(k - j))
        L.append("v_mul_lo_u32 %s, v0, %%[ninv]" % r(k))
        L.append("v_and_b32_e32 %s, 0x1fffffff, %s" % (r(k), r(k)))
        mac(r(k), p(0))
        L.append("v_lshrrev_b64 v[0:1], 29, v[0:1]")
    for k in range(NL, 2 * NL - 1):
        for t in range(N):
            for i in range(k - NL + 1, NL):
                mac(a(i, t), b(k - i, t))
        for j in range(k - NL + 1, NL):
            mac(r(j), p(k - j))
        L.append("v_and_b32_e32 %s, 0x1fffffff, v0" % r(k - NL))
        L.append("v_lshrrev_b64 v[0:1], 29, v[0:1]")
    L.append("v_mov_b32_e32 %s, v0" % r(NL - 1))
    return L


def emit_sqr():
    """a * a with the cross products taken once against the doubled operand d = 2a (operands %[a<i>], %[d<i>]): column k sums
    a_i * d_(k-i) over i < k-i plus a_(k/2)^2 -- 45 multiply-adds instead of 81; the reduction half is unchanged."""
    L = []
    a = lambda i: "%%[a%d]" % i
    d = lambda i: "%%[d%d]" % i
    r = lambda i: "%%[r%d]" % i
    p = lambda i: "%%[p%d]" % i
    first = [True]

    def mac(x, y):
        if first[0]:
            L.append("v_mad_u64_u32 v[0:1], vcc, %s, %s, 0" % (x, y))
            first[0] = False
        else:
            L.append("v_mad_u64_u32 v[0:1], vcc, %s, %s, v[0:1]" % (x, y))

    def column(k):
        lo, hi = max(0, k - NL + 1), min(k, NL - 1)
        for i in range(lo, hi + 1):
            j = k - i
            if i < j:
                mac(a(i), d(j))
            elif i == j:
                mac(a(i), a(i))

    for k in range(NL):
        column(k)
        for j in range(k):
            mac(r(j), p(k - j))
        L.append("v_mul_lo_u32 %s, v0, %%[ninv]" % r(k))
        L.append("v_and_b32_e32 %s, 0x1fffffff, %s" % (r(k), r(k)))
        mac(r(k), p(0))
        L.append("v_lshrrev_b64 v[0:1], 29, v[0:1]")
    for k in range(NL, 2 * NL - 1):
        column(k)
        for j in range(k - NL + 1, NL):
            mac(r(j), p(k - j))
        L.append("v_and_b32_e32 %s, 0x1fffffff, v0" % r(k - NL))
        L.append("v_lshrrev_b64 v[0:1], 29, v[0:1]")
    L.append("v_mov_b32_e32 %s, v0" % r(NL - 1))
    return L


def rename_stream(lines, t):
    """Operands of one single-product stream renamed for slot t of a two-product block: %[aI] -> %[aI_t], %[bI] / %[dI] / %[rI] likewise;
    slot 1 accumulates in v[2:3] instead of v[0:1] (modulus limbs and ninv are shared)."""
    import re
    out = []
    for l in lines:
        l = re.sub(r"%\[([abdr])(\d)\]", lambda m: "%%[%s%s_%d]" % (m.group(1), m.group(2), t), l)
        if t == 1:
            l = l.replace("v[0:1]", "v[2:3]").replace(" v0", " v2")
        out.append(l)
    return out


def emit_pair(sqr=False):
    """TWO INDEPENDENT products in one block, their instruction streams interleaved one to one: every product is a single dependent chain of
    multiply-adds on one accumulator, and four waves per SIMD do not quite cover that chain's latency (990 cycles per product at 4 waves, 915 at 8:
    tools/ubench4.hip) -- a second, independent chain in the same wave does, at no extra instruction."""
    a = rename_stream(emit_sqr() if sqr else emit(1), 0)
    b = rename_stream(emit_sqr() if sqr else emit(1), 1)
    out = []
    for x, y in zip(a, b):
        out += [x, y]
    return out


def main():
    print("// GENERATED by gen_ff29_asm.py -- do not edit; see that file for the schedule.")
    for name, lines in (("ZKMI_MONT_MUL29_X2_ASM", emit_pair(False)), ("ZKMI_MONT_SQR29_X2_ASM", emit_pair(True))):
        print("// two independent %s interleaved: %d instructions" % ("squarings" if "SQR" in name else "products", len(lines)))
        print("#define %s \\" % name)
        for i, l in enumerate(lines):
            end = " \\" if i + 1 < len(lines) else ""
            print('    "%s\\n\\t"%s' % (l, end))
    lines = emit_sqr()
    print("// squaring: %d instructions: %d v_mad_u64_u32" % (len(lines), sum("v_mad_u64" in l for l in lines)))
    print("#define ZKMI_MONT_SQR29_ASM \\")
    for i, l in enumerate(lines):
        end = " \\" if i + 1 < len(lines) else ""
        print('    "%s\\n\\t"%s' % (l, end))
    for N in (1, 2, 3, 4):
        lines = emit(N)
        print("// N = %d: %d instructions: %d v_mad_u64_u32, %d v_mul_lo_u32, %d v_lshrrev_b64, %d v_and_b32" % (
            N, len(lines), sum("v_mad_u64" in l for l in lines), sum("v_mul_lo" in l for l in lines),
            sum("v_lshrrev_b64" in l for l in lines), sum("v_and_b32" in l for l in lines)))
        print("#define ZKMI_MONT_MUL29%s_ASM \\" % ("" if N == 1 else "_N%d" % N))
        for i, l in enumerate(lines):
            end = " \\" if i + 1 < len(lines) else ""
            print('    "%s\\n\\t"%s' % (l, end))


if __name__ == "__main__":
    main()
