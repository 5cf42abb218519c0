#!/usr/bin/env python
"""Prints the inline-asm bodies of mh_key_block4<NI, KN> (monohair_amd/csrc/pmvo_search.hip): four taps x NI items of the
search kernel's key body as one hand-ordered block per NI = 4, 3, 2, 1.

Order inside a half (two taps of the same parity, items j = 0..NI-1): the instructions of one kind for all items, then the
next kind, so that every result is used NI - 1 or more instructions after it is produced; the second tap's products are
issued between the first tap's sum and subtraction.  Even taps g[0], g[2] fold into ke, odd taps g[1], g[3] into ko.

    python tools/gen_key_blocks.py > /tmp/blocks.txt     # paste between the braces of mh_key_block4
"""


def block4(n):
    lines = []

    def half(ta, tb, acc):   # taps ta (place i0) and tb (place i1) into accumulator acc
        r = range(n)
        for j in r: lines.append(f"v_mul_f32_e32 %[a{j}], %[t{ta}x], %[x{j}]")
        for j in r: lines.append(f"v_mul_f32_e32 %[b{j}], %[t{ta}y], %[y{j}]")
        for j in r: lines.append(f"v_mul_f32_e32 %[c{j}], %[t{tb}x], %[x{j}]")
        for j in r: lines.append(f"v_add_f32_e32 %[a{j}], %[a{j}], %[b{j}]")
        for j in r: lines.append(f"v_mul_f32_e32 %[b{j}], %[t{tb}y], %[y{j}]")
        for j in r: lines.append(f"v_sub_f32_e64 %[a{j}], %[cc], |%[a{j}]|")
        for j in r: lines.append(f"v_add_f32_e32 %[c{j}], %[c{j}], %[b{j}]")
        for j in r: lines.append(f"v_lshl_or_b32 %[a{j}], %[a{j}], 5, %[i0]")
        for j in r: lines.append(f"v_sub_f32_e64 %[c{j}], %[cc], |%[c{j}]|")
        for j in r: lines.append(f"v_lshl_or_b32 %[c{j}], %[c{j}], 5, %[i1]")
        for j in r: lines.append(f"v_min3_u32 %[{acc}{j}], %[{acc}{j}], %[a{j}], %[c{j}]")

    half(0, 2, "e")
    half(1, 3, "o")
    return lines


def emit(n):
    lines = block4(n)
    body = "\n".join('            "%s%s"' % (l, "\\n\\t" if i < len(lines) - 1 else "") for i, l in enumerate(lines))
    outs = ", ".join([f'[e{j}] "+v"(ke[{j}])' for j in range(n)] + [f'[o{j}] "+v"(ko[{j}])' for j in range(n)] +
                     [f'[{c}{j}] "=&v"({c}[{j}])' for c in "abc" for j in range(n)])
    ins = ", ".join([f'[t{u}x] "v"(g[{u}].x), [t{u}y] "v"(g[{u}].y)' for u in range(4)] +
                    [f'[x{j}] "v"(DX[{j}]), [y{j}] "v"(DY[{j}])' for j in range(n)] +
                    ['[cc] "s"(MH_KEY_C)', '[i0] "s"(ib)', '[i1] "s"(ib + 1)'])
    head = "if" if n == 4 else "else if"
    return (f"    {head} constexpr (NI == {n}) {{\n        asm volatile(\n{body}\n            : {outs}\n            : {ins}\n"
            f'            : "memory");\n    }}\n')


if __name__ == "__main__":
    print("".join(emit(n) for n in (4, 3, 2, 1)), end="")
