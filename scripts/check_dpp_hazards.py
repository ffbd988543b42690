#!/usr/bin/env python3
"""Build-time check of the inline-assembly DPP instructions in kernels.hip (hipcc -S output):
a DPP operand (src0) must not have been written by a vector-ALU instruction in the two issue
slots in front of it (gfx9 'VALU write VGPR -> DPP read' hazard, 2 wait states).  The compiler
pads its own DPP instructions but does not look into inline assembly, so the asm pads itself
(fmac_bc_pivot) and this script checks the result.  Usage: check_dpp_hazards.py kernels.s [kernel ...]"""
import re
import sys


def regs(tok):
    tok = tok.strip()
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2:]
    bad_total = 0
    for m in re.finditer(r'^(_ZN3spx\w+):.*?\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want and not any(w in name for w in want):
            continue
        lines = [l.strip() for l in body.split('\n')]
        lines = [l for l in lines if l and not l.startswith(';') and not l.startswith('.') and not l.endswith(':')]
        ndpp = sum('_dpp' in l for l in lines)
        if not ndpp:
            continue
        bad = 0
        for i, l in enumerate(lines):
            if '_dpp' not in l:
                continue
            ops = l.split(None, 1)[1].split(',')
            src = regs(ops[1].split()[0])
            # wait states: s_nop N counts N + 1, any other instruction 1
            ws = 0
            for p in reversed(lines[max(0, i - 4):i]):
                if ws >= 2:
                    break
                if p.startswith('s_nop'):
                    ws += int(p.split()[1]) + 1
                    continue
                if p.startswith('v_') and regs(p.split(None, 1)[1].split(',')[0]) & src:
                    bad += 1
                    print('  HAZARD', name[:40], ':', p, '->', l)
                    break
                ws += 1
        # transcendental result forwarding: the instruction right behind a v_rcp / v_rsq / v_sqrt_f64 must
        # not read its destination (1 wait state; the inline assembly carries an s_nop 0 for it)
        for i, l in enumerate(lines[:-1]):
            if not re.match(r'v_(rcp|rsq|sqrt)_f64', l):
                continue
            dst = regs(l.split(None, 1)[1].split(',')[0])
            nxt = lines[i + 1]
            if nxt.startswith('v_') and len(nxt.split(None, 1)) > 1:
                srcs = set()
                for tok in nxt.split(None, 1)[1].split(',')[1:]:
                    srcs |= regs(tok.strip().lstrip('-|').split()[0].rstrip('|')) if tok.strip() else set()
                if dst & srcs:
                    bad += 1
                    print('  HAZARD (trans result forwarding)', name[:40], ':', l, '->', nxt)
        nfmac = sum('v_fmac_f64_dpp' in l for l in lines)
        nscr = sum('scratch_' in l for l in lines)
        print(f'{name[:60]}: {len(lines)} instructions, {ndpp} DPP ({nfmac} v_fmac_f64_dpp), {nscr} scratch accesses, {bad} hazards')
        bad_total += bad
    return 1 if bad_total else 0


if __name__ == '__main__':
    sys.exit(main())
