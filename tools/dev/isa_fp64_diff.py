#!/usr/bin/env python3
"""isa_fp64_diff.py a.s b.s [filter] -- do two builds of the device code do the same floating-point arithmetic?
For every kernel present in both assembly files (hipcc -S --cuda-device-only) the sequence of fp64 opcodes is compared; with
`--many` the kernels NAME_many of ONE file are compared with NAME (same template arguments).  HIP contracts a*b+c across
statements (-ffp-contract=fast), so a refactoring that leaves the source arithmetic alone can still move a fused multiply-add:
the runs-in-step kernels must be the one-run kernels bit for bit, and this is the check that needs no GPU."""
import difflib, re, sys

def kernels(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1); out[cur] = []; continue
        if cur is None: continue
        if '.end_amdhsa_kernel' in line or line.startswith('\t.section'):
            cur = None; continue
        t = line.split()
        if t and re.match(r'v_\w*f64', t[0]): out[cur].append(t[0].replace('_e32', '').replace('_e64', '').replace('v_fmac_f64', 'v_fma_f64'))
    return out

def cmp(a, b):
    # the scheduler may reorder independent instructions: compare as multisets per window as well as in order
    if a == b: return "identical"
    from collections import Counter
    ca, cb = Counter(a), Counter(b)
    if ca == cb: return "same opcode counts, other order (%d ops)" % len(a)
    d = {k: cb.get(k, 0) - ca.get(k, 0) for k in set(ca) | set(cb) if cb.get(k, 0) != ca.get(k, 0)}
    return "DIFFERENT: " + ", ".join("%s %+d" % kv for kv in sorted(d.items()))

def main():
    args = [x for x in sys.argv[1:] if not x.startswith('--')]
    many = '--many' in sys.argv
    flt = args[2] if len(args) > 2 else (args[1] if many and len(args) > 1 else '')
    A = kernels(args[0])
    if many:
        for k in sorted(A):
            m = re.match(r'_Z(\d+)(\w+?)_many(I.*E)v', k) or re.match(r'_Z(\d+)(\w+?)_many()v?', k)
            if not m or (flt and flt not in k): continue
            base = m.group(2)
            solo = [q for q in A if re.match(r'_Z\d+' + re.escape(base) + re.escape(m.group(3)) + r'(v|E)', q) and '_many' not in q]
            if not solo: solo = [q for q in A if re.match(r'_Z\d+' + re.escape(base) + r'(I|v|7|P)', q) and '_many' not in q and (not m.group(3) or m.group(3) in q)]
            for q in solo[:1]: print("%-70s vs %-60s %s" % (k[:70], q[:60], cmp(A[q], A[k])))
        return
    B = kernels(args[1])
    for k in sorted(A):
        if k in B and (not flt or flt in k):
            r = cmp(A[k], B[k])
            if r != "identical" or '-v' in sys.argv: print("%-90s %s" % (k[:90], r))

if __name__ == "__main__":
    main()
