#!/usr/bin/env python3
"""Static instruction mix of one kernel (no GPU needed).  Input: the device assembly of a .hip file,
   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 --cuda-device-only -S file.hip -o file.s
   python3 scripts/isa_count.py file.s <substring of the mangled kernel name> [--loop]
--loop restricts the count to the largest backward-branch region (the row / pixel loop of the sweep kernels)."""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
loop = "--loop" in sys.argv
text = open(src).read().split("\n")
i = 0
while i < len(text):
    m = re.match(r"^(_Z\w+):", text[i])
    if not (m and pat in m.group(1)):
        i += 1
        continue
    name = m.group(1)
    body = []
    i += 1
    while i < len(text) and not text[i].startswith(".Lfunc_end"):
        body.append(text[i]); i += 1
    ins, labels = [], {}
    for l in body:
        lm = re.match(r"^(\.LBB\w+):", l)
        if lm:
            labels[lm.group(1)] = len(ins); continue
        im = re.match(r"^\s+([a-z]\w+)\s*(.*?)\s*(;.*)?$", l)
        if im and not im.group(1).startswith("."):
            ins.append((im.group(1), im.group(2)))
    lo, hi = 0, len(ins)
    if loop:
        best = (0, 0, 0)
        for k, (op, args) in enumerate(ins):
            if op.startswith("s_cbranch") or op == "s_branch":
                t = args.strip()
                if t in labels and labels[t] <= k and k - labels[t] > best[0]:
                    best = (k - labels[t], labels[t], k + 1)
        if best[0]:
            lo, hi = best[1], best[2]
    cls, ops = {}, {}
    for op, args in ins[lo:hi]:
        if op.startswith("v_"): c = "VALU"
        elif op.startswith("s_waitcnt"): c = "s_waitcnt"
        elif op.startswith("s_"): c = "SALU"
        elif op.startswith("ds_"): c = "LDS"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "VMEM"
        else: c = "other"
        cls[c] = cls.get(c, 0) + 1
        ops[op] = ops.get(op, 0) + 1
    top = sorted(ops.items(), key=lambda kv: -kv[1])[:24]
    print(name, "instructions %d%s" % (hi - lo, " (largest loop)" if loop else ""))
    print("  ", " ".join("%s=%d" % kv for kv in sorted(cls.items())))
    print("  ", " ".join("%s:%d" % kv for kv in top))
