"""Dev: print the instruction stream of one kernel from a hipcc --save-temps .s file in compressed form (one letter per instruction: M matrix,
v vector ALU, e transcendental, L LDS read, W LDS write, G vector memory, D LDS-DMA, s scalar, w s_waitcnt, B barrier, n s_nop, b branch),
per basic block:  python3 scratch/isa_stream.py file.s kernel_substring [--full]"""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
full = "--full" in sys.argv
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cls(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): return "e"
    if op.startswith("v_"): return "v"
    if op.startswith("ds_read") or op.startswith("ds_load") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"): return "L"
    if op.startswith("ds_"): return "W"
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and "lds" in op): return "D"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "G"
    if op == "s_waitcnt": return "w"
    if op == "s_barrier": return "B"
    if op == "s_nop": return "n"
    if op.startswith(("s_cbranch", "s_branch")): return "b"
    if op.startswith("s_"): return "s"
    return "?"
blk, out, counts = "entry", [], {}
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith((";", "//", ".")): 
        if re.match(r"^\.LBB\d+_\d+:", t):
            out.append((blk, counts)); blk, counts = t.split(":")[0], {}
            out[-1] = out[-1]
        continue
    if re.match(r"^\.?LBB\d+_\d+:", t):
        out.append((blk, counts)); blk, counts = t.split(":")[0], {}
        continue
    op = t.split()[0]
    c = cls(op)
    counts.setdefault("stream", []).append(c if not full else t)
    counts[c] = counts.get(c, 0) + 1
out.append((blk, counts))
for name, c in out:
    st = c.pop("stream", [])
    if not st: continue
    print("== %s  n=%d  %s" % (name, len(st), " ".join("%s:%d" % kv for kv in sorted(c.items()))))
    if full: print("\n".join(st))
    else:
        s = "".join(st)
        for i in range(0, len(s), 160): print("   " + s[i:i + 160])
