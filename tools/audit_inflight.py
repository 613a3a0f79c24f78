"""Static audit of asm-issued loads (cdna_hip_programming.md, 'What hipcc does not do', item 1): hipcc treats the VGPR destination
of an inline-asm load as written at the end of the statement, so it may read, copy or reuse it before the data lands.  For every
loop of a kernel that holds MFMAs, walk the disassembly twice around the back edge, keep the in-order queue of vector-memory
instructions that the explicit `s_waitcnt vmcnt(N)` have not retired yet, and report every instruction that names a destination
register of a load still in that queue.
    python tools/audit_inflight.py [object file] [kernel name substring]      default: csrc/build/conv_wino_x3.o x3w
Round 4: the shipped conv_wino_x3w_kernel and two builds of it that fail beside the SPyNet stream all audit
clean -- whatever that hazard is, it is not a register of an in-flight load being touched."""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e2fgvi_amd import build as B

obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(B.CSRC, "build", "conv_wino_x3.o")
want = sys.argv[2] if len(sys.argv) > 2 else "x3w"
VM = re.compile(r"^(buffer|global|flat|scratch)_(load|store|atomic)")


def regs(o):
    r = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", o):
        r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", o):
        r.add(int(m.group(1)))
    return r


total = 0
for name, ins in B._kernels(B.device_isa(obj)).items():
    if want not in name:
        continue
    at = {a: i for i, (a, m, o) in enumerate(ins)}
    loops = []
    for i, (a, m, o) in enumerate(ins[:-1]):
        if m.startswith("s_cbranch") or m == "s_branch":
            try:
                off = int(o.split()[0])
            except ValueError:
                continue
            off -= 65536 if off > 32767 else 0
            tgt = ins[i + 1][0] + 4 * off
            if tgt <= a and tgt in at:
                loops.append((at[tgt], i))
    for s, e in loops:
        body = ins[s:e + 1]
        nm = sum(1 for _, m, _ in body if "mfma" in m)
        if nm < 10 or len(body) > 2000:          # K loops only (the outer persistent / epilogue loops carry no counted waits)
            continue
        pending, bad = [], 0
        for rnd in range(2):
            for a, m, o in body:
                if VM.match(m):
                    lds = o.split()[-1] == "lds"
                    pending.append((regs(o.split(",")[0]) if ("load" in m and not lds) else None, a))
                elif m == "s_waitcnt" and "vmcnt" in o:
                    n = int(re.search(r"vmcnt\((\d+)\)", o).group(1))
                    pending = pending[len(pending) - n:] if n else []
                else:
                    r = regs(o)
                    for d, pa in pending:
                        if d and d & r:
                            if rnd:
                                print("%s: %06x %s %s touches v%s of the load at %06x" % (name[:60], a, m, o, sorted(d & r), pa))
                                bad += 1
                            break
        print("%s loop %06x-%06x: %d instructions, %d MFMAs, %d findings" % (name[:60], body[0][0], body[-1][0], len(body), nm, bad))
        total += bad
sys.exit(1 if total else 0)
