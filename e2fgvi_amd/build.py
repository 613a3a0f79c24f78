"""Build libe2fgvi_hip.so (gfx950) in-tree with hipcc.  ``python -m e2fgvi_amd.build``."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libe2fgvi_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]
# No packed-fp32 VALU (v_pk_{mul,add,fma}_f32) in kernels that may run beside the bf16 MFMA tiles on another stream.
# Measured on MI355X (tools/probe/overlap_probe.hip, profiles/r02_overlap_probe_*.txt, DESIGN.md "Stream overlap"): a
# packed-fp32 instruction that consumes registers freshly written by vector-memory loads returns wrong values in lanes
# 48-63 when its wave shares a SIMD with a wave that streams 2x2 v_mfma_f32_32x32x16_bf16 tiles fed by ds_read_b128 --
# regardless of s_waitcnt / idle cycles / cache policy, never with scalar fp32 instructions on the same registers.
NOPK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# (source, object, extra flags).  The HBM-bound kernels lose nothing without packed math; the fp32 conv kernels keep it in
# their normal build (6-8 % on the MFMA kernels' transforms / epilogues, profiles/r02_c2_*) and get a second, packed-free
# build for the side-stream launches (SPyNet).  mdcn.hip: its sampler waves do arithmetic on freshly loaded offset / mask / flow
# words while the other waves of the SAME workgroup stream LDS-fed bf16 MFMA tiles -- with packed math the 64-row / two-K-group
# tile returned wrong rows 24-31 (lanes 48-63 of the sampler wave) in a few launches per hundred.
# attention_bf16.hip / conv_tail.hip (round 3, advisor): their fp32 VALU works on MFMA results and LDS data today, but both
# units hold bf16 MFMA streams, so they are kept free of packed fp32 as well (measured neutral) -- a later epilogue edit that
# touches VMEM-fresh registers then cannot re-enter the hazard silently; 200-launch bit-identity reruns in the GPU tests.
# conv_wino.o / conv_wino4.o (round 6, advisor): the fp32 Winograd kernels build their A operands with VALU adds on patch words and
# run wherever the kernel table (or a table miss, or E2FGVI_TILE_TABLE=0) names them -- encoder.layers.14 in front of the join
# beside SPyNet's split-operand GEMM, the propagation split's side-stream layers beside the main stream's split-operand deformable
# conv: they held 48-992 v_pk_{mul,add,fma}_f32 per kernel.  Both objects are built packed-free now and checked like the others.
UNITS = [("error.hip", "error.o", []), ("conv.hip", "conv.o", []), ("conv.hip", "conv_nopk.o", NOPK + ["-DE2_NOPK_VARIANT"]),
         ("conv_bf16x.hip", "conv_bf16x.o", NOPK), ("conv_wino.hip", "conv_wino.o", NOPK), ("conv_wino.hip", "conv_wino_x3.o", NOPK + ["-DE2_WINO_X3=1"]), ("conv_wino4.hip", "conv_wino4.o", NOPK), ("conv_tail.hip", "conv_tail.o", NOPK),
         ("mdcn.hip", "mdcn.o", NOPK), ("attention.hip", "attention.o", []),
         ("attention_bf16.hip", "attention_bf16.o", NOPK), ("attention_x3.hip", "attention_x3.o", NOPK), ("misc.hip", "misc.o", NOPK),
         ("video.hip", "video.o", NOPK), ("metrics.hip", "metrics.o", NOPK)]
NOPK_OBJECTS = ("conv_nopk.o", "conv_bf16x.o", "conv_wino.o", "conv_wino4.o", "conv_wino_x3.o", "conv_tail.o", "mdcn.o", "attention_bf16.o", "attention_x3.o", "misc.o", "video.o", "metrics.o")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return "hipcc"


def _stale(out, deps):
    return (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=False, nopk_all=False, lib=None, tag=None, defines=()):
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path.
    nopk_all: build every unit without packed-fp32 VALU (A/B measurements) into ``lib``.
    tag / defines: an A/B build with extra -D flags into csrc/build_<tag>/ and csrc/libe2fgvi_hip_<tag>.so (run it with
    E2FGVI_LIB=<that path>); the product library is the build without either."""
    hipcc = _hipcc()
    if tag:
        nopk_all = False
    objdir = os.path.join(CSRC, ("build_" + tag) if tag else ("build_nopk" if nopk_all else "build"))
    os.makedirs(objdir, exist_ok=True)
    out = lib or (os.path.join(CSRC, "libe2fgvi_hip_%s.so" % tag) if tag else
                  os.path.join(CSRC, "libe2fgvi_hip_nopk.so") if nopk_all else LIB)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "e2fgvi_hip.h"), os.path.abspath(__file__)]
    jobs = []
    for src, obj, extra in UNITS:
        sp, op = os.path.join(CSRC, src), os.path.join(objdir, obj)
        if nopk_all and not any(f == "-packed-fp32-ops" for f in extra):
            extra = extra + NOPK
        if force or _stale(op, [sp] + headers):
            jobs.append([hipcc] + FLAGS + extra + list(defines) + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, obj) for _, obj, _ in UNITS]
    if force or jobs or _stale(out, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    if jobs:
        verify_isa(objdir)
        if not nopk_all:
            verify_wino_waits(objdir)
            verify_exit_reuse(objdir)
            verify_no_scratch(objdir)
    return out


def build_variant(tag, *defines):
    """python -c 'from e2fgvi_amd import build; build.build_variant("trunc", "-DE2_SPLIT_RNE=0")'"""
    return build(tag=tag, defines=defines)


def device_isa(obj):
    """Disassembly of the gfx950 code object embedded in a host object file."""
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="e2isa")
    try:
        local = os.path.join(d, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", os.path.basename(obj)], cwd=d, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(d) if "gfx950" in f]
        if not dev:
            return ""
        return subprocess.run([OBJDUMP, "-d", os.path.join(d, dev[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)


def verify_isa(objdir=None):
    """Build-time check of the generated code: the units that must be free of packed-fp32 VALU instructions are."""
    import re
    objdir = objdir or os.path.join(CSRC, "build")
    if not os.path.exists(OBJDUMP):
        return
    for obj in NOPK_OBJECTS:
        isa = device_isa(os.path.join(objdir, obj))
        bad = re.findall(r"v_pk_(?:mul|add|fma)_f32", isa)
        if bad or not isa:
            raise RuntimeError("%s: %s" % (obj, "%d packed-fp32 instructions in a unit that must have none" % len(bad) if isa
                                           else "no gfx950 code object found"))


def verify_no_scratch(objdir=None, obj="conv_bf16x.o", marker="ELb0ELi2ELb1EEE"):
    """The ping-pong instantiations of conv_bf16x_kernel (template tail `false, 2, true`) keep 128 accumulator registers in place
    across four barriers per K-step; two ways of writing that loop made hipcc spill 414 registers or copy the kernel arguments to
    scratch without a warning (csrc/conv_bf16x.hip, PP).  Fails the build when one of them has a private segment or a spill."""
    import re
    import shutil
    import tempfile
    objdir = objdir or os.path.join(CSRC, "build")
    readelf = os.path.join(os.path.dirname(OBJDUMP), "llvm-readelf")
    if not (os.path.exists(OBJDUMP) and os.path.exists(readelf)):
        return 0
    d = tempfile.mkdtemp(prefix="e2elf")
    try:
        shutil.copy(os.path.join(objdir, obj), os.path.join(d, obj))
        subprocess.run([OBJDUMP, "--offloading", obj], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(d) if "gfx950" in f]
        notes = subprocess.run([readelf, "--notes", os.path.join(d, dev[0])], check=True, capture_output=True, text=True).stdout if dev else ""
    finally:
        shutil.rmtree(d, ignore_errors=True)
    found = 0
    for block in notes.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name or marker not in name.group(1):
            continue
        found += 1
        priv = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1))
        if priv or spill:
            raise RuntimeError("%s: %s: private segment %d bytes, %d spilled VGPRs in a ping-pong kernel" % (obj, name.group(1)[:70], priv, spill))
    if not found:
        raise RuntimeError("verify_no_scratch: no ping-pong instantiation found in %s (mangled-name marker changed?)" % obj)
    return found


def _kernels(isa):
    """{kernel name: [(address, mnemonic, operand text)]} of a disassembly"""
    import re
    out, cur = {}, None
    for line in isa.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def check_kernel_waits(obj, name, ins):
    """One kernel's instruction list [(address, mnemonic, operands)]: every marked explicit wait inside a loop must carry
    the number of vector-memory instructions issued since the previous marked wait (cyclically over the back edge).
    Returns the number of loops checked; raises RuntimeError on a mismatch."""
    import re
    vmem = re.compile(r"^(buffer|global|flat|scratch)_(load|store|atomic)")
    marks = [i for i in range(len(ins) - 1) if ins[i][1] == "s_waitcnt" and ins[i + 1][1] == "s_waitcnt"
             and ins[i][2] == ins[i + 1][2] and "vmcnt" in ins[i][2]]
    if not marks:
        return 0
    addr_to_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    spans = []
    for i, (a, mn, ops_) in enumerate(ins):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            # objdump prints the target as the unsigned 16-bit word offset from the next instruction
            try:
                off = int(ops_.split()[0])
            except (ValueError, IndexError):
                continue
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if off < 0 and tgt in addr_to_idx:
                spans.append((addr_to_idx[tgt], i))
    loops = 0
    # a kernel that opts in to another wait discipline, by (mangled) name: conv_wino_x3w_kernel reloads its single-buffered weight
    # planes in place and claims them group by group (below).  (Round 3's two-trips-ahead LDS-DMA variant of conv_wino_kernel and
    # its rule left with the kernel in round 6.)
    rolling = "conv_wino_x3w_kernel" in name
    for lo, hi in spans:
        inner = [m for m in marks if lo <= m <= hi]
        if not inner or any(lo <= l2 and h2 <= hi and (l2, h2) != (lo, hi) and any(l2 <= m <= h2 for m in inner)
                            for l2, h2 in spans):
            continue                                   # no marked wait, or an inner loop owns them
        loops += 1
        if rolling:
            # The plane loads of the next stage are the LAST vector-memory instructions of a trip, G per position; the next
            # trip claims them position by position with vmcnt(n_0 = (P - 1) G), ..., vmcnt(0), nothing issued in between.
            counts = [int(re.search(r"vmcnt\((\d+)\)", ins[m][2]).group(1)) for m in inner]
            for a, b in zip(inner, inner[1:]):
                if any(vmem.match(ins[k][1]) for k in range(a + 2, b)):
                    raise RuntimeError("%s: %s: vector-memory instruction between the group waits at 0x%x and 0x%x"
                                       % (obj, name[:60], ins[a][0], ins[b][0]))
            g = counts[0] - counts[1] if len(counts) > 1 else counts[0]
            if counts[-1] != 0 or g <= 0 or any(c != counts[0] - j * g for j, c in enumerate(counts)):
                raise RuntimeError("%s: %s: group waits %s are not (P-1) G, ..., G, 0" % (obj, name[:60], counts))
            tail = [k for k in list(range(inner[-1] + 2, hi + 1)) + list(range(lo, inner[0])) if vmem.match(ins[k][1])]
            planes = tail[-(counts[0] + g):]
            if len(planes) != counts[0] + g or any(ins[k][1] != "buffer_load_dwordx4" or " lds" in ins[k][2] for k in planes):
                raise RuntimeError("%s: %s: the %d vector-memory instructions in front of the group waits are not the plane "
                                   "loads" % (obj, name[:60], counts[0] + g))
            continue
        for j, m in enumerate(inner):
            n = int(re.search(r"vmcnt\((\d+)\)", ins[m][2]).group(1))
            if j:
                rng = range(inner[j - 1] + 2, m)
            else:                                       # over the back edge: tail of the loop + its head
                rng = list(range(inner[-1] + 2, hi + 1)) + list(range(lo, m))
            cnt = sum(1 for k in rng if vmem.match(ins[k][1]))
            if cnt != n:
                raise RuntimeError("%s: %s: explicit s_waitcnt vmcnt(%d) at 0x%x follows %d vector-memory instructions "
                                   "since the previous explicit wait" % (obj, name[:60], n, ins[m][0], cnt))
    return loops


def verify_wino_waits(objdir=None, objects=("conv_wino.o", "conv_wino_x3.o", "conv_wino4.o")):
    """The Winograd kernels issue their weight loads through pinned inline asm and wait for them with an explicit
    ``s_waitcnt vmcnt(N)`` whose N is the number of vector-memory instructions issued since (conv_wino.hip): a count the
    compiler does not maintain.  This check re-derives it from the generated code on every build.  The explicit waits are
    marked by being issued twice in a row; in every K loop (a backward branch around marked waits) the wait that follows
    another must carry exactly the number of VMEM instructions between the two -- cyclically over the loop's back edge --
    because the loads a wait claims were issued immediately before the previous marked wait.  Raises on a mismatch
    (a compiler that adds, removes or moves a load in the loop), returns the number of loops checked."""
    objdir = objdir or os.path.join(CSRC, "build")
    if not os.path.exists(OBJDUMP):
        return 0
    loops = 0
    for obj in objects:
        for name, ins in _kernels(device_isa(os.path.join(objdir, obj))).items():
            loops += check_kernel_waits(obj, name, ins)
    if not loops:
        raise RuntimeError("verify_wino_waits: no K loop with marked waits found (disassembly format changed?)")
    return loops


def _vregs(text):
    """vector registers named in an operand text"""
    import re
    r = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        r.add(int(m.group(1)))
    return r


def _successors(ins, at):
    """control-flow successors (instruction indices) of every instruction of a kernel; objdump prints a branch target as the
    unsigned 16-bit word offset from the next instruction.  Raises on an indirect jump (nothing here has one)."""
    succ = []
    for i, (a, m, o) in enumerate(ins):
        if m in ("s_endpgm", "s_endpgm_saved"):
            succ.append(())
        elif m == "s_branch" or m.startswith("s_cbranch"):
            off = int(o.split()[0])
            off -= 65536 if off > 32767 else 0
            tgt = at.get(a + 4 + 4 * off)
            if tgt is None:
                raise RuntimeError("branch at 0x%x leaves the kernel" % a)
            succ.append((tgt,) if m == "s_branch" else (tgt, i + 1))
        elif m.startswith("s_setpc") or m.startswith("s_swappc"):
            raise RuntimeError("indirect jump at 0x%x: control flow cannot be followed" % a)
        else:
            succ.append((i + 1,) if i + 1 < len(ins) else ())
    return succ


def check_exit_reuse(obj, name, ins, min_mfma=8):
    """For hipcc the destination of an inline-asm load is written when the statement ends.  The weight loads a K loop issues for
    the stage PAST the end are still in flight when the loop exits; if the compiler reuses their registers (it did: for the
    epilogue's address arithmetic, hoisted above the kernel's own `s_waitcnt vmcnt(0)`) the late data lands on top of the new values
    whenever memory is slow -- DESIGN.md C4.  The sources name those registers behind a post-loop wait; this check re-derives from
    the disassembly that it worked:

      * weight registers = destinations of non-LDS buffer loads that feed an MFMA's B operand anywhere in the kernel;
      * K loops = the strongly connected components of the control-flow graph that hold >= min_mfma MFMAs;
      * from every control-flow edge that LEAVES such a span the walk follows the code -- unconditional branches, both arms of
        conditional ones -- until the first wait that contains vmcnt(0) on each path; any vector instruction on the way that
        reads or writes a weight register is an error, and so is reaching s_endpgm (or another K loop) without that wait;
      * the wave roles of these kernels (`switch (wave)`, wave = threadIdx.x >> 6) are compiled as exec-masked if / else regions
        although every lane of a wave takes the same arm.  The walk carries that fact: passing the flip to an `else` arm
        (s_andn2_saveexec_b64, or the s_xor_b64 exec, exec, ... that follows an s_or_saveexec_b64) with live lanes leaves none (the wave took the `then` arm with
        all its lanes), until the region's end restores the mask (s_or_b64 exec, exec, ...); with no live lane vector instructions
        do nothing, s_cbranch_execz is taken and s_cbranch_execnz is not.  Without this the walk would run from one role's loop
        exit straight into the next role's prologue (the round-4 version of this check stopped there, at that prologue's wait,
        and could pass without having seen the shared epilogue).

    Returns the number of exit edges walked; raises RuntimeError on a violation."""
    import re
    at = {a: i for i, (a, _, _) in enumerate(ins)}
    succ = _successors(ins, at)
    is_mfma = [("mfma" in m) for _, m, _ in ins]
    breg = set()
    for (_, m, o), f in zip(ins, is_mfma):
        if f:
            breg |= _vregs([x.strip() for x in o.split(",")][2])
    wregs = set()
    for _, m, o in ins:
        if re.match(r"buffer_load_dword", m) and not o.rstrip().endswith(" lds"):
            d = _vregs(o.split(",")[0])
            if d & breg:
                wregs |= d
    if not wregs:
        return 0
    # K loops: the strongly connected components of the control-flow graph that hold >= min_mfma MFMAs (Tarjan, iterative).  Address
    # ranges of backward branches would not do: block placement puts parts of the epilogue in front of the loops.
    n = len(ins)
    index, low, comp = [-1] * n, [0] * n, [-1] * n
    stack, onstack, counter, ncomp = [], [False] * n, 0, 0
    for root in range(n):
        if index[root] >= 0:
            continue
        work = [(root, 0)]
        while work:
            v, pi = work.pop()
            if pi == 0:
                index[v] = low[v] = counter
                counter += 1
                stack.append(v)
                onstack[v] = True
            recurse = False
            for j in range(pi, len(succ[v])):
                w = succ[v][j]
                if index[w] < 0:
                    work.append((v, j + 1))
                    work.append((w, 0))
                    recurse = True
                    break
                if onstack[w]:
                    low[v] = min(low[v], index[w])
            if recurse:
                continue
            if low[v] == index[v]:
                while True:
                    w = stack.pop()
                    onstack[w] = False
                    comp[w] = ncomp
                    if w == v:
                        break
                ncomp += 1
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
    members = {}
    for k in range(n):
        members.setdefault(comp[k], []).append(k)
    loops = [m for m in members.values() if len(m) > 1 and sum(1 for k in m if is_mfma[k]) >= min_mfma]
    inside = [False] * n
    for m in loops:
        for k in m:
            inside[k] = True
    edges = 0
    for m in loops:
        mine = set(m)
        starts = set()
        for k in m:
            for t in succ[k]:
                if t not in mine:
                    starts.add(t)
        for st in sorted(starts):
            edges += 1
            seen, todo = set(), [(st, False)]
            while todo:
                k, dead = todo.pop()                 # dead: no live lane (the wave is passing over another role's region)
                if (k, dead) in seen:
                    continue
                seen.add((k, dead))
                a, m, o = ins[k]
                if m == "s_waitcnt" and "vmcnt(0)" in o:
                    continue
                if m.startswith("s_endpgm") or (inside[k] and not dead):
                    raise RuntimeError("%s: %s: the path from the K-loop exit at 0x%x reaches 0x%x (%s) without an s_waitcnt vmcnt(0): "
                                       "weight loads may still be in flight" % (obj, name[:60], ins[st][0], a,
                                                                                "another K loop" if inside[k] else m))
                if not dead and not m.startswith("s_") and _vregs(o) & wregs:
                    raise RuntimeError("%s: %s: 0x%x %s %s touches a register of a weight load that may still be in flight behind the "
                                       "K loop left at 0x%x (no vmcnt(0) yet)" % (obj, name[:60], a, m, o, ins[st][0]))
                flip = m == "s_andn2_saveexec_b64" or (m == "s_xor_b64" and o.startswith("exec"))
                if flip and not dead:
                    dead = True
                elif m in ("s_or_b64", "s_mov_b64") and o.startswith("exec"):
                    dead = False
                nxt = succ[k]
                if dead and m == "s_cbranch_execz":
                    nxt = nxt[:1]
                elif dead and m == "s_cbranch_execnz":
                    nxt = nxt[1:]
                todo.extend((t, dead) for t in nxt)
    return edges


EXIT_REUSE_OBJECTS = ("conv_wino.o", "conv_wino_x3.o", "conv_wino4.o")


def verify_exit_reuse(objdir=None, objects=EXIT_REUSE_OBJECTS):
    """check_exit_reuse over every Winograd kernel of the library (all of them prefetch their weights past the end of the K loop
    through inline asm); returns the number of K-loop exit edges walked"""
    objdir = objdir or os.path.join(CSRC, "build")
    if not os.path.exists(OBJDUMP):
        return 0
    n = 0
    for obj in objects:
        for name, ins in _kernels(device_isa(os.path.join(objdir, obj))).items():
            if "conv_wino" not in name or "pack_" in name:
                continue
            k = check_exit_reuse(obj, name, ins)
            if not k:
                raise RuntimeError("verify_exit_reuse: no K-loop exit found in %s: %s (disassembly format changed?)" % (obj, name[:60]))
            n += k
    if not n:
        raise RuntimeError("verify_exit_reuse: no Winograd kernel found")
    return n


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, nopk_all="--nopk-all" in sys.argv))
