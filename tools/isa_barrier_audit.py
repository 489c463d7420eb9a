"""Static audit of the gfx950 code objects: is there an s_barrier that a wave can reach with one of its own LDS WRITES still in
flight (no s_waitcnt lgkmcnt covering it on some path)?  The other waves' reads behind the barrier may then see the old bytes.

Why this exists: hipcc of ROCm 7.2 drops the `s_waitcnt lgkmcnt(0)` that __syncthreads()'s
workgroup release fence asks for when the barrier is a loop header and the pending write sits on the back edge (the entry edge has
nothing pending).  On gfx90a+ the waitcnt pass does not force a wait at s_barrier (back-off barrier), so nothing else catches it.
That was the "unexplained nondeterminism" of the register-resident Sinkhorn: `misc[0] = b_dust` written at the end of iteration k,
read by every wave after the barrier at the top of iteration k + 1 -- stale (iteration k - 1's value) when the write was still
queued behind other workgroups' LDS traffic.

    python tools/isa_barrier_audit.py file.s [...]         (hipcc -S --cuda-device-only output)
    python tools/isa_barrier_audit.py --build              (every .hip of ur-mvo_amd/csrc, product flags)

Dataflow per function over the basic-block graph.  State = is an LDS write pending, how many DS operations were issued after the
youngest pending write (DS operations complete in order, so `lgkmcnt(N)` retires the write once N <= that count and no scalar
memory operation -- which returns out of order -- sits among them).  LDS-DMA (`... lds`) is tracked by vmcnt and deliberately
carried across barriers by the pipelined GEMMs: not flagged."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAP = 32
DS_WRITE = re.compile(r"^ds_(write|add|sub|min|max|and|or|xor|inc|dec|mskor|cmpst|wrxchg|swizzle_never|append|consume|bpermute_never)")
BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+|s_setpc_b64|s_endpgm)\b")


def parse_functions(text):
    funcs, cur, name = {}, None, None
    for ln, line in enumerate(text.split("\n"), 1):
        s = line.split(";")[0].rstrip()
        m = re.match(r"^([A-Za-z_][\w.$]*):\s*$", s)
        if m and not m.group(1).startswith(".L"):
            name = m.group(1)
            cur = funcs.setdefault(name, [])
            continue
        if cur is None:
            continue
        if re.match(r"^\.Lfunc_end", s):
            cur, name = None, None
            continue
        t = s.strip()
        if not t or t.startswith(".") and not t.endswith(":"):
            continue
        cur.append((ln, t))
    return {k: v for k, v in funcs.items() if any("s_barrier" in t for _, t in v)}


def blocks_of(ins):
    """[(label or None, [(ln, text)...])], label -> block index"""
    blocks, cur, label = [], [], None
    for ln, t in ins:
        m = re.match(r"^(\.L[\w.$]+):$", t)
        if m:
            if cur or label is not None:
                blocks.append((label, cur))
            cur, label = [], m.group(1)
            continue
        cur.append((ln, t))
        if BRANCH.match(t):
            blocks.append((label, cur))
            cur, label = [], None
    if cur or label is not None:
        blocks.append((label, cur))
    return blocks


def transfer(state, t, hits, ln):
    pend, young, smem = state
    op = t.split()[0]
    if op == "s_waitcnt":
        m = re.search(r"lgkmcnt\((\d+)\)", t)
        n = None
        if m:
            n = int(m.group(1))
        elif re.match(r"^s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", t):
            v = int(t.split()[1], 0)
            n = (v >> 8) & 0xF
        if n is not None and pend and (n == 0 or (not smem and young >= n)):
            return (False, 0, False)
        return state
    if op == "s_barrier":
        if pend:
            hits.add(ln)
        return state
    if op.startswith("ds_"):
        if DS_WRITE.match(op):
            return (True, 0, False)
        return (pend, min(young + 1, CAP), smem) if pend else state
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op in ("s_memtime", "s_memrealtime", "s_dcache_inv", "s_dcache_wb") or op.startswith("s_atc") or op.startswith("s_scratch") or op.startswith("s_store"):
        return (pend, young, True) if pend else state
    if op.startswith("flat_"):
        return (pend, young, True) if pend else state
    return state


def merge(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if not a[0]:
        return b
    if not b[0]:
        return a
    return (True, min(a[1], b[1]), a[2] or b[2])


def audit(ins):
    blocks = blocks_of(ins)
    index = {lab: i for i, (lab, _) in enumerate(blocks) if lab}
    succ = []
    for i, (_, body) in enumerate(blocks):
        s = []
        last = body[-1][1] if body else ""
        op = last.split()[0] if last else ""
        if op == "s_endpgm" or op == "s_setpc_b64":
            pass
        elif op == "s_branch":
            tgt = last.split()[1]
            if tgt in index:
                s.append(index[tgt])
        else:
            if op.startswith("s_cbranch"):
                tgt = last.split()[-1]
                if tgt in index:
                    s.append(index[tgt])
            if i + 1 < len(blocks):
                s.append(i + 1)
        succ.append(s)
    inn = [None] * len(blocks)
    inn[0] = (False, 0, False)
    hits = set()
    work = [0]
    while work:
        i = work.pop()
        st = inn[i]
        for ln, t in blocks[i][1]:
            st = transfer(st, t, hits, ln)
        for j in succ[i]:
            m = merge(inn[j], st)
            if m != inn[j]:
                inn[j] = m
                work.append(j)
    return sorted(hits)


def audit_text(text):
    out = []
    for name, ins in parse_functions(text).items():
        for ln in audit(ins):
            out.append((name, ln))
    return out


def compile_to_asm(src, extra=()):
    tmp = tempfile.NamedTemporaryFile(suffix=".s", delete=False)
    tmp.close()
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-Wno-pass-failed",
           "-S", "--cuda-device-only", "-o", tmp.name, src, *extra]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(tmp.name).read()
    os.unlink(tmp.name)
    return text


def main(argv):
    files = []
    extra = []
    if "--exp" in argv:
        extra = ["-DURF_EXPERIMENTS"]
    if "--build" in argv:
        d = os.path.join(ROOT, "ur-mvo_amd", "csrc")
        files = [(f, None) for f in sorted(os.listdir(d)) if f.endswith(".hip")]
        texts = [(f, compile_to_asm(os.path.join(d, f), extra)) for f, _ in files]
    else:
        texts = [(f, open(f).read()) for f in argv if not f.startswith("--")]
    bad = 0
    for f, text in texts:
        lines = text.split("\n")
        res = audit_text(text)
        nb = sum(1 for l in lines if re.match(r"^\s*s_barrier\b", l))
        print(f"{f}: {nb} s_barrier, {len(res)} reachable with an LDS write in flight")
        for name, ln in res:
            bad += 1
            print(f"   {name}: line {ln}: {lines[ln - 1].strip()}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
