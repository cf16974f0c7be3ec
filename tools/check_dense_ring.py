"""Static check of k_dense_fused's register ring (bz_kernels.h): its tile loads are inline asm, invisible to the compiler's
own wait bookkeeping, so nothing else may touch a load's destination registers between the load and the `s_waitcnt vmcnt(N)`
statement that retires it; the same holds for the early mailbox polls (scalar loads retired by an asm `s_waitcnt lgkmcnt(0)`).  What CAN be checked on the generated code without a control-flow analysis, and is what went wrong
once during development (launch bounds that capped the registers at 64 made the compiler spill ring registers right behind
their loads): no spill code at all in any instantiation, and every ring wait is vmcnt(4 KP) — two tiles left in flight.
That no live value is copied or re-used in between is covered by the parity tests (tests/test_gpu_dense.py: a register read
before its data arrived is a wrong gradient).

    python tools/check_dense_ring.py          (exit code 1 on a finding; tests/test_host_logic.py runs it)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
INST = [("float", 1), ("float", 2), ("float", 4), ("double", 1), ("double", 2)]


def check_function(name, lines, kp):
    """No spill code (a spill of a ring register between its load and its wait would store data that has not arrived), and
    every ring wait leaves exactly two tiles in flight."""
    findings = []
    text = [l.split(";")[0].strip() for _, l in lines]
    for (ln, _), s in zip(lines, text):
        if s.startswith("scratch_") or re.match(r"buffer_(store|load)_dword.*offen", s):
            findings.append(f"{name}: line {ln}: spill code `{s}`")
    # the early polls (scalar loads issued at the top of a step, retired by an asm wait after the tile's products): between
    # such a load and the next `s_waitcnt lgkmcnt(0)` in program order nothing may name its destination registers
    code = [(ln, s) for (ln, _), s in zip(lines, text) if s and not s.endswith(":") and not s.startswith((".", ";"))]
    nearly = 0
    for i, (ln, s) in enumerate(code):
        m = re.match(r"s_load_dwordx8 s\[(\d+):(\d+)\], .* glc$", s)
        if not m or (i + 1 < len(code) and code[i + 1][1].startswith("s_waitcnt lgkmcnt(0)")):
            continue
        nearly += 1
        lo, hi = int(m.group(1)), int(m.group(2))
        for ln2, s2 in code[i + 1:]:
            if s2.startswith("s_waitcnt lgkmcnt(0)"):
                break
            regs = set()
            for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", s2):
                regs.update(range(int(a), int(b) + 1))
            regs.update(int(a) for a in re.findall(r"\bs(\d+)\b", s2))
            if any(lo <= r <= hi for r in regs) and not re.match(r"s_load_dwordx8 s\[\d+:\d+\], .* glc$", s2):
                findings.append(f"{name}: line {ln2}: `{s2}` touches s[{lo}:{hi}] of the early poll issued at line {ln}")
                break
    if nearly == 0:
        findings.append(f"{name}: no early poll found (expected one or two per step)")
    waits = [int(m.group(1)) for s in text for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)$", s)] if m]
    ring_waits = [w for w in waits if w not in (0, 1)]
    if not ring_waits or any(w != 4 * kp for w in ring_waits):
        findings.append(f"{name}: ring waits {sorted(set(ring_waits))}, expected vmcnt({4 * kp}) = two tiles of 2 rows x {kp} packs")
    return findings


def main():
    src = "#include \"%s\"\n" % os.path.join(ROOT, "bazinga.jl_amd", "csrc", "bz_kernels.h")
    for t, kp in INST:
        src += f"template __global__ void bz::k_dense_fused<{t}, {kp}>(bz::DenseFusedArgs<{t}>, bz::ElemParams<{t}>);\n"
    with tempfile.TemporaryDirectory() as td:
        cu, asm = os.path.join(td, "ring.hip"), os.path.join(td, "ring.s")
        open(cu, "w").write(src)
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I/opt/rocm/include",
                               "--cuda-device-only", "-S", cu, "-o", asm], cwd=os.path.join(ROOT, "bazinga.jl_amd", "csrc"))
        text = open(asm).read().splitlines()
    funcs, cur, name = {}, None, None
    for i, l in enumerate(text):
        m = re.match(r"^(_ZN2bz13k_dense_fused\w+):", l)
        if m:
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is not None:
            if l.startswith(".Lfunc_end"):
                cur = None
                continue
            cur.append((i + 1, l))
    if len(funcs) != len(INST):
        print(f"expected {len(INST)} k_dense_fused functions, found {len(funcs)}")
        return 1
    bad = []
    for name, lines in funcs.items():
        nload = sum(1 for _, l in lines if "global_load_dwordx4" in l and " nt" in l)
        nwait = sum(1 for _, l in lines if re.search(r"s_waitcnt vmcnt\((4|8|16)\)", l))
        kp = int(re.search(r"fusedI[fd]Li(\d)E", name).group(1))
        f = check_function(name, lines, kp)
        print(f"{name}: {nload} ring loads, {nwait} ring waits, {len(f)} findings")
        bad += f
    for b in bad[:20]:
        print(b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
