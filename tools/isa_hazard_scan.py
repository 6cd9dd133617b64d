"""Scan gfx950 ISA for the one hazard hipcc does not guard: an SGPR written by the VALU (v_readfirstlane / v_readlane -- also the reload of a spilled SGPR)
and read by a VMEM instruction INSIDE INLINE ASM fewer than five wait states later (the hazard recogniser does not look inside inline asm).

    python tools/isa_hazard_scan.py                       # compiles terrain_diffusion_amd/csrc/engine.hip (every kernel of the library, ~5 min) and scans it
    python tools/isa_hazard_scan.py file.s [file2.s ...]   # scans ISA produced with  hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S

Exit status 1 if a violation is found.  tests/test_isa_contract.py runs the same scan on the wide tile's file in the CPU suite (20 s)."""
import os
import re
import subprocess
import sys
import tempfile


def sregs(text):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bs(\d+)\b", text))
    return out


def functions(path):
    funcs, name = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1); funcs[name] = []
        elif line.startswith(".Lfunc_end"):
            name = None
        elif name:
            funcs[name].append(line)
    return funcs


def scan(body):
    """[(writer, reader, wait states)] for one function body (list of ISA lines)"""
    code, in_asm = [], False
    for line in body:
        t = line.strip()
        if "#ASMSTART" in t:
            in_asm = True; continue
        if "#ASMEND" in t:
            in_asm = False; continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        code.append((t.split(";")[0].strip(), in_asm))
    bad = []
    for i, (t, a) in enumerate(code):
        if not a or not re.match(r"(global|buffer|flat|scratch)_", t):
            continue
        need = sregs(t)
        if not need:
            continue
        states = 0
        for j in range(i - 1, max(-1, i - 8), -1):
            u = code[j][0]
            if u.startswith("v_readfirstlane_b32") or u.startswith("v_readlane_b32"):
                if sregs(u.split(",")[0]) & need and states < 5:
                    bad.append((u, t, states))
            m = re.match(r"s_nop (\d+)", u)
            states += (int(m.group(1)) + 1) if m else 1
            if states >= 5:
                break
    return bad


def main(argv):
    paths = argv[1:]
    if not paths:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        csrc = os.path.join(root, "terrain_diffusion_amd", "csrc")
        out = os.path.join(tempfile.mkdtemp(), "engine.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", '-DTD_CSRC_SHA="scan"', "-I" + csrc, "--cuda-device-only", "-S", os.path.join(csrc, "engine.hip"), "-o", out], check=True)
        paths = [out]
    total = 0
    for p in paths:
        fs = functions(p)
        for fn, body in fs.items():
            bad = scan(body)
            total += len(bad)
            for w, r, n in bad[:4]:
                print(f"{fn[:100]}: `{w}` -> `{r}` after {n} wait states")
        print(f"{p}: {len(fs)} functions scanned")
    print("violations:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
