"""Compiler-contract checks on the ISA of the wide conv tile (csrc/conv_glds_wide.hip), on the CPU: hipcc cross-compiles gfx950 without a GPU.

The kernel issues its halo-patch loads by INLINE ASM (so that hipcc's wait-count pass does not guard them with waits that would drain the LDS-DMA weight
stream) and tells the compiler from where it may use the loaded registers with an empty asm "pin" behind a counted s_waitcnt.  Two things the compiler is free
to do would silently break that -- both happened during round 6 (profiles/r06_wide_tile_persistent_loop.txt):
  * touch (copy, spill, overwrite) an asm-loaded register between the load and its pin: the data has not landed yet;
  * put a v_readfirstlane right in front of an inline-asm VMEM instruction that reads the SGPR it wrote: the hazard recogniser does not look inside inline
    asm, and the hardware needs five wait states there.
The test compiles a copy of the source whose pin carries a marker comment (the only change: `; tdw-pin %0` instead of an empty asm string -- same
constraints, same code) and scans every default (non-persistent) instantiation."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "terrain_diffusion_amd", "csrc")
PIN = 'asm volatile("" : "+v"(av[it_]));'


def _regs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


def _sregs(text):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bs(\d+)\b", text))
    return out


@pytest.fixture(scope="module")
def wide_isa(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("isa")
    src = open(os.path.join(CSRC, "conv_glds_wide.hip")).read()
    assert src.count(PIN) == 1, "the pin of the asm-loaded patch registers moved: update this test"
    (d / "conv_glds_wide.hip").write_text(src.replace(PIN, 'asm volatile("; tdw-pin %0" : "+v"(av[it_]));'))
    (d / "u.hip").write_text('#include <hip/hip_runtime.h>\n#include <algorithm>\n#include "%s"\n' % (d / "conv_glds_wide.hip"))
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "--cuda-device-only", "-S", str(d / "u.hip"), "-o", str(d / "u.s")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    funcs, name, body = {}, None, []
    for line in (d / "u.s").read_text().split("\n"):
        m = re.match(r"^(_ZN2td21conv_glds_kernel_wide\w+):", line)
        if m:
            name, body = m.group(1), []
            funcs[name] = body
        elif line.startswith(".Lfunc_end"):
            name = None
        elif name:
            body.append(line)
    # template arguments <T, BN, TAIL, PERS>: ...ELb<TAIL>ELb<PERS>E...; the default plan launches PERS = 0 only
    return {k: v for k, v in funcs.items() if re.search(r"ELb[01]ELb0E", k)}


def test_asm_loaded_registers_are_untouched_until_their_pin(wide_isa):
    assert len(wide_isa) == 8, sorted(wide_isa)   # {bf16, fp16} x {96, 64} x {3x3 only, with a 1x1 tail}
    for fn, body in wide_isa.items():
        in_asm, flight, loads, pins, bad = False, {}, 0, 0, []
        for i, line in enumerate(body):
            t = line.strip()
            if "#ASMSTART" in t:
                in_asm = True; continue
            if "#ASMEND" in t:
                in_asm = False; continue
            if in_asm:
                m = re.match(r"global_load_dword(x4)? (v\[\d+:\d+\]|v\d+),", t)
                if m:   # (the LDS-DMA form, global_load_lds_*, has no destination register)
                    for r in _regs(m.group(2)):
                        flight[r] = i
                    loads += 1
                m = re.match(r"; tdw-pin (v\[\d+:\d+\])", t)
                if m:
                    for r in _regs(m.group(1)):
                        flight.pop(r, None)
                    pins += 1
                continue
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            code = t.split(";")[0]
            hit = _regs(code) & set(flight)
            if hit:
                bad.append((i, t, sorted(hit)[:4]))
        assert loads >= 18 and pins >= 18, (fn, loads, pins)   # six pieces: the first patch, and the two requests of a 64-channel unit
        assert not bad, (fn, bad[:5])
        assert not flight, (fn, "asm loads without a pin behind them", sorted(flight)[:8])


def test_no_readfirstlane_right_in_front_of_an_asm_vmem_instruction(wide_isa):
    for fn, body in wide_isa.items():
        code = []   # (text, inside inline asm)
        in_asm = False
        for line in body:
            t = line.strip()
            if "#ASMSTART" in t:
                in_asm = True; continue
            if "#ASMEND" in t:
                in_asm = False; continue
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            code.append((t.split(";")[0].strip(), in_asm))
        for i, (t, a) in enumerate(code):
            if not a or not t.startswith("global_load"):
                continue
            need = _sregs(t)
            if not need:
                continue
            states = 0   # wait states between a VALU write of one of those SGPRs and this instruction: five are required
            for j in range(i - 1, max(-1, i - 8), -1):
                u = code[j][0]
                if u.startswith("v_readfirstlane_b32") or u.startswith("v_readlane_b32"):
                    dst = _sregs(u.split(",")[0])
                    assert not (dst & need) or states >= 5, (fn, u, t, states)
                m = re.match(r"s_nop (\d+)", u)
                states += (int(m.group(1)) + 1) if m else 1
                if states >= 5:
                    break
