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
    return funcs


def test_asm_loaded_registers_are_untouched_until_their_pin(wide_isa):
    # template arguments <T, BN, TAIL, PERS>: ...ELb<TAIL>ELb<PERS>E...; the default plan launches PERS = 0 only.  (In the persistent instantiation the two
    # staging roles live in different wave-uniform branches, whose register use a linear scan cannot tell apart: its guard is the bit-identity test on the GPU.)
    default = {k: v for k, v in wide_isa.items() if re.search(r"ELb[01]ELb0E", k)}
    assert len(default) == 8 and len(wide_isa) == 10, sorted(wide_isa)   # {bf16, fp16} x {96, 64} x {3x3 only, with a 1x1 tail}; + the persistent 64-cout tile
    for fn, body in default.items():
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


def test_no_valu_written_sgpr_right_in_front_of_an_asm_vmem_instruction(wide_isa):
    """Every instantiation, the persistent one included (its DMA / load forms carry the wait states themselves: TD_GLDS16G).  The same scan over the whole
    library: python tools/isa_hazard_scan.py (~5 min of compilation)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_hazard_scan", os.path.join(ROOT, "tools", "isa_hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    for fn, body in wide_isa.items():
        bad = mod.scan(body)
        assert not bad, (fn, bad[:4])
