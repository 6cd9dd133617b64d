"""Instruction-class stream of one kernel from a `hipcc -save-temps` .s file: one letter per instruction (M MFMA, e v_exp, v other VALU, r / w LDS read / write,
d other LDS, G global load, S global store, W s_waitcnt, B s_barrier, n s_nop, X scratch, J branch, s other scalar), one line per basic block, plus the kernel's
VGPR / LDS / scratch sizes.  Used to check that the pipelined attention loop keeps its MFMAs in the shadow of the softmax's vector work (round 6).
  python tools/isa_stream.py file.s <substring of the mangled kernel name> [--resources-only]"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    txt = open(path).read()
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S):
        if key in m.group(1):
            g = lambda k: int(re.search(k + r"\s+(\d+)", m.group(2)).group(1))
            res[m.group(1)] = (g(r"\.amdhsa_next_free_vgpr"), g(r"\.amdhsa_group_segment_fixed_size"), g(r"\.amdhsa_private_segment_fixed_size"))
    for k, (v, l, p) in res.items():
        print(f"{k[:90]}: vgpr {v} lds {l} scratch {p}")
    if "--resources-only" in sys.argv:
        return
    name = next(iter(res))
    body = txt[txt.index("\n" + name + ":"):txt.index(".amdhsa_kernel " + name)]
    cls = [("v_mfma", "M"), ("v_exp", "e"), ("ds_read", "r"), ("ds_load", "r"), ("ds_write", "w"), ("ds_store", "w"), ("ds_", "d"), ("global_load", "G"), ("buffer_load", "G"),
           ("global_store", "S"), ("buffer_store", "S"), ("s_waitcnt", "W"), ("s_barrier", "B"), ("s_nop", "n"), ("scratch_", "X"), ("v_", "v"), ("s_cbranch", "J"), ("s_branch", "J"), ("s_", "s")]
    cur = ""
    for line in body.split("\n")[1:]:
        t = line.strip()
        if t.startswith(".LBB"):
            print(cur); cur = ""; print(t.split(";")[0].strip(), end=" ")
            continue
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        cur += next((c for p, c in cls if op.startswith(p)), "?")
    print(cur)


if __name__ == "__main__":
    main()
