"""Host logic of the split-K planner (csrc/td_device.h: conv_set_kbounds), compiled with hipcc as a host program (no GPU needed)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kb(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    exe = str(tmp_path_factory.mktemp("kb") / "kbounds")
    subprocess.check_call(["hipcc", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "terrain_diffusion_amd", "csrc"),
                           os.path.join(ROOT, "tests", "kbounds_harness.cpp"), "-o", exe], stderr=subprocess.DEVNULL)

    def run(weighted, ksplit, segs, chunk=64):
        args = [exe, str(int(weighted)), str(ksplit), str(chunk)] + [str(v) for s in segs for v in s]
        out = subprocess.check_output(args).decode().split()
        return (out[0] == "1"), [int(v) for v in out[1:]]
    return run


def _steps(bounds, segs, chunk=64):
    wt = [taps for C, taps in segs for _ in range(C // chunk)]
    return [sum(wt[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]


def test_uniform_groups_keep_the_floor_split(kb):
    for n, ks in ((12, 4), (27, 8), (36, 32), (5, 5), (9, 2)):
        for taps in (9, 1):
            ok, b = kb(True, ks, [(n * 64, taps)])
            assert ok and b == [s * n // ks for s in range(ks + 1)], (n, ks, taps, b)


def test_mixed_groups_are_balanced_by_k_steps(kb):
    segs = [(768, 9), (1536, 1)]                      # 8x8-level decoder conv_res1: 12 3x3 groups + 24 1x1 groups
    ok, b = kb(True, 2, segs)
    assert ok and b == [0, 7, 36] and _steps(b, segs) == [63, 69]
    ok, u = kb(False, 2, segs)
    assert ok and u == [0, 18, 36] and _steps(u, segs) == [114, 18]
    for segs, ks in (([(384, 9), (960, 1)], 16), ([(192, 9), (576, 1)], 6), ([(192, 9), (384, 9), (576, 1)], 8), ([(576, 9), (1344, 1)], 10)):
        n = sum(C // 64 for C, _ in segs)
        ok, b = kb(True, ks, segs)
        ok2, u = kb(False, ks, segs)
        assert ok and ok2 and b[0] == 0 and b[-1] == n and len(b) == ks + 1
        assert all(x < y for x, y in zip(b[:-1], b[1:])), b           # every slice keeps at least one K-group
        assert max(_steps(b, segs)) <= max(_steps(u, segs)), (b, u)     # never worse than the even-groups split


def test_other_chunk_sizes_and_refusals(kb):
    ok, b = kb(True, 3, [(96, 9), (64, 1)], chunk=32)     # the fp32 kernels' 32-channel K-chunk
    assert ok and b[0] == 0 and b[-1] == 5 and all(x < y for x, y in zip(b[:-1], b[1:]))
    assert kb(True, 40, [(768, 9)])[0] is False            # more slices than K-groups
    assert kb(True, 65, [(64 * 100, 1)])[0] is False       # more than 64 slices
