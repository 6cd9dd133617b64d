"""CPU: libtd_seam.so (include/td_seam.h) loads and exports what the header declares; its shard plan (host arithmetic, no GPU) equals
parallel.ShardPlan list by list; the message cuts of sender and receiver agree pairwise (an RCCL exchange with unequal counts would hang);
argument errors are refused with a message.  No RCCL call is made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [  # H, W, tile, stride, world
    (288, 288, 64, None, 1), (288, 288, 64, None, 2), (288, 288, 64, None, 4), (288, 288, 64, None, 8), (160, 224, 64, None, 2), (160, 224, 64, None, 3),
    (1056, 1056, 64, None, 8), (1056, 1056, 64, None, 6), (40, 56, 16, None, 2), (40, 56, 16, None, 4), (100, 37, 16, 8, 3), (64, 640, 64, None, 4),
    (640, 64, 64, None, 5), (96, 96, 64, 32, 4), (200, 200, 48, 16, 8), (130, 70, 32, 24, 2), (512, 384, 128, 64, 6), (33, 33, 16, 5, 7),
]


def _declared():
    text = open(os.path.join(ROOT, "include", "td_seam.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(td_seam_[a-z0-9_]+)\s*\(", text)))


def test_seam_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from terrain_diffusion_amd import seam
    lib = C.CDLL(seam.LIB_PATH)
    declared = _declared()
    assert len(declared) == 15, declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/td_seam.h but not exported"
    assert set(seam.EXPORTS) == set(declared), set(seam.EXPORTS) ^ set(declared)


@pytest.mark.parametrize("extended", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_c_shard_plan_equals_the_python_plan(case, extended):
    from terrain_diffusion_amd import seam
    from terrain_diffusion_amd.parallel import ShardPlan
    H, W, tile, stride, world = case
    try:
        py = ShardPlan(H, W, tile, world, stride=stride, extended=extended)
    except ValueError as e:   # more ranks than windows along both axes: both sides must refuse, with the same sentence
        with pytest.raises(seam.TdError, match=re.escape(str(e))):
            seam.CShardPlan(H, W, tile, world, stride=stride, extended=extended)
        return
    c = seam.CShardPlan(H, W, tile, world, stride=stride, extended=extended)
    assert (c.pr, c.pc) == (py.pr, py.pc) and (c.n_rows, c.n_cols) == (len(py.h_starts), len(py.w_starts))
    assert c.h_starts == py.h_starts and c.w_starts == py.w_starts
    assert c.regions == [tuple(r) for r in py.regions]
    for r in range(world):
        own, peers = c.windows_of(r, seam.OWN)
        assert own == py.windows[r] and set(peers) <= {r}
        need, owners = c.windows_of(r, seam.NEEDED)
        assert need == py.needed[r] and owners == [py.owner[w] for w in need]
        sends, dst = c.windows_of(r, seam.SENDS)
        want = [(w, d) for (s, d), wins in sorted(py.sends.items()) if s == r for w in wins]
        assert list(zip(sends, dst)) == want
        recvs, src = c.windows_of(r, seam.RECVS)
        want = [(w, s) for (s, d), wins in sorted(py.sends.items()) if d == r for w in wins]
        assert list(zip(recvs, src)) == want


@pytest.mark.parametrize("case", CASES)
def test_message_cuts_of_sender_and_receiver_agree(case):
    """for every ordered pair (s, d): the byte counts s posts towards d equal, message by message, what d posts from s; the send offsets cover
    exactly the crossing windows in s's own-order array, the receive offsets d's slot array without gaps."""
    from terrain_diffusion_amd import seam
    H, W, tile, stride, world = case
    wb = 5 * tile * tile * 4
    for extended in (False, True):
        try:
            c = seam.CShardPlan(H, W, tile, world, stride=stride, extended=extended)
        except seam.TdError:
            return
        msgs = [c.messages(r, wb) for r in range(world)]
        for s in range(world):
            own, _ = c.windows_of(s, seam.OWN)
            local = {w: i for i, w in enumerate(own)}
            sent_wins, sent_to = c.windows_of(s, seam.SENDS)
            for d in range(world):
                if d == s:
                    continue
                out = [(off, n) for peer, off, n in msgs[s][0] if peer == d]
                inn = [(off, n) for peer, off, n in msgs[d][1] if peer == s]
                assert [n for _, n in out] == [n for _, n in inn], (s, d)
                # the sender's messages, unrolled to window indices, are the crossing windows in order
                idx = [off // wb + k for off, n in out for k in range(n // wb)]
                assert idx == [local[w] for w, to in zip(sent_wins, sent_to) if to == d]
            # receive slots: contiguous from 0
            pos = 0
            for _, off, n in msgs[s][1]:
                assert off == pos and n % wb == 0
                pos += n
            assert pos == len(c.windows_of(s, seam.RECVS)[0]) * wb
        assert all(peer != r for r in range(world) for lst in msgs[r] for peer, _, _ in lst)


def test_neighbouring_windows_travel_as_one_message():
    """configs[3]'s 32x32 window grid on 8 ranks (2x4 mesh, 16x8 windows per block): a block's last window row is contiguous in its own-order
    array -> ONE message to the neighbour below, not one per window; the last column is strided -> one message per window."""
    from terrain_diffusion_amd import seam
    c = seam.CShardPlan(1056, 1056, 64, 8)
    assert (c.pr, c.pc, c.n_rows, c.n_cols) == (2, 4, 32, 32)
    wb = 5 * 64 * 64 * 4
    sends, _ = c.messages(0, wb)
    wins, dst = c.windows_of(0, seam.SENDS)
    assert len(sends) < len(wins)
    below = c.pc   # rank of the block under rank 0
    to_below = [n for peer, _, n in sends if peer == below]
    assert len(to_below) == 1 and to_below[0] == wb * sum(1 for d in dst if d == below)


def test_seam_argument_errors_are_refused_with_a_message():
    from terrain_diffusion_amd import seam
    lib = seam.lib()
    h = C.c_void_p()
    assert lib.td_seam_plan_create(0, 64, 64, 0, 1, 0, C.byref(h)) == -1 and b"positive" in lib.td_seam_last_error()
    with pytest.raises(seam.TdError, match="cannot place 64 ranks on a 3x3 window grid"):
        seam.CShardPlan(128, 128, 64, 64)
    c = seam.CShardPlan(128, 128, 64, 2)
    assert lib.td_seam_plan_windows(c._h, 2, seam.OWN, None, None, 0) == -1 and b"rank outside" in lib.td_seam_last_error()
    assert lib.td_seam_plan_windows(c._h, 0, 9, None, None, 0) == -1 and b"kind" in lib.td_seam_last_error()
    ns, nr = C.c_int(), C.c_int()
    assert lib.td_seam_plan_messages(c._h, 0, 0, None, C.byref(ns), None, C.byref(nr), 0) == -1 and b"window_bytes" in lib.td_seam_last_error()
    assert lib.td_seam_exchange(None, None, None, 0, None, None, 0, None) == -1 and b"NULL communicator" in lib.td_seam_last_error()
    assert lib.td_seam_exchange_windows(None, c._h, None, None, 4, None) == -1
    assert lib.td_seam_comm_adopt(None, C.byref(h)) == -1
    assert lib.td_seam_unique_id(None) == -1


def test_exchange_windows_rejects_a_foreign_communicator():
    """parallel.exchange_windows with seam_comm: the communicator must be this rank's, of the plan's world (checked before any RCCL call)."""
    import torch
    from terrain_diffusion_amd import seam
    from terrain_diffusion_amd.parallel import ShardPlan, exchange_windows
    plan = ShardPlan(160, 224, 64, 2)
    fake = seam.SeamComm(None, world=1, rank=0, device=0)
    with pytest.raises(ValueError, match="seam_comm is rank 0 of 1"):
        exchange_windows(plan, 0, torch.zeros(len(plan.windows[0]), 5, 64, 64), seam_comm=fake)


def test_headers_are_plain_c_and_a_c_host_reads_the_same_plan(tmp_path):
    """include/td_engine.h and include/td_seam.h compile as C99 (no C++ in the boundary), and a C program linked against libtd_seam.so
    (tests/seam_host.c: the non-Python host the header is for) prints the plan parallel.ShardPlan computes."""
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    from terrain_diffusion_amd import seam
    from terrain_diffusion_amd.parallel import ShardPlan
    inc = os.path.join(ROOT, "include")
    for h in ("td_engine.h", "td_seam.h"):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", os.path.join(inc, h)])
    exe = str(tmp_path / "seam_host")
    libdir = os.path.dirname(seam.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, os.path.join(ROOT, "tests", "seam_host.c"), "-o", exe,
                           "-L", libdir, "-ltd_seam", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    wb = 5 * 64 * 64 * 4
    for (H, W, tile, stride, world, extended) in ((1056, 1056, 64, 0, 8, 0), (1056, 1056, 64, 0, 8, 1), (160, 224, 64, 0, 2, 0), (100, 37, 16, 8, 3, 1)):
        py = ShardPlan(H, W, tile, world, stride=stride or None, extended=bool(extended))
        c = seam.CShardPlan(H, W, tile, world, stride=stride or None, extended=bool(extended))
        for rank in range(world):
            out = subprocess.run([exe, *map(str, (H, W, tile, stride, world, extended, rank, wb))], capture_output=True, text=True, timeout=60)
            assert out.returncode == 0, out.stderr
            lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
            assert lines["mesh"] == f"{py.pr} {py.pc} {len(py.h_starts)} {len(py.w_starts)}"
            assert lines["region"] == " ".join(str(v) for v in py.regions[rank])

            def fmt(wins, peers):
                return f"{len(wins)}:" + "".join(f" {i},{j}@{p_}" for (i, j), p_ in zip(wins, peers))
            assert lines["own"] == fmt(py.windows[rank], [rank] * len(py.windows[rank]))
            assert lines["needed"] == fmt(py.needed[rank], [py.owner[w] for w in py.needed[rank]])
            snd = [(w, d) for (s_, d), wins in sorted(py.sends.items()) if s_ == rank for w in wins]
            rcv = [(w, s_) for (s_, d), wins in sorted(py.sends.items()) if d == rank for w in wins]
            assert lines["sends"] == fmt([w for w, _ in snd], [d for _, d in snd])
            assert lines["recvs"] == fmt([w for w, _ in rcv], [s_ for _, s_ in rcv])
            ms, mr = c.messages(rank, wb)
            assert lines["send_msgs"] == f"{len(ms)}:" + "".join(f" {p_}:{o}+{n}" for p_, o, n in ms)
            assert lines["recv_msgs"] == f"{len(mr)}:" + "".join(f" {p_}:{o}+{n}" for p_, o, n in mr)
