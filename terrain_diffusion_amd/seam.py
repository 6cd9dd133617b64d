"""ctypes binding of libtd_seam.so (C-ABI declared in include/td_seam.h): the shard plan and the RCCL seam exchange for hosts that shard
through the C-ABI.  The Python samplers of parallel.py use torch.distributed by default; passing `seam_comm=SeamComm.create(...)` sends the
window outputs through td_seam_exchange_windows instead — ONE grouped ncclSend/ncclRecv on the stream the engine runs on.

The library is separate from libtd_engine.so so that the engine carries no RCCL dependency.  No fallback: a missing library raises.
"""
import ctypes as C
import os

import torch

from ._lib import TdError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtd_seam.so")

OWN, NEEDED, SENDS, RECVS = 0, 1, 2, 3
ID_BYTES = 128


class SeamMsg(C.Structure):
    _fields_ = [("peer", C.c_int32), ("reserved", C.c_int32), ("offset", C.c_int64), ("bytes", C.c_int64)]


_P = C.c_void_p
_I32P = C.POINTER(C.c_int32)
_MSGP = C.POINTER(SeamMsg)
_SIGS = {
    "td_seam_last_error": (C.c_char_p, []),
    "td_seam_plan_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "td_seam_plan_destroy": (None, [_P]),
    "td_seam_plan_mesh": (C.c_int, [_P, _I32P]),
    "td_seam_plan_region": (C.c_int, [_P, C.c_int, _I32P]),
    "td_seam_plan_starts": (C.c_int, [_P, C.c_int, _I32P, C.c_int]),
    "td_seam_plan_windows": (C.c_int, [_P, C.c_int, C.c_int, _I32P, _I32P, C.c_int]),
    "td_seam_plan_messages": (C.c_int, [_P, C.c_int, C.c_int64, _MSGP, C.POINTER(C.c_int), _MSGP, C.POINTER(C.c_int), C.c_int]),
    "td_seam_unique_id": (C.c_int, [_P]),
    "td_seam_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "td_seam_comm_adopt": (C.c_int, [_P, C.POINTER(_P)]),
    "td_seam_comm_destroy": (None, [_P]),
    "td_seam_comm_info": (C.c_int, [_P, _I32P]),
    "td_seam_exchange": (C.c_int, [_P, _P, _MSGP, C.c_int, _P, _MSGP, C.c_int, _P]),
    "td_seam_exchange_windows": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TdError(f"{LIB_PATH} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()'). There is no fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(l, name)
            f.restype, f.argtypes = res, args
        _lib = l
    return _lib


def check(code):
    if code < 0:
        raise TdError(f"libtd_seam: {lib().td_seam_last_error().decode()} (code {code})")
    return code


class CShardPlan:
    """td_seam_plan: the C-ABI's restatement of parallel.ShardPlan (same constructor meaning, same lists)."""

    def __init__(self, H, W, tile_size, world, stride=None, extended=False):
        self._h = _P()
        check(lib().td_seam_plan_create(int(H), int(W), int(tile_size), int(stride or 0), int(world), int(bool(extended)), C.byref(self._h)))
        self.H, self.W, self.size, self.world, self.extended = H, W, tile_size, world, bool(extended)
        m = (C.c_int32 * 4)()
        check(lib().td_seam_plan_mesh(self._h, m))
        self.pr, self.pc, self.n_rows, self.n_cols = (int(v) for v in m)
        self.h_starts, self.w_starts = self._starts(0), self._starts(1)
        self.regions = [self._region(r) for r in range(world)]

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.td_seam_plan_destroy(self._h)
            self._h = None

    def _starts(self, axis):
        n = check(lib().td_seam_plan_starts(self._h, axis, None, 0))
        buf = (C.c_int32 * n)()
        check(lib().td_seam_plan_starts(self._h, axis, buf, n))
        return [int(v) for v in buf]

    def _region(self, rank):
        g = (C.c_int32 * 4)()
        check(lib().td_seam_plan_region(self._h, rank, g))
        return tuple(int(v) for v in g)

    def windows_of(self, rank, kind):
        """[(ic, jc)], [peer] of one of the lists OWN / NEEDED / SENDS / RECVS."""
        n = check(lib().td_seam_plan_windows(self._h, rank, kind, None, None, 0))
        ij, peer = (C.c_int32 * (2 * max(n, 1)))(), (C.c_int32 * max(n, 1))()
        check(lib().td_seam_plan_windows(self._h, rank, kind, ij, peer, n))
        return [(int(ij[2 * k]), int(ij[2 * k + 1])) for k in range(n)], [int(peer[k]) for k in range(n)]

    def messages(self, rank, window_bytes):
        """(sends, recvs) as lists of (peer, offset, bytes)."""
        ns, nr = C.c_int(), C.c_int()
        check(lib().td_seam_plan_messages(self._h, rank, window_bytes, None, C.byref(ns), None, C.byref(nr), 0))
        cap = max(ns.value, nr.value, 1)
        s, r = (SeamMsg * cap)(), (SeamMsg * cap)()
        check(lib().td_seam_plan_messages(self._h, rank, window_bytes, s, C.byref(ns), r, C.byref(nr), cap))
        return ([(m.peer, m.offset, m.bytes) for m in s[:ns.value]], [(m.peer, m.offset, m.bytes) for m in r[:nr.value]])


def _msg_array(msgs):
    arr = (SeamMsg * max(len(msgs), 1))()
    for k, (peer, off, nbytes) in enumerate(msgs):
        arr[k].peer, arr[k].offset, arr[k].bytes = peer, off, nbytes
    return arr


class SeamComm:
    """td_seam_comm: one RCCL communicator per process / GPU."""

    def __init__(self, handle, world, rank, device):
        self._h, self.world, self.rank, self.device = handle, world, rank, device
        self._plans = {}

    @classmethod
    def create(cls, device, group=None):
        """Collective over the ranks of `group` (torch.distributed, any backend: it only carries the 128-byte id); world 1 without it."""
        import torch.distributed as dist
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = C.create_string_buffer(ID_BYTES)
        if rank == 0:
            check(lib().td_seam_unique_id(ident))
        if world > 1:
            box = [ident.raw]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = C.create_string_buffer(box[0], ID_BYTES)
        h = _P()
        check(lib().td_seam_comm_create(idx, world, rank, ident, C.byref(h)))
        return cls(h, world, rank, idx)

    def close(self):
        if self._h:
            lib().td_seam_comm_destroy(self._h)
            self._h = None

    # a communicator is a device resource: usable as `with SeamComm.create(...) as comm:`; a forgotten one is destroyed with the object
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: the library may already be gone
            pass

    def info(self):
        v = (C.c_int32 * 3)()
        check(lib().td_seam_comm_info(self._h, v))
        return tuple(int(x) for x in v)

    def plan_for(self, plan):
        """The td_seam_plan of a parallel.ShardPlan (cached)."""
        if isinstance(plan, CShardPlan):
            return plan
        key = (plan.H, plan.W, plan.size, plan.world, plan.extended, tuple(plan.h_starts), tuple(plan.w_starts))
        if key not in self._plans:
            c = CShardPlan(plan.H, plan.W, plan.size, plan.world, stride=plan.stride, extended=plan.extended)
            if c.h_starts != list(plan.h_starts) or c.w_starts != list(plan.w_starts):
                raise TdError("the C shard plan's window grid differs from the Python plan's")
            self._plans[key] = c
        return self._plans[key]

    def exchange(self, send_buf, sends, recv_buf, recvs, stream=None):
        """td_seam_exchange: lists of (peer, byte offset, bytes) into two device tensors; enqueue-only on `stream` (default: torch's current)."""
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        check(lib().td_seam_exchange(self._h, send_buf.data_ptr() if send_buf is not None else None, _msg_array(sends), len(sends),
                                     recv_buf.data_ptr() if recv_buf is not None else None, _msg_array(recvs), len(recvs), st.cuda_stream))

    def exchange_windows(self, plan, my_tiles):
        """The C-ABI form of parallel.exchange_windows: {(ic, jc): tile} for every window this rank's region needs."""
        if not my_tiles.is_cuda or not my_tiles.is_contiguous():
            raise TdError("td_seam_exchange_windows moves device memory: my_tiles must be a contiguous CUDA tensor")
        cplan = self.plan_for(plan)
        own, _ = cplan.windows_of(self.rank, OWN)
        if my_tiles.shape[0] != len(own):
            raise TdError(f"my_tiles holds {my_tiles.shape[0]} windows, the plan gives rank {self.rank} {len(own)}")
        wins, _ = cplan.windows_of(self.rank, RECVS)
        recv = torch.empty((len(wins),) + tuple(my_tiles.shape[1:]), dtype=my_tiles.dtype, device=my_tiles.device)
        wb = my_tiles[0].numel() * my_tiles.element_size() if len(own) else recv[0].numel() * recv.element_size()
        st = torch.cuda.current_stream(my_tiles.device)
        check(lib().td_seam_exchange_windows(self._h, cplan._h, my_tiles.data_ptr(), recv.data_ptr(), wb, st.cuda_stream))
        local = {w: i for i, w in enumerate(own)}
        need, owners = cplan.windows_of(self.rank, NEEDED)
        have = {w: my_tiles[local[w]] for w, o in zip(need, owners) if o == self.rank}
        for k, w in enumerate(wins):
            have[w] = recv[k]
        return have
