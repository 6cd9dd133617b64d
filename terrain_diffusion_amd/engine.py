"""Engine handle + pointer plumbing.  torch is used only as the owner of host/device buffers."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check

_engines = {}


class Engine:
    """One engine per GPU (td_engine_create); single caller thread per handle."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        check(lib().td_engine_create(int(device_id), C.byref(self._h)))
        self.device_id = int(device_id)
        self._async = False

    def set_async(self, on):
        """option "async": calls on device tensors only enqueue (see on_stream)."""
        self.set_option("async", 1 if on else 0)

    def set_option(self, key, value):
        check(lib().td_engine_set_option(self._h, key.encode(), int(value)))
        if key == "async":   # the shadow on_stream restores on exit follows the documented C-ABI knob, whoever sets it (round-5 advisor)
            self._async = bool(int(value))

    def synchronize(self):
        check(lib().td_engine_synchronize(self._h))

    def profile_read(self, reset=True):
        """(conv_ms, conv_launches, other_ms, other_launches) accumulated while option 'profile' was 1."""
        a, b, c, d = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
        check(lib().td_engine_profile_read(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), int(reset)))
        return a.value, b.value, c.value, d.value

    def profile_read_glds(self, reset=True):
        """(ms, algorithmic_flop, launches) of the td::conv_glds_kernel launches recorded in profile mode."""
        a, b, c = C.c_double(), C.c_double(), C.c_int64()
        check(lib().td_engine_profile_read_glds(self._h, C.byref(a), C.byref(b), C.byref(c), int(reset)))
        return a.value, b.value, c.value

    def profile_ops(self):
        """[(label, ms, launches)] per fused op, accumulated in profile mode."""
        buf = C.create_string_buffer(1 << 20)
        check(lib().td_engine_profile_dump(self._h, buf, len(buf)))
        rows = [l.split("\t") for l in buf.value.decode().splitlines() if l]
        return [(r[0], float(r[1]), int(r[2])) for r in rows]

    @property
    def stream(self):
        return lib().td_engine_stream(self._h)

    def set_stream(self, stream=None):
        """Caller-supplied stream (td_engine_set_stream): a torch.cuda.Stream / raw hipStream_t handle, or None for the engine's own stream.
        torch's DEFAULT stream is the legacy NULL stream, which hipGraph capture refuses: pass a stream you created."""
        h = getattr(stream, "cuda_stream", stream)
        if stream is not None and not h:
            raise ValueError("the legacy default (NULL) stream cannot carry the engine's captured graphs: create a torch.cuda.Stream()")
        check(lib().td_engine_set_stream(self._h, C.c_void_p(int(h)) if h else None))
        if h:
            _SHARED_STREAM[self.device_id] = int(h)
        else:
            _SHARED_STREAM.pop(self.device_id, None)

    def on_stream(self, stream, asynchronous=True):
        """Context manager: run the engine on `stream`; with asynchronous=True calls on device tensors only enqueue (option "async").
        A torch.cuda.Stream is also made torch's CURRENT stream for the duration (the producers of the engine's inputs and the consumers of its
        outputs must be ordered on that same stream: `ptr()` skips its host synchronisation exactly when torch's current stream is the stream the
        engine launches on).  With a raw hipStream_t handle the caller is responsible for that ordering.  One Engine per device (get_engine)."""
        eng = self

        class _Ctx:
            def __enter__(self_):
                # re-entrant: remember what the engine ran on, and restore exactly that on exit (a sharded sampler called inside a caller's own
                # on_stream block must not drop the caller's stream / enqueue-only mode for the rest of that block)
                self_.prev_handle = _SHARED_STREAM.get(eng.device_id)
                self_.prev_async = eng._async
                self_.tctx = None
                if isinstance(stream, torch.cuda.Stream):
                    # torch side streams do not wait for the previously current stream: tensors produced there just before entry (conditioning
                    # inputs, tiles cached by an earlier call) would be read on `stream` unordered, and ptr() no longer synchronises the host
                    # once the engine shares torch's current stream
                    stream.wait_stream(torch.cuda.current_stream(stream.device))
                    self_.tctx = torch.cuda.stream(stream)
                    self_.tctx.__enter__()
                try:
                    eng.set_stream(stream)
                    eng.set_async(asynchronous)
                except BaseException:
                    if self_.tctx is not None:
                        self_.tctx.__exit__(None, None, None)
                    raise
                return eng

            def __exit__(self_, *exc):
                eng.set_async(False)
                eng.set_stream(self_.prev_handle)     # drains the stream and releases the staging buffers
                eng.set_async(self_.prev_async)
                if self_.tctx is not None:
                    self_.tctx.__exit__(*exc)
                    # whoever continues on the outer stream sees everything enqueued inside the block
                    torch.cuda.current_stream(stream.device).wait_stream(stream)
                return False
        return _Ctx()

    def close(self):
        if self._h:
            lib().td_engine_destroy(self._h)
            self._h = C.c_void_p()


def get_engine(device=None) -> Engine:
    """Engine for a torch device spec ('cuda', 'cuda:1', int, None).  Raises on 'cpu': no CPU path exists."""
    if device is None:
        idx = 0
    elif isinstance(device, int):
        idx = device
    else:
        d = torch.device(device)
        if d.type != "cuda":
            raise RuntimeError("terrain_diffusion_amd runs on MI355X only (device must be 'cuda[:i]'); there is no CPU fallback")
        idx = d.index or 0
    if idx not in _engines:
        _engines[idx] = Engine(idx)
    return _engines[idx]


_SHARED_STREAM = {}   # device index -> raw handle of the caller stream the engine currently launches on (Engine.set_stream)


def engine_on_current_stream(device) -> bool:
    """True when the engine of `device` launches on torch's CURRENT stream (Engine.set_stream / on_stream): work enqueued by torch on that
    stream -- RCCL transfers included -- is then ordered with the engine's kernels by the stream itself and needs no host synchronisation."""
    d = torch.device(device)
    idx = d.index if d.index is not None else torch.cuda.current_device()
    return _SHARED_STREAM.get(idx) == torch.cuda.current_stream(d).cuda_stream


def ptr(t):
    """raw pointer of a contiguous fp32 torch tensor / numpy array (host or device), or None.
    C-ABI contract: device buffers handed to the engine must be COMPLETE (the engine launches on its own non-blocking stream and
    does not know the caller's streams) and results are complete on return.  Torch produces tensors asynchronously on its current
    stream (uploads, stack/clone/contiguous), so a device pointer is only taken after that stream has been synchronised."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return C.c_void_p(t.ctypes.data)
    assert t.is_contiguous()
    if t.is_cuda:
        cs = torch.cuda.current_stream(t.device)
        # ... unless the engine has been put ON that stream (Engine.set_stream): then stream order is all that is needed
        if _SHARED_STREAM.get(t.device.index if t.device.index is not None else torch.cuda.current_device()) != cs.cuda_stream:
            cs.synchronize()
    return C.c_void_p(t.data_ptr())


def f32(t, device=None):
    """contiguous fp32 tensor (keeps device unless `device` given)."""
    t = torch.as_tensor(t)
    t = t.to(dtype=torch.float32)
    if device is not None:
        t = t.to(device)
    return t.contiguous()
