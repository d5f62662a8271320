"""K-PEER on the host side: the exchange buffers of the data-parallel step's small messages, mapped into every rank.

Each rank allocates one slot buffer and one flag buffer in device memory (plain hipMalloc allocations of their own, so
that hipIpcGetMemHandle names exactly them), the 64-byte IPC handles travel once through torch.distributed
(all_gather_object), every rank maps the others' buffers (hipIpcOpenMemHandle: peer access over xGMI; on one GPU shared by
two processes -- the test box -- the same device memory) and keeps the two pointer tables on the device.  After that an
exchange is ONE kernel launch (include/dcahip.h: dcahip_peer_exchange), capturable with the step.

The HIP runtime is reached through ctypes on the libamdhip64 this process has ALREADY loaded (torch's), never a second copy.
"""
import ctypes

import torch

from . import hip

_HIP = None
IPC_HANDLE_BYTES = 64
MAX_SPIN = 4000000           # polls before a missing peer is reported (a few seconds; a healthy exchange needs a few hundred)


class _IpcHandle(ctypes.Structure):          # hipIpcMemHandle_t: 64 opaque bytes, passed BY VALUE to hipIpcOpenMemHandle
    _fields_ = [('reserved', ctypes.c_char * IPC_HANDLE_BYTES)]


def _runtime():
    """ctypes handle of the HIP runtime library mapped into this process."""
    global _HIP
    if _HIP is None:
        path = None
        with open('/proc/self/maps') as f:
            for line in f:
                if 'libamdhip64' in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError('the HIP runtime is not loaded in this process (no GPU build of torch?)')
        L = ctypes.CDLL(path)
        L.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        L.hipFree.argtypes = [ctypes.c_void_p]
        L.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        L.hipIpcGetMemHandle.argtypes = [ctypes.POINTER(_IpcHandle), ctypes.c_void_p]
        L.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), _IpcHandle, ctypes.c_uint]
        L.hipIpcCloseMemHandle.argtypes = [ctypes.c_void_p]
        L.hipDeviceSynchronize.argtypes = []
        _HIP = L
    return _HIP


def _check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed with HIP error %d' % (what, rc))


class PeerExchange:
    """Collective constructor (every rank of `group` calls it).  gather(out, local) / reduce(t): see dcahip_peer_exchange."""

    def __init__(self, rank, world, nmax, group=None, device=None):
        import torch.distributed as dist
        self.rank, self.world, self.nmax = int(rank), int(world), int(nmax)
        self.dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.L = hip.lib()
        rt = _runtime()
        self.slot_bytes = int(self.L.dcahip_peer_slot_bytes(self.world, self.nmax))
        self.flag_bytes = int(self.L.dcahip_peer_flag_bytes(self.world))
        self._own, self._opened = [], []
        ptrs = []
        for nbytes in (self.slot_bytes, self.flag_bytes):
            p = ctypes.c_void_p()
            _check(rt.hipMalloc(ctypes.byref(p), nbytes), 'hipMalloc')
            _check(rt.hipMemset(p, 0, nbytes), 'hipMemset')
            self._own.append(p)
            ptrs.append(p.value)
        _check(rt.hipDeviceSynchronize(), 'hipDeviceSynchronize')
        handles = []
        for p in self._own:
            h = _IpcHandle()
            _check(rt.hipIpcGetMemHandle(ctypes.byref(h), p), 'hipIpcGetMemHandle')
            handles.append(bytes(bytearray(h)))
        everyone = [None] * self.world
        dist.all_gather_object(everyone, tuple(handles), group=group)
        slot_ptrs, flag_ptrs = [], []
        for q in range(self.world):
            if q == self.rank:
                slot_ptrs.append(ptrs[0]); flag_ptrs.append(ptrs[1])
                continue
            opened = []
            for hb in everyone[q]:
                h = _IpcHandle.from_buffer_copy(hb)
                p = ctypes.c_void_p()
                _check(rt.hipIpcOpenMemHandle(ctypes.byref(p), h, 1), 'hipIpcOpenMemHandle (rank %d -> %d)' % (self.rank, q))
                self._opened.append(p)
                opened.append(p.value)
            slot_ptrs.append(opened[0]); flag_ptrs.append(opened[1])
        self.slots = torch.tensor(slot_ptrs, dtype=torch.int64).to(self.dev)
        self.flags = torch.tensor(flag_ptrs, dtype=torch.int64).to(self.dev)
        self.epoch = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        dist.barrier(group=group)               # nobody exchanges before everybody has mapped everything

    def _launch(self, local, out, n, reduce):
        assert local.dtype == torch.float32 and out.dtype == torch.float32 and local.is_cuda and out.is_cuda
        assert 0 < n <= self.nmax
        hip.check(self.L.dcahip_peer_exchange(hip.ptr(local), int(n), hip.ptr(self.slots), hip.ptr(self.flags), self.rank,
                                              self.world, self.nmax, hip.ptr(self.epoch), hip.ptr(out), int(reduce),
                                              hip.ptr(self.status), MAX_SPIN, hip.stream()), 'peer_exchange')

    def gather(self, out, local):
        """out [world * n] = every rank's `local` [n], in rank order."""
        n = local.numel()
        assert out.numel() >= self.world * n and local.is_contiguous() and out.is_contiguous()
        self._launch(local, out, n, 0)
        return out

    def reduce(self, t):
        """t [n] <- sum over the ranks, added in rank order (the same bits on every rank)."""
        assert t.is_contiguous()
        self._launch(t, t, t.numel(), 1)
        return t

    def check(self):
        """Host synchronisation point: raises if a peer went missing in an exchange since the last check."""
        if int(self.status.item()) != 0:
            raise RuntimeError('K-PEER: a rank did not arrive at an exchange (rank %d of %d gave up waiting)' % (self.rank, self.world))

    def close(self):
        rt = _runtime()
        torch.cuda.synchronize()
        for p in self._opened:
            rt.hipIpcCloseMemHandle(p)
        for p in self._own:
            rt.hipFree(p)
        self._opened, self._own = [], []
