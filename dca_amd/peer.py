"""K-PEER on the host side: the exchange buffers of the data-parallel step's small messages, mapped into every rank.

Each rank allocates one slot buffer and one flag buffer in FINE-GRAINED device memory (hipExtMallocWithFlags(
hipDeviceMallocFinegrained): allocations of their own, so that hipIpcGetMemHandle names exactly them; fine-grained because a
peer's stores arrive over xGMI behind the owner's L2 -- in a coarse-grained hipMalloc allocation the owner's spinning loads are
not guaranteed to see them, system-scope atomics notwithstanding; RCCL allocates its flag buffers the same way),
the 64-byte IPC handles travel once through torch.distributed
(all_gather_object), every rank maps the others' buffers (hipIpcOpenMemHandle: peer access over xGMI; on one GPU shared by
two processes -- the test box -- the same device memory) and keeps the two pointer tables on the device.  After that an
exchange is ONE kernel launch (include/dcahip.h: dcahip_peer_exchange), capturable with the step.

Before the first real exchange the constructor runs a SELF CHECK: a few gathers and reductions of known vectors with a short
timeout, compared on the host; the ranks agree on the outcome through the library's communicator.  Any failure (a handle
that does not map, a store that never becomes visible, a wrong value) raises PeerUnavailable on EVERY rank, and the caller
(dist.Comm.enable_peer_exchange) stays on RCCL and says so.

The HIP runtime is reached through ctypes on the libamdhip64 this process has ALREADY loaded (torch's), never a second copy.
"""
import ctypes

import torch

from . import hip

_HIP = None
IPC_HANDLE_BYTES = 64
TIMEOUT_US = 60 * 1000 * 1000        # a peer that has not arrived after a minute is reported (status + NaN results): rank skew
                                     # from a checkpoint write, a first graph capture or a page-fault stall is seconds at most
SELF_CHECK_TIMEOUT_US = 2 * 1000 * 1000
HIP_DEVICE_MALLOC_FINEGRAINED = 0x1  # hipDeviceMallocFinegrained (hip_runtime_api.h)
# hsa_amd_memory_pool_global_flag_t bits hsa_amd_pointer_info reports for an allocation (hsa/hsa_ext_amd.h)
HSA_FLAG_FINE_GRAINED, HSA_FLAG_COARSE_GRAINED, HSA_FLAG_EXTENDED_SCOPE_FINE_GRAINED = 2, 4, 8
_HSA = None


class PeerUnavailable(RuntimeError):
    """K-PEER cannot be used between these ranks (allocation, mapping or the self check failed): stay on the collectives."""


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
        L.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
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


class _HsaPointerInfo(ctypes.Structure):     # hsa_amd_pointer_info_t up to global_flags (hsa/hsa_ext_amd.h)
    _fields_ = [('size', ctypes.c_uint32), ('type', ctypes.c_int), ('agentBaseAddress', ctypes.c_void_p),
                ('hostBaseAddress', ctypes.c_void_p), ('sizeInBytes', ctypes.c_size_t), ('userData', ctypes.c_void_p),
                ('agentOwner', ctypes.c_uint64), ('global_flags', ctypes.c_uint32), ('_pad', ctypes.c_uint32 * 9)]


def memory_flags(ptr):
    """The effective hsa_amd_memory_pool_global_flag_t bits of the allocation behind a device pointer (hsa_amd_pointer_info on
    the ROCr runtime this process has loaded), or None when the runtime cannot be asked."""
    global _HSA
    try:
        if _HSA is None:
            path = None
            with open('/proc/self/maps') as f:
                for line in f:
                    if 'libhsa-runtime64' in line:
                        path = line.split()[-1]
                        break
            if path is None:
                return None
            _HSA = ctypes.CDLL(path)
            _HSA.hsa_amd_pointer_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(_HsaPointerInfo), ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p]
        info = _HsaPointerInfo()
        info.size = ctypes.sizeof(_HsaPointerInfo)
        if _HSA.hsa_amd_pointer_info(ctypes.c_void_p(ptr), ctypes.byref(info), None, None, None) != 0:
            return None
        if info.size < _HsaPointerInfo.global_flags.offset + 4 or info.type == 0:       # older runtime / unknown pointer
            return None
        return int(info.global_flags)
    except (OSError, AttributeError):
        return None


class PeerExchange:
    """Collective constructor (every rank of `group` calls it); raises PeerUnavailable on EVERY rank if any rank could not
    set the exchange up or its self check failed.  gather(out, local) / reduce(t): see dcahip_peer_exchange."""

    def __init__(self, rank, world, nmax, group=None, device=None, self_check=True):
        import torch.distributed as dist
        self.rank, self.world, self.nmax = int(rank), int(world), int(nmax)
        self.dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.L = hip.lib()
        self.timeout_us = TIMEOUT_US
        self._own, self._opened = [], []
        self.slots = self.flags = None
        err = None
        handles = ()
        try:
            handles = self._allocate()
        except Exception as e:                  # noqa: BLE001 -- whatever went wrong here, the other ranks must not hang below
            err = 'rank %d: %s' % (self.rank, e)
        everyone = [None] * self.world
        dist.all_gather_object(everyone, (err, handles), group=group)
        errs = [e for e, _ in everyone if e]
        if not errs:
            try:
                self._map([h for _, h in everyone])
            except Exception as e:              # noqa: BLE001
                err = 'rank %d: %s' % (self.rank, e)
        self.epoch = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # nobody exchanges before everybody has mapped everything -- and everybody learns whether everybody could
        errs = self._agree(errs + ([err] if err and err not in errs else []), group)
        if not errs and self_check:
            errs = self._agree(self._self_check(), group)
        if errs:
            self.close()
            raise PeerUnavailable('K-PEER is not usable between these ranks: ' + '; '.join(errs))

    def _agree(self, errs, group):
        """Every rank's error list, the same on every rank (an object all-gather on the library's communicator)."""
        import torch.distributed as dist
        everyone = [None] * self.world
        dist.all_gather_object(everyone, list(errs), group=group)
        out = []
        for e in everyone:
            out += [x for x in e if x not in out]
        return out

    def _allocate(self):
        rt = _runtime()
        self.slot_bytes = int(self.L.dcahip_peer_slot_bytes(self.world, self.nmax))
        self.flag_bytes = int(self.L.dcahip_peer_flag_bytes(self.world))
        self._ptrs = []
        for nbytes in (self.slot_bytes, self.flag_bytes):
            p = ctypes.c_void_p()
            _check(rt.hipExtMallocWithFlags(ctypes.byref(p), nbytes, HIP_DEVICE_MALLOC_FINEGRAINED),
                   'hipExtMallocWithFlags(hipDeviceMallocFinegrained)')
            self._own.append(p)
            _check(rt.hipMemset(p, 0, nbytes), 'hipMemset')
            self._ptrs.append(p.value)
        _check(rt.hipDeviceSynchronize(), 'hipDeviceSynchronize')
        # the memory type is the correctness argument of the cross-GPU wait (csrc/dcahip_peer.hip): refuse anything the
        # runtime reports as coarse-grained (None: this runtime cannot be asked -- the allocation flag stands)
        self.mem_flags = [memory_flags(p) for p in self._ptrs]
        fine = HSA_FLAG_FINE_GRAINED | HSA_FLAG_EXTENDED_SCOPE_FINE_GRAINED
        if any(f is not None and ((f & HSA_FLAG_COARSE_GRAINED) or not (f & fine)) for f in self.mem_flags):
            raise RuntimeError('exchange buffers are not fine-grained memory (pool flags %s)' % self.mem_flags)
        handles = []
        for p in self._own:
            h = _IpcHandle()
            _check(rt.hipIpcGetMemHandle(ctypes.byref(h), p), 'hipIpcGetMemHandle')
            handles.append(bytes(bytearray(h)))
        return tuple(handles)

    def _map(self, everyone):
        rt = _runtime()
        slot_ptrs, flag_ptrs = [], []
        for q in range(self.world):
            if q == self.rank:
                slot_ptrs.append(self._ptrs[0]); flag_ptrs.append(self._ptrs[1])
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

    def _self_check(self, rounds=4):
        """Ping-pong before the first real exchange: `rounds` gathers and reductions of vectors every rank can predict, a short
        timeout, results compared on the host.  Returns this rank's error list (empty: fine)."""
        errs = []
        n = min(self.nmax, 8)
        keep, self.timeout_us = self.timeout_us, SELF_CHECK_TIMEOUT_US
        try:
            for k in range(rounds):
                local = torch.arange(n, dtype=torch.float32, device=self.dev) + float(1000 * self.rank + 10 * k)
                out = torch.zeros(self.world * n, dtype=torch.float32, device=self.dev)
                self.gather(out, local)
                red = local.clone()
                self.reduce(red)
                want = torch.cat([torch.arange(n, dtype=torch.float32) + float(1000 * q + 10 * k) for q in range(self.world)])
                got, gred = out.cpu(), red.cpu()                      # (synchronises the stream)
                if int(self.status.item()) != 0:
                    errs.append('rank %d: self check round %d timed out waiting for a peer' % (self.rank, k))
                    break
                if not torch.equal(got, want) or not torch.equal(gred, want.view(self.world, n).sum(0)):
                    errs.append('rank %d: self check round %d read wrong values' % (self.rank, k))
                    break
        except Exception as e:                  # noqa: BLE001
            errs.append('rank %d: self check raised %s' % (self.rank, e))
        finally:
            self.timeout_us = keep
        return errs

    def _launch(self, local, out, n, reduce):
        assert local.dtype == torch.float32 and out.dtype == torch.float32 and local.is_cuda and out.is_cuda
        assert 0 < n <= self.nmax
        hip.check(self.L.dcahip_peer_exchange(hip.ptr(local), int(n), hip.ptr(self.slots), hip.ptr(self.flags), self.rank,
                                              self.world, self.nmax, hip.ptr(self.epoch), hip.ptr(out), int(reduce),
                                              hip.ptr(self.status), int(self.timeout_us), hip.stream()), 'peer_exchange')

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
        """Host synchronisation point: raises if a peer went missing in an exchange since the last check (the results of that
        exchange were NaN: nothing computed from it is usable)."""
        if int(self.status.item()) != 0:
            raise RuntimeError('K-PEER: a rank did not arrive at an exchange within %.0f s (rank %d of %d gave up waiting; the '
                               'exchange returned NaN)' % (self.timeout_us / 1e6, self.rank, self.world))

    def close(self):
        rt = _runtime()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        for p in self._opened:
            rt.hipIpcCloseMemHandle(p)
        for p in self._own:
            rt.hipFree(p)
        self._opened, self._own = [], []
