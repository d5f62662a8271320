"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" for the CPU tests).

The reference is single-process: data parallelism over cells is a capability
this implementation adds.  Cells are independent given the parameters, so the path shards by
rows with exactly three exchanges per optimizer step:
  1. the flat fp32 gradient buffer in two all-reduce buckets: [heads | loss] as soon as the heads'
     backward is done (overlaps the hidden stack's backward), [hidden layers] at the end,
  2. per BatchNormalization layer, forward: an all-gather of one (mean, M2) pair per rank
     (merged with Chan's formula -> identical to single-GPU statistics of the global batch),
  3. per BatchNormalization layer, backward: an all-reduce of [sum dy, sum dy*xhat].
"""
import os

import numpy as np
import torch
import torch.distributed as dist


class TorchDistComm:
    dp = True             # the data-parallel step (exchanges in place) -- also with ONE rank (init_from_env(force=True))

    def __init__(self, group=None):
        assert dist.is_initialized()
        self.group = group
        self.world = dist.get_world_size(group)
        # the exchanges may be captured into hipGraphs with the step only over RCCL (gloo stages through the host)
        self.capturable = dist.get_backend(group) == 'nccl'          # (K-PEER's launches are capturable whatever the backend)
        self.rank = dist.get_rank(group)
        # a second communicator (its own RCCL stream) for the bulk gradient bucket: on the main
        # one it would queue the small SyncBN exchanges of the backward pass behind 15 MB
        self.bulk = dist.new_group(ranks=list(range(dist.get_world_size())) if group is None else None) \
            if group is None else group

    # K-PEER (dca_amd/peer.py): the small fp32 exchanges (SyncBN statistics) as one kernel launch over IPC-mapped buffers
    # instead of a library call; None = the library's collectives (the default: EngineConfig.dp_peer_exchange)
    peer = None

    def enable_peer_exchange(self, nmax):
        """Collective.  Small float32 device tensors (<= nmax elements) of all_gather_into(name='all_gather_small') and
        all_reduce_sum go through dcahip_peer_exchange from here on."""
        from .peer import PeerExchange, PeerUnavailable
        if self.peer is not None and self.peer.nmax < nmax:
            self.peer.close()
            self.peer = None
        if self.peer is None:
            try:
                self.peer = PeerExchange(self.rank, self.world, nmax, group=self.group)
            except PeerUnavailable as e:
                # (raised on every rank alike: allocation, mapping or the init-time ping-pong failed somewhere) -- the small
                # exchanges stay on the library's collectives, and the run says so
                import warnings
                warnings.warn('dca_amd: %s -- the SyncBN exchanges stay on RCCL collectives' % e, RuntimeWarning)
                self.peer = None
        return self.peer

    def _peer_takes(self, t, n):
        return (self.peer is not None and t.is_cuda and t.dtype == torch.float32 and 0 < n <= self.peer.nmax and t.is_contiguous())

    # every exchange can be timed: with .timer = {} the communicator brackets each call (and each wait for an
    # asynchronous one) with events on the compute stream -- what the stream spends there is the EXPOSED communication
    # time of the step (bench.py --gpus N reports it per category)
    timer = None

    class _Span:
        def __init__(self, comm, name, t):
            self.comm, self.name, self.cuda = comm, name, t.is_cuda

        def __enter__(self):
            if self.comm.timer is None:
                return self
            # inside a graph capture the call is only counted: events recorded there have no time of their own
            self.captured = self.cuda and torch.cuda.is_current_stream_capturing()
            if self.captured:
                return self
            if self.cuda:
                self.s = torch.cuda.Event(enable_timing=True); self.e = torch.cuda.Event(enable_timing=True)
                self.s.record()
            else:
                import time
                self.t0 = time.perf_counter()
            return self

        def __exit__(self, *a):
            if self.comm.timer is None:
                return False
            if self.captured:
                self.comm.timer.setdefault(self.name, []).append(None)
                return False
            if self.cuda:
                self.e.record()
                self.comm.timer.setdefault(self.name, []).append((self.s, self.e))
            else:
                import time
                self.comm.timer.setdefault(self.name, []).append(time.perf_counter() - self.t0)
            return False

    def timer_summary(self):
        """{category: (calls, total ms)} of the spans recorded since .timer was set."""
        out = {}
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        for k, lst in (self.timer or {}).items():
            ms = [x * 1e3 if isinstance(x, float) else x[0].elapsed_time(x[1]) for x in lst if x is not None]
            out[k] = (len(lst), float(sum(ms)))
        return out

    def all_reduce_sum(self, t):
        if self._peer_takes(t, t.numel()):
            with self._Span(self, 'peer_reduce_small', t):
                self.peer.reduce(t)
            return t
        with self._Span(self, 'all_reduce_%s' % ('small' if t.numel() <= 4096 else 'bucket'), t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_sum_async(self, t):
        """Starts the all-reduce on the communication stream and returns a handle; wait() makes
        the current stream wait for it.  Lets the head-gradient bucket (75 % of the bytes, ready
        first) travel over xGMI while the hidden stack's backward still computes."""
        self._async_probe = t
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.bulk, async_op=True)

    def wait(self, work):
        with self._Span(self, 'wait_async_bucket', self._async_probe):
            work.wait()

    def all_gather(self, t):
        flat = t.contiguous().view(-1)
        out = torch.empty(self.world * flat.numel(), dtype=t.dtype, device=t.device)
        with self._Span(self, 'all_gather_small' if flat.numel() <= 4096 else 'all_gather', t):
            dist.all_gather_into_tensor(out, flat, group=self.group)
        return out.view((self.world,) + tuple(t.shape))

    def reduce_scatter_sum(self, inp, out):
        """out [len / world] = this rank's shard of the element-wise sum of inp over the ranks."""
        with self._Span(self, 'reduce_scatter', inp):
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group)
        return out

    def all_gather_into(self, out, shard, name='all_gather_params'):
        """out = concatenation of every rank's shard (shard may be a slice of out)."""
        src = shard.clone() if shard.data_ptr() >= out.data_ptr() and \
            shard.data_ptr() < out.data_ptr() + out.numel() * out.element_size() else shard
        if name == 'all_gather_small' and self._peer_takes(src, src.numel()) and out.is_contiguous():
            with self._Span(self, 'peer_gather_small', out):
                self.peer.gather(out, src)
            return out
        with self._Span(self, name, out):
            dist.all_gather_into_tensor(out, src, group=self.group)
        return out


def init_from_env(backend=None, force=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract) and
    binds this process to its GPU. Returns a communicator (SingleProcess if WORLD_SIZE <= 1).
    force (or DCA_AMD_DIST_FORCE=1): a real communicator with ONE rank as well -- the data-parallel step with its RCCL
    calls, streams and waits on a one-GPU box (tests/test_dist_gpu.py)."""
    from .engine import SingleProcess
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if force is None:
        force = os.environ.get('DCA_AMD_DIST_FORCE', '0') == '1'
    if world <= 1 and not force:
        return SingleProcess()
    if world <= 1:
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('MASTER_PORT', '29533')
    local_rank = int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))
    if backend is None:
        # DCA_AMD_DIST_BACKEND=gloo: functional multi-rank runs on a box with fewer GPUs than ranks
        backend = os.environ.get('DCA_AMD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        if backend == 'nccl' and local_rank >= ndev:
            raise RuntimeError('rank %d has no GPU (%d visible): RCCL needs one GPU per rank' % (local_rank, ndev))
        torch.cuda.set_device(local_rank % ndev)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if not dist.is_initialized():
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend=backend, **kw)
    return TorchDistComm()


def shard(n, world, rank):
    """Contiguous block partition of n rows: (start, count) of `rank`; sizes differ by <= 1."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def local_order(global_perm, start, count):
    """Order in which this rank visits its own rows: the subsequence of the global shuffled
    index array that falls into [start, start+count), re-based to local row numbers.  With one
    rank this is the reference's shuffled index_array itself."""
    gp = np.asarray(global_perm)
    m = (gp >= start) & (gp < start + count)
    return (gp[m] - start).astype(np.int32)
