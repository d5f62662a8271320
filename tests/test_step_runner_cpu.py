"""The fit loop's step runner (dca_amd/train.py::_StepRunner) without a GPU: which steps run eagerly, which are captured,
which graph is replayed -- with a fake engine and a fake capture that records what a replay would execute."""
import pytest

from dca_amd import train as T


class FakeOps:
    device_type = 'cuda'


class FakeComm:
    def __init__(self, dp, capturable=True):
        self.dp, self.capturable, self.world, self.rank = dp, capturable, 2 if dp else 1, 0


class FakeEngine:
    def __init__(self, dp=False, capturable=True):
        self.ops, self.comm = FakeOps(), FakeComm(dp, capturable)
        self.log = []                # ('step', b, b_global, counts) / ('counts', counts)
        self.capturing = False

    def train_step(self, b, b_global, world_counts, rows_per_slot):
        self.log.append(('captured' if self.capturing else 'step', b, b_global, tuple(world_counts)))

    def set_world_counts(self, world_counts):
        self.log.append(('counts', tuple(world_counts)))


class FakeGraph:
    def __init__(self, eng, steps):
        self.eng, self.steps, self.replays = eng, steps, 0

    def replay(self):
        self.replays += 1
        self.eng.log.extend(('replayed',) + s[1:] for s in self.steps)


def make_runner(eng, fail_on=None):
    r = T._StepRunner(eng, True)
    captures = []

    def capture(args, k):
        if fail_on is not None and len(captures) == fail_on:
            captures.append(None)
            raise RuntimeError('capture refused')
        eng.capturing = True
        n0 = len(eng.log)
        for _ in range(k):
            eng.train_step(*args)
        eng.capturing = False
        steps = eng.log[n0:]
        del eng.log[n0:]                                   # a capture executes nothing
        g = FakeGraph(eng, steps)
        captures.append(g)
        return g
    r._capture = capture
    return r, captures


def executed(eng):
    return [e for e in eng.log if e[0] in ('step', 'replayed')]


def test_single_gpu_first_step_eager_then_graphs_of_eight_and_one():
    eng = FakeEngine()
    r, caps = make_runner(eng)
    r.run(32, 32, [32], 32, n=21)                          # 1 eager + 2 x 8 + 4 x 1
    assert [e[0] for e in executed(eng)] == ['step'] + ['replayed'] * 20
    assert [len(g.steps) for g in caps] == [8, 1] and [g.replays for g in caps] == [2, 4]
    r.run(32, 32, [32], 32, n=8)                           # the graphs are kept
    assert len(caps) == 2 and caps[0].replays == 3
    r.run(7, 7, [7], 32, n=1)                              # a new batch size: eager, nothing captured
    assert executed(eng)[-1] == ('step', 7, 7, (7,)) and len(caps) == 2
    r.run(7, 7, [7], 32, n=1)                              # its second visit: captured and replayed
    assert len(caps) == 3 and caps[2].replays == 1
    assert not any(e[0] == 'counts' for e in eng.log)      # no communicator: no rank counts


def test_data_parallel_keys_hold_the_rank_counts_and_counts_are_set_before_every_replay():
    eng = FakeEngine(dp=True)
    r, caps = make_runner(eng)
    r.run(5, 10, [5, 5], 5, n=10)                          # 1 eager + 8 + 1
    r.run(5, 9, [5, 4], 5, n=2)                            # same local batch, other peer count: a shape of its own
    assert [len(g.steps) for g in caps] == [8, 1, 1]
    assert caps[0].steps[0] == ('captured', 5, 10, (5, 5)) and caps[2].steps[0] == ('captured', 5, 9, (5, 4))
    ex = executed(eng)
    assert [e[0] for e in ex] == ['step'] + ['replayed'] * 9 + ['step', 'replayed']
    assert ex[10][2:] == (9, (5, 4)) and ex[11][2:] == (9, (5, 4))
    # every run() sets the counts first (host -> device copy outside the graphs)
    first = [i for i, e in enumerate(eng.log) if e[0] == 'counts']
    assert eng.log[first[0]] == ('counts', (5, 5)) and eng.log[first[1]] == ('counts', (5, 4)) and len(first) == 2
    assert first[0] == 0 and eng.log[first[1] + 1][0] == 'step'
    r.run(0, 4, [0, 4], 5, n=3)                            # an exhausted shard takes part eagerly
    assert [e[:2] for e in executed(eng)[-3:]] == [('step', 0)] * 3 and len(caps) == 3


def test_data_parallel_without_a_capturable_backend_or_with_the_switch_off_stays_eager(monkeypatch):
    eng = FakeEngine(dp=True, capturable=False)            # gloo
    r, caps = make_runner(eng)
    r.run(4, 8, [4, 4], 4, n=20)
    assert not caps and [e[0] for e in executed(eng)] == ['step'] * 20
    monkeypatch.setenv('DCA_AMD_DP_GRAPH', '0')
    eng = FakeEngine(dp=True)
    r, caps = make_runner(eng)
    r.run(4, 8, [4, 4], 4, n=20)
    assert not caps and len(executed(eng)) == 20


def test_a_refused_capture_leaves_the_data_parallel_run_eager_and_complete(capsys):
    eng = FakeEngine(dp=True)
    r, caps = make_runner(eng, fail_on=0)
    r.run(4, 8, [4, 4], 4, n=12)                           # 1 eager, capture of 8 refused -> 11 eager
    assert [e[0] for e in executed(eng)] == ['step'] * 12 and r.use_graph is False
    assert 'capture of the data-parallel step failed' in capsys.readouterr().err
    r.run(4, 8, [4, 4], 4, n=9)
    assert len(executed(eng)) == 21 and caps == [None]
    # without a communicator a refused capture is an error of the caller's setup and is raised
    eng = FakeEngine()
    r, caps = make_runner(eng, fail_on=0)
    with pytest.raises(RuntimeError):
        r.run(32, 32, [32], 32, n=12)


def test_captures_are_thread_local(monkeypatch):
    """_StepRunner._capture itself, with torch's graph objects replaced: the capture must run in thread-local error mode
    (RCCL's watchdog thread polls the events of earlier collectives; under the global mode that call from another thread
    invalidates a capture in progress), on a side stream that waits for / is waited for by the current one."""
    import contextlib
    import torch
    seen = {}

    class G:
        pass

    class S:
        def wait_stream(self, other):
            seen.setdefault('waits', []).append('side')

    class Cur:
        def wait_stream(self, other):
            seen.setdefault('waits', []).append('current')

    @contextlib.contextmanager
    def fake_graph(g, stream=None, capture_error_mode='global', **kw):
        seen['mode'], seen['stream'] = capture_error_mode, stream
        yield

    @contextlib.contextmanager
    def fake_stream(s):
        yield

    monkeypatch.setattr(torch.cuda, 'CUDAGraph', G)
    monkeypatch.setattr(torch.cuda, 'Stream', S)
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda: Cur())
    monkeypatch.setattr(torch.cuda, 'graph', fake_graph)
    monkeypatch.setattr(torch.cuda, 'stream', fake_stream)
    eng = FakeEngine(dp=True)
    r = T._StepRunner(eng, True)
    g = r._capture((32, 64, [32, 32], 32), 3)
    assert isinstance(g, G) and seen['mode'] == 'thread_local' and isinstance(seen['stream'], S)
    assert seen['waits'] == ['side', 'current'] and [e[0] for e in eng.log] == ['step'] * 3
