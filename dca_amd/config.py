"""Every knob of the MI355X path in ONE place.

The product runs on the defaults.  The fields below exist because a parity test or a measurement tool compares two
forms of the same step; each names the environment variable that sets it (read once, here: `EngineConfig.from_env()`),
what it selects, and the test that holds the non-default form to the default one.  Nothing else in the package reads
the environment for a kernel / path decision (the torch.distributed rendezvous variables and DCA_AMD_DIST_* -- which
communicator to build -- are dist.py's).

Tested combinations (tests/, all against the same oracle numbers):
  default                                   every -m gpu test
  stack = 'coop' | 'off'                    test_engine_gpu.py::test_fused_hidden_stack_equals_the_per_operation_kernels
  bwd_chain = False                         test_engine_gpu.py::test_single_workgroup_backward_chain_equals_the_per_layer_kernels
  wide_planes = False                       test_engine_gpu.py::test_wide_network_step[planes=False]
  wide_h2 = False                           test_engine_gpu.py::test_wide_network_step (both plane arithmetics against the oracle)
  dp_sharded_opt = True                     test_dp_gloo.py::test_sharded_optimizer_equals_the_all_reduce_path,
                                            test_dp_gloo.py::test_checkpoints_of_the_sharded_optimizer_...,
                                            test_dp_gpu.py::test_rccl_communicator_with_one_rank_...[sharded]
  dp_graph = False                          test_dp_gpu.py::test_rccl_communicator_with_one_rank_... (eager == captured, bit for bit),
                                            test_step_runner_cpu.py
  device_prep = False                       test_prep_gpu.py (host path == K-PREP)
  fused_write = False                       test_api_gpu.py::test_fused_predict_writer_writes_the_files_of_predict_then_write
  dp_peer_exchange = True                   test_dp_gpu.py::test_peer_exchange_two_processes_on_one_gpu (raw exchanges, then a fit == the
                                            library-collective fit, bit for bit)
Engine attributes a test sets directly instead (no knob): use_fused (K-HEADS vs separate kernels:
test_fused_and_separate_heads_agree_stepwise, test_full_size_step_fused_equals_separate).
"""
import os
from dataclasses import dataclass, fields


def _flag(name, default):
    v = os.environ.get(name)
    return default if v is None else v not in ('0', '', 'off', 'false', 'False')


@dataclass
class EngineConfig:
    # K-STACK, the hidden stack at throughput batches: 'steps' = one launch per batch-wide dependency (9 launches for
    # 64-32-64), 'coop' = one cooperative launch per direction (grid barriers), 'off' = one launch per operation (22)
    stack: str = 'steps'                    # DCA_AMD_STACK
    # the hidden stack's backward as ONE single-workgroup launch at batches of at most 64 rows (the reference default 32)
    bwd_chain: bool = True                  # DCA_AMD_BWD_CHAIN
    # decoders wider than 64 units: every large product from pre-split bf16 planes (False: transposed fp32 operand copies)
    wide_planes: bool = True                # DCA_AMD_WIDE_PLANES
    # ... and those planes as TWO fp16 pieces with three products per fp32 product (half the matrix instructions; False: three
    # bf16 pieces, six products)
    wide_h2: bool = True                    # DCA_AMD_WIDE_H2
    # data parallel: reduce-scatter -> per-rank clip + RMSprop on its shard -> all-gather (instead of two all-reduce buckets)
    dp_sharded_opt: bool = False            # DCA_AMD_DP_SHARDED_OPT
    # data parallel: capture the steps (RCCL exchanges included) into hipGraphs
    dp_graph: bool = True                   # DCA_AMD_DP_GRAPH
    # data parallel: the SyncBN exchanges (<= 2 h floats each) as one kernel launch over IPC-mapped peer buffers (K-PEER)
    # instead of RCCL calls; off until a multi-GPU node has measured it (functionally proven with two processes on one GPU)
    dp_peer_exchange: bool = False          # DCA_AMD_DP_PEER
    # io.normalize on the GPU (K-PREP) when one is present
    device_prep: bool = True                # DCA_AMD_DEVICE_PREP
    # the command line's predict + write as one streaming pass (gene x cell blocks formatted while the next one computes)
    fused_write: bool = True                # DCA_AMD_FUSED_WRITE
    # ---- measured constants (no environment variable; DESIGN.md holds the measurements)
    graph_steps: int = 8                    # consecutive training steps per hipGraph launch (fit loop and bench)
    sparse_dw_min: int = 512                # batch rows from which the first layer's weight gradient reads the byte store
    lut_fwd_min: int = 1024                 # ... and its forward product (looked-up operand on the matrix pipe; 32 / 64 units)
    enc0_nt_min: int = 256                  # batch rows from which the first product runs in the NT form on a transposed W0
    predict_chunk: int = 1024               # rows per device -> host chunk of predict()

    _ENV = {'stack': 'DCA_AMD_STACK', 'bwd_chain': 'DCA_AMD_BWD_CHAIN', 'wide_planes': 'DCA_AMD_WIDE_PLANES', 'wide_h2': 'DCA_AMD_WIDE_H2',
            'dp_sharded_opt': 'DCA_AMD_DP_SHARDED_OPT', 'dp_graph': 'DCA_AMD_DP_GRAPH', 'device_prep': 'DCA_AMD_DEVICE_PREP',
            'fused_write': 'DCA_AMD_FUSED_WRITE', 'dp_peer_exchange': 'DCA_AMD_DP_PEER'}

    @classmethod
    def from_env(cls):
        c = cls()
        for f in fields(cls):
            env = cls._ENV.get(f.name)
            if env is None:
                continue
            if f.type is bool or isinstance(getattr(c, f.name), bool):
                setattr(c, f.name, _flag(env, getattr(c, f.name)))
            elif os.environ.get(env) is not None:
                setattr(c, f.name, type(getattr(c, f.name))(os.environ[env]))
        if c.stack not in ('steps', 'coop', 'off'):
            raise ValueError('DCA_AMD_STACK must be steps, coop or off (got %r)' % c.stack)
        return c


def current():
    """The configuration of this process (the environment is read at every call: tests switch forms between engines)."""
    return EngineConfig.from_env()
