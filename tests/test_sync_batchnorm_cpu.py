"""Host logic of the modeling.sync_batchnorm package: the replica-thread rendez-vous (reference
modeling/sync_batchnorm/comm.py:8-94) and the DataParallel replication callbacks (replicate.py:17-48).  No GPU."""
import copy
import pickle
import threading

import pytest
import torch

from modeling.sync_batchnorm import comm, replicate
from modeling.sync_batchnorm import SynchronizedBatchNorm1d, SynchronizedBatchNorm2d, SynchronizedBatchNorm3d


def test_future_result_is_a_one_slot_mailbox():
    f = comm.FutureResult()
    got = []
    t = threading.Thread(target=lambda: got.append(f.get()))
    t.start()
    f.put(41)
    t.join(10)
    assert got == [41]
    f.put(1)
    with pytest.raises(AssertionError):       # comm.py:18 - the previous result has not been fetched
        f.put(2)
    assert f.get() == 1
    f.wait_taken()                            # returns at once on an empty slot


def _sum_callback(messages):
    """Master callback in the style of _SynchronizedBatchNorm._data_parallel_master (batchnorm.py:80-101): sees the
    identifier-ordered list with the master first, answers every participant."""
    assert messages[0][0] == 0 and [i for i, _ in messages] == sorted(i for i, _ in messages)
    total = sum(m for _, m in messages)
    return [(i, (total, i)) for i, _ in messages]


def test_sync_master_rounds_with_three_replica_threads():
    master = comm.SyncMaster(_sum_callback)
    pipes = [master.register_slave(i) for i in (1, 2, 3)]
    assert master.nr_slaves == 3 and [p.identifier for p in pipes] == [1, 2, 3]
    replies = {}

    def replica(pipe, rounds):
        for r in range(rounds):
            replies[(pipe.identifier, r)] = pipe.run_slave(10 * pipe.identifier + r)

    threads = [threading.Thread(target=replica, args=(p, 2)) for p in pipes]
    for t in threads:
        t.start()
    out = [master.run_master(100 + r) for r in range(2)]
    for t in threads:
        t.join(10)
        assert not t.is_alive()
    for r in range(2):
        total = 100 + r + sum(10 * i + r for i in (1, 2, 3))
        assert out[r] == (total, 0)
        for i in (1, 2, 3):
            assert replies[(i, r)] == (total, i)
    # the first registration after a round starts a new replica set (comm.py:63-68)
    p = master.register_slave(7)
    assert master.nr_slaves == 1 and p.identifier == 7


def test_sync_master_copies_carry_only_the_callback():
    master = comm.SyncMaster(_sum_callback)
    master.register_slave(1)
    for clone in (copy.deepcopy(master), pickle.loads(pickle.dumps(master))):
        assert clone.nr_slaves == 0
        assert clone.run_master(5) == (5, 0)          # no replicas registered: the master alone


def test_sync_master_rejects_a_callback_whose_first_reply_is_not_the_masters():
    master = comm.SyncMaster(lambda messages: [(1, None)])
    with pytest.raises(AssertionError):
        master.run_master(0)


class _Leaf(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.seen = []

    def __data_parallel_replicate__(self, ctx, copy_id):
        self.seen.append((ctx, copy_id))
        if copy_id == 0:
            ctx.master = self
        else:
            assert ctx.master is not None            # the master copy is always visited first


def _model():
    return torch.nn.Sequential(_Leaf(), torch.nn.ReLU(), torch.nn.Sequential(_Leaf()))


def test_replication_callbacks_share_one_context_per_submodule():
    clones = [_model() for _ in range(3)]
    replicate.execute_replication_callbacks(clones)
    leaves = [[m for m in c.modules() if isinstance(m, _Leaf)] for c in clones]
    for j in range(2):
        ctxs = [leaves[k][j].seen[0][0] for k in range(3)]
        assert all(isinstance(c, replicate.CallbackContext) for c in ctxs)
        assert ctxs[0] is ctxs[1] is ctxs[2] and ctxs[0].master is leaves[0][j]
        assert [leaves[k][j].seen[0][1] for k in range(3)] == [0, 1, 2]
    assert leaves[0][0].seen[0][0] is not leaves[0][1].seen[0][0]


def test_patch_replication_callback_wraps_replicate():
    made = []

    def fake_replicate(module, device_ids):
        made.append(list(device_ids))
        return [copy.deepcopy(module) for _ in device_ids]

    for build in (lambda m: torch.nn.DataParallel(m), lambda m: replicate.DataParallelWithCallback(m)):
        model = _model()
        dp = build(model)
        dp.replicate = fake_replicate if not isinstance(dp, replicate.DataParallelWithCallback) else dp.replicate
        if isinstance(dp, replicate.DataParallelWithCallback):
            # already patched at construction: put the fake underneath a fresh patch
            dp.replicate = fake_replicate
        replicate.patch_replication_callback(dp)
        clones = dp.replicate(model, [0, 1])
        assert len(clones) == 2 and made[-1] == [0, 1]
        for k, c in enumerate(clones):
            for leaf in (m for m in c.modules() if isinstance(m, _Leaf)):
                assert leaf.seen[-1][1] == k
    assert "replicate" in replicate.DataParallelWithCallback(_model()).__dict__     # patched per instance
    with pytest.raises(AssertionError):
        replicate.patch_replication_callback(torch.nn.Linear(2, 2))


@pytest.mark.parametrize("cls,shape,bad", [(SynchronizedBatchNorm1d, (4, 6), (4, 6, 2, 2)),
                                           (SynchronizedBatchNorm2d, (2, 6, 3, 3), (2, 6)),
                                           (SynchronizedBatchNorm3d, (2, 6, 2, 3, 3), (2, 6, 3, 3))])
def test_synchronized_batchnorm_shape_errors_and_no_cpu_path(cls, shape, bad):
    m = cls(6)
    assert list(m.state_dict().keys()) == ["weight", "bias", "running_mean", "running_var", "num_batches_tracked"]
    with pytest.raises(ValueError):                   # batchnorm.py:127-131 and siblings
        m(torch.zeros(bad))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(shape))
