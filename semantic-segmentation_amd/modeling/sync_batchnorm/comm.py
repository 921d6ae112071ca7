"""Replica-thread rendez-vous with the public names of the reference's modeling/sync_batchnorm/comm.py:8-94
(FutureResult, SlavePipe, SyncMaster).

One-process-per-GPU training never needs it - there the all-reduce IS the rendez-vous (snws.ops, harness.GradBuckets) -
but code written against the replica-thread protocol keeps working: per round the master gathers one message from
every registered replica, runs its callback on the identifier-ordered list [(0, master message), (id, message), ...]
and hands each participant the reply the callback addressed to it.  Built on one mailbox type: a slot that is
filled once, emptied once, and lets the filler wait until it has been emptied (which is what keeps two rounds apart).
"""
import threading

__all__ = ['FutureResult', 'SlavePipe', 'SyncMaster']


class FutureResult(object):
    """One-slot mailbox.  put() requires an empty slot; get() blocks for the value and empties the slot;
    wait_taken() blocks until the last value put has been fetched."""

    def __init__(self):
        self._lock = threading.Condition()
        self._box = []

    def put(self, result):
        with self._lock:
            if self._box:
                raise AssertionError("FutureResult.put: the previous result has not been fetched")
            self._box.append(result)
            self._lock.notify_all()

    def get(self):
        with self._lock:
            self._lock.wait_for(lambda: self._box)
            value = self._box.pop()
            self._lock.notify_all()
            return value

    def wait_taken(self):
        with self._lock:
            self._lock.wait_for(lambda: not self._box)


class SlavePipe(object):
    """A replica's handle on its master: run_slave(message) -> the reply addressed to this replica."""

    def __init__(self, master, identifier):
        self.identifier = identifier
        self._master = master
        self.result = FutureResult()

    def run_slave(self, msg):
        self._master._deliver(self.identifier, msg)
        return self.result.get()


class SyncMaster(object):
    """master_callback(messages) -> [(identifier, reply), ...] with the master's own entry (identifier 0) first."""

    def __init__(self, master_callback):
        self._master_callback = master_callback
        self._pipes = {}
        self._inbox = {}
        self._arrived = threading.Condition()
        self._in_round = False

    def __getstate__(self):          # replicas are created by copying modules: only the callback travels
        return {'master_callback': self._master_callback}

    def __setstate__(self, state):
        self.__init__(state['master_callback'])

    @property
    def nr_slaves(self):
        return len(self._pipes)

    def register_slave(self, identifier):
        with self._arrived:
            if self._in_round:       # first registration after a round: a new set of replicas
                if self._inbox:
                    raise AssertionError("SyncMaster: messages left over from the previous round")
                self._pipes.clear()
                self._in_round = False
            pipe = self._pipes[identifier] = SlavePipe(self, identifier)
        return pipe

    def _deliver(self, identifier, msg):
        with self._arrived:
            self._inbox[identifier] = msg
            self._arrived.notify_all()

    def run_master(self, master_msg):
        with self._arrived:
            self._in_round = True
            self._arrived.wait_for(lambda: len(self._inbox) == len(self._pipes))
            messages = [(0, master_msg)] + sorted(self._inbox.items(), key=lambda kv: kv[0])
            self._inbox.clear()
        replies = self._master_callback(messages)
        if replies[0][0] != 0:
            raise AssertionError("SyncMaster: the callback's first reply must be the master's")
        for identifier, reply in replies[1:]:
            self._pipes[identifier].result.put(reply)
        for identifier, _ in replies[1:]:
            self._pipes[identifier].result.wait_taken()
        return replies[0][1]
