"""Thread rendez-vous primitives with the reference's names (modeling/sync_batchnorm/comm.py:8-94: FutureResult,
SlavePipe, SyncMaster).  One-process-per-GPU training does not need them - the all-reduce is the rendez-vous - but
code written against the replica-thread protocol keeps working: a master collects one message per registered slave,
runs its callback on the identifier-sorted list and hands every participant its own reply."""
import collections
import queue
import threading

__all__ = ['FutureResult', 'SlavePipe', 'SyncMaster']


class FutureResult(object):
    """A one-slot mailbox: put() once, get() blocks until the value is there and empties the slot."""

    def __init__(self):
        self._cond = threading.Condition()
        self._value = None
        self._full = False

    def put(self, result):
        with self._cond:
            assert not self._full, 'Previous result has\'t been fetched.'
            self._value, self._full = result, True
            self._cond.notify()

    def get(self):
        with self._cond:
            while not self._full:
                self._cond.wait()
            value, self._value, self._full = self._value, None, False
            return value


_MasterRegistry = collections.namedtuple('MasterRegistry', ['result'])


class SlavePipe(collections.namedtuple('_SlavePipeBase', ['identifier', 'queue', 'result'])):
    """The slave's end: send a message to the master, wait for the reply, acknowledge."""

    def run_slave(self, msg):
        self.queue.put((self.identifier, msg))
        reply = self.result.get()
        self.queue.put(True)
        return reply


class SyncMaster(object):
    """The master's end.  master_callback(list of (identifier, message)) -> list of (identifier, reply), with
    identifier 0 being the master itself."""

    def __init__(self, master_callback):
        self._master_callback = master_callback
        self._queue = queue.Queue()
        self._registry = collections.OrderedDict()
        self._activated = False

    def __getstate__(self):
        return {'master_callback': self._master_callback}

    def __setstate__(self, state):
        self.__init__(state['master_callback'])

    def register_slave(self, identifier):
        if self._activated:     # a new replication round starts: forget the previous replicas
            assert self._queue.empty(), 'Queue is not clean before next initialization.'
            self._activated = False
            self._registry.clear()
        future = FutureResult()
        self._registry[identifier] = _MasterRegistry(future)
        return SlavePipe(identifier, self._queue, future)

    def run_master(self, master_msg):
        self._activated = True
        messages = [(0, master_msg)] + [self._queue.get() for _ in range(self.nr_slaves)]
        replies = self._master_callback(messages)
        assert replies[0][0] == 0, 'The first result should belongs to the master.'
        for ident, reply in replies:
            if ident != 0:
                self._registry[ident].result.put(reply)
        for _ in range(self.nr_slaves):
            assert self._queue.get() is True
        return replies[0][1]

    @property
    def nr_slaves(self):
        return len(self._registry)
