"""ctypes face of the library's own RCCL collectives (snws_comm_*), for hosts that do not run torch.distributed.
The package's modules use torch.distributed (backend "nccl" = RCCL) when a process group exists; this class is the
same three exchanges for a host that only has the C ABI: rank 0 calls Comm.unique_id(), ships the 128 bytes to the
other ranks by any means, and every rank builds Comm(id, rank, world) on its own GPU."""
import ctypes

import torch

from . import _lib as L


class Comm:
    def __init__(self, unique_id, rank, world):
        if len(unique_id) != 128:
            raise ValueError("Comm: the unique id is 128 bytes")
        self._id = ctypes.create_string_buffer(bytes(unique_id), 128)
        h = ctypes.c_void_p()
        L.call("snws_comm_init", ctypes.cast(self._id, ctypes.c_void_p), int(rank), int(world), ctypes.byref(h))
        self._h, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        L.call("snws_comm_unique_id", ctypes.cast(buf, ctypes.c_void_p))
        return buf.raw

    def all_reduce(self, t, op="sum"):
        """In-place all-reduce of a contiguous float32 (sum) or int64 (sum / max) device tensor."""
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError("Comm.all_reduce: contiguous device tensor expected")
        st = torch.cuda.current_stream(t.device).cuda_stream
        if t.dtype == torch.float32 and op == "sum":
            L.call("snws_comm_allreduce_f32", self._h, t.data_ptr(), t.numel(), st)
        elif t.dtype == torch.int64 and op in ("sum", "max"):
            L.call("snws_comm_allreduce_i64", self._h, t.data_ptr(), t.numel(), 0 if op == "sum" else 1, st)
        else:
            raise TypeError("Comm.all_reduce: float32 sum or int64 sum/max")
        return t

    def close(self):
        if self._h is not None:
            L.call("snws_comm_destroy", self._h)
            self._h = None
