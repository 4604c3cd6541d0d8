"""A communicator of this library's own on RCCL, driven through its C API on the CALLER's stream.

Why: torch.nn.SyncBatchNorm (train_SROIE.py:202-203 `convert_sync_batchnorm`) needs one all-reduce of 2C + 1 doubles in front of every
BatchNorm's normalisation and one of 2C doubles inside its backward -- 80 tiny, blocking collectives per step at resnet-34.  Through
torch.distributed each of them costs ~74 us on an MI355X (measured with a one-rank process group, DESIGN.md section 6: 5.9 ms per 34 ms
step): ProcessGroupNCCL runs collectives on a stream of its own, so every call is an event hand-over from the compute stream, the
collective's kernel, and an event hand-over back, plus the dispatcher.  `ncclAllReduce` enqueued directly on the compute stream is one
kernel in stream order: nothing to hand over, and the statistics no longer queue behind the 32 MB gradient buckets that
ProcessGroupNCCL's stream is busy with during backward.

The communicator is built once per process (rank 0 draws the unique id, torch.distributed's default group carries it to the others) and
only ever used from the stream the model runs on; the gradient buckets stay on torch.distributed's communicator (vbg.optim.FlatReducer).
Two communicators have kernels in flight at the same time during backward -- both are a handful of workgroups on a 256-CU device and
co-reside; the program order of the statistics collectives is the model's own forward / backward order, identical on every rank.

`backend="nccl"` of torch.distributed IS this library on ROCm (the same librccl.so torch loaded); nothing here touches CUDA / NCCL
proper."""
import ctypes as C
import os

import torch
import torch.distributed as dist

_NCCL_SUM = 0
_DT = {torch.float64: 8, torch.float32: 7, torch.int32: 2, torch.int64: 4}     # ncclDataType_t (nccl.h / rccl.h)


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_LIB = [None]


def _lib():
    if _LIB[0] is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = C.CDLL(path)             # (already mapped by torch: the same library instance)
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        _LIB[0] = lib
    return _LIB[0]


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_lib().ncclGetErrorString(rc).decode()} ({rc})")


class DirectComm:
    """ncclComm_t over the ranks of torch.distributed's default group, one per process; collectives are enqueued on the stream the caller
    is on (`torch.cuda.current_stream`)."""

    def __init__(self, device):
        assert dist.is_initialized()
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        lib = _lib()
        uid = _UniqueId()
        box = [None]
        if self.rank == 0:
            # (a failure on rank 0 still reaches the broadcast below -- as None -- so that nobody waits for an id that never comes)
            if lib.ncclGetUniqueId(C.byref(uid)) == 0:
                box = [C.string_at(C.addressof(uid), 128)]          # (raw bytes: .value would stop at the first NUL)
        dist.broadcast_object_list(box, src=0)
        self.comm = C.c_void_p()
        if box[0] is None:
            raise RuntimeError("ncclGetUniqueId failed on rank 0")
        C.memmove(C.addressof(uid), box[0], 128)
        with torch.cuda.device(self.device):
            _check(lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.calls = 0

    def all_reduce_(self, t: torch.Tensor):
        """in-place sum over the ranks, in stream order on the current stream of t's device"""
        assert t.is_cuda and t.is_contiguous() and t.dtype in _DT
        stream = torch.cuda.current_stream(t.device).cuda_stream
        _check(_lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DT[t.dtype], _NCCL_SUM, self.comm, C.c_void_p(stream)), "ncclAllReduce")
        self.calls += 1
        return t

    def nranks(self) -> int:
        """ranks of the communicator as RCCL itself counts them (ncclCommCount)"""
        n = C.c_int(0)
        lib = _lib()
        lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _check(lib.ncclCommCount(self.comm, C.byref(n)), "ncclCommCount")
        return int(n.value)

    def destroy(self):
        if self.comm:
            _lib().ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


def available() -> bool:
    """the default group runs on RCCL (backend "nccl"): a communicator of this library's own can be built beside it"""
    try:
        return dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl" and torch.cuda.is_available()
    except Exception:
        return False
