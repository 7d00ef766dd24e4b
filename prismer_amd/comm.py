"""ctypes binding of libprismer_comm.so (include/prismer_comm.h): the library's own RCCL communicator for the gradient
buckets.  Used by the Trainer with `transport='native'`; the default transport is torch.distributed (backend 'nccl' = the
same RCCL).  The 128-byte rendezvous token is shipped through the torch.distributed process group the launcher set up."""
import ctypes as C
import os

import torch

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.COMM_LIB
        known_bad, err = _build.comm_failed_before()            # (a marker of another comm.cpp / compiler does not count: retry)
        if not os.path.isfile(path) and not known_bad:
            _build.build_comm()                                 # comm.cpp only: never touches the compute library a process has loaded
            known_bad, err = _build.comm_failed_before()
        if not os.path.isfile(path):
            raise RuntimeError('libprismer_comm.so is not available on this machine (built without <rccl/rccl.h>?); '
                               "use transport='torch.distributed'.  Compiler output: " + (err[-600:] or 'none recorded'))
        L = C.CDLL(path)
        L.ph_comm_last_error.restype = C.c_char_p
        L.ph_comm_unique_id.argtypes = [C.c_void_p]
        L.ph_comm_init.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.ph_allreduce_bucket.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ph_reduce_scatter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ph_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ph_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.ph_comm_world.argtypes = [C.c_void_p]
        L.ph_comm_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


EXPORTS = ['ph_comm_unique_id', 'ph_comm_init', 'ph_allreduce_bucket', 'ph_reduce_scatter', 'ph_all_gather', 'ph_broadcast', 'ph_comm_world', 'ph_comm_destroy',
           'ph_comm_last_error']
F32, BF16 = 0, 1


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what}: {lib().ph_comm_last_error().decode()} (code {rc})')


class NativeComm:
    def __init__(self, rank, world, unique_id: bytes):
        assert len(unique_id) == 128
        self.rank, self.world = rank, world
        h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, 128)
        _check(lib().ph_comm_init(rank, world, buf, C.byref(h)), 'ph_comm_init')
        self.handle = h

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(lib().ph_comm_unique_id(buf), 'ph_comm_unique_id')
        return buf.raw

    @classmethod
    def from_process_group(cls, group, device):
        """rank 0 mints the token; it travels as a uint8 tensor over the existing process group (any backend)"""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        backend = dist.get_backend(group)
        dev = device if backend == 'nccl' else torch.device('cpu')
        t = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8))
        dist.broadcast(t, 0, group=group)
        torch.cuda.set_device(device)
        return cls(rank, world, bytes(t.cpu().tolist()))

    def all_reduce_(self, t):
        """in-place SUM of a contiguous fp32 / bf16 device tensor, on torch's current stream"""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        _check(lib().ph_allreduce_bucket(self.handle, t.data_ptr(), t.numel(), BF16 if t.dtype == torch.bfloat16 else F32,
                                         torch.cuda.current_stream().cuda_stream), 'ph_allreduce_bucket')
        return t

    @staticmethod
    def _dt(t):
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        return BF16 if t.dtype == torch.bfloat16 else F32

    def reduce_scatter(self, out, inp):
        """out[c] = SUM over ranks of chunk `rank` of inp[world * c] (torch.distributed.reduce_scatter_tensor), on torch's current stream"""
        assert inp.numel() == self.world * out.numel() and inp.dtype == out.dtype
        _check(lib().ph_reduce_scatter(self.handle, inp.data_ptr(), out.data_ptr(), out.numel(), self._dt(out), torch.cuda.current_stream().cuda_stream),
               'ph_reduce_scatter')
        return out

    def all_gather(self, out, inp):
        """out[world * c] = the ranks' inp[c] in rank order (torch.distributed.all_gather_into_tensor)"""
        assert out.numel() == self.world * inp.numel() and inp.dtype == out.dtype
        _check(lib().ph_all_gather(self.handle, inp.data_ptr(), out.data_ptr(), inp.numel(), self._dt(inp), torch.cuda.current_stream().cuda_stream),
               'ph_all_gather')
        return out

    def broadcast_(self, t, root):
        _check(lib().ph_broadcast(self.handle, t.data_ptr(), t.numel(), self._dt(t), int(root), torch.cuda.current_stream().cuda_stream), 'ph_broadcast')
        return t

    def destroy(self):
        if self.handle:
            _check(lib().ph_comm_destroy(self.handle), 'ph_comm_destroy')
            self.handle = None
