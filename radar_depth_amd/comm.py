"""Native data-parallel communicator: the RCCL collectives of the C ABI (include/radar_depth_hip.h, rd_comm_*) bootstrapped
from whatever rendezvous the launcher already has.

    comm.init_from_torch_distributed()     # rank 0 creates the RCCL token, torch.distributed (gloo or nccl) ships it
    comm.init_from_file(path, rank, world) # or through a shared file (no torch.distributed at all)

After that HipTrainStep averages gradients with rd_allreduce_bucket on its own communication stream, event-chained behind each
backward segment; torch.distributed's all_reduce stays available as the cross-check (HipTrainStep(comm="torch"))."""
import ctypes as C
import os
import time

from ._lib import check, lib

TOKEN_BYTES = 128
RD_DTYPE_F32, RD_DTYPE_BF16 = 0, 1


def world():
    return int(lib().rd_comm_world())


def rank():
    return int(lib().rd_comm_rank())


def initialised():
    return world() > 0


def _join(token, rank_, world_):
    buf = (C.c_ubyte * TOKEN_BYTES).from_buffer_copy(bytes(token))
    check(lib().rd_comm_init(buf, rank_, world_), "rd_comm_init")


def _new_token():
    buf = (C.c_ubyte * TOKEN_BYTES)()
    check(lib().rd_comm_unique_id(buf), "rd_comm_unique_id")
    return bytes(buf)


def init_from_torch_distributed():
    """torch.distributed must be initialised (any backend) and the device selected (torch.cuda.set_device)."""
    import torch
    import torch.distributed as dist
    assert dist.is_initialized(), "initialise torch.distributed first (it only carries the 128-byte RCCL token)"
    if initialised():
        return
    r, w = dist.get_rank(), dist.get_world_size()
    box = [_new_token() if r == 0 else None]
    dist.broadcast_object_list(box, src=0)
    _join(box[0], r, w)


def init_from_file(path, rank_, world_, timeout_s=120.0, job_id=None):
    """Rendezvous through a file on a shared filesystem: rank 0 writes the token atomically, the others poll for it.
    job_id (any string every rank of THIS run agrees on -- e.g. the launcher's job / run id) is written next to the token and
    compared on read, so a stale file of an earlier run at the same path is never accepted -- pass one whenever a path can be
    reused.  Without it the only protection is that rank 0 removes an existing file before writing and removes its own once every
    rank has joined (ncclCommInitRank returns when all ranks did), so a CLEANLY finished run leaves nothing behind; use a fresh
    path per run then."""
    if initialised():
        return
    tag = ("" if job_id is None else str(job_id)).encode()
    if rank_ == 0:
        if os.path.exists(path):
            if job_id is None:
                import warnings
                warnings.warn("radar_depth_amd.comm.init_from_file: %s already exists (a crashed earlier run?) and no job_id was given: a rank "
                              "that read the stale token before this one is replaced will hang in ncclCommInitRank -- pass job_id" % path)
            os.unlink(path)
        token = _new_token()
        with open(path + ".tmp", "wb") as f:
            f.write(token + tag)
        os.replace(path + ".tmp", path)
    else:
        t0 = time.time()
        while True:
            try:
                with open(path, "rb") as f:
                    blob = f.read()
                if len(blob) >= TOKEN_BYTES and blob[TOKEN_BYTES:] == tag:
                    token = blob[:TOKEN_BYTES]
                    break
            except OSError:
                pass
            if time.time() - t0 > timeout_s:
                raise TimeoutError("no RCCL token for this run at %s after %.0f s" % (path, timeout_s))
            time.sleep(0.01)
    _join(token, rank_, world_)
    if rank_ == 0:
        try:
            os.unlink(path)
        except OSError:
            pass


def allreduce_(tensor, stream, lo=0, hi=None):
    """In-place sum of tensor[lo:hi] (a flat fp32 / bf16 CUDA tensor) over all ranks, asynchronous on `stream` (c_void_p)."""
    import torch
    hi = tensor.numel() if hi is None else hi
    dt = {torch.float32: RD_DTYPE_F32, torch.bfloat16: RD_DTYPE_BF16}[tensor.dtype]
    check(lib().rd_allreduce_bucket(C.c_void_p(tensor.data_ptr() + lo * tensor.element_size()), C.c_int64(hi - lo), dt, stream),
          "rd_allreduce_bucket")


def broadcast_(tensor, stream, root=0):
    import torch
    dt = {torch.float32: RD_DTYPE_F32, torch.bfloat16: RD_DTYPE_BF16}.get(tensor.dtype)
    if dt is None:                       # e.g. the int64 num_batches_tracked counters: as raw fp32 words
        n_words = tensor.numel() * tensor.element_size() // 4
        check(lib().rd_broadcast(C.c_void_p(tensor.data_ptr()), C.c_int64(n_words), RD_DTYPE_F32, root, stream), "rd_broadcast")
        return
    check(lib().rd_broadcast(C.c_void_p(tensor.data_ptr()), C.c_int64(tensor.numel()), dt, root, stream), "rd_broadcast")


def destroy():
    check(lib().rd_comm_destroy(), "rd_comm_destroy")
