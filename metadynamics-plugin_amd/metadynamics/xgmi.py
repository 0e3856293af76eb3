"""xGMI mailbox (``mtd_comm_*`` of include/mtd_abi.h): set-up over a torch.distributed process group.

The mailbox carries the few doubles a particle-sharded bias step exchanges by direct stores between
the GPUs of one node; torch.distributed is only the control plane here (it gathers the 64-byte IPC handles once and
lets the ranks agree on the outcome of the self test).  If any rank cannot map a peer or the self test fails, every
rank gets ``None`` back and the caller keeps the RCCL all-reduce — never a CPU path.
"""
import ctypes as C
import os
import socket


class Mailbox:
    """one rank's end of the mailbox; ``handle`` is the opaque ``mtd_comm*``"""

    def __init__(self, handle, rank, world):
        from . import _abi
        self._abi, self.lib = _abi, _abi.load()
        self.handle, self.rank, self.world = handle, rank, world

    def all_reduce(self, tensor):
        """in-place sum over ranks of a float64 device tensor of at most ``max_doubles`` elements, on the current stream's
        device queue (the null stream, like every call of this library made from Python)"""
        self._abi.check(self.lib.mtd_comm_allreduce_small(self.handle, tensor.data_ptr(), int(tensor.numel()), None))
        return tensor

    def barrier(self):
        """all ranks' earlier work on the null stream is complete (and visible in their shared buffers) before anything
        enqueued after this call runs: a one-double exchange"""
        if getattr(self, "_token", None) is None:
            import torch
            self._token = torch.zeros(1, dtype=torch.float64, device="cuda")
        self.all_reduce(self._token)

    def share(self, dist, nbytes):
        """a buffer of ``nbytes`` on every rank that all ranks can read: returns (local address, [address of rank r's
        buffer as mapped here]); collective over ``dist``"""
        import torch
        dev = _control_device(dist)
        local, slot, h = C.c_void_p(), C.c_uint(), (C.c_ubyte * 64)()
        self._abi.check(self.lib.mtd_comm_share(self.handle, int(nbytes), C.byref(local), C.byref(slot), h))
        handles = torch.empty(self.world * 64, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(handles, torch.frombuffer(bytearray(bytes(h)), dtype=torch.uint8).to(dev))
        peers = (C.c_void_p * self.world)()
        self._abi.check(self.lib.mtd_comm_open(self.handle, slot.value, handles.cpu().numpy().tobytes(), peers))
        return local.value, [peers[r] for r in range(self.world)]

    def timeouts(self):
        n = C.c_uint()
        self._abi.check(self.lib.mtd_comm_status(self.handle, C.byref(n), None))
        return n.value

    def close(self):
        if self.handle:
            self._abi.check(self.lib.mtd_comm_destroy(self.handle))
            self.handle = None


def attach(dist, exec_conf, mailbox, communicator=None):
    """Make ``exec_conf`` (``_metadynamics.ExecutionConfiguration``) the execution configuration of a DOMAIN-DECOMPOSED run: the
    mailbox plays the role of HOOMD's MPI communicator for the per-step sums of every collective variable, ``dist`` is the control
    plane the host classes use at set-up (an all-gather of IPC handles: the mesh CV's exported buffers), ``communicator`` an
    optional ``mtd_rccl`` handle (``sharded.RcclAllReduce(dist).handle``) for the large per-step buffers — without it they go
    through the mailbox's exported buffers (``mtd_comm_allreduce_pull``)."""
    import torch
    dev = _control_device(dist)
    world = dist.get_world_size()

    def allgather(blob):
        out = torch.empty(world * len(blob), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(out, torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev))
        return out.cpu().numpy().tobytes()

    exec_conf.setMailbox(mailbox.handle.value)
    exec_conf.setAllgather(allgather)
    if communicator is not None:
        exec_conf.setCommunicator(communicator.value if hasattr(communicator, "value") else int(communicator))


_last_failure = None


def last_failure():
    """why the last :func:`connect` of this process handed back ``None`` (a short sentence), or ``None``"""
    return _last_failure


def _control_device(dist):
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def connect(dist, max_doubles=64, self_test=True):
    """Create this rank's mailbox, exchange the IPC handles over ``dist``, map the peers and run a self test.
    Returns a :class:`Mailbox`, or ``None`` on every rank when the mailbox cannot be used (more than 8 ranks, ranks on
    different hosts, a peer that cannot be mapped, a failed self test, or MTD_XGMI_MAILBOX=0)."""
    import torch
    from . import _abi
    global _last_failure
    _last_failure = None
    lib = _abi.load()
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _control_device(dist)

    def agree(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    usable = os.environ.get("MTD_XGMI_MAILBOX", "1") != "0" and world <= 8
    # one node only: IPC handles mean nothing on another host
    host = socket.gethostname().encode()[:63].ljust(64, b"\0")
    names = torch.empty(world * 64, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(names, torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev))
    names = bytes(names.cpu().numpy().tobytes())
    usable = usable and all(names[64 * r:64 * r + 64] == host for r in range(world))
    if not agree(usable):
        _last_failure = ("switched off (MTD_XGMI_MAILBOX=0)" if os.environ.get("MTD_XGMI_MAILBOX", "1") == "0"
                         else "more than 8 ranks or ranks on different hosts")
        return None

    h = C.c_void_p()
    rc = lib.mtd_comm_create(C.byref(h), rank, world, int(max_doubles))
    mine = (C.c_ubyte * 64)()
    if rc == 0 and world > 1:
        rc = lib.mtd_comm_handle(h, mine)
    handles = torch.empty(world * 64, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(handles, torch.frombuffer(bytearray(bytes(mine)), dtype=torch.uint8).to(dev))
    if rc == 0 and world > 1:
        blob = handles.cpu().numpy().tobytes()
        rc = lib.mtd_comm_connect(h, blob)
    if os.environ.get("MTD_XGMI_TEST_FAIL_RANK") == str(rank):
        rc = -3                                 # test hook: this rank pretends it could not map its peers
    ok = agree(rc == 0)
    box = Mailbox(h, rank, world) if h else None
    if ok and self_test:
        ok = agree(_self_test(box, torch))
    if not ok:
        # HIP IPC needs HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts with dmabuf-only IPC, set before the runtime initialises
        # (metadynamics._abi.load does when nobody chose a value; a process that initialised HIP earlier has to export it)
        _last_failure = ("a rank could not export or map a peer's mailbox (status %d on rank %d; HSA_ENABLE_IPC_MODE_LEGACY=%s)"
                         % (rc, rank, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))) if rc != 0 or box is None else "the self test failed"
        # every rank leaves the same way (the barrier is collective: ranks whose mailbox never came up take part too)
        torch.cuda.synchronize()
        dist.barrier()
        if box is not None:
            box.close()
        return None
    return box


def _self_test(box, torch):
    """three exchanges of known values (both parities of the slot buffers); bounded by the mailbox's own timeout"""
    try:
        for it in range(3):
            v = torch.tensor([box.rank + 1.0 + it, -2.5 * (box.rank + 1), 1e-3 * it], dtype=torch.float64, device="cuda")
            box.all_reduce(v)
            torch.cuda.synchronize()
            w = box.world
            tri = w * (w + 1) / 2.0
            want = [tri + it * w, -2.5 * tri, 1e-3 * it * w]
            got = v.cpu().tolist()
            if box.timeouts() != 0 or any(abs(a - b) > 1e-12 * max(1.0, abs(b)) for a, b in zip(got, want)):
                return False
        return True
    except Exception:
        return False
