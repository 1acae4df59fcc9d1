"""Multi-GPU sharding of the fill hot path: sites (or seeds) are independent (Dispatcher.iterate, src/Filler.cpp:824,844), so
each rank takes a contiguous slice, the index is replicated, and only the results travel: byte payloads are gathered on one
rank with an all_gather of sizes followed by a padded gather (RCCL over xGMI with backend "nccl", gloo in the CPU tests)."""
import threading

import numpy as np


def shard_range(n_items, rank, world):
    """contiguous slice [lo, hi) of n_items for this rank; slices differ by at most one item"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_bytes(payload, dst=0, device=None):
    """payload: 1-D uint8 numpy array of this rank.  Returns the list of all ranks' payloads on rank dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    mine = torch.from_numpy(np.ascontiguousarray(payload, dtype=np.uint8)).to(dev)
    sz = torch.tensor([mine.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(sz) for _ in range(world)]
    dist.all_gather(sizes, sz)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=dev)
    pad[: mine.numel()] = mine
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:n].cpu().numpy() for b, n in zip(bufs, sizes)]


class PipelinedGather:
    """Per-step gather of a variable-length byte payload on rank dst that overlaps with the following steps: the payload is written
    into a page-locked staging buffer, copied to the device on a side stream and gathered asynchronously (RCCL over xGMI; gloo with host
    tensors in the CPU tests), double buffered.  Buffers have a fixed capacity agreed on beforehand; the first 16 bytes carry the
    payload length."""

    HEADER = 16

    def __init__(self, capacity, dst=0, device=None, depth=2):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank, self.dst, self.depth = dist.get_world_size(), dist.get_rank(), dst, depth
        self.dev = device if device is not None else torch.device("cpu")
        self.on_gpu = self.dev.type == "cuda"
        n = int(capacity) + self.HEADER
        self.stage = [torch.empty(n, dtype=torch.uint8) for _ in range(depth)]
        if self.on_gpu:
            self.stage = [t.pin_memory() for t in self.stage]
            self.dbuf = [torch.empty(n, dtype=torch.uint8, device=self.dev) for _ in range(depth)]
            self.stream = torch.cuda.Stream(device=self.dev)
        else:
            self.dbuf = self.stage
            self.stream = None
        self.recv = [[torch.empty(n, dtype=torch.uint8, device=self.dev) for _ in range(self.world)] for _ in range(depth)] if self.rank == dst else None
        self.work = [None] * depth
        self.events = [None] * depth  # GPU: completion of buffer j's gather on the side stream
        self.cur = 0
        self.done = -1  # buffer index of the last submitted step
        self.lock = threading.Lock()
        # ownership of the staging buffers: a buffer is free, or owned by a caller (acquire .. submit), or in flight (its gather queued);
        # acquire() hands out only a buffer nobody owns, waiting for the oldest one in flight when none is free
        self.cond = threading.Condition(self.lock)
        self.free = list(range(depth))
        self.in_flight = []  # submitted buffers, oldest first

    def _wait(self, j):
        """blocks until the gather of buffer j (already taken off the in-flight list by the caller) is complete; called WITHOUT the lock, so
        that one caller waiting for its buffer does not keep the others from acquiring and submitting theirs"""
        if self.on_gpu:
            if self.events[j] is not None:
                self.events[j].synchronize()  # recorded on the side stream right behind this buffer's gather: nothing later is waited for
        elif self.work[j] is not None:
            self.work[j].wait()
        self.work[j] = None

    def acquire(self):
        """(j, numpy view of the payload area of staging buffer j); the buffer belongs to the caller until submit(nbytes, j).
        Thread-safe: several steps may be in flight (depth must exceed their number for the gathers to overlap with them)."""
        wait_for = None
        with self.cond:
            while True:
                if self.free:
                    j = self.free.pop(0)
                    break
                if self.in_flight:
                    j = wait_for = self.in_flight.pop(0)  # the oldest gather: once it is done its buffers may be written again
                    break
                self.cond.wait()  # every buffer is in the hands of another caller: one of them will submit
        if wait_for is not None:
            self._wait(wait_for)
        return j, self.stage[j].numpy()[self.HEADER:]

    def device_area(self, j):
        """(address, capacity) of the payload area of DEVICE buffer j (after acquire()): a producer on the device -- the fill writing its
        sequences with mtg_fill_prepared_serial_device -- fills it in place and calls submit(nbytes, j, on_device=True); the payload then
        never crosses PCIe on its way to the gather"""
        if not self.on_gpu:
            raise RuntimeError("device buffers exist only with a GPU backend")
        return self.dbuf[j].data_ptr() + self.HEADER, self.dbuf[j].numel() - self.HEADER

    def buffer(self):
        """numpy view of this step's payload area (capacity bytes); valid until submit()  (one step at a time)"""
        self.cur, view = self.acquire()
        return view

    def submit(self, nbytes, j=None, on_device=False, host_copy=False):
        """queue the gather of buffer j.  on_device: the payload was produced in the device buffer (device_area); host_copy: it is also
        brought to the page-locked staging buffer of this rank (so that a rank's results are in its host memory as they are without a
        gather), on the side stream, next to the gather"""
        j = self.cur if j is None else j
        if nbytes + self.HEADER > self.stage[j].numel():
            raise ValueError("payload of %d bytes exceeds the agreed capacity" % nbytes)
        self.stage[j].numpy()[: self.HEADER].view(np.int64)[0] = nbytes
        with self.cond:  # collectives are issued one at a time, in the same number on every rank
            if self.on_gpu:
                self.stream.wait_stream(self.torch.cuda.current_stream(self.dev))
                with self.torch.cuda.stream(self.stream):
                    if on_device:  # the payload is in dbuf[j] already: only its length goes up
                        self.dbuf[j][: self.HEADER].copy_(self.stage[j][: self.HEADER], non_blocking=True)
                        if host_copy and nbytes:
                            self.stage[j][self.HEADER: self.HEADER + nbytes].copy_(self.dbuf[j][self.HEADER: self.HEADER + nbytes], non_blocking=True)
                    else:
                        self.dbuf[j].copy_(self.stage[j], non_blocking=True)
                    self.work[j] = self.dist.gather(self.dbuf[j], self.recv[j] if self.rank == self.dst else None, dst=self.dst, async_op=True)
                    self.work[j].wait()  # the side stream waits for the collective (the host does not)
                    if self.events[j] is None:
                        self.events[j] = self.torch.cuda.Event()
                    self.events[j].record(self.stream)
            else:
                self.work[j] = self.dist.gather(self.dbuf[j], self.recv[j] if self.rank == self.dst else None, dst=self.dst, async_op=True)
            self.done = j
            self.in_flight.append(j)  # the caller's ownership ends here: copy and gather are queued
            self.cond.notify()

    def drain(self):
        while True:
            with self.cond:
                if not self.in_flight:
                    break
                j = self.in_flight.pop(0)
            self._wait(j)
            with self.cond:
                self.free.append(j)
                self.cond.notify()
        if self.on_gpu:
            self.torch.cuda.synchronize(self.dev)

    def last(self):
        """after drain(): the payloads of the last submitted step, one numpy array per rank (rank dst only)"""
        if self.rank != self.dst or self.done < 0:
            return None
        out = []
        for t in self.recv[self.done]:
            a = t.cpu().numpy()
            n = int(a[: self.HEADER].view(np.int64)[0])
            out.append(a[self.HEADER: self.HEADER + n].copy())
        return out
