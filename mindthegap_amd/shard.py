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


def strong_plan(total_sites, batch_sites, rank, world):
    """Strong scaling of one fixed site set (bench.py, BASELINE config 5): the sites are sharded contiguously over the ranks (shard_range), a
    rank cuts its shard into batches of at most `batch_sites` sites, and a batch's GLOBAL index is its position in site order -- rank r's
    batches follow those of the ranks before it -- so that rank 0 can put gathered payloads back in input order by their tags.  Returns
    {my: [(global batch index, first site, end site)], per_rank: batches of every rank, n_batches_job, max_per_rank}: a rank with fewer
    batches than max_per_rank pads its gathers (every rank must issue the same collectives)."""
    lo, hi = shard_range(total_sites, rank, world)
    bs = min(batch_sites, max(1, (total_sites + world - 1) // world))
    per_rank = []
    for r in range(world):
        l, h = shard_range(total_sites, r, world)
        per_rank.append((h - l + bs - 1) // bs)
    first = sum(per_rank[:rank])
    my = [(first + i, s, min(s + bs, hi)) for i, s in enumerate(range(lo, hi, bs))]
    return dict(my=my, per_rank=per_rank, n_batches_job=sum(per_rank), max_per_rank=max(per_rank) if per_rank else 0, batch_sites=bs)


def gather_slots_for(max_per_rank, in_flight):
    """(slots, depth) of the gather bench.py uses for a rank with up to max_per_rank batches per step: several payloads per collective once a
    rank has three or more batches per step (what a rank pays per gather is its orchestration, not its bytes)"""
    slots = min(6, max_per_rank) if max_per_rank >= 3 else 1
    depth = max(in_flight, 1) + 1 if slots == 1 else max(3, (max(in_flight, 1) + slots - 1) // slots + 2)
    return slots, depth


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
    payload length and the payload's TAG (the global index of the batch it holds: several batches are in flight and the callers of a
    rank race, so payloads reach the destination in any order -- the tag is what lets it put them back in input order).

    on_arrival (rank dst only): called with [(rank, tag, payload tensor)] for every completed gather, before its buffers are reused, by
    whichever thread waits for that gather; the tensors live where the collective left them (device memory with RCCL)."""

    HEADER = 16

    def __init__(self, capacity, dst=0, device=None, depth=2, on_arrival=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank, self.dst, self.depth = dist.get_world_size(), dist.get_rank(), dst, depth
        self.prefix = 0  # bytes of the payload area that are always taken from the staging buffer, even when the payload was produced on the device (SlottedGather's slot table)
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
        self.on_arrival = on_arrival
        self.arrival_lock = threading.Lock()
        self.delivered = [True] * depth  # buffer j's payloads have been handed to on_arrival
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
        if self.on_arrival is not None and self.rank == self.dst and not self.delivered[j]:
            with self.arrival_lock:  # one delivery at a time: the consumer need not be thread-safe
                if not self.delivered[j]:
                    self.delivered[j] = True
                    self.on_arrival(self.payloads(j))

    def payloads(self, j):
        """[(rank, tag, payload tensor)] of the completed gather in buffer j (rank dst)"""
        # one copy to the host for the heads of all ranks' buffers (the gather's own 16 bytes and the first 64 of the payload, a relocatable
        # batch's header): a synchronisation per rank and field was most of what an arrival cost the thread that waited for it
        H = self.HEADER + WIRE_HEADER
        heads = self.torch.stack([t[:H] for t in self.recv[j]]).cpu().numpy() if self.recv[j][0].numel() >= H else None
        out = []
        for r, t in enumerate(self.recv[j]):
            hd = (heads[r, : self.HEADER] if heads is not None else t[: self.HEADER].cpu().numpy()).view(np.int64)
            item = GatherItem((r, int(hd[1]), t[self.HEADER: self.HEADER + int(hd[0])]))
            item.head = heads[r, self.HEADER:].copy() if heads is not None and int(hd[0]) >= WIRE_HEADER else None
            out.append(item)
        return out

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

    def submit(self, nbytes, j=None, on_device=False, host_copy=False, tag=-1):
        """queue the gather of buffer j.  on_device: the payload was produced in the device buffer (device_area); host_copy: it is also
        brought to the page-locked staging buffer of this rank (so that a rank's results are in its host memory as they are without a
        gather), on the side stream, next to the gather"""
        j = self.cur if j is None else j
        if nbytes + self.HEADER > self.stage[j].numel():
            raise ValueError("payload of %d bytes exceeds the agreed capacity" % nbytes)
        hd = self.stage[j].numpy()[: self.HEADER].view(np.int64)
        hd[0], hd[1] = nbytes, tag
        self.delivered[j] = False
        with self.cond:  # collectives are issued one at a time, in the same number on every rank
            if self.on_gpu:
                self.stream.wait_stream(self.torch.cuda.current_stream(self.dev))
                with self.torch.cuda.stream(self.stream):
                    if on_device:  # the payload is in dbuf[j] already: only its length (and the slot table, if any) goes up
                        self.dbuf[j][: self.HEADER + self.prefix].copy_(self.stage[j][: self.HEADER + self.prefix], non_blocking=True)
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


class SlottedGather:
    """Several payloads per collective.  What a rank pays per gather is not the bytes but the orchestration: a handful of stream operations and a
    collective call per batch, issued under one lock by whichever caller thread finished its batch -- with six batches in flight on one
    process that was 0.3 ms per batch, a third of the step (profiles/r03_dry_one_rank_rccl.json: 0.71 x the plain rate before a second GPU
    exists).  Here a buffer has `slots` payload areas and a slot table; callers acquire a SLOT, fill it (on the device or in the staging
    buffer) and submit it; the buffer's one gather is issued by the caller that submits its last slot.  Payloads keep their tags, so the
    receiving side sees exactly what it saw before, `slots` at a time.  acquire() / device_area() / submit() / drain() as PipelinedGather,
    the first value acquire() returns being an opaque slot handle.  Every rank must submit the same number of slots (drain() fills up the
    open buffer with empty slots, tag -1)."""

    SLOT_HDR = 16

    def __init__(self, capacity, slots=3, dst=0, device=None, depth=3, on_arrival=None):
        self.K = max(1, int(slots))
        self.cap = (int(capacity) + 63) & ~63
        self.user_arrival = on_arrival
        self.inner = PipelinedGather(self.K * self.SLOT_HDR + self.K * self.cap, dst=dst, device=device, depth=depth, on_arrival=self._arrive if on_arrival is not None else None)
        self.inner.prefix = self.K * self.SLOT_HDR
        self.on_gpu = self.inner.on_gpu
        self.lock = threading.Lock()
        self.open_lock = threading.Lock()  # held by the one caller that fetches the next buffer (it may wait for a gather; submitters must not wait for it)
        self.open = None       # buffer that still has slots to hand out
        self.next_slot = 0
        self.submitted = {}    # buffer -> slots submitted so far
        self.mode = {}         # buffer -> payloads produced on the device?

    def _slot_off(self, s):
        return self.K * self.SLOT_HDR + s * self.cap

    def acquire(self):
        while True:
            with self.lock:
                if self.open is not None:
                    j, s = self.open, self.next_slot
                    self.next_slot += 1
                    if self.next_slot == self.K:
                        self.open = None
                    break
            with self.open_lock:  # the next buffer: fetched outside self.lock, so that callers that only want to submit a slot get through
                with self.lock:
                    if self.open is not None:
                        continue
                j2, _ = self.inner.acquire()  # may wait for the oldest gather in flight
                with self.lock:
                    self.open, self.next_slot = j2, 0
                    self.submitted[j2] = 0
                    self.mode[j2] = None
        view = self.inner.stage[j].numpy()[PipelinedGather.HEADER + self._slot_off(s): PipelinedGather.HEADER + self._slot_off(s) + self.cap]
        return j * self.K + s, view

    def device_area(self, h):
        j, s = divmod(h, self.K)
        base, _cap = self.inner.device_area(j)
        return base + self._slot_off(s), self.cap

    def submit(self, nbytes, h, on_device=False, host_copy=False, tag=-1):
        if nbytes > self.cap:
            raise ValueError("payload of %d bytes exceeds the agreed capacity" % nbytes)
        j, s = divmod(h, self.K)
        tab = self.inner.stage[j].numpy()[PipelinedGather.HEADER: PipelinedGather.HEADER + self.K * self.SLOT_HDR].view(np.int64)
        tab[2 * s], tab[2 * s + 1] = nbytes, tag
        with self.lock:
            if nbytes:
                if self.mode[j] is None:
                    self.mode[j] = bool(on_device)
                elif self.mode[j] != bool(on_device):
                    raise ValueError("the payloads of one buffer must all be produced on the device, or all in the staging buffer")
            self.submitted[j] += 1
            last = self.submitted[j] == self.K
        if last:  # this caller issues the buffer's gather
            self.inner.submit(self.K * self.SLOT_HDR + self.K * self.cap, j, on_device=bool(self.mode[j]), tag=int(tab[1]))

    def flush(self):
        """the open buffer leaves with its remaining slots empty.  Only ever pads an EXISTING open buffer: its remaining slots are taken under the
        lock in one go, and nothing here fetches a fresh buffer (round 4 re-checked `open` outside acquire(): a thread taking the last slot in
        between made flush pad a new buffer -- one collective more on this rank only, a hang).  Not to be called while other threads acquire()."""
        with self.lock:
            if self.open is None:
                return
            j, first = self.open, self.next_slot
            self.open = None
            self.next_slot = self.K
        for s in range(first, self.K):
            self.submit(0, j * self.K + s, tag=-1)

    def drain(self):
        self.flush()
        self.inner.drain()

    def _arrive(self, items):
        """inner payloads (one per rank) -> the slots' payloads, with their tags; the slot tables and the payloads' 64-byte heads come to the host in two copies"""
        import torch
        K, H = self.K, self.SLOT_HDR
        tabs = torch.stack([t[: K * H] for (_r, _tag, t) in items]).cpu().numpy().view(np.int64)  # [rank, 2K]
        offs = [self._slot_off(s) for s in range(K)]
        heads = torch.stack([t[o: o + WIRE_HEADER] for (_r, _tag, t) in items for o in offs]).cpu().numpy()  # [rank * K, 64]
        out = []
        for ri, (r, _tag, t) in enumerate(items):
            for s in range(K):
                nb, tg = int(tabs[ri, 2 * s]), int(tabs[ri, 2 * s + 1])
                it = GatherItem((r, tg, t[offs[s]: offs[s] + nb]))
                it.head = heads[ri * K + s].copy() if nb >= WIRE_HEADER else None
                out.append(it)
        self.user_arrival(out)


# ---- checking a payload where it arrived -------------------------------------------------------------------------------------------------
_C1 = np.uint64(0x9E3779B97F4A7C15).astype(np.int64)
_C2 = np.uint64(0xBF58476D1CE4E5B9).astype(np.int64)
WIRE_MAGIC = 0x3145524957474D54
WIRE_HEADER = 64


class GatherItem(tuple):
    """(rank, tag, payload tensor) of a completed gather; .head = the payload's first 64 bytes on the host (numpy uint8) when it has them"""
    head = None


def wire_check(payload, checksum=True, head=None):
    """validates a relocatable batch (include/mtg_fill.h: mtg_wire_header) on the device (or host) the tensor lives on, without moving it:
    magic, sizes and (checksum=True) the checksum of the body recomputed with tensor arithmetic (64-bit wrap-around = the C side's; five passes
    over the payload).  Returns the header as a dict with 'ok'."""
    import torch
    n = payload.numel()
    if n < WIRE_HEADER:
        return {"ok": False, "why": "short"}
    hd = (head if head is not None else payload[:WIRE_HEADER].cpu().numpy()).view(np.uint64)  # head: already on the host (GatherItem.head)
    h = {"magic": int(hd[0]), "tag": int(hd[1]), "n_gaps": int(hd[2]), "n_filled": int(hd[3]), "seq_bytes": int(hd[4]), "ext_bytes": int(hd[5]),
         "total_bytes": int(hd[6]), "checksum": int(hd[7])}
    if h["magic"] != WIRE_MAGIC or h["total_bytes"] != n or n % 8:
        h.update(ok=False, why="header")
        return h
    if not checksum:  # header only: what the receiving rank can afford for every payload inside a timed step
        h.update(ok=True, why="")
        return h
    body = payload[WIRE_HEADER:].view(torch.int64)
    idx = torch.arange(body.numel(), dtype=torch.int64, device=body.device)
    s = int((((body ^ (idx * int(_C1))) * int(_C2)).sum()).item()) & 0xFFFFFFFFFFFFFFFF
    h.update(ok=(s == h["checksum"]), why="" if s == h["checksum"] else "checksum")
    return h


class OrderedArrivals:
    """rank dst: payloads arrive tagged with their global batch index, in any order; take() hands them out in index order.  The payloads
    are copied to host memory on arrival (the gather's buffers are reused)."""

    def __init__(self, n_batches, check=None):
        self.n = n_batches
        self.check = check  # payload tensor -> header dict with 'ok' (wire_check for a bare relocatable batch)
        self.have = {}
        self.next = 0
        self.cond = threading.Condition()
        self.bad = []

    def arrive(self, items):
        with self.cond:
            for rank, tag, t in items:
                if tag < 0:
                    continue  # a rank with fewer batches than the others sends empty payloads to keep the collectives in step
                if self.check is not None:
                    h = self.check(t)
                    if not h["ok"] or h["tag"] != tag:
                        self.bad.append((rank, tag, h.get("why")))
                self.have[tag] = t.cpu().numpy().copy()
            self.cond.notify_all()

    def take(self, timeout=600.0):
        """the payload of the next batch in input order (blocks until it has arrived); None after the last"""
        with self.cond:
            if self.next >= self.n:
                return None
            if not self.cond.wait_for(lambda: self.next in self.have, timeout=timeout):
                raise RuntimeError("batch %d never arrived" % self.next)
            p = self.have.pop(self.next)
            self.next += 1
            return p


_RC = dict(zip("ACGTacgt", "TGCAtgca"))


def revcomp_dropping(s):
    """revcomp_sequence (src/Utils.cpp:44-77): complements A, C, G, T in either case and DROPS every other character"""
    return "".join(_RC[c] for c in reversed(s) if c in _RC)


def fill_bkpt_sharded(idx, sites, out_prefix, params=None, batch_sites=100000, device=None, extend=False, filter=False, fwd_only=False, in_flight=2, sample="index"):
    """`MindTheGap fill -bkpt` over the ranks of a torch.distributed job (one process per GPU, the index replicated, all ranks on one node
    and one file system): the sites -- the same list of (name, name_r, source, target) on every rank -- are cut into batches of batch_sites,
    batch b goes to rank b mod world (cyclic), every rank fills its batches (forward attempt, reverse attempt for the unfilled
    sites) and formats them with the tool's own writers (mtg_format_bkpt).  No result byte travels between the ranks: after every round
    (one batch per rank) an all_gather of the four pieces' SIZES gives every rank the file offsets of that round's batches in input order,
    and each rank writes its own pieces with pwrite -- the files are those of the single-process tool, byte for byte
    (src/Filler.cpp:682-683 writes under a lock in completion order; input order is its -nb-cores 1 order).  The fill of a rank's next
    batch runs while a helper thread formats and writes the previous one (in_flight batches are held at most).
    Returns the number of sites (rank 0) / None."""
    import os
    import queue
    import torch
    import torch.distributed as dist
    from . import lib as L
    rank, world = dist.get_rank(), dist.get_world_size()
    params = params or L.FillParams()
    nb = (len(sites) + batch_sites - 1) // batch_sites
    rounds = (nb + world - 1) // world
    cdev = device if device is not None else torch.device("cpu")
    names = [".insertions.fasta", ".info.txt", ".insertions.vcf"] + ([".extensions.fasta"] if extend else [])
    keys = ["fasta", "info", "vcf"] + (["ext"] if extend else [])
    header = L.vcf_header(sample, out_prefix)
    if rank == 0:  # the files exist (and hold the VCF header) before anyone writes a piece
        for nm in names:
            with open(out_prefix + nm, "wb") as f:
                if nm == ".insertions.vcf":
                    f.write(header)
    dist.barrier()
    fds = [os.open(out_prefix + nm, os.O_WRONLY) for nm in names]
    base = [len(header) if nm == ".insertions.vcf" else 0 for nm in names]  # file offset of the next round, per file

    def one(b):
        """fills and formats batch b; returns the text pieces in the order of `names`"""
        s0, s1 = b * batch_sites, min(len(sites), (b + 1) * batch_sites)
        gaps = [L.Gap(st[2], st[3], [(st[3], st[1], False)], is_anchor_repeated=bool(st[4]) if len(st) > 4 else False) for st in sites[s0:s1]]
        h, nf, _ = idx.fill_prepared(L.Index.prepare_gaps(gaps), params, want_seqs=False)
        unf = [] if fwd_only else [j for j in range(s1 - s0) if nf[j] == 0]
        rev_index = np.full(s1 - s0, -1, dtype=np.int64)
        h2 = None
        if unf:
            rg = []
            for q, j in enumerate(unf):
                st = sites[s0 + j]
                name, src, tgt = st[0], st[2], st[3]
                rs, rt = revcomp_dropping(tgt), revcomp_dropping(src)
                rg.append(L.Gap(rs, rt, [(rt, name, False)], is_anchor_repeated=bool(st[4]) if len(st) > 4 else False, reverse=True))
                rev_index[j] = q
            h2, _, _ = idx.fill_prepared(L.Index.prepare_gaps(rg), params, want_seqs=False)
        text = L.format_bkpt([st[:4] for st in sites[s0:s1]], h, h2, rev_index, filter=filter, extend=extend)
        idx.free_results(h)
        if h2 is not None:
            idx.free_results(h2)
        return [text[k] for k in keys]

    # the writer of this rank: takes (round, pieces) in round order, agrees on the offsets with the other ranks' writers, writes
    todo = queue.Queue(maxsize=max(1, in_flight))
    err = []

    def writer():
        try:
            for i in range(rounds):
                item = todo.get()
                if item is None:
                    return
                pieces = item if item else None
                mine = torch.tensor([len(p) for p in pieces] if pieces is not None else [0] * len(names), dtype=torch.int64, device=cdev)
                allsz = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allsz, mine)  # one small collective per round, issued by every rank's writer in the same order
                allsz = torch.stack(allsz).cpu().numpy()  # [rank, file]
                for f in range(len(names)):
                    off = base[f] + int(allsz[:rank, f].sum())
                    if pieces is not None and len(pieces[f]):
                        mv, done = memoryview(pieces[f]), 0
                        while done < len(mv):
                            done += os.pwrite(fds[f], mv[done:], off + done)
                    base[f] += int(allsz[:, f].sum())
        except BaseException as e:  # surfaced below
            err.append(e)

    wt = threading.Thread(target=writer)
    wt.start()

    def hand_over(item):
        """puts item on the writer's queue without ever blocking behind a writer that has died (a failed pwrite, a collective that timed out):
        the put is retried in short waits that look at the writer's state; False = the writer is gone"""
        while True:
            if err or not wt.is_alive():
                return False
            try:
                todo.put(item, timeout=0.2)
                return True
            except queue.Full:
                continue

    failure = None
    try:
        for i in range(rounds):
            b = i * world + rank
            # a rank without a batch in the last round still takes part in the exchange of sizes
            if not hand_over(one(b) if b < nb else _EMPTY_ROUND):
                break
    except BaseException as e:  # the fill failed: the writer leaves (the other ranks' writers wait for this rank's sizes until their timeout)
        failure = e
    # the sentinel: only needed when the writer has not seen all its rounds; never a blocking put (round 4 could hang here with a full queue and a dead writer)
    if (failure is not None or err) and wt.is_alive():
        hand_over(None)
    wt.join()
    if failure is not None:
        for fd in fds:
            os.close(fd)
        raise failure
    for fd in fds:
        os.close(fd)
    if err:
        raise err[0]
    dist.barrier()
    return len(sites) if rank == 0 else None


_EMPTY_ROUND = ()
