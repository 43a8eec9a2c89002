"""Row-sharded optimal assignment across GPUs.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  Rank r owns cost
rows [r*rps, (r+1)*rps) x all n columns (rps = ceil(n / world)); prices and column owners are
replicated.  Per bidding round there is exactly ONE exchange step: a MAX all-reduce of the n
packed 64-bit bid keys (512 KiB at n = 65 536) — every rank then applies the identical update, so
no second exchange and no termination message is needed (rounds are a fixed count; a converged
round makes all later device kernels exit immediately).  The few rows left free are finished by
the augmenting-path workgroup on rank 0, which reads the other ranks' compressed shards in place
through hipIpc peer mappings over xGMI (fallback: gather copies).  owner[] is then broadcast and
each rank sums its rows' costs; one SUM all-reduce of a scalar gives the total.

The reference has no multi-process path for the solve (its only parallel construct is the
8-process pool finder, findpool.c:138-164); this design is new, not a translation.

`solve_sharded` is written against a small shard interface so that the collective logic is
testable on CPU with gloo (tests/test_sharded_gloo.py supplies a numpy model of a shard); the
product shard is `HipShard`, which needs the HIP library and a GPU.
"""
import ctypes
import os

import numpy as np

from . import _ffi

DEFAULT_ROUNDS = 12  # same as td_assign (TD_MAX_ROUNDS)


def shard_bounds(n, world, rank):
    """rows owned by `rank`: (row0, nrows, rows_per_shard) with a uniform shard height."""
    rps = (n + world - 1) // world
    row0 = min(n, rank * rps)
    return row0, max(0, min(rps, n - row0)), rps


class HipShard:
    """One rank's rows on its GPU, through the C ABI (td_shard_*)."""

    def __init__(self, n, row0, nrows, cost_rows, share_torch_stream=True):
        import torch
        self.torch = torch
        self.lib = _ffi.lib()
        # Run the library AND the collectives on one dedicated, non-null torch stream: RCCL orders
        # its collectives with torch's current stream, so bid -> all_reduce -> apply needs no host
        # synchronisation.  (torch's default stream has handle 0, which td_set_stream reads as
        # "use the library's own stream": sharing it would leave kernels and collectives unordered.)
        self.shared_stream = bool(share_torch_stream)
        self.stream = None
        if self.shared_stream and HipShard._stream_owner is not None:
            # td_set_stream is process-global: a second shard installing ITS stream would move the first one's
            # kernels off the stream its collectives are ordered with
            raise _ffi.TdError("another HipShard already shares its torch stream with the library: close it first, or pass "
                               "share_torch_stream=False")
        if self.shared_stream:
            self.stream = torch.cuda.Stream()
            self.stream.wait_stream(torch.cuda.current_stream())   # inputs written on the caller's stream
            assert self.stream.cuda_stream != 0
            _ffi.check(self.lib.td_set_stream(ctypes.c_void_p(self.stream.cuda_stream)))
            HipShard._stream_owner = self
        self.n, self.row0, self.nrows = n, row0, nrows
        self._cost = cost_rows  # keep alive: the library reads it again for the total
        h = ctypes.c_void_p()
        _ffi.check(self.lib.td_shard_create(n, row0, nrows, _ffi.addr(cost_rows) if nrows else None,
                                            ctypes.byref(h)))
        self.h = h
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._opened = []
        self._handle_cache = {}   # 64-byte hipIpc handle -> mapped pointer (rank 0: peers' compressed shards)
        self.ipc_opens = 0

    def stream_ctx(self):
        """context in which solve_sharded issues its collectives and tensor ops"""
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def close(self):
        for p in self._opened:
            self.lib.td_ipc_close(ctypes.c_void_p(p))
        self._opened = []
        self._handle_cache = {}
        if self.h:
            self.lib.td_shard_destroy(self.h)
            self.h = None
        if self.stream is not None:
            self.stream.synchronize()
            if HipShard._stream_owner is self:
                self.lib.td_set_stream(None)   # back to the library's own stream (the state a HipShard found)
                HipShard._stream_owner = None
            self.stream = None

    # -- compression width agreement
    def compress(self, bytes_per_cell):
        fits = ctypes.c_int(0)
        _ffi.check(self.lib.td_shard_compress(self.h, bytes_per_cell, ctypes.byref(fits)))
        return bool(fits.value)

    def range(self):
        r = ctypes.c_int64(0)
        _ffi.check(self.lib.td_shard_range(self.h, ctypes.byref(r)))
        return int(r.value)

    def begin(self, global_range=-1):
        _ffi.check(self.lib.td_shard_begin(self.h, int(global_range)))

    def new_keys(self):
        return self.torch.zeros(self.lib.td_shard_keys_len(self.h) + 16, dtype=self.torch.int64, device=self.device)

    def bid(self, rnd, keys):
        _ffi.check(self.lib.td_shard_bid(self.h, rnd, keys.data_ptr()))
        if not self.shared_stream:
            _ffi.check(self.lib.td_synchronize())  # the collective runs on torch's stream

    def apply(self, rnd, keys):
        _ffi.check(self.lib.td_shard_apply(self.h, rnd, keys.data_ptr()))
        if not self.shared_stream:
            _ffi.check(self.lib.td_synchronize())

    # -- all rounds in one C call, the collective issued by the library itself (td_shard_rounds)
    _comm_key = None   # (world, rank) of the library's communicator in this process
    _stream_owner = None   # the one HipShard whose torch stream is installed in the (process-global) library context

    @classmethod
    def destroy_comm(cls):
        """td_comm_destroy + forget the cached communicator key (call before the process group goes away, or
        before td.shutdown(): a later native_comm() builds a fresh communicator)."""
        if cls._comm_key is not None:
            _ffi.lib().td_comm_destroy()
            cls._comm_key = None

    def native_comm(self, dist):
        """Builds (once per process) the library's own RCCL communicator: rank 0 makes the 128-byte
        id, torch.distributed broadcasts it.  Returns False when RCCL cannot be used (gloo test runs,
        TD_SHARD_NATIVE=0)."""
        if os.environ.get("TD_SHARD_NATIVE", "1") == "0" or dist.get_backend() != "nccl":
            return False
        world, rank = dist.get_world_size(), dist.get_rank()
        if HipShard._comm_key == (world, rank):
            return True
        buf = (ctypes.c_ubyte * 128)()
        if rank == 0:
            _ffi.check(self.lib.td_comm_unique_id(buf))
        t = self.torch.tensor(list(buf), dtype=self.torch.uint8, device=self.device)
        if world > 1:
            dist.broadcast(t, 0)
        raw = bytes(t.cpu().numpy().tobytes())
        idb = (ctypes.c_ubyte * 128).from_buffer_copy(raw)
        self.torch.cuda.current_stream().synchronize()
        _ffi.check(self.lib.td_comm_init(world, rank, idb))
        HipShard._comm_key = (world, rank)
        return True

    def rounds_native(self, rounds, keys):
        _ffi.check(self.lib.td_shard_rounds(self.h, int(rounds), keys.data_ptr()))

    # -- finisher support
    def export_handle(self):
        """64-byte hipIpc handle of the compressed shard (uint8 tensor on the device)."""
        ptr = ctypes.c_void_p()
        nbytes = ctypes.c_uint64()
        _ffi.check(self.lib.td_shard_cc(self.h, ctypes.byref(ptr), ctypes.byref(nbytes)))
        buf = (ctypes.c_ubyte * 64)()
        rc = self.lib.td_ipc_export(ptr, buf)
        ok = rc == 0
        return self.torch.tensor(list(buf) + [1 if ok else 0], dtype=self.torch.uint8, device=self.device)

    def cc_ref(self):
        ptr = ctypes.c_void_p()
        _ffi.check(self.lib.td_shard_cc(self.h, ctypes.byref(ptr), None))
        return int(ptr.value or 0)

    def empty_owner(self):
        return self.torch.empty(self.n, dtype=self.torch.int32, device=self.device)

    def open_handle(self, handle_u8):
        """Peer mapping of another rank's compressed shard.  A peer's workspace is grow-only, so across the solves of
        one shard object its handle stays the same: the mapping is opened ONCE per distinct handle and reused
        (hipIpcOpenMemHandle per solve costs a driver call per peer and leaves one more mapping open each time)."""
        raw = bytes(handle_u8[:64].cpu().numpy().tobytes())
        cached = self._handle_cache.get(raw)
        if cached is not None:
            return cached
        buf = (ctypes.c_ubyte * 64).from_buffer_copy(raw)
        out = ctypes.c_void_p()
        rc = self.lib.td_ipc_open(buf, ctypes.byref(out))
        if rc != 0:
            return None
        self._opened.append(out.value)
        self._handle_cache[raw] = out.value
        self.ipc_opens += 1
        return out.value

    def cc_copy(self):
        """a torch uint8 copy of the compressed shard (gather fallback)"""
        ptr = ctypes.c_void_p()
        nbytes = ctypes.c_uint64()
        _ffi.check(self.lib.td_shard_cc(self.h, ctypes.byref(ptr), ctypes.byref(nbytes)))
        t = self.torch.empty(max(1, nbytes.value), dtype=self.torch.uint8, device=self.device)
        if nbytes.value:
            _ffi.check(self.lib.td_memcpy(t.data_ptr(), ptr, nbytes.value))
        return t

    def finish(self, refs, rps):
        """refs[k]: device address (own / hipIpc-mapped) or a torch uint8 device tensor (gathered copy)"""
        ptrs = [r.data_ptr() if hasattr(r, "data_ptr") else int(r) for r in refs]
        arr = (ctypes.c_void_p * len(ptrs))(*[ctypes.c_void_p(p) for p in ptrs])
        _ffi.check(self.lib.td_shard_finish(self.h, len(ptrs), arr, rps))

    def get_owner(self):
        t = self.torch.empty(self.n, dtype=self.torch.int32, device=self.device)
        _ffi.check(self.lib.td_shard_owner(self.h, t.data_ptr(), 0))
        return t

    def set_owner(self, owner):
        _ffi.check(self.lib.td_shard_owner(self.h, owner.data_ptr(), 1))

    def get_price(self):
        t = self.torch.empty(self.n, dtype=self.torch.int64, device=self.device)
        _ffi.check(self.lib.td_shard_price(self.h, t.data_ptr(), 0))
        return t

    def empty_price(self):
        return self.torch.empty(self.n, dtype=self.torch.int64, device=self.device)

    def set_price(self, price):
        _ffi.check(self.lib.td_shard_price(self.h, price.data_ptr(), 1))

    def totals(self, want_dual):
        tot = ctypes.c_int64(0)
        dual = ctypes.c_int64(0)
        _ffi.check(self.lib.td_shard_total(self.h, ctypes.byref(tot), ctypes.byref(dual) if want_dual else None))
        return int(tot.value), int(dual.value)

    def totals_dev(self, want_dual):
        """{partial total, partial dual bound, error flags} as a device int64 tensor, queued without a host round trip"""
        t = self.torch.empty(3, dtype=self.torch.int64, device=self.device)
        if not self.shared_stream:
            self.torch.cuda.current_stream().synchronize()
        _ffi.check(self.lib.td_shard_total_dev(self.h, t.data_ptr(), 1 if want_dual else 0))
        if not self.shared_stream:
            _ffi.check(self.lib.td_synchronize())
        return t

    def row_to_col(self):
        r = np.empty(self.nrows, np.int32)
        _ffi.check(self.lib.td_shard_row_to_col(self.h, r.ctypes.data if self.nrows else None))
        return r

    def scalar_tensor(self, values, dtype=None):
        return self.torch.tensor(values, dtype=dtype or self.torch.int64, device=self.device)

    def fused_round0(self, on=True):
        """the caller promises the constant-row exchange in every solve: a wide shard's compress pass may write round 0's
        bids itself (td_shard_options)"""
        _ffi.check(self.lib.td_shard_options(self.h, 1 if on else 0))

    # -- block-local start (csrc/td_blocks.h): phase A on the shard's own diagonal blocks, then ONE exchange
    def blocks_start(self, on=True):
        lazy = 4 if os.environ.get("TD_LAZY_CC", "1") != "0" else 0   # the narrow copy outside the diagonal slices only if phase A leaves rows
        _ffi.check(self.lib.td_shard_options(self.h, (3 | lazy) if on else 1))

    def compress_spec(self):
        """the 1-byte compress pass of the block-local start without waiting for its width flag (it travels in the segment)"""
        _ffi.check(self.lib.td_shard_compress_spec(self.h))

    def blocks_pending(self):
        return bool(self.lib.td_shard_blocks_pending(self.h))

    def phase_a(self):
        _ffi.check(self.lib.td_shard_phase_a(self.h))

    def state_segment(self, rps, fits):
        """this rank's segment of the exchange after phase A (int32 device tensor)"""
        seg = self.torch.empty(int(self.lib.td_shard_state_words(self.h, int(rps))), dtype=self.torch.int32, device=self.device)
        if not self.shared_stream:
            self.torch.cuda.current_stream().synchronize()
        _ffi.check(self.lib.td_shard_state_export(self.h, int(rps), 1 if fits else 0, seg.data_ptr()))
        if not self.shared_stream:
            _ffi.check(self.lib.td_synchronize())   # the exchange runs on torch's stream
        return seg

    def place_const(self):
        _ffi.check(self.lib.td_shard_place_const(self.h))

    def state_import(self, world, rank, rps, allseg):
        """the gathered segments -> the other slices' owners, the constant-row mask; returns the summary"""
        out = (ctypes.c_int64 * 6)()
        _ffi.check(self.lib.td_shard_state_import(self.h, int(world), int(rank), int(rps), allseg.data_ptr(), out))
        return {"fit": bool(out[0]), "ran": bool(out[1]), "left": int(out[2]), "nconst": int(out[3]), "range": int(out[4]), "word6": int(out[5])}

    # -- constant rows sit out the solve (td_shard_const_rows)
    def const_mask(self):
        """zeroed device mask of n ints with this shard's constant rows set (after compress); the caller sums it"""
        m = self.torch.zeros(self.n, dtype=self.torch.int32, device=self.device)
        if not self.shared_stream:
            self.torch.cuda.current_stream().synchronize()   # the fill runs on torch's stream, the kernel on the library's
        _ffi.check(self.lib.td_shard_const_rows(self.h, m.data_ptr(), 0))
        if not self.shared_stream:
            _ffi.check(self.lib.td_synchronize())
        return m

    def set_const_mask(self, mask):
        _ffi.check(self.lib.td_shard_const_rows(self.h, mask.data_ptr(), 1))
        if not self.shared_stream:
            _ffi.check(self.lib.td_synchronize())

    # -- the sorted matching of line-metric matrices over row shards (td_line_shard_*)
    def line_ws(self):
        return self.torch.empty(int(self.lib.td_line_shard_ws_words(self.n)), dtype=self.torch.int64, device=self.device)

    def line_phase(self, phase, ws):
        """runs one phase on this shard's rows; returns the (offset, length) of ws to SUM over all shards"""
        off, ln = ctypes.c_int64(0), ctypes.c_int64(0)
        _ffi.check(self.lib.td_line_shard_phase(phase, self.n, self.row0, self.nrows, _ffi.addr(self._cost) if self.nrows else None,
                                                ws.data_ptr(), ctypes.byref(off), ctypes.byref(ln)))
        if not self.shared_stream:
            _ffi.check(self.lib.td_synchronize())   # the exchange runs on torch's stream
        return int(off.value), int(ln.value)

    def line_plausible(self, ws):
        """after phase 0's exchange: did the anchors' owner find the first two rows compatible with a line metric"""
        return int(ws[4].item()) != 0    # ctl word LC_PLAUS

    def line_result(self, ws):
        """(accepted, total, local row_to_col) after the last exchange"""
        r = np.empty(self.nrows, np.int32)
        tot, acc = ctypes.c_int64(0), ctypes.c_int32(0)
        _ffi.check(self.lib.td_line_shard_result(self.n, self.row0, self.nrows, ws.data_ptr(), r.ctypes.data if self.nrows else None,
                                                 ctypes.byref(tot), ctypes.byref(acc)))
        return bool(acc.value), int(tot.value), r


def _staged(dist, t):
    """gloo has no device collectives on every build: stage device tensors through the host."""
    return dist.get_backend() == "gloo" and getattr(t, "is_cuda", False)


_NEED_FENCE = True


def _fence(t):
    """RCCL collectives are ordered with torch's current stream only; when the library runs on its
    own stream the result must be complete before the next library call (HipShard with
    share_torch_stream=True makes this unnecessary)."""
    if _NEED_FENCE and getattr(t, "is_cuda", False):
        import torch
        torch.cuda.current_stream().synchronize()


def all_reduce(dist, t, op):
    if _staged(dist, t):
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)
        _fence(t)


def broadcast(dist, t, src):
    if _staged(dist, t):
        c = t.cpu()
        dist.broadcast(c, src)
        t.copy_(c)
    else:
        dist.broadcast(t, src)
        _fence(t)


def all_gather_equal(dist, t):
    """all_gather of equally sized tensors -> list"""
    world = dist.get_world_size()
    if _staged(dist, t):
        c = t.cpu()
        out = [c.new_empty(c.shape) for _ in range(world)]
        dist.all_gather(out, c)
        return [o.to(t.device) for o in out]
    out = [t.new_empty(t.shape) for _ in range(world)]
    dist.all_gather(out, t)
    _fence(t)
    return out


def all_gather_cat(dist, t):
    """all_gather of equally sized 1-D tensors -> one tensor, rank order"""
    world = dist.get_world_size()
    if _staged(dist, t):
        c = t.cpu()
        out = c.new_empty(world * c.numel())
        dist.all_gather(list(out.chunk(world)), c)
        return out.to(t.device)
    out = t.new_empty(world * t.numel())
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, t)
    else:
        dist.all_gather(list(out.chunk(world)), t)
    _fence(out)
    return out


BLOCKS = 8   # diagonal blocks of the block-local start (a rank owns BLOCKS / world of them)


def blocks_ok(n, world):
    """can every rank start block-locally: whole blocks per rank, chunk-aligned column slices, rows wide enough for the
    compress pass that writes the zero-slice bids (k_compress_reg<.., BID0>: n >= 12 288)"""
    return BLOCKS % world == 0 and n % (16 * BLOCKS) == 0 and 12288 <= n <= 65536


LINE_PHASES = 4


def line_sharded(shards, dist):
    """The line-metric path of td_assign over row shards: `shards` = the shards THIS process drives (one
    per rank in production; several in one process for single-GPU tests and timing), `dist` = torch.distributed or
    None.  Four phases, each followed by ONE SUM all-reduce of a segment every shard writes disjointly (O(n) words;
    the O(n^2 / world) certificate pass stays local).  Returns (total, [local row_to_col per shard]) when every
    shard's rows certify the sorted matching - it is then optimal whatever the matrix was - else None (the caller
    runs the general sharded solve)."""
    world = dist.get_world_size() if dist is not None else 1
    wss = [sh.line_ws() for sh in shards]
    for phase in range(LINE_PHASES):
        segs = []
        for sh, ws in zip(shards, wss):
            off, ln = sh.line_phase(phase, ws)
            segs.append(ws[off:off + ln])
        red = segs[0]
        for sg in segs[1:]:
            red += sg
        if world > 1:
            all_reduce(dist, red, dist.ReduceOp.SUM)
        for sg in segs[1:]:
            sg.copy_(red)
        if len(segs) > 1:
            _fence(red)
        if phase == 0 and hasattr(shards[0], "line_plausible") and not shards[0].line_plausible(wss[0]):
            return None   # refused after ONE exchange (every rank reads the same summed word)
    out = [sh.line_result(ws) for sh, ws in zip(shards, wss)]
    if not all(acc for acc, _, _ in out):
        return None
    return out[0][1], [r for _, _, r in out]


def solve_sharded(shard, dist, rounds=DEFAULT_ROUNDS, want_dual=False, use_ipc=None):
    """Collective part of the sharded solve; `shard` implements the HipShard interface and `dist`
    is torch.distributed (initialised). Returns (local row_to_col, total[, dual])."""
    ctx = shard.stream_ctx() if hasattr(shard, "stream_ctx") else None
    if ctx is None:
        return _solve_sharded(shard, dist, rounds, want_dual, use_ipc)
    with ctx:
        return _solve_sharded(shard, dist, rounds, want_dual, use_ipc)


def _solve_sharded(shard, dist, rounds, want_dual, use_ipc):
    global _NEED_FENCE
    world, rank = dist.get_world_size(), dist.get_rank()
    n = shard.n
    _, _, rps = shard_bounds(n, world, rank)
    MIN, MAX, SUM = dist.ReduceOp.MIN, dist.ReduceOp.MAX, dist.ReduceOp.SUM
    _NEED_FENCE = not getattr(shard, "shared_stream", False)
    # 0. the sorted matching first (td_assign's order): O(n) exchanged, one local pass over the rows; accepted only
    #    when every rank's rows certify it (total == dual bound by construction)
    solve_sharded.last_path = "auction"
    defer0 = hasattr(shard, "const_mask") and os.environ.get("TD_DEFER_CONST", "1") != "0"
    can_blocks0 = shard.blocks_ok(n, world) if hasattr(shard, "blocks_ok") else blocks_ok(n, world)
    blocks_first = (defer0 and hasattr(shard, "phase_a") and os.environ.get("TD_SHARD_BLOCKS", "1") != "0" and can_blocks0 and
                    hasattr(shard, "compress_spec"))
    want_line = hasattr(shard, "line_phase") and n >= 2 and os.environ.get("TD_LINE", "1") != "0"
    line_ws0 = None
    if want_line and blocks_first:
        # the block-local start follows: the line attempt's first test (do rows 0 and 1 look like a line metric at all) is queued
        # here, its verdict rides in the one all-gather below instead of an exchange + a read-back of its own
        line_ws0 = shard.line_ws()
        shard.line_phase(0, line_ws0)
    elif want_line:
        got = line_sharded([shard], dist)
        solve_sharded.last_path = "line" if got is not None else "auction"
        if got is not None:
            total, (r2c,) = got
            return (r2c, total, total) if want_dual else (r2c, total)
    defer = hasattr(shard, "const_mask") and os.environ.get("TD_DEFER_CONST", "1") != "0"
    # 1a. block-local start (csrc/td_blocks.h): the 1-byte attempt begins on every rank's own diagonal blocks — zero
    #     cells only, no price moves, nothing exchanged — and the ranks then meet ONCE (an all-gather that carries the
    #     width flag, the owners of each rank's column slice and its constant rows).  On tie-heavy instances
    #     (perf.jl) nothing is left for the rounds.
    widths = (1, 2, 4)
    left = None
    can_blocks = shard.blocks_ok(n, world) if hasattr(shard, "blocks_ok") else blocks_ok(n, world)
    if defer and hasattr(shard, "phase_a") and os.environ.get("TD_SHARD_BLOCKS", "1") != "0" and can_blocks:
        shard.blocks_start(True)
        if hasattr(shard, "compress_spec"):
            shard.compress_spec()      # no wait for the width flag: it is word 0 of the segment
            fits = True
        else:
            fits = shard.compress(1)
        if fits and shard.blocks_pending():
            shard.phase_a()
        seg = shard.state_segment(rps, fits)
        if line_ws0 is not None:
            seg[6:7].copy_(line_ws0[4:5])   # LC_PLAUS of the rank that owns row 0 (zero elsewhere)
        allseg = all_gather_cat(dist, seg) if (world > 1 or os.environ.get("TD_SHARD_FORCE_AR")) else seg
        summ = shard.state_import(world, rank, rps, allseg)
        if line_ws0 is not None and summ.get("word6", 0) != 0:
            # rows 0 and 1 are compatible with a line metric: the full attempt (four small exchanges, one local pass); the
            # block-local state stays valid if it is refused after all
            got = line_sharded([shard], dist)
            if got is not None:
                solve_sharded.last_path = "line"
                total, (r2c,) = got
                return (r2c, total, total) if want_dual else (r2c, total)
        if summ["fit"] and summ["ran"]:
            left = summ["left"]
            shard.begin(summ["range"])
            solve_sharded.last_path = "blocks"
        else:
            shard.blocks_start(False)
            widths = (1, 2, 4) if summ["fit"] else (2, 4)
    if left is None:
        # 1b. agree on the storage width (every rank must use the same one)
        if hasattr(shard, "fused_round0"):
            shard.fused_round0(defer)   # round 0 out of the compress pass needs the constant-row exchange below
        for width in widths:
            flag = shard.scalar_tensor([1 if shard.compress(width) else 0])
            if world > 1:
                all_reduce(dist, flag, MIN)
            if int(flag[0].item()) == 1:
                break
        else:
            raise _ffi.TdError("row cost range exceeds 2^32-2 on some rank")
        # constant rows (dummy cabs of a padded model) sit out the rounds and the searches, as in td_assign: one SUM
        # all-reduce of an n-int mask tells every rank (the finisher's above all) which rows they are
        if defer:
            mask = shard.const_mask()
            if world > 1:
                all_reduce(dist, mask, SUM)
            shard.set_const_mask(mask)
        # the packed-key range guard needs the largest row range of ANY rank (td_assign's TD_ERANGE rule)
        grange = -1
        if hasattr(shard, "range"):
            rt = shard.scalar_tensor([shard.range()])
            if world > 1:
                all_reduce(dist, rt, MAX)
            grange = int(rt[0].item())
        shard.begin(grange)
    solve_sharded.last_left = left
    if left == 0:
        # every row that bids has its column: no rounds, no finisher, no peer mappings.  Deferred constant rows take the
        # never-owned columns — the same placement on every rank from the replicated state — then on to the totals
        if summ["nconst"] > 0:
            shard.place_const()
        return _sharded_totals(shard, dist, world, want_dual, False)
    # 2. Jacobi bidding rounds: ONE exchange step per round
    keys = shard.new_keys()
    if hasattr(shard, "native_comm") and getattr(shard, "shared_stream", False) and shard.native_comm(dist):
        # one C call: bid -> RCCL MAX all-reduce -> apply per round, all on the library's stream
        shard.rounds_native(rounds, keys)
    else:
        for r in range(rounds):
            shard.bid(r, keys)
            if world > 1:
                all_reduce(dist, keys, MAX)
            shard.apply(r, keys)
    # 3. finisher on rank 0 over peer-mapped (or gathered) shards
    if use_ipc is None:
        use_ipc = os.environ.get("TD_SHARD_GATHER", "0") != "1"
    refs = None
    if world == 1:
        refs = [shard.cc_ref()]
    else:
        ipc_ok = shard.scalar_tensor([0])
        if use_ipc and hasattr(shard, "export_handle"):
            hs = all_gather_equal(dist, shard.export_handle())
            if rank == 0:
                good = all(int(x[64].item()) == 1 for x in hs)
                refs = [shard.cc_ref()]
                for k in range(1, world):
                    p = shard.open_handle(hs[k]) if good else None
                    if p is None:
                        good = False
                        break
                    refs.append(p)
                ipc_ok = shard.scalar_tensor([1 if good else 0])
            broadcast(dist, ipc_ok, 0)
        if int(ipc_ok[0].item()) != 1:
            # fallback: every shard's compressed rows are copied to rank 0 (padded to one size)
            mine = shard.cc_copy()
            sz = shard.scalar_tensor([int(mine.numel())])
            all_reduce(dist, sz, MAX)
            padded = mine.new_zeros(int(sz[0].item()))
            padded[:mine.numel()] = mine
            refs = all_gather_equal(dist, padded)
    if rank == 0:
        shard.finish(refs, rps)
        owner = shard.get_owner()
    else:
        owner = shard.empty_owner()
    if world > 1:
        broadcast(dist, owner, 0)
        if rank != 0:
            shard.set_owner(owner)
    del refs
    return _sharded_totals(shard, dist, world, want_dual, True)


def _sharded_totals(shard, dist, world, want_dual, finisher_ran):
    rank = dist.get_rank()
    SUM = dist.ReduceOp.SUM
    if world > 1 and want_dual and finisher_ran:  # the finisher moved prices on rank 0: the certificate needs them everywhere
        price = shard.get_price() if rank == 0 else shard.empty_price()
        broadcast(dist, price, 0)
        if rank != 0:
            shard.set_price(price)
    # 4. totals: each rank sums its own rows
    if hasattr(shard, "totals_dev"):
        # the partial sums stay on the device: one SUM all-reduce of three words, ONE read-back (word 2: error flags of any rank)
        t = shard.totals_dev(want_dual)
        if world > 1 or os.environ.get("TD_SHARD_FORCE_AR"):
            all_reduce(dist, t, SUM)
        vals = t.tolist()
        if vals[2] != 0:
            raise _ffi.TdError("sharded solve: a device-side consistency check failed on some rank (%d flags)" % vals[2])
        r2c = shard.row_to_col()
        return (r2c, int(vals[0]), int(vals[1])) if want_dual else (r2c, int(vals[0]))
    tot, dual = shard.totals(want_dual)
    t = shard.scalar_tensor([tot, dual])
    if world > 1:
        all_reduce(dist, t, SUM)
    r2c = shard.row_to_col()
    if want_dual:
        return r2c, int(t[0].item()), int(t[1].item())
    return r2c, int(t[0].item())


def solve_shards_in_process(shards, rounds=DEFAULT_ROUNDS, want_dual=True, blocks=True, call=None, fused_round0=True):
    """The steps of solve_sharded over ALL the shards of an instance driven by one process on one GPU (tests, per-phase
    timing: tools/r4_shard_time.py): torch.cat / torch.maximum stand in for the all-gather / the MAX all-reduce, the
    library calls are the ones a rank makes.  `call(name, shard_index, fn)` wraps every library call (default: runs it).
    Returns (row_to_col of all rows, total, dual, info)."""
    import torch
    call = call or (lambda name, k, fn: fn())
    world, n = len(shards), shards[0].n
    rps = shard_bounds(n, world, 0)[2]
    sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)
    info = {"path": "auction", "left": None}
    left = None
    can_blocks = shards[0].blocks_ok(n, world) if hasattr(shards[0], "blocks_ok") else blocks_ok(n, world)
    if blocks and hasattr(shards[0], "phase_a") and can_blocks:
        for s in shards:
            s.blocks_start(True)
        fits = [call("compress", k, lambda s=s: s.compress(1)) for k, s in enumerate(shards)]
        for k, (s, f) in enumerate(zip(shards, fits)):
            if f and s.blocks_pending():
                call("phase_a", k, s.phase_a)
        segs = [call("export", k, lambda s=s, f=f: s.state_segment(rps, f)) for k, (s, f) in enumerate(zip(shards, fits))]
        allseg = torch.cat(segs)
        sync()
        summs = [call("import", k, lambda s=s, k=k: s.state_import(world, k, rps, allseg)) for k, s in enumerate(shards)]
        assert all(x == summs[0] for x in summs), summs
        if summs[0]["fit"] and summs[0]["ran"]:
            left = summs[0]["left"]
            for s in shards:
                s.begin(summs[0]["range"])
            info["path"] = "blocks"
        else:
            for s in shards:
                s.blocks_start(False)
    if left is None:
        for s in shards:
            if hasattr(s, "fused_round0"):
                s.fused_round0(fused_round0)
        for width in (1, 2, 4):
            oks = [call("compress", k, lambda s=s: s.compress(width)) for k, s in enumerate(shards)]
            if all(oks):
                break
        masks = [s.const_mask() for s in shards]
        for m in masks[1:]:
            masks[0] += m
        sync()
        for s in shards:
            s.set_const_mask(masks[0])
        grange = max(s.range() for s in shards) if hasattr(shards[0], "range") else -1
        for k, s in enumerate(shards):
            call("begin", k, lambda s=s: s.begin(grange))
    info["left"] = left
    finisher_ran = False
    if left == 0 and summs[0]["nconst"] > 0:
        for k, s in enumerate(shards):
            call("place_const", k, s.place_const)
    if left != 0:
        keys = [s.new_keys() for s in shards]
        for rnd in range(rounds):
            for k, (s, ky) in enumerate(zip(shards, keys)):
                call("bid%d" % rnd, k, lambda s=s, ky=ky: s.bid(rnd, ky))
            red = keys[0].clone()
            for ky in keys[1:]:
                red = torch.maximum(red, ky)
            sync()
            for k, (s, ky) in enumerate(zip(shards, keys)):
                ky.copy_(red)
                sync()
                call("apply%d" % rnd, k, lambda s=s, ky=ky: s.apply(rnd, ky))
        call("finish", 0, lambda: shards[0].finish([s.cc_ref() for s in shards], rps))
        finisher_ran = True
        owner = shards[0].get_owner()
        price = shards[0].get_price() if want_dual else None
        sync()
        for s in shards[1:]:
            s.set_owner(owner)
            if want_dual:
                s.set_price(price)
    tot = dual = 0
    parts = []
    for k, s in enumerate(shards):
        t, d = call("totals", k, lambda s=s: s.totals(want_dual))
        tot += t
        dual += d
        parts.append(s.row_to_col())
    info["finisher_ran"] = finisher_ran
    return np.concatenate(parts), tot, dual, info


def assign_sharded(cost_rows, n, want_dual=False, rounds=DEFAULT_ROUNDS, dist=None):
    """Convenience entry: this rank's rows (torch CUDA int32 [nrows, n] or numpy) -> result."""
    if dist is None:
        import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    row0, nrows, _ = shard_bounds(n, world, rank)
    sh = HipShard(n, row0, nrows, cost_rows)
    try:
        return solve_sharded(sh, dist, rounds, want_dual)
    finally:
        sh.close()


def pool_fanout(k, demand, dist, children=8, finder=None, merger=None, timeout=60.0):
    """findpool.c's 8-way fan-out mapped to GPUs: child t (a first-pick-up slice, pool_n.c:243-246)
    runs on rank t % world, the lists travel to rank 0, rank 0 merges (findpool.c:73-98,166-172) and
    hands the result back.  No data-path collective: like findpool.c, which polls its children's flag
    files and gives up after 60 s (findpool.c:149-169 "ERROR: not all threads have returned results"),
    the lists and the result go through the process group's key-value store (a few KiB per rank) and
    every wait has `timeout` seconds — a rank that died or hangs makes the others raise TdError naming
    it instead of blocking in a collective for ever.
    `finder(k, demand, child)` -> records, `merger(k, n, lists)` -> records default to the GPU entry
    points; the gloo test injects host models."""
    import datetime
    import pickle
    import time
    if finder is None or merger is None:
        from . import dispatch
        finder = finder or (lambda kk, dd, child: dispatch.find_pool_n(kk, dd, child=child, children=children)[0])
        merger = merger or dispatch.merge_pools
    world, rank = dist.get_world_size(), dist.get_rank()
    from torch.distributed import distributed_c10d as c10d
    store = c10d._get_default_store()
    pool_fanout._seq = getattr(pool_fanout, "_seq", 0) + 1      # every rank makes the same calls in the same order
    key = "td_pool_fanout/%d/" % pool_fanout._seq
    mine = {t: np.asarray(finder(k, demand, t)).reshape(-1, 2 * k + 1).tolist() for t in range(children) if t % world == rank}
    store.set(key + "r%d" % rank, pickle.dumps(mine))
    limit = datetime.timedelta(seconds=float(timeout))
    if rank == 0:
        keys = [key + "r%d" % r for r in range(world)]
        try:
            store.wait(keys, limit)
        except Exception:
            missing = [r for r in range(world) if not store.check([key + "r%d" % r])]
            store.set(key + "result", pickle.dumps(("error", missing)))
            raise _ffi.TdError("pool fan-out: rank(s) %s did not deliver their slices within %.0f s (findpool.c:165-168 gives up the "
                               "same way)" % (missing, float(timeout)))
        lists = {}
        for kk in keys:
            lists.update(pickle.loads(store.get(kk)))
        out = np.asarray(merger(k, len(demand), [lists[t] for t in range(children)])).reshape(-1, 2 * k + 1).tolist()
        store.set(key + "result", pickle.dumps(("ok", out)))
        # the readers acknowledge; the last one (or, after the timeout, nobody) leaves rank 0 to delete the keys of this
        # call — a long simulation does not pile one merged list per tick up in the store
        deadline = time.monotonic() + float(timeout)
        while world > 1 and store.add(key + "ack", 0) < world - 1 and time.monotonic() < deadline:
            time.sleep(0.0005)
        for kk in keys + [key + "result", key + "ack"]:
            try:
                store.delete_key(kk)
            except Exception:
                pass
    else:
        # rank 0 may still be in its own finder when this rank is done, then waits up to `timeout` for the slowest rank
        # and merges: its answer is due within two timeouts of NOW (a wait of timeout + 5 s measured from this rank's
        # finish could expire on a healthy run)
        try:
            store.wait([key + "result"], limit + limit + datetime.timedelta(seconds=5))
        except Exception:
            raise _ffi.TdError("pool fan-out: rank 0 did not hand the merged pools back within %.0f s" % (2 * float(timeout) + 5))
        status, out = pickle.loads(store.get(key + "result"))
        store.add(key + "ack", 1)
        if status != "ok":
            raise _ffi.TdError("pool fan-out: rank(s) %s did not deliver their slices within %.0f s" % (out, float(timeout)))
    return np.asarray(out, np.int32).reshape(-1, 2 * k + 1)


# ----------------------------------------------------------------------------------------
# row-sharded LCM
# ----------------------------------------------------------------------------------------
INT64_MAX = 2**63 - 1


class HipLcmShard:
    """One rank's rows for the sharded lowest-cost method (td_lcm_shard_*)."""

    def __init__(self, n, row0, nrows, cost_rows, stop_value_on=0, stop_value=0):
        self.lib = _ffi.lib()
        self.n, self.row0, self.nrows = n, row0, nrows
        self._cost = cost_rows
        h = ctypes.c_void_p()
        _ffi.check(self.lib.td_lcm_shard_create(n, row0, nrows, _ffi.addr(cost_rows) if nrows else None, int(stop_value_on),
                                                int(stop_value), ctypes.byref(h)))
        self.h = h

    def local_min(self):
        out = (ctypes.c_int64 * 3)()
        _ffi.check(self.lib.td_lcm_shard_local_min(self.h, out))
        return (int(out[0]), int(out[1]), int(out[2]))

    def take(self, row, col):
        _ffi.check(self.lib.td_lcm_shard_take(self.h, int(row), int(col)))

    # rounds of locally dominant cells: int64 vectors of length n on `device` (torch tensors)
    device = "cuda"

    def round_colmin(self, limit, out):
        _ffi.check(self.lib.td_lcm_shard_round_colmin(self.h, int(limit), _ffi.addr(out)))
        self._fence(out)

    def round_apply(self, limit, colmin, out):
        # `colmin` comes out of torch work (torch.minimum over in-process shards, the MIN all-reduce): RCCL and torch
        # order that on torch's stream only, the library's kernel runs on its own — wait for it first (ADVICE r3:
        # without this k_lcmsh_apply could read a partially reduced vector on a multi-GPU run)
        self._fence(colmin)
        _ffi.check(self.lib.td_lcm_shard_round_apply(self.h, int(limit), _ffi.addr(colmin), _ffi.addr(out)))
        self._fence(out)

    @staticmethod
    def _fence(t):
        # the library queued the kernel that writes `t` on ITS stream; torch (element-wise min / max, the
        # collective) reads it on its own: wait for the device before torch touches the vector
        if getattr(t, "is_cuda", False):
            import torch
            torch.cuda.synchronize(t.device)

    def round_commit(self, taken):
        self._fence(taken)   # (torch.maximum / the MAX all-reduce that produced `taken`)
        _ffi.check(self.lib.td_lcm_shard_round_commit(self.h, _ffi.addr(taken)))

    def close(self):
        if self.h:
            self.lib.td_lcm_shard_destroy(self.h)
            self.h = None


LCM_NONE_MIN = 2**63 - 1     # round_colmin: column without a candidate
LCM_NONE_MAX = -2**63        # round_apply: column not taken in this round


def lcm_sharded(shards, dist, n, mask, threshold=-1, stop_value_on=0, stop_value=0, stop_size=-1, sum_below=INT64_MAX,
                max_pairs=None, by_pick=False, force_collectives=False):
    """The lowest-cost method over row shards, stop rules exactly as td_lcm / k_lcm_loop
    (greedy_opt.py:61-82, heuristic.py:24-33, Simulator.java:523-549).  `shards`: the shard(s) this rank
    owns; `dist`: torch.distributed or None for a single process.

    ROUNDS of locally dominant cells: a live cell that is the first minimum of its row and
    of its column under the reference's order (value, row, column) is taken by the sequential greedy before
    anything else of its row or column, so every round takes all of them at once — two all-reduces of n
    int64 keys per round (MIN of the column minima, MAX of the taken keys) and a few tens of rounds per
    tick instead of one exchange per pick (~700).  The picks, sorted by the reference's order, ARE the
    sequential greedy's picks in its order; the stop rules are then applied to that sequence.  The vectors
    live on the shards' device, so the collectives run on whatever backend owns it (RCCL for HIP shards).
    by_pick=True keeps the one-exchange-per-pick driver (shards with local_min / take only);
    force_collectives=True issues the collectives with one rank too (a one-GPU test of the RCCL path).
    Returns (total, rows, cols, last_min), identical on every rank."""
    if by_pick or not hasattr(shards[0], "round_colmin"):
        return _lcm_sharded_by_pick(shards, dist, n, mask, threshold, stop_value_on, stop_value, stop_size, sum_below, max_pairs,
                                    force_collectives)
    import torch
    world = dist.get_world_size() if dist is not None else 1
    coll = dist is not None and (world > 1 or force_collectives)
    dev = getattr(shards[0], "device", "cpu")
    cand_limit = int(stop_value) if stop_value_on else LCM_NONE_MIN     # Simulator.java:529-537: cells >= big_cost are never looked at
    limit = min(cand_limit, int(mask))                                   # only masked-valued cells remain -> stop
    if threshold >= 0:
        limit = min(limit, int(threshold) + 1)                           # greedy_opt.py:68-69: stop when the minimum is > threshold
    colmin = torch.empty(n, dtype=torch.int64, device=dev)
    taken = torch.empty(n, dtype=torch.int64, device=dev)
    tmp = torch.empty(n, dtype=torch.int64, device=dev) if len(shards) > 1 else None
    picked = torch.full((n,), LCM_NONE_MAX, dtype=torch.int64, device=dev)

    def col_minima(lim):
        shards[0].round_colmin(lim, colmin)
        for sh in shards[1:]:
            sh.round_colmin(lim, tmp)
            torch.minimum(colmin, tmp, out=colmin)
        if coll:
            dist.all_reduce(colmin, op=dist.ReduceOp.MIN)

    rounds = 0
    while True:
        col_minima(limit)
        shards[0].round_apply(limit, colmin, taken)
        for sh in shards[1:]:
            sh.round_apply(limit, colmin, tmp)
            torch.maximum(taken, tmp, out=taken)
        if coll:
            dist.all_reduce(taken, op=dist.ReduceOp.MAX)
        if not bool((taken != LCM_NONE_MAX).any()):
            break
        rounds += 1
        for sh in shards:
            sh.round_commit(taken)
        torch.maximum(picked, taken, out=picked)
    keys = picked.cpu().numpy()
    cols_all = np.nonzero(keys != LCM_NONE_MAX)[0]
    vals = keys[cols_all] >> 32
    rows_all = keys[cols_all] & 0xFFFFFFFF
    order = np.lexsort((cols_all, rows_all, vals))      # the reference's order: value, then row, then column
    iters = n if max_pairs is None else min(n, max_pairs)
    pairs_r, pairs_c = [], []
    total, size = 0, n
    last_min = None
    stopped = False
    for k in order[:iters]:
        v = int(vals[k])
        last_min = v
        pairs_r.append(int(rows_all[k]))
        pairs_c.append(int(cols_all[k]))
        if v < sum_below:
            total += v
        size -= 1
        if stop_size >= 0 and size == stop_size:   # Simulator.java:544-545
            stopped = True
            break
    if not stopped and len(pairs_r) < iters:
        # the loop looks once more: the smallest live cell the scan still sees decides what last_min reads
        col_minima(cand_limit)
        m = int(colmin.min()) if n else LCM_NONE_MIN
        last_min = (int(stop_value) if stop_value_on else int(mask)) if m == LCM_NONE_MIN else (m >> 32)
    elif last_min is None:
        last_min = int(stop_value)
    lcm_sharded.last_rounds = rounds
    return total, pairs_r, pairs_c, last_min


def _lcm_sharded_by_pick(shards, dist, n, mask, threshold=-1, stop_value_on=0, stop_value=0, stop_size=-1, sum_below=INT64_MAX,
                         max_pairs=None, force_collectives=False):
    """One exchange per pick (the first sharded driver; kept as the comparator of the rounds)."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    dev = getattr(shards[0], "device", "cpu")
    pairs_r, pairs_c = [], []
    total, size = 0, n
    last_min = stop_value
    iters = n if max_pairs is None else min(n, max_pairs)
    for _ in range(iters):
        best = min((s.local_min() for s in shards), default=(INT64_MAX, -1, -1))
        if dist is not None and (world > 1 or force_collectives):
            # on the shards' device: an RCCL-only process group has no backend for CPU tensors
            t = torch.tensor(best, dtype=torch.int64, device=dev)
            out = [torch.empty(3, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(out, t)
            best = min(tuple(int(x) for x in o.tolist()) for o in out)
        v, r, c = best
        if v == INT64_MAX:                      # nothing left to look at
            last_min = stop_value if stop_value_on else mask
            break
        last_min = v
        if threshold >= 0 and v > threshold:    # greedy_opt.py:68-69
            break
        if stop_value_on and v >= stop_value:   # Simulator.java:538
            break
        if v >= mask:                           # only masked-valued cells remain
            break
        pairs_r.append(r)
        pairs_c.append(c)
        if v < sum_below:
            total += v
        size -= 1
        for s in shards:
            s.take(r, c)
        if stop_size >= 0 and size == stop_size:   # Simulator.java:544-545
            break
    return total, pairs_r, pairs_c, last_min
