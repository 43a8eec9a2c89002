"""numpy model of ONE shard of the row-sharded auction (test infrastructure).

Implements the interface `taxidispatcher_amd.sharded.solve_sharded` drives (the product
implementation is HipShard, which needs a GPU) so that the collective logic — shard bounds, one
MAX all-reduce of packed keys per round, finisher hand-off, owner broadcast, SUM of totals — runs
under gloo on CPU.  Same packed key format as the device: (price << 20) | (global_row + 1).
"""
import numpy as np
import torch

ROW_BITS = 20
LIMITS = {1: 254, 2: 65534, 4: 2**32 - 2}


class ModelShard:
    def __init__(self, n, row0, nrows, cost_rows):
        self.n, self.row0, self.nrows = n, row0, nrows
        self.cost = np.asarray(cost_rows, dtype=np.int64).reshape(nrows, n)
        self.width = 0

    def compress(self, width):
        self._ready = False
        if self.nrows == 0:
            self.width = width
            self.rowmin = np.zeros(0, np.int64)
            self.cc = np.zeros((0, self.n), np.int64)
            return True
        rng = self.cost.max(1) - self.cost.min(1)
        if (rng > LIMITS[width]).any():
            return False
        self.width = width
        self.rowmin = self.cost.min(1)
        self.cc = self.cost - self.rowmin[:, None]
        return True

    # -- numpy model of the block-local start (csrc/td_blocks.h): phase A on the shard's own diagonal blocks (zero cells
    #    only, no price moves), then ONE exchange of segments [16 header words | owners of the column slice | constant-row
    #    flags].  Tie-breaks differ from the device's; the driver logic and the exactness argument are what is modelled.
    def blocks_ok(self, n, world):
        from taxidispatcher_amd import sharded
        return sharded.BLOCKS % world == 0 and n % sharded.BLOCKS == 0 and n >= 2 * sharded.BLOCKS

    def blocks_start(self, on=True):
        self._blocks = bool(on)
        self._ready = False

    def blocks_pending(self):
        return bool(getattr(self, "_blocks", False)) and self.width == 1

    def phase_a(self):
        from taxidispatcher_amd import sharded
        n = self.n
        rpb = n // sharded.BLOCKS
        self.p = np.zeros(n, np.int64)
        self.owner = np.full(n, -1, np.int64)
        self.r2c = np.full(self.nrows, -1, np.int64)
        self.r2c[self.cost.max(1) == self.cost.min(1)] = -2   # constant rows are deferred
        for lb in range(self.nrows // rpb):
            c0 = self.row0 + lb * rpb               # the block's columns = its rows' global ids
            rows = range(lb * rpb, (lb + 1) * rpb)
            for lr in rows:                          # greedy on the zero cells
                if self.r2c[lr] != -1:
                    continue
                z = np.nonzero((self.cc[lr, c0:c0 + rpb] == 0) & (self.owner[c0:c0 + rpb] < 0))[0]
                if len(z):
                    j = c0 + int(z[(lr * 7) % len(z)])
                    self.owner[j] = self.row0 + lr
                    self.r2c[lr] = j
            for lr in rows:                          # two hops: free row -> zero column -> its owner -> free zero column
                if self.r2c[lr] != -1:
                    continue
                done = False
                for j in c0 + np.nonzero(self.cc[lr, c0:c0 + rpb] == 0)[0]:
                    r = int(self.owner[j]) - self.row0
                    if r < 0:
                        continue
                    esc = np.nonzero((self.cc[r, c0:c0 + rpb] == 0) & (self.owner[c0:c0 + rpb] < 0))[0]
                    if len(esc):
                        jp = c0 + int(esc[0])
                        self.owner[jp] = self.row0 + r
                        self.r2c[r] = jp
                        self.owner[j] = self.row0 + lr
                        self.r2c[lr] = j
                        done = True
                        break
                _ = done
        self._ready = True

    def place_const(self):
        crow, fcol = np.nonzero(self.cmask)[0], np.nonzero(self.owner < 0)[0]
        assert len(crow) == len(fcol)
        owner = self.owner.copy()
        owner[fcol] = crow
        self.set_owner(torch.from_numpy(owner.astype(np.int32)))

    def state_segment(self, rps, fits):
        seg = np.zeros(16 + 2 * rps, np.int64)
        ran = bool(getattr(self, "_ready", False))
        rng = int((self.cost.max(1) - self.cost.min(1)).max()) if self.nrows else 0
        seg[0:6] = [1 if fits else 0, 1 if ran else 0, int((self.r2c == -1).sum()) if ran else -1,
                    int((self.cost.max(1) == self.cost.min(1)).sum()) if self.nrows else 0, rng & 0xFFFFFFFF, rng >> 32]
        seg[16:16 + rps] = -1
        if ran:
            seg[16:16 + self.nrows] = self.owner[self.row0:self.row0 + self.nrows]
        if self.nrows:
            seg[16 + rps:16 + rps + self.nrows] = (self.cost.max(1) == self.cost.min(1))
        return torch.from_numpy(seg.astype(np.int32))

    def state_import(self, world, rank, rps, allseg):
        a = allseg.numpy().astype(np.int64).reshape(world, 16 + 2 * rps)
        n = self.n
        cm = np.zeros(n, bool)
        for r in range(world):
            lo, hi = r * rps, min(n, (r + 1) * rps)
            if r != rank and getattr(self, "_ready", False):
                self.owner[lo:hi] = a[r, 16:16 + hi - lo]
            cm[lo:hi] = a[r, 16 + rps:16 + rps + hi - lo] != 0
        self.cmask = cm
        left = sum(int(x) if x >= 0 else n for x in a[:, 2])
        return {"fit": bool(a[:, 0].all()), "ran": bool(a[:, 1].all()), "left": left, "nconst": int(a[:, 3].sum()),
                "range": int(((a[:, 5] << 32) | (a[:, 4] & 0xFFFFFFFF)).max())}

    def begin(self, global_range=-1):
        if getattr(self, "_ready", False):   # phase A has initialised the state
            return
        self.p = np.zeros(self.n, np.int64)
        self.owner = np.full(self.n, -1, np.int64)
        self.r2c = np.full(self.nrows, -1, np.int64)
        if getattr(self, "cmask", None) is not None:
            self.r2c[self.cmask[self.row0:self.row0 + self.nrows]] = -2   # deferred: never bid, never searched

    # constant rows sit out the solve (model of td_shard_const_rows)
    def const_mask(self):
        m = torch.zeros(self.n, dtype=torch.int32)
        if self.nrows:
            m[self.row0:self.row0 + self.nrows] = torch.from_numpy((self.cost.max(1) == self.cost.min(1)).astype(np.int32))
        return m

    def set_const_mask(self, mask):
        self.cmask = mask.numpy().astype(bool)

    def new_keys(self):
        return torch.zeros(self.n + 16, dtype=torch.int64)

    def bid(self, rnd, keys):
        k = keys.numpy()
        owned = (self.owner >= 0).astype(np.int64)
        for lr in np.nonzero(self.r2c == -1)[0]:
            key = 2 * (self.cc[lr] + self.p) + owned
            rot = (int(lr + self.row0) * 7919 + rnd * 104729) % self.n
            order = np.roll(np.arange(self.n), -rot)
            j1 = int(order[np.argmin(key[order])])
            k1 = key[j1]
            rest = np.delete(key, j1)
            k2 = rest.min() if rest.size else k1
            inc = (k2 >> 1) - (k1 >> 1)
            if (k1 & 1) and inc == 0 and rnd == 0:
                continue
            newp = self.p[j1] + inc
            k[j1] = max(k[j1], (int(newp) << ROW_BITS) | (int(lr) + self.row0 + 1))

    def apply(self, rnd, keys):
        k = keys.numpy()
        for j in np.nonzero(k[:self.n])[0]:
            row = int(k[j] & ((1 << ROW_BITS) - 1)) - 1
            newp = int(k[j] >> ROW_BITS)
            old = int(self.owner[j])
            if self.row0 <= old < self.row0 + self.nrows:
                self.r2c[old - self.row0] = -1
            self.owner[j] = row
            if self.row0 <= row < self.row0 + self.nrows:
                self.r2c[row - self.row0] = j
            self.p[j] = newp
        k[:] = 0

    def cc_ref(self):
        return self.cc_copy()

    def cc_copy(self):
        return torch.from_numpy(np.ascontiguousarray(self.cc).view(np.uint8).reshape(-1).copy())

    def finish(self, refs, rps):
        n = self.n
        rows = []
        for k, t in enumerate(refs):
            nr = max(0, min(rps, n - k * rps))
            rows.append(t.numpy()[:nr * n * 8].view(np.int64).reshape(nr, n))
        cc = np.concatenate(rows, 0)
        p, owner = self.p, self.owner
        r2c = np.full(n, -1, np.int64)
        r2c[owner[owner >= 0]] = np.nonzero(owner >= 0)[0]
        cmask = getattr(self, "cmask", None)
        if cmask is None:
            cmask = np.zeros(n, bool)
        for f in np.nonzero((r2c < 0) & ~cmask)[0]:
            d = cc[f] + p
            pred = np.full(n, f, np.int64)
            scanned = np.zeros(n, bool)
            while True:
                key = np.where(scanned, np.iinfo(np.int64).max, 2 * d + (owner >= 0))
                j = int(np.argmin(key))
                if owner[j] < 0:
                    break
                scanned[j] = True
                o = int(owner[j])
                h = d[j] + cc[o] + p - (cc[o, j] + p[j])
                upd = (h < d) & ~scanned
                d[upd] = h[upd]
                pred[upd] = o
            p[scanned] += d[j] - d[scanned]
            while True:
                i = int(pred[j])
                owner[j] = i
                j, r2c[i] = int(r2c[i]), j
                if i == f:
                    break
        # the deferred constant rows take the columns nobody owns, k-th row <- k-th free column
        crow, fcol = np.nonzero(cmask)[0], np.nonzero(owner < 0)[0]
        assert len(crow) == len(fcol)
        owner[fcol] = crow
        self.set_owner(torch.from_numpy(owner.astype(np.int32)))

    def get_owner(self):
        return torch.from_numpy(self.owner.astype(np.int32))

    def empty_owner(self):
        return torch.empty(self.n, dtype=torch.int32)

    def set_owner(self, owner):
        self.owner = owner.numpy().astype(np.int64)
        self.r2c[:] = -1
        for j, o in enumerate(self.owner):
            if self.row0 <= o < self.row0 + self.nrows:
                self.r2c[o - self.row0] = j

    def get_price(self):
        return torch.from_numpy(self.p.copy())

    def empty_price(self):
        return torch.empty(self.n, dtype=torch.int64)

    def set_price(self, price):
        self.p = price.numpy().copy()

    def totals(self, want_dual):
        tot = int(self.cost[np.arange(self.nrows), self.r2c].sum()) if self.nrows else 0
        dual = 0
        if want_dual:
            dual = int((self.rowmin + (self.cc + self.p).min(1)).sum()) if self.nrows else 0
            if self.row0 == 0:
                dual -= int(self.p.sum())
        return tot, dual

    def row_to_col(self):
        return self.r2c.astype(np.int32)

    def scalar_tensor(self, values, dtype=None):
        return torch.tensor(values, dtype=dtype or torch.int64)

    # -- numpy model of td_line_shard_* (the sorted matching over row shards): same four phases and exchanged
    #    segments as the device path; the replicated part (sort, prices) is recomputed from the summed segments
    def line_ws(self):
        n = self.n
        return torch.zeros(16 + 2 * n + n + 2 * n + 2, dtype=torch.int64)

    def line_phase(self, phase, ws):
        n, c, w = self.n, self.cost, ws.numpy()
        A, RK, FB = 16, 16 + 2 * n, 16 + 3 * n
        TOT, FAIL = FB + 2 * n, FB + 2 * n + 1
        if phase == 0:
            w[:] = 0
            if self.row0 == 0 and self.nrows > 0:
                d = np.abs(c[0] - c[0, 0])
                q = int(np.argmax(d))
                e = np.abs(c[:, 0] - c[0, 0]) + np.abs(c[:, q] - c[0, q])
                i2 = int(np.argmax(e))
                k1 = c[0, 0] ** 2 - c[0, q] ** 2
                k2 = c[i2, 0] ** 2 - c[i2, q] ** 2
                ok = bool(d[q] > 0 and e[i2] > 0)
                if ok and self.nrows >= 2:   # the probe's first test on rows 0 and 1 (k_lsh_anchor)
                    m = min(n, 1024)
                    x, y = c[0, :m], c[1, :m]

                    def span_ok(D):
                        return bool(((x + y == D) | (np.abs(x - y) == D)).all())
                    ok = span_ok(np.abs(x - y).max()) or span_ok((x + y).min())
                w[0:4] = [q, i2, int(k2 < k1), int(ok)]
                w[A:A + n] = c[0]
                w[A + n:A + 2 * n] = c[i2]
            return 0, A + 2 * n
        if phase == 1:
            q = int(w[0])
            if self.nrows:
                w[RK + self.row0:RK + self.row0 + self.nrows] = c[:, 0] ** 2 - c[:, q] ** 2 + (1 << 62)
            return RK, n
        if phase == 2:
            ck = w[A:A + n] ** 2 - w[A + n:A + 2 * n] ** 2
            if w[2]:
                ck = -ck
            self.sig = np.argsort(w[RK:RK + n], kind="stable")
            self.tau = np.argsort(ck, kind="stable")
            inv = np.empty(n, np.int64)
            inv[self.sig] = np.arange(n)
            for t in range(self.nrows):
                k = int(inv[self.row0 + t])
                ckk = c[t, self.tau[k]]
                if k + 1 < n:
                    w[FB + k] = c[t, self.tau[k + 1]] - ckk
                if k > 0:
                    w[FB + n + k - 1] = c[t, self.tau[k - 1]] - ckk
                w[TOT] += ckk
            self.line_r2c = self.tau[inv[self.row0:self.row0 + self.nrows]]
            return FB, 2 * n + 1
        if phase == 3:
            F, B = w[FB:FB + n], w[FB + n:FB + 2 * n]
            bad = (w[3] == 0) or bool((F[:n - 1] + B[:n - 1] < 0).any())
            v = np.zeros(n, np.int64)
            v[self.tau] = np.concatenate(([0], np.cumsum(F[:n - 1])))
            if not bad and self.nrows:
                red = c - v[None, :]
                tight = red[np.arange(self.nrows), self.line_r2c]
                bad = bool((red.min(1) != tight).any())
            w[FAIL] = int(bad)
            return FAIL, 1
        raise ValueError(phase)

    def line_plausible(self, ws):
        return int(ws.numpy()[3]) != 0

    def line_result(self, ws):
        n = self.n
        w = ws.numpy()
        return bool(w[16 + 5 * n + 1] == 0), int(w[16 + 5 * n]), self.line_r2c.astype(np.int32)

    def close(self):
        pass


class ModelLcmShard:
    """numpy model of td_lcm_shard_* (one rank's rows of the sharded lowest-cost method)."""

    def __init__(self, n, row0, nrows, cost_rows, stop_value_on=0, stop_value=0):
        self.n, self.row0, self.nrows = n, row0, nrows
        self.c = np.asarray(cost_rows, np.int64).reshape(nrows, n).copy()
        self.limit = stop_value if stop_value_on else None
        self.row_live = np.ones(nrows, bool)
        self.col_live = np.ones(n, bool)

    def local_min(self):
        best = (2**63 - 1, -1, -1)
        for i in np.nonzero(self.row_live)[0]:
            row = np.where(self.col_live, self.c[i], 2**62)
            if self.limit is not None:
                row = np.where(row < self.limit, row, 2**62)
            j = int(np.argmin(row))                     # first minimum: smallest column
            if row[j] < 2**62:
                best = min(best, (int(row[j]), self.row0 + int(i), j))
        return best

    def take(self, row, col):
        if self.row0 <= row < self.row0 + self.nrows:
            self.row_live[row - self.row0] = False
        self.col_live[col] = False

    # ---- rounds of locally dominant cells (td_lcm_shard_round_*): torch int64 vectors on the CPU
    device = "cpu"
    NONE_MIN, NONE_MAX = 2**63 - 1, -2**63

    def _row_first_min(self, i):
        row = np.where(self.col_live, self.c[i], 2**62)
        if self.limit is not None:
            row = np.where(row < self.limit, row, 2**62)
        j = int(np.argmin(row))
        return (int(row[j]), j) if row[j] < 2**62 else None

    def round_colmin(self, limit, out):
        res = np.full(self.n, self.NONE_MIN, np.int64)
        for i in np.nonzero(self.row_live)[0]:
            if self._row_first_min(i) is None:      # no candidate left in the row: it sits out (like rowbest == INF)
                continue
            ok = self.col_live & (self.c[i] < limit)
            keys = (self.c[i] << 32) | (self.row0 + int(i))
            res = np.where(ok, np.minimum(res, keys), res)
        out.copy_(__import__("torch").from_numpy(res))

    def round_apply(self, limit, colmin, out):
        cm = colmin.numpy()
        res = np.full(self.n, self.NONE_MAX, np.int64)
        for i in np.nonzero(self.row_live)[0]:
            fm = self._row_first_min(i)
            if fm is None or fm[0] >= limit:
                continue
            key = (fm[0] << 32) | (self.row0 + int(i))
            if cm[fm[1]] == key:
                res[fm[1]] = key
        out.copy_(__import__("torch").from_numpy(res))

    def round_commit(self, taken):
        t = taken.numpy()
        for c in np.nonzero(t != self.NONE_MAX)[0]:
            self.take(int(t[c] & 0xFFFFFFFF), int(c))
