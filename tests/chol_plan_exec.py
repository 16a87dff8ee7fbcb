"""TEST INFRASTRUCTURE — a numpy executor of the symbolic sparse-Cholesky plan (semantic_slam_amd/csrc/chol_plan.hpp).

It walks the exported plan records (pieces, internal levels, work items, update records, multi-block reductions) exactly in
the order the HIP kernels of sslam_chol.hip do — external updates from the finished factor, internal levels out of a
piece-local mirror, partial tiles summed in item order — but with dense numpy block arithmetic.  Run on a CPU-only box it
pins the *plan* (offsets, piece cuts, item lists) against dense linear algebra before any GPU time is spent."""
import ctypes as C

import numpy as np

COL = np.dtype([(n, "<i4") for n in ("xoff", "yoff", "dim", "graph", "b0", "nb", "nbi", "base", "f0", "f1", "piece", "ilevel")])
BLK = np.dtype([(n, "<i4") for n in ("off", "src", "xoff_row", "yoff_row", "coldiag", "colyoff", "info", "as0")])
ASRC = np.dtype([("uoff", "<i4"), ("uyoff", "<i4")])
FWD = np.dtype([("off", "<i4"), ("yoff", "<i4")])
UITEM = np.dtype([(n, "<i4") for n in ("u0", "n", "uoff", "flags", "s0", "ns", "uyoff", "pad")])
UMB = np.dtype([(n, "<i4") for n in ("uoff", "ps0", "n", "info", "s0", "ns", "uyoff", "pad")])
UPD = np.dtype([("ab", "<u4"), ("xk", "<u4")])     # 8-byte update record: piece-local operand offsets | local y offset / target + flags (chol_plan.hpp UpdMeta)
ITEM = np.dtype([(n, "<i4") for n in ("u0", "n", "tloff", "flags")])
MB = np.dtype([(n, "<i4") for n in ("tloff", "ps0", "n", "info")])
RCOL = np.dtype([("u0", "<i4"), ("n", "<i4")])
ILV = np.dtype([(n, "<i4") for n in ("c0", "c1", "b0", "b1", "it0", "it1", "mb0", "mb1")])
PIECE = np.dtype([(n, "<i4") for n in ("graph", "c0", "nc", "b0", "nb", "lbase", "lsize", "y0", "ysize", "ilv0", "nilv", "iit0", "nit_i",
                                       "iu0", "nu_i", "imb0", "nimb", "as0", "nas", "uit0", "nuit", "umb0", "numb",
                                       "uu0", "nuu", "us0", "nus", "n36", "n18", "nint", "pad3", "pad4", "nu4", "nu2", "nu1", "pad5")])
K_DI6, K_DK6, K_DIAG, K_DJ6 = 1 << 28, 1 << 29, 1 << 30, 1 << 31


def upd_fields(r):
    """(local offset of L_ik, of L_jk, local y offset of column k, flags) of a target-major update record"""
    ab, xk = int(r["ab"]), int(r["xk"])
    return ab & 0xFFFF, ab >> 16, xk & 0x1FFF, xk


def upd_fields_right(r):
    """(local offset of L_ik, of L_jk, local offset of the target block, local y offset of its column, flags) of a right-looking record"""
    ab, xk = int(r["ab"]), int(r["xk"])
    return ab & 0xFFFF, ab >> 16, xk & 0xFFFF, (xk >> 16) & 0xFFF, xk
B_FMT, B_DIAG, B_ROWIN = 1 << 8, 1 << 9, 1 << 10


class Plan:
    def __init__(self, lib, graphs):
        arr = (C.c_void_p * len(graphs))(*[g._h for g in graphs])
        self._lib = lib
        self._h = lib.sslam_debug_plan_create(arr, len(graphs))
        if not self._h:
            raise RuntimeError(lib.sslam_last_error().decode())
        g = self._get
        self.col, self.blk, self.upd = g("col", COL), g("blk", BLK), g("upd", UPD)
        self.item, self.mb, self.ilv, self.piece = g("item", ITEM), g("mb", MB), g("ilv", ILV), g("piece", PIECE)
        self.asrc, self.usrc, self.fwd, self.uitem, self.umb = g("asrc", ASRC), g("usrc", ASRC), g("fwd", FWD), g("uitem", UITEM), g("umb", UMB)
        self.rcol, self.rupd = g("rcol", RCOL), g("rupd", UPD)     # right-looking update lists of the tail pieces
        for n in ("lvl_ptr", "lvl_cols", "plv_ptr", "plv_pieces", "tail_ptr", "tail_pieces", "plv_lds_f", "plv_lds_b", "plv_nt", "plv_cls", "ppoff", "plblk", "scalars"):
            setattr(self, n, g(n, np.int32))
        s = self.scalars
        (self.ncol, self.nlevels, self.dim, self.B, self.npiece, self.lnz, self.tail_lds_f, self.tail_lds_b, self.nt_leaf, self.nt_tail,
         self.h_total, self.nPr, self.nLr, self.hll_base, self.hpp_off_base, self.hpl_base, self.unz) = [int(v) for v in s[:17]]
        lib.sslam_debug_plan_destroy(self._h)
        self._h = None

    def _get(self, name, dtype):
        n = self._lib.sslam_debug_plan_array(self._h, name.encode(), None, 0)
        if n < 0:
            raise RuntimeError(self._lib.sslam_last_error().decode())
        buf = np.zeros(max(n, 1), np.uint8)
        self._lib.sslam_debug_plan_array(self._h, name.encode(), buf.ctypes.data, n)
        return buf[:n].view(dtype).copy()

    # ---- piece order of the device: launches by depth, then every graph's tail
    def piece_order(self):
        order = [int(p) for p in self.plv_pieces]
        order += [int(p) for p in self.tail_pieces]
        return order

    # ---- device layout of H from a dense matrix in internal row order (pose rows 6 each, then landmark rows 3 each)
    def pack_H(self, Hd):
        nPr, nLr = self.nPr, self.nLr
        out = np.zeros(self.h_total)
        xo = lambda r: 6 * r if r < nPr else 6 * nPr + 3 * (r - nPr)
        for r in range(nPr):
            out[36 * r:36 * r + 36] = Hd[6 * r:6 * r + 6, 6 * r:6 * r + 6].ravel()
        for l in range(nLr):
            o = xo(nPr + l)
            out[self.hll_base + 9 * l:self.hll_base + 9 * l + 9] = Hd[o:o + 3, o:o + 3].ravel()
        pp = self.ppoff.reshape(-1, 2)
        for i, (a, c) in enumerate(pp):
            out[self.hpp_off_base + 36 * i:self.hpp_off_base + 36 * i + 36] = Hd[6 * a:6 * a + 6, 6 * c:6 * c + 6].ravel()
        pl = self.plblk.reshape(-1, 2)
        for i, (rp, rl) in enumerate(pl):
            o = xo(nPr + rl)
            out[self.hpl_base + 18 * i:self.hpl_base + 18 * i + 18] = Hd[6 * rp:6 * rp + 6, o:o + 3].ravel()
        return out

    # ---- factorisation + fused forward substitution, piece by piece
    def factor(self, Hdev, bvec, lam, right=False):
        """right: the tail pieces apply their internal updates column by column (rcol / rupd: what k_chol_tail does by default)
        instead of target by target (item / upd / mb)"""
        Lval = np.zeros(self.lnz + 64)
        Uval = np.full(self.unz + 64, np.nan)      # every update-matrix entry must be written before it is read
        y = np.zeros(self.dim + 8)
        ok = True
        for p in self.piece_order():   # PieceMeta.pad5: class of the piece (0 leaf, 1 mid, 2 tail); mid and tail pieces carry the right-looking lists
            ok &= self._factor_piece(self.piece[p], Hdev, bvec, lam, Lval, Uval, y, right=right and int(self.piece[p]["pad5"]) >= 1)
        self.Uval = Uval
        return Lval, y, ok

    def _tile_sum(self, upd, Ls, Ys, lofs, yofs, di, dj, diag):
        acc = np.zeros((di, dj)); accy = np.zeros(di)
        for r in upd:   # offsets are piece-local (lofs / yofs: the piece's base, kept for the callers' signatures)
            ua, ub, yk, pk = upd_fields(r)
            dk = 6 if pk & K_DK6 else 3
            assert bool(pk & K_DI6) == (di == 6) and bool(pk & K_DJ6) == (dj == 6) and bool(pk & K_DIAG) == diag
            A = Ls[ua:ua + di * dk].reshape(di, dk)
            Bm = Ls[ub:ub + dj * dk].reshape(dj, dk)
            acc += A @ Bm.T
            if diag:
                accy += A @ Ys[yk:yk + dk]
        return acc, accy

    def _run_items(self, items, upd, Ls, Ys, lofs, yofs, smL, smY, part):
        for im in items:
            u = upd[im["u0"]:im["u0"] + im["n"]]
            assert len(u) == im["n"] and im["n"] > 0
            pk0 = int(u[0]["xk"])
            di = 6 if pk0 & K_DI6 else 3
            dj = 6 if pk0 & K_DJ6 else 3
            diag = bool(pk0 & K_DIAG)
            acc, accy = self._tile_sum(u, Ls, Ys, lofs, yofs, di, dj, diag)
            fl = int(im["flags"])
            if fl & 1:
                smL[im["tloff"]:im["tloff"] + di * dj] -= acc.ravel()
                if diag:
                    yl = fl >> 12
                    smY[yl:yl + dj] -= accy
            else:
                part[(fl >> 1) & 0x7FF] = (acc, accy)

    def _reduce(self, mbs, smL, smY, part):
        for mm in mbs:
            di, dj = int(mm["info"]) & 15, (int(mm["info"]) >> 4) & 15
            acc = np.zeros((di, dj)); accy = np.zeros(di)
            for q in range(mm["n"]):
                a, ay = part[mm["ps0"] + q]
                acc += a; accy += ay
            smL[mm["tloff"]:mm["tloff"] + di * dj] -= acc.ravel()
            if int(mm["info"]) & B_DIAG:
                yl = int(mm["info"]) >> 12
                smY[yl:yl + dj] -= accy

    def _factor_piece(self, pm, Hdev, bvec, lam, Lval, Uval, y, right=False):
        lbase, y0 = int(pm["lbase"]), int(pm["y0"])
        smL = np.zeros(pm["lsize"]); smY = np.zeros(pm["ysize"])
        sAsm = self.asrc[pm["as0"]:pm["as0"] + pm["nas"]]
        # gather: A + lambda I, rhs, minus the children's update-matrix blocks
        for bm in self.blk[pm["b0"]:pm["b0"] + pm["nb"]]:
            info = int(bm["info"])
            di, dj, nas = info & 15, (info >> 4) & 15, (info >> 16) & 255
            v = np.zeros((di, dj))
            if bm["src"] >= 0:
                raw = Hdev[bm["src"]:bm["src"] + di * dj]
                v = raw.reshape(dj, di).T.copy() if info & B_FMT else raw.reshape(di, dj).copy()
            rhs = None
            if info & B_DIAG:
                v += lam * np.eye(dj)
                assert bm["off"] == bm["coldiag"]
                rhs = bvec[bm["xoff_row"]:bm["xoff_row"] + dj].copy()
            for a in sAsm[bm["as0"]:bm["as0"] + nas]:
                blk = Uval[a["uoff"]:a["uoff"] + di * dj]
                assert not np.isnan(blk).any()
                v -= blk.reshape(di, dj)
                if a["uyoff"] >= 0:
                    assert info & B_DIAG
                    uy = Uval[a["uyoff"]:a["uyoff"] + dj]
                    assert not np.isnan(uy).any()
                    rhs -= uy
            if info & B_DIAG:
                smY[bm["colyoff"] - y0:bm["colyoff"] - y0 + dj] = rhs
            smL[bm["off"] - lbase:bm["off"] - lbase + di * dj] = v.ravel()
        sUpd = self.upd[pm["iu0"]:pm["iu0"] + pm["nu_i"]]          # the LDS copies the kernel makes
        sMb = self.mb[pm["imb0"]:pm["imb0"] + pm["nimb"]]
        ok = True
        for lv in self.ilv[pm["ilv0"]:pm["ilv0"] + pm["nilv"]]:
            part = {}
            if not right:
                self._run_items(self.item[pm["iit0"] + lv["it0"]:pm["iit0"] + lv["it1"]], sUpd, smL, smY, lbase, y0, smL, smY, part)
                self._reduce(sMb[lv["mb0"]:lv["mb1"]], smL, smY, part)
            for cm in self.col[lv["c0"]:lv["c1"]]:
                d = int(cm["dim"]); o = cm["base"] - lbase; yl = cm["yoff"] - y0
                S = smL[o:o + d * d].reshape(d, d)
                S = np.tril(S) + np.tril(S, -1).T
                try:
                    Lj = np.linalg.cholesky(S)
                except np.linalg.LinAlgError:
                    ok = False
                    Lj = np.eye(d)
                smL[o:o + d * d] = Lj.ravel()
                smY[yl:yl + d] = np.linalg.solve(Lj, smY[yl:yl + d])
            for bm in self.blk[lv["b0"]:lv["b1"]]:
                info = int(bm["info"])
                if info & B_DIAG:
                    continue
                di, dj = info & 15, (info >> 4) & 15
                o = bm["off"] - lbase; od = bm["coldiag"] - lbase
                Lj = smL[od:od + dj * dj].reshape(dj, dj)
                Vb = smL[o:o + di * dj].reshape(di, dj)
                smL[o:o + di * dj] = np.linalg.solve(Lj, Vb.T).T.ravel()
            if right:   # the finished columns update every later block of the piece
                for c in range(int(lv["c0"]), int(lv["c1"])):
                    rc = self.rcol[c]
                    cm = self.col[c]
                    dk = int(cm["dim"]); yk = int(cm["yoff"]) - y0
                    recs = self.rupd[int(pm["pad3"]) + int(rc["u0"]):int(pm["pad3"]) + int(rc["u0"]) + int(rc["n"])]
                    assert len(recs) == rc["n"]
                    for r in recs:
                        ua, ub, tl, yl, pk = upd_fields_right(r)
                        di = 6 if pk & K_DI6 else 3
                        dj = 6 if pk & K_DJ6 else 3
                        assert (6 if pk & K_DK6 else 3) == dk
                        A = smL[ua:ua + di * dk].reshape(di, dk)
                        Bm = smL[ub:ub + dj * dk].reshape(dj, dk)
                        smL[tl:tl + di * dj] -= (A @ Bm.T).ravel()
                        if pk & K_DIAG:
                            assert ua == ub
                            smY[yl:yl + dj] -= A @ smY[yk:yk + dk]
        # update matrix of the piece: own updates (sources in the piece) + the children's blocks
        part = {}
        uupd = self.upd[pm["uu0"]:pm["uu0"] + pm["nuu"]]
        usrc = self.usrc[pm["us0"]:pm["us0"] + pm["nus"]]
        for im in self.uitem[pm["uit0"]:pm["uit0"] + pm["nuit"]]:
            fl = int(im["flags"])
            di = 6 if fl & (1 << 12) else 3
            dj = 6 if fl & (1 << 13) else 3
            diag = bool(fl & (1 << 14))
            u = uupd[im["u0"]:im["u0"] + im["n"]]
            assert len(u) == im["n"]
            for r in u:
                assert upd_fields(r)[0] < pm["lsize"] and upd_fields(r)[1] < pm["lsize"] and upd_fields(r)[2] < pm["ysize"]
            acc, accy = self._tile_sum(u, smL, smY, lbase, y0, di, dj, diag)
            if fl & 1:
                assert im["s0"] + im["ns"] <= pm["nus"]
                for a in usrc[im["s0"]:im["s0"] + im["ns"]]:
                    blk = Uval[a["uoff"]:a["uoff"] + di * dj]
                    assert not np.isnan(blk).any()
                    acc += blk.reshape(di, dj)
                    if diag and a["uyoff"] >= 0:
                        accy += Uval[a["uyoff"]:a["uyoff"] + dj]
                Uval[im["uoff"]:im["uoff"] + di * dj] = acc.ravel()
                if diag:
                    Uval[im["uyoff"]:im["uyoff"] + dj] = accy
            else:
                part[(fl >> 1) & 0x7FF] = (acc, accy)
        for mm in self.umb[pm["umb0"]:pm["umb0"] + pm["numb"]]:
            di, dj = int(mm["info"]) & 15, (int(mm["info"]) >> 4) & 15
            diag = bool(int(mm["info"]) & B_DIAG)
            acc = np.zeros((di, dj)); accy = np.zeros(di)
            for q in range(mm["n"]):
                a, ay = part[mm["ps0"] + q]
                acc += a; accy += ay
            for a in usrc[mm["s0"]:mm["s0"] + mm["ns"]]:
                acc += Uval[a["uoff"]:a["uoff"] + di * dj].reshape(di, dj)
                if diag and a["uyoff"] >= 0:
                    accy += Uval[a["uyoff"]:a["uyoff"] + dj]
            Uval[mm["uoff"]:mm["uoff"] + di * dj] = acc.ravel()
            if diag:
                Uval[mm["uyoff"]:mm["uyoff"] + dj] = accy
        Lval[lbase:lbase + pm["lsize"]] = smL
        y[y0:y0 + pm["ysize"]] = smY
        return ok

    # ---- backward substitution, pieces top-down
    def backward(self, Lval, y):
        x = np.zeros(self.dim)
        for p in reversed(self.piece_order()):
            pm = self.piece[p]
            lbase, y0 = int(pm["lbase"]), int(pm["y0"])
            smX = y[y0:y0 + pm["ysize"]].copy()
            cols = self.col[pm["c0"]:pm["c0"] + pm["nc"]]
            for cm in cols:
                d = int(cm["dim"]); yl = cm["yoff"] - y0
                for bm in self.blk[cm["b0"] + cm["nbi"]:cm["b0"] + cm["nb"]]:
                    assert not (int(bm["info"]) & B_ROWIN)
                    di = int(bm["info"]) & 15
                    Lb = Lval[bm["off"]:bm["off"] + di * d].reshape(di, d)
                    smX[yl:yl + d] -= Lb.T @ x[bm["xoff_row"]:bm["xoff_row"] + di]
            for lv in self.ilv[pm["ilv0"]:pm["ilv0"] + pm["nilv"]][::-1]:
                for cm in self.col[lv["c0"]:lv["c1"]]:
                    d = int(cm["dim"]); yl = cm["yoff"] - y0
                    acc = np.zeros(d)
                    for bm in self.blk[cm["b0"] + 1:cm["b0"] + cm["nbi"]]:
                        assert int(bm["info"]) & B_ROWIN
                        di = int(bm["info"]) & 15
                        Lb = Lval[bm["off"]:bm["off"] + di * d].reshape(di, d)
                        yr = bm["yoff_row"] - y0
                        assert 0 <= yr and yr + di <= pm["ysize"]
                        acc += Lb.T @ smX[yr:yr + di]
                    Lj = Lval[cm["base"]:cm["base"] + d * d].reshape(d, d)
                    smX[yl:yl + d] = np.linalg.solve(Lj.T, smX[yl:yl + d] - acc)
            for cm in cols:
                d = int(cm["dim"])
                x[cm["xoff"]:cm["xoff"] + d] = smX[cm["yoff"] - y0:cm["yoff"] - y0 + d]
        return x

    # level-scheduled forward substitution through the row lists (what the multi right-hand-side kernel does)
    def forward_rows(self, Lval, rhs_x):
        y = np.zeros(self.dim)
        for l in range(self.nlevels):
            for j in self.lvl_cols[self.lvl_ptr[l]:self.lvl_ptr[l + 1]]:
                cm = self.col[j]; d = int(cm["dim"])
                a = rhs_x[cm["xoff"]:cm["xoff"] + d].copy()
                for fm in self.fwd[cm["f0"]:cm["f1"]]:
                    dk = 6 if fm["off"] < 0 else 3
                    o = int(fm["off"]) & 0x7FFFFFFF
                    a -= Lval[o:o + d * dk].reshape(d, dk) @ y[fm["yoff"]:fm["yoff"] + dk]
                Lj = Lval[cm["base"]:cm["base"] + d * d].reshape(d, d)
                y[cm["yoff"]:cm["yoff"] + d] = np.linalg.solve(Lj, a)
        return y

    # dense L in elimination order (tests)
    def dense_L(self, Lval):
        Ld = np.zeros((self.dim, self.dim))
        for cm in self.col:
            d = int(cm["dim"])
            for bm in self.blk[cm["b0"]:cm["b0"] + cm["nb"]]:
                di = int(bm["info"]) & 15
                Ld[bm["yoff_row"]:bm["yoff_row"] + di, cm["yoff"]:cm["yoff"] + d] = Lval[bm["off"]:bm["off"] + di * d].reshape(di, d)
        return Ld

    def perm_x_to_y(self):
        """index array p with  v_y[p_y] = v_x[p_x]  (x: internal row order, y: elimination order)"""
        px = np.zeros(self.dim, np.int64); py = np.zeros(self.dim, np.int64)
        k = 0
        for cm in self.col:
            d = int(cm["dim"])
            px[k:k + d] = np.arange(cm["xoff"], cm["xoff"] + d); py[k:k + d] = np.arange(cm["yoff"], cm["yoff"] + d)
            k += d
        return px, py
