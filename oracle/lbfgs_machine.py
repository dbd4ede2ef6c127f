"""ORACLE (test infrastructure): the reference's optimiser restated as an explicit
per-frame state machine -- the exact specification the HIP kernel
`sfx_lbfgs_tick` (csrc/lbfgs.hip) implements.

Restates, for ONE frame and ONE stage:
  * FittingMonitor.run_fitting            smplifyx/fitting.py:147-217
  * LBFGS.step (strong-Wolfe branch)      smplifyx/optimizers/lbfgs_ls.py:256-445
  * _strong_Wolfe / _cubic_interpolate    smplifyx/optimizers/lbfgs_ls.py:11-167
  * optim_factory 'lbfgsls' defaults      smplifyx/optimizers/optim_factory.py:50-52
    (max_eval = max_iter*5//4, tolerance_grad 1e-5, tolerance_change 1e-9, history 100)

Pinned against the real reference by tests/golden/lbfgs_*.npz (tools/make_goldens.py).

The reference mixes Python floats (double) with 0-d float32 tensors; PyTorch then
computes in float32 whenever a tensor takes part and in double otherwise.  `Sc`
reproduces that rule so that every branch decision is taken on the same rounded
numbers: a scalar is (value, is_tensor); an operation with a tensor operand is carried
out in the working dtype after rounding both operands to it.

The machine is *pull based*: `x_trial` is where the next closure evaluation is wanted;
`feed(f, g)` consumes the result.  One `feed` = one closure evaluation = one tick of
the batched device engine.
"""
import numpy as np

PH_ENTRY, PH_BRACKET, PH_ZOOM, PH_DONE = 0, 1, 2, 3


class Sc(object):
    """Dual-typed scalar: Python float (double) or 0-d tensor of dtype `dt`."""
    __slots__ = ("v", "t")
    dt = np.float32

    def __init__(self, v, t):
        self.v = (Sc.dt(v) if t else np.float64(v))
        self.t = bool(t)

    def __repr__(self):
        return "%s(%r)" % ("T" if self.t else "P", float(self.v))


def P(v):
    return Sc(v, False)


def T(v):
    return Sc(v, True)


def _bin(a, b, op):
    if a.t or b.t:
        x, y = Sc.dt(a.v), Sc.dt(b.v)
        with np.errstate(all="ignore"):
            return Sc(op(x, y), True)
    with np.errstate(all="ignore"):
        return Sc(op(np.float64(a.v), np.float64(b.v)), False)


def add(a, b): return _bin(a, b, lambda x, y: x + y)
def sub(a, b): return _bin(a, b, lambda x, y: x - y)
def mul(a, b): return _bin(a, b, lambda x, y: x * y)
def div(a, b): return _bin(a, b, lambda x, y: x / y)


def _cmp(a, b):
    if a.t or b.t:
        return Sc.dt(a.v), Sc.dt(b.v)
    return np.float64(a.v), np.float64(b.v)


def lt(a, b): x, y = _cmp(a, b); return bool(x < y)
def le(a, b): x, y = _cmp(a, b); return bool(x <= y)
def gt(a, b): x, y = _cmp(a, b); return bool(x > y)
def ge(a, b): x, y = _cmp(a, b); return bool(x >= y)
def pmax(a, b): return b if gt(b, a) else a      # Python max(a, b)
def pmin(a, b): return b if lt(b, a) else a      # Python min(a, b)
def sabs(a): return Sc(abs(a.v), a.t)
def neg(a): return Sc(-a.v, a.t)


def cubic_interpolate(x1, f1, g1, x2, f2, g2, bounds=None):
    """Minimiser of the cubic through (x1,f1,g1),(x2,f2,g2), clamped to bounds
    (lbfgs_ls.py:11-36)."""
    if bounds is not None:
        lo, hi = bounds
    else:
        lo, hi = (x1, x2) if le(x1, x2) else (x2, x1)
    d1 = sub(add(g1, g2), div(mul(P(3), sub(f1, f2)), sub(x1, x2)))
    d2sq = sub(mul(d1, d1), mul(g1, g2))
    if ge(d2sq, P(0)):
        d2 = Sc(np.sqrt(d2sq.v), d2sq.t)
        if le(x1, x2):
            mp = sub(x2, mul(sub(x2, x1), div(sub(add(g2, d2), d1),
                                              add(sub(g2, g1), mul(P(2), d2)))))
        else:
            mp = sub(x1, mul(sub(x1, x2), div(sub(add(g1, d2), d1),
                                              add(sub(g1, g2), mul(P(2), d2)))))
        return pmin(pmax(mp, lo), hi)
    return div(add(lo, hi), P(2.0))


class StageMachine(object):
    """run_fitting + LBFGS('lbfgsls') for one frame and one stage on a flat vector."""

    def __init__(self, x0, groups=None, maxiters=30, ftol=1e-9, gtol=1e-9, lr=1.0,
                 max_iter=None, max_eval=None, history=100, tol_grad=1e-5, tol_change=1e-9,
                 c1=1e-4, c2=0.9, max_ls=25, dtype=np.float32, reuse_entry_eval=False,
                 fma=True):
        self.dt = dtype
        Sc.dt = dtype
        self.x = np.array(x0, dtype=dtype).copy()
        self.N = self.x.size
        # parameter groups (offset, length, has_grad) for the gtol test of fitting.py:191
        self.groups = groups if groups is not None else [(0, self.N, True)]
        self.maxiters = maxiters
        self.ftol, self.gtol, self.lr = ftol, gtol, lr
        self.max_iter = maxiters if max_iter is None else max_iter
        self.max_eval = self.max_iter * 5 // 4 if max_eval is None else max_eval      # lbfgs_ls.py:262
        self.history = history
        self.tol_grad, self.tol_change = tol_grad, tol_change
        self.c1, self.c2, self.max_ls = c1, c2, max_ls
        self.reuse = reuse_entry_eval
        self.fma = fma
        # optimiser state (lbfgs_ls.py:293-300)
        self.n_iter_total = 0
        self.func_evals = 0
        self.d = None
        self.t = None
        self.Y, self.S, self.ro = [], [], []
        self.H_diag = P(1)
        self.prev_g = None
        # run_fitting state
        self.outer = 0
        self.prev_loss_outer = None
        self.result = None
        self.phase = PH_ENTRY
        self.x_trial = self.x.copy()
        self.g_last = None
        self.cache = None              # (f, g) valid at self.x
        self.evals = 0                 # closure evaluations actually requested
        self.trace = []
        self.records = []              # same records as the device trace (include/sfx.h: sfx_batch_trace)

    # ---- helpers -----------------------------------------------------------------------
    def _axpy(self, x, a, d):
        """x + a*d in working precision (ATen add_(alpha) vector path is an fma)."""
        a = self.dt(a.v)
        if self.fma and self.dt == np.float32:
            return (x.astype(np.float64) + np.float64(a) * d.astype(np.float64)).astype(np.float32)
        return (x + a * d).astype(self.dt)

    def _dot(self, a, b):
        return T(np.dot(a, b))

    def _trial(self, t):
        return self._axpy(self.x_init, t, self.d)

    # ---- protocol ----------------------------------------------------------------------
    @property
    def done(self):
        return self.phase == PH_DONE

    def feed(self, f, g):
        """Consume closure(x_trial) = (f, g)."""
        Sc.dt = self.dt
        self.evals += 1
        g = np.asarray(g, dtype=self.dt).copy()
        self.g_last = g
        fP = P(self.dt(f))             # float(closure()) : python float holding the fp value
        if self.phase == PH_ENTRY:
            self._on_entry(fP, g)
        elif self.phase == PH_BRACKET:
            self._on_bracket(fP, g)
        elif self.phase == PH_ZOOM:
            self._on_zoom(fP, g)
        else:
            raise RuntimeError("machine is done")
        # entry evaluations at an unchanged point may be served from the cache
        while self.phase == PH_ENTRY and self.reuse and self.cache is not None:
            cf, cg = self.cache
            self.g_last = cg
            self._on_entry(cf, cg.copy())

    # ---- LBFGS.step prologue (lbfgs_ls.py:279-290) ---------------------------------------
    def _on_entry(self, fP, g):
        self.cache = (fP, g.copy())    # (f, g) at self.x: valid until x moves
        self.func_evals += 1
        self.orig_loss = fP
        self.loss = fP
        self.g = g
        self.cur_evals = 1
        self.n_iter = 0
        if le(T(np.abs(g).max()), P(self.tol_grad)):
            return self._end_step()
        self._iter_head()

    # ---- one L-BFGS iteration up to the first line-search evaluation (:304-397) ------------
    def _iter_head(self):
        if not (self.n_iter < self.max_iter):
            return self._end_step()
        self.n_iter += 1
        self.n_iter_total += 1
        g = self.g
        if self.n_iter_total == 1:
            d = -g
            self.Y, self.S, self.ro = [], [], []
            self.H_diag = P(1)
        else:
            y = (g - self.prev_g).astype(self.dt)
            s = (self.d * self.dt(self.t.v)).astype(self.dt)
            ys = self._dot(y, s)
            if gt(ys, P(1e-10)):
                if len(self.Y) == self.history:
                    self.Y.pop(0); self.S.pop(0); self.ro.pop(0)
                self.Y.append(y); self.S.append(s)
                self.ro.append(div(P(1.), ys))
                self.H_diag = div(ys, self._dot(y, y))
            k = len(self.Y)
            al = [None] * k
            q = -g
            for i in range(k - 1, -1, -1):
                al[i] = mul(self._dot(self.S[i], q), self.ro[i])
                q = self._axpy(q, neg(al[i]), self.Y[i])
            r = (q * self.dt(self.H_diag.v)).astype(self.dt)
            for i in range(k):
                be = mul(self._dot(self.Y[i], r), self.ro[i])
                r = self._axpy(r, sub(al[i], be), self.S[i])
            d = r
        self.d = d
        self.prev_g = g.copy()
        self.prev_loss = self.loss
        if self.n_iter_total == 1:
            t = mul(pmin(P(1.), div(P(1.), T(np.abs(g).sum(dtype=self.dt)))), P(self.lr))
        else:
            t = P(self.lr)
        self.t = t
        gtd = self._dot(g, d)
        if gt(gtd, P(-self.tol_change)):
            return self._end_step()
        # ---- strong-Wolfe set-up (lbfgs_ls.py:39-52) ----
        self.x_init = self.x.copy()
        self.ls_f0, self.ls_g0, self.ls_gtd0 = self.loss, g.copy(), gtd
        self.d_norm = T(np.abs(d).max())
        self.ls_evals = 0
        self.ls_iter = 0
        self.t_prev, self.f_prev, self.g_prev, self.gtd_prev = P(0), self.loss, g.copy(), gtd
        self.phase = PH_BRACKET
        self.x_trial = self._trial(t)

    def _armijo_fail(self, f_new, t):
        return gt(f_new, add(self.ls_f0, mul(mul(P(self.c1), t), self.ls_gtd0)))

    def _curv_ok(self, gtd_new):
        return le(sabs(gtd_new), mul(P(-self.c2), self.ls_gtd0))

    # ---- bracket phase (lbfgs_ls.py:54-100) ----------------------------------------------
    def _on_bracket(self, f_new, g_new):
        self.ls_evals += 1
        t = self.t
        gtd_new = self._dot(g_new, self.d)
        if self.ls_iter == self.max_ls:
            # loop budget exhausted: no tests on the last point, bracket = [0, t] (:96-100)
            return self._start_zoom([P(0), t], [self.ls_f0, f_new],
                                    [self.ls_g0, g_new], [self.ls_gtd0, gtd_new], False)
        if self._armijo_fail(f_new, t) or (self.ls_iter > 1 and ge(f_new, self.f_prev)):
            return self._start_zoom([self.t_prev, t], [self.f_prev, f_new],
                                    [self.g_prev, g_new], [self.gtd_prev, gtd_new], False)
        if self._curv_ok(gtd_new):
            return self._start_zoom([t], [f_new], [g_new], [gtd_new], True)
        if ge(gtd_new, P(0)):
            return self._start_zoom([self.t_prev, t], [self.f_prev, f_new],
                                    [self.g_prev, g_new], [self.gtd_prev, gtd_new], False)
        lo = add(t, mul(P(0.01), sub(t, self.t_prev)))
        hi = mul(t, P(10))
        t_next = cubic_interpolate(self.t_prev, self.f_prev, self.gtd_prev, t, f_new, gtd_new,
                                   bounds=(lo, hi))
        self.t_prev, self.f_prev, self.g_prev, self.gtd_prev = t, f_new, g_new, gtd_new
        self.t = t_next
        self.ls_iter += 1
        self.x_trial = self._trial(t_next)

    def _start_zoom(self, br, bf, bg, bgtd, done):
        self.br, self.bf, self.bg, self.bgtd = list(br), list(bf), [np.array(v) for v in bg], list(bgtd)
        self.ls_done = done
        self.insuf = False
        self.low, self.high = (0, 1) if le(self.bf[0], self.bf[-1]) else (1, 0)
        self._zoom_next()

    # ---- zoom phase (lbfgs_ls.py:102-167) --------------------------------------------------
    def _zoom_next(self):
        if self.ls_done or not (self.ls_iter < self.max_iter):
            return self._finish_ls()
        br = self.br
        t = cubic_interpolate(br[0], self.bf[0], self.bgtd[0], br[1], self.bf[1], self.bgtd[1])
        bmax = br[1] if gt(br[1], br[0]) else br[0]          # Python max(list)
        bmin = br[1] if lt(br[1], br[0]) else br[0]          # Python min(list)
        eps = mul(P(0.1), sub(bmax, bmin))
        if lt(pmin(sub(bmax, t), sub(t, bmin)), eps):
            if self.insuf or ge(t, bmax) or le(t, bmin):
                if lt(sabs(sub(t, bmax)), sabs(sub(t, bmin))):
                    t = sub(bmax, eps)
                else:
                    t = add(bmin, eps)
                self.insuf = False
            else:
                self.insuf = True
        else:
            self.insuf = False
        self.t = t
        self.phase = PH_ZOOM
        self.x_trial = self._trial(t)

    def _on_zoom(self, f_new, g_new):
        self.ls_evals += 1
        self.ls_iter += 1
        t = self.t
        gtd_new = self._dot(g_new, self.d)
        lo, hi = self.low, self.high
        if self._armijo_fail(f_new, t) or ge(f_new, self.bf[lo]):
            self.br[hi], self.bf[hi], self.bg[hi], self.bgtd[hi] = t, f_new, g_new.copy(), gtd_new
            self.low, self.high = (0, 1) if le(self.bf[0], self.bf[1]) else (1, 0)
        else:
            if self._curv_ok(gtd_new):
                self.ls_done = True
            elif ge(mul(gtd_new, sub(self.br[hi], self.br[lo])), P(0)):
                self.br[hi], self.bf[hi], self.bg[hi], self.bgtd[hi] = \
                    self.br[lo], self.bf[lo], self.bg[lo].copy(), self.bgtd[lo]
            self.br[lo], self.bf[lo], self.bg[lo], self.bgtd[lo] = t, f_new, g_new.copy(), gtd_new
        if lt(mul(sabs(sub(self.br[1], self.br[0])), self.d_norm), P(self.tol_change)):
            return self._finish_ls()
        self._zoom_next()

    # ---- after the line search (lbfgs_ls.py:398-434) ----------------------------------------
    def _finish_ls(self):
        lo = self.low
        t = self.br[lo]
        self.loss = self.bf[lo]
        self.g = self.bg[lo].copy()
        self.t = t
        self.x = self._axpy(self.x_init, t, self.d)
        self.cache = (self.loss, self.g.copy())
        opt_cond = le(T(np.abs(self.g).max()), P(self.tol_grad))
        self.cur_evals += self.ls_evals
        self.func_evals += self.ls_evals
        self.trace.append((self.n_iter_total, float(self.loss.v), float(t.v), self.ls_evals))
        self.records.append((0, float(t.v), float(self.loss.v), self.ls_evals))
        if self.n_iter == self.max_iter:
            return self._end_step()
        if self.cur_evals >= self.max_eval:
            return self._end_step()
        if opt_cond:
            return self._end_step()
        step = (self.d * self.dt(t.v)).astype(self.dt)
        if le(T(np.abs(step).max()), P(self.tol_change)):
            return self._end_step()
        if lt(sabs(sub(self.loss, self.prev_loss)), P(self.tol_change)):
            return self._end_step()
        self._iter_head()

    # ---- run_fitting bookkeeping after optimizer.step (fitting.py:175-217) ---------------------
    def _end_step(self):
        loss = float(self.orig_loss.v)           # step() returns the ENTRY loss
        self.records.append((1, loss, self.func_evals, self.n_iter_total))
        n = self.outer
        stop = False
        if np.isnan(loss) or np.isinf(loss):
            stop = True                           # 'break' BEFORE prev_loss is updated
            self.result = self.prev_loss_outer
            return self._finish_stage()
        if n > 0 and self.prev_loss_outer is not None and self.ftol > 0:
            rel = (self.prev_loss_outer - loss) / max(abs(self.prev_loss_outer), abs(loss), 1)
            if rel <= self.ftol:
                self.result = self.prev_loss_outer
                return self._finish_stage()
        gl = self.g_last
        if all(abs(self.dt(gl[o:o + n_].max())) < self.gtol
               for (o, n_, has) in self.groups if has):
            self.result = self.prev_loss_outer
            return self._finish_stage()
        self.prev_loss_outer = loss
        self.outer += 1
        if self.outer >= self.maxiters:
            self.result = self.prev_loss_outer
            return self._finish_stage()
        self.phase = PH_ENTRY
        self.x_trial = self.x.copy()

    def _finish_stage(self):
        self.records.append((2, float("nan") if self.result is None else float(self.result), self.evals, 0))
        self.phase = PH_DONE
        self.x_trial = self.x.copy()


def run_stage(closure, x0, **kw):
    """Drive a StageMachine with `closure(x) -> (f, g)`; returns the machine."""
    m = StageMachine(x0, **kw)
    while not m.done:
        f, g = closure(m.x_trial.copy())
        m.feed(f, g)
    return m
