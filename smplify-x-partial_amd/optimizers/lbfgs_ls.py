"""Drop-in handle for smplifyx/optimizers/lbfgs_ls.py's LBFGS (strong-Wolfe).  The algorithm
itself -- two-loop recursion over the last history_size pairs, first-step scaling min(1, 1/|g|_1) * lr, bracket
and zoom phases of the strong-Wolfe search, all stopping rules -- runs on the GPU in
csrc/lbfgs.hip (specification: oracle/lbfgs_machine.py).  This object carries the hyper-
parameters and forwards `.step(closure)` to the engine batch bound to the closure."""


class LBFGS(object):
    def __init__(self, params, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-5, tolerance_change=1e-9,
                 history_size=100, line_search_fn=None):
        if line_search_fn != "strong_Wolfe":
            raise RuntimeError("only 'strong_Wolfe' is supported")
        if history_size > 400 or history_size < 1:
            raise NotImplementedError("history_size must be 1..400 (the device keeps the alphas of the two-loop recursion in LDS)")
        self._params = list(params)
        self.lr, self.max_iter = lr, max_iter
        self.max_eval = max_iter * 5 // 4 if max_eval is None else max_eval
        self.tolerance_grad, self.tolerance_change, self.history_size = tolerance_grad, tolerance_change, history_size
        self.param_groups = [dict(params=self._params, lr=lr, max_iter=max_iter, max_eval=self.max_eval,
                                  tolerance_grad=tolerance_grad, tolerance_change=tolerance_change,
                                  history_size=history_size, line_search_fn=line_search_fn)]
        self._closure = None

    def _bind(self, engine_closure):
        self._closure = engine_closure

    def zero_grad(self, set_to_none=True):
        for p in self._params:
            p.grad = None

    def step(self, closure):
        """One optimisation step (up to max_iter L-BFGS iterations / max_eval evaluations) on
        device; returns the loss at entry, like the reference."""
        from ..fitting import EngineClosure
        c = closure if isinstance(closure, EngineClosure) else self._closure
        if c is None:
            raise TypeError("LBFGS.step needs a closure made by FittingMonitor.create_fitting_closure "
                            "(or a lambda wrapping one, after create_fitting_closure(optimizer, ...))")
        c._check_params(self._params)
        return c.step(stage=getattr(c, "_fb_stage", None) or 0)
