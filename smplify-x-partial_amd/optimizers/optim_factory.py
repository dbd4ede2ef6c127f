"""Drop-in for smplifyx/optimizers/optim_factory.py:27-65.  Every shipped configuration uses
optim_type 'lbfgsls' (e.g. cfg_files/fit_smplx_smplifyx.yaml:39); the adam/sgd/rmsprop/plain
lbfgs branches of the reference are outside the accelerated path."""
from .lbfgs_ls import LBFGS as LBFGSLs


def create_optimizer(parameters, optim_type="lbfgs", lr=1e-3, momentum=0.9, use_nesterov=True, beta1=0.9,
                     beta2=0.999, epsilon=1e-8, use_locking=False, weight_decay=0.0, centered=False,
                     rmsprop_alpha=0.99, maxiters=20, gtol=1e-6, ftol=1e-9, **kwargs):
    if optim_type == "lbfgsls":
        return LBFGSLs(parameters, lr=lr, max_iter=maxiters, line_search_fn="strong_Wolfe"), False
    if optim_type in ("adam", "lbfgs", "rmsprop", "sgd"):
        raise NotImplementedError("optim_type %r is not on the MI355X path (no shipped cfg uses it); "
                                  "use 'lbfgsls'" % optim_type)
    raise ValueError("Optimizer {} not supported!".format(optim_type))
