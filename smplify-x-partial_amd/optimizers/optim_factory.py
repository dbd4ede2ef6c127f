"""Drop-in for smplifyx/optimizers/optim_factory.py:27-65.  Every shipped configuration uses
optim_type 'lbfgsls' (e.g. cfg_files/fit_smplx_smplifyx.yaml:39): that optimiser runs on the device
(csrc/lbfgs_body.h).  'adam' / 'lbfgs' / 'rmsprop' / 'sgd' return the torch.optim objects the reference
returns; FittingMonitor.run_fitting then drives them from the host, with every closure evaluation (loss
and the .grad of the optimised tensors) still computed by the HIP closure kernel."""
import torch.optim as optim

from .lbfgs_ls import LBFGS as LBFGSLs


def create_optimizer(parameters, optim_type="lbfgs", lr=1e-3, momentum=0.9, use_nesterov=True, beta1=0.9,
                     beta2=0.999, epsilon=1e-8, use_locking=False, weight_decay=0.0, centered=False,
                     rmsprop_alpha=0.99, maxiters=20, gtol=1e-6, ftol=1e-9, **kwargs):
    if optim_type == "lbfgsls":
        return LBFGSLs(parameters, lr=lr, max_iter=maxiters, line_search_fn="strong_Wolfe"), False
    if optim_type == "adam":
        return optim.Adam(parameters, lr=lr, betas=(beta1, beta2), weight_decay=weight_decay), False
    if optim_type == "lbfgs":
        return optim.LBFGS(parameters, lr=lr, max_iter=maxiters), False
    if optim_type == "rmsprop":        # (the reference passes `epsilon=`, which torch.optim.RMSprop rejects: `eps` meant)
        return optim.RMSprop(parameters, lr=lr, eps=epsilon, alpha=rmsprop_alpha, weight_decay=weight_decay,
                             momentum=momentum, centered=centered), False
    if optim_type == "sgd":
        return optim.SGD(parameters, lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=use_nesterov), False
    raise ValueError("Optimizer {} not supported!".format(optim_type))
