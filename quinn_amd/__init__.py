"""quinn_amd -- MI355X-native hot path of QUiNN (sandialabs/quinn).

Batched log-posterior / gradient, ELBO and ensemble evaluation of an MLP over many
independent chains / MC samples / members at once, behind QUiNN's wrapper API
(NN_MCMC, NN_VI, NN_Ens, AMCMC, HMC, MALA).  The compute path is the C-ABI library
quinn_amd/lib/libquinn_amd.so (hand-written gfx950 HIP kernels); there is no CPU fallback.
"""
from ._lib import QuinnAmdError, build  # noqa: F401

__version__ = "0.1"
