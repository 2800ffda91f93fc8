"""Metropolis-adjusted Langevin with states and momenta resident on the GPU (throughput engine).

The reference's `MALA.sampler` (quinn/mcmc/mala.py:24-53) draws p ~ N(0, I), proposes
    x' = x + (eps^2 / 2) grad(x) + eps p,       p' = p + eps (grad(x) + grad(x')) / 2,
and hands K = |p|^2 / 2, K' = |p'|^2 / 2 to the MH test of `MCMCBase.run` (mcmc.py:65-85) -- which is, term for term, one
leapfrog step: half kick p + (eps / 2) grad(x), drift x + eps (that), half kick with grad(x') (the reference's own note:
"MALA is actually exactly HMC with L = 1").  So the device engine IS the device HMC engine with one leapfrog step:

    qn_hmc_begin   momenta (in-kernel Philox keyed by the GLOBAL chain id), K partials, half kick with the cached gradient
                   of the current state, drift  ->  the Langevin proposal
    qn_mlp_sse_fwdbwd at the proposal  ->  qn_hmc_leap (last: half kick + K' partials)
    qn_hmc_accept  MH test with the kinetic terms, state / gradient / MAP / history rows, device step counter

ONE gradient launch per step where the reference takes two gradients and one log-posterior (the gradient at the current
state is the accepted proposal's, the proposal's log-posterior comes with its gradient); no torch op and no host
synchronisation inside a step.  Chains agree with the host `MALA` in distribution, not bit for bit (Philox instead of
numpy's MT19937)."""
from .device_hmc import DeviceHMC


class DeviceMALA(DeviceHMC):
    """Args as `MALA` (epsilon: step size, default 0.05) plus the engine's seed / chain0 / use_graph / groups."""

    def __init__(self, op, sigma, epsilon=0.05, seed=0, chain0=0, use_graph=False, groups=None):
        super().__init__(op, sigma, epsilon=epsilon, L=1, seed=seed, chain0=chain0, use_graph=use_graph, groups=groups)
