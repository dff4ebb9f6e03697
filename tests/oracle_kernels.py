"""Oracle-backed stand-ins for the resident containers of qampy_amd.core.equalisation.hip_equalisation (CPU test runs of the
host layer: the host logic is the product's, the kernels are the checker's)."""
import numpy as np

from oracle import oracle


class OracleField:                          # stands in for hip_equalisation.ResidentField (the capture resident in HBM)
    def __init__(self, E, defer=False):
        self.E = E

    def finish(self):
        pass

    def train(self, *a):
        return oracle.train_equaliser(self.E, *a)

    def apply(self, os, wx, modes=None):
        return oracle.apply_filter_to_signal(self.E, os, np.ascontiguousarray(wx), modes)


class OracleJobs:                           # stands in for hip_equalisation.ResidentJobs: one call per job, one output mode each
    def __init__(self, slices, job_modes):
        self.slices = [np.ascontiguousarray(x) for x in slices]
        self.job_modes = [int(m) for m in job_modes]

    def train(self, TrSyms, Niter, os, mu, wx, adaptive, symbols, method):
        for E, m in zip(self.slices, self.job_modes):
            _, wx, _ = oracle.train_equaliser(E, TrSyms, Niter, os, mu, wx, np.array([m]), adaptive, symbols, method)
        return wx

    def train_bank(self, TrSyms, Niter, os, mu, bank, adaptive, symbols, method):
        res = np.array(bank, copy=True)
        for j, (E, m) in enumerate(zip(self.slices, self.job_modes)):
            _, w, _ = oracle.train_equaliser(E, TrSyms, Niter, os, mu, np.ascontiguousarray(res[j]), np.array([m]), adaptive, symbols, method)
            res[j] = w
        return res

    def apply(self, os, wx):
        return np.array([oracle.apply_filter_to_signal(E, os, np.ascontiguousarray(wx), np.array([m]))[0] for E, m in zip(self.slices, self.job_modes)])
