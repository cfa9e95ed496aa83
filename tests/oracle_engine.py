"""TEST-ONLY engine with the same methods as the product's HipEngine, backed by the oracle, so that
the host logic above the C ABI (iteration loop, source partition over ranks, reduction of the rate
grids) can be exercised on CPU with torch.distributed/gloo.  Never imported by the product."""
import numpy as np
import torch

import oracle as orc


class OracleEngine:
    def __init__(self, mesh, otables):
        self.mesh = tuple(mesh)
        self.ncell = int(np.prod(mesh))
        self.T = otables
        self.buf = torch.zeros(4 * self.ncell + 48, dtype=torch.float64)

    def set_tables(self, t):
        pass

    def set_step(self, mat, grid, cosmo):
        self._mat, self._grid, self._cosmo = mat, grid, cosmo

    def set_sources(self, src):
        self._src = src

    def upload_state(self, mat):
        m, g, c, s = self._mat, self._grid, self._cosmo, self._src
        self.st = orc.Step(g.mesh, g.dr, g.vol, c.zred, c.H0, c.Omega0, m.isothermal, m.temper_val, m.clumping,
                           s.srcpos, s.NormFlux, s.S_star, m.ndens, m.reccoef)
        self.s = orc.State(self.st, mat.xh, mat.xhe, mat.temperature_grid)

    def begin_step(self):
        orc.begin_step(self.s)

    def set_rates_to_zero(self):
        self.s.phih[:] = 0
        self.s.phihe[:] = 0
        self.s.phiheat[:] = 0
        self.loss = 0.0
        self.nbox = 0

    def pass_sources(self, first=1, stride=1):
        for ns in range(first, self.st.c.nsrc + 1, stride):
            nbox, loss = orc.do_source_accumulate(self.T, self.st, self.s, ns)
            self.loss += loss
            self.nbox += nbox

    def rates_buffer(self):
        n = self.ncell
        b = self.buf.numpy()
        b[:n], b[n:3 * n], b[3 * n:4 * n] = self.s.phih, self.s.phihe, self.s.phiheat
        b[4 * n:] = 0
        b[4 * n] = self.loss
        b[4 * n + 47] = self.nbox
        return self.buf

    def rates_reduced(self):
        n = self.ncell
        b = self.buf.numpy()
        self.s.phih[:], self.s.phihe[:], self.s.phiheat[:] = b[:n], b[n:3 * n], b[3 * n:4 * n]
        self.loss, self.nbox = float(b[4 * n]), int(round(b[4 * n + 47]))

    def synchronize(self):
        pass

    def global_pass(self, dt):
        return orc.global_pass(self.T, self.st, self.s, dt)

    def end_step(self):
        n = self.ncell
        self.s.xh[:] = self.s.xh_intermed
        self.s.xhe[:] = self.s.xhe_intermed
        if self.s.temperature is not None:
            self.s.temperature[2 * n:] = self.s.temperature[:n]

    def evolve3d(self, dt):
        n = orc.evolve3d(self.T, self.st, self.s, dt)
        return n, self.s.conv_flags

    def download_state(self, mat):
        mat.xh, mat.xhe = self.s.xh.copy(), self.s.xhe.copy()
        if self.s.temperature is not None:
            mat.temperature_grid = self.s.temperature.copy()

    def download_rates(self):
        loss = np.zeros(47)
        loss[0] = getattr(self, "loss", self.s.photon_loss[0])
        return dict(phih_grid=self.s.phih.copy(), phihe_grid=self.s.phihe.copy(), phiheat=self.s.phiheat.copy(),
                    photon_loss=loss, sum_nbox=getattr(self, "nbox", self.s.c.sum_nbox))

    def download_iter_state(self):
        return dict(xh_av=self.s.xh_av.copy(), xhe_av=self.s.xhe_av.copy(), xh_intermed=self.s.xh_intermed.copy(),
                    xhe_intermed=self.s.xhe_intermed.copy())
