"""Synthetic CKD model + training scenes for the optimize_lut tests, and the CPU oracle of the
total cost function (composition of oracle_ckd.c pieces exactly as solve_adept.cpp:72-211 does)."""
import ctypes as C

import numpy as np

CONC = {"none": 0, "linear": 1, "lut": 2, "relative-linear": 3}


from ecckd_amd.synthetic import ckd_model as make_model, ckd_scenes as make_scenes  # noqa: E402,F401


class Oracle:
    """Forward cost J(x) on the CPU: solve_adept.cpp:72-211 + :273 from oracle_ckd.c pieces."""

    def __init__(self, pyoracle, model, scenes, cfg):
        self.o, self.m, self.scenes, self.cfg = pyoracle, model, scenes, cfg
        L = pyoracle.lib()
        L.orc_calc_cost_function_ckd_lw.restype = C.c_double
        self.L = L
        self.active = [i for i, g in enumerate(model["gases"]) if g.get("active", True)]
        self.sizes = [model["gases"][i]["molar_abs"].size for i in self.active]
        self.x0 = np.concatenate([self._logk(model["gases"][i]["molar_abs"].ravel()) for i in self.active])
        self._prior = None

    @staticmethod
    def _logk(k):
        with np.errstate(divide="ignore"):
            return np.where(k > 0.0, np.log(np.where(k > 0.0, k, 1.0)), -1.0e20)

    def coeffs(self, x):
        out = [g["molar_abs"] for g in self.m["gases"]]
        off = 0
        for i, n in zip(self.active, self.sizes):
            xx = x[off:off + n]
            out[i] = np.where(xx > -1.0e20, np.exp(np.minimum(xx, 700.0)), 0.0).reshape(self.m["gases"][i]["molar_abs"].shape)
            off += n
        return out

    def optical_depth(self, x, scene):
        m, P = self.m, self.o._p
        ks = self.coeffs(x)
        p, T, vmr = (np.ascontiguousarray(scene[k]) for k in ("pressure_hl", "temperature_hl", "vmr_fl"))
        ncol, nhl = p.shape
        nlay = nhl - 1
        nt, np_ = m["temperature"].shape
        ng = m["planck_function"].shape[1]
        t_fl = np.ascontiguousarray((T[:, :-1] * p[:, :-1] + T[:, 1:] * p[:, 1:]) / (p[:, :-1] + p[:, 1:]))
        total = np.zeros((ncol, nlay, ng))
        tmp = np.empty((ncol, nlay, ng))
        for i, g in enumerate(m["gases"]):
            present = scene.get("gas_present") is None or scene["gas_present"][i]
            if not present and g["conc"] != "none":
                continue                                   # solve_adept.cpp:55-67
            v = np.ascontiguousarray(vmr[:, i, :]) if (present and g["conc"] != "none") else None
            k = np.ascontiguousarray(ks[i])
            vl = np.ascontiguousarray(g["vmr"]) if g.get("vmr") is not None else None
            rc = self.L.orc_ckd_optical_depth(C.c_int(ng), C.c_int(nt), C.c_int(np_), P(np.ascontiguousarray(m["log_pressure"])),
                                              P(np.ascontiguousarray(m["temperature"])), C.c_int(CONC[g["conc"]]),
                                              C.c_int(k.shape[0] if g["conc"] == "lut" else 1), P(vl),
                                              C.c_double(g.get("reference_vmr", 0.0)), P(k), C.c_int(ncol), C.c_int(nlay),
                                              P(p), P(t_fl), P(v), P(tmp))
            assert rc == 0
            total += tmp
        return total

    def planck(self, T):
        m, P = self.m, self.o._p
        T = np.ascontiguousarray(T).ravel()
        ng = m["planck_function"].shape[1]
        out = np.empty((T.size, ng))
        self.L.orc_ckd_planck(C.c_int(m["planck_function"].shape[0]), P(np.ascontiguousarray(m["temperature_planck"])),
                              P(np.ascontiguousarray(m["planck_function"])), C.c_int(ng), C.c_int(T.size), P(T), P(out))
        return out

    def fluxes(self, x, scene, unclamped=False):
        """CKD fluxes per g: (ncol, 2, nhl, ng), LblFluxes::calc_ckd_fluxes.  unclamped: on the optical depths as they come
        out of the tables (the "relative_to" evaluation, optimize_lut.cpp:229-234) instead of the cost function's clamped ones."""
        od = self.optical_depth(x, scene)
        if not unclamped:
            od = np.maximum(od, 0.0)
        T = scene["temperature_hl"]
        ncol, nhl = T.shape
        ng = od.shape[2]
        out = np.empty((ncol, 2, nhl, ng))
        for c in range(ncol):
            pl = self.planck(T[c])
            d, u = self.o.radiative_transfer_lw(pl, od[c], np.ones(ng), pl[-1])
            out[c, 0], out[c, 1] = d, u
        return out

    def band_fluxes(self, x, scene):
        f = self.fluxes(x, scene)
        ib = self.m["iband_per_g"]
        nband = self.m["nband"]
        return np.stack([f[:, :, :, ib == b].sum(-1) for b in range(nband)], axis=-1)  # (ncol,2,nhl,nband)

    def cost_rt(self, x):
        cfg, P = self.cfg, self.o._p
        J = 0.0
        nband = self.m["nband"]
        ib = np.ascontiguousarray(self.m["iband_per_g"], dtype=np.int32)
        for scene in self.scenes:
            od = self.optical_depth(x, scene)
            neg = od < 0.0
            if neg.any():                                  # solve_adept.cpp:107-116
                J += cfg["negative_od_penalty"] * np.sum(od[neg] ** 2)
                od = np.where(neg, 0.0, od)
            p, T = scene["pressure_hl"], scene["temperature_hl"]
            ncol, nhl = p.shape
            nlay = nhl - 1
            ng = od.shape[2]
            for c in range(ncol):
                pw = cfg["pressure_weight_power"]
                lw = (np.sqrt(p[c, 1:]) - np.sqrt(p[c, :-1])) if pw == 0.5 else (p[c, 1:] ** pw - p[c, :-1] ** pw)
                lw = np.ascontiguousarray(lw / lw.sum())
                pl = self.planck(T[c])
                fd = np.ascontiguousarray(scene["flux_dn"][c])
                fu = np.ascontiguousarray(scene["flux_up"][c])
                hr = self.o.heating_rate(p[c], fd, fu)
                sfd = scene.get("spectral_flux_dn_surf")
                sfu = scene.get("spectral_flux_up_toa")
                J += self.L.orc_calc_cost_function_ckd_lw(
                    C.c_int(nlay), C.c_int(ng), C.c_int(nband), P(np.ascontiguousarray(p[c])), P(pl),
                    P(np.ones(nband)), P(np.ascontiguousarray(pl[-1])), P(np.ascontiguousarray(od[c])), P(fd), P(fu),
                    P(np.ascontiguousarray(hr)), P(np.ascontiguousarray(sfd[c])) if sfd is not None else None,
                    P(np.ascontiguousarray(sfu[c])) if sfu is not None else None, C.c_double(cfg["flux_weight"]),
                    C.c_double(cfg["flux_profile_weight"]), C.c_double(cfg["broadband_weight"]),
                    C.c_double(cfg["spectral_boundary_weight"]), P(lw),
                    P(np.ascontiguousarray(scene["relative_flux_dn"][c])) if scene.get("relative_flux_dn") is not None else None,
                    P(np.ascontiguousarray(scene["relative_flux_up"][c])) if scene.get("relative_flux_up") is not None else None,
                    ib.ctypes.data_as(C.POINTER(C.c_int)))
        return J

    def cost_grad_rt(self, x):
        """cost_rt and its gradient with respect to x = ln k by the oracle's hand-written reverse mode (oracle_adjoint.c):
        per profile dJ/d(optical depth), the penalty of negative optical depths and the clamp (solve_adept.cpp:107-116),
        the transpose of the look-up-table interpolation, then dJ/dx = dJ/dk * k (solve_adept.cpp:276-283)."""
        cfg, P, L, m = self.cfg, self.o._p, self.L, self.m
        L.orc_calc_cost_function_ckd_lw_ad.restype = C.c_double
        nband = m["nband"]
        ib = np.ascontiguousarray(m["iband_per_g"], dtype=np.int32)
        nt, np_ = m["temperature"].shape
        ks = self.coeffs(x)
        dk = [np.zeros_like(np.ascontiguousarray(k, dtype=np.float64)) for k in ks]
        J = 0.0
        for scene in self.scenes:
            od = self.optical_depth(x, scene)
            neg = od < 0.0
            p, T, vmr = (np.ascontiguousarray(scene[k]) for k in ("pressure_hl", "temperature_hl", "vmr_fl"))
            ncol, nhl = p.shape
            nlay, ng = nhl - 1, od.shape[2]
            d_od = np.zeros_like(od)
            if neg.any():
                J += cfg["negative_od_penalty"] * np.sum(od[neg] ** 2)
            odc = np.where(neg, 0.0, od)
            for c in range(ncol):
                pw = cfg["pressure_weight_power"]
                lw = (np.sqrt(p[c, 1:]) - np.sqrt(p[c, :-1])) if pw == 0.5 else (p[c, 1:] ** pw - p[c, :-1] ** pw)
                lw = np.ascontiguousarray(lw / lw.sum())
                dc = np.zeros((nlay, ng))
                J += self._profile_cost_ad(scene, c, odc[c], lw, ib, nband, dc)
                d_od[c] = dc
            # the clamped cells pass nothing on but the penalty's derivative (:110-113)
            d_od = np.where(neg, 2.0 * cfg["negative_od_penalty"] * od, d_od)
            d_od = np.ascontiguousarray(d_od)
            t_fl = np.ascontiguousarray((T[:, :-1] * p[:, :-1] + T[:, 1:] * p[:, 1:]) / (p[:, :-1] + p[:, 1:]))
            for i, g in enumerate(m["gases"]):
                present = scene.get("gas_present") is None or scene["gas_present"][i]
                if not present and g["conc"] != "none":
                    continue
                v = np.ascontiguousarray(vmr[:, i, :]) if (present and g["conc"] != "none") else None
                vl = np.ascontiguousarray(g["vmr"]) if g.get("vmr") is not None else None
                rc = L.orc_ckd_optical_depth_ad(C.c_int(ng), C.c_int(nt), C.c_int(np_), P(np.ascontiguousarray(m["log_pressure"])),
                                                P(np.ascontiguousarray(m["temperature"])), C.c_int(CONC[g["conc"]]),
                                                C.c_int(dk[i].shape[0] if g["conc"] == "lut" else 1), P(vl),
                                                C.c_double(g.get("reference_vmr", 0.0)), C.c_int(ncol), C.c_int(nlay), P(p), P(t_fl),
                                                P(v), P(d_od), P(dk[i]))
                assert rc == 0
        grad = np.zeros_like(x)
        off = 0
        for i, n in zip(self.active, self.sizes):
            xx = x[off:off + n]
            kk = np.asarray(ks[i]).ravel()
            grad[off:off + n] = np.where(xx > -1.0e20, dk[i].ravel() * kk, 0.0)
            off += n
        return J, grad

    def _profile_cost_ad(self, scene, c, od_c, lw, ib, nband, dc):
        """cost of profile c of `scene` and dc += d cost / d optical depth (oracle_adjoint.c, longwave)"""
        cfg, P, L = self.cfg, self.o._p, self.L
        p, T = np.ascontiguousarray(scene["pressure_hl"]), np.ascontiguousarray(scene["temperature_hl"])
        nlay, ng = od_c.shape
        pl = self.planck(T[c])
        fd, fu = np.ascontiguousarray(scene["flux_dn"][c]), np.ascontiguousarray(scene["flux_up"][c])
        hr = np.ascontiguousarray(self.o.heating_rate(p[c], fd, fu))
        sfd, sfu = scene.get("spectral_flux_dn_surf"), scene.get("spectral_flux_up_toa")
        rd, ru = scene.get("relative_flux_dn"), scene.get("relative_flux_up")
        return L.orc_calc_cost_function_ckd_lw_ad(
            C.c_int(nlay), C.c_int(ng), C.c_int(nband), P(np.ascontiguousarray(p[c])), P(pl), P(np.ones(nband)),
            P(np.ascontiguousarray(pl[-1])), P(np.ascontiguousarray(od_c)), P(fd), P(fu), P(hr),
            P(np.ascontiguousarray(sfd[c])) if sfd is not None else None,
            P(np.ascontiguousarray(sfu[c])) if sfu is not None else None, C.c_double(cfg["flux_weight"]),
            C.c_double(cfg["flux_profile_weight"]), C.c_double(cfg["broadband_weight"]),
            C.c_double(cfg["spectral_boundary_weight"]), P(lw),
            P(np.ascontiguousarray(rd[c])) if rd is not None else None,
            P(np.ascontiguousarray(ru[c])) if ru is not None else None, ib.ctypes.data_as(C.POINTER(C.c_int)), P(dc))

    def prior_matrices(self):
        """Dense inverse covariance per active gas exactly as create_error_covariances builds it
        (ckd_model.cpp:693-713, :760-780): pow(corr, |index difference|), LAPACK inverse, < 1e-6 zeroed."""
        if self._prior is None:
            cfg = self.cfg
            nt, np_ = self.m["temperature"].shape
            mats = []
            for i in self.active:
                g = self.m["gases"][i]
                nconc = g["molar_abs"].shape[0] if g["conc"] == "lut" else 1
                ic, it, ip = np.meshgrid(np.arange(nconc), np.arange(nt), np.arange(np_), indexing="ij")
                ic, it, ip = ic.ravel(), it.ravel(), ip.ravel()
                B = (cfg["temperature_corr"] ** np.abs(it[:, None] - it[None, :]) *
                     cfg["pressure_corr"] ** np.abs(ip[:, None] - ip[None, :]))
                if nconc > 1:
                    B = B * cfg["conc_corr"] ** np.abs(ic[:, None] - ic[None, :])
                Binv = np.linalg.inv(B)
                Binv[np.abs(Binv) < 1.0e-6] = 0.0
                mats.append(Binv)
            self._prior = mats
        return self._prior

    def cost_prior(self, x, sigma):
        """calc_background_cost_function (ckd_model.cpp:840-877) with dense matrices; returns (J, grad)."""
        ng = self.m["planck_function"].shape[1]
        J = 0.0
        grad = np.zeros_like(x)
        off = 0
        for Binv, n in zip(self.prior_matrices(), self.sizes):
            dx = (x[off:off + n] - self.x0[off:off + n]).reshape(-1, ng)
            gl = (Binv @ dx) / sigma ** 2
            J += 0.5 * np.sum(dx * gl)
            grad[off:off + n] = gl.ravel()
            off += n
        return J, grad


def make_model_sw(seed=0, **kw):
    """Shortwave variant of the synthetic model: solar irradiance + Rayleigh per g instead of a Planck LUT."""
    m = make_model(seed=seed, **kw)
    ng = m["planck_function"].shape[1]
    rs = np.random.RandomState(seed + 99)
    ssi = rs.uniform(0.5, 1.5, ng)
    m["solar_irradiance"] = 1340.0 * ssi / ssi.sum()        # differs from the scenes' tsi: exercises tsi_scaling
    m["rayleigh_molar_scattering"] = 10.0 ** rs.uniform(-7.0, -5.5, ng)
    m["planck_function"] = None
    m["temperature_planck"] = None
    m["ng"] = ng
    for g in m["gases"]:                                     # shortwave gases absorb less strongly
        for k in ("molar_abs", "min_molar_abs", "max_molar_abs"):
            g[k] = g[k] * 0.002
    return m


def make_scenes_sw(model, albedo, mu0=(1.0, 0.6, 0.25, 0.1), tsi=1361.0, boundary_weights=None, **kw):
    """lbl_fluxes.cpp:68-148: every profile repeated per solar zenith angle (done here on the host)."""
    base = make_scenes(model, **kw)
    out = []
    for s in base:
        ncol = s["pressure_hl"].shape[0]
        rep = lambda a: np.ascontiguousarray(np.repeat(a, len(mu0), axis=0))
        t = dict(pressure_hl=rep(s["pressure_hl"]), temperature_hl=rep(s["temperature_hl"]), vmr_fl=rep(s["vmr_fl"]),
                 gas_present=s["gas_present"], mu0=np.tile(np.asarray(mu0, dtype=np.float64), ncol), tsi=tsi,
                 albedo=np.asarray(albedo, dtype=np.float64))
        if boundary_weights is not None:
            t["spectral_boundary_weights"] = np.asarray(boundary_weights, dtype=np.float64)
        out.append(t)
    return out


class OracleSW(Oracle):
    """Shortwave branch of solve_adept.cpp:172-200 from oracle_ckd.c / oracle_rt.c pieces."""

    def __init__(self, pyoracle, model, scenes, cfg):
        ng = model["ng"]
        shim = dict(model, planck_function=np.zeros((2, ng)))   # Oracle only reads its shape
        super().__init__(pyoracle, shim, scenes, cfg)
        self.L.orc_calc_cost_function_ckd_sw.restype = C.c_double

    def optical_depth(self, x, scene):
        od = super().optical_depth(x, scene)
        ray = self.m.get("rayleigh_molar_scattering")
        if ray is not None:                                  # ckd_model.h:242-252
            p = scene["pressure_hl"]
            moles = (p[:, 1:] - p[:, :-1]) * (1.0 / (9.80665 * 0.001 * 28.970))
            od = od + moles[:, :, None] * np.asarray(ray)[None, None, :]
        return od

    def ssi(self, scene):
        return np.ascontiguousarray(scene["tsi"] / self.m["solar_irradiance"].sum() * self.m["solar_irradiance"])

    def _profile_cost_ad(self, scene, c, od_c, lw, ib, nband, dc):
        """cost of profile c and dc += d cost / d optical depth: the shortwave reverse mode of oracle_adjoint.c"""
        cfg, P, L = self.cfg, self.o._p, self.L
        L.orc_calc_cost_function_ckd_sw_ad.restype = C.c_double
        p = np.ascontiguousarray(scene["pressure_hl"])
        nlay, ng = od_c.shape
        fd, fu = np.ascontiguousarray(scene["flux_dn"][c]), np.ascontiguousarray(scene["flux_up"][c])
        hr = np.ascontiguousarray(self.o.heating_rate(p[c], fd, None))
        sfd, sbw = scene.get("spectral_flux_dn_surf"), scene.get("spectral_boundary_weights")
        use_b = sfd is not None and sbw is not None
        rd, ru = scene.get("relative_flux_dn"), scene.get("relative_flux_up")
        return L.orc_calc_cost_function_ckd_sw_ad(
            C.c_int(nlay), C.c_int(ng), C.c_int(nband), C.c_double(scene["mu0"][c]), P(np.ascontiguousarray(p[c])),
            P(self.ssi(scene)), P(np.ascontiguousarray(scene["albedo"])), P(np.ascontiguousarray(od_c)), P(fd), P(fu), P(hr),
            P(np.ascontiguousarray(sfd[c])) if use_b else None, C.c_double(cfg["flux_weight"]),
            C.c_double(cfg["flux_profile_weight"]), C.c_double(cfg["broadband_weight"]),
            P(np.ascontiguousarray(sbw)) if use_b else None, P(lw),
            P(np.ascontiguousarray(rd[c])) if rd is not None else None,
            P(np.ascontiguousarray(ru[c])) if ru is not None else None, ib.ctypes.data_as(C.POINTER(C.c_int)), P(dc))

    def fluxes(self, x, scene):
        od = np.maximum(self.optical_depth(x, scene), 0.0)
        ncol, nlay, ng = od.shape
        out = np.zeros((ncol, 2, nlay + 1, ng))
        alb = scene["albedo"]
        alb_g = np.ascontiguousarray(alb[self.m["iband_per_g"]])
        for c in range(ncol):
            if np.all(alb <= 0.0):
                out[c, 0] = self.o.radiative_transfer_direct_sw(scene["mu0"][c], self.ssi(scene), od[c])
            else:
                out[c, 0], out[c, 1] = self.o.radiative_transfer_norayleigh_sw(scene["mu0"][c], self.ssi(scene), od[c], alb_g)
        return out

    def cost_rt(self, x):
        cfg, P = self.cfg, self.o._p
        J = 0.0
        nband = self.m["nband"]
        ib = np.ascontiguousarray(self.m["iband_per_g"], dtype=np.int32)
        for scene in self.scenes:
            od = self.optical_depth(x, scene)
            neg = od < 0.0
            if neg.any():
                J += cfg["negative_od_penalty"] * np.sum(od[neg] ** 2)
                od = np.where(neg, 0.0, od)
            p = scene["pressure_hl"]
            ncol, nhl = p.shape
            nlay, ng = nhl - 1, od.shape[2]
            ssi = self.ssi(scene)
            for c in range(ncol):
                pw = cfg["pressure_weight_power"]
                lw = (np.sqrt(p[c, 1:]) - np.sqrt(p[c, :-1])) if pw == 0.5 else (p[c, 1:] ** pw - p[c, :-1] ** pw)
                lw = np.ascontiguousarray(lw / lw.sum())
                fd = np.ascontiguousarray(scene["flux_dn"][c])
                fu = np.ascontiguousarray(scene["flux_up"][c])
                hr = np.ascontiguousarray(self.o.heating_rate(p[c], fd, None))   # direct beam only
                sfd = scene.get("spectral_flux_dn_surf")
                sbw = scene.get("spectral_boundary_weights")
                use_b = sfd is not None and sbw is not None
                J += self.L.orc_calc_cost_function_ckd_sw(
                    C.c_int(nlay), C.c_int(ng), C.c_int(nband), C.c_double(scene["mu0"][c]), P(np.ascontiguousarray(p[c])),
                    P(ssi), P(np.ascontiguousarray(scene["albedo"])), P(np.ascontiguousarray(od[c])), P(fd), P(fu), P(hr),
                    P(np.ascontiguousarray(sfd[c])) if use_b else None, C.c_double(cfg["flux_weight"]),
                    C.c_double(cfg["flux_profile_weight"]), C.c_double(cfg["broadband_weight"]),
                    P(np.ascontiguousarray(sbw)) if use_b else None, P(lw),
                    P(np.ascontiguousarray(scene["relative_flux_dn"][c])) if scene.get("relative_flux_dn") is not None else None,
                    P(np.ascontiguousarray(scene["relative_flux_up"][c])) if scene.get("relative_flux_up") is not None else None,
                    ib.ctypes.data_as(C.POINTER(C.c_int)))
        return J
