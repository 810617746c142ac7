"""GPU parity for scale_lut: per-g-point sums of the LBL direct-beam spectral flux on the
device, and the scaling of the look-up tables (scale_lut.cpp:117-189, ckd_model.cpp:1151-1176) against a
numpy restatement over the CPU oracle's CKD optical depths."""
import numpy as np
import pytest
import torch

import ckd_synth

pytestmark = pytest.mark.gpu


def test_gmap_sum_rows(ctx):
    from ecckd_amd import api
    rs = np.random.RandomState(4)
    n, ng, nrows = 70001, 9, 7
    g_point = rs.randint(-1, ng, n).astype(np.int32)
    g_point[g_point == 4] = 5                                   # an empty g point
    wn = np.linspace(250.0, 50000.0, n)
    dwn = np.full(n, wn[1] - wn[0])
    dev = lambda a: torch.as_tensor(a, device=ctx.device)
    gm = api.GPointMap(ctx, dev(g_point), ng, dev(wn), dev(dwn))
    for dtype, rtol in ((np.float64, 1e-13), (np.float32, 1e-13)):
        rows = rs.uniform(0.0, 2.0, (nrows, n)).astype(dtype)
        got = gm.sum_rows(dev(rows))
        want = np.stack([[rows[r, g_point == g].astype(np.float64).sum() for g in range(ng)] for r in range(nrows)])
        assert np.allclose(got, want, rtol=rtol, atol=0) and np.all(got[:, 4] == 0.0)
    gm.close()


def test_scale_lut(ctx, oracle):
    from ecckd_amd import api
    model = ckd_synth.make_model_sw(seed=8)
    ng = model["ng"]
    ngas = len(model["gases"])
    scene = ckd_synth.make_scenes(model, nscene=1, ncol=1, nlay=20)[0]
    p, T, vmr = scene["pressure_hl"][0], scene["temperature_hl"][0], scene["vmr_fl"][0]
    present = np.array([1, 1, 1, 0, 1], dtype=np.int32)         # ch4 not in the LBL file's constituent list
    mu0 = 0.5
    # "LBL" direct-beam flux sums per g point: a perturbed version of what the model itself gives, and one
    # g point whose beam is extinguished half way down
    rs = np.random.RandomState(3)
    t_fl = 0.5 * (T[:-1] + T[1:])                              # scale_lut.cpp:108
    sc1 = dict(pressure_hl=p[None], temperature_hl=T[None], vmr_fl=vmr[None], gas_present=present)
    orc = ckd_synth.Oracle(oracle, dict(model, planck_function=np.zeros((2, ng))), [sc1], {})

    def od_total_ref():                                          # :137-182 with the plain-mean temperature
        import ctypes as C
        P = oracle._p
        m = model
        nt, np_ = m["temperature"].shape
        total = np.zeros((1, p.size - 1, ng))
        tmp = np.empty_like(total)
        for i, g in enumerate(m["gases"]):
            if not present[i]:
                continue
            k = np.ascontiguousarray(g["molar_abs"])
            vl = np.ascontiguousarray(g["vmr"]) if g.get("vmr") is not None else None
            v = np.ascontiguousarray(vmr[i][None]) if g["conc"] != "none" else None
            rc = orc.L.orc_ckd_optical_depth(C.c_int(ng), C.c_int(nt), C.c_int(np_), P(np.ascontiguousarray(m["log_pressure"])),
                                             P(np.ascontiguousarray(m["temperature"])), C.c_int(ckd_synth.CONC[g["conc"]]),
                                             C.c_int(k.shape[0] if g["conc"] == "lut" else 1), P(vl),
                                             C.c_double(g.get("reference_vmr", 0.0)), P(k), C.c_int(1), C.c_int(p.size - 1),
                                             P(np.ascontiguousarray(p[None])), P(np.ascontiguousarray(t_fl[None])), P(v), P(tmp))
            assert rc == 0
            total += tmp
        return total[0]

    od_tot = od_total_ref()
    od_lbl = od_tot * np.exp(0.3 * rs.normal(size=od_tot.shape))
    flux = np.empty((p.size, ng))
    flux[0] = mu0 * rs.uniform(5.0, 30.0, ng)
    for l in range(p.size - 1):
        flux[l + 1] = flux[l] * np.exp(-od_lbl[l] / mu0)
    flux[12:, 3] = 0.0
    # numpy restatement of scale_lut.cpp:117-133, :186-187 and ckd_model.cpp:1151-1176
    with np.errstate(divide="ignore", invalid="ignore"):
        od_best = np.where(flux[1:] <= 0.0, -1.0, -mu0 * np.log(np.where(flux[1:] > 0, flux[1:], 1.0) / flux[:-1]))
    scaling = np.where(od_best <= 0.0, 1.0, od_best / od_tot)
    x = np.log(0.5 * (p[:-1] + p[1:]))
    xi = model["log_pressure"]
    j = np.clip(np.searchsorted(x, xi, side="left") - 1, 0, x.size - 2)
    w = ((xi - x[j]) / (x[j + 1] - x[j]))[:, None]
    local = (1.0 - w) * scaling[j] + w * scaling[j + 1]
    outs, got_scaling = api.scale_lut(ctx, model, flux, p, T, vmr, present, mu0)
    assert np.allclose(got_scaling, scaling, rtol=1e-10)
    assert np.all(got_scaling[11:, 3] == 1.0)
    for g, out in zip(model["gases"], outs):
        want = np.clip(g["molar_abs"] * local, g["min_molar_abs"], g["max_molar_abs"])
        assert np.allclose(out, want, rtol=1e-10, atol=1e-300), g["name"]
    clipped = sum(int(np.sum((o == g["min_molar_abs"]) | (o == g["max_molar_abs"]))) for g, o in zip(model["gases"], outs))
    assert clipped > 0                                           # the [min, max] clamp is exercised
