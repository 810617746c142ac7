"""GPU parity for the arithmetic of read_spectrum / read_merged_spectrum (row a1) and the erythemal weights of
LblFluxes::read (row a21), against numpy restatements that cite the reference lines."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_derive_d_wavenumber(ctx):
    from ecckd_amd import api
    rs = np.random.RandomState(1)
    wn = np.cumsum(rs.uniform(1e-4, 2e-3, 100003)) + 250.0
    # read_spectrum.cpp:55-65
    want = np.empty_like(wn)
    want[1:-1] = 0.5 * (wn[2:] - wn[:-2])
    want[0] = 0.5 * want[1]
    want[-1] = 0.5 * want[-2]
    got = api.derive_d_wavenumber(ctx, torch.as_tensor(wn, device=ctx.device)).cpu().numpy()
    assert np.array_equal(got, want)


def test_merge_scaling_rules():
    from ecckd_amd import api, EcckdError
    p = np.exp(np.linspace(np.log(1.0), np.log(101325.0), 31))
    pfl = 0.5 * (p[1:] + p[:-1])
    vmr = np.linspace(3e-4, 4e-4, 30)
    # scalar rules, read_merged_spectrum.cpp:132-147
    for kw, s_want in ((dict(), 1.0), (dict(scaling=2.5), 2.5), (dict(conc=0.0, scaling=3.0), 0.0),
                       (dict(conc=8e-4, reference_surface_vmr=4e-4), 2.0)):
        sp, vo = api.merge_scaling(p, vmr_fl=vmr, **kw)
        assert np.all(sp == s_want) and np.array_equal(vo, vmr * s_want if s_want != 1.0 else vmr)
    with pytest.raises(EcckdError) as e:
        api.merge_scaling(p, conc=8e-4, vmr_fl=vmr)           # no reference_surface_mole_fraction in the file
    assert e.value.code == 147
    # requested concentration profile, :117-131: interp in pressure, ends clamped
    pc = np.array([50.0, 500.0, 5000.0, 50000.0, 80000.0])
    cr = np.array([1e-6, 2e-6, 5e-6, 4e-6, 3e-6])
    sp, vo = api.merge_scaling(p, vmr_fl=vmr, pressure_conc=pc, conc_req=cr)
    want = np.interp(pfl, pc, cr)                              # np.interp clamps the ends like :124-125
    assert np.allclose(vo, want, rtol=1e-15) and np.allclose(sp, want / vmr, rtol=1e-15)
    assert vo[0] == cr[0] and vo[-1] == cr[-1]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_merge_spectrum(ctx, dtype):
    from ecckd_amd import api
    rs = np.random.RandomState(2)
    nlay, n = 12, 50021
    ods = [rs.lognormal(-3, 3, (nlay, n)).astype(dtype) for _ in range(3)]
    profs = [np.ones(nlay), np.full(nlay, 0.37), rs.uniform(0.5, 2.0, nlay)]
    merged = None
    for od, sp in zip(ods, profs):
        merged = api.merge_spectrum(ctx, torch.as_tensor(od, device=ctx.device), sp, merged)
    ctx.synchronize()
    want = np.zeros((nlay, n))
    for i, (od, sp) in enumerate(zip(ods, profs)):              # :152-166, product and sum rounded separately
        term = od.astype(np.float64) * sp[:, None]
        want = term if i == 0 else want + term
    assert np.array_equal(merged.cpu().numpy(), want)


def test_erythemal_spectrum_per_g_point(ctx, oracle):
    from ecckd_amd import api
    rs = np.random.RandomState(5)
    n, ng = 60000, 8
    wn = np.linspace(20000.0, 45000.0, n)                      # 500 nm .. 222 nm: covers all three branches
    dwn = np.empty(n)
    dwn[1:-1] = 0.5 * (wn[2:] - wn[:-2]); dwn[0] = 0.5 * dwn[1]; dwn[-1] = 0.5 * dwn[-2]
    g_point = rs.randint(0, ng, n).astype(np.int32)
    g_point[g_point == 6] = 2                                  # an empty g point -> 0/0
    dev = lambda a: torch.as_tensor(a, device=ctx.device)
    gm = api.GPointMap(ctx, dev(g_point), ng, dev(wn), dev(dwn))
    got = gm.erythemal_spectrum()
    # lbl_fluxes.cpp:198-230
    wl = 1.0e7 / wn
    ery = np.zeros(n)
    ery[(wl > 250.0) & (wl <= 298.0)] = 1.0
    m = (wl > 298.0) & (wl <= 328.0); ery[m] = 10.0 ** (0.094 * (298.0 - wl[m]))
    m = (wl > 328.0) & (wl <= 400.0); ery[m] = 10.0 ** (0.015 * (140.0 - wl[m]))
    ery = np.sqrt(ery)
    planck = oracle.planck_function([5777.0], wn, dwn)[0]
    for g in range(ng):
        idx = g_point == g
        if not idx.any():
            assert np.isnan(got[g])
        else:
            assert got[g] == pytest.approx(np.sum(ery[idx] * planck[idx]) / np.sum(planck[idx]), rel=1e-11)
    gm.close()
