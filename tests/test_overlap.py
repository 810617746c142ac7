"""a14: overlap of the gases' g points.  The host part (ecckd_overlap_g_points) runs without a
GPU and is compared with the pure-Python restatement of single_gas_data.cpp:24-124; the two
per-wavenumber maps are GPU kernels compared with numpy `where` passes as the reference does."""
import numpy as np
import pytest


def _case(seed, ngas=3, nband=4):
    rs = np.random.RandomState(seed)
    ngp = rs.randint(1, 6, (ngas, nband))
    svs = []
    for i in range(ngas):
        sv = []
        for b in range(nband):
            sv.extend(np.sort(rs.uniform(-0.5, 12.0, ngp[i, b])).tolist())   # ascending inside a band
        svs.append(np.array(sv))
    return ngp, svs


@pytest.mark.parametrize("seed", range(6))
def test_overlap_matches_restatement(oracle, seed):
    from ecckd_amd import api
    ngp, svs = _case(seed)
    ng, band, gmin, gmax = api.overlap_g_points(ngp, svs)
    ong, oband, ogmin, ogmax = oracle.overlap_g_points(ngp, svs)
    assert ng == ong == int(ngp.sum()) - ngp.shape[1] * (ngp.shape[0] - 1)     # Hogan (2010) Eq. 7
    assert np.array_equal(band, oband) and np.array_equal(gmin, ogmin) and np.array_equal(gmax, ogmax)
    # every single-gas g point is covered, ranges never decrease
    for i in range(ngp.shape[0]):
        assert gmin[i].min() == 0 and gmax[i].max() == ngp[i].sum() - 1
        assert np.all(np.diff(gmax[i]) >= 0)


def test_single_gas_is_identity():
    from ecckd_amd import api
    ng, band, gmin, gmax = api.overlap_g_points([[3, 2]], [np.arange(5.0)])
    assert ng == 5 and band.tolist() == [0, 0, 0, 1, 1]
    assert gmin[0].tolist() == gmax[0].tolist() == [0, 1, 2, 3, 4]


@pytest.mark.gpu
def test_per_wavenumber_maps(ctx, oracle):
    import torch
    from ecckd_amd import api
    rs = np.random.RandomState(3)
    nwav, ngas, nband = 50000, 3, 2
    ngp, svs = _case(11, ngas, nband)
    half = nwav // 2
    gas_gp_dev, gas_gp = [], []
    for i in range(ngas):
        # a rank permutation inside each band and rank ranges splitting each band into ngp[i,b] pieces
        rank = np.concatenate([rs.permutation(half), half + rs.permutation(nwav - half)]).astype(np.int32)
        r1, r2 = [], []
        for b, (lo, hi) in enumerate([(0, half), (half, nwav)]):
            cuts = np.sort(rs.choice(np.arange(lo + 1, hi - 1), ngp[i, b] - 1, replace=False)) if ngp[i, b] > 1 else []
            edges = [lo] + list(cuts) + [hi]
            r1 += edges[:-1]
            r2 += [e - 1 for e in edges[1:]]
        g = api.gas_g_point(ctx, torch.as_tensor(rank, device=ctx.device), r1, r2)
        expect = np.full(nwav, -1, dtype=np.int32)
        for ig, (a, b_) in enumerate(zip(r1, r2)):                      # single_gas_data.h:59-61
            expect[(rank >= a) & (rank <= b_)] = ig
        assert np.array_equal(g.cpu().numpy(), expect)
        gas_gp_dev.append(g)
        gas_gp.append(expect)
    ng, band, gmin, gmax = api.overlap_g_points(ngp, svs)
    gp, nun = api.merge_g_points(ctx, gas_gp_dev, gmin, gmax)
    expect = np.full(nwav, -1, dtype=np.int32)
    for ig in range(ng):                                                # find_g_points.cpp:1463-1470
        found = np.ones(nwav, dtype=bool)
        for i in range(ngas):
            found &= ~((gas_gp[i] < gmin[i, ig]) | (gas_gp[i] > gmax[i, ig]))
        expect[found] = ig
    assert np.array_equal(gp.cpu().numpy(), expect)
    assert nun == int((expect < 0).sum())
