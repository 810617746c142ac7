"""The find_g_points job BASELINE configs[1] names - "LW FSCK all well-mixed gases" - on synthetic spectra that are resident in
HBM: one band over 0-3260 cm-1, the gases composite, h2o, o3, co2, ch4, n2o, every gas with a background MERGED
from several spectra as the shipped configurations do (test/find_g_points_lw.sh:176-236 climate, :262-285 nwp): the sum of
scaling x optical depth over the background files, accumulated in DOUBLE (read_merged_spectrum.cpp:135-166), scalings from
`background_conc` / reference_surface_mole_fraction (:135-141) where the script gives concentrations.

The job per gas, as the reference's gas loop (find_g_points.cpp:655-1266) and the reorder_spectrum run in front of it:
    merged background (K: ecckd_merge_spectrum_dev)  ->  sorting key + stable sort of the target spectrum (K1 + K3)
    ->  gas preparation (K4; the first gas's Planck matrix is reused, :529, :970-984)  ->  band search (K5)
then the overlap of the gases' g points and the merged g-point map (:1452-1483).  bench.py, tools/gases_probe.py and the
full-size tests drive it through pipeline.find_g_points_resident, gas after gas or with the gases' searches side by side.
"""
import numpy as np

from . import api, pipeline, synthetic as syn

# spectrum "files" of the job: name -> (seed offset, column scale of the synthetic line spectrum, reference surface mole fraction)
SPECTRA = {
    "composite_present": (0, 30.0, None),
    "h2o_median": (1, 100.0, None),
    "h2o_minimum": (2, 12.0, None),
    "o3_median": (3, 10.0, None),
    "o3_minimum": (4, 3.0, None),
    "co2_present": (5, 50.0, 415e-6),
    "ch4_present": (6, 5.0, 1921e-9),
    "n2o_present": (7, 5.0, 332e-9),
    "o2n2_constant": (8, 1.0, None),
}
# gas -> (target spectrum, [(background spectrum, background_conc or -1)]): the structure of test/find_g_points_lw.sh:176-236 with
# the composite of :280-285; concentrations as there (co2 180 ppmv, ch4 350 ppbv, n2o 190 ppbv: the minima of the range)
GASES = [
    ("composite", "composite_present", [("h2o_minimum", -1.0), ("o3_minimum", -1.0)]),
    ("h2o", "h2o_median", [("composite_present", -1.0), ("o3_minimum", -1.0)]),
    ("o3", "o3_median", [("composite_present", -1.0), ("h2o_minimum", -1.0)]),
    ("co2", "co2_present", [("h2o_minimum", -1.0), ("o3_minimum", -1.0), ("ch4_present", 350e-9), ("n2o_present", 190e-9),
                            ("o2n2_constant", -1.0)]),
    ("ch4", "ch4_present", [("h2o_minimum", -1.0), ("o3_minimum", -1.0), ("co2_present", 180e-6), ("n2o_present", 190e-9),
                            ("o2n2_constant", -1.0)]),
    ("n2o", "n2o_present", [("h2o_minimum", -1.0), ("o3_minimum", -1.0), ("co2_present", 180e-6), ("ch4_present", 350e-9),
                            ("o2n2_constant", -1.0)]),
]


class FsckJob:
    """The spectra of the job on the device (FLOAT, as the CKDMIP files store them) and its steps."""

    def __init__(self, ctx, nwav=7_200_000, nlay=54, ngas=6, nlines=12000, seed=None, spectra="lines"):
        import torch
        self.ctx, self.nwav, self.nlay = ctx, nwav, nlay
        self.gases = GASES[:ngas]
        self.names = [g[0] for g in self.gases]
        self.background_names = {g[0]: [b for b, _ in g[2]] for g in self.gases}
        dev = ctx.device
        seed = syn.SEED_BASE + 1 if seed is None else seed
        self.p = syn.pressure_grid(nlay)
        self.wn_h, self.dwn_h = syn.wavenumber_grid(nwav)
        self.wn = torch.as_tensor(self.wn_h, device=dev)
        self.dwn = torch.as_tensor(self.dwn_h, device=dev)
        self.t_ideal = api.idealised_temperature(self.p)
        self.t_file = syn.temperature_profile(self.p)
        needed = sorted({g[1] for g in self.gases} | {b for g in self.gases for b, _ in g[2]})
        targets = {g[1] for g in self.gases}
        self.od = {}
        for name in needed:
            k, scale, _ = SPECTRA[name]
            if spectra == "legacy":
                self.od[name] = syn.optical_depth(torch, self.p, self.wn, seed + 17 * k, nlines=32, column_scale=scale, device=dev,
                                                  chunk=1 << 20)
            elif name in targets:
                self.od[name] = syn.optical_depth_lines(torch, self.p, self.wn, seed + 17 * k, nlines=nlines, column_scale=scale, device=dev)
            else:                              # background-only spectra: fewer lines, no exactly-zero columns (bench.make_inputs)
                self.od[name] = syn.optical_depth_lines(torch, self.p, self.wn, seed + 1000 + 17 * k, nlines=max(nlines // 3, 1),
                                                        column_scale=scale, zero_fraction=0.0, nclusters=5, device=dev)
        self.begin, self.end = np.array([0]), np.array([nwav - 1])
        self.keys = {}

    def close(self):
        self.od.clear()
        self.keys.clear()

    def merged_background(self, gi):
        """read_merged_spectrum for the gas's background_input list: DOUBLE (nlay, nwav) on the device."""
        merged = None
        for name, conc in self.gases[gi][2]:
            ref = SPECTRA[name][2]
            sp, _ = api.merge_scaling(self.p, conc=conc, reference_surface_vmr=ref if ref is not None else -1.0)
            merged = api.merge_spectrum(self.ctx, self.od[name], sp, merged)
        return merged

    def load_gas(self, gi):
        """What find_g_points has of a gas before its preparation: target, merged background, ordering (the gas's
        reorder_spectrum step, K1 + K3, is part of the job)."""
        ctx = self.ctx
        od = self.od[self.gases[gi][1]]
        bg = self.merged_background(gi)
        key, _ = api.reorder_key_lw(ctx, self.p, self.t_ideal, self.wn, self.dwn, od, 0.5)
        rnk, _ = api.stable_argsort_bands(ctx, key, self.begin, self.end, want_ordered=False)
        return dict(pressure_hl=self.p, temperature_hl=self.t_file, wn=self.wn, dwn=self.dwn, rank=rnk, od=od, bg=bg,
                    sorting_variable=key, band_begin=self.begin, band_end=self.end, min_g_points=np.ones(1, dtype=int),
                    max_g_points=np.full(1, 256))

    def run(self, tolerance=0.0161, tolerance_tolerance=0.01, max_iterations=60, gases_side_by_side=0, merged_map=True):
        """One find_g_points job: every gas loaded, prepared and searched, overlap and merged map -> the result dict of
        pipeline.find_g_points_resident (points = wavenumber points worked through: one pass per gas + what the searches swept)."""
        return pipeline.find_g_points_resident(self.ctx, self.names, self.load_gas, 1, tolerance, None, "transmission", 0.0, 0.0,
                                               tolerance_tolerance, max_iterations, rank=0, world_size=1, merged_map=merged_map,
                                               gases_side_by_side=gases_side_by_side)

    # --- the pieces, for probes that time the search of the SAME prepared gases twice ---
    def prepare(self):
        gases = []
        self._prepared = []
        first = None
        for gi in range(len(self.gases)):
            g = self.load_gas(gi)
            reuse = first.view_ptr("planck_hl")[0] if first is not None else None
            gas, sv, _ = pipeline._prepare_gas(self.ctx, g, "transmission", 0.0, 0.0, reuse, None)
            g.pop("bg")
            first = first or gas
            gases.append(gas)
            self._prepared.append((g, sv))
        return gases

    def search(self, gases, tolerance, tolerance_tolerance, max_iterations, max_concurrent=0):
        req = [dict(ibegin=[0], iend=[self.nwav - 1], heating_rate_tolerance=[tolerance], options=[dict(min_g_points=1, max_g_points=256)])
               for _ in gases]
        return api.find_g_gases(gases, req, tolerance_tolerance, max_iterations, max_concurrent=max_concurrent)
