#!/usr/bin/env python3
"""The shipped path of the headline job: BASELINE configs[1] - "LW FSCK all well-mixed gases" - run by the command-line TOOLS on
CKDMIP-format files, at the scripts' own settings (test/do_all_lw.sh:59-60 fsck tolerance 0.0161 K/d; test/find_g_points_lw.sh:
averaging_method transmission, tolerance_tolerance 0.01, flux_weight 0, max_iterations 60; the gas blocks of :176-236 with the
composite of :280-285: every gas a background_input list of 2-5 files, co2 / ch4 / n2o with background_conc):

    bin/reorder_spectrum  x 6   (one fresh process per gas, as test/reorder_spectrum_lw.sh loops)
    bin/find_g_points           (ONE process, six gases: files read, backgrounds merged, gases prepared, searches side by side)

the same spectra as bench.py's resident job (ecckd_amd/fsck_job.py), written as nine one-column spectrum files.  Reported: wall
time per process (start-up, HIP initialisation, file I/O included), the wavenumber-points/s of the whole (points worked through
by the two stages over the sum of their wall times), find_g_points again with gases_side_by_side=1 (the reference's gas loop) and
whether the two g-points files are identical.

TEST / BENCH INFRASTRUCTURE.  bench.py calls run(); standalone:  python tools/fsck_tools_bench.py [--nwav 7200000]
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def run(ctx, nwav=7_200_000, nlay=54, tolerance=0.0161, workdir=None, keep=False, nlines=12000, compare_gas_after_gas=True, ext="nc"):
    import e2e_bench
    from ecckd_amd import fsck_job, ncio
    top = workdir or tempfile.mkdtemp(prefix="ecckd_fsck_")
    os.makedirs(top, exist_ok=True)
    bindir = os.path.join(ROOT, "bin")
    out = {}
    try:
        job = fsck_job.FsckJob(ctx, nwav, nlay, ngas=6, nlines=nlines)
        need = len(job.od) * nlay * nwav * 4 * 1.1
        free = shutil.disk_usage(top).free
        if free < need:
            raise RuntimeError("fsck tools bench: %s has %.1f GB free, the spectra need %.1f GB" % (top, free / 1e9, need / 1e9))
        t0 = time.perf_counter()
        for name, od in job.od.items():
            vmr = fsck_job.SPECTRA[name][2] or 1.0e-3
            e2e_bench._write_spectrum(os.path.join(top, name + ".nc"), name.split("_")[0], job.p, [job.t_file], job.wn_h, od.cpu().numpy(), vmr)
        names, gases = job.names, job.gases
        job.close()
        # the tools are child processes on the SAME device: hand back what this process has parked (torch's cache, the blocks the
        # library's allocator keeps for the next gas) - a six-gas job wants ~90 GB for itself
        import torch
        torch.cuda.empty_cache()
        ctx.trim_cache()
        os.sync()
        write_s = time.perf_counter() - t0
        procs = []

        def tool(name, *args):
            t1 = time.perf_counter()
            r = subprocess.run([os.path.join(bindir, name), *[str(a) for a in args]], cwd=top, capture_output=True, text=True, timeout=3600,
                               env=dict(os.environ, ECCKD_LOG_TIMES="1"))
            dt = time.perf_counter() - t1
            if r.returncode != 0:
                raise RuntimeError(f"{name} failed ({r.returncode}): {r.stderr[-2000:]}")
            stamps = re.findall(r"^\[\s*([0-9.]+)\]", r.stdout, flags=re.M)
            rec = {"tool": name, "args": " ".join(str(a) for a in args)[:80], "seconds": round(dt, 3),
                   "last_log_stamp": float(stamps[-1]) if stamps else None}
            if name == "find_g_points":       # where its time goes: the tool's own clock in front of its log lines
                rec["log"] = [ln[:110] for ln in r.stdout.splitlines() if ln.startswith("[") and "g point " not in ln and "Band " not in ln][:120]
            procs.append(rec)
            return dt, r

        t_reorder = 0.0
        for g, target, _ in gases:
            t_reorder += tool("reorder_spectrum", f"input={target}.nc", f"output=order_{g}.{ext}", "wavenumber1=0", "wavenumber2=3260")[0]
        with open(os.path.join(top, "find_g.cfg"), "w") as f:
            f.write("iprofile 0\naveraging_method \"transmission\"\ntolerance_tolerance 0.01\nflux_weight 0.0\nmax_iterations 60\n"
                    "heating_rate_tolerance %g\ngases %s\n" % (tolerance, " ".join(names)))
            for g, target, bgs in gases:
                f.write("\\begin %s\n  input %s.nc\n  reordering_input order_%s.%s\n  background_input \"%s\"\n" %
                        (g, target, g, ext, "\n".join(b + ".nc" for b, _ in bgs)))
                if any(c >= 0 for _, c in bgs):
                    f.write("  background_conc %s\n" % " ".join("%g" % c if c >= 0 else "-1" for _, c in bgs))
                f.write("\\end %s\n" % g)
        t_find, r = tool("find_g_points", "find_g.cfg", f"output=gpoints.{ext}")
        gp = ncio.read_g_points(os.path.join(top, "gpoints." + ext))
        comp = [float(v) for v in re.findall(r"computational cost = ([0-9.eE+-]+)", r.stdout)]
        ngs = [int(v) for v in re.findall(r": (\d+) g points, computational cost", r.stdout)]
        out = {"workload": "configs[1] by the tools on files: 6 x bin/reorder_spectrum + ONE bin/find_g_points over %s, backgrounds of "
                           "%s files merged in double, nwav=%d, nlay=%d, fsck band 0-3260 cm-1, tolerance %g K/d, tolerance_tolerance "
                           "0.01, max_iterations 60" % (" ".join(names), "/".join(str(len(b)) for _, _, b in gases), nwav, nlay, tolerance),
               "seconds": {"reorder_spectrum_x6": round(t_reorder, 3), "find_g_points": round(t_find, 3)},
               "output_format": "NetCDF-4, per-wavenumber variables deflated (the scripts' *.h5 names)" if ext == "h5" else "NetCDF classic (*.nc names)",
               "output_bytes": {"order_files": int(sum(os.path.getsize(os.path.join(top, f"order_{g}.{ext}")) for g, _, _ in gases)),
                                "g_points_file": os.path.getsize(os.path.join(top, "gpoints." + ext))},
               "ng_merged": int(np.max(gp["g_point"])) + 1, "ng_per_gas": ngs, "n_pass_reference_counter": sum(comp),
               "input_bytes": int(sum(os.path.getsize(os.path.join(top, n + ".nc")) for n in fsck_job.SPECTRA if os.path.exists(os.path.join(top, n + ".nc")))),
               "setup_seconds_not_timed": {"writing_spectra_and_sync": round(write_s, 1)}}
        if compare_gas_after_gas:
            ref_bytes = open(os.path.join(top, "gpoints." + ext), "rb").read()
            t_seq, _ = tool("find_g_points", "find_g.cfg", f"output=gpoints_seq.{ext}", "gases_side_by_side=1")
            seq = ncio.read_g_points(os.path.join(top, "gpoints_seq." + ext))
            out["seconds"]["find_g_points_gas_after_gas"] = round(t_seq, 3)
            out["side_by_side_speedup_of_the_tool"] = t_seq / t_find
            out["g_point_maps_identical"] = bool(np.array_equal(seq["g_point"], gp["g_point"]))
            del ref_bytes
        out["tool_processes"] = procs
    finally:
        if not keep and workdir is None:
            shutil.rmtree(top, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nwav", type=int, default=7_200_000)
    ap.add_argument("--nlay", type=int, default=54)
    ap.add_argument("--nlines", type=int, default=12000)
    ap.add_argument("--tolerance", type=float, default=0.0161)
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--netcdf4", action="store_true", help="name the ordering and g-points files *.h5 as the scripts do: NetCDF-4, deflated")
    args = ap.parse_args()
    from ecckd_amd import api
    with api.Context(0) as ctx:
        print(json.dumps(run(ctx, args.nwav, args.nlay, args.tolerance, args.workdir, args.keep, args.nlines, ext="h5" if args.netcdf4 else "nc")))


if __name__ == "__main__":
    main()
